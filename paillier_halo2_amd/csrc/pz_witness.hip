// pz_witness.hip -- K4: expansion of a mul_mod step trace into the advice / lookup cell stream
// (Fr Montgomery, 32 B per cell) that halo2-lib's Context holds after BigUintChip::mul_mod emitted
// its constraints.  Replaces the per-cell CPU pushes behind the dependency call sites
// /root/reference/src/paillier.rs:39-57 and src/bench.rs:44-74.
//
// Layout = paillier_halo2_amd/layout.py::mul_mod_cells (segments: assign q/n/r with range-check
// digit splits | limb convolution a*b | limb convolution q*n | qn + r | carry chain of
// is_equal_muled | r < n); the value semantics are restated in oracle/pyref.py::expand_mul_mod_cells.
//
// Step records are 64-bit words; limbs of any width 16..90 bits are cut out of them into LDS.
// One workgroup (4 waves) per step.  The two convolutions are 76 % of the cells: a wave owns a
// product limb (row), lanes own the terms, partial sums come from a 192-bit wave prefix scan, every
// lane writes a contiguous 96/192-byte run -> fully coalesced streaming stores.  Operand limbs are
// converted to Montgomery form once per step into LDS.  The short serial carry / borrow chains run
// on one lane between two barriers.  Bound: HBM write bandwidth, 32 B per cell.
#include <array>
#include <vector>

#include "fp.cuh"
#include "pz_internal.h"

struct U192 {
    u64 w[3];
};
__device__ __forceinline__ U192 u_make(u64 a, u64 b = 0, u64 c = 0) {
    U192 r;
    r.w[0] = a; r.w[1] = b; r.w[2] = c;
    return r;
}
__device__ __forceinline__ U192 u_add(const U192& a, const U192& b) {
    U192 r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        u64 t = a.w[i] + b.w[i];
        u64 c1 = t < a.w[i];
        u64 t2 = t + c;
        c = c1 | (t2 < t);
        r.w[i] = t2;
    }
    return r;
}
__device__ __forceinline__ U192 u_sub(const U192& a, const U192& b, bool& borrow) {
    U192 r;
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        u64 t = a.w[i] - b.w[i];
        u64 b1 = a.w[i] < b.w[i];
        u64 t2 = t - br;
        br = b1 | (t < br);
        r.w[i] = t2;
    }
    borrow = br != 0;
    return r;
}
__device__ __forceinline__ bool u_eq(const U192& a, const U192& b) {
    return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2];
}
__device__ __forceinline__ U192 u_shr(const U192& a, unsigned s) {  // s < 192
    U192 r = a;
    while (s >= 64) {
        r.w[0] = r.w[1]; r.w[1] = r.w[2]; r.w[2] = 0;
        s -= 64;
    }
    if (s) {
        r.w[0] = (r.w[0] >> s) | (r.w[1] << (64 - s));
        r.w[1] = (r.w[1] >> s) | (r.w[2] << (64 - s));
        r.w[2] >>= s;
    }
    return r;
}
__device__ __forceinline__ U192 u_shl(const U192& a, unsigned s) {  // s < 192
    U192 r = a;
    while (s >= 64) {
        r.w[2] = r.w[1]; r.w[1] = r.w[0]; r.w[0] = 0;
        s -= 64;
    }
    if (s) {
        r.w[2] = (r.w[2] << s) | (r.w[1] >> (64 - s));
        r.w[1] = (r.w[1] << s) | (r.w[0] >> (64 - s));
        r.w[0] <<= s;
    }
    return r;
}
__device__ __forceinline__ U192 u_lowbits(const U192& a, unsigned bits) {  // a mod 2^bits, bits <= 192
    U192 r = a;
    if (bits < 64) { r.w[0] &= (1ull << bits) - 1; r.w[1] = 0; r.w[2] = 0; }
    else if (bits < 128) { if (bits > 64) r.w[1] &= (1ull << (bits - 64)) - 1; else r.w[1] = 0; r.w[2] = 0; }
    else if (bits < 192) { if (bits > 128) r.w[2] &= (1ull << (bits - 128)) - 1; else r.w[2] = 0; }
    return r;
}
struct S192 {
    U192 m;
    bool neg;
};
__device__ __forceinline__ S192 s_sub(const U192& a, const U192& b) {  // a - b
    S192 r;
    bool br;
    r.m = u_sub(a, b, br);
    r.neg = br;
    if (br) {
        bool d;
        r.m = u_sub(b, a, d);
    }
    return r;
}
__device__ __forceinline__ S192 s_addu(const S192& a, const U192& b) {  // a + b, b >= 0
    if (!a.neg) {
        S192 r;
        r.m = u_add(a.m, b);
        r.neg = false;
        return r;
    }
    return s_sub(b, a.m);
}
__device__ __forceinline__ Fr fr_from_u(const U192& v) {
    Fr x;
    x.v[0] = (u32)v.w[0]; x.v[1] = (u32)(v.w[0] >> 32);
    x.v[2] = (u32)v.w[1]; x.v[3] = (u32)(v.w[1] >> 32);
    x.v[4] = (u32)v.w[2]; x.v[5] = (u32)(v.w[2] >> 32);
    x.v[6] = 0; x.v[7] = 0;
    if ((v.w[0] | v.w[1] | v.w[2]) == 0) return x;
    return fp_to_mont(x);
}
__device__ __forceinline__ Fr fr_from_s(const S192& v) {
    Fr x = fr_from_u(v.m);
    return v.neg ? fp_neg(x) : x;
}
__device__ __forceinline__ void mul64w(u64 a, u64 b, u64& hi, u64& lo) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 t = (u64)a0 * b1 + (p00 >> 32);
    const u64 u = (u64)a1 * b0 + (u32)t;
    lo = (u << 32) | (u32)p00;
    hi = (u64)a1 * b1 + (t >> 32) + (u >> 32);
}

// product of two limbs of at most 96 bits (x = x0 + x1 2^64, x1 < 2^32): < 2^192.  `wide` is kernel-uniform
// (limb_bits > 64); the 64-bit-limb configuration pays one 64x64 product as before.
__device__ __forceinline__ U192 u_mul_limb(const U192& x, const U192& y, bool wide) {
    u64 hi, lo;
    mul64w(x.w[0], y.w[0], hi, lo);
    U192 r = u_make(lo, hi, 0);
    if (wide) {
        u64 h1, l1, h2, l2;
        mul64w(x.w[0], y.w[1], h1, l1);
        mul64w(x.w[1], y.w[0], h2, l2);
        r = u_add(r, u_make(0, l1, h1));
        r = u_add(r, u_make(0, l2, h2));
        r = u_add(r, u_make(0, 0, x.w[1] * y.w[1]));
    }
    return r;
}
// limb i (W bits) of the little-endian 64-bit-word integer `words[0..nw)`
__device__ __forceinline__ void limb_extract(const u64* __restrict__ words, unsigned nw, unsigned i, unsigned W, u64 out[2]) {
    const unsigned o = i * W, wi = o >> 6, sh = o & 63;
    const u64 w0 = wi < nw ? words[wi] : 0, w1 = wi + 1 < nw ? words[wi + 1] : 0, w2 = wi + 2 < nw ? words[wi + 2] : 0;
    U192 v = u_lowbits(u_shr(u_make(w0, w1, w2), sh), W);
    out[0] = v.w[0];
    out[1] = v.w[1];
}

// Where a cell of the stream lives: the stream is cut into the circuit's columns of `rows` usable rows, and a column
// may be stored with a larger stride (2^k: room for the blinding rows) -- cell c sits at c + (c / rows) * pad.
// rows == 0: the stream is dense.
// starts != nullptr: halo2-lib's BREAK-POINT layout of the advice stream (assign_with_constraints of the dependency, SURVEY tag [D]):
// column j holds the cells starts[j] .. starts[j + 1] from row 0 on -- the cell a column ends with is ALSO row 0 of the next one
// (a gate must not straddle two columns: the column breaks where a 4-cell gate would, and the shared cell carries the value
// across, tied by a copy constraint).  starts[0] = 0, starts[n_cols] = the stream's length; a column takes at most rows + 1 cells
// (rows = max_rows - 1 new ones), so c / rows never overshoots the column and the search walks forward from there.  pad = the
// column stride in this mode.
struct CellPtr {
    Fr* base;
    size_t idx, rows, pad;
    const u64* starts;
    unsigned ncols;
    __device__ __forceinline__ CellPtr operator+(size_t o) const {
        CellPtr r = *this;
        r.idx += o;
        return r;
    }
    __device__ __forceinline__ explicit operator bool() const { return base != nullptr; }
    __device__ __forceinline__ size_t plain_col() const {
        // one division per cell: 32-bit whenever the stream has fewer than 2^32 cells (a 64-bit division is ~100 instructions)
        return ((idx | rows) >> 32) == 0 ? (size_t)((u32)idx / (u32)rows) : idx / rows;
    }
    __device__ __forceinline__ Fr* addr() const {   // (dense / plain cut only)
        if (!rows) return base + idx;
        return base + idx + plain_col() * pad;
    }
};
__device__ __forceinline__ void fp_store(const CellPtr& p, const Fr& v) {
    if (!p.starts) {
        fp_store(p.addr(), v);
        return;
    }
    size_t col = p.plain_col();
    if (col >= p.ncols) col = p.ncols - 1;
    while (col + 1 < p.ncols && p.starts[col + 1] <= p.idx) ++col;
    const size_t row = p.idx - p.starts[col];
    if (row < p.pad) fp_store(p.base + col * p.pad + row, v);            // (row < stride always for a table built by pz_circuit_break_points)
    if (row == 0 && col > 0) {                                         // the previous column's last cell is this one too
        const size_t prow = p.idx - p.starts[col - 1];
        if (prow < p.pad) fp_store(p.base + (col - 1) * p.pad + prow, v);
    }
}

struct ExpP {
    unsigned L, D, lb;
    unsigned W, L64;                          // circuit limb width; 64-bit words per big integer of a step record
    unsigned k64, rem64, rc64_adv, rc64_lk;   // range_check(limb, W)   (names from the 64-bit-limb first version)
    unsigned cb, kcb, remcb, rccb_adv, rccb_lk;  // range_check(carry, cb)
    size_t off_assign, off_ab, off_qn, off_add, off_eq, off_lt, cells;
    size_t lk_assign, lk_eq, lk_lt, lookups;
    u64 max_w[3];  // L*(2^W-1)^2 + (2^W-1)
    size_t cell0, lk0;        // stream index of this launch's first advice / lookup cell
    size_t rows, pad;         // column cut of the stream (CellPtr); 0, 0 = dense
    const u64* bp_starts;     // != nullptr: the advice stream in break-point layout (CellPtr::starts); the lookup stream keeps the plain cut
    size_t bp_div, bp_stride; // max_rows - 1, column stride
    unsigned bp_ncols;
    size_t pair_stride, odd_off;   // != 0: steps come in pairs (uniform-shape circuit): step s sits at
                                   // cell0 + (s / 2) * pair_stride + (s & 1) * odd_off instead of cell0 + s * cells
};

__device__ __forceinline__ CellPtr adv_ptr(const ExpP& P, Fr* advice, size_t idx) {
    if (P.bp_starts) return CellPtr{advice, idx, P.bp_div, P.bp_stride, P.bp_starts, P.bp_ncols};
    return CellPtr{advice, idx, P.rows, P.pad, nullptr, 0};
}

// A cell's value as an integer: +-m, or the field inverse of that.  The helpers below pick the VALUE of a cell (cheap integer
// selects, even when the lanes of a wave sit at different positions of a pattern) and the caller converts it to Montgomery
// form ONCE (cv_to_fr): with the conversion inside every switch arm a wave paid one Montgomery product per arm.
struct CV {
    U192 m;
    bool neg, inv;
};
__device__ __forceinline__ CV cv_u(const U192& m) { return CV{m, false, false}; }
__device__ __forceinline__ CV cv_s(const S192& v) { return CV{v.m, v.neg, false}; }
__device__ __forceinline__ CV cv_inv(const S192& v) { return CV{v.m, v.neg, true}; }
__device__ __forceinline__ CV cv_one() { return CV{u_make(1), false, false}; }
__device__ __forceinline__ CV cv_zero() { return CV{u_make(0), false, false}; }
__device__ __forceinline__ Fr cv_to_fr(const CV& c) {
    Fr x = fr_from_u(c.m);
    if (c.neg) x = fp_neg(x);
    if (c.inv) x = fp_inv(x);   // is_zero's inverse cell of a non-zero difference: rare
    return x;
}


// position p of RangeChip::range_check(x, bits): advice pattern
__device__ __forceinline__ CV rc_adv_cell(const U192& x, unsigned bits, unsigned lb, unsigned p) {
    const unsigned k = (bits + lb - 1) / lb, rem = bits % lb;
    const unsigned body = k > 1 ? 1 + 3 * (k - 1) : 0;
    if (p < body) {
        if (p == 0) return cv_u(u_lowbits(x, lb));
        const unsigned g = (p - 1) / 3 + 1, w = (p - 1) % 3;
        if (w == 0) return cv_u(u_lowbits(u_shr(x, lb * g), lb));
        if (w == 1) return cv_u(u_shl(u_make(1), lb * g));
        return cv_u(u_lowbits(x, lb * (g + 1)));
    }
    const unsigned t = p - body;  // tail gate
    const U192 last = u_lowbits(u_shr(x, lb * (k - 1)), lb);
    if (t == 0) return cv_zero();
    if (rem == 1) return cv_u(last);
    if (t == 1) return cv_u(last);
    if (t == 2) return cv_u(u_make(1ull << (lb - rem)));
    return cv_u(u_shl(last, lb - rem));
}
__device__ __forceinline__ CV rc_lk_cell(const U192& x, unsigned bits, unsigned lb, unsigned p) {
    const unsigned k = (bits + lb - 1) / lb, rem = bits % lb;
    if (p < k) return cv_u(u_lowbits(u_shr(x, lb * p), lb));
    return cv_u(u_shl(u_lowbits(u_shr(x, lb * (k - 1)), lb), lb - rem));
}

// 8-cell is_zero pattern on the (signed) difference d, then positions 0..11 of is_equal(x, y)
__device__ __forceinline__ CV is_equal_cell(const U192& x, const U192& y, unsigned p) {
    S192 d = s_sub(x, y);
    const bool z = !d.neg && (d.m.w[0] | d.m.w[1] | d.m.w[2]) == 0;
    switch (p) {
        case 0: return cv_s(d);
        case 1: return cv_u(y);
        case 2: return cv_one();
        case 3: return cv_u(x);
        case 4: return z ? cv_one() : cv_zero();
        case 5: return cv_s(d);
        case 6: return z ? cv_one() : cv_inv(d);
        case 7: return cv_one();
        case 8: return cv_zero();
        case 9: return cv_s(d);
        case 10: return z ? cv_one() : cv_zero();
        default: return cv_zero();
    }
}
// 22 cells of div_mod_unsafe(v, 2^W)
__device__ __forceinline__ CV div_mod_cell(const U192& v, unsigned p, unsigned W) {
    const U192 qd = u_shr(v, W);
    const U192 rd = u_lowbits(v, W);
    const U192 prod = u_shl(qd, W);
    switch (p) {
        case 0: return cv_u(qd);
        case 1: return cv_u(rd);
        case 2: return cv_zero();
        case 3: return cv_u(qd);
        case 4: return cv_u(u_shl(u_make(1), W));
        case 5: return cv_u(prod);
        case 6: return cv_u(rd);  // v - prod
        case 7: return cv_u(prod);
        case 8: return cv_one();
        case 9: return cv_u(v);
        default: return is_equal_cell(rd, rd, p - 10);
    }
}

#define EXP_THREADS 256
#define EXP_MAXL 128

__global__ __launch_bounds__(EXP_THREADS) void k_witness_expand(ExpP P, const u64* __restrict__ steps,
                                                                const u64* __restrict__ modulus,
                                                                Fr* __restrict__ advice, Fr* __restrict__ lookup) {
    __shared__ u64 s_a[EXP_MAXL][2], s_b[EXP_MAXL][2], s_q[EXP_MAXL][2], s_r[EXP_MAXL][2], s_n[EXP_MAXL][2];
    __shared__ Fr s_am[EXP_MAXL], s_bm[EXP_MAXL], s_qm[EXP_MAXL], s_nm[EXP_MAXL];
    __shared__ u64 s_pab[2 * EXP_MAXL][3], s_pqn[2 * EXP_MAXL][3];
    __shared__ u64 s_carry[2 * EXP_MAXL + 1][2], s_accx[2 * EXP_MAXL + 1][2];
    __shared__ unsigned char s_eqbit[2 * EXP_MAXL + 2], s_borrow[EXP_MAXL + 1];
    const unsigned L = P.L, D = P.D, lb = P.lb, W = P.W;
    const bool wide = W > 64;
    const size_t step = blockIdx.x;
    const u64* st = steps + step * 4 * (size_t)P.L64;
    const CellPtr adv = adv_ptr(P, advice, P.cell0 + (P.pair_stride ? (step >> 1) * P.pair_stride + (step & 1) * P.odd_off : step * P.cells));
    const CellPtr lk{lookup, P.lk0 + step * P.lookups, P.rows, P.pad, nullptr, 0};
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const U192 MAXV = u_make(P.max_w[0], P.max_w[1], P.max_w[2]);
    const U192 BASE = u_shl(u_make(1), W);
#define LIMB(X, i) u_make((X)[i][0], (X)[i][1])

    for (unsigned i = tid; i < L; i += EXP_THREADS) {
        limb_extract(st, P.L64, i, W, s_a[i]);
        limb_extract(st + P.L64, P.L64, i, W, s_b[i]);
        limb_extract(st + 2 * (size_t)P.L64, P.L64, i, W, s_q[i]);
        limb_extract(st + 3 * (size_t)P.L64, P.L64, i, W, s_r[i]);
        limb_extract(modulus, P.L64, i, W, s_n[i]);
        s_am[i] = fr_from_u(LIMB(s_a, i));
        s_bm[i] = fr_from_u(LIMB(s_b, i));
        s_qm[i] = fr_from_u(LIMB(s_q, i));
        s_nm[i] = fr_from_u(LIMB(s_n, i));
    }
    __syncthreads();

    // ---- the two limb convolutions: wave per product limb (row), lanes own terms
    const unsigned PER = (D + 63) / 64;
    for (int which = 0; which < 2; ++which) {
        const u64(*xs)[2] = which ? s_q : s_a;
        const u64(*ys)[2] = which ? s_n : s_b;
        const Fr* xm = which ? s_qm : s_am;
        const Fr* ym = which ? s_nm : s_bm;
        u64(*pout)[3] = which ? s_pqn : s_pab;
        const CellPtr seg = adv + (which ? P.off_qn : P.off_ab);
        if (seg && tid == 0) fp_store(seg, fp_zero<FrTag>());  // load_zero
        for (unsigned i = wave; i < D; i += EXP_THREADS / 64) {
            const size_t row = 1 + (size_t)i + 3 * ((size_t)i * (i + 1) / 2);
            // local inclusive prefix over this lane's PER terms
            U192 pre[4];      // PER <= 4 (D <= 2 * EXP_MAXL - 1): unrolled so that pre[] stays in registers
            U192 run = u_make(0);
#pragma unroll
            for (unsigned t = 0; t < 4; ++t) {
                if (t < PER) {
                    const unsigned j = lane * PER + t;
                    U192 p = u_make(0);
                    if (j <= i && j < L && (i - j) < L) p = u_mul_limb(LIMB(xs, j), LIMB(ys, i - j), wide);
                    run = u_add(run, p);
                }
                pre[t] = run;
            }
            // wave inclusive scan of lane totals
            U192 tot = run;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                U192 o;
                o.w[0] = __shfl_up(tot.w[0], off, 64);
                o.w[1] = __shfl_up(tot.w[1], off, 64);
                o.w[2] = __shfl_up(tot.w[2], off, 64);
                if (lane >= (unsigned)off) tot = u_add(tot, o);
            }
            bool dummy;
            const U192 excl = u_sub(tot, run, dummy);
            if (seg && lane == 0) fp_store(seg + row, fp_zero<FrTag>());
#pragma unroll
            for (unsigned t = 0; t < 4; ++t) {
                const unsigned j = lane * PER + t;
                if (t < PER && j <= i) {
                    const U192 s = u_add(excl, pre[t]);
                    if (seg) {
                        const CellPtr c = seg + row + 1 + 3 * (size_t)j;
                        fp_store(c, j < L ? xm[j] : fp_zero<FrTag>());
                        fp_store(c + 1, (i - j) < L ? ym[i - j] : fp_zero<FrTag>());
                        fp_store(c + 2, fr_from_u(s));
                    }
                    if (j == i) {
                        pout[i][0] = s.w[0]; pout[i][1] = s.w[1]; pout[i][2] = s.w[2];
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- serial chains (one lane): carries of is_equal_muled, borrows of r < n
    if (tid == 0) {
        U192 carry = u_make(0), accx = u_make(0);
        unsigned eq = 1;
        s_eqbit[0] = 1;
        for (unsigned i = 0; i < D; ++i) {
            s_carry[i][0] = carry.w[0]; s_carry[i][1] = carry.w[1];
            s_accx[i][0] = accx.w[0]; s_accx[i][1] = accx.w[1];
            const U192 A = u_make(s_pab[i][0], s_pab[i][1], s_pab[i][2]);
            U192 Bq = u_make(s_pqn[i][0], s_pqn[i][1], s_pqn[i][2]);
            if (i < L) Bq = u_add(Bq, LIMB(s_r, i));
            S192 s = s_addu(s_addu(s_sub(A, Bq), carry), MAXV);  // >= 0 for a consistent step
            const U192 t = u_add(accx, MAXV);
            const unsigned e = (!s.neg && u_eq(u_lowbits(s.m, W), u_lowbits(t, W))) ? 1u : 0u;
            eq &= e;
            s_eqbit[i + 1] = (unsigned char)eq;
            carry = u_shr(s.m, W);
            accx = u_shr(t, W);
        }
        s_carry[D][0] = carry.w[0]; s_carry[D][1] = carry.w[1];
        s_accx[D][0] = accx.w[0]; s_accx[D][1] = accx.w[1];
        s_eqbit[D + 1] = (unsigned char)(eq & (u_eq(carry, accx) ? 1u : 0u));
        unsigned borrow = 0;
        for (unsigned i = 0; i < L; ++i) {
            s_borrow[i] = (unsigned char)borrow;
            // r_i < n_i + borrow  (n_i + borrow may be 2^W)
            bool lt;
            (void)u_sub(LIMB(s_r, i), u_add(LIMB(s_n, i), u_make(borrow)), lt);
            borrow = lt ? 1u : 0u;
        }
        s_borrow[L] = (unsigned char)borrow;
    }
    __syncthreads();

    // ---- segment: assign_integer(q), (n), (r)
    {
        const unsigned per_int = L + L * P.rc64_adv;
        for (unsigned t = tid; t < 3 * per_int; t += EXP_THREADS) {
            const unsigned which = t / per_int, u = t % per_int;
            const u64(*X)[2] = which == 0 ? s_q : (which == 1 ? s_n : s_r);
            CV v;
            if (u < L) v = cv_u(LIMB(X, u));
            else {
                const unsigned i = (u - L) / P.rc64_adv, p = (u - L) % P.rc64_adv;
                v = rc_adv_cell(LIMB(X, i), W, lb, p);
            }
            if (adv) fp_store(adv + P.off_assign + t, cv_to_fr(v));
        }
        if (lk)
            for (unsigned t = tid; t < 3 * L * P.rc64_lk; t += EXP_THREADS) {
                const unsigned which = t / (L * P.rc64_lk), u = t % (L * P.rc64_lk);
                const u64(*X)[2] = which == 0 ? s_q : (which == 1 ? s_n : s_r);
                fp_store(lk + P.lk_assign + t, cv_to_fr(rc_lk_cell(LIMB(X, u / P.rc64_lk), W, lb, u % P.rc64_lk)));
            }
    }
    // ---- segment: qn + r
    if (adv)
        for (unsigned t = tid; t < 4 * L; t += EXP_THREADS) {
            const unsigned i = t / 4, p = t % 4;
            const U192 qn = u_make(s_pqn[i][0], s_pqn[i][1], s_pqn[i][2]);
            CV v;
            if (p == 0) v = cv_u(qn);
            else if (p == 1) v = cv_one();
            else if (p == 2) v = cv_u(LIMB(s_r, i));
            else v = cv_u(u_add(qn, LIMB(s_r, i)));
            fp_store(adv + P.off_add + t, cv_to_fr(v));
        }
    // ---- segment: is_equal_muled
    {
        const unsigned per = 75 + P.rccb_adv;  // limbs 0..D-2; the last limb has 75 + 16
        if (adv && tid == 0) {
            fp_store(adv + P.off_eq, fp_zero<FrTag>());
            fp_store(adv + P.off_eq + 1, fp_one<FrTag>());
        }
        const unsigned total = (D - 1) * per + 75 + 16;
        for (unsigned t = tid; t < total; t += EXP_THREADS) {
            unsigned i, p;
            if (t < (D - 1) * per) { i = t / per; p = t % per; }
            else { i = D - 1; p = t - (D - 1) * per; }
            const U192 A = u_make(s_pab[i][0], s_pab[i][1], s_pab[i][2]);
            U192 Bq = u_make(s_pqn[i][0], s_pqn[i][1], s_pqn[i][2]);
            if (i < L) Bq = u_add(Bq, LIMB(s_r, i));
            const U192 carry = u_make(s_carry[i][0], s_carry[i][1]);
            const U192 accx = u_make(s_accx[i][0], s_accx[i][1]);
            const S192 diff = s_sub(A, Bq);
            const S192 dc = s_addu(diff, carry);
            const S192 ssum = s_addu(dc, MAXV);
            const U192 tt = u_add(accx, MAXV);
            const U192 ncarry = u_make(s_carry[i + 1][0], s_carry[i + 1][1]);
            CV v;
            if (p < 4) {
                v = p == 0 ? cv_s(diff) : p == 1 ? cv_u(Bq) : p == 2 ? cv_one() : cv_u(A);
            } else if (p < 11) {
                switch (p - 4) {
                    case 0: v = cv_s(diff); break;
                    case 1: v = cv_u(carry); break;
                    case 2: v = cv_one(); break;
                    case 3: v = cv_s(dc); break;
                    case 4: v = cv_u(MAXV); break;
                    case 5: v = cv_one(); break;
                    default: v = cv_s(ssum); break;
                }
            } else if (p < 33) {
                v = div_mod_cell(ssum.m, p - 11, W);
            } else if (p < 37) {
                v = p == 33 ? cv_u(accx) : p == 34 ? cv_one() : p == 35 ? cv_u(MAXV) : cv_u(tt);
            } else if (p < 59) {
                v = div_mod_cell(tt, p - 37, W);
            } else if (p < 71) {
                v = is_equal_cell(u_lowbits(ssum.m, W), u_lowbits(tt, W), p - 59);
            } else if (p < 75) {
                const unsigned e = u_eq(u_lowbits(ssum.m, W), u_lowbits(tt, W)) ? 1u : 0u;
                const unsigned in = s_eqbit[i], out = s_eqbit[i + 1];
                v = p == 71 ? cv_zero() : p == 72 ? (in ? cv_one() : cv_zero())
                    : p == 73 ? (e ? cv_one() : cv_zero()) : (out ? cv_one() : cv_zero());
            } else if (i < D - 1) {
                v = rc_adv_cell(ncarry, P.cb, lb, p - 75);
            } else {
                const U192 qacc = u_make(s_accx[D][0], s_accx[D][1]);
                if (p < 75 + 12) v = is_equal_cell(ncarry, qacc, p - 75);
                else {
                    const unsigned e = u_eq(ncarry, qacc) ? 1u : 0u;
                    const unsigned in = s_eqbit[D], out = s_eqbit[D + 1];
                    const unsigned pp = p - 87;
                    v = pp == 0 ? cv_zero() : pp == 1 ? (in ? cv_one() : cv_zero())
                        : pp == 2 ? (e ? cv_one() : cv_zero()) : (out ? cv_one() : cv_zero());
                }
            }
            if (adv) fp_store(adv + P.off_eq + 2 + t, cv_to_fr(v));
        }
        if (lk)
            for (unsigned t = tid; t < (D - 1) * P.rccb_lk; t += EXP_THREADS) {
                const unsigned i = t / P.rccb_lk, p = t % P.rccb_lk;
                fp_store(lk + P.lk_eq + t, cv_to_fr(rc_lk_cell(u_make(s_carry[i + 1][0], s_carry[i + 1][1]), P.cb, lb, p)));
            }
    }
    // ---- segment: r < n
    {
        const unsigned per = 11 + P.rc64_adv;
        for (unsigned t = tid; t < L * per; t += EXP_THREADS) {
            const unsigned i = t / per, p = t % per;
            const unsigned borrow = s_borrow[i], lt = s_borrow[i + 1];
            const U192 nb = u_add(LIMB(s_n, i), u_make(borrow));
            bool br;
            const U192 shift = u_sub(u_add(LIMB(s_r, i), BASE), nb, br);  // r_i - nb + 2^W
            const U192 outv = u_lowbits(shift, W);
            CV v;
            switch (p) {
                case 0: v = cv_u(LIMB(s_n, i)); break;
                case 1: v = cv_one(); break;
                case 2: v = borrow ? cv_one() : cv_zero(); break;
                case 3: v = cv_u(nb); break;
                case 4: v = cv_u(shift); break;
                case 5: v = lt ? cv_one() : cv_zero(); break;
                case 6: v = cv_u(outv); break;
                case 7: v = cv_u(LIMB(s_r, i)); break;
                case 8: v = lt ? cv_one() : cv_zero(); break;
                case 9: v = cv_u(BASE); break;
                case 10: v = cv_u(lt ? u_add(LIMB(s_r, i), BASE) : LIMB(s_r, i)); break;
                default: v = rc_adv_cell(outv, W, lb, p - 11); break;
            }
            if (adv) fp_store(adv + P.off_lt + t, cv_to_fr(v));
        }
        if (adv && tid == 0) fp_store(adv + P.off_lt + (size_t)L * per, s_borrow[L] ? fp_one<FrTag>() : fp_zero<FrTag>());
        if (lk)
            for (unsigned t = tid; t < L * P.rc64_lk; t += EXP_THREADS) {
                const unsigned i = t / P.rc64_lk, p = t % P.rc64_lk;
                bool br;
                const U192 shift = u_sub(u_add(LIMB(s_r, i), BASE), u_add(LIMB(s_n, i), u_make(s_borrow[i])), br);
                fp_store(lk + P.lk_lt + t, cv_to_fr(rc_lk_cell(u_lowbits(shift, W), W, lb, p)));
            }
    }
}

#undef LIMB

// ------------------------------------------------------------------------------------------------
// host: layout arithmetic (mirrors paillier_halo2_amd/layout.py::mul_mod_cells)
// ------------------------------------------------------------------------------------------------
static void rc_counts(unsigned bits, unsigned lb, unsigned& k, unsigned& rem, unsigned& adv, unsigned& lk) {
    k = (bits + lb - 1) / lb;
    rem = bits % lb;
    adv = k == 1 ? 0 : 1 + 3 * (k - 1);
    lk = k;
    if (rem == 1) adv += 4;
    else if (rem > 1) { adv += 4; lk += 1; }
}

// 192-bit host helpers for MAX = L*(2^W-1)^2 + (2^W-1) = (L << 2W) - (L << (W+1)) + L + (1 << W) - 1
static void h_shl(const uint64_t a[3], unsigned s, uint64_t r[3]) {
    uint64_t t[3] = {a[0], a[1], a[2]};
    while (s >= 64) { t[2] = t[1]; t[1] = t[0]; t[0] = 0; s -= 64; }
    if (s) { t[2] = (t[2] << s) | (t[1] >> (64 - s)); t[1] = (t[1] << s) | (t[0] >> (64 - s)); t[0] <<= s; }
    r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}
static void h_add(uint64_t a[3], const uint64_t b[3]) {
    unsigned __int128 c = 0;
    for (int i = 0; i < 3; ++i) { c += (unsigned __int128)a[i] + b[i]; a[i] = (uint64_t)c; c >>= 64; }
}
static void h_sub(uint64_t a[3], const uint64_t b[3]) {
    uint64_t br = 0;
    for (int i = 0; i < 3; ++i) {
        const uint64_t t = a[i] - b[i], b1 = a[i] < b[i], t2 = t - br;
        br = b1 | (t < br);
        a[i] = t2;
    }
}
static unsigned h_bits(const uint64_t a[3]) {
    for (int i = 2; i >= 0; --i)
        if (a[i]) return 64 * i + (64 - __builtin_clzll(a[i]));
    return 0;
}

static int make_params(uint32_t L, uint32_t limb_bits, uint32_t lb, ExpP& P) {
    // limb_bits = 64 is the reference bench's choice (bench.rs:140, paillier.rs:116); its add test runs 88-bit limbs
    // (paillier.rs:186-187).  A limb travels as two 64-bit words and a product limb as three: 2W + log2(L) + 3 <= 192.
    if (limb_bits < 16 || limb_bits > 90) return PZ_ERR_UNSUPPORTED;
    if (L < 2 || L > EXP_MAXL) return PZ_ERR_UNSUPPORTED;
    if (lb < 4 || lb > 32) return PZ_ERR_INVALID;
    {
        unsigned lg = 0;
        while ((1u << lg) < L) ++lg;
        if (2 * limb_bits + lg + 3 > 192) return PZ_ERR_UNSUPPORTED;
    }
    memset(&P, 0, sizeof P);
    P.L = L;
    P.D = 2 * L - 1;
    P.lb = lb;
    P.W = limb_bits;
    P.L64 = (L * limb_bits + 63) / 64;
    rc_counts(limb_bits, lb, P.k64, P.rem64, P.rc64_adv, P.rc64_lk);
    if (P.k64 < 2) return PZ_ERR_INVALID;
    const uint64_t Lw[3] = {L, 0, 0}, one[3] = {1, 0, 0};
    uint64_t mx[3], t[3];
    h_shl(Lw, 2 * limb_bits, mx);
    h_shl(Lw, limb_bits + 1, t);
    h_sub(mx, t);
    h_add(mx, Lw);
    h_shl(one, limb_bits, t);
    h_add(mx, t);
    h_sub(mx, one);
    P.max_w[0] = mx[0]; P.max_w[1] = mx[1]; P.max_w[2] = mx[2];
    h_shl(mx, 1, t);  // 2*MAX
    const unsigned bits = h_bits(t);
    P.cb = bits - limb_bits;
    rc_counts(P.cb, lb, P.kcb, P.remcb, P.rccb_adv, P.rccb_lk);
    size_t off = 0;
    P.off_assign = off;
    off += 3 * ((size_t)L + (size_t)L * P.rc64_adv);
    size_t per_mul = 1;
    for (unsigned i = 0; i < P.D; ++i) per_mul += 1 + 3 * ((size_t)i + 1);
    P.off_ab = off; off += per_mul;
    P.off_qn = off; off += per_mul;
    P.off_add = off; off += 4 * (size_t)L;
    P.off_eq = off; off += 2 + (size_t)P.D * 75 + (size_t)(P.D - 1) * P.rccb_adv + 16;
    P.off_lt = off; off += (size_t)L * (11 + P.rc64_adv) + 1;
    P.cells = off;
    P.lk_assign = 0;
    P.lk_eq = 3 * (size_t)L * P.rc64_lk;
    P.lk_lt = P.lk_eq + (size_t)(P.D - 1) * P.rccb_lk;
    P.lookups = P.lk_lt + (size_t)L * P.rc64_lk;
    return PZ_OK;
}

extern "C" int pz_witness_cells_per_step(uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits,
                                         size_t* advice_cells, size_t* lookup_cells) {
    ExpP P;
    PZCHK(make_params(limbs, limb_bits, lookup_bits, P));
    if (advice_cells) *advice_cells = P.cells;
    if (lookup_cells) *lookup_cells = P.lookups;
    return PZ_OK;
}

extern "C" int pz_witness_expand_dev(pz_ctx* ctx, uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits,
                                     const uint64_t* d_steps, size_t n_steps, const uint64_t* d_modulus,
                                     uint64_t* d_advice, uint64_t* d_lookup) {
    if (!ctx || (n_steps && (!d_steps || !d_modulus))) return PZ_ERR_INVALID;
    ExpP P;
    PZCHK(make_params(limbs, limb_bits, lookup_bits, P));
    if (!n_steps || (!d_advice && !d_lookup)) return PZ_OK;
    PZ_ENTER(ctx);
    pz_timer tm(ctx, PZ_T_EXPAND);
    hipLaunchKernelGGL(k_witness_expand, dim3((unsigned)n_steps), dim3(EXP_THREADS), 0, ctx->stream, P, d_steps,
                       d_modulus, (Fr*)d_advice, (Fr*)d_lookup);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

// ------------------------------------------------------------------------------------------------
// The rest of the circuit: every cell the drivers bench.rs:33-75 / :77-117 push that is
// NOT inside a mul_mod -- the input assignments, square + refresh of n, load_zero, the constant cells of
// pow_mod_fixed_exp, the assignment of res and assert_equal_fresh -- in the layout of
// paillier_halo2_amd/layout.py::circuit_cells (values: oracle/pyref.py::expand_circuit_cells).  A few thousand to a few
// hundred thousand cells per proof: one workgroup, the serial carry chain of refresh on one lane.
// ------------------------------------------------------------------------------------------------
#define CIRC_MAXF 256   // fresh limbs of n^2 (2 Ln)
#define CIRC_MAXJ 3     // pieces a product limb is cut into (2W + log2(Ln) bits -> at most 3 limbs)

struct CircP {
    ExpP e;                         // shape of the mul_mod steps (L = 2 Ln): range-check parameters, limb width
    unsigned Ln, nf, words_n, kind;
    unsigned rc_adv, rc_lk;         // range_check(limb, W)
    size_t a_assign[4], a_square, a_refresh, a_zero, a_pow[2], a_res, a_eq;   // advice offsets of the segments
    size_t l_assign[4], l_refresh, l_res;                                      // lookup offsets
    unsigned char inc[CIRC_MAXF];
};

__global__ __launch_bounds__(256) void k_circuit_misc(CircP C, const u64* __restrict__ inputs /* n | g | x | y | res */,
                                                      const u64* __restrict__ cval /* the circuit's result, L64 words */,
                                                      Fr* __restrict__ advice, Fr* __restrict__ lookup) {
    const CellPtr adv = adv_ptr(C.e, advice, 0), lk{lookup, 0, C.e.rows, C.e.pad, nullptr, 0};
    __shared__ u64 s_in[4][EXP_MAXL / 2 + 1][2];       // limbs of n, g, x, y
    __shared__ u64 s_res[CIRC_MAXF][2], s_c[CIRC_MAXF][2], s_fresh[CIRC_MAXF][2];
    __shared__ u64 s_sq[CIRC_MAXF][3];                 // limbs of n * n (unreduced convolution sums)
    __shared__ u64 s_dm[CIRC_MAXF][CIRC_MAXJ][3];      // refresh: the value each div_mod_unsafe divides
    __shared__ u64 s_ad[CIRC_MAXF][CIRC_MAXJ][3];      // refresh: limb i + j before the j-th remainder is added
    __shared__ unsigned s_roff[CIRC_MAXF + 1];
    __shared__ unsigned char s_eq[CIRC_MAXF + 1];
    const unsigned Ln = C.Ln, nf = C.nf, W = C.e.W, lb = C.e.lb, tid = threadIdx.x;
    const bool wide = W > 64;
    const unsigned D = 2 * Ln - 1;
#define LIMB(X, i) u_make((X)[i][0], (X)[i][1])
    for (unsigned t = tid; t < 4 * Ln; t += blockDim.x) limb_extract(inputs + (size_t)(t / Ln) * C.words_n, C.words_n, t % Ln, W, s_in[t / Ln][t % Ln]);
    for (unsigned t = tid; t < nf; t += blockDim.x) {
        limb_extract(inputs + 4 * (size_t)C.words_n, C.e.L64, t, W, s_res[t]);
        limb_extract(cval, C.e.L64, t, W, s_c[t]);
    }
    __syncthreads();
    // ---- square(n): one lane per product limb, cells written as the row is summed
    if (tid == 0) fp_store(adv + C.a_square, fp_zero<FrTag>());
    for (unsigned i = tid; i < D; i += blockDim.x) {
        const size_t row = 1 + (size_t)i + 3 * ((size_t)i * (i + 1) / 2);
        const CellPtr c = adv + C.a_square + row;
        fp_store(c, fp_zero<FrTag>());
        U192 sum = u_make(0);
        for (unsigned j = 0; j <= i; ++j) {
            const bool in = j < Ln && (i - j) < Ln;
            if (in) sum = u_add(sum, u_mul_limb(LIMB(s_in[0], j), LIMB(s_in[0], i - j), wide));
            fp_store(c + 1 + 3 * (size_t)j, j < Ln ? fr_from_u(LIMB(s_in[0], j)) : fp_zero<FrTag>());
            fp_store(c + 2 + 3 * (size_t)j, (i - j) < Ln ? fr_from_u(LIMB(s_in[0], i - j)) : fp_zero<FrTag>());
            fp_store(c + 3 + 3 * (size_t)j, fr_from_u(sum));
        }
        s_sq[i][0] = sum.w[0]; s_sq[i][1] = sum.w[1]; s_sq[i][2] = sum.w[2];
    }
    for (unsigned i = D + tid; i < nf; i += blockDim.x) { s_sq[i][0] = 0; s_sq[i][1] = 0; s_sq[i][2] = 0; }
    __syncthreads();
    // ---- serial chains: refresh's carries, assert_equal_fresh's running bit
    if (tid == 0) {
        unsigned off = 1;
        for (unsigned i = 0; i < nf; ++i) {
            s_roff[i] = off;
            U192 limb = u_make(s_sq[i][0], s_sq[i][1], s_sq[i][2]);
            for (unsigned j = 0; j <= C.inc[i]; ++j) {
                s_dm[i][j][0] = limb.w[0]; s_dm[i][j][1] = limb.w[1]; s_dm[i][j][2] = limb.w[2];
                const U192 nrem = u_lowbits(limb, W);
                if (j == 0) {
                    s_sq[i][0] = nrem.w[0]; s_sq[i][1] = nrem.w[1]; s_sq[i][2] = 0;
                } else if (i + j < nf) {
                    s_ad[i][j][0] = s_sq[i + j][0]; s_ad[i][j][1] = s_sq[i + j][1]; s_ad[i][j][2] = s_sq[i + j][2];
                    const U192 nv = u_add(u_make(s_sq[i + j][0], s_sq[i + j][1], s_sq[i + j][2]), nrem);
                    s_sq[i + j][0] = nv.w[0]; s_sq[i + j][1] = nv.w[1]; s_sq[i + j][2] = nv.w[2];
                }
                limb = u_shr(limb, W);
                off += 22 + (j ? 4 : 0);
            }
            s_fresh[i][0] = s_sq[i][0]; s_fresh[i][1] = s_sq[i][1];
        }
        s_roff[nf] = off;
        unsigned eq = 1;
        s_eq[0] = 1;
        for (unsigned i = 0; i < nf; ++i) {
            eq &= u_eq(LIMB(s_c, i), LIMB(s_res, i)) ? 1u : 0u;
            s_eq[i + 1] = (unsigned char)eq;
        }
    }
    __syncthreads();
    // ---- assign_integer of n, g, x, y (Ln limbs) and of res (2 Ln limbs)
    for (unsigned which = 0; which < 5; ++which) {
        const unsigned nl = which < 4 ? Ln : nf;
        const u64(*X)[2] = which < 4 ? s_in[which] : s_res;
        const CellPtr a = adv + (which < 4 ? C.a_assign[which] : C.a_res);
        const CellPtr l = lk + (which < 4 ? C.l_assign[which] : C.l_res);
        for (unsigned t = tid; t < nl * (1 + C.rc_adv); t += blockDim.x) {
            Fr v;
            if (t < nl) v = fr_from_u(LIMB(X, t));
            else v = cv_to_fr(rc_adv_cell(LIMB(X, (t - nl) / C.rc_adv), W, lb, (t - nl) % C.rc_adv));
            fp_store(a + t, v);
        }
        if (l)
            for (unsigned t = tid; t < nl * C.rc_lk; t += blockDim.x) fp_store(l + t, cv_to_fr(rc_lk_cell(LIMB(X, t / C.rc_lk), W, lb, t % C.rc_lk)));
    }
    // ---- refresh
    if (tid == 0) fp_store(adv + C.a_refresh, fp_zero<FrTag>());
    for (unsigned i = 0; i < nf; ++i) {
        const unsigned len = s_roff[i + 1] - s_roff[i];
        for (unsigned t = tid; t < len; t += blockDim.x) {
            // blocks: j = 0: 22 cells; j > 0: 22 + 4 cells
            unsigned j = 0, p = t;
            if (p >= 22) { j = 1 + (p - 22) / 26; p = (p - 22) % 26; }
            const U192 v = u_make(s_dm[i][j][0], s_dm[i][j][1], s_dm[i][j][2]);
            Fr x;
            if (p < 22) x = cv_to_fr(div_mod_cell(v, p, W));
            else {
                const U192 before = u_make(s_ad[i][j][0], s_ad[i][j][1], s_ad[i][j][2]), nrem = u_lowbits(v, W);
                x = p == 22 ? fr_from_u(before) : p == 23 ? fp_one<FrTag>() : p == 24 ? fr_from_u(nrem) : fr_from_u(u_add(before, nrem));
            }
            fp_store(adv + C.a_refresh + s_roff[i] + t, x);
        }
    }
    for (unsigned t = tid; t < nf * C.rc_adv; t += blockDim.x)
        fp_store(adv + C.a_refresh + s_roff[nf] + t, cv_to_fr(rc_adv_cell(LIMB(s_fresh, t / C.rc_adv), W, lb, t % C.rc_adv)));
    if (lk)
        for (unsigned t = tid; t < nf * C.rc_lk; t += blockDim.x)
            fp_store(lk + C.l_refresh + t, cv_to_fr(rc_lk_cell(LIMB(s_fresh, t / C.rc_lk), W, lb, t % C.rc_lk)));
    // ---- load_zero; assign_constant(1) + load_zero of both pow_mod_fixed_exp
    if (tid == 0) {
        fp_store(adv + C.a_zero, fp_zero<FrTag>());
        if (C.kind != 1)
            for (int k = 0; k < 2; ++k) {
                fp_store(adv + C.a_pow[k], fp_one<FrTag>());
                fp_store(adv + C.a_pow[k] + 1, fp_zero<FrTag>());
            }
        fp_store(adv + C.a_eq, fp_zero<FrTag>());
        fp_store(adv + C.a_eq + 1, fp_one<FrTag>());
    }
    // ---- assert_equal_fresh(c, res)
    for (unsigned t = tid; t < 16 * nf; t += blockDim.x) {
        const unsigned i = t / 16, p = t % 16;
        Fr v;
        if (p < 12) v = cv_to_fr(is_equal_cell(LIMB(s_c, i), LIMB(s_res, i), p));
        else {
            const unsigned e = u_eq(LIMB(s_c, i), LIMB(s_res, i)) ? 1u : 0u, in = s_eq[i], out = s_eq[i + 1];
            v = p == 12 ? fp_zero<FrTag>() : p == 13 ? (in ? fp_one<FrTag>() : fp_zero<FrTag>())
                : p == 14 ? (e ? fp_one<FrTag>() : fp_zero<FrTag>()) : (out ? fp_one<FrTag>() : fp_zero<FrTag>());
        }
        fp_store(adv + C.a_eq + 2 + t, v);
    }
#undef LIMB
}

// ---- uniform-shape circuit: FlexGate::num_to_bits of the message's limbs and the limb-wise select after
// every mul_mod(acc, sq) of pow_mod.  grid.x = limb of m (num_to_bits) / exponent bit (select).
__global__ __launch_bounds__(256) void k_circuit_bits(ExpP P, unsigned Ln, unsigned words_n, const u64* __restrict__ m_words,
                                                      size_t cell_base, size_t limb_stride, Fr* __restrict__ advice) {
    const unsigned li = blockIdx.x, W = P.W;
    u64 lw[2];
    limb_extract(m_words, words_n, li, W, lw);
    const U192 x = u_make(lw[0], lw[1]);
    const CellPtr a = adv_ptr(P, advice, cell_base + (size_t)li * limb_stride);
    const unsigned nip = 1 + 3 * (W - 1);
    for (unsigned t = threadIdx.x; t < nip + 4 * W; t += blockDim.x) {
        Fr v;
        if (t == 0) v = fr_from_u(u_lowbits(x, 1));
        else if (t < nip) {
            const unsigned i = (t - 1) / 3 + 1, w = (t - 1) % 3;
            v = w == 0 ? fr_from_u(u_lowbits(u_shr(x, i), 1)) : w == 1 ? fr_from_u(u_shl(u_make(1), i)) : fr_from_u(u_lowbits(x, i + 1));
        } else {
            const unsigned i = (t - nip) / 4, w = (t - nip) % 4;
            v = w == 0 ? fp_zero<FrTag>() : fr_from_u(u_lowbits(u_shr(x, i), 1));
        }
        fp_store(a + t, v);
    }
}
__global__ __launch_bounds__(256) void k_circuit_select(ExpP P, unsigned Ln, unsigned words_n, const u64* __restrict__ m_words,
                                                        const u64* __restrict__ steps, size_t cell_base, size_t limb_stride,
                                                        size_t bit_stride, size_t nbits_cells, Fr* __restrict__ advice) {
    const unsigned i = blockIdx.x, W = P.W, L = P.L;        // exponent bit i = limb li, bit bi
    const unsigned li = i / W, bi = i % W;
    u64 lw[2];
    limb_extract(m_words, words_n, li, W, lw);
    const unsigned bit = (unsigned)(u_lowbits(u_shr(u_make(lw[0], lw[1]), bi), 1).w[0]);
    const u64* st = steps + (size_t)(2 * i) * 4 * P.L64;    // the mul_mod(acc, sq) step: a = acc, r = muled
    // cells of this bit: after its limb's num_to_bits block and its mul_mod step
    const CellPtr a = adv_ptr(P, advice, cell_base + (size_t)li * limb_stride + nbits_cells + (size_t)bi * bit_stride + P.cells);
    for (unsigned t = threadIdx.x; t < 8 * L; t += blockDim.x) {
        const unsigned limb = t / 8, p = t % 8;
        u64 aw[2], mw[2];
        limb_extract(st, P.L64, limb, W, aw);                           // acc
        limb_extract(st + 3 * (size_t)P.L64, P.L64, limb, W, mw);       // muled
        const U192 acc = u_make(aw[0], aw[1]), mul = u_make(mw[0], mw[1]);
        const S192 d = s_sub(mul, acc);
        Fr v;
        switch (p) {
            case 0: case 6: v = fr_from_s(d); break;
            case 1: v = fp_one<FrTag>(); break;
            case 2: case 4: v = fr_from_u(acc); break;
            case 3: v = fr_from_u(mul); break;
            case 5: v = bit ? fp_one<FrTag>() : fp_zero<FrTag>(); break;
            default: v = fr_from_u(bit ? mul : acc); break;
        }
        fp_store(a + t, v);
    }
}

// RefreshAux::new(limb_bits, l, r).increased_limbs_vec with the maximal limb values tracked as bit lengths + exact
// small big-integers (3 x 64-bit words are enough: a product limb is below 2^(2W + 8))
static int refresh_aux_host(unsigned W, unsigned nl, unsigned nr, unsigned char* inc, unsigned* n_out) {
    const unsigned d = nl + nr - 1;
    std::vector<std::array<uint64_t, 3>> muled;
    uint64_t mx2[3], one[3] = {1, 0, 0}, t[3];
    // (2^W - 1)^2 = 2^(2W) - 2^(W+1) + 1
    h_shl(one, 2 * W, mx2);
    h_shl(one, W + 1, t);
    h_sub(mx2, t);
    h_add(mx2, one);
    for (unsigned i = 0; i < d; ++i) {
        unsigned cnt = 0;
        for (unsigned j = 0; j < nl; ++j)
            if (i >= j && i - j < nr) ++cnt;
        std::array<uint64_t, 3> v = {0, 0, 0};
        for (unsigned k = 0; k < cnt; ++k) h_add(v.data(), mx2);
        muled.push_back(v);
    }
    unsigned cur = 0, n = 0;
    while (cur < muled.size()) {
        if (n >= CIRC_MAXF) return PZ_ERR_UNSUPPORTED;
        const unsigned bits = h_bits(muled[cur].data());
        unsigned chunks = (bits + W - 1) / W;
        if (chunks == 0) chunks = 1;
        if (chunks > CIRC_MAXJ) return PZ_ERR_UNSUPPORTED;
        inc[n++] = (unsigned char)(chunks - 1);
        uint64_t val[3] = {muled[cur][0], muled[cur][1], muled[cur][2]};
        for (unsigned i = 0; i < chunks; ++i) {
            // piece = val mod 2^W ; val >>= W
            uint64_t piece[3] = {val[0], val[1], val[2]};
            if (W < 64) { piece[0] &= (1ull << W) - 1; piece[1] = 0; piece[2] = 0; }
            else if (W == 64) { piece[1] = 0; piece[2] = 0; }
            else { piece[1] &= (1ull << (W - 64)) - 1; piece[2] = 0; }
            unsigned sft = W;
            while (sft >= 64) { val[0] = val[1]; val[1] = val[2]; val[2] = 0; sft -= 64; }
            if (sft) { val[0] = (val[0] >> sft) | (val[1] << (64 - sft)); val[1] = (val[1] >> sft) | (val[2] << (64 - sft)); val[2] >>= sft; }
            if (cur + i < muled.size()) {
                if (i == 0) muled[cur] = {piece[0], piece[1], piece[2]};
                else h_add(muled[cur + i].data(), piece);
            } else muled.push_back({piece[0], piece[1], piece[2]});
        }
        ++cur;
    }
    *n_out = n;
    return PZ_OK;
}

static int make_circuit_params(int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lb, size_t ng, size_t nr, CircP& C,
                               size_t* adv_total, size_t* lk_total, size_t step_off[3]) {
    if (kind < 0 || kind > 2) return PZ_ERR_INVALID;
    if (limbs_n < 1 || 2 * limbs_n > CIRC_MAXF || 2 * limbs_n > EXP_MAXL) return PZ_ERR_UNSUPPORTED;
    memset(&C, 0, sizeof C);
    PZCHK(make_params(2 * limbs_n, limb_bits, lb, C.e));
    C.Ln = limbs_n;
    C.kind = (unsigned)kind;
    C.words_n = (limbs_n * limb_bits + 63) / 64;
    C.rc_adv = C.e.rc64_adv;
    C.rc_lk = C.e.rc64_lk;
    PZCHK(refresh_aux_host(limb_bits, limbs_n, limbs_n, C.inc, &C.nf));
    if (C.nf != 2 * limbs_n) return PZ_ERR_UNSUPPORTED;   // the chip's mul_mod needs n^2 at the operands' limb count
    size_t a = 0, l = 0;
    const size_t asg_a = (size_t)limbs_n * (1 + C.rc_adv), asg_l = (size_t)limbs_n * C.rc_lk;
    for (int k = 0; k < 4; ++k) { C.a_assign[k] = a; C.l_assign[k] = l; a += asg_a; l += asg_l; }
    C.a_square = a;
    a += 1;
    for (unsigned i = 0; i < 2 * limbs_n - 1; ++i) a += 1 + 3 * ((size_t)i + 1);
    C.a_refresh = a; C.l_refresh = l;
    a += 1;
    for (unsigned i = 0; i < C.nf; ++i) a += ((size_t)C.inc[i] + 1) * 22 + (size_t)C.inc[i] * 4;
    a += (size_t)C.nf * C.rc_adv;
    l += (size_t)C.nf * C.rc_lk;
    C.a_zero = a; a += 1;
    size_t lstep[3] = {0, 0, 0};
    if (kind == 0) {
        C.a_pow[0] = a; a += 2; step_off[0] = a; lstep[0] = l; a += ng * C.e.cells; l += ng * C.e.lookups;
        C.a_pow[1] = a; a += 2; step_off[1] = a; lstep[1] = l; a += nr * C.e.cells; l += nr * C.e.lookups;
    } else if (kind == 2) {
        // uniform shape: per limb of m [num_to_bits | per bit: mul_mod, select (8 cells per limb), square_mod]
        const size_t m_bits = (size_t)limbs_n * limb_bits;
        if (ng != 2 * m_bits) return PZ_ERR_INVALID;
        C.a_pow[0] = a; a += 2; step_off[0] = a; lstep[0] = l;
        a += (size_t)limbs_n * (7 * (size_t)limb_bits - 2) + m_bits * (2 * C.e.cells + 8 * (size_t)C.e.L);
        l += ng * C.e.lookups;
        C.a_pow[1] = a; a += 2; step_off[1] = a; lstep[1] = l; a += nr * C.e.cells; l += nr * C.e.lookups;
    } else if (ng || nr) return PZ_ERR_INVALID;
    step_off[2] = a; lstep[2] = l; a += C.e.cells; l += C.e.lookups;
    C.a_res = a; C.l_res = l;
    a += 2 * asg_a; l += 2 * asg_l;
    C.a_eq = a;
    a += 2 + 16 * (size_t)C.nf;
    *adv_total = a;
    *lk_total = l;
    // step_off[0..2]: advice offsets of the three runs of mul_mod steps (g^m, r^n, final); [3..5]: their lookup offsets
    memcpy(step_off + 3, lstep, sizeof lstep);
    return PZ_OK;
}

// RefreshAux::new(limb_bits, l, r).increased_limbs_vec for the host mirror (paillier.rs:40-44)
extern "C" int pz_refresh_aux(uint32_t limb_bits, uint32_t num_limbs_l, uint32_t num_limbs_r, uint8_t* increased_limbs,
                              uint32_t capacity, uint32_t* n_out) {
    if (!increased_limbs || !n_out || limb_bits < 16 || limb_bits > 90 || !num_limbs_l || !num_limbs_r) return PZ_ERR_INVALID;
    unsigned char inc[CIRC_MAXF];
    unsigned n = 0;
    PZCHK(refresh_aux_host(limb_bits, num_limbs_l, num_limbs_r, inc, &n));
    if (n > capacity) return PZ_ERR_CAPACITY;
    memcpy(increased_limbs, inc, n);
    *n_out = n;
    return PZ_OK;
}

// cells one chip operation pushes (the terms pz_circuit_cells sums): op 0 assign_integer(limbs), 1 square(limbs),
// 2 refresh of a limbs x limbs product, 3 load_zero / load_constant (one cell), 4 mul_mod(limbs), 5 assert_equal_fresh(limbs)
extern "C" int pz_op_cells(int op, uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits, size_t* advice_cells,
                           size_t* lookup_cells) {
    size_t a = 0, l = 0;
    unsigned k, rem, rc_a, rc_l;
    if (limb_bits < 16 || limb_bits > 90 || lookup_bits < 4 || lookup_bits > 32) return PZ_ERR_INVALID;
    rc_counts(limb_bits, lookup_bits, k, rem, rc_a, rc_l);
    switch (op) {
        case 0: a = (size_t)limbs * (1 + rc_a); l = (size_t)limbs * rc_l; break;
        case 1:
            a = 1;
            for (unsigned i = 0; i + 1 < 2 * limbs; ++i) a += 1 + 3 * ((size_t)i + 1);
            break;
        case 2: {
            unsigned char inc[CIRC_MAXF];
            unsigned n = 0;
            PZCHK(refresh_aux_host(limb_bits, limbs, limbs, inc, &n));
            a = 1;
            for (unsigned i = 0; i < n; ++i) a += ((size_t)inc[i] + 1) * 22 + (size_t)inc[i] * 4;
            a += (size_t)n * rc_a;
            l = (size_t)n * rc_l;
            break;
        }
        case 3: a = 1; break;
        case 4: {
            ExpP P;
            PZCHK(make_params(limbs, limb_bits, lookup_bits, P));
            a = P.cells;
            l = P.lookups;
            break;
        }
        case 5: a = 2 + 16 * (size_t)limbs; break;
        default: return PZ_ERR_INVALID;
    }
    if (advice_cells) *advice_cells = a;
    if (lookup_cells) *lookup_cells = l;
    return PZ_OK;
}

extern "C" int pz_circuit_cells(int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits, size_t n_steps_g,
                                size_t n_steps_r, size_t* advice_cells, size_t* lookup_cells) {
    CircP C;
    size_t a, l, so[6];
    PZCHK(make_circuit_params(kind, limbs_n, limb_bits, lookup_bits, n_steps_g, n_steps_r, C, &a, &l, so));
    if (advice_cells) *advice_cells = a;
    if (lookup_cells) *lookup_cells = l;
    return PZ_OK;
}

// halo2-lib's column break rule (assign_with_constraints of the dependency [D]), from the selector mask of the advice stream: walking
// the stream with row_offset r inside the current column, cell i is placed at (column, r); then, if (q[i] && r + 4 > max_rows) ||
// r >= max_rows - 1, the column ends with this cell, the next column starts with a COPY of it at row 0 (tied by an equality
// constraint) and -- if q[i] -- its gate is enabled there, not in the column it left.  starts[j] = the stream index of column j's row 0.
extern "C" int pz_circuit_break_points(const uint8_t* gate_mask, size_t n_cells, size_t max_rows, uint64_t* starts_out, size_t capacity,
                                       size_t* n_cols) {
    if (!gate_mask || !n_cols || max_rows < 8) return PZ_ERR_INVALID;
    size_t cols = 0, s = 0;
    for (;;) {
        if (starts_out) {
            if (cols >= capacity) return PZ_ERR_CAPACITY;
            starts_out[cols] = s;
        }
        ++cols;
        // the column ends at the first i with (q[i] && i - s + 4 > max_rows) || i - s >= max_rows - 1
        size_t end = s + max_rows - 1;
        for (size_t i = s + max_rows - 3; i < s + max_rows - 1 && i < n_cells; ++i)
            if (gate_mask[i]) {
                // the dependency asserts that no gate started one or two cells earlier (overlaps are by exactly three cells)
                if ((i >= 1 && gate_mask[i - 1] && i - 1 > s) || (i >= 2 && gate_mask[i - 2] && i - 2 > s)) return PZ_ERR_UNSUPPORTED;
                end = i;
                break;
            }
        if (end >= n_cells) break;     // the stream ends inside this column
        s = end;
    }
    if (starts_out) {
        if (cols >= capacity) return PZ_ERR_CAPACITY;
        starts_out[cols] = n_cells;
    }
    *n_cols = cols;
    return PZ_OK;
}

static int circuit_expand_impl(pz_ctx* ctx, int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits,
                               const uint64_t* inputs, const uint64_t* d_steps, size_t n_steps_g, size_t n_steps_r,
                               const uint64_t* d_modulus, uint64_t* d_advice, uint64_t* d_lookup, size_t rows, size_t col_stride,
                               const uint64_t* d_col_starts, size_t n_adv_cols, size_t max_rows) {
    if (!ctx || !inputs || !d_steps || !d_modulus || !d_advice) return PZ_ERR_INVALID;
    if ((rows == 0) != (col_stride == 0) || col_stride < rows) return PZ_ERR_INVALID;
    CircP C;
    size_t a, l, so[6];
    PZCHK(make_circuit_params(kind, limbs_n, limb_bits, lookup_bits, n_steps_g, n_steps_r, C, &a, &l, so));
    C.e.rows = rows;
    C.e.pad = col_stride - rows;
    if (d_col_starts) {
        if (n_adv_cols == 0 || n_adv_cols > 0xffffffffu || max_rows < 8 || max_rows > col_stride) return PZ_ERR_INVALID;
        C.e.bp_starts = (const u64*)d_col_starts;
        C.e.bp_ncols = (unsigned)n_adv_cols;
        C.e.bp_div = max_rows - 1;
        C.e.bp_stride = col_stride;
    }
    PZ_ENTER(ctx);
    const size_t in_words = 4 * (size_t)C.words_n + C.e.L64;
    void* d_in;
    PZCHK(pz_ws_get(ctx, WS_MISC, in_words * 8 + 64, &d_in));
    // `inputs` is the caller's (pageable) memory and may go away when this returns: the few hundred bytes go through a pinned
    // staging block of the context, so the copy is truly asynchronous and reads nothing of the caller's afterwards
    PZCHK(pz_upload_small_async(ctx, d_in, inputs, in_words * 8));
    pz_timer tm(ctx, PZ_T_EXPAND);
    const size_t rec = 4 * (size_t)C.e.L64;   // words per step record
    const size_t runs[3] = {kind != 1 ? n_steps_g : 0, kind != 1 ? n_steps_r : 0, 1};
    size_t first = 0;
    for (int k = 0; k < 3; ++k) {
        if (runs[k]) {
            ExpP P = C.e;
            P.cell0 = so[k];
            P.lk0 = so[3 + k];
            if (kind == 2 && k == 0) {
                // the g^m steps come in (mul_mod, square_mod) pairs with the select cells in between and a num_to_bits block
                // in front of every limb's 64 bits: launch limb by limb
                const size_t W = limb_bits, nbc = 7 * W - 2, bit_stride = 2 * C.e.cells + 8 * (size_t)C.e.L;
                const size_t limb_stride = nbc + W * bit_stride;
                hipLaunchKernelGGL(k_circuit_bits, dim3(limbs_n), dim3(256), 0, ctx->stream, C.e, limbs_n, C.words_n,
                                   (const u64*)d_in + 2 * (size_t)C.words_n, so[0], limb_stride, (Fr*)d_advice);
                hipLaunchKernelGGL(k_circuit_select, dim3((unsigned)(limbs_n * W)), dim3(256), 0, ctx->stream, C.e, limbs_n, C.words_n,
                                   (const u64*)d_in + 2 * (size_t)C.words_n, d_steps, so[0], limb_stride, bit_stride, nbc, (Fr*)d_advice);
                for (unsigned li = 0; li < limbs_n; ++li) {
                    ExpP Q = P;
                    Q.cell0 = so[0] + li * limb_stride + nbc;
                    Q.lk0 = so[3] + (size_t)li * 2 * W * C.e.lookups;
                    Q.pair_stride = bit_stride;
                    Q.odd_off = C.e.cells + 8 * (size_t)C.e.L;
                    hipLaunchKernelGGL(k_witness_expand, dim3((unsigned)(2 * W)), dim3(EXP_THREADS), 0, ctx->stream, Q,
                                       d_steps + (size_t)li * 2 * W * rec, d_modulus, (Fr*)d_advice, (Fr*)d_lookup);
                }
                first += runs[k];
                continue;
            }
            hipLaunchKernelGGL(k_witness_expand, dim3((unsigned)runs[k]), dim3(EXP_THREADS), 0, ctx->stream, P, d_steps + first * rec,
                               d_modulus, (Fr*)d_advice, (Fr*)d_lookup);
        }
        first += runs[k];
    }
    // the circuit's result c = remainder of the last step (the final mul_mod)
    const uint64_t* d_c = d_steps + (first - 1) * rec + 3 * (size_t)C.e.L64;
    hipLaunchKernelGGL(k_circuit_misc, dim3(1), dim3(256), 0, ctx->stream, C, (const u64*)d_in, (const u64*)d_c, (Fr*)d_advice,
                       (Fr*)d_lookup);
    HIPCHK(ctx, hipGetLastError());
    return PZ_OK;
}

extern "C" int pz_circuit_expand_dev(pz_ctx* ctx, int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits,
                                     const uint64_t* inputs, const uint64_t* d_steps, size_t n_steps_g, size_t n_steps_r,
                                     const uint64_t* d_modulus, uint64_t* d_advice, uint64_t* d_lookup, size_t rows,
                                     size_t col_stride) {
    return circuit_expand_impl(ctx, kind, limbs_n, limb_bits, lookup_bits, inputs, d_steps, n_steps_g, n_steps_r, d_modulus, d_advice,
                               d_lookup, rows, col_stride, nullptr, 0, 0);
}
extern "C" int pz_circuit_expand_cols_dev(pz_ctx* ctx, int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits,
                                          const uint64_t* inputs, const uint64_t* d_steps, size_t n_steps_g, size_t n_steps_r,
                                          const uint64_t* d_modulus, uint64_t* d_advice, uint64_t* d_lookup,
                                          const uint64_t* d_col_starts, size_t n_adv_cols, size_t max_rows, size_t lookup_rows,
                                          size_t col_stride) {
    if (!d_col_starts || lookup_rows == 0 || lookup_rows > col_stride) return PZ_ERR_INVALID;
    return circuit_expand_impl(ctx, kind, limbs_n, limb_bits, lookup_bits, inputs, d_steps, n_steps_g, n_steps_r, d_modulus, d_advice,
                               d_lookup, lookup_rows, col_stride, d_col_starts, n_adv_cols, max_rows);
}

// host-pointer form (SURVEY.md section 8b `pz_witness_expand`): stages the trace up, expands on the device in groups
// of steps and copies the cell streams back; the prover pipeline keeps everything resident and uses the _dev form.
extern "C" int pz_witness_expand(pz_ctx* ctx, uint32_t limbs, uint32_t limb_bits, uint32_t lookup_bits, const uint64_t* steps,
                                 size_t n_steps, const uint64_t* modulus, uint64_t* advice_out, uint64_t* lookup_out) {
    if (!ctx || (n_steps && (!steps || !modulus))) return PZ_ERR_INVALID;
    ExpP P;
    PZCHK(make_params(limbs, limb_bits, lookup_bits, P));
    if (!n_steps || (!advice_out && !lookup_out)) return PZ_OK;
    PZ_ENTER(ctx);
    const size_t rec = 4 * (size_t)P.L64 * 8;  // bytes per step record
    // groups of steps bounded to ~512 MiB of cells
    size_t group = ((size_t)512 << 20) / (P.cells * 32 + 1);
    if (group == 0) group = 1;
    if (group > n_steps) group = n_steps;
    void *d_steps, *d_mod, *d_adv = nullptr, *d_lk = nullptr;
    PZCHK(pz_ws_get(ctx, WS_IO_A, group * rec + 64, &d_steps));
    PZCHK(pz_ws_get(ctx, WS_MISC, (size_t)P.L64 * 8, &d_mod));
    if (advice_out) PZCHK(pz_ws_get(ctx, WS_IO_B, group * P.cells * 32, &d_adv));
    if (lookup_out) PZCHK(pz_ws_get(ctx, WS_IO_C, group * P.lookups * 32 + 32, &d_lk));
    HIPCHK(ctx, hipMemcpyAsync(d_mod, modulus, (size_t)P.L64 * 8, hipMemcpyHostToDevice, ctx->stream));
    for (size_t s0 = 0; s0 < n_steps; s0 += group) {
        const size_t ns = n_steps - s0 < group ? n_steps - s0 : group;
        HIPCHK(ctx, hipMemcpyAsync(d_steps, (const char*)steps + s0 * rec, ns * rec, hipMemcpyHostToDevice, ctx->stream));
        PZCHK(pz_witness_expand_dev(ctx, limbs, limb_bits, lookup_bits, (const uint64_t*)d_steps, ns, (const uint64_t*)d_mod,
                                    (uint64_t*)d_adv, (uint64_t*)d_lk));
        if (advice_out)
            HIPCHK(ctx, hipMemcpyAsync((char*)advice_out + s0 * P.cells * 32, d_adv, ns * P.cells * 32, hipMemcpyDeviceToHost,
                                       ctx->stream));
        if (lookup_out)
            HIPCHK(ctx, hipMemcpyAsync((char*)lookup_out + s0 * P.lookups * 32, d_lk, ns * P.lookups * 32, hipMemcpyDeviceToHost,
                                       ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PZ_OK;
}
