// pz_structure.hip -- the STRUCTURE of the reference's circuits as halo2's keygen sees it, generated behind the C ABI, ON THE DEVICE:
// selector positions, the copy-constraint permutation (as the (column, row) every cell maps to), the constants column and the
// break-point column layout -- what `keygen_vk` / `keygen_pk` extract by running `synthesize` of halo2-lib's builder over the drivers
// (/root/reference/src/bench.rs:33-117 with PaillierChip::{encrypt, add}, src/paillier.rs:32-85; reached from bench.rs:161-175).
// The structure depends on the SHAPE only: key size, limb width, lookup bits and -- paillier.rs:50-55 pulls m and n out of the witness
// and hands them to pow_mod_fixed_exp -- the BITS of the two exponents, so a new message of the reference's circuit needs a new
// structure before keygen.  This is the compiled counterpart of paillier_halo2_amd/circuit_structure.py (same walk, same order of
// constants, same cycles: tests/test_gpu_structure.py compares them array for array), so that a caller with no Python -- the reference's
// Rust prover patched at INTEGRATION.md point D, tests/cpp/prove_connected -- gets structure -> pz_pk_create_dev -> proof from the
// library alone.
//
// How: a value-free walk of ONE mul_mod block on the host records, per cell, the cell it copies (a block-local cell, an operand limb, a
// limb of the refreshed n^2, a constant), its gate windows and lookups (~65 000 cells at config c2); the circuit's ~6 000 identical
// blocks are NEVER materialised on the host: a kernel computes, for each of the 4 x 10^8 cells, what it copies from the template and
// the per-block operand tables (12 MB).  Equality classes become cycles of sigma by pointer jumping + ONE radix sort of
// (class, position) keys (rocPRIM through hipCUB: the only library primitive; everything else is elementwise / gather work bound by
// HBM, no arithmetic on field elements at all).
// The per-primitive patterns restate halo2-lib / biguint-halo2 [D] exactly as K4 (pz_witness.hip) does: layout parity with the
// reference's floating dependency versions is unpinned.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <vector>

#include "pz_internal.h"

namespace {
typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

// ---------------------------------------------------------------------------------------------- 256-bit constants (host)
struct C256 {
    u64 w[4];
    bool operator<(const C256& o) const {
        for (int i = 3; i >= 0; --i)
            if (w[i] != o.w[i]) return w[i] < o.w[i];
        return false;
    }
};
C256 c_small(u64 v) { return C256{{v, 0, 0, 0}}; }
C256 c_pow2(unsigned bits) {
    C256 r{{0, 0, 0, 0}};
    r.w[bits / 64] = (u64)1 << (bits % 64);
    return r;
}
C256 c_add(const C256& a, const C256& b) {
    C256 r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (unsigned __int128)a.w[i] + b.w[i];
        r.w[i] = (u64)c;
        c >>= 64;
    }
    return r;
}
C256 c_sub(const C256& a, const C256& b) {
    C256 r;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - br;
        r.w[i] = (u64)d;
        br = (d >> 64) & 1;
    }
    return r;
}
C256 c_mul_small(const C256& a, u64 k) {
    C256 r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (unsigned __int128)a.w[i] * k;
        r.w[i] = (u64)c;
        c >>= 64;
    }
    return r;
}
unsigned c_bits(const C256& a) {
    for (int i = 3; i >= 0; --i)
        if (a.w[i]) return 64 * i + (64 - __builtin_clzll(a.w[i]));
    return 0;
}
// L * (2^W - 1)^2 + (2^W - 1): the bound of one product limb plus a remainder limb (is_equal_muled's carries are sized by it)
C256 c_max_word(unsigned L, unsigned W) {
    const C256 m = c_sub(c_pow2(W), c_small(1));
    const C256 m2 = c_add(c_sub(c_pow2(2 * W), c_pow2(W + 1)), c_small(1));
    return c_add(c_mul_small(m2, L), m);
}

// ---------------------------------------------------------------------------------------------- the walk (host)
// A cell reference: >= 0 a stream index (global, or local inside a block template); NONE; or an operand limb of the block (kind, j).
const i64 NONE = -1;
enum { EXT_A = 0, EXT_B = 1, EXT_N = 2, EXT_S = 3 };
inline i64 ext_ref(int kind, int j) { return -(1000 + (i64)kind * 65536 + j); }
inline bool is_ext(i64 v) { return v <= -1000; }
inline int ext_kind(i64 v) { return (int)((-v - 1000) >> 16); }
inline int ext_j(i64 v) { return (int)((-v - 1000) & 0xffff); }

struct Walk {
    i64 base = 0;
    std::vector<i64> src;          // NONE | index | ext_ref
    std::vector<int32_t> cidx;     // -1 or index into cvals
    std::vector<C256> cvals;
    std::vector<i64> gates;        // local positions
    std::vector<i64> lk;           // cell references, lookup-stream order
    explicit Walk(i64 b = 0) : base(b) {}
    i64 put(i64 s = NONE) {
        src.push_back(s);
        cidx.push_back(-1);
        return base + (i64)src.size() - 1;
    }
    i64 putc(const C256& v) {
        src.push_back(NONE);
        cidx.push_back((int32_t)cvals.size());
        cvals.push_back(v);
        return base + (i64)src.size() - 1;
    }
    void pair(i64 s, i64 copy) {
        if (s == NONE) return;
        src[(size_t)(copy - base)] = s;
    }
    void gate(i64 cell) { gates.push_back(cell - base); }
    size_t n() const { return src.size(); }
};

// RangeChip::range_check: digits + running recomposition, then the top digit's tail gate; -> the cell holding the value
i64 range_check(Walk& w, i64 xcell, unsigned bits, unsigned lb) {
    const unsigned k = (bits + lb - 1) / lb, rem = bits % lb;
    i64 last_cell = xcell, holder = xcell;
    if (k > 1) {
        const i64 c0 = w.put();
        w.lk.push_back(c0);
        w.gate(c0);
        i64 acc_cell = c0;
        for (unsigned gi = 1; gi < k; ++gi) {
            last_cell = w.put();
            w.lk.push_back(last_cell);
            w.putc(c_pow2(lb * gi));
            acc_cell = w.put();
            if (gi < k - 1) w.gate(acc_cell);
        }
        w.pair(xcell, acc_cell);
        holder = acc_cell;
    } else {
        w.lk.push_back(xcell);
    }
    if (rem == 1) {
        const i64 z = w.putc(c_small(0));
        w.gate(z);
        w.put(last_cell); w.put(last_cell); w.put(last_cell);
    } else if (rem > 1) {
        const i64 z = w.putc(c_small(0));
        w.gate(z);
        w.put(last_cell);
        w.putc(c_pow2(lb - rem));
        w.lk.push_back(w.put());
    }
    return holder;
}
std::vector<i64> assign(Walk& w, unsigned nl, unsigned W, unsigned lb) {
    std::vector<i64> cells(nl);
    for (unsigned i = 0; i < nl; ++i) cells[i] = w.put();
    for (i64 c : cells) range_check(w, c, W, lb);
    return cells;
}
std::vector<i64> mul_cells(Walk& w, const std::vector<i64>& xs, const std::vector<i64>& ys, unsigned D) {
    const i64 zc = w.putc(c_small(0));
    std::vector<i64> xe(xs), ye(ys);
    xe.resize(D, zc);
    ye.resize(D, zc);
    std::vector<i64> prod;
    for (unsigned i = 0; i < D; ++i) {
        const i64 z = w.putc(c_small(0));
        w.gate(z);
        i64 cell = NONE;
        for (unsigned j = 0; j <= i; ++j) {
            w.put(xe[j]);
            w.put(ye[i - j]);
            cell = w.put();
            if (j < i) w.gate(cell);
        }
        prod.push_back(cell);
    }
    return prod;
}
void is_equal(Walk& w, i64 xc, i64 yc) {
    const i64 c_d = w.put();
    w.gate(c_d);
    w.put(yc); w.putc(c_small(1)); w.put(xc);
    const i64 c_z = w.put();
    w.gate(c_z);
    const i64 c_a = w.put(c_d);
    w.put(); w.putc(c_small(1));
    const i64 z2 = w.putc(c_small(0));
    w.gate(z2);
    w.put(c_a); w.put(c_z); w.putc(c_small(0));
}
void div_mod(Walk& w, i64 vcell, unsigned W, i64* qd, i64* rd) {
    const i64 c_qd = w.put();
    const i64 c_rd = w.put();
    const i64 z = w.putc(c_small(0));
    w.gate(z);
    w.put(c_qd); w.putc(c_pow2(W));
    const i64 c_pr = w.put();
    const i64 d = w.put();
    w.gate(d);
    w.put(c_pr); w.putc(c_small(1)); w.put(vcell);
    is_equal(w, c_rd, NONE);
    *qd = c_qd;
    *rd = c_rd;
}
// BigUintChip::mul_mod: assign q, n, r; the two limb convolutions; qn + r; is_equal_muled's carry chain; r < n.  -> r's cells
std::vector<i64> mul_mod(Walk& w, const std::vector<i64>& a, const std::vector<i64>& b, const std::vector<i64>& nfresh, unsigned L, unsigned W,
                         unsigned lb) {
    const std::vector<i64> ql = assign(w, L, W, lb), nl = assign(w, L, W, lb), rl = assign(w, L, W, lb);
    for (unsigned i = 0; i < L; ++i) w.pair(nfresh[i], nl[i]);
    const unsigned D = 2 * L - 1;
    const std::vector<i64> p_ab = mul_cells(w, a, b, D), p_qn = mul_cells(w, ql, nl, D);
    std::vector<i64> qnr(p_qn);
    for (unsigned i = 0; i < L; ++i) {
        const i64 g = w.put(p_qn[i]);
        w.gate(g);
        w.putc(c_small(1)); w.put(rl[i]);
        qnr[i] = w.put();
    }
    const C256 MAX = c_max_word(L, W);
    const unsigned cb = c_bits(c_add(MAX, MAX)) - W;
    const i64 c_zero = w.putc(c_small(0)), c_one = w.putc(c_small(1));
    i64 carry = c_zero, accx = c_zero, eq_cell = c_one;
    for (unsigned i = 0; i < D; ++i) {
        const i64 c_diff = w.put();
        w.gate(c_diff);
        w.put(qnr[i]); w.putc(c_small(1)); w.put(p_ab[i]);
        i64 g = w.put(c_diff);
        w.gate(g);
        w.put(carry); w.putc(c_small(1));
        const i64 s1 = w.put();
        w.gate(s1);
        w.putc(MAX); w.putc(c_small(1));
        const i64 c_s = w.put();
        i64 new_carry, cmod, q_acc, mod_acc;
        div_mod(w, c_s, W, &new_carry, &cmod);
        g = w.put(accx);
        w.gate(g);
        w.putc(c_small(1)); w.putc(MAX);
        const i64 c_t = w.put();
        div_mod(w, c_t, W, &q_acc, &mod_acc);
        is_equal(w, cmod, mod_acc);
        g = w.putc(c_small(0));
        w.gate(g);
        w.put(eq_cell); w.put();
        eq_cell = w.put();
        accx = q_acc;
        if (i < D - 1) {
            range_check(w, new_carry, cb, lb);
        } else {
            is_equal(w, new_carry, accx);
            g = w.putc(c_small(0));
            w.gate(g);
            w.put(eq_cell); w.put();
            eq_cell = w.put();
        }
        carry = new_carry;
    }
    i64 borrow = c_zero;
    for (unsigned i = 0; i < L; ++i) {
        i64 g = w.put(nl[i]);
        w.gate(g);
        w.putc(c_small(1)); w.put(borrow); w.put();
        w.put();
        const i64 c_lt = w.put();
        const i64 c_out = w.put();
        g = w.put(rl[i]);
        w.gate(g);
        w.put(c_lt); w.putc(c_pow2(W)); w.put();
        range_check(w, c_out, W, lb);
        borrow = c_lt;
    }
    w.put(borrow);
    return rl;
}

// ---------------------------------------------------------------------------------------------- templates and parts (host)
struct Template {
    size_t cells = 0;
    std::vector<int32_t> sol;       // local source index, or the cell's own index
    std::vector<int8_t> kind;       // -1 | EXT_*
    std::vector<int16_t> limb;      // operand limb of an ext cell
    std::vector<int32_t> cid;       // -1 | id of the constant the cell is loaded as
    std::vector<u8> mask;
    std::vector<int32_t> lkpos;     // local positions, lookup-stream order
    std::vector<i64> r_cells;       // local positions of the block's outputs
    std::vector<C256> const_vals;   // in template order (ids are assigned by the stream walk)
    std::vector<int32_t> const_pos;
};
Template make_template(Walk& w, const std::vector<i64>& outs) {
    Template t;
    t.cells = w.n();
    t.sol.resize(t.cells); t.kind.assign(t.cells, -1); t.limb.assign(t.cells, 0); t.cid.assign(t.cells, -1); t.mask.assign(t.cells, 0);
    for (size_t i = 0; i < t.cells; ++i) {
        const i64 s = w.src[i];
        t.sol[i] = (int32_t)i;
        if (is_ext(s)) {
            t.kind[i] = (int8_t)ext_kind(s);
            t.limb[i] = (int16_t)ext_j(s);
        } else if (s >= 0) {
            t.sol[i] = (int32_t)s;
        }
        if (w.cidx[i] >= 0) {
            t.const_pos.push_back((int32_t)i);
            t.const_vals.push_back(w.cvals[(size_t)w.cidx[i]]);
        }
    }
    for (i64 g : w.gates) t.mask[(size_t)g] = 1;
    for (i64 c : w.lk) t.lkpos.push_back((int32_t)c);
    t.r_cells = outs;
    return t;
}
std::vector<i64> ext_vec(int kind, unsigned L) {
    std::vector<i64> v(L);
    for (unsigned j = 0; j < L; ++j) v[j] = ext_ref(kind, (int)j);
    return v;
}
Template block_template(unsigned L, unsigned W, unsigned lb) {
    Walk w;
    const std::vector<i64> rl = mul_mod(w, ext_vec(EXT_A, L), ext_vec(EXT_B, L), ext_vec(EXT_N, L), L, W, lb);
    return make_template(w, rl);
}
// one exponent bit of the uniform-shape circuit: mul_mod(acc, sq), the limb-wise select(bit, product, acc), square_mod(sq);
// outputs: the NEW acc (select outputs) then the NEW sq, L cells each
Template uniform_template(unsigned L, unsigned W, unsigned lb) {
    Walk w;
    const std::vector<i64> A = ext_vec(EXT_A, L), B = ext_vec(EXT_B, L), Nf = ext_vec(EXT_N, L);
    const std::vector<i64> mul = mul_mod(w, A, B, Nf, L, W, lb);
    std::vector<i64> outs;
    for (unsigned t = 0; t < L; ++t) {
        const i64 c_d = w.put();
        w.gate(c_d);
        w.putc(c_small(1)); w.put(ext_ref(EXT_A, (int)t)); w.put(mul[t]);
        const i64 g = w.put(ext_ref(EXT_A, (int)t));
        w.gate(g);
        w.put(ext_ref(EXT_S, 0)); w.put(c_d);
        outs.push_back(w.put());
    }
    const std::vector<i64> sq = mul_mod(w, B, B, Nf, L, W, lb);
    outs.insert(outs.end(), sq.begin(), sq.end());
    return make_template(w, outs);
}

struct Part {
    i64 cell_off = 0, lk_off = 0, n_cells = 0, n_lk = 0;
    int tmpl = -1;                    // -1: dense
    i64 ns = 0;
    std::vector<i64> src;             // dense: absolute source (self if none), -(1 + id) for constants
    std::vector<u8> mask;             // dense
    std::vector<i64> lk;              // dense: absolute cells
    std::vector<i64> a, b, s;         // blocks: [ns][L] operand cells, [ns] the bit's cell (uniform)
};

struct Stream {
    std::vector<Part> parts;
    std::vector<Template> tmpls;
    std::vector<C256> constants;
    std::map<C256, int> const_id;
    std::vector<i64> fresh;
    i64 n_cells = 0, n_lk = 0, result_cell = 0;
    size_t n_steps_g = 0, n_steps_r = 0;
    std::vector<std::vector<int32_t>> tmpl_cids;   // per template: constant id per const_pos entry
    int cid(const C256& v) {
        auto it = const_id.find(v);
        if (it != const_id.end()) return it->second;
        const int id = (int)constants.size();
        constants.push_back(v);
        const_id[v] = id;
        return id;
    }
    // a finished walked part -> arrays
    i64 flush(Walk& w) {
        Part p;
        p.cell_off = w.base;
        p.lk_off = n_lk;
        p.n_cells = (i64)w.n();
        p.src.resize(w.n());
        p.mask.assign(w.n(), 0);
        for (size_t i = 0; i < w.n(); ++i) {
            p.src[i] = w.src[i] == NONE ? w.base + (i64)i : w.src[i];
            if (w.cidx[i] >= 0) p.src[i] = -(1 + (i64)cid(w.cvals[(size_t)w.cidx[i]]));
        }
        for (i64 g : w.gates) p.mask[(size_t)g] = 1;
        p.lk = w.lk;
        p.n_lk = (i64)w.lk.size();
        n_lk += p.n_lk;
        n_cells = w.base + (i64)w.n();
        parts.push_back(std::move(p));
        return n_cells;
    }
    void bind_template_constants(int t) {
        std::vector<int32_t> ids;
        for (const C256& v : tmpls[(size_t)t].const_vals) ids.push_back(cid(v));
        if (tmpl_cids.size() <= (size_t)t) tmpl_cids.resize((size_t)t + 1);
        tmpl_cids[(size_t)t] = ids;
        Template& T = tmpls[(size_t)t];
        for (size_t i = 0; i < T.const_pos.size(); ++i) T.cid[(size_t)T.const_pos[i]] = ids[i];
    }
    // ns blocks of template t from stream index off; -> the r cells of block `which` are r_of(off, t, which)
    i64 blocks(i64 off, int t, std::vector<i64>&& a, std::vector<i64>&& b, std::vector<i64>&& s, i64 ns) {
        const Template& T = tmpls[(size_t)t];
        Part p;
        p.cell_off = off;
        p.lk_off = n_lk;
        p.tmpl = t;
        p.ns = ns;
        p.n_cells = ns * (i64)T.cells;
        p.n_lk = ns * (i64)T.lkpos.size();
        p.a = std::move(a); p.b = std::move(b); p.s = std::move(s);
        n_lk += p.n_lk;
        n_cells = off + p.n_cells;
        parts.push_back(std::move(p));
        return n_cells;
    }
};

int build_stream(int kind, unsigned Ln, unsigned W, unsigned lb, const u64* exp_g, const u64* exp_r, Stream& S) {
    const unsigned L = 2 * Ln;
    S.tmpls.reserve(2);                  // (references to the templates stay valid when the uniform circuit adds the second one)
    S.tmpls.push_back(block_template(L, W, lb));
    const Template& tm = S.tmpls[0];
    // ---- prefix: the four assign_integer, square, refresh, load_zero
    Walk w(0);
    const std::vector<i64> n_c = assign(w, Ln, W, lb), g_c = assign(w, Ln, W, lb), x_c = assign(w, Ln, W, lb), y_c = assign(w, Ln, W, lb);
    const std::vector<i64> prod = mul_cells(w, n_c, n_c, 2 * Ln - 1);
    std::vector<u8> inc(4 * Ln + 8);
    uint32_t n_inc = 0;
    PZCHK(pz_refresh_aux(W, Ln, Ln, inc.data(), (uint32_t)inc.size(), &n_inc));
    w.putc(c_small(0));
    std::vector<i64> cur(prod);
    cur.resize(n_inc, NONE);
    for (uint32_t i = 0; i < n_inc; ++i) {
        i64 limb = cur[i];
        for (unsigned j = 0; j <= inc[i]; ++j) {
            i64 qd, rd;
            div_mod(w, limb, W, &qd, &rd);
            if (j == 0) {
                cur[i] = rd;
            } else {
                const i64 g = w.put(cur[i + j]);
                w.gate(g);
                w.putc(c_small(1)); w.put(rd);
                cur[i + j] = w.put();
            }
            limb = qd;
        }
    }
    for (i64 c : cur) {
        const i64 holder = range_check(w, c, W, lb);
        S.fresh.push_back(c != NONE ? c : holder);
    }
    if (S.fresh.size() != L) return PZ_ERR_UNSUPPORTED;   // the refreshed n^2 has l + r limbs for every shape of this circuit
    const i64 zero = w.putc(c_small(0));
    auto ext_l = [&](const std::vector<i64>& limbs) {
        std::vector<i64> v(limbs);
        v.resize(L, zero);
        return v;
    };
    i64 off = S.flush(w);
    S.bind_template_constants(0);
    auto r_of = [&](i64 boff, const Template& T, i64 blk, size_t which) { return boff + blk * (i64)T.cells + T.r_cells[which]; };

    std::vector<i64> gm, rn;   // the cells holding g^m and r^n
    if (kind == 0 || kind == 2) {
        struct Chain { std::vector<i64> base; const u64* e; };
        Chain chains[2] = {{ext_l(g_c), exp_g}, {ext_l(y_c), exp_r}};
        int first_chain = 0;
        if (kind == 2) {
            // g^m over ALL Ln * W bits of m IN the circuit: assign_constant(1), load_zero, then per limb of m num_to_bits and per bit the
            // (mul_mod, select, square_mod) block -- the same shape for every message
            S.tmpls.push_back(uniform_template(L, W, lb));
            S.bind_template_constants(1);
            const Template& ut = S.tmpls[1];
            Walk wc(off);
            const i64 one = wc.putc(c_small(1)), z2 = wc.putc(c_small(0));
            off = S.flush(wc);
            std::vector<i64> acc_cells(L, z2), sq_cells = ext_l(g_c);
            acc_cells[0] = one;
            for (unsigned li = 0; li < Ln; ++li) {
                Walk wb(off);
                std::vector<i64> bit_cells{wb.put()};
                wb.gate(bit_cells[0]);
                i64 acc_cell = bit_cells[0];
                for (unsigned i = 1; i < W; ++i) {
                    bit_cells.push_back(wb.put());
                    wb.putc(c_pow2(i));
                    acc_cell = wb.put();
                    if (i < W - 1) wb.gate(acc_cell);
                }
                wb.pair(x_c[li], acc_cell);
                for (i64 bc : bit_cells) {
                    const i64 g_ = wb.putc(c_small(0));
                    wb.gate(g_);
                    wb.put(bc); wb.put(bc); wb.put(bc);
                }
                off = S.flush(wb);
                // the limb's W bits: blocks chained through acc (select outputs) and sq (square remainders)
                std::vector<i64> a((size_t)W * L), b((size_t)W * L), s(W);
                for (unsigned t = 0; t < W; ++t) {
                    for (unsigned j = 0; j < L; ++j) {
                        a[(size_t)t * L + j] = t == 0 ? acc_cells[j] : r_of(off, ut, t - 1, j);
                        b[(size_t)t * L + j] = t == 0 ? sq_cells[j] : r_of(off, ut, t - 1, L + j);
                    }
                    s[t] = bit_cells[t];
                }
                for (unsigned j = 0; j < L; ++j) {
                    acc_cells[j] = r_of(off, ut, W - 1, j);
                    sq_cells[j] = r_of(off, ut, W - 1, L + j);
                }
                off = S.blocks(off, 1, std::move(a), std::move(b), std::move(s), W);
            }
            S.n_steps_g = 2 * (size_t)Ln * W;
            gm = acc_cells;
            first_chain = 1;
        }
        for (int ci = first_chain; ci < 2; ++ci) {
            Walk wc(off);
            const i64 one = wc.putc(c_small(1)), z2 = wc.putc(c_small(0));
            off = S.flush(wc);
            // pow_mod_fixed_exp's schedule: per bit the squaring step (cur, cur); on a set bit then (acc, cur).  Which BLOCK produced
            // each operand is structure: block indices first, the operand cells follow from them (-1: the base, -2: the constant one)
            const u64* e = chains[ci].e;
            int top = (int)((Ln * W + 63) / 64) - 1;      // the exponent: ceil(limbs_n * limb_bits / 64) words
            while (top >= 0 && e[top] == 0) --top;
            const size_t bits = top < 0 ? 0 : (size_t)top * 64 + (64 - __builtin_clzll(e[top]));
            std::vector<i64> a_blk, b_blk;
            i64 sq_blk = -1, acc_blk = -2, t = 0;
            for (size_t bi = 0; bi < bits; ++bi) {
                a_blk.push_back(sq_blk); b_blk.push_back(sq_blk);
                const i64 cur_blk = sq_blk;
                sq_blk = t++;
                if ((e[bi / 64] >> (bi % 64)) & 1) {
                    a_blk.push_back(acc_blk); b_blk.push_back(cur_blk);
                    acc_blk = t++;
                }
            }
            const i64 ns = t;
            (ci == 0 ? S.n_steps_g : S.n_steps_r) = (size_t)ns;
            auto operand = [&](i64 blk, unsigned j) -> i64 {
                if (blk == -2) return j == 0 ? one : z2;
                if (blk == -1) return chains[ci].base[j];
                return r_of(off, tm, blk, j);
            };
            std::vector<i64> a((size_t)ns * L), b((size_t)ns * L);
            for (i64 q = 0; q < ns; ++q)
                for (unsigned j = 0; j < L; ++j) {
                    a[(size_t)q * L + j] = operand(a_blk[(size_t)q], j);
                    b[(size_t)q * L + j] = operand(b_blk[(size_t)q], j);
                }
            std::vector<i64> res(L);
            for (unsigned j = 0; j < L; ++j) res[j] = operand(acc_blk, j);
            if (ns) off = S.blocks(off, 0, std::move(a), std::move(b), {}, ns);
            (ci == 0 ? gm : rn) = res;
        }
    } else {
        gm = ext_l(x_c);
        rn = ext_l(y_c);
    }
    const i64 fin_off = off;
    off = S.blocks(off, 0, std::vector<i64>(gm), std::vector<i64>(rn), {}, 1);
    // ---- suffix: assign_integer(res), assert_equal_fresh
    Walk ws(off);
    const std::vector<i64> res_c = assign(ws, L, W, lb);
    ws.putc(c_small(0));
    i64 eq_cell = ws.putc(c_small(1));
    for (unsigned j = 0; j < L; ++j) {
        is_equal(ws, r_of(fin_off, tm, 0, j), res_c[j]);
        const i64 g = ws.putc(c_small(0));
        ws.gate(g);
        ws.put(eq_cell); ws.put();
        eq_cell = ws.put();
    }
    off = S.flush(ws);
    // assert_equal_fresh's result is constrained to the constant 1 (bench.rs:74)
    Part& last = S.parts.back();
    last.src[(size_t)(eq_cell - last.cell_off)] = -(1 + (i64)S.cid(c_small(1)));
    S.result_cell = eq_cell;
    return PZ_OK;
}

// gate mask of the stream at cell i (host; for the break points: only a few cells per column are inspected)
struct MaskAt {
    const Stream& S;
    u8 operator()(i64 i) const {
        size_t lo = 0, hi = S.parts.size();
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (S.parts[mid].cell_off <= i) lo = mid;
            else hi = mid;
        }
        const Part& p = S.parts[lo];
        if (p.tmpl < 0) return p.mask[(size_t)(i - p.cell_off)];
        const Template& T = S.tmpls[(size_t)p.tmpl];
        return T.mask[(size_t)((i - p.cell_off) % (i64)T.cells)];
    }
};
// halo2-lib's column break rule (pz_circuit_break_points, pz_witness.hip), on the mask accessor
int break_points(const MaskAt& mask, i64 n_cells, i64 max_rows, std::vector<i64>& starts) {
    starts.clear();
    i64 s = 0;
    for (;;) {
        starts.push_back(s);
        i64 end = s + max_rows - 1;
        for (i64 i = s + max_rows - 3; i < s + max_rows - 1 && i < n_cells; ++i)
            if (mask(i)) {
                if ((i >= 1 && mask(i - 1) && i - 1 > s) || (i >= 2 && mask(i - 2) && i - 2 > s)) return PZ_ERR_UNSUPPORTED;
                end = i;
                break;
            }
        if (end >= n_cells) break;
        s = end;
    }
    starts.push_back(n_cells);
    return PZ_OK;
}

// ---------------------------------------------------------------------------------------------- device side
struct DTmpl {
    int cells, lks, L;
    const int32_t* sol;
    const int8_t* kind;
    const int16_t* limb;
    const int32_t* cid;
    const u8* mask;
    const int32_t* lkpos;
};
struct DPart {
    i64 cell_off, lk_off, n_cells, n_lk;
    int tmpl;
    const i64* src;
    const u8* mask;
    const i64* lk;
    const i64 *a, *b, *s;
};
struct DStream {
    const DPart* parts;
    int n_parts;
    const DTmpl* tmpls;
    const i64* fresh;
    const i64* starts;      // n_used + 1
    int n_used;
    i64 NC, NL, NK, n, max_rows, A, Lk;
};
__device__ __forceinline__ int part_of_cell(const DStream& D, i64 c) {
    int lo = 0, hi = D.n_parts;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (D.parts[mid].cell_off <= c) lo = mid;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int part_of_lookup(const DStream& D, i64 t) {
    // parts without lookups share their lk_off with the next part: take the LAST part whose lk_off <= t, then step back over empty ones
    int lo = 0, hi = D.n_parts;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (D.parts[mid].lk_off <= t) lo = mid;
        else hi = mid;
    }
    while (lo > 0 && D.parts[lo].n_lk == 0) --lo;
    return lo;
}
__device__ __forceinline__ i64 cell_pos(const DStream& D, i64 c) {
    int lo = 0, hi = D.n_used;   // the column whose start is the last one <= c (a shared break cell: row 0 of the later column)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (D.starts[mid] <= c) lo = mid;
        else hi = mid;
    }
    return (i64)lo * D.n + (c - D.starts[lo]);
}
// node ids: [0, NC) advice cells | [NC, NC + NL) lookup-advice cells | NK constants | n_used - 1 copies of the break cells
__global__ __launch_bounds__(256) void k_struct_nodes(DStream D, i64 T, u32* __restrict__ src, u32* __restrict__ pos) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    i64 s, p;
    if (i < D.NC) {
        const DPart& P = D.parts[part_of_cell(D, i)];
        const i64 rel = i - P.cell_off;
        if (P.tmpl < 0) {
            s = P.src[rel];
        } else {
            const DTmpl& Tm = D.tmpls[P.tmpl];
            const i64 blk = rel / Tm.cells;
            const int l = (int)(rel - blk * Tm.cells);
            const int kd = Tm.kind[l], cid = Tm.cid[l];
            if (cid >= 0) s = -(1 + (i64)cid);
            else if (kd < 0) s = P.cell_off + blk * Tm.cells + Tm.sol[l];
            else if (kd == EXT_A) s = P.a[blk * Tm.L + Tm.limb[l]];
            else if (kd == EXT_B) s = P.b[blk * Tm.L + Tm.limb[l]];
            else if (kd == EXT_N) s = D.fresh[Tm.limb[l]];
            else s = P.s[blk];
        }
        if (s < 0) s = D.NC + D.NL - 1 - s;   // -(1 + id) -> constant node NC + NL + id
        p = cell_pos(D, i);
    } else if (i < D.NC + D.NL) {
        const i64 t = i - D.NC;
        const DPart& P = D.parts[part_of_lookup(D, t)];
        const i64 rel = t - P.lk_off;
        if (P.tmpl < 0) {
            s = P.lk[rel];
        } else {
            const DTmpl& Tm = D.tmpls[P.tmpl];
            const i64 blk = rel / Tm.lks;
            s = P.cell_off + blk * Tm.cells + Tm.lkpos[rel - blk * Tm.lks];
        }
        p = (D.A + t / D.max_rows) * D.n + t % D.max_rows;
    } else if (i < D.NC + D.NL + D.NK) {
        s = i;
        p = (D.A + D.Lk) * D.n + (i - D.NC - D.NL);
    } else {
        const i64 j = i - (D.NC + D.NL + D.NK) + 1;          // the cell column j starts with is also the last cell of column j - 1
        s = D.starts[j];
        p = (j - 1) * D.n + (D.starts[j] - D.starts[j - 1]);
    }
    src[i] = (u32)s;
    pos[i] = (u32)p;
}
__global__ __launch_bounds__(256) void k_struct_jump(const u32* __restrict__ in, u32* __restrict__ out, i64 T, unsigned* __restrict__ changed) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const u32 a = in[i], b = in[a];
    out[i] = b;
    if (a != b) *changed = 1;
}
__global__ __launch_bounds__(256) void k_struct_touch(const u32* __restrict__ root, i64 T, u8* __restrict__ touched) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    const u32 r = root[i];
    if (r != (u32)i) {
        touched[i] = 1;
        touched[r] = 1;
    }
}
__global__ __launch_bounds__(256) void k_struct_keys(const u32* __restrict__ members, i64 M, const u32* __restrict__ root, const u32* __restrict__ pos,
                                                     u64 span, u64* __restrict__ keys) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const u32 id = members[j];
    keys[j] = (u64)root[id] * span + pos[id];
}
__global__ __launch_bounds__(256) void k_struct_identity(u32* __restrict__ map_col, u32* __restrict__ map_row, i64 total, i64 n) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    map_col[i] = (u32)(i / n);
    map_row[i] = (u32)(i % n);
}
// every class becomes one cycle: a member maps to the next member of its class in (column, row) order, the last one to the first --
// found by a binary search for the class's first key (keys are sorted; only the LAST member of a class searches)
__global__ __launch_bounds__(256) void k_struct_cycles(const u64* __restrict__ keys, i64 M, u64 span, i64 n, u32* __restrict__ map_col,
                                                       u32* __restrict__ map_row) {
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const u64 kj = keys[j], cls = kj / span, base = cls * span, p = kj - base;
    u64 nxt;
    if (j + 1 < M && keys[j + 1] < base + span) {
        nxt = keys[j + 1] - base;
    } else {
        i64 lo = -1, hi = j;                 // the first index whose key is >= base lies in (lo, hi]
        while (hi - lo > 1) {
            const i64 mid = (lo + hi) >> 1;
            if (keys[mid] >= base) hi = mid;
            else lo = mid;
        }
        nxt = keys[hi] - base;
    }
    map_col[p] = (u32)(nxt / (u64)n);
    map_row[p] = (u32)(nxt % (u64)n);
}
__global__ __launch_bounds__(256) void k_struct_selectors(DStream D, const u32* __restrict__ pos, u8* __restrict__ sel) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D.NC) return;
    const DPart& P = D.parts[part_of_cell(D, c)];
    const i64 rel = c - P.cell_off;
    const u8 g = P.tmpl < 0 ? P.mask[rel] : D.tmpls[P.tmpl].mask[rel % D.tmpls[P.tmpl].cells];
    if (g) sel[pos[c]] = 1;   // (the gate of a shared break cell is enabled in the column it starts: pos is that column's row 0)
}

// device buffers of one call, freed on every path out
struct DevBufs {
    std::vector<void*> bufs;
    ~DevBufs() {
        for (void* d : bufs) (void)pz_hip_free(d);
    }
    template <class T> int get(pz_ctx* ctx, size_t count, T** out) {
        void* d = nullptr;
        HIPCHK(ctx, pz_hip_malloc(ctx, &d, (count ? count : 1) * sizeof(T)));   // (out of memory: the pz_dev_alloc block cache is released, once)
        bufs.push_back(d);
        *out = (T*)d;
        return PZ_OK;
    }
    template <class T> int up(pz_ctx* ctx, const std::vector<T>& v, const T** out) {
        T* d = nullptr;
        PZCHK(get(ctx, v.size(), &d));
        if (!v.empty()) HIPCHK(ctx, hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        *out = d;
        return PZ_OK;
    }
    void drop(void* d) {
        auto it = std::find(bufs.begin(), bufs.end(), d);
        if (it != bufs.end()) {
            (void)pz_hip_free(d);
            bufs.erase(it);
        }
    }
};
}   // namespace

struct pz_structure {
    pz_ctx* ctx = nullptr;
    u8* d_selectors = nullptr;
    u32 *d_map_col = nullptr, *d_map_row = nullptr;
    u64* d_starts = nullptr;
    std::vector<u64> constants;   // canonical integers, 4 words each
    std::vector<u64> starts;      // host copy (n_adv + 1)
    size_t n_adv = 0, n_used = 0, n_lk = 0, max_rows = 0, n_cells = 0, n_lookups = 0, n_steps_g = 0, n_steps_r = 0;
    uint32_t k = 0;
};

extern "C" int pz_structure_free(pz_structure* st) {
    if (!st) return PZ_OK;
    if (st->ctx) {
        std::lock_guard<std::recursive_mutex> lock(st->ctx->mu);
        (void)hipSetDevice(st->ctx->device);
        (void)hipStreamSynchronize(st->ctx->stream);
        for (void* d : {(void*)st->d_selectors, (void*)st->d_map_col, (void*)st->d_map_row, (void*)st->d_starts})
            if (d) (void)pz_hip_free(d);
    }
    delete st;
    return PZ_OK;
}

extern "C" int pz_structure_info(const pz_structure* st, size_t* n_adv, size_t* n_adv_filled, size_t* n_lk, size_t* max_rows, size_t* n_constants,
                                 size_t* n_cells, size_t* n_lookups, size_t* n_steps_g, size_t* n_steps_r) {
    if (!st) return PZ_ERR_INVALID;
    if (n_adv) *n_adv = st->n_adv;
    if (n_adv_filled) *n_adv_filled = st->n_used;
    if (n_lk) *n_lk = st->n_lk;
    if (max_rows) *max_rows = st->max_rows;
    if (n_constants) *n_constants = st->constants.size() / 4;
    if (n_cells) *n_cells = st->n_cells;
    if (n_lookups) *n_lookups = st->n_lookups;
    if (n_steps_g) *n_steps_g = st->n_steps_g;
    if (n_steps_r) *n_steps_r = st->n_steps_r;
    return PZ_OK;
}

extern "C" int pz_structure_arrays(const pz_structure* st, const uint8_t** d_selectors, const uint32_t** d_map_col, const uint32_t** d_map_row,
                                   const uint64_t** d_col_starts, const uint64_t** constants, const uint64_t** col_starts_host) {
    if (!st) return PZ_ERR_INVALID;
    if (d_selectors) *d_selectors = st->d_selectors;
    if (d_map_col) *d_map_col = st->d_map_col;
    if (d_map_row) *d_map_row = st->d_map_row;
    if (d_col_starts) *d_col_starts = st->d_starts;
    if (constants) *constants = st->constants.data();
    if (col_starts_host) *col_starts_host = st->starts.data();
    return PZ_OK;
}

extern "C" int pz_circuit_structure_dev(pz_ctx* ctx, int kind, uint32_t limbs_n, uint32_t limb_bits, uint32_t lookup_bits, uint32_t k,
                                        const uint64_t* exp_g, const uint64_t* exp_r, size_t minimum_rows, uint32_t blinding_factors,
                                        pz_structure** out) {
    if (!ctx || !out || kind < 0 || kind > 2 || limbs_n == 0 || limbs_n > 64) return PZ_ERR_INVALID;
    if (limb_bits < 16 || limb_bits > 90 || lookup_bits == 0 || lookup_bits >= k || k < 4 || k > 24 || lookup_bits >= limb_bits) return PZ_ERR_INVALID;
    if ((kind != 1 && !exp_r) || (kind == 0 && !exp_g)) return PZ_ERR_INVALID;
    const i64 n = (i64)1 << k;
    const i64 unusable = (i64)blinding_factors + 3;
    if (unusable + 8 > n || (i64)minimum_rows >= n) return PZ_ERR_INVALID;
    *out = nullptr;
    PZ_ENTER(ctx);
    Stream S;
    try {
        PZCHK(build_stream(kind, limbs_n, limb_bits, lookup_bits, exp_g, exp_r, S));
    } catch (const std::bad_alloc&) {
        return PZ_ERR_OOM;
    }
    // ---- the row budget (paillier_halo2_amd/layout.py RowBudget): columns are FILLED to max_rows = 2^k - (blinding_factors + 3), their
    // NUMBER is what calculate_params(Some(minimum_rows)) configures
    const i64 max_rows = n - unusable, count_rows = n - (i64)minimum_rows;
    const i64 NC = S.n_cells, NL = S.n_lk, NK = (i64)S.constants.size();
    std::vector<i64> starts;
    PZCHK(break_points(MaskAt{S}, NC, max_rows, starts));
    const i64 A_used = (i64)starts.size() - 1;
    const i64 A = std::max(A_used, (NC + count_rows - 1) / count_rows);
    const i64 Lk = std::max((NL + max_rows - 1) / max_rows, (NL + count_rows - 1) / count_rows);
    const i64 m = A + Lk + 1;
    if (NK > max_rows || Lk < 1) return PZ_ERR_UNSUPPORTED;
    const i64 T = NC + NL + NK + (A_used - 1), span = m * n;
    if (T >= ((i64)1 << 31) || span >= ((i64)1 << 31)) return PZ_ERR_UNSUPPORTED;   // node ids and flat positions are 32-bit (and hipCUB counts in int)
    std::unique_ptr<pz_structure> st(new (std::nothrow) pz_structure);
    if (!st) return PZ_ERR_OOM;
    st->k = k;
    st->n_adv = (size_t)A; st->n_used = (size_t)A_used; st->n_lk = (size_t)Lk; st->max_rows = (size_t)max_rows;
    st->n_cells = (size_t)NC; st->n_lookups = (size_t)NL; st->n_steps_g = S.n_steps_g; st->n_steps_r = S.n_steps_r;
    for (const C256& c : S.constants) st->constants.insert(st->constants.end(), c.w, c.w + 4);
    st->starts.assign(starts.begin(), starts.end());
    st->starts.resize((size_t)A + 1, (u64)NC);      // a configured column the cells do not reach starts and ends at the stream's end

    DevBufs tmp;
    // ---- tables
    std::vector<DTmpl> h_tm(S.tmpls.size());
    for (size_t t = 0; t < S.tmpls.size(); ++t) {
        const Template& Tm = S.tmpls[t];
        DTmpl& d = h_tm[t];
        d.cells = (int)Tm.cells; d.lks = (int)Tm.lkpos.size(); d.L = (int)(2 * limbs_n);
        PZCHK(tmp.up(ctx, Tm.sol, &d.sol)); PZCHK(tmp.up(ctx, Tm.kind, &d.kind)); PZCHK(tmp.up(ctx, Tm.limb, &d.limb));
        PZCHK(tmp.up(ctx, Tm.cid, &d.cid)); PZCHK(tmp.up(ctx, Tm.mask, &d.mask)); PZCHK(tmp.up(ctx, Tm.lkpos, &d.lkpos));
    }
    std::vector<DPart> h_parts(S.parts.size());
    for (size_t p = 0; p < S.parts.size(); ++p) {
        const Part& P = S.parts[p];
        DPart& d = h_parts[p];
        d.cell_off = P.cell_off; d.lk_off = P.lk_off; d.n_cells = P.n_cells; d.n_lk = P.n_lk; d.tmpl = P.tmpl;
        PZCHK(tmp.up(ctx, P.src, &d.src)); PZCHK(tmp.up(ctx, P.mask, &d.mask)); PZCHK(tmp.up(ctx, P.lk, &d.lk));
        PZCHK(tmp.up(ctx, P.a, &d.a)); PZCHK(tmp.up(ctx, P.b, &d.b)); PZCHK(tmp.up(ctx, P.s, &d.s));
    }
    DStream D;
    PZCHK(tmp.up(ctx, h_parts, &D.parts));
    PZCHK(tmp.up(ctx, h_tm, &D.tmpls));
    PZCHK(tmp.up(ctx, S.fresh, &D.fresh));
    PZCHK(tmp.up(ctx, starts, &D.starts));
    D.n_parts = (int)h_parts.size(); D.n_used = (int)A_used;
    D.NC = NC; D.NL = NL; D.NK = NK; D.n = n; D.max_rows = max_rows; D.A = A; D.Lk = Lk;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // the host vectors above go out of use only now

    // ---- outputs
    HIPCHK(ctx, pz_hip_malloc(ctx, (void**)&st->d_selectors, (size_t)(A * n)));
    st->ctx = ctx;                                      // from here pz_structure_free releases what was allocated
    struct Guard {                                      // (a failing step below frees the half-built structure)
        std::unique_ptr<pz_structure>& s;
        bool armed = true;
        ~Guard() {
            if (armed) pz_structure_free(s.release());
        }
    } guard{st};
    HIPCHK(ctx, pz_hip_malloc(ctx, (void**)&st->d_map_col, (size_t)span * 4));
    HIPCHK(ctx, pz_hip_malloc(ctx, (void**)&st->d_map_row, (size_t)span * 4));
    HIPCHK(ctx, pz_hip_malloc(ctx, (void**)&st->d_starts, ((size_t)A + 1) * 8));
    HIPCHK(ctx, hipMemcpyAsync(st->d_starts, st->starts.data(), ((size_t)A + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(st->d_selectors, 0, (size_t)(A * n), ctx->stream));

    const unsigned TB = 256;
    auto grid = [&](i64 count) { return dim3((unsigned)((count + TB - 1) / TB)); };
    // ---- nodes: what each copies, where it sits
    u32 *src = nullptr, *src2 = nullptr, *pos = nullptr;
    PZCHK(tmp.get(ctx, (size_t)T, &src)); PZCHK(tmp.get(ctx, (size_t)T, &src2)); PZCHK(tmp.get(ctx, (size_t)T, &pos));
    hipLaunchKernelGGL(k_struct_nodes, grid(T), dim3(TB), 0, ctx->stream, D, T, src, pos);
    hipLaunchKernelGGL(k_struct_selectors, grid(NC), dim3(TB), 0, ctx->stream, D, (const u32*)pos, st->d_selectors);
    HIPCHK(ctx, hipGetLastError());
    // ---- roots by pointer jumping (chains are a few links long)
    unsigned* d_changed = nullptr;
    PZCHK(tmp.get(ctx, 1, &d_changed));
    for (int it = 0; it < 64; ++it) {
        HIPCHK(ctx, hipMemsetAsync(d_changed, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_struct_jump, grid(T), dim3(TB), 0, ctx->stream, (const u32*)src, src2, T, d_changed);
        HIPCHK(ctx, hipGetLastError());
        unsigned changed = 0;
        HIPCHK(ctx, hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::swap(src, src2);
        if (!changed) break;
        if (it == 63) return PZ_ERR_INTERNAL;
    }
    const u32* root = src;
    // ---- members of non-trivial classes
    u8* touched = nullptr;
    PZCHK(tmp.get(ctx, (size_t)T, &touched));
    HIPCHK(ctx, hipMemsetAsync(touched, 0, (size_t)T, ctx->stream));
    hipLaunchKernelGGL(k_struct_touch, grid(T), dim3(TB), 0, ctx->stream, root, T, touched);
    HIPCHK(ctx, hipGetLastError());
    u32* members = src2;                                 // (the jump's second buffer is free now)
    i64* d_M = nullptr;
    PZCHK(tmp.get(ctx, 1, &d_M));
    {
        size_t tb = 0;
        hipcub::CountingInputIterator<u32> ids(0);
        HIPCHK(ctx, hipcub::DeviceSelect::Flagged(nullptr, tb, ids, touched, members, d_M, (int)T, ctx->stream));
        char* t_ = nullptr;
        PZCHK(tmp.get(ctx, tb, &t_));
        HIPCHK(ctx, hipcub::DeviceSelect::Flagged(t_, tb, ids, touched, members, d_M, (int)T, ctx->stream));
    }
    i64 M = 0;
    HIPCHK(ctx, hipMemcpyAsync(&M, d_M, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    tmp.drop(touched);
    hipLaunchKernelGGL(k_struct_identity, grid(span), dim3(TB), 0, ctx->stream, st->d_map_col, st->d_map_row, span, n);
    HIPCHK(ctx, hipGetLastError());
    if (M > 0) {
        // ---- sorted by (class, position): ONE radix sort of the combined key (class < T < 2^32, position < m n < 2^32)
        u64 *keys = nullptr, *keys2 = nullptr;
        PZCHK(tmp.get(ctx, (size_t)M, &keys));
        hipLaunchKernelGGL(k_struct_keys, grid(M), dim3(TB), 0, ctx->stream, (const u32*)members, M, root, (const u32*)pos, (u64)span, keys);
        HIPCHK(ctx, hipGetLastError());
        // the node arrays are done with once the keys exist: released BEFORE the sort's second buffer is allocated (peak 25 GB instead of 45
        // at config c5, where the generator runs beside a cached 119-GB key)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        tmp.drop(src);
        tmp.drop(src2);
        tmp.drop(pos);
        PZCHK(tmp.get(ctx, (size_t)M, &keys2));
        unsigned key_bits = 1;
        while (key_bits < 64 && (((u64)T * (u64)span) >> key_bits) != 0) ++key_bits;
        {
            size_t tb = 0;
            HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tb, (const u64*)keys, keys2, (int)M, 0, (int)key_bits, ctx->stream));
            char* t_ = nullptr;
            PZCHK(tmp.get(ctx, tb, &t_));
            HIPCHK(ctx, hipcub::DeviceRadixSort::SortKeys(t_, tb, (const u64*)keys, keys2, (int)M, 0, (int)key_bits, ctx->stream));
        }
        // ---- the cycles
        hipLaunchKernelGGL(k_struct_cycles, grid(M), dim3(TB), 0, ctx->stream, (const u64*)keys2, M, (u64)span, n, st->d_map_col, st->d_map_row);
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    guard.armed = false;
    *out = st.release();
    return PZ_OK;
}
