#!/usr/bin/env python3
"""Generates fp29_gen.cuh: the Montgomery product / square of fp29.cuh (9 limbs x 29 bits, R' = 2^261).

Product scanning with ONE 64-bit column accumulator.  In this radix no v_mad_u64_u32 can overflow its 64-bit
addend (fp29.cuh: limb bounds), so a column is a plain chain of mads -- no carry-out is ever consumed, no v_addc,
no wait states.  hipcc left alone re-associates the sums into ~17 parallel accumulators joined by
v_lshl_add_u64 (a ~9-cycle instruction on gfx950) and a v_mov per join; the mads of a column are therefore
emitted as inline asm (<= 30 operands per statement: a column is split into its a*b part and its m*p part), the
m_k computation, the masks and the 29-bit shifts stay C (one v_mul_lo_u32, v_and_b32, v_lshrrev_b64 each).
Run: python gen_fp29.py > fp29_gen.cuh"""

N = 9


def mads(pairs):
    """pairs: list of (exprA, consA, exprB, consB) -> one asm statement accumulating into acc"""
    lines, ops = [], []
    idx = 1
    for (xa, ca, xb, cb) in pairs:
        lines.append('v_mad_u64_u32 %%0, vcc, %%%d, %%%d, %%0' % (idx, idx + 1))
        ops.append('"%s"(%s)' % (ca, xa))
        ops.append('"%s"(%s)' % (cb, xb))
        idx += 2
    return '    asm("%s"\n        : "+v"(acc)\n        : %s\n        : "vcc");' % ('\\n\\t'.join(lines), ', '.join(ops))


def emit_mul(out, name, square, b_scalar=False):
    """b_scalar: b is a wave-uniform constant given as nine SGPR-resident limbs (const u32 (&b)[9], e.g. a kernel argument): every
    a_i * b_j mad reads ONE scalar operand, and the nine VGPRs a per-lane copy of the constant would occupy are free"""
    if square:
        out.append("template <class T> __device__ __forceinline__ F29<T> %s(const F29<T>& a) {" % name)
        out.append("    u32 d[9];   // doubled limbs for the cross terms\n#pragma unroll\n    for (int i = 0; i < 9; ++i) d[i] = a.v[i] << 1;")
    elif b_scalar:
        out.append("template <class T> __device__ __forceinline__ F29<T> %s(const F29<T>& a, const u32 (&b)[9]) {" % name)
    else:
        out.append("template <class T> __device__ __forceinline__ F29<T> %s(const F29<T>& a, const F29<T>& b) {" % name)
    out.append("    u64 acc = 0;\n    u32 m0, m1, m2, m3, m4, m5, m6, m7, m8;\n    F29<T> r;")

    def ab_pairs(k):
        ps = []
        if not square:
            for i in range(max(0, k - (N - 1)), min(k, N - 1) + 1):
                ps.append(("a.v[%d]" % i, "v", ("b[%d]" if b_scalar else "b.v[%d]") % (k - i), "s" if b_scalar else "v"))
        else:
            lo = max(0, k - (N - 1))
            for i in range(lo, (k + 1) // 2):      # i < k - i
                ps.append(("d[%d]" % i, "v", "a.v[%d]" % (k - i), "v"))
            if k % 2 == 0:
                ps.append(("a.v[%d]" % (k // 2), "v", "a.v[%d]" % (k // 2), "v"))
        return ps

    for k in range(N):
        out.append(mads(ab_pairs(k)))
        mp = [("m%d" % i, "v", "P29<T>::P(%d)" % (k - i), "s") for i in range(k)]
        if mp:
            out.append(mads(mp))
        out.append("    m%d = ((u32)acc * P29<T>::INV) & F29_MASK;" % k)
        out.append(mads([("m%d" % k, "v", "P29<T>::P(0)", "s")]))
        out.append("    acc >>= 29;")
    for k in range(N, 2 * N - 1):
        out.append(mads(ab_pairs(k)))
        out.append(mads([("m%d" % i, "v", "P29<T>::P(%d)" % (k - i), "s") for i in range(k - (N - 1), N)]))
        out.append("    r.v[%d] = (u32)acc & F29_MASK;" % (k - N))
        out.append("    acc >>= 29;")
    out.append("    r.v[8] = (u32)acc;   // < 2^29: the result is below 2^261")
    out.append("    return r;\n}")


def emit_mul2(out, name, b_scalar=False):
    """(a*b + c*d) / R' with ONE reduction: both products' columns go into the same accumulator.  Legal when
    9*(max a limb * max b limb + max c limb * max d limb) + 2^59.8 < 2^64; result tight, value < (va*vb + vc*vd)/(169 p) + p.
    b_scalar: b is a wave-uniform constant in SGPRs (see emit_mul)"""
    if b_scalar:
        out.append("template <class T> __device__ __forceinline__ F29<T> %s(const F29<T>& a, const u32 (&b)[9], const F29<T>& c, const F29<T>& d) {" % name)
    else:
        out.append("template <class T> __device__ __forceinline__ F29<T> %s(const F29<T>& a, const F29<T>& b, const F29<T>& c, const F29<T>& d) {" % name)
    out.append("    u64 acc = 0;\n    u32 m0, m1, m2, m3, m4, m5, m6, m7, m8;\n    F29<T> r;")

    def pairs(x, y, k):
        if b_scalar and y == "b":
            return [("%s.v[%d]" % (x, i), "v", "b[%d]" % (k - i), "s") for i in range(max(0, k - (N - 1)), min(k, N - 1) + 1)]
        return [("%s.v[%d]" % (x, i), "v", "%s.v[%d]" % (y, k - i), "v") for i in range(max(0, k - (N - 1)), min(k, N - 1) + 1)]

    for k in range(N):
        out.append(mads(pairs("a", "b", k)))
        out.append(mads(pairs("c", "d", k)))
        mp = [("m%d" % i, "v", "P29<T>::P(%d)" % (k - i), "s") for i in range(k)]
        if mp:
            out.append(mads(mp))
        out.append("    m%d = ((u32)acc * P29<T>::INV) & F29_MASK;" % k)
        out.append(mads([("m%d" % k, "v", "P29<T>::P(0)", "s")]))
        out.append("    acc >>= 29;")
    for k in range(N, 2 * N - 1):
        out.append(mads(pairs("a", "b", k)))
        out.append(mads(pairs("c", "d", k)))
        out.append(mads([("m%d" % i, "v", "P29<T>::P(%d)" % (k - i), "s") for i in range(k - (N - 1), N)]))
        out.append("    r.v[%d] = (u32)acc & F29_MASK;" % (k - N))
        out.append("    acc >>= 29;")
    out.append("    r.v[8] = (u32)acc;   // < 2^29: (va*vb + vc*vd) / 2^261 + p stays below 2^261 for the documented operand values")
    out.append("    return r;\n}")


def emit_dot(out, name, terms):
    """sum_{t < terms} a[t] * b[t] / R' with ONE reduction (dot products: evaluation at a point, linear combinations).  Legal when
    9 * sum_t (max a[t] limb * max b[t] limb) + 2^59.8 < 2^64 -- e.g. four tight x tight terms; result tight,
    value < sum(va * vb) / (169 p) + p"""
    out.append("template <class T> __device__ __forceinline__ F29<T> %s(const F29<T> (&a)[%d], const F29<T> (&b)[%d]) {" % (name, terms, terms))
    out.append("    u64 acc = 0;\n    u32 m0, m1, m2, m3, m4, m5, m6, m7, m8;\n    F29<T> r;")

    def pairs(t, k):
        return [("a[%d].v[%d]" % (t, i), "v", "b[%d].v[%d]" % (t, k - i), "v") for i in range(max(0, k - (N - 1)), min(k, N - 1) + 1)]

    for k in range(N):
        for t in range(terms):
            out.append(mads(pairs(t, k)))
        mp = [("m%d" % i, "v", "P29<T>::P(%d)" % (k - i), "s") for i in range(k)]
        if mp:
            out.append(mads(mp))
        out.append("    m%d = ((u32)acc * P29<T>::INV) & F29_MASK;" % k)
        out.append(mads([("m%d" % k, "v", "P29<T>::P(0)", "s")]))
        out.append("    acc >>= 29;")
    for k in range(N, 2 * N - 1):
        for t in range(terms):
            out.append(mads(pairs(t, k)))
        out.append(mads([("m%d" % i, "v", "P29<T>::P(%d)" % (k - i), "s") for i in range(k - (N - 1), N)]))
        out.append("    r.v[%d] = (u32)acc & F29_MASK;" % (k - N))
        out.append("    acc >>= 29;")
    out.append("    r.v[8] = (u32)acc;")
    out.append("    return r;\n}")


def emit_mulc(out, name):
    """Product by a PRECOMPUTED CONSTANT w (twiddles, challenges), Barrett / Shoup style: with wq = floor(w * 2^261 / p) known,
        q = floor(a * wq / 2^261)  (only the columns 7 .. 16 of the product: 53 mads; the dropped columns move q by less than one)
        r = a * w - q * p  =  (a * w + q * pbar) mod 2^261,  pbar = 2^261 - p   (low halves only: 45 + 45 mads)
    143 multiplier instructions instead of f29_mul's 171 + 9 (a Montgomery product needs the FULL a * b and the FULL m * p).  No
    Montgomery factor: r = a * w mod p in a's own domain (w is the plain constant).  Bounds: a's limbs below 2^31 (value < 2^263), w
    and wq tight; q is short of the true quotient by at most 4 + 1, so r < 6p, limbs tight (masked)."""
    out.append("template <class T> __device__ __forceinline__ F29<T> %s(const F29<T>& a, const F29<T>& w, const F29<T>& wq) {" % name)
    out.append("    u64 acc = 0;\n    u32 q0, q1, q2, q3, q4, q5, q6, q7, q8;\n    F29<T> r;")
    for k in range(7, 2 * N - 1):
        out.append(mads([("a.v[%d]" % i, "v", "wq.v[%d]" % (k - i), "v") for i in range(max(0, k - (N - 1)), min(k, N - 1) + 1)]))
        if k >= N:
            out.append("    q%d = (u32)acc & F29_MASK;" % (k - N))
        out.append("    acc >>= 29;")
    out.append("    q8 = (u32)acc;")
    out.append("    acc = 0;")
    for k in range(N):
        out.append(mads([("a.v[%d]" % i, "v", "w.v[%d]" % (k - i), "v") for i in range(k + 1)]))
        out.append(mads([("q%d" % i, "v", "f29_pbar_limb<T>(%d)" % (k - i), "s") for i in range(k + 1)]))
        out.append("    r.v[%d] = (u32)acc & F29_MASK;" % k)
        if k < N - 1:
            out.append("    acc >>= 29;")
    out.append("    return r;\n}")


out = ["// GENERATED by gen_fp29.py -- do not edit.  Included by fp29.cuh."]
emit_mul(out, "f29_mul", False)
out.append("")
emit_mul(out, "f29_sqr", True)
out.append("")
emit_mul2(out, "f29_mul2")
out.append("")
emit_dot(out, "f29_dot4", 4)
out.append("")
emit_mul(out, "f29_mul_s", False, b_scalar=True)
out.append("")
emit_mul2(out, "f29_mul2_s", b_scalar=True)
out.append("")
emit_mulc(out, "f29_mulc")
print("\n".join(out))
