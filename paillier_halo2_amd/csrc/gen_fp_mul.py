#!/usr/bin/env python3
"""Generates fp_mul_gen.cuh: the Montgomery product of fp.cuh as finely integrated product scanning
with ONE inline-asm statement per column (plus one for the m_k*p_0 term), so hipcc's one-wait-state
boundary pad after every asm statement is paid 24 times per product instead of 128.
Run: python gen_fp_mul.py > fp_mul_gen.cuh"""

# gfx940/gfx950 hazard: a VALU write of an SGPR -- VCC included -- needs 2 wait states before a VALU reads
# that SGPR (LLVM GCNHazardRecognizer "VALUWriteSGPRVALURead", hasVDecCoExecHazard targets; hipcc -O3 pads every
# v_add_co/v_addc_co pair it emits itself with `s_nop 1`).  hipcc does not pad inside an asm string, so the
# carries are software-pipelined: mad t writes its carry to one of three rotating SGPR pairs and the v_addc
# that folds it into the third accumulator word is issued two instructions later; a block with a single
# product has nothing to put in between and carries an explicit `s_nop 1`.
# WAIT=False reproduces round 1's back-to-back mad/addc pair (no wait states) under another name, for the
# issue-rate microbenchmark only (pz_ubench_fqmul_variant): what the two wait states cost.
MAD = 'v_mad_u64_u32 %0, %{c}, %{x}, %{y}, %0'
ADDC = 'v_addc_co_u32_e64 %1, vcc, 0, %1, %{c}'


WAIT = True


def block(pairs, hi_zero=False):
    """pairs: list of ((expr, constraint), (expr, constraint)); operands: %0 lo, %1 hi, %2..%4 carries.
    hi_zero: the third accumulator word is known to be zero on entry (start of a column): the first v_addc
    then adds into the constant 0 and `hi` is a pure output, which saves the v_mov that would clear it."""
    n = len(pairs)
    lines = []
    ops = []
    idx = 5
    first = [True]

    def addc(c):
        if hi_zero and first[0]:
            first[0] = False
            return 'v_addc_co_u32_e64 %1, vcc, 0, 0, %{c}'.format(c=c)
        return ADDC.format(c=c)

    for t, ((xa, ca), (xb, cb)) in enumerate(pairs):
        lines.append(MAD.format(c=2 + t % 3, x=idx, y=idx + 1))
        if t >= 2:
            lines.append(addc(2 + (t - 2) % 3))
        ops.append('"%s"(%s)' % (ca, xa))
        ops.append('"%s"(%s)' % (cb, xb))
        idx += 2
    if n == 1:
        # single product (m_k * p_0): carry through VCC, two wait states between the write and the carry-in read
        lines[0] = 'v_mad_u64_u32 %0, vcc, %5, %6, %0'
        if WAIT:
            lines.append('s_nop 1')
        lines.append('v_addc_co_u32_e64 %1, vcc, 0, 0, vcc' if hi_zero else 'v_addc_co_u32_e64 %1, vcc, 0, %1, vcc')
    elif n == 2:
        lines.append('s_nop 0')
        lines.append(addc(2))
        lines.append(addc(3))
    else:
        lines.append(addc(2 + (n - 2) % 3))
        lines.append(addc(2 + (n - 1) % 3))
    body = '\\n\\t'.join(lines)
    hi_c = '"=&v"(hi)' if hi_zero else '"+&v"(hi)'
    return ('    asm("%s"\n        : "+&v"(lo), %s, "=&s"(c0), "=&s"(c1), "=&s"(c2)\n        : %s\n        : "vcc");'
            % (body, hi_c, ', '.join(ops)))


def emit_mul(out, name):
    out.append("template <class T> __device__ __forceinline__ Fp<T> %s(const Fp<T>& a, const Fp<T>& b) {" % name)
    out.append("    u64 lo = 0;\n    u32 hi;\n    u32 m0, m1, m2, m3, m4, m5, m6, m7;\n    u64 c0, c1, c2;  // carry-out SGPR pairs\n    Fp<T> r;")
    for k in range(8):
        pairs = []
        for i in range(k + 1):
            pairs.append((("a.v[%d]" % i, "v"), ("b.v[%d]" % (k - i), "v")))
        for i in range(k):
            pairs.append((("m%d" % i, "v"), ("FieldParams<T>::P(%d)" % (k - i), "s")))
        out.append(block(pairs, hi_zero=True))
        out.append("    m%d = (u32)lo * FieldParams<T>::INV;" % k)
        out.append(block([(("m%d" % k, "v"), ("FieldParams<T>::P(0)", "s"))]))
        out.append("    lo = (lo >> 32) | ((u64)hi << 32);")
    for k in range(8, 16):
        pairs = []
        for i in range(k - 7, 8):
            pairs.append((("a.v[%d]" % i, "v"), ("b.v[%d]" % (k - i), "v")))
            pairs.append((("m%d" % i, "v"), ("FieldParams<T>::P(%d)" % (k - i), "s")))
        if pairs:
            out.append(block(pairs, hi_zero=True))
        else:
            out.append("    hi = 0;")
        out.append("    r.v[%d] = (u32)lo;" % (k - 8))
        out.append("    lo = (lo >> 32) | ((u64)hi << 32);")
    out.append("    // a, b < 2p and R = 2^256 > 4p => result < 1.76p: stays in the lazy range [0, 2p) without a final subtraction")
    out.append("    return r;\n}")


def emit_from_mont(out):
    # Montgomery reduction alone (a * R^-1 mod p == fp_mul(a, 1) without the 56 products by the zero limbs of 1): the
    # canonical form of an element, needed once per scalar by the MSM digit passes
    out.append("template <class T> __device__ __forceinline__ Fp<T> fp_from_mont(const Fp<T>& a) {")
    out.append("    u64 lo = 0;\n    u32 hi;\n    u32 m0, m1, m2, m3, m4, m5, m6, m7;\n    u64 c0, c1, c2;  // carry-out SGPR pairs\n    const u32 one = 1u;\n    Fp<T> r;")
    for k in range(8):
        pairs = [(("a.v[%d]" % k, "v"), ("one", "s"))]
        for i in range(k):
            pairs.append((("m%d" % i, "v"), ("FieldParams<T>::P(%d)" % (k - i), "s")))
        out.append(block(pairs, hi_zero=True))
        out.append("    m%d = (u32)lo * FieldParams<T>::INV;" % k)
        out.append(block([(("m%d" % k, "v"), ("FieldParams<T>::P(0)", "s"))]))
        out.append("    lo = (lo >> 32) | ((u64)hi << 32);")
    for k in range(8, 16):
        pairs = []
        for i in range(k - 7, 8):
            pairs.append((("m%d" % i, "v"), ("FieldParams<T>::P(%d)" % (k - i), "s")))
        if pairs:
            out.append(block(pairs, hi_zero=True))
        else:
            out.append("    hi = 0;")
        out.append("    r.v[%d] = (u32)lo;" % (k - 8))
        out.append("    lo = (lo >> 32) | ((u64)hi << 32);")
    out.append("    fp_reduce_once(r);  // canonical: the digit passes read the integer value")
    out.append("    return r;\n}")


out = ["// GENERATED by gen_fp_mul.py -- do not edit.  Included by fp.cuh."]
emit_mul(out, "fp_mul")
out.append("")
emit_from_mont(out)
out.append("")
out.append("#ifdef PZ_FP_MUL_VARIANTS  // timing-only variant for pz_ubench_fqmul_variant (pz_core.hip); never used for results")
WAIT = False
emit_mul(out, "fp_mul_nowait")
out.append("#endif")
print("\n".join(out))
