"""Static instruction census of the K2 kernels' ISA (hipcc -S, gfx950): multiplier instructions against everything else.
Usage: python profiles/probes/ntt_static_census.py  (writes to stdout; no GPU needed)"""
import collections, os, re, subprocess, tempfile

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "paillier_halo2_amd", "csrc", "pz_ntt.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "pz_ntt.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    txt = open(out).read()
kern, cur = {}, None
for line in txt.splitlines():
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        kern[cur] = []
    elif cur and line.strip() and not line.strip().startswith((";", ".")):
        kern[cur].append(line.strip())
    if line.strip().startswith("s_endpgm"):
        cur = None
print("# static instruction census of pz_ntt.hip's kernels (hipcc -O3 -S --offload-arch=gfx950); counts are per kernel BODY (loops are unrolled over the")
print("# NTT_LB loads and the radix-4 stage pairs are loop bodies executed logR/2 times: the ratios are what matters, not the totals)")
for k, body in kern.items():
    if "k_ntt" not in k:
        continue
    ops = collections.Counter(l.split()[0] for l in body)
    valu = sum(v for o, v in ops.items() if o.startswith("v_"))
    mult = ops["v_mad_u64_u32"] + ops["v_mul_lo_u32"]
    lds = sum(v for o, v in ops.items() if o.startswith("ds_"))
    vmem = sum(v for o, v in ops.items() if o.startswith(("global_", "buffer_")))
    salu = sum(v for o, v in ops.items() if o.startswith("s_") and o not in ("s_waitcnt", "s_nop", "s_barrier"))
    rest = sorted(((o, v) for o, v in ops.items() if o.startswith("v_") and o not in ("v_mad_u64_u32", "v_mul_lo_u32")), key=lambda kv: -kv[1])[:10]
    print("%s\n  %d instructions: VALU %d = %d v_mad_u64_u32 + %d v_mul_lo_u32 (%.1f %% multiplier) + %d other; LDS %d, VMEM %d, SALU %d, s_nop %d, s_waitcnt %d, s_barrier %d"
          % (k[:48], len(body), valu, ops["v_mad_u64_u32"], ops["v_mul_lo_u32"], 100.0 * mult / max(1, valu), valu - mult, lds, vmem, salu, ops["s_nop"], ops["s_waitcnt"], ops["s_barrier"]))
    print("  other VALU: " + ", ".join("%s %d" % kv for kv in rest))
