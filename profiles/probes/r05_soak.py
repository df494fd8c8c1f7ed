"""Round-5 soak (evidence, not a unit test): what changed this round under sustained and perturbed load.
  1. K3's two-workgroup protocol (roles by arrival, release / acquire hand-off): random batches of encrypts at 128 ... 3072-bit keys,
     WHILE a second context keeps the chip busy with transforms (different issue timing than the lone-kernel tests), every ciphertext
     against Python's pow and sampled steps by a*b == q*n^2 + r;
  2. K2 on constant pairs: inverse(forward(x)) == x on random columns at 2^10 ... 2^22, coset transforms included;
  3. the connected proof at the reference's bench shape: N proofs in a row with fresh hashing transcripts, EVERY one checked as the
     verifier would.
Usage: python profiles/probes/r05_soak.py [seconds per part, default 40]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import paillier_halo2_amd as pz
from oracle import cref, pyref as P
cref.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
eng = pz.Engine(0); eng.bind_torch_stream()
noise = pz.Engine(0)
sn = torch.cuda.Stream(); noise.set_stream(sn.cuda_stream)
rng = random.Random(0x50AC)
R = P.FR_R
F = lambda v: cref.fr_ints_to_mont([v % R])[0]

# ---- 1. K3 under load
k = 16; n = 1 << k
xn = torch.randint(0, 1 << 62, (32, n, 4), dtype=torch.int64, device="cuda"); xn[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
t_end, calls, encs = time.time() + budget, 0, 0
while time.time() < t_end:
    for _ in range(6):   # keep ~30 ms of transforms queued on the other context
        noise.ntt_dev(xn.data_ptr(), 32, 4 * n, F(P.fr_omega(k)), k, None, None)
    bits = rng.choice([128, 256, 1024, 2048, 3072])
    Ln = bits // 64
    batch = rng.choice([1, 2, 3, 7]) if bits >= 1024 else rng.choice([1, 5, 33, 200])
    ins = []
    for _ in range(batch):
        nn = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        m = rng.getrandbits(rng.choice([1, 17, bits])) % nn
        ins.append((nn, rng.randrange(1, nn), m, rng.randrange(1, nn)))
    pack = lambda j: np.stack([cref.int_to_limbs(t[j], Ln) for t in ins])
    c, steps, ng, nr = eng.paillier_encrypt(Ln, pack(0), pack(1), pack(2), pack(3))
    for i, (nn, g, m, r) in enumerate(ins):
        assert cref.limbs_to_int(c[i]) == P.paillier_enc_native(nn, g, m, r), ("K3 value", bits, batch, i)
        tot = int(ng[i]) + int(nr[i]) + 1
        for s_ in sorted({0, tot // 2, tot - 1}):
            a, b, q, rr = (cref.limbs_to_int(steps[i, s_, j]) for j in range(4))
            assert a * b == q * nn * nn + rr and rr < nn * nn, ("K3 step", bits, i, s_)
    calls += 1; encs += batch
torch.cuda.synchronize()
print("[soak] K3 under transform load: %d calls, %d encrypts, all equal to pow()" % (calls, encs), flush=True)

# ---- 2. K2 round trips
t_end, trips = time.time() + budget, 0
while time.time() < t_end:
    lg = rng.choice([10, 13, 15, 17, 18, 19, 20, 22])
    nc = 1 if lg >= 20 else rng.choice([1, 3, 8])
    m_ = 1 << lg
    x = torch.randint(0, 1 << 62, (nc, m_, 4), dtype=torch.int64, device="cuda"); x[:, :, 3] &= 0x0FFFFFFFFFFFFFFF
    eng.fr_convert_dev(x.data_ptr(), nc * m_, True)      # canonical Montgomery elements
    y = x.clone()
    w = P.fr_omega(lg)
    g = rng.choice([None, 7, P.FR_GENERATOR ** 3 % R])
    eng.ntt_dev(y.data_ptr(), nc, 4 * m_, F(w), lg, F(g) if g else None, None)
    eng.ntt_dev(y.data_ptr(), nc, 4 * m_, F(pow(w, -1, R)), lg, None, F(pow(m_, -1, R)))
    if g:
        eng.fr_distribute_powers_dev(y.data_ptr(), nc, 4 * m_, m_, F(pow(g, -1, R)))
    eng.sync()
    assert torch.equal(x, y), ("K2 round trip", lg, nc, g)
    trips += 1
print("[soak] K2 inverse(forward(x)) == x: %d round trips (2^10 ... 2^22, plain and coset)" % trips, flush=True)

# ---- 3. connected proofs, every one verified
import bench_connected
wl = bench_connected.ConnectedWorkload(eng, torch, 128, 14, 0x5042)
t_end, ok = time.time() + budget, 0
while time.time() < t_end:
    wl.run(1, timed=False)
    v = wl.verify(cref)
    assert v["verified"] is True, v
    ok += 1
print("[soak] connected proofs (128-bit, k = 14): %d of %d verified" % (ok, ok), flush=True)
wl.release()
print("[soak] done")
