"""Does a pinned H2D copy on one stream overlap kernels on another?  46 ms of pz_msm_g1_dev beside a 1 GiB upload."""
import sys
import time

import torch

sys.path.insert(0, ".")
import paillier_halo2_amd as pz
from paillier_halo2_amd import consts

eng = pz.Engine(0)
eng.bind_torch_stream()
k, n = 17, 1 << 17
gen = torch.Generator(device="cuda")
gen.manual_seed(1)
x = torch.randint(-(1 << 63), (1 << 63) - 1, (256 * n, 4), dtype=torch.int64, device="cuda", generator=gen)
x[:, 3] &= 0x0FFFFFFFFFFFFFFF
d_b = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
eng.srs_setup_g1_dev(k, consts.fr_mont_limbs(123456789), consts.fr_mont_limbs(consts.fr_omega(k)), 0, d_b.data_ptr())
eng.sync()
bases = eng.load_bases_dev(d_b.data_ptr(), n)
out = torch.zeros((256, 12), dtype=torch.int64, device="cuda")
h = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
h.view(torch.int64).view(-1, 4)[: 256 * n].copy_(x.cpu())   # the same uniform scalars on the host
d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
s2 = torch.cuda.Stream()


def msm():
    eng.msm_dev(bases, x.data_ptr(), 256, n, 4 * n, out.data_ptr())


def copy(pieces):
    with torch.cuda.stream(s2):
        if pieces == 1:
            d.copy_(h, non_blocking=True)
        else:
            sz = (1 << 30) // pieces
            for i in range(pieces):
                d[i * sz:(i + 1) * sz].copy_(h[i * sz:(i + 1) * sz], non_blocking=True)


def t(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


print("msm alone %.1f ms   copy alone %.1f ms (1 piece) %.1f ms (256 pieces)" % (t(msm), t(lambda: copy(1)), t(lambda: copy(256))))
print("msm then copy enqueued: %.1f ms (1 piece)  %.1f ms (256 pieces)" % (t(lambda: (msm(), copy(1))), t(lambda: (msm(), copy(256)))))
print("copy then msm enqueued: %.1f ms (1 piece)  %.1f ms (256 pieces)" % (t(lambda: (copy(1), msm())), t(lambda: (copy(256), msm()))))

# the pipeline of pz_msm_g1_batch rebuilt from Python: uploads on s2, msm_dev per group on the engine's stream
stage = [torch.empty((256 * n, 4), dtype=torch.int64, device="cuda") for _ in range(2)]
hcols = h.view(torch.int64).view(-1, 4)[: 256 * n]
main = torch.cuda.current_stream()
groups = [(0, 32), (32, 64), (96, 128), (224, 32)]


def pipe():
    done = [None, None]
    for g, (c0, nc) in enumerate(groups):
        b = g & 1
        with torch.cuda.stream(s2):
            if done[b] is not None:
                s2.wait_event(done[b])
            stage[b][: nc * n].copy_(hcols[c0 * n:(c0 + nc) * n], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(s2)
        main.wait_event(ev)
        eng.msm_dev(bases, stage[b].data_ptr(), nc, n, 4 * n, out[c0:].data_ptr())
        done[b] = torch.cuda.Event(); done[b].record(main)


print("python pipeline (4 groups, uploads beside kernels): %.1f ms" % t(pipe))
cols = [hcols[j * n:(j + 1) * n].numpy().view("uint64") for j in range(256)]
print("pz_msm_g1_batch, 256 pinned columns: %.1f ms" % t(lambda: eng.msm_batch(bases, cols)))

# the same call from pageable host memory (what a prover that does not page-lock its columns hands over)
import numpy as np
pcols = [np.array(c, copy=True) for c in cols]
print("pz_msm_g1_batch, 256 pageable columns: %.1f ms" % t(lambda: eng.msm_batch(bases, pcols)))

w_inv = consts.fr_mont_limbs(pow(consts.fr_omega(k), -1, consts.FR_R))
print("pz_ntt_fr_batch 2^17, 256 pinned columns: %.1f ms" % t(lambda: eng.ntt_batch_inplace(cols, w_inv, k)))
print("pz_ntt_fr_batch 2^17, 256 pageable columns: %.1f ms" % t(lambda: eng.ntt_batch_inplace(pcols, w_inv, k)))
