"""paillier_halo2_amd -- MI355X-native (gfx950, hand-written HIP) hot path of the
Paillier-in-Halo2 prover: K1 G1 MSM, K2 Fr NTT, K3 big-integer modexp witness trace, K4 witness
cell expansion, behind the C ABI of include/pz.h (csrc/libpz_hip.so).  Python here is plumbing
(ctypes + torch device memory / torch.distributed); see DESIGN.md."""
from ._lib import PzError, SO_PATH, build, lib  # noqa: F401
from .engine import Bases, Engine  # noqa: F401
