"""bench.py's headline line end to end, at a small shape (256-bit n, k = 13) so that it runs in seconds under `pytest -m gpu`: the
contract's fields, the connected proof verified by the checker leg, `roofline` / `cpu_baseline` present, the extras' legs; and the
same loop through torch.distributed.run with the RCCL (nccl) process group of one rank -- the path the driver's N = 2, 4, 8 runs take."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (extra, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]           # ONE JSON line on stdout
    return json.loads(lines[0])


SMALL = ["--enc-bits", "256", "--k", "13", "--steps", "3", "--warmup", "1"]


def test_headline_line_small_shape():
    d = _bench(SMALL + ["--headline-only"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "proofs/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    assert "NOT the headline configuration" in d["metric"]                      # only 2048-bit / k = 17 carries BASELINE.json's metric name
    assert d["config"]["scope"].startswith("one connected proof") and d["config"]["minimum_rows"] == 20 and d["config"]["max_rows"] == (1 << 13) - 9
    assert d["verified"] is True and d["verification"]["shplonk_identity_on_the_proofs_commitments"] is True and d["comparable"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "k_msm_accumulate" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["launches"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["connected_tail"]["field_mults_per_s_all_threads"] > 0
    assert set(d["phases_ms_per_proof"]) >= {"advice_commit", "products_commit", "quotient", "evaluations", "multiopen"}
    assert d["memory_gb"]["proving_key_streamed"] is False and d["counts"]["minimum_rows"] == 20


def test_headline_line_streamed_key_and_minimum_rows_9():
    env_key = "PZ_CONNECTED_STREAMED_KEY"
    old = os.environ.get(env_key)
    os.environ[env_key] = "0"
    try:
        d = _bench(SMALL + ["--headline-only", "--minimum-rows", "9", "--no-cpu-baseline", "--no-verify"])
    finally:
        if old is None:
            del os.environ[env_key]
        else:
            os.environ[env_key] = old
    assert d["config"]["proving_key_streamed"] is True and d["memory_gb"]["of_which_extended_forms"] == 0.0
    assert d["config"]["minimum_rows"] == 9 and d["verified"] is None and d["value"] > 0
    assert d["config"]["env_switches"].get(env_key) == "0"                      # every PZ_* switch in effect is echoed


def test_headline_line_through_the_rccl_group_of_one():
    d = _bench(SMALL + ["--headline-only", "--no-cpu-baseline", "--gpus", "1", "--force-dist"])
    assert d["backend"] == "nccl" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert d["value"] > 0 and d["config"]["scope"].startswith("one connected proof")


def test_default_line_legs_small_shape():
    """the default run's other legs are present for the headline configuration only; at a small custom shape the line still carries the
    compiled prover's figure (a child process over the C ABI) beside the headline"""
    d = _bench(SMALL + ["--no-cpu-baseline"])
    assert "compiled_prover" in d and d["compiled_prover"]["quotient_degree_ok"] is True and d["compiled_prover"]["ms_per_step"] > 0
    assert d["compiled_stepper"]["quotient_degree_ok"] is True and "stepper" in d["compiled_stepper"]["via"]
    assert "hot_path_only" not in d and "fresh_message_cpp" not in d


def test_c3_add_circuit_is_a_connected_line_too():
    """BASELINE config c3 (PaillierChip::add, /root/reference/src/paillier.rs:62-85 through bench.rs:77-117) as a connected proof: one key, distinct
    ciphertext pairs per proof, verified -- here at 256-bit / k = 12; `--hot-path-headline` keeps rounds 1-5's hot-path line"""
    d = _bench(["--workload", "c3", "--enc-bits", "256", "--k", "12", "--steps", "3", "--warmup", "1", "--headline-only"])
    assert d["verified"] is True and d["config"]["scope"].startswith("one connected proof") and "add" in d["metric"]
    assert d["config"]["mul_mod_steps"] == 1 and d["value"] > 0
    h = _bench(["--workload", "c3", "--enc-bits", "256", "--k", "12", "--steps", "3", "--warmup", "1", "--hot-path-headline", "--no-cpu-baseline", "--no-dropin",
                "--no-tail"])
    assert h["config"]["scope"].startswith("hot path only") and h["verified"] is True
