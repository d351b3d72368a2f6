"""The committed bench line (the newest profiles/rNN_bench.json, written by `python bench.py` on an MI355X) keeps the driver's contract: metric / unit /
value / timing fields, the `roofline` object of the dominant kernel and the `cpu_baseline` object — and its numbers are mutually consistent.
Runs on CPU: it reads the committed evidence, not the GPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    import glob
    import re
    cands = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json"))]
    path = max(cands, key=lambda p: int(re.search(r"r(\d+)_bench", p).group(1)))
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_bench_line_has_the_contract_fields():
    d = _line()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["metric"].split(",")[0] in base["metric"]          # BASELINE.json's metric, without its "1/2/4/8 GPU" suffix
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("cfg2") and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in r, k
    assert (r["bound"] in ("hbm", "mfma") or r["bound"].startswith("mfma-")) and r["unit"] in ("GB/s", "TFLOP/s")
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1


def test_bench_line_is_self_consistent():
    d = _line()
    r = d["roofline"]
    # value = interior points x steps / elapsed; ms_per_step = elapsed / steps
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / d["config"]["interior_points"] - 1.0) < 1e-6
    # frac = achieved / peak on the pipe that executes the hidden-layer products
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the dominant kernel cannot take longer than the step it is part of
    assert 0 < r["kernel_ms"] <= d["ms_per_step"]
    if "frac_fp32_equiv" in r:                   # r04 format: priced on the executing pipe, frac <= 1 by construction
        assert 0 < r["frac"] < 1.0
        assert abs(r["frac_fp32_equiv"] - r["executed_flops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12 / 157.3) < 1e-6
        if r["bound"].startswith("mfma-bf16"):
            assert r["peak"] == 2500.0
            assert abs(r["achieved"] - r["executed_bf16_mfma_flops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
            # six bf16 MFMAs per fp32 product block: the executed bf16 flops are close to 6 x the hidden-layer share of the fp32-equivalent flops
            assert 4.0 < r["executed_bf16_mfma_flops_per_launch"] / r["executed_flops_per_launch"] < 6.5
        else:
            assert r["peak"] == 157.3
        assert d["value_incl_theta_h2d"] is None or 0 < d["value_incl_theta_h2d"] <= d["value"] * 1.02
        assert len(d["cpu_baseline"]["thread_counts_tried"]) >= 1
    else:
        assert abs(r["achieved"] - r["executed_flops_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
        if r.get("gemm"):
            assert 0 < r["frac_mixed_pipes"] < 1.0 and r["mixed_pipe_peak"] >= r["peak"]
        else:
            assert 0 < r["frac"] < 1.0
    # measured HBM-side traffic (PMC) is at least the algorithmic bytes
    if r["traffic"] is not None:
        assert r["traffic"] >= r["algorithmic_bytes_per_launch"]
    # the five term losses of the run are the full-size golden's (float64 oracle, same theta and sets) to fp32 accuracy
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_full.npz"))
    ref = np.asarray(g["losses_stencil"], dtype=float)
    got = np.asarray(d["loss_terms"], dtype=float)
    assert got.shape == ref.shape and np.max(np.abs(got - ref) / np.abs(ref)) < 1e-5
