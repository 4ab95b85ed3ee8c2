"""The bench line's contract (task brief, section 4): the newest committed `profiles/round*_bench_v*.json` is a line bench.py printed
on an MI355X; its keys, units and internal arithmetic are checked here so a change to bench.py that breaks the contract shows
up in the CPU suite."""
import glob
import json
import os

from conftest import ROOT


def _newest():
    """the default line of the newest round that committed one (profiles/roundN_bench_final.json; rounds 1 - 3 named them _vK)"""
    finals = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_bench_final.json")), key=lambda p: int(os.path.basename(p).split("_")[0][5:]))
    if finals:
        return json.load(open(finals[-1]))

    def key(p):
        b = os.path.basename(p)
        return (int(b.split("_")[0][5:]), int(b.split("_v")[1].split("_")[0].split(".")[0]))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_bench_v*.json")), key=key)
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_fields():
    d = _newest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"].replace("\u00d7", " x ").replace("  ", " ") or "GN iters/sec" in d["metric"]
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1


def test_bench_line_arithmetic():
    d = _newest()
    W, it = d["config"]["windows_per_gpu"], d["config"]["iterations_per_step"]
    # value = whole-job window-iterations / time of the timed steps
    assert abs(d["value"] - d["n_gpus"] * W * it / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    # achieved = algorithmic bytes per launch (per-window figure x windows of one launch) / average launch duration of the dominant kernel
    assert abs(r["achieved"] - r["algorithmic_bytes_per_window_iteration"] * W / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["kernel"] in d["kernels"] and abs(d["kernels"][r["kernel"]]["avg_ms"] - r["kernel_avg_ms"]) < 1e-12
    # the dominant kernel's measured HBM traffic is a per-launch byte count of plausible size
    assert r["traffic"] is None or 0.1 * r["algorithmic_bytes_per_window_iteration"] * W < r["traffic"] < 50 * r["algorithmic_bytes_per_window_iteration"] * W


def test_the_line_checks_what_it_timed():
    """parity_sample of the committed line looks at the first, a middle and the LAST windows of the timed batch and at every window's cost;
    cpu_baseline names its host and carries the compiled reference's Evaluate pass; the counters behind roofline.traffic belong to the
    kernels that ran."""
    d = _newest()
    W = d["config"]["windows_per_gpu"]
    ps = d["parity_sample"]
    idx = ps["window_indices"]
    assert idx[0] == 0 and idx[-1] == W - 1 and any(W // 4 < i < 3 * W // 4 for i in idx)
    assert max(ps["max_state_err"], ps["max_cost_rel"]) <= ps["tolerance"] == 1e-8
    aw = ps["all_windows"]
    assert aw["n"] == W and aw["ok"] and aw["outside_0.1x_10x_of_median"] == 0 and aw["states_finite"]
    c = d["cpu_baseline"]
    assert c["cpu_model"] and c["nproc"] >= c["cores"]
    re = c["reference_evaluate"]
    assert re["kind"] == "reference" and re["residual_blocks_per_pass"] > 3000 and 0.1 < re["ms_per_evaluation_pass"] < 100
    assert d["roofline"]["traffic_source"]["stale"] is False
    hi = d["host_inclusive"]
    one = hi.get("as_one_batch", hi)     # (lines from before the call pipelined itself carry the one batch only)
    assert abs(one["value"] - hi["windows"] * d["config"]["iterations_per_step"] / (one["ms"]["total"] * 1e-3)) < 1e-6 * one["value"]
    if one is not hi:
        assert abs(hi["value"] - hi["windows"] * d["config"]["iterations_per_step"] / (hi["ms"] * 1e-3)) < 1e-6 * hi["value"]
        assert hi["bitwise_equal_to_one_batch"] is True and hi["value"] >= one["value"]
    assert hi["value"] < d["value"]        # (the hand-over costs: it can never look faster than the resident line)


def test_sample_indices_cover_head_middle_and_tail():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    for W in (1, 2, 5, 9, 256, 32768):
        idx = bench.sample_indices(W)
        assert idx == sorted(set(idx)) and idx[0] == 0 and idx[-1] == W - 1 and all(0 <= i < W for i in idx) and len(idx) <= 9
    assert bench.sample_indices(32768) == [0, 1, 2, 16383, 16384, 16385, 32765, 32766, 32767]
