"""The bench line's contract (task brief, section 4): the newest committed `profiles/round*_bench_v*.json` is a line bench.py printed
on an MI355X; its keys, units and internal arithmetic are checked here so a change to bench.py that breaks the contract shows
up in the CPU suite."""
import glob
import json
import os

from conftest import ROOT


def _newest():
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
