"""The bench line's contract, checked on the line the round committed (profiles/r04_bench_default.json = stdout of `python bench.py` on an MI355X):
every field the driver and the judge read is there, and the numbers are consistent with each other."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields_and_consistent_arithmetic():
    j = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert base["metric"].startswith(j["metric"]) and j["unit"] == "audio-seconds/sec"  # BASELINE.json's metric names the train-step rate first
    assert j["n_gpus"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "bf16" and "synthetic" in j["data"] and "workload" in j["config"] and "model" not in j["config"]
    # 256 clips of 30 s per step
    assert abs(j["value"] - 256 * 30.0 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic flops of one launch / its average duration
    assert abs(r["achieved"] - r["alg_flops_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) / r["achieved"] < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    # the dominant kernel's launches fit into the step, and the step's rate is below the peak
    assert r["launches_per_step"] * r["avg_launch_us"] * 1e-3 < j["ms_per_step"]
    assert 0.0 < j["step_frac_of_mfma_peak"] < 1.0 and 0.0 < j["executed_over_algorithmic_flops"] <= 1.0
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == j["unit"] and c["cores"] >= 1 and c["value"] > 0
    # the line's own parity block: the timed micro-batch through the span step equals the plain step
    sp = j["parity"]["span_step_vs_plain_step"]
    assert sp["loss_span"] == sp["loss_full"] and sp["grad_rel_l2"] < 1e-4
