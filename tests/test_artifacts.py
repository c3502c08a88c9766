"""CPU-only: the committed measurement artefacts keep the shape the bench contract asks for."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    b = json.load(open(os.path.join(ROOT, "profiles", "r01_final_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["unit"] == "verifies/s" and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert "workload" in b["config"] and "model" not in b["config"]
    r = b["roofline"]
    assert r["bound"] == "valu" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert b["value"] >= 1.0e6                       # BASELINE.json target: >= 1 M verifies/s on one MI355X


def test_rocprof_summary_agrees_with_the_bench_line():
    """The dominant kernel's rocprofv3 average and the HIP-event figure behind roofline.achieved are the same run
    class: they must agree within a few percent (both are in profiles/)."""
    b = json.load(open(os.path.join(ROOT, "profiles", "r01_final_bench.json")))
    rows = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_final_kernel_stats.csv")))}
    avg_ms = float(rows["k_miller_verify"]["AverageNs"]) / 1e6
    mads = (8612 + 3784) * 136 * 262144
    assert abs(mads / (avg_ms * 1e-3) / 1e12 / b["roofline"]["achieved"] - 1) < 0.05
    t = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
    assert t["kernels"]["k_miller_verify"]["hbm_bytes_per_launch"] == b["roofline"]["traffic"]
