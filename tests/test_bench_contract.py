"""CPU-only: bench.py's launch contract (the parts that run without a GPU) and the shape of the bench line and
rocprof summary committed under profiles/ (fields and internal consistency only -- no performance numbers are asserted)."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")


def test_bench_rejects_world_size_mismatch():
    """WORLD_SIZE set by a launcher must equal --gpus: the check runs before anything touches the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in CLEAN}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (out.stderr + out.stdout)


def test_bench_parent_spawns_n_ranks_and_relays_their_exit_code():
    """Plain `python bench.py --gpus 2` with WORLD_SIZE unset starts two ranks itself (torch.distributed.run).  Here there is
    no GPU, so both ranks must fail loudly at Engine creation (no CPU fallback) and the parent must relay the failure."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_gpu_sharded.py")
    env = {k: v for k, v in os.environ.items() if k not in CLEAN}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--backend", "gloo", "--tuples-per-gpu", "64"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    text = out.stderr + out.stdout
    assert "torch.distributed" in text or "ChildFailedError" in text or "rank" in text.lower()


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_committed_bench_line_has_the_contract_fields():
    b = json.load(open(_latest("r*_final_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["unit"] == "verifies/s" and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert "workload" in b["config"] and "model" not in b["config"]
    r = b["roofline"]
    assert r["bound"] == "valu" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = b["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and "sample" in c and "value" in c


def test_rocprof_summary_agrees_with_the_bench_line():
    """The dominant kernel's rocprofv3 average and the HIP-event figure behind roofline.achieved come from the same run
    class and must agree within a few percent; the work per launch is read from the line itself, not hard-coded."""
    bpath = _latest("r*_final_bench.json")
    tag = os.path.basename(bpath).split("_")[0]
    b = json.load(open(bpath))
    rows = {r["Name"].split("(")[0]: r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_final_kernel_stats.csv")))}
    avg_ms = float(rows[b["roofline"]["kernel"]]["AverageNs"]) / 1e6
    alg = b["algorithmic_fp_mul_per_verify"]
    per_tuple = alg.get("dominant_kernel") or (alg["miller_variable_pair"] + alg["miller_fixed_pair_lines"])
    mads = per_tuple * 136 * b["config"]["tuples_per_gpu"]
    assert abs(mads / (avg_ms * 1e-3) / 1e12 / b["roofline"]["achieved"] - 1) < 0.05
    if b["roofline"].get("traffic") is not None:
        t = json.load(open(os.path.join(ROOT, "profiles", tag + "_traffic.json")))
        assert t["kernels"][b["roofline"]["kernel"]]["hbm_bytes_per_launch"] == b["roofline"]["traffic"]


def test_scripts_compile():
    """every helper script under scripts/ (GPU benchmarks, profiling summaries, differential runs) at least parses"""
    import py_compile
    d = os.path.join(ROOT, "scripts")
    for f in sorted(os.listdir(d)):
        if f.endswith(".py"):
            py_compile.compile(os.path.join(d, f), doraise=True)


def test_executed_mads_and_rank_fields_of_the_committed_lines():
    """roofline.executed_mads_per_tuple is the hostsim count committed beside it (scripts/executed_mads.py), priced consistently;
    the committed N > 1 rehearsal line carries the per-rank spread and the all-reduce time."""
    b = json.load(open(_latest("r*_final_bench.json")))
    r = b["roofline"]
    if r.get("executed_mads_per_tuple") is not None:
        e = json.load(open(os.path.join(ROOT, "profiles", "r03_executed_mads.json")))
        ph = [v for k, v in e["phases"].items() if "(" + r["kernel"] + ")" in k]
        assert len(ph) == 1 and ph[0]["executed_mads"] == r["executed_mads_per_tuple"]
        m = e["mads_per_op"]
        assert ph[0]["executed_mads"] == (m["fp_mul"] * ph[0]["fp_mul"] + m["fp_sqr"] * ph[0]["fp_sqr"] + m["fp_dot2"] * ph[0]["fp_dot2"]
                                          + m["fp_lc_term"] * ph[0]["fp_lc_terms"])
        assert abs(r["achieved_executed"] / r["achieved"] - r["executed_over_algorithmic"]) < 0.01
    n2 = glob.glob(os.path.join(ROOT, "profiles", "r03_n2_rehearsal_one_gpu.json"))
    if n2:
        j = json.loads(open(n2[0]).read().strip().splitlines()[-1])
        assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "cpu_baseline" not in j
        for k in ("ms_per_step_min", "ms_per_step_max", "allreduce_ms_min", "allreduce_ms_max", "datagen_s_max"):
            assert k in j["ranks"], k
        assert j["ranks"]["ms_per_step_min"] <= j["ranks"]["ms_per_step_max"] <= j["ms_per_step"] * 1.01
