"""Copy the round-end profile set from gpurun_out/prof_final into profiles/ and derive profiles/<tag>_traffic.json
and the per-kernel VALU-busy / wait fractions quoted in DESIGN.md.  Usage: python scripts/summarize_profiles.py [tag=r02]"""
import collections, csv, json, os, shutil, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_final")
DST = os.path.join(ROOT, "profiles")
os.makedirs(os.path.join(DST, TAG + "_pmc_final"), exist_ok=True)
shutil.copy(os.path.join(SRC, "stats", "bench_kernel_stats.csv"), os.path.join(DST, TAG + "_final_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "pmc_sq", "sq_counter_collection.csv"), os.path.join(DST, TAG + "_pmc_final", "SQ_WAVES_counter_collection.csv"))
shutil.copy(os.path.join(SRC, "pmc_fetch", "fetch_counter_collection.csv"), os.path.join(DST, TAG + "_pmc_final", "FETCH_SIZE_counter_collection.csv"))
shutil.copy(os.path.join(SRC, "pmc_write", "write_counter_collection.csv"), os.path.join(DST, TAG + "_pmc_final", "WRITE_SIZE_counter_collection.csv"))
line = [l for l in open(os.path.join(SRC, "bench.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(DST, TAG + "_final_bench.json"), "w"), indent=1)


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); launches[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in launches.items()}


f, fl = per_kernel(os.path.join(SRC, "pmc_fetch", "fetch_counter_collection.csv"))
w, wl = per_kernel(os.path.join(SRC, "pmc_write", "write_counter_collection.csv"))
out = {"_source": "profiles/" + TAG + "_pmc_final/{FETCH,WRITE}_SIZE_counter_collection.csv (rocprofv3 --pmc, separate passes, bench.py --steps 1 "
                  "--warmup 0, final build of the round); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch (FETCH_SIZE/WRITE_SIZE in KiB; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM'); averages over the launches of each kernel in the run", "kernels": {}}
for k in f:
    fk, wk = f[k]["FETCH_SIZE"] / fl[k], w[k]["WRITE_SIZE"] / max(wl.get(k, 1), 1)
    out["kernels"][k] = {"fetch_kib": round(fk), "write_kib": round(wk), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump(out, open(os.path.join(DST, TAG + "_traffic.json"), "w"), indent=1)
# the bench line above was produced before this traffic file existed: carry the PMC figure of the same build into it
bj = json.load(open(os.path.join(DST, TAG + "_final_bench.json")))
dom = bj["roofline"]["kernel"]
t = out["kernels"][dom]["hbm_bytes_per_launch"]
bj["roofline"]["traffic"] = t
bj["roofline"]["traffic_note"] = ("PMC FETCH_SIZE/WRITE_SIZE of the same kernel at the same batch size (profiles/%s_traffic.json): %.1f KB per tuple "
                                  "against ~0.6 KB algorithmic (sig 64 B, H 108 B, f out 432 B; line tables from cache)" % (TAG, t / 262144 / 1024.0))
for extra in ("bench_exact.json", "configs.json", "valu_peak.json", "rlc.json", "rlc_1m.json"):
    src = os.path.join(SRC, extra)
    if os.path.exists(src):
        txt = open(src).read()
        if extra == "bench_exact.json":
            txt = json.dumps(json.loads([l for l in txt.splitlines() if l.startswith("{")][-1]), indent=1)
        open(os.path.join(DST, TAG + "_" + extra.replace("bench_exact", "exact_path_bench")), "w").write(txt)
json.dump(bj, open(os.path.join(DST, TAG + "_final_bench.json"), "w"), indent=1)
s, sl = per_kernel(os.path.join(SRC, "pmc_sq", "sq_counter_collection.csv"))
for k, v in sorted(s.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc = v["SQ_WAVE_CYCLES"] or 1
    print("%-18s launches %2d  VALU-busy %5.1f%%  wait_any %5.1f%%  wait_inst %5.1f%%  cycles/VALU-inst %.2f   HBM %.2f GB/launch" % (
        k, sl[k], 100 * v["SQ_ACTIVE_INST_VALU"] / wc, 100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc,
        wc / max(v["SQ_INSTS_VALU"], 1) , out["kernels"].get(k, {}).get("hbm_bytes_per_launch", 0) / 1e9))


# ---- round 3: the same three PMC passes for the exact per-tuple path (BLSBN254_AUTO_PREPARE=0) and for the RLC path
summary_lines = []
for setname in ("exact", "rlc"):
    fdir, wdir, sdir = (os.path.join(SRC, "%s_%s" % (setname, x)) for x in ("fetch", "write", "sq"))
    if not (os.path.isdir(fdir) and os.path.isdir(wdir) and os.path.isdir(sdir)):
        continue
    for sub, name, cname in ((fdir, "fetch", "FETCH_SIZE"), (wdir, "write", "WRITE_SIZE"), (sdir, "sq", "SQ_WAVES")):
        shutil.copy(os.path.join(sub, "%s_counter_collection.csv" % name), os.path.join(DST, TAG + "_pmc_final", "%s_%s_counter_collection.csv" % (setname, cname)))
    f2, fl2 = per_kernel(os.path.join(fdir, "fetch_counter_collection.csv"))
    w2, wl2 = per_kernel(os.path.join(wdir, "write_counter_collection.csv"))
    s2, sl2 = per_kernel(os.path.join(sdir, "sq_counter_collection.csv"))
    sec = {}
    for k in f2:
        fk, wk = f2[k]["FETCH_SIZE"] / fl2[k], w2[k]["WRITE_SIZE"] / max(wl2.get(k, 1), 1)
        sec[k] = {"fetch_kib": round(fk), "write_kib": round(wk), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    out["kernels_" + setname] = sec
    for k, v in sorted(s2.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
        wc = v["SQ_WAVE_CYCLES"] or 1
        line = "[%s] %-24s launches %2d  VALU-busy %5.1f%%  wait_any %5.1f%%  wait_inst %5.1f%%  cycles/VALU-inst %.2f   HBM %.3f GB/launch" % (
            setname, k, sl2[k], 100 * v["SQ_ACTIVE_INST_VALU"] / wc, 100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc,
            wc / max(v["SQ_INSTS_VALU"], 1), sec.get(k, {}).get("hbm_bytes_per_launch", 0) / 1e9)
        print(line); summary_lines.append(line)
json.dump(out, open(os.path.join(DST, TAG + "_traffic.json"), "w"), indent=1)
for extra in ("bench_distinct.json", "small_batches.json", "host_api.json"):
    src = os.path.join(SRC, extra)
    if os.path.exists(src):
        txt = open(src).read()
        if extra == "bench_distinct.json":
            txt = json.dumps(json.loads([l for l in txt.splitlines() if l.startswith("{")][-1]), indent=1)
        open(os.path.join(DST, TAG + "_" + extra.replace("bench_distinct", "distinct_keys_bench")), "w").write(txt)
rs = os.path.join(SRC, "rlc_stats", "rlc_kernel_stats.csv")
if os.path.exists(rs):
    shutil.copy(rs, os.path.join(DST, TAG + "_rlc_kernel_stats.csv"))
# the headline set's table as text too
with open(os.path.join(DST, TAG + "_pmc_summary.txt"), "w") as fh:
    fh.write("rocprofv3 --pmc passes of the final build (scripts/profile_r03.sh): per kernel, summed over its launches in ONE step\n")
    fh.write("VALU-busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; HBM = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch (MI355X_MICROARCH.md)\n\n")
    for k, v in sorted(s.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
        wc = v["SQ_WAVE_CYCLES"] or 1
        fh.write("[headline] %-24s launches %2d  VALU-busy %5.1f%%  wait_any %5.1f%%  wait_inst %5.1f%%  cycles/VALU-inst %.2f   HBM %.3f GB/launch\n" % (
            k, sl[k], 100 * v["SQ_ACTIVE_INST_VALU"] / wc, 100 * v["SQ_WAIT_ANY"] / wc, 100 * v["SQ_WAIT_INST_ANY"] / wc,
            wc / max(v["SQ_INSTS_VALU"], 1), out["kernels"].get(k, {}).get("hbm_bytes_per_launch", 0) / 1e9))
    for line in summary_lines:
        fh.write(line + "\n")
