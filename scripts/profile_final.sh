#!/bin/bash
# Round-end profile set (copied into profiles/<tag>_* by scripts/summarize_profiles.py <tag>): rocprofv3 kernel stats of the bench command, then three separate PMC passes
# (SQ counters, FETCH_SIZE, WRITE_SIZE) of one bench step, then the plain bench line.  Run through gpurun.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/prof_final
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side-paths > $O/stats.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_FLAT --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-paths > $O/pmc_sq.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-paths > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-paths > $O/pmc_write.log 2>&1 &&
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err &&
BLSBN254_AUTO_PREPARE=0 timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_exact.json 2> $O/bench_exact.err &&
timeout -k 10 300 python3 scripts/bench_configs.py > $O/configs.json 2> $O/configs.err &&
timeout -k 10 300 python3 scripts/bench_rlc.py 262144 8,0 2,3 > $O/rlc.json 2> $O/rlc.err &&
timeout -k 10 300 python3 scripts/bench_rlc.py 1048576 0 > $O/rlc_1m.json 2> $O/rlc_1m.err &&
timeout -k 10 200 ./bench_micro/valu_peak > $O/valu_peak.json 2> $O/valu_peak.err
echo rc=$?
tail -n 1 $O/bench.json | cut -c1-300
