#!/bin/bash
# Round-3 profile set (copied into profiles/r03_* by scripts/summarize_profiles.py r03): rocprofv3 kernel stats of the bench command,
# then separate PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE) of one bench step -- for the headline path, for the exact
# per-tuple path (BLSBN254_AUTO_PREPARE=0) and for the RLC path (scripts/run_rlc_once.py) -- then the plain bench lines and
# the secondary configurations.  Run through gpurun; every step appends a line to $O/progress.log.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/prof_final
mkdir -p $O
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_FLAT"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side-paths"
step() { echo "$(date +%T) $1" >> $O/progress.log; }
step stats && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side-paths > $O/stats.log 2>&1 &&
step pmc_sq && timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- $B > $O/pmc_sq.log 2>&1 &&
step pmc_fetch && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o fetch -- $B > $O/pmc_fetch.log 2>&1 &&
step pmc_write && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o write -- $B > $O/pmc_write.log 2>&1 &&
export BLSBN254_AUTO_PREPARE=0 &&
step exact_sq && timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/exact_sq -o sq -- $B > $O/exact_sq.log 2>&1 &&
step exact_fetch && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/exact_fetch -o fetch -- $B > $O/exact_fetch.log 2>&1 &&
step exact_write && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/exact_write -o write -- $B > $O/exact_write.log 2>&1 &&
step exact_bench && timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_exact.json 2> $O/bench_exact.err &&
unset BLSBN254_AUTO_PREPARE &&
step rlc_stats && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rlc_stats -o rlc -- python3 scripts/run_rlc_once.py 5 > $O/rlc_stats.log 2>&1 &&
step rlc_sq && timeout -k 10 300 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/rlc_sq -o sq -- python3 scripts/run_rlc_once.py 1 > $O/rlc_sq.log 2>&1 &&
step rlc_fetch && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/rlc_fetch -o fetch -- python3 scripts/run_rlc_once.py 1 > $O/rlc_fetch.log 2>&1 &&
step rlc_write && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/rlc_write -o write -- python3 scripts/run_rlc_once.py 1 > $O/rlc_write.log 2>&1 &&
step bench && timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err &&
step distinct && timeout -k 10 400 python3 bench.py --key-pool 0 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_distinct.json 2> $O/bench_distinct.err &&
step configs && timeout -k 10 300 python3 scripts/bench_configs.py > $O/configs.json 2> $O/configs.err &&
step rlc && timeout -k 10 300 python3 scripts/bench_rlc.py 262144 0 2,3 > $O/rlc.json 2> $O/rlc.err &&
step rlc_1m && timeout -k 10 300 python3 scripts/bench_rlc.py 1048576 0 > $O/rlc_1m.json 2> $O/rlc_1m.err &&
step small && timeout -k 10 400 python3 scripts/bench_small.py > $O/small_batches.json 2> $O/small.err &&
step host_api && timeout -k 10 200 python3 scripts/bench_host_api.py > $O/host_api.json 2> $O/host_api.err &&
step valu_peak && timeout -k 10 200 ./bench_micro/valu_peak > $O/valu_peak.json 2> $O/valu_peak.err
echo rc=$?
step done
tail -n 1 $O/bench.json | cut -c1-300
