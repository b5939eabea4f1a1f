#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprofv3 kernel stats + HBM counters.  Usage: gpurun -- 'bash tools/gpu_round.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench.log 2>&1
timeout 300 python bench.py --num-envs 1024 --no-contact --no-cpu-baseline > $O/bench_cfg2.log 2>&1
rm -rf $O/prof $O/pmc_fetch $O/pmc_write && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --steps 64 --warmup 32 --no-cpu-baseline > $O/rocprof.log 2>&1)
# HBM traffic counters, each in its own pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o pmc -- python $R/bench.py --steps 32 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o pmc -- python $R/bench.py --steps 32 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1)
# summaries on the box (the raw rocprofv3 databases exceed what is merged back)
python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats.txt 2>&1
python $R/tools/pmc_summary.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_summary.json > /dev/null 2>&1
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -1 $O/bench.log | cut -c1-1500; tail -1 $O/bench_cfg2.log | cut -c1-300; head -8 $O/rocprof_stats.txt
