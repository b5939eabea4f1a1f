#!/bin/bash
# The short GPU-box round (after a change that leaves the 8192-env kernel alone): parity tests, smoke, bench (default + driver command +
# the variants that run the other build), rocprofv3 kernel stats of the default command.  Usage: gpurun -- 'bash tools/gpu_round_short.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
for v in "--num-envs 1024 --no-contact" "--num-envs 1024" "--num-envs 4096" "--num-envs 4096 --kernel-build 1" "--num-envs 4096 --racket-ball" "--racket-ball" "--solver tgs" "--djokovic"; do
  echo "[$v] $(timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1)"
done > $O/bench_variants.log 2>&1
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --no-cpu-baseline > $O/rocprof.log 2>&1)
python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats_default_cmd.txt 2>&1
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --no-cpu-baseline --num-envs 4096 > $O/rocprof2.log 2>&1)
python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats_4096_envs.txt 2>&1
rm -rf $O/prof
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -1 $O/bench.log | cut -c1-600; cut -c1-200 $O/bench_variants.log; head -5 $O/rocprof_stats_default_cmd.txt; head -5 $O/rocprof_stats_4096_envs.txt
