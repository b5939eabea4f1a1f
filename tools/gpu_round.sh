#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprofv3 kernel stats.  Usage: gpurun -- 'bash tools/gpu_round.sh'
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
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --steps 64 --warmup 32 --no-cpu-baseline > $O/rocprof.log 2>&1)
find $O/prof -name "*stats*" | head > $O/prof_files.txt
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -3; tail -2 $O/bench.log; tail -1 $O/bench_cfg2.log; tail -5 $O/rocprof.log
