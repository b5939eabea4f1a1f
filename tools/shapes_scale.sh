#!/bin/bash
# f-3 at the reference's scale (one body shape per clip, thousands of clips): bench + rocprofv3 kernel stats + HBM counters of the physics
# kernel at 64 / 2048 / 8192 shapes, one shape for all envs beside them.  Usage: gpurun -- 'bash tools/shapes_scale.sh' -> gpurun_out/shapes/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/shapes
mkdir -p $O
cd $R
export TMPDIR=/tmp
for S in ${SHAPES:-0 64 2048 8192}; do
  if [ $S = 0 ]; then A=""; else A="--per-clip-shapes --num-shapes $S"; fi
  T0=$SECONDS
  timeout 900 python bench.py --no-cpu-baseline $A > $O/bench_$S.log 2> $O/bench_$S.err
  echo "wall $((SECONDS - T0)) s (process start to exit: import, task construction, 64 + 320 steps)" >> $O/bench_$S.err
  echo "[shapes $S] $(tail -1 $O/bench_$S.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f M env-steps/s  kernel %.4f ms  alive %.3f' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['alive_fraction_at_end']))")  $(grep -h "compiled on the device\|wall" $O/bench_$S.err | tr '\n' ' ')"
  if [ -n "${BENCH_ONLY:-}" ]; then continue; fi
  rm -rf $O/prof && mkdir -p $O/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --no-cpu-baseline --steps 96 --warmup 32 $A > $O/rocprof_$S.log 2>&1)
  python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats_$S.txt 2>&1
  rm -rf $O/prof
  PMC_ARGS="$A" bash $R/tools/pmc_probe.sh > $O/pmc_$S.log 2>&1
  cp $R/gpurun_out/pmc_summary.json $O/pmc_summary_$S.json 2>/dev/null
  echo "[shapes $S] physics kernel: $(grep physics_ll $O/rocprof_stats_$S.txt | head -1 | awk '{print "avg_us", $(NF-1)}')  $(python -c "import json; p=json.load(open('$O/pmc_summary_$S.json')); print('hbm bytes/launch calibrated', p.get('physics_kernel_hbm_bytes_calibrated'), 'fetch_x2', p.get('physics_kernel_hbm_bytes_per_launch_fetch_x2'))" 2>/dev/null)"
done 2>&1 | tee $O/summary.txt
