#!/bin/bash
# One GPU-box round: parity tests, smoke, bench (+ variants), rocprofv3 kernel stats, HBM counters, SQ counters, wave timeline.
# Usage: gpurun -- 'bash tools/gpu_round.sh'; the summaries land in gpurun_out/ (copy what is to be kept into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -s > $O/pytest_full.log 2>&1
tail -30 $O/pytest_full.log > $O/pytest_gpu.log
grep -h "^\[selection\]\|^\[schedules\]" $O/pytest_full.log > $O/selection.log; rm -f $O/pytest_full.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1   # the driver's command: carries the whole_epoch block
for v in "--num-envs 1024 --no-contact" "--djokovic" "--racket-ball" "--racket-ball --joint-limits 0" "--racket-ball --ball-body-contacts 0" "--racket-ball --substep-jobs 0" "--racket-ball --per-clip-shapes" "--per-clip-shapes" "--solver tgs" "--groups 2" "--freeze-terminated" "--action-noise 0.03" "--num-envs 32768 --steps 96 --warmup 32" "--substep-jobs 0"; do
  echo "[$v] $(timeout 300 python bench.py --no-cpu-baseline $v 2>&1 | tail -1)"
done > $O/bench_variants.log 2>&1
timeout 600 python bench.py --ppo --ppo-epochs 3 > $O/bench_ppo.log 2>&1
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --no-cpu-baseline > $O/rocprof.log 2>&1)
python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats_default_cmd.txt 2>&1
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/rocprof2.log 2>&1)
python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats_driver_cmd.txt 2>&1
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --racket-ball --steps 64 --warmup 32 --no-cpu-baseline > $O/rocprof_rb.log 2>&1)
python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats_racket_ball.txt 2>&1
rm -rf $O/prof
bash $R/tools/pmc_probe.sh > $O/pmc.log 2>&1
bash $R/tools/hbm_calib.sh > $O/hbm_calib.log 2>&1
timeout 900 python tools/parity_sweep.py 2048 > $O/parity_sweep.log 2>&1
bash $R/tools/valu_probe.sh > $O/valu.log 2>&1
V2P_DEBUG=1 V2P_WAVE_TIMES=$O/wave_times.bin timeout 300 python bench.py --steps 29 --warmup 0 --no-cpu-baseline --substep-jobs 0 > $O/wt.log 2>&1
python tools/wave_times.py $O/wave_times.bin > $O/wave_times.txt 2>&1; rm -f $O/wave_times.bin
V2P_DEBUG=1 V2P_PHASE_TIMING=1 timeout 300 python bench.py --steps 64 --warmup 0 --no-cpu-baseline 2>&1 | grep phase > $O/phase.log
V2P_PHASE_HEAVY=1 V2P_DEBUG=1 V2P_PHASE_TIMING=1 timeout 300 python bench.py --steps 64 --warmup 0 --no-cpu-baseline 2>&1 | grep phase > $O/phase_heavy.log
[ -f variants/libv2p_nowalk.so ] && bash tools/walk_ab.sh > $O/walk_ab.log 2>&1
python tools/epoch_profile.py --epochs 4 > $O/epoch_profile.txt 2>&1
timeout 600 python tools/limit_cost.py > $O/limit_cost.txt 2>&1
timeout 900 python tools/soak.py 6000 2>&1 | tail -4 > $O/soak.log
timeout 900 python tools/soak.py 6000 racket 2>&1 | tail -4 > $O/soak_racket_ball.log
timeout 600 python bench.py --ppo --ppo-epochs 60 2> $O/ppo_learning.log > /dev/null
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -1 $O/bench.log | cut -c1-1500; cut -c1-260 $O/bench_variants.log; tail -1 $O/bench_ppo.log | cut -c1-400; head -8 $O/rocprof_stats_default_cmd.txt
