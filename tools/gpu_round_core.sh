#!/bin/bash
# The core of tools/gpu_round.sh (what the bench line and DESIGN 5 quote): parity tests, smoke, bench (+ driver command, variants), rocprofv3 kernel
# stats of three commands, HBM counters, SQ counters, epoch profile, parity sweep.  Usage: gpurun -- 'bash tools/gpu_round_core.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -s > $O/pytest_full.log 2>&1
tail -30 $O/pytest_full.log > $O/pytest_gpu.log
grep -h "^\[selection\]\|^\[schedules\]" $O/pytest_full.log > $O/selection.log; grep -h "^\[rows\]\|^\[close\]\|^\[vfric\]\|^\[limits\]\|^\[racket-ball" $O/pytest_full.log > $O/rows.log; rm -f $O/pytest_full.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 600 python bench.py > $O/bench.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
for v in "--num-envs 1024 --no-contact" "--num-envs 1024" "--num-envs 4096" "--djokovic" "--racket-ball" "--racket-ball --solver tgs" "--racket-ball --joint-limits 0" "--racket-ball --ball-body-contacts 0" "--racket-ball --per-clip-shapes" "--per-clip-shapes" "--solver tgs" "--friction-frame velocity" "--groups 2" "--action-noise 0.03" "--num-envs 16384 --steps 96 --warmup 32" "--num-envs 32768 --steps 96 --warmup 32" "--num-envs 65536 --steps 64 --warmup 32" "--substep-jobs 0"; do
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
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python $R/bench.py --racket-ball --solver tgs --steps 64 --warmup 32 --no-cpu-baseline > $O/rocprof_rbt.log 2>&1)
python $R/tools/rocprof_summary.py $O/prof/stats_results.db > $O/rocprof_stats_racket_ball_tgs.txt 2>&1
rm -rf $O/prof
bash $R/tools/pmc_probe.sh > $O/pmc.log 2>&1
bash $R/tools/valu_probe.sh > $O/valu.log 2>&1
python tools/epoch_profile.py --epochs 4 > $O/epoch_profile.txt 2>&1
V2P_DEBUG=1 V2P_PHASE_TIMING=1 timeout 300 python bench.py --steps 64 --warmup 0 --no-cpu-baseline 2>&1 | grep phase > $O/phase.log
timeout 900 python tools/parity_sweep.py 2048 > $O/parity_sweep.log 2>&1
timeout 1500 python tools/parity_sweep.py 16384 2>&1 | grep -v "^\[rows\]" > $O/parity_sweep_16k.log
timeout 600 python tools/soak.py 4000 2>&1 | tail -3 > $O/soak.log
timeout 600 python tools/soak.py 4000 racket 2>&1 | tail -3 > $O/soak_racket_ball.log
timeout 900 python tools/soak.py 8000 racket 8192 tgs 2>&1 | tail -3 > $O/soak_racket_ball_tgs.log
timeout 600 python tools/soak.py 4000 racket 8192 tgs velocity 2>&1 | tail -3 > $O/soak_racket_ball_tgs_vfric.log
# round 5: two processes on one GPU, the rollout's kernel budget, per-clip shapes at scale (own script), MFMA microbenchmark, host CPU scaling
timeout 900 python tools/soak2.py 2000 > $O/soak_two_processes.log 2>&1
timeout 300 python tools/play_profile.py 4 > $O/play_wall.log 2>&1
rm -rf $O/prof && mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o play -- python $R/tools/play_profile.py 4 > $O/play_rocprof.log 2>&1)
python $R/tools/play_summary.py $O/prof/play_results.db 5 > $O/play_steps_kernels.txt 2>&1
rm -rf $O/prof
[ -x $R/tools/ubench/mfma_congruence ] && $R/tools/ubench/mfma_congruence > $O/mfma_congruence.txt 2>&1
timeout 300 python tools/cpu_scaling_probe.py 4096 > $O/cpu_scaling_default.txt 2>&1
# (bash tools/shapes_scale.sh: bench + rocprofv3 + PMC at 1 / 64 / 2048 / 8192 body shapes -> gpurun_out/shapes/)
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -1 $O/bench.log | cut -c1-700; cut -c1-220 $O/bench_variants.log; tail -1 $O/bench_ppo.log | cut -c1-300; head -5 $O/rocprof_stats_default_cmd.txt; tail -3 $O/parity_sweep.log; cat $O/soak.log
