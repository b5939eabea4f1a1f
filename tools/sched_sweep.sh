#!/bin/bash
# launch-schedule sweep on the GPU box: pair_mix_permille x job_mono_permille at 8192 envs.  Usage: gpurun -- 'bash tools/sched_sweep.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for mix in ${MIXES:-0 125 250 375 500}; do for mono in ${MONOS:-125 250 400}; do
  echo "mix=$mix mono=$mono $(timeout 200 python bench.py --no-cpu-baseline --pair-mix $mix --job-mono $mono ${EXTRA:-} 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('%10.0f env-steps/s  kernel ms %.4f' % (d['value'], d['roofline']['kernel_ms']))")"
done; done | tee $O/sched_sweep.log
