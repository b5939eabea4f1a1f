#!/bin/bash
# A/B of library variants on the GPU box: every variants/libv2p_*.so is copied over the in-tree library in turn and benched
# (the default build first).  Usage: gpurun -- 'bash tools/variants.sh [bench args]'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
L=vid2player3d_amd/libv2p_rollout.so
cp $L /tmp/default.so
run() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('%-28s %10.0f env-steps/s  ms/step %.4f  kernel ms %.4f' % ('$NAME', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
except Exception as e: print('$NAME', 'FAILED', e)"; }
NAME=default; run "$@" | tee $O/variants.log
for v in variants/libv2p_*.so; do
  [ -f "$v" ] || continue
  cp $v $L; NAME=$(basename $v .so | sed 's/libv2p_//'); run "$@" | tee -a $O/variants.log
done
cp /tmp/default.so $L
