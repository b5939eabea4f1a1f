#!/bin/bash
# Parity sweep (tools/parity_sweep.py) of library variants on the GPU box: the default build first (with a dump of its outlier envs and
# the env-per-lane schedule on the same states), then every variants/libv2p_*.so in turn.  usage: gpurun -- 'bash tools/parity_ab.sh [envs]'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
N=${1:-2048}
L=vid2player3d_amd/libv2p_rollout.so
cp $L /tmp/default.so
timeout 900 python tools/parity_sweep.py $N --dump $O/outliers.npz > $O/parity_default.log 2>&1
timeout 900 python tools/parity_sweep.py $N --schedule env_per_lane --fixtures "pd only,standing,fallen,fast,low" > $O/parity_env_per_lane.log 2>&1
for v in variants/libv2p_*.so; do
  [ -f "$v" ] || continue
  cp $v $L; NAME=$(basename $v .so | sed 's/libv2p_//')
  timeout 900 python tools/parity_sweep.py $N --fixtures "pd only,standing,fallen,fast,low,limits" > $O/parity_$NAME.log 2>&1
done
cp /tmp/default.so $L
grep -h "^\[cond\]" $O/parity_default.log | cut -c1-400
for f in $O/parity_*.log; do echo "== $f"; grep "^\[sweep\]" $f | cut -c1-230; done
