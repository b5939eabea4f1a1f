#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for L in 64 32 16 8 4; do
  echo "lanes=$L" >> gpurun_out/sweep.log
  V2P_LANES_PER_WAVE=$L timeout 300 python bench.py --steps 96 --warmup 32 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
