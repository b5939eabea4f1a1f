#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; rm -f gpurun_out/sweep.log
timeout 600 python -m pytest tests/test_gpu_physics.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_phys.log
tail -5 gpurun_out/pytest_phys.log
for L in 8 4 16 32; do
  echo "envs_per_block=$L" >> gpurun_out/sweep.log
  V2P_DEBUG=1 V2P_ENVS_PER_BLOCK=$L timeout 300 python bench.py --steps 96 --warmup 32 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
