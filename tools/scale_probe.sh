#!/bin/bash
# physics kernel time vs number of envs: flat below some N = bound by the slowest wave (critical path), linear = throughput bound
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for n in ${NS:-1024 2048 4096 8192 16384 32768}; do
echo -n "N=$n: "
timeout 300 python bench.py --steps 96 --warmup 32 --num-envs $n --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'env-steps/s  kernel ms', round(d['roofline']['kernel_ms'],3), ' alive', d['config']['alive_fraction_at_end'])"
done
