#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for cfg in "8 1024" "8 2048" "8 4096" "8 8192" "32 2048" "32 8192" "32 16384" "32 32768" "16 8192" "16 16384"; do
set -- $cfg
echo -n "L=$1 N=$2: "
V2P_ENVS_PER_BLOCK=$1 timeout 300 python bench.py --steps 96 --warmup 32 --num-envs $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],3), d['config']['alive_fraction_at_end'])"
done
