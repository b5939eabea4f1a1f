#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for D in ${DBGS:-0}; do for L in ${LS:-32 8}; do
echo "dbg=$D L=$L"
V2P_DBG=$D V2P_DEBUG=1 V2P_PHASE_TIMING=1 V2P_ENVS_PER_BLOCK=$L timeout 300 python bench.py --steps 64 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "phase|value" | cut -c1-1400 | sed 's/"unit".*//'
done; done
