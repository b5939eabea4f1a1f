#!/bin/bash
# HBM traffic of the engine's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (kernel trace only) -> gpurun_out/pmc_summary.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/pmc_fetch $O/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o pmc -- python $R/bench.py --steps 32 --warmup 0 --no-cpu-baseline ${PMC_ARGS:-} > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o pmc -- python $R/bench.py --steps 32 --warmup 0 --no-cpu-baseline ${PMC_ARGS:-} > $O/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_fetch/pmc_results.db $O/pmc_write/pmc_results.db $O/pmc_summary.json | head -12
rm -rf $O/pmc_fetch $O/pmc_write
