#!/bin/bash
# SQ counters of the physics kernel in separate passes (kernel trace only); GROUPS_ overrides the counter groups ("a b|c d"), VALU_ARGS adds bench.py
# arguments (e.g. "--racket-ball"), VALU_OUT names the summary (default valu_summary.json)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
DEF="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES|SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU|SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32|SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM|GRBM_GUI_ACTIVE"
IFS='|' read -ra GR <<< "${GROUPS_:-$DEF}"
i=0
for grp in "${GR[@]}"; do
i=$((i+1)); rm -rf $O/pmc_valu_$i
timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $O/pmc_valu_$i -o pmc -- python $R/bench.py --steps 16 --warmup 0 --no-cpu-baseline ${VALU_ARGS:-} > $O/pmc_valu_$i.log 2>&1
done
python $R/tools/valu_summary.py $O > $O/${VALU_OUT:-valu_summary.json}; rm -rf $O/pmc_valu_*/; cat $O/${VALU_OUT:-valu_summary.json}
