#!/bin/bash
# A/B of workgroup shape / register budget of the link-per-lane kernel on the GPU box (rebuilds in place)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${CFGS:-"1 2" "1 3"}; do set -- $cfg
V2P_FLAGS_PHYSICS_LL="-O3 -DV2P_LL_WPB=$1 -DV2P_LL_WPS=$2" python -m vid2player3d_amd.build --force > /dev/null 2>&1
echo "WPB=$1 WPS=$2: $(python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-130)"
done
python -m vid2player3d_amd.build --force > /dev/null 2>&1
