#!/bin/bash
# A/B of compile-time switches of the link-per-lane kernel on the GPU box: FLAGSETS="-DX=1|-DX=2" (rebuilds in place, restores the default build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
IFS='|' read -ra FS <<< "${FLAGSETS:-}"
for f in "${FS[@]}"; do
V2P_FLAGS_PHYSICS_LL="-O3 $f" python -m vid2player3d_amd.build --force > /dev/null 2>&1
echo "$f: $(python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c77-100)"
done
python -m vid2player3d_amd.build --force > /dev/null 2>&1
