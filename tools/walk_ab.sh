#!/bin/bash
# A/B of the in-tree library against variants/libv2p_<name>.so (default nowalk) on identical inputs (tools/walk_ab.py).
# The variant: V2P_FLAGS_PHYSICS_LL="-O3 -DV2P_LL_WALK=0" python -c "from vid2player3d_amd import build; build.build(force=True, lib_out='variants/libv2p_nowalk.so', tag='_nowalk')"
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=vid2player3d_amd/libv2p_rollout.so; V=variants/libv2p_${1:-nowalk}.so
cp $L /tmp/default.so
echo "== in-tree"; python tools/walk_ab.py /tmp/a.npz ${NENV:-512} 2>&1 | grep -v "^\[selection\]\|amdgpu.ids"
cp $V $L
echo "== $V"; python tools/walk_ab.py /tmp/b.npz ${NENV:-512} 2>&1 | grep -v "^\[selection\]\|amdgpu.ids"
cp /tmp/default.so $L
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/a.npz"), np.load("/tmp/b.npz")
for k in ("standing", "fallen", "fast", "low"):
    d = np.abs(a[k + "_dvel"] - b[k + "_dvel"]).max(axis=1)
    print("%-9s in-tree vs variant, dof vel per env: median %.2e p90 %.2e p99 %.2e max %.2e; envs > 1e-3: %d of %d" % (k, np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max(), (d > 1e-3).sum(), len(d)))
PY
