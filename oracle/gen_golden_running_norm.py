"""Golden vectors for RunningNorm in TRAINING mode, recorded from the reference's own class
(/root/reference/embodied_pose/models/running_norm.py:5-43): three batches through forward() of a fresh module (statistics updated
before each normalisation), the outputs and the buffers after every batch.  Run in the build container:
    python oracle/gen_golden_running_norm.py        -> tests/golden/running_norm.npz
"""
import importlib.util
import os
import sys

sys.dont_write_bytecode = True  # never leave __pycache__ in the read-only reference mount

import numpy as np
import torch

REF = "/root/reference/embodied_pose/models/running_norm.py"
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("ref_running_norm", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

torch.manual_seed(5)
dim = 48
rn = mod.RunningNorm(dim, clip=5.0)
rn.train()
out = {}
scales = torch.rand(dim) * 3 + 0.1
for k, m in enumerate((16, 5, 33)):
    x = torch.randn(m, dim) * scales + torch.randn(dim) * (k + 1)
    y = rn(x)
    out["x%d" % k] = x.numpy().astype(np.float32)
    out["y%d" % k] = y.numpy().astype(np.float32)
    out["mean%d" % k] = rn.mean.numpy().copy()
    out["var%d" % k] = rn.var.numpy().copy()
    out["std%d" % k] = rn.std.numpy().copy()
    out["n%d" % k] = np.int64(rn.n.item())
rn.eval()
x = torch.randn(7, dim) * 4
out["x_eval"] = x.numpy().astype(np.float32)
out["y_eval"] = rn(x).numpy().astype(np.float32)
np.savez(os.path.join(HERE, "..", "tests", "golden", "running_norm.npz"), **out)
print("wrote tests/golden/running_norm.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})
