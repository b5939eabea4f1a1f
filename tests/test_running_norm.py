"""RunningNorm mirror (vid2player3d_amd/learning.py) against vectors recorded from the reference's own module in training mode
(oracle/gen_golden_running_norm.py -> tests/golden/running_norm.npz).  Pure torch: runs without a GPU."""
import os

import numpy as np
import torch

from vid2player3d_amd.learning import RunningNorm

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "running_norm.npz"))


def test_training_mode_updates_then_normalises_like_the_reference():
    rn = RunningNorm(48, clip=5.0)
    assert int(rn.n) == 0
    x = torch.as_tensor(G["x0"])
    rn.eval()
    assert torch.equal(rn(x), x), "n == 0: nothing is normalised (running_norm.py:36)"
    rn.train()
    for k in range(3):
        y = rn(torch.as_tensor(G["x%d" % k]))
        assert int(rn.n) == int(G["n%d" % k])
        np.testing.assert_allclose(rn.mean.numpy(), G["mean%d" % k], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rn.var.numpy(), G["var%d" % k], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rn.std.numpy(), G["std%d" % k], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(y.numpy(), G["y%d" % k], rtol=1e-6, atol=1e-6)
    rn.eval()
    n_before = int(rn.n)
    np.testing.assert_allclose(rn(torch.as_tensor(G["x_eval"])).numpy(), G["y_eval"], rtol=1e-6, atol=1e-6)
    assert int(rn.n) == n_before, "eval mode leaves the statistics alone"


def test_clip_and_flags():
    rn = RunningNorm(4, clip=None)
    rn(torch.tensor([[0.0, 1.0, 2.0, 3.0], [100.0, 1.0, -2.0, 3.0]]))
    rn.eval()
    y = rn(torch.tensor([[1e6, 1.0, 0.0, 3.0]]))
    assert float(y[0, 0]) > 5.0, "a falsy clip means no clamp (running_norm.py:41)"
    rn2 = RunningNorm(4, demean=False, destd=True, clip=5.0)
    rn2(torch.tensor([[2.0, 2.0, 2.0, 2.0], [4.0, 4.0, 4.0, 4.0]]))
    rn2.eval()
    y2 = rn2(torch.tensor([[1.0, 1.0, 1.0, 1.0]]))
    assert torch.allclose(y2, torch.ones(1, 4) / (1.0 + 1e-8)), "demean=False: only the scale is applied"
