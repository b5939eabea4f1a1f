"""The five rl_games 1.1.4 helpers the PPO path leans on (neglogp, policy_kl, the masked mean of apply_masks, the RunningMeanStd value
normaliser and its moment merge) against INDEPENDENT closed forms - scipy's normal log-density, torch.distributions' Gaussian KL, a
hand-computed 2 x 3 example, numpy statistics of the concatenated data - for the product's functions (vid2player3d_amd/ppo.py) AND for
the restatement the golden generator runs the reference's methods against (oracle/ref_shim/rl_games_restated.py).

rl_games itself is absent (third-party, pinned 1.1.4 in the reference's install.sh:2, no network): both were written from its published
source by the same author, so `tests/golden/ppo_trace.npz` compares those helpers with themselves.  What THIS file pins is the mathematics
each helper claims; what it cannot pin is rl_games' conventions where they depart from the textbook - the 1e-5 guards inside policy_kl,
the divisor of the masked mean (the mask's number of elements: anchored on the reference's OWN twin, im_agent.py:573), the value
normaliser's prior (count 1, mean 0, var 1), its unbiased batch variance and epsilon inside the square root.  The golden keys that rest
on those conventions are listed in tests/test_ppo_reference.py (SELF_REFERENTIAL_KEYS)."""
import numpy as np
import pytest
import torch
from scipy import stats

from oracle.ref_shim import rl_games_restated as R
from vid2player3d_amd import ppo


def _gauss(seed, n=7, d=5):
    g = torch.Generator().manual_seed(seed)
    mu0, mu1 = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
    ls0, ls1 = 0.3 * torch.randn(n, d, generator=g) - 1.0, 0.3 * torch.randn(n, d, generator=g) - 1.0
    x = mu0 + torch.exp(ls0) * torch.randn(n, d, generator=g)
    return x, mu0, ls0, mu1, ls1


def test_neglogp_is_minus_the_normal_log_density():
    x, mu, ls, _, _ = _gauss(1)
    want = -stats.norm.logpdf(x.numpy(), loc=mu.numpy(), scale=np.exp(ls.numpy())).sum(axis=-1)
    got = ppo.neglogp(x, mu, torch.exp(ls), ls).numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
    shim = R.neglogp(x, mu, torch.exp(ls), ls).numpy() if hasattr(R, "neglogp") else None
    if shim is not None:
        np.testing.assert_allclose(shim, want, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("fn", [ppo.policy_kl, lambda *a: R.policy_kl(*a, reduce=False)])
def test_policy_kl_is_the_gaussian_kl_up_to_its_guards(fn):
    _, mu0, ls0, mu1, ls1 = _gauss(2)
    s0, s1 = torch.exp(ls0), torch.exp(ls1)
    want = torch.distributions.kl_divergence(torch.distributions.Normal(mu0, s0), torch.distributions.Normal(mu1, s1)).sum(dim=-1)
    got = fn(mu0, s0, mu1, s1)
    # the 1e-5 guards (inside the log and in the denominator) move a term by at most ~1e-5 / sigma1^2 relative: sigma ~ e^-1
    assert got.shape == want.shape and torch.allclose(got, want, rtol=2e-4, atol=2e-4)
    assert torch.allclose(fn(mu0, s0, mu0, s0), torch.zeros(mu0.shape[0]), atol=2e-4)  # KL(p || p) = 0 up to the guards


def test_masked_mean_divides_by_the_number_of_elements_of_the_mask():
    """hand-computed: x = [[1,2,3],[4,5,6]] flattened per sample to losses of shape [6, 1]; alive = [1,0,1,1,0,1]: the alive losses sum
    to 1 + 3 + 4 + 6 = 14; over the 6 ELEMENTS of the mask (im_agent.py:573's own form), not over its sum 4."""
    x = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]).unsqueeze(1)
    alive = torch.tensor([1.0, 0.0, 1.0, 1.0, 0.0, 1.0])
    assert float(ppo.masked_mean(x, alive.unsqueeze(1))) == pytest.approx(14.0 / 6.0)
    (got,), n = R.apply_masks([x], alive)
    assert float(got) == pytest.approx(14.0 / 6.0) and n == 6
    # the reference's twin, evaluated literally (embodied_pose/agents/im_agent.py:573): (kl_dist * alive).sum() / alive.numel()
    kl_dist = x.squeeze(1)
    assert float((kl_dist * alive).sum() / alive.numel()) == pytest.approx(14.0 / 6.0)


def _merged_moments_numpy(a, b):
    """What a population-style merge of two batches that ENTER with their unbiased variances must hold: mean of the concatenation;
    M2 = sum of squared deviations of the concatenation + var_unbiased(a) + var_unbiased(b)  (n var_unb - n var_pop = var_unb)."""
    c = np.concatenate([a, b])
    m2 = ((c - c.mean()) ** 2).sum() + a.var(ddof=1) + b.var(ddof=1)
    return c.mean(), m2 / len(c), len(c)


def test_value_normaliser_merges_moments_like_numpy_on_the_concatenation():
    rng = np.random.default_rng(3)
    a, b = rng.normal(2.0, 3.0, size=40), rng.normal(-1.0, 0.5, size=25)
    want_mean, want_var, want_n = _merged_moments_numpy(a, b)
    # product: start the normaliser AT batch a's moments (prior replaced), take batch b in
    v = ppo.ValueMeanStd("cpu")
    v.running_mean.fill_(a.mean()); v.running_var.fill_(a.var(ddof=1)); v.count.fill_(len(a))
    v.update(torch.tensor(b).unsqueeze(1))
    assert float(v.count) == want_n and float(v.running_mean) == pytest.approx(want_mean, rel=1e-12) and float(v.running_var) == pytest.approx(want_var, rel=1e-12)
    # restatement used by the golden generator
    m, var, n = R.RunningMeanStd._update_mean_var_count_from_moments(torch.tensor(a.mean()), torch.tensor(a.var(ddof=1)), torch.tensor(float(len(a))),
                                                                     torch.tensor(b.mean()), torch.tensor(b.var(ddof=1)), len(b))
    assert float(n) == want_n and float(m) == pytest.approx(want_mean, rel=1e-12) and float(var) == pytest.approx(want_var, rel=1e-12)


def test_value_normaliser_round_trip_and_clamp():
    v = ppo.ValueMeanStd("cpu")
    v.running_mean.fill_(1.5); v.running_var.fill_(4.0); v.count.fill_(100.0)
    x = torch.tensor([[-3.0], [0.0], [2.5], [400.0]])
    y = v(x)
    scale = float(np.sqrt(4.0 + 1e-5))
    np.testing.assert_allclose(y.numpy()[:3, 0], (np.array([-3.0, 0.0, 2.5]) - 1.5) / scale, rtol=1e-6)
    assert float(y[3]) == 5.0  # normalised values are clamped to +-5
    back = v(y, unnorm=True)
    np.testing.assert_allclose(back.numpy()[:3, 0], [-3.0, 0.0, 2.5], rtol=1e-5, atol=1e-6)
    assert float(back[3]) == pytest.approx(5.0 * scale + 1.5, rel=1e-6)
