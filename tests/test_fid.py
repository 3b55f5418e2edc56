"""CPU: Frechet distance against the vectors captured from the reference's own
calculate_frechet_distance (tests/golden/fid.npz), the oracle's eigenvalue restatement and closed forms."""
import os

import numpy as np
import pytest

from oracle import fid_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fid.npz')


def test_matches_reference_vectors():
    from pcgan_amd.util.fid import activation_statistics, frechet_distance
    g = np.load(GOLD)
    for i in range(4):
        m1, s1 = activation_statistics(g['act1_%d' % i])
        m2, s2 = activation_statistics(g['act2_%d' % i])
        got, want = frechet_distance(m1, s1, m2, s2), float(g['fid_%d' % i])
        assert abs(got - want) <= 1e-9 * abs(want), (i, got, want)
        assert abs(frechet_distance(m1, s1, m1, s1) - float(g['fid_self_%d' % i])) <= 1e-6
        if i < 3:       # full-rank cases: the oracle's eigenvalue route agrees too
            assert abs(R.frechet_eig(m1, s1, m2, s2) - want) <= 1e-6 * abs(want)


def test_closed_forms_and_errors():
    from pcgan_amd.util.fid import frechet_distance
    rng = np.random.default_rng(0)
    mu1, mu2 = rng.normal(size=12), rng.normal(size=12)
    v1, v2 = rng.uniform(0.1, 3, size=12), rng.uniform(0.1, 3, size=12)
    assert abs(frechet_distance(mu1, np.diag(v1), mu2, np.diag(v2)) - R.frechet_diagonal(mu1, v1, mu2, v2)) < 1e-9
    assert abs(frechet_distance(2.0, 4.0, -1.0, 1.0) - (9.0 + (2.0 - 1.0) ** 2)) < 1e-12      # scalars are promoted
    with pytest.raises(AssertionError):
        frechet_distance(mu1, np.diag(v1), mu2[:5], np.diag(v2))
    with pytest.raises(AssertionError):
        frechet_distance(mu1, np.diag(v1), mu2, np.diag(v2)[:5, :5])


def test_get_activations_uses_whole_batches_and_pools():
    import torch
    from pcgan_amd.util.fid import get_activations
    x = torch.arange(10 * 3 * 4 * 4, dtype=torch.float32).reshape(10, 3, 4, 4) / 1000
    act = get_activations(x, lambda b: b * 2, batch_size=4)
    assert act.shape == (8, 3) and act.dtype == np.float64
    assert np.allclose(act, (x[:8] * 2).mean(dim=(2, 3)).double().numpy())
    assert get_activations(x, lambda b: b.flatten(1)[:, :5], batch_size=64).shape == (10, 5)
    with pytest.raises(ValueError):
        get_activations(x, lambda b: b.flatten(1), batch_size=5, dims=7)
