"""ORACLE (test infrastructure, not product code).

Frechet distance of the reference (compute_fid_score.py:126-180) restated two ways for the tests: `frechet_eig` through
the eigenvalues of S1 S2 (tr sqrt(S1 S2) = sum of sqrt of its eigenvalues -- independent of scipy's sqrtm), and closed
forms for commuting (diagonal) covariances.  Pinned by tests/golden/fid.npz, written by oracle/make_golden.py --only fid
from the reference's own calculate_frechet_distance."""
import numpy as np


def frechet_eig(mu1, s1, mu2, s2):
    ev = np.linalg.eigvals(s1.dot(s2))
    tr_root = np.sqrt(np.clip(ev.real, 0, None)).sum()
    d = mu1 - mu2
    return d.dot(d) + np.trace(s1) + np.trace(s2) - 2 * tr_root


def frechet_diagonal(mu1, v1, mu2, v2):
    """covariances diag(v1), diag(v2): sum (mu1-mu2)^2 + sum (sqrt(v1) - sqrt(v2))^2"""
    return ((mu1 - mu2) ** 2).sum() + ((np.sqrt(v1) - np.sqrt(v2)) ** 2).sum()
