"""Frechet distance between two sets of feature activations (reference compute_fid_score.py:126-205, the evaluation
metric of SURVEY.md 8f rank 4).  Host code, as in the reference: the statistics are a (dims x dims) problem solved once
per evaluation with scipy.  The Inception-v3 weights the reference downloads (models/inception.py:60) cannot be fetched
here, so the feature extractor is an argument; `compute_fid_score.py` at the repo root wires the pieces together."""
import numpy as np
from scipy import linalg


def activation_statistics(act):
    """mean and (unbiased) covariance over samples of an (n, dims) activation matrix (compute_fid_score.py:183-205)"""
    act = np.asarray(act, dtype=np.float64)
    return act.mean(axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """||mu1 - mu2||^2 + tr(S1) + tr(S2) - 2 tr((S1 S2)^(1/2)); when the matrix square root of the product is not
    finite the covariances get `eps` on their diagonals first; a numerically complex root is accepted when its
    diagonal is real to 1e-3 (compute_fid_score.py:126-180)."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape:
        raise AssertionError('Training and test mean vectors have different lengths')
    if sigma1.shape != sigma2.shape:
        raise AssertionError('Training and test covariances have different dimensions')
    root, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(root).all():
        print('fid calculation produces singular product; adding %s to diagonal of cov estimates' % eps)
        ridge = np.eye(sigma1.shape[0]) * eps
        root = linalg.sqrtm((sigma1 + ridge).dot(sigma2 + ridge))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component {}'.format(np.max(np.abs(root.imag))))
        root = root.real
    delta = mu1 - mu2
    return delta.dot(delta) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(root)


def get_activations(images, model, batch_size=64, dims=None):
    """(n, 3, H, W) float images in [0, 1] -> (n_used, dims) float64 features; whole batches only, like the reference
    (compute_fid_score.py:67-123).  `model(batch)` returns (b, dims) or (b, dims, h, w) (spatially averaged)."""
    import torch
    n = images.shape[0]
    if batch_size > n:
        print('Warning: batch size is bigger than the data size. Setting batch size to data size')
        batch_size = n
    feats = []
    with torch.no_grad():
        for i in range(n // batch_size):
            f = model(images[i * batch_size:(i + 1) * batch_size])
            if f.dim() == 4:
                f = f.mean(dim=(2, 3))
            feats.append(f.detach().double().cpu().numpy())
    out = np.concatenate(feats, axis=0)
    if dims is not None and out.shape[1] != dims:
        raise ValueError('feature extractor returned %d dims, expected %d' % (out.shape[1], dims))
    return out
