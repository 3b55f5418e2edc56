"""Comparison helpers shared by the parity tests."""
import torch


def rel_err(got, ref):
    """max-abs error relative to the largest reference magnitude (both moved to fp64 CPU)."""
    g = got.detach().double().cpu()
    r = ref.detach().double().cpu()
    assert g.shape == r.shape, 'shape %s vs %s' % (tuple(g.shape), tuple(r.shape))
    scale = r.abs().max().item() + 1e-30
    return (g - r).abs().max().item() / scale


def assert_close(got, ref, tol, name=''):
    e = rel_err(got, ref)
    assert e == e and e <= tol, '%s: relative error %.3e > %.1e' % (name, e, tol)
    return e
