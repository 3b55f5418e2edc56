"""Comparison helpers shared by the parity tests."""
import torch


def rel_err(got, ref):
    """max-abs error relative to the largest reference magnitude (both moved to fp64 CPU)."""
    g = got.detach().double().cpu()
    r = ref.detach().double().cpu()
    assert g.shape == r.shape, 'shape %s vs %s' % (tuple(g.shape), tuple(r.shape))
    scale = r.abs().max().item() + 1e-30
    return (g - r).abs().max().item() / scale


def assert_close(got, ref, tol, name='', atol=0.0):
    """max|got-ref| <= tol * max|ref| + atol.  `atol` is only used for tensors whose true value
    is 0 (gradients of biases cancelled by a following normalisation: fp32 noise)."""
    g = got.detach().double().cpu()
    r = ref.detach().double().cpu()
    assert g.shape == r.shape, '%s: shape %s vs %s' % (name, tuple(g.shape), tuple(r.shape))
    err = (g - r).abs().max().item() if g.numel() else 0.0
    scale = r.abs().max().item() if r.numel() else 0.0
    assert err == err and err <= tol * scale + atol, \
        '%s: max abs error %.3e > %.1e * %.3e + %.1e' % (name, err, tol, scale, atol)
    return err / (scale + 1e-30)
