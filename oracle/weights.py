"""ORACLE helper: deterministic, reference-free weight generation.

`fill_state_dict(sd, seed)` overwrites every entry of a state_dict with values drawn from a
numpy RandomState keyed by (seed, crc32(key)), so the reference nets (in make_golden.py),
the oracle nets and the HIP nets receive bit-identical weights without any of them having to
ship a checkpoint (a ResNet-18 is 45 MB).
"""
import zlib

import numpy as np
import torch


def _rs(seed, key):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31 - 1))


def fill_state_dict(sd, seed=0):
    out = {}
    for key, t in sd.items():
        rs = _rs(seed, key)
        shape = tuple(t.shape)
        if key.endswith('num_batches_tracked'):
            v = np.zeros(shape, dtype=np.int64)
        elif key.endswith('running_var'):
            v = rs.uniform(0.5, 1.5, size=shape)
        elif key.endswith('running_mean'):
            v = rs.normal(0.0, 0.1, size=shape)
        elif t.dim() == 1 and key.endswith('weight'):      # norm scale
            v = 1.0 + 0.1 * rs.normal(size=shape)
        elif t.dim() == 1:                                   # bias / norm shift
            v = 0.05 * rs.normal(size=shape)
        else:                                                # conv / linear weight: He-style scale
            fan_in = int(np.prod(shape[1:]))
            v = rs.normal(size=shape) * np.sqrt(2.0 / fan_in)
        out[key] = torch.from_numpy(np.asarray(v)).to(t.dtype)
    return out


def seeded_tensor(shape, seed, lo=-1.0, hi=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(lo, hi, size=shape).astype(np.float32))


def seeded_normal(shape, seed, std=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.normal(size=shape) * std).astype(np.float32))


def damp_generator_head(sd, factor=0.05):
    """Step fixtures only: shrink the generator's last conv so its tanh output stays in the linear range.
    With He-scaled random weights the fake image saturates to exactly +-1 over large areas; the encoder's
    ReLU masks and max-pool arg-max then sit on exact ties and the step's gradients become a discontinuous
    function of 1e-6-level rounding differences -- a property of the random fixture, not of any kernel."""
    sd = dict(sd)
    sd['model.26.weight'] = sd['model.26.weight'] * factor
    return sd
