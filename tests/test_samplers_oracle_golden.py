"""SURVEY 8(f) rank 4, first half: the samplers `sample_from_prior` / `sample_from_label` of wsgan_emb (reference
models/wsgan_emb_model.py:279-298).  Pins the ORACLE's restatement (oracle/step_ref.py) to vectors captured from the
reference's own methods (`oracle/make_golden.py --only samplers` -> tests/golden/samplers.npz): default, noisy and
MC-dropout (bayesian) encoders.  CPU only; same torch and same RNG stream => float32 rounding (1e-5)."""
import os

import numpy as np
import pytest
import torch

from oracle.make_golden import SAMPLER_VARIANTS
from test_oracle_golden import build_oracle_step, oracle_set_input
from util_cmp import assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
BINS = [-1.0, 0.0, 1.5]


def run_oracle_samplers(variant):
    m = build_oracle_step(variant)
    torch.manual_seed(1234)
    oracle_set_input(m, variant, 0)
    m.forward()
    torch.manual_seed(777)
    out = {}
    with torch.no_grad():
        out['prior'] = m.sample_from_prior()
        out['embedding_B'] = m.embedding_B
        for label in range(3):
            out['label%d' % label] = m.sample_from_label(label, BINS)
    return m, out


@pytest.mark.parametrize('variant', SAMPLER_VARIANTS)
def test_oracle_samplers_match_reference(variant):
    torch.set_num_threads(4)
    gold = np.load(os.path.join(GOLD, 'samplers.npz'))
    m, out = run_oracle_samplers(variant)
    for k, v in out.items():
        assert_close(v, torch.from_numpy(gold['%s/%s' % (variant, k)]), 2e-5, '%s %s' % (variant, k))
    assert tuple(out['prior'].shape) == (4, 3, 32, 32) and tuple(out['label0'].shape) == (4, 3, 32, 32)
    # the passes ran in train mode: G's InstanceNorm and E's BatchNorm running statistics moved exactly as the reference's
    for tag, net in (('G', m.netG), ('E', m.netE)):
        for k, v in net.state_dict().items():
            if 'running' in k or 'num_batches' in k:
                ref = gold['%s/after%s/%s' % (variant, tag, k)]
                a = v.double()
                assert abs(float(a.sum()) - ref[0]) <= 1e-5 * (ref[1] + 1.0), '%s after the samplers: %s %s' % (variant, tag, k)
