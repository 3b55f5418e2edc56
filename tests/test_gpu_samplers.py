"""GPU parity of the samplers (reference models/wsgan_emb_model.py:279-298): `sample_from_prior()` and
`sample_from_label(l)` of the HIP model against (a) the vectors captured from the reference's own methods
(tests/golden/samplers.npz) and (b) the oracle run side by side, for the default, noisy and MC-dropout encoders (the
oracle's Dropout2d keep masks are replayed: a GPU RNG stream cannot match a CPU one).  Tolerance: images 2e-4 of the
largest magnitude, ratings 2e-4; running-statistics checksums 1e-3."""
import os

import numpy as np
import pytest
import torch

from oracle import networks_ref as N
from oracle.make_golden import SAMPLER_VARIANTS, step_batch
from test_gpu_step import build_hip_model
from test_samplers_oracle_golden import BINS, run_oracle_samplers
from util_cmp import assert_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize('variant', SAMPLER_VARIANTS)
def test_samplers_match_reference_and_oracle(variant, tmp_path, dev):
    from pcgan_amd.hip import nn as hnn
    gold = np.load(os.path.join(GOLD, 'samplers.npz'))
    N.Dropout2dRec.record = []
    try:
        oracle, want = run_oracle_samplers(variant)
    finally:
        masks, N.Dropout2dRec.record = N.Dropout2dRec.record, None
    model, opt = build_hip_model(variant, tmp_path)
    assert list(model.embedding_bins) == BINS
    hnn.Dropout2d.mask_source = iter(masks) if masks else None
    try:
        torch.manual_seed(1234)
        model.set_input(step_batch(variant, 0))
        model.forward()
        with torch.no_grad():
            got = {'prior': model.sample_from_prior(), 'embedding_B': model.embedding_B}
            for label in range(3):
                got['label%d' % label] = model.sample_from_label(label)
        torch.cuda.synchronize()
        if masks:
            assert next(hnn.Dropout2d.mask_source, None) is None, 'the HIP samplers consumed fewer dropout masks than the oracle drew'
    finally:
        hnn.Dropout2d.mask_source = None
    for k, v in got.items():
        assert_close(v, want[k], 2e-4, '%s %s vs oracle' % (variant, k))
        assert_close(v, torch.from_numpy(gold['%s/%s' % (variant, k)]), 2e-4, '%s %s vs reference golden' % (variant, k))
    assert tuple(got['prior'].shape) == (4, 3, 32, 32)
    assert not got['label0'].requires_grad
    for tag, net in (('G', model.netG), ('E', model.netE)):
        for k, v in net.state_dict().items():
            if 'running' in k or 'num_batches' in k:
                ref = gold['%s/after%s/%s' % (variant, tag, k)]
                a = v.double()
                assert abs(float(a.sum()) - ref[0]) <= 1e-3 * (ref[1] + 1.0), '%s after the samplers: %s %s' % (variant, tag, k)
