import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'allow_nonfinite: the test feeds inf / NaN (or a stale operand maximum) to an fp16-route kernel on purpose')


def pytest_collection_modifyitems(config, items):
    # GPU tests must never silently pass on a box without a GPU: they are deselected by
    # `-m "not gpu"` on CPU; if somebody runs them anyway without a device they fail loudly
    # inside the HIP path (no fallback), which is the intended behaviour.
    pass


@pytest.fixture(scope='session')
def dev():
    import torch
    assert torch.cuda.is_available(), 'GPU test selected but no GPU is visible'
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _fp16_route_never_overflows(request):
    """After EVERY GPU test: no fp16-route convolution may have produced inf / NaN (hip/ops.py: the non-finite sentinel) -- a
    stale operand maximum would overflow an fp16 piece and poison its results silently otherwise."""
    yield
    if request.node.get_closest_marker('gpu') is None:
        return
    import torch
    if not torch.cuda.is_available():
        return
    from pcgan_amd.hip import ops
    n = ops.nonfinite_count()
    if request.node.get_closest_marker('allow_nonfinite') is None:
        assert n == 0, '%d wave(s) of fp16-route kernels produced inf / NaN during this test' % n
