import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cap_cpu_threads():
    """the oracle side of every parity test is stock PyTorch on the CPU: on a GPU box torch defaults to all 128 hardware threads of
    the host while the container's share is ~16 -- the oversubscribed pool made the oracle steps several times slower (round 4: the
    30-step trajectory test took 4:44 min, most of it in the CPU oracle).  Cap the pool at the usable CPUs (at most 16)."""
    try:
        import torch
        sys.path.insert(0, ROOT)
        import bench
        n, _ = bench.host_cpu_budget()
        torch.set_num_threads(max(1, min(16, n)))
    except Exception:      # noqa: BLE001  (a convenience, never a reason to fail collection)
        pass


def pytest_configure(config):
    _cap_cpu_threads()
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
    stale operand maximum would overflow an fp16 piece and poison its results silently otherwise -- and no audited operand maximum
    may differ from the tensor it is attached to (round 4: the too-LARGE direction loses precision without any inf)."""
    gpu = request.node.get_closest_marker('gpu') is not None
    if gpu:
        import torch
        if torch.cuda.is_available():
            from pcgan_amd.hip import ops
            if 'PCGAN_AMAX_AUDIT' not in os.environ:
                ops.AMAX_AUDIT_EVERY = 1        # the suite audits EVERY consumption of attached operand maxima (production: every 64th)
    yield
    if not gpu:
        return
    import torch
    if not torch.cuda.is_available():
        return
    from pcgan_amd.hip import ops
    n = ops.nonfinite_count()
    stale = ops.stale_maxima_count()
    if request.node.get_closest_marker('allow_nonfinite') is None:
        assert n == 0, '%d wave(s) of fp16-route kernels produced inf / NaN during this test' % n
        assert stale == (0, 0, 0), 'operand maxima that no longer describe their tensor (too small, too large, other): %r' % (stale,)
