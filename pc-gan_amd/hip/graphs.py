"""hipGraph replay of a frozen network's forward under no_grad.

The step calls the frozen Elo encoder on the two real image sets without building an autograd graph -- twice per step by default, 20
times per step with MC dropout (config 4: T = 10 passes per image set at batch 8).  Alone, such a pass is bound by the host's launch
rate, not by the GPU: 1.40 ms eager against 0.76 ms as a graph at batch 8, 3.49 against 1.42 ms at batch 32
(`scripts/graph_e_probe.py`).  `GraphedNoGrad(fn)` runs fn eagerly for its first calls of an input shape (allocator, packed weights,
launch plans settle), captures the next call with torch.cuda.graph -- the captured launch sequence is the eager one, kernel for
kernel, including the in-place BatchNorm running-statistics updates of a train-mode net -- and replays it afterwards.  The C-ABI
makes that possible: no hidden allocation, no host synchronisation, device-side scalars.

Dropout2d inside a captured net: the keep flags of every site are slices of ONE static buffer that is refilled by a single
`bernoulli_` launch in front of each replay (independent draws per pass, as in eager mode).

Not used while autograd would record the call (an input or a parameter requires gradients), while a parity test injects masks (`Dropout2d.mask_source`), under torch.distributed, or with
PCGAN_GRAPH_NOGRAD=0.
A capture is tied to the packed-weight epoch, to the train / eval mode of every sub-module and to the versions of the parameters and
buffers it reads: anything that re-packs every weight (load_networks, broadcast), `net.eval()` / `net.train()`, or a direct
`load_state_dict` / `load_pretrained` starts over with eager calls instead of replaying the other mode's launches or stale packed
weights.  A capture that fails (out of the flag arena, an op that cannot be captured) marks its key "eager only": the call is served by
fn(x) and never retried."""
import os

import torch

from . import nn as hnn
from . import ops
from . import parallel

ENABLED = os.environ.get('PCGAN_GRAPH_NOGRAD', '1') != '0'
STATS = {'eager': 0, 'captured': 0, 'replayed': 0}


def _map(out, f):
    if isinstance(out, torch.Tensor):
        return f(out)
    if isinstance(out, (tuple, list)):
        return type(out)(_map(o, f) for o in out)
    return out


class FlagArena(object):
    """static keep-flag storage of the Dropout2d sites of one captured forward"""

    def __init__(self, device, size):
        self.buf = torch.ones(max(int(size), 1), dtype=torch.float32, device=device)
        self.used = 0
        self.p = None

    def take(self, n, p):
        assert self.p is None or self.p == p, 'one dropout rate per captured net'
        self.p = p
        o = self.used
        self.used = o + n
        if self.used > self.buf.numel():     # (sized from the warm-up calls: only a net whose dropout sites change between calls gets here)
            raise RuntimeError('pcgan_amd: dropout flag arena too small (%d > %d)' % (self.used, self.buf.numel()))
        return self.buf[o:o + n]

    def refill(self):
        if self.used:
            self.buf[:self.used].bernoulli_(1.0 - self.p)


def _module_stamp(fn):
    """(train / eval pattern, parameter versions) of a module: what a captured launch sequence silently depends on.  Buffers are
    left out on purpose: a train-mode pass updates its own running statistics in place, through raw pointers, inside the replay."""
    if not isinstance(fn, torch.nn.Module):
        return None
    mode = 0
    for m in fn.modules():
        mode = (mode * 3 + (1 if m.training else 2)) % 1000000007
    ver = 0
    for p_ in fn.parameters():      # (tensor version: load_state_dict / copy_; _pcgan_wepoch: an optimizer writing through raw pointers)
        ver += p_._version + p_.__dict__.get('_pcgan_wepoch', 0) + (p_.data_ptr() & 0xffff)
    return mode, ver


class GraphedNoGrad(object):
    def __init__(self, fn, warm=2):
        self.fn = fn
        self.warm = warm
        self.state = {}

    def _records_nothing(self, x):
        """no autograd graph would be built for this call: grad mode off, or neither the input nor any parameter asks for gradients
        (the frozen encoder on real images inside backward_G: config 4's ten MC-dropout passes of the z_rec term)"""
        if not torch.is_grad_enabled():
            return True
        if x.requires_grad:
            return False
        return not (isinstance(self.fn, torch.nn.Module) and any(p.requires_grad for p in self.fn.parameters()))

    def __call__(self, x):
        if (not ENABLED or not (isinstance(x, torch.Tensor) and x.is_cuda) or not self._records_nothing(x)
                or hnn.Dropout2d.mask_source is not None or torch.cuda.is_current_stream_capturing() or parallel.is_distributed()):
            return self.fn(x)      # (under torch.distributed: RCCL's watchdog thread and stream capture do not mix; eager there)
        with torch.no_grad():
            return self._call_no_grad(x)

    def _call_no_grad(self, x):
        key = (tuple(x.shape), x.dtype, ops._PACK_EPOCH[0], ops.BF16X6, ops.HSPLIT, ops.HGEMM, ops.BSPLIT_MIN_PIXELS, _module_stamp(self.fn))
        ent = self.state.get(key)
        if ent is None:
            if len(self.state) > 8:      # shapes / modes keep changing (a last partial batch, ...): do not hoard graph pools
                self.state.clear()
            ent = self.state[key] = {'calls': 0, 'flags': 0}
        if ent.get('eager_only'):
            STATS['eager'] += 1
            return self.fn(x)
        if 'graph' not in ent:
            if ent['calls'] < self.warm:
                ent['calls'] += 1
                STATS['eager'] += 1
                hnn.Dropout2d.flag_demand = demand = [0]      # how many keep flags one pass draws (sizes the capture's arena)
                try:
                    return self.fn(x)
                finally:
                    hnn.Dropout2d.flag_demand = None
                    ent['flags'] = max(ent['flags'], demand[0])
            inp = x.clone()
            arena = FlagArena(x.device, ent['flags'] if ent['calls'] > 0 else 1 << 17)      # (no warm-up call measured the demand: a generous default)
            g = torch.cuda.CUDAGraph()
            hnn.Dropout2d.flag_arena = arena
            try:
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    out = self.fn(inp)
            except Exception as exc:      # never abort training over an optimisation: this key stays eager
                ent['eager_only'] = True
                STATS['capture_failed'] = STATS.get('capture_failed', 0) + 1
                import warnings
                warnings.warn('pcgan_amd: hipGraph capture of a no-grad pass failed (%s); this shape runs eagerly' % (exc,))
                hnn.Dropout2d.flag_arena = None
                torch.cuda.synchronize()
                return self.fn(x)
            finally:
                hnn.Dropout2d.flag_arena = None
            ent.update(graph=g, inp=inp, out=out, arena=arena)
            STATS['captured'] += 1
        else:
            ent['inp'].copy_(x)
        ent['arena'].refill()
        ent['graph'].replay()
        STATS['replayed'] += 1
        return _map(ent['out'], lambda t: t.clone())      # the graph's output buffers are overwritten by the next replay
