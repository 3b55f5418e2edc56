"""Tensor-level wrappers over the C-ABI: validate, allocate outputs/workspace through
torch's caching allocator (device memory plumbing only) and launch on torch's current
stream.  No arithmetic happens in Python or in torch here.
"""
import ctypes
import os

import torch

from . import lib as _L
from .lib import ConvDesc, ResBlockDesc, ACT_NONE, F32, BF16

_vp = ctypes.c_void_p


def _p(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


_DEVICE_INDEX = []


def _raw_stream():
    """the current HIP stream handle as an integer (torch.cuda.current_stream() builds a Stream object: ~9 us, 650 calls per step)"""
    if not _DEVICE_INDEX:
        _DEVICE_INDEX.append(torch.cuda.current_device())     # one process drives one GPU
    return torch._C._cuda_getCurrentRawStream(_DEVICE_INDEX[0])


def _stream():
    return _vp(_raw_stream())


def _chk(*tensors):
    """fp32 tensors (parameters, statistics, per-sample targets ...): contiguous, on the GPU"""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('pcgan_amd: tensor on %s -- the HIP path needs GPU tensors (no CPU fallback)' % t.device)
        if t.dtype != torch.float32:
            raise RuntimeError('pcgan_amd: expected float32, got %s' % t.dtype)
        if not t.is_contiguous():
            raise RuntimeError('pcgan_amd: tensor must be contiguous')


_DTYPES = {torch.float32: F32, torch.bfloat16: BF16}


def _act(*tensors):
    """ACTIVATION tensors (and their gradients): fp32 or bf16 storage, all of one type; returns the C-ABI dtype code"""
    code = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('pcgan_amd: tensor on %s -- the HIP path needs GPU tensors (no CPU fallback)' % t.device)
        c = _DTYPES.get(t.dtype)
        if c is None:
            raise RuntimeError('pcgan_amd: activation tensors are float32 or bfloat16, got %s' % t.dtype)
        if code is not None and c != code:
            raise RuntimeError('pcgan_amd: activation tensors of one call must share a storage type (float32 and bfloat16 mixed)')
        code = c
        if not t.is_contiguous():
            raise RuntimeError('pcgan_amd: tensor must be contiguous')
    return F32 if code is None else code


def cast(x, dtype):
    """storage cast fp32 <-> bf16 (round to nearest even): the bf16 path's boundary"""
    src = _act(x)
    if x.dtype == dtype:
        return x
    y = torch.empty_like(x, dtype=dtype)
    _L.check(_L.load().pcgan_cast(_p(x), src, _p(y), _DTYPES[dtype], x.numel(), _stream()), 'cast')
    return y


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def conv_out_size(H, k, stride, pad):
    return (H + 2 * pad - k) // stride + 1


def make_desc(N, C, H, W, K, R, S, stride, pad, pad_mode, dtype=F32):
    return ConvDesc(N, C, H, W, K, R, S, stride, pad, pad_mode,
                    conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad), dtype)


# ---------------------------------------------------------------- side stream for parameter gradients
# In a backward pass the data gradient of a layer feeds the next layer while its weight / bias gradients feed
# nobody until the optimizer runs.  They are launched on a second HIP stream, so the hardware overlaps them
# with the data-gradient chain (fills each kernel's ramp-up / tail and the small launches in between).  The
# main stream joins at the end of the backward pass (autograd callback) -- results and summation order are
# unchanged (all parameter-gradient kernels run in issue order on the one side stream).
SIDE_STREAM = os.environ.get('PCGAN_SIDE_STREAM', '1') != '0'
_side = {}
_side_state = {'dirty': False, 'queued': False}


class fork_side(object):
    """with fork_side(t1, t2, ...): launches inside go to the side stream, after everything already queued on
    the current stream; the tensors are kept alive until the side stream is done with them."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        cur = torch.cuda.current_stream()
        dev = cur.device
        st = _side.get(dev)
        if st is None:
            st = _side[dev] = torch.cuda.Stream(device=dev)
        st.wait_stream(cur)
        for t in self.tensors:
            t.record_stream(st)
        self.ctx = torch.cuda.stream(st)
        self.ctx.__enter__()
        _side_state['dirty'] = True
        if not _side_state['queued']:
            try:    # join automatically when the running backward pass ends
                torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
                _side_state['queued'] = True
            except Exception:      # not inside a backward pass (or no such hook): FusedAdam.step / zero_grad join
                pass
        return st

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def side_stream_for(cur):
    """the parameter-gradient stream of `cur`'s device (made on first use)"""
    st = _side.get(cur.device)
    if st is None:
        st = _side[cur.device] = torch.cuda.Stream(device=cur.device)
    return st


def mark_side_used():
    """work was queued on the side stream by a composite call (which forks inside the library): join when the backward pass ends"""
    _side_state['dirty'] = True
    if not _side_state['queued']:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
            _side_state['queued'] = True
        except Exception:
            pass


# ---------------------------------------------------------------- branch streams (independent sub-networks)
# In backward_G the discriminator, the AlexNet identity branch and the Elo encoder all read the same fake image and
# meet again only in the scalar loss sum.  Each branch can run on its own stream: autograd replays every backward node
# on the stream of its forward op and synchronises where the gradients meet, so forward AND backward of the branches
# overlap (their mid-size kernels and launch-latency-bound BatchNorm chains fill each other's gaps).  Different nets
# only -- two passes through ONE BatchNorm net must stay ordered (running statistics).
BRANCH_STREAMS = os.environ.get('PCGAN_BRANCH_STREAMS', '1') != '0'
_branch = {}
# Which branch NAMES share one stream.  HIP maps streams onto a few hardware queues (4 by default) and streams on one queue serialise;
# which streams end up together used to follow from their creation order -- round 4 measured the same step at 1062 / 1210 / 750 img/s
# for three such accidents (profiles/r04_experiments.txt).  The mapping is now stated: the generator's two passes are a dependency
# chain (rec_A = G(fake_B)) and share the stream 'G'; with the main stream, the encoder's stream 'E' (which also carries the batch
# uploads) and the parameter-gradient stream that makes four streams for four queues.  PCGAN_STREAM_ALIAS="name:target,..." re-maps
# for A/B runs (target `main` = run that branch on the current stream).
STREAM_ALIAS = {'G1': 'G', 'G2': 'G'}
for _kv in os.environ.get('PCGAN_STREAM_ALIAS', '').split(','):
    if ':' in _kv:
        STREAM_ALIAS[_kv.split(':')[0].strip()] = _kv.split(':')[1].strip()


class branch(object):
    """with branch('E') as b: ... loss = f(...) ; b.join(loss): run the body on the named stream after everything
    queued so far on the current stream; join() makes the current stream wait and hands the tensors over.
    after = a list of events / streams: the body only waits for THOSE (the readiness of its inputs, `ready_event`) instead of for the whole
    current stream -- work that depends on nothing the current stream still has queued (the frozen encoder on the next batch) then
    runs beside the tail of the previous step."""

    def __init__(self, name, enabled=True, after=None):
        self.name = STREAM_ALIAS.get(name, name)
        self.after = after
        self.on = (enabled and BRANCH_STREAMS and torch.cuda.is_available() and self.name != 'main'
                   and name not in os.environ.get('PCGAN_BRANCH_OFF', '').split(','))

    def __enter__(self):
        if self.on:
            self.cur = torch.cuda.current_stream()
            key = (self.cur.device, self.name)
            st = _branch.get(key)
            if st is None:
                st = _branch[key] = torch.cuda.Stream(device=self.cur.device)
            if self.after is None:
                st.wait_stream(self.cur)
            else:
                for ev in self.after:      # events, or whole streams (another branch whose results the body reads)
                    if isinstance(ev, torch.cuda.Stream):
                        st.wait_stream(ev)
                    else:
                        st.wait_event(ev)
            self.st = st
            self.ctx = torch.cuda.stream(st)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            return self.ctx.__exit__(*exc)

    def reads(self, *tensors):
        """tensors allocated on other streams that the body reads: the caching allocator must not hand their memory out again (after
        the owner drops them while the host runs ahead) before this stream is done with them"""
        if self.on:
            for t in tensors:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(self.st)

    def join(self, *tensors):
        if self.on:
            self.cur.wait_stream(self.st)
            for t in tensors:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(self.cur)


def upload_stream(device):
    """the stream input batches are copied / cast on (base_model.to_act): not ordered behind the step still running on the main stream.
    It IS the encoder's branch stream ('E'): the frozen encoder's passes over the new batch are the first consumers of an upload and wait
    for nothing else, and a stream of its own is not free on this runtime -- HIP streams are mapped onto a few hardware queues (4 by
    default), streams that share a queue serialise, and WHICH streams share one follows from the set of streams in use: round 4
    measured the same step at 1200 img/s without a dedicated upload stream and 1062 img/s with one (the generator's ahead-of-step pass
    and the parameter-gradient stream then shared a queue), and 750 img/s with GPU_MAX_HW_QUEUES=8 (everything concurrent: the matrix
    kernels of four streams thrash the chip) -- profiles/r04_experiments.txt."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (device, STREAM_ALIAS.get('E', 'E'))
    st = _branch.get(key)
    if st is None:
        st = _branch[key] = torch.cuda.Stream(device=device)
    return st


def mark_ready(t, stream=None):
    """PRODUCER side: declare that device tensor `t` holds its contents once everything queued so far on `stream` (default: the current
    stream) has run -- to_act's upload, the loader's --gpu_transform output, a custom loader that fills a device buffer.  Whoever
    rewrites the buffer afterwards (also through a raw pointer or `.data`, which do not bump the tensor version) calls this again."""
    ev = torch.cuda.Event()
    ev.record(stream if stream is not None else torch.cuda.current_stream(t.device))
    t._pcgan_ready = (t._version, ev)
    return ev


def ready_event(t):
    """CONSUMER side: an event after which the device tensor `t` holds its current contents.  The producer's event when it attached one
    (mark_ready) and the tensor version is still the one it was attached at; otherwise an event recorded NOW on the current stream,
    i.e. "after everything queued so far" -- the plain stream order, never cached.  (Round 3 recorded an event at first sight and kept
    it while the version stayed the same: a buffer rewritten behind the version counter kept a stale "ready")"""
    ent = t.__dict__.get('_pcgan_ready')
    if ent is not None and ent[0] == t._version:
        return ent[1]
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))
    return ev


_DEFER_JOIN = [0]


class defer_side_join(object):
    """with defer_side_join(): the automatic join at the end of a backward pass is skipped -- the parameter-gradient stream keeps
    running behind the main stream and whoever consumes the gradients orders itself explicitly (FusedAdam.step_on_grad_stream: the
    update is queued ON that stream, behind the weight-gradient kernels, and leaves an event).  Without it the main stream idles at the
    end of every backward pass until the last weight gradient has finished (the side stream is the longer chain of backward_G)."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        if self.enabled:
            _DEFER_JOIN[0] += 1
        return self

    def __exit__(self, *exc):
        if self.enabled:
            _DEFER_JOIN[0] -= 1


def join_side_stream(force=False):
    """Make the current stream wait for everything launched on the side stream."""
    _side_state['queued'] = False
    if _DEFER_JOIN[0] and not force:
        return
    if _side_state['dirty']:
        for st in _side.values():
            torch.cuda.current_stream(st.device).wait_stream(st)
        _side_state['dirty'] = False


# ---------------------------------------------------------------- convolution family
# Packed-weight cache.  The implicit-GEMM kernels read the weights re-tiled ("packed"); a step runs every
# net several times between two optimizer updates, so the packed copy is kept per (weight tensor, pass) and
# re-made when the weights may have changed: the stamp is (global epoch, tensor version, storage address).
# The epoch is bumped by everything that writes parameters behind autograd's back (the Adam kernels, which
# update the flat buffer through raw pointers; model.set_input / load_networks / broadcast as a backstop).
_PACK_EPOCH = [0]


def invalidate_packed_weights(params=None):
    """packed copies (and weight maxima) of `params` -- or, without a list, of every weight tensor -- are out of date"""
    if params is None:
        _PACK_EPOCH[0] += 1
        return
    for p in params:     # the optimizer that wrote them: nets it does not own (the frozen encoder, AlexNet) keep their packs
        p._pcgan_wepoch = p.__dict__.get('_pcgan_wepoch', 0) + 1


def _weight_stamp(w):
    return (_PACK_EPOCH[0], w.__dict__.get('_pcgan_wepoch', 0), w._version, w.data_ptr(), tuple(w.shape))


# The residual-block convolutions (forward, data gradient, weight gradient) run on the bf16 matrix pipe with every fp32
# operand split exactly into three bf16 pieces, six piece products per term (csrc/bf16x6_conv.hip): fp32-level error at
# 1.5x the speed of the fp32 MFMA kernels.  PCGAN_BF16X6=0 routes them back to the fp32 MFMA implicit GEMM (A/B runs).
BF16X6 = os.environ.get('PCGAN_BF16X6', '1') == '1'
BSPLIT_MIN_PIXELS = 16384    # output pixels (N*P*Q) from which the one-tile-shape split kernels fill the chip (tests lower it)
PASS_FWD_BSPLIT = 100    # cache keys only
PASS_BWD_BSPLIT = 101
PASS_FWD_HSPLIT = 102
PASS_BWD_HSPLIT = 103
PASS_FWD_HGEMM = 104     # pcgan_conv2d_hgemm_pack: the fp32 packed image with pre-split (two fp16 pieces) weights
PASS_BWD_HGEMM = 105
PASS_FWD_THIN = 106      # pcgan_conv2d_thin_pack: <= 4 gathered channels (7x7 stems, first PatchGAN layer; data gradient of the 64 -> 3 head)
PASS_BWD_THIN = 107
# Which split the fp32 residual convolutions (forward, data gradient) take on the matrix pipe: 'f16' = two scaled fp16 pieces, three
# products (csrc/bf16x6_conv.hip, "fp16 route"); 'bf16' = three bf16 pieces, six products.  Same measured error, half the MFMAs.
HSPLIT = os.environ.get('PCGAN_SPLIT', 'f16') == 'f16'
# ... and whether every other convolution with a multiple of 16 gathered channels runs the fp16 two-piece form of the packed
# implicit GEMM (csrc/igemm_conv.hip hgemm_kernel) instead of the fp32 MFMA one
HGEMM = os.environ.get('PCGAN_HGEMM', '1') == '1'
# ... and the convolutions that gather <= 4 channels the window kernel of csrc/thin_conv.hip instead of igemm2_kernel<.., 4> on fp32 MFMA
THIN = os.environ.get('PCGAN_THIN', '1') == '1'
THIN_WGRAD = os.environ.get('PCGAN_THIN_WGRAD', '1') == '1'     # weight gradients of the 3- / 4-channel layers on the matrix-pipe kernel too
WGRAD_INLINE_BIG = os.environ.get('PCGAN_WGRAD_INLINE_BIG', '1') != '0'
THIN_MASK = int(os.environ.get('PCGAN_THIN_MASK', '7'))     # debugging: 1 = stride-1 forward, 2 = stride-2 forward, 4 = data gradient


AMAX_STATS = {'attached': 0, 'computed': 0}     # operand maxima handed over by the producing kernel / taken by an absmax pass
ROUTE_STATS = {}     # (pass, route) -> number of convolution calls that took it: tests assert the kernels under test are the production ones


def _count_route(pass_, route):
    k = (pass_, route)
    ROUTE_STATS[k] = ROUTE_STATS.get(k, 0) + 1


def _want_maxima(dt, C):
    """does a norm kernel leave the plane / channel maxima of its output for the fp16 route of the next convolution?
    (fp32 tensors whose channel count a matrix-pipe convolution can gather: a multiple of 16)"""
    return HSPLIT and dt == F32 and C % 16 == 0


def _attach_amax(t, pmax, event=None):
    """maxima written by the kernel that wrote `t` itself need no event: whoever may read `t` is already ordered behind that kernel.
    Maxima taken by a SEPARATE pass (amax_of below) carry the event recorded behind that pass."""
    t._pcgan_amax = (t._version, pmax, _raw_stream(), event)


def amax_of(x):
    """fp32 device tensor of partial maxima of |x| (the consumer takes the largest): the per-plane maxima its producer attached
    (`_pcgan_amax`, valid for the tensor version they were attached at) or the single value of one pcgan_absmax pass."""
    ent = x.__dict__.get('_pcgan_amax')
    if ent is not None and ent[0] == x._version:
        AMAX_STATS['attached'] += 1
        if ent[2] != _raw_stream():      # consumed on another stream than it was produced on
            cur = torch.cuda.current_stream()
            if len(ent) > 3 and ent[3] is not None:
                # taken by an absmax pass on that other stream (a weight gradient on the parameter-gradient stream computes the maxima
                # of an un-annotated dy, the data gradient on the main stream finds them attached): being ordered behind the producer
                # of x says nothing about that pass -- wait for it
                cur.wait_event(ent[3])
            ent[1].record_stream(cur)
        if AMAX_AUDIT_EVERY and AMAX_STATS['attached'] % AMAX_AUDIT_EVERY == 0:
            _audit_amax(x, ent[1])       # (after the wait: the audit reads the claim on THIS stream -- it caught exactly that race in its first run)
        return ent[1]
    AMAX_STATS['computed'] += 1
    lib = _L.load()
    slots = int(lib.pcgan_absmax_slots(x.numel()))
    out = torch.empty(slots, dtype=torch.float32, device=x.device)
    _L.check(lib.pcgan_absmax(_p(x), x.numel(), _DTYPES[x.dtype], _p(out), slots, _stream()), 'absmax')
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    _attach_amax(x, out, ev)      # a second consumer of the same tensor version (forward and weight gradient of one layer) reuses the pass
    return out


# Audit of attached maxima (round 4, VERDICT r3 "weak" 2).  `_pcgan_amax` is trusted on the tensor version, which writes through `.data`
# or a raw pointer do not bump.  A stale claim that is too SMALL overflows fp16 and the non-finite sentinel reports it; one that is too
# LARGE only loses precision, silently.  Whether a claim is too large cannot be decided from a tile (a sparse tile looks the same): it
# takes the largest value of the WHOLE tensor -- so every AMAX_AUDIT_EVERY-th consumption of attached maxima re-takes them with one
# pcgan_absmax pass (4 us per 33 MB) and pcgan_amax_audit compares on the device (no host synchronisation); check_nonfinite() raises on
# a non-zero count.  Default: every 64th (~5 audits per step: a code path that rewrites tensors behind the version counter is caught
# within a few steps); the GPU test suite audits EVERY consumption (tests/conftest.py); 0 = off.
AMAX_AUDIT_EVERY = int(os.environ.get('PCGAN_AMAX_AUDIT', '64') or 0)
AMAX_STATS['audited'] = 0


def _audit_amax(x, claimed):
    _sentinel()
    cnt = _SENTINEL.get('audit')
    if cnt is None or x.dtype != torch.float32:
        return
    lib = _L.load()
    slots = int(lib.pcgan_absmax_slots(x.numel()))
    fresh = torch.empty(slots, dtype=torch.float32, device=x.device)
    _L.check(lib.pcgan_absmax(_p(x), x.numel(), F32, _p(fresh), slots, _stream()), 'absmax')
    _L.check(lib.pcgan_amax_audit(_p(claimed), claimed.numel(), _p(fresh), slots, _vp(cnt.data_ptr()), _stream()), 'amax_audit')
    AMAX_STATS['audited'] += 1


def _packed_weights(lib, d, pass_, w, cache, want_rowmax=False):
    """the packed copy of w for one pass; want_rowmax (hgemm route): (packed, rowmax) -- the pre-split pack also writes the largest
    magnitude of every weight row, which the convolution's epilogue needs"""
    key = (pass_, d.stride, d.pad, d.pad_mode, d.dtype)
    stamp = _weight_stamp(w)
    ent = cache.get(key)
    if ent is not None and ent[0] == stamp:
        if ent[3] != _raw_stream():      # packed on another stream (branch streams): order this use after the pack
            torch.cuda.current_stream().wait_event(ent[2])
        return (ent[1], ent[4]) if want_rowmax else ent[1]
    cur = torch.cuda.current_stream()
    if pass_ in (PASS_FWD_HSPLIT, PASS_BWD_HSPLIT):
        nb = int(lib.pcgan_conv2d_hsplit_packed_bytes(ctypes.byref(d), _L.PASS_FWD if pass_ == PASS_FWD_HSPLIT else _L.PASS_BWD_DATA))
    elif pass_ == PASS_FWD_BSPLIT:
        nb = int(lib.pcgan_conv2d_bsplit_packed_bytes(ctypes.byref(d)))
    elif pass_ == PASS_BWD_BSPLIT:
        nb = int(lib.pcgan_conv2d_bsplit_dgrad_packed_bytes(ctypes.byref(d)))
    elif pass_ in (PASS_FWD_HGEMM, PASS_BWD_HGEMM):
        nb = int(lib.pcgan_conv2d_packed_bytes(ctypes.byref(d), _L.PASS_FWD if pass_ == PASS_FWD_HGEMM else _L.PASS_BWD_DATA))
    elif pass_ in (PASS_FWD_THIN, PASS_BWD_THIN):
        nb = int(lib.pcgan_conv2d_thin_packed_bytes(ctypes.byref(d), _L.PASS_FWD if pass_ == PASS_FWD_THIN else _L.PASS_BWD_DATA))
    else:
        nb = int(lib.pcgan_conv2d_packed_bytes(ctypes.byref(d), pass_))
    nb = max(nb, 256)
    if ent is not None and ent[1].numel() == nb and ent[1].device == w.device:
        buf = ent[1]
        if ent[3] != cur.cuda_stream:      # re-pack into a buffer another stream may still be reading
            for st in list(_side.values()) + list(_branch.values()):
                cur.wait_stream(st)
    else:
        buf = _ws(nb, w.device)
    rowmax = None
    if pass_ in (PASS_FWD_HGEMM, PASS_BWD_HGEMM):
        rows = d.K if pass_ == PASS_FWD_HGEMM else d.C
        rowmax = ent[4] if (ent is not None and len(ent) > 4 and ent[4] is not None and ent[4].numel() == rows and buf is ent[1]) else \
            torch.empty(rows, dtype=torch.float32, device=w.device)
    if pass_ in (PASS_FWD_HSPLIT, PASS_BWD_HSPLIT):
        _L.check(lib.pcgan_conv2d_hsplit_pack(ctypes.byref(d), _L.PASS_FWD if pass_ == PASS_FWD_HSPLIT else _L.PASS_BWD_DATA, _p(w), _p(buf),
                                              _stream()), 'conv2d_hsplit_pack')
    elif pass_ == PASS_FWD_BSPLIT:
        _L.check(lib.pcgan_conv2d_bsplit_pack(ctypes.byref(d), _p(w), _p(buf), _stream()), 'conv2d_bsplit_pack')
    elif pass_ == PASS_BWD_BSPLIT:
        _L.check(lib.pcgan_conv2d_bsplit_dgrad_pack(ctypes.byref(d), _p(w), _p(buf), _stream()), 'conv2d_bsplit_dgrad_pack')
    elif pass_ in (PASS_FWD_THIN, PASS_BWD_THIN):
        _L.check(lib.pcgan_conv2d_thin_pack(ctypes.byref(d), _L.PASS_FWD if pass_ == PASS_FWD_THIN else _L.PASS_BWD_DATA, _p(w), _p(buf),
                                            _stream()), 'conv2d_thin_pack')
    elif pass_ in (PASS_FWD_HGEMM, PASS_BWD_HGEMM):
        _L.check(lib.pcgan_conv2d_hgemm_pack(ctypes.byref(d), _L.PASS_FWD if pass_ == PASS_FWD_HGEMM else _L.PASS_BWD_DATA, _p(w), _p(rowmax),
                                             _p(buf), _stream()), 'conv2d_hgemm_pack')
    else:
        _L.check(lib.pcgan_conv2d_pack_weights(ctypes.byref(d), pass_, _p(w), _p(buf), _stream()), 'conv2d_pack_weights')
    ev = torch.cuda.Event()
    ev.record(cur)
    cache[key] = (stamp, buf, ev, cur.cuda_stream, rowmax)
    return (buf, rowmax) if want_rowmax else buf


# Per-shape launch plans: descriptor, workspace size and route of a convolution call are functions of the shape and of the
# routing switches only -- worked out once (4-5 ctypes queries, ~10 us) and looked up afterwards (the step makes ~330 such calls).
_PLANS = {}


def clear_plans():
    """drop every cached launch plan (after lib.set_option changed what the library supports / routes)"""
    _PLANS.clear()
    _RB_PLANS.clear()


class _Plan(object):
    __slots__ = ('d', 'dref', 'P', 'Q', 'ws_bytes', 'route', 'pack_pass')


# Non-finite sentinel of the fp16 route (include/pcgan_hip.h: pcgan_set_nonfinite_counter): one device word the kernels count
# inf / NaN-producing waves into.  A stale operand maximum (a tensor rewritten through `.data` or a raw pointer after a norm kernel
# attached its maxima: `_pcgan_amax` is trusted on the tensor version) overflows fp16 and lands here instead of passing silently.
NONFINITE_CHECK = os.environ.get('PCGAN_NONFINITE_CHECK', '1') != '0'
_SENTINEL = {}


def _sentinel():
    if not NONFINITE_CHECK or 'word' in _SENTINEL or not torch.cuda.is_available():
        return
    _SENTINEL['word'] = torch.zeros(1, dtype=torch.int32, device='cuda')
    _SENTINEL['audit'] = torch.zeros(3, dtype=torch.int32, device='cuda')      # pcgan_amax_audit: [claim too small, too large by 2^8, other mismatch]
    _L.check(_L.load().pcgan_set_nonfinite_counter(_vp(_SENTINEL['word'].data_ptr())), 'set_nonfinite_counter')


def nonfinite_count(reset=True):
    """waves of fp16-route kernels that produced inf / NaN since the last reset (synchronises)"""
    w = _SENTINEL.get('word')
    if w is None:
        return 0
    n = int(w.item())
    if n and reset:
        w.zero_()
    return n


def stale_maxima_count(reset=True):
    """(claims too small, claims too large by >= 2^8, other mismatches) found by the audits since the last reset (synchronises)"""
    a = _SENTINEL.get('audit')
    if a is None:
        return (0, 0, 0)
    v = tuple(int(t) for t in a.tolist())
    if any(v) and reset:
        a.zero_()
    return v


def check_nonfinite(where=''):
    """raise if an fp16-route convolution produced non-finite values since the last check, or if an audit found operand maxima that no
    longer describe their tensor.  Called where the host synchronises anyway (BaseModel.get_current_losses, the GPU test suite after
    every test)."""
    small, large, other = stale_maxima_count()
    if small or large or other:
        raise RuntimeError('pcgan_amd: stale operand maxima%s: %d audited tensor(s) held MORE than the maximum attached to them (fp16 overflow), '
                           '%d held less than 2^-8 of it (the fp16 route silently loses >= 8 bits of every product), %d differed otherwise -- '
                           'a tensor was rewritten through .data / a raw pointer after a kernel attached its maxima; rewrite it through an '
                           'op, or drop `_pcgan_amax`' % ((' (' + where + ')') if where else '', small, large, other))
    n = nonfinite_count()
    if n:
        raise RuntimeError('pcgan_amd: %d wave(s) of the fp16-route convolutions produced inf / NaN%s -- an operand exceeded the maximum its '
                           'scale was derived from (a tensor rewritten through .data / a raw pointer after its maxima were attached?) or the '
                           'inputs were already non-finite' % (n, (' (' + where + ')') if where else ''))


# ---- the route table -------------------------------------------------------------------------------------------------------------------
# Which kernel family a convolution call takes is DECLARED here: per pass an ordered list of (route, packed-weight pass, condition,
# workspace query); the first entry whose condition holds wins.  A condition sees one context object `c`: the shape (N, C, H, W, K, R,
# S, stride, pad, pad_mode, P, Q, dt, no_bias), the routing switches as they stand, the library handle and the descriptor.  The
# measurements behind the order are in profiles/ (kernel dashboards, rNN_experiments.txt); the conditions that are pure shape limits of a
# kernel live in the library (`*_supported`).
class _RouteCtx(object):
    pass


def _ws_generic(pass_):
    return lambda c: int(c.lib.pcgan_conv2d_workspace_bytes(c.dref, pass_))


def _fills_chip(c):
    """one-tile-shape kernels without split-K (window / per-tap split kernels): only where whole tiles fill the chip"""
    return c.split and c.pixels >= BSPLIT_MIN_PIXELS


_FWD_ROUTES = (
    # residual-block convolutions, fp32 tensors: window kernel, two fp16 pieces
    ('hsplit', PASS_FWD_HSPLIT, lambda c: _fills_chip(c) and c.K % 128 == 0 and c.lib.pcgan_conv2d_bsplit_supported(c.dref) and c.f16
     and c.lib.pcgan_conv2d_hsplit_supported(c.dref, _L.PASS_FWD), _ws_generic(_L.PASS_FWD)),
    # every other fp32 convolution with a multiple of 16 gathered channels: packed implicit GEMM, two fp16 pieces (before the three-piece
    # kernel below: the encoder's 128-channel 28x28 layers were on its six products per term, 0.062 ms against 0.047)
    ('hgemm', PASS_FWD_HGEMM, lambda c: c.f16 and HGEMM and c.lib.pcgan_conv2d_hgemm_supported(c.dref, _L.PASS_FWD), _ws_generic(_L.PASS_FWD)),
    # three-piece bf16 split (PCGAN_SPLIT=bf16) / bf16 tensors: window or per-tap kernel
    ('bsplit', PASS_FWD_BSPLIT, lambda c: _fills_chip(c) and c.K % 128 == 0 and c.lib.pcgan_conv2d_bsplit_supported(c.dref), _ws_generic(_L.PASS_FWD)),
    # <= 4 gathered channels (7x7 stems, first PatchGAN layer)
    ('thin', PASS_FWD_THIN, lambda c: c.f16 and THIN and (THIN_MASK & (1 if c.stride == 1 else 2)) and c.lib.pcgan_conv2d_thin_supported(c.dref, _L.PASS_FWD),
     _ws_generic(_L.PASS_FWD)),
    ('packed', _L.PASS_FWD, lambda c: True, _ws_generic(_L.PASS_FWD)),       # fp32-MFMA implicit GEMM / small-M kernels
)

_DGRAD_ROUTES = (
    ('hsplit', PASS_BWD_HSPLIT, lambda c: c.split and c.no_bias and c.C % 128 == 0 and c.N * c.H * c.W >= BSPLIT_MIN_PIXELS
     and c.lib.pcgan_conv2d_bsplit_dgrad_supported(c.dref) and c.f16 and c.lib.pcgan_conv2d_hsplit_supported(c.dref, _L.PASS_BWD_DATA),
     _ws_generic(_L.PASS_BWD_DATA)),
    ('bsplit', PASS_BWD_BSPLIT, lambda c: c.split and c.no_bias and c.C % 128 == 0 and c.N * c.H * c.W >= BSPLIT_MIN_PIXELS
     and c.lib.pcgan_conv2d_bsplit_dgrad_supported(c.dref), _ws_generic(_L.PASS_BWD_DATA)),
    ('hgemm', PASS_BWD_HGEMM, lambda c: c.f16 and HGEMM and c.lib.pcgan_conv2d_hgemm_supported(c.dref, _L.PASS_BWD_DATA), _ws_generic(_L.PASS_BWD_DATA)),
    # the data gradient of the 64 -> 3 head as a forward-form convolution (no pack cache: the generic call; needs the padded-grid workspace)
    ('thin', PASS_BWD_THIN, lambda c: c.f16 and THIN and (THIN_MASK & 4) and c.no_bias and c.lib.pcgan_conv2d_thin_supported(c.dref, _L.PASS_BWD_DATA),
     lambda c: max(int(c.lib.pcgan_conv2d_workspace_bytes(c.dref, _L.PASS_BWD_DATA)), int(c.lib.pcgan_conv2d_thin_workspace_bytes(c.dref, _L.PASS_BWD_DATA)))),
    ('packed', _L.PASS_BWD_DATA, lambda c: True, _ws_generic(_L.PASS_BWD_DATA)),
)


def _wgrad_on_matrix_pipe(c):
    """fp16 / bf16 matrix-pipe weight gradient (hsplit_wgrad_kernel): the residual convolutions from the host's routing threshold on,
    every other eligible layer (stride 1 / 2, output width handled by the kernel) from 4096 output pixels on"""
    res_like = c.K == 256 and c.R == 3 and c.S == 3 and c.stride == 1 and c.pad_mode == 1
    # 3- / 4-channel inputs: only the generator's 7x7 stride-1 stem gains (0.266 -> 0.215 ms); the PatchGAN's first layer and the encoder's
    # strided stem are faster on the fp32-MFMA kernels (scripts/time_thin.py)
    thin_stem = c.C <= 4 and THIN_WGRAD and c.R == 7 and c.S == 7 and c.stride == 1
    # zero padding <= 1 is applied inside the gather (no padded copy): those layers go there whatever their size; a layer that needs the
    # padded copy only while it costs less than the matrix pipe saves (80 MB; PCGAN_WGRAD_INLINE_BIG=0 applies the limit to all, for A/B)
    no_copy = WGRAD_INLINE_BIG and bool(c.lib.pcgan_conv2d_hsplit_wgrad_inline(c.dref))
    cheap_pad = (c.C % 16 == 0 or thin_stem) and (no_copy or c.N * c.C * c.H * c.W * 4 <= 80 * 1000 * 1000)
    return bool(HSPLIT and BF16X6 and c.lib.pcgan_conv2d_hsplit_wgrad_supported(c.dref)
                and (c.pixels >= BSPLIT_MIN_PIXELS if (res_like or not HGEMM) else c.pixels >= min(BSPLIT_MIN_PIXELS, 4096))
                and (res_like or (HGEMM and cheap_pad))
                # (a ragged output width under a half-empty 128-row tile loses to the fp32 kernel: ResNet-18 layer1, 64 -> 64 at 56 x 56,
                # 0.092 vs 0.076 ms)
                and (c.K >= 128 or c.Q % 16 == 0))


_WGRAD_ROUTES = (
    ('hsplit', None, _wgrad_on_matrix_pipe, lambda c: int(c.lib.pcgan_conv2d_hsplit_wgrad_workspace_bytes(c.dref))),
    ('bsplit', None, lambda c: c.split and c.K in (128, 256) and c.N * c.H * c.W >= BSPLIT_MIN_PIXELS and c.lib.pcgan_conv2d_bsplit_wgrad_supported(c.dref),
     lambda c: int(c.lib.pcgan_conv2d_bsplit_wgrad_workspace_bytes(c.dref))),
    ('generic', None, lambda c: True, _ws_generic(_L.PASS_BWD_WEIGHT)),      # fp32-MFMA / small-M weight-gradient kernels
)
ROUTE_TABLE = {_L.PASS_FWD: _FWD_ROUTES, _L.PASS_BWD_DATA: _DGRAD_ROUTES, _L.PASS_BWD_WEIGHT: _WGRAD_ROUTES}


def _plan(pass_, N, C, H, W, K, R, S, stride, pad, pad_mode, dt, no_bias=True):
    key = (pass_, N, C, H, W, K, R, S, stride, pad, pad_mode, dt, no_bias, BF16X6, HSPLIT, HGEMM, THIN, THIN_WGRAD, BSPLIT_MIN_PIXELS)
    p = _PLANS.get(key)
    if p is not None:
        return p
    lib = _L.load()
    _sentinel()
    p = _Plan()
    p.d = make_desc(N, C, H, W, K, R, S, stride, pad, pad_mode, dt)
    p.dref = ctypes.byref(p.d)
    p.P, p.Q = p.d.P, p.d.Q
    c = _RouteCtx()
    c.N, c.C, c.H, c.W, c.K, c.R, c.S, c.stride, c.pad, c.pad_mode, c.dt, c.no_bias = N, C, H, W, K, R, S, stride, pad, pad_mode, dt, no_bias
    c.P, c.Q, c.pixels = p.P, p.Q, N * p.P * p.Q
    c.lib, c.dref = lib, p.dref
    c.split = BF16X6 or dt == BF16          # the matrix-pipe split kernels may take the call (always for bf16 tensors)
    c.f16 = HSPLIT and dt == F32            # fp32 tensors on the fp16 two-piece route
    for route, pack_pass, cond, ws in ROUTE_TABLE[pass_]:
        if cond(c):
            p.route, p.pack_pass, p.ws_bytes = route, pack_pass, ws(c)
            break
    _PLANS[key] = p
    return p


def conv2d_fwd(x, w, bias, stride, pad, pad_mode=0, act=ACT_NONE, slope=0.0, pack_cache=None):
    """pack_cache: a dict owned by the caller (one per weight tensor) that keeps the packed weights between
    calls; None packs inside the call."""
    _chk(w, bias)
    dt = _act(x)
    lib = _L.load()
    N, C, H, W = x.shape
    K, C2, R, S = w.shape
    assert C == C2, 'conv2d_fwd: channel mismatch %d vs %d' % (C, C2)
    pl = _plan(_L.PASS_FWD, N, C, H, W, K, R, S, stride, pad, pad_mode, dt)
    d = pl.dref
    y = torch.empty((N, K, pl.P, pl.Q), dtype=x.dtype, device=x.device)
    ws = _ws(pl.ws_bytes, x.device)
    if pack_cache is not None:
        if pl.route == 'hgemm':
            pk, wmax = _packed_weights(lib, pl.d, pl.pack_pass, w, pack_cache, True)
        else:
            pk = _packed_weights(lib, pl.d, pl.pack_pass, w, pack_cache)
        _count_route('fwd', pl.route)
        if pl.route == 'hgemm':
            xmax = amax_of(x)
            _L.check(lib.pcgan_conv2d_fwd_packed_hsplit(d, _p(x), _p(xmax), xmax.numel(), _p(pk), _p(wmax), _p(bias), _p(y), act,
                                                        float(slope), _p(ws), ws.numel(), _stream()), 'conv2d_fwd_packed_hsplit')
        elif pl.route == 'hsplit':
            xmax = amax_of(x)
            _L.check(lib.pcgan_conv2d_fwd_hsplit(d, _p(x), _p(xmax), xmax.numel(), _p(pk), _p(bias), _p(y), act, float(slope), _stream()),
                     'conv2d_fwd_hsplit')
        elif pl.route == 'bsplit':
            _L.check(lib.pcgan_conv2d_fwd_bsplit(d, _p(x), _p(pk), _p(bias), _p(y), act, float(slope), _stream()), 'conv2d_fwd_bsplit')
        elif pl.route == 'thin':
            xmax = amax_of(x)
            _L.check(lib.pcgan_conv2d_fwd_thin(d, _p(x), _p(xmax), xmax.numel(), _p(pk), _p(bias), _p(y), act, float(slope), _stream()),
                     'conv2d_fwd_thin')
        else:
            _L.check(lib.pcgan_conv2d_fwd_packed(d, _p(x), _p(pk), _p(bias), _p(y), act, float(slope), _p(ws), ws.numel(), _stream()),
                     'conv2d_fwd_packed')
        return y
    _L.check(lib.pcgan_conv2d_fwd(d, _p(x), _p(w), _p(bias), _p(y), act, float(slope), _p(ws), ws.numel(), _stream()), 'conv2d_fwd')
    return y


def conv2d_bwd_data(dy, w, in_hw, stride, pad, pad_mode=0, bias=None, pack_cache=None):
    """dx[N][C][H][W] for a conv with weight w[K][C][R][S]; in_hw = (H, W) of the conv input."""
    _chk(w, bias)
    dt = _act(dy)
    lib = _L.load()
    N, K, P, Q = dy.shape
    K2, C, R, S = w.shape
    assert K == K2, 'conv2d_bwd_data: channel mismatch'
    H, W = in_hw
    pl = _plan(_L.PASS_BWD_DATA, N, C, H, W, K, R, S, stride, pad, pad_mode, dt, bias is None)
    d = pl.dref
    assert (pl.P, pl.Q) == (P, Q), 'conv2d_bwd_data: geometry mismatch %s vs %s' % ((pl.P, pl.Q), (P, Q))
    dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device)
    ws = _ws(pl.ws_bytes, dy.device)
    if pack_cache is not None:
        if pl.route == 'hgemm':
            pk, wmax = _packed_weights(lib, pl.d, pl.pack_pass, w, pack_cache, True)
        else:
            pk = _packed_weights(lib, pl.d, pl.pack_pass, w, pack_cache)
        _count_route('dgrad', pl.route)
        if pl.route == 'hsplit':
            dmax = amax_of(dy)
            _L.check(lib.pcgan_conv2d_bwd_data_hsplit(d, _p(dy), _p(dmax), dmax.numel(), _p(pk), _p(dx), _stream()), 'conv2d_bwd_data_hsplit')
        elif pl.route == 'bsplit':
            _L.check(lib.pcgan_conv2d_bwd_data_bsplit(d, _p(dy), _p(pk), _p(dx), _stream()), 'conv2d_bwd_data_bsplit')
        elif pl.route == 'thin':
            dmax = amax_of(dy)
            _L.check(lib.pcgan_conv2d_bwd_data_thin(d, _p(dy), _p(dmax), dmax.numel(), _p(pk), _p(dx), _p(ws), ws.numel(), _stream()),
                     'conv2d_bwd_data_thin')
        elif pl.route == 'hgemm':
            dmax = amax_of(dy)
            _L.check(lib.pcgan_conv2d_bwd_data_packed_hsplit(d, _p(dy), _p(dmax), dmax.numel(), _p(pk), _p(wmax), _p(bias), _p(dx),
                                                             _p(ws), ws.numel(), _stream()), 'conv2d_bwd_data_packed_hsplit')
        else:
            _L.check(lib.pcgan_conv2d_bwd_data_packed(d, _p(dy), _p(pk), _p(bias), _p(dx), _p(ws), ws.numel(), _stream()),
                     'conv2d_bwd_data_packed')
        return dx
    _L.check(lib.pcgan_conv2d_bwd_data(d, _p(dy), _p(w), _p(bias), _p(dx), _p(ws), ws.numel(), _stream()), 'conv2d_bwd_data')
    return dx


def conv2d_bwd_weight(x, dy, w_shape, stride, pad, pad_mode=0, accumulate_into=None):
    """dw, or -- with accumulate_into = the parameter's gradient buffer -- `accumulate_into += dw` in place."""
    _chk(accumulate_into)
    dt = _act(x, dy)
    lib = _L.load()
    N, C, H, W = x.shape
    K, C2, R, S = w_shape
    assert C == C2 and dy.shape[1] == K
    if dt == BF16 and SMALLM_WGRAD_VIA_FP32 and K <= 3 and stride == 1 and R >= 3 and C >= 16:
        # WORKAROUND, cause not established (round 4): the <= 3-output-channel strip kernel (the generator head's weight gradient, 64 -> 3,
        # 7x7) takes 0.59 ms on bf16 tensors against 0.22 on fp32 tensors of the same values -- same instruction count, same grid; dword
        # instead of 16-bit loads changed nothing.  The fp32 kernel on the up-cast tensors gives bit-identical results (bf16 values are
        # exact in fp32, same order of accumulation) in 0.25 ms incl. the two casts.  PCGAN_SMALLM_WGRAD_VIA_FP32=0 switches it off.
        x, dy, dt = x.float(), dy.float(), F32
    pl = _plan(_L.PASS_BWD_WEIGHT, N, C, H, W, K, R, S, stride, pad, pad_mode, dt)
    d = pl.dref
    assert (pl.P, pl.Q) == tuple(dy.shape[2:]), 'conv2d_bwd_weight: geometry mismatch'
    if accumulate_into is not None:
        assert tuple(accumulate_into.shape) == (K, C, R, S)
        dw = accumulate_into
    else:
        dw = torch.empty((K, C, R, S), dtype=torch.float32, device=x.device)
    acc = int(accumulate_into is not None)
    ws = _ws(pl.ws_bytes, x.device)
    _count_route('wgrad', pl.route)
    if pl.route == 'hsplit':
        if dt == F32:
            xmax, dmax = amax_of(x), amax_of(dy)
            nx, nd = xmax.numel(), dmax.numel()
        else:            # bf16 tensors: one product, no scaling
            xmax = dmax = None
            nx = nd = 0
        _L.check(lib.pcgan_conv2d_bwd_weight_hsplit(d, _p(x), _p(xmax), nx, _p(dy), _p(dmax), nd, _p(dw), acc,
                                                    _p(ws), ws.numel(), _stream()), 'conv2d_bwd_weight_hsplit')
    elif pl.route == 'bsplit':
        _L.check(lib.pcgan_conv2d_bwd_weight_bsplit(d, _p(x), _p(dy), _p(dw), acc, _p(ws), ws.numel(), _stream()), 'conv2d_bwd_weight_bsplit')
    else:
        _L.check(lib.pcgan_conv2d_bwd_weight(d, _p(x), _p(dy), _p(dw), acc, _p(ws), ws.numel(), _stream()), 'conv2d_bwd_weight')
    return dw


# ---------------------------------------------------------------- composite: one ResnetBlock per call
# (include/pcgan_hip.h "composite"): the same launches as the per-op sequence conv -> IN+ReLU -> conv -> IN+skip (and its backward)
# from ONE ctypes call -- the host side of a config-2 step spends ~12 of its 27 ms on the 18 blocks x 2 generator passes.
SMALLM_WGRAD_VIA_FP32 = os.environ.get('PCGAN_SMALLM_WGRAD_VIA_FP32', '1') != '0'
COMPOSITE = os.environ.get('PCGAN_COMPOSITE', '1') != '0'
TRUNK = os.environ.get('PCGAN_TRUNK', '1') != '0'      # runs of ResnetBlocks in ONE library call (pcgan_restrunk_*); 0: one call per block
COMPOSITE_STATS = {'fwd': 0, 'bwd': 0}
_RB_PLANS = {}
_FORK_EVENTS = {}
_WGRAD_WS = {}


class _RBPlan(object):
    __slots__ = ('d', 'dref', 'ok', 'conv', 'ws_bytes', 'fwd_pass', 'bwd_pass', 'half')


def resblock_plan(N, C, H, W, eps, momentum, dt=F32):
    key = (N, C, H, W, float(eps), float(momentum), dt, BF16X6, HSPLIT, HGEMM, BSPLIT_MIN_PIXELS)
    p = _RB_PLANS.get(key)
    if p is None:
        lib = _L.load()
        p = _RBPlan()
        p.d = ResBlockDesc(N, C, H, W, float(eps), float(momentum), dt)
        p.dref = ctypes.byref(p.d)
        # the composite is the per-op sequence: taken exactly where the host would route all three passes to the kernels it calls -- fp32
        # tensors: the fp16 two-piece route; bf16 tensors (round 4): the one-product window kernels + the matrix-pipe weight gradient
        want = ('hsplit', 'hsplit', 'hsplit') if dt == F32 else ('bsplit', 'bsplit', 'hsplit')
        got = tuple(_plan(ps, N, C, H, W, C, 3, 3, 1, 1, 1, dt).route for ps in (_L.PASS_FWD, _L.PASS_BWD_DATA, _L.PASS_BWD_WEIGHT))
        p.ok = bool(COMPOSITE and lib.pcgan_resblock_supported(p.dref) and got == want)
        p.conv = _plan(_L.PASS_FWD, N, C, H, W, C, 3, 3, 1, 1, 1, dt).d if p.ok else None
        p.ws_bytes = int(lib.pcgan_resblock_wgrad_workspace_bytes(p.dref)) if p.ok else 0
        p.fwd_pass, p.bwd_pass = (PASS_FWD_HSPLIT, PASS_BWD_HSPLIT) if dt == F32 else (PASS_FWD_BSPLIT, PASS_BWD_BSPLIT)
        p.half = dt != F32
        _RB_PLANS[key] = p
    return p


def _fork_event(device):
    ev = _FORK_EVENTS.get(device)
    if ev is None:
        box = _vp(0)
        _L.check(_L.load().pcgan_event_create(ctypes.byref(box)), 'event_create')
        ev = _FORK_EVENTS[device] = _vp(box.value)
    return ev


def resblock_fwd(pl, x, w1, b1, w2, b2, rm1, rv1, rm2, rv2, pack1, pack2):
    """returns out, (y1, h, y2, stats, amax, x_amax): everything the backward call needs besides x"""
    _chk(w1, b1, w2, b2, rm1, rv1, rm2, rv2)
    _act(x)
    lib = _L.load()
    N, C = x.shape[0], x.shape[1]
    pk1 = _packed_weights(lib, pl.conv, pl.fwd_pass, w1, pack1)
    pk2 = _packed_weights(lib, pl.conv, pl.fwd_pass, w2, pack2)
    xmax = None if pl.half else amax_of(x)      # (bf16 tensors: one product, no operand scaling)
    y1, h, y2, out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    stats = torch.empty(4 * N * C, dtype=torch.float32, device=x.device)
    amax = None if pl.half else torch.empty(2 * N * C, dtype=torch.float32, device=x.device)
    _L.check(lib.pcgan_resblock_fwd(pl.dref, _p(x), _p(xmax), 0 if pl.half else xmax.numel(), _p(pk1), _p(b1), _p(pk2), _p(b2), _p(rm1), _p(rv1),
                                    _p(rm2), _p(rv2), _p(y1), _p(h), _p(y2), _p(out), _p(stats), _p(amax), _stream()), 'resblock_fwd')
    route = 'bsplit' if pl.half else 'hsplit'
    _count_route('fwd', route)
    _count_route('fwd', route)
    COMPOSITE_STATS['fwd'] += 1
    if not pl.half:
        _attach_amax(out, amax[N * C:])
    return out, (y1, h, y2, stats, amax, xmax)


def resblock_bwd(pl, dout, x, saved, w1, w2, dw1, db1, dw2, db2, pack1, pack2):
    """dx (skip connection included); the parameter gradients are ADDED into dw* / db* on the parameter-gradient stream"""
    _chk(dw1, db1, dw2, db2)
    _act(dout, x)
    y1, h, y2, stats, amax, xmax = saved
    lib = _L.load()
    N, C = x.shape[0], x.shape[1]
    pk1b = _packed_weights(lib, pl.conv, pl.bwd_pass, w1, pack1)
    pk2b = _packed_weights(lib, pl.conv, pl.bwd_pass, w2, pack2)
    cur = torch.cuda.current_stream()
    side = side_stream_for(cur)
    wkey = (x.device, pl.ws_bytes)
    ws = _WGRAD_WS.get(wkey)
    if ws is None:     # one workspace for every weight-gradient launch of this shape: they run in issue order on the one side stream
        with torch.cuda.stream(side):
            ws = _WGRAD_WS[wkey] = _ws(pl.ws_bytes, x.device)
    dy2, dh, dy1, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    scratch = torch.empty(5 * N * C, dtype=torch.float32, device=x.device)
    for t in (x, h, dy2, dy1, scratch, amax, xmax):      # read by the side stream after this call returns
        if t is not None:
            t.record_stream(side)
    _L.check(lib.pcgan_resblock_bwd(pl.dref, _p(dout), _p(x), _p(xmax), 0 if pl.half else xmax.numel(), _p(y1), _p(h), _p(amax), _p(y2), _p(stats), _p(pk1b),
                                    _p(pk2b), _p(dw1), _p(db1), _p(dw2), _p(db2), _p(dy2), _p(dh), _p(dy1), _p(dx), _p(scratch), _p(ws),
                                    ws.numel(), _vp(cur.cuda_stream), _vp(side.cuda_stream), _fork_event(x.device)), 'resblock_bwd')
    for _ in range(2):
        _count_route('dgrad', 'bsplit' if pl.half else 'hsplit')
        _count_route('wgrad', 'hsplit')
    if not pl.half:
        AMAX_STATS['attached'] += 6        # (x, h, dy2 x 2, dy1 x 2: the operand maxima all came from the norm kernels)
    PLANE_SUM_STATS['fused'] += 2
    COMPOSITE_STATS['bwd'] += 1
    mark_side_used()
    return dx


_LAST_TRUNK = {}


def _ptr_array(items):
    return (ctypes.c_void_p * len(items))(*[(t.data_ptr() if t is not None else None) for t in items])


def restrunk_fwd(pl, x, blocks):
    """a run of ResnetBlocks in one library call (pcgan_restrunk_fwd).  blocks: [(w1, b1, w2, b2, rm1, rv1, rm2, rv2, pack1, pack2)].
    returns the last block's output (a slice of the stacked outputs, plane maxima attached) and what the backward call needs"""
    lib = _L.load()
    nb = len(blocks)
    N, C = x.shape[0], x.shape[1]
    for b in blocks:
        _chk(*b[:8])
    _act(x)
    pk1 = [_packed_weights(lib, pl.conv, pl.fwd_pass, b[0], b[8]) for b in blocks]
    pk2 = [_packed_weights(lib, pl.conv, pl.fwd_pass, b[2], b[9]) for b in blocks]
    xmax = None if pl.half else amax_of(x)
    shape = (nb,) + tuple(x.shape)
    y1, h, y2, out = (torch.empty(shape, dtype=x.dtype, device=x.device) for _ in range(4))
    stats = torch.empty(nb * 4 * N * C, dtype=torch.float32, device=x.device)
    amax = torch.empty(nb * 2 * N * C, dtype=torch.float32, device=x.device)
    _L.check(lib.pcgan_restrunk_fwd(pl.dref, nb, _p(x), _p(xmax), 0 if pl.half else xmax.numel(), _ptr_array(pk1), _ptr_array([b[1] for b in blocks]),
                                    _ptr_array(pk2), _ptr_array([b[3] for b in blocks]), _ptr_array([b[4] for b in blocks]),
                                    _ptr_array([b[5] for b in blocks]), _ptr_array([b[6] for b in blocks]), _ptr_array([b[7] for b in blocks]),
                                    _p(y1), _p(h), _p(y2), _p(out), _p(stats), _p(amax), _stream()), 'restrunk_fwd')
    rk = ('fwd', 'bsplit' if pl.half else 'hsplit')
    ROUTE_STATS[rk] = ROUTE_STATS.get(rk, 0) + 2 * nb
    COMPOSITE_STATS['fwd'] += nb
    COMPOSITE_STATS['trunk_fwd'] = COMPOSITE_STATS.get('trunk_fwd', 0) + 1
    last = out[nb - 1]
    if not pl.half:
        AMAX_STATS['attached'] += 2 * nb - 1      # (every convolution but the first took its operand maxima from the norm kernel in front of it)
        _attach_amax(last, amax[(2 * nb - 1) * N * C:])
        _LAST_TRUNK['amax'] = last.__dict__['_pcgan_amax']      # (autograd may hand the caller another tensor object for a view output)
    return last, (y1, h, y2, out, stats, amax, xmax)


def restrunk_bwd(pl, dout, x, saved, blocks):
    """blocks: [(w1, w2, dw1, db1, dw2, db2, pack1, pack2)]; dx of the chain's input, parameter gradients ADDED on the side stream"""
    y1, h, y2, out, stats, amax, xmax = saved
    lib = _L.load()
    nb = len(blocks)
    N, C = x.shape[0], x.shape[1]
    _act(dout, x)
    pk1b = [_packed_weights(lib, pl.conv, pl.bwd_pass, b[0], b[6]) for b in blocks]
    pk2b = [_packed_weights(lib, pl.conv, pl.bwd_pass, b[1], b[7]) for b in blocks]
    cur = torch.cuda.current_stream()
    side = side_stream_for(cur)
    wkey = (x.device, pl.ws_bytes)
    ws = _WGRAD_WS.get(wkey)
    if ws is None:
        with torch.cuda.stream(side):
            ws = _WGRAD_WS[wkey] = _ws(pl.ws_bytes, x.device)
    shape = (nb,) + tuple(x.shape)
    dy2, dy1 = torch.empty(shape, dtype=x.dtype, device=x.device), torch.empty(shape, dtype=x.dtype, device=x.device)
    dh, dx = torch.empty_like(x), torch.empty_like(x)
    dxs = torch.empty((2,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    scratch = torch.empty(nb * 5 * N * C, dtype=torch.float32, device=x.device)
    for t in (x, h, out, dy2, dy1, scratch, amax, xmax):      # read by the side stream after this call returns
        if t is not None:
            t.record_stream(side)
    _L.check(lib.pcgan_restrunk_bwd(pl.dref, nb, _p(dout), _p(x), _p(xmax), 0 if pl.half else xmax.numel(), _p(y1), _p(h), _p(y2), _p(out), _p(stats), _p(amax),
                                    _ptr_array(pk1b), _ptr_array(pk2b), _ptr_array([b[2] for b in blocks]), _ptr_array([b[3] for b in blocks]),
                                    _ptr_array([b[4] for b in blocks]), _ptr_array([b[5] for b in blocks]), _p(dy2), _p(dh), _p(dy1), _p(dxs),
                                    _p(dx), _p(scratch), _p(ws), ws.numel(), _vp(cur.cuda_stream), _vp(side.cuda_stream),
                                    _fork_event(x.device)), 'restrunk_bwd')
    dk = ('dgrad', 'bsplit' if pl.half else 'hsplit')
    ROUTE_STATS[dk] = ROUTE_STATS.get(dk, 0) + 2 * nb
    ROUTE_STATS[('wgrad', 'hsplit')] = ROUTE_STATS.get(('wgrad', 'hsplit'), 0) + 2 * nb
    if not pl.half:
        AMAX_STATS['attached'] += 6 * nb
    PLANE_SUM_STATS['fused'] += 2 * nb
    COMPOSITE_STATS['bwd'] += nb
    COMPOSITE_STATS['trunk_bwd'] = COMPOSITE_STATS.get('trunk_bwd', 0) + 1
    mark_side_used()
    return dx


# ---------------------------------------------------------------- kernel timer (bench.py's roofline block)
TIMER_KINDS = {'res_fwd': 0, 'res_dgrad': 1, 'res_wgrad': 2, 'res_wgrad_main': 3}


def timer_enable(capacity):
    _L.check(_L.load().pcgan_timer_enable(int(capacity)), 'timer_enable')


def timer_read(kind, cap=65536):
    buf = (ctypes.c_float * cap)()
    n = _L.load().pcgan_timer_read(TIMER_KINDS[kind], buf, cap)
    if n < 0:
        raise RuntimeError('pcgan_timer_read failed')
    return [float(buf[i]) for i in range(n)]


# ---------------------------------------------------------------- pointwise
def channel_sum(x, accumulate_into=None):
    _chk(accumulate_into)
    dt = _act(x)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    out = accumulate_into if accumulate_into is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    psum = getattr(x, '_pcgan_plane_sums', None)
    if psum is not None and psum.numel() == N * C and psum.device == x.device:
        # x is the dx an instance-norm backward just produced: its plane sums exist already (instnorm_bwd)
        PLANE_SUM_STATS['fused'] += 1
        if getattr(x, '_pcgan_plane_sums_stream', None) != _raw_stream():     # (the parameter-gradient side stream)
            psum.record_stream(torch.cuda.current_stream())
        _L.check(_L.load().pcgan_sum_planes(_p(psum), _p(out), N, C, int(accumulate_into is not None), _stream()), 'sum_planes')
        return out
    PLANE_SUM_STATS['full'] += 1
    scratch = torch.empty(N * C, dtype=torch.float32, device=x.device)
    _L.check(_L.load().pcgan_channel_sum(_p(x), _p(out), _p(scratch), N, C, HW, int(accumulate_into is not None), dt,
                                         _stream()), 'channel_sum')
    return out


def act_fwd(x, act, slope=0.0):
    dt = _act(x)
    y = torch.empty_like(x)
    _L.check(_L.load().pcgan_act_fwd(_p(x), _p(y), x.numel(), act, float(slope), dt, _stream()), 'act_fwd')
    return y


def act_bwd(dy, y, act, slope=0.0):
    dt = _act(dy, y)
    dx = torch.empty_like(dy)
    _L.check(_L.load().pcgan_act_bwd(_p(dy), _p(y), _p(dx), dy.numel(), act, float(slope), dt, _stream()), 'act_bwd')
    return dx


def add(a, b):
    dt = _act(a, b)
    assert a.shape == b.shape
    y = torch.empty_like(a)
    _L.check(_L.load().pcgan_add(_p(a), _p(b), _p(y), a.numel(), dt, _stream()), 'add')
    return y


def scale(x, scalar_dev=None, alpha=1.0):
    _chk(scalar_dev)
    dt = _act(x)
    y = torch.empty_like(x)
    _L.check(_L.load().pcgan_scale(_p(x), _p(scalar_dev), float(alpha), _p(y), x.numel(), dt, _stream()), 'scale')
    return y


def concat_z(img, z):
    _chk(z)                   # ratings stay fp32
    dt = _act(img)
    N, C, H, W = img.shape
    zb, nz = z.shape[0], z.shape[1]
    out = torch.empty((N, C + nz, H, W), dtype=img.dtype, device=img.device)
    _L.check(_L.load().pcgan_concat_z(_p(img), _p(z), _p(out), N, C, nz, H * W, zb, dt, _stream()), 'concat_z')
    return out


def channel_scale(x, mask_nc, scale_):
    _chk(mask_nc)
    dt = _act(x)
    N, C = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    _L.check(_L.load().pcgan_channel_scale(_p(x), _p(mask_nc), _p(y), N * C, x.numel() // (N * C), float(scale_), dt,
                                           _stream()), 'channel_scale')
    return y


# ---------------------------------------------------------------- normalisation
def plane_stats(x):
    dt = _act(x)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
    m2 = torch.empty(N * C, dtype=torch.float32, device=x.device)
    _L.check(_L.load().pcgan_plane_stats(_p(x), _p(mean), _p(m2), N * C, HW, dt, _stream()), 'plane_stats')
    return mean, m2


def bn_merge(mean_nc, m2_nc, N, C, HW, running_mean, running_var, momentum):
    _chk(mean_nc, m2_nc, running_mean, running_var)
    mean_c = torch.empty(C, dtype=torch.float32, device=mean_nc.device)
    var_c = torch.empty(C, dtype=torch.float32, device=mean_nc.device)
    _L.check(_L.load().pcgan_bn_merge(_p(mean_nc), _p(m2_nc), _p(mean_c), _p(var_c), _p(running_mean),
                                      _p(running_var), N, C, HW, float(momentum), _stream()), 'bn_merge')
    return mean_c, var_c


def bn_stats_merged(x, running_mean, running_var, batches, ticket, momentum):
    """batch statistics of a train-mode BatchNorm2d in ONE launch (plane_stats + bn_merge + batch counter: the last-arriving
    workgroup of every channel merges its N plane statistics); returns mean_c, var_c (biased)"""
    _chk(running_mean, running_var)
    dt = _act(x)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    assert ticket.dtype == torch.int32 and ticket.numel() >= C and ticket.is_cuda
    if batches is not None:
        assert batches.dtype == torch.int64 and batches.is_cuda
    part = torch.empty(2 * N * C, dtype=torch.float32, device=x.device)
    mc = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    _L.check(_L.load().pcgan_bn_stats_merged(_p(x), _p(part), _vp(part.data_ptr() + 4 * N * C), _p(mc), _vp(mc.data_ptr() + 4 * C),
                                             _p(running_mean), _p(running_var), _p(batches), _p(ticket), N, C, HW, float(momentum), dt,
                                             _stream()), 'bn_stats_merged')
    return mc[:C], mc[C:]


def bn_bwd_stats_reduced(dy, x, y, mean, var, eps, act, slope, ticket):
    """s1_c = sum g, s2_c = sum g * xhat over N, H, W in ONE launch (norm_bwd_stats + bn_bwd_reduce)"""
    _chk(mean, var)
    dt = _act(dy, x, y)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    assert ticket.dtype == torch.int32 and ticket.numel() >= C and ticket.is_cuda
    part = torch.empty(2 * N * C, dtype=torch.float32, device=x.device)
    sc = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    _L.check(_L.load().pcgan_bn_bwd_stats_reduced(_p(dy), _p(x), _p(y), _p(mean), _p(var), _p(part), _vp(part.data_ptr() + 4 * N * C), _p(sc),
                                                  _vp(sc.data_ptr() + 4 * C), _p(ticket), N, C, HW, float(eps), act, float(slope), dt,
                                                  _stream()), 'bn_bwd_stats_reduced')
    return sc[:C], sc[C:]


def in_running_update(mean_nc, m2_nc, running_mean, running_var, N, C, HW, momentum):
    _chk(mean_nc, m2_nc, running_mean, running_var)
    _L.check(_L.load().pcgan_in_running_update(_p(mean_nc), _p(m2_nc), _p(running_mean), _p(running_var), N, C, HW,
                                               float(momentum), _stream()), 'in_running_update')


def norm_act_fwd(x, mean, var, gamma, beta, residual, per_plane, eps, act, slope):
    _chk(mean, var, gamma, beta)
    dt = _act(x, residual)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    y = torch.empty_like(x)
    pmax = torch.empty(N * C, dtype=torch.float32, device=x.device) if _want_maxima(dt, C) else None
    _L.check(_L.load().pcgan_norm_act_fwd(_p(x), _p(mean), _p(var), _p(gamma), _p(beta), _p(residual), _p(y), _p(pmax), N, C,
                                          HW, int(per_plane), float(eps), act, float(slope), dt, _stream()),
             'norm_act_fwd')
    if pmax is not None:
        _attach_amax(y, pmax)
    return y


def norm_bwd_stats(dy, x, y, mean, var, per_plane, eps, act, slope):
    _chk(mean, var)
    dt = _act(dy, x, y)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    s1 = torch.empty(N * C, dtype=torch.float32, device=x.device)
    s2 = torch.empty(N * C, dtype=torch.float32, device=x.device)
    _L.check(_L.load().pcgan_norm_bwd_stats(_p(dy), _p(x), _p(y), _p(mean), _p(var), _p(s1), _p(s2), N, C, HW,
                                            int(per_plane), float(eps), act, float(slope), dt, _stream()),
             'norm_bwd_stats')
    return s1, s2


def bn_bwd_reduce(s1_nc, s2_nc, N, C):
    _chk(s1_nc, s2_nc)
    s1 = torch.empty(C, dtype=torch.float32, device=s1_nc.device)
    s2 = torch.empty(C, dtype=torch.float32, device=s1_nc.device)
    _L.check(_L.load().pcgan_bn_bwd_reduce(_p(s1_nc), _p(s2_nc), _p(s1), _p(s2), N, C, _stream()), 'bn_bwd_reduce')
    return s1, s2


def norm_bwd_apply(dy, x, y, mean, var, gamma, s1, s2, per_plane, eps, act, slope, want_residual_grad):
    _chk(mean, var, gamma, s1, s2)
    dt = _act(dy, x, y)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_residual_grad else None
    pmax = torch.empty(N * C, dtype=torch.float32, device=x.device) if _want_maxima(dt, C) else None
    _L.check(_L.load().pcgan_norm_bwd_apply(_p(dy), _p(x), _p(y), _p(mean), _p(var), _p(gamma), _p(s1), _p(s2),
                                            _p(dx), _p(dres), _p(pmax), N, C, HW, int(per_plane), float(eps), act,
                                            float(slope), dt, _stream()), 'norm_bwd_apply')
    if pmax is not None:
        _attach_amax(dx, pmax)
    return dx, dres


BN_FUSED_MAX = int(os.environ.get('PCGAN_BN_FUSED_MAX', 8192))   # elements per channel (N * HW) up to which BatchNorm runs as one launch per pass


def bn_fwd_fused(x, gamma, beta, residual, running_mean, running_var, batches, momentum, eps, act, slope):
    """training-mode BatchNorm2d (+ residual + activation) of a small tensor in one launch; returns y, mean, var (biased)"""
    _chk(gamma, beta, running_mean, running_var)
    dt = _act(x, residual)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    var = torch.empty(C, dtype=torch.float32, device=x.device)
    if batches is not None:
        assert batches.dtype == torch.int64 and batches.is_cuda
    cmax = torch.empty(C, dtype=torch.float32, device=x.device) if _want_maxima(dt, C) else None
    _L.check(_L.load().pcgan_bn_fwd_fused(_p(x), _p(gamma), _p(beta), _p(residual), _p(y), _p(mean), _p(var), _p(running_mean),
                                          _p(running_var), _p(batches), _p(cmax), N, C, HW, float(momentum), float(eps), act, float(slope),
                                          dt, _stream()), 'bn_fwd_fused')
    if cmax is not None:
        _attach_amax(y, cmax)
    return y, mean, var


def bn_bwd_fused(dy, x, y, mean, var, gamma, eps, act, slope, want_dx, want_dres):
    _chk(mean, var, gamma)
    dt = _act(dy, x, y)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    dx = torch.empty_like(x) if want_dx else None
    dres = torch.empty_like(x) if want_dres else None
    s1 = torch.empty(C, dtype=torch.float32, device=x.device)
    s2 = torch.empty(C, dtype=torch.float32, device=x.device)
    cmax = torch.empty(C, dtype=torch.float32, device=x.device) if want_dx and _want_maxima(dt, C) else None
    _L.check(_L.load().pcgan_bn_bwd_fused(_p(dy), _p(x), _p(y), _p(mean), _p(var), _p(gamma), _p(dx), _p(dres), _p(s1), _p(s2), _p(cmax),
                                          N, C, HW, float(eps), act, float(slope), dt, _stream()), 'bn_bwd_fused')
    if cmax is not None:
        _attach_amax(dx, cmax)
    return dx, dres, s1, s2


def instnorm_fwd(x, residual, eps, act, slope):
    """fused instance norm forward: returns y, mean[N*C], m2[N*C]"""
    dt = _act(x, residual)
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    y = torch.empty_like(x)
    mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
    m2 = torch.empty(N * C, dtype=torch.float32, device=x.device)
    lib = _L.load()
    # the largest magnitude of y for the fp16 route of the convolution that reads it (residual blocks: width 32 / 64 planes)
    slot = torch.empty(N * C, dtype=torch.float32, device=x.device) if _want_maxima(dt, C) and lib.pcgan_instnorm_fused(HW) else None
    _L.check(lib.pcgan_instnorm_fwd(_p(x), _p(residual), _p(y), _p(mean), _p(m2), _p(slot), N, C, HW, float(eps), act,
                                    float(slope), dt, _stream()), 'instnorm_fwd')
    if slot is not None:
        _attach_amax(y, slot)
    return y, mean, m2


PLANE_SUM_STATS = {'fused': 0, 'full': 0}     # bias gradients finished from plane sums / by a full channel_sum pass


def instnorm_bwd(dy, x, y, mean, m2, eps, act, slope):
    """dx; where the plane runs in the register-resident kernel dx also carries `_pcgan_plane_sums` ([N*C] sums of dx over
    each plane, free there): the convolution in front of the norm finishes its bias gradient from them (channel_sum)."""
    _chk(mean, m2)
    dt = _act(dy, x, y)
    lib = _L.load()
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    dx = torch.empty_like(x)
    fused = bool(lib.pcgan_instnorm_fused(HW))
    psum = torch.empty(N * C, dtype=torch.float32, device=x.device) if fused else None
    ws = None if fused else torch.empty(2 * N * C, dtype=torch.float32, device=x.device)
    slot = torch.empty(N * C, dtype=torch.float32, device=x.device) if fused and _want_maxima(dt, C) else None
    _L.check(lib.pcgan_instnorm_bwd(_p(dy), _p(x), _p(y), _p(mean), _p(m2), _p(dx), _p(psum), _p(slot), _p(ws), N, C, HW, float(eps),
                                    act, float(slope), dt, _stream()), 'instnorm_bwd')
    if psum is not None:
        dx._pcgan_plane_sums = psum
        dx._pcgan_plane_sums_stream = _raw_stream()
    if slot is not None:
        _attach_amax(dx, slot)
    return dx


# ---------------------------------------------------------------- pooling / resize
def maxpool_fwd(x, k, stride, pad):
    dt = _act(x)
    N, C, H, W = x.shape
    P, Q = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    y = torch.empty((N, C, P, Q), dtype=x.dtype, device=x.device)
    arg = torch.empty((N, C, P, Q), dtype=torch.int32, device=x.device)
    _L.check(_L.load().pcgan_maxpool_fwd(_p(x), _p(y), _vp(arg.data_ptr()), N * C, H, W, k, stride, pad, P, Q, dt,
                                         _stream()), 'maxpool_fwd')
    return y, arg


def maxpool_bwd(dy, arg, in_hw, k, stride, pad):
    dt = _act(dy)
    N, C, P, Q = dy.shape
    H, W = in_hw
    dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device)
    _L.check(_L.load().pcgan_maxpool_bwd(_p(dy), _vp(arg.data_ptr()), _p(dx), N * C, H, W, k, stride, pad, P, Q, dt,
                                         _stream()), 'maxpool_bwd')
    return dx


def global_pool_fwd(x, is_max):
    dt = _act(x)
    N, C, H, W = x.shape
    y = torch.empty((N, C, 1, 1), dtype=x.dtype, device=x.device)
    arg = torch.empty((N, C), dtype=torch.int32, device=x.device) if is_max else None
    _L.check(_L.load().pcgan_global_pool_fwd(_p(x), _p(y), _vp(arg.data_ptr()) if is_max else _vp(0), N * C, H * W,
                                             int(is_max), dt, _stream()), 'global_pool_fwd')
    return y, arg


def global_pool_bwd(dy, arg, in_hw, is_max):
    dt = _act(dy)
    N, C = dy.shape[0], dy.shape[1]
    H, W = in_hw
    dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device)
    _L.check(_L.load().pcgan_global_pool_bwd(_p(dy), _vp(arg.data_ptr()) if is_max else _vp(0), _p(dx), N * C, H * W,
                                             int(is_max), dt, _stream()), 'global_pool_bwd')
    return dx


def bilinear_fwd(x, size):
    dt = _act(x)
    N, C, H, W = x.shape
    P, Q = size
    y = torch.empty((N, C, P, Q), dtype=x.dtype, device=x.device)
    _L.check(_L.load().pcgan_bilinear_fwd(_p(x), _p(y), N * C, H, W, P, Q, dt, _stream()), 'bilinear_fwd')
    return y


def bilinear_bwd(dy, in_hw):
    dt = _act(dy)
    N, C, P, Q = dy.shape
    H, W = in_hw
    dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device)
    _L.check(_L.load().pcgan_bilinear_bwd(_p(dy), _p(dx), N * C, H, W, P, Q, dt, _stream()), 'bilinear_bwd')
    return dx


# ---------------------------------------------------------------- losses / optimizer
def _loss(fn_name, a, b, n_or_N, per_n, want_grad, dt):
    lib = _L.load()
    loss = torch.empty((), dtype=torch.float32, device=a.device)       # the loss itself is always fp32
    grad = torch.empty_like(a) if want_grad else None
    ws = _ws(lib.pcgan_loss_workspace_bytes(a.numel()), a.device)
    fn = getattr(lib, fn_name)
    if fn_name == 'pcgan_bce_loss':
        st = fn(_p(a), _p(b), _p(loss), _p(grad), n_or_N, per_n, 1.0, _p(ws), ws.numel(), dt, _stream())
    else:
        st = fn(_p(a), _p(b), _p(loss), _p(grad), a.numel(), 1.0, _p(ws), ws.numel(), dt, _stream())
    _L.check(st, fn_name)
    return loss, grad


def bce_loss(pred, target_n, want_grad=True):
    """mean BCE of pred[N][...] against target_n[N] broadcast over each sample."""
    _chk(target_n)
    dt = _act(pred)
    N = pred.shape[0]
    assert target_n.numel() == N
    return _loss('pcgan_bce_loss', pred, target_n, N, pred.numel() // N, want_grad, dt)


def l1_loss(a, b, want_grad=True):
    dt = _act(a, b)
    assert a.shape == b.shape
    return _loss('pcgan_l1_loss', a, b, 0, 1, want_grad, dt)


def mse_loss(a, b, want_grad=True):
    dt = _act(a, b)
    assert a.shape == b.shape
    return _loss('pcgan_mse_loss', a, b, 0, 1, want_grad, dt)


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    _chk(param, grad, exp_avg, exp_avg_sq)
    invalidate_packed_weights()
    _L.check(_L.load().pcgan_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr),
                                       float(beta1), float(beta2), float(eps), int(step), _stream()), 'adam_step')


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, lr_dev, step_dev, beta1, beta2, eps, params=None):
    """params: the parameter tensors that are views of `param` (their packed copies go out of date; None: everybody's)"""
    _chk(param, grad, exp_avg, exp_avg_sq, lr_dev)
    invalidate_packed_weights(params)
    _L.check(_L.load().pcgan_adam_step_dev(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(),
                                           _p(lr_dev), _vp(step_dev.data_ptr()), float(beta1), float(beta2),
                                           float(eps), _stream()), 'adam_step_dev')


def device_info():
    cu = ctypes.c_int(0)
    buf = ctypes.create_string_buffer(64)
    _L.check(_L.load().pcgan_device_info(ctypes.byref(cu), buf, 64), 'device_info')
    return cu.value, buf.value.decode()
