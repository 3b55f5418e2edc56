"""torch.autograd.Function wrappers around the HIP kernels.

Each Function saves exactly what its backward needs and is re-entrant (the generator is
applied twice per step with the first output feeding the second pass; the discriminator
sees the same fake image in two different graphs).  Autograd bookkeeping is the only
thing torch does here -- every forward/backward computation is a libpcgan_hip.so kernel.
"""
import torch

from . import ops
from .lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID  # noqa: F401


def _c(t):
    return t if (t is None or t.is_contiguous()) else t.contiguous()


def _fused_grad_target(p):
    """The parameter's slice of a FusedAdam flat gradient buffer, if it has one: the weight-/bias-gradient
    kernels then add into it directly and autograd receives None (no separate `grad += dw` pass, no dw
    allocation).  FusedAdam marks its parameters with `_pcgan_fused_grad`."""
    if p is not None and getattr(p, '_pcgan_fused_grad', False) and p.grad is not None and p.grad.is_contiguous():
        return p.grad
    return None


def _pack_cache(p):
    """Per-parameter dict holding the packed copies of a conv weight (see ops._packed_weights)."""
    if p is None or not isinstance(p, torch.nn.Parameter):
        return None
    c = p.__dict__.get('_pcgan_pack')
    if c is None:
        c = p.__dict__['_pcgan_pack'] = {}
    return c


# ---------------------------------------------------------------------------- conv
class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, pad_mode, act, slope):
        x, w, b = _c(x), _c(w), _c(b)
        ctx.pack = _pack_cache(w)
        y = ops.conv2d_fwd(x, w, b, stride, pad, pad_mode, act, slope, pack_cache=ctx.pack)
        ctx.cfg = (stride, pad, pad_mode, act, slope)
        ctx.has_bias = b is not None
        ctx.params = (w, b)          # the Parameter objects (for their fused gradient buffers)
        ctx.x_amax = x.__dict__.get('_pcgan_amax')     # operand maxima its producer left (fp16 route): a saved tensor comes back without them
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        if ctx.x_amax is not None and ctx.x_amax[0] == x._version and '_pcgan_amax' not in x.__dict__:
            x._pcgan_amax = ctx.x_amax
        stride, pad, pad_mode, act, slope = ctx.cfg
        dy = _c(dy)
        if act != ACT_NONE:
            dy = ops.act_bwd(dy, y, act, slope)
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        wt = _fused_grad_target(ctx.params[0]) if want_w else None
        bt = _fused_grad_target(ctx.params[1]) if want_b else None
        if ops.SIDE_STREAM and (not want_w or wt is not None) and (not want_b or bt is not None) and (want_w or want_b):
            # both go straight into the optimizer's gradient buffer: nobody in this backward pass reads them.  Forked
            # BEFORE the data gradient is queued, so the two kernels of this layer may run side by side.
            with ops.fork_side(x, dy):
                if want_w:
                    ops.conv2d_bwd_weight(x, dy, tuple(w.shape), stride, pad, pad_mode, accumulate_into=wt)
                if want_b:
                    ops.channel_sum(dy, accumulate_into=bt)
            if ctx.needs_input_grad[0]:
                dx = ops.conv2d_bwd_data(dy, w, (x.shape[2], x.shape[3]), stride, pad, pad_mode, pack_cache=ctx.pack)
            return dx, None, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_bwd_data(dy, w, (x.shape[2], x.shape[3]), stride, pad, pad_mode, pack_cache=ctx.pack)
        if want_w:
            dw = ops.conv2d_bwd_weight(x, dy, tuple(w.shape), stride, pad, pad_mode, accumulate_into=wt)
            if wt is not None:
                dw = None
        if want_b:
            db = ops.channel_sum(dy, accumulate_into=bt)
            if bt is not None:
                db = None
        return dx, dw, db, None, None, None, None, None


class _ResBlockFn(torch.autograd.Function):
    """One ResnetBlock (reference models/networks.py:616-652) as ONE autograd node and ONE library call per pass
    (ops.resblock_fwd / resblock_bwd: the launches of the per-op path, bit for bit).  Taken by nn-level code only when the block's
    shape is on the fp16 route and its parameters feed a FusedAdam gradient buffer (or nothing requires gradients)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, pl, in1, in2):
        upd1 = in1.training and in1.running_mean is not None
        upd2 = in2.training and in2.running_mean is not None
        pack1, pack2 = _pack_cache(w1), _pack_cache(w2)
        out, saved = ops.resblock_fwd(pl, x, w1, b1, w2, b2, in1.running_mean if upd1 else None, in1.running_var if upd1 else None,
                                      in2.running_mean if upd2 else None, in2.running_var if upd2 else None, pack1, pack2)
        ctx.pl, ctx.saved, ctx.packs, ctx.params = pl, saved, (pack1, pack2), (w1, b1, w2, b2)
        ctx.save_for_backward(x, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2 = ctx.saved_tensors
        p1, pb1, p2, pb2 = ctx.params
        tw1, tb1, tw2, tb2 = (_fused_grad_target(p) for p in (p1, pb1, p2, pb2))
        assert tw1 is not None and tw2 is not None and (pb1 is None or tb1 is not None) and (pb2 is None or tb2 is not None), \
            'pcgan_amd: the composite residual block needs FusedAdam gradient buffers (checked in forward)'
        dx = ops.resblock_bwd(ctx.pl, _c(dout), x, ctx.saved, w1, w2, tw1, tb1, tw2, tb2, ctx.packs[0], ctx.packs[1])
        ctx.saved = None
        return dx, None, None, None, None, None, None, None


def resblock_composite_ok(x, conv1, conv2, in1, in2):
    """the launch plan if this ResnetBlock call can take the composite path, else None"""
    if not (ops.COMPOSITE and ops.SIDE_STREAM and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 4 and x.is_contiguous()):
        return None
    if in1.eps != in2.eps or in1.momentum != in2.momentum or in1.momentum is None:
        return None
    for m in (in1, in2):       # plane statistics (train mode, or no running statistics at all)
        if m.affine or not (m.training or not m.track_running_stats):
            return None
    N, C, H, W = x.shape
    for c in (conv1, conv2):
        if tuple(c.weight.shape) != (C, C, 3, 3) or c.stride != (1, 1) or c.padding != (0, 0) or c.weight.dtype != torch.float32:
            return None
    if (conv1.bias is None) != (conv2.bias is None):
        return None
    pl = ops.resblock_plan(N, C, H, W, in1.eps, in1.momentum, ops.F32 if x.dtype == torch.float32 else ops.BF16)
    if not pl.ok:
        return None
    params = [p for p in (conv1.weight, conv1.bias, conv2.weight, conv2.bias) if p is not None]
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        # training: every parameter's gradient goes straight into its optimizer's flat buffer, and x carries a gradient
        if not (x.requires_grad and all(p.requires_grad and _fused_grad_target(p) is not None for p in params)):
            return None
    return pl


def resblock(x, conv1, conv2, in1, in2, pl):
    return _ResBlockFn.apply(x, conv1.weight, conv1.bias, conv2.weight, conv2.bias, pl, in1, in2)


class _ResTrunkFn(torch.autograd.Function):
    """A run of ResnetBlocks (the generator's nine, reference models/networks.py:589-592) as ONE autograd node and ONE library call per
    pass (ops.restrunk_fwd / restrunk_bwd = the launches of the per-block composite calls, bit for bit).  params: (w1, b1, w2, b2) per
    block, flat, so that autograd sees them."""

    @staticmethod
    def forward(ctx, x, pl, norms, *params):
        nb = len(params) // 4
        blocks = []
        for i in range(nb):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            in1, in2 = norms[i]
            u1 = in1.training and in1.running_mean is not None
            u2 = in2.training and in2.running_mean is not None
            blocks.append((w1, b1, w2, b2, in1.running_mean if u1 else None, in1.running_var if u1 else None,
                           in2.running_mean if u2 else None, in2.running_var if u2 else None, _pack_cache(w1), _pack_cache(w2)))
        out, saved = ops.restrunk_fwd(pl, x, blocks)
        ctx.pl, ctx.saved, ctx.params = pl, saved, params
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        params = ctx.params
        blocks = []
        for i in range(len(params) // 4):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            t = [_fused_grad_target(p) for p in (w1, b1, w2, b2)]
            assert t[0] is not None and t[2] is not None and (b1 is None or t[1] is not None) and (b2 is None or t[3] is not None), \
                'pcgan_amd: the composite residual trunk needs FusedAdam gradient buffers (checked in forward)'
            blocks.append((w1, w2, t[0], t[1], t[2], t[3], _pack_cache(w1), _pack_cache(w2)))
        dx = ops.restrunk_bwd(ctx.pl, _c(dout), x, ctx.saved, blocks)
        ctx.saved = None
        return (dx, None, None) + (None,) * len(params)


def restrunk(x, mods):
    """mods: consecutive ResnetBlock modules with the composite layout; the chain through ONE call when every block qualifies (same
    plan), else None (the caller runs them one by one)"""
    if not (ops.TRUNK and len(mods) >= 2):
        return None
    pl = None
    params, norms = [], []
    for m in mods:
        cb = m.conv_block
        p = resblock_composite_ok(x, cb[1], cb[5], cb[2], cb[6])
        if p is None or (pl is not None and p is not pl):
            return None
        pl = p
        params += [cb[1].weight, cb[1].bias, cb[5].weight, cb[5].bias]
        norms.append((cb[2], cb[6]))
    y = _ResTrunkFn.apply(x, pl, norms, *params)
    ent = ops._LAST_TRUNK.pop('amax', None)
    if ent is not None and '_pcgan_amax' not in y.__dict__ and ent[0] == y._version:
        y._pcgan_amax = ent       # the plane maxima of the chain's output (the next convolution's operand scale)
    return y


def conv2d(x, w, b=None, stride=1, pad=0, pad_mode=0, act=ACT_NONE, slope=0.0):
    """nn.Conv2d (optionally preceded by nn.ReflectionPad2d(pad): pad_mode=1) with the
    following pointwise activation fused into the epilogue."""
    return _Conv2dFn.apply(x, w, b, stride, pad, pad_mode, act, slope)


class _ConvTranspose2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d(w[Cin][Cout][R][S]) expressed through the conv entry points of the
    conv (C=Cout -> K=Cin) whose data-gradient it is."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad):
        x, w, b = _c(x), _c(w), _c(b)
        R, S = w.shape[2], w.shape[3]
        Ho = (x.shape[2] - 1) * stride - 2 * pad + R + out_pad
        Wo = (x.shape[3] - 1) * stride - 2 * pad + S + out_pad
        ctx.pack = _pack_cache(w)
        y = ops.conv2d_bwd_data(x, w, (Ho, Wo), stride, pad, 0, bias=b, pack_cache=ctx.pack)
        ctx.cfg = (stride, pad)
        ctx.has_bias = b is not None
        ctx.params = (w, b)
        ctx.x_amax = x.__dict__.get('_pcgan_amax')     # (see _Conv2dFn)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.cfg
        dy = _c(dy)
        if ctx.x_amax is not None and ctx.x_amax[0] == x._version and '_pcgan_amax' not in x.__dict__:
            x._pcgan_amax = ctx.x_amax
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        wt = _fused_grad_target(ctx.params[0]) if want_w else None
        bt = _fused_grad_target(ctx.params[1]) if want_b else None
        if ops.SIDE_STREAM and (not want_w or wt is not None) and (not want_b or bt is not None) and (want_w or want_b):
            # as in _Conv2dFn: both go straight into the optimizer's gradient buffer, on the parameter-gradient stream (all
            # accumulations into that buffer are issued in order on the one stream)
            with ops.fork_side(x, dy):
                if want_w:
                    ops.conv2d_bwd_weight(dy, x, tuple(w.shape), stride, pad, 0, accumulate_into=wt)
                if want_b:
                    ops.channel_sum(dy, accumulate_into=bt)
            if ctx.needs_input_grad[0]:
                dx = ops.conv2d_fwd(dy, w, None, stride, pad, 0, pack_cache=ctx.pack)
            return dx, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_fwd(dy, w, None, stride, pad, 0, pack_cache=ctx.pack)
        if want_w:
            dw = ops.conv2d_bwd_weight(dy, x, tuple(w.shape), stride, pad, 0, accumulate_into=wt)
            if wt is not None:
                dw = None
        if want_b:
            db = ops.channel_sum(dy, accumulate_into=bt)
            if bt is not None:
                db = None
        return dx, dw, db, None, None, None


def conv_transpose2d(x, w, b=None, stride=2, pad=1, out_pad=1):
    return _ConvTranspose2dFn.apply(x, w, b, stride, pad, out_pad)


# ---------------------------------------------------------------------------- norms
class _InstanceNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, running_mean, running_var, momentum, eps, act, slope, training):
        x, residual = _c(x), _c(residual)
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        if training or running_mean is None:
            y, mean, var = ops.instnorm_fwd(x, residual, eps, act, slope)    # var holds the plane M2
            per_plane = True
            if training and running_mean is not None:
                ops.in_running_update(mean, var, running_mean, running_var, N, C, HW, momentum)
        else:
            mean, var, per_plane = running_mean, running_var, False
            y = ops.norm_act_fwd(x, mean, var, None, None, residual, per_plane, eps, act, slope)
        ctx.cfg = (eps, act, slope, per_plane, residual is not None)
        ctx.save_for_backward(x, y if act != ACT_NONE else None, mean, var)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var = ctx.saved_tensors
        eps, act, slope, per_plane, has_res = ctx.cfg
        if not per_plane:
            raise NotImplementedError('pcgan_amd: backward through eval-mode InstanceNorm is not on the hot path')
        dy = _c(dy)
        dx = dres = None
        if ctx.needs_input_grad[0]:
            dx = ops.instnorm_bwd(dy, x, y, mean, var, eps, act, slope)
        if has_res and ctx.needs_input_grad[1]:
            dres = dy if act == ACT_NONE else ops.act_bwd(dy, y, act, slope)
        return dx, dres, None, None, None, None, None, None, None


def instance_norm_act(x, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, act=ACT_NONE, slope=0.0,
                      residual=None, training=True):
    """act( InstanceNorm2d(affine=False)(x) + residual ), running statistics updated in place."""
    return _InstanceNormActFn.apply(x, residual, running_mean, running_var, momentum, eps, act, slope, training)


class _BatchNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, act, slope, training, batches, tickets=None):
        x, residual = _c(x), _c(residual)
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        fused = training and N * HW <= ops.BN_FUSED_MAX and N * HW > 1
        if fused:      # one launch: statistics + running update + batch counter + normalise / activation
            y, mean, var = ops.bn_fwd_fused(x, gamma, beta, residual, running_mean, running_var, batches, momentum, eps, act, slope)
        else:
            if training and tickets is not None and N * HW > 1:
                # statistics + Chan merge + running statistics + batch counter in one launch (last-arriver merge, csrc/norm.hip)
                mean, var = ops.bn_stats_merged(x, running_mean, running_var, batches, tickets[0], momentum)
            elif training:
                mean_nc, m2_nc = ops.plane_stats(x)
                mean, var = ops.bn_merge(mean_nc, m2_nc, N, C, HW, running_mean, running_var, momentum)
                if batches is not None:
                    batches.add_(1)
            else:
                mean, var = running_mean, running_var
            y = ops.norm_act_fwd(x, mean, var, gamma, beta, residual, False, eps, act, slope)
        ctx.cfg = (eps, act, slope, training, residual is not None, fused)
        ctx.tickets = tickets
        ctx.save_for_backward(x, y if act != ACT_NONE else None, mean, var, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, gamma = ctx.saved_tensors
        eps, act, slope, training, has_res, fused = ctx.cfg
        if not training:
            raise NotImplementedError('pcgan_amd: backward through eval-mode BatchNorm is not on the hot path')
        dy = _c(dy)
        N, C = x.shape[0], x.shape[1]
        want_res = has_res and ctx.needs_input_grad[3]
        want_dx = ctx.needs_input_grad[0]
        dx = dres = None
        if fused:
            dx, dres, s1, s2 = ops.bn_bwd_fused(dy, x, y, mean, var, gamma, eps, act, slope, want_dx,
                                                want_res and act != ACT_NONE)
        else:
            if ctx.tickets is not None:
                s1, s2 = ops.bn_bwd_stats_reduced(dy, x, y, mean, var, eps, act, slope, ctx.tickets[1])
            else:
                s1n, s2n = ops.norm_bwd_stats(dy, x, y, mean, var, False, eps, act, slope)
                s1, s2 = ops.bn_bwd_reduce(s1n, s2n, N, C)
            if want_dx or (want_res and act != ACT_NONE):
                dx, dres = ops.norm_bwd_apply(dy, x, y, mean, var, gamma, s1, s2, False, eps, act, slope,
                                              want_res and act != ACT_NONE)
        if want_res and dres is None:
            dres = dy
        dgamma = s2 if ctx.needs_input_grad[1] else None
        dbeta = s1 if ctx.needs_input_grad[2] else None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, act=ACT_NONE, slope=0.0,
                   residual=None, training=True, batches=None, tickets=None):
    """act( BatchNorm2d(affine=True)(x) + residual ) with batch statistics in train mode; `batches` = the module's
    num_batches_tracked counter (incremented on the device, inside the fused kernel when the tensor is small)."""
    return _BatchNormActFn.apply(x, gamma, beta, residual, running_mean, running_var, momentum, eps, act, slope,
                                 training, batches, tickets)


# ---------------------------------------------------------------------------- pointwise
class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        y = ops.act_fwd(_c(x), act, slope)
        ctx.cfg = (act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, slope = ctx.cfg
        return ops.act_bwd(_c(dy), y, act, slope), None, None


def activation(x, act, slope=0.0):
    return x if act == ACT_NONE else _ActFn.apply(x, act, slope)


class _ConcatZFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, z):
        img = _c(img)
        z2 = _c(z.reshape(z.size(0), z.size(1)))
        ctx.shapes = (img.shape, z.shape)
        return ops.concat_z(img, z2)

    @staticmethod
    def backward(ctx, dy):
        ishape, zshape = ctx.shapes
        C = ishape[1]
        dimg = dz = None
        if ctx.needs_input_grad[0]:
            dimg = dy[:, :C].contiguous()        # strided device copy (memory plumbing)
        if ctx.needs_input_grad[1]:
            dzc = dy[:, C:].contiguous()
            N, nz = dzc.shape[0], dzc.shape[1]
            # per-(n, j) plane sums: reuse the channel-sum kernel on a [1][N*nz][HW] view
            s = ops.channel_sum(dzc.view(1, N * nz, dzc.shape[2], dzc.shape[3]))
            s = s.view(N, nz)
            if zshape[0] == 1:
                s = ops.channel_sum(s.t().contiguous().view(1, nz, N, 1)).view(1, nz)
            dz = s.reshape(zshape)
        return dimg, dz


def concat_z(img, z):
    """torch.cat((img, z broadcast over H,W), 1); z is (B or 1, nz, 1, 1)."""
    return _ConcatZFn.apply(img, z)


class _ChannelScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask_nc, scale):
        ctx.scale = scale
        ctx.save_for_backward(mask_nc)
        return ops.channel_scale(_c(x), mask_nc, scale)

    @staticmethod
    def backward(ctx, dy):
        (mask_nc,) = ctx.saved_tensors
        return ops.channel_scale(_c(dy), mask_nc, ctx.scale), None, None


def dropout2d(x, p, training=True, mask=None):
    """nn.Dropout2d: whole (n,c) planes zeroed with probability p, survivors scaled by 1/(1-p).
    `mask` (N*C keep flags) can be injected for parity tests; otherwise it is drawn with
    torch's device RNG (host-side RNG plumbing, cannot match a CPU stream anyway)."""
    if not training or p <= 0.0:
        return x
    N, C = x.shape[0], x.shape[1]
    if mask is None:
        mask = _keep_flags(N * C, p, x.device)
    return _ChannelScaleFn.apply(x, mask.reshape(-1).contiguous(), 1.0 / (1.0 - p))


# Keep flags are independent Bernoulli(1 - p) draws from torch's device generator: drawn 65536 at a time and handed out in
# consecutive slices (a Bayesian step calls Dropout2d 540 times -- 18 sites x 30 encoder passes -- with 100-4000 flags each: one
# bernoulli_ launch per ~2 encoder passes instead of one per call).  Slices are never reused.
_FLAG_POOL = {}
_FLAG_POOL_SIZE = 1 << 16


def _keep_flags(n, p, device):
    if n > _FLAG_POOL_SIZE // 4:
        return torch.empty(n, dtype=torch.float32, device=device).bernoulli_(1.0 - p)
    # A re-seed (torch.manual_seed) must not be followed by leftover flags of the old stream: the pool remembers the generator's
    # seed and its Philox offset right after the refill; a different seed, or an offset that went BACKWARDS (same seed again),
    # means the generator was re-seeded since -> refill, so a seeded MC-dropout run is reproducible from its seed.
    cuda = device.type == 'cuda'
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()] if cuda else torch.default_generator
    seed, offset = gen.initial_seed(), (gen.get_offset() if cuda else 0)
    key = (device, float(p), torch.cuda.current_stream(device).cuda_stream if cuda else 0)
    ent = _FLAG_POOL.get(key)
    if ent is None or ent[1] + n > _FLAG_POOL_SIZE or ent[2] != seed or offset < ent[3]:
        buf = torch.empty(_FLAG_POOL_SIZE, dtype=torch.float32, device=device).bernoulli_(1.0 - p)
        ent = _FLAG_POOL[key] = [buf, 0, seed, gen.get_offset() if cuda else 0]
    o = ent[1]
    ent[1] = o + n
    return ent[0][o:o + n]


def reset_flag_pool():
    """drop every pre-drawn keep flag (call after re-seeding by other means than torch.manual_seed, e.g. set_rng_state)"""
    _FLAG_POOL.clear()


# ---------------------------------------------------------------------------- pooling / resize
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        x = _c(x)
        y, arg = ops.maxpool_fwd(x, k, stride, pad)
        ctx.cfg = (tuple(x.shape[2:]), k, stride, pad)
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        hw, k, stride, pad = ctx.cfg
        return ops.maxpool_bwd(_c(dy), arg, hw, k, stride, pad), None, None, None


def max_pool2d(x, k, stride, pad=0):
    return _MaxPoolFn.apply(x, k, stride, pad)


class _GlobalPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, is_max):
        x = _c(x)
        y, arg = ops.global_pool_fwd(x, is_max)
        ctx.cfg = (tuple(x.shape[2:]), is_max)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        hw, is_max = ctx.cfg
        return ops.global_pool_bwd(_c(dy), arg, hw, is_max), None


def global_pool(x, is_max):
    """nn.AvgPool2d(H) / nn.MaxPool2d(H) over the whole plane."""
    return _GlobalPoolFn.apply(x, bool(is_max))


class _BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        x = _c(x)
        ctx.hw = tuple(x.shape[2:])
        return ops.bilinear_fwd(x, (size, size))

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear_bwd(_c(dy), ctx.hw), None


def upsample2d(x, size):
    """util.upsample2d (util/util.py:111-117): identity if size <= 0 or already that size."""
    if size <= 0 or x.size(2) == size:
        return x
    return _BilinearFn.apply(x, int(size))


# ---------------------------------------------------------------------------- storage casts (bf16 path boundary)
class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return ops.cast(_c(x), dtype)

    @staticmethod
    def backward(ctx, dy):
        return ops.cast(_c(dy), ctx.src), None


def cast(x, dtype):
    """x in another activation storage type (torch.float32 / torch.bfloat16), differentiable; identity if it already is"""
    return x if x.dtype == dtype else _CastFn.apply(x, dtype)


# ---------------------------------------------------------------------------- losses
class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, a, b):
        a, b = _c(a), _c(b)
        want = a.requires_grad
        if kind == 'bce':
            loss, grad = ops.bce_loss(a, b, want)
        elif kind == 'l1':
            loss, grad = ops.l1_loss(a, b, want)
        else:
            loss, grad = ops.mse_loss(a, b, want)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, dl):
        (grad,) = ctx.saved_tensors
        if grad is None:
            return None, None, None
        # d loss / d a, scaled by the upstream scalar which stays on the device
        return None, ops.scale(grad, _c(dl).reshape(1)), None


def bce_loss(pred, target_n):
    """nn.BCELoss(mean) of pred[N,...] against a per-sample target (float tensor [N])."""
    return _LossFn.apply('bce', pred, target_n)


def l1_loss(a, b):
    return _LossFn.apply('l1', a, b)


def mse_loss(a, b):
    return _LossFn.apply('mse', a, b)
