"""Fused Adam over one flat fp32 buffer per optimizer.

All parameters of a network are re-homed as views into one contiguous buffer (and their
.grad as views into a twin buffer), so that
  * the whole update is ONE HBM-bound kernel launch (pcgan_adam_step_dev: 4 reads + 3 writes
    per element) instead of ~60 per-tensor launches, and
  * the data-parallel gradient exchange is ONE RCCL all-reduce per optimizer
    (pcgan_amd.hip.parallel).
Learning rate and step counter live in device memory so the step is hipGraph-capturable.

Semantics: torch.optim.Adam(lr, betas=(beta1, 0.999), eps=1e-8), no weight decay / amsgrad
(reference models/wsgan_emb_model.py:153-163).
"""
import torch

from . import ops

_ALIGN = 64  # floats; keeps every parameter 256-byte aligned inside the flat buffer


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        if not params:
            raise ValueError('FusedAdam: empty parameter list')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        ps = self.param_groups[0]['params']
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError('FusedAdam: all parameters must be fp32 on one device')
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._views = []
        with torch.no_grad():
            for p, o in zip(ps, offs):
                n = p.numel()
                self.flat[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                gv = self.gflat[o:o + n].view(p.shape)
                p.grad = gv
                p._pcgan_fused_grad = True     # conv backward accumulates straight into this view
                self._views.append((p, gv))
        self.numel = sum(p.numel() for p in ps)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self._lr_host = float(lr)
        self._step_host = 0
        self.updated = None       # event behind the last update when it ran on the gradient stream (step_on_grad_stream)

    def zero_grad(self, set_to_none=False):
        """Zero the flat gradient buffer (one memset) and keep every .grad a view into it so that
        autograd accumulates in place."""
        if self.updated is not None and self.gflat.is_cuda:
            # the last update ran ON the parameter-gradient stream (behind every kernel that wrote these gradients): waiting for IT is
            # enough -- a full join would stall this stream behind the other optimizer's weight gradients still in flight there
            torch.cuda.current_stream(self.gflat.device).wait_event(self.updated)
        else:
            ops.join_side_stream()
        self.gflat.zero_()
        for p, gv in self._views:
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv

    def rebind(self):
        """Re-home parameters whose storage was replaced behind our back (e.g. net.to())."""
        with torch.no_grad():
            o = 0
            for p, gv in self._views:
                n = p.numel()
                view = self.flat[o:o + n].view(p.shape)
                if p.data.data_ptr() != view.data_ptr():
                    view.copy_(p.data)
                    p.data = view
                o += (n + _ALIGN - 1) // _ALIGN * _ALIGN

    def step_on_grad_stream(self, before=None):
        """The same update, queued on the parameter-gradient stream (`before`: a callable run on that stream first -- data parallel: the
        all-reduce of the flat gradient buffer, which then also runs off the main stream): behind the weight- / bias-gradient kernels that stream still holds
        and behind everything the current stream has queued so far (gradients autograd accumulated there: BatchNorm's affine
        parameters).  The current stream does NOT wait: it goes on with work that needs neither these gradients nor the new weights
        (wsgan_emb: backward_D after the generator's update, the next step's forward after the discriminator's).  `self.updated` is the
        event consumers of the new weights wait for.  Same kernel, same inputs, same order on that stream: bit-identical results."""
        cur = torch.cuda.current_stream(self.flat.device)
        side = ops.side_stream_for(cur)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if before is not None:
                before()
            self.step(_joined=True)
            ev = torch.cuda.Event()
            ev.record(side)
        self.updated = ev
        ops._side_state['dirty'] = True
        return ev

    @torch.no_grad()
    def step(self, closure=None, _joined=False):
        if not _joined:
            ops.join_side_stream(force=True)      # parameter-gradient kernels run on a side stream
            self.updated = None
        g = self.param_groups[0]
        lr = float(g['lr'])
        if lr != self._lr_host:           # scheduler changed it (once per epoch)
            self.lr_dev.fill_(lr)
            self._lr_host = lr
        for p, gv in self._views:
            if p.grad is not None and p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)          # somebody replaced .grad; fold it back
                p.grad = gv
        if self.flat.is_cuda:
            ops.adam_step_dev(self.flat, self.gflat, self.exp_avg, self.exp_avg_sq, self.lr_dev, self.step_dev,
                              g['betas'][0], g['betas'][1], g['eps'], params=[p for p, _ in self._views])
        else:
            raise RuntimeError('FusedAdam: parameters are on %s; the HIP path needs a GPU (no CPU fallback)'
                               % self.flat.device)
        self._step_host += 1
