"""BaseModel -- the plugin base class every model derives from; method and attribute surface of the reference
(models/base_model.py:7-159), organised for one-process-per-GPU execution:

  * networks stay on their GPU while checkpoints are written (the reference moves them to the CPU and back, which
    would tear the parameters out of the fused optimizer's flat buffer);
  * under torch.distributed only rank 0 writes checkpoints;
  * checkpoint file names and state_dict key layout are the reference's (`<which_epoch>_net_<name>.pth`,
    SURVEY.md appendix B), so files interchange in both directions.

Conventions a subclass relies on: networks are attributes `net<X>` listed in `model_names`, scalar losses are attributes
`loss_<name>` listed in `loss_names`, images to show are attributes listed in `visual_names`, optimizers go in
`self.optimizers`.
"""
import os
from collections import OrderedDict

import torch

from . import networks
from ..hip import ops as hip_ops
from ..hip import parallel

_STALE_NORM_KEYS = ('running_mean', 'running_var', 'num_batches_tracked')


class BaseModel(object):
    # ------------------------------------------------------------------ plugin hooks (overridden by the models)
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def name(self):
        return 'BaseModel'

    def to_act(self, image):
        """a loader image batch on the device in the model's activation storage type (one cast kernel under --dtype bf16).  Copies and
        casts run on an upload stream and the result carries its readiness event (hip/ops.py: ready_event), so that consumers which do
        not depend on the step still in flight (the frozen encoder's passes over the new batch) need not queue behind it."""
        from ..hip import ops
        if not (isinstance(image, torch.Tensor) and self.device.type == 'cuda'):
            return image.to(self.device)
        if image.is_cuda and image.dtype == self.act_dtype:
            return image      # resident already: ready when its producer said so (ops.mark_ready), else in plain stream order
        cur = torch.cuda.current_stream(self.device)
        up = ops.upload_stream(self.device)
        src_ev = ops.ready_event(image) if image.is_cuda else None
        with torch.cuda.stream(up):
            if src_ev is not None:
                up.wait_event(src_ev)
                image.record_stream(up)
            out = image.to(self.device, non_blocking=True)
            if out.dtype != self.act_dtype:
                out = ops.cast(out.contiguous(), self.act_dtype)
            ev = torch.cuda.Event()
            ev.record(up)
        out._pcgan_ready = (out._version, ev)      # (= ops.mark_ready(out, up) with the event recorded inside the stream context)
        cur.wait_event(ev)
        out.record_stream(cur)
        return out

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def optimize_parameters(self):
        pass

    # ------------------------------------------------------------------ construction
    def initialize(self, opt):
        self.opt = opt
        self.isTrain = opt.isTrain
        self.gpu_ids = opt.gpu_ids
        self.use_gpu = bool(opt.gpu_ids) and torch.cuda.is_available()
        self.device = torch.device('cuda:%d' % opt.gpu_ids[0]) if opt.gpu_ids else torch.device('cpu')
        # storage type of activations (build-only --dtype: fp32 | bf16); parameters, statistics, losses, Adam: always fp32
        self.act_dtype = torch.bfloat16 if getattr(opt, 'dtype', 'fp32') == 'bf16' else torch.float32
        self.Tensor = torch.cuda.FloatTensor if self.use_gpu else torch.Tensor
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        for attr in ('loss_names', 'model_names', 'load_model_names', 'visual_names', 'image_paths', 'optimizers'):
            setattr(self, attr, [])
        self.current_iter = 0
        self.current_batch_size = opt.batchSize

    def setup(self, opt, parser=None):
        """learning-rate schedulers (training); weights from disk (testing or --continue_train); network summary"""
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(o, opt) for o in self.optimizers]
        if not self.isTrain or opt.continue_train:
            self.load_networks(opt.which_epoch)
        self.print_networks(opt.verbose)

    # ------------------------------------------------------------------ networks by name
    def _nets(self, names=None):
        """(name, module) of every string entry of `names` (default: model_names)"""
        for name in (self.model_names if names is None else names):
            if isinstance(name, str):
                yield name, getattr(self, 'net' + name)

    def _checkpoint(self, which_epoch, name):
        return os.path.join(self.save_dir, '%s_net_%s.pth' % (which_epoch, name))

    def eval(self):
        for _, net in self._nets():
            net.eval()

    def set_requires_grad(self, nets, requires_grad=False):
        for net in (nets if isinstance(nets, list) else [nets]):
            if net is None:
                continue
            for p in net.parameters():
                p.requires_grad = requires_grad

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name, net in self._nets():
            if verbose:
                print(net)
            count = sum(p.numel() for p in net.parameters())
            print('[Network %s] Total number of parameters : %.3f M' % (name, count / 1e6))
        print('-----------------------------------------------')

    # ------------------------------------------------------------------ checkpoints
    def sync_parameter_updates(self):
        """order the current stream behind optimizer updates still queued on the parameter-gradient stream (hip/optim.py:
        step_on_grad_stream) -- called wherever parameters are read outside the training step (checkpoints, visuals, tests)"""
        if torch.cuda.is_available():
            hip_ops.join_side_stream(force=True)

    def save_networks(self, which_epoch):
        if parallel.is_distributed() and torch.distributed.get_rank() != 0:
            return
        self.sync_parameter_updates()
        os.makedirs(self.save_dir, exist_ok=True)
        for name, net in self._nets():
            weights = getattr(net, 'module', net).state_dict()
            torch.save(OrderedDict((k, v.detach().cpu()) for k, v in weights.items()), self._checkpoint(which_epoch, name))

    def load_networks(self, which_epoch):
        self.sync_parameter_updates()
        for name, net in self._nets(self.load_model_names or self.model_names):
            net = getattr(net, 'module', net)
            path = self._checkpoint(which_epoch, name)
            print('loading the model from %s' % path)
            state_dict = torch.load(path, map_location='cpu')
            # checkpoints written before PyTorch 0.4 carry InstanceNorm statistics the module may not track: drop the
            # ones this net has no slot for (reference models/base_model.py:109-117, 134-135)
            slots = net.state_dict()
            for key in [k for k in state_dict if k not in slots and k.rsplit('.', 1)[-1] in _STALE_NORM_KEYS]:
                del state_dict[key]
            net.load_state_dict(state_dict)
        from ..hip import ops
        ops.invalidate_packed_weights()

    # ------------------------------------------------------------------ what the training loop reads
    def test(self):
        with torch.no_grad():
            self.forward()

    def get_image_paths(self):
        return self.image_paths

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            scheduler.step()
        print('learning rate = %.7f' % self.optimizers[0].param_groups[0]['lr'])

    def get_current_visuals(self):
        return OrderedDict((name, getattr(self, name)) for name in self.visual_names if isinstance(name, str))

    def get_current_losses(self):
        """float(...) of every loss_<name>: the only device->host synchronisation of the loop (train.py reads it every
        print_freq iterations)."""
        out = OrderedDict((name, float(getattr(self, 'loss_' + name).detach() if isinstance(getattr(self, 'loss_' + name), torch.Tensor)
                                        else getattr(self, 'loss_' + name))) for name in self.loss_names if isinstance(name, str))
        hip_ops.check_nonfinite('get_current_losses')      # the fp16 route's overflow sentinel: the device is in sync here anyway
        return out
