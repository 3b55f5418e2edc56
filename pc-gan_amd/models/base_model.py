"""BaseModel -- the plugin base class of the reference (models/base_model.py:7-159), same
method and attribute surface, rebuilt for one-process-per-GPU execution:

  * networks stay on their GPU while checkpoints are written (the reference moves them to the
    CPU and back, which would tear the parameters out of the fused optimizer's flat buffer);
  * only rank 0 writes checkpoints under torch.distributed;
  * checkpoint file names and state_dict key layout are the reference's
    (`<which_epoch>_net_<name>.pth`), so files are interchangeable.
"""
import os
from collections import OrderedDict

import torch

from . import networks
from ..hip import parallel


class BaseModel():
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def name(self):
        return 'BaseModel'

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.use_gpu = len(opt.gpu_ids) > 0 and torch.cuda.is_available()
        self.Tensor = torch.cuda.FloatTensor if self.use_gpu else torch.Tensor
        self.isTrain = opt.isTrain
        self.device = torch.device('cuda:{}'.format(self.gpu_ids[0])) if self.gpu_ids else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names = []
        self.model_names = []
        self.load_model_names = []
        self.visual_names = []
        self.image_paths = []
        self.optimizers = []
        self.current_iter = 0
        self.current_batch_size = opt.batchSize

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def setup(self, opt, parser=None):
        """create schedulers; load networks when testing / continuing; print networks"""
        if self.isTrain:
            self.schedulers = [networks.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
        if not self.isTrain or opt.continue_train:
            self.load_networks(opt.which_epoch)
        self.print_networks(opt.verbose)

    def _nets(self, names=None):
        for name in (self.model_names if names is None else names):
            if isinstance(name, str):
                yield name, getattr(self, 'net' + name)

    def eval(self):
        for _, net in self._nets():
            net.eval()

    def test(self):
        with torch.no_grad():
            self.forward()

    def get_image_paths(self):
        return self.image_paths

    def optimize_parameters(self):
        pass

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        print('learning rate = %.7f' % lr)

    def get_current_visuals(self):
        ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str):
                ret[name] = getattr(self, name)
        return ret

    def get_current_losses(self):
        """float(...) of every loss_<name>: the only device->host synchronisation of the loop
        (every print_freq iterations in train.py)."""
        ret = OrderedDict()
        for name in self.loss_names:
            if isinstance(name, str):
                ret[name] = float(getattr(self, 'loss_' + name))
        return ret

    def save_networks(self, which_epoch):
        if parallel.is_distributed() and torch.distributed.get_rank() != 0:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        for name, net in self._nets():
            net = getattr(net, 'module', net)
            path = os.path.join(self.save_dir, '%s_net_%s.pth' % (which_epoch, name))
            torch.save(OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict().items()), path)

    def load_networks(self, which_epoch):
        names = self.load_model_names if len(self.load_model_names) > 0 else self.model_names
        for name, net in self._nets(names):
            net = getattr(net, 'module', net)
            path = os.path.join(self.save_dir, '%s_net_%s.pth' % (which_epoch, name))
            print('loading the model from %s' % path)
            state_dict = torch.load(path, map_location='cpu')
            # pre-0.4 InstanceNorm checkpoints carry running stats the module may not track
            # (reference models/base_model.py:109-117,134-135)
            own = net.state_dict()
            for key in list(state_dict.keys()):
                if key not in own and key.rsplit('.', 1)[-1] in ('running_mean', 'running_var',
                                                                 'num_batches_tracked'):
                    state_dict.pop(key)
            net.load_state_dict(state_dict)
        from ..hip import ops
        ops.invalidate_packed_weights()

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name, net in self._nets():
            n = sum(p.numel() for p in net.parameters())
            if verbose:
                print(net)
            print('[Network %s] Total number of parameters : %.3f M' % (name, n / 1e6))
        print('-----------------------------------------------')

    def set_requires_grad(self, nets, requires_grad=False):
        if not isinstance(nets, list):
            nets = [nets]
        for net in nets:
            if net is not None:
                for param in net.parameters():
                    param.requires_grad = requires_grad
