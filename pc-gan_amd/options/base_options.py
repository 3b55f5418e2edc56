"""Option parser -- same flags, defaults and two-pass structure as the reference
(options/base_options.py:10-143): base flags, then the model's modify_commandline_options,
then the dataset's; `parse()` resolves gpu_ids / attr_bins and writes opt_<phase>.txt and
cmd_<phase>.txt under <checkpoints_dir>/<name>/.

Build-only additions (do not change any existing name): --seed, --local_rank, --gpu_transform.  Under torchrun
(LOCAL_RANK set) gpu_ids becomes [LOCAL_RANK]: one process drives one GPU.
"""
import argparse
import os
import sys

import torch

from .. import models
from .. import data
from ..util import util

_INF = float('inf')

# (flag, kwargs); help strings abbreviated
_BASE_FLAGS = [
    ('--dataroot', dict(required=True, help='path to images')),
    ('--sourcefile_A', dict(type=str, default='')),
    ('--sourcefile_B', dict(type=str, default='')),
    ('--batchSize', dict(type=int, default=1, help='input batch size')),
    ('--loadSize', dict(type=int, default=286, help='scale images to this size')),
    ('--fineSize', dict(type=int, default=256, help='then crop to this size')),
    ('--input_nc', dict(type=int, default=3)),
    ('--output_nc', dict(type=int, default=3)),
    ('--ngf', dict(type=int, default=64)),
    ('--ndf', dict(type=int, default=64)),
    ('--which_model_netD', dict(type=str, default='basic')),
    ('--which_model_netG', dict(type=str, default='resnet_9blocks')),
    ('--n_layers_G', dict(type=int, default=7)),
    ('--n_layers_D', dict(type=int, default=3)),
    ('--gpu_ids', dict(type=str, default='0', help='e.g. 0 ; -1 for CPU (host plumbing only: no HIP path)')),
    ('--name', dict(type=str, default='experiment_name')),
    ('--dataset_mode', dict(type=str, default='unaligned')),
    ('--model', dict(type=str, default='cycle_gan')),
    ('--which_direction', dict(type=str, default='AtoB')),
    ('--nThreads', dict(default=4, type=int)),
    ('--checkpoints_dir', dict(type=str, default='./checkpoints')),
    ('--norm', dict(type=str, default='instance')),
    ('--serial_batches', dict(action='store_true')),
    ('--display_winsize', dict(type=int, default=256)),
    ('--display_id', dict(type=int, default=1)),
    ('--display_server', dict(type=str, default='http://localhost')),
    ('--display_port', dict(type=int, default=8097)),
    ('--dropout', dict(type=float, default=0)),
    ('--max_dataset_size', dict(type=int, default=_INF)),
    ('--transforms', dict(type=str, default='resize_and_crop')),
    ('--affineScale', dict(nargs='+', type=float, default=[0.95, 1.05])),
    ('--affineDegrees', dict(type=float, default=5)),
    ('--use_color_jitter', dict(action='store_true')),
    ('--no_flip', dict(action='store_true')),
    ('--init_type', dict(type=str, default='normal')),
    ('--verbose', dict(action='store_true')),
    ('--suffix', dict(default='', type=str)),
    ('--num_Ds', dict(type=int, default=2)),
    ('--nl', dict(type=str, default='relu')),
    ('--upsample', dict(type=str, default='basic')),
    ('--num_classes', dict(type=int, default=None)),
    ('--attr_bins', dict(type=str, default='[]')),
    ('--load_model_names', dict(type=str, nargs='+', default=[])),
    ('--sorted', dict(action='store_true')),
    # build-only
    ('--seed', dict(type=int, default=None, help='(pcgan_amd) seed torch/numpy if given')),
    ('--dtype', dict(type=str, default='fp32', choices=['fp32', 'bf16'],
                     help='(pcgan_amd) storage type of activations and their gradients: fp32, or bf16 with fp32 master '
                          'weights / accumulation / statistics / losses / Adam (BASELINE config 3)')),
    ('--local_rank', dict(type=int, default=None, help='(pcgan_amd) set by launchers; LOCAL_RANK env wins')),
    ('--gpu_transform', dict(action='store_true', help='(pcgan_amd) resize / crop / flip / normalise on the GPU '
                                                       '(bit-exact with the PIL path); workers only decode')),
]


class BaseOptions():
    def __init__(self):
        self.initialized = False

    def initialize(self, parser):
        for flag, kw in _BASE_FLAGS:
            parser.add_argument(flag, **kw)
        self.initialized = True
        return parser

    def gather_options(self):
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        parser = self.initialize(parser)
        opt, _ = parser.parse_known_args()
        parser = models.get_option_setter(opt.model)(parser, self.isTrain)
        opt, _ = parser.parse_known_args()          # again, with the model's new defaults
        parser = data.get_option_setter(opt.dataset_mode)(parser, self.isTrain)
        self.parser = parser
        return parser.parse_args()

    def print_options(self, opt):
        lines = ['----------------- Options ---------------']
        for k, v in sorted(vars(opt).items()):
            default = self.parser.get_default(k)
            comment = '\t[default: %s]' % str(default) if v != default else ''
            lines.append('{:>25}: {:<30}{}'.format(str(k), str(v), comment))
        lines.append('----------------- End -------------------')
        message = '\n'.join(lines)
        print(message)
        if int(os.environ.get('RANK', '0')) != 0:
            return
        expr_dir = os.path.join(opt.checkpoints_dir, opt.name)
        util.mkdirs(expr_dir)
        with open(os.path.join(expr_dir, 'opt_%s.txt' % opt.phase), 'wt') as f:
            f.write(message + '\n')
        with open(os.path.join(expr_dir, 'cmd_%s.txt' % opt.phase), 'wt') as f:
            vis = os.getenv('HIP_VISIBLE_DEVICES') or os.getenv('CUDA_VISIBLE_DEVICES')
            if vis:
                f.write('HIP_VISIBLE_DEVICES=%s ' % vis)
            f.write(' '.join(sys.argv) + '\n')

    def parse(self):
        opt = self.gather_options()
        opt.isTrain = self.isTrain
        if opt.suffix:
            opt.name = opt.name + '_' + opt.suffix.format(**vars(opt))
        self.print_options(opt)

        ids = [int(s) for s in opt.gpu_ids.split(',')]
        opt.gpu_ids = [i for i in ids if i >= 0]
        if 'LOCAL_RANK' in os.environ and len(opt.gpu_ids) > 0 and int(os.environ.get('WORLD_SIZE', '1')) > 1:
            opt.gpu_ids = [int(os.environ['LOCAL_RANK']) % max(1, torch.cuda.device_count())]   # one process <-> one GPU
        if len(opt.gpu_ids) > 1:
            raise RuntimeError('pcgan_amd runs one process per GPU: launch with torchrun --nproc-per-node %d '
                               'instead of --gpu_ids %s' % (len(opt.gpu_ids), ','.join(map(str, opt.gpu_ids))))
        if len(opt.gpu_ids) > 0:
            torch.cuda.set_device(opt.gpu_ids[0])
        opt.attr_bins = util.str2list(opt.attr_bins)
        if opt.seed is not None:
            import numpy as np
            import random
            random.seed(opt.seed)
            np.random.seed(opt.seed)
            torch.manual_seed(opt.seed)
        self.opt = opt
        return self.opt
