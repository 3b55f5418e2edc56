"""WSGANCycleModel -- the reference's attribute-conditioned cycle model (models/wsgan_cycle_model.py:13-269) on the
HIP path: G(image, attribute) with an unconditional PatchGAN, an AlexNet identity term and an encoder E that is TRAINED
here (it regresses the attribute back: x -> y -> x and y -> x -> y cycles).  SURVEY.md 8(f) rank 1 / config 5.

Same plugin surface as the reference (flags, defaults, loss / visual / model names, step order: D first, then G and E
together).  Differences, all stated where they occur: generators outside the hot path (the reference's default
`unet_128`) raise NotImplementedError from define_G -- pass `--which_model_netG resnet_9blocks`; `--use_bicycle_E`
(another encoder family) raises; data parallelism is one process per GPU + RCCL all-reduce of the three flat
gradient buffers instead of nn.DataParallel.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import networks
from .base_model import BaseModel
from ..hip import ops as hip_ops
from ..hip import parallel
from ..hip.optim import FusedAdam
from ..util.util import upsample2d

_COMMON_FLAGS = [   # reference models/wsgan_cycle_model.py:18-32
    ('--norm_G', dict(type=str, default='instance')),
    ('--norm_D', dict(type=str, default='batch')),
    ('--embedding_nc', dict(type=int, default=1)),
    ('--which_model_netE', dict(type=str, default='resnet18')),
    ('--use_bicycle_E', dict(action='store_true')),
    ('--pooling_E', dict(type=str, default='max')),
    ('--cnn_dim_E', dict(type=int, nargs='+', default=[64, 1])),
    ('--cnn_pad_E', dict(type=int, default=1)),
    ('--cnn_relu_slope_E', dict(type=float, default=0.7)),
    ('--fineSize_E', dict(type=int, default=224)),
    ('--pretrained_model_path_E', dict(type=str, default='pretrained_models/resnet18-5c106cde.pth')),
    ('--attr_mean', dict(type=float, nargs='*', default=[0.0])),
    ('--attr_std', dict(type=float, nargs='*', default=[100])),
    ('--display_visuals', dict(action='store_true')),
]
_TRAIN_FLAGS = [    # :33-41
    ('--lambda_x', dict(type=float, default=1.0)),
    ('--lambda_y', dict(type=float, default=1.0)),
    ('--lambda_IP', dict(type=float, default=1.0)),
    ('--which_model_netIP', dict(type=str, default='alexnet')),
    ('--fineSize_IP', dict(type=int, default=224)),
    ('--pretrained_model_path_IP', dict(type=str, default='pretrained_models/alexnet-owt-4df8aa71.pth')),
    ('--no_trick', dict(action='store_true')),
    ('--identity_preserving_criterion', dict(type=str, default='mse')),
]
_DEFAULT_OVERRIDES = dict(pool_size=0, no_lsgan=True, norm='instance', dataset_mode='wsgan_cycle',
                          which_model_netG='unet_128', which_model_netD='n_layers', n_layers_D=4, batchSize=10,
                          loadSize=140, fineSize=128, display_visuals=True, save_epoch_freq=2)   # :44-55


class WSGANCycleModel(BaseModel):
    def name(self):
        return 'WSGANCycleModel'

    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        for flag, kw in _COMMON_FLAGS + (_TRAIN_FLAGS if is_train else []):
            parser.add_argument(flag, **kw)
        parser.set_defaults(**_DEFAULT_OVERRIDES)
        return parser

    # ------------------------------------------------------------------ construction (reference :59-146)
    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        assert opt.input_nc == opt.output_nc
        assert opt.embedding_nc == 1
        self.attr_bins = opt.attr_bins
        self.loss_names = ['G_GAN', 'G_IP', 'cycle_x', 'cycle_y', 'D_real', 'D_fake']
        self.visual_names = ['real_x', 'fake_x', 'rec_x'] if self.isTrain else ['real_x']
        self.model_names = ['G', 'E', 'D'] if self.isTrain else ['G', 'E']

        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.embedding_nc, opt.ngf,
                                      which_model_netG=opt.which_model_netG, norm=opt.norm_G, nl=opt.nl,
                                      dropout=opt.dropout, init_type=opt.init_type, gpu_ids=self.gpu_ids,
                                      upsample=opt.upsample)
        if opt.use_bicycle_E:
            raise NotImplementedError('pcgan_amd: --use_bicycle_E (define_E_bicycle) is outside the MI355X hot path')
        self.netE = networks.define_E(opt.which_model_netE, 3, init_type=opt.init_type, pooling=opt.pooling_E,
                                      cnn_dim=opt.cnn_dim_E, cnn_pad=opt.cnn_pad_E,
                                      cnn_relu_slope=opt.cnn_relu_slope_E, gpu_ids=self.gpu_ids)
        if self.isTrain and not opt.continue_train:
            getattr(self.netE, 'module', self.netE).load_base(opt.pretrained_model_path_E)

        if self.isTrain:
            # unconditional discriminator: nz = 0, called D(image)   (:98-100)
            self.netD = networks.define_D(opt.output_nc, 0, opt.ndf, opt.which_model_netD, opt.n_layers_D, opt.norm_D,
                                          opt.no_lsgan, opt.init_type, num_Ds=opt.num_Ds, gpu_ids=self.gpu_ids)
            self.netIP = networks.define_IP(opt.which_model_netIP, opt.input_nc, self.gpu_ids)
            getattr(self.netIP, 'module', self.netIP).load_pretrained(opt.pretrained_model_path_IP)

            assert opt.pool_size == 0
            self.criterionGAN = networks.GANLoss(use_lsgan=not opt.no_lsgan, tensor=self.Tensor)
            self.criterionL1 = networks.L1Loss()
            crit = opt.identity_preserving_criterion.lower()
            if crit == 'mse':
                self.criterionIP = networks.MSELoss()
            elif crit == 'l1':
                self.criterionIP = networks.L1Loss()
            else:
                raise NotImplementedError('Not Implemented')
            self.criterionCycle = networks.L1Loss()
            self.criterionAR = networks.MSELoss()

            for net in (self.netG, self.netD, self.netE, self.netIP):
                parallel.broadcast_parameters(net)
            betas = (opt.beta1, 0.999)
            self.optimizer_G = FusedAdam(self.netG.parameters(), lr=opt.lr, betas=betas)
            self.optimizer_E = FusedAdam(self.netE.parameters(), lr=opt.lr, betas=betas)
            self.optimizer_D = FusedAdam(self.netD.parameters(), lr=opt.lr, betas=betas)
            self.optimizers = [self.optimizer_G, self.optimizer_D, self.optimizer_E]   # the reference's order (:128-130)
            self.set_requires_grad(self.netIP, False)   # no optimizer in the reference either: its gradients are unused

        self.attr_mean, self.attr_std = opt.attr_mean, opt.attr_std
        mean, std = opt.attr_mean[0], opt.attr_std[0]
        self.attr_normalize = lambda x: (x - mean) / std
        if opt.display_visuals:
            self.pre_generate_embeddings()
        if self.isTrain:
            self.transform_IP = networks.Normalize((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))
            self.transform_E = networks.Normalize((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))

    def pre_generate_embeddings(self):
        arr = np.array(self.opt.attr_bins).reshape(len(self.opt.attr_bins), 1, 1, 1, 1)
        self.fixed_embeddings = [self.attr_normalize(torch.Tensor(arr[i]).to(self.device)) for i in range(arr.shape[0])]

    # ------------------------------------------------------------------ the step (reference :148-256)
    def set_input(self, input):
        self.real_x = self.to_act(input['A'])
        if self.isTrain:
            self.real_y = self.attr_normalize(input['B_attr'].to(self.device, non_blocking=True))
            self.image_paths = input['B_paths']
            self.real_x_IP = upsample2d(self.real_x, self.opt.fineSize_IP)
            self.real_x_E = upsample2d(self.real_x, self.opt.fineSize_E)
        else:
            self.image_paths = input['A_paths']
            if 'B_attr' in input:
                self.real_y = self.attr_normalize(input['B_attr'].to(self.device, non_blocking=True))
                self.image_paths = input['B_paths']
        self.current_iter += 1
        self.current_batch_size = int(self.real_x.size(0))

    def forward(self):
        # (the encoder input is NOT passed through transform_E here, as in the reference :170)
        self.fake_x = self.netG(self.real_x, self.real_y)
        self.fake_x_IP = upsample2d(self.fake_x, self.opt.fineSize_IP)
        self.fake_x_E = upsample2d(self.fake_x, self.opt.fineSize_E)
        self.fake_y = self.netE(self.real_x_E)
        self.rec_x = self.netG(self.real_x, self.fake_y)
        self.rec_y = self.netE(self.fake_x_E)

    def test(self):
        return

    def sample_from_prior(self):
        return self.netG(self.real_x, self.real_y)

    def sample_from_label(self, label):
        attr_B = torch.Tensor([self.attr_bins[label]]).reshape(1, 1, 1, 1).to(self.device)
        return self.netG(self.real_x, self.attr_normalize(attr_B))

    def backward_D(self):
        self.loss_D_fake = self.criterionGAN(self.netD(self.fake_x.detach()), False)
        self.loss_D_real = self.criterionGAN(self.netD(self.real_x), True)
        self.loss_D = (self.loss_D_fake + self.loss_D_real) * 0.5
        self.loss_D.backward()

    def backward_GE(self):
        o = self.opt
        b_ip = None
        if o.lambda_IP > 0.0:
            with hip_ops.branch('IP') as b_ip:     # AlexNet branch beside the discriminator branch (different nets)
                with torch.no_grad():
                    feature_A = self.netIP(self.transform_IP(self.real_x_IP))
                self.loss_G_IP = self.criterionIP(self.netIP(self.transform_IP(self.fake_x_IP)), feature_A) * o.lambda_IP
        else:
            self.loss_G_IP = 0.0
        self.loss_G_GAN = self.criterionGAN(self.netD(self.fake_x), True)
        self.loss_cycle_x = self.criterionCycle(self.rec_x, self.real_x) * o.lambda_x if o.lambda_x > 0.0 else 0.0
        self.loss_cycle_y = self.criterionAR(self.rec_y, self.real_y) * o.lambda_y if o.lambda_y > 0.0 else 0.0
        if b_ip is not None:
            b_ip.join(self.loss_G_IP)
        self.loss_G = self.loss_G_GAN + self.loss_G_IP + self.loss_cycle_x + self.loss_cycle_y
        self.loss_G.backward()

    def optimize_parameters(self):
        self.forward()
        # update D
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        parallel.sync_gradients(self.optimizer_D)
        self.optimizer_D.step()
        parallel.ddp_check(self.optimizer_D, 'D')
        # update G, E
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.optimizer_E.zero_grad()
        self.backward_GE()
        parallel.sync_gradients(self.optimizer_G)
        parallel.sync_gradients(self.optimizer_E)
        self.optimizer_G.step()
        self.optimizer_E.step()
        parallel.ddp_check(self.optimizer_G, 'G')
        parallel.ddp_check(self.optimizer_E, 'E')

    def get_current_visuals(self):
        self.set_requires_grad(self.netG, False)
        ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str):
                ret[name] = getattr(self, name)
        if self.opt.display_visuals:
            for L, embedding in enumerate(self.fixed_embeddings):
                ret['attr_' + str(L)] = self.netG(self.real_x[0:1, ...], embedding)
        self.set_requires_grad(self.netG, True)
        return ret
