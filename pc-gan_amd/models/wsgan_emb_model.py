"""WSGANEmbModel -- the PC-GAN (`--model wsgan_emb`) training step on the HIP layer set.

Plugin surface, option flags, loss names, step ordering (G first, then D), and every optional
branch follow the reference (models/wsgan_emb_model.py:16-497); the arithmetic runs through
pcgan_amd.hip.  What is different by design:

  * one process drives one GPU; under torch.distributed each optimizer's flat gradient buffer
    is averaged with ONE RCCL all-reduce before its step (pcgan_amd.hip.parallel);
  * Adam is the fused flat-buffer kernel (pcgan_amd.hip.optim.FusedAdam);
  * per-sample GAN targets are gathered from a device-resident look-up table instead of being
    rebuilt from a Python list every step (keeps the step free of host syncs);
  * `--lr_E > 0` (update_G_and_E): the reference's own branch fails on torch >= 1.5 with an in-place-modification error
    (SURVEY.md D13); here it runs with the semantics of the PyTorch it was written for -- the optimizers step without telling
    autograd, and the second backward pass through the RETAINED graph reads the updated weights where a backward formula reads
    a weight and the old forward pass's saved activations elsewhere (oracle/step_ref.py: AdamThroughData; parity unpinned, the
    semantics are DEFINED there).  FusedAdam updates the flat buffer through raw pointers, which is exactly that.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import networks
from .base_model import BaseModel
from ..hip import ops as hip_ops
from ..hip import parallel
from ..hip.optim import FusedAdam
from ..util.util import upsample2d, str2list, str2bool, resample, compute_mu_and_var

_DDP_OVERLAP = os.environ.get('PCGAN_DDP_OVERLAP', '0') == '1'
# second generator pass on its own stream: measured +2 % with fp32 tensors, -2.7 % with bf16 tensors (same box A/B) -> by storage type
_G2_BRANCH = {'1': True, '0': False}.get(os.environ.get('PCGAN_G2_BRANCH', ''), None)
# the frozen encoder's two passes over the new batch on the encoder's stream, waiting only for the batch itself
_E_AHEAD = os.environ.get('PCGAN_E_AHEAD', '1') == '1'
# ... likewise the generator's FIRST pass: it needs the batch, the encoder's rating and the generator's weights as the last
# optimizer_G.step() left them -- not the discriminator update still queued behind it.  (AlexNet's features of real_A moved ahead the
# same way cost 1.8 ms per step: that pass is better placed where it is, beside the discriminator branch of backward_G.)
_G1_AHEAD = os.environ.get('PCGAN_G1_AHEAD', '1') == '1'
# the AlexNet identity term on its own stream beside the discriminator branch: round 2's default; with the cross-step overlap four
# streams then crowd the same phase of backward_G and the step is 4.5 % SLOWER (1093 vs 1143 img/s, same box) -> off
_IP_BRANCH = os.environ.get('PCGAN_IP_BRANCH', '0') == '1'
# Round 4 experiment, default OFF (PCGAN_ADAM_ON_GRAD_STREAM=1 switches it on): the two Adam updates queued ON the parameter-gradient
# stream, behind the weight gradients, with no join of that stream at the end of a backward pass (hip/ops.py: defer_side_join,
# hip/optim.py: step_on_grad_stream) -- backward_D then starts while the generator's last weight gradients and its update are still
# running, the next step's forward while the discriminator's are; consumers of the new weights wait for the update's event.  Bit-identical
# (tests/test_gpu_concurrency.py runs it) and NEUTRAL on the step: 1166 / 1166 and 1204 / 1209 img/s in two same-box A/B runs
# (profiles/r04_experiments.txt) -- the step is bound by the total time of its matrix-pipe kernels, not by the join.  Single process
# only (under torch.distributed the all-reduce sits between backward and update).
_ADAM_ON_GRAD_STREAM = os.environ.get('PCGAN_ADAM_ON_GRAD_STREAM', '0') == '1'
# Data parallel, the default (PCGAN_DDP_GRAD_STREAM=0 restores the blocking all-reduce on the main stream): the same form with the all-reduce of the flat gradient buffer queued on the gradient
# stream in front of the update -- RCCL orders the collective behind that stream only, so the generator's 45.5 MB all-reduce and its
# Adam pass run under backward_D on the main stream, the discriminator's under the next step's forward passes (the main stream waits
# for `optimizer.updated` where it reads the new weights).  Unlike PCGAN_DDP_OVERLAP=1 the order of the step is untouched.  One rank
# over RCCL and two ranks over gloo give bit-equal results (tests/test_gpu_ddp.py) and on one rank the step time is unchanged
# (profiles/r04_experiments.txt section 10); the gain needs a fabric to be measured on.  bench.py --gpus N compares the replicas'
# parameter hashes after its timed steps.
_DDP_GRAD_STREAM = os.environ.get('PCGAN_DDP_GRAD_STREAM', '1') == '1'

MAGIC_EPS = 1e-20

# (flag, kwargs) -- same names, types, defaults as reference models/wsgan_emb_model.py:22-64
_COMMON_FLAGS = [
    ('--norm_G', dict(type=str, default='instance')),
    ('--norm_D', dict(type=str, default='batch')),
    ('--embedding_nc', dict(type=int, default=1)),
    ('--which_model_netE', dict(type=str, default='resnet18')),
    ('--pooling_E', dict(type=str, default='avg')),
    ('--cnn_dim_E', dict(type=int, nargs='+', default=[32, 1])),
    ('--no_cnn_E', dict(action='store_true')),
    ('--cnn_pad_E', dict(type=int, default=1)),
    ('--cnn_relu_slope_E', dict(type=float, default=0.7)),
    ('--fineSize_E', dict(type=int, default=224)),
    ('--pretrained_model_path_E', dict(type=str, default='pretrained_models/embedding_encoder.pth')),
    ('--embedding_mean', dict(type=float, nargs='*', default=[0.0])),
    ('--embedding_std', dict(type=float, nargs='*', default=[1.0])),
    ('--embedding_bins', dict(type=str, default='[]')),
    ('--display_visuals', dict(action='store_true')),
    ('--noisy', dict(type=str2bool, default=False)),
    ('--noisy_D', dict(type=str2bool, default=True)),
    ('--noisy_rec', dict(type=str2bool, default=True)),
    ('--noisy_var_type', dict(type=str, default='')),
    ('--bayesian', dict(type=str2bool, default=False)),
    ('--bnn_dropout', dict(type=float, default=0.)),
    ('--bnn_T', dict(type=int, default=10)),
    ('--use_projection', dict(type=str2bool, default=True)),
    ('--sample_embedding_B', dict(type=str2bool, default=False)),
]
_TRAIN_FLAGS = [
    ('--lambda_L1', dict(type=float, default=0.0)),
    ('--lambda_IP', dict(type=float, default=1.0)),
    ('--lambda_z', dict(type=float, default=1.0)),
    ('--lambda_A', dict(type=float, default=0.5)),
    ('--lambda_A_GAN', dict(type=float, default=0.0)),
    ('--lambda_theta_D', dict(type=float, default=0.0)),
    ('--lambda_theta_E', dict(type=float, default=0.0)),
    ('--which_model_netIP', dict(type=str, default='alexnet')),
    ('--pretrained_model_path_IP', dict(type=str, default='pretrained_models/alexnet-owt-4df8aa71.pth')),
    ('--fineSize_IP', dict(type=int, default=224)),
    ('--lr_E', dict(type=float, default=0.0)),
    ('--use_real_A', dict(action='store_true')),
    ('--identity_preserving_criterion', dict(type=str, default='mse')),
    ('--relabel_D', dict(type=int, nargs='*', default=[0, 1, 0])),
    ('--no_mixed_label_D', dict(action='store_true')),
    ('--weight_label_D', dict(nargs='*', type=float, default=[0.5, 0, 0.5])),
    ('--detach_fake_B', dict(action='store_true')),
    ('--update_logvar_E', dict(type=str2bool, default=False)),
]
_DEFAULT_OVERRIDES = dict(pool_size=0, no_lsgan=True, norm='instance', dataset_mode='wsgan_emb',
                          which_model_netG='unet_128', which_model_netD='n_layers', n_layers_D=4, batchSize=10,
                          loadSize=128, fineSize=128, display_visuals=True, save_epoch_freq=2)


class WSGANEmbModel(BaseModel):
    def name(self):
        return 'WSGANEmbModel'

    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        for flag, kw in _COMMON_FLAGS + (_TRAIN_FLAGS if is_train else []):
            parser.add_argument(flag, **kw)
        parser.set_defaults(**_DEFAULT_OVERRIDES)
        return parser

    # ------------------------------------------------------------------ construction
    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        assert opt.input_nc == opt.output_nc
        self.attr_bins = opt.attr_bins
        self.embedding_bins = str2list(opt.embedding_bins)
        if opt.no_cnn_E:
            opt.cnn_dim_E = []
        if 'a' in opt.noisy_var_type and not opt.noisy:
            raise RuntimeError('Aleatoric only available when noisy is True.')
        if 'e' in opt.noisy_var_type and not opt.bayesian:
            raise RuntimeError('Epistemic only available when bayesian is True.')
        self.loss_names = ['G_GAN', 'G_GAN_cycle', 'G_IP', 'G_L1', 'G_cycle', 'z_rec',
                           'D_real_right', 'D_real_wrong', 'D_fake']
        self.visual_names = ['real_A', 'fake_B', 'real_B', 'rec_A'] if self.isTrain else ['real_A']
        self.model_names = ['G', 'D', 'E'] if self.isTrain else ['G', 'E']
        self.load_model_names = opt.load_model_names

        self.netG = networks.define_G(opt.input_nc, opt.output_nc, opt.embedding_nc, opt.ngf,
                                      which_model_netG=opt.which_model_netG, norm=opt.norm_G, nl=opt.nl,
                                      dropout=opt.dropout, init_type=opt.init_type, gpu_ids=self.gpu_ids,
                                      upsample=opt.upsample, n_layers_G=opt.n_layers_G)
        self.netE = networks.define_E(opt.which_model_netE, 3, init_type=opt.init_type, pooling=opt.pooling_E,
                                      cnn_dim=opt.cnn_dim_E, cnn_pad=opt.cnn_pad_E,
                                      cnn_relu_slope=opt.cnn_relu_slope_E, gpu_ids=self.gpu_ids,
                                      fine_size_E=opt.fineSize_E, noisy=opt.noisy, bnn_dropout=opt.bnn_dropout)
        if self.isTrain and not opt.continue_train:
            getattr(self.netE, 'module', self.netE).load_pretrained(opt.pretrained_model_path_E)

        if self.isTrain:
            self.netD = networks.define_D(opt.output_nc, opt.embedding_nc, opt.ndf, opt.which_model_netD,
                                          opt.n_layers_D, opt.norm_D, opt.no_lsgan, opt.init_type,
                                          num_Ds=opt.num_Ds, gpu_ids=self.gpu_ids)
            self.netIP = networks.define_IP(opt.which_model_netIP, opt.input_nc, self.gpu_ids)
            if opt.pretrained_model_path_IP and opt.lambda_IP > 0:
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
            self.criterionRec = networks.MSELoss()
            self.criterionCycle = networks.L1Loss()

            # every replica starts from rank 0's weights (DataParallel replicates device 0)
            for net in (self.netG, self.netD, self.netE, self.netIP):
                parallel.broadcast_parameters(net)

            self.optimizer_G = self.make_optimizer(self.netG.parameters(), opt.lr, (opt.beta1, 0.999))
            self.optimizer_D = self.make_optimizer(self.netD.parameters(), opt.lr, (opt.beta1, 0.999))
            self.optimizers = [self.optimizer_G, self.optimizer_D]
            if opt.lr_E > 0.0:      # reference models/wsgan_emb_model.py:157-163
                if opt.update_logvar_E:
                    assert opt.noisy
                    e_params = self.netE.cnn_logvar.parameters()
                else:
                    e_params = self.netE.parameters()
                self.optimizer_E = self.make_optimizer(e_params, opt.lr_E, (opt.beta1, 0.999))
                self.optimizers.append(self.optimizer_E)
            else:
                self.set_requires_grad(self.netE, False)
            self.set_requires_grad(self.netIP, False)   # frozen in the reference too: it has no optimizer

        mean, std = opt.embedding_mean[0], opt.embedding_std[0]
        self.embedding_normalize = lambda x: (x - mean) / std

        if opt.display_visuals:
            self.pre_generate_embeddings(self.embedding_bins)

        if self.isTrain:
            self.transform_IP = networks.Normalize((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))
        self.transform_E = networks.Normalize((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))

        if self.isTrain:
            self.relabel_D = opt.relabel_D
            if len(opt.weight_label_D) > 0:
                assert len(opt.weight_label_D) == len(opt.relabel_D)
                total = sum(opt.weight_label_D)
                self.weight_label_D = [w / total for w in opt.weight_label_D]
            else:
                self.weight_label_D = None
            self._relabel_lut = torch.tensor(np.array(self.relabel_D, dtype=np.float32), device=self.device)

    def make_optimizer(self, params, lr, betas):
        """torch.optim.Adam of the reference (models/wsgan_emb_model.py:153-154) as the fused kernel."""
        return FusedAdam(params, lr=lr, betas=betas)

    def pre_generate_embeddings(self, embeddings_list):
        arr = np.array(embeddings_list)
        self.fixed_embeddings = [
            self.embedding_normalize(torch.Tensor(arr[i].reshape([1, 1, 1, 1])).to(self.device))
            for i in range(arr.shape[0])]

    # ------------------------------------------------------------------ the step
    def set_input(self, input):
        if self.isTrain:
            if not self.opt.no_mixed_label_D:
                self.real_A = self.to_act(input['A'])
                self.real_B = self.to_act(input['B'])
                self.image_paths = input['B_paths']
                self.label_AB = input['label']
            else:
                L = int(np.random.choice(range(len(self.relabel_D)), p=self.weight_label_D))
                self.label_AB = [L]
                self.real_A = self.to_act(input[str(L) + '_A'])
                self.real_B = self.to_act(input[str(L) + '_B'])
                self.image_paths = input[str(L) + '_B_paths']
            lab = self.label_AB if isinstance(self.label_AB, torch.Tensor) else torch.as_tensor(self.label_AB)
            self._label_dev = lab.to(self.device, dtype=torch.int64, non_blocking=True).reshape(-1)
        else:
            self.real_A = self.to_act(input['A'])
            self.image_paths = input['A_paths']
            if 'B' in input:
                self.real_B = self.to_act(input['B'])
                self.image_paths = input['B_paths']
        self.current_iter += 1
        self.current_batch_size = int(self.real_A.size(0))

    def _encode(self, x_E):
        """E on one image set under the (bayesian, noisy) mode; returns (y, var or None) where var
        is what `resample` needs for the configured noisy_var_type."""
        o = self.opt
        x = self.transform_E(x_E)
        # the frozen encoder on a real image set (no autograd graph): hipGraph replay of the same launches (hip/graphs.py)
        E = self._graphed('E', self.netE)      # (replays only when the call records no autograd graph; eager otherwise)
        if not o.bayesian and not o.noisy:
            return E(x), None
        if not o.bayesian and o.noisy:
            y, logvar = E(x)
            return y, (torch.exp(logvar) if 'a' in o.noisy_var_type else None)
        if o.bayesian and not o.noisy:
            y, y_var = compute_mu_and_var(E, x, o.bnn_T, False)
            return y, (y_var if 'e' in o.noisy_var_type else None)
        y, y_var, y_s2 = compute_mu_and_var(E, x, o.bnn_T, True)
        return y, (y_s2 + y_var if 'a' in o.noisy_var_type else None)

    def _graphed(self, name, net):
        """GraphedNoGrad wrapper of a frozen net (made on first use)"""
        cache = self.__dict__.setdefault('_graphed_nets', {})
        g = cache.get(name)
        if g is None:
            from ..hip.graphs import GraphedNoGrad
            g = cache[name] = GraphedNoGrad(net)
        return g

    def forward(self):
        o = self.opt
        if self.isTrain:
            self._wait_updated(self.optimizer_G)      # generator passes on this stream (and on branches forked from it): after its last update
            if o.lr_E > 0.0:
                self._wait_updated(self.optimizer_E)
        self.real_A_IP = upsample2d(self.real_A, o.fineSize_IP)
        frozen = o.lr_E <= 0.0
        # The frozen encoder's passes over the two real image sets depend on the batch and on the encoder's own state only (weights
        # that no optimizer touches, BatchNorm running statistics that only encoder passes update -- all of them on this stream, in
        # the reference's order): they run on the encoder's branch stream and wait for the BATCH (ops.ready_event), not for what the
        # main stream still has queued, so they fill the tail of the previous step (backward_D, Adam) instead of starting the next.
        ahead = frozen and _E_AHEAD and self.real_A.is_cuda
        after = [hip_ops.ready_event(self.real_A), hip_ops.ready_event(self.real_B)] if ahead else None
        with hip_ops.branch('E', enabled=ahead, after=after) as eb:
            eb.reads(self.real_A, self.real_B)
            self.real_A_E = upsample2d(self.real_A, o.fineSize_E)
            self.real_B_E = upsample2d(self.real_B, o.fineSize_E)
            with torch.set_grad_enabled(not frozen):   # E is frozen: no graph, nothing saved for backward
                y_A, var_A = self._encode(self.real_A_E)
                y_B, var_B = self._encode(self.real_B_E)
                if var_A is not None:       # resample order: A then B, as the reference draws them
                    self.resample_A = self.embedding_normalize(resample(y_A, var_A))
                    self.resample_B = self.embedding_normalize(resample(y_B, var_B))
            self.y_A, self.y_B = y_A, y_B
            self.embedding_A = self.embedding_normalize(y_A)
            self.embedding_B = self.embedding_normalize(y_B)
        self._e_on_branch = bool(ahead and eb.on)
        eb.join(self.real_A_E, self.real_B_E, self.y_A, self.y_B, self.embedding_A, self.embedding_B,
                getattr(self, 'resample_A', None), getattr(self, 'resample_B', None))
        if frozen:
            self.y_A, self.y_B = self.y_A.detach(), self.y_B.detach()
            self.embedding_A, self.embedding_B = self.embedding_A.detach(), self.embedding_B.detach()
            if o.noisy_var_type:
                self.resample_A, self.resample_B = self.resample_A.detach(), self.resample_B.detach()
        g_done = getattr(self, '_g_updated', None)
        g1 = self.isTrain and _G1_AHEAD and ahead and eb.on and g_done is not None
        with hip_ops.branch('G1', enabled=g1, after=[hip_ops.ready_event(self.real_A), eb.st, g_done] if g1 else None) as bg:
            bg.reads(self.real_A, self.embedding_B)
            self.fake_B = self.netG(self.real_A, self.embedding_B)
        bg.join(self.fake_B)
        self.fake_B_IP = upsample2d(self.fake_B, o.fineSize_IP)
        self.fake_B_E = upsample2d(self.fake_B, o.fineSize_E)
        # the second generator pass only meets the other consumers of fake_B (D / IP / E branches of backward_G) in the
        # loss sum: it runs on its own stream beside them (forward here, its backward nodes on the same stream)
        # (default since round 3, PCGAN_G2_BRANCH=0 switches it off: -2 % step time; the second pass's kernels share the GPU with
        # the branches, so their in-step timings include that overlap -- bench.py reports the alone timings beside them)
        g2 = _G2_BRANCH if _G2_BRANCH is not None else (self.act_dtype == torch.float32)
        with hip_ops.branch('G2', enabled=g2) as self._rec_branch:
            self._rec_branch.reads(self.fake_B, self.embedding_A)
            self.rec_A = self.netG(self.fake_B.detach() if o.detach_fake_B else self.fake_B, self.embedding_A)

    def _join_rec(self):
        b = getattr(self, '_rec_branch', None)
        if b is not None:
            b.join(self.rec_A)
            self._rec_branch = None

    def test(self):
        self.sync_parameter_updates()
        self._g_updated = None
        if hasattr(self, 'real_B'):
            if 'real_B' not in self.visual_names:
                self.visual_names += ['real_B', 'fake_B']
            with torch.no_grad():
                y_B, _ = self._encode(upsample2d(self.real_B, self.opt.fineSize_E))
                self.embedding_B = self.embedding_normalize(y_B.detach())
                self.fake_B = self.netG(self.real_A, self.embedding_B)

    def sample_from_prior(self):
        self.sync_parameter_updates()
        self._g_updated = None
        y_B, _ = self._encode(upsample2d(self.real_B, self.opt.fineSize_E))
        self.embedding_B = self.embedding_normalize(y_B.detach())
        return self.netG(self.real_A, self.embedding_B)

    def sample_from_label(self, label):
        self.sync_parameter_updates()
        self._g_updated = None
        emb_B = torch.Tensor([self.embedding_bins[label]]).reshape(1, 1, 1, 1).to(self.device)
        return self.netG(self.real_A, self.embedding_normalize(emb_B))

    def _rating_for_D(self):
        o = self.opt
        return self.resample_B if (o.noisy_var_type and o.noisy_D) else self.embedding_B

    def backward_D(self):
        o = self.opt
        pred_fake = self.netD(self.fake_B.detach(), self._rating_for_D().detach())
        self.loss_D_fake = self.criterionGAN(pred_fake, False)
        img = self.real_A if o.use_real_A else self.real_B
        z_right = self.embedding_A if o.use_real_A else self.embedding_B
        z_wrong = self.embedding_B if o.use_real_A else self.embedding_A
        self.loss_D_real_right = self.criterionGAN(self.netD(img, z_right.detach()), True)
        # "real image, wrong rating": per-sample target relabel_D[label] (device-side LUT gather)
        target = self._relabel_lut.index_select(0, self._label_dev)
        self.loss_D_real_wrong = self.criterionGAN(self.netD(img, z_wrong.detach()), target)
        self.loss_D = (self.loss_D_fake + (self.loss_D_real_right + self.loss_D_real_wrong) * 0.5) * 0.5
        self.loss_D.backward()

    def _common_G_losses(self):
        o = self.opt
        # the identity branch (AlexNet: no running statistics, order-free) runs beside the discriminator branch
        b_ip = None
        if o.lambda_IP > 0.0:
            with hip_ops.branch('IP', enabled=_IP_BRANCH) as b_ip:
                b_ip.reads(self.real_A_IP, self.fake_B_IP)
                with torch.no_grad():
                    feature_A = self.netIP(self.transform_IP(self.real_A_IP))
                self.loss_G_IP = self.criterionIP(self.netIP(self.transform_IP(self.fake_B_IP)), feature_A) * o.lambda_IP
        else:
            self.loss_G_IP = 0.0
        self.loss_G_GAN = self.criterionGAN(self.netD(self.fake_B, self._rating_for_D()), True)
        self._join_rec()
        if o.lambda_A_GAN > 0.0:
            self.loss_G_GAN_cycle = self.criterionGAN(self.netD(self.rec_A, self.embedding_A), True) * o.lambda_A_GAN
        else:
            self.loss_G_GAN_cycle = 0.0
        self.loss_G_L1 = self.criterionL1(self.fake_B, self.real_A) * o.lambda_L1 if o.lambda_L1 > 0.0 else 0.0
        self.loss_G_cycle = self.criterionCycle(self.rec_A, self.real_A) * o.lambda_A if o.lambda_A > 0.0 else 0.0
        return b_ip

    def backward_G(self):
        o = self.opt
        b_e = None
        if o.lambda_z > 0.0:
            # the Elo-encoder branch, beside the discriminator / identity branches.  In the bayesian + noisy mode the prediction comes
            # from real_A_E (reference quirk D10): everything it reads was produced on the encoder's stream by forward(), so the branch
            # need not wait for the generator passes queued on the main stream
            d10 = o.bayesian and o.noisy and o.lr_E <= 0.0 and getattr(self, '_e_on_branch', False)
            b_e = hip_ops.branch('E', after=[] if d10 else None)
            b_e.__enter__()
            b_e.reads(self.fake_B_E, self.real_A_E, self.y_B)
            try:
                self._z_rec_loss()
            finally:
                b_e.__exit__(None, None, None)
        else:
            self.loss_z_rec = 0.0
        b_ip = self._common_G_losses()
        if b_e is not None:
            b_e.join(self.loss_z_rec)
        if b_ip is not None:
            b_ip.join(self.loss_G_IP)
        self.loss_G = self.loss_G_GAN + self.loss_G_IP + self.loss_G_L1 + self.loss_G_cycle + self.loss_z_rec \
            + self.loss_G_GAN_cycle
        if isinstance(self.loss_G, torch.Tensor) and self.loss_G.requires_grad:
            self.loss_G.backward()

    def backward_GE(self):
        """reference models/wsgan_emb_model.py:331-369: the generator's losses without the rating reconstruction; the graph is kept
        for backward_G_alone"""
        b_ip = self._common_G_losses()
        if b_ip is not None:
            b_ip.join(self.loss_G_IP)
        self.loss_G = self.loss_G_GAN + self.loss_G_IP + self.loss_G_L1 + self.loss_G_cycle + self.loss_G_GAN_cycle
        self.loss_G.backward(retain_graph=True)

    def backward_G_alone(self):
        """reference models/wsgan_emb_model.py:439-449: the rating reconstruction, back-propagated through the encoder's NEW pass
        over fake_B and then through the retained graph of fake_B (updated weights, old activations: see the module docstring)"""
        o = self.opt
        if o.lambda_z > 0.0:
            pred = self.netE(self.transform_E(self.fake_B_E))
            pred = self.embedding_normalize(pred[0] if o.noisy else pred)
            self.loss_z_rec = self.criterionRec(pred, self.embedding_B.detach()) * o.lambda_z
            self.loss_z_rec.backward()
        else:
            self.loss_z_rec = 0.0

    def _z_rec_loss(self):
        """loss_z_rec of backward_G (reference models/wsgan_emb_model.py:400-435), its four encoder variants."""
        o = self.opt
        y_var = y_logvar = None
        if not o.bayesian and not o.noisy:
            pred_y = self.netE(self.transform_E(self.fake_B_E))
        elif not o.bayesian and o.noisy:
            pred_y, y_logvar = self.netE(self.transform_E(self.fake_B_E))
            if 'a' in o.noisy_var_type:
                y_var = torch.exp(y_logvar)
        elif o.bayesian and not o.noisy:
            pred_y, y_var = compute_mu_and_var(self.netE, self.transform_E(self.fake_B_E), o.bnn_T, False)
            if 'e' in o.noisy_var_type:
                y_logvar = torch.log(y_var + MAGIC_EPS)
        else:
            # reference quirk kept: the prediction comes from real_A_E, so this term
            # carries no gradient to G
            # (real images, frozen encoder: nothing to differentiate -- the ten MC-dropout passes replay the captured forward)
            pred_y, y_var_, y_s2_ = compute_mu_and_var(self._graphed('E', self.netE) if o.lr_E <= 0.0 else self.netE,
                                                       self.transform_E(self.real_A_E), o.bnn_T, True)
            y_var = torch.zeros_like(pred_y)
            if 'a' in o.noisy_var_type:
                y_var = y_var + y_s2_
            if 'e' in o.noisy_var_type:
                y_var = y_var + y_var_
            y_logvar = torch.log(y_var + MAGIC_EPS)
        if o.noisy_var_type and o.noisy_rec:
            self.loss_z_rec = ((pred_y - self.y_B).pow(2) / y_var.detach() + y_logvar.detach()).sum() \
                / pred_y.size()[0] * 0.5 * o.lambda_z
        else:
            self.loss_z_rec = self.criterionRec(pred_y, self.y_B) * o.lambda_z

    def _on_grad_stream(self, optimizer):
        """may this optimizer's update run on the parameter-gradient stream?  (the stock FusedAdam.step on a GPU, one process; an
        instance-level `step` -- the tests' gradient grabbers -- takes the joined path and reads finished gradients)"""
        on = _DDP_GRAD_STREAM if parallel.is_distributed() else _ADAM_ON_GRAD_STREAM
        return on and hip_ops.SIDE_STREAM and isinstance(optimizer, FusedAdam) and 'step' not in optimizer.__dict__ and optimizer.flat.is_cuda

    def _step(self, optimizer, name):
        if self._on_grad_stream(optimizer):
            dist = parallel.is_distributed()
            optimizer.step_on_grad_stream(before=(lambda: parallel.allreduce_mean_(optimizer.gflat)) if dist else None)
            if dist and parallel.DDP_CHECK_EVERY:
                self._wait_updated(optimizer)      # the replica hash reads the new parameters on this stream
        else:
            hip_ops.join_side_stream(force=True)
            parallel.sync_gradients(optimizer)
            optimizer.step()
        parallel.ddp_check(optimizer, name)

    def _wait_updated(self, optimizer):
        """the current stream reads this optimizer's parameters next: order it behind an update still queued on the gradient stream"""
        ev = getattr(optimizer, 'updated', None)
        if ev is not None and self.real_A.is_cuda:
            torch.cuda.current_stream(self.real_A.device).wait_event(ev)

    def update_D(self):
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        with hip_ops.defer_side_join(self._on_grad_stream(self.optimizer_D)):
            self.backward_D()
        self._step(self.optimizer_D, 'D')

    def update_G(self):
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self._wait_updated(self.optimizer_D)      # backward_G runs the discriminator: after its last update
        with hip_ops.defer_side_join(self._on_grad_stream(self.optimizer_G)):
            self.backward_G()
        self._step(self.optimizer_G, 'G')
        self._mark_g_updated()

    def update_G_and_E(self):
        """reference models/wsgan_emb_model.py:463-476"""
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.optimizer_E.zero_grad()
        self._wait_updated(self.optimizer_D)
        self.loss_z_rec = 0.0
        self.backward_GE()
        self._step(self.optimizer_G, 'G')
        self._step(self.optimizer_E, 'E')
        if self.opt.lambda_z > 0.0:      # update G only
            self.optimizer_G.zero_grad()
            self.optimizer_E.zero_grad()
            self._wait_updated(self.optimizer_G)
            self._wait_updated(self.optimizer_E)
            self.backward_G_alone()
            self._step(self.optimizer_G, 'G')
        self._mark_g_updated()

    def _mark_g_updated(self):
        """the point in the current stream after which the generator's weights are those of this update (forward(): G1 branch)"""
        if self.real_A.is_cuda:
            if getattr(self.optimizer_G, 'updated', None) is not None:       # the update runs on the gradient stream: its own event
                self._g_updated = self.optimizer_G.updated
                return
            self._g_updated = torch.cuda.Event()
            self._g_updated.record(torch.cuda.current_stream(self.real_A.device))

    def optimize_parameters(self):
        self.forward()
        if self.opt.lr_E > 0.0:
            self.update_G_and_E()
            self.update_D()
            return
        if not (parallel.is_distributed() and _DDP_OVERLAP):
            self.update_G()          # (under torch.distributed each ends in a blocking all-reduce of its flat gradient buffer)
            self.update_D()
            return
        # PCGAN_DDP_OVERLAP=1 (opt-in until one multi-GPU RCCL run has confirmed bit-equal replicas and the overlap):
        # data parallel: the generator's gradient all-reduce is launched after backward_G and runs under backward_D, which neither
        # reads the generator's weights (fake_B is detached and was computed in forward()) nor touches its gradients; both optimizer
        # steps follow -- same arithmetic as update_G(); update_D() (reference models/wsgan_emb_model.py:451-461, 478-484)
        self.set_requires_grad(self.netD, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        finish_G = parallel.sync_gradients(self.optimizer_G, async_op=True)
        self.set_requires_grad(self.netD, True)
        self.optimizer_D.zero_grad()
        self.backward_D()
        finish_G()
        self.optimizer_G.step()
        parallel.ddp_check(self.optimizer_G, 'G')
        self._mark_g_updated()
        parallel.sync_gradients(self.optimizer_D)
        self.optimizer_D.step()
        parallel.ddp_check(self.optimizer_D, 'D')

    def get_current_visuals(self):
        self._join_rec()
        self.sync_parameter_updates()
        self._g_updated = None      # generator passes outside the step move its running statistics: the next forward() queues behind them
        self.set_requires_grad(self.netG, False)
        ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str):
                ret[name] = getattr(self, name)
        if self.opt.display_visuals:
            for i, emb in enumerate(self.fixed_embeddings):
                ret['attr_' + str(i)] = self.netG(self.real_A[0:1, ...], emb)
        self.set_requires_grad(self.netG, True)
        return ret
