"""ORACLE (test infrastructure, not product code).

CPU restatement of WSGANEmbModel's training step -- set_input / forward / backward_G /
backward_D / update_G / update_D / optimize_parameters -- as plain functions over the oracle
nets (reference models/wsgan_emb_model.py:193-259, 300-329, 371-437, 451-484), plus the small
tensor utilities it calls (reference util/util.py:111-171) and the integer helpers of row a13
(util/util.py:76-93).  Pinned against the reference's own outputs by tests/test_oracle_golden.py.

Randomness (resample eps, MC-dropout masks) is drawn exactly where the reference draws it,
so under the same torch.manual_seed the CPU random streams coincide; every draw is also
recorded in `self.draws` so the GPU product can be fed the identical numbers.
"""
import ast
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

MAGIC_EPS = 1e-20


# ----------------------------------------------------------------------------- a13 integer helpers
def get_attr_label(attr, bins):
    """util/util.py:76-81 -- first L with bins[L] <= attr < bins[L+1]; falls through to
    len(bins)-2 when nothing matches; None when len(bins) < 2."""
    L = None
    for L in range(len(bins) - 1):
        if (attr >= bins[L]) and (attr < bins[L + 1]):
            break
    return L


def str2list(s):
    """util/util.py:84-93"""
    assert isinstance(s, str)
    s = s.strip()
    if s.endswith(('.npy', '.npz')):
        return np.load(s)
    assert s.startswith('[') and s.endswith(']')
    return ast.literal_eval(s)


def relabel(relabel_D, label_AB):
    """models/wsgan_emb_model.py:324"""
    return [relabel_D[int(L)] for L in label_AB]


def lambda_lr(epoch, epoch_count, niter, niter_decay):
    """models/networks.py:59-61"""
    return 1.0 - max(0, epoch + 1 + epoch_count - niter) / float(niter_decay + 1)


# ----------------------------------------------------------------------------- tensor utils
def upsample2d(x, size):
    """util/util.py:111-117"""
    if size <= 0 or x.size(2) == size:
        return x
    return F.interpolate(input=x, size=(size, size), mode='bilinear', align_corners=True)


def gan_loss(pred, target, lsgan=False):
    """models/networks.py:386-420"""
    if not isinstance(target, list):
        target = [target]
    vals = [(1 if t else 0) if isinstance(t, bool) else t for t in target]
    t = torch.tensor(np.array(vals).reshape(len(vals), 1, 1, 1), dtype=pred.dtype).expand_as(pred)
    return F.mse_loss(pred, t) if lsgan else F.binary_cross_entropy(pred, t)


DEFAULTS = dict(
    fineSize_E=224, fineSize_IP=224, embedding_mean=[0.0], embedding_std=[1.0],
    noisy=False, noisy_D=True, noisy_rec=True, noisy_var_type='', bayesian=False, bnn_T=10,
    lambda_L1=0.0, lambda_IP=1.0, lambda_z=1.0, lambda_A=0.5, lambda_A_GAN=0.0, lr_E=0.0, update_logvar_E=False,
    use_real_A=False, relabel_D=[0, 1, 0], detach_fake_B=False, lr=2e-4, beta1=0.5,
    identity_preserving_criterion='mse')


class AdamThroughData(torch.optim.Optimizer):
    """torch.optim.Adam's update (betas, eps 1e-8, no weight decay / amsgrad) applied through `p.data`, i.e. without telling
    autograd that the parameter changed -- how optimizers stepped in the PyTorch (<= 0.4) the reference was written for.  Only the
    `--lr_E > 0` branch needs it: update_G_and_E steps G and E and then back-propagates through the graph it RETAINED
    (models/wsgan_emb_model.py:463-476); torch >= 1.5 refuses that, the old one silently used the updated weights
    where a backward formula reads a weight and the activations saved by the old forward pass elsewhere.  The optimizer's
    SEMANTICS are defined here, not taken from a run of the reference's era; with it substituted for torch.optim.Adam the
    reference's own update_G_and_E / backward_GE / backward_G_alone run on torch 2.10, and that run pins the restatement
    below (oracle/make_golden.py golden_step_lr_E -> tests/golden/step_lr_E.npz)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g['params']:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        for g in self.param_groups:
            b1, b2 = g['betas']
            for p in g['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st['t'], st['m'], st['v'] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st['t'] += 1
                m, v, t = st['m'], st['v'], st['t']
                grad = p.grad
                m.mul_(b1).add_(grad, alpha=1 - b1)
                v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
                bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
                denom = (v.sqrt() / (bc2 ** 0.5)).add_(g['eps'])
                p.data.addcdiv_(m, denom, value=-g['lr'] / bc1)


class WSGANEmbStepRef:
    """One object = the reference model's tensors and optimizers, restated."""

    def __init__(self, netG, netD, netE, netIP, **opts):
        o = dict(DEFAULTS)
        o.update(opts)
        self.opt = SimpleNamespace(**o)
        self.netG, self.netD, self.netE, self.netIP = netG, netD, netE, netIP
        # models/wsgan_emb_model.py:153-165
        self.optimizer_G = torch.optim.Adam(netG.parameters(), lr=self.opt.lr, betas=(self.opt.beta1, 0.999))
        self.optimizer_D = torch.optim.Adam(netD.parameters(), lr=self.opt.lr, betas=(self.opt.beta1, 0.999))
        if self.opt.lr_E > 0.0:
            # models/wsgan_emb_model.py:157-163 under the defined semantics of AdamThroughData (parity unpinned, see there)
            betas = (self.opt.beta1, 0.999)
            self.optimizer_G = AdamThroughData(netG.parameters(), self.opt.lr, betas)
            e_params = netE.cnn_logvar.parameters() if self.opt.update_logvar_E else netE.parameters()
            self.optimizer_E = AdamThroughData(e_params, self.opt.lr_E, betas)
        else:
            for p in netE.parameters():
                p.requires_grad = False
        self.draws = []          # every random tensor drawn, in order
        self.inject = None       # optional iterator of tensors to use instead of drawing
        self.grads_G = self.grads_D = None

    # -- randomness -------------------------------------------------------------
    def _randn_like(self, t):
        if self.inject is not None:
            eps = next(self.inject).to(t.dtype).view_as(t)
        else:
            eps = torch.randn_like(t)
        self.draws.append(eps.detach().clone())
        return eps

    def embedding_normalize(self, x):
        """models/wsgan_emb_model.py:167"""
        return (x - self.opt.embedding_mean[0]) / self.opt.embedding_std[0]

    def resample(self, mu, var):
        """util/util.py:136-139"""
        std = torch.sqrt(var)
        return mu + self._randn_like(std) * std

    def compute_mu_and_var(self, x, T, noisy):
        """util/util.py:153-171"""
        y_mu, y_sq, s2_mu = 0., 0., 0.
        for _ in range(T):
            if noisy:
                y, logs2 = self.netE(x)
                s2_mu = s2_mu + 1. / T * torch.exp(logs2)
            else:
                y = self.netE(x)
            y_mu = y_mu + 1. / T * y
            y_sq = y_sq + 1. / T * y ** 2
        y_var = y_sq - y_mu ** 2
        return (y_mu, y_var, s2_mu) if noisy else (y_mu, y_var)

    # -- the step ---------------------------------------------------------------
    def set_input(self, real_A, real_B, label_AB):
        """models/wsgan_emb_model.py:193-212 (mixed-label branch)"""
        self.real_A, self.real_B, self.label_AB = real_A, real_B, label_AB

    def set_input_no_mixed(self, batch, weight_label_D=(0.5, 0.0, 0.5)):
        """models/wsgan_emb_model.py:199-207 (`--no_mixed_label_D`): ONE label for the whole batch, drawn from numpy's
        global generator with the normalised --weight_label_D; the batch dict carries one pair set per label."""
        total = float(sum(weight_label_D))
        p = [w / total for w in weight_label_D]
        L = np.random.choice(range(len(self.opt.relabel_D)), p=p)
        self.label_AB = [L]
        self.real_A, self.real_B = batch[str(L) + '_A'], batch[str(L) + '_B']

    def get_current_visuals(self, fixed_ratings):
        """models/wsgan_emb_model.py:486-497 with --display_visuals: G on real_A[0:1] once per fixed rating, in whatever
        mode G is in (train mode during training: InstanceNorm running statistics move)."""
        for p in self.netG.parameters():
            p.requires_grad = False
        ret = {'real_A': self.real_A, 'fake_B': self.fake_B, 'real_B': self.real_B, 'rec_A': self.rec_A}
        for i, r in enumerate(fixed_ratings):
            emb = self.embedding_normalize(torch.tensor([float(r)], dtype=self.real_A.dtype).reshape(1, 1, 1, 1))
            ret['attr_%d' % i] = self.netG(self.real_A[0:1, ...], emb)
        for p in self.netG.parameters():
            p.requires_grad = True
        return ret

    def sample_from_prior(self):
        """models/wsgan_emb_model.py:279-292: the rating of real_B (the noisy / bayesian branches read the real_B_E attribute
        that forward() left) -> normalised -> G(real_A, rating)"""
        o = self.opt
        real_B_E = upsample2d(self.real_B, o.fineSize_E)
        if not o.bayesian and not o.noisy:
            y_B = self.netE(real_B_E)
        elif not o.bayesian and o.noisy:
            y_B, _ = self.netE(self.real_B_E)
        elif o.bayesian and not o.noisy:
            y_B, _ = self.compute_mu_and_var(self.real_B_E, o.bnn_T, False)
        else:
            y_B, _, _ = self.compute_mu_and_var(self.real_B_E, o.bnn_T, True)
        self.embedding_B = self.embedding_normalize(y_B.detach())
        return self.netG(self.real_A, self.embedding_B)

    def sample_from_label(self, label, embedding_bins):
        """models/wsgan_emb_model.py:294-298: the bin centre of `label` as a (1, 1, 1, 1) rating broadcast over the batch"""
        emb_B = torch.tensor([float(embedding_bins[label])], dtype=self.real_A.dtype).reshape(1, 1, 1, 1)
        return self.netG(self.real_A, self.embedding_normalize(emb_B))

    def forward(self):
        """models/wsgan_emb_model.py:214-259 (transform_E / transform_IP are identities)"""
        o = self.opt
        self.real_A_IP = upsample2d(self.real_A, o.fineSize_IP)
        self.real_A_E = upsample2d(self.real_A, o.fineSize_E)
        self.real_B_E = upsample2d(self.real_B, o.fineSize_E)
        if not o.bayesian and not o.noisy:
            y_A = self.netE(self.real_A_E)
            y_B = self.netE(self.real_B_E)
        elif not o.bayesian and o.noisy:
            y_A, logvar_A = self.netE(self.real_A_E)
            y_B, logvar_B = self.netE(self.real_B_E)
            if 'a' in o.noisy_var_type:
                self.resample_A = self.embedding_normalize(self.resample(y_A, torch.exp(logvar_A)))
                self.resample_B = self.embedding_normalize(self.resample(y_B, torch.exp(logvar_B)))
        elif o.bayesian and not o.noisy:
            y_A, y_A_var = self.compute_mu_and_var(self.real_A_E, o.bnn_T, False)
            y_B, y_B_var = self.compute_mu_and_var(self.real_B_E, o.bnn_T, False)
            if 'e' in o.noisy_var_type:
                self.resample_A = self.embedding_normalize(self.resample(y_A, y_A_var))
                self.resample_B = self.embedding_normalize(self.resample(y_B, y_B_var))
        else:
            y_A, y_A_var, y_A_s2 = self.compute_mu_and_var(self.real_A_E, o.bnn_T, True)
            y_B, y_B_var, y_B_s2 = self.compute_mu_and_var(self.real_B_E, o.bnn_T, True)
            if 'a' in o.noisy_var_type:
                self.resample_A = self.embedding_normalize(self.resample(y_A, y_A_s2 + y_A_var))
                self.resample_B = self.embedding_normalize(self.resample(y_B, y_B_s2 + y_B_var))
        self.y_A, self.y_B = y_A, y_B
        self.embedding_A = self.embedding_normalize(y_A)
        self.embedding_B = self.embedding_normalize(y_B)
        if o.lr_E <= 0.0:
            self.y_A, self.y_B = self.y_A.detach(), self.y_B.detach()
            self.embedding_A, self.embedding_B = self.embedding_A.detach(), self.embedding_B.detach()
            if o.noisy_var_type:
                self.resample_A = self.resample_A.detach()
                self.resample_B = self.resample_B.detach()
        self.fake_B = self.netG(self.real_A, self.embedding_B)
        self.fake_B_IP = upsample2d(self.fake_B, o.fineSize_IP)
        self.fake_B_E = upsample2d(self.fake_B, o.fineSize_E)
        src = self.fake_B.detach() if o.detach_fake_B else self.fake_B
        self.rec_A = self.netG(src, self.embedding_A)

    def backward_GE(self):
        """models/wsgan_emb_model.py:331-369: the generator's losses without the rating reconstruction, graph retained"""
        self._common_G_losses()
        self.loss_G = self.loss_G_GAN + self.loss_G_IP + self.loss_G_L1 + self.loss_G_cycle + self.loss_G_GAN_cycle
        self.loss_G.backward(retain_graph=True)

    def backward_G_alone(self):
        """models/wsgan_emb_model.py:439-449: the rating reconstruction through the RETAINED graph of fake_B"""
        o = self.opt
        self.loss_z_rec = 0.0
        if o.lambda_z > 0.0:
            pred = self.netE(self.fake_B_E)
            pred = self.embedding_normalize(pred[0] if o.noisy else pred)
            self.loss_z_rec = F.mse_loss(pred, self.embedding_B.detach()) * o.lambda_z
            self.loss_z_rec.backward()

    def _common_G_losses(self):
        o = self.opt
        zB = self.resample_B if (o.noisy_var_type and o.noisy_D) else self.embedding_B
        self.loss_G_GAN = gan_loss(self.netD(self.fake_B, zB), True)
        self.loss_G_GAN_cycle = 0.0
        if o.lambda_A_GAN > 0.0:
            self.loss_G_GAN_cycle = gan_loss(self.netD(self.rec_A, self.embedding_A), True) * o.lambda_A_GAN
        self.loss_G_L1 = F.l1_loss(self.fake_B, self.real_A) * o.lambda_L1 if o.lambda_L1 > 0.0 else 0.0
        self.loss_G_IP = 0.0
        if o.lambda_IP > 0.0:
            feature_A = self.netIP(self.real_A_IP).detach()
            crit = F.mse_loss if o.identity_preserving_criterion.lower() == 'mse' else F.l1_loss
            self.loss_G_IP = crit(self.netIP(self.fake_B_IP), feature_A) * o.lambda_IP
        self.loss_G_cycle = F.l1_loss(self.rec_A, self.real_A) * o.lambda_A if o.lambda_A > 0.0 else 0.0

    def backward_G(self):
        """models/wsgan_emb_model.py:371-437"""
        o = self.opt
        self._common_G_losses()
        self.loss_z_rec = 0.0
        if o.lambda_z > 0.0:
            if not o.bayesian and not o.noisy:
                pred_y = self.netE(self.fake_B_E)
            elif not o.bayesian and o.noisy:
                pred_y, y_logvar = self.netE(self.fake_B_E)
                if 'a' in o.noisy_var_type:
                    y_var = torch.exp(y_logvar)
            elif o.bayesian and not o.noisy:
                pred_y, y_var = self.compute_mu_and_var(self.fake_B_E, o.bnn_T, False)
                if 'e' in o.noisy_var_type:
                    y_logvar = torch.log(y_var + MAGIC_EPS)
            else:
                # reference quirk: prediction from real_A_E, not fake_B_E
                pred_y, y_var_, y_s2_ = self.compute_mu_and_var(self.real_A_E, o.bnn_T, True)
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
                self.loss_z_rec = F.mse_loss(pred_y, self.y_B) * o.lambda_z
        self.loss_G = self.loss_G_GAN + self.loss_G_IP + self.loss_G_L1 + self.loss_G_cycle + self.loss_z_rec \
            + self.loss_G_GAN_cycle
        self.loss_G.backward()

    def backward_D(self):
        """models/wsgan_emb_model.py:300-329"""
        o = self.opt
        zB = self.resample_B if (o.noisy_var_type and o.noisy_D) else self.embedding_B
        self.loss_D_fake = gan_loss(self.netD(self.fake_B.detach(), zB.detach()), False)
        img = self.real_A if o.use_real_A else self.real_B
        z_right = self.embedding_A if o.use_real_A else self.embedding_B
        z_wrong = self.embedding_B if o.use_real_A else self.embedding_A
        self.loss_D_real_right = gan_loss(self.netD(img, z_right.detach()), True)
        self.loss_D_real_wrong = gan_loss(self.netD(img, z_wrong.detach()), relabel(o.relabel_D, self.label_AB))
        self.loss_D = (self.loss_D_fake + (self.loss_D_real_right + self.loss_D_real_wrong) * 0.5) * 0.5
        self.loss_D.backward()

    def optimize_parameters(self):
        """models/wsgan_emb_model.py:451-484 (G first, then D: SURVEY D7)"""
        self.forward()
        for p in self.netD.parameters():
            p.requires_grad = False
        grab = lambda net: {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in net.named_parameters()}  # noqa: E731
        if self.opt.lr_E > 0.0:
            # update_G_and_E, models/wsgan_emb_model.py:463-476 (AdamThroughData: defined semantics, parity unpinned)
            self.optimizer_G.zero_grad()
            self.optimizer_E.zero_grad()
            self.backward_GE()
            self.grads_G, self.grads_E = grab(self.netG), grab(self.netE)
            self.optimizer_G.step()
            self.optimizer_E.step()
            if self.opt.lambda_z > 0.0:
                self.optimizer_G.zero_grad()
                self.optimizer_E.zero_grad()
                self.backward_G_alone()
                self.grads_G_alone = grab(self.netG)
                self.optimizer_G.step()
        else:
            self.optimizer_G.zero_grad()
            self.backward_G()
            self.grads_G = grab(self.netG)
            self.optimizer_G.step()
        for p in self.netD.parameters():
            p.requires_grad = True
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.grads_D = {k: p.grad.detach().clone() for k, p in self.netD.named_parameters()}
        self.optimizer_D.step()

    LOSS_NAMES = ['G_GAN', 'G_GAN_cycle', 'G_IP', 'G_L1', 'G_cycle', 'z_rec', 'D_real_right', 'D_real_wrong',
                  'D_fake']

    def losses(self):
        """models/base_model.py:87-93"""
        return {n: float(v.detach() if isinstance(v, torch.Tensor) else v) for n, v in ((n, getattr(self, 'loss_' + n)) for n in self.LOSS_NAMES)}


# ============================================================================= wsgan_cycle
CYCLE_DEFAULTS = dict(fineSize_E=224, fineSize_IP=224, attr_mean=[0.0], attr_std=[100.0], lambda_x=1.0, lambda_y=1.0,
                      lambda_IP=1.0, lr=2e-4, beta1=0.5, identity_preserving_criterion='mse')


class WSGANCycleStepRef:
    """CPU restatement of WSGANCycleModel's step (reference models/wsgan_cycle_model.py:148-256): unconditional D,
    encoder E trained together with G; D is updated first, then G and E."""

    LOSS_NAMES = ['G_GAN', 'G_IP', 'cycle_x', 'cycle_y', 'D_real', 'D_fake']

    def __init__(self, netG, netD, netE, netIP, **opts):
        o = dict(CYCLE_DEFAULTS)
        o.update(opts)
        self.opt = SimpleNamespace(**o)
        self.netG, self.netD, self.netE, self.netIP = netG, netD, netE, netIP
        betas = (self.opt.beta1, 0.999)                                        # :125-127
        self.optimizer_G = torch.optim.Adam(netG.parameters(), lr=self.opt.lr, betas=betas)
        self.optimizer_E = torch.optim.Adam(netE.parameters(), lr=self.opt.lr, betas=betas)
        self.optimizer_D = torch.optim.Adam(netD.parameters(), lr=self.opt.lr, betas=betas)
        self.grads = {}

    def attr_normalize(self, x):
        """:134"""
        return (x - self.opt.attr_mean[0]) / self.opt.attr_std[0]

    def set_input(self, A, B_attr):
        """:148-163 (training branch)"""
        self.real_x = A
        self.real_y = self.attr_normalize(B_attr)
        self.real_x_IP = upsample2d(self.real_x, self.opt.fineSize_IP)
        self.real_x_E = upsample2d(self.real_x, self.opt.fineSize_E)

    def forward(self):
        """:165-171 -- the encoder input is not normalised"""
        self.fake_x = self.netG(self.real_x, self.real_y)
        self.fake_x_IP = upsample2d(self.fake_x, self.opt.fineSize_IP)
        self.fake_x_E = upsample2d(self.fake_x, self.opt.fineSize_E)
        self.fake_y = self.netE(self.real_x_E)
        self.rec_x = self.netG(self.real_x, self.fake_y)
        self.rec_y = self.netE(self.fake_x_E)

    def backward_D(self):
        """:187-201"""
        self.loss_D_fake = gan_loss(self.netD(self.fake_x.detach()), False)
        self.loss_D_real = gan_loss(self.netD(self.real_x), True)
        self.loss_D = (self.loss_D_fake + self.loss_D_real) * 0.5
        self.loss_D.backward()

    def backward_GE(self):
        """:203-238 (transform_IP is the identity: SURVEY D8)"""
        o = self.opt
        self.loss_G_GAN = gan_loss(self.netD(self.fake_x), True)
        if o.lambda_IP > 0.0:
            feature_A = self.netIP(self.real_x_IP).detach()
            crit = F.mse_loss if o.identity_preserving_criterion.lower() == 'mse' else F.l1_loss
            self.loss_G_IP = crit(self.netIP(self.fake_x_IP), feature_A) * o.lambda_IP
        else:
            self.loss_G_IP = 0.0
        self.loss_cycle_x = F.l1_loss(self.rec_x, self.real_x) * o.lambda_x if o.lambda_x > 0.0 else 0.0
        self.loss_cycle_y = F.mse_loss(self.rec_y, self.real_y) * o.lambda_y if o.lambda_y > 0.0 else 0.0
        self.loss_G = self.loss_G_GAN + self.loss_G_IP + self.loss_cycle_x + self.loss_cycle_y
        self.loss_G.backward()

    def optimize_parameters(self):
        """:240-256"""
        self.forward()
        for p in self.netD.parameters():
            p.requires_grad = True
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.grads['D'] = {k: p.grad.detach().clone() for k, p in self.netD.named_parameters()}
        self.optimizer_D.step()
        for p in self.netD.parameters():
            p.requires_grad = False
        self.optimizer_G.zero_grad()
        self.optimizer_E.zero_grad()
        self.backward_GE()
        self.grads['G'] = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in self.netG.named_parameters()}
        self.grads['E'] = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in self.netE.named_parameters()}
        self.optimizer_G.step()
        self.optimizer_E.step()

    def losses(self):
        return {n: float(getattr(self, 'loss_' + n)) for n in self.LOSS_NAMES}
