"""ORACLE (test infrastructure, not product code).

CPU restatement of the Elo-encoder training step (reference siamese.py:526-553, 577-669 with
models/networks.py:473-482 and :872-992: the deterministic recipe and its reparameterised / MC-dropout variants): SiameseNetwork = SiameseFeature trunk + head applied to
two images, score = y1 - y2, prob = sigmoid(score), draw-aware binary NLL, Adam(lr) on trunk + head.
Pinned against vectors captured from the reference's own SiameseNetwork by tests/test_siamese_oracle_golden.py."""
import torch

from . import networks_ref as N

MAGIC_EPS = 1e-20


def binary_nll(prob, label):
    """models/networks.py:473-482 (the reference class cannot be constructed without a GPU: its __init__ calls .cuda())"""
    lut = torch.tensor([0.0, 0.5, 1.0], dtype=prob.dtype)
    target = lut[label].reshape(prob.size(0), 1, 1, 1).expand(prob.size(0), 1, prob.size(2), prob.size(3))
    return -(target * torch.log(prob + MAGIC_EPS) + (1 - target) * torch.log(1 - prob + MAGIC_EPS)).mean()


def reparameterize(mu, logvar, draws=None):
    """util/util.py:130-133"""
    std = torch.exp(0.5 * logvar)
    eps = torch.randn_like(std)
    if draws is not None:
        draws.append(eps.detach().clone())
    return mu + eps * std


class SiameseTrainRef:
    """deterministic recipe by default; noisy / rsample / lb_or_mc / bnn_dropout / T_train / M select the reparameterised and
    MC-dropout branches of the trainer's iteration (siamese.py:577-669).  Randomness is drawn where the reference draws it
    (Dropout2d masks inside the net, one randn_like per reparameterize call): under the same torch seed the numbers coincide with
    the reference's; every eps is recorded in `self.draws`, the masks by networks_ref.Dropout2dRec.record."""

    def __init__(self, cnn_dim=(32, 1), pooling='avg', slope=0.7, lr=2e-4, noisy=False, rsample=True, lb_or_mc='lb',
                 bnn_dropout=0.0, T_train=1, M=1, lr_sigma=2e-7):
        self.noisy, self.rsample, self.lb_or_mc, self.bayesian = noisy, rsample, lb_or_mc, bnn_dropout > 0
        self.T, self.M = (T_train if bnn_dropout > 0 else 1), M
        self.net = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18', bnn_dropout), pooling, cnn_dim, 1, slope, noisy, bnn_dropout)
        params = list(self.net.base.parameters()) + list(self.net.cnn.parameters())      # siamese.py:545-551
        self.optimizer = torch.optim.Adam(params, lr=lr)
        self.optimizer_sigma = torch.optim.Adam(self.net.cnn_logvar.parameters(), lr=lr_sigma) if noisy else None   # :552-553
        self.draws = []

    def _pair(self, img0, img1):
        a, b = self.net(img0), self.net(img1)
        return (a[0], b[0], a[1], b[1]) if self.noisy else (a, b, None, None)

    def step(self, img0, img1, label):
        """siamese.py:577-586, 596-669"""
        self.optimizer.zero_grad()
        if self.noisy:
            self.optimizer_sigma.zero_grad()
        loss = 0.0
        for _ in range(self.T):
            y1, y2, lv1, lv2 = self._pair(img0, img1)
            if self.noisy and self.rsample:
                if self.lb_or_mc == 'mc':
                    prob = 0.0
                    for _m in range(self.M):
                        score = reparameterize(y1, lv1, self.draws) - reparameterize(y2, lv2, self.draws)
                        prob = prob + 1. / self.M * torch.sigmoid(score)
                    loss = loss + 1. / self.T * binary_nll(prob, label)
                else:
                    for _m in range(self.M):
                        score = reparameterize(y1, lv1, self.draws) - reparameterize(y2, lv2, self.draws)
                        prob = torch.sigmoid(score)
                        loss = loss + 1. / (self.T * self.M) * binary_nll(prob, label)
            elif self.noisy:       # models/networks.py:985-989: std of the difference, then score / (std + eps)
                std = torch.sqrt(torch.exp(0.5 * lv1).pow(2) + torch.exp(0.5 * lv2).pow(2))
                self.score = y1 - y2
                prob = torch.sigmoid(self.score / (std + MAGIC_EPS))
                loss = loss + 1. / self.T * binary_nll(prob, label)
            else:
                self.score = y1 - y2
                prob = torch.sigmoid(self.score)
                loss = loss + 1. / self.T * binary_nll(prob, label)
        self.y1, self.y2, self.prob, self.loss = y1, y2, prob, loss
        self.loss.backward()
        self.grads = {k: p.grad.detach().clone() for k, p in self.net.named_parameters() if p.grad is not None}
        self.optimizer.step()
        if self.noisy:
            self.optimizer_sigma.step()
        return float(self.loss)
