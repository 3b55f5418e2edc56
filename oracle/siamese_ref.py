"""ORACLE (test infrastructure, not product code).

CPU restatement of the Elo-encoder training step (reference siamese.py:526-540, 577-586, 660-669 with
models/networks.py:473-482 and :872-992, deterministic recipe): SiameseNetwork = SiameseFeature trunk + head applied to
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


class SiameseTrainRef:
    def __init__(self, cnn_dim=(32, 1), pooling='avg', slope=0.7, lr=2e-4):
        self.net = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), pooling, cnn_dim, 1, slope, False)
        params = list(self.net.base.parameters()) + list(self.net.cnn.parameters())      # siamese.py:545-551
        self.optimizer = torch.optim.Adam(params, lr=lr)

    def step(self, img0, img1, label):
        """siamese.py:577-586, 660-669"""
        self.optimizer.zero_grad()
        self.y1, self.y2 = self.net(img0), self.net(img1)
        self.score = self.y1 - self.y2
        self.prob = torch.sigmoid(self.score)
        self.loss = binary_nll(self.prob, label)
        self.loss.backward()
        self.grads = {k: p.grad.detach().clone() for k, p in self.net.named_parameters() if p.grad is not None}
        self.optimizer.step()
        return float(self.loss)
