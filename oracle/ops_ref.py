"""ORACLE (test infrastructure, not product code).

CPU restatement, op by op, of the torch.nn calls the reference's hot path makes.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product path (pc-gan_amd/) never does.

Every function states the reference call site it restates (file:line relative to the
reference root).  The arithmetic is stock PyTorch-CPU -- the same third-party dependency
the reference itself executes (the reference pins no torch version; this container has
torch 2.10.0) -- optionally in float64 so tests can judge fp32 results against an
fp64 twin (tolerance basis).
"""
import torch
import torch.nn.functional as F


def conv2d(x, w, b=None, stride=1, pad=0, pad_mode=0):
    """nn.ReflectionPad2d(pad) + nn.Conv2d(padding=0)   (models/networks.py:578-579, 624-633)
    or nn.Conv2d(padding=pad)                            (models/networks.py:586-587, 748-772)."""
    if pad_mode == 1 and pad > 0:
        x = F.pad(x, (pad, pad, pad, pad), mode='reflect')
        pad = 0
    return F.conv2d(x, w, b, stride=stride, padding=pad)


def conv_transpose2d(x, w, b=None, stride=2, pad=1, out_pad=1):
    """nn.ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1)  (models/networks.py:597-600)."""
    return F.conv_transpose2d(x, w, b, stride=stride, padding=pad, output_padding=out_pad)


def activation(x, act, slope=0.0):
    """nn.ReLU / nn.LeakyReLU / nn.Tanh / nn.Sigmoid  (models/networks.py:581,605,749,775)."""
    if act == 0:
        return x
    if act == 1:
        return F.relu(x)
    if act == 2:
        return F.leaky_relu(x, slope)
    if act == 3:
        return torch.tanh(x)
    if act == 4:
        return torch.sigmoid(x)
    raise ValueError(act)


def instance_norm(x, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, training=True):
    """nn.InstanceNorm2d(affine=False, track_running_stats=True)  (models/networks.py:26)."""
    return F.instance_norm(x, running_mean, running_var, None, None, use_input_stats=training,
                           momentum=momentum, eps=eps)


def batch_norm(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, training=True):
    """nn.BatchNorm2d(affine=True) -- train mode on the hot path
    (models/networks.py:24, models/resnet.py:47,51,136)."""
    return F.batch_norm(x, running_mean, running_var, gamma, beta, training=training, momentum=momentum, eps=eps)


def max_pool2d(x, k, stride, pad):
    """nn.MaxPool2d  (models/resnet.py:138; models/networks.py:1225,1228,1235)."""
    return F.max_pool2d(x, k, stride, pad)


def global_pool(x, is_max):
    """nn.AvgPool2d(H) / nn.MaxPool2d(H)  (models/networks.py:1056-1059)."""
    return F.max_pool2d(x, x.size(2)) if is_max else F.avg_pool2d(x, x.size(2))


def upsample2d(x, size):
    """util.upsample2d: bilinear, align_corners=True, identity when sizes match or target <= 0
    (util/util.py:111-117)."""
    if size <= 0 or x.size(2) == size:
        return x
    return F.interpolate(x, size=(size, size), mode='bilinear', align_corners=True)


def concat_z(img, z):
    """z.view(B',nz,1,1).expand(B,nz,H,W); cat on channels  (models/networks.py:610-611, 781-782)."""
    z_img = z.view(z.size(0), z.size(1), 1, 1).expand(img.size(0), z.size(1), img.size(2), img.size(3))
    return torch.cat((img, z_img), 1)


def dropout2d_with_mask(x, mask_nc, p):
    """nn.Dropout2d(p) with the Bernoulli keep-mask made explicit (models/resnet.py:38-51)."""
    return x * mask_nc.view(x.size(0), x.size(1), 1, 1) / (1.0 - p)


def bce_loss(pred, target_n):
    """GANLoss with nn.BCELoss: per-sample target reshaped (n,1,1,1) and expanded
    (models/networks.py:395-420)."""
    t = target_n.view(-1, 1, 1, 1).to(pred.dtype).expand_as(pred)
    return F.binary_cross_entropy(pred, t)


def l1_loss(a, b):
    """nn.L1Loss  (models/wsgan_emb_model.py:141,149)."""
    return F.l1_loss(a, b)


def mse_loss(a, b):
    """nn.MSELoss  (models/wsgan_emb_model.py:143,148)."""
    return F.mse_loss(a, b)


def adam_reference(params, grads, lr, beta1, beta2=0.999, eps=1e-8, steps=1):
    """torch.optim.Adam(lr, betas=(beta1, 0.999))  (models/wsgan_emb_model.py:153-154):
    applies `steps` updates with the same grads; returns the updated params."""
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.Adam(ps, lr=lr, betas=(beta1, beta2), eps=eps)
    for _ in range(steps):
        for p, g in zip(ps, grads):
            p.grad = g.clone()
        opt.step()
    return [p.detach() for p in ps]
