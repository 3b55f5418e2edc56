"""Loader image pipeline: the GPU kernel (pcgan_image_transform) against the PIL path it replaces, UTKFace geometry
(200x200 -> 143x143 bicubic -> 128x128 crop), one training batch of pairs (64 images).
Algorithmic bytes per image: 200*200*3 read + 3*128*128*4 written = 316.6 KB."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from pcgan_amd.data.gpu_transform import GpuTransform
from pcgan_amd.data.base_dataset import get_transform


class O(object):
    loadSize, fineSize, transforms, isTrain, no_flip = 143, 128, 'resize_and_crop', True, False


n = 64
rng = np.random.default_rng(0)
arrs = [rng.integers(0, 256, (200, 200, 3), dtype=np.uint8) for _ in range(n)]
dev = torch.device('cuda:0')
tf = GpuTransform(O, dev)
aug = [(random.randint(0, 15), random.randint(0, 15), random.randint(0, 1)) for _ in range(n)]
imgs = [torch.from_numpy(a) for a in arrs]
for _ in range(3):
    tf(imgs, aug)
torch.cuda.synchronize()
# kernel alone: inputs resident
import ctypes
from pcgan_amd.hip import lib as L
g = tf.geometry(200, 200, 3)
src = torch.stack(imgs).to(dev)
a = torch.tensor([list(x) + [i] for i, x in enumerate(aug)], dtype=torch.int32, device=dev)
out = torch.empty(n, 3, 128, 128, device=dev)
st = torch.cuda.current_stream().cuda_stream
def launch():
    L.check(L.load().pcgan_image_transform(ctypes.byref(g.desc), src.data_ptr(), g.kh.data_ptr(), g.bh.data_ptr(), g.kv.data_ptr(),
                                           g.bv.data_ptr(), a.data_ptr(), out.data_ptr(), n, g.band, g.rows, st), 'image_transform')
for _ in range(5):
    launch()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100):
    launch()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 100
bytes_ = n * (200 * 200 * 3 + 3 * 128 * 128 * 4)
print('kernel: %.1f us per batch of %d  = %.0f images/s, %.1f GB/s algorithmic (HBM-bound op; 8 TB/s peak)' % (ms * 1e3, n, n / ms * 1e3, bytes_ / ms / 1e6))
t0 = time.perf_counter()
for _ in range(20):
    tf(imgs, aug)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print('host call incl. stacking + H2D of the decoded bytes: %.2f ms per batch = %.0f images/s' % (dt * 1e3, n / dt))
pil = get_transform(O)
pimgs = [Image.fromarray(x) for x in arrs]
t0 = time.perf_counter()
for im in pimgs:
    pil(im)
dt = (time.perf_counter() - t0)
print('PIL path, 1 core: %.2f ms per image = %.0f images/s' % (dt / n * 1e3, n / dt))
if os.environ.get('PCGAN_PROFILE_HOST'):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        tf(imgs, aug)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
