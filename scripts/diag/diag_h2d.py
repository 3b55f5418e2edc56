"""where the host time of an image-batch upload goes (GPU box): stack / pageable H2D / pinned staging variants"""
import time, numpy as np, torch
n = 64
imgs = [torch.from_numpy(np.random.randint(0, 256, (200, 200, 3), dtype=np.uint8)) for _ in range(n)]
dev = torch.device('cuda:0')
torch.zeros(1, device=dev); torch.cuda.synchronize()
def t(label, fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%-50s issue %.2f ms  incl. sync %.2f ms' % (label, (t1 - t0) / reps * 1e3, (t2 - t0) / reps * 1e3))
t('torch.stack (pageable)', lambda: torch.stack(imgs))
st = torch.stack(imgs)
t('pageable .to(dev)', lambda: st.to(dev, non_blocking=True))
pin = torch.empty((n, 200, 200, 3), dtype=torch.uint8).pin_memory()
t('pinned .to(dev)', lambda: pin.to(dev, non_blocking=True))
def per_image():
    for j, im in enumerate(imgs):
        pin[j].copy_(im)
t('64 x pin[j].copy_(im)', per_image)
t('pin.copy_(stacked)', lambda: pin.copy_(st))
pn = pin.numpy()
def per_image_np():
    for j, im in enumerate(imgs):
        pn[j] = im.numpy()
t('64 x numpy assignment into pinned', per_image_np)
t('torch.stack(out=pinned)', lambda: torch.stack(imgs, out=pin))
print('threads', torch.get_num_threads())
torch.set_num_threads(1)
t('64 x pin[j].copy_(im), 1 thread', per_image)
t('torch.stack (pageable), 1 thread', lambda: torch.stack(imgs))
