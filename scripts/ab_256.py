"""A/B of the three routes of the fp32 convolutions (fp16 two-piece split = default, three-piece bf16 split, fp32 MFMA) on BASELINE
configs 4 and 5 at 256x256: the switches are flipped inside ONE process, rounds interleaved (rule 24 of the HIP guide), wall time and host-side issue time per step."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pcgan_amd.hip import ops
from pcgan_amd.options.train_options import TrainOptions
from pcgan_amd.models import create_model, networks

tmp = tempfile.mkdtemp()
STEPS = int(os.environ.get('AB_STEPS', 6))
ROUNDS = int(os.environ.get('AB_ROUNDS', 3))


def parse(argv):
    old, sys.argv = sys.argv, argv
    so, sys.stdout = sys.stdout, open(os.devnull, 'w')
    try:
        opt = TrainOptions().parse()
        m = create_model(opt); m.setup(opt)
    finally:
        sys.stdout.close(); sys.argv, sys.stdout = old, so
    return m, opt


torch.manual_seed(0)
e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7, noisy=True, bnn_dropout=0.2)
torch.save(e.state_dict(), tmp + '/E.pth'); torch.save(e.base.model.state_dict(), tmp + '/base.pth')
torch.save(networks.define_IP('alexnet', 3).state_dict(), tmp + '/IP.pth')
common = ['--dataroot', 'synthetic', '--checkpoints_dir', tmp, '--gpu_ids', '0', '--which_model_netG', 'resnet_9blocks',
          '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--fineSize', '256', '--loadSize', '256', '--display_id', '-1',
          '--pretrained_model_path_IP', tmp + '/IP.pth']


def ab(name, m, batch, nimg):
    dev = torch.device('cuda:0')
    # pinned host batch, uploaded by set_input every step (BaseModel.to_act: the upload carries the readiness event the ahead-of-step encoder
    # passes wait for; a resident tensor nobody declared ready -- ops.mark_ready -- is taken in plain stream order and does not run ahead)
    batch = {k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    routes = [('fp16x3 (default)', True, True), ('bf16x6', True, False), ('fp32 MFMA', False, False)]
    res = {r[0]: [] for r in routes}
    host = {r[0]: [] for r in routes}

    def select(r):
        ops.BF16X6, ops.HSPLIT = r[1], r[2]
    for r in routes:          # warm every route (packed weights, allocator)
        select(r)
        for _ in range(2):
            m.set_input(batch); m.optimize_parameters()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for r in routes:
            select(r)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(STEPS):
                m.set_input(batch); m.optimize_parameters()
            t1 = time.perf_counter()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            res[r[0]].append((t2 - t0) / STEPS * 1e3); host[r[0]].append((t1 - t0) / STEPS * 1e3)
    select(routes[0])
    for r in routes:
        w = sorted(res[r[0]]); h = sorted(host[r[0]])
        print('%s  %-17s wall ms/step min %.1f median %.1f  (%.1f img/s)   host issue ms/step min %.1f median %.1f' % (
            name, r[0], w[0], w[len(w) // 2], nimg / w[len(w) // 2] * 1e3, h[0], h[len(h) // 2]), flush=True)


m, opt = parse(['x', '--model', 'wsgan_emb', '--name', 'c4', '--batchSize', '8', '--noisy', 'true', '--bayesian', 'true',
                '--bnn_dropout', '0.2', '--noisy_var_type', 'ae', '--pretrained_model_path_E', tmp + '/E.pth'] + common)
ab('config 4 (wsgan_emb 256x256 bs8, bayesian+noisy ae, T=10)', m, bench.synthetic_batch(8, 256, 0), 8)
del m
torch.cuda.empty_cache()
m, opt = parse(['x', '--model', 'wsgan_cycle', '--name', 'c5', '--batchSize', '16', '--attr_bins', '[10,30,50]',
                '--pretrained_model_path_E', tmp + '/base.pth'] + common)
g = torch.Generator().manual_seed(1)
batch = {'A': torch.rand(16, 3, 256, 256, generator=g) * 2 - 1, 'B_attr': torch.rand(16, 1, 1, 1, generator=g) * 100,
         'A_paths': [''] * 16, 'B_paths': [''] * 16}
ab('config 5 (wsgan_cycle 256x256 bs16)', m, batch, 16)
