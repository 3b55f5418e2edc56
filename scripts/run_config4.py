"""BASELINE config 4 (wsgan_emb 256x256 bs 8, bayesian + noisy ae, T = 10): N steps -- for rocprofv3 --kernel-trace --stats"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pcgan_amd.options.train_options import TrainOptions
from pcgan_amd.models import create_model, networks
tmp = tempfile.mkdtemp()
dev = torch.device('cuda:0')
old, sys.argv = sys.argv, None
torch.manual_seed(0)
e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7, noisy=True, bnn_dropout=0.2)
torch.save(e.state_dict(), tmp + '/E.pth')
torch.save(networks.define_IP('alexnet', 3).state_dict(), tmp + '/IP.pth')
sys.argv = ['x', '--model', 'wsgan_emb', '--name', 'c4', '--batchSize', '8', '--noisy', 'true', '--bayesian', 'true', '--bnn_dropout', '0.2',
            '--noisy_var_type', 'ae', '--pretrained_model_path_E', tmp + '/E.pth', '--dataroot', 'synthetic', '--checkpoints_dir', tmp, '--gpu_ids', '0',
            '--which_model_netG', 'resnet_9blocks', '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--fineSize', '256', '--loadSize', '256',
            '--display_id', '-1', '--pretrained_model_path_IP', tmp + '/IP.pth']
so, sys.stdout = sys.stdout, open(os.devnull, 'w')
try:
    opt = TrainOptions().parse()
    m = create_model(opt); m.setup(opt)
finally:
    sys.stdout.close(); sys.stdout = so
b = bench.synthetic_batch(8, 256, 0)
b = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
n = int(old[1]) if len(old) > 1 else 8
for i in range(n + 2):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    m.set_input(b); m.optimize_parameters()
torch.cuda.synchronize()
print('config 4: %.1f ms/step' % ((time.perf_counter() - t0) / n * 1e3))
