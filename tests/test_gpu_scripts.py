"""GPU: the reference's driver scripts run unchanged command lines end to end on synthetic data: test.py (reference
test.py:9-37) writes one PNG per visual and the result page from a checkpoint train.py's model wrote."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_test_py_writes_results(dev, tmp_path):
    import torch
    from PIL import Image
    from pcgan_amd.models import networks
    ck = tmp_path / 'ck' / 'run'
    os.makedirs(ck)
    torch.manual_seed(0)
    G = networks.define_G(3, 3, 1, 8, which_model_netG='resnet_2blocks', norm='instance', init_type='normal', gpu_ids=[])
    E = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    torch.save(G.state_dict(), ck / 'latest_net_G.pth')
    torch.save(E.state_dict(), ck / 'latest_net_E.pth')
    cmd = [sys.executable, os.path.join(ROOT, 'test.py'), '--dataroot', 'synthetic', '--model', 'wsgan_emb', '--name', 'run',
           '--checkpoints_dir', str(tmp_path / 'ck'), '--results_dir', str(tmp_path / 'res'), '--which_model_netG', 'resnet_2blocks',
           '--ngf', '8', '--fineSize', '32', '--loadSize', '32', '--fineSize_E', '64', '--how_many', '3', '--gpu_ids', '0']
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    img_dir = tmp_path / 'res' / 'run' / 'test_latest' / 'images'
    names = sorted(os.listdir(img_dir))
    assert len(names) == 9 and all(n.endswith(('_real_A.png', '_real_B.png', '_fake_B.png')) for n in names), names
    im = np.asarray(Image.open(img_dir / names[0]))
    assert im.shape == (32, 32, 3) and im.dtype == np.uint8
    page = open(tmp_path / 'res' / 'run' / 'test_latest' / 'index.html').read()
    assert page.count('<h3>') == 3 and names[0] in page
    assert os.path.exists(ck / 'test_net_G.pth')
