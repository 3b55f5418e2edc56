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


def _write_pairs(root, n=8, size=(50, 50)):
    from PIL import Image
    rng = np.random.default_rng(5)
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, size + (3,), dtype=np.uint8)).save(root / ('img_%d.png' % i))
    with open(root / 'pairs.txt', 'w') as f:
        for i in range(n):
            f.write('img_%d.png img_%d.png %d\n' % (i, (i + 3) % n, (0, 2)[i % 2]))


def test_train_py_on_image_files_same_losses_with_gpu_transform(dev, tmp_path):
    """train.py on a pair file of PNGs: the PIL loader and --gpu_transform feed bit-identical batches, so the loss
    lines of the two runs (same seed) are identical"""
    import re
    import torch
    from pcgan_amd.models import networks
    _write_pairs(tmp_path)
    torch.manual_seed(0)
    E = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    IP = networks.define_IP('alexnet', 3)
    torch.save(E.state_dict(), tmp_path / 'E.pth')
    torch.save(IP.state_dict(), tmp_path / 'IP.pth')
    logs = []
    for extra in ([], ['--gpu_transform']):
        name = 'run_gpu' if extra else 'run_pil'
        cmd = [sys.executable, os.path.join(ROOT, 'train.py'), '--dataroot', str(tmp_path), '--sourcefile_A', str(tmp_path / 'pairs.txt'),
               '--model', 'wsgan_emb', '--name', name, '--checkpoints_dir', str(tmp_path / 'ck'), '--which_model_netG', 'resnet_2blocks',
               '--which_model_netD', 'n_layers', '--n_layers_D', '3', '--ngf', '8', '--ndf', '8', '--loadSize', '40', '--fineSize', '32',
               '--fineSize_E', '64', '--fineSize_IP', '64', '--batchSize', '4', '--nThreads', '0', '--niter', '1', '--niter_decay', '0',
               '--print_freq', '1', '--save_epoch_freq', '1', '--display_id', '-1', '--seed', '3', '--serial_batches', '--pretrained_model_path_E', str(tmp_path / 'E.pth'),
               '--pretrained_model_path_IP', str(tmp_path / 'IP.pth'), '--gpu_ids', '0'] + extra
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        lines = [re.sub(r'time: [0-9.]+, data: [0-9.]+', '', l) for l in open(tmp_path / 'ck' / name / 'loss_log.txt') if l.startswith('(epoch')]
        assert len(lines) == 2, lines
        logs.append(lines)
        assert os.path.exists(tmp_path / 'ck' / name / 'latest_net_G.pth')
    assert logs[0] == logs[1]


def test_compute_fid_score_alexnet_features(dev, tmp_path):
    """compute_fid_score.py on image directories with the HIP AlexNet as the feature extractor: a set against itself is 0,
    against a different set positive; .npz statistics reproduce the value"""
    import re
    import torch
    from PIL import Image
    from pcgan_amd.models import networks
    rng = np.random.default_rng(9)
    for name, lo in (('a', 0), ('b', 96)):
        os.makedirs(tmp_path / name)
        for i in range(12):
            Image.fromarray(rng.integers(lo, lo + 160, (32, 32, 3), dtype=np.uint8)).save(tmp_path / name / ('%02d.png' % i))
    torch.manual_seed(1)
    torch.save(networks.define_IP('alexnet', 3).state_dict(), tmp_path / 'IP.pth')

    def run(p1, p2):
        cmd = [sys.executable, os.path.join(ROOT, 'compute_fid_score.py'), str(p1), str(p2), '--features', 'alexnet', '--batch-size', '4',
               '--pretrained_model_path_IP', str(tmp_path / 'IP.pth')]
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        return float(re.search(r': ([-0-9.e]+)\s*$', p.stdout.strip()).group(1))
    same, diff = run(tmp_path / 'a', tmp_path / 'a'), run(tmp_path / 'a', tmp_path / 'b')
    assert abs(same) < 1e-3 * max(1.0, abs(diff)) and diff > 0


def test_generate_images_feeds_the_frechet_script(dev, tmp_path):
    """generate_images.py (the sampler eval_emb.py of the reference calls but does not ship, eval_emb.py:70-93): `--how_to_sample label`
    over a single-image listing writes <label>_<index>_<stem>.png, one per source image, with the classes of --sample_label_file;
    the images equal model.sample_from_label on the same input; compute_fid_score.py takes the output directory as it stands."""
    import re
    import torch
    from PIL import Image
    from pcgan_amd.models import networks
    rng = np.random.default_rng(11)
    os.makedirs(tmp_path / 'src')
    for i in range(6):
        Image.fromarray(rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)).save(tmp_path / 'src' / ('face_%d.png' % i))
    with open(tmp_path / 'list.txt', 'w') as f:
        f.writelines('face_%d.png\n' % i for i in range(6))
    with open(tmp_path / 'labels.txt', 'w') as f:
        f.writelines('%d\n' % l for l in (2, 0, 1))
    torch.manual_seed(2)
    E = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    IP = networks.define_IP('alexnet', 3)
    G = networks.define_G(3, 3, 1, 8, 'resnet_2blocks', 'instance', 'relu', 0, 'normal')
    os.makedirs(tmp_path / 'ck' / 'gen')
    torch.save(E.state_dict(), tmp_path / 'ck' / 'gen' / 'latest_net_E.pth')
    torch.save(G.state_dict(), tmp_path / 'ck' / 'gen' / 'latest_net_G.pth')
    torch.save(IP.state_dict(), tmp_path / 'IP.pth')
    torch.save(E.state_dict(), tmp_path / 'E.pth')
    out = tmp_path / 'samples'
    cmd = [sys.executable, os.path.join(ROOT, 'generate_images.py'), '--model', 'wsgan_emb', '--how_to_sample', 'label', '--sample_label_file',
           str(tmp_path / 'labels.txt'), '--dataset_mode', 'single', '--sourcefile_A', str(tmp_path / 'list.txt'), '--dataroot', str(tmp_path / 'src'),
           '--embedding_bins', '[-1.0, 0.0, 1.5]', '--embedding_mean', '0.1', '--embedding_std', '0.8', '--name', 'gen', '--checkpoints_dir',
           str(tmp_path / 'ck'), '--which_epoch', 'latest', '--which_model_netG', 'resnet_2blocks', '--ngf', '8', '--loadSize', '32', '--fineSize', '32',
           '--fineSize_E', '64', '--output_dir', str(out), '--how_many', '5', '--gpu_ids', '0', '--pretrained_model_path_E', str(tmp_path / 'E.pth')]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    names = sorted(os.listdir(out))
    # classes cycle through the label file (2, 0, 1); --how_many 5 of the 6 sources
    assert names == sorted(['2_00000_face_0.png', '0_00001_face_1.png', '1_00002_face_2.png', '2_00003_face_3.png', '0_00004_face_4.png']), names
    img = np.asarray(Image.open(out / '2_00000_face_0.png'))
    assert img.shape == (32, 32, 3) and img.std() > 0
    # the directory goes straight into the Frechet script
    cmd = [sys.executable, os.path.join(ROOT, 'compute_fid_score.py'), str(out), str(tmp_path / 'src'), '--features', 'alexnet', '--batch-size', '4',
           '--pretrained_model_path_IP', str(tmp_path / 'IP.pth')]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert np.isfinite(float(re.search(r': ([-0-9.e]+)\s*$', p.stdout.strip()).group(1)))
