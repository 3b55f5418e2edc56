"""Writes seeded-random stand-ins for the checkpoints the training scripts expect (none ship offline):
  <out>/embedding_encoder.pth   SiameseFeature(ResNet-18, avg pool, cnn_dim [32, 1])  -> --pretrained_model_path_E (wsgan_emb)
  <out>/resnet18_base.pth       plain ResNet-18 trunk state_dict                       -> --pretrained_model_path_E (wsgan_cycle)
  <out>/alexnet.pth             AlexNetFeature                                         -> --pretrained_model_path_IP
usage: python scripts/make_random_pretrained.py <out_dir>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                   # noqa: E402
from pcgan_amd.models import networks          # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else 'pretrained_models'
os.makedirs(out, exist_ok=True)
torch.manual_seed(0)
e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
torch.save(e.state_dict(), os.path.join(out, 'embedding_encoder.pth'))
torch.save(e.base.model.state_dict(), os.path.join(out, 'resnet18_base.pth'))
torch.save(networks.define_IP('alexnet', 3).state_dict(), os.path.join(out, 'alexnet.pth'))
print('wrote', sorted(os.listdir(out)))
