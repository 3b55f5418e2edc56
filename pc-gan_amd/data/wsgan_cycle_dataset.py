"""Unpaired image + attribute dataset of the reference (data/wsgan_cycle_dataset.py:10-62): images from
`--sourcefile_A`, attribute values parsed from the file names listed in `--sourcefile_B` ("<attr>_<rest>", e.g. an
age), drawn independently.  Batch dict: 'A' (C,S,S), 'B_attr' (1,1,1), 'A_paths', 'B_paths'.

`--dataroot synthetic` short-circuits to seeded synthetic samples of the same dict layout (U[-1,1) images,
attributes U[0,100))."""
import os.path
import random

import torch

from .base_dataset import BaseDataset, get_transform, decode_raw
from ..util.util import get_attr_value


class WSGANCycleDataset(BaseDataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def name(self):
        return 'WSGANCycleDataset'

    def initialize(self, opt):
        self.opt = opt
        self.root = opt.dataroot
        self.synthetic = (opt.dataroot == 'synthetic')
        if self.synthetic:
            self.A_size = self.B_size = int(min(opt.max_dataset_size, 64 * opt.batchSize))
            return
        with open(opt.sourcefile_A, 'r') as f:
            lines_A = f.readlines()
        with open(opt.sourcefile_B, 'r') as f:
            lines_B = f.readlines()
        self.A_paths = [os.path.join(self.root, p.rstrip('\n').split()[0]) for p in lines_A]
        self.B_paths = [p.rstrip('\n').split()[0] for p in lines_B]      # attribute source only: not joined with the root
        self.A_size, self.B_size = len(self.A_paths), len(self.B_paths)
        self.transform = get_transform(opt)
        self.raw = bool(getattr(opt, 'gpu_transform', False))     # workers decode, the loader finishes the batch on the GPU

    def __getitem__(self, index):
        o = self.opt
        if self.synthetic:
            g = torch.Generator().manual_seed(4321 + index)
            A = torch.rand(o.input_nc, o.fineSize, o.fineSize, generator=g) * 2 - 1
            attr = torch.rand(1, generator=g) * 100.0
            return {'A': A, 'B_attr': attr.reshape(1, 1, 1), 'A_paths': 'synthetic_A_%d' % index,
                    'B_paths': '%d_synthetic_B' % int(attr)}
        from PIL import Image
        A_path = self.A_paths[index % self.A_size]
        index_B = index % self.B_size if o.serial_batches else random.randint(0, self.B_size - 1)
        B_path = self.B_paths[index_B]
        B_attr = torch.Tensor([get_attr_value(B_path)]).reshape(1, 1, 1)
        if self.raw:
            A_raw, A_aug = decode_raw(Image.open(A_path).convert('RGB'), o)
            return {'A_raw': A_raw, 'A_aug': A_aug, 'B_attr': B_attr, 'A_paths': A_path, 'B_paths': B_path}
        A = self.transform(Image.open(A_path).convert('RGB'))
        if o.input_nc == 1:
            A = (A[0] * 0.299 + A[1] * 0.587 + A[2] * 0.114).unsqueeze(0)
        return {'A': A, 'B_attr': B_attr, 'A_paths': A_path, 'B_paths': B_path}

    def reshuffle(self, rng=random):
        if not self.synthetic:
            rng.shuffle(self.A_paths)
            rng.shuffle(self.B_paths)

    def __len__(self):
        # the reference reshuffles both lists every time len() is taken (:53-59); under torch.distributed the loader does it once
        # per epoch with a generator shared by all ranks (`external_shuffle`, data/__init__.py) and len() leaves the lists alone
        if not getattr(self, 'external_shuffle', False):
            self.reshuffle()
        return max(self.A_size, self.B_size)
