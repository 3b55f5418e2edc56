"""Pairwise-comparison dataset of the reference (data/wsgan_emb_dataset.py:9-82): each line of
`--sourcefile_A` is "<fileA> <fileB> <label>" with label 0: A<B, 1: A=B, 2: A>B.

`--dataroot synthetic` short-circuits to seeded synthetic batches of the same dict layout
(no datasets ship with the build)."""
import os.path
import random

import torch

from .base_dataset import BaseDataset, get_transform, decode_raw


class WSGANEmbDataset(BaseDataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def name(self):
        return 'WSGANEmbDataset'

    def initialize(self, opt):
        self.opt = opt
        self.root = opt.dataroot
        self.synthetic = (opt.dataroot == 'synthetic')
        if self.synthetic:
            self.size = int(min(opt.max_dataset_size, 64 * opt.batchSize))
            return
        with open(opt.sourcefile_A, 'r') as f:
            self.sourcefile = [line.rstrip('\n') for line in f.readlines()]
        if opt.no_mixed_label_D:
            per = {L: [] for L in range(len(opt.relabel_D))}
            for line in self.sourcefile:
                per[int(line.split()[2])].append(line)
            self.sourcefiles = {L: v for L, v in per.items() if v}
            self.size = min(max(len(v) for v in self.sourcefiles.values()), opt.max_dataset_size)
        else:
            self.size = min(len(self.sourcefile), opt.max_dataset_size)
        self.transform = get_transform(opt)
        self.raw = bool(getattr(opt, 'gpu_transform', False))

    def _pair(self, line):
        from PIL import Image
        a, b, lab = line.split()[:3]
        pa, pb = os.path.join(self.root, a), os.path.join(self.root, b)
        if self.raw:
            # workers only decode and draw the augmentation (same draws, same order as the PIL path); the per-pixel work
            # is one launch per batch in the loader (gpu_transform.py), which also does the gray mix
            A, B = decode_raw(Image.open(pa).convert('RGB'), self.opt), decode_raw(Image.open(pb).convert('RGB'), self.opt)
            return A, B, int(lab), pa, pb
        A = self.transform(Image.open(pa).convert('RGB'))
        B = self.transform(Image.open(pb).convert('RGB'))
        if self.opt.input_nc == 1:
            A = (A[0] * 0.299 + A[1] * 0.587 + A[2] * 0.114).unsqueeze(0)
        if self.opt.output_nc == 1:
            B = (B[0] * 0.299 + B[1] * 0.587 + B[2] * 0.114).unsqueeze(0)
        return A, B, int(lab), pa, pb

    def __getitem__(self, index):
        o = self.opt
        if self.synthetic:
            g = torch.Generator().manual_seed(1234 + index)
            s = o.fineSize
            A = torch.rand(o.input_nc, s, s, generator=g) * 2 - 1
            B = torch.rand(o.output_nc, s, s, generator=g) * 2 - 1
            label = int(torch.randint(0, 2, (1,), generator=g)) * 2     # "diff" pairs only: {0, 2}
            return {'A': A, 'B': B, 'label': label, 'A_paths': 'synthetic_A_%d' % index,
                    'B_paths': 'synthetic_B_%d' % index}
        if not o.no_mixed_label_D:
            A, B, lab, pa, pb = self._pair(self.sourcefile[index])
            if self.raw:
                return {'A_raw': A[0], 'A_aug': A[1], 'B_raw': B[0], 'B_aug': B[1], 'label': lab, 'A_paths': pa, 'B_paths': pb}
            return {'A': A, 'B': B, 'label': lab, 'A_paths': pa, 'B_paths': pb}
        ret = {}
        for L, lines in self.sourcefiles.items():
            A, B, _, pa, pb = self._pair(lines[index % len(lines)])
            if self.raw:
                ret[str(L) + '_A_raw'], ret[str(L) + '_A_aug'] = A
                ret[str(L) + '_B_raw'], ret[str(L) + '_B_aug'] = B
            else:
                ret[str(L) + '_A'], ret[str(L) + '_B'] = A, B
            ret[str(L) + '_A_paths'], ret[str(L) + '_B_paths'] = pa, pb
        return ret

    def reshuffle(self, rng=random):
        """the reference's pair-list reshuffle (wsgan_emb_dataset.py:72-79), with the generator given"""
        if self.synthetic:
            return
        if not self.opt.no_mixed_label_D:
            rng.shuffle(self.sourcefile)
        else:
            for L in self.sourcefiles:
                rng.shuffle(self.sourcefiles[L])

    def __len__(self):
        # the reference reshuffles its pair list every time len() is taken (wsgan_emb_dataset.py:72-79).  Under
        # torch.distributed every rank must hold the same list order, so the loader takes that over (`external_shuffle`:
        # one reshuffle per epoch from __iter__ with a generator keyed by (shared seed, epoch), data/__init__.py) and len()
        # -- which a single rank may call on its own, e.g. for logging -- leaves the list alone
        if not getattr(self, 'external_shuffle', False):
            self.reshuffle()
        return self.size
