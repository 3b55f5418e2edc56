"""Single-image dataset of the reference (data/single_dataset.py:8-56): the images of `--sourcefile_A` (first token of each line,
under --dataroot) or of the --dataroot directory, optionally --sorted; yields {'A', 'A_paths'} -- what the image generator
(generate_images.py) feeds to `set_input` for `sample_from_label`.  `--dataroot synthetic`: seeded images."""
import os.path
import random

import torch

from .base_dataset import BaseDataset, get_transform

IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp')


class SingleDataset(BaseDataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def name(self):
        return 'SingleImageDataset'

    def initialize(self, opt):
        self.opt = opt
        self.root = opt.dataroot
        self.synthetic = opt.dataroot == 'synthetic'
        if self.synthetic:
            self.A_paths = ['synthetic_A_%d.png' % i for i in range(int(min(opt.max_dataset_size, 64)))]
            return
        if opt.sourcefile_A:
            with open(opt.sourcefile_A, 'r') as f:
                self.A_paths = [os.path.join(self.root, line.rstrip('\n').split()[0]) for line in f.readlines() if line.strip()]
        else:       # image_folder.make_dataset of the reference: every image file under the root
            self.A_paths = [os.path.join(d, n) for d, _, names in sorted(os.walk(self.root)) for n in names if n.lower().endswith(IMG_EXTENSIONS)]
        if opt.sorted:
            self.A_paths = sorted(self.A_paths)
        self.transform = get_transform(opt)

    def __getitem__(self, index):
        o = self.opt
        nc = o.output_nc if getattr(o, 'which_direction', 'AtoB') == 'BtoA' else o.input_nc
        if self.synthetic:
            g = torch.Generator().manual_seed(2468 + index)
            return {'A': torch.rand(nc, o.fineSize, o.fineSize, generator=g) * 2 - 1, 'A_paths': self.A_paths[index]}
        from PIL import Image
        A_path = self.A_paths[index]
        A = self.transform(Image.open(A_path).convert('RGB'))
        if nc == 1:
            A = (A[0] * 0.299 + A[1] * 0.587 + A[2] * 0.114).unsqueeze(0)
        return {'A': A, 'A_paths': A_path}

    def reshuffle(self, rng=random):
        if not self.synthetic:
            rng.shuffle(self.A_paths)

    def __len__(self):
        # The reference reshuffles on EVERY len() (data/single_dataset.py:45-47), also under --sorted -- so its --sorted order only
        # survives until the DataLoader asks for the length.  INTENTIONAL DEVIATION (deviations): --sorted keeps the sorted
        # order here (generate_images.py / test.py / siamese.py --mode embedding rely on a fixed order so that the i-th output belongs
        # to the i-th path); without --sorted the reference's reshuffle-on-len() behaviour is kept.
        if not getattr(self, 'external_shuffle', False) and not self.opt.sorted:
            self.reshuffle()
        return len(self.A_paths)
