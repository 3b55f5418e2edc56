"""Data layer surface of the reference (data/__init__.py:5-69): dataset registry by
`--dataset_mode`, `CreateDataLoader(opt).load_data()` yielding dict batches with the keys the
model's set_input reads ('A', 'B', 'label', 'A_paths', 'B_paths').

Under torch.distributed (one process per GPU) the loader is sharded at the SAMPLER: all ranks walk one
permutation of the data set per epoch (shared seed: `--seed`, else drawn on rank 0 and broadcast), cut it into
global batches of `--batchSize`, and rank r decodes / transforms / uploads only samples [r*B/n, (r+1)*B/n) of each
global batch -- the contiguous slice DataParallel's scatter would hand to device r (reference
models/networks.py:96-102).  The last partial global batch of an epoch is dropped when there is more than one rank
(a ragged batch cannot be split evenly; a single process keeps it, like the reference's DataLoader).
"""
import random

import torch
import torch.utils.data

from .base_dataset import BaseDataset
from ..util.registry import find_plugin


def find_dataset_using_name(dataset_name):
    return find_plugin(__name__, 'dataset', dataset_name, BaseDataset)


def get_option_setter(dataset_name):
    return find_dataset_using_name(dataset_name).modify_commandline_options


def create_dataset(opt):
    dataset = find_dataset_using_name(opt.dataset_mode)()
    dataset.initialize(opt)
    print('dataset [%s] was created' % dataset.name())
    return dataset


def _collate_keep_raw(samples):
    """default collate, except that decoded images ('*_raw', possibly of different sizes) stay a list"""
    raw = {k: [s[k] for s in samples] for k in samples[0] if k.endswith('_raw')}
    rest = torch.utils.data.default_collate([{k: v for k, v in s.items() if not k.endswith('_raw')} for s in samples])
    rest.update(raw)
    return rest


class RankShardedBatchSampler(torch.utils.data.Sampler):
    """Batch sampler of one rank: a seeded per-epoch permutation shared by all ranks, cut into global batches of
    `global_batch`; yields this rank's contiguous slice of every FULL global batch."""

    def __init__(self, n_samples, global_batch, world, rank, shuffle, seed):
        assert global_batch % world == 0, '--batchSize %d (the global batch) must be divisible by the %d ranks' % (global_batch, world)
        self.n, self.B, self.world, self.rank = int(n_samples), int(global_batch), int(world), int(rank)
        self.per = self.B // self.world
        self.shuffle, self.seed, self.epoch = bool(shuffle), int(seed), 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def order(self):
        if not self.shuffle:
            return list(range(self.n))
        g = torch.Generator().manual_seed(self.seed + 7919 * self.epoch)
        return torch.randperm(self.n, generator=g).tolist()

    def __len__(self):
        return self.n // self.B

    def __iter__(self):
        order = self.order()
        for b in range(self.n // self.B):
            lo = b * self.B + self.rank * self.per
            yield order[lo:lo + self.per]


def shared_seed(opt):
    """`--seed` if given, else one number drawn on rank 0 and broadcast (every rank must shuffle alike)."""
    from ..hip import parallel
    if getattr(opt, 'seed', None) is not None:
        return int(opt.seed)
    box = [random.SystemRandom().randrange(1 << 31)]
    if parallel.is_distributed():
        torch.distributed.broadcast_object_list(box, src=0)
    return int(box[0])


class CustomDatasetDataLoader(object):
    """Iterable over dict batches, capped at --max_dataset_size samples (reference data/__init__.py:42-69)."""

    def name(self):
        return 'CustomDatasetDataLoader'

    def initialize(self, opt, world=None, rank=None):
        from ..hip import parallel
        self.opt = opt
        if world is None:
            world, rank = (torch.distributed.get_world_size(), torch.distributed.get_rank()) if parallel.is_distributed() else (1, 0)
        self.world, self.rank = world, rank
        self.dataset = create_dataset(opt)
        self.gpu_transform = None
        if getattr(opt, 'gpu_transform', False) and opt.dataroot != 'synthetic':
            from .gpu_transform import GpuTransform
            self.gpu_transform = GpuTransform(opt, 'cuda:%d' % opt.gpu_ids[0] if opt.gpu_ids else 'cpu')
        collate = _collate_keep_raw if self.gpu_transform else None
        self.sampler = None
        # pinned batches: set_input's upload (BaseModel.to_act) is then an asynchronous copy on the upload stream whose event the
        # ahead-of-step encoder passes wait for -- the form bench.py measures; from pageable memory the copy blocks the host.
        # (The reference's loader, data/__init__.py:58-62, does not pin.)
        pin = bool(torch.cuda.is_available() and getattr(opt, 'gpu_ids', None))
        if world > 1:
            seed = self.seed = shared_seed(opt)
            # the data set reshuffles its pair list whenever its length is taken (reference quirk); here every rank must hold the
            # same list order (index i = the same pair everywhere), so the loader reshuffles once per epoch from __iter__ with a
            # generator keyed by (shared seed, epoch) and a bare len() -- rank-0-only logging -- changes nothing
            self.dataset.external_shuffle = True
            n_samples = min(len(self.dataset), opt.max_dataset_size)
            # everything drawn PER SAMPLE must differ between ranks (DataParallel draws crop / flip / resample noise / dropout
            # independently over the global batch): python's and torch's generators -- and through torch's the DataLoader's
            # worker base seed -- continue from seed + rank.  numpy's global generator stays shared: --no_mixed_label_D draws ONE
            # label for the whole global batch from it (reference models/wsgan_emb_model.py:199-207).
            random.seed(seed * 1000003 + rank + 1)
            torch.manual_seed(seed * 1000003 + rank + 1)
            import numpy as np
            np.random.seed(seed % (1 << 32))
            self.sampler = RankShardedBatchSampler(n_samples, opt.batchSize, world, rank, not opt.serial_batches, seed)
            self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_sampler=self.sampler, num_workers=int(opt.nThreads),
                                                          collate_fn=collate, pin_memory=pin)
        else:
            self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, num_workers=int(opt.nThreads),
                                                          shuffle=not opt.serial_batches, collate_fn=collate, pin_memory=pin)
        self._epochs = 0
        return self

    def _finish_on_gpu(self, batch):
        """'<key>_raw' + '<key>_aug' -> '<key>': one image-pipeline launch per image set (A with input_nc channels, B
        with output_nc, the gray mix of the pair dataset included)"""
        for raw_key in [k for k in batch if k.endswith('_raw')]:
            key = raw_key[:-4]
            channels = self.opt.input_nc if key.endswith('A') else self.opt.output_nc
            batch[key] = self.gpu_transform(batch.pop(raw_key), batch.pop(key + '_aug'), out_channels=channels)
        return batch

    def load_data(self):
        return self

    def __len__(self):
        return min(len(self.dataset), self.opt.max_dataset_size)

    def __iter__(self):
        if self.sampler is not None:
            # one reshuffle of the pair list per epoch, in the parent and BEFORE the workers fork: identical on every rank
            self.dataset.reshuffle(random.Random(self.seed * 7919 + self._epochs))
            self.sampler.set_epoch(self._epochs)
            self._epochs += 1
        seen = 0
        for batch in self.dataloader:
            if seen >= self.opt.max_dataset_size:
                return
            seen += self.opt.batchSize          # (the global batch: every rank stops at the same point)
            yield self._finish_on_gpu(batch) if self.gpu_transform else batch


def CreateDataLoader(opt, world=None, rank=None):
    """world / rank default to the torch.distributed group (1 / 0 without one)."""
    return CustomDatasetDataLoader().initialize(opt, world, rank)
