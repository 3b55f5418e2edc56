"""Data layer surface of the reference (data/__init__.py:5-69): dataset registry by
`--dataset_mode`, `CreateDataLoader(opt).load_data()` yielding dict batches with the keys the
model's set_input reads ('A', 'B', 'label', 'A_paths', 'B_paths').

Under torch.distributed every rank draws the same global batch order and keeps its contiguous
slice (DataParallel's scatter), see pcgan_amd.hip.parallel.shard_batch.
"""
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


class CustomDatasetDataLoader(object):
    """Iterable over dict batches, capped at --max_dataset_size samples (reference data/__init__.py:42-69)."""

    def name(self):
        return 'CustomDatasetDataLoader'

    def initialize(self, opt):
        self.opt = opt
        self.dataset = create_dataset(opt)
        self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, num_workers=int(opt.nThreads),
                                                      shuffle=not opt.serial_batches)
        return self

    def load_data(self):
        return self

    def __len__(self):
        return min(len(self.dataset), self.opt.max_dataset_size)

    def __iter__(self):
        seen = 0
        for batch in self.dataloader:
            if seen >= self.opt.max_dataset_size:
                return
            seen += self.opt.batchSize
            yield batch


def CreateDataLoader(opt):
    return CustomDatasetDataLoader().initialize(opt)
