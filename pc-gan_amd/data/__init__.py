"""Data layer surface of the reference (data/__init__.py:5-69): dataset registry by
`--dataset_mode`, `CreateDataLoader(opt).load_data()` yielding dict batches with the keys the
model's set_input reads ('A', 'B', 'label', 'A_paths', 'B_paths').

Under torch.distributed every rank draws the same global batch order and keeps its contiguous
slice (DataParallel's scatter), see pcgan_amd.hip.parallel.shard_batch.
"""
import importlib

import torch.utils.data

from .base_dataset import BaseDataset


def find_dataset_using_name(dataset_name):
    module_name = '%s.%s_dataset' % (__name__, dataset_name)
    try:
        lib = importlib.import_module(module_name)
    except ModuleNotFoundError as e:
        if e.name != module_name:
            raise
        raise NotImplementedError('pcgan_amd: dataset mode [%s] is outside the MI355X hot path '
                                  '(available: wsgan_emb, wsgan_cycle)' % dataset_name)
    target = dataset_name.replace('_', '') + 'dataset'
    found = None
    for name, cls in vars(lib).items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseDataset):
            found = cls
    if found is None:
        print('In %s.py, there should be a subclass of BaseDataset with class name that matches %s in lowercase.'
              % (module_name, target))
        exit(0)
    return found


def get_option_setter(dataset_name):
    return find_dataset_using_name(dataset_name).modify_commandline_options


def create_dataset(opt):
    instance = find_dataset_using_name(opt.dataset_mode)()
    instance.initialize(opt)
    print('dataset [%s] was created' % instance.name())
    return instance


class CustomDatasetDataLoader():
    def name(self):
        return 'CustomDatasetDataLoader'

    def initialize(self, opt):
        self.opt = opt
        self.dataset = create_dataset(opt)
        self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize,
                                                      shuffle=not opt.serial_batches,
                                                      num_workers=int(opt.nThreads))

    def load_data(self):
        return self

    def __len__(self):
        return min(len(self.dataset), self.opt.max_dataset_size)

    def __iter__(self):
        for i, batch in enumerate(self.dataloader):
            if i * self.opt.batchSize >= self.opt.max_dataset_size:
                break
            yield batch


def CreateDataLoader(opt):
    loader = CustomDatasetDataLoader()
    loader.initialize(opt)
    return loader
