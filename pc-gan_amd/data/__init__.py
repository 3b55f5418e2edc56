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


def _collate_keep_raw(samples):
    """default collate, except that decoded images ('*_raw', possibly of different sizes) stay a list"""
    raw = {k: [s[k] for s in samples] for k in samples[0] if k.endswith('_raw')}
    rest = torch.utils.data.default_collate([{k: v for k, v in s.items() if not k.endswith('_raw')} for s in samples])
    rest.update(raw)
    return rest


class CustomDatasetDataLoader(object):
    """Iterable over dict batches, capped at --max_dataset_size samples (reference data/__init__.py:42-69)."""

    def name(self):
        return 'CustomDatasetDataLoader'

    def initialize(self, opt):
        self.opt = opt
        self.dataset = create_dataset(opt)
        self.gpu_transform = None
        if getattr(opt, 'gpu_transform', False) and opt.dataroot != 'synthetic':
            from .gpu_transform import GpuTransform
            self.gpu_transform = GpuTransform(opt, 'cuda:%d' % opt.gpu_ids[0] if opt.gpu_ids else 'cpu')
        self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, num_workers=int(opt.nThreads),
                                                      shuffle=not opt.serial_batches,
                                                      collate_fn=_collate_keep_raw if self.gpu_transform else None)
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
        seen = 0
        for batch in self.dataloader:
            if seen >= self.opt.max_dataset_size:
                return
            seen += self.opt.batchSize
            yield self._finish_on_gpu(batch) if self.gpu_transform else batch


def CreateDataLoader(opt):
    return CustomDatasetDataLoader().initialize(opt)
