"""reference options/test_options.py (same flags and defaults)."""
from .base_options import BaseOptions

_TEST_FLAGS = [
    ('--ntest', dict(type=int, default=float('inf'))),
    ('--results_dir', dict(type=str, default='./results/')),
    ('--aspect_ratio', dict(type=float, default=1.0)),
    ('--phase', dict(type=str, default='test')),
    ('--which_epoch', dict(type=str, default='latest')),
    ('--how_many', dict(type=int, default=20000)),
    # inception-score / image-generation options of the reference's offline tools (parsed, unused here)
    ('--which_model_IS', dict(type=str, default='inception_v3')),
    ('--batchSize_IS', dict(type=int, default=32)),
    ('--pretrained_model_path_IS', dict()),
    ('--splits', dict(type=int, default=10)),
    ('--result_path', dict(type=str, default='')),
    ('--how_to_sample', dict(type=str, choices=['prior', 'label'], default='prior')),
    ('--sample_label_file', dict(type=str, default='')),
    ('--output_dir', dict(type=str)),
]


class TestOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        for flag, kw in _TEST_FLAGS:
            parser.add_argument(flag, **kw)
        parser.set_defaults(model='test')
        parser.set_defaults(loadSize=parser.get_default('fineSize'))
        self.isTrain = False
        return parser
