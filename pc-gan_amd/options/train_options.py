"""reference options/train_options.py:5-29 (same flags and defaults)."""
from .base_options import BaseOptions

_TRAIN_FLAGS = [
    ('--display_freq', dict(type=int, default=50)),
    ('--display_ncols', dict(type=int, default=4)),
    ('--update_html_freq', dict(type=int, default=1000)),
    ('--save_all_images', dict(action='store_true')),
    ('--print_freq', dict(type=int, default=50)),
    ('--save_latest_freq', dict(type=int, default=1000)),
    ('--save_epoch_freq', dict(type=int, default=5)),
    ('--continue_train', dict(action='store_true')),
    ('--epoch_count', dict(type=int, default=1)),
    ('--phase', dict(type=str, default='train')),
    ('--which_epoch', dict(type=str, default='latest')),
    ('--niter', dict(type=int, default=50)),
    ('--niter_decay', dict(type=int, default=50)),
    ('--beta1', dict(type=float, default=0.5)),
    ('--lr', dict(type=float, default=0.0002)),
    ('--no_lsgan', dict(action='store_true')),
    ('--pool_size', dict(type=int, default=50)),
    ('--no_html', dict(action='store_true')),
    ('--lr_policy', dict(type=str, default='lambda')),
    ('--lr_decay_iters', dict(type=int, default=50)),
]


class TrainOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        for flag, kw in _TRAIN_FLAGS:
            parser.add_argument(flag, **kw)
        self.isTrain = True
        return parser
