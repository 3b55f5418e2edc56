#!/usr/bin/env python
"""Training driver with the reference's command line and cadence (reference train.py:9-61): same options, same
print / save frequencies, the same `time` and `data` fields in the loss line; a plain-text loss log replaces the
visdom / HTML visualiser (those dependencies do not exist here).

    python train.py --dataroot synthetic --model wsgan_emb --which_model_netG resnet_9blocks \
        --which_model_netD n_layers --n_layers_D 3 --batchSize 32 --pretrained_model_path_E E.pth ...
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...      # one process per GPU, RCCL
"""
import math
import os
import time

from pcgan_amd.data import CreateDataLoader
from pcgan_amd.hip import parallel
from pcgan_amd.models import create_model
from pcgan_amd.options.train_options import TrainOptions


class LossLog(object):
    """rank 0 prints the loss line and appends it to <checkpoints_dir>/<name>/loss_log.txt"""

    def __init__(self, opt, rank):
        self.path = os.path.join(opt.checkpoints_dir, opt.name, 'loss_log.txt')
        self.active = rank == 0
        self.write('================ Training Loss (%s) ================' % time.strftime('%c'), echo=False)

    def write(self, line, echo=True):
        if not self.active:
            return
        if echo:
            print(line)
        with open(self.path, 'a') as f:
            f.write(line + '\n')


def train_epoch(epoch, batches, model, opt, log, counters, world, rank):
    t_epoch = time.time()
    t_after_prev = time.time()
    in_epoch = 0
    for batch in batches:
        t_iter = time.time()
        t_data = t_iter - t_after_prev                    # time spent waiting for the loader
        model.set_input(batch)                            # (already this rank's slice of the global batch: data/__init__.py)
        model.optimize_parameters()
        counters['total'] += 1
        in_epoch += 1
        n = counters['total']
        if n % opt.display_freq == 0:
            model.get_current_visuals()                   # runs G on the fixed ratings in train mode, like the reference
        if n % opt.print_freq == 0:
            per_image = (time.time() - t_iter) / opt.batchSize
            fields = ' '.join('%s: %.3f' % kv for kv in model.get_current_losses().items())
            log.write('(epoch: %d, iters: %d, time: %.3f, data: %.3f) %s' % (epoch, in_epoch, per_image, t_data, fields))
        if n % opt.save_latest_freq == 0:
            print('saving the latest model (epoch %d, total_iter %d)' % (epoch, n))
            model.save_networks('latest')
        t_after_prev = time.time()
    if epoch % opt.save_epoch_freq == 0:
        print('saving the model at the end of epoch %d, iters %d' % (epoch, counters['total']))
        model.save_networks('latest')
        model.save_networks(epoch)
    print('End of epoch %d / %d \t Time Taken: %d sec' % (epoch, opt.niter + opt.niter_decay, time.time() - t_epoch))


def main():
    world, rank, _ = parallel.init_process_group()
    opt = TrainOptions().parse()
    loader = CreateDataLoader(opt)
    batches = loader.load_data()
    print('#training images = %d' % len(loader))
    opt.num_iter_per_epoch = math.ceil(len(loader) / opt.batchSize)
    model = create_model(opt)
    model.setup(opt)
    log = LossLog(opt, rank)
    counters = {'total': 0}
    for epoch in range(opt.epoch_count, opt.niter + opt.niter_decay + 1):
        train_epoch(epoch, batches, model, opt, log, counters, world, rank)
        model.update_learning_rate()


if __name__ == '__main__':
    main()
