#!/usr/bin/env python
"""Training driver with the reference's command line (train.py:9-61): parse options, build the
loader and the model, run the epoch/iteration loop with the same print/save cadence and the same
`time` / `data` log fields, plain-text loss log instead of visdom/HTML (those deps are absent).

    python train.py --dataroot synthetic --model wsgan_emb --which_model_netG resnet_9blocks \
        --which_model_netD n_layers --n_layers_D 3 --batchSize 32 --pretrained_model_path_E E.pth ...
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...      # one process per GPU, RCCL
"""
import math
import os
import time

import torch

from pcgan_amd.options.train_options import TrainOptions
from pcgan_amd.data import CreateDataLoader
from pcgan_amd.models import create_model
from pcgan_amd.hip import parallel


def shard(data, world, rank):
    """this rank's contiguous slice of the global batch (DataParallel's scatter)"""
    if world == 1:
        return data
    out = {}
    for k, v in data.items():
        n = v.shape[0] if isinstance(v, torch.Tensor) else len(v)
        per = n // world
        out[k] = v[rank * per:(rank + 1) * per]
    return out


if __name__ == '__main__':
    world, rank, local = parallel.init_process_group()
    opt = TrainOptions().parse()
    data_loader = CreateDataLoader(opt)
    dataset = data_loader.load_data()
    dataset_size = len(data_loader)
    print('#training images = %d' % dataset_size)
    total_iter = 0
    num_iter_per_epoch = math.ceil(dataset_size / opt.batchSize)
    opt.num_iter_per_epoch = num_iter_per_epoch
    model = create_model(opt)
    model.setup(opt)
    log_path = os.path.join(opt.checkpoints_dir, opt.name, 'loss_log.txt')
    if rank == 0:
        with open(log_path, 'a') as f:
            f.write('================ Training Loss (%s) ================\n' % time.strftime('%c'))

    for epoch in range(opt.epoch_count, opt.niter + opt.niter_decay + 1):
        epoch_start_time = time.time()
        iter_data_time = time.time()
        epoch_iter = 0
        for i, data in enumerate(dataset):
            iter_start_time = time.time()
            if total_iter % opt.print_freq == 0:
                t_data = iter_start_time - iter_data_time
            model.set_input(shard(data, world, rank))
            model.optimize_parameters()
            total_iter += 1
            epoch_iter += 1
            if total_iter % opt.display_freq == 0:
                model.get_current_visuals()      # runs G on the fixed ratings like the reference (train mode)
            if total_iter % opt.print_freq == 0:
                losses = model.get_current_losses()
                t = (time.time() - iter_start_time) / opt.batchSize
                msg = '(epoch: %d, iters: %d, time: %.3f, data: %.3f) ' % (epoch, epoch_iter, t, t_data)
                msg += ' '.join('%s: %.3f' % kv for kv in losses.items())
                if rank == 0:
                    print(msg)
                    with open(log_path, 'a') as f:
                        f.write(msg + '\n')
            if total_iter % opt.save_latest_freq == 0:
                print('saving the latest model (epoch %d, total_iter %d)' % (epoch, total_iter))
                model.save_networks('latest')
            iter_data_time = time.time()
        if epoch % opt.save_epoch_freq == 0:
            print('saving the model at the end of epoch %d, iters %d' % (epoch, total_iter))
            model.save_networks('latest')
            model.save_networks(epoch)
        print('End of epoch %d / %d \t Time Taken: %d sec' % (epoch, opt.niter + opt.niter_decay,
                                                              time.time() - epoch_start_time))
        model.update_learning_rate()
