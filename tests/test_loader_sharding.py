"""CPU, world size 2 over gloo: the loader is sharded at the sampler (pc-gan_amd/data/__init__.py).

Two real processes run CreateDataLoader on an ODD-sized pair list (11 pairs, global batch 4) WITHOUT --seed -- the shared
seed is drawn on rank 0 and broadcast.  Checked: per epoch both ranks walk ONE permutation (their samples are disjoint and
come from the list, each pair at most once), every rank decodes only batchSize / world samples per step, the last partial
global batch is dropped (no ragged shard, no abort), the order changes from epoch to epoch, and rank r holds the r-th
contiguous slice of each global batch (DataParallel's scatter, reference models/networks.py:96-102) -- verified by
rebuilding the global order from the shared seed in the parent."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from PIL import Image

N_PAIRS = 11
GLOBAL_BATCH = 4


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_pairs(tmp):
    rng = np.random.default_rng(3)
    for i in range(N_PAIRS + 1):
        Image.fromarray(rng.integers(0, 256, (24, 24, 3), dtype=np.uint8)).save(os.path.join(tmp, 'img_%d.png' % i))
    with open(os.path.join(tmp, 'pairs.txt'), 'w') as f:
        for i in range(N_PAIRS):
            f.write('img_%d.png img_%d.png %d\n' % (i, i + 1, (0, 2, 1)[i % 3]))


def _opt(tmp, extra=()):
    from pcgan_amd.options.train_options import TrainOptions
    argv = ['train.py', '--dataroot', tmp, '--model', 'wsgan_emb', '--gpu_ids', '-1', '--checkpoints_dir', os.path.join(tmp, 'ck'),
            '--sourcefile_A', os.path.join(tmp, 'pairs.txt'), '--loadSize', '20', '--fineSize', '16', '--nThreads', '1',
            '--batchSize', str(GLOBAL_BATCH)] + list(extra)
    old, sys.argv = sys.argv, argv
    so, sys.stdout = sys.stdout, open(os.devnull, 'w')
    try:
        return TrainOptions().parse()
    finally:
        sys.stdout.close()
        sys.argv, sys.stdout = old, so


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import multiprocessing
    multiprocessing.set_start_method('fork', force=True)     # DataLoader workers fork, as under torchrun (this child was spawned)
    from pcgan_amd.hip import parallel
    from pcgan_amd.data import CreateDataLoader
    parallel.init_process_group('gloo')
    opt = _opt(tmp)
    loader = CreateDataLoader(opt)
    assert loader.world == world and loader.rank == rank
    n_images = len(loader)
    import random
    draw = (random.random(), float(torch.rand(1)), float(np.random.rand()))     # per-sample generators after the loader is up
    epochs = []
    for ep in range(2):
        if rank == 0:          # rank-0-only logging takes len() on its own: must not desynchronise the pair lists
            len(loader), len(loader.dataset)
        batches = []
        for b in loader.load_data():
            assert tuple(b['A'].shape) == (GLOBAL_BATCH // world, 3, 16, 16) and len(b['A_paths']) == GLOBAL_BATCH // world
            batches.append([(os.path.basename(a), os.path.basename(c), int(l)) for a, c, l in zip(b['A_paths'], b['B_paths'], b['label'])])
        epochs.append(batches)
    torch.save({'epochs': epochs, 'seed': loader.sampler.seed, 'n': n_images, 'draw': draw}, os.path.join(tmp, 'rank%d.pt' % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_hold_disjoint_slices_of_one_permutation(tmp_path):
    tmp = str(tmp_path)
    _make_pairs(tmp)
    mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp, 'rank%d.pt' % i)) for i in range(2)]
    assert r[0]['seed'] == r[1]['seed'], 'ranks must share the shuffle seed (broadcast from rank 0)'
    assert r[0]['n'] == r[1]['n'] == N_PAIRS
    # crop / flip / noise draws are per sample: python's and torch's generators differ between ranks (DataParallel draws them
    # independently over the global batch); numpy's stays shared (--no_mixed_label_D draws one label per GLOBAL batch from it)
    assert r[0]['draw'][0] != r[1]['draw'][0] and r[0]['draw'][1] != r[1]['draw'][1]
    assert r[0]['draw'][2] == r[1]['draw'][2]
    lines = {('img_%d.png' % i, 'img_%d.png' % (i + 1), (0, 2, 1)[i % 3]) for i in range(N_PAIRS)}
    orders = []
    for ep in range(2):
        b0, b1 = r[0]['epochs'][ep], r[1]['epochs'][ep]
        assert len(b0) == len(b1) == N_PAIRS // GLOBAL_BATCH, 'the partial last global batch is dropped on every rank'
        seen = []
        for s0, s1 in zip(b0, b1):
            assert len(s0) == len(s1) == GLOBAL_BATCH // 2
            seen += s0 + s1                   # global batch = rank 0's slice followed by rank 1's
        assert len(set(seen)) == len(seen) == (N_PAIRS // GLOBAL_BATCH) * GLOBAL_BATCH, 'a pair was decoded twice in one epoch'
        assert set(seen) <= lines
        orders.append(seen)
    assert orders[0] != orders[1], 'the permutation must change from epoch to epoch'


def test_sampler_slices_are_contiguous_parts_of_the_global_batches():
    from pcgan_amd.data import RankShardedBatchSampler
    world, B, n = 4, 8, 37
    per_rank = []
    for rank in range(world):
        s = RankShardedBatchSampler(n, B, world, rank, True, 1234)
        s.set_epoch(3)
        per_rank.append(list(s))
        assert len(s) == n // B
    ref = RankShardedBatchSampler(n, B, 1, 0, True, 1234)
    ref.set_epoch(3)
    for b, g in enumerate(ref):          # the single-process order, cut into global batches
        assert sum((per_rank[r][b] for r in range(world)), []) == g
    serial = RankShardedBatchSampler(n, B, world, 1, False, 0)
    assert next(iter(serial)) == [2, 3]
    with pytest.raises(AssertionError):
        RankShardedBatchSampler(n, 6, world, 0, True, 0)
