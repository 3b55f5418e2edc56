"""Child process of tests/test_gpu_ddp.py (not a test module): one rank of a data-parallel step of the PRODUCT model.

    RANK / WORLD_SIZE / MASTER_* in the environment, PCGAN_DIST_BACKEND=gloo (several ranks share the one GPU of the box).
    argv: <out.pt> <lo> <hi> [fp32|bf16]   -- this rank steps samples [lo, hi) of the fixed 4-sample batch, activations stored as given
    (bf16 x ranks = BASELINE configs[2]); PCGAN_DDP_OVERLAP=1 selects the overlapped all-reduce of the generator's gradients

Writes the flat G / D gradient buffers as they are when each optimizer steps (i.e. after the all-reduce), the flat
parameter buffers after the step and the losses."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    out, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dtype = sys.argv[4] if len(sys.argv) > 4 else 'fp32'
    from pcgan_amd.hip import parallel
    import bench
    world, rank, _ = parallel.init_process_group()
    torch.cuda.set_device(0)
    tmp = tempfile.mkdtemp(prefix='pcgan_ddp_%d_' % rank)
    # every rank builds from its own seed: broadcast_parameters must then install rank 0's weights everywhere
    model, opt = bench.build_model(0, hi - lo, 32, tmp, seed=7 + (rank if world > 1 else 0), ngf=8, ndf=8, fine_e=64, n_blocks=2, dtype=dtype)
    grabbed = {}
    # the flat gradient buffers as the update kernel reads them (after the all-reduce), grabbed on whichever stream the update is queued
    # on (the main stream, or the parameter-gradient stream under PCGAN_DDP_GRAD_STREAM=1)
    from pcgan_amd.hip import ops
    tags = {model.optimizer_G.flat.data_ptr(): 'gG', model.optimizer_D.flat.data_ptr(): 'gD'}
    real_adam = ops.adam_step_dev

    def adam(flat, gflat, *a, **k):
        grabbed[tags[flat.data_ptr()]] = gflat.detach().clone()
        return real_adam(flat, gflat, *a, **k)
    ops.adam_step_dev = adam
    b = bench.synthetic_batch(4, 32, 0)
    batch = {k: v[lo:hi] for k, v in b.items()}
    model.set_input(batch)
    model.optimize_parameters()
    torch.cuda.synchronize()
    grabbed['gG'], grabbed['gD'] = grabbed['gG'].cpu(), grabbed['gD'].cpu()
    grabbed['pG'] = model.optimizer_G.flat.detach().clone().cpu()
    grabbed['pD'] = model.optimizer_D.flat.detach().clone().cpu()
    grabbed['losses'] = dict(model.get_current_losses())
    grabbed['distributed'] = parallel.is_distributed()
    grabbed['identity'] = parallel.rank_identity()
    torch.save(grabbed, out)
    if parallel.is_distributed():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
