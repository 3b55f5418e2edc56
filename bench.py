#!/usr/bin/env python
"""bench.py -- images/sec of one full wsgan_emb optimize_parameters() (G+D adversarial step incl.
the Elo-encoder and AlexNet identity terms) on synthetic 128x128 batches.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 9-block ResnetGenerator + 3-layer PatchGAN (BatchNorm, sigmoid)
+ ResNet-18 Elo encoder @224 + AlexNet IP @224, default loss weights, fp32, batch 32 PER GPU
(weak scaling: global batch = 32 x N), synthetic U[-1,1) images, labels in {0,2}, seeded-random
"pretrained" E/IP weights (no checkpoints ship offline).

One JSON line on rank 0: value = whole-job images/s; `roofline` = the three kernels of the 256->256 3x3 residual
convolution (forward, data gradient, weight gradient INCLUDING its reduce launch; 36 launches each per step;
fp32 contraction as two scaled fp16 pieces / three products on the f16 matrix pipe by default): every launch inside the timed
region is bracketed by HIP events on its launch stream inside the library (pcgan_timer_*), the slowest of the three is the
dominant kernel the top-level fields describe.  The step runs on several streams (parameter gradients, branches of backward_G, the
next step's encoder passes and first generator pass beside the previous step's backward_D), so an in-region duration is that of a
kernel SHARING the GPU; `ms_per_launch_alone` / `frac_alone` are the same launches with those streams off (3 extra steps after the
region).  `cpu_baseline` = the oracle's PyTorch-CPU step on this host's cores, bounded sample.
`python bench.py --dtype bf16` is the same line for BASELINE configs[2] per GPU (bf16 activations).
"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMG_FULL = 183.95      # 2 G + 4 D + 3 E + 2 IP fwd and all backward terms @128^2
FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
BF16_SPLIT_PRODUCTS = 6          # bf16 piece products that stand for one fp32 product (csrc/bf16x6_conv.hip)
F16_MFMA_PEAK_TFLOPS = 2500.0    # dense f16 MFMA (v_mfma_f32_32x32x16_f16): same rate as bf16
F16_SPLIT_PRODUCTS = 3           # fp16 piece products that stand for one fp32 product (the default route, "fp16 route" in bf16x6_conv.hip)
PER_GPU_BATCH = 32
SIZE = 128
TRAFFIC_FILE = os.path.join(ROOT, 'profiles', 'r04_residual_kernel_traffic.json')   # written from the --pmc passes of this round
N_BATCHES = 4                    # distinct synthetic batches cycled through (pinned host memory; uploaded INSIDE the timed region)


def lib_option(key):
    """a routing option of the library as this run had it (pcgan_get_option)"""
    from pcgan_amd.hip import lib as _lib
    return _lib.get_option(key)


def measured_traffic(route):
    """HBM bytes per launch of the three residual-convolution kernels from this round's separate rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE cannot share a pass), committed under profiles/: {'fwd' | 'dgrad' | 'wgrad': {...}}; None when
    the file is absent or was counted on another route."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        if t.get('route') != route:
            return None
        out = {}
        for k in ('fwd', 'dgrad', 'wgrad'):
            e = t[k]
            out[k] = {'bytes_per_launch': int(e['fetch_bytes'] + e['write_bytes']), 'fetch_bytes': int(e['fetch_bytes']),
                      'write_bytes': int(e['write_bytes']), 'algorithmic_bytes': int(e['algorithmic_bytes']), 'kernels': e.get('kernels')}
        out['source'] = t['source']
        out['mfma_busy'] = t.get('mfma_busy')      # {'fwd' | 'dgrad' | 'wgrad': SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x XCDs x 4 SIMDs ...)}: see the file
        return out
    except (OSError, KeyError, ValueError, TypeError):
        return None


def build_model(device_index, batch, size, tmpdir, seed=0, ngf=64, ndf=64, fine_e=224, n_blocks=9, dtype='fp32'):
    from pcgan_amd.options.train_options import TrainOptions
    from pcgan_amd.models import create_model, networks
    torch.manual_seed(seed)
    # seeded-random stand-ins for the pretrained Elo encoder / AlexNet (none available offline)
    e = networks.define_E('resnet18', 3, 'normal', 'avg', [32, 1], 1, 0.7)
    ip = networks.define_IP('alexnet', 3)
    e_path, ip_path = os.path.join(tmpdir, 'E.pth'), os.path.join(tmpdir, 'IP.pth')
    torch.save(e.state_dict(), e_path)
    torch.save(ip.state_dict(), ip_path)
    argv = ['bench.py', '--dataroot', 'synthetic', '--model', 'wsgan_emb', '--name', 'bench',
            '--checkpoints_dir', tmpdir, '--gpu_ids', str(device_index),
            '--which_model_netG', 'resnet_%dblocks' % n_blocks, '--which_model_netD', 'n_layers', '--n_layers_D', '3',
            '--ngf', str(ngf), '--ndf', str(ndf), '--fineSize', str(size), '--loadSize', str(size),
            '--fineSize_E', str(fine_e), '--fineSize_IP', str(fine_e), '--batchSize', str(batch),
            '--pretrained_model_path_E', e_path, '--pretrained_model_path_IP', ip_path, '--display_id', '-1',
            '--dtype', dtype]
    old, sys.argv = sys.argv, argv
    stdout, sys.stdout = sys.stdout, open(os.devnull, 'w')
    try:
        opt = TrainOptions().parse()
        model = create_model(opt)
        model.setup(opt)
    finally:
        sys.stdout.close()
        sys.argv, sys.stdout = old, stdout
    return model, opt


def synthetic_batch(batch, size, rank, it=0):
    g = torch.Generator().manual_seed(1234 + rank + 1000 * it)
    A = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    B = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    label = torch.randint(0, 2, (batch,), generator=g) * 2
    return {'A': A, 'B': B, 'label': label, 'A_paths': [''] * batch, 'B_paths': [''] * batch}


RES_CONV_KEY = (PER_GPU_BATCH, 256, 32, 32, 256, 3, 3, 1, 1, 1)   # the residual-block convolution: 36 forward launches per step
RES_CONV_FLOP = 2.0 * PER_GPU_BATCH * 32 * 32 * 256 * 256 * 9


def host_cpu_budget():
    """(usable CPUs, why): min of the scheduler affinity and the cgroup CPU quota -- a GPU box reports all 128 hardware threads of its
    host while the container may run on a 16-CPU share; a thread pool sized to the former is oversubscribed 8x (round 3's baseline)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    why = 'affinity %d' % n
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                q = max(1, int(float(quota) / float(period) + 0.5))
                if q < n:
                    n, why = q, why + ', cgroup quota %d' % q
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and per > 0 and q // per < n:
            n, why = max(1, q // per), why + ', cgroup v1 quota %d' % (q // per)
    except (OSError, ValueError):
        pass
    return n, why


def _progress(msg):
    print('[bench %6.1f s] %s' % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_baseline(batch=8, sweep_batch=4, final_steps=3, sweep_budget_s=45.0):
    """The oracle's CPU step (reference-equivalent PyTorch-CPU path) on a bounded sample, at the BEST thread count this box offers:
    torch.set_num_threads is swept over {usable CPUs, 16, 32, 8, 64, all hardware threads} capped at 4 x the usable CPUs -- most promising first, 1 untimed + 1 timed
    step each at batch `sweep_batch`, stopped when `sweep_budget_s` is used up -- then 1 untimed + `final_steps` timed steps at the
    winner at batch `batch`; value = median of those.  A stated baseline must be the best the host's cores can do: round 3 ran on
    torch's default of 128 threads (the box's hardware threads; the container's CPU share is smaller) and got half of what 8 give."""
    from oracle import networks_ref as N
    from oracle import step_ref as S
    from oracle import weights as W
    default_threads = torch.get_num_threads()
    usable, why = host_cpu_budget()
    hw = os.cpu_count() or usable
    cand = []
    for t in (usable, 16, 32, 8, 64, hw):
        # never more than 4 x the usable CPUs: a 128-thread OpenMP pool on a 16-CPU share does not merely run slowly, its spin-waiting
        # barriers stall for minutes (round 4: the sweep's 128-thread setting hung the bench until the box's silence watchdog killed it)
        if 1 <= t <= min(hw, 4 * usable) and t not in cand:
            cand.append(t)
    G = N.ResnetGeneratorRef(3, 3, 1, 64, 'instance', 9)
    D = N.NLayerDiscriminatorRef(3, 1, 64, 3, 'batch', True)
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
    IP = N.AlexNetFeatureRef(3, 'None')
    for i, net in enumerate((G, D, E, IP)):
        net.load_state_dict(W.fill_state_dict(net.state_dict(), 50 + i))
    m = S.WSGANEmbStepRef(G, D, E, IP)

    def feed(n):
        b = synthetic_batch(n, SIZE, 0)
        m.set_input(b['A'], b['B'], [int(v) for v in b['label']])

    def one():
        t0 = time.perf_counter()
        m.optimize_parameters()
        return time.perf_counter() - t0
    t_begin = time.perf_counter()
    sweep = {}
    try:
        feed(sweep_batch)
        for t in cand:
            torch.set_num_threads(t)
            one()                              # untimed: oneDNN primitives (first setting), pool resize, per-thread scratch
            sweep[t] = one()
            _progress('cpu_baseline sweep: %d threads %.2f img/s (batch %d)' % (t, sweep_batch / sweep[t], sweep_batch))
            if time.perf_counter() - t_begin > sweep_budget_s:      # bounded: the most promising settings came first
                break
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        feed(batch)
        one()
        times = sorted(one() for _ in range(final_steps))
    finally:
        torch.set_num_threads(default_threads)
    med = times[len(times) // 2]
    return {'value': round(batch / med, 3), 'unit': 'images/sec', 'cores': best, 'kind': 'port',
            'best': round(batch / times[0], 3), 'worst': round(batch / times[-1], 3),
            'threads_swept': {str(t): round(sweep_batch / v, 3) for t, v in sweep.items()},
            'host_cpus': {'hardware_threads': hw, 'usable': usable, 'basis': why, 'torch_default_threads': default_threads},
            'sample': '%d timed steps (1 untimed) of the oracle CPU step at batch %d, 128x128, same nets/flags, at the best of %d thread '
                      'counts (sweep: 1 untimed + 1 timed step each at batch %d, most promising first, %.0f s budget); value = median '
                      'step at %d threads; %.0f s of CPU work in all' % (
                          len(times), batch, len(sweep), sweep_batch, sweep_budget_s, best, time.perf_counter() - t_begin)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                    help='activation storage type; the headline (BASELINE configs[1]) is fp32, bf16 is BASELINE configs[2] per GPU')
    ap.add_argument('--resident-inputs', action='store_true',
                    help='A/B only: keep the synthetic batches in HBM (no upload inside the timed region); the headline uploads them')
    ap.add_argument('--no-experiment', action='store_true',
                    help='skip the extra (not headline) run with the residual convolutions routed back to the fp32 MFMA kernels')
    args = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: native libraries write to file descriptor 1 behind Python's back (RCCL prints
    # a five-line version banner on rank 0's stdout when its communicator comes up -- seen on the first RCCL run, round 4), so the
    # descriptor itself points at stderr until the line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from pcgan_amd.hip import parallel
    world, rank, local = parallel.init_process_group()
    assert world == args.gpus, 'launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
    dev_index = local % torch.cuda.device_count()   # (one GPU per rank on a real node; a 1-GPU box can only rehearse)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)

    tmpdir = tempfile.mkdtemp(prefix='pcgan_bench_')
    with contextlib.redirect_stdout(sys.stderr):      # the reference's 'initialize network with ...' notices: stdout carries the JSON line only
        model, opt = build_model(dev_index, PER_GPU_BATCH, SIZE, tmpdir, dtype=args.dtype)
    # N_BATCHES distinct batches in PINNED host memory: set_input uploads them INSIDE the timed region (BaseModel.to_act: async copy on
    # the upload stream, 2 x 6.3 MB + the labels per step -- SURVEY 8(a2) is part of the step; round 3 parked two batches in HBM)
    batches = [synthetic_batch(PER_GPU_BATCH, SIZE, rank, it) for it in range(N_BATCHES)]
    batches = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]
    if args.resident_inputs:      # A/B only (round 3's form): batches parked in HBM before the region, declared ready by their producer
        from pcgan_amd.hip import ops as _ops
        batches = [{k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]
        torch.cuda.synchronize()
        for b in batches:
            _ops.mark_ready(b['A'])
            _ops.mark_ready(b['B'])

    def step(i):
        model.set_input(batches[i % N_BATCHES])
        model.optimize_parameters()

    from pcgan_amd.hip import ops
    _progress('model built, warm-up')
    for i in range(args.warmup):
        step(i)
    dist_on = parallel.is_distributed()     # more than one rank, or the one-rank RCCL rehearsal (PCGAN_FORCE_COLLECTIVES=1)
    _progress('timed region')
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream around every launch of the three residual-convolution kernels inside the timed region
    # (recorded inside the library, pcgan_timer_*: whichever host path -- per-op call or composite -- issues the launch)
    ops.timer_enable(36 * args.steps + 8)
    if dist_on:
        parallel.COMM_TIMER = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_issued = time.perf_counter() - t0     # host side done issuing (the GPU may still be running): launch-bound if ~ dt
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    dt = dt_rank = time.perf_counter() - t0
    per_rank = None
    if dist_on:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank view (diagnostics of the first multi-GPU runs): own wall time and the time inside the gradient all-reduces
        # (collective + waiting for the slowest rank), HIP events on the launch stream
        comm, parallel.COMM_TIMER = parallel.COMM_TIMER, None
        comm_ms = sum(a.elapsed_time(b) for a, b in comm) / args.steps
        mine = {'rank': rank, 'ms_per_step': round(dt_rank / args.steps * 1e3, 3), 'allreduce_ms_per_step': round(comm_ms, 3),
                'allreduces_per_step': round(len(comm) / args.steps, 2), 'host_issue_ms_per_step': round(t_issued / args.steps * 1e3, 3)}
        mine.update(parallel.rank_identity(device))       # device uuid / PCI bus / RCCL version: the record proves N distinct GPUs
        # replicas must be bit-identical after the timed steps (same start, same averaged gradients, same Adam arithmetic): compared
        # once, outside the timed region; a divergence puts every rank's hash on stderr and `false` in the line
        model.sync_parameter_updates()
        try:
            mine['replicas_bit_equal'] = bool(parallel.ddp_check(model.optimizer_G, 'bench_G', every=1, exit_on_divergence=False)
                                              and parallel.ddp_check(model.optimizer_D, 'bench_D', every=1, exit_on_divergence=False))
        except parallel.ReplicaDivergenceError:       # (raised on every rank alike; each has printed its hash) the line still goes out, flagged
            mine['replicas_bit_equal'] = False
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)
        uuids = {(r['host'], r.get('device_uuid'), r.get('pci_bus_id')) for r in per_rank}
        if torch.distributed.get_backend() == 'nccl':
            assert len(uuids) == world, 'bench.py --gpus %d: ranks share a GPU (%d distinct devices): %r' % (world, len(uuids), sorted(uuids))
    losses = model.get_current_losses()
    assert all(v == v and abs(v) < 1e6 for v in losses.values()), 'non-finite loss: %r' % losses
    _progress('timed region done: %.2f ms/step' % (dt / args.steps * 1e3))

    timed = {k: ops.timer_read(k) for k in ('res_fwd', 'res_dgrad', 'res_wgrad', 'res_wgrad_main')}
    # the same kernels running ALONE: three more steps with the parameter-gradient and branch streams off (in the step the data
    # gradient on the main stream and the weight gradient on the side stream share the GPU, so each in-region duration above holds
    # the other kernel's work too)
    saved_streams = (ops.SIDE_STREAM, ops.BRANCH_STREAMS)
    ops.SIDE_STREAM = ops.BRANCH_STREAMS = False
    try:
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
    finally:
        ops.SIDE_STREAM, ops.BRANCH_STREAMS = saved_streams
    alone = {k: ops.timer_read(k) for k in ('res_fwd', 'res_dgrad', 'res_wgrad', 'res_wgrad_main')}
    ops.timer_enable(0)
    # host work per step: time to ISSUE one step starting from an idle GPU.  (Inside the timed region the HIP queue back-pressures
    # the host about one step ahead of the GPU, so the in-region issue time tracks the GPU time from below, not the host's work.)
    idle_issue = []
    for i in range(5):
        torch.cuda.synchronize()
        th = time.perf_counter()
        step(i)
        idle_issue.append(time.perf_counter() - th)
    torch.cuda.synchronize()
    idle_issue.sort()
    host_ms = idle_issue[len(idle_issue) // 2] * 1e3
    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    value = PER_GPU_BATCH * world * args.steps / dt
    assert all(len(timed[k]) > 0 for k in ('res_fwd', 'res_dgrad', 'res_wgrad')), 'the residual convolutions were not launched inside the timed region'
    kern = {k: {'ms_per_launch': sum(v) / len(v), 'launches_timed': len(v)} for k, v in timed.items() if v}
    conv_flop = RES_CONV_FLOP
    dominant = max(('res_fwd', 'res_dgrad', 'res_wgrad'), key=lambda k: kern[k]['ms_per_launch'])
    conv_ms, conv_launches = kern[dominant]['ms_per_launch'], kern[dominant]['launches_timed']
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12
    split = ops.BF16X6
    hsplit = split and ops.HSPLIT
    bf16 = args.dtype == 'bf16'
    shape = '256->256 3x3 reflect @32x32, bs32'
    if bf16:
        # bf16 storage, one bf16 product per term: 16x the fp32 MFMA rate makes the kernels LDS / issue bound -> judged against HBM
        route, peak, basis = 'bf16', None, None
        names = {'res_fwd': 'bsplit_halo_kernel<BH_FWD,PK_BF16,bf16,32> (bf16 activations and weights, v_mfma_f32_32x32x16_bf16, fp32 accumulate)',
                 'res_dgrad': 'bsplit_halo_kernel<BH_DGRAD,PK_BF16,bf16,32>',
                 'res_wgrad': ('rowring_wgrad_bf16_kernel + wgd_reduce_kernel (row ring in LDS, one bf16 product per tap, loads four stages ahead)'
                               if lib_option('wgrad_rowring') in (1, 2) else 'bsplit_pad_wave_kernel + hsplit_wgrad_kernel<256,1,bf16,2> + bsplit_wgrad_reduce_kernel')}
    elif hsplit:
        # the fp32 contraction runs as 3 fp16 piece products per term (two scaled fp16 pieces per operand) on the f16 matrix pipe:
        # the MFMA roofline in ALGORITHMIC (fp32) FLOP is the dense f16 peak / 3
        route, peak = 'f16x2', F16_MFMA_PEAK_TFLOPS / F16_SPLIT_PRODUCTS
        basis = 'dense f16 MFMA %.0f TFLOP/s / %d piece products per fp32 product' % (F16_MFMA_PEAK_TFLOPS, F16_SPLIT_PRODUCTS)
        names = {'res_fwd': 'bsplit_halo_kernel<BH_FWD,PK_F16X2,float,32> (two scaled fp16 pieces per operand, 3 x v_mfma_f32_32x32x16_f16 per '
                            'K=16 step, 256 x 128 tile, input window split once per 16-channel chunk)',
                 'res_dgrad': 'bsplit_halo_kernel<BH_DGRAD,PK_F16X2,float,32> (the same window kernel on dy with its sum rows / columns)',
                 'res_wgrad': ('rowring_wgrad_kernel + wgd_reduce_kernel (column tile = 32 input channels x all nine taps walking down a 16-pixel strip: '
                               'padded rows in a ring of four LDS slots, each element of x loaded and split once per strip; dy straight from memory; '
                               '32 splits of the pixel reduction, fixed-order reduce into the gradient buffer)') if lib_option('wgrad_rowring') else
                              ('hsplit_wgrad_kernel<256,1,float,1> + bsplit_wgrad_reduce_kernel (x gathered with the reflection applied in '
                               'the loads, 14 splits of the pixel reduction, fixed-order reduce into the gradient buffer)')}
    elif split:
        route, peak = 'bf16x3', BF16_MFMA_PEAK_TFLOPS / BF16_SPLIT_PRODUCTS
        basis = 'dense bf16 MFMA %.0f TFLOP/s / %d piece products per fp32 product' % (BF16_MFMA_PEAK_TFLOPS, BF16_SPLIT_PRODUCTS)
        names = {'res_fwd': 'bsplit_halo_kernel<BH_FWD,PK_BF16X3,float,32> (exact 3-piece bf16 split, 6 x v_mfma_f32_32x32x16_bf16 per K=16 step)',
                 'res_dgrad': 'bsplit_halo_kernel<BH_DGRAD,PK_BF16X3,float,32>',
                 'res_wgrad': 'bsplit_pad_reflect + bsplit_pack_dy + bsplit_conv_fwd_kernel<BS_WGRAD> + reduce'}
    else:
        route, peak, basis = 'fp32_mfma', FP32_MFMA_PEAK_TFLOPS, 'fp32 MFMA v_mfma_f32_32x32x2_f32'
        names = {'res_fwd': 'igemm2_kernel<1,128,128,16> (FWD_REFLECT, 128x128 tile, 16-channel K stages)',
                 'res_dgrad': 'igemm2_kernel (reflect data gradient, three row classes)', 'res_wgrad': 'wgrad2_kernel + wgrad_reduce_kernel'}
    traffic = measured_traffic(route)
    tkey = {'res_fwd': 'fwd', 'res_dgrad': 'dgrad', 'res_wgrad': 'wgrad'}
    alg_bytes_bf16 = 2 * PER_GPU_BATCH * 256 * 32 * 32 * 2 + 256 * 256 * 9 * 2     # x + y as bf16, bf16 weights

    def kernel_entry(k):
        ms = kern[k]['ms_per_launch']
        tf = conv_flop / (ms * 1e-3) / 1e12
        e = {'kernel': names[k] + ' ' + shape, 'ms_per_launch': round(ms, 4), 'launches_timed': kern[k]['launches_timed'],
             'achieved_tflops': round(tf, 2)}
        if peak is not None:
            e['frac'] = round(tf / peak, 4)
        if alone.get(k):
            ms_a = sum(alone[k]) / len(alone[k])
            e['ms_per_launch_alone'] = round(ms_a, 4)
            if peak is not None:
                e['frac_alone'] = round(conv_flop / (ms_a * 1e-3) / 1e12 / peak, 4)
        if k == 'res_wgrad' and 'res_wgrad_main' in kern:
            e['ms_main_kernel_only'] = round(kern['res_wgrad_main']['ms_per_launch'], 4)
            if alone.get('res_wgrad_main'):
                e['ms_main_kernel_only_alone'] = round(sum(alone['res_wgrad_main']) / len(alone['res_wgrad_main']), 4)
        t = traffic.get(tkey[k]) if traffic else None
        e['traffic'] = t['bytes_per_launch'] if t else None
        if t:
            e['traffic_detail'] = t
        # share of the kernel's active cycles its matrix pipe is busy (rocprofv3 --pmc, from the same committed file as `traffic`)
        busy = traffic.get('mfma_busy') if traffic else None
        e['mfma_busy'] = busy.get(tkey[k]) if busy else None
        return e
    per_kernel = {k: kernel_entry(k) for k in ('res_fwd', 'res_dgrad', 'res_wgrad')}
    dom = per_kernel[dominant]
    if bf16:
        roof = {'bound': 'hbm', 'kernel': dom['kernel'], 'dominant': dominant, 'achieved': round(alg_bytes_bf16 / (conv_ms * 1e-3) / 1e9, 1),
                'peak': 8000.0, 'unit': 'GB/s', 'frac': round(alg_bytes_bf16 / (conv_ms * 1e-3) / 1e9 / 8000.0, 4),
                'ms_per_launch': round(conv_ms, 4), 'launches_timed': conv_launches, 'algorithmic_bytes_per_launch': alg_bytes_bf16,
                'flop_per_launch': conv_flop, 'mfma_tflops': round(achieved, 1), 'traffic': dom['traffic'], 'kernels': per_kernel}
    else:
        roof = {'bound': 'mfma', 'kernel': dom['kernel'], 'dominant': dominant,
                'dominant_rule': 'the slowest of the three residual-convolution kernels (36 launches each per step) by its launch duration '
                                 'inside the timed region; all three under "kernels".  In the region the data gradient (main stream) and '
                                 'the weight gradient (parameter-gradient stream) run CONCURRENTLY and share the GPU: "ms_per_launch_alone" / '
                                 '"frac_alone" are the same launches with those streams off (3 extra steps after the region)',
                'frac_alone': dom.get('frac_alone'), 'ms_per_launch_alone': dom.get('ms_per_launch_alone'),
                'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                'ms_per_launch': round(conv_ms, 4), 'launches_timed': conv_launches, 'flop_per_launch': conv_flop, 'peak_basis': basis,
                'clock_note': 'peak quoted at the 2.4 GHz boost clock; in steady state the fp32 step is held at ~2.0 GHz / ~1255 W by the '
                              'chip\'s power management (profiles/r04_power.txt, scripts/power_sample.sh): the same bound at that clock is 17 % lower',
                # memory-side bytes per launch from this round's separate rocprofv3 --pmc passes (profiles/README.md); null when no
                # counter file of this round is committed
                'traffic': dom['traffic'], 'traffic_detail': dom.get('traffic_detail'), 'mfma_busy': dom.get('mfma_busy'),
                'kernels': per_kernel}
    step_tflops = GFLOP_PER_IMG_FULL * 1e9 * value / world / 1e12        # algorithmic FLOP of the whole step per GPU and second
    out = {
        'metric': 'images/sec (G+D step) 128x128 bs32 per GPU', 'value': round(value, 3), 'unit': 'images/sec',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16' if bf16 else 'f32',
        # what the matrix pipe actually multiplies (dtype above = the storage / interface type of every tensor)
        'arithmetic': ('bf16 x bf16 products, fp32 accumulate' if bf16 else
                       '2 scaled fp16 pieces per fp32 operand x 3 piece products (hi*hi, hi*lo, lo*hi), fp32 accumulate: ~22-bit significand '
                       '(error vs float64 at the fp32 MFMA level, tests/test_gpu_bf16x6.py)' if hsplit else
                       '3 bf16 pieces per fp32 operand x 6 piece products, fp32 accumulate' if split else 'fp32 MFMA'),
        'data': 'synthetic', 'input': ('%d distinct batches RESIDENT in HBM (A/B run, not the headline form)' % N_BATCHES) if args.resident_inputs else
        '%d distinct pinned-host batches, uploaded inside the timed region (set_input)' % N_BATCHES,
        'config': {'workload': 'wsgan_emb UTKFace-shaped 128x128 bs32/GPU %s: 9-block ResnetGenerator + 3-layer '
                               'PatchGAN + ResNet-18 Elo encoder@224 + AlexNet IP@224, full optimize_parameters()' % (
                                   'bf16 activations (fp32 master weights, accumulation, statistics, losses, Adam)' if bf16 else 'fp32'),
                   'global_batch': PER_GPU_BATCH * world, 'parallelism': 'dp%d' % world},
        # host work: median time to issue one step from an idle GPU (5 steps after the region); in the region the launch queue's
        # back-pressure makes the issue time follow the GPU time (second number)
        'host_issue_ms_per_step': round(host_ms, 3),
        'host_issue_in_region_ms_per_step': round(t_issued / args.steps * 1e3, 3),
        # whole-step algorithmic FLOP rate (183.95 GFLOP per image) against the bound of the route the convolutions run
        # on -- THE step-level roofline fraction -- and, for orientation only, against the fp32 MFMA peak (a pipe the default route
        # does not use: that ratio is not a roofline fraction and may exceed 1)
        'step_algorithmic_tflops': round(step_tflops, 2),
        'step_frac_of_route_bound': round(step_tflops / peak, 4) if peak else None,
        'step_flop_over_fp32_mfma_peak_not_a_roofline_fraction': round(step_tflops / FP32_MFMA_PEAK_TFLOPS, 4),
        'roofline': roof,
        'losses': {k: round(v, 5) for k, v in losses.items()},
    }
    if per_rank is not None:
        out['per_rank'] = per_rank
    if world == 1 and not args.no_experiment and ops.BF16X6 and not bf16:
        # NOT the headline: the same K steps once more with the 108 residual-convolution launches on the two other routes -- the
        # exact three-piece bf16 split (PCGAN_SPLIT=bf16, the default before the fp16 route) and the fp32 MFMA implicit GEMM
        # (PCGAN_BF16X6=0, round 1's default) -- the A/B behind the default
        def rerun(what, switch):
            _progress('re-timing on the route ' + switch)
            for i in range(args.warmup):
                step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            l2 = model.get_current_losses()
            assert all(v == v and abs(v) < 1e6 for v in l2.values()), 'non-finite loss (%s): %r' % (switch, l2)
            return {'value': round(PER_GPU_BATCH * args.steps / dt2, 3), 'unit': 'images/sec', 'ms_per_step': round(dt2 / args.steps * 1e3, 3),
                    'default': False, 'switch': switch, 'what': what}
        saved = (ops.BF16X6, ops.HSPLIT)
        try:
            if ops.HSPLIT:
                ops.HSPLIT = False
                out['route_bf16x6'] = rerun('the residual convolutions on the exact three-piece bf16 split, 6 products', 'PCGAN_SPLIT=bf16')
            ops.BF16X6 = False
            out['route_fp32_mfma'] = rerun('the residual convolutions on v_mfma_f32_32x32x2_f32 (round-1 default)', 'PCGAN_BF16X6=0')
        finally:
            ops.BF16X6, ops.HSPLIT = saved
    if world == 1 and not args.no_cpu_baseline:
        _progress('cpu baseline')
        out['cpu_baseline'] = cpu_baseline()
    _progress('done')
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + '\n').encode())
    os.close(json_fd)


if __name__ == '__main__':
    try:
        main()
    finally:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
