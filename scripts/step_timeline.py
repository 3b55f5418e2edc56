"""Steady-state schedule of one optimize_parameters() from a handful of HIP events (no profiler: the host keeps its pace): where each
phase of the step begins and ends on the main stream, when the generator stream finishes its two forward passes, when the
parameter-gradient stream drains -- averaged over steps, in ms from the step's first event."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pcgan_amd.hip import ops
dev = torch.device('cuda:0')
model, opt = bench.build_model(0, 32, 128, tempfile.mkdtemp())
batches = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in bench.synthetic_batch(32, 128, 0, i).items()} for i in range(4)]
marks = []


def rec(name, stream=None):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(stream if stream is not None else torch.cuda.current_stream())
    marks.append((name, ev))


def wrap(obj, attr, before=None, after=None):
    f = getattr(obj, attr)

    def g(*a, **k):
        if before:
            rec(before)
        r = f(*a, **k)
        if after:
            rec(after)
        return r
    setattr(obj, attr, g)


wrap(model, 'forward', 'forward: begin (main)', 'forward: issued, main has waited for fake_B')
wrap(model, 'backward_G', 'backward_G: begin', 'backward_G: done on main (graph traversed, branches joined)')
wrap(model.optimizer_G, 'zero_grad', None, None)
wrap(model, 'backward_D', 'backward_D: begin', 'backward_D: done on main')
og, od = model.optimizer_G.step, model.optimizer_D.step


def step_g():
    rec('side stream drained of G gradients', ops.side_stream_for(torch.cuda.current_stream()))
    og()
    rec('Adam G done')


def step_d():
    rec('side stream drained of D gradients', ops.side_stream_for(torch.cuda.current_stream()))
    od()
    rec('Adam D done')


model.optimizer_G.step, model.optimizer_D.step = step_g, step_d
netG_fwd = model.netG.forward
calls = {'n': 0}


def g_fwd(*a, **k):
    calls['n'] += 1
    r = netG_fwd(*a, **k)
    rec('generator pass %d forward done (its stream)' % (1 + (calls['n'] - 1) % 2))
    return r


model.netG.forward = g_fwd
for i in range(6):
    model.set_input(batches[i % 4]); model.optimize_parameters()
torch.cuda.synchronize()
marks.clear()
N = 12
for i in range(N):
    rec('step begin')
    model.set_input(batches[i % 4]); model.optimize_parameters()
torch.cuda.synchronize()
# split into steps
steps, cur = [], None
for name, ev in marks:
    if name == 'step begin':
        cur = []
        steps.append(cur)
    cur.append((name, ev))
names = [n for n, _ in steps[0]]
print('%-70s %8s' % ('event', 'ms after the step\'s begin on the main stream (mean of %d steps)' % (N - 2)))
for j, n in enumerate(names):
    ts = [s[0][1].elapsed_time(s[j][1]) for s in steps[1:-1] if len(s) == len(names)]
    print('%-70s %8.2f' % (n, sum(ts) / len(ts)))
d = [steps[i][0][1].elapsed_time(steps[i + 1][0][1]) for i in range(1, N - 1)]
print('step to step: %.2f ms' % (sum(d) / len(d)))
