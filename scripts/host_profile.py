"""cProfile of the host side of the bench step (where the ~32 ms of issue time per step go): top functions by own time"""
import cProfile, pstats, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
tmp = tempfile.mkdtemp()
model, opt = bench.build_model(0, 32, 128, tmp)
b = bench.synthetic_batch(32, 128, 0)
b = {k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}     # pinned host batch uploaded by set_input, as in bench.py
def step():
    model.set_input(b); model.optimize_parameters()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumtime').print_stats(22)
