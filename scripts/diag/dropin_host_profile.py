"""Where the host time of the drop-in head step goes (cProfile over 300 steps of WeightEstimator.head_step + backward)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd.head import WeightEstimator
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state
dev = torch.device("cuda:0")
s = CONFIGS["C2"]
ep = {k: v.to(dev).contiguous() for k, v in make_episodes(s).items()}
we = WeightEstimator(s.K, s.c).to(dev).eval()
we.load_state_dict(make_estimator_state(s.K, s.c))
labels = ep["label"].reshape(s.N, s.way).contiguous()
Fq = ep["Fq"].clone().requires_grad_(True); Fs = ep["Fs"].clone().requires_grad_(True)
def fused():
    we.zero_grad(set_to_none=True); Fq.grad = Fs.grad = None
    lk, lp, full, binl = we.head_step(Fq, Fs, ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"])
    lk.backward()
for _ in range(20): fused()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): fused()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue us/step %.1f, with final sync %.1f" % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fused()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
