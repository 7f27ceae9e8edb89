"""Host time of the drop-in head step, forward and backward apart (no profiler), and the GPU-bound rate."""
import sys, os, time
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
leaves = [Fq, Fs] + list(we.parameters())
tf = tb = tz = 0.0
def fused(timeit=False):
    global tf, tb, tz
    t0 = time.perf_counter()
    for p in leaves: p.grad = None
    t1 = time.perf_counter()
    lk, lp, full, binl = we.head_step(Fq, Fs, ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"])
    t2 = time.perf_counter()
    lk.backward()
    t3 = time.perf_counter()
    if timeit: tz += t1 - t0; tf += t2 - t1; tb += t3 - t2
for _ in range(50): fused()
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for _ in range(n): fused(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("us/step: enqueue %.1f (grad reset %.1f, forward %.1f, backward %.1f), with final sync %.1f" % (
    (t1 - t0) / n * 1e6, tz / n * 1e6, tf / n * 1e6, tb / n * 1e6, (t2 - t0) / n * 1e6))
