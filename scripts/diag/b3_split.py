"""Diagnostic: time of the backward with and without the dFs blocks (phases of HeadStepEngine.run_backward):
phase 1 = B1 + B2 + the parameter-gradient blocks of B3, phase 2 = B3's dFs blocks (+ the chunk writer)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd.engine import HeadStepEngine
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
s = CONFIGS[cfg]
dev = torch.device("cuda:0")
fdt = torch.bfloat16 if s.feat_dtype == "bf16" else torch.float32
ep = {k: v.to(dev).contiguous() for k, v in make_episodes(s).items()}
params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(s.K, s.c).items()}
eng = HeadStepEngine(s.N, s.way, s.shot, s.K, s.c, s.h, s.w, s.H, s.W, dev, fdt)
labels = ep["label"].reshape(s.N, s.way).contiguous()
eng.use_bound(eng.make_bound(ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params, drop=None))
st = torch.cuda.current_stream().cuda_stream
def run(phases):
    eng.run_forward(st)
    for p in phases: eng.run_backward(p, st)
for _ in range(30): run((0,))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
res = {"whole": [], "p1": [], "p2": []}
for it in range(60):
    eng.run_forward(st); ev[0].record(); eng.run_backward(0, st); ev[1].record()
    eng.run_forward(st); ev[2].record(); eng.run_backward(1, st); ev[3].record()
    e4 = torch.cuda.Event(enable_timing=True); e5 = torch.cuda.Event(enable_timing=True)
    e4.record(); eng.run_backward(2, st); e5.record()
    torch.cuda.synchronize()
    res["whole"].append(ev[0].elapsed_time(ev[1]) * 1e3); res["p1"].append(ev[2].elapsed_time(ev[3]) * 1e3); res["p2"].append(e4.elapsed_time(e5) * 1e3)
import statistics
print(cfg, {k: round(statistics.median(v), 1) for k, v in res.items()}, "us (backward: whole / without dFs blocks / dFs blocks only)")
