"""Experiment: classifier backward roles timed apart (stand-alone kernel, C2 batch 8)."""
import sys, torch
sys.path.insert(0, '.')
from denet_amd import ops
N, K, C, h, w = 8, 21, 256, 41, 41
dev = torch.device("cuda:0")
Fq = torch.randn(N, C, h, w, device=dev).relu()
Wn = torch.randn(N, K, C, device=dev)
da = torch.randn(N, K, h, w, device=dev); db = torch.randn(N, K, h, w, device=dev)
scale = torch.ones(N, device=dev)
def t(fn, reps=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
from denet_amd._lib import call, feat_code, query
n_part = query("denet_classify_bwd_parts", h * w)
dFq = torch.empty_like(Fq); part = torch.empty(N, n_part, K, C, device=dev)
both = lambda: call("denet_classify_bwd", Fq, feat_code(Fq), Wn, da, db, scale, None, N, K, C, h * w, dFq, part)
only_dfq = lambda: call("denet_classify_bwd", Fq, feat_code(Fq), Wn, da, db, scale, None, N, K, C, h * w, dFq, None)
only_dwn = lambda: call("denet_classify_bwd", Fq, feat_code(Fq), Wn, da, db, scale, None, N, K, C, h * w, None, part)
empty = lambda: call("denet_batch_mean", scale, N, 1, scale[:1].clone())
print(f"both roles in one block {t(both):.1f} us | dFq only {t(only_dfq):.1f} us | dWn only {t(only_dwn):.1f} us")
