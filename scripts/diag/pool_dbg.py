import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from denet_amd import ops
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
dev = torch.device("cuda:0")
ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias", "map.linear.weight", "map.linear.bias"]
s = EpisodeShape("t", N=2, way=1, shot=2, K=21, c=64, h=9, w=11, H=33, W=41, config_index=3, lam=1.0)
ep = make_episodes(s, ignore_rows=1); state = make_estimator_state(s.K, s.c)
def pool(): return {k: len(v) for k, v in ops.workspace_pool._free.items()}
for it in range(3):
    P = [state[k].to(dev).requires_grad_(True) for k in ORDER]
    Fq = ep["Fq"].to(dev).requires_grad_(True); Fs = ep["Fs"].to(dev).requires_grad_(True)
    lk, lp, logits, binl = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), ep["label"].reshape(s.N, 1).to(dev), ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
    print("after fwd", it, pool())
    lk.backward()
    print("after bwd", it, pool())
with torch.no_grad():
    out = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), ep["label"].reshape(s.N, 1).to(dev), ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
print("after no_grad fwd", pool())
