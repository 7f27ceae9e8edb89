import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from denet_amd import ops
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
from oracle import denet_oracle as O
ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias", "map.linear.weight", "map.linear.bias"]
dev = torch.device("cuda:0")
for (h, w, H, W) in [(1, 7, 5, 31), (1, 7, 1, 31), (7, 1, 31, 5), (1, 1, 4, 4)]:
    s = EpisodeShape("t", N=2, way=2, shot=2, K=21, c=64, h=h, w=w, H=H, W=W, config_index=7, lam=1.0)
    ep = make_episodes(s); state = make_estimator_state(s.K, s.c)
    P = [state[k].to(dev).requires_grad_(True) for k in ORDER]
    Fq = ep["Fq"].to(dev).requires_grad_(True); Fs = ep["Fs"].to(dev).requires_grad_(True)
    lab = ep["label"].reshape(s.N, s.way).to(dev)
    try:
        lk, lp, logits, binl = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
        lk.backward(); torch.cuda.synchronize()
    except Exception as e:
        print(h, w, H, W, "raised:", str(e)[:100]); continue
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    print(h, w, H, W, "ours: lk", float(lk), "lp", float(lp), "dFq nan", bool(torch.isnan(Fq.grad).any()), "| ref: lk", float(ref["loss_k"]), "lp", float(ref["loss_p"]), "dFq nan", bool(torch.isnan(ref["dFq"]).any()))
