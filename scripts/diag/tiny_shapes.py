"""Diagnostic: the fused step against the oracle on the smallest feature maps (P = 4 .. 12), K on both sides of the
bf16-pipe thresholds, both feature types: the shapes where the backward falls back from the 8-position fragments."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import rel_err
from denet_amd import ops, _lib
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
from oracle import denet_oracle as O
ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias", "map.linear.weight", "map.linear.bias"]
dev = torch.device("cuda:0")
bad = 0
for (h, w) in [(2, 2), (2, 3), (1, 7), (2, 4), (3, 3), (3, 4)]:
    for K in (21, 33, 81, 96, 97):
        for fd in ("f32", "bf16"):
            s = EpisodeShape("t", N=2, way=2, shot=2, K=K, c=64, h=h, w=w, H=4 * h + 1, W=4 * w + 3, config_index=7, lam=1.0, feat_dtype=fd)
            ep = make_episodes(s); state = make_estimator_state(s.K, s.c)
            P = [state[k].to(dev).requires_grad_(True) for k in ORDER]
            Fq = ep["Fq"].to(dev).requires_grad_(True); Fs = ep["Fs"].to(dev).requires_grad_(True)
            lab = ep["label"].reshape(s.N, s.way).to(dev)
            try:
                lk, lp, logits, binl = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
                lk.backward(); torch.cuda.synchronize()
            except Exception as e:
                print(h, w, K, fd, "raised", str(e)[:80]); continue
            ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
            lim = 8e-3 if fd == "bf16" else 1e-3
            errs = {"logits": rel_err(logits.cpu(), ref["logits_full"]), "dFq": rel_err(Fq.grad.float().cpu(), ref["dFq"]),
                    "dFs": rel_err(Fs.grad.float().cpu(), ref["dFs"]), "dW": max(rel_err(t.grad.cpu(), ref["d:" + k]) for k, t in zip(ORDER, P))}
            pl = _lib.head_step_plan(s.N, s.way, s.shot, K, 64, h, w, s.H, s.W, _lib.BF16 if fd == "bf16" else _lib.F32)
            ok = errs["logits"] < 1e-3 and errs["dFq"] < lim and errs["dFs"] < lim and errs["dW"] < 1e-3
            bad += not ok
            print(h, w, K, fd, "pipe", pl["bwd_bf16_pipe"], "persist", pl["bwd_persistent"], {k: f"{v:.1e}" for k, v in errs.items()}, "" if ok else "FAIL")
print("failures:", bad)
sys.exit(1 if bad else 0)
