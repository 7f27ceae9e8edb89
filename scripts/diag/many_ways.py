"""Diagnostic: the fused step against the oracle with more ways than a block has waves (stage_wn's second round),
duplicate labels among the ways, K on both sides of the layout thresholds, both feature types."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import rel_err
from denet_amd import ops
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
from oracle import denet_oracle as O
ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias", "map.linear.weight", "map.linear.bias"]
dev = torch.device("cuda:0")
bad = 0
for way in (4, 5, 8, 9, 12):
    for K in (21, 33, 81, 97):
        for fd in ("f32", "bf16"):
            for dup in (False, True):
                s = EpisodeShape("t", N=2, way=way, shot=1, K=K, c=256 if K == 81 else 64, h=9, w=11, H=33, W=41, config_index=9, lam=1.0, feat_dtype=fd)
                ep = make_episodes(s, ignore_rows=1); state = make_estimator_state(s.K, s.c)
                if dup:
                    lab = ep["label"].reshape(s.N, s.way)
                    lab[:, -1] = lab[:, 0]
                    ep["label"] = lab.reshape(ep["label"].shape)
                P = [state[k].to(dev).requires_grad_(True) for k in ORDER]
                Fq = ep["Fq"].to(dev).requires_grad_(True); Fs = ep["Fs"].to(dev).requires_grad_(True)
                lab = ep["label"].reshape(s.N, s.way).to(dev)
                lk, lp, logits, binl = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
                lk.backward(); torch.cuda.synchronize()
                ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
                lim = 8e-3 if fd == "bf16" else 1e-3
                errs = {"logits": rel_err(logits.cpu(), ref["logits_full"]), "bin": rel_err(binl.cpu(), ref["logits_binary"]), "lk": rel_err(lk.detach().cpu(), ref["loss_k"]), "lp": rel_err(lp.cpu(), ref["loss_p"]),
                        "dFq": rel_err(Fq.grad.float().cpu(), ref["dFq"]), "dFs": rel_err(Fs.grad.float().cpu(), ref["dFs"]), "dW": max(rel_err(t.grad.cpu(), ref["d:" + k]) for k, t in zip(ORDER, P))}
                ok = all(errs[k] < 1e-3 for k in ("logits", "bin", "lk", "lp", "dW")) and errs["dFq"] < lim and errs["dFs"] < lim
                bad += not ok
                if not ok: print(way, K, fd, dup, {k: f"{v:.1e}" for k, v in errs.items()}, "FAIL")
print("failures:", bad)
sys.exit(1 if bad else 0)
