"""Diagnostic: degenerate label maps against the oracle -- binary query labels all ignored (focal loss), an empty
support mask, a query map of ONE class, every class weight zero except one."""
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
def same(a, b, tol=2e-5):
    a = torch.as_tensor(a).detach().float().cpu(); b = torch.as_tensor(b).detach().float().cpu()
    if torch.isnan(b).any() or torch.isnan(a).any():
        return bool((torch.isnan(a) == torch.isnan(b)).all()) and (rel_err(torch.nan_to_num(a), torch.nan_to_num(b)) < tol)
    return rel_err(a, b) < tol
bad = 0
for name in ("focal_all_ignored", "empty_support", "one_class_query", "one_hot_weights", "yq_all_255_one_episode"):
    for K in (21, 81):
        s = EpisodeShape("t", N=3, way=2, shot=2, K=K, c=64, h=9, w=11, H=33, W=41, config_index=3, lam=1.0)
        ep = make_episodes(s, ignore_rows=1); state = make_estimator_state(s.K, s.c)
        if name == "focal_all_ignored": ep["Yq"][:] = 255
        if name == "yq_all_255_one_episode": ep["Yq"][: s.way] = 255
        if name == "empty_support": ep["Ys"][0] = 0
        if name == "one_class_query": ep["Yq_full"][:] = 3
        if name == "one_hot_weights": ep["wgt"][:] = 0; ep["wgt"][:, 5] = 1.0
        P = [state[k].to(dev).requires_grad_(True) for k in ORDER]
        Fq = ep["Fq"].to(dev).requires_grad_(True); Fs = ep["Fs"].to(dev).requires_grad_(True)
        lab = ep["label"].reshape(s.N, s.way).to(dev)
        lk, lp, logits, binl = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
        lk.backward(); torch.cuda.synchronize()
        ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
        res = {"lk": same(lk, ref["loss_k"]), "lp": same(lp, ref["loss_p"], 1e-4), "logits": same(logits, ref["logits_full"]), "dFq": same(Fq.grad, ref["dFq"]), "dFs": same(Fs.grad, ref["dFs"])}
        for k, t in zip(ORDER, P): res["d:" + k] = same(t.grad, ref["d:" + k])
        ok = all(res.values()); bad += not ok
        print(name, K, "lk", float(lk), float(ref["loss_k"]), "lp", float(lp), float(ref["loss_p"]), "OK" if ok else ("FAIL " + str([k for k, v in res.items() if not v])))
print("failures:", bad)
sys.exit(1 if bad else 0)
