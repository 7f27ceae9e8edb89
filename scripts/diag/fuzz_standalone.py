"""Diagnostic: the STAND-ALONE kernel chain (ops.EstimatorFunction + ops.UpsampleWCEFunction + ops.upsample_focal:
what forward_with_binary + loss_denet run when a custom loss is configured) against the oracle on random shapes."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import rel_err
from denet_amd import ops
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
from oracle import denet_oracle as O
ORDER = ["gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias", "map.linear.weight", "map.linear.bias", "weight"]
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0; worst = {}
for it in range(count):
    h, w = rng.randint(2, 12), rng.randint(2, 14)
    if rng.random() < 0.25: h, w = rng.randint(15, 45), rng.randint(15, 45)
    H, W = h + rng.randint(0, 40), w + rng.randint(0, 40)
    if rng.random() < 0.5: H, W = (h - 1) * 4 + 1, (w - 1) * 8 + 1
    H, W = max(H, 2), max(W, 2)
    s = EpisodeShape(f"fz{it}", N=rng.randint(1, 3), way=rng.randint(1, 3), shot=rng.choice([1, 2, 5, 8]),
                     K=rng.choice([4, 8, 21, 32, 33, 64, 81, 96, 97, 128]), c=rng.choice([64, 256]), h=h, w=w, H=H, W=W,
                     config_index=300 + it, lam=rng.choice([1.0, 2.5]), feat_dtype=rng.choice(["f32", "f32", "bf16"]))
    ep = make_episodes(s, ignore_rows=min(2, H - 1)); state = make_estimator_state(s.K, s.c)
    P_ = {k: state[k].to(dev).contiguous().requires_grad_(True) for k in ORDER}
    Fq = ep["Fq"].to(dev).requires_grad_(True); Fs = ep["Fs"].to(dev).requires_grad_(True)
    lab = ep["label"].reshape(s.N, s.way).to(dev)
    logits, binl, Wn = ops.EstimatorFunction.apply(Fq, Fs, P_["weight"], P_["gam.gate.0.weight"], P_["gam.gate.0.bias"], P_["gam.gate.2.weight"],
                                                   P_["gam.gate.2.bias"], P_["map.linear.weight"], P_["map.linear.bias"], ep["Ys"].to(dev), lab, None, None, True)
    loss = ops.UpsampleWCEFunction.apply(logits, ep["Yq_full"].to(dev), ep["wgt"].to(dev).contiguous(), False)
    lp = ops.upsample_focal(binl, ep["Yq"].to(dev))
    loss.backward(); torch.cuda.synchronize()
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    errs = {"logits": rel_err(logits.detach().cpu(), ref["logits_full"]), "bin": rel_err(binl.detach().cpu(), ref["logits_binary"]), "lk": rel_err(loss.detach().cpu(), ref["loss_k"]),
            "lp": rel_err(lp.cpu(), ref["loss_p"]), "dFq": rel_err(Fq.grad.float().cpu(), ref["dFq"]), "dFs": rel_err(Fs.grad.float().cpu(), ref["dFs"])}
    for k in ORDER: errs["d:" + k] = rel_err(P_[k].grad.cpu(), ref["d:" + k])
    lim = {k: (8e-3 if s.feat_dtype == "bf16" and k in ("dFq", "dFs") else 1e-3) for k in errs}
    b = {k: v for k, v in errs.items() if not v < lim[k]}
    bad += bool(b)
    for k, v in errs.items():
        if lim[k] == 1e-3: worst[k] = max(worst.get(k, 0), v)
    if b: print(it, s, {k: f"{v:.1e}" for k, v in b.items()}, "FAIL", flush=True)
print("worst:", {k: f"{v:.1e}" for k, v in worst.items()}); print("failures:", bad)
sys.exit(1 if bad else 0)
