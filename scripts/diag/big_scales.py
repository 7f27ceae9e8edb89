#!/usr/bin/env python3
"""Diagnostic (GPU): large up-sampling factors (16x .. 80x) through the fused step against the oracle: each case
either matches or is refused with DenetHipError (unsupported geometry) -- never a fault."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import rel_err                                     # noqa: E402
from denet_amd import ops, _lib                                 # noqa: E402
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state   # noqa: E402
from oracle import denet_oracle as O                            # noqa: E402  (checker)

ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
         "map.linear.weight", "map.linear.bias"]

dev = torch.device("cuda:0")
bad = 0
for i, (h, w, H, W, K) in enumerate([(21, 21, 321, 321, 21), (11, 11, 321, 321, 21), (5, 5, 321, 321, 21),
                                     (5, 7, 400, 97, 81), (3, 3, 193, 193, 9), (2, 2, 129, 65, 4), (1, 1, 64, 64, 21),
                                     (41, 41, 641, 641, 21), (6, 40, 500, 41, 33)]):
    s = EpisodeShape(f"bs{i}", N=2, way=1, shot=2, K=K, c=64, h=h, w=w, H=H, W=W, config_index=200 + i)
    ep = make_episodes(s, ignore_rows=2)
    state = make_estimator_state(s.K, s.c)
    P = [state[k].to(dev).requires_grad_(True) for k in ORDER]
    Fq = ep["Fq"].to(dev).requires_grad_(True)
    Fs = ep["Fs"].to(dev).requires_grad_(True)
    lab = ep["label"].reshape(s.N, s.way).to(dev)
    try:
        lk, lp, logits, binl = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), lab, ep["Yq"].to(dev),
                                                          ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
        lk.backward()
        torch.cuda.synchronize()
    except _lib.DenetHipError as e:
        print(f"{h}x{w}->{H}x{W} K={K}: refused ({e})", flush=True)
        continue
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state, None, None)
    errs = {"loss_k": rel_err(lk.detach().cpu(), ref["loss_k"]), "loss_p": rel_err(lp.cpu(), ref["loss_p"]),
            "dFq": rel_err(Fq.grad.cpu(), ref["dFq"]), "dFs": rel_err(Fs.grad.cpu(), ref["dFs"]),
            "logits": rel_err(logits.cpu(), ref["logits_full"])}
    ok = all(v < 1e-3 for v in errs.values())
    bad += not ok
    print(f"{h}x{w}->{H}x{W} K={K}: max={max(errs.values()):.2e} ({max(errs, key=errs.get)})" + ("" if ok else f" FAIL {errs}"), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
