#!/usr/bin/env python3
"""Randomised parity sweep (GPU): the fused head step against the oracle on random small shapes --
class counts on both sides of every kernel threshold (8, 32, 64, 96, 128), power-of-two and other
up-sampling scales, 1-3 ways, 1-8 shots, both channel counts.  Prints the worst relative error per tensor."""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import rel_err                                     # noqa: E402
from denet_amd import ops                                       # noqa: E402
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state   # noqa: E402
from oracle import denet_oracle as O                            # noqa: E402  (checker)

ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
         "map.linear.weight", "map.linear.bias"]


def sweep(seed=0, count=24, verbose=True):
    """-> (worst relative error per tensor, list of failing cases)."""
    dev = torch.device("cuda:0")
    rng = random.Random(seed)
    worst, failures = {}, []
    for it in range(count):
        h, w = rng.randint(3, 12), rng.randint(3, 14)
        if rng.random() < 0.25:                                     # several position tiles / tile groups, ragged last tile
            h, w = rng.randint(15, 45), rng.randint(15, 45)
        if rng.random() < 0.5:                                      # power-of-two scale: (H-1) = (h-1) * 2^j
            j = rng.choice([1, 2, 3])
            H, W = (h - 1) * 2 ** j + 1, (w - 1) * 2 ** rng.choice([1, 2, 3]) + 1
        else:
            H, W = h + rng.randint(0, 40), w + rng.randint(0, 40)
        s = EpisodeShape(f"fz{it}", N=rng.randint(1, 3), way=rng.randint(1, 3), shot=rng.choice([1, 2, 3, 5, 8]),
                         K=rng.choice([4, 7, 8, 9, 21, 32, 33, 64, 65, 81, 96, 97, 128]), c=rng.choice([64, 256]),
                         h=h, w=w, H=H, W=W, config_index=100 + it, lam=rng.choice([1.0, 2.5]),
                         feat_dtype=rng.choice(["f32", "f32", "bf16"]))
        ep = make_episodes(s, ignore_rows=min(2, H - 1))
        state = make_estimator_state(s.K, s.c)
        P = [state[k].to(dev).requires_grad_(True) for k in ORDER]
        Fq = ep["Fq"].to(dev).requires_grad_(True)
        Fs = ep["Fs"].to(dev).requires_grad_(True)
        lab = ep["label"].reshape(s.N, s.way).to(dev)
        # train mode in half of the cases: injected dropout multipliers (keep-prob 0.5, scale 2), the
        # same on both sides (model/common.py:231,326)
        gm = wm = None
        if rng.random() < 0.5:
            tg = torch.Generator().manual_seed(9000 + it)
            gm = 2.0 * torch.bernoulli(torch.full((s.N * s.way * s.shot, s.c), 0.5), generator=tg)
            wm = 2.0 * torch.bernoulli(torch.full((s.N, s.K, s.c), 0.5), generator=tg)
        lk, lp, logits, binl = ops.HeadStepFunction.apply(
            Fq, Fs, *P, ep["Ys"].to(dev), lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev),
            None if gm is None else gm.to(dev), None if wm is None else wm.to(dev))
        lk.backward()
        torch.cuda.synchronize()
        ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state,
                              None if gm is None else gm.reshape(-1, s.c, 1, 1), wm)
        errs = {"logits_full": rel_err(logits.cpu(), ref["logits_full"]), "logits_binary": rel_err(binl.cpu(), ref["logits_binary"]),
                "loss_k": rel_err(lk.detach().cpu(), ref["loss_k"]), "loss_p": rel_err(lp.cpu(), ref["loss_p"]),
                "dFq": rel_err(Fq.grad.float().cpu(), ref["dFq"]), "dFs": rel_err(Fs.grad.float().cpu(), ref["dFs"])}
        for k, t in zip(ORDER, P):
            errs["d:" + k] = rel_err(t.grad.cpu(), ref["d:" + k])
        # bf16 features: the feature gradients are rounded to bf16 on the way out (8 significant bits)
        lim = {k: (8e-3 if s.feat_dtype == "bf16" and k in ("dFq", "dFs") else 1e-3) for k in errs}
        bad = {k: v for k, v in errs.items() if not v < lim[k]}
        if bad:
            failures.append((s, bad))
        if verbose:
          print(f"{it:3d} N={s.N} way={s.way} shot={s.shot} K={s.K} c={s.c} {h}x{w}->{H}x{W} {s.feat_dtype}{' masks' if gm is not None else ''} max={max(errs.values()):.2e}"
              + (f"  FAIL {bad}" if bad else ""), flush=True)
        for k, v in errs.items():
            if lim[k] == 1e-3:
                worst[k] = max(worst.get(k, 0.0), v)
    return worst, failures


if __name__ == "__main__":
    worst, failures = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 24)
    print("worst:", {k: f"{v:.2e}" for k, v in worst.items()})
    sys.exit(1 if failures else 0)
