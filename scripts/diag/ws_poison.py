"""Does any kernel of the fused step read workspace it did not write in the same step?  Run the step on the same
inputs with the workspace pre-filled with zeros, with NaNs and with a large constant; every output must be
bit-identical."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import _lib
from denet_amd.engine import HeadStepEngine, PARAM_ORDER
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state

dev = torch.device("cuda:0")
shapes = [EpisodeShape("t", N=4, way=1, shot=2, K=21, c=256, h=9, w=9, H=65, W=65, config_index=5),
          EpisodeShape("u", N=2, way=2, shot=3, K=40, c=64, h=7, w=9, H=50, W=70, config_index=6),
          EpisodeShape("c2", N=8, way=1, shot=5, K=21, c=256, h=41, w=41, H=321, W=321, config_index=2)]
for s in shapes:
    ep = {k: v.to(dev).contiguous() for k, v in make_episodes(s, ignore_rows=2).items()}
    params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(s.K, s.c).items()}
    lab = ep["label"].reshape(s.N, s.way).contiguous()
    eng = HeadStepEngine(s.N, s.way, s.shot, s.K, s.c, s.h, s.w, s.H, s.W, dev)
    for drop in (None, _lib.Dropout(1234, 0.5, 0.5)):
        res = []
        for fill in (0.0, float("nan"), 1e30):
            eng.ws.view(torch.float32).fill_(fill)
            eng.grad_bucket.fill_(fill); eng.dFq.fill_(fill); eng.dFs.fill_(fill)
            eng.forward(ep["Fq"], ep["Fs"], ep["Ys"], lab, ep["Yq"], ep["Yq_full"], ep["wgt"], params, drop=drop)
            eng.backward(ep["Fq"], lab, params, drop=drop)
            torch.cuda.synchronize()
            res.append([t.clone() for t in (eng.losses, eng.logits_full, eng.logits_binary, eng.dFq, eng.dFs, eng.grad_bucket)])
        names = ["losses", "logits", "binary", "dFq", "dFs", "param grads"]
        for i, nm in enumerate(names):
            for j in (1, 2):
                a, b = res[0][i], res[j][i]
                same = torch.equal(a, b)
                if not same:
                    bad = (a != b) | (a.isnan() != b.isnan())
                    print(s.name, "drop" if drop else "eval", nm, "fill", j, "DIFFERS:", int(bad.sum()), "of", a.numel(),
                          "first idx", bad.reshape(-1).nonzero()[:5].reshape(-1).tolist(), "nan:", int(b.isnan().sum()))
        if drop is None or True:
            off = 0
            for nm in PARAM_ORDER:
                n = eng.grads[nm].numel()
                d = (res[0][5][off:off + n] != res[1][5][off:off + n]).sum().item()
                if d: print("   ", nm, "differs in", d)
                off += n
    print(s.name, "done")
