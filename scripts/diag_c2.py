"""Diagnostic: stage-by-stage backward errors at full C2 size (GPU vs oracle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from denet_amd import ops
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state
from oracle import denet_oracle as O

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
s = CONFIGS[cfg]
dev = torch.device("cuda:0")
ep = make_episodes(s)
state = make_estimator_state(s.K, s.c)
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
# oracle with retained logits grad
Fq = ep["Fq"].float().clone().requires_grad_(True)
Fs = ep["Fs"].float().clone().requires_grad_(True)
pp = {k: v.clone().requires_grad_(True) for k, v in state.items()}
full, binl, Wn = O.head_forward(Fq, Fs, ep["Ys"], ep["label"], pp)
full.retain_grad(); Wn.retain_grad()
up = O.upsample(full, (s.H, s.W))
lk = O.loss_knowledge(up, ep["Yq_full"], ep["wgt"])
lk.backward()
# gpu pieces
logits_g = full.detach().to(dev).requires_grad_(True)
loss, per, scale, da, db = ops.upsample_wce(logits_g, ep["Yq_full"].to(dev), ep["wgt"].to(dev).contiguous(), grad=True)
dl = ops.combine_slabs(da, db, scale, None)
print("loss", rel(loss, lk), "dlogits", rel(dl, full.grad))
err = (dl.cpu() - full.grad).abs()
idx = err.reshape(-1).argmax().item()
print(" worst idx", idx, "n,k,y,x =", [int(v) for v in torch.unravel_index(torch.tensor(idx), err.shape)], float(err.reshape(-1)[idx]), float(full.grad.abs().max()))
# per-row error profile
e_rows = err.amax(dim=(0, 1, 3)); print(" err by row:", [f"{v:.1e}" for v in e_rows.tolist()][:45])
e_cols = err.amax(dim=(0, 1, 2)); print(" err by col:", [f"{v:.1e}" for v in e_cols.tolist()][:45])
dFq_g, part, n_part = ops.classify_bwd(ep["Fq"].to(dev), Wn.detach().to(dev).contiguous(), full.grad.to(dev).contiguous())
print("dFq (oracle dlogits in)", rel(dFq_g, Fq.grad), "dWn", rel(part.sum(1), Wn.grad))
e2 = (dFq_g.cpu() - Fq.grad).abs().reshape(s.N, s.c, -1)
print(" dFq err by position (max over n,c) worst 10:", torch.topk(e2.amax(dim=(0, 1)), 10))
