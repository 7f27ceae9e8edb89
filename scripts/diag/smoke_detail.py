"""Diagnostic: element-wise error detail of smoke()'s comparison for the gate layer-1 weight gradient."""
import sys
sys.path.insert(0, ".")
import torch
from denet_amd import ops
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
from oracle import denet_oracle as O

dev = torch.device("cuda:0")
shape = EpisodeShape("smoke", N=2, way=1, shot=5, K=21, c=256, h=17, w=17, H=129, W=129, config_index=77, lam=2.0)
ep = make_episodes(shape, ignore_rows=4)
state = make_estimator_state(shape.K, shape.c)
ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
ref64 = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state,
                    dtype=torch.float64) if "--f64" in sys.argv else None
order = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias", "map.linear.weight", "map.linear.bias"]
P = [state[k].to(dev).requires_grad_(True) for k in order]
Fq = ep["Fq"].to(dev).requires_grad_(True)
Fs = ep["Fs"].to(dev).requires_grad_(True)
lk, lp, logits, binl = ops.HeadStepFunction.apply(Fq, Fs, *P, ep["Ys"].to(dev), ep["label"].reshape(shape.N, 1).to(dev), ep["Yq"].to(dev),
                                                  ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
lk.backward()
torch.cuda.synchronize()
for k, t in zip(order, P):
    a, b = t.grad.detach().double().cpu().flatten(), ref["d:" + k].double().flatten()
    bound = b.abs() + 1e-3 * b.abs().max()
    r = (a - b).abs() / bound
    i = int(r.argmax())
    msg = f"{k}: worst ratio {float(r[i]):.2e} at {i}: gpu {float(a[i]):.6e} oracle {float(b[i]):.6e} max|b| {float(b.abs().max()):.3e} norm-rel {float((a-b).norm()/b.norm()):.2e}"
    if ref64 is not None:
        c = ref64["d:" + k].double().flatten()
        msg += f" | f64 {float(c[i]):.6e}; oracle-f32 vs f64 worst ratio {float(((b - c).abs() / (c.abs() + 1e-3 * c.abs().max())).max()):.2e}, gpu vs f64 {float(((a - c).abs() / (c.abs() + 1e-3 * c.abs().max())).max()):.2e}"
    print(msg)
