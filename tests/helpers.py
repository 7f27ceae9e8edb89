"""Shared helpers for the parity tests (fixtures -> tensors)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = ["eval_1way_5shot", "eval_2way_3shot", "eval_1way_1shot", "train_1way_5shot",
         "planted_1way_1shot"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, way, shot, K, c, h, w, H, W = [int(v) for v in z["shape"]]
    ep = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in:")}
    params = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}
    gate_mask = weight_mask = None
    if "mask:gate" in z.files:
        gate_mask = torch.from_numpy(z["mask:gate"])             # [B,c]
        weight_mask = torch.from_numpy(z["mask:weight"])         # [N,K,c]
    dims = dict(N=N, way=way, shot=shot, K=K, c=c, h=h, w=w, H=H, W=W)
    return z, dims, ep, params, gate_mask, weight_mask


def rel_err(a, b):
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_err(a, b, rtol=1e-3, afloor=1e-6):
    """Element-wise bound: the worst |a-b| / (rtol*|b| + afloor*max|b|).  <= 1 means every entry is within
    `rtol` RELATIVE to itself, with an absolute floor for entries that are tiny next to the tensor's
    largest (north_star: "within 1e-3 relative"; rel_err above only bounds the error by the largest entry)."""
    a = torch.as_tensor(a).detach().double()
    b = torch.as_tensor(b).detach().double()
    bound = rtol * b.abs() + afloor * b.abs().max().clamp_min(1e-30)
    return float(((a - b).abs() / bound).max())


def assert_close(a, b, tol, name="", rtol=1e-3, afloor=1e-6):
    """max-normalised error below `tol` AND every entry within rtol of itself (+ floor)."""
    r, e = rel_err(a, b), elem_err(a, b, rtol, afloor)
    assert r < tol and e <= 1.0, f"{name}: max-normalised error {r:.3e} (bar {tol:.1e}), element-wise {e:.3f} of its bound"
