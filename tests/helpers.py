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
