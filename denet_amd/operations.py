"""Reshape helpers and binary-logit extraction with the reference's names and
argument meaning (utils/operations.py:261-338 of the reference)."""
from __future__ import annotations

import torch

from . import ops


def merge_first_k_dim(x: torch.Tensor, dims):
    """(a, b, c, d, e), dims=(0, 1, 2) -> (a*b*c, d, e).  utils/operations.py:261-289."""
    dims = tuple(dims)
    assert len(dims) >= 2, 'length of dims must greater than 2'
    assert all(dims[i] - dims[i - 1] == 1 for i in range(1, len(dims))), 'dims not continuous'
    shape = list(x.size())
    merged = 1
    for i in dims:
        merged *= shape[i]
    return x.reshape([merged] + [shape[i] for i in range(len(shape)) if i not in dims])


def depart_first_dim(x: torch.Tensor, dims):
    """(a1*a2, b, c) , dims=(a1, a2) -> (a1, a2, b, c).  utils/operations.py:292-310."""
    return x.reshape(list(dims) + list(x.size())[1:])


def get_binary_logits(logits_full: torch.Tensor, label: torch.Tensor, base: bool = True):
    """(N, K, h, w), (N) -> (N, 2, h, w): channel 0 = max over the other K-1 rows (row 0 when
    base=False), channel 1 = row `label`.  utils/operations.py:313-338.  The result stays in the autograd
    graph, but a backward pass through it raises: the reference never back-propagates through the binary
    logits (trainer/trainers.py:972-974) and no gradient kernel exists for them."""
    N = logits_full.shape[0]
    return ops.BinaryLogitsFunction.apply(logits_full, label.reshape(N, 1), 1, base)


def process_nway(Yq_pre: torch.Tensor, way: int):
    """Fuse the 2*way logits of a class-agnostic model into per-way binary scores
    (trainer/utils.py:128-154): (N, 2*way, h, w) -> (N, way, 2, h, w).  Channel 2i is way i's
    background, 2i+1 its foreground; the shared background is the max over the way backgrounds,
    and way i's binary pair is get_binary_logits([bg, fg_0..fg_{way-1}], label = i + 1).

    Pinned by tests/golden/trainer_glue.npz: trainer/utils.py itself cannot be imported in the build
    container (it pulls in torchvision through model.head), so oracle/gen_golden.py restates its
    index_select / max / cat glue around the reference's own get_binary_logits; the result must be
    bit-identical (tests/test_gpu_api.py)."""
    N = Yq_pre.shape[0]
    x = Yq_pre.detach().float()
    background = x[:, 0::2].max(dim=1, keepdim=True)[0]
    scores = torch.cat([background, x[:, 1::2]], dim=1).contiguous()          # (N, way + 1, h, w)
    labels = torch.arange(1, way + 1, device=x.device, dtype=torch.int64).repeat(N, 1)   # (N, way)
    out = ops.binary_logits(scores, labels, way=way, base=True)                # (N*way, 2, h, w)
    return out.reshape(N, way, 2, *x.shape[-2:])
