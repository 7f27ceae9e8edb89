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
    base=False), channel 1 = row `label`.  utils/operations.py:313-338.  Forward only: the
    reference never back-propagates through the binary logits (trainer/trainers.py:972-974)."""
    N = logits_full.shape[0]
    return ops.binary_logits(logits_full.detach(), label.reshape(N, 1), way=1, base=base)
