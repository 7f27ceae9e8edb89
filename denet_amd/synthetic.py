"""Seeded synthetic episodes with the tensor contract the DENet head receives.

Contract (after dataset/voc_sbd.py:501-551 and
dataset/transforms.py:17-22 of the reference): the trainer hands the model

    Ys      [N, way, shot, H, W] int64 in {0, 1}
    Yq      [N, H, W] (way == 1) or [N, way, H, W] int64 in {0, 1}
    cls     [N] (way == 1) or [N, way] int64
    Yq_full [N, H, W] int64 in [0, K) or 255 (ignored)

and the feature producer (model/head/denet.py:56-73) hands the head

    Fs [N, way, shot, c, h, w], Fq [N, c, h, w]   (post-ReLU, so >= 0).

Everything is drawn from ``torch.Generator().manual_seed(1234 + config_index)``
on the CPU so that the GPU box, this container and the golden fixtures see the
same bits.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict

import torch


@dataclass(frozen=True)
class EpisodeShape:
    name: str
    N: int          # episodes per batch
    way: int
    shot: int
    K: int          # maximum_num_classes
    c: int          # feature depth
    h: int          # feature rows
    w: int          # feature cols
    H: int          # label rows
    W: int          # label cols
    config_index: int = 0
    lam: float = 1.0     # trainer's `lambda` re-weighting of the novel class
    feat_dtype: str = "f32"

    def asdict(self):
        return asdict(self)


# BASELINE.md §3 table.
CONFIGS = {
    "C1": EpisodeShape("C1", N=8, way=1, shot=1, K=21, c=256, h=41, w=41, H=321, W=321, config_index=1),
    "C2": EpisodeShape("C2", N=8, way=1, shot=5, K=21, c=256, h=41, w=41, H=321, W=321, config_index=2),
    "C3": EpisodeShape("C3", N=20, way=2, shot=5, K=81, c=256, h=41, w=41, H=321, W=321, config_index=3),
    "C4a": EpisodeShape("C4a", N=20, way=1, shot=1, K=21, c=256, h=41, w=41, H=321, W=321, config_index=4),
    "C4b": EpisodeShape("C4b", N=20, way=1, shot=1, K=81, c=256, h=41, w=41, H=321, W=321, config_index=4),
    "C5": EpisodeShape("C5", N=8, way=2, shot=5, K=81, c=256, h=81, w=81, H=642, W=642, config_index=5,
                       feat_dtype="bf16"),
}


def make_episodes(shape: EpisodeShape, *, planted: bool = False, seed_offset: int = 0,
                  ignore_rows: int = 8):
    """Return a dict of CPU tensors for one batch of episodes.

    planted=True builds features with class structure so that the upsampled
    arg-max has a wide top-2 margin (SURVEY.md §7 "hard parts"): every query
    pixel is a noisy copy of one of K fixed channel signatures and the
    estimator's class weights are expected to be set to those signatures by
    the caller (see ``planted_state``).
    """
    g = torch.Generator().manual_seed(1234 + shape.config_index + seed_offset)
    N, way, shot, K, c, h, w, H, W = (shape.N, shape.way, shape.shot, shape.K, shape.c,
                                      shape.h, shape.w, shape.H, shape.W)
    Fs = torch.relu(torch.randn(N, way, shot, c, h, w, generator=g))
    Fq = torch.relu(torch.randn(N, c, h, w, generator=g))
    Ys = torch.randint(0, 2, (N, way, shot, H, W), generator=g, dtype=torch.int64)
    if way == 1:
        Yq = torch.randint(0, 2, (N, H, W), generator=g, dtype=torch.int64)
    else:
        Yq = torch.randint(0, 2, (N, way, H, W), generator=g, dtype=torch.int64)
    # >= 1 support mask all zero (exercises den + 1e-3) and >= 1 query mask all
    # zero (exercises the BinaryIOU skip, utils/metrics.py:29-30).
    Ys[0, 0, 0] = 0
    if way == 1:
        Yq[N - 1] = 0
    else:
        Yq[N - 1, 0] = 0
    # `way` distinct labels per episode in [1, K)
    label = torch.stack([torch.randperm(K - 1, generator=g)[:way] + 1 for _ in range(N)], 0)
    Yq_full = torch.randint(0, K, (N, H, W), generator=g, dtype=torch.int64)
    if ignore_rows > 0:
        Yq_full[:, :min(ignore_rows, H)] = 255
    wgt = torch.ones(N, K)
    if way == 1:
        wgt[torch.arange(N), label[:, 0]] = shape.lam

    if planted:
        # K channel signatures, well separated; query pixels follow a coarse
        # class map so bilinear blends stay inside one class almost everywhere.
        sig = torch.zeros(K, c)
        for k in range(K):
            # class-dependent amplitude: two classes' blended logits then never
            # cross exactly on the 1/8-pixel grid of the bilinear up-sampling
            sig[k, (k * (c // K)):(k * (c // K) + max(c // K, 1))] = 0.6 + 0.0937 * k
        cells_y = torch.randint(0, K, (N, (h + 3) // 4, (w + 3) // 4), generator=g)
        cmap = cells_y.repeat_interleave(4, 1).repeat_interleave(4, 2)[:, :h, :w]      # [N,h,w]
        Fq = sig[cmap].permute(0, 3, 1, 2).contiguous() + 0.05 * torch.rand(N, c, h, w, generator=g)
        # supports: foreground pixels carry the signature of the episode label
        for n in range(N):
            for j in range(way):
                Fs[n, j] = 0.05 * torch.rand(shot, c, h, w, generator=g) + sig[label[n, j]].view(1, c, 1, 1)
        Ys[:] = 1
        Ys[0, 0, 0] = 0
    if shape.feat_dtype == "bf16":
        Fs = Fs.to(torch.bfloat16)
        Fq = Fq.to(torch.bfloat16)
    out = dict(Fs=Fs, Fq=Fq, Ys=Ys, Yq=Yq, label=label if way > 1 else label[:, 0],
               Yq_full=Yq_full, wgt=wgt)
    if planted:
        out["signatures"] = sig
    return out


def make_estimator_state(K: int, c: int, seed: int = 7):
    """Random-init parameters with the reference estimator's names and shapes
    (`weight [K,c]`, three 1x1 convs `[c,c,1,1]` + bias)."""
    g = torch.Generator().manual_seed(seed)
    bound = 1.0 / (c ** 0.5)

    def u(*s):
        return (torch.rand(*s, generator=g) * 2 - 1) * bound

    return {
        "weight": u(K, c),
        "gam.gate.0.weight": u(c, c, 1, 1),
        "gam.gate.0.bias": u(c),
        "gam.gate.2.weight": u(c, c, 1, 1),
        "gam.gate.2.bias": u(c),
        "map.linear.weight": u(c, c, 1, 1),
        "map.linear.bias": u(c),
    }
