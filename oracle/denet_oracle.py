"""CPU restatement of the DENet head hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  ``denet_amd`` never imports anything under ``oracle/``.

It restates, with stock PyTorch CPU ops in the reference's op order, what the
reference head computes (citations are into /root/reference).  It is pinned
against golden fixtures captured from the reference's own code run in the
build container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``); the
reference has no test suite of its own for this path.

Nothing here is copied from the reference: the functions are written from the
algorithm description in SURVEY.md §8a and checked against the fixtures.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

EPS_POOL = 1e-3          # model/common.py:170 (eps of MaskedAveragePooling.forward)
IGNORE = 255             # loss/losses.py:19,47
FOCAL_ALPHA = 0.25       # loss/losses.py:47
FOCAL_GAMMA = 2


# --------------------------------------------------------------------------
# index rules (verified against ATen by the fixtures)
# --------------------------------------------------------------------------
def nearest_src_index(out_size: int, in_size: int) -> np.ndarray:
    """F.interpolate(mode='nearest') source index for each destination index.
    src = min(floor(dst * fp32(in/out)), in-1)    (model/common.py:190,238)."""
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = np.floor(dst * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


def bilinear_taps(out_size: int, in_size: int):
    """align_corners=True bilinear taps (trainer/trainers.py:955-956):
    src = dst*(in-1)/(out-1), i0=floor(src), i1=min(i0+1,in-1), frac=src-i0."""
    if out_size > 1:
        scale = np.float32(in_size - 1) / np.float32(out_size - 1)
    else:
        scale = np.float32(0)
    src = np.arange(out_size, dtype=np.float32) * scale
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    frac = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, frac


# --------------------------------------------------------------------------
# K1: mask resize + masked pooling      (model/common.py:188-198, 234-246)
# --------------------------------------------------------------------------
def resize_mask(Ys: torch.Tensor, h: int, w: int, dtype=torch.float32) -> torch.Tensor:
    """[B,H,W] int -> [B,1,h,w] float, nearest."""
    return F.interpolate(Ys.unsqueeze(1).to(dtype), size=(h, w), mode="nearest")


def gam_attention(Fs: torch.Tensor, Ys: torch.Tensor) -> torch.Tensor:
    """att = mean over all h*w positions of mask*Fs  -> [B,c,1,1]
    (GAM.forward, model/common.py:241-243)."""
    h, w = Fs.shape[-2:]
    return F.adaptive_avg_pool2d(resize_mask(Ys, h, w, Fs.dtype) * Fs, (1, 1))


def gate(att, p, gate_mask=None):
    """g = sigmoid(dropout(conv2(relu(conv1(att)))))  (model/common.py:225-232).
    gate_mask, if given, is the already-scaled dropout multiplier [B,c,1,1]."""
    z = F.conv2d(att, p["gam.gate.0.weight"], p["gam.gate.0.bias"])
    z = F.relu(z)
    z = F.conv2d(z, p["gam.gate.2.weight"], p["gam.gate.2.bias"])
    if gate_mask is not None:
        z = z * gate_mask
    return torch.sigmoid(z)


def masked_avg_pool_project(Fs_g, Ys, p):
    """(sum(mask*F)/(sum(mask)+1e-3)) -> 1x1 conv   (model/common.py:188-196)."""
    h, w = Fs_g.shape[-2:]
    m = resize_mask(Ys, h, w, Fs_g.dtype)
    num = torch.sum(m * Fs_g, dim=(2, 3), keepdim=True)
    den = torch.sum(m, dim=(2, 3), keepdim=True)
    proto = num / (den + EPS_POOL)
    return F.conv2d(proto, p["map.linear.weight"], p["map.linear.bias"])


# --------------------------------------------------------------------------
# K2: dynamic extension of the class-weight matrix (model/common.py:304-368)
# --------------------------------------------------------------------------
def extend_weights(Fs, Ys, labels, p, gate_mask=None, weight_mask=None):
    """Fs [N,way,shot,c,h,w], Ys [N,way,shot,H,W], labels [N,way] ->
    weights [N,K,c].  gate_mask [N*way*shot,c,1,1] and weight_mask [N,K,c]
    are the scaled dropout multipliers (train mode) or None (eval)."""
    N, way, shot, c, h, w = Fs.shape
    B = N * way * shot
    Fm = Fs.reshape(B, c, h, w)
    Ym = Ys.reshape(B, *Ys.shape[-2:])
    g = gate(gam_attention(Fm, Ym), p, gate_mask)
    proto = masked_avg_pool_project(g * Fm, Ym, p)               # [B,c,1,1]
    proto = proto.reshape(N, way, shot, c)
    out = []
    for n in range(N):
        Wn = p["weight"].clone()
        if weight_mask is not None:
            Wn = Wn * weight_mask[n]
        pbar = proto[n].mean(dim=1)                              # [way,c]
        for j in range(way):                                     # later way wins
            Wn[int(labels[n, j])] = pbar[j]
        out.append(Wn)
    return torch.stack(out, 0)


# --------------------------------------------------------------------------
# K3 / K3e: classifier and binary-logit extraction
# --------------------------------------------------------------------------
def classify(Fq, Wn):
    """logits[n,k,y,x] = sum_c Fq[n,c,y,x]*Wn[n,k,c]  (model/common.py:390-394)."""
    N, c, h, w = Fq.shape
    x = Fq.permute(0, 2, 3, 1).reshape(N, h * w, c)
    return torch.bmm(x, Wn.transpose(1, 2)).transpose(1, 2).reshape(N, -1, h, w)


def binary_logits(logits_full, label, base=True):
    """[N,K,h,w],[N] -> [N,2,h,w]; ch0 = max over the other K-1 rows (or row 0
    when base=False), ch1 = row `label`  (utils/operations.py:313-338)."""
    N, K = logits_full.shape[:2]
    idx = torch.arange(N)
    fg = logits_full[idx, label]
    if base:
        keep = torch.arange(K).unsqueeze(0) != label.unsqueeze(1)             # [N,K]
        neg = torch.finfo(logits_full.dtype).min
        bg = torch.where(keep[:, :, None, None], logits_full,
                         torch.full_like(logits_full, neg)).max(dim=1)[0]
    else:
        bg = logits_full[:, 0]
    return torch.stack([bg, fg], dim=1)


def process_nway(Yq_pre, way):
    """(N, 2*way, h, w) -> (N, way, 2, h, w): shared background = max over the way backgrounds (even
    channels), then binary_logits([bg, fg_0..], label = i + 1) per way  (trainer/utils.py:128-154)."""
    N = Yq_pre.shape[0]
    bg = Yq_pre[:, 0::2].max(dim=1, keepdim=True)[0]
    scores = torch.cat([bg, Yq_pre[:, 1::2]], dim=1)
    return torch.stack([binary_logits(scores, torch.full([N], i + 1, dtype=torch.int64)) for i in range(way)], dim=1)


def head_forward(Fq, Fs, Ys, label, p, gate_mask=None, weight_mask=None):
    """DENet.forward after the feature producer (model/head/denet.py:57-93).
    label is [N] (way==1) or [N,way]. Returns (logits_full, logits_binary, Wn)."""
    way = Fs.shape[1]
    lab = label.unsqueeze(-1) if label.dim() == 1 else label
    Wn = extend_weights(Fs, Ys, lab, p, gate_mask, weight_mask)
    full = classify(Fq, Wn)
    if way > 1:
        b = torch.stack([binary_logits(full, lab[:, i]) for i in range(way)], dim=1)
        b = b.reshape(-1, *b.shape[2:])
    else:
        b = binary_logits(full, lab[:, 0])
    return full, b, Wn


# --------------------------------------------------------------------------
# K4 / K5: upsample + losses
# --------------------------------------------------------------------------
def upsample(x, size):
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)


def weighted_ce_2d(up, target, weight=None):
    """cross_entropy2d on already-upsampled logits (loss/losses.py:39-44)."""
    K = up.shape[1]
    x = up.permute(0, 2, 3, 1).reshape(-1, K)
    return F.cross_entropy(x, target.reshape(-1), weight=weight, ignore_index=IGNORE)


def loss_knowledge(full_up, Yq_full, wgt, dtype=torch.float32):
    """mean over episodes of per-episode weighted CE (trainer/utils.py:270-277)."""
    N = full_up.shape[0]
    tot = 0.0
    for n in range(N):
        tot = tot + weighted_ce_2d(full_up[n:n + 1].to(dtype), Yq_full[n:n + 1].long(), wgt[n].to(dtype))
    return tot / N


def loss_pattern(bin_up, Yq, dtype=torch.float32):
    """binary focal loss, alpha .25 gamma 2 (loss/losses.py:47-79)."""
    x = bin_up.to(dtype).permute(0, 2, 3, 1).reshape(-1, bin_up.shape[1])
    t = Yq.long().reshape(-1)
    prob = F.softmax(x, dim=-1)
    foc = (1 - prob) ** FOCAL_GAMMA * torch.log(prob)
    a = torch.tensor([FOCAL_ALPHA, 1 - FOCAL_ALPHA], dtype=dtype)
    return F.nll_loss(foc, t, weight=a, ignore_index=IGNORE)


def head_step(Fq, Fs, Ys, label, Yq, Yq_full, wgt, p, gate_mask=None, weight_mask=None,
              backward=True, dtype=torch.float32):
    """Everything the bench boundary covers: forward, both
    losses, and loss_k.backward() (trainer/trainers.py:951-975).  Returns a
    dict with logits, losses and, if backward, grads of Fq, Fs and params.
    dtype=torch.float64 runs the same lines in double: the yardstick for how far fp32 arithmetic in ANY
    order sits from the exact result (tests use it to size the absolute floor of element-wise bounds)."""
    Fq = Fq.detach().to(dtype).requires_grad_(backward)
    Fs = Fs.detach().to(dtype).requires_grad_(backward)
    pp = {k: v.detach().clone().to(dtype).requires_grad_(backward) for k, v in p.items()}
    gate_mask = None if gate_mask is None else gate_mask.to(dtype)
    weight_mask = None if weight_mask is None else weight_mask.to(dtype)
    full, binl, Wn = head_forward(Fq, Fs, Ys, label, pp, gate_mask, weight_mask)
    size = Yq_full.shape[-2:]
    full_up = upsample(full, size)
    bin_up = upsample(binl.to(dtype), size)
    lk = loss_knowledge(full_up, Yq_full, wgt, dtype)
    lp = loss_pattern(bin_up, Yq.reshape(-1, *size), dtype)
    out = dict(logits_full=full.detach(), logits_binary=binl.detach(), weights=Wn.detach(),
               full_up=full_up.detach(), bin_up=bin_up.detach(),
               loss_k=lk.detach(), loss_p=lp.detach())
    if backward:
        lk.backward()
        out["dFq"] = Fq.grad
        out["dFs"] = Fs.grad
        for k, v in pp.items():
            out["d:" + k] = v.grad
    return out


# --------------------------------------------------------------------------
# M1 / M2: metrics (utils/metrics.py)
# --------------------------------------------------------------------------
def confusion_matrix(label_true: np.ndarray, pred: np.ndarray, K: int) -> np.ndarray:
    """mat[t,p] += 1 for 0 <= t < K  (utils/metrics.py:59-67). int64 [K,K]."""
    t = label_true.reshape(-1)
    q = pred.reshape(-1)
    ok = (t >= 0) & (t < K)
    return np.bincount(K * t[ok].astype(np.int64) + q[ok], minlength=K * K).reshape(K, K)


def full_iou_from_mat(mat: np.ndarray, eps=1e-8):
    """(utils/metrics.py:69-76)"""
    hist = mat.astype(float)
    d = np.diag(hist)
    den = np.clip(hist.sum(1) + hist.sum(0) - d, eps, np.inf)
    iu = d / den
    return float(np.nanmean(iu)), iu


def binary_iou_counts(label_true: np.ndarray, pred: np.ndarray):
    """Per sample (skip, inter, union) (utils/metrics.py:10-31).
    label_true, pred: [N,H,W] ints."""
    out = []
    for t, q in zip(label_true, pred):
        if t.sum() == 0:
            out.append((True, 0, 0))
            continue
        s = t + q
        out.append((False, int((s == 2).sum()), int(((s == 1) | (s == 2)).sum())))
    return out


class BinaryIOUOracle:
    """dict-insertion-order accumulation (utils/metrics.py:4-50)."""

    def __init__(self):
        self.inter, self.union = {}, {}

    def update(self, label_true, pred, cls_ids):
        for (skip, i, u), c in zip(binary_iou_counts(label_true, pred), cls_ids):
            if skip:
                continue
            c = int(c)
            self.inter[c] = self.inter.get(c, 0.0) + float(i)
            self.union[c] = self.union.get(c, 0.0) + float(u)

    def mean_iou(self):
        return float(np.mean([self.inter[c] / self.union[c] for c in self.inter]))

    def mean_bf_iou(self):
        return sum(self.inter.values()) / sum(self.union.values())


def min_top2_margin(up: torch.Tensor) -> float:
    """Smallest top-1/top-2 gap over all pixels (SURVEY.md §7 hard parts)."""
    top = up.topk(2, dim=1)[0]
    return float((top[:, 0] - top[:, 1]).min())


# ---- train-mode dropout (nn.Dropout at model/common.py:231 and :274/:326) ----------------------------
# The reference draws its masks from torch's generator, so no implementation can reproduce its stream;
# what it fixes is the distribution: multiplier 0 with probability p, 1/(1-p) otherwise.  The library's
# stream is Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; Random123), restated here and pinned by
# Random123's published known-answer vectors (tests/test_oracle_golden.py).
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counter (c0..c3) and key (k0, k1) as uint32 arrays / ints -> four uint32 arrays (as uint64)."""
    c0, c1, c2, c3 = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    k0, k1, m32, s32 = np.uint64(k0), np.uint64(k1), np.uint64(0xFFFFFFFF), np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> s32) ^ c1 ^ k0) & m32, p1 & m32, ((p0 >> s32) ^ c3 ^ k1) & m32, p0 & m32
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c0, c1, c2, c3


def dropout_mask(seed: int, stream_id: int, p: float, n: int) -> np.ndarray:
    """include/denet_hip.h `denet_dropout_mask`: key = seed, counter = (i // 4, stream_id, 0, 0), element i takes
    output word i % 4; dropped iff word < round(p * 2^32); kept elements carry float32(1 / (1 - p))."""
    if not 0.0 < p < 1.0:
        return np.ones(n, dtype=np.float32)
    groups = (n + 3) // 4
    idx = np.arange(groups, dtype=np.uint64)
    zero = np.zeros(groups, dtype=np.uint64)
    r = np.stack(philox4x32_10(idx, zero + np.uint64(stream_id), zero, zero, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF),
                 axis=1).reshape(-1)[:n]
    thresh = min(int(float(np.float32(p)) * 4294967296.0 + 0.5), 4294967295)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(r < np.uint64(thresh), np.float32(0.0), scale).astype(np.float32)
