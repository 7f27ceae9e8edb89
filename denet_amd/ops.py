"""Host-side operators over the C ABI (include/denet_hip.h).

Two layers:
  * thin functional wrappers, one per kernel, that allocate outputs with torch
    (device memory is torch's job here, nothing else) and enqueue on torch's
    current stream;
  * torch.autograd.Function classes that give the head, the losses and the
    fused head step the reference's differentiable behaviour.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import call, feat_code, query

IGNORE_INDEX = 255
DROPOUT_GATE, DROPOUT_WEIGHT = _lib.DROPOUT_GATE, _lib.DROPOUT_WEIGHT


def _f32(*shape, device):
    return torch.empty(*shape, dtype=torch.float32, device=device)


def _as_i64(t):
    return t if t.dtype == torch.int64 else t.long()


# --------------------------------------------------------------------------
# K0 / K1
# --------------------------------------------------------------------------
def mask_resize(Ys: torch.Tensor, h: int, w: int):
    """[B,H,W] int64 -> (mask [B,h*w] f32, cnt [B] f32).  model/common.py:189-190."""
    Ys = _as_i64(Ys).contiguous()
    B, H, W = Ys.shape
    mask = _f32(B, h * w, device=Ys.device)
    cnt = _f32(B, device=Ys.device)
    call("denet_mask_resize", Ys, B, H, W, h, w, mask, cnt)
    return mask, cnt


def masked_sum(Fs: torch.Tensor, mask: torch.Tensor):
    """Fs [B,C,h,w] (f32/bf16), mask [B,P] -> S [B,C] f32."""
    Fs = Fs.contiguous()
    B, C = Fs.shape[:2]
    P = Fs.shape[2] * Fs.shape[3]
    S = _f32(B, C, device=Fs.device)
    call("denet_masked_sum_fwd", Fs, feat_code(Fs), mask, B, C, P, S)
    return S


def masked_pool(Fs: torch.Tensor, Ys: torch.Tensor):
    """K0 + K1 in one launch: Fs [B,C,h,w], Ys [B,H,W] int64 -> (mask [B,P], cnt [B], S [B,C])."""
    Fs = Fs.contiguous()
    Ys = _as_i64(Ys).contiguous()
    B, C, h, w = Fs.shape
    H, W = Ys.shape[-2:]
    mask, cnt, S = _f32(B, h * w, device=Fs.device), _f32(B, device=Fs.device), _f32(B, C, device=Fs.device)
    call("denet_masked_pool_fwd", Fs, feat_code(Fs), Ys, B, C, H, W, h, w, mask, cnt, S)
    return mask, cnt, S


def masked_sum_bwd(mask, coef, shape, dtype, chunked=False):
    """dFs = mask x coef (K1b).  chunked: the large-stream writer (denet_masked_sum_bwd_chunked), same results."""
    B, C, h, w = shape
    dFs = torch.empty(B, C, h, w, dtype=dtype, device=mask.device)
    call("denet_masked_sum_bwd_chunked" if chunked else "denet_masked_sum_bwd", mask, coef, B, C, h * w, dFs,
         feat_code(dFs))
    return dFs


# --------------------------------------------------------------------------
# K2
# --------------------------------------------------------------------------
def proto_extend(S, cnt, params, labels, N, way, shot, K, P, gate_mask=None, weight_mask=None, proto=None):
    """-> (Wn [N,K,C], saved (hidden, gate, pool)).  proto: optional [N*way,C] output."""
    W1, b1, W2, b2, Wl, bl, weight = params
    C = S.shape[1]
    dev = S.device
    hidden, gate, pool = _f32(S.shape[0], C, device=dev), _f32(S.shape[0], C, device=dev), _f32(S.shape[0], C, device=dev)
    Wn = _f32(N, K, C, device=dev)
    call("denet_proto_extend_fwd", S, cnt, W1, b1, W2, b2, Wl, bl, weight, labels, gate_mask, weight_mask,
         N, way, shot, K, C, P, hidden, gate, pool, Wn, proto)
    return Wn, (hidden, gate, pool)


def proto_extend_bwd(dWn_part, n_part, S, cnt, saved, params, labels, N, way, shot, K, P,
                     gate_mask=None, weight_mask=None):
    hidden, gate, pool = saved
    W1, b1, W2, b2, Wl, bl, weight = params
    C = S.shape[1]
    B = S.shape[0]
    dev = S.device
    dW1, dW2, dWl = (torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(Wl))
    db1, db2, dbl = (_f32(C, device=dev), _f32(C, device=dev), _f32(C, device=dev))
    dweight = _f32(K, C, device=dev)
    coef = _f32(B, C, device=dev)
    nbytes = query("denet_proto_extend_bwd_workspace", N, way, shot, K, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    call("denet_proto_extend_bwd", dWn_part, n_part, S, cnt, hidden, gate, pool, W1, W2, Wl, labels,
         gate_mask, weight_mask, N, way, shot, K, C, P, dW1, db1, dW2, db2, dWl, dbl, dweight, coef, ws, nbytes)
    return dict(dW1=dW1, db1=db1, dW2=dW2, db2=db2, dWl=dWl, dbl=dbl, dweight=dweight, coef=coef)


# --------------------------------------------------------------------------
# K3
# --------------------------------------------------------------------------
def classify(Fq, Wn, labels=None, way=0, base=True):
    """Fq [N,C,h,w], Wn [N,K,C] -> logits [N,K,h,w] (+ binary [N*way,2,h,w] if labels)."""
    Fq = Fq.contiguous()
    N, C, h, w = Fq.shape
    K = Wn.shape[1]
    logits = _f32(N, K, h, w, device=Fq.device)
    binl = _f32(N * way, 2, h, w, device=Fq.device) if labels is not None else None
    call("denet_classify_fwd", Fq, feat_code(Fq), Wn, labels, N, way, K, C, h * w, 1 if base else 0, logits, binl)
    return logits, binl


def classify_shared(Fq, W):
    Fq = Fq.contiguous()
    N, C, h, w = Fq.shape
    K = W.shape[0]
    logits = _f32(N, K, h, w, device=Fq.device)
    call("denet_classify_shared_fwd", Fq, feat_code(Fq), W.contiguous(), N, K, C, h * w, logits)
    return logits


def binary_logits(logits, labels, way=1, base=True):
    """get_binary_logits (utils/operations.py:313-338): logits [N,K,h,w], labels [N,way] -> [N*way,2,h,w]."""
    logits = logits.float().contiguous()
    N, K, h, w = logits.shape
    labels = _as_i64(labels).contiguous()
    out = _f32(N * way, 2, h, w, device=logits.device)
    call("denet_binary_logits", logits, labels, N, way, K, h * w, 1 if base else 0, out)
    return out


def batch_mean(x):
    """mean over dim 0 (model/common.py:371)."""
    x = x.float().contiguous()
    out = _f32(*x.shape[1:], device=x.device)
    call("denet_batch_mean", x, x.shape[0], x[0].numel(), out)
    return out


def classify_bwd(Fq, Wn, dlog_a, dlog_b=None, scale=None, gmul=None, need_dFq=True, need_dWn=True):
    Fq = Fq.contiguous()
    N, C, h, w = Fq.shape
    K = Wn.shape[1]
    P = h * w
    n_part = query("denet_classify_bwd_parts", P)
    dFq = torch.empty_like(Fq) if need_dFq else None
    dWn_part = _f32(N, n_part, K, C, device=Fq.device) if need_dWn else None
    call("denet_classify_bwd", Fq, feat_code(Fq), Wn, dlog_a, dlog_b, scale, gmul, N, K, C, P, dFq, dWn_part)
    return dFq, dWn_part, n_part


# --------------------------------------------------------------------------
# K4 / K5
# --------------------------------------------------------------------------
def upsample_wce(logits, Yq_full, wgt=None, grad=False, pooled=False, reduction='mean'):
    """-> (loss scalar tensor, per_episode [N], scale [N], dlog_a, dlog_b).  reduction='sum': the loss is the
    weighted numerator summed over the batch and `scale` is None (the slabs are its gradient as they stand)."""
    logits = logits.contiguous()
    N, K, h, w = logits.shape
    Y = _as_i64(Yq_full).contiguous().reshape(N, *Yq_full.shape[-2:])
    H, W = Y.shape[-2:]
    dev = logits.device
    parts = query("denet_upsample_loss_parts", K, h, w, H, W, 1 if grad else 0)
    if parts < 0:
        raise _lib.DenetHipError(f"denet_upsample_loss_parts: {_lib.strerror(parts)}")
    part = _f32(N, parts, 2, device=dev)
    da = torch.empty_like(logits) if grad else None
    db = torch.empty_like(logits) if grad else None
    call("denet_upsample_wce_fwd", logits, Y, wgt, N, K, h, w, H, W, part, da, db)
    if reduction == 'sum':
        return part[:, :, 0].double().sum().float(), None, None, da, db
    out = _f32(2, device=dev)
    per = _f32(N, device=dev)
    scale = _f32(N, device=dev)
    call("denet_loss_finalize", part, None, N, 0, parts, 0, 1 if pooled else 0, out, per, scale)
    return out[0], per, scale, da, db


def upsample_focal(logits_binary, Yq, alpha=0.25, gamma=2.0, reduction='mean'):
    lb = logits_binary.contiguous()
    NB, two, h, w = lb.shape
    assert two == 2
    Y = _as_i64(Yq).contiguous().reshape(NB, *Yq.shape[-2:])
    H, W = Y.shape[-2:]
    parts = query("denet_upsample_loss_parts", 2, h, w, H, W, 0)
    if parts < 0:
        raise _lib.DenetHipError(f"denet_upsample_loss_parts: {_lib.strerror(parts)}")
    part = _f32(NB, parts, 2, device=lb.device)
    call("denet_upsample_focal_fwd", lb, Y, NB, h, w, H, W, alpha, gamma, part)
    if reduction == 'sum':
        return part[:, :, 0].double().sum().float()
    out = _f32(2, device=lb.device)
    call("denet_loss_finalize", None, part, 0, NB, 0, parts, 0, out, None, None)
    return out[1]


def upsample_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) for [N,C,h,w] f32."""
    x = x.contiguous()
    N, C, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    out = _f32(N, C, H, W, device=x.device)
    step = 32768
    xf, of = x.reshape(N * C, h, w), out.reshape(N * C, H, W)
    for i in range(0, N * C, step):
        call("denet_upsample_bilinear", xf[i:i + step], min(step, N * C - i), h, w, H, W, of[i:i + step])
    return out


def combine_slabs(a, b, scale, gmul):
    N = a.shape[0]
    out = torch.empty_like(a)
    call("denet_combine_slabs", a, b, scale, gmul, N, a[0].numel(), out)
    return out


# --------------------------------------------------------------------------
# metrics
# --------------------------------------------------------------------------
def confmat_update(mat, label_trues, preds):
    """mat [K,K] int64 (device) += confusion counts.  preds [N,K,H,W] or low-res [N,K,h,w]."""
    K = mat.shape[0]
    yt = _as_i64(label_trues).contiguous()
    preds = preds.float().contiguous()
    N = preds.shape[0]
    H, W = yt.shape[-2:]
    if tuple(preds.shape[-2:]) == (H, W):
        call("denet_confmat", preds, yt, N, K, H * W, mat)
    else:
        call("denet_confmat_fused", preds, yt, N, K, preds.shape[2], preds.shape[3], H, W, mat)
    return mat


def binary_counts(label_trues, preds2):
    """-> int64 [N,3] = (sum true, inter, union).  preds2 [N,2,H,W] or low-res [N,2,h,w]."""
    yt = _as_i64(label_trues).contiguous()
    preds2 = preds2.float().contiguous()
    N = preds2.shape[0]
    H, W = yt.shape[-2:]
    counts = torch.zeros(N, 3, dtype=torch.int64, device=preds2.device)
    if tuple(preds2.shape[-2:]) == (H, W):
        call("denet_binary_iou_counts", preds2, yt, N, H * W, counts)
    else:
        call("denet_binary_iou_counts_fused", preds2, yt, N, preds2.shape[2], preds2.shape[3], H, W, counts)
    return counts


# --------------------------------------------------------------------------
# autograd: estimator (extend + infer), reference model/common.py:401-426
# --------------------------------------------------------------------------
class EstimatorFunction(torch.autograd.Function):
    """(Fq, Fs, weight, W1, b1, W2, b2, Wl, bl | Ys, labels, masks) -> (logits_full, logits_binary, Wn).

    logits_binary and Wn are returned non-differentiable: the reference only ever
    back-propagates loss_k, which depends on logits_full (trainer/trainers.py:972-974).
    """

    @staticmethod
    def forward(ctx, Fq, Fs, weight, W1, b1, W2, b2, Wl, bl, Ys, labels, gate_mask, weight_mask, base):
        N, way, shot, C, h, w = Fs.shape
        K = weight.shape[0]
        B, P = N * way * shot, h * w
        mask, cnt, S = masked_pool(Fs.contiguous().reshape(B, C, h, w), Ys.reshape(B, *Ys.shape[-2:]))
        params = tuple(t.contiguous() for t in (W1, b1, W2, b2, Wl, bl, weight))
        labels = _as_i64(labels).contiguous()
        Wn, saved = proto_extend(S, cnt, params, labels, N, way, shot, K, P, gate_mask, weight_mask)
        logits, binl = classify(Fq, Wn, labels, way, base)
        ctx.save_for_backward(Fq, mask, cnt, S, Wn, labels, *saved, *params,
                              *(t for t in (gate_mask, weight_mask) if t is not None))
        ctx.has_masks = (gate_mask is not None, weight_mask is not None)
        ctx.dims = (N, way, shot, K, C, h, w)
        ctx.fs_dtype = Fs.dtype
        ctx.mark_non_differentiable(binl, Wn)
        ctx.set_materialize_grads(False)
        return logits, binl, Wn

    @staticmethod
    def backward(ctx, dlogits, dbin, dWn_unused):
        if dbin is not None:
            raise NotImplementedError("gradient through logits_binary is not implemented "
                                      "(the reference never back-propagates loss_p)")
        N, way, shot, K, C, h, w = ctx.dims
        sv = ctx.saved_tensors
        Fq, mask, cnt, S, Wn, labels, hidden, gate, pool, W1, b1, W2, b2, Wl, bl, weight = sv[:16]
        rest = list(sv[16:])
        gate_mask = rest.pop(0) if ctx.has_masks[0] else None
        weight_mask = rest.pop(0) if ctx.has_masks[1] else None
        if dlogits is None:
            return (None,) * 14
        dFq, dWn_part, n_part = classify_bwd(Fq, Wn, dlogits.contiguous().float())
        g = proto_extend_bwd(dWn_part, n_part, S, cnt, (hidden, gate, pool), (W1, b1, W2, b2, Wl, bl, weight),
                             labels, N, way, shot, K, h * w, gate_mask, weight_mask)
        dFs = masked_sum_bwd(mask, g["coef"], (N * way * shot, C, h, w), ctx.fs_dtype)
        dFs = dFs.reshape(N, way, shot, C, h, w)
        return (dFq, dFs, g["dweight"], g["dW1"], g["db1"], g["dW2"], g["db2"], g["dWl"], g["dbl"],
                None, None, None, None, None)


class MaskedSumFunction(torch.autograd.Function):
    """S[b,c] = sum_p mask[b,p] * Fs[b,c,p] (kernel K1) with its rank-1 backward dFs = mask x dS (kernel K1b):
    the differentiable piece behind the stand-alone MaskedAveragePooling / GAM modules (model/common.py:188-246)."""

    @staticmethod
    def forward(ctx, Fs, mask):
        ctx.save_for_backward(mask)
        ctx.fs_shape, ctx.fs_dtype = Fs.shape, Fs.dtype
        return masked_sum(Fs, mask)

    @staticmethod
    def backward(ctx, dS):
        (mask,) = ctx.saved_tensors
        return masked_sum_bwd(mask, dS.contiguous().float(), ctx.fs_shape, ctx.fs_dtype), None


class BinaryLogitsFunction(torch.autograd.Function):
    """get_binary_logits (utils/operations.py:313-338) as a graph node: the forward is kernel K3e; a backward
    through it raises instead of silently cutting the graph (the reference never back-propagates loss_p,
    trainer/trainers.py:972-974, and no gradient kernel exists for it here)."""

    @staticmethod
    def forward(ctx, logits, labels, way, base):
        return binary_logits(logits, labels, way=way, base=base)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("gradient through get_binary_logits is not implemented: the reference only "
                                  "back-propagates loss_k (trainer/trainers.py:972-974)")


class ExtendFunction(torch.autograd.Function):
    """WeightEstimator.extend(mode='training') alone (model/common.py:359-368):
    (Fs, weight, W1, b1, W2, b2, Wl, bl | Ys, labels, masks) -> (Wn [N,K,C], proto [N*way,C])."""

    @staticmethod
    def forward(ctx, Fs, weight, W1, b1, W2, b2, Wl, bl, Ys, labels, gate_mask, weight_mask):
        N, way, shot, C, h, w = Fs.shape
        K = weight.shape[0]
        B, P = N * way * shot, h * w
        mask, cnt, S = masked_pool(Fs.contiguous().reshape(B, C, h, w), Ys.reshape(B, *Ys.shape[-2:]))
        params = tuple(t.contiguous() for t in (W1, b1, W2, b2, Wl, bl, weight))
        labels = _as_i64(labels).contiguous()
        proto = _f32(N * way, C, device=Fs.device)
        Wn, saved = proto_extend(S, cnt, params, labels, N, way, shot, K, P, gate_mask, weight_mask, proto)
        ctx.save_for_backward(mask, cnt, S, labels, *saved, *params,
                              *(t for t in (gate_mask, weight_mask) if t is not None))
        ctx.has_masks = (gate_mask is not None, weight_mask is not None)
        ctx.dims = (N, way, shot, K, C, h, w)
        ctx.fs_dtype = Fs.dtype
        ctx.mark_non_differentiable(proto)
        ctx.set_materialize_grads(False)
        return Wn, proto

    @staticmethod
    def backward(ctx, dWn, dproto_unused):
        N, way, shot, K, C, h, w = ctx.dims
        if dWn is None:
            return (None,) * 12
        sv = ctx.saved_tensors
        mask, cnt, S, labels, hidden, gate, pool, W1, b1, W2, b2, Wl, bl, weight = sv[:14]
        rest = list(sv[14:])
        gate_mask = rest.pop(0) if ctx.has_masks[0] else None
        weight_mask = rest.pop(0) if ctx.has_masks[1] else None
        g = proto_extend_bwd(dWn.contiguous().float(), 1, S, cnt, (hidden, gate, pool),
                             (W1, b1, W2, b2, Wl, bl, weight), labels, N, way, shot, K, h * w, gate_mask, weight_mask)
        dFs = masked_sum_bwd(mask, g["coef"], (N * way * shot, C, h, w), ctx.fs_dtype).reshape(N, way, shot, C, h, w)
        return (dFs, g["dweight"], g["dW1"], g["db1"], g["dW2"], g["db2"], g["dWl"], g["dbl"], None, None, None, None)


class ClassifyFunction(torch.autograd.Function):
    """WeightEstimator.infer (model/common.py:390-399): (Fq [N,C,h,w], Wn [N,K,C] or W [K,C]) -> logits."""

    @staticmethod
    def forward(ctx, Fq, Wn):
        shared = Wn.dim() == 2
        Wn_c = Wn.float().contiguous()
        logits = classify_shared(Fq, Wn_c) if shared else classify(Fq, Wn_c)[0]
        ctx.save_for_backward(Fq, Wn_c)
        ctx.shared = shared
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        Fq, Wn = ctx.saved_tensors
        N = Fq.shape[0]
        Wb = Wn.unsqueeze(0).expand(N, -1, -1).contiguous() if ctx.shared else Wn
        dFq, dWn_part, n_part = classify_bwd(Fq, Wb, dlogits.contiguous().float(),
                                             need_dFq=ctx.needs_input_grad[0], need_dWn=ctx.needs_input_grad[1])
        dWn = None
        if dWn_part is not None:
            K, C = Wb.shape[1:]
            dWn = _f32(N, K, C, device=Fq.device)
            call("denet_sum_parts", dWn_part, N, n_part, K * C, dWn)
            if ctx.shared:
                dWn = dWn.sum(0)
        return dFq, dWn


# --------------------------------------------------------------------------
# autograd: losses on (possibly low-res) logits, reference loss/losses.py
# --------------------------------------------------------------------------
class UpsampleWCEFunction(torch.autograd.Function):
    """cross_entropy2d with the bilinear up-sampling fused in (loss/losses.py:19-44).
    pooled=True: one weighted mean over the batch; False: mean of per-sample means
    (trainer/utils.py:270-277)."""

    @staticmethod
    def forward(ctx, logits, target, wgt, pooled, reduction='mean'):
        need = logits.requires_grad
        loss, per, scale, da, db = upsample_wce(logits, target, wgt, grad=need, pooled=pooled, reduction=reduction)
        if need:
            ctx.has_scale = scale is not None
            ctx.save_for_backward(*((scale,) if scale is not None else ()), da, db)
        return loss

    @staticmethod
    def backward(ctx, gout):
        sv = ctx.saved_tensors
        scale, (da, db) = (sv[0] if ctx.has_scale else None), sv[-2:]
        g = gout.contiguous().float().reshape(1)
        return combine_slabs(da, db, scale, g), None, None, None, None


class UpsampleFocalFunction(torch.autograd.Function):
    """binary_focal_loss with the up-sampling fused in (loss/losses.py:47-79); forward only."""

    @staticmethod
    def forward(ctx, logits_binary, target, alpha, gamma, reduction='mean'):
        return upsample_focal(logits_binary, target, alpha, gamma, reduction)

    @staticmethod
    def backward(ctx, gout):
        raise NotImplementedError("binary_focal_loss backward is not implemented: the reference only "
                                  "logs loss_p and never back-propagates it (trainer/trainers.py:972-974)")


# --------------------------------------------------------------------------
# autograd: the fused head step (T1)
# --------------------------------------------------------------------------
_ws_bytes_cache = {}


def _step_workspace_bytes(*dims):
    """denet_head_step_workspace for a shape, remembered (one ctypes call less per step)."""
    n = _ws_bytes_cache.get(dims)
    if n is None:
        n = query("denet_head_step_workspace", *dims)
        if n < 0:
            raise _lib.DenetHipError(f"denet_head_step_workspace: {_lib.strerror(n)}")
        _ws_bytes_cache[dims] = n
    return n


def _check_step_shapes(Fq, Fs, params, Ys, labels, Yq, Yq_full, wgt, gate_mask, weight_mask):
    """The kernels index raw memory from (N, way, shot, K, C, h, w, H, W) alone: a tensor of another size is an
    out-of-bounds access on the GPU, so the sizes are checked on the host (element counts: views are fine)."""
    N, way, shot, C, h, w = Fs.shape
    K = params[0].shape[0]
    H, W = Ys.shape[-2:]

    def need(name, t, numel, tail=None):
        if t is None:
            return
        if t.numel() != numel or (tail is not None and tuple(t.shape[-len(tail):]) != tuple(tail)):
            raise ValueError(f"head step: {name} has shape {tuple(t.shape)}, expected {numel} elements"
                             + (f" ending in {tuple(tail)}" if tail is not None else ""))
    need("Fq", Fq, N * C * h * w, (C, h, w))
    need("Ys", Ys, N * way * shot * H * W)
    need("labels", labels, N * way)
    # one (H, W) describes all three label maps (the reference's loader crops them alike)
    need("Yq", Yq, N * way * H * W, (H, W))
    need("Yq_full", Yq_full, N * H * W, (H, W))
    need("weight (loss class weights)", wgt, N * K)
    for name, t, shp in zip(("weight", "W1", "b1", "W2", "b2", "Wl", "bl"), params,
                            ((K, C), (C, C), (C,), (C, C), (C,), (C, C), (C,))):
        need(name, t, int(torch.Size(shp).numel()))          # (the gate's 1x1 convolutions hold [C, C, 1, 1])
    need("gate_mask", gate_mask, N * way * shot * C)
    need("weight_mask", weight_mask, N * K * C)


class _WorkspacePool:
    """Step workspaces by (device, stream, bytes): the fused step's scratch slab (~150 MB at C2) is handed from
    one step to the next instead of going through the allocator on every call.  A forward takes a slab, the
    matching backward (or a forward that needs no gradient) puts it back; everything is enqueued on one stream,
    so the next step's kernels are ordered behind the last reader.  A graph that is dropped without a backward
    simply frees its slab."""

    def __init__(self):
        self._free = {}

    def take(self, nbytes, device):
        key = (device.index, _lib.current_stream_handle(device.index), int(nbytes))
        free = self._free.get(key)
        if free:
            return key, free.pop()
        return key, torch.empty(nbytes, dtype=torch.uint8, device=device)

    def give(self, key, ws):
        free = self._free.setdefault(key, [])
        if len(free) < 2:
            free.append(ws)

    def clear(self):
        self._free.clear()


workspace_pool = _WorkspacePool()


class HeadStepFunction(torch.autograd.Function):
    """One call for forward + both losses; backward runs K3b/K2b/K1b from the
    gradient slabs the forward already wrote.  Returns
    (loss_k, loss_p, logits_full, logits_binary); only loss_k is differentiable.
    This is the node behind WeightEstimator.head_step / DENet.forward_step / DENetTrainer.step, i.e. the
    reference's model(...) -> F.interpolate x2 -> loss_func -> loss_k.backward() sequence
    (trainer/trainers.py:951-975) as seven launches."""

    @staticmethod
    def forward(ctx, Fq, Fs, weight, W1, b1, W2, b2, Wl, bl, Ys, labels, Yq, Yq_full, wgt, gate_mask, weight_mask,
                drop=None):
        """drop: (seed, p_gate, p_weight) -- train-mode dropout generated in the kernels; the backward
        regenerates the same multipliers from it."""
        N, way, shot, C, h, w = Fs.shape
        K = weight.shape[0]
        dev = Fq.device
        ctx.drop = _lib.Dropout(*drop) if drop is not None else None
        Fq, Fs = Fq.contiguous(), Fs.contiguous()
        params = tuple(t.contiguous() for t in (weight, W1, b1, W2, b2, Wl, bl))
        labels = _as_i64(labels).contiguous()
        Ys = _as_i64(Ys).contiguous()
        H, W = Ys.shape[-2:]
        Yq = _as_i64(Yq).contiguous() if Yq is not None else None
        Yq_full = _as_i64(Yq_full).contiguous() if Yq_full is not None else None
        _check_step_shapes(Fq, Fs, params, Ys, labels, Yq, Yq_full, wgt, gate_mask, weight_mask)
        need = any(ctx.needs_input_grad[:9])          # (ops.head_step detaches the inputs under no_grad)
        nbytes = _step_workspace_bytes(N, way, shot, K, C, h, w, H, W)
        ws_key, ws = workspace_pool.take(nbytes, dev)
        logits = _f32(N, K, h, w, device=dev)
        binl = _f32(N * way, 2, h, w, device=dev)
        # (both entries are written by the step when both label maps are given: no memset launch then)
        losses = _f32(2, device=dev) if (Yq is not None and Yq_full is not None) else \
            torch.zeros(2, dtype=torch.float32, device=dev)
        call("denet_head_step_train_fwd", Fq, Fs, feat_code(Fq, Fs), Ys, labels, Yq, Yq_full, wgt, *params,
             gate_mask, weight_mask, ctx.drop, N, way, shot, K, C, h, w, H, W, 1 if need else 0, logits, binl, losses,
             ws, nbytes, None)
        ctx.dims = (N, way, shot, K, C, h, w, H, W)
        ctx.have_focal = Yq is not None
        if need:
            ctx.ws, ctx.ws_key = ws, ws_key
        else:
            ctx.ws = None
            workspace_pool.give(ws_key, ws)          # nothing will read it again
        ctx.fs_shape = Fs.shape
        ctx.save_for_backward(Fq, labels, params[1], params[3], params[5],
                              *(t for t in (gate_mask, weight_mask) if t is not None))
        ctx.has_masks = (gate_mask is not None, weight_mask is not None)
        loss_k, loss_p = losses[0], losses[1]
        ctx.mark_non_differentiable(loss_p, logits, binl)
        ctx.set_materialize_grads(False)
        return loss_k, loss_p, logits, binl

    @staticmethod
    def backward(ctx, g_k, g_p, g_l, g_b):
        if g_k is None:
            return (None,) * 17
        if ctx.ws is None:
            raise RuntimeError("HeadStepFunction: the step's workspace went back to the pool after the first "
                               "backward; a second backward through the same step is not supported")
        N, way, shot, K, C, h, w, H, W = ctx.dims
        sv = list(ctx.saved_tensors)
        Fq, labels, W1, W2, Wl = sv[:5]
        rest = sv[5:]
        gate_mask = rest.pop(0) if ctx.has_masks[0] else None
        weight_mask = rest.pop(0) if ctx.has_masks[1] else None
        dev = Fq.device
        dFq = torch.empty_like(Fq)
        dFs = torch.empty(ctx.fs_shape, dtype=Fq.dtype, device=dev)
        dweight = _f32(K, C, device=dev)
        dW1, dW2, dWl = torch.empty_like(W1), torch.empty_like(W2), torch.empty_like(Wl)
        db1, db2, dbl = _f32(C, device=dev), _f32(C, device=dev), _f32(C, device=dev)
        g = g_k.contiguous().float().reshape(1)
        call("denet_head_step_train_bwd", Fq, feat_code(Fq, dFq, dFs), labels, W1, W2, Wl, gate_mask, weight_mask,
             ctx.drop, g,
             N, way, shot, K, C, h, w, H, W, dFq, dFs, dweight, dW1, db1, dW2, db2, dWl, dbl, None,
             1 if ctx.have_focal else 0, ctx.ws, ctx.ws.numel(), None)
        workspace_pool.give(ctx.ws_key, ctx.ws)
        ctx.ws = None
        return (dFq, dFs, dweight, dW1, db1, dW2, db2, dWl, dbl) + (None,) * 8


def head_step(*args):
    """HeadStepFunction.apply for callers that may run under torch.no_grad() (validation steps).  Inside an autograd
    Function's forward the grad mode is always off and `ctx.needs_input_grad` still mirrors the inputs'
    requires_grad, so the node cannot see a surrounding no_grad by itself: it would write the gradient slabs
    nobody reads and keep its workspace out of the pool.  Detached inputs say what the caller means."""
    if not torch.is_grad_enabled():
        args = tuple(a.detach() if isinstance(a, torch.Tensor) else a for a in args)
    return HeadStepFunction.apply(*args)


def head_forward(Fq, Fs, params, Ys, labels, gate_mask=None, weight_mask=None, drop=None):
    """Forward only, no losses: K0..K3e through the fused step's three forward launches (F1 pooling + split-K,
    F2 gate / projection, F3 Wn assembly + classifier + binary logits) instead of the stand-alone kernel chain.
    What WeightEstimator.forward_with_binary runs when no gradient is asked for (test / inference loops,
    trainer/trainers.py:1110-1141).  params = (weight, W1, b1, W2, b2, Wl, bl).  -> (logits_full, logits_binary)."""
    N, way, shot, C, h, w = Fs.shape
    K = params[0].shape[0]
    dev = Fq.device
    Fq, Fs = Fq.contiguous(), Fs.contiguous()
    params = tuple(t.detach().contiguous() for t in params)
    labels = _as_i64(labels).contiguous()
    Ys = _as_i64(Ys).contiguous()
    H, W = Ys.shape[-2:]
    _check_step_shapes(Fq, Fs, params, Ys, labels, None, None, None, gate_mask, weight_mask)
    nbytes = _step_workspace_bytes(N, way, shot, K, C, h, w, H, W)
    ws_key, ws = workspace_pool.take(nbytes, dev)
    logits = _f32(N, K, h, w, device=dev)
    binl = _f32(N * way, 2, h, w, device=dev)
    losses = _f32(2, device=dev)
    call("denet_head_step_train_fwd", Fq, Fs, feat_code(Fq, Fs), Ys, labels, None, None, None, *params,
         gate_mask, weight_mask, _lib.Dropout(*drop) if drop is not None else None, N, way, shot, K, C, h, w, H, W,
         0, logits, binl, losses, ws, nbytes, None)
    workspace_pool.give(ws_key, ws)
    return logits, binl


def dropout_mask(seed, stream_id, p, shape, device):
    """The multipliers the train-mode step generates for (seed, stream id, p), as a tensor
    (denet_dropout_mask): feeds the stand-alone ops' mask pointers with the same dropout."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    call("denet_dropout_mask", seed, stream_id, p, out.numel(), out)
    return out
