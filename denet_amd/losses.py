"""Losses with the reference's names, arguments and error behaviour
(loss/losses.py:5-79, loss/__init__.py:7-27), on the fused HIP kernels K4/K5.

When the logits are smaller than the target (the DENet case: 41x41 logits,
321x321 labels) the reference first materialises the bilinear up-sampling
(loss/losses.py:34-35); here the up-sampling happens inside the loss kernel and
the [N,K,H,W] tensor is never written.  Equal sizes go through the same kernel
(the bilinear map is then the identity).
"""
from __future__ import annotations

import functools

import torch
import torch.nn.functional as F

from . import ops


def _match_sizes(input, target):
    """The reference's size handling (loss/losses.py:20-37) -> (input, target [N,H,W])."""
    n, c, h, w = input.size()
    if target.dim() != 3:
        target = target.squeeze(1)
    nt, ht, wt = target.size()
    if h > ht and w > wt:
        # label up-sampling branch (dead on the DENet path, trainer/trainers.py:954-956 up-samples
        # the logits first): nearest resize of an integer map, stock PyTorch on the device.
        target = F.interpolate(target.unsqueeze(1).float(), size=(h, w), mode="nearest").long().squeeze(1)
    elif h < ht and w < wt:
        pass                                   # bilinear up-sampling is fused into the kernel
    elif h != ht and w != wt:
        raise Exception("Only support upsampling")
    return input, target


def _remap_ignore(target, ignore_index):
    """The kernels narrow labels to bytes and treat 255 as "ignore" (the reference's default and only value,
    loss/losses.py:19,47).  Any other ignore_index is mapped onto 255 first; class ids are < K <= 128, so 255
    never names a class."""
    if ignore_index == ops.IGNORE_INDEX:
        return target
    return torch.where(target == ignore_index, torch.full_like(target, ops.IGNORE_INDEX), target)


def _check_reduction(reduction):
    if reduction not in ('mean', 'sum', 'none'):
        raise ValueError("{} is not a valid value for reduction".format(reduction))


def cross_entropy2d(input, target, weight=None, ignore_index=255, reduction='mean'):
    """loss/losses.py:19-44.  'mean' and 'sum' run the fused up-sampling + CE kernel; 'none' (a per-pixel map,
    which no caller in the reference asks for) materialises the up-sampled logits with stock device ops."""
    _check_reduction(reduction)
    input, target = _match_sizes(input, target)
    n, c = input.shape[:2]
    if reduction == 'none':
        if input.shape[-2:] != target.shape[-2:]:
            input = F.interpolate(input, size=target.shape[-2:], mode="bilinear", align_corners=True)
        flat = input.permute(0, 2, 3, 1).contiguous().view(-1, c)
        w = None if weight is None else weight.to(input.device, input.dtype)
        return F.cross_entropy(flat, target.reshape(-1), weight=w, ignore_index=ignore_index, reduction='none')
    wgt = None
    if weight is not None:
        wgt = weight.to(input.device, torch.float32).reshape(1, c).expand(n, c).contiguous()
    return ops.UpsampleWCEFunction.apply(input.float(), _remap_ignore(target.long(), ignore_index), wgt, True,
                                         reduction)


def binary_focal_loss(input, target, alpha=0.25, gamma=2, ignore_index=255, reduction='mean'):
    """loss/losses.py:47-79 (forward only on the fused kernel, like every use of it in the reference;
    reduction='none' goes through stock device ops and is differentiable)."""
    _check_reduction(reduction)
    input, target = _match_sizes(input, target)
    if input.shape[1] != 2:
        raise ValueError("binary_focal_loss expects 2 channels, got %d" % input.shape[1])
    if reduction == 'none':
        if input.shape[-2:] != target.shape[-2:]:
            input = F.interpolate(input, size=target.shape[-2:], mode="bilinear", align_corners=True)
        flat = input.permute(0, 2, 3, 1).contiguous().view(-1, 2)
        probs = F.softmax(flat, dim=-1)
        w = torch.tensor([alpha, 1 - alpha], device=input.device, dtype=flat.dtype)
        return F.nll_loss((1 - probs) ** gamma * torch.log(probs), target.reshape(-1), weight=w,
                          ignore_index=ignore_index, reduction='none')
    return ops.UpsampleFocalFunction.apply(input.float(), _remap_ignore(target.long(), ignore_index), float(alpha),
                                           float(gamma), reduction)


def loss_general(input, target):
    return cross_entropy2d(input.float(), target.long())


def loss_denet(input, target, mode, weight=None):
    """loss/losses.py:9-16."""
    if mode == 'knowledge':
        loss = cross_entropy2d(input.float(), target.long(), weight=weight)
    elif mode == 'pattern':
        loss = binary_focal_loss(input.float(), target.long())
    else:
        raise ValueError("mode should be either `knowledge` or `pattern`, but found %s" % mode)
    return loss


def loss_denet_batch(Yq_full_pre, Yq_full, Yq_pre, Yq, weight):
    """What DENet_loss_wrapper computes (trainer/utils.py:264-281) in two launches instead of a
    Python loop over episodes: loss_k = mean_n CE(pre[n], tar[n], weight[n]); loss_p = focal."""
    n, c = Yq_full_pre.shape[:2]
    _, tk = _match_sizes(Yq_full_pre, Yq_full)
    wgt = weight.to(Yq_full_pre.device, torch.float32).reshape(n, c).contiguous() if weight is not None else None
    loss_k = ops.UpsampleWCEFunction.apply(Yq_full_pre.float(), tk.long(), wgt, False)
    _, tp = _match_sizes(Yq_pre, Yq.reshape(-1, *Yq.shape[-2:]))
    loss_p = ops.UpsampleFocalFunction.apply(Yq_pre.float().detach(), tp.long(), 0.25, 2.0)
    return loss_k, loss_p


key2loss = {
    'loss_general': loss_general,
    'loss_denet': loss_denet,
}


def get_loss(loss_dict):
    """loss/__init__.py:13-27."""
    if loss_dict is None:
        print("Using default cross entropy loss")
        return cross_entropy2d
    loss_name = loss_dict['name']
    loss_params = {k: v for k, v in loss_dict.items() if k != 'name'}
    if loss_name not in key2loss:
        raise NotImplementedError('Loss {} not implemented'.format(loss_name))
    print('Using loss: {} with params {}'.format(loss_name, loss_params))
    return functools.partial(key2loss[loss_name], **loss_params)
