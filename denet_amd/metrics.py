"""FullIOU / BinaryIOU with the reference's interface (utils/metrics.py:4-88)
on device-side integer counting (kernels M1/M2).

Differences from the reference, all invisible in the results:
  * predictions never leave the device: ``update`` takes the logits either
    already up-sampled ([N,K,H,W], what the reference's trainer passes) or at
    feature resolution ([N,K,h,w]); the latter are up-sampled on the fly inside
    the kernel with ATen's bilinear arithmetic;
  * the confusion matrix is an int64 device tensor until it is read (``.mat``
    returns the same float64 numpy array the reference keeps);
  * BinaryIOU keeps per-sample (sum_true, inter, union) triples on the device
    and folds them into the per-class dictionaries -- in first-seen order, as
    the reference's dict does -- when a result is read, so ``update`` does not
    synchronise;
  * ``all_reduce`` merges the counts of episode-sharded ranks:
    integer sums, hence bit-identical for any number of GPUs.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class FullIOU(object):
    def __init__(self, num_classes=21):
        self.num_classes = num_classes
        self._mat = None               # int64 [K,K] on the device
        self.cls_iou = None

    def update(self, label_trues, label_preds):
        """label_trues [N,H,W] int, label_preds [N,K,H,W] or [N,K,h,w] float (device)."""
        n = self.num_classes
        if label_preds.shape[1] != n:
            raise ValueError("label_preds has %d channels, num_classes is %d" % (label_preds.shape[1], n))
        if self._mat is None or self._mat.shape[0] != n:
            self._mat = torch.zeros(n, n, dtype=torch.int64, device=label_preds.device)
        yt = label_trues.reshape(-1, *label_trues.shape[-2:]).to(label_preds.device)
        ops.confmat_update(self._mat, yt, label_preds.detach())
        self.cls_iou = None

    def update_counts(self, mat_increment):
        """Add an integer [K,K] matrix of counts (what one `update` contributes)."""
        inc = torch.as_tensor(mat_increment, dtype=torch.int64)
        if self._mat is None:
            self._mat = torch.zeros(self.num_classes, self.num_classes, dtype=torch.int64, device=inc.device)
        self._mat += inc.to(self._mat.device)
        self.cls_iou = None

    @property
    def mat(self):
        """float64 numpy [K,K], as utils/metrics.py:63-67 keeps it (None before the first update)."""
        return None if self._mat is None else self._mat.cpu().numpy().astype(np.float64)

    def counts(self):
        """Exact integer matrix (numpy int64)."""
        return self._mat.cpu().numpy()

    def all_reduce(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self._mat, op=dist.ReduceOp.SUM, group=group)
        self.cls_iou = None

    def mean_iou(self, eps=1e-8):
        hist = self.mat
        numerator = np.diag(hist)
        denominator = hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist)
        denominator = np.clip(denominator, eps, np.inf)
        iu = numerator / denominator
        self.cls_iou = {cls_id: iou for cls_id, iou in enumerate(iu)}
        return np.nanmean(iu)

    def class_iou(self):
        if self.cls_iou is None:
            self.mean_iou()
        return self.cls_iou

    def mean_subclasses_iou(self, subclasses):
        mean = [v if k in subclasses else 0 for k, v in self.class_iou().items()]
        return sum(mean) / len(mean)

    def subclasses_iou(self, subclasses):
        return {k: v for k, v in self.class_iou().items() if k in subclasses}


class BinaryIOU(object):
    def __init__(self):
        self._inter = dict()
        self._union = dict()
        self._pending = []             # (counts [N,3] int64 device, cls_ids numpy, order keys numpy)
        self._seen = 0                 # samples seen so far on this rank (default ordering key)
        self._first = dict()           # class -> ordering key of its first non-skipped sample
        self.cls_iou = None

    def update(self, label_trues, label_pred, cls_ids, order_keys=None):
        """label_trues [N,H,W] {0,1}, label_pred [N,2,H,W] or [N,2,h,w] logits, cls_ids [N].
        order_keys [N] (optional, episode-sharded runs): each sample's position in the
        single-process update order; see all_reduce."""
        yt = label_trues.reshape(-1, *label_trues.shape[-2:]).to(label_pred.device)
        self.update_counts(ops.binary_counts(yt, label_pred.detach()), cls_ids, order_keys)

    def update_counts(self, counts, cls_ids, order_keys=None):
        """counts int64 [N,3] = (sum_true, inter, union) per sample, as kernel M2 returns them."""
        if isinstance(cls_ids, torch.Tensor):
            cls_ids = cls_ids.detach().cpu().numpy()
        n = counts.shape[0]
        keys = np.arange(self._seen, self._seen + n) if order_keys is None else np.asarray(order_keys).reshape(-1)
        self._pending.append((counts, np.asarray(cls_ids).reshape(-1).copy(), keys))
        self._seen += n
        self.cls_iou = None

    def _fold(self):
        for counts, cls_ids, keys in self._pending:
            rows = counts.cpu().tolist() if isinstance(counts, torch.Tensor) else np.asarray(counts).tolist()
            for (st, inter, union), c, key in zip(rows, cls_ids.tolist(), keys.tolist()):
                if st == 0:            # no foreground in the mask: skipped (utils/metrics.py:29-30)
                    continue
                if c not in self._inter:
                    self._inter[c] = float(inter)
                    self._union[c] = float(union)
                    self._first[c] = int(key)
                else:
                    self._inter[c] += float(inter)
                    self._union[c] += float(union)
        self._pending = []

    @property
    def inter(self):
        self._fold()
        return self._inter

    @property
    def union(self):
        self._fold()
        return self._union

    def all_reduce(self, num_classes, group=None):
        """Merge episode-sharded ranks: sum inter/union per class, min of the first-seen global
        index, then rebuild the dictionaries in global first-seen order so that mean_iou's
        float64 summation order equals a single-process run (utils/metrics.py:34-36)."""
        import torch.distributed as dist
        self._fold()
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return
        dev = self._pending_device()
        big = np.iinfo(np.int64).max
        sums = torch.zeros(2, num_classes, dtype=torch.int64)
        first = torch.full((num_classes,), big, dtype=torch.int64)
        for c in self._inter:
            sums[0, c] = int(self._inter[c])
            sums[1, c] = int(self._union[c])
            first[c] = self._first[c]
        sums, first = sums.to(dev), first.to(dev)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(first, op=dist.ReduceOp.MIN, group=group)
        sums, first = sums.cpu().numpy(), first.cpu().numpy()
        order = [int(c) for c in np.argsort(first, kind="stable") if first[c] != big]
        self._inter = {c: float(sums[0, c]) for c in order}
        self._union = {c: float(sums[1, c]) for c in order}
        self._first = {c: int(first[c]) for c in order}
        self.cls_iou = None

    def _pending_device(self):
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def mean_iou(self):
        inter, union = self.inter, self.union
        self.cls_iou = {cls_id: inter[cls_id] / union[cls_id] for cls_id in inter.keys()}
        return float(np.mean(list(self.cls_iou.values())))

    def class_iou(self):
        if self.cls_iou is None:
            self.mean_iou()
        return self.cls_iou

    def mean_bf_iou(self):
        inter, union = self.inter, self.union
        inter_sum = sum([inter[cls_id] for cls_id in inter.keys()])
        union_sum = sum([union[cls_id] for cls_id in union.keys()])
        return inter_sum / union_sum
