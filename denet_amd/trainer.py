"""DENetTrainer: the reference's trainer for the head (trainer/trainers.py:876-1141,
wrappers trainer/utils.py:251-303) restated on the HIP path.

Kept from the reference: the config layout (`data` / `model` / `train` or
`test` sections of one YAML), `step()`'s return order
``(Yq_pre, Yq_full_pre, loss, loss_k, loss_p)`` in train/val and
``(Yq_pre, Yq_full_pre)`` in test mode, train()/eval() switching, the `lambda`
re-weighting of the novel class (1-way only), only ``loss_k`` back-propagated,
per-way BinaryIOU updates for way > 1, the checkpoint dictionary
(trainer/trainers.py:174-183) and `fix_randomness` for testing.

`step()` returns `Yq_pre [N*way,2,H,W]` and `Yq_full_pre [N,K,H,W]` up-sampled to the label resolution,
as the reference does (trainer/trainers.py:954-956,1003-1005); the loss itself never reads those
tensors (the loss kernels up-sample on the fly from the low-res logits).  `materialize_upsampled: false`
in the `train` / `test` section is the opt-out for loops that only feed the metrics: the outputs then stay
at feature resolution and the metric kernels up-sample on the fly as well.

Changed on purpose: tensorboard / matplotlib output, the PIL datasets and the ImageNet backbone are
outside this path -- episodes come from `data.*_dataset: synthetic` and the backbone may be the stub
`SyntheticBackbone` (`model.backbone: synthetic`).

With WORLD_SIZE > 1 (torchrun) every rank owns a contiguous block of each episode batch (batch sizes
must be multiples of the world size: `loss_k` is a mean over the local episodes, so only equal shards make
the averaged gradient the gradient of the global batch); gradients are averaged with one flat all-reduce
per step and the metric counts are summed once at the end (denet_amd/sharding.py).  The running B-IOU /
F-IOU that train() prints every `log_steps` are rank 0's shard only (labelled so); train_val() / test()
print all-reduced values.  BatchNorm of the conv stack normalises with per-shard batch statistics unless
`train.sync_bn: true` converts it to SyncBatchNorm; either way the running buffers are averaged over the
ranks before every validation pass and checkpoint (sharding.sync_buffers), so all ranks evaluate, and rank 0
saves, one model.
"""
from __future__ import annotations

import functools
import math
import os
import random
import time
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import ops, sharding
from .head import DENet, freeze_weights, get_architecture, key2arch
from .losses import get_loss, loss_denet_batch
from .metrics import BinaryIOU as bIOU, FullIOU as fIOU
from .schedulers import get_scheduler, tick
from .synthetic import EpisodeShape, make_episodes


# ---- small host utilities (trainer/utils.py:20-118) ---------------------------------------
class Timer(object):
    def __init__(self):
        self.start = time.time()

    def measure(self, p=1):
        x = (time.time() - self.start) / max(p, 1e-12)
        return '%dh %02dm %02ds' % (x // 3600, x % 3600 // 60, x % 60)


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


# ---- the stub feature producer and the synthetic episode stream ---------------------------
class SyntheticBackbone(nn.Module):
    """Frozen stand-in for the dilated ResNet-50 layer3 output: images [*,3,H,W] ->
    {'layer3': [*,1024,h,w]} with h = (H-1)//8 + 1 (321 -> 41, as model/backbone/backbones.py:177-179)."""

    def __init__(self, seed=0):
        super().__init__()
        self.conv = nn.Conv2d(3, 1024, kernel_size=9, stride=8, padding=4)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            self.conv.weight.copy_(torch.randn(self.conv.weight.shape, generator=g) * 0.05)
            self.conv.bias.zero_()

    def forward(self, x):
        return {'layer3': torch.relu(self.conv(x))}


class SyntheticEpisodes(object):
    """Batches with the tensor contract of dataset/voc_sbd.py:501-551:
    (Is, Ys, Iq, Yq, cls_idx, Ys_full, Yq_full).  Deterministic per batch index."""

    def __init__(self, num_batches, batch_size, way, shot, K, img_size, seed=0, split='train'):
        self.num_batches, self.batch_size = num_batches, batch_size
        self.way, self.shot, self.K, self.img = way, shot, K, img_size
        self.seed, self.split = seed, split

    def __len__(self):
        return self.num_batches

    def batch(self, idx):
        N, way, shot, K, S = self.batch_size, self.way, self.shot, self.K, self.img
        shape = EpisodeShape("syn", N=N, way=way, shot=shot, K=K, c=1, h=1, w=1, H=S, W=S,
                             config_index=self.seed)
        ep = make_episodes(shape, seed_offset=7919 * (idx + 1), ignore_rows=8 if self.split == 'train' else 0)
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        Is = torch.randn(N, way, shot, 3, S, S, generator=g)
        Iq = torch.randn(N, 3, S, S, generator=g)
        Ys_full = ep["Ys"] * ep["label"].reshape(N, way, 1, 1, 1)
        return Is, ep["Ys"], Iq, ep["Yq"], ep["label"], Ys_full, ep["Yq_full"]

    def __iter__(self):
        for i in range(self.num_batches):
            yield self.batch(i)


key2loader = {'synthetic': SyntheticEpisodes}


def get_loader(name):
    if name not in key2loader:
        raise NotImplementedError('Loader {} not implemented (only synthetic episodes are in scope)'.format(name))
    return key2loader[name]


# ---- wrappers (trainer/utils.py:251-303) --------------------------------------------------
def DENet_loss_wrapper(loss_func, arch):
    """Given (Yq_pre, Yq, Ys_pre, Ys, Yq_full_pre, Yq_full, Ys_full_pre, Ys_full, weight) return
    (knowledge loss, pattern loss).  The per-episode Python loop of the reference
    (trainer/utils.py:270-277) is one kernel launch here when `loss_func` is this package's
    loss_denet; any other callable is applied episode by episode as the reference does."""
    if arch is None:
        raise ValueError('Arch name must specify!')
    if arch not in key2arch:
        raise NotImplementedError('Architecture {} not implemented'.format(arch))
    from .losses import loss_denet
    base = loss_func.func if isinstance(loss_func, functools.partial) else loss_func
    extra = loss_func.keywords if isinstance(loss_func, functools.partial) else {}

    # the package's own loss_denet with the reference's defaults: DENetTrainer.step may then run forward +
    # up-sampling + this wrapper as the fused head step (same arithmetic, one autograd node)
    fusable = base is loss_denet and not extra

    @functools.wraps(base)
    def wrapper(*args, **kwargs):
        if arch not in ['DENet']:
            raise ValueError('Architecture {} not supported'.format(arch))
        Yq_pre, Yq, _, _, Yq_full_pre, Yq_full, _, _, weight = args
        if base is loss_denet and not extra and not kwargs:
            return loss_denet_batch(Yq_full_pre, Yq_full, Yq_pre, Yq, weight)
        loss_k = 0.0
        batch_size = Yq_full_pre.shape[0]
        for pre, tar, w in zip(Yq_full_pre.chunk(batch_size, dim=0), Yq_full.chunk(batch_size, dim=0), weight):
            loss_k += loss_func(pre, tar, weight=w, mode='knowledge')
        loss_k /= batch_size
        loss_p = loss_func(Yq_pre, Yq, mode='pattern')
        return loss_k, loss_p

    wrapper.fused_head_step = fusable
    return wrapper


def DENet_forward_wrapper(model, arch, *args, **kwargs):
    """Given (Is, Ys, Iq, Yq, cls_id, Ys_full, Yq_full) return (Yq_pre, Ys_pre, Yq_full_pre, Ys_full_pre)."""
    if arch is None:
        raise ValueError('Arch name must specify!')
    if arch not in key2arch:
        raise NotImplementedError('Architecture {} not implemented'.format(arch))
    if arch in ['DENet']:
        Is, Ys, Iq, Yq, cls_id, Ys_full, Yq_full = args
        outs = model(Is, Ys, Iq, cls_id, **kwargs)
        return outs[1], None, outs[0], None
    raise ValueError('Architecture {} not supported'.format(arch))


def nway_fusion_update(b_iou, pre, Yq, cls_idx, way, order_base=None):
    """The n-way test fusion of the class-agnostic trainers (PatternTrainer.test,
    trainer/trainers.py:838-866): `pre` (N, 2*way, H, W) = the `way` per-way binary predictions
    concatenated along the channels, fused by process_nway into (N, way, 2, H, W), then one
    BinaryIOU.update per way with that way's query mask Yq[:, i] and class ids cls_idx[:, i].
    order_base: global position of this batch's first (way 0) update in a sharded run (see
    BinaryIOU.all_reduce); None in a single process."""
    from .operations import process_nway
    fused = process_nway(pre, way)
    N = pre.shape[0]
    for i in range(way):
        keys = None if order_base is None else order_base + i * N + np.arange(N)
        cls = torch.select(torch.as_tensor(cls_idx), 1, i)
        b_iou.update(torch.select(Yq, 1, i), torch.select(fused, 1, i).contiguous(), cls.cpu().numpy(), order_keys=keys)
    return fused


def _make_optimizer(name, params, **cfg):
    table = {'sgd': torch.optim.SGD, 'adam': torch.optim.Adam, 'adamw': torch.optim.AdamW,
             'rmsprop': torch.optim.RMSprop, 'adagrad': torch.optim.Adagrad}
    if name.lower() not in table:
        raise NotImplementedError('Optimizer {} not implemented'.format(name))
    return table[name.lower()](params, **cfg)


class DENetTrainer(object):
    def __init__(self, train_config=None, test_config=None):
        if not (train_config is None and test_config is not None or
                train_config is not None and test_config is None):
            raise ValueError('Only support one of train config or test config.')
        cfg = train_config or test_config
        self.rank, self.world, self.local_rank = sharding.init_from_env()
        if not torch.cuda.is_available():
            raise ValueError('DENetTrainer needs a GPU: the HIP head has no CPU path')
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.model_cfg, self.data_cfg = cfg['model'], cfg['data']
        self.max_num_classes = self.model_cfg['maximum_num_classes']
        model_kwargs = deepcopy(self.model_cfg)
        arch = model_kwargs.pop('arch')
        if model_kwargs.get('backbone') == 'synthetic':
            model_kwargs['backbone'] = SyntheticBackbone()
        self.model = get_architecture(arch)(**model_kwargs).to(self.dev)
        self.shot = self.data_cfg['shot']
        self.materialize = bool(cfg.get('train', cfg.get('test', {})).get('materialize_upsampled', True))
        self.logdir = None
        self.best_bIOU, self.best_step, self.start_epoch = 0, 1, 1
        seed = self.data_cfg.get('seed', 0)
        img = self.data_cfg['img_size']

        if train_config is not None:
            self.train_cfg = train_config['train']
            self.testing = False
            self.way = 1                                   # training is 1-way only (dataset/voc_sbd.py:286-287)
            if self.world > 1 and self.train_cfg.get('sync_bn', False):
                self.model = nn.SyncBatchNorm.convert_sync_batchnorm(self.model)
            opt = deepcopy(self.train_cfg['optimizer'])
            self.optimizer = _make_optimizer(opt.pop('name'), [p for p in self.model.parameters() if p.requires_grad],
                                             **opt)
            self.scheduler = get_scheduler(self.optimizer, scheduler_dict=self.train_cfg.get('scheduler', None))
            self.bucket = sharding.GradBucket(self.model.parameters())
            self.loss_func = DENet_loss_wrapper(get_loss(self.train_cfg['loss']), arch=arch)
            self.num_steps = self.train_cfg['num_steps']
            self.log_steps = self.train_cfg.get('log_steps', 10)
            self.val_steps = self.train_cfg.get('val_steps', 0)
            self.lambda_ = self.train_cfg.get('lambda', 1.0)
            bs = self.data_cfg['batch_size']
            self._check_shardable(bs, self.data_cfg.get('batch_size_val') or bs)
            self.train_loader = get_loader(self.data_cfg['train_dataset'])(
                self.data_cfg.get('batches_per_epoch', self.num_steps), bs, 1, self.shot, self.max_num_classes, img,
                seed=seed, split='train')
            self.val_loader = get_loader(self.data_cfg.get('val_dataset', 'synthetic'))(
                self.data_cfg.get('val_batches', 2), self.data_cfg.get('batch_size_val') or bs, 1, self.shot,
                self.max_num_classes, img, seed=seed + 1, split='val')
            self.num_epoch = math.ceil(self.num_steps / len(self.train_loader))
            self.b_iou, self.f_iou = bIOU(), fIOU(num_classes=self.max_num_classes)
            self.meters = {k: AverageMeter() for k in ('loss', 'loss_k', 'loss_p')}
            self.logdir = self.train_cfg.get('logdir')
            if self.train_cfg.get('resume'):
                # continue a run from its latest / best checkpoint (trainer/trainers.py:71-80,185-195: model and
                # scheduler are restored, the optimizer is not)
                self._load_checkpoints(self.train_cfg['resume'], is_testing=False)
        else:
            self.test_cfg = test_config['test']
            self.testing = True
            self.way = self.data_cfg['way']
            self._check_shardable(self.data_cfg['batch_size'])
            self.test_loader = get_loader(self.data_cfg['dataset'])(
                self.data_cfg.get('num_batches', 50), self.data_cfg['batch_size'], self.way, self.shot,
                self.max_num_classes, img, seed=seed, split='test')
            ckpt = self.test_cfg.get('ckpt_file')
            if ckpt:
                self._load_checkpoints(ckpt, is_testing=True)

    def _check_shardable(self, *batch_sizes):
        """An empty or unequal shard would leave one rank out of a collective (a hang) or change the mean
        the all-reduce forms: refuse it up front, on every rank alike."""
        for bs in batch_sizes:
            if bs % self.world:
                raise ValueError('batch size %d is not a multiple of the world size %d' % (bs, self.world))

    # ---- checkpoints: the reference's dictionary (trainer/trainers.py:174-195) -------------
    def _save_checkpoints(self, iter_idx, path):
        if self.rank != 0:
            return
        torch.save({'best_step': self.best_step, 'epoch': iter_idx, 'best_bIOU': self.best_bIOU,
                    'model': self.model.state_dict(), 'optimizer': self.optimizer.state_dict(),
                    'scheduler': self.scheduler.state_dict()}, path)

    def _load_checkpoints(self, path, is_testing=False):
        d = torch.load(path, map_location=self.dev, weights_only=True)
        self.start_epoch, self.best_step, self.best_bIOU = d['epoch'], d['best_step'], d['best_bIOU']
        self.model.load_state_dict(d['model'])
        if not is_testing and d.get('scheduler'):
            # resume (trainer/trainers.py:192-194); LambdaLR keeps its factor functions out of the state
            self.scheduler.load_state_dict(d['scheduler'])

    # ---- one step (trainer/trainers.py:936-1005) --------------------------------------------
    def step(self, data, train=True, **kwargs):
        self.model.train() if train else self.model.eval()
        data = sharding.shard_batch(tuple(data), self.rank, self.world)
        Is, Ys, Iq, Yq, cls_idx, Ys_full, Yq_full = [t.to(self.dev, non_blocking=True) for t in data]
        fused = not self.testing and getattr(self.loss_func, 'fused_head_step', False)
        if fused:
            # model(...) -> two F.interpolate -> loss_func (trainer/trainers.py:951-968) as one call: the conv
            # stack, then ONE fused head node (DENet.forward_step -> ops.HeadStepFunction, seven launches)
            loss_weighted = torch.ones(Is.shape[0], self.max_num_classes, device=self.dev)
            # re-weight the loss, only support 1-way currently (trainer/trainers.py:962-963)
            loss_weighted[torch.arange(loss_weighted.shape[0], device=self.dev), cls_idx.reshape(-1)] = self.lambda_
            loss_k, loss_p, Yq_full_pre, Yq_pre = self.model.forward_step(Is, Ys, Iq, cls_idx, Yq, Yq_full,
                                                                          loss_weighted)
        else:
            Yq_pre, Ys_pre, Yq_full_pre, Ys_full_pre = DENet_forward_wrapper(
                self.model, self.model_cfg['arch'], Is, Ys, Iq, Yq, cls_idx, Ys_full, Yq_full)
        Yq_pre = Yq_pre.float()
        out_b, out_f = Yq_pre, Yq_full_pre
        if self.materialize:
            out_b = ops.upsample_bilinear(Yq_pre.detach(), Yq.shape[-2:])
            out_f = ops.upsample_bilinear(Yq_full_pre.detach(), Yq.shape[-2:])
        if self.testing:
            return out_b.detach(), out_f.detach()
        if not fused:
            loss_weighted = torch.ones(Yq_full_pre.shape[:2], device=self.dev)
            # re-weight the loss, only support 1-way currently (trainer/trainers.py:962-963)
            loss_weighted[torch.arange(loss_weighted.shape[0], device=self.dev), cls_idx.reshape(-1)] = self.lambda_
            loss_k, loss_p = self.loss_func(Yq_pre, Yq, Ys_pre, Ys, Yq_full_pre, Yq_full, Ys_full_pre, Ys_full,
                                            loss_weighted)
        if train:
            self.bucket.zero()
            loss_k.backward()                       # only the knowledge loss (trainer/trainers.py:972-974)
            self.bucket.all_reduce_mean()
            self.optimizer.step()
        lk, lp = torch.stack([loss_k.detach(), loss_p.detach()]).tolist()     # one host sync
        return out_b.detach(), out_f.detach(), lk + lp, lk, lp

    # ---- loops ------------------------------------------------------------------------------
    def train(self):
        timer = Timer()
        where = 'Train' if self.world == 1 else 'Train (rank-0 shard)'
        for epoch_idx in range(self.start_epoch, self.num_epoch + 1):
            # per-epoch reset of the running metrics / meters and the scheduler tick (trainer/trainers.py:205-209)
            self.b_iou, self.f_iou = bIOU(), fIOU(num_classes=self.max_num_classes)
            for m in self.meters.values():
                m.reset()
            # pinned to the epoch like the reference's scheduler.step(epoch=epoch_idx) (trainer/trainers.py:209):
            # a run resumed in epoch e (last_epoch = e in the checkpoint) must not tick to e + 1
            tick(self.scheduler, epoch_idx)
            for batch_idx, batch in enumerate(self.train_loader, 1):
                step_idx = (epoch_idx - 1) * len(self.train_loader) + batch_idx
                Yq_pre, Yq_full_pre, loss, loss_k, loss_p = self.step(batch, train=True)
                local = sharding.shard_batch(tuple(batch), self.rank, self.world)
                self.b_iou.update(local[3], Yq_pre, local[4].numpy())
                self.f_iou.update(local[6], Yq_full_pre)
                for k, v in (('loss', loss), ('loss_k', loss_k), ('loss_p', loss_p)):
                    self.meters[k].update(v)
                if self.rank == 0 and step_idx % self.log_steps == 0:
                    print('Epoch %d (%2d/%2d), Step %d, %s, Loss %.4f, B-IOU %.4f, F-IOU %.4f, lr %.2e, %s' % (
                        epoch_idx, batch_idx, len(self.train_loader), step_idx, where, loss, self.b_iou.mean_iou(),
                        self.f_iou.mean_iou(), self.scheduler.get_last_lr()[0], timer.measure()))
                if self.val_steps and step_idx % self.val_steps == 0:
                    self.train_val(epoch_idx, step_idx)
                if step_idx == self.num_steps:
                    return

    def train_val(self, iter_idx, step_idx):
        """trainer/trainers.py:1056-1108: no_grad, loss computed, metrics, best/latest checkpoints."""
        meters = {k: AverageMeter() for k in ('loss', 'loss_k', 'loss_p')}
        b_iou, f_iou = bIOU(), fIOU(num_classes=self.max_num_classes)
        with torch.no_grad():
            sharding.sync_buffers(self.model)
            for j, data in enumerate(self.val_loader):
                Yq_pre, Yq_full_pre, loss, loss_k, loss_p = self.step(data, train=False)
                local = sharding.shard_batch(tuple(data), self.rank, self.world)
                n_local = local[3].shape[0]
                lo, _ = sharding.shard_range(data[3].shape[0], self.rank, self.world)
                keys = j * data[3].shape[0] + lo + np.arange(n_local)
                b_iou.update(local[3], Yq_pre, local[4].numpy(), order_keys=keys)
                f_iou.update(local[6], Yq_full_pre)
                for k, v in (('loss', loss), ('loss_k', loss_k), ('loss_p', loss_p)):
                    meters[k].update(v)
        b_iou.all_reduce(self.max_num_classes)
        f_iou.all_reduce()
        mb = b_iou.mean_iou()
        if self.rank == 0:
            print('Epoch %d, Step %d, Eval, Loss %.4f, B-IOU %.4f, F-IOU %.4f' % (
                iter_idx, step_idx, meters['loss'].avg, mb, f_iou.mean_iou()))
        if self.logdir:
            os.makedirs(self.logdir, exist_ok=True)
            if mb > self.best_bIOU:
                self.best_bIOU, self.best_step = mb, step_idx
                self._save_checkpoints(iter_idx, os.path.join(self.logdir, 'best_model.bin'))
            self._save_checkpoints(iter_idx, os.path.join(self.logdir, 'latest_model.bin'))
        return mb, f_iou.mean_iou()

    def fix_randomness(self):
        """trainer/trainers.py:251-260."""
        freeze_weights(self.model)
        random.seed(0)
        torch.manual_seed(0)
        torch.cuda.manual_seed_all(0)

    def test(self):
        """trainer/trainers.py:1110-1141.  Returns (B-IOU, F-IOU) besides printing them."""
        self.fix_randomness()
        b_iou, f_iou = bIOU(), fIOU(num_classes=self.max_num_classes)
        with torch.no_grad():
            for j, batch in enumerate(self.test_loader):
                Yq_pre, Yq_full_pre = self.step(batch, train=False)
                local = sharding.shard_batch(tuple(batch), self.rank, self.world)
                Yq, cls_idx, Yq_full = local[3], local[4], local[6]
                n_glob, n_local = batch[3].shape[0], Yq.shape[0]
                lo, _ = sharding.shard_range(n_glob, self.rank, self.world)
                if self.way > 1:
                    Yq_pre = Yq_pre.reshape([n_local, self.way, *Yq_pre.shape[-3:]])
                    for i in range(self.way):
                        keys = (j * self.way + i) * n_glob + lo + np.arange(n_local)
                        b_iou.update(Yq[:, i], Yq_pre[:, i].contiguous(), cls_idx[:, i].numpy(), order_keys=keys)
                else:
                    keys = j * n_glob + lo + np.arange(n_local)
                    b_iou.update(Yq, Yq_pre, cls_idx.numpy(), order_keys=keys)
                f_iou.update(Yq_full, Yq_full_pre)
        b_iou.all_reduce(self.max_num_classes)
        f_iou.all_reduce()
        mb, mf = b_iou.mean_iou(), float(f_iou.mean_iou())
        if self.rank == 0:
            print('Fold %d, way %d, shot %d, B-IOU: %.4f, F-IOU: %.4f' % (
                self.data_cfg.get('fold', 0), self.way, self.shot, mb, mf))
            print('Class B-IOU: ', sorted(b_iou.class_iou().items()))
            print('Class F-IOU: ', sorted(f_iou.class_iou().items()))
        return mb, mf


key2trainer = {'DENetTrainer': DENetTrainer}


def get_trainer(name):
    """trainer/__init__.py:6-16 (only the DENet trainer is in scope)."""
    print("Using trainer: {}".format(name))
    return key2trainer[name]
