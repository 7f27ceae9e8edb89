"""Host-side mirror of the reference's DENet head modules, on the HIP kernels.

Same class names, constructor arguments, sub-module / parameter names (so a
reference ``state_dict`` loads unchanged) and call signatures as

    model/common.py:149-200   MaskedAveragePooling
    model/common.py:211-246   GAM
    model/common.py:249-441   WeightEstimator
    model/common.py:130-146   DeepLabHead (+ RegASPP :61-91), stock PyTorch-ROCm convs
    model/head/denet.py:8-93  DENet

What differs, on purpose:
  * the estimator runs on libdenet_hip.so (no CPU path; CPU tensors are rejected);
  * class labels are read on the device: no ``label.item()`` sync per way
    (model/common.py:328), except in ``support_free`` mode, whose ``record``
    dictionary lives on the host in the reference too;
  * dropout masks (gate: model/common.py:231, class weights: :326) come from the
    library's counter-based generator (Philox4x32-10, include/denet_hip.h
    "T1 in train mode"), keyed by ONE 64-bit seed per forward that is drawn from
    torch's CPU generator (so ``torch.manual_seed`` still fixes the run): same
    distribution as ``nn.Dropout``, different random stream.  The fused step
    (``HeadStepEngine`` / ``HeadStepFunction`` with ``drop=``) generates the same
    multipliers inside its kernels from the same seed;
  * the binary logits are not differentiable (the reference only ever calls
    ``loss_k.backward()``: trainer/trainers.py:972-974).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, sharding
from .operations import depart_first_dim, merge_first_k_dim


def freeze_weights(module):
    for p in module.parameters():
        p.requires_grad = False


class MaskedAveragePooling(nn.Module):
    """model/common.py:149-200.  Inside WeightEstimator only its ``linear``
    parameters are used (the pooling is fused into kernels K1/K2); called on its
    own it runs the HIP mask-resize and masked-sum kernels plus the 1x1 projection,
    differentiable w.r.t. the embedding and ``linear`` as in the reference."""

    def __init__(self, depth=None, project=True):
        super().__init__()
        self.depth = depth
        self.linear = None
        if project and depth is not None:
            self.linear = nn.Conv2d(in_channels=depth, out_channels=depth, kernel_size=1)

    def forward(self, embedding, mask, pooling=True, eps=1e-3):
        B, c, h, w = embedding.shape
        m, cnt = ops.mask_resize(mask, h, w)
        if not pooling:
            return m.reshape(B, 1, h, w) * embedding
        S = ops.MaskedSumFunction.apply(embedding, m)
        proto = (S / (cnt.unsqueeze(1) + eps)).reshape(B, c, 1, 1)
        return self.linear(proto) if self.linear is not None else proto


class GAM(nn.Module):
    """Guided attention module, model/common.py:211-246.  Inside WeightEstimator the gate is
    fused into kernel K2; called on its own it is differentiable like the reference's module (masked mean on
    the HIP kernels, gate and the channel-wise product as stock modules)."""

    def __init__(self, in_channels, drop_rate=0.5):
        super().__init__()
        self.DEPTH = in_channels
        self.DROP_RATE = drop_rate
        self.gate = nn.Sequential(
            nn.Conv2d(in_channels, in_channels, kernel_size=1),
            nn.ReLU(),
            nn.Conv2d(in_channels, in_channels, kernel_size=1),
            nn.Dropout(p=drop_rate),
            nn.Sigmoid())

    @staticmethod
    def mask(embedding, mask):
        B, c, h, w = embedding.shape
        m, _ = ops.mask_resize(mask, h, w)
        return m.reshape(B, 1, h, w) * embedding

    def forward(self, *x):
        Fs, Ys = x
        B, c, h, w = Fs.shape
        m, _ = ops.mask_resize(Ys, h, w)
        att = (ops.MaskedSumFunction.apply(Fs, m) / float(h * w)).reshape(B, c, 1, 1)
        return self.gate(att) * Fs


class WeightEstimator(nn.Module):
    """model/common.py:249-441."""

    def __init__(self, num_classes, depth, drop_rate=0.5, support_free=False):
        super().__init__()
        self.num_classes = num_classes
        self.depth = depth
        self.sfree = support_free
        self.gam = GAM(in_channels=depth)
        self.map = MaskedAveragePooling(depth=depth)
        self.dropout = nn.Dropout(p=drop_rate)
        self.weight = nn.Parameter(0.01 * torch.randn([num_classes, depth]), requires_grad=True)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.record = {_: False for _ in range(num_classes)}

    # ---- helpers -------------------------------------------------------------------------
    def _params(self):
        g0, g2, lin = self.gam.gate[0], self.gam.gate[2], self.map.linear
        return (self.weight, g0.weight, g0.bias, g2.weight, g2.bias, lin.weight, lin.bias)

    def dropout_config(self, seed=None):
        """(seed, p_gate, p_weight) of the next train-mode forward -- what the fused step takes as
        ``drop`` -- or None in eval mode.  The seed comes from torch's CPU generator (no device sync)."""
        if not self.training:
            return None
        if seed is None:
            # every rank of a torchrun job starts torch's generator from the same default seed, so the draw
            # alone would give all shards bit-identical masks: draw * world + rank is distinct per rank and step
            world, rank = sharding.world(), sharding.rank()
            seed = int(torch.randint(0, 2 ** 62 // world, (1,), dtype=torch.int64).item()) * world + rank
        return seed, float(self.gam.gate[3].p), float(self.dropout.p)

    def _dropout_masks(self, B, N, device, seed=None):
        """Already-scaled multipliers (0 or 1/(1-p)) as tensors, or None in eval mode: the values the
        fused step would generate in its kernels for the same seed (denet_dropout_mask)."""
        cfg = self.dropout_config(seed)
        if cfg is None:
            return None, None
        seed, pg, pw = cfg
        self.last_dropout_seed = seed

        def draw(stream, shape, p):
            if p <= 0.0:
                return None
            if p >= 1.0:
                return torch.zeros(shape, device=device)
            return ops.dropout_mask(seed, stream, p, shape, device)

        return (draw(ops.DROPOUT_GATE, (B, self.depth), pg),
                draw(ops.DROPOUT_WEIGHT, (N, self.num_classes, self.depth), pw))

    @staticmethod
    def _labels2d(labels, N):
        labels = torch.as_tensor(labels)
        return labels.reshape(N, -1).long().contiguous()

    # ---- reference API -------------------------------------------------------------------
    def extend(self, Fs, Ys, labels, mode="training", gate_mask=None, weight_mask=None):
        """Fs (N, way, shot, c, h, w), Ys (N, way, shot, H, W), labels (N, way) ->
        'training': (N, num_class, c); 'deploy': (num_class, c).  model/common.py:341-372."""
        assert mode in ['training', 'deploy'], 'mode `%s` not supported' % mode
        if mode == 'deploy':
            return ops.batch_mean(self.extend(Fs, Ys, labels, mode='training').detach())
        N, way, shot = Fs.shape[:3]
        labels = self._labels2d(labels, N).to(Fs.device)
        if gate_mask is None and weight_mask is None:
            gate_mask, weight_mask = self._dropout_masks(N * way * shot, N, Fs.device)
        if self.sfree:
            return self._extend_support_free(Fs, Ys, labels, gate_mask, weight_mask)
        Wn, _ = ops.ExtendFunction.apply(Fs, *self._params(), Ys, labels, gate_mask, weight_mask)
        return Wn

    @torch.no_grad()
    def _extend_support_free(self, Fs, Ys, labels, gate_mask, weight_mask):
        """model/common.py:326-338 with support_free=True: a class row is inserted only the
        first time the class is seen and every episode's (dropout-perturbed) matrix is persisted
        into self.weight, so episodes are processed in order.  Forward only."""
        N, way = labels.shape
        _, proto = ops.ExtendFunction.apply(Fs.detach(), *[p.detach() for p in self._params()], Ys, labels,
                                            gate_mask, None)
        proto = proto.reshape(N, way, -1)
        lab = labels.tolist()                      # the reference syncs here too (label[_].item())
        out = []
        for n in range(N):
            weight = self.weight.detach().clone()
            if weight_mask is not None:
                weight = weight * weight_mask[n]
            for j in range(way):
                l = lab[n][j]
                if not self.record[l]:
                    weight[l] = proto[n, j]
                    self.record[l] = True
                    self.weight.data = weight
            out.append(weight)
        return torch.stack(out, dim=0)

    def infer(self, embedding, weight, mode="training"):
        """(N, c, h, w) x (N, num_class, c) | (num_class, c) -> (N, num_class, h, w).
        model/common.py:374-399.  As in the reference, 'deploy' classifies with self.weight."""
        assert mode in ['training', 'deploy'], 'mode `%s` not supported' % mode
        if mode == 'training':
            return ops.ClassifyFunction.apply(embedding, weight)
        return ops.ClassifyFunction.apply(embedding, self.weight)

    def forward(self, *x, mode='training'):
        """model/common.py:401-441."""
        assert mode in ['training', 'deploy'], 'mode `%s` not supported' % mode
        if mode == 'training':
            embedding, Fs, Ys, labels = x
            if self.sfree:
                return self.infer(embedding, self.extend(Fs, Ys, labels, mode=mode), mode=mode)
            return self.forward_with_binary(embedding, Fs, Ys, labels)[0]
        assert len(x) == 1 or len(x) == 4, 'in deploy mode, support extend and infer together or infer only'
        if len(x) == 4:
            embedding, Fs, Ys, labels = x
            self.weight.data = self.extend(Fs, Ys, labels, mode=mode).data
        else:
            embedding = x[0]
        return self.infer(embedding, weight=self.weight, mode=mode)

    # ---- fused path used by DENet.forward ---------------------------------------------
    def forward_with_binary(self, embedding, Fs, Ys, labels, base=True, gate_mask=None, weight_mask=None):
        """extend + infer + get_binary_logits for every way in one autograd node (kernels
        K0-K3e).  Returns (logits_full (N,K,h,w), logits_binary (N*way,2,h,w))."""
        N, way, shot = Fs.shape[:3]
        labels = self._labels2d(labels, N).to(Fs.device)
        if self.sfree:
            full = self.infer(embedding, self.extend(Fs, Ys, labels), mode='training')
            return full, ops.binary_logits(full.detach(), labels, way=way, base=base)
        need_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (embedding, Fs) + self._params())
        if not need_grad and base and self._fused_ok(Fs, Ys):
            # forward only (test / inference loops): the fused step's three forward launches, dropout (train
            # mode without gradients) generated in the kernels
            drop = self.dropout_config() if gate_mask is None and weight_mask is None else None
            if drop is None or max(drop[1:]) < 1.0:          # (p = 1 has no in-kernel form: masks of zeros below)
                return ops.head_forward(embedding, Fs, self._params(), Ys, labels, gate_mask, weight_mask, drop)
        if gate_mask is None and weight_mask is None:
            gate_mask, weight_mask = self._dropout_masks(N * way * shot, N, Fs.device)
        full, binl, _ = ops.EstimatorFunction.apply(embedding, Fs, *self._params(), Ys, labels, gate_mask,
                                                    weight_mask, base)
        return full, binl

    def _fused_ok(self, Fs, Ys):
        """Whether the fused step's kernels take this shape (denet_head_step_workspace says so)."""
        N, way, shot, C, h, w = Fs.shape
        H, W = Ys.shape[-2:]
        return C in (64, 256) and ops.query("denet_head_step_workspace", N, way, shot, self.num_classes, C, h, w,
                                            H, W) > 0

    def head_step(self, embedding, Fs, Ys, labels, Yq, Yq_full, weight=None, gate_mask=None, weight_mask=None):
        """The reference's training-step sequence for the head -- estimator(...) + get_binary_logits
        (model/head/denet.py:79-91), both F.interpolate calls (trainer/trainers.py:954-956) and
        DENet_loss_wrapper (trainer/utils.py:264-281) -- as ONE autograd node on the fused seven-launch step
        (ops.HeadStepFunction): returns (loss_k, loss_p, logits_full (N,K,h,w), logits_binary (N*way,2,h,w));
        loss_k.backward() runs the three backward launches.  weight: (N,K) per-episode class weights of the
        cross entropy (trainer/trainers.py:961-963) or None.  Train mode draws ONE dropout seed; the masks
        are generated inside the kernels (and regenerated in the backward)."""
        N, way, shot = Fs.shape[:3]
        labels = self._labels2d(labels, N).to(Fs.device)
        if self.sfree:
            # support_free persists weights episode by episode (model/common.py:332-338): forward only,
            # on the stand-alone kernels
            from .losses import loss_denet_batch
            full, binl = self.forward_with_binary(embedding, Fs, Ys, labels)
            loss_k, loss_p = loss_denet_batch(full, Yq_full, binl, Yq, weight)
            return loss_k, loss_p, full, binl
        drop = None
        if gate_mask is None and weight_mask is None:
            drop = self.dropout_config()
            if drop is not None and max(drop[1:]) >= 1.0:    # p = 1: everything dropped, as tensors of zeros
                gate_mask, weight_mask = self._dropout_masks(N * way * shot, N, Fs.device, seed=drop[0])
                drop = None
            if drop is not None:
                self.last_dropout_seed = drop[0]
        wgt = None if weight is None else weight.to(Fs.device, torch.float32).reshape(N, self.num_classes).contiguous()
        Yq2 = None if Yq is None else Yq.reshape(N * way, *Yq.shape[-2:])
        return ops.head_step(embedding, Fs, *self._params(), Ys, labels, Yq2, Yq_full, wgt, gate_mask, weight_mask, drop)


# ------------------------------------------------------------------------------------------
# Feature-producer tail (stock PyTorch-ROCm convolutions; SURVEY.md §2: out of scope for HIP,
# kept so that state_dict keys match: embedding.0.*, deeplab_head.aspp.convs.*, ...).
# ------------------------------------------------------------------------------------------
class _ASPPBranch(nn.Sequential):
    def __init__(self, cin, cout, dilation, p):
        if dilation == 0:
            conv = nn.Conv2d(cin, cout, 1)
        else:
            conv = nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation)
        super().__init__(conv, nn.ReLU(), nn.Dropout2d(p=p))


class _ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout, p):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1), nn.ReLU(), nn.Dropout2d(p=p))

    def forward(self, x):
        size = x.shape[-2:]
        return F.interpolate(super().forward(x), size=size, mode='bilinear', align_corners=False)


class RegASPP(nn.Module):
    """model/common.py:61-91: ASPP without BN, with dropout."""

    def __init__(self, in_channels=256, atrous_rates=(6, 12, 18), p=0.5):
        super().__init__()
        out_channels = 256
        branches = [_ASPPBranch(in_channels, out_channels, 0, p)]
        branches += [_ASPPBranch(in_channels, out_channels, r, p) for r in atrous_rates]
        branches.append(_ASPPPooling(in_channels, out_channels, p))
        self.convs = nn.ModuleList(branches)
        self.project = nn.Sequential(nn.Conv2d(5 * out_channels, out_channels, 1), nn.ReLU(), nn.Dropout(p=p))

    def forward(self, x):
        return self.project(torch.cat([conv(x) for conv in self.convs], dim=1))


class DeepLabHead(nn.Module):
    """model/common.py:130-146."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.aspp = RegASPP(in_channels, (6, 12, 18))
        self.conv1 = nn.Conv2d(256, 256, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(256)
        self.relu = nn.ReLU()

    def forward(self, x, relu=True):
        out = self.conv1(self.aspp(x))
        return self.relu(self.bn(out)) if relu else out


def _torchvision_backbone(name):
    """model/backbone/backbones.py:175-186 (ResNet50, dilated layer3/4, IntermediateLayerGetter).
    torchvision and its pretrained weights are not part of this image: raise loudly."""
    try:
        import torchvision                                              # noqa: F401
        from torchvision.models._utils import IntermediateLayerGetter
    except ImportError as e:                                            # pragma: no cover
        raise ImportError(
            f"DENet(backbone={name!r}) needs torchvision for the frozen feature producer; pass an nn.Module "
            "that maps images to {'layer3': [*,1024,h,w]} as `backbone` instead") from e
    if name == 'ResNet50':                                              # pragma: no cover
        net = torchvision.models.resnet50(weights="IMAGENET1K_V1", replace_stride_with_dilation=[False, True, True])
        return IntermediateLayerGetter(net, return_layers={'layer3': 'layer3'})
    raise NotImplementedError(f"backbone {name!r}: only ResNet50 is wired to torchvision here")   # pragma: no cover


class DENet(nn.Module):
    """Dynamic Extension Net, model/head/denet.py:8-93.

    ``backbone`` is the reference's name ('ResNet50' / 'DenseNet121', needs
    torchvision) or, as an extension, any nn.Module returning ``{'layer3': features}``
    with 1024 channels (used by the tests and by synthetic-feature runs)."""

    def __init__(self, backbone="ResNet50", maximum_num_classes=21, depth=256, drop_rate=0.5,
                 support_free=False, visualize=True, feature_dtype=torch.float32):
        super().__init__()
        # feature_dtype (extension): torch.bfloat16 makes the producer's tail run under
        # bf16 autocast, so `features()` hands bf16 NCHW straight to the head's bf16 kernels (fp32
        # accumulation inside) -- no fp32 feature map, no cast pass between producer and head
        assert feature_dtype in (torch.float32, torch.bfloat16), 'feature_dtype must be float32 or bfloat16'
        self.feature_dtype = feature_dtype
        if isinstance(backbone, str):
            assert backbone in ['ResNet50', 'DenseNet121'], 'get unsupported backbone "%s" for DENet.' % backbone
            backbone = _torchvision_backbone(backbone)
        self.maximum_num_classes = maximum_num_classes
        self.depth = depth
        self.visualize = visualize
        self.vis = dict()
        self.backbone = backbone
        freeze_weights(self.backbone)
        self.embedding = nn.Sequential(
            nn.Conv2d(1024, self.depth, kernel_size=3, padding=2, dilation=2),
            nn.ReLU(),
            nn.Dropout2d(p=drop_rate))
        self.deeplab_head = DeepLabHead(256)
        self.estimator = WeightEstimator(self.maximum_num_classes, self.depth, support_free=support_free)

    def features(self, images):
        """backbone['layer3'] -> embedding -> deeplab_head (model/head/denet.py:60-62,71-73)."""
        layer3 = self.backbone(images)['layer3']
        if self.feature_dtype == torch.bfloat16 and layer3.is_cuda:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                f = self.deeplab_head(self.embedding(layer3), relu=True)
            return f.to(torch.bfloat16).contiguous()        # (already bf16 under autocast: no copy)
        return self.deeplab_head(self.embedding(layer3), relu=True)

    def forward(self, *x):
        Is, Ys, Iq, label = x
        N, way, shot = Is.size()[:3]
        Fs = self.features(merge_first_k_dim(Is, dims=(0, 1, 2)))
        Fs = depart_first_dim(Fs, dims=(N * way, shot))
        if self.visualize and way == 1:
            self.vis.update({'hidden_Fs': torch.select(Fs, dim=1, index=0).clone().detach()})
        Fs = depart_first_dim(Fs, dims=(N, way))
        Fq = self.features(Iq)
        if self.visualize:
            self.vis.update({'hidden_Fq': Fq.clone().detach()})
        # knowledge logits and, per way, the pattern (binary) logits: one fused node
        logits_full, logits_binary = self.estimator.forward_with_binary(Fq, Fs, Ys, label)
        return logits_full, logits_binary

    def forward_step(self, Is, Ys, Iq, label, Yq, Yq_full, weight=None):
        """forward() + the two up-samplings + DENet_loss_wrapper of the reference's training step
        (trainer/trainers.py:951-968) in one call: (loss_k, loss_p, logits_full, logits_binary).  The head
        part is ONE fused autograd node (WeightEstimator.head_step); `loss_k.backward()` then reaches the
        conv stack through dFq / dFs.  forward() stays what inference calls."""
        N, way, shot = Is.size()[:3]
        Fs = self.features(merge_first_k_dim(Is, dims=(0, 1, 2)))
        Fs = depart_first_dim(Fs, dims=(N * way, shot))
        if self.visualize and way == 1:
            self.vis.update({'hidden_Fs': torch.select(Fs, dim=1, index=0).clone().detach()})
        Fs = depart_first_dim(Fs, dims=(N, way))
        Fq = self.features(Iq)
        if self.visualize:
            self.vis.update({'hidden_Fq': Fq.clone().detach()})
        return self.estimator.head_step(Fq, Fs, Ys, label, Yq, Yq_full, weight)


key2arch = {'DENet': DENet}


def get_architecture(arch_name=None):
    """model/head/__init__.py:14-24 (only the DENet entry is in scope here)."""
    if arch_name is None:
        print("Using default DENet model")
        return DENet
    if arch_name not in key2arch:
        raise NotImplementedError('Model {} not implemented'.format(arch_name))
    print('Using model: {}'.format(arch_name))
    return key2arch[arch_name]
