"""Generate tests/golden/*.npz by running the REFERENCE's own code on CPU.

Run in the build container only (needs /root/reference; the GPU box has
neither the reference nor any need for this script):

    python oracle/gen_golden.py

Recipe: `utils`, `loss` import normally from /root/reference;
`model/common.py` is loaded by file path because `model/__init__.py` pulls in
torchvision, which this image lacks.  `np.float` (removed in numpy >= 1.24) is
aliased to `float` for `BinaryIOU._update` (utils/metrics.py:12-13).

Only DATA is written: inputs, parameters and the reference's outputs.
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state, CONFIGS  # noqa: E402

if not hasattr(np, "float"):
    np.float = float                      # shim, see module docstring

import utils.metrics as ref_metrics       # noqa: E402  (reference)
import utils.operations as ref_ops        # noqa: E402  (reference)
import loss.losses as ref_losses          # noqa: E402  (reference)

_spec = importlib.util.spec_from_file_location("ref_common", os.path.join(REF, "model/common.py"))
ref_common = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_common)

OUT = os.path.join(ROOT, "tests", "golden")


def build_estimator(K, c, state):
    we = ref_common.WeightEstimator(num_classes=K, depth=c)
    we.load_state_dict(state)
    return we


def ref_denet_tail(we, Fq, Fs, Ys, label):
    """model/head/denet.py:57-93 with the feature producer removed: the only
    lines of DENet.forward that touch the head (cannot import the class: it
    needs torchvision + a pretrained download)."""
    way = Fs.shape[1]
    lab = label.unsqueeze(-1) if way == 1 else label
    full = we(Fq, Fs, Ys, lab, mode="training")
    if way > 1:
        b = torch.stack([ref_ops.get_binary_logits(full, lab[:, i]) for i in range(way)], 1)
        b = ref_ops.merge_first_k_dim(b, (0, 1))
    else:
        b = ref_ops.get_binary_logits(full, lab[:, 0], base=True)
    return full, b


def ref_losses_like_trainer(full, binl, Yq, Yq_full, wgt):
    """trainer/trainers.py:954-956 + trainer/utils.py:264-281 (restated glue,
    calling the reference's loss functions)."""
    size = Yq_full.shape[-2:]
    b_up = F.interpolate(binl.float(), size=size, mode="bilinear", align_corners=True)
    f_up = F.interpolate(full, size=size, mode="bilinear", align_corners=True)
    N = f_up.shape[0]
    lk = 0.0
    for pre, tar, w in zip(f_up.chunk(N, 0), Yq_full.chunk(N, 0), wgt):
        lk = lk + ref_losses.loss_denet(pre, tar, weight=w, mode="knowledge")
    lk = lk / N
    lp = ref_losses.loss_denet(b_up, Yq.reshape(-1, *size), mode="pattern")
    return f_up, b_up, lk, lp


def capture_case(shape: EpisodeShape, train: bool, fname: str, seed_offset=0, planted=False):
    ep = make_episodes(shape, seed_offset=seed_offset, planted=planted, ignore_rows=3)
    state = make_estimator_state(shape.K, shape.c)
    if planted:
        state["weight"] = ep["signatures"].clone()
    we = build_estimator(shape.K, shape.c, state)
    we.train(train)
    rec = {}
    if train:
        torch.manual_seed(99 + seed_offset)
        wmasks = []

        def hook_gate(mod, inp, out):
            rec["gate_mask"] = torch.where(inp[0] != 0, out / inp[0], torch.zeros_like(out)).detach()

        def hook_w(mod, inp, out):
            wmasks.append(torch.where(inp[0] != 0, out / inp[0], torch.zeros_like(out)).detach())

        we.gam.gate[3].register_forward_hook(hook_gate)
        we.dropout.register_forward_hook(hook_w)
    Fq = ep["Fq"].float().clone().requires_grad_(True)
    Fs = ep["Fs"].float().clone().requires_grad_(True)
    full, binl = ref_denet_tail(we, Fq, Fs, ep["Ys"], ep["label"])
    f_up, b_up, lk, lp = ref_losses_like_trainer(full, binl, ep["Yq"], ep["Yq_full"], ep["wgt"])
    lk.backward()
    d = {("in:" + k): v.float().numpy() if v.is_floating_point() else v.numpy() for k, v in ep.items()}
    d.update({("param:" + k): v.numpy() for k, v in state.items()})
    d["shape"] = np.array([shape.N, shape.way, shape.shot, shape.K, shape.c, shape.h, shape.w,
                           shape.H, shape.W], dtype=np.int64)
    d["out:logits_full"] = full.detach().numpy()
    d["out:logits_binary"] = binl.detach().numpy()
    d["out:loss_k"] = lk.detach().numpy()
    d["out:loss_p"] = lp.detach().numpy()
    d["out:dFq"] = Fq.grad.numpy()
    d["out:dFs"] = Fs.grad.numpy()
    for k, v in we.named_parameters():
        d["grad:" + k] = v.grad.numpy()
    # K1 intermediates (att, pooled prototype before the linear layer) in eval
    if not train:
        with torch.no_grad():
            B = shape.N * shape.way * shape.shot
            Fm = ep["Fs"].float().reshape(B, shape.c, shape.h, shape.w)
            Ym = ep["Ys"].reshape(B, shape.H, shape.W)
            att = F.adaptive_avg_pool2d(ref_common.GAM.mask(Fm, Ym), (1, 1))
            g = we.gam.gate(att)
            proto = we.map(g * Fm, Ym)
            lab = ep["label"].unsqueeze(-1) if shape.way == 1 else ep["label"]
            d["out:att"] = att.reshape(B, shape.c).numpy()
            d["out:gate"] = g.reshape(B, shape.c).numpy()
            d["out:proto"] = proto.reshape(B, shape.c).numpy()
            d["out:weights"] = we.extend(ep["Fs"].float(), ep["Ys"], lab).numpy()
    else:
        d["mask:gate"] = rec["gate_mask"].reshape(-1, shape.c).numpy()
        d["mask:weight"] = torch.stack(wmasks, 0).numpy()
    # metrics on the reference's own upsampled logits
    fi = ref_metrics.FullIOU(num_classes=shape.K)
    fi.update(ep["Yq_full"], f_up.detach())
    d["out:confmat"] = fi.mat.astype(np.int64)
    d["out:full_miou"] = np.float64(fi.mean_iou())
    bi = ref_metrics.BinaryIOU()
    if shape.way > 1:
        bu = b_up.detach().reshape(shape.N, shape.way, *b_up.shape[-3:])
        for i in range(shape.way):
            bi.update(ep["Yq"][:, i], bu[:, i], ep["label"][:, i].numpy())
    else:
        bi.update(ep["Yq"], b_up.detach(), ep["label"].numpy())
    keys = list(bi.inter.keys())
    d["out:biou_keys"] = np.array(keys, dtype=np.int64)
    d["out:biou_inter"] = np.array([bi.inter[k] for k in keys], dtype=np.float64)
    d["out:biou_union"] = np.array([bi.union[k] for k in keys], dtype=np.float64)
    d["out:biou_mean"] = np.float64(bi.mean_iou()) if keys else np.float64("nan")
    d["out:pred_full"] = f_up.detach().argmax(1).numpy().astype(np.uint8)
    d["out:pred_bin"] = b_up.detach().max(1)[1].numpy().astype(np.uint8)
    top = f_up.detach().topk(2, dim=1)[0]
    d["out:min_margin"] = np.float32((top[:, 0] - top[:, 1]).min())
    np.savez_compressed(os.path.join(OUT, fname), **d)
    print(fname, "loss_k=%.6f loss_p=%.6f margin=%.3g" % (float(lk), float(lp), float(d["out:min_margin"])))


def capture_metrics():
    """M1/M2 directly at the metric boundary (predictions in -> counts out)."""
    g = torch.Generator().manual_seed(4321)
    K, N, H, W = 7, 3, 19, 23
    d = {}
    fi = ref_metrics.FullIOU(num_classes=K)
    bi = ref_metrics.BinaryIOU()
    for u in range(3):
        yt = torch.randint(0, K, (N, H, W), generator=g)
        yt[:, :2] = 255
        if u == 1:
            yt[0, 5] = 254         # another out-of-range value, dropped by `< n`
        lp = torch.randn(N, K, H, W, generator=g)
        lp[:, 3] = lp[:, 2]        # exact ties: first index must win
        fi.update(yt, lp)
        yb = torch.randint(0, 2, (N, H, W), generator=g)
        if u == 0:
            yb[1] = 0              # skipped sample
        lb = torch.randn(N, 2, H, W, generator=g)
        lb[:, 1, :3] = lb[:, 0, :3]
        cls = np.array([5, 2, 5]) if u != 2 else np.array([9, 2, 1])
        bi.update(yb, lb, cls)
        d[f"in:yt{u}"], d[f"in:lp{u}"] = yt.numpy(), lp.numpy()
        d[f"in:yb{u}"], d[f"in:lb{u}"], d[f"in:cls{u}"] = yb.numpy(), lb.numpy(), cls
    d["out:mat"] = fi.mat.astype(np.int64)
    d["out:miou"] = np.float64(fi.mean_iou())
    d["out:class_iou"] = np.array([fi.class_iou()[k] for k in range(K)])
    d["out:sub_miou"] = np.float64(fi.mean_subclasses_iou([1, 3, 4]))
    keys = list(bi.inter.keys())
    d["out:biou_keys"] = np.array(keys, dtype=np.int64)
    d["out:biou_inter"] = np.array([bi.inter[k] for k in keys])
    d["out:biou_union"] = np.array([bi.union[k] for k in keys])
    d["out:biou_mean"] = np.float64(bi.mean_iou())
    d["out:biou_bf"] = np.float64(bi.mean_bf_iou())
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **d)
    print("metrics.npz", d["out:miou"], d["out:biou_mean"])


def capture_index_tables():
    """ATen's own index rules, observed through F.interpolate."""
    d = {}
    for (H, h) in [(321, 41), (642, 81), (65, 9), (50, 7), (70, 9), (33, 33), (5, 9)]:
        src = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1).expand(1, 1, H, 2)
        near = F.interpolate(src, size=(h, 2), mode="nearest")[0, 0, :, 0]
        d[f"nearest_{H}_{h}"] = near.numpy().astype(np.int64)
        # bilinear h -> H: interpolate a ramp and a one-hot basis to read taps
        eye = torch.eye(h).view(1, h, h, 1).expand(1, h, h, 2).contiguous()
        up = F.interpolate(eye, size=(H, 2), mode="bilinear", align_corners=True)[0, :, :, 0]  # [h,H]
        d[f"bilinear_{h}_{H}"] = up.numpy()
    np.savez_compressed(os.path.join(OUT, "index_tables.npz"), **d)
    print("index_tables.npz")


def capture_fullsize_checksums():
    """C2 at full size: outputs reduced to a few numbers + sampled entries."""
    shape = CONFIGS["C2"]
    ep = make_episodes(shape)
    state = make_estimator_state(shape.K, shape.c)
    we = build_estimator(shape.K, shape.c, state).eval()
    Fq = ep["Fq"].clone().requires_grad_(True)
    Fs = ep["Fs"].clone().requires_grad_(True)
    full, binl = ref_denet_tail(we, Fq, Fs, ep["Ys"], ep["label"])
    f_up, b_up, lk, lp = ref_losses_like_trainer(full, binl, ep["Yq"], ep["Yq_full"], ep["wgt"])
    lk.backward()
    g = torch.Generator().manual_seed(5)

    def sample(t, n=4096):
        flat = t.detach().reshape(-1)
        idx = torch.randint(0, flat.numel(), (n,), generator=g)
        return idx.numpy(), flat[idx].numpy()

    d = {"out:loss_k": lk.detach().numpy(), "out:loss_p": lp.detach().numpy()}
    for name, t in [("logits_full", full), ("logits_binary", binl), ("dFq", Fq.grad), ("dFs", Fs.grad)] + \
                   [("grad:" + k, v.grad) for k, v in we.named_parameters()]:
        i, v = sample(t)
        d["idx:" + name], d["val:" + name] = i, v
        d["sum:" + name] = np.float64(t.detach().double().sum())
        d["abs:" + name] = np.float64(t.detach().double().abs().sum())
    fi = ref_metrics.FullIOU(num_classes=shape.K)
    fi.update(ep["Yq_full"], f_up.detach())
    d["out:confmat"] = fi.mat.astype(np.int64)
    top = f_up.detach().topk(2, dim=1)[0]
    d["out:min_margin"] = np.float32((top[:, 0] - top[:, 1]).min())
    np.savez_compressed(os.path.join(OUT, "c2_fullsize.npz"), **d)
    print("c2_fullsize.npz loss_k=%.6f loss_p=%.6f" % (float(lk), float(lp)))


def capture_api_cases():
    """Secondary API surface (SURVEY.md §8a K2', §8f N1) and the loss functions called the way a
    user of the reference calls them: deploy mode, support_free, get_binary_logits(base=False),
    cross_entropy2d / binary_focal_loss on a batch (low-res and same-size), state_dict keys."""
    import json
    s = EpisodeShape("api", N=3, way=2, shot=3, K=9, c=64, h=7, w=9, H=50, W=70, config_index=104)
    ep = make_episodes(s, ignore_rows=3)
    # make labels overlap across episodes and inside one episode (support_free keeps the FIRST)
    ep["label"] = torch.tensor([[3, 5], [5, 2], [4, 4]])
    state = make_estimator_state(s.K, s.c)
    d = {("in:" + k): v.float().numpy() if v.is_floating_point() else v.numpy() for k, v in ep.items()}
    d.update({("param:" + k): v.numpy() for k, v in state.items()})
    d["shape"] = np.array([s.N, s.way, s.shot, s.K, s.c, s.h, s.w, s.H, s.W], dtype=np.int64)
    Fq, Fs = ep["Fq"].float(), ep["Fs"].float()
    with torch.no_grad():
        # ---- deploy: extend + infer, then infer only (model/common.py:427-441)
        we = build_estimator(s.K, s.c, state).eval()
        d["out:deploy_logits"] = we(Fq, Fs, ep["Ys"], ep["label"], mode="deploy").numpy()
        d["out:deploy_weight"] = we.weight.data.clone().numpy()
        d["out:deploy_infer_only"] = we(Fq.flip(0), mode="deploy").numpy()
        # ---- support_free (model/common.py:326-338), two consecutive batches
        wf = ref_common.WeightEstimator(num_classes=s.K, depth=s.c, support_free=True)
        wf.load_state_dict(state)
        wf.eval()
        d["out:sfree_weights1"] = wf.extend(Fs, ep["Ys"], ep["label"]).numpy()
        d["out:sfree_persist1"] = wf.weight.data.clone().numpy()
        lab2 = torch.tensor([[1, 3], [6, 7], [2, 8]])
        d["in:label2"] = lab2.numpy()
        d["out:sfree_logits2"] = wf(Fq, Fs.flip(0), ep["Ys"].flip(0), lab2, mode="training").numpy()
        d["out:sfree_persist2"] = wf.weight.data.clone().numpy()
        d["out:sfree_record"] = np.array([int(wf.record[k]) for k in range(s.K)], dtype=np.int64)
        # ---- get_binary_logits(base=False) (utils/operations.py:334-336)
        we2 = build_estimator(s.K, s.c, state).eval()
        full = we2(Fq, Fs, ep["Ys"], ep["label"], mode="training")
        d["out:full"] = full.numpy()
        d["out:bin_base_false"] = ref_ops.get_binary_logits(full, ep["label"][:, 1], base=False).numpy()
        d["out:bin_base_true"] = ref_ops.get_binary_logits(full, ep["label"][:, 1], base=True).numpy()
    # ---- losses on a batch, with gradients w.r.t. the logits
    g = torch.Generator().manual_seed(77)
    cw = torch.rand(s.K, generator=g) + 0.5
    d["in:class_weight"] = cw.numpy()
    for tag, x in (("low", full.clone()), ("same", torch.randn(s.N, s.K, s.H, s.W, generator=g))):
        x = x.requires_grad_(True)
        l = ref_losses.cross_entropy2d(x, ep["Yq_full"], weight=cw)
        l.backward()
        d[f"in:ce_{tag}"] = x.detach().numpy()
        d[f"out:ce_{tag}"] = l.detach().numpy()
        d[f"out:ce_{tag}_grad"] = x.grad.numpy()
    d["out:loss_general"] = ref_losses.loss_general(full, ep["Yq_full"]).numpy()
    xb = torch.randn(s.N, 2, s.H, s.W, generator=g)
    d["in:focal_same"] = xb.numpy()
    d["out:focal_same"] = ref_losses.binary_focal_loss(xb, ep["Yq"][:, 0].contiguous()).numpy()
    d["out:focal_low"] = ref_losses.loss_denet(torch.from_numpy(d["out:bin_base_true"]), ep["Yq"][:, 1].contiguous(), mode="pattern").numpy()
    # ---- state_dict layout a reference checkpoint carries
    emb = torch.nn.Sequential(torch.nn.Conv2d(1024, 256, kernel_size=3, padding=2, dilation=2), torch.nn.ReLU(),
                              torch.nn.Dropout2d(p=0.5))
    keys = {}
    for prefix, mod in (("embedding.", emb), ("deeplab_head.", ref_common.DeepLabHead(256)),
                        ("estimator.", ref_common.WeightEstimator(21, 256))):
        for k, v in mod.state_dict().items():
            keys[prefix + k] = list(v.shape)
    d["state_dict_layout"] = np.array(json.dumps(keys))
    np.savez_compressed(os.path.join(OUT, "api_cases.npz"), **d)
    print("api_cases.npz", len(keys), "state_dict entries")


def ref_process_nway(Yq_pre, way):
    """trainer/utils.py:128-154 (the module cannot be imported: it pulls in torchvision through model.head):
    the index_select / max / cat glue restated around the REFERENCE's own get_binary_logits."""
    N = Yq_pre.shape[0]
    background = Yq_pre.index_select(1, torch.tensor([2 * i for i in range(way)])).max(dim=1, keepdim=True)[0]
    foreground = Yq_pre.index_select(1, torch.tensor([2 * i + 1 for i in range(way)]))
    scores = torch.cat([background, foreground], dim=1)
    outs = [ref_ops.get_binary_logits(scores, torch.full([N], i + 1, dtype=torch.int64)) for i in range(way)]
    return torch.stack(outs, dim=1)


def capture_trainer_glue():
    """T1 / N4: what DENetTrainer.step returns (the two logit tensors up-sampled by ATen,
    trainer/trainers.py:954-956) and the n-way fusion of PatternTrainer.test (trainer/trainers.py:831-873:
    per-way class-agnostic predictions -> cat -> process_nway -> per-way BinaryIOU with the reference's
    own metric class)."""
    s = EpisodeShape("glue", N=2, way=2, shot=3, K=9, c=64, h=7, w=9, H=50, W=70, config_index=105)
    ep = make_episodes(s, ignore_rows=3)
    state = make_estimator_state(s.K, s.c)
    d = {("in:" + k): v.float().numpy() if v.is_floating_point() else v.numpy() for k, v in ep.items()}
    d.update({("param:" + k): v.numpy() for k, v in state.items()})
    d["shape"] = np.array([s.N, s.way, s.shot, s.K, s.c, s.h, s.w, s.H, s.W], dtype=np.int64)
    with torch.no_grad():
        we = build_estimator(s.K, s.c, state).eval()
        full, binl = ref_denet_tail(we, ep["Fq"].float(), ep["Fs"].float(), ep["Ys"], ep["label"])
        size = ep["Yq_full"].shape[-2:]
        d["out:full_up"] = F.interpolate(full, size=size, mode="bilinear", align_corners=True).numpy()
        d["out:bin_up"] = F.interpolate(binl.float(), size=size, mode="bilinear", align_corners=True).numpy()
    # n-way fusion: `way` class-agnostic predictions per episode (what PatternTrainer.step returns per way)
    g = torch.Generator().manual_seed(606)
    way, N, H, W = 3, 4, 33, 41
    bi = ref_metrics.BinaryIOU()
    for u in range(2):
        pre = torch.randn(N, 2 * way, H, W, generator=g)
        pre[:, 2, :4] = pre[:, 0, :4]                     # ties between way backgrounds
        Yq = torch.randint(0, 2, (N, way, H, W), generator=g)
        if u == 0:
            Yq[1, 2] = 0                                  # skipped (sample, way)
        cls = torch.tensor([[3, 7, 1], [7, 2, 3], [5, 5, 1], [2, 8, 4]]) if u == 0 else \
            torch.tensor([[9, 1, 2], [3, 3, 3], [6, 2, 7], [1, 4, 8]])
        fused = ref_process_nway(pre, way)
        for i in range(way):
            bi.update(torch.select(Yq, 1, i), torch.select(fused, 1, i), torch.select(cls, 1, i).numpy())
        d[f"in:nway_pre{u}"], d[f"in:nway_Yq{u}"], d[f"in:nway_cls{u}"] = pre.numpy(), Yq.numpy(), cls.numpy()
        d[f"out:nway_fused{u}"] = fused.numpy()
    keys = list(bi.inter.keys())
    d["out:nway_keys"] = np.array(keys, dtype=np.int64)
    d["out:nway_inter"] = np.array([bi.inter[k] for k in keys])
    d["out:nway_union"] = np.array([bi.union[k] for k in keys])
    d["out:nway_mean"] = np.float64(bi.mean_iou())
    d["out:nway_bf"] = np.float64(bi.mean_bf_iou())
    np.savez_compressed(os.path.join(OUT, "trainer_glue.npz"), **d)
    print("trainer_glue.npz", d["out:nway_mean"], d["out:nway_bf"])


def capture_loss_variants():
    """loss/losses.py:19-79 with the arguments no reference call site uses but the signatures take:
    reduction 'sum' / 'none' and an ignore_index other than 255, for both loss functions, on low-res logits
    (the up-sampling branch) -- inputs taken from api_cases.npz so the two fixtures share them."""
    z = np.load(os.path.join(OUT, "api_cases.npz"))
    full = torch.from_numpy(z["out:full"])
    yf = torch.from_numpy(z["in:Yq_full"])
    yq = torch.from_numpy(z["in:Yq"])[:, 1].contiguous()
    binl = torch.from_numpy(z["out:bin_base_true"])
    cw = torch.from_numpy(z["in:class_weight"])
    d = {}
    x = full.clone().requires_grad_(True)
    l = ref_losses.cross_entropy2d(x, yf, weight=cw, reduction='sum')
    l.backward()
    d["out:ce_sum"], d["out:ce_sum_grad"] = l.detach().numpy(), x.grad.numpy()
    d["out:ce_none"] = ref_losses.cross_entropy2d(full, yf, weight=cw, reduction='none').numpy()
    # ignore_index = 2: class 2 is skipped; 255 is then an ordinary (out-of-range) label for torch, so the
    # rows that carry it are relabelled 0 first
    yf2 = torch.where(yf == 255, torch.zeros_like(yf), yf)
    d["in:Yq_full_no255"] = yf2.numpy()
    x = full.clone().requires_grad_(True)
    l = ref_losses.cross_entropy2d(x, yf2, weight=cw, ignore_index=2)
    l.backward()
    d["out:ce_ign2"], d["out:ce_ign2_grad"] = l.detach().numpy(), x.grad.numpy()
    d["out:focal_sum"] = ref_losses.binary_focal_loss(binl, yq, reduction='sum').numpy()
    d["out:focal_none"] = ref_losses.binary_focal_loss(binl, yq, reduction='none').numpy()
    d["out:focal_ign1_a03"] = ref_losses.binary_focal_loss(binl, yq, alpha=0.3, ignore_index=1).numpy()
    np.savez_compressed(os.path.join(OUT, "loss_variants.npz"), **d)
    print("loss_variants.npz", {k: float(np.asarray(v).reshape(-1)[0]) for k, v in d.items() if k.startswith("out:") and np.asarray(v).size < 4})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "glue":     # only the fixture added in round 2
        capture_trainer_glue()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "loss_variants":     # only the fixture added in round 3
        capture_loss_variants()
        sys.exit(0)
    torch.set_num_threads(8)
    s1 = EpisodeShape("g1", N=2, way=1, shot=5, K=21, c=64, h=9, w=9, H=65, W=65, config_index=101, lam=2.5)
    s2 = EpisodeShape("g2", N=2, way=2, shot=3, K=9, c=64, h=7, w=9, H=50, W=70, config_index=102)
    s3 = EpisodeShape("g3", N=3, way=1, shot=1, K=5, c=64, h=9, w=9, H=65, W=65, config_index=103)
    capture_case(s1, train=False, fname="eval_1way_5shot.npz")
    capture_case(s2, train=False, fname="eval_2way_3shot.npz")
    capture_case(s3, train=False, fname="eval_1way_1shot.npz")
    capture_case(s1, train=True, fname="train_1way_5shot.npz", seed_offset=1)
    capture_case(s3, train=False, fname="planted_1way_1shot.npz", planted=True, seed_offset=2)
    capture_metrics()
    capture_index_tables()
    capture_fullsize_checksums()
    capture_api_cases()
    capture_trainer_glue()
    capture_loss_variants()
