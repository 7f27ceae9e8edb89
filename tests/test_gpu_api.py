"""GPU: the host-side mirror of the reference interface (denet_amd.head / losses /
metrics / trainer) driven the way a user of the reference drives it, against the
golden fixtures captured from the reference's own classes."""
import numpy as np
import pytest
import torch

from helpers import CASES, GOLDEN, assert_close, load_case, rel_err

pytestmark = pytest.mark.gpu
TIGHT = 2e-5
TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _estimator(params, K, c, dev, **kw):
    from denet_amd.head import WeightEstimator
    we = WeightEstimator(K, c, **kw)
    we.load_state_dict(params)
    return we.to(dev)


@pytest.mark.parametrize("name", [c for c in CASES if not c.startswith("train")])
def test_weight_estimator_module_eval(name, dev):
    z, d, ep, params, gm, wm = load_case(name)
    we = _estimator(params, d["K"], d["c"], dev).eval()
    N, way = d["N"], d["way"]
    lab = ep["label"].reshape(N, way).to(dev)
    Fq, Fs, Ys = ep["Fq"].float().to(dev), ep["Fs"].float().to(dev), ep["Ys"].to(dev)
    # forward() = extend + infer (model/common.py:421-426)
    logits = we(Fq, Fs, Ys, lab, mode="training")
    assert rel_err(logits.cpu(), z["out:logits_full"]) < TIGHT
    # the two halves on their own, and differentiable through both
    Fq.requires_grad_(True)
    Fs.requires_grad_(True)
    Wn = we.extend(Fs, Ys, lab)
    if "out:weights" in z.files:
        assert rel_err(Wn.detach().cpu(), z["out:weights"]) < TIGHT
    logits2 = we.infer(Fq, Wn)
    # (the no-grad forward above ran the fused step's forward launches -- sliced gate / projection chain --,
    # extend + infer the stand-alone kernels: same arithmetic, different summation order inside the mat-vecs)
    assert rel_err(logits2.detach(), logits.detach()) < 2e-6
    from denet_amd.losses import loss_denet_batch
    full, binl = we.forward_with_binary(Fq, Fs, Ys, lab)
    assert rel_err(binl.cpu(), z["out:logits_binary"]) < TIGHT
    lk, lp = loss_denet_batch(logits2, ep["Yq_full"].to(dev), binl, ep["Yq"].to(dev), ep["wgt"].to(dev))
    lk.backward()
    assert rel_err(lk.detach().cpu(), z["out:loss_k"]) < 1e-4
    assert rel_err(lp.cpu(), z["out:loss_p"]) < 1e-4
    assert rel_err(Fq.grad.cpu(), z["out:dFq"]) < TOL
    assert rel_err(Fs.grad.cpu(), z["out:dFs"]) < TOL
    for k, p in we.named_parameters():
        assert rel_err(p.grad.cpu(), z["grad:" + k]) < TOL, k


def test_weight_estimator_train_mode_with_injected_masks(dev):
    z, d, ep, params, gm, wm = load_case("train_1way_5shot")
    we = _estimator(params, d["K"], d["c"], dev).train()
    lab = ep["label"].reshape(d["N"], d["way"]).to(dev)
    full, binl = we.forward_with_binary(ep["Fq"].float().to(dev), ep["Fs"].float().to(dev), ep["Ys"].to(dev), lab,
                                        gate_mask=gm.to(dev), weight_mask=wm.to(dev).contiguous())
    assert rel_err(full.cpu(), z["out:logits_full"]) < TIGHT
    # self-drawn masks: keep-probability 0.5, scale 2 (nn.Dropout(0.5)); statistics only
    g, w_ = we._dropout_masks(64, 4, dev)
    assert set(g.unique().tolist()) <= {0.0, 2.0} and set(w_.unique().tolist()) <= {0.0, 2.0}
    assert 0.4 < float((g > 0).float().mean()) < 0.6 and w_.shape == (4, d["K"], d["c"])
    we.eval()
    assert we._dropout_masks(64, 4, dev) == (None, None)


def test_deploy_support_free_and_base_false(dev):
    """SURVEY.md §8a K2' / §8f N1 against the reference's own run (tests/golden/api_cases.npz)."""
    from denet_amd.head import WeightEstimator
    from denet_amd.operations import get_binary_logits
    z = np.load(GOLDEN + "/api_cases.npz")
    N, way, shot, K, c, h, w, H, W = [int(v) for v in z["shape"]]
    params = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}
    Fq, Fs = torch.from_numpy(z["in:Fq"]).to(dev), torch.from_numpy(z["in:Fs"]).to(dev)
    Ys, lab = torch.from_numpy(z["in:Ys"]).to(dev), torch.from_numpy(z["in:label"]).to(dev)
    # deploy: persist the batch mean of the extended weights, classify with it
    we = _estimator(params, K, c, dev).eval()
    out = we(Fq, Fs, Ys, lab, mode="deploy")
    assert rel_err(out.cpu(), z["out:deploy_logits"]) < TIGHT
    assert rel_err(we.weight.data.cpu(), z["out:deploy_weight"]) < TIGHT
    assert rel_err(we(Fq.flip(0).contiguous(), mode="deploy").cpu(), z["out:deploy_infer_only"]) < TIGHT
    # support_free: first insertion wins and persists, across two batches
    wf = WeightEstimator(K, c, support_free=True)
    wf.load_state_dict(params)
    wf = wf.to(dev).eval()
    w1 = wf.extend(Fs, Ys, lab)
    assert rel_err(w1.cpu(), z["out:sfree_weights1"]) < TIGHT
    assert rel_err(wf.weight.data.cpu(), z["out:sfree_persist1"]) < TIGHT
    lab2 = torch.from_numpy(z["in:label2"]).to(dev)
    l2 = wf(Fq, Fs.flip(0).contiguous(), Ys.flip(0).contiguous(), lab2, mode="training")
    assert rel_err(l2.cpu(), z["out:sfree_logits2"]) < TIGHT
    assert rel_err(wf.weight.data.cpu(), z["out:sfree_persist2"]) < TIGHT
    assert [int(wf.record[k]) for k in range(K)] == list(z["out:sfree_record"])
    # get_binary_logits, both variants
    full = torch.from_numpy(z["out:full"]).to(dev)
    assert torch.equal(get_binary_logits(full, lab[:, 1], base=False).cpu(), torch.from_numpy(z["out:bin_base_false"]))
    assert torch.equal(get_binary_logits(full, lab[:, 1], base=True).cpu(), torch.from_numpy(z["out:bin_base_true"]))


def test_loss_functions_called_like_the_reference(dev):
    from denet_amd import losses
    z = np.load(GOLDEN + "/api_cases.npz")
    yf = torch.from_numpy(z["in:Yq_full"]).to(dev)
    cw = torch.from_numpy(z["in:class_weight"]).to(dev)
    for tag in ("low", "same"):
        x = torch.from_numpy(z[f"in:ce_{tag}"]).to(dev).requires_grad_(True)
        loss = losses.cross_entropy2d(x, yf, weight=cw)
        loss.backward()
        assert rel_err(loss.detach().cpu(), z[f"out:ce_{tag}"]) < 1e-4, tag
        assert rel_err(x.grad.cpu(), z[f"out:ce_{tag}_grad"]) < TOL, tag
    full = torch.from_numpy(z["out:full"]).to(dev)
    assert rel_err(losses.loss_general(full, yf).cpu(), z["out:loss_general"]) < 1e-4
    yq = torch.from_numpy(z["in:Yq"]).to(dev)
    xb = torch.from_numpy(z["in:focal_same"]).to(dev)
    assert rel_err(losses.binary_focal_loss(xb, yq[:, 0].contiguous()).cpu(), z["out:focal_same"]) < 1e-4
    low = torch.from_numpy(z["out:bin_base_true"]).to(dev)
    assert rel_err(losses.loss_denet(low, yq[:, 1].contiguous(), mode="pattern").cpu(), z["out:focal_low"]) < 1e-4
    f = losses.get_loss({"name": "loss_denet"})
    assert rel_err(f(full, yf, mode="knowledge").cpu(), z["out:loss_general"]) < 1e-4


def test_metric_classes_on_device(dev):
    """FullIOU / BinaryIOU fed like the reference's trainer feeds them (up-sampled logits)."""
    from denet_amd.metrics import BinaryIOU, FullIOU
    z = np.load(GOLDEN + "/metrics.npz")
    K = z["out:mat"].shape[0]
    fi, bi = FullIOU(num_classes=K), BinaryIOU()
    for u in range(3):
        fi.update(torch.from_numpy(z[f"in:yt{u}"]).to(dev), torch.from_numpy(z[f"in:lp{u}"]).to(dev))
        bi.update(torch.from_numpy(z[f"in:yb{u}"]).to(dev), torch.from_numpy(z[f"in:lb{u}"]).to(dev), z[f"in:cls{u}"])
    assert np.array_equal(fi.mat, z["out:mat"].astype(np.float64))
    assert fi.mean_iou() == float(z["out:miou"])
    assert list(bi.inter.keys()) == list(z["out:biou_keys"])
    assert bi.mean_iou() == float(z["out:biou_mean"]) and bi.mean_bf_iou() == float(z["out:biou_bf"])


def test_denet_module_with_stub_backbone(dev):
    """DENet.forward end to end (model/head/denet.py:52-93): stock PyTorch-ROCm producer, HIP head,
    against the same modules' producer on the CPU feeding the oracle's head."""
    from denet_amd.head import DENet
    from denet_amd.trainer import SyntheticBackbone, SyntheticEpisodes
    from oracle import denet_oracle as O
    torch.manual_seed(0)
    m = DENet(backbone=SyntheticBackbone(), maximum_num_classes=21).eval()
    Is, Ys, Iq, Yq, cls, Ys_full, Yq_full = SyntheticEpisodes(1, 2, 1, 2, 21, 65, split='val').batch(0)
    with torch.no_grad():
        N, way, shot = Is.shape[:3]
        Fs = m.features(Is.reshape(-1, *Is.shape[3:])).reshape(N, way, shot, 256, 9, 9)
        Fq = m.features(Iq)
        state = {k: v.detach().clone() for k, v in m.estimator.state_dict().items()}
        ref_full, ref_bin, _ = O.head_forward(Fq, Fs, Ys, cls, state)
    m = m.to(dev)
    with torch.no_grad():
        full, binl = m(Is.to(dev), Ys.to(dev), Iq.to(dev), cls.to(dev))
    assert full.shape == (2, 21, 9, 9) and binl.shape == (2, 2, 9, 9)
    assert rel_err(full.cpu(), ref_full) < TOL          # MIOpen vs oneDNN convolutions upstream
    assert rel_err(binl.cpu(), ref_bin) < TOL
    assert set(m.vis) == {"hidden_Fs", "hidden_Fq"}


def test_trainer_train_val_test_roundtrip(dev, tmp_path):
    """DENetTrainer: a few optimisation steps lower the loss on a fixed synthetic stream, the
    checkpoint has the reference's keys, and test() runs the 2-way path with per-way BinaryIOU."""
    from denet_amd.trainer import DENetTrainer, get_trainer
    cfg = {
        "model": {"arch": "DENet", "backbone": "synthetic", "maximum_num_classes": 21},
        "data": {"train_dataset": "synthetic", "val_dataset": "synthetic", "img_size": 65, "shot": 2,
                 "batch_size": 4, "batches_per_epoch": 3, "val_batches": 1},
        "train": {"trainer": "DENetTrainer", "device": 0, "num_steps": 6, "log_steps": 100, "val_steps": 6,
                  "lambda": 2.0, "logdir": str(tmp_path),
                  "optimizer": {"name": "sgd", "lr": 0.05, "momentum": 0.9}, "loss": {"name": "loss_denet"}},
    }
    assert get_trainer("DENetTrainer") is DENetTrainer
    torch.manual_seed(0)
    tr = DENetTrainer(cfg)
    batch = tr.train_loader.batch(0)
    with torch.no_grad():
        first = tr.step(batch, train=False)                # eval mode: deterministic loss
    # reference-shaped outputs by default: up-sampled to the label size (trainer/trainers.py:954-956,1003-1005)
    assert len(first) == 5 and first[0].shape == (4, 2, 65, 65) and first[1].shape == (4, 21, 65, 65)
    assert abs(first[2] - (first[3] + first[4])) < 1e-5
    before = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    out = tr.step(batch, train=True)
    assert all(np.isfinite(v) for v in out[2:])
    # one optimiser step moved every trainable tensor (head through the HIP backward, conv stack
    # through dFq/dFs), and left the frozen backbone alone; labels are random, so the loss itself
    # sits at log(K) and is not expected to fall
    after = tr.model.state_dict()
    for k, v in before.items():
        moved = not torch.equal(after[k], v)
        if k.startswith("backbone."):
            assert not moved, k
        elif k.endswith("weight") or k.endswith("bias"):
            assert moved, k
    assert float(tr.bucket.flat.abs().sum()) > 0
    tr.train()
    ck = torch.load(str(tmp_path / "latest_model.bin"), weights_only=True)
    assert set(ck) == {"best_step", "epoch", "best_bIOU", "model", "optimizer", "scheduler"}
    assert "estimator.gam.gate.0.weight" in ck["model"]
    tcfg = {
        "model": {"arch": "DENet", "backbone": "synthetic", "maximum_num_classes": 21},
        "data": {"dataset": "synthetic", "img_size": 65, "fold": 0, "way": 2, "shot": 2, "batch_size": 4,
                 "num_batches": 2},
        "test": {"trainer": "DENetTrainer", "device": 0, "id": "t", "best": True,
                 "ckpt_file": str(tmp_path / "latest_model.bin")},
    }
    te = DENetTrainer(train_config=None, test_config=tcfg)
    out = te.step(te.test_loader.batch(0), train=False)
    assert len(out) == 2 and out[0].shape == (8, 2, 65, 65) and out[1].shape == (4, 21, 65, 65)
    mb, mf = te.test()
    assert 0.0 <= mb <= 1.0 and 0.0 <= mf <= 1.0
    # opt-out: outputs stay at feature resolution, the metric kernels up-sample on the fly -> same counts
    tcfg["test"]["materialize_upsampled"] = False
    tl = DENetTrainer(train_config=None, test_config=tcfg)
    out = tl.step(tl.test_loader.batch(0), train=False)
    assert out[0].shape == (8, 2, 9, 9) and out[1].shape == (4, 21, 9, 9)
    mb2, mf2 = tl.test()
    # the two paths may only disagree on pixels whose top-2 margin is below 1e-4: count those
    # pixels on the materialised outputs and bound the difference of the two paths' COUNTS by them
    from denet_amd.metrics import BinaryIOU, FullIOU
    fa, fb, ba, bb, sub_f, sub_b = FullIOU(21), FullIOU(21), BinaryIOU(), BinaryIOU(), 0, 0
    for j in range(len(te.test_loader)):
        batch = te.test_loader.batch(j)
        Yq, cls_idx, Yq_full = batch[3], batch[4], batch[6]
        up_b, up_f = te.step(batch, train=False)
        lo_b, lo_f = tl.step(batch, train=False)
        fa.update(Yq_full, up_f)
        fb.update(Yq_full, lo_f)
        top = up_f.topk(2, dim=1)[0]
        sub_f += int((((top[:, 0] - top[:, 1]) <= 1e-4).cpu() & (Yq_full != 255)).sum())
        n, way = Yq.shape[:2]
        ub, lb = up_b.reshape(n, way, *up_b.shape[-3:]), lo_b.reshape(n, way, *lo_b.shape[-3:])
        for i in range(way):
            ba.update(Yq[:, i], ub[:, i].contiguous(), cls_idx[:, i].numpy())
            bb.update(Yq[:, i], lb[:, i].contiguous(), cls_idx[:, i].numpy())
        sub_b += int(((up_b[:, 0] - up_b[:, 1]).abs() <= 1e-4).sum())
    assert int(np.abs(fa.counts() - fb.counts()).sum()) <= 2 * sub_f          # a flip moves one count to another bin
    assert list(ba.inter) == list(bb.inter)
    assert sum(abs(ba.inter[c] - bb.inter[c]) + abs(ba.union[c] - bb.union[c]) for c in ba.inter) <= 2 * sub_b
    if sub_f == 0 and sub_b == 0:
        assert mb2 == mb and mf2 == mf
    print(f"materialised vs fused metric paths: {sub_f} / {sub_b} sub-margin pixels (full / binary)")
    with pytest.raises(ValueError):
        DENetTrainer(cfg, tcfg)


def test_step_outputs_match_reference_upsampling(dev, golden_dir):
    """T1: the tensors DENetTrainer.step returns are the reference's F.interpolate(bilinear,
    align_corners=True) of both logit tensors (trainer/trainers.py:954-956), captured from ATen."""
    from denet_amd import ops
    from denet_amd.head import WeightEstimator
    z = np.load(golden_dir + "/trainer_glue.npz")
    N, way, shot, K, c, h, w, H, W = [int(v) for v in z["shape"]]
    we = WeightEstimator(K, c).to(dev).eval()
    we.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")})
    with torch.no_grad():
        full, binl = we.forward_with_binary(torch.from_numpy(z["in:Fq"]).to(dev), torch.from_numpy(z["in:Fs"]).to(dev),
                                            torch.from_numpy(z["in:Ys"]).to(dev), torch.from_numpy(z["in:label"]).to(dev))
        full_up = ops.upsample_bilinear(full, (H, W)).cpu()
        bin_up = ops.upsample_bilinear(binl, (H, W)).cpu()
    assert full_up.shape == (N, K, H, W) and bin_up.shape == (N * way, 2, H, W)
    assert_close(full_up, z["out:full_up"], 2e-5, "full_up")
    assert_close(bin_up, z["out:bin_up"], 2e-5, "bin_up")


def test_process_nway_and_nway_fusion_match_reference(dev, golden_dir):
    """N4: process_nway (trainer/utils.py:128-154) bit for bit against the fixture built around the
    reference's own get_binary_logits, and the n-way test fusion of PatternTrainer.test
    (trainer/trainers.py:831-873) with per-way BinaryIOU against the reference's metric class."""
    from denet_amd.metrics import BinaryIOU
    from denet_amd.operations import process_nway
    from denet_amd.trainer import nway_fusion_update
    z = np.load(golden_dir + "/trainer_glue.npz")
    bi = BinaryIOU()
    for u in range(2):
        pre = torch.from_numpy(z[f"in:nway_pre{u}"]).to(dev)
        way = pre.shape[1] // 2
        assert torch.equal(process_nway(pre, way).cpu(), torch.from_numpy(z[f"out:nway_fused{u}"]))
        nway_fusion_update(bi, pre, torch.from_numpy(z[f"in:nway_Yq{u}"]), torch.from_numpy(z[f"in:nway_cls{u}"]), way)
    assert list(bi.inter.keys()) == z["out:nway_keys"].tolist()                  # dict insertion order
    assert [bi.inter[k] for k in bi.inter] == z["out:nway_inter"].tolist()
    assert [bi.union[k] for k in bi.union] == z["out:nway_union"].tolist()
    assert bi.mean_iou() == float(z["out:nway_mean"]) and bi.mean_bf_iou() == float(z["out:nway_bf"])


def test_standalone_gam_and_map_are_differentiable(dev):
    """model/common.py:170-246: the reference's GAM / MaskedAveragePooling are ordinary differentiable modules
    that other heads reuse; called on their own here they must carry gradients to the embedding and to their
    parameters (oracle = the same lines on CPU autograd)."""
    from denet_amd.head import GAM, MaskedAveragePooling
    from oracle import denet_oracle as O
    z, d, ep, params, gm, wm = load_case("eval_2way_3shot")
    B = d["N"] * d["way"] * d["shot"]
    Fm = ep["Fs"].float().reshape(B, d["c"], d["h"], d["w"])
    Ym = ep["Ys"].reshape(B, d["H"], d["W"])
    gam, mp = GAM(d["c"]).to(dev).eval(), MaskedAveragePooling(depth=d["c"]).to(dev)
    sd = {k[4:]: v for k, v in params.items() if k.startswith("gam.")}
    gam.load_state_dict(sd)
    mp.load_state_dict({k[4:]: v for k, v in params.items() if k.startswith("map.")})
    x = Fm.to(dev).requires_grad_(True)
    out = mp(gam(x, Ym.to(dev)), Ym.to(dev))                    # gate * Fs -> masked average -> linear
    out.square().sum().backward()
    # oracle
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xc = Fm.clone().requires_grad_(True)
    ref = O.masked_avg_pool_project(O.gate(O.gam_attention(xc, Ym), p) * xc, Ym, p)
    ref.square().sum().backward()
    assert_close(out.detach().cpu(), ref.detach(), 2e-5, "proto")
    assert_close(x.grad.cpu(), xc.grad, 1e-4, "d embedding")
    assert_close(mp.linear.weight.grad.cpu(), p["map.linear.weight"].grad, 1e-4, "d map.linear.weight")
    assert_close(gam.gate[0].weight.grad.cpu(), p["gam.gate.0.weight"].grad, 1e-4, "d gam.gate.0.weight")
    # the binary logits stay in the graph, and a backward through them fails loudly
    from denet_amd.operations import get_binary_logits
    lg = torch.randn(2, 5, 4, 4, device=dev, requires_grad=True)
    b = get_binary_logits(lg, torch.tensor([1, 3], device=dev))
    assert b.requires_grad
    with pytest.raises(NotImplementedError):
        b.sum().backward()


def test_denet_bf16_features_from_the_producer(dev):
    """N2: DENet(feature_dtype=bfloat16): the conv stack emits bf16 NCHW and the head's bf16 kernels consume it
    (fp32 accumulate).  Parity: the oracle head fed the SAME bf16-rounded features (1e-3 on logits / parameter
    gradients, 8e-3 on the feature gradient, which is rounded to bf16 on output), and one forward at the real
    producer shape [*,1024,41,41] -> 256 channels at C2's batch."""
    from denet_amd.head import DENet
    from denet_amd.trainer import SyntheticBackbone, SyntheticEpisodes
    from oracle import denet_oracle as O
    torch.manual_seed(0)
    m = DENet(backbone=SyntheticBackbone(), maximum_num_classes=21, feature_dtype=torch.bfloat16).to(dev).eval()
    Is, Ys, Iq, Yq, cls, Ys_full, Yq_full = SyntheticEpisodes(1, 2, 1, 2, 21, 65, split='val').batch(0)
    N, way, shot = Is.shape[:3]
    Fs = m.features(Is.reshape(-1, *Is.shape[3:]).to(dev)).reshape(N, way, shot, 256, 9, 9)
    Fq = m.features(Iq.to(dev))
    assert Fs.dtype == torch.bfloat16 and Fq.dtype == torch.bfloat16 and Fq.is_contiguous()
    Fq_l, Fs_l = Fq.detach().clone().requires_grad_(True), Fs.detach().clone().requires_grad_(True)
    full, binl = m.estimator.forward_with_binary(Fq_l, Fs_l, Ys.to(dev), cls.to(dev))
    assert full.dtype == torch.float32
    g = torch.Generator().manual_seed(3)
    cot = torch.randn(full.shape, generator=g)
    (full * cot.to(dev)).sum().backward()
    # oracle on the same bf16-rounded features
    state = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.estimator.state_dict().items()}
    Fq_c, Fs_c = Fq.detach().float().cpu().requires_grad_(True), Fs.detach().float().cpu().requires_grad_(True)
    ref_full, ref_bin, _ = O.head_forward(Fq_c, Fs_c, Ys, cls, state)
    (ref_full * cot).sum().backward()
    assert rel_err(full.detach().cpu(), ref_full.detach()) < 1e-3
    assert rel_err(binl.detach().cpu(), ref_bin.detach()) < 1e-3
    assert Fq_l.grad.dtype == torch.bfloat16
    assert rel_err(Fq_l.grad.float().cpu(), Fq_c.grad) < 8e-3 and rel_err(Fs_l.grad.float().cpu(), Fs_c.grad) < 8e-3
    for k, p in m.estimator.named_parameters():
        assert rel_err(p.grad.cpu(), state[k].grad) < 1e-3, k
    # the real shape: 8 episodes, 1-way 5-shot, 321 x 321 images -> [*,1024,41,41] -> [*,256,41,41] bf16
    Is8, Ys8, Iq8, _, cls8, _, _ = SyntheticEpisodes(1, 8, 1, 5, 21, 321, split='val').batch(0)
    with torch.no_grad():
        l3 = m.backbone(Iq8.to(dev))['layer3']
        assert l3.shape == (8, 1024, 41, 41)
        full8, bin8 = m(Is8.to(dev), Ys8.to(dev), Iq8.to(dev), cls8.to(dev))
    assert full8.shape == (8, 21, 41, 41) and bin8.shape == (8, 2, 41, 41) and torch.isfinite(full8).all()
    assert m.vis["hidden_Fq"].dtype == torch.bfloat16


def test_smoke_entry_point(dev):
    """__graft_entry__.smoke() itself (the driver runs it before the bench): fused step vs the oracle."""
    import __graft_entry__ as g
    g.smoke()
