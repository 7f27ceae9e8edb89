"""GPU: the reference's own call sequence for a training step -- model(...) -> two F.interpolate -> loss_func ->
loss_k.backward() (trainer/trainers.py:951-975) -- lands on the fused seven-launch head step when driven through
the drop-in API (DENet.forward_step / WeightEstimator.head_step / DENetTrainer.step), plus the loss-function
arguments the reference's signatures take but none of its call sites use (loss/losses.py:19,47)."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, load_case, rel_err

pytestmark = pytest.mark.gpu
ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
         "map.linear.weight", "map.linear.bias"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _feature_stub_model(K, c, dev):
    """DENet whose producer is the identity: the 'images' ARE the feature maps, so forward_step's gradients
    w.r.t. its image arguments are dFs / dFq and can be compared with the oracle's."""
    from denet_amd.head import DENet

    class FeatureStub(DENet):
        def features(self, images):
            return images

    m = FeatureStub(backbone=torch.nn.Identity(), maximum_num_classes=K, depth=c, visualize=False)
    return m.to(dev)


@pytest.mark.parametrize("name", ["eval_1way_5shot", "eval_2way_3shot"])
def test_forward_step_is_one_fused_node_and_matches_the_reference(name, dev):
    """forward_step == the fixture captured from the reference's estimator + up-sampling + loss wrapper +
    autograd, and bit-equal to ops.HeadStepFunction on the same tensors (it IS that node)."""
    from denet_amd import ops
    z, d, ep, params, gm, wm = load_case(name)
    N, way = d["N"], d["way"]
    m = _feature_stub_model(d["K"], d["c"], dev).eval()
    m.estimator.load_state_dict(params)
    Fs = ep["Fs"].float().to(dev).requires_grad_(True)
    Fq = ep["Fq"].float().to(dev).requires_grad_(True)
    lab = ep["label"].reshape(N, way).to(dev)
    lk, lp, full, binl = m.forward_step(Fs, ep["Ys"].to(dev), Fq, lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev),
                                        ep["wgt"].to(dev))
    assert type(lk.grad_fn).__name__ == "HeadStepFunctionBackward"
    lk.backward()
    assert rel_err(full.cpu(), z["out:logits_full"]) < 2e-5
    assert rel_err(binl.cpu(), z["out:logits_binary"]) < 2e-5
    assert rel_err(lk.detach().cpu(), z["out:loss_k"]) < 1e-4 and rel_err(lp.cpu(), z["out:loss_p"]) < 1e-4
    assert_close(Fq.grad.cpu(), z["out:dFq"], 1e-3, "dFq")
    assert_close(Fs.grad.cpu(), z["out:dFs"], 1e-3, "dFs")
    got = {k: p.grad.clone() for k, p in m.estimator.named_parameters()}
    for k, g in got.items():
        assert rel_err(g.cpu(), z["grad:" + k]) < 1e-3, k
    # the same tensors through the autograd node directly: bit for bit
    P = [dict(m.estimator.named_parameters())[k].detach().clone().requires_grad_(True) for k in ORDER]
    Fs2, Fq2 = Fs.detach().clone().requires_grad_(True), Fq.detach().clone().requires_grad_(True)
    lk2, lp2, full2, bin2 = ops.HeadStepFunction.apply(
        Fq2, Fs2, *P, ep["Ys"].to(dev), lab, ep["Yq"].reshape(N * way, d["H"], d["W"]).to(dev),
        ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
    lk2.backward()
    assert torch.equal(lk2, lk) and torch.equal(lp2, lp) and torch.equal(full2, full) and torch.equal(bin2, binl)
    assert torch.equal(Fs2.grad, Fs.grad) and torch.equal(Fq2.grad, Fq.grad)
    for k, p in zip(ORDER, P):
        assert torch.equal(p.grad, got[k]), k
    # validation steps: no graph under no_grad, same numbers, and the workspace goes straight back to the pool
    with torch.no_grad():
        lk3, lp3, full3, _ = m.forward_step(Fs, ep["Ys"].to(dev), Fq, lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev),
                                            ep["wgt"].to(dev))
    assert lk3.grad_fn is None and torch.equal(full3, full) and rel_err(lk3.cpu(), lk.detach().cpu()) < 1e-6
    assert sum(len(v) for v in ops.workspace_pool._free.values()) >= 1
    # ... also when this is the process's first step (nothing left in the pool by earlier tests): the node sees the
    # surrounding no_grad (ops.head_step), runs the forward without gradient slabs and hands its workspace back
    ops.workspace_pool.clear()
    with torch.no_grad():
        lk4, _, full4, _ = m.forward_step(Fs, ep["Ys"].to(dev), Fq, lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev),
                                          ep["wgt"].to(dev))
    assert lk4.grad_fn is None and torch.equal(full4, full) and rel_err(lk4.cpu(), lk.detach().cpu()) < 1e-6
    assert sum(len(v) for v in ops.workspace_pool._free.values()) == 1


def test_train_mode_forward_step_draws_in_kernel_dropout(dev):
    """Train mode: ONE seed per step, masks generated in the kernels; equals the same step fed the masks
    denet_dropout_mask writes for that seed through the injected-mask pointers (the path the reference fixture
    train_1way_5shot pins, tests/test_gpu_dropout.py)."""
    from denet_amd import ops
    z, d, ep, params, gm, wm = load_case("train_1way_5shot")
    N, way, shot, K, c = d["N"], d["way"], d["shot"], d["K"], d["c"]
    m = _feature_stub_model(K, c, dev).train()
    m.estimator.load_state_dict(params)
    Fs, Fq = ep["Fs"].float().to(dev), ep["Fq"].float().to(dev)
    lab = ep["label"].reshape(N, way).to(dev)
    args = (ep["Ys"].to(dev), Fq, lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev))
    torch.manual_seed(5)
    lk, lp, full, binl = m.forward_step(Fs, *args)
    lk.backward()
    seed = m.estimator.last_dropout_seed
    g1 = {k: p.grad.clone() for k, p in m.estimator.named_parameters()}
    torch.manual_seed(6)
    lkb, _, fullb, _ = m.forward_step(Fs, *args)
    assert m.estimator.last_dropout_seed != seed and not torch.equal(fullb, full)      # a new seed every step
    gmask = ops.dropout_mask(seed, ops.DROPOUT_GATE, 0.5, (N * way * shot, c), dev)
    wmask = ops.dropout_mask(seed, ops.DROPOUT_WEIGHT, 0.5, (N, K, c), dev)
    m.zero_grad()
    lk2, lp2, full2, bin2 = m.estimator.head_step(Fq, Fs, args[0], lab, args[3], args[4], args[5],
                                                  gate_mask=gmask, weight_mask=wmask)
    lk2.backward()
    assert torch.equal(full2, full) and torch.equal(lk2, lk) and torch.equal(bin2, binl)
    for k, p in m.estimator.named_parameters():
        assert torch.equal(p.grad, g1[k]), k


def test_trainer_step_runs_the_fused_step(dev, tmp_path):
    """DENetTrainer.step (train and val) = conv stack + ONE HeadStepFunction: the estimator gradients the
    trainer's bucket holds after a train step equal, bit for bit, those of the hand-driven sequence
    features -> HeadStepFunction(drop=seed) -> backward under the same RNG state; the loss equals the oracle's on
    the same features; resuming restores the scheduler at the checkpoint's epoch."""
    from denet_amd import ops
    from denet_amd.trainer import DENetTrainer
    from oracle import denet_oracle as O
    cfg = {
        "model": {"arch": "DENet", "backbone": "synthetic", "maximum_num_classes": 21},
        "data": {"train_dataset": "synthetic", "val_dataset": "synthetic", "img_size": 65, "shot": 2,
                 "batch_size": 4, "batches_per_epoch": 2, "val_batches": 1},
        "train": {"trainer": "DENetTrainer", "device": 0, "num_steps": 6, "log_steps": 100, "val_steps": 2,
                  "lambda": 2.0, "logdir": str(tmp_path),
                  "optimizer": {"name": "sgd", "lr": 0.05, "momentum": 0.9},
                  "scheduler": {"name": "StepLR", "step_size": 1, "gamma": 0.5},
                  "loss": {"name": "loss_denet"}},
    }
    torch.manual_seed(0)
    tr = DENetTrainer(cfg)
    assert tr.loss_func.fused_head_step
    batch = tr.train_loader.batch(0)
    Is, Ys, Iq, Yq, cls, Ys_full, Yq_full = [t.to(dev) for t in batch]
    N, way, shot = Is.shape[:3]
    wgt = torch.ones(N, 21, device=dev)
    wgt[torch.arange(N, device=dev), cls.reshape(-1)] = 2.0
    model, est = tr.model, tr.model.estimator
    names = ["estimator." + k for k in ORDER]
    params = dict(model.named_parameters())

    # --- eval: loss against the oracle on the same features
    model.eval()
    with torch.no_grad():
        Fs = model.features(Is.reshape(-1, *Is.shape[3:])).reshape(N, way, shot, 256, 9, 9)
        Fq = model.features(Iq)
        out = tr.step(batch, train=False)
    state = {k: v.detach().cpu() for k, v in est.state_dict().items()}
    ref = O.head_step(Fq.cpu(), Fs.cpu(), batch[1], batch[4], batch[3], batch[6], wgt.cpu(), state, backward=False)
    assert abs(out[3] - float(ref["loss_k"])) < 1e-4 * abs(float(ref["loss_k"]))
    assert abs(out[4] - float(ref["loss_p"])) < 1e-4 * abs(float(ref["loss_p"]))
    assert out[0].shape == (4, 2, 65, 65) and out[1].shape == (4, 21, 65, 65)

    # --- train: the hand-driven sequence first (parameters untouched), then the trainer's own step
    model.train()
    torch.manual_seed(123)
    Fs = model.features(Is.reshape(-1, *Is.shape[3:])).reshape(N, way, shot, 256, 9, 9)
    Fq = model.features(Iq)
    drop = est.dropout_config()
    lk, lp, _, _ = ops.HeadStepFunction.apply(Fq, Fs, *[params[n] for n in names], Ys, cls.reshape(N, way),
                                              Yq.reshape(N * way, 65, 65), Yq_full, wgt, None, None, drop)
    tr.bucket.zero()
    lk.backward()
    hand = {n: params[n].grad.clone() for n in names}
    hand_loss = float(lk.detach())
    # (MIOpen's convolutions are not bit-reproducible from call to call -- the same features() call twice differs
    # in the last bits -- so the bit-for-bit comparison runs on cached features, the real producer within 1e-5)
    real_features = model.features
    cache = {N * way * shot: Fs.detach().reshape(-1, 256, 9, 9), N: Fq.detach()}      # (8 support images, 4 queries)
    model.features = lambda images: cache[images.shape[0]].clone().requires_grad_(True)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(123)
    out = tr.step(batch, train=True)
    model.features = real_features
    assert est.last_dropout_seed == drop[0]
    assert out[3] == hand_loss
    for n in names:
        assert torch.equal(params[n].grad, hand[n]), n
    # ... and with the real producer: same gradients up to the convolutions' noise, and dFq / dFs reach the stack
    model.load_state_dict(sd)
    torch.manual_seed(123)
    out = tr.step(batch, train=True)
    assert abs(out[3] - hand_loss) < 1e-5 * abs(hand_loss)
    for n in names:
        assert rel_err(params[n].grad, hand[n]) < 1e-4, n
    assert float(params["embedding.0.weight"].grad.abs().sum()) > 0

    # --- resume: epoch-pinned scheduler (trainer/trainers.py:209): lr of epoch e after a reload = lr of epoch e
    tr.train()
    lrs = {}
    ck = torch.load(str(tmp_path / "latest_model.bin"), weights_only=True)
    e = ck["epoch"]
    cfg2 = {**cfg, "train": {**cfg["train"], "resume": str(tmp_path / "latest_model.bin")}}
    tr2 = DENetTrainer(cfg2)
    assert tr2.start_epoch == e and tr2.scheduler.last_epoch == ck["scheduler"]["last_epoch"]
    from denet_amd.schedulers import tick
    tick(tr2.scheduler, e)
    assert abs(tr2.optimizer.param_groups[0]["lr"] - 0.05 * 0.5 ** e) < 1e-12


def test_loss_arguments_beyond_the_reference_defaults(dev):
    """reduction='sum' / 'none' and ignore_index != 255 for both losses, against the reference's own functions
    (tests/golden/loss_variants.npz, written by oracle/gen_golden.py loss_variants)."""
    from denet_amd import losses
    za, z = np.load(GOLDEN + "/api_cases.npz"), np.load(GOLDEN + "/loss_variants.npz")
    full = torch.from_numpy(za["out:full"]).to(dev)
    yf = torch.from_numpy(za["in:Yq_full"]).to(dev)
    cw = torch.from_numpy(za["in:class_weight"]).to(dev)
    x = full.clone().requires_grad_(True)
    l = losses.cross_entropy2d(x, yf, weight=cw, reduction='sum')
    l.backward()
    assert rel_err(l.detach().cpu(), z["out:ce_sum"]) < 1e-5
    assert_close(x.grad.cpu(), z["out:ce_sum_grad"], 1e-3, "d ce_sum")
    none = losses.cross_entropy2d(full, yf, weight=cw, reduction='none')
    assert_close(none.cpu(), z["out:ce_none"], 2e-5, "ce_none", afloor=1e-5)
    yf2 = torch.from_numpy(z["in:Yq_full_no255"]).to(dev)
    x = full.clone().requires_grad_(True)
    l = losses.cross_entropy2d(x, yf2, weight=cw, ignore_index=2)
    l.backward()
    assert rel_err(l.detach().cpu(), z["out:ce_ign2"]) < 1e-5
    assert_close(x.grad.cpu(), z["out:ce_ign2_grad"], 1e-3, "d ce_ign2")
    yq = torch.from_numpy(za["in:Yq"])[:, 1].contiguous().to(dev)
    binl = torch.from_numpy(za["out:bin_base_true"]).to(dev)
    assert rel_err(losses.binary_focal_loss(binl, yq, reduction='sum').cpu(), z["out:focal_sum"]) < 1e-5
    assert_close(losses.binary_focal_loss(binl, yq, reduction='none').cpu(), z["out:focal_none"], 2e-5, "focal_none",
                 afloor=1e-5)
    assert rel_err(losses.binary_focal_loss(binl, yq, alpha=0.3, ignore_index=1).cpu(), z["out:focal_ign1_a03"]) < 1e-5
    with pytest.raises(ValueError):
        losses.cross_entropy2d(full, yf, reduction='median')


def test_standalone_pooling_long_rows_fp32(dev):
    """Stand-alone denet_masked_sum_fwd on fp32 rows of more than 2048 positions: the rolling loop over a
    row-major mask image (the fused step parks the gate weights behind that image; stand-alone there is
    padding instead)."""
    from denet_amd import ops
    g = torch.Generator().manual_seed(11)
    B, C, h, w = 3, 64, 50, 53                       # P = 2650 > 2048, odd
    Fs = torch.randn(B, C, h, w, generator=g)
    mask = (torch.rand(B, h * w, generator=g) > 0.4).float()
    S = ops.masked_sum(Fs.to(dev), mask.to(dev))
    ref = (Fs.reshape(B, C, -1).double() * mask.unsqueeze(1).double()).sum(-1)
    assert rel_err(S.cpu(), ref) < 2e-6


def test_head_step_follows_the_callers_stream(dev):
    """The library launches on the stream torch has current (denet_amd._lib.current_stream_handle), forward AND
    backward (autograd runs the backward node under the forward's stream).  The inputs are produced on a side
    stream BEHIND a few milliseconds of other work and consumed there at once: kernels launched on any other
    stream would read the zeros the buffers held before."""
    from denet_amd import ops
    from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
    from oracle import denet_oracle as O
    order = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
             "map.linear.weight", "map.linear.bias"]
    s = EpisodeShape("side", N=2, way=1, shot=3, K=21, c=64, h=17, w=19, H=65, W=73, config_index=31, lam=1.0)
    ep = make_episodes(s, ignore_rows=2)
    state = make_estimator_state(s.K, s.c)
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    pinned = {k: ep[k].pin_memory() for k in ("Fq", "Fs")}
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        Fq = torch.zeros_like(ep["Fq"], device=dev)
        Fs = torch.zeros_like(ep["Fs"], device=dev)
        busy = torch.randn(4096, 4096, device=dev)
        for _ in range(20):                                   # ~ms of work in front of the copies
            busy = busy @ busy * 1e-4
        Fq.copy_(pinned["Fq"], non_blocking=True)
        Fs.copy_(pinned["Fs"], non_blocking=True)
        Fq.requires_grad_(True)
        Fs.requires_grad_(True)
        P = [state[k].to(dev).requires_grad_(True) for k in order]
        lk, lp, logits, binl = ops.HeadStepFunction.apply(
            Fq, Fs, *P, ep["Ys"].to(dev), ep["label"].reshape(s.N, 1).to(dev), ep["Yq"].to(dev),
            ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
        lk.backward()
    side.synchronize()
    assert rel_err(logits.cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(lk.detach().cpu(), ref["loss_k"]) < 1e-4
    assert rel_err(Fq.grad.cpu(), ref["dFq"]) < 2e-5 and rel_err(Fs.grad.cpu(), ref["dFs"]) < 2e-5
    for k, t in zip(order, P):
        assert rel_err(t.grad.cpu(), ref["d:" + k]) < 2e-5, k


def test_head_step_rejects_mismatched_shapes(dev):
    """The kernels index raw memory from the step's sizes; a tensor of another size must be refused on the host
    (ValueError) instead of being read out of bounds: label maps of another size than Ys, a short weight vector,
    a feature map of another channel count."""
    from denet_amd import ops
    from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
    order = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
             "map.linear.weight", "map.linear.bias"]
    s = EpisodeShape("chk", N=2, way=1, shot=2, K=21, c=64, h=9, w=11, H=33, W=41, config_index=5, lam=1.0)
    ep = {k: v.to(dev) for k, v in make_episodes(s, ignore_rows=1).items()}
    P = [v.to(dev) for v in (make_estimator_state(s.K, s.c)[k] for k in order)]
    lab = ep["label"].reshape(s.N, 1)

    def run(**over):
        a = dict(Fq=ep["Fq"], Fs=ep["Fs"], Ys=ep["Ys"], Yq=ep["Yq"], Yq_full=ep["Yq_full"], wgt=ep["wgt"], P=P)
        a.update(over)
        return ops.head_step(a["Fq"], a["Fs"], *a["P"], a["Ys"], lab, a["Yq"], a["Yq_full"], a["wgt"], None, None)

    run()                                                                   # the matching shapes pass
    with pytest.raises(ValueError, match="Yq_full"):
        run(Yq_full=ep["Yq_full"][:, :-1])
    with pytest.raises(ValueError, match="Yq"):
        run(Yq=ep["Yq"][..., :-2])
    with pytest.raises(ValueError, match="class weights"):
        run(wgt=ep["wgt"][:, :-1])
    with pytest.raises(ValueError, match="Fq"):
        run(Fq=ep["Fq"][:, :32])
    with pytest.raises(ValueError, match="W2"):
        run(P=P[:3] + [P[3][:, :32]] + P[4:])
