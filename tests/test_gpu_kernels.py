"""GPU parity, kernel by kernel, through the C ABI (denet_amd.ops -> libdenet_hip.so)
against the CPU oracle and the golden fixtures captured from the reference.

Tolerances: fp32 logits / losses / gradients within 1e-3 relative
(BASELINE.json north_star); in practice they agree to ~1e-6.  Integer work
(mask resize, confusion matrices, IoU counts) is bit-exact.
"""
import numpy as np
import pytest
import torch

from helpers import CASES, assert_close, load_case, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-3
TIGHT = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _params(params, dev):
    order = ["gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
             "map.linear.weight", "map.linear.bias", "weight"]
    return tuple(params[k].to(dev).contiguous() for k in order)


@pytest.mark.parametrize("name", CASES)
def test_forward_stages(name, dev):
    from denet_amd import ops
    from oracle import denet_oracle as O
    z, d, ep, params, gm, wm = load_case(name)
    N, way, shot, K, c, h, w, H, W = (d[k] for k in ("N", "way", "shot", "K", "c", "h", "w", "H", "W"))
    B, P = N * way * shot, h * w
    Fs = ep["Fs"].float().reshape(B, c, h, w)
    Ys = ep["Ys"].reshape(B, H, W)
    # --- K0: mask resize, bit exact vs ATen nearest
    mask, cnt = ops.mask_resize(Ys.to(dev), h, w)
    m_ref = O.resize_mask(Ys, h, w).reshape(B, P)
    assert torch.equal(mask.cpu(), m_ref)
    assert torch.equal(cnt.cpu(), m_ref.sum(1))
    # --- K1
    S = ops.masked_sum(Fs.to(dev), mask)
    S_ref = (m_ref.unsqueeze(1) * Fs.reshape(B, c, P)).sum(-1)
    assert rel_err(S.cpu(), S_ref) < TIGHT
    if "out:att" in z.files:
        assert rel_err(S.cpu() / P, z["out:att"]) < TIGHT
    # --- K2
    lab = ep["label"].reshape(N, way).to(dev)
    gmd = gm.to(dev) if gm is not None else None
    wmd = wm.to(dev).contiguous() if wm is not None else None
    Wn, saved = ops.proto_extend(S, cnt, _params(params, dev), lab, N, way, shot, K, P, gmd, wmd)
    if "out:gate" in z.files:
        assert rel_err(saved[1].cpu(), z["out:gate"]) < TIGHT
    Wn_ref = O.extend_weights(ep["Fs"].float(), ep["Ys"], ep["label"].reshape(N, way), params,
                              gm.reshape(B, c, 1, 1) if gm is not None else None, wm)
    assert rel_err(Wn.cpu(), Wn_ref) < TIGHT
    if "out:weights" in z.files:
        assert rel_err(Wn.cpu(), z["out:weights"]) < TIGHT
    # --- K3 + K3e
    logits, binl = ops.classify(ep["Fq"].float().to(dev), Wn, lab, way)
    assert rel_err(logits.cpu(), z["out:logits_full"]) < TIGHT
    assert rel_err(binl.cpu(), z["out:logits_binary"]) < TIGHT
    # --- K4 / K5
    loss_k, per, scale, _, _ = ops.upsample_wce(logits, ep["Yq_full"].to(dev), ep["wgt"].to(dev).contiguous())
    assert rel_err(loss_k.cpu(), z["out:loss_k"]) < 1e-4
    loss_p = ops.upsample_focal(binl, ep["Yq"].to(dev))
    assert rel_err(loss_p.cpu(), z["out:loss_p"]) < 1e-4
    # --- plain up-sampling vs ATen
    up = ops.upsample_bilinear(logits, (H, W))
    up_ref = O.upsample(torch.from_numpy(z["out:logits_full"]), (H, W))
    assert rel_err(up.cpu(), up_ref) < TIGHT


@pytest.mark.parametrize("name", CASES)
def test_backward_chain(name, dev):
    from denet_amd import ops
    z, d, ep, params, gm, wm = load_case(name)
    N, way, shot, K, c, h, w, H, W = (d[k] for k in ("N", "way", "shot", "K", "c", "h", "w", "H", "W"))
    order = ["gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
             "map.linear.weight", "map.linear.bias", "weight"]
    P_ = {k: params[k].to(dev).contiguous().requires_grad_(True) for k in order}
    Fq = ep["Fq"].float().to(dev).requires_grad_(True)
    Fs = ep["Fs"].float().to(dev).requires_grad_(True)
    gmd = gm.to(dev) if gm is not None else None
    wmd = wm.to(dev).contiguous() if wm is not None else None
    lab = ep["label"].reshape(N, way).to(dev)
    logits, binl, Wn = ops.EstimatorFunction.apply(
        Fq, Fs, P_["weight"], P_["gam.gate.0.weight"], P_["gam.gate.0.bias"], P_["gam.gate.2.weight"],
        P_["gam.gate.2.bias"], P_["map.linear.weight"], P_["map.linear.bias"], ep["Ys"].to(dev), lab, gmd, wmd, True)
    loss = ops.UpsampleWCEFunction.apply(logits, ep["Yq_full"].to(dev), ep["wgt"].to(dev).contiguous(), False)
    loss.backward()
    assert rel_err(loss.detach().cpu(), z["out:loss_k"]) < 1e-4
    assert rel_err(Fq.grad.cpu(), z["out:dFq"]) < TOL
    assert rel_err(Fs.grad.cpu(), z["out:dFs"]) < TOL
    for k in order:
        assert_close(P_[k].grad.cpu(), z["grad:" + k], TOL, k)


@pytest.mark.parametrize("name", CASES)
def test_fused_head_step(name, dev):
    from denet_amd import ops
    z, d, ep, params, gm, wm = load_case(name)
    N, way = d["N"], d["way"]
    order = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
             "map.linear.weight", "map.linear.bias"]
    P_ = [params[k].to(dev).contiguous().requires_grad_(True) for k in order]
    Fq = ep["Fq"].float().to(dev).requires_grad_(True)
    Fs = ep["Fs"].float().to(dev).requires_grad_(True)
    gmd = gm.to(dev) if gm is not None else None
    wmd = wm.to(dev).contiguous() if wm is not None else None
    lab = ep["label"].reshape(N, way).to(dev)
    lk, lp, logits, binl = ops.HeadStepFunction.apply(
        Fq, Fs, *P_, ep["Ys"].to(dev), lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev),
        ep["wgt"].to(dev).contiguous(), gmd, wmd)
    lk.backward()
    assert rel_err(logits.cpu(), z["out:logits_full"]) < TIGHT
    assert rel_err(binl.cpu(), z["out:logits_binary"]) < TIGHT
    assert rel_err(lk.detach().cpu(), z["out:loss_k"]) < 1e-4
    assert rel_err(lp.cpu(), z["out:loss_p"]) < 1e-4
    assert_close(logits.cpu(), z["out:logits_full"], TIGHT, "logits_full")
    assert_close(Fq.grad.cpu(), z["out:dFq"], TOL, "dFq")
    assert_close(Fs.grad.cpu(), z["out:dFs"], TOL, "dFs")
    for k, t in zip(order, P_):
        assert_close(t.grad.cpu(), z["grad:" + k], TOL, k)


def test_metrics_boundary_bit_exact(dev, golden_dir):
    """Predictions in -> counts out, against the reference's own FullIOU/BinaryIOU."""
    from denet_amd import ops
    z = np.load(golden_dir + "/metrics.npz")
    K = z["out:mat"].shape[0]
    mat = torch.zeros(K, K, dtype=torch.int64, device=dev)
    inter, union, order = {}, {}, []
    for u in range(3):
        ops.confmat_update(mat, torch.from_numpy(z[f"in:yt{u}"]).to(dev), torch.from_numpy(z[f"in:lp{u}"]).to(dev))
        cnt = ops.binary_counts(torch.from_numpy(z[f"in:yb{u}"]).to(dev), torch.from_numpy(z[f"in:lb{u}"]).to(dev)).cpu()
        for (st, i, un), c in zip(cnt.tolist(), z[f"in:cls{u}"].tolist()):
            if st == 0:
                continue
            if c not in inter:
                order.append(c)
            inter[c] = inter.get(c, 0) + i
            union[c] = union.get(c, 0) + un
    assert np.array_equal(mat.cpu().numpy(), z["out:mat"])
    assert order == list(z["out:biou_keys"])
    assert [float(inter[c]) for c in order] == list(z["out:biou_inter"])
    assert [float(union[c]) for c in order] == list(z["out:biou_union"])


@pytest.mark.parametrize("name", CASES)
def test_metrics_on_upsampled_logits(name, dev):
    """API-level metric kernels on the reference's own up-sampled logits are bit exact."""
    from denet_amd import ops
    from oracle import denet_oracle as O
    z, d, ep, params, gm, wm = load_case(name)
    K, H, W = d["K"], d["H"], d["W"]
    f_up = O.upsample(torch.from_numpy(z["out:logits_full"]), (H, W))
    mat = torch.zeros(K, K, dtype=torch.int64, device=dev)
    ops.confmat_update(mat, ep["Yq_full"].to(dev), f_up.to(dev))
    assert np.array_equal(mat.cpu().numpy(), z["out:confmat"])
    b_up = O.upsample(torch.from_numpy(z["out:logits_binary"]), (H, W))
    cnt = ops.binary_counts(ep["Yq"].reshape(-1, H, W).to(dev), b_up.to(dev)).cpu().numpy()
    ref = O.binary_iou_counts(ep["Yq"].reshape(-1, H, W).numpy(), b_up.max(1)[1].numpy())
    for (st, i, u), (skip, ri, ru) in zip(cnt.tolist(), ref):
        assert (st == 0) == skip
        if not skip:
            assert (i, u) == (ri, ru)


def test_fused_confmat_planted_margin(dev, golden_dir):
    """End-to-end bit-exact confusion matrix from low-res logits, on the planted-margin
    episode (min top-2 margin asserted >> fp32 noise)."""
    from denet_amd import ops
    z, d, ep, params, gm, wm = load_case("planted_1way_1shot")
    assert float(z["out:min_margin"]) > 1e-4
    K = d["K"]
    mat = torch.zeros(K, K, dtype=torch.int64, device=dev)
    ops.confmat_update(mat, ep["Yq_full"].to(dev), torch.from_numpy(z["out:logits_full"]).to(dev))
    assert np.array_equal(mat.cpu().numpy(), z["out:confmat"])


@pytest.mark.parametrize("name", ["eval_1way_5shot", "eval_2way_3shot"])
def test_engine_deferred_losses_match_eager(name, dev):
    """The allocation-free engine with the loss reduction folded into the first backward launch
    (what bench.py times) gives the same bits as the autograd path, and is stable across steps."""
    from denet_amd import ops
    from denet_amd.engine import HeadStepEngine, PARAM_ORDER
    z, d, ep, params, gm, wm = load_case(name)
    N, way, shot, K, c, h, w, H, W = (d[k] for k in ("N", "way", "shot", "K", "c", "h", "w", "H", "W"))
    P_ = {k: params[k].to(dev).contiguous() for k in PARAM_ORDER}
    Fq, Fs = ep["Fq"].float().to(dev), ep["Fs"].float().to(dev)
    lab = ep["label"].reshape(N, way).to(dev)
    Ys, Yq, Yf, wgt = ep["Ys"].to(dev), ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev).contiguous()
    eng = HeadStepEngine(N, way, shot, K, c, h, w, H, W, dev)
    snaps = []
    for defer in (False, True, True):
        eng.losses.zero_()
        eng.forward(Fq, Fs, Ys, lab, Yq, Yf, wgt, P_, with_grad=True, defer_losses=defer)
        eng.backward(Fq, lab, P_)
        torch.cuda.synchronize()
        snaps.append([eng.losses.clone(), eng.dFq.clone(), eng.dFs.clone(), eng.grad_bucket.clone(),
                      eng.logits_full.clone(), eng.logits_binary.clone()])
    # the backward in two calls (what bench.py does with more than one GPU: the gradient all-reduce
    # starts after phase 1): phase 1 must leave dFs alone and complete everything else, phase 2 only dFs
    eng.losses.zero_()
    eng.forward(Fq, Fs, Ys, lab, Yq, Yf, wgt, P_, with_grad=True, defer_losses=True)
    eng.dFs.fill_(-7.0)
    eng.backward(Fq, lab, P_, phase=1)
    torch.cuda.synchronize()
    assert torch.all(eng.dFs == -7.0)
    assert torch.equal(eng.grad_bucket, snaps[0][3]) and torch.equal(eng.dFq, snaps[0][1])
    eng.grad_bucket.fill_(3.0)
    eng.backward(Fq, lab, P_, phase=2)
    torch.cuda.synchronize()
    assert torch.all(eng.grad_bucket == 3.0)
    eng.grad_bucket.copy_(snaps[0][3])
    snaps.append([eng.losses.clone(), eng.dFq.clone(), eng.dFs.clone(), eng.grad_bucket.clone(),
                  eng.logits_full.clone(), eng.logits_binary.clone()])
    # the pre-bound step (bench.py's loop: arguments converted once) is the same two library calls
    eng.bind_step(Fq, Fs, Ys, lab, Yq, Yf, wgt, P_)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        eng.losses.zero_(); eng.dFs.zero_(); eng.dFq.zero_(); eng.grad_bucket.zero_()
        eng.run_forward(st)
        eng.run_backward(0, st)
        torch.cuda.synchronize()
    snaps.append([eng.losses.clone(), eng.dFq.clone(), eng.dFs.clone(), eng.grad_bucket.clone(),
                  eng.logits_full.clone(), eng.logits_binary.clone()])
    for other in snaps[1:]:
        for a, b in zip(snaps[0], other):
            assert torch.equal(a, b)
    assert rel_err(snaps[0][0][0].cpu(), z["out:loss_k"]) < 1e-4
    assert rel_err(snaps[0][0][1].cpu(), z["out:loss_p"]) < 1e-4
    assert rel_err(snaps[0][1].cpu(), z["out:dFq"]) < TOL
    assert rel_err(snaps[0][2].reshape(ep["Fs"].shape).cpu(), z["out:dFs"]) < TOL
    for k in PARAM_ORDER:
        assert rel_err(eng.grads[k].cpu(), z["grad:" + k]) < TOL, k


@pytest.mark.parametrize("geom", [(9, 9, 65, 65), (5, 13, 33, 97), (6, 7, 41, 50), (7, 9, 7, 9), (4, 5, 7, 5)])
def test_loss_kernel_extreme_logits(geom, dev):
    """Logit swings of hundreds of nats between neighbouring positions: every exponential of a pixel
    underflows against the blended bound (exact-max fallback), and on power-of-two scales (the first two
    geometries) the multiplication chains of the uniform-step path are too steep and the block falls
    back to the generic gradient pass.  Loss and gradient must still match stock PyTorch."""
    import torch.nn.functional as F
    from denet_amd import ops
    h, w, H, W = geom
    N, K = 3, 5
    g = torch.Generator().manual_seed(77 + h)
    logits = torch.randn(N, K, h, w, generator=g)
    hot = torch.randint(0, K, (N, h, w), generator=g)
    logits[0] += 400.0 * F.one_hot(hot[0], K).permute(2, 0, 1)         # a different class wins by 400 at each position
    logits[1] += 120.0 * torch.randn(K, h, w, generator=g)            # steep but mostly representable
    y = torch.randint(0, K, (N, H, W), generator=g)
    y[:, :3] = 255
    wgt = torch.ones(N, K)
    wgt[:, 2] = 2.5
    ref_in = logits.clone().requires_grad_(True)
    up = F.interpolate(ref_in, size=(H, W), mode="bilinear", align_corners=True)
    per = [F.cross_entropy(up[n:n + 1], y[n:n + 1], weight=wgt[n], ignore_index=255) for n in range(N)]
    ref = torch.stack(per).mean()
    ref.backward()
    loss, per_dev, scale, da, db = ops.upsample_wce(logits.to(dev), y.to(dev), wgt.to(dev), grad=True)
    dl = ops.combine_slabs(da, db, scale, None)
    torch.cuda.synchronize()
    assert rel_err(loss.cpu(), ref.detach()) < 1e-5
    assert rel_err(per_dev.cpu(), torch.stack(per).detach()) < 1e-5
    assert rel_err(dl.cpu(), ref_in.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,h,w", [(3, 64, 5, 7), (2, 256, 41, 41), (5, 64, 9, 9), (1, 64, 2, 2), (2, 128, 33, 31)])
def test_chunked_dfs_writer_equals_the_row_writer(dev, dtype, B, C, h, w):
    """K1b's two writers: dFs = mask x coef by rows (denet_masked_sum_bwd) and by contiguous chunks of the flat
    tensor (denet_masked_sum_bwd_chunked, the fused step's writer for large bf16 streams): bit-identical, on shapes
    whose rows straddle the 16-byte vectors in every phase (odd P), whose last chunk is partial, and on a span
    shorter than one chunk; and both equal the product itself."""
    from denet_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C + h)
    mask = (torch.rand(B, h * w, generator=g) > 0.3).float().to(dev)
    coef = torch.randn(B, C, generator=g).to(dev)
    a = ops.masked_sum_bwd(mask, coef, (B, C, h, w), dtype)
    b = ops.masked_sum_bwd(mask, coef, (B, C, h, w), dtype, chunked=True)
    assert torch.equal(a, b)
    ref = (coef.unsqueeze(2) * mask.unsqueeze(1)).reshape(B, C, h, w).to(dtype)
    assert torch.equal(b, ref)


@pytest.mark.parametrize("N,way,K,C,h,w,base", [
    (2, 2, 81, 256, 9, 9, True),       # odd P: the last position rides in the high half of its dword
    (3, 1, 33, 64, 8, 9, True),        # two row tiles, C = 64 (4 k-steps, one request)
    (2, 3, 64, 256, 16, 17, False),    # K a multiple of 32 (no zero rows), base = False takes row 0
    (1, 2, 96, 64, 1, 2, True),        # the smallest P the kernel takes; K = the most its LDS planes hold
    (5, 2, 81, 256, 27, 31, True),     # several blocks per episode, a partly idle last block
])
@pytest.mark.parametrize("feat", ["bf16", "f32"])
def test_classifier_on_the_bf16_matrix_pipe(N, way, K, C, h, w, base, feat, dev):
    """bf16 features with 32 < K <= 96 run on the bf16 matrix pipe (classify_fwd_wide4_bf16_kernel): the fp32 class
    weights enter the MFMA as three bf16 planes whose sum is the fp32 value exactly, so every product is exact and
    the result differs from a float64 product only by fp32 accumulation (model/common.py:390-394).  fp32 features
    of the same shapes (the fp32-MFMA layouts) are held to the same bounds.  Binary logits
    (utils/operations.py:313-338) are selections of the kernel's own logits: bit-exact against them.  Labels
    include a duplicate and one outside [0, K) (ch1 = 0, as the oracle)."""
    from denet_amd import _lib, ops
    from oracle import denet_oracle as O
    pl = _lib.head_step_plan(N, way, 1, K, C, h, w, 4 * h, 4 * w, _lib.BF16 if feat == "bf16" else _lib.F32)
    assert (pl["classify_layout"] == _lib.CLS_WAVE4_BF16) == (feat == "bf16")
    g = torch.Generator().manual_seed(1000 * K + h * w)
    Fq = torch.randn(N, C, h, w, generator=g)
    # (fp32: a spread of magnitudes over the channels too)
    Fq = Fq.bfloat16() if feat == "bf16" else Fq * torch.logspace(1, -3, C).reshape(1, C, 1, 1)
    # weights with full 24-bit significands and a spread of magnitudes (the low plane must matter)
    Wn = (torch.randn(N, K, C, generator=g) * torch.logspace(-3, 1, C)).contiguous()
    lab = torch.randint(0, K, (N, way), generator=g)
    if way > 1:
        lab[0, 1] = lab[0, 0]
    logits, binl = ops.classify(Fq.to(dev), Wn.to(dev), lab.to(dev), way, base=base)
    ref = O.classify(Fq.double(), Wn.double())
    assert rel_err(logits.cpu(), ref) < 2e-6
    # element-wise: each logit within fp32 accumulation error of ITS OWN magnitude scale sum |w x|
    mag = O.classify(Fq.double().abs(), Wn.double().abs())
    assert float(((logits.cpu().double() - ref).abs() / mag).max()) < 1e-6
    lg = logits.cpu()
    for n in range(N):
        for jw in range(way):
            want = O.binary_logits(lg[n:n + 1], lab[n:n + 1, jw], base=base)
            assert torch.equal(binl.cpu()[n * way + jw], want[0]), (n, jw)
    # a label outside [0, K): ch1 = 0 and ch0 = the max over all rows (base) -- the stand-alone kernel's convention
    lab2 = lab.clone()
    lab2[0, 0] = K + 3
    _, b2 = ops.classify(Fq.to(dev), Wn.to(dev), lab2.to(dev), way, base=base)
    b_ref = ops.binary_logits(logits, lab2.to(dev), way, base=base)
    assert torch.equal(b2, b_ref)
    # the shared-weight entry point (mode='deploy', model/common.py:369-372) takes the same kernel
    sh = ops.classify_shared(Fq.to(dev), Wn[0].to(dev))
    assert rel_err(sh.cpu(), O.classify(Fq.double(), Wn[:1].double().expand(N, K, C))) < 2e-6


def test_episode_with_every_label_ignored(dev):
    """An episode whose whole query map is the ignore label: the reference's per-episode CE is 0 / 0 = NaN
    (trainer/utils.py:272-277) while its gradient is exactly zero -- ATen's nll_loss backward skips ignored targets,
    so nothing is multiplied by the infinite 1 / den.  The fused step must not turn that into NaN gradients for
    the OTHER episodes of the batch (parameter gradients sum over episodes)."""
    from denet_amd import ops
    from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
    from oracle import denet_oracle as O
    order = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
             "map.linear.weight", "map.linear.bias"]
    for K in (21, 81):                                   # both backward kernels (fp32 instruction / bf16 pipe)
        s = EpisodeShape("ign", N=3, way=1, shot=2, K=K, c=64, h=9, w=10, H=33, W=37, config_index=55, lam=1.0)
        ep = make_episodes(s, ignore_rows=2)
        ep["Yq_full"][1] = 255                           # episode 1: nothing to learn from
        state = make_estimator_state(s.K, s.c)
        P = [state[k].to(dev).requires_grad_(True) for k in order]
        Fq = ep["Fq"].to(dev).requires_grad_(True)
        Fs = ep["Fs"].to(dev).requires_grad_(True)
        lk, lp, logits, binl = ops.HeadStepFunction.apply(
            Fq, Fs, *P, ep["Ys"].to(dev), ep["label"].reshape(s.N, 1).to(dev), ep["Yq"].to(dev),
            ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
        lk.backward()
        ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
        assert torch.isnan(ref["loss_k"]) and torch.isnan(lk)                      # the loss is NaN on both sides
        assert not torch.isnan(ref["dFq"]).any()
        assert rel_err(Fq.grad.cpu(), ref["dFq"]) < TIGHT and rel_err(Fs.grad.cpu(), ref["dFs"]) < TIGHT
        assert float(Fq.grad[1].abs().max()) == 0.0                                # the ignored episode: exactly zero
        for k, t in zip(order, P):
            assert rel_err(t.grad.cpu(), ref["d:" + k]) < TIGHT, k
