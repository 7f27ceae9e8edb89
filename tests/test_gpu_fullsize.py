"""GPU: parity at BASELINE.json's full sizes.

  * C2 (1-way 5-shot, K=21, 256x41x41, 321x321): the fused head step against the checksums and
    sampled entries captured from the reference itself (tests/golden/c2_fullsize.npz).
  * C3 (2-way 5-shot, K=81, N=20) and C5 (2x spatial, bf16 features): against the oracle run here
    on the same seeded episodes (seconds on the host), plus size-independent properties:
    linearity of the classifier, conservation of pixel counts in the confusion matrix, zero
    gradient outside the support masks, idempotence of the metric accumulation.
  * edge cases the reference's semantics define: empty support mask, an episode whose labels are
    all ignored (loss = 0/0 = NaN there too), one-episode batches, non-square feature maps.
"""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, rel_err

pytestmark = pytest.mark.gpu
ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
         "map.linear.weight", "map.linear.bias"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _run_step(shape, dev, ep=None, backward=True):
    from denet_amd import ops
    from denet_amd.synthetic import make_episodes, make_estimator_state
    ep = ep or make_episodes(shape)
    state = make_estimator_state(shape.K, shape.c)
    P = [state[k].to(dev).requires_grad_(backward) for k in ORDER]
    Fq = ep["Fq"].to(dev).requires_grad_(backward)
    Fs = ep["Fs"].to(dev).requires_grad_(backward)
    lab = ep["label"].reshape(shape.N, shape.way).to(dev)
    lk, lp, logits, binl = ops.HeadStepFunction.apply(
        Fq, Fs, *P, ep["Ys"].to(dev), lab, ep["Yq"].to(dev), ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
    if backward:
        lk.backward()
    torch.cuda.synchronize()
    return ep, state, dict(loss_k=lk.detach(), loss_p=lp, logits_full=logits, logits_binary=binl, Fq=Fq, Fs=Fs, P=P)


def test_c2_fullsize_against_reference_checksums(dev):
    from denet_amd.synthetic import CONFIGS
    z = np.load(GOLDEN + "/c2_fullsize.npz")
    ep, state, out = _run_step(CONFIGS["C2"], dev)
    assert rel_err(out["loss_k"].cpu(), z["out:loss_k"]) < 2e-5
    assert rel_err(out["loss_p"].cpu(), z["out:loss_p"]) < 2e-5
    got = {"logits_full": out["logits_full"], "logits_binary": out["logits_binary"], "dFq": out["Fq"].grad,
           "dFs": out["Fs"].grad}
    got.update({"grad:" + k: t.grad for k, t in zip(ORDER, out["P"])})
    for name, t in got.items():
        flat = t.detach().reshape(-1).cpu()
        ref = torch.from_numpy(z["val:" + name])
        # sampled entries: max-normalised and element-wise (the floor is relative to the sample's largest entry)
        assert_close(flat[torch.from_numpy(z["idx:" + name])], ref, 1e-3, name)
        scale = float(z["abs:" + name])
        assert abs(float(flat.double().sum()) - float(z["sum:" + name])) < 1e-4 * scale + 1e-12, name
        assert abs(float(flat.double().abs().sum()) - scale) < 1e-4 * scale + 1e-12, name


def _confmat_check(logits, ep, K, dev, margin=1e-4):
    """Fused confusion matrix vs the oracle's up-sample + arg-max: pixels whose top-2 margin is
    above `margin` must be counted identically; the rest may flip and are reported."""
    from denet_amd import ops
    from oracle import denet_oracle as O
    H, W = ep["Yq_full"].shape[-2:]
    up = O.upsample(logits.detach().cpu().float(), (H, W))
    top = up.topk(2, dim=1)[0]
    safe = (top[:, 0] - top[:, 1]) > margin
    yt = ep["Yq_full"].clone()
    sub = int((~safe & (yt != 255)).sum())
    yt[~safe] = 255                                       # take the sub-margin pixels out of both sides
    mat = torch.zeros(K, K, dtype=torch.int64, device=dev)
    ops.confmat_update(mat, yt.to(dev), logits)
    ref = O.confusion_matrix(yt.numpy(), up.argmax(1).numpy(), K)
    assert np.array_equal(mat.cpu().numpy(), ref)
    assert int(mat.sum()) == int(((yt >= 0) & (yt < K)).sum())            # every valid pixel counted once
    ops.confmat_update(mat, yt.to(dev), logits)                            # accumulation: twice = 2x
    assert np.array_equal(mat.cpu().numpy(), 2 * ref)
    return sub


def test_c3_forward_and_metrics_against_oracle(dev):
    from denet_amd import ops
    from denet_amd.synthetic import CONFIGS
    from oracle import denet_oracle as O
    s = CONFIGS["C3"]
    ep, state, out = _run_step(s, dev, backward=False)
    with torch.no_grad():
        ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state,
                          backward=False)
    assert rel_err(out["logits_full"].cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(out["logits_binary"].cpu(), ref["logits_binary"]) < 2e-5
    assert rel_err(out["loss_k"].cpu(), ref["loss_k"]) < 1e-4
    assert rel_err(out["loss_p"].cpu(), ref["loss_p"]) < 1e-4
    sub = _confmat_check(out["logits_full"], ep, s.K, dev)
    print(f"C3: {sub} sub-margin pixels excluded of {ep['Yq_full'].numel()}")
    # BinaryIOU counts per way, skip rule included
    H, W = s.H, s.W
    bup = O.upsample(ref["logits_binary"], (H, W))
    safe = (bup[:, 1] - bup[:, 0]).abs() > 1e-4
    yq = ep["Yq"].reshape(-1, H, W)
    cnt = ops.binary_counts(yq.to(dev), out["logits_binary"]).cpu().numpy()
    pred = bup.max(1)[1].numpy()
    for i, (t, q, ok) in enumerate(zip(yq.numpy(), pred, safe.numpy())):
        assert cnt[i, 0] == t.sum()
        slack = int((~ok).sum())
        assert abs(int(cnt[i, 1]) - int(((t + q) == 2).sum())) <= slack
        assert abs(int(cnt[i, 2]) - int((((t + q) == 1) | ((t + q) == 2)).sum())) <= slack
    assert (cnt[:, 0] == 0).any()                         # the planted all-zero query mask is present


def test_c5_bf16_large_resolution_against_oracle(dev):
    from denet_amd.synthetic import CONFIGS, EpisodeShape
    from oracle import denet_oracle as O
    s = EpisodeShape(**{**CONFIGS["C5"].asdict(), "N": 2})           # full resolution, 2 episodes: host oracle in seconds
    ep, state, out = _run_step(s, dev)
    assert out["Fq"].dtype == torch.bfloat16 and out["Fs"].grad.dtype == torch.bfloat16
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    # bf16 inputs are exact in fp32; accumulation is fp32 on both sides
    assert rel_err(out["logits_full"].cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(out["loss_k"].cpu(), ref["loss_k"]) < 1e-4
    assert rel_err(out["loss_p"].cpu(), ref["loss_p"]) < 1e-4
    # feature gradients are rounded to bf16 on the way out (8 significant bits)
    assert rel_err(out["Fq"].grad.float().cpu(), ref["dFq"]) < 8e-3
    assert rel_err(out["Fs"].grad.float().cpu(), ref["dFs"]) < 8e-3
    for k, t in zip(ORDER, out["P"]):
        assert rel_err(t.grad.cpu(), ref["d:" + k]) < 1e-3, k
    _confmat_check(out["logits_full"], ep, s.K, dev)


def _plan(s):
    from denet_amd import _lib
    return _lib.head_step_plan(s.N, s.way, s.shot, s.K, s.c, s.h, s.w, s.H, s.W,
                               _lib.BF16 if s.feat_dtype == "bf16" else _lib.F32)


def _check_step_against_oracle(s, dev, gtol):
    from oracle import denet_oracle as O
    ep, state, out = _run_step(s, dev)
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    assert rel_err(out["logits_full"].cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(out["logits_binary"].cpu(), ref["logits_binary"]) < 2e-5
    assert rel_err(out["loss_k"].cpu(), ref["loss_k"]) < 1e-4
    assert rel_err(out["loss_p"].cpu(), ref["loss_p"]) < 1e-4
    assert rel_err(out["Fq"].grad.float().cpu(), ref["dFq"]) < gtol
    assert rel_err(out["Fs"].grad.float().cpu(), ref["dFs"]) < gtol
    for k, t in zip(ORDER, out["P"]):
        assert rel_err(t.grad.cpu(), ref["d:" + k]) < 1e-3, k
    return ep, out


def test_c3_fwd_bwd_at_the_benchmarked_batch(dev):
    """C3 forward AND backward at the 20 episodes profiles/*_bench_c3.json times, against the oracle (about a
    second on the host).  The launchers pick kernel variants from the sizes alone; the plan query reports what
    this shape runs, so the test fails if a threshold moves the benchmark onto a path it does not check:
    flat mask resize + non-temporal Fs loads (344 MB), classifier layout (a) with 12 blocks per episode,
    chain loss kernel on 8-column cells, persistent backward with 6 tile groups on the bf16 matrix pipe, non-temporal dFq stores,
    the flat non-temporal fp32 dFs writer."""
    from denet_amd import _lib
    from denet_amd.synthetic import CONFIGS
    s = CONFIGS["C3"]
    assert s.N == 20
    pl = _plan(s)
    assert pl["pool_launches"] == 2 and pl["pool_nt_loads"] == 1 and pl["pool_long_rows"] == 0
    assert pl["classify_layout"] == _lib.CLS_UNITS and pl["classify_groups"] == 12
    assert pl["loss_chain_cell"] == 8 and 1 < (s.K + pl["loss_chunk"] - 1) // pl["loss_chunk"] <= 5
    assert pl["bwd_persistent"] == 1 and pl["bwd_groups"] == 6 and pl["bwd_dfq_nt"] == 1 and pl["bwd_bf16_pipe"] == 1
    assert pl["bwd_xcd_pairs"] == 1
    assert pl["dfs_nt"] == 1 and pl["dfs_flat_writer"] == 1 and pl["dfs_chunked"] == 0
    _check_step_against_oracle(s, dev, 2e-5)


def test_c5_at_the_benchmarked_batch(dev):
    """C5 at the 8 episodes profiles/*_bench_c5.json times (the N = 2 case above stays below every size
    threshold): long-row bf16 pooling with non-temporal loads (269 MB), flat mask resize, the bf16-pipe classifier,
    chain loss kernel on 9-column cells, persistent backward with 15 tile groups on the bf16 matrix pipe, the chunked bf16 dFs writer
    (coefficient blocks + one block per contiguous chunk, non-temporal stores)."""
    from denet_amd import _lib
    from denet_amd.synthetic import CONFIGS
    s = CONFIGS["C5"]
    assert s.N == 8 and s.feat_dtype == "bf16"
    pl = _plan(s)
    assert pl["pool_launches"] == 2 and pl["pool_nt_loads"] == 1 and pl["pool_long_rows"] == 1
    assert pl["classify_layout"] == _lib.CLS_WAVE4_BF16
    assert pl["loss_chain_cell"] == 9
    assert pl["bwd_persistent"] == 1 and pl["bwd_groups"] == 15 and pl["bwd_dfq_nt"] == 1 and pl["bwd_bf16_pipe"] == 1
    assert pl["bwd_xcd_pairs"] == 1
    assert pl["dfs_nt"] == 1 and pl["dfs_flat_writer"] == 0 and pl["dfs_chunked"] == 1
    ep, out = _check_step_against_oracle(s, dev, 8e-3)
    assert out["Fs"].grad.dtype == torch.bfloat16


def test_c2_plan_at_the_headline_batch():
    """The headline configuration's variants (everything cached, in-kernel mask gather, eight-wave classifier),
    which test_c2_fullsize_against_reference_checksums checks against the reference's own run."""
    from denet_amd import _lib
    from denet_amd.synthetic import CONFIGS
    pl = _plan(CONFIGS["C2"])
    assert pl["pool_launches"] == 1 and pl["pool_nt_loads"] == 0 and pl["classify_layout"] == _lib.CLS_TILE8
    assert pl["loss_chain_cell"] == 8 and pl["loss_chunk"] == 21
    assert pl["bwd_groups"] == 14 and pl["dfs_nt"] == 0 and pl["bwd_dfq_nt"] == 1 and pl["bwd_bf16_pipe"] == 0
    assert pl["bwd_xcd_pairs"] == 1


@pytest.mark.parametrize("feat_dtype", ["f32", "bf16"])
def test_c2_sixteen_episodes_batch_paths_against_oracle(dev, feat_dtype):
    """C2's shape at 16 episodes takes the batch paths of the K <= 32 kernels: 432 position tiles -> the
    wave-per-tile forward classifier (from 384 tiles on), 80 support images -> the flat mask resize with the count
    written by the pooling kernel (from 64 images on), 7 tile groups per episode in the persistent backward; bf16
    also runs the planar mask image of both streaming kernels on short rows."""
    from denet_amd.synthetic import CONFIGS, EpisodeShape
    from oracle import denet_oracle as O
    s = EpisodeShape(**{**CONFIGS["C2"].asdict(), "N": 16, "feat_dtype": feat_dtype})
    ep, state, out = _run_step(s, dev)
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    gtol = 8e-3 if feat_dtype == "bf16" else 2e-5           # feature gradients are rounded to bf16 on the way out
    assert rel_err(out["logits_full"].cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(out["logits_binary"].cpu(), ref["logits_binary"]) < 2e-5
    assert rel_err(out["loss_k"].cpu(), ref["loss_k"]) < 1e-4
    assert rel_err(out["loss_p"].cpu(), ref["loss_p"]) < 1e-4
    assert rel_err(out["Fq"].grad.float().cpu(), ref["dFq"]) < gtol
    assert rel_err(out["Fs"].grad.float().cpu(), ref["dFs"]) < gtol
    for k, t in zip(ORDER, out["P"]):
        assert rel_err(t.grad.cpu(), ref["d:" + k]) < 1e-3, k


def test_properties_at_full_size(dev):
    """Size-independent properties on C2: classifier linearity, masked-gradient support."""
    from denet_amd import ops
    from denet_amd.synthetic import CONFIGS, make_episodes
    s = CONFIGS["C2"]
    ep = make_episodes(s)
    Fq = ep["Fq"].to(dev)
    Wn = torch.randn(s.N, s.K, s.c, device=dev)
    a, _ = ops.classify(Fq, Wn)
    b, _ = ops.classify(2.0 * Fq, Wn)
    assert torch.equal(b, 2.0 * a)                        # scaling by a power of two is exact
    c, _ = ops.classify(Fq, Wn.flip(1).contiguous())
    assert torch.equal(c, a.flip(1))                      # class rows are independent
    _, _, out = _run_step(s, dev, ep=ep)
    B, P = s.N * s.way * s.shot, s.h * s.w
    mask, cnt = ops.mask_resize(ep["Ys"].reshape(B, s.H, s.W).to(dev), s.h, s.w)
    dFs = out["Fs"].grad.reshape(B, s.c, P)
    assert float((dFs * (1 - mask).unsqueeze(1)).abs().max()) == 0.0      # no gradient outside the mask
    assert float(cnt[0]) == 0.0 and torch.isfinite(out["logits_full"]).all()   # empty mask: den + 1e-3


def test_edge_cases(dev):
    from denet_amd import ops
    from denet_amd.synthetic import EpisodeShape, make_episodes
    from oracle import denet_oracle as O
    # one episode, non-square maps, position count not a multiple of the 64-wide tile
    s = EpisodeShape("edge", N=1, way=1, shot=1, K=3, c=64, h=5, w=13, H=33, W=97, config_index=9, lam=0.5)
    ep = make_episodes(s, ignore_rows=2)
    ep["Ys"][0, 0, 0, ::2] = 1                            # make_episodes empties the first support mask
    _, state, out = _run_step(s, dev, ep=ep)
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    assert rel_err(out["logits_full"].cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(out["loss_k"].cpu(), ref["loss_k"]) < 1e-4
    assert float(ref["dFs"].abs().max()) > 0 and rel_err(out["Fs"].grad.cpu(), ref["dFs"]) < 1e-3
    assert rel_err(out["Fq"].grad.cpu(), ref["dFq"]) < 1e-3
    # an episode whose labels are all ignored: weighted mean over nothing = NaN, as in the reference
    s2 = EpisodeShape("edge2", N=2, way=1, shot=2, K=5, c=64, h=9, w=9, H=65, W=65, config_index=10)
    ep2 = make_episodes(s2, ignore_rows=2)
    ep2["Yq_full"][1] = 255
    _, state2, out2 = _run_step(s2, dev, ep=ep2, backward=False)
    ref2 = O.head_step(ep2["Fq"], ep2["Fs"], ep2["Ys"], ep2["label"], ep2["Yq"], ep2["Yq_full"], ep2["wgt"], state2,
                       backward=False)
    assert torch.isnan(ref2["loss_k"]) and torch.isnan(out2["loss_k"].cpu())
    assert rel_err(out2["loss_p"].cpu(), ref2["loss_p"]) < 1e-4
    # confusion matrix ignores every out-of-range label, not only 255
    yt = torch.tensor([[[0, 1, 2, 7, -1, 255, 254, 3]]])
    pr = torch.zeros(1, 3, 1, 8)
    pr[0, 2] = 1.0
    mat = torch.zeros(3, 3, dtype=torch.int64, device=dev)
    ops.confmat_update(mat, yt.to(dev), pr.to(dev))
    assert mat.cpu().tolist() == [[0, 0, 1], [0, 0, 1], [0, 0, 1]]


def test_limits_and_duplicate_labels(dev):
    """The documented limits (K = 128, shot = 8) and the reference's overwrite order when two ways of an
    episode carry the same label (model/common.py:327-334: the later way wins), forward and backward."""
    from denet_amd.synthetic import EpisodeShape, make_episodes
    from oracle import denet_oracle as O
    s = EpisodeShape("limits", N=2, way=3, shot=8, K=128, c=64, h=6, w=7, H=41, W=50, config_index=11)
    ep = make_episodes(s, ignore_rows=1)
    ep["label"][0] = torch.tensor([5, 9, 5])              # ways 0 and 2 collide: way 2's prototype is kept
    ep["label"][1] = torch.tensor([127, 1, 64])
    _, state, out = _run_step(s, dev, ep=ep)
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    assert rel_err(out["logits_full"].cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(out["logits_binary"].cpu(), ref["logits_binary"]) < 2e-5
    assert rel_err(out["loss_k"].cpu(), ref["loss_k"]) < 1e-4
    assert rel_err(out["loss_p"].cpu(), ref["loss_p"]) < 1e-4
    assert rel_err(out["Fq"].grad.cpu(), ref["dFq"]) < 1e-3
    assert rel_err(out["Fs"].grad.cpu(), ref["dFs"]) < 1e-3
    # the losing way (episode 0, way 0) gets no gradient at all, like the overwritten row in the reference
    assert float(out["Fs"].grad[0, 0].abs().max()) == 0.0 and float(ref["dFs"][0, 0].abs().max()) == 0.0
    for k, t in zip(ORDER, out["P"]):
        assert rel_err(t.grad.cpu(), ref["d:" + k]) < 1e-3, k


def test_many_ways_and_wide_channels(dev):
    """Sliced-chain corners: more ways than the classifier block has thread groups (the novel rows are
    then assembled in a second pass, K <= 32 runs eight waves per block), duplicate and out-of-range
    labels among them, and shot = 8 at C = 256 (the widest gate/projection instantiation)."""
    from denet_amd.synthetic import EpisodeShape, make_episodes
    from oracle import denet_oracle as O
    s = EpisodeShape("ways", N=2, way=10, shot=1, K=21, c=64, h=5, w=6, H=33, W=41, config_index=13)
    ep = make_episodes(s, ignore_rows=1)
    ep["label"][0] = torch.tensor([1, 2, 3, 4, 5, 6, 7, 8, 9, 3])        # way 9 overrides way 2
    ep["label"][1] = torch.tensor([20, 19, 18, 17, 16, 15, 14, 13, 12, 11])
    _, state, out = _run_step(s, dev, ep=ep)
    ref = O.head_step(ep["Fq"], ep["Fs"], ep["Ys"], ep["label"], ep["Yq"], ep["Yq_full"], ep["wgt"], state)
    assert rel_err(out["logits_full"].cpu(), ref["logits_full"]) < 2e-5
    assert rel_err(out["logits_binary"].cpu(), ref["logits_binary"]) < 2e-5
    assert rel_err(out["Fq"].grad.cpu(), ref["dFq"]) < 1e-3
    assert rel_err(out["Fs"].grad.cpu(), ref["dFs"]) < 1e-3
    assert float(out["Fs"].grad[0, 2].abs().max()) == 0.0                 # the overridden way
    for k, t in zip(ORDER, out["P"]):
        assert rel_err(t.grad.cpu(), ref["d:" + k]) < 1e-3, k

    s2 = EpisodeShape("wide", N=2, way=2, shot=8, K=40, c=256, h=7, w=9, H=50, W=70, config_index=14, lam=1.5)
    ep2 = make_episodes(s2, ignore_rows=1)
    _, state2, out2 = _run_step(s2, dev, ep=ep2)
    ref2 = O.head_step(ep2["Fq"], ep2["Fs"], ep2["Ys"], ep2["label"], ep2["Yq"], ep2["Yq_full"], ep2["wgt"], state2)
    assert rel_err(out2["logits_full"].cpu(), ref2["logits_full"]) < 2e-5
    assert rel_err(out2["loss_k"].cpu(), ref2["loss_k"]) < 1e-4
    assert rel_err(out2["loss_p"].cpu(), ref2["loss_p"]) < 1e-4
    assert rel_err(out2["Fq"].grad.cpu(), ref2["dFq"]) < 1e-3
    assert rel_err(out2["Fs"].grad.cpu(), ref2["dFs"]) < 1e-3
    for k, t in zip(ORDER, out2["P"]):
        assert rel_err(t.grad.cpu(), ref2["d:" + k]) < 1e-3, k


def test_unsupported_shapes_fail_loudly(dev):
    """No fallback path: shapes outside the kernels' limits raise instead of computing something else."""
    from denet_amd import _lib, ops
    from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state
    for kw in (dict(c=128), dict(shot=9), dict(K=129)):
        s = EpisodeShape(**{**dict(name="bad", N=1, way=1, shot=1, K=5, c=64, h=5, w=5, H=17, W=17, config_index=12),
                            **kw})
        ep = make_episodes(s, ignore_rows=1)
        state = make_estimator_state(s.K, s.c)
        P = [state[k].to(dev) for k in ORDER]
        with pytest.raises(_lib.DenetHipError):
            ops.HeadStepFunction.apply(ep["Fq"].to(dev), ep["Fs"].to(dev), *P, ep["Ys"].to(dev),
                                       ep["label"].reshape(s.N, s.way).to(dev), ep["Yq"].to(dev),
                                       ep["Yq_full"].to(dev), ep["wgt"].to(dev), None, None)
    with pytest.raises(_lib.DenetHipError):
        ops.masked_sum(torch.zeros(1, 64, 4, 4, dtype=torch.float16, device=dev), torch.zeros(1, 16, device=dev))


@pytest.mark.parametrize("K", [21, 81])
def test_c4_eval_1000_episodes_sharded_counts(dev, K):
    """C4 (BASELINE.json configs[3]): test-time mIoU over 1000 synthetic episodes in batches of 20
    (dataset/voc_sbd.py:268, config/test_config_denet.template:25), through FullIOU / BinaryIOU.
    Size-independent properties at the full size: the counts of G = 2, 4, 8 contiguous episode shards
    add up to the single-stream matrix bit for bit (what the int64 all-reduce of SURVEY.md §8e does),
    every valid pixel is counted exactly once, mean_iou is the same float64 for every G, and
    BinaryIOU's first-seen class order is restored from the shards' order keys.  The first two batches
    are also checked against the oracle (up-sample + arg-max + bincount on the host)."""
    from denet_amd.metrics import BinaryIOU, FullIOU
    from oracle import denet_oracle as O
    E, NB, h, w, H, W = 1000, 20, 41, 41, 321, 321

    def batch(b):
        g = torch.Generator(device=dev).manual_seed(4000 + 97 * K + b)
        logits = 3.0 * torch.randn(NB, K, h, w, device=dev, generator=g)
        yt = torch.randint(0, K, (NB, H, W), device=dev, generator=g)
        yt[:, :8] = 255
        binl = 3.0 * torch.randn(NB, 2, h, w, device=dev, generator=g)
        yq = torch.randint(0, 2, (NB, H, W), device=dev, generator=g)
        if b % 7 == 0:
            yq[3] = 0                                     # no foreground: BinaryIOU skips the sample
        cls = torch.randint(1, K, (NB,), device=dev, generator=g)
        return logits, yt, binl, yq, cls

    nbatch = E // NB
    per_batch, valid = [], 0
    single_f, single_b = FullIOU(K), BinaryIOU()
    for b in range(nbatch):
        logits, yt, binl, yq, cls = batch(b)
        single_f.update(yt, logits)
        single_b.update(yq, binl, cls)
        one = FullIOU(K)
        one.update(yt, logits)
        per_batch.append(one.counts())
        valid += int((yt != 255).sum())
        if b < 2 and K == 21:
            up = O.upsample(logits.cpu(), (H, W))
            top = up.topk(2, dim=1)[0]
            assert float((top[:, 0] - top[:, 1]).min()) >= 0.0
            ytc = yt.cpu().clone()
            ytc[(top[:, 0] - top[:, 1]) <= 1e-4] = 255    # sub-margin pixels out of both sides
            chk = FullIOU(K)
            chk.update(ytc.to(dev), logits)
            assert np.array_equal(chk.counts(), O.confusion_matrix(ytc.numpy(), up.argmax(1).numpy(), K))
    ref = single_f.counts()
    assert int(ref.sum()) == valid
    assert np.array_equal(sum(per_batch), ref)
    miou, biou = single_f.mean_iou(), single_b.mean_iou()
    order = list(single_b.inter.keys())
    for G in (2, 4, 8):
        shard = nbatch // G                               # contiguous blocks of the episode stream per rank
        mats, bs = [], []
        for r in range(G):
            f, bi = FullIOU(K), BinaryIOU()
            lo = r * shard
            hi = nbatch if r == G - 1 else lo + shard
            for b in range(lo, hi):
                logits, yt, binl, yq, cls = batch(b)
                f.update(yt, logits)
                bi.update(yq, binl, cls, order_keys=np.arange(b * NB, (b + 1) * NB))
            mats.append(f.counts())
            bs.append(bi)
        merged = FullIOU(K)
        merged.update_counts(sum(mats))
        assert np.array_equal(merged.counts(), ref)
        assert merged.mean_iou() == miou                  # same integers -> same float64
        # what BinaryIOU.all_reduce does with SUM / MIN collectives, on the host
        inter, union, first = {}, {}, {}
        for bi in bs:
            for c in bi.inter:
                inter[c] = inter.get(c, 0.0) + bi.inter[c]
                union[c] = union.get(c, 0.0) + bi.union[c]
                first[c] = min(first.get(c, 1 << 62), bi._first[c])
        seq = sorted(first, key=lambda c: first[c])
        assert seq == order
        assert float(np.mean([inter[c] / union[c] for c in seq])) == biou
        assert all(inter[c] == single_b.inter[c] and union[c] == single_b.union[c] for c in seq)
