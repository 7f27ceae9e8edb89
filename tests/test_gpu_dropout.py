"""GPU: train-mode dropout generated inside the kernels (include/denet_hip.h "T1 in train mode").

The reference's dropout (nn.Dropout at model/common.py:231 and :326) draws from torch's generator, so its
stream cannot be reproduced; what is checked here:
  * the library's multipliers are bit-identical to the oracle's Philox restatement (itself pinned by the
    Random123 known-answer vectors on the CPU side), take the values {0, 1/(1-p)} and keep 1-p of the elements;
  * the fused train-mode step equals, bit for bit, the same step fed those multipliers through the
    injected-mask pointers -- the path the reference fixture `train_1way_5shot` pins;
  * the module path (WeightEstimator.train()) and the fused step agree for one seed.
"""
import numpy as np
import pytest
import torch

from denet_amd import _lib, ops
from denet_amd.engine import HeadStepEngine, PARAM_ORDER
from denet_amd.synthetic import CONFIGS, EpisodeShape, make_episodes, make_estimator_state
from oracle import denet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("p,stream,n", [(0.5, ops.DROPOUT_GATE, 1 << 20), (0.5, ops.DROPOUT_WEIGHT, 1 << 20),
                                        (0.25, ops.DROPOUT_WEIGHT, 1001), (0.9, ops.DROPOUT_GATE, 40 * 256)])
def test_dropout_mask_matches_the_oracle_stream(p, stream, n, dev):
    seed = 0x1234_5678_9ABC_DEF0
    m = ops.dropout_mask(seed, stream, p, (n,), dev).cpu().numpy()
    ref = O.dropout_mask(seed, stream, p, n)
    assert np.array_equal(m, ref)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    assert set(np.unique(m).tolist()) <= {0.0, float(scale)}
    if n >= 1 << 20:
        keep = float((m != 0).mean())
        assert abs(keep - (1.0 - p)) < 0.01, keep            # keep-rate 0.5 +- 0.01
        assert abs(float(m.mean()) - 1.0) < 0.02             # E[multiplier] = 1
    # another seed / the other stream give another mask; p = 0 gives ones
    other = ops.dropout_mask(seed + 1, stream, p, (n,), dev).cpu().numpy()
    assert not np.array_equal(other, m)
    assert np.array_equal(ops.dropout_mask(seed, stream, 0.0, (n,), dev).cpu().numpy(), np.ones(n, np.float32))


def test_dropout_rejects_bad_p(dev):
    out = torch.empty(16, device=dev)
    for p in (1.0, -0.1, float("nan")):
        with pytest.raises(_lib.DenetHipError):
            _lib.call("denet_dropout_mask", 1, ops.DROPOUT_GATE, p, 16, out)
    with pytest.raises(_lib.DenetHipError):
        _lib.call("denet_dropout_mask", 1, 7, 0.5, 16, out)


SMALL = EpisodeShape("S", N=3, way=2, shot=2, K=9, c=64, h=13, w=11, H=50, W=45, config_index=11)
WIDE = EpisodeShape("Wd", N=3, way=2, shot=3, K=81, c=256, h=21, w=21, H=161, W=161, config_index=12)


def _run(shape, dev, *, drop=None, masks=(None, None), fdt=torch.float32):
    ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
    params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(shape.K, shape.c).items()}
    N, way = shape.N, shape.way
    labels = ep["label"].reshape(N, way).contiguous()
    Fq, Fs = ep["Fq"].to(fdt), ep["Fs"].to(fdt)
    eng = HeadStepEngine(N, way, shape.shot, shape.K, shape.c, shape.h, shape.w, shape.H, shape.W, dev, fdt)
    eng.forward(Fq, Fs, ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params, gate_mask=masks[0],
                weight_mask=masks[1], drop=drop)
    eng.backward(Fq, labels, params, gate_mask=masks[0], weight_mask=masks[1], drop=drop)
    torch.cuda.synchronize()
    out = {"logits": eng.logits_full, "binl": eng.logits_binary, "losses": eng.losses, "dFq": eng.dFq, "dFs": eng.dFs}
    out.update({"g:" + k: eng.grads[k] for k in PARAM_ORDER})
    return {k: v.detach().clone() for k, v in out.items()}


@pytest.mark.parametrize("shape,fdt", [(SMALL, torch.float32), (CONFIGS["C2"], torch.float32),
                                       (WIDE, torch.float32), (WIDE, torch.bfloat16)],
                         ids=["small-c64", "C2", "K81", "K81-bf16"])
def test_train_step_equals_injected_masks_of_the_same_seed(shape, fdt, dev):
    seed, pg, pw = 0xABCDEF0123, 0.5, 0.5
    B = shape.N * shape.way * shape.shot
    gm = ops.dropout_mask(seed, ops.DROPOUT_GATE, pg, (B, shape.c), dev)
    wm = ops.dropout_mask(seed, ops.DROPOUT_WEIGHT, pw, (shape.N, shape.K, shape.c), dev)
    a = _run(shape, dev, drop=_lib.Dropout(seed, pg, pw), fdt=fdt)
    b = _run(shape, dev, masks=(gm, wm), fdt=fdt)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # dropout is really on (differs from eval mode), and a mask pointer wins over the generator
    e = _run(shape, dev, fdt=fdt)
    assert not torch.equal(a["logits"], e["logits"])
    assert not torch.equal(a["g:weight"], e["g:weight"])
    ones_g, ones_w = torch.ones_like(gm), torch.ones_like(wm)
    c = _run(shape, dev, drop=_lib.Dropout(seed, pg, pw), masks=(ones_g, ones_w), fdt=fdt)
    for k in e:
        assert torch.equal(c[k], e[k]), k
    # another seed: another step
    d = _run(shape, dev, drop=_lib.Dropout(seed + 1, pg, pw), fdt=fdt)
    assert not torch.equal(a["logits"], d["logits"])
    # only one site on: p = 0 at the other leaves it without dropout
    g_only = _run(shape, dev, drop=_lib.Dropout(seed, pg, 0.0), fdt=fdt)
    g_ref = _run(shape, dev, masks=(gm, None), fdt=fdt)
    for k in g_only:
        assert torch.equal(g_only[k], g_ref[k]), k


def test_train_step_bad_p_is_rejected(dev):
    with pytest.raises(_lib.DenetHipError):
        _run(SMALL, dev, drop=_lib.Dropout(1, 1.0, 0.5))


def test_module_train_mode_uses_the_library_stream(dev):
    """WeightEstimator.train(): its masks are the library's for the seed it drew (torch.manual_seed fixes it),
    so the module path reproduces the fused train-mode step of that seed."""
    from denet_amd.head import WeightEstimator
    shape = SMALL
    ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
    state = make_estimator_state(shape.K, shape.c)
    we = WeightEstimator(shape.K, shape.c)
    we.load_state_dict(state)
    we = we.to(dev).train()
    labels = ep["label"].reshape(shape.N, shape.way)
    torch.manual_seed(5)
    full1, _ = we.forward_with_binary(ep["Fq"], ep["Fs"], ep["Ys"], labels)
    seed = we.last_dropout_seed
    torch.manual_seed(5)
    full2, _ = we.forward_with_binary(ep["Fq"], ep["Fs"], ep["Ys"], labels)
    assert torch.equal(full1, full2) and we.last_dropout_seed == seed
    full3, _ = we.forward_with_binary(ep["Fq"], ep["Fs"], ep["Ys"], labels)
    assert not torch.equal(full1, full3)                       # the next draw
    fused = _run(shape, dev, drop=_lib.Dropout(seed, 0.5, 0.5))
    err = (fused["logits"] - full1.detach()).abs().max().item()
    assert err <= 1e-4 * full1.abs().max().item(), err        # same masks; fused vs stand-alone kernels
    we.eval()
    assert we.dropout_config() is None
