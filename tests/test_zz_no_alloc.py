"""The no-allocation entry point (include/denet_hip.h: denet_proto_extend_bwd_static) against the workspace
entry point it is a variant of, through the C ABI: same inputs -> bit-identical outputs, on two calls in a
row, the second with a smaller problem (the static scratch keeps the first call's contents)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(N, way, shot, K, C, n_part, seed, dev):
    g = torch.Generator().manual_seed(seed)
    B = N * way * shot
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    d = dict(dWn_part=r(N, n_part, K, C) * 0.1, S=r(B, C).abs(), cnt=(torch.rand(B, generator=g) * 50 + 1).to(dev),
             hidden=r(B, C).relu(), gate=torch.sigmoid(r(B, C)), pool=r(B, C), W1=r(C, C) * 0.05, W2=r(C, C) * 0.05,
             Wl=r(C, C) * 0.05, gate_mask=(torch.bernoulli(torch.full((B, C), 0.5), generator=g) * 2).to(dev),
             weight_mask=(torch.bernoulli(torch.full((N, K, C), 0.5), generator=g) * 2).to(dev))
    d["labels"] = torch.stack([torch.randperm(K - 1, generator=g)[:way] + 1 for _ in range(N)]).to(dev)
    return d


def _outputs(K, C, B, dev):
    z = lambda *s: torch.full(s, float("nan"), device=dev)
    return [z(C, C), z(C), z(C, C), z(C), z(C, C), z(C), z(K, C), z(B, C)]


def _run(name, d, N, way, shot, K, C, P, n_part, dev, with_ws):
    from denet_amd import _lib
    outs = _outputs(K, C, N * way * shot, dev)
    args = [d["dWn_part"], n_part, d["S"], d["cnt"], d["hidden"], d["gate"], d["pool"], d["W1"], d["W2"], d["Wl"],
            d["labels"], d["gate_mask"], d["weight_mask"], N, way, shot, K, C, P, *outs]
    if with_ws:
        nbytes = _lib.query("denet_proto_extend_bwd_workspace", N, way, shot, K, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        args += [ws, nbytes]
    _lib.call(name, *args)
    torch.cuda.synchronize()
    return outs


def test_static_scratch_entry_point_matches_workspace_entry_point():
    from denet_amd import _lib
    dev = torch.device("cuda:0")
    limit = _lib.query("denet_proto_extend_bwd_static_limit")
    assert limit == 1 << 20
    # first call: C2's shape with dWn given as 27 position-tile slabs; second call: a smaller problem
    for (N, way, shot, K, C, P, n_part, seed) in [(8, 1, 5, 21, 256, 1681, 27, 1), (2, 2, 3, 9, 64, 63, 1, 2)]:
        assert _lib.query("denet_proto_extend_bwd_workspace", N, way, shot, K, C) <= limit
        d = _problem(N, way, shot, K, C, n_part, seed, dev)
        ref = _run("denet_proto_extend_bwd", d, N, way, shot, K, C, P, n_part, dev, True)
        got = _run("denet_proto_extend_bwd_static", d, N, way, shot, K, C, P, n_part, dev, False)
        for a, b in zip(got, ref):
            assert torch.isfinite(b).all()
            assert torch.equal(a, b)          # same kernels, same order of operations: bit-identical
    # beyond the static limit: a clear error, nothing launched
    N, way, shot, K, C = 64, 2, 5, 81, 256
    assert _lib.query("denet_proto_extend_bwd_workspace", N, way, shot, K, C) > limit
    d = _problem(N, way, shot, K, C, 1, 3, dev)
    with pytest.raises(_lib.DenetHipError, match="static buffer"):
        _run("denet_proto_extend_bwd_static", d, N, way, shot, K, C, 100, 1, dev, False)
