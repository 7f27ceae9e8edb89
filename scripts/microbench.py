#!/usr/bin/env python3
"""Per-kernel timing through the C ABI (HIP events around `reps` back-to-back launches).
Tuning tool, not the headline benchmark."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from denet_amd import ops, _lib                         # noqa: E402
from denet_amd.synthetic import CONFIGS, EpisodeShape, make_episodes, make_estimator_state  # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--episodes", type=int, default=0)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    s = CONFIGS[args.config]
    if args.episodes:
        s = EpisodeShape(**{**s.asdict(), "N": args.episodes})
    dev = torch.device("cuda:0")
    ep = {k: v.to(dev).contiguous() for k, v in make_episodes(s).items()}
    st = make_estimator_state(s.K, s.c)
    order = ["gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias",
             "map.linear.weight", "map.linear.bias", "weight"]
    params = tuple(st[k].to(dev).contiguous() for k in order)
    N, way, shot, K, C, h, w, H, W = s.N, s.way, s.shot, s.K, s.c, s.h, s.w, s.H, s.W
    B, P = N * way * shot, h * w
    Fs = ep["Fs"].reshape(B, C, h, w)
    Ys = ep["Ys"].reshape(B, H, W)
    lab = ep["label"].reshape(N, way).contiguous()
    res = {}

    def want(name):
        return not args.only or name in args.only.split(",")

    mask, cnt = ops.mask_resize(Ys, h, w)
    S = ops.masked_sum(Fs, mask)
    Wn, saved = ops.proto_extend(S, cnt, params, lab, N, way, shot, K, P)
    logits, binl = ops.classify(ep["Fq"], Wn, lab, way)
    loss, per, scale, da, db = ops.upsample_wce(logits, ep["Yq_full"], ep["wgt"], grad=True)
    dFq, dWn_part, n_part = ops.classify_bwd(ep["Fq"], Wn, da, db, scale)
    g = ops.proto_extend_bwd(dWn_part, n_part, S, cnt, saved, params, lab, N, way, shot, K, P)

    if want("mask_resize"):
        res["mask_resize"] = timeit(lambda: ops.mask_resize(Ys, h, w))
    if want("masked_sum_fwd"):
        res["masked_sum_fwd"] = timeit(lambda: ops.masked_sum(Fs, mask))
    if want("proto_extend_fwd"):
        res["proto_extend_fwd"] = timeit(lambda: ops.proto_extend(S, cnt, params, lab, N, way, shot, K, P))
    if want("classify_fwd"):
        res["classify_fwd"] = timeit(lambda: ops.classify(ep["Fq"], Wn, lab, way))
    if want("wce_fwd"):
        res["wce_fwd_nograd"] = timeit(lambda: ops.upsample_wce(logits, ep["Yq_full"], ep["wgt"], grad=False))
        res["wce_fwd_grad"] = timeit(lambda: ops.upsample_wce(logits, ep["Yq_full"], ep["wgt"], grad=True))
    if want("focal_fwd"):
        res["focal_fwd"] = timeit(lambda: ops.upsample_focal(binl, ep["Yq"]))
    if want("classify_bwd"):
        res["classify_bwd"] = timeit(lambda: ops.classify_bwd(ep["Fq"], Wn, da, db, scale))
        res["classify_bwd_dFq_only"] = timeit(lambda: ops.classify_bwd(ep["Fq"], Wn, da, db, scale, need_dWn=False))
        res["classify_bwd_dWn_only"] = timeit(lambda: ops.classify_bwd(ep["Fq"], Wn, da, db, scale, need_dFq=False))
    if want("proto_extend_bwd"):
        res["proto_extend_bwd"] = timeit(lambda: ops.proto_extend_bwd(dWn_part, n_part, S, cnt, saved, params, lab,
                                                                      N, way, shot, K, P))
    if want("masked_sum_bwd"):
        res["masked_sum_bwd"] = timeit(lambda: ops.masked_sum_bwd(mask, g["coef"], (B, C, h, w), torch.float32))
    if want("metrics"):
        mat = torch.zeros(K, K, dtype=torch.int64, device=dev)
        up = ops.upsample_bilinear(logits, (H, W))
        res["upsample_bilinear"] = timeit(lambda: ops.upsample_bilinear(logits, (H, W)))
        res["confmat_api"] = timeit(lambda: ops.confmat_update(mat, ep["Yq_full"], up))
        res["confmat_fused"] = timeit(lambda: ops.confmat_update(mat, ep["Yq_full"], logits))
        res["biou_fused"] = timeit(lambda: ops.binary_counts(ep["Yq"].reshape(-1, H, W), binl))
    for k, v in res.items():
        print(f"{k:28s} {v:9.2f} us")


if __name__ == "__main__":
    main()
