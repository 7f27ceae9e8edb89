#!/usr/bin/env python3
"""C4 (BASELINE.json configs[3]): test-time mIoU evaluation, metrics only.  1000 synthetic episodes in
batches of 20 (dataset/voc_sbd.py:268, config/test_config_denet.template:25) through FullIOU / BinaryIOU,
(a) API level -- up-sampled [N,K,H,W] logits in, what the reference's trainer passes
(trainer/trainers.py:1026-1029) -- and (b) fused from the low-res logits; one int64 all-reduce of the
counts at the end when launched with more than one rank.  Prints one JSON line per K.

    python scripts/bench_eval.py                      # one GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_eval.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from denet_amd import ops                                  # noqa: E402
from denet_amd.metrics import BinaryIOU, FullIOU            # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)

E, NB, h, w, H, W = 1000, 20, 41, 41, 321, 321
nbatch = E // NB
mine = range(rank * nbatch // world, (rank + 1) * nbatch // world)     # contiguous blocks of the episode stream
for K in (21, 81):
    g = torch.Generator(device=dev).manual_seed(4000 + K)
    logits = 3.0 * torch.randn(NB, K, h, w, device=dev, generator=g)
    binl = 3.0 * torch.randn(NB, 2, h, w, device=dev, generator=g)
    yt = torch.randint(0, K, (NB, H, W), device=dev, generator=g)
    yq = torch.randint(0, 2, (NB, H, W), device=dev, generator=g)
    cls = torch.randint(1, K, (NB,), device=dev, generator=g)
    up, upb = ops.upsample_bilinear(logits, (H, W)), ops.upsample_bilinear(binl, (H, W))
    res = {}
    for name, (pl, pb) in (("api", (up, upb)), ("fused", (logits, binl))):
        f, b = FullIOU(K), BinaryIOU()
        for _ in range(3):
            f.update(yt, pl); b.update(yq, pb, cls)
        f, b = FullIOU(K), BinaryIOU()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in mine:
            f.update(yt, pl)
            b.update(yq, pb, cls, order_keys=range(i * NB, (i + 1) * NB))
        if world > 1:
            f.all_reduce(); b.all_reduce(K)
        miou, biou = f.mean_iou(), b.mean_iou()            # reads the counts: synchronises
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        per_ep = (K * H * W * 4 + H * W * 8 + 2 * H * W * 4 + H * W * 8) if name == "api" else \
                 (K * h * w * 4 + H * W * 8 + 2 * h * w * 4 + H * W * 8)
        res[name] = {"episodes_per_s": round(E / dt, 1), "ms_total": round(dt * 1e3, 2),
                     "bytes_per_episode": per_ep, "hbm_frac_of_8TBs": round(E / dt * per_ep / 8e12, 4),
                     "mIoU": miou, "binary_mIoU": biou}
    if rank == 0:
        print(json.dumps({"workload": f"C4: 1000 episodes, batch {NB}, K={K}, metrics only", "n_gpus": world, **res}))
if world > 1:
    sys.stdout.flush()
    time.sleep(0.5)
    os._exit(0)
