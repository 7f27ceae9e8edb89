#!/usr/bin/env python3
"""Headline benchmark: DENet head episodes/s, fwd+bwd, 1-way 5-shot (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the head over one batch of synthetic episodes already
resident in HBM: K0..K5 forward, both losses, and the backward of loss_k down
to dFq, dFs and the seven estimator parameter gradients (SURVEY.md §8d
"boundary being timed").  Feature producer and optimizer are excluded, as in
the metric's definition.  With N > 1 each rank owns its own batch (episodes are
independent: weak scaling) and the step contains ONE RCCL all-reduce (AVG) of
the flat parameter-gradient bucket, enqueued behind the backward.

Rank 0 prints one JSON line.  `roofline` is for the dominant kernel, timed with
HIP events on the launch stream inside the timed region; `cpu_baseline` is the
oracle (CPU restatement of the reference head, stock PyTorch CPU ops) on the
same batch, on this host's cores.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from denet_amd import _lib                                    # noqa: E402
from denet_amd.engine import HeadStepEngine, PARAM_ORDER      # noqa: E402
from denet_amd.hiprt import Event                             # noqa: E402
from denet_amd.synthetic import CONFIGS, EpisodeShape, make_episodes, make_estimator_state  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def kernel_bytes(s: EpisodeShape, sF: int):
    """Algorithmic bytes per launch of each kernel of the step."""
    N, way, shot, K, C = s.N, s.way, s.shot, s.K, s.c
    B, P, Q, NB = N * way * shot, s.h * s.w, s.H * s.W, N * way
    n_part = (P + 63) // 64
    if P >= 4:
        # persistent blocks write one dWn slab per GROUP of position tiles (dev_classify.hpp,
        # cls_bwd_wide_groups): 14 per episode at C2, 6 at C3, 15 at C5, instead of 27 / 27 / 103
        tgt = 512 if (K <= 32 and 2 * N * n_part >= 1536) else 256
        tpb = (2 * N * n_part + tgt - 1) // tgt
        n_part = (n_part + tpb - 1) // tpb
    nsl, G = C // 16, N * way
    return {
        # Fs stream + mask gather + S, W1 slices (each element of W1 once per image) + layer-1 partials
        "masked_pool_fwd": B * C * P * sF + B * P * 8 + B * P * 4 + B * C * 4 + B * C * C * 4 + B * nsl * C * 4,
        # per (group, slice) block: the group's layer-1 partials + 16 rows of W2 + 16 columns of Wl; outputs
        "proto_extend_fwd": G * nsl * (shot * nsl * C + 32 * C) * 4 + 3 * B * C * 4 + G * nsl * C * 4,
        "classify_fwd": N * C * P * sF + K * C * 4 + G * nsl * C * 4 + N * K * C * 4 + N * K * P * 4 + NB * 2 * P * 4,
        "loss_pair_fwd": N * Q * 8 + N * K * P * 4 + 2 * N * K * P * 4 + NB * Q * 8 + NB * 2 * P * 4,
        "loss_finalize": (N + NB) * s.h * 8,
        # dWn slabs || dFq: Fq twice (once per role), the two gradient slabs twice, dFq out, slabs out
        "bwd_dwn": 2 * N * C * P * sF + 4 * N * K * P * 4 + N * C * P * sF + N * n_part * K * C * 4 + N * K * C * 4,
        # per (group, slice) block: the label row of every slab + 16 columns of Wl + 16 rows of W2; outputs
        # ... and the dweight role: every slab once
        "bwd_chain_dfq": G * nsl * (n_part * C + 32 * C) * 4 + 2 * B * C * 4 + B * nsl * C * 4
                         + N * n_part * K * C * 4 + K * C * 4,
        "bwd_params_dfs": B * C * P * sF + B * P * 4 + B * C * C * 4 + B * nsl * C * 4 + 6 * B * C * 4
                          + 3 * C * C * 4,
    }


def step_bytes_per_episode(s: EpisodeShape, sF: int) -> float:
    """SURVEY.md §8d formula (compulsory traffic per episode, fwd + bwd, with loss_p)."""
    w_, sh, K, C = s.way, s.shot, s.K, s.c
    P, Q = s.h * s.w, s.H * s.W
    fwd = w_ * sh * C * P * sF + w_ * sh * P * 8 + C * P * sF + K * P * 4 + 2 * w_ * P * 4 + Q * 8 + w_ * Q * 8
    bwd = Q * 8 + K * P * 4 + 2 * K * P * 4 + 2 * C * P * sF + w_ * sh * C * P * sF + w_ * sh * P * 8
    return float(fwd + bwd)


def pmc_traffic(kernel: str, config: str, episodes: int):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json,
    written by scripts/pmc_summary.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    runs of this same command; reads doubled per MI355X_MICROARCH.md §HBM).  None if the passes were
    taken at another batch size or are absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    run = d.get("runs", {}).get(f"{config}/{episodes}")
    if run is None:
        return None
    return run.get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--episodes", type=int, default=0, help="episodes per GPU (default: the config's batch)")
    ap.add_argument("--graph", type=int, default=0, help="replay the step from a captured HIP graph")
    ap.add_argument("--probe", default="auto", help="kernel to time with HIP events (name | auto | none)")
    ap.add_argument("--probe-every", type=int, default=8,
                    help="bracket the probed kernel with HIP events on every n-th timed step (an event pair "
                         "costs a few microseconds of stream time, so not on every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU-baseline sample (seconds)")
    ap.add_argument("--breakdown", type=int, default=1, help="also report every kernel's duration (untimed pass)")
    ap.add_argument("--collective", action="store_true",
                    help="with one rank: still create the RCCL group and run the step's gradient all-reduce "
                         "(exercises the N > 1 code path on a one-GPU box)")
    ap.add_argument("--settle", type=int, default=2000,
                    help="untimed steps run back to back before the W warm-up steps (set-up: sustained clocks)")
    ap.add_argument("--train-mode", action="store_true",
                    help="time the step with train-mode dropout generated inside the kernels (p = 0.5 at both "
                         "sites, a new seed every step) instead of eval mode; the default run reports it beside "
                         "the eval-mode value as `train_mode`")
    args = ap.parse_args()
    if args.train_mode and args.graph:
        raise SystemExit("--train-mode with --graph would replay one captured seed; run them separately")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; the modulo only matters for a dry run of the N > 1 path on a one-GPU box
    # (DENET_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device)
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("DENET_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    # gloo (dry runs only) has no AVG: it sums, which is the same exchange
    reduce_op = None if dist is None else (dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM)

    # Dry runs over gloo (two ranks sharing one GPU) reduce through host memory: gloo's DEVICE-tensor
    # all-reduce hangs intermittently in that configuration with stock torch kernels as the only producers
    # (scripts/diag_gloo_shared_gpu.py, mode `torch`: both ranks parked in all_reduce after 20-40 calls), so the
    # round-1 hang was not the step's stream ordering.  RCCL (one rank per GPU) reduces the device bucket in place.
    host_reduce = dist is not None and dist.get_backend() != "nccl"

    shape = CONFIGS[args.config]
    if args.episodes > 0:
        shape = EpisodeShape(**{**shape.asdict(), "N": args.episodes})
    ep_cpu = make_episodes(shape, seed_offset=1000 * rank)
    state_cpu = make_estimator_state(shape.K, shape.c)
    fdt = torch.bfloat16 if shape.feat_dtype == "bf16" else torch.float32
    sF = 2 if fdt == torch.bfloat16 else 4
    ep = {k: v.to(dev).contiguous() for k, v in ep_cpu.items()}
    params = {k: v.to(dev).contiguous() for k, v in state_cpu.items()}
    N, way = shape.N, shape.way
    labels = ep["label"].reshape(N, way).contiguous()
    eng = HeadStepEngine(N, way, shape.shot, shape.K, shape.c, shape.h, shape.w, shape.H, shape.W, dev, fdt)
    stream = torch.cuda.current_stream().cuda_stream

    # the step's buffers never change: bind the arguments once, so a step costs ~20 us of host time
    # (two library calls) instead of ~40 us of ctypes argument conversion
    # train mode: WeightEstimator's two dropouts (model/common.py:231,326), generated in the kernels from a seed
    # the host advances every step (the library reads the struct at each call)
    drop = _lib.Dropout(0x5EED0000 + rank, 0.5, 0.5)
    bind_args = (ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params)
    eng.bind_step(*bind_args, drop=drop if args.train_mode else None)
    train_now = [args.train_mode]

    def step():
        st = torch.cuda.current_stream().cuda_stream if args.graph else stream     # capture runs on a side stream
        if train_now[0]:
            drop.seed += world
        eng.run_forward(st)
        eng.run_backward(0, st)
        if dist is not None:
            # ONE all-reduce of the flat parameter-gradient bucket, enqueued behind the backward.
            # Measured on a one-rank RCCL group (scripts/diag_enqueue.py): splitting the last launch
            # (DENET_BWD_NO_DFS / ONLY_DFS) to overlap the all-reduce with the dFs stream costs 7 us for
            # the extra launch and 12 us more for running two hardware queues side by side -- 128.6 us
            # per step against 110.2 us for this order.
            if host_reduce:
                buf = eng.grad_bucket.cpu()
                dist.all_reduce(buf, op=reduce_op)
                eng.grad_bucket.copy_(buf)
            else:
                dist.all_reduce(eng.grad_bucket, op=reduce_op)

    dbg = (lambda m: print(f"[bench rank {rank}] {m}", file=sys.stderr, flush=True)) if os.environ.get("DENET_BENCH_DEBUG") else (lambda m: None)
    if os.environ.get("DENET_BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(30, exit=True)       # a hung rank prints its stack
    dbg("engine ready")
    # ---- warm-up (also loads code objects, sets LDS attributes) ----
    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    dbg("warm-up done")

    # ---- per-kernel breakdown (untimed): HIP events around each kernel in turn ----
    kb = kernel_bytes(shape, sF)
    breakdown = {}
    if args.breakdown or args.probe == "auto":
        e0, e1 = Event(), Event()
        for name in _lib.KERNEL_IDS:
            if name == "loss_finalize":
                continue                          # folded into the first backward launch (defer_losses)
            ts = []
            eng.set_probe(name, e0, e1)
            for _ in range(10):
                step()
                e1.synchronize()
                ts.append(e0.elapsed_ms(e1) * 1e3)
            breakdown[name] = statistics.median(ts)
        eng.set_probe(None, None, None)
    dbg("breakdown done")
    probe = args.probe
    if probe == "auto":
        probe = max(breakdown, key=breakdown.get)
    if probe == "none":
        probe = None

    runner = step
    graph = None
    if args.graph and world == 1:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            step()
        runner = graph.replay
        runner()
        torch.cuda.synchronize()

    # ---- timed region ----
    pairs = {}
    if probe and graph is None:
        # at least 8 samples whatever --steps is (the driver runs 20 steps)
        every = max(1, min(args.probe_every, args.steps // 8))
        # step 0 is not probed: the queue is empty there, the start event is reached before the host has
        # enqueued the kernel and the pair reads the host's launch time too (25 us against 18.5 for F1 at C2)
        first = 1 if args.steps > 8 * every else 0
        pairs = {i: (Event(), Event()) for i in range(first, args.steps, every)}
        if every < args.probe_every:
            # short regions (the driver's 20 steps): exactly 8 samples -- every pair costs ~3 us of stream time
            pairs = dict(list(pairs.items())[:8])
    # no collector pause inside a timed loop: re-binding for the probes allocates closures, and a full collection
    # of torch's object graph (measured 42 ms, once) would be charged to whichever step triggers it
    gc.collect()
    gc.disable()
    # the W warm-up steps run HERE, right before the timed region: the breakdown pass and the collection above
    # leave the GPU idle for tens of milliseconds, and the first steps after an idle period run below the
    # sustained clocks (20 timed steps read 95.3 us/step after the idle gap, 90.5 us in a 300-step run)
    # Settling run first (untimed set-up, like the breakdown pass above): so far the GPU has idled between short
    # synchronised bursts, and W = 5 warm-up steps (0.4 ms) do not bring it to its sustained state.
    for _ in range(0 if host_reduce else args.settle):      # (gloo dry runs reduce through host memory: ms per step)
        runner()
    for _ in range(max(args.warmup, 2)):
        runner()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dbg_ev = None
    if os.environ.get("DENET_BENCH_DEBUG") and graph is None:
        dbg_ev = (Event(), Event())
    t0 = time.perf_counter()
    if dbg_ev:
        dbg_ev[0].record(stream)
    host_us = []
    for i in range(args.steps):
        h0 = time.perf_counter()
        if pairs:
            if i in pairs:
                eng.set_probe(probe, *pairs[i])
            elif (i - 1) in pairs:
                eng.set_probe(None, None, None)
        runner()
        host_us.append((time.perf_counter() - h0) * 1e6)
    if dbg_ev:
        dbg_ev[1].record(stream)
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    if dbg_ev:
        dbg("timed region: host %.1f us (enqueue done at %.1f us), first-to-last event on the stream %.1f us" % (
            (t1 - t0) * 1e6, (t_enq - t0) * 1e6, dbg_ev[0].elapsed_ms(dbg_ev[1]) * 1e3))
    dbg("host us per step: median %.1f, max %.1f, over 200 us: %s" % (
        statistics.median(host_us), max(host_us), [(i, round(h)) for i, h in enumerate(host_us) if h > 200][:20]))
    eng.set_probe(None, None, None)
    dbg("timed region done")
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_reduce else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the same step in train mode (in-kernel dropout), timed the same way, reported beside the main line ----
    train_line = None
    if not args.train_mode and graph is None:
        eng.bind_step(*bind_args, drop=drop)
        train_now[0] = True
        for _ in range(5):
            step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el2 = time.perf_counter() - t2
        if dist is not None:
            t = torch.tensor([el2], dtype=torch.float64, device="cpu" if host_reduce else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        train_line = {"value": round(N * world * args.steps / el2, 1), "unit": "episodes/s",
                      "ms_per_step": round(el2 / args.steps * 1e3, 4),
                      "dropout": "Philox4x32-10 in F2 / F3 / B2, p=0.5 gate and class weights, new seed per step",
                      "loss_k": float(eng.losses[0].item())}
        train_now[0] = False
        gc.enable()
        eng.bind_step(*bind_args, drop=None)
        step()                                   # eng.losses back to the eval-mode values reported below
        torch.cuda.synchronize()

    gc.enable()
    probe_us = None
    if pairs:
        samples = [a.elapsed_ms(b) * 1e3 for a, b in pairs.values()]
        dbg("probe samples (us, in step order): %s" % [round(x, 1) for x in samples])
        probe_us = statistics.median(samples)
    elif probe:
        probe_us = breakdown.get(probe)

    losses = eng.losses.cpu().tolist()
    episodes = N * world * args.steps
    value = episodes / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    roofline = None
    if probe and probe_us:
        ach = kb[probe] / (probe_us * 1e-6) / 1e9
        # the dominant kernel (the fused up-sampling + loss launch) is bound by VALU issue, not by HBM or
        # MFMA; its HBM fraction is reported because the contract prices kernels against HBM or MFMA only
        traffic = pmc_traffic(probe, args.config, N)
        two_launches = probe == "masked_pool_fwd" and (N * way * shape.shot >= 64 or shape.h * shape.w >= 4096)
        if two_launches and traffic is not None:
            traffic += pmc_traffic("mask_resize", args.config, N) or 0
        roofline = {"kernel": probe, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": kb[probe], "us_per_launch": round(probe_us, 2),
                    "probed_steps": len(pairs) if pairs else 0, "us_per_launch_stat": "median of the probed steps"}
        if two_launches:
            # from 64 support images (or 64 x 64 features) on, the pooling stage is two launches and the probe
            # brackets both; rocprofv3 lists mask_resize_flat_kernel and masked_sum_fwd_kernel apart
            roofline["launches"] = ["mask_resize", "masked_pool_fwd"]
    # the two Fs-sized HBM streams, from the untimed per-kernel pass (event pair around each launch)
    streams = {k: {"us_per_launch": round(breakdown[k], 2), "bytes_per_launch": kb[k],
                   "achieved": round(kb[k] / (breakdown[k] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(kb[k] / (breakdown[k] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
               for k in ("masked_pool_fwd", "bwd_params_dfs") if k in breakdown}
    spb = step_bytes_per_episode(shape, sF)
    step_roof = {"bytes_per_episode": spb, "achieved": round(value / world * spb / 1e9, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(value / world * spb / 1e9 / HBM_PEAK_GBS, 4)}

    # the box's own stream-copy ceiling (read + write bytes of a 512 MiB device copy), for reading the
    # fractions above against what this HBM delivers rather than the 8 TB/s specification
    copy_gbs = None
    if rank == 0:
        a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(10):
            b.copy_(a)
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = round(10 * 2 * a.numel() / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
        del a, b

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.collective:
        from oracle import denet_oracle as O            # checker only: CPU baseline leg
        nthreads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1))
        torch.set_num_threads(nthreads)
        ts = []
        budget = time.perf_counter() + args.cpu_seconds
        while len(ts) < 4 or (time.perf_counter() < budget and len(ts) < 400):
            c0 = time.perf_counter()
            O.head_step(ep_cpu["Fq"], ep_cpu["Fs"], ep_cpu["Ys"], ep_cpu["label"], ep_cpu["Yq"], ep_cpu["Yq_full"],
                        ep_cpu["wgt"], state_cpu)
            ts.append(time.perf_counter() - c0)
        med = statistics.median(ts[1:])
        cpu = {"value": round(N / med, 3), "unit": "episodes/s", "cores": nthreads, "kind": "port",
               "sample": f"1 warm-up + {len(ts) - 1} timed steps (~{sum(ts):.0f} s) of the same {args.config} batch "
                         f"({N} episodes), oracle head_step (fwd + both losses + loss_k.backward), "
                         f"median {med * 1e3:.0f} ms/step"}

    if rank == 0:
        out = {
            "metric": "DENet head episodes/sec fwd+bwd (1-way 5-shot)" if args.config == "C2"
                      else f"DENet head episodes/sec fwd+bwd ({args.config})",
            "value": round(value, 1), "unit": "episodes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "settle_steps": args.settle, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if sF == 4 else "bf16-in/f32-acc",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {way}-way {shape.shot}-shot head fwd+bwd, K={shape.K}, "
                                   f"features {shape.c}x{shape.h}x{shape.w}, labels {shape.H}x{shape.W}",
                       "episodes_per_gpu": N, "global_batch": N * world, "parallelism": f"dp{world} (episode-sharded)",
                       "graph_replay": bool(graph),
                       "mode": "train (in-kernel dropout, p=0.5)" if args.train_mode else "eval (no dropout)"},
            "roofline": roofline, "step_roofline": step_roof, "hbm_stream_kernels": streams,
            "hbm_copy_measured_gbs": copy_gbs, "cpu_baseline": cpu, "train_mode": train_line,
            "kernels_us": {k: round(v, 2) for k, v in breakdown.items()},
            "loss_k": losses[0], "loss_p": losses[1],
        }
        print(json.dumps(out))
    if dist is not None:
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
