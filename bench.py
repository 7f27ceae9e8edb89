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
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch and denet_amd are imported inside run(): with `--gpus N` and no WORLD_SIZE this process is only the
# launcher of N fresh rank processes and must not have touched the GPU (or loaded the library) itself

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
LLC_BYTES = 256 << 20        # last-level (Infinity) cache: the rotating leg's batches must exceed it together


def kernel_bytes(s, sF: int):
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


def kernel_bytes_compulsory(s, sF: int):
    """SURVEY.md §8d bytes per launch: each input read once per pass that needs it, each output written once;
    parameters and their slices (L2-resident, 0.81 MB) and split-K partials excluded.  This is what
    `roofline.achieved` is priced on; kernel_bytes() above (with the parameter re-reads and partial outputs a
    launch really touches) stays in the line as `bytes_incl_params`."""
    N, way, shot, K, C = s.N, s.way, s.shot, s.K, s.c
    B, P, Q, NB = N * way * shot, s.h * s.w, s.H * s.W, N * way
    return {
        "masked_pool_fwd": B * C * P * sF + B * P * 8,                        # Fs + the sampled mask elements
        "proto_extend_fwd": 0,
        "classify_fwd": N * C * P * sF + N * K * P * 4 + NB * 2 * P * 4,      # Fq in, both logit tensors out
        "loss_pair_fwd": N * Q * 8 + NB * Q * 8 + N * K * P * 4 + NB * 2 * P * 4 + 2 * N * K * P * 4,
        "loss_finalize": 0,
        "bwd_dwn": 2 * N * K * P * 4 + N * C * P * sF + N * C * P * sF,       # slabs re-read, Fq re-read, dFq out
        "bwd_chain_dfq": 0,
        "bwd_params_dfs": B * C * P * sF + B * P * 8,                         # dFs out + the mask row (§8d: w*s*P*8)
    }


def step_bytes_per_episode(s, sF: int) -> float:
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node to run on, one rank each.  Under torchrun (WORLD_SIZE set) it must equal "
                         "the world size; without it and N > 1 this process starts the N ranks itself")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4a", "C4b", "C5"])
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
    ap.add_argument("--no-rotating", action="store_true",
                    help="skip the second timed leg that rotates over resident batches larger than the last-level cache")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the leg that drives the same batch through the reference-shaped Python API")
    ap.add_argument("--dry-run", action="store_true",
                    help="join the process group, check that all ranks are there, print one line and stop before "
                         "the first kernel (CPU test of the launch path: DENET_BENCH_BACKEND=gloo)")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (torch.distributed.run, one per
    GPU, rendezvous on 127.0.0.1), relay rank 0's JSON line and return the children's exit code.  This process
    has not touched the GPU: nothing here imports torch.cuda state or the HIP library."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC for RCCL (see the environment notes)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out or out.startswith('{"dry_run"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    run(args)


def run(args):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}: one rank per GPU")
    backend = os.environ.get("DENET_BENCH_BACKEND", "nccl")

    if args.dry_run:
        # launch-path check: every rank joins, the group counts them, rank 0 reports; no kernel, no GPU needed
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            t = torch.ones(1, device=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
            t = torch.ones(1)
        dist.all_reduce(t)
        joined = int(t.item())
        assert joined == dist.get_world_size() == args.gpus, (joined, dist.get_world_size(), args.gpus)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": dist.get_world_size(), "ranks_joined": joined,
                              "backend": backend}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    from denet_amd import _lib
    from denet_amd.engine import HeadStepEngine
    from denet_amd.hiprt import Event
    from denet_amd.synthetic import CONFIGS, EpisodeShape, make_episodes, make_estimator_state

    if args.train_mode and args.graph:
        raise SystemExit("--train-mode with --graph would replay one captured seed; run them separately")
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

    # every rank on a GPU of its own: the line's n_gpus is the group's size, and the devices are distinct
    devices = [torch.cuda.current_device()]
    if dist is not None:
        assert dist.get_world_size() == world == args.gpus or args.collective, (dist.get_world_size(), world, args.gpus)
        onehot = torch.zeros(max(torch.cuda.device_count(), world), dtype=torch.int64,
                             device="cpu" if host_reduce else dev)
        onehot[torch.cuda.current_device()] = 1
        dist.all_reduce(onehot)
        devices = [i for i, c in enumerate(onehot.tolist()) for _ in range(c)]
        if not host_reduce and int(onehot.max()) > 1:
            raise SystemExit(f"two ranks share a GPU (ranks per device: {onehot.tolist()}): one rank per GPU")

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
    bound_eval = eng.make_bound(*bind_args, drop=None)
    bound_train = eng.make_bound(*bind_args, drop=drop)
    eng.use_bound(bound_train if args.train_mode else bound_eval)
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

    def timed(fn, steps):
        """`steps` calls of fn bracketed by barrier + synchronize on both sides; MAX over ranks, seconds."""
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if host_reduce else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

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
    kc = kernel_bytes_compulsory(shape, sF)
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
    host_us = []

    def timed_step(i):
        h0 = time.perf_counter()
        if pairs:
            if i in pairs:
                eng.set_probe(probe, *pairs[i])
            elif (i - 1) in pairs:
                eng.set_probe(None, None, None)
        runner()
        host_us.append((time.perf_counter() - h0) * 1e6)

    elapsed = timed(timed_step, args.steps)
    dbg("host us per step: median %.1f, max %.1f, over 200 us: %s" % (
        statistics.median(host_us), max(host_us), [(i, round(h)) for i, h in enumerate(host_us) if h > 200][:20]))
    eng.set_probe(None, None, None)
    dbg("timed region done")
    losses = eng.losses.cpu().tolist()

    # ---- second timed leg: the same step ROTATING over resident batches that together exceed the last-level
    # cache.  The main leg re-reads one batch every step, and at C2 x 8 its ~200 MB working set stays in the
    # 256 MB Infinity Cache (what a training loop over fresh batches would not see); here every step's inputs
    # come from HBM.  Same engine (outputs, gradients, workspace), R pre-bound input sets.
    rotating = None
    if not args.no_rotating and graph is None and not host_reduce:
        in_bytes = sum(ep[k].numel() * ep[k].element_size() for k in ("Fq", "Fs", "Ys", "Yq", "Yq_full", "wgt"))
        R = max(3, -(-int(1.25 * LLC_BYTES) // in_bytes))
        bounds = [bound_train if args.train_mode else bound_eval]
        keep = []
        for j in range(1, R):
            e2 = {k: v.to(dev).contiguous() for k, v in make_episodes(shape, seed_offset=1000 * rank + 31 * j).items()}
            keep.append(e2)
            bounds.append(eng.make_bound(e2["Fq"], e2["Fs"], e2["Ys"], e2["label"].reshape(N, way).contiguous(),
                                         e2["Yq"], e2["Yq_full"], e2["wgt"], params,
                                         drop=drop if args.train_mode else None))

        def rot_step(i):
            eng.use_bound(bounds[i % R])
            step()

        for i in range(max(args.warmup, 2 * R)):
            rot_step(i)
        el_r = timed(rot_step, args.steps)
        eng.use_bound(bounds[0])
        rotating = {"value": round(N * world * args.steps / el_r, 1), "unit": "episodes/s",
                    "ms_per_step": round(el_r / args.steps * 1e3, 4), "batches": R,
                    "input_bytes_per_batch": in_bytes,
                    "note": "inputs rotate over %d resident batches (%.0f MB in total, last-level cache 256 MiB); "
                            "outputs / gradients / workspace are the engine's own buffers" % (R, R * in_bytes / 1e6)}
        del keep, bounds

    # ---- train mode (in-kernel dropout) beside eval mode: alternating segments of `steps` steps each, so that
    # both readings come from the same number of steps under the same clocks (a single 20-step region carries
    # +-2 % of noise, more than the ~1.5 us the generator costs) ----
    train_line = None
    if not args.train_mode and graph is None:
        seg = {False: [], True: []}
        for rep in range(3):
            for mode in (False, True):
                eng.use_bound(bound_train if mode else bound_eval)
                train_now[0] = mode
                for _ in range(5):
                    step()
                seg[mode].append(timed(lambda i: step(), args.steps))
        train_now[0] = False
        el_e, el_t = statistics.median(seg[False]), statistics.median(seg[True])
        torch.cuda.synchronize()
        train_line = {"value": round(N * world * args.steps / el_t, 1), "unit": "episodes/s",
                      "ms_per_step": round(el_t / args.steps * 1e3, 4),
                      "ms_per_step_eval_interleaved": round(el_e / args.steps * 1e3, 4),
                      "segments": "3 alternating pairs of %d-step regions (eval, train), medians" % args.steps,
                      "dropout": "Philox4x32-10 in F2 / F3 / B2, p=0.5 gate and class weights, new seed per step",
                      "loss_k": float(eng.losses[0].item())}
        eng.use_bound(bound_eval)
    gc.enable()

    probe_us = None
    if pairs:
        samples = [a.elapsed_ms(b) * 1e3 for a, b in pairs.values()]
        dbg("probe samples (us, in step order): %s" % [round(x, 1) for x in samples])
        probe_us = statistics.median(samples)
    elif probe:
        probe_us = breakdown.get(probe)

    episodes = N * world * args.steps
    value = episodes / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    roofline = None
    if probe and probe_us:
        # algorithmic (compulsory) bytes of the launch / its measured duration
        nbytes = kc[probe] or kb[probe]
        ach = nbytes / (probe_us * 1e-6) / 1e9
        # the dominant kernel (the fused up-sampling + loss launch) is bound by VALU issue, not by HBM or
        # MFMA; its HBM fraction is reported because the contract prices kernels against HBM or MFMA only
        traffic = pmc_traffic(probe, args.config, N)
        two_launches = probe == "masked_pool_fwd" and (N * way * shape.shot >= 64 or shape.h * shape.w >= 4096)
        if two_launches and traffic is not None:
            traffic += pmc_traffic("mask_resize", args.config, N) or 0
        roofline = {"kernel": probe, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": "profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes of this command, committed; not re-measured in this run)"
                                      if traffic is not None else None,
                    "bytes_per_launch": nbytes, "bytes_definition": "SURVEY.md 8d compulsory bytes",
                    "bytes_incl_params": kb[probe], "us_per_launch": round(probe_us, 2),
                    "probed_steps": len(pairs) if pairs else 0, "us_per_launch_stat": "median of the probed steps"}
        if two_launches:
            # from 64 support images (or 64 x 64 features) on, the pooling stage is two launches and the probe
            # brackets both; rocprofv3 lists mask_resize_flat_kernel and masked_sum_fwd_kernel apart
            roofline["launches"] = ["mask_resize", "masked_pool_fwd"]
    # the two Fs-sized HBM streams, from the untimed per-kernel pass (event pair around each launch)
    streams = {k: {"us_per_launch": round(breakdown[k], 2), "bytes_per_launch": kc[k],
                   "achieved": round(kc[k] / (breakdown[k] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(kc[k] / (breakdown[k] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
               for k in ("masked_pool_fwd", "bwd_params_dfs") if k in breakdown}
    spb = step_bytes_per_episode(shape, sF)
    step_roof = {"bytes_per_episode": spb, "achieved": round(value / world * spb / 1e9, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(value / world * spb / 1e9 / HBM_PEAK_GBS, 4)}
    if rotating is not None:
        step_roof["frac_rotating"] = round(rotating["value"] / world * spb / 1e9 / HBM_PEAK_GBS, 4)

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

    # ---- the drop-in API on the same batch (rank 0): what a user of the reference calls ----
    dropin = None
    if rank == 0 and not args.no_dropin and graph is None:
        dropin = dropin_leg(torch, shape, ep, labels, state_cpu, dev, fdt)

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
        n_gpus = dist.get_world_size() if dist is not None else 1
        assert n_gpus == world
        ws_mb = (eng.ws_bytes + eng.dFs.numel() * eng.dFs.element_size() + eng.dFq.numel() * eng.dFq.element_size()
                 + sum(ep[k].numel() * ep[k].element_size() for k in ep)) / 1e6
        out = {
            "metric": "DENet head episodes/sec fwd+bwd (1-way 5-shot)" if args.config == "C2"
                      else f"DENet head episodes/sec fwd+bwd ({args.config})",
            "value": round(value, 1), "unit": "episodes/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "settle_steps": args.settle, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32" if sF == 4 else "bf16-in/f32-acc",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {way}-way {shape.shot}-shot head fwd+bwd, K={shape.K}, "
                                   f"features {shape.c}x{shape.h}x{shape.w}, labels {shape.H}x{shape.W}",
                       "episodes_per_gpu": N, "global_batch": N * world, "parallelism": f"dp{world} (episode-sharded)",
                       "graph_replay": bool(graph), "devices": devices,
                       "mode": "train (in-kernel dropout, p=0.5)" if args.train_mode else "eval (no dropout)",
                       "residency": "one batch re-read every step; working set %.0f MB %s the 256 MiB last-level "
                                    "cache (see value_rotating)" % (ws_mb, "fits" if ws_mb * 1e6 < LLC_BYTES else "exceeds")},
            "value_rotating": rotating["value"] if rotating else None, "rotating": rotating,
            "roofline": roofline, "step_roofline": step_roof, "hbm_stream_kernels": streams,
            "hbm_copy_measured_gbs": copy_gbs, "cpu_baseline": cpu, "train_mode": train_line, "dropin": dropin,
            "kernels_us": {k: round(v, 2) for k, v in breakdown.items()},
            # which variant each launch ran (denet_head_step_plan: the launchers' own decision functions)
            "kernel_plan": _lib.head_step_plan(N, way, shape.shot, shape.K, shape.c, shape.h, shape.w, shape.H, shape.W,
                                               _lib.BF16 if shape.feat_dtype == "bf16" else _lib.F32),
            "loss_k": losses[0], "loss_p": losses[1],
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()


def dropin_leg(torch, shape, ep, labels, state_cpu, dev, fdt, steps=200):
    """The same batch through the reference-shaped Python API, fwd + both losses + loss_k.backward(), us per
    step (host-synchronised loop of `steps` steps after 10 warm-up steps):
      * `head_step`: WeightEstimator.head_step -- what DENet.forward_step / DENetTrainer.step run: ONE
        HeadStepFunction node = the seven launches the headline times, plus torch's autograd / allocator and the
        ctypes argument conversion on the host;
      * `standalone`: WeightEstimator.forward_with_binary + loss_denet_batch + backward -- the stand-alone
        kernel chain (one autograd node per reference call), the drop-in path of rounds 1-2."""
    from denet_amd.head import WeightEstimator
    from denet_amd.losses import loss_denet_batch
    we = WeightEstimator(shape.K, shape.c).to(dev).eval()
    we.load_state_dict(state_cpu)
    Fq = ep["Fq"].detach().clone().requires_grad_(True)
    Fs = ep["Fs"].detach().clone().requires_grad_(True)

    def fused():
        lk, lp, full, binl = we.head_step(Fq, Fs, ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"])
        lk.backward()

    def standalone():
        full, binl = we.forward_with_binary(Fq, Fs, ep["Ys"], labels)
        lk, lp = loss_denet_batch(full, ep["Yq_full"], binl, ep["Yq"], ep["wgt"])
        lk.backward()

    res = {}
    leaves = [Fq, Fs] + list(we.parameters())
    gc.collect()
    gc.disable()                # (as in the main timed region: no collector pause charged to a step)
    for name, fn in (("head_step", fused), ("standalone", standalone)):
        for _ in range(10):
            for t in leaves:
                t.grad = None
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for t in leaves:
                t.grad = None
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res[name + "_us_per_step"] = round((time.perf_counter() - t0) / steps * 1e6, 1)
        res[name + "_host_enqueue_us"] = round((t1 - t0) / steps * 1e6, 1)
    gc.enable()
    res["steps"] = steps
    res["episodes_per_s_head_step"] = round(shape.N / (res["head_step_us_per_step"] * 1e-6), 1)
    res["note"] = ("eval mode, same C-ABI kernels as the headline; head_step = WeightEstimator.head_step "
                   "(DENet.forward_step / DENetTrainer.step), standalone = forward_with_binary + loss_denet_batch")
    return res


if __name__ == "__main__":
    main()
