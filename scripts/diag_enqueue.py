#!/usr/bin/env python3
"""CPU-side enqueue cost of one head step (is the bench loop GPU-bound?), and of the N > 1 variant
(two backward calls + an async all-reduce on a one-rank RCCL group)."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from denet_amd.engine import HeadStepEngine
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
s = CONFIGS["C2"]
ep = {k: v.to(dev).contiguous() for k, v in make_episodes(s).items()}
params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(s.K, s.c).items()}
labels = ep["label"].reshape(s.N, s.way).contiguous()
eng = HeadStepEngine(s.N, s.way, s.shot, s.K, s.c, s.h, s.w, s.H, s.W, dev)

def fwd():
    eng.forward(ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params, with_grad=True, defer_losses=True)
def step1():
    fwd(); eng.backward(ep["Fq"], labels, params)
def stepN():
    fwd(); eng.backward(ep["Fq"], labels, params, phase=1)
    w = dist.all_reduce(eng.grad_bucket, op=dist.ReduceOp.AVG, async_op=True)
    eng.backward(ep["Fq"], labels, params, phase=2); w.wait()
eng.bind_step(ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params)
st = torch.cuda.current_stream().cuda_stream
def bstep1():
    eng.run_forward(st); eng.run_backward(0, st)
def bstepN():
    eng.run_forward(st); eng.run_backward(1, st)
    w = dist.all_reduce(eng.grad_bucket, op=dist.ReduceOp.AVG, async_op=True)
    eng.run_backward(2, st); w.wait()
def bstepA():
    eng.run_forward(st); eng.run_backward(0, st)
    dist.all_reduce(eng.grad_bucket, op=dist.ReduceOp.AVG)
def bstepS():
    eng.run_forward(st); eng.run_backward(0, st)
    dist.all_reduce(eng.grad_bucket, op=dist.ReduceOp.SUM)
def bstepC():
    eng.run_forward(st); eng.run_backward(1, st); eng.run_backward(2, st)
def bstepD():
    eng.run_forward(st); eng.run_backward(1, st)
    dist.all_reduce(eng.grad_bucket, op=dist.ReduceOp.AVG)
    eng.run_backward(2, st)
side = torch.cuda.Stream()
ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()
def bstepE():
    # dFs on a side stream, all-reduce (sync op) on the main one
    eng.run_forward(st); eng.run_backward(1, st)
    ev1.record(); side.wait_event(ev1)
    eng.run_backward(2, side.cuda_stream)
    dist.all_reduce(eng.grad_bucket, op=dist.ReduceOp.AVG)
    ev2.record(side); torch.cuda.current_stream().wait_event(ev2)
for f in (bstep1, bstepN, bstepA, bstepC, bstepD, bstepE):
    for _ in range(20): f()
    torch.cuda.synchronize()
    cpu = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); cpu.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): f()
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / 500 * 1e6
    print(f"{f.__name__}: CPU enqueue median {statistics.median(cpu):.1f} us (min {min(cpu):.1f}); steady-state {thr:.1f} us/step", flush=True)
# component costs (CPU, queue drained before each sample)
import collections
acc = collections.defaultdict(list)
for _ in range(50):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    fwd(); t.append(time.perf_counter())
    eng.backward(ep["Fq"], labels, params, phase=1); t.append(time.perf_counter())
    w = dist.all_reduce(eng.grad_bucket, op=dist.ReduceOp.AVG, async_op=True); t.append(time.perf_counter())
    eng.backward(ep["Fq"], labels, params, phase=2); t.append(time.perf_counter())
    w.wait(); t.append(time.perf_counter())
    for name, a, b in zip(["fwd", "bwd1", "all_reduce", "bwd2", "wait"], t[:-1], t[1:]):
        acc[name].append((b - a) * 1e6)
print({k: round(statistics.median(v), 1) for k, v in acc.items()}, flush=True)
os._exit(0)
