"""Diagnostic: host and device time of the bound fused step, eval vs train mode (in-kernel dropout)."""
import sys, time
sys.path.insert(0, ".")
import torch
from denet_amd import _lib
from denet_amd.engine import HeadStepEngine
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state

dev = torch.device("cuda:0")
shape = CONFIGS["C2"]
ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(shape.K, shape.c).items()}
N, way = shape.N, shape.way
labels = ep["label"].reshape(N, way).contiguous()
eng = HeadStepEngine(N, way, shape.shot, shape.K, shape.c, shape.h, shape.w, shape.H, shape.W, dev)
args = (ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params)
drop = _lib.Dropout(123, 0.5, 0.5)
st = torch.cuda.current_stream().cuda_stream
for mode, d, bump in (("eval", None, False), ("train", drop, True), ("train-fixed-seed", drop, False), ("eval", None, False)):
    eng.bind_step(*args, drop=d)
    for _ in range(20):
        eng.run_forward(st); eng.run_backward(0, st)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(300):
        h0 = time.perf_counter()
        if bump:
            drop.seed += 1
        eng.run_forward(st); eng.run_backward(0, st)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{mode}: {el / 300 * 1e6:.1f} us/step wall, host enqueue {host / 300 * 1e6:.1f} us/step", flush=True)

# cost of re-binding (set_probe) with and without the dropout struct, and of probed steps
from denet_amd.hiprt import Event
e0, e1 = Event(), Event()
for mode, d in (("eval", None), ("train", drop)):
    eng.bind_step(*args, drop=d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.set_probe("masked_pool_fwd", e0, e1)
        eng.set_probe(None, None, None)
    print(f"{mode}: set_probe pair {(time.perf_counter() - t0) / 50 * 1e6:.1f} us", flush=True)
    for probed in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(300):
            if probed and i % 8 == 0:
                eng.set_probe("masked_pool_fwd", e0, e1)
            elif probed and i % 8 == 1:
                eng.set_probe(None, None, None)
            if d is not None:
                drop.seed += 1
            eng.run_forward(st); eng.run_backward(0, st)
        torch.cuda.synchronize()
        print(f"{mode}: probed={probed} {(time.perf_counter() - t0) / 300 * 1e6:.1f} us/step", flush=True)

# separate the re-bind from the event records, and find the slow step
for what in ("rebind-only", "probe"):
    eng.bind_step(*args, drop=drop)
    torch.cuda.synchronize()
    per = []
    for i in range(40):
        if i % 8 == 0:
            if what == "probe":
                eng.set_probe("masked_pool_fwd", e0, e1)
            else:
                eng._rebind()
        elif i % 8 == 1:
            if what == "probe":
                eng.set_probe(None, None, None)
            else:
                eng._rebind()
        drop.seed += 1
        t0 = time.perf_counter()
        eng.run_forward(st)
        t1 = time.perf_counter()
        eng.run_backward(0, st)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        per.append((round((t1 - t0) * 1e6), round((t2 - t1) * 1e6), round((t3 - t2) * 1e6)))
    print(what, per[6:20], flush=True)

# async loop, train mode, probed: where does the host spend its time?
for mode, d in (("eval", None), ("train", drop)):
    eng.bind_step(*args, drop=d)
    torch.cuda.synchronize()
    acc = {"probe": 0.0, "fwd": 0.0, "bwd": 0.0}
    slow = []
    T0 = time.perf_counter()
    for i in range(300):
        a = time.perf_counter()
        if i % 8 == 0:
            eng.set_probe("masked_pool_fwd", e0, e1)
        elif i % 8 == 1:
            eng.set_probe(None, None, None)
        b = time.perf_counter()
        eng.run_forward(st)
        c = time.perf_counter()
        eng.run_backward(0, st)
        e = time.perf_counter()
        acc["probe"] += b - a; acc["fwd"] += c - b; acc["bwd"] += e - c
        if e - a > 200e-6:
            slow.append((i, round((b - a) * 1e6), round((c - b) * 1e6), round((e - c) * 1e6)))
    torch.cuda.synchronize()
    print(mode, f"{(time.perf_counter() - T0) / 300 * 1e6:.1f} us/step;", {k: round(v / 300 * 1e6, 1) for k, v in acc.items()}, slow[:12], flush=True)

# which ingredient makes the probed + seed-bumped loop slow?
def loop(tag, d, bump, probe_kernel="masked_pool_fwd", fresh_events=False, stride=1):
    eng.bind_step(*args, drop=d)
    torch.cuda.synchronize()
    evs = [(Event(), Event()) for _ in range(40)] if fresh_events else None
    T0 = time.perf_counter()
    for i in range(300):
        if i % 8 == 0:
            a, b = evs[i // 8] if evs else (e0, e1)
            eng.set_probe(probe_kernel, a, b)
        elif i % 8 == 1:
            eng.set_probe(None, None, None)
        if bump:
            d.seed += stride
        eng.run_forward(st); eng.run_backward(0, st)
    torch.cuda.synchronize()
    print(tag, f"{(time.perf_counter() - T0) / 300 * 1e6:.1f} us/step", flush=True)

loop("train bump", _lib.Dropout(123, 0.5, 0.5), True)
loop("train bump fresh events", _lib.Dropout(123, 0.5, 0.5), True, fresh_events=True)
loop("train bump probe=classify_fwd", _lib.Dropout(123, 0.5, 0.5), True, probe_kernel="classify_fwd")
loop("p=0 bump", _lib.Dropout(123, 0.0, 0.0), True)
loop("gate only bump", _lib.Dropout(123, 0.5, 0.0), True)
loop("weight only bump", _lib.Dropout(123, 0.0, 0.5), True)
loop("train bump big stride", _lib.Dropout(1 << 40, 0.5, 0.5), True, stride=1 << 33)

loop("seed 0x5EED0000", _lib.Dropout(0x5EED0000, 0.5, 0.5), True, fresh_events=True)
loop("seed 723", _lib.Dropout(723, 0.5, 0.5), True)
loop("seed 0x5EED0000 no bump", _lib.Dropout(0x5EED0000, 0.5, 0.5), False)
loop("seed 123 again", _lib.Dropout(123, 0.5, 0.5), True)
