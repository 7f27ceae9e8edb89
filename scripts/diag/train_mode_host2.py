"""Diagnostic: mimic bench.py --train-mode --probe masked_pool_fwd (fresh engine, warm-up, probed timed loop)."""
import sys, time
sys.path.insert(0, ".")
import torch
from denet_amd import _lib
from denet_amd.engine import HeadStepEngine
from denet_amd.hiprt import Event
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state

dev = torch.device("cuda:0")
shape = CONFIGS["C2"]
ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(shape.K, shape.c).items()}
N, way = shape.N, shape.way
labels = ep["label"].reshape(N, way).contiguous()
args = (ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params)
st = torch.cuda.current_stream().cuda_stream
for variant in sys.argv[1:] or ["bench"]:
    eng = HeadStepEngine(N, way, shape.shot, shape.K, shape.c, shape.h, shape.w, shape.H, shape.W, dev)
    drop = _lib.Dropout(0x5EED0000, 0.5, 0.5)
    eng.bind_step(*args, drop=drop)
    for _ in range(20):
        drop.seed += 1
        eng.run_forward(st); eng.run_backward(0, st)
    torch.cuda.synchronize()
    pairs = {i: (Event(), Event()) for i in range(0, 300, 8)}
    slow = []
    acc = [0.0, 0.0, 0.0]
    T0 = time.perf_counter()
    for i in range(300):
        a = time.perf_counter()
        if variant != "noprobe":
            if i in pairs:
                eng.set_probe("masked_pool_fwd", *pairs[i])
            elif (i - 1) in pairs:
                eng.set_probe(None, None, None)
        b = time.perf_counter()
        drop.seed += 1
        eng.run_forward(st)
        c = time.perf_counter()
        eng.run_backward(0, st)
        e = time.perf_counter()
        acc[0] += b - a; acc[1] += c - b; acc[2] += e - c
        if e - a > 300e-6:
            slow.append((i, round((b - a) * 1e6), round((c - b) * 1e6), round((e - c) * 1e6)))
    torch.cuda.synchronize()
    print(variant, f"{(time.perf_counter() - T0) / 300 * 1e6:.1f} us/step; host per step (probe, fwd, bwd) us:",
          [round(x / 300 * 1e6, 1) for x in acc], "slow calls:", slow[:10], flush=True)
    if pairs and variant != "noprobe":
        print("  probed kernel us:", [round(a.elapsed_ms(b) * 1e3, 1) for a, b in list(pairs.values())[:10]])
