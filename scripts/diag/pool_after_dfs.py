"""Diagnostic: duration of the pooling kernel (F1) inside the step loop, with and without the dFs stream of the
previous step's last launch (DENET_BWD_NO_DFS), C3 shapes: how much of F1's in-loop time is the write-back of
the 344 MB of dFs lines the previous launch left dirty in L2 / MALL."""
import statistics, sys
sys.path.insert(0, ".")
import torch
from denet_amd.engine import HeadStepEngine
from denet_amd.hiprt import Event
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state

dev = torch.device("cuda:0")
for cfg in sys.argv[1:] or ["C3"]:
    shape = CONFIGS[cfg]
    fdt = torch.bfloat16 if shape.feat_dtype == "bf16" else torch.float32
    ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
    params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(shape.K, shape.c).items()}
    N, way = shape.N, shape.way
    labels = ep["label"].reshape(N, way).contiguous()
    eng = HeadStepEngine(N, way, shape.shot, shape.K, shape.c, shape.h, shape.w, shape.H, shape.W, dev, fdt)
    eng.bind_step(ep["Fq"].to(fdt), ep["Fs"].to(fdt), ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params)
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = Event(), Event()
    for phase, tag in ((0, "full backward (dFs written)"), (1, "backward without the dFs blocks")):
        for _ in range(5):
            eng.run_forward(st); eng.run_backward(phase, st)
        ts = []
        eng.set_probe("masked_pool_fwd", e0, e1)
        for _ in range(20):
            eng.run_forward(st); eng.run_backward(phase, st)
            e1.synchronize()
            ts.append(e0.elapsed_ms(e1) * 1e3)
        eng.set_probe(None, None, None)
        print(f"{cfg}: masked_pool_fwd after {tag}: median {statistics.median(ts[2:]):.1f} us", flush=True)
