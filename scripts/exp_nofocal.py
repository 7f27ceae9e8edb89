"""Experiment: step time with and without the focal tiles in the loss launch."""
import sys, time, torch
sys.path.insert(0, '.')
from denet_amd.engine import HeadStepEngine
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state
shape = CONFIGS["C2"]
dev = torch.device("cuda:0")
ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(shape.K, shape.c).items()}
N, way = shape.N, shape.way
labels = ep["label"].reshape(N, way).contiguous()
eng = HeadStepEngine(N, way, shape.shot, shape.K, shape.c, shape.h, shape.w, shape.H, shape.W, dev)
for focal in (True, False, True, False):
    def step():
        eng.forward(ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"] if focal else None, ep["Yq_full"], ep["wgt"], params,
                    with_grad=True, defer_losses=True)
        eng.backward(ep["Fq"], labels, params)
    for _ in range(20): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("focal" if focal else "no focal", f"{(t1 - t0) / 300 * 1e6:.1f} us/step")
