"""Diagnostic: duration of the loss launch (F4) of the fused step for different label-map statistics:
uniformly random labels (bench.py's synthetic data), piecewise-constant maps (what a segmentation data set
holds: blobs of 40 x 40 pixels), and a single class everywhere (the worst case for the LDS moment atomics)."""
import statistics, sys
sys.path.insert(0, ".")
import torch
from denet_amd.engine import HeadStepEngine
from denet_amd.hiprt import Event
from denet_amd.synthetic import CONFIGS, make_episodes, make_estimator_state

dev = torch.device("cuda:0")
for cfg in sys.argv[1:] or ["C2", "C3"]:
    shape = CONFIGS[cfg]
    fdt = torch.bfloat16 if shape.feat_dtype == "bf16" else torch.float32
    ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
    params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(shape.K, shape.c).items()}
    N, way, K, H, W = shape.N, shape.way, shape.K, shape.H, shape.W
    labels = ep["label"].reshape(N, way).contiguous()
    g = torch.Generator().manual_seed(3)
    blobs = torch.randint(0, K, (N, (H + 39) // 40, (W + 39) // 40), generator=g)
    blobs = blobs.repeat_interleave(40, 1).repeat_interleave(40, 2)[:, :H, :W].contiguous().to(dev)
    bin_blobs = (blobs % 2).contiguous()
    pats = {"random": (ep["Yq_full"], ep["Yq"]),
            "blobs 40x40": (blobs, bin_blobs if way == 1 else ep["Yq"]),
            "one class": (torch.full_like(ep["Yq_full"], 3), torch.ones_like(ep["Yq"]))}
    eng = HeadStepEngine(N, way, shape.shot, K, shape.c, shape.h, shape.w, H, W, dev, fdt)
    st = torch.cuda.current_stream().cuda_stream
    e0, e1 = Event(), Event()
    for tag, (yf, yq) in pats.items():
        eng.bind_step(ep["Fq"].to(fdt), ep["Fs"].to(fdt), ep["Ys"], labels, yq, yf, ep["wgt"], params)
        for _ in range(5):
            eng.run_forward(st); eng.run_backward(0, st)
        ts = []
        eng.set_probe("loss_pair_fwd", e0, e1)
        for _ in range(20):
            eng.run_forward(st); eng.run_backward(0, st)
            e1.synchronize()
            ts.append(e0.elapsed_ms(e1) * 1e3)
        eng.set_probe(None, None, None)
        torch.cuda.synchronize()
        print(f"{cfg}: F4 with {tag} labels: median {statistics.median(ts[2:]):.1f} us, loss_k {eng.losses[0].item():.4f}", flush=True)
