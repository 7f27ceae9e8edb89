"""Diagnostic: rate of the dFs stream (last backward launch) for feature rows that are / are not multiples of
128 bytes: P = 41*41 = 1681 floats (rows start at arbitrary 4-byte offsets) against P = 40*40 = 1600 (every row
starts on a 128-byte line)."""
import statistics, sys
sys.path.insert(0, ".")
import torch
from denet_amd.engine import HeadStepEngine
from denet_amd.hiprt import Event
from denet_amd.synthetic import EpisodeShape, make_episodes, make_estimator_state

dev = torch.device("cuda:0")
for (h, H) in ((41, 321), (40, 313)):
    for N in (8, 40):
        shape = EpisodeShape("x", N=N, way=1, shot=5, K=21, c=256, h=h, w=h, H=H, W=H, config_index=2)
        ep = {k: v.to(dev).contiguous() for k, v in make_episodes(shape).items()}
        params = {k: v.to(dev).contiguous() for k, v in make_estimator_state(shape.K, shape.c).items()}
        labels = ep["label"].reshape(N, 1).contiguous()
        eng = HeadStepEngine(N, 1, 5, 21, 256, h, h, H, H, dev)
        eng.bind_step(ep["Fq"], ep["Fs"], ep["Ys"], labels, ep["Yq"], ep["Yq_full"], ep["wgt"], params)
        st = torch.cuda.current_stream().cuda_stream
        e0, e1 = Event(), Event()
        out = {}
        for kern in ("bwd_params_dfs", "masked_pool_fwd"):
            for _ in range(5):
                eng.run_forward(st); eng.run_backward(0, st)
            ts = []
            eng.set_probe(kern, e0, e1)
            for _ in range(20):
                eng.run_forward(st); eng.run_backward(0, st)
                e1.synchronize()
                ts.append(e0.elapsed_ms(e1) * 1e3)
            eng.set_probe(None, None, None)
            t = statistics.median(ts[2:])
            out[kern] = (t, N * 5 * 256 * h * h * 4 / (t * 1e-6) / 1e12)
        print(f"P = {h}x{h}, {N} episodes: " + ", ".join(f"{k} {v[0]:.1f} us = {v[1]:.2f} TB/s" for k, v in out.items()), flush=True)
