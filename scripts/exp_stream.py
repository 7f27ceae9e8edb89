"""Experiment: the two Fs-sized streams (K0+K1 pooling, K1b dFs) stand-alone, at batch 8 and 128."""
import sys, torch
sys.path.insert(0, '.')
from denet_amd import ops
dev = torch.device("cuda:0")
def t(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
from denet_amd._lib import call, feat_code
for B in (40, 200, 640):
    C, h, w, H, W = 256, 41, 41, 321, 321
    Fs = torch.randn(B, C, h, w, device=dev)
    Ys = torch.randint(0, 2, (B, H, W), device=dev)
    mask, cnt, S = torch.empty(B, h * w, device=dev), torch.empty(B, device=dev), torch.empty(B, C, device=dev)
    coef = torch.randn(B, C, device=dev); dFs = torch.empty_like(Fs)
    f = lambda: call("denet_masked_pool_fwd", Fs, feat_code(Fs), Ys, B, C, H, W, h, w, mask, cnt, S)
    g = lambda: call("denet_masked_sum_bwd", mask, coef, B, C, h * w, dFs, feat_code(dFs))
    c = lambda: dFs.copy_(Fs)
    nb = Fs.numel() * 4
    tf, tg, tc = t(f), t(g), t(c)
    print(f"B={B}: pool fwd {tf:.1f} us ({nb / tf / 1e6:.2f} TB/s) | dFs {tg:.1f} us ({nb / tg / 1e6:.2f} TB/s) | torch copy {tc:.1f} us ({2 * nb / tc / 1e6:.2f} TB/s r+w)")
