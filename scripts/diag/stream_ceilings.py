"""Diagnostic: what plain torch kernels reach on this box for a pure write stream, a pure read stream and a copy,
at the sizes of the step's two Fs-sized streams (C2: 69 MB, C3: 344 MB, C2x128: 1.1 GB)."""
import torch
dev = torch.device("cuda:0")
for mb in (69, 344, 1100):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    res = {}
    for name, fn, nbytes in (("fill (write)", lambda: a.fill_(1.0), 4 * n), ("sum (read)", lambda: a.sum(), 4 * n),
                              ("copy (read+write)", lambda: b.copy_(a), 8 * n), ("mul_ (read+write in place)", lambda: a.mul_(1.0001), 8 * n)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"{mb} MB:", ", ".join(f"{k} {v:.2f} TB/s" for k, v in res.items()), flush=True)
