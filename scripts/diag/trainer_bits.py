import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from denet_amd import ops
from denet_amd.trainer import DENetTrainer
ORDER = ["weight", "gam.gate.0.weight", "gam.gate.0.bias", "gam.gate.2.weight", "gam.gate.2.bias", "map.linear.weight", "map.linear.bias"]
dev = torch.device("cuda:0")
cfg = {"model": {"arch": "DENet", "backbone": "synthetic", "maximum_num_classes": 21},
       "data": {"train_dataset": "synthetic", "val_dataset": "synthetic", "img_size": 65, "shot": 2, "batch_size": 4, "batches_per_epoch": 2, "val_batches": 1},
       "train": {"trainer": "DENetTrainer", "device": 0, "num_steps": 6, "log_steps": 100, "val_steps": 2, "lambda": 2.0,
                 "optimizer": {"name": "sgd", "lr": 0.05, "momentum": 0.9}, "loss": {"name": "loss_denet"}}}
torch.manual_seed(0)
tr = DENetTrainer(cfg)
batch = tr.train_loader.batch(0)
Is, Ys, Iq, Yq, cls, Ys_full, Yq_full = [t.to(dev) for t in batch]
N, way, shot = Is.shape[:3]
wgt = torch.ones(N, 21, device=dev); wgt[torch.arange(N, device=dev), cls.reshape(-1)] = 2.0
model, est = tr.model, tr.model.estimator
names = ["estimator." + k for k in ORDER]
params = dict(model.named_parameters())
model.train()
def hand():
    torch.manual_seed(123)
    Fs = model.features(Is.reshape(-1, *Is.shape[3:])).reshape(N, way, shot, 256, 9, 9)
    Fq = model.features(Iq)
    drop = est.dropout_config()
    lk, lp, _, _ = ops.HeadStepFunction.apply(Fq, Fs, *[params[n] for n in names], Ys, cls.reshape(N, way), Yq.reshape(N * way, 65, 65), Yq_full, wgt, None, None, drop)
    tr.bucket.zero()
    lk.backward()
    return float(lk), {n: params[n].grad.clone() for n in names}, Fs.detach().clone(), Fq.detach().clone()
l1, g1, Fs1, Fq1 = hand()
l2, g2, Fs2, Fq2 = hand()
print("hand vs hand: loss", l1 == l2, "Fs", torch.equal(Fs1, Fs2), "Fq", torch.equal(Fq1, Fq2))
for n in names:
    print("  ", n, torch.equal(g1[n], g2[n]), float((g1[n] - g2[n]).abs().max()))
w0 = {n: params[n].detach().clone() for n in names}
torch.manual_seed(123)
out = tr.step(batch, train=True)
print("trainer loss", out[3] == l1)
for n in names:
    print("  ", n, torch.equal(params[n].grad, g1[n]), float((params[n].grad - g1[n]).abs().max()), "param moved", not torch.equal(params[n].detach(), w0[n]))
