"""CPU, world_size 2, gloo: the N > 1 path of SURVEY.md §8e.

Episodes are sharded in contiguous blocks; the only exchanges are the integer
metric all-reduce at the end of evaluation and the flat gradient all-reduce per
training step.  The merged metrics must be bit-identical to a single-process
run over the whole episode stream (integer sums; BinaryIOU's first-seen class
order restored from the all-reduced `min` of the order keys).
"""
import os
import socket
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _episode_stream(K=7, batches=5, N=6, H=11, W=13, way=2):
    """Predictions and labels of a small evaluation run (CPU numpy): per batch
    (true_full [N,H,W], pred_full [N,H,W], per way: true_bin, pred_bin [N,H,W], cls [N,way])."""
    g = np.random.default_rng(321)
    out = []
    for b in range(batches):
        tf = g.integers(0, K, (N, H, W))
        tf[:, 0] = 255
        pf = g.integers(0, K, (N, H, W))
        tb = g.integers(0, 2, (N, way, H, W))
        tb[g.integers(0, N), g.integers(0, way)] = 0            # a skipped sample
        pb = g.integers(0, 2, (N, way, H, W))
        cls = g.integers(1, K, (N, way))
        out.append((tf, pf, tb, pb, cls))
    return out


def _counts(tb, pb):
    s = tb + pb
    n = tb.shape[0]
    return torch.from_numpy(np.stack([tb.reshape(n, -1).sum(1), (s == 2).reshape(n, -1).sum(1),
                                      ((s == 1) | (s == 2)).reshape(n, -1).sum(1)], 1).astype(np.int64))


def _evaluate(rank, world, K=7, way=2):
    """The loop of DENetTrainer.test (denet_amd/trainer.py) with the kernels' outputs injected."""
    from denet_amd.metrics import BinaryIOU, FullIOU
    from denet_amd.sharding import shard_range
    from oracle import denet_oracle as O
    fi, bi = FullIOU(num_classes=K), BinaryIOU()
    for j, (tf, pf, tb, pb, cls) in enumerate(_episode_stream(K=K, way=way)):
        n_glob = tf.shape[0]
        lo, hi = shard_range(n_glob, rank, world)
        fi.update_counts(O.confusion_matrix(tf[lo:hi], pf[lo:hi], K))
        for i in range(way):
            keys = (j * way + i) * n_glob + lo + np.arange(hi - lo)
            bi.update_counts(_counts(tb[lo:hi, i], pb[lo:hi, i]), cls[lo:hi, i], order_keys=keys)
    return fi, bi


def _worker(rank, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      LOCAL_RANK=str(rank))
    from denet_amd import sharding
    r, w, _ = sharding.init_from_env(backend="gloo")
    assert (r, w) == (rank, WORLD)
    fi, bi = _evaluate(rank, WORLD)
    fi.all_reduce()
    bi.all_reduce(num_classes=7)
    # gradient bucket: mean over ranks of per-rank gradients
    lin = torch.nn.Linear(4, 3)
    with torch.no_grad():
        lin.weight.fill_(0.5)
        lin.bias.zero_()
    bucket = sharding.GradBucket(lin.parameters())
    x = torch.arange(8.0).reshape(2, 4) + 10 * rank
    lin(x).pow(2).mean().backward()
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    # BatchNorm running buffers: every rank saw its own shard's statistics; sync_buffers makes them one model
    bn = torch.nn.BatchNorm2d(3)
    bn.train()
    bn(torch.randn(4, 3, 5, 5, generator=torch.Generator().manual_seed(7 + rank)) * (1 + rank) + rank)
    bn_local = {k: v.clone() for k, v in bn.state_dict().items()}
    sharding.sync_buffers(bn)
    # train-mode dropout seed: torch's generator starts from the same state on every rank, the seeds must not
    from denet_amd.head import WeightEstimator
    torch.manual_seed(0)
    we = WeightEstimator(5, 64).train()
    seeds = [we.dropout_config()[0] for _ in range(3)]
    torch.save({"mat": fi.counts(), "miou": float(fi.mean_iou()), "keys": list(bi.inter.keys()),
                "bn_local": bn_local, "bn_synced": {k: v.clone() for k, v in bn.state_dict().items()},
                "seeds": seeds,
                "inter": list(bi.inter.values()), "union": list(bi.union.values()), "biou": bi.mean_iou(),
                "bf": bi.mean_bf_iou(), "grad_local": local, "grad_avg": bucket.flat.clone()},
               os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_metrics_and_gradients_match_single_process():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(_free_port(), tmp), nprocs=WORLD, join=True)
        res = [torch.load(os.path.join(tmp, f"r{r}.pt"), weights_only=False) for r in range(WORLD)]
    fi, bi = _evaluate(0, 1)                                   # single process over every episode
    for r in res:
        assert np.array_equal(r["mat"], fi.counts())
        assert r["miou"] == float(fi.mean_iou())
        assert r["keys"] == list(bi.inter.keys())              # first-seen order restored
        assert r["inter"] == list(bi.inter.values()) and r["union"] == list(bi.union.values())
        assert r["biou"] == bi.mean_iou() and r["bf"] == bi.mean_bf_iou()
    avg = (res[0]["grad_local"] + res[1]["grad_local"]) / 2
    assert torch.allclose(res[0]["grad_avg"], avg, rtol=0, atol=1e-6)
    assert torch.equal(res[0]["grad_avg"], res[1]["grad_avg"])
    assert not torch.equal(res[0]["grad_local"], res[1]["grad_local"])
    # buffers: mean of the ranks' running statistics, identical everywhere; the step counter from rank 0
    for k in ("running_mean", "running_var"):
        mean = (res[0]["bn_local"][k] + res[1]["bn_local"][k]) / 2
        assert not torch.equal(res[0]["bn_local"][k], res[1]["bn_local"][k])
        assert torch.allclose(res[0]["bn_synced"][k], mean, rtol=0, atol=1e-6)
        assert torch.equal(res[0]["bn_synced"][k], res[1]["bn_synced"][k])
    assert int(res[1]["bn_synced"]["num_batches_tracked"]) == int(res[0]["bn_local"]["num_batches_tracked"])
    # dropout seeds: same torch generator state on both ranks, distinct seeds per rank and per step
    assert len(set(res[0]["seeds"]) | set(res[1]["seeds"])) == 6
    assert [s % WORLD for s in res[0]["seeds"]] == [0, 0, 0] and [s % WORLD for s in res[1]["seeds"]] == [1, 1, 1]
