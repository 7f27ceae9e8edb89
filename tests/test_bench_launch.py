"""CPU: the launch path of `bench.py --gpus N` (episodes/s at 1/2/4/8 GPUs) and the host-side
pieces of a resumed training run.  No kernel runs here: `--dry-run` stops after the ranks have joined."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e["DENET_BENCH_BACKEND"] = "gloo"
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True,
                          text=True, timeout=300)


def test_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (the driver's form, no torchrun): the parent starts two rank processes, they
    join one group, and rank 0's line says n_gpus = 2."""
    r = _bench("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"dry_run": True, "n_gpus": 2, "ranks_joined": 2, "backend": "gloo"}


def test_gpus_flag_must_match_the_world_size_under_torchrun():
    """Under torchrun the flag cannot silently disagree with the group: --gpus 2 in a world of 1 is an error."""
    r = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


def test_single_rank_dry_run():
    r = _bench("--gpus", "1", "--dry-run", env={"MASTER_PORT": "29577"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_scheduler_resume_keeps_the_epoch_schedule(tmp_path):
    """trainer/trainers.py:185-195,209: a checkpoint stores the scheduler at epoch e; a resumed run restores it
    and ticks with the epoch number, so epoch e (and every later one) gets the rate an uninterrupted run had.
    StepLR (closed form) and the reference's PolynomialLR."""
    from denet_amd.schedulers import get_scheduler, tick

    def make(cfg):
        p = [torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD(p, lr=0.1)
        return opt, get_scheduler(opt, dict(cfg))

    for cfg in ({"name": "StepLR", "step_size": 2, "gamma": 0.5},
                {"name": "PolynomialLR", "max_iter": 8, "decay_iter": 2, "gamma": 0.9}):
        opt, sch = make(cfg)
        full = []
        for e in range(1, 9):
            tick(sch, e)
            full.append(opt.param_groups[0]["lr"])
            if e == 4:
                torch.save({"scheduler": sch.state_dict(), "epoch": e}, tmp_path / "ck.bin")
        ck = torch.load(tmp_path / "ck.bin", weights_only=True)
        opt2, sch2 = make(cfg)                       # fresh optimizer: the reference does not restore it either
        sch2.load_state_dict(ck["scheduler"])
        resumed = []
        for e in range(ck["epoch"], 9):              # train() restarts at start_epoch = the saved epoch
            tick(sch2, e)
            resumed.append(opt2.param_groups[0]["lr"])
        assert resumed == full[ck["epoch"] - 1:], (cfg["name"], resumed, full)
