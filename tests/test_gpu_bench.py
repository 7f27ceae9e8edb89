"""bench.py end to end on one GPU, with the N > 1 code path switched on: a one-rank RCCL group, the
gradient-bucket all-reduce inside the step, the max-over-ranks timing, and a normal teardown
(destroy_process_group, exit code 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_step_with_rccl_allreduce_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                        "--collective", "--no-cpu-baseline"], capture_output=True, text=True, timeout=110, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["value"] > 0 and d["unit"] == "episodes/s"
    assert d["roofline"]["probed_steps"] >= 8                  # the driver's 20-step run still gets >= 8 samples
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] in ("hbm", "mfma")
    assert abs(d["loss_k"] - 3.109) < 0.01                     # the synthetic C2 batch: ln(21) = 3.04 plus noise
