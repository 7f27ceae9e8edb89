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
    # the fraction is SURVEY.md 8d's compulsory bytes / the probed duration / the 8 TB/s peak, by hand
    rf = d["roofline"]
    assert abs(rf["frac"] - rf["bytes_per_launch"] / (rf["us_per_launch"] * 1e-6) / 1e9 / rf["peak"]) < 2e-3
    if rf["kernel"] == "masked_pool_fwd":
        assert rf["bytes_per_launch"] == 40 * 256 * 1681 * 4 + 40 * 1681 * 8          # B*C*P*4 + B*P*8
    assert d["value_rotating"] is None or d["value_rotating"] > 0
    assert d["dropin"]["head_step_us_per_step"] > 0 and d["dropin"]["standalone_us_per_step"] > 0
    # the variants the launchers chose for the benchmarked shape travel with the line
    assert d["kernel_plan"]["pool_launches"] in (1, 2) and "classify_layout" in d["kernel_plan"]
    assert abs(d["loss_k"] - 3.109) < 0.01                     # the synthetic C2 batch: ln(21) = 3.04 plus noise


def test_bench_gpus_flag_starts_two_ranks_and_runs_the_step():
    """`python bench.py --gpus 2` without torchrun (the driver's form): two rank processes are started, both run
    the timed step, the gradient bucket is reduced between them and rank 0's line says n_gpus = 2.  On a one-GPU
    box the ranks share the card, which RCCL refuses: the rehearsal goes over gloo (reduction through host
    memory); the 8-GPU RCCL run is the driver's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["DENET_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--settle", "0", "--no-cpu-baseline", "--no-dropin", "--no-rotating"],
                       capture_output=True, text=True, timeout=170, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["config"]["parallelism"].startswith("dp2")
    assert len(d["config"]["devices"]) == 2 and d["value"] > 0
    assert abs(d["loss_k"] - 3.109) < 0.01
