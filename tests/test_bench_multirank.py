"""bench.py's N > 1 code path, executed for real through its own self-spawn — on a box with ONE GPU.

`--shared-gpu` (DRM_BENCH_SHARED_GPU=1) lets the ranks share the visible device; RCCL refuses two ranks on one device, so
the collectives go through gloo (device buffers staged through pinned host memory).  Everything else is the code the 8-GPU
driver run executes: respawn under torch.distributed.run on 127.0.0.1, rank / device set-up, per-rank shards, the timed
region with barrier + synchronize + max over ranks, the gather inside the step, the JSON line from rank 0.
`--verify-gather`: rank 0 rebuilds all ranks' inputs, runs them as ONE single-rank launch and compares the gathered
buffer with it bit for bit.

CPU (not gpu): argument handling only (the compute has no CPU path).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_bench(*flags, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):   # the spawn path, not an inherited rendezvous
        env.pop(k, None)
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout, env=env, cwd=ROOT)
    out = done.stdout.decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert done.returncode == 0, (done.returncode, out[-2000:], done.stderr.decode()[-4000:])
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0: %r" % lines
    return json.loads(lines[0])


def test_shared_gpu_flag_selects_gloo():
    import bench
    a = bench.parse_args(["--gpus", "2", "--shared-gpu"])
    assert a.backend == "gloo" and a.shared_gpu
    assert bench.parse_args(["--gpus", "2"]).backend == "nccl"


def test_gather_flag_per_config():
    import bench
    assert bench.parse_args([]).gather is False and bench.parse_args(["--gather"]).gather is True     # metric config: a switch
    assert bench.parse_args(["--gather", "none"]).gather is False
    assert bench.parse_args(["--config", "3"]).gather is None                                           # config 3: a mode
    assert bench.parse_args(["--config", "3", "--gather"]).gather == "all"
    assert bench.parse_args(["--config", "3", "--gather", "tau"]).gather == "tau"
    from differentiable_robot_model_amd.distributed import gather_model_us
    shard, tau = 131072 * 56, 131072 * 28
    assert gather_model_us("none", shard, tau, 8) == 0.0 and gather_model_us("all", shard, tau, 1) == 0.0
    assert abs(gather_model_us("all", shard, tau, 8) - 47.97) < 0.1 and abs(gather_model_us("tau", shard, tau, 8) - 23.99) < 0.1
    assert gather_model_us("root", shard, tau, 8) == gather_model_us("all", shard, tau, 8)


@pytest.mark.gpu
def test_config3_two_ranks_sharing_the_gpu_gather_equals_one_launch():
    line = run_bench("--gpus", "2", "--config", "3", "--gather", "all", "--steps", "5", "--warmup", "2", "--shared-gpu", "--verify-gather")
    rows = (1 << 20) // 2
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 5 and line["scaling"] == "strong"
    assert line["config"]["batch_per_gpu"] == rows and line["config"]["gather"] == "all"
    assert line["config"]["gather_bytes_per_rank"] == rows * 56
    assert line["gather_verified"] is True          # all-gather, torques-only all-gather and gather-to-root, bit for bit vs one launch
    assert line["distributed"]["backend"] == "gloo" and line["distributed"]["shared_gpu"] is True
    assert line["value"] > 0 and line["compute_us_per_step"] > 0 and line["step_us_with_gather"] >= line["compute_us_per_step"] * 0.5
    modes = line["gather_modes"]
    assert set(modes) == {"none", "tau", "root", "all"} and modes["all"]["value"] == line["value"]
    assert modes["tau"]["gather_bytes_per_rank"] == rows * 28 and modes["none"]["gather_bytes_per_rank"] == 0
    assert modes["none"]["value"] >= modes["all"]["value"] and modes["all"]["gather_model_us"] > modes["tau"]["gather_model_us"] > 0


@pytest.mark.gpu
def test_config3_headline_is_the_sharded_step():
    line = run_bench("--gpus", "2", "--config", "3", "--steps", "5", "--warmup", "2", "--shared-gpu", "--no-cpu-baseline")
    assert line["config"]["gather"] == "none" and line["config"]["gather_bytes_per_rank"] == 0
    assert line["value"] == line["gather_modes"]["none"]["value"] and line["gather_us_per_step"] < line["compute_us_per_step"]


@pytest.mark.gpu
def test_metric_two_ranks_sharing_the_gpu_with_gather():
    line = run_bench("--gpus", "2", "--gather", "--steps", "5", "--warmup", "2", "--shared-gpu", "--verify-gather",
                     "--no-large", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 2 * 65536 and line["config"]["gather"] is True
    assert line["config"]["gather_bytes_per_rank"] == 65536 * 196
    assert line["gather_verified"] is True and "cpu_baseline" not in line and "roofline_large" not in line


@pytest.mark.gpu
def test_metric_two_ranks_sharing_the_gpu_no_collective():
    """The driver's own form for N > 1 (weak scaling, no data-path collective, hipGraph replay next to an initialised
    process group): `--gpus 2 --steps 20 --warmup 5`."""
    line = run_bench("--gpus", "2", "--steps", "20", "--warmup", "5", "--shared-gpu")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 20 and line["warmup"] == 5
    assert line["config"]["launch"].startswith("hipGraph") and line["config"]["gather"] is False
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_refuses_more_gpus_than_the_node_has():
    import torch
    n = torch.cuda.device_count() + 1
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert done.returncode != 0 and b"refusing" in done.stderr
