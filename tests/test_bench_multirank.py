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
    """Run bench.py; check the stdout contract (exactly ONE compact, strict-JSON line from rank 0) and return the FULL record of the
    run (--detail), which carries the compact line's fields and everything they were cut from."""
    import tempfile
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):   # the spawn path, not an inherited rendezvous
        env.pop(k, None)
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail", detail] + list(flags), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, timeout=timeout, env=env, cwd=ROOT)
        out = done.stdout.decode()
        every = [l for l in out.splitlines() if l.strip()]
        lines = [l for l in every if l.startswith("{")]      # (gloo's C++ side prints "[Gloo] Rank r is connected ..." to stdout)
        assert done.returncode == 0, (done.returncode, out[-2000:], done.stderr.decode()[-4000:])
        assert len(lines) == 1 and every[-1] == lines[0], "exactly ONE JSON line, from rank 0, and it is the LAST line: %r" % every
        assert len(lines[0]) < 4096, len(lines[0])
        compact = json.loads(lines[0], parse_constant=lambda name: pytest.fail("non-finite %s in the bench line" % name))
        with open(detail) as f:
            full = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "ranks_seen"):
        assert key in compact, key
        if key not in ("config", "roofline"):
            assert compact[key] == full[key] or abs(compact[key] - full[key]) <= 1e-6 * abs(full[key]), key
    return full


def test_compact_line_stays_small_and_strict():
    """bench.compact_line: whatever the run collected, stdout carries < 4 KB of strict JSON with the contract fields."""
    import bench
    nan = float("nan")
    prose = "x" * 3000
    full = {"metric": "m", "value": 1.23456789012e10, "unit": "evals/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 4.6e-3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks_seen": 1,
            "config": {"workload": "w", "batch_per_gpu": 65536, "note": prose},
            "roofline": {"bound": "hbm", "achieved": 3500.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.4375, "traffic": nan,
                         "kernel": "k", "launch_us": 4.2, "note": prose, "steady_state": {"launch_us": 3.7, "frac": 0.49, "steps": 200}},
            "roofline_large": [{"batch": 1 << 22, "launch_us": 140.0, "frac": 0.84, "note": prose}, {"batch": 1 << 24, "skipped": "x"}],
            "cpu_baseline": {"value": 1.2e6, "unit": "evals/s", "cores": 1, "kind": "reference", "sample": prose,
                             "port": {"value": 3e7, "cores": 128, "sample": prose},
                             "reference": {"gpu_vs_reference_max_abs": {"pos": 2.4e-7, "quat_sign_flips": 0}, "blob": prose}},
            "configs": {"legs": [{"name": n, "own_kernel": True, "own_kernel_path": "default", "workload": prose,
                                  "roofline": {"frac": 0.5, "launch_us": float("inf")}} for n in
                                 ("config2", "config3_shard", "config3_whole", "config4", "config5")],
                        "learn_dynamics_step": {"workload": prose, "graph_step_us": 131.0, "graph_step_fused_adam_us": 44.2,
                                                "without_table_links_us": 101.0},
                        "api_eager_us_per_call": {"compute_forward_kinematics": {"us_per_call": 6.0}, "note": prose}}}
    line = bench.compact_line(full)
    text = json.dumps(line, allow_nan=False)          # raises on NaN / Infinity
    assert len(text) < 4096 and prose not in text
    assert line["roofline"]["traffic"] is None and line["configs_launch_us"]["c2"] is None
    assert line["configs_launch_us"]["dyn_graph_step_fused_adam"] == 44.2 and line["configs_launch_us"]["dyn_graph_step_fused_adam_no_abi13"] == 101.0
    assert line["configs_frac"] == {"c2": 0.5, "c3_shard": 0.5, "c3_whole": 0.5, "c4": 0.5, "c5": 0.5}
    assert line["configs_own_kernel"]["c4"] == "default" and line["roofline_large"] == {"4194304": {"launch_us": 140.0, "frac": 0.84}}
    assert line["cpu_baseline"]["sample"].endswith("...") and line["cpu_baseline"]["port_cores"] == 128
    assert line["api_eager_us_per_call"] == {"forward_kinematics": 6.0}


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
    assert set(modes) == {"none", "tau", "root", "all", "p2p"} and modes["all"]["value"] == pytest.approx(line["value"], rel=1e-9)
    # round 6: the one-sided gather (drm_fk_rnea_put + distributed.PeerGather: IPC-mapped peer buffers, the stores leave from the
    # fused kernel's epilogue) — verified bit for bit with the collectives above (gather_verified), one launch per step, and far
    # cheaper than a host-staged collective even with both ranks on one device
    assert modes["p2p"]["one_sided"] is True and modes["p2p"]["hipgraph"] is True
    assert line["one_sided_gather"]["in_kernel"] is bool(line["own_kernel"])      # (the arm's own fused kernel stores to the peers itself;
    assert line["own_kernel"] == "default" or os.environ.get("DRM_SPECIALIZE") == "0"   # under DRM_SPECIALIZE=0: compute, then copies)
    assert modes["p2p"]["step_us_device"] < 0.2 * modes["all"]["step_us_device"]
    assert modes["tau"]["gather_bytes_per_rank"] == rows * 28 and modes["none"]["gather_bytes_per_rank"] == 0
    assert modes["none"]["value"] >= modes["all"]["value"] and modes["all"]["gather_model_us"] > modes["tau"]["gather_model_us"] > 0


@pytest.mark.gpu
def test_config3_one_sided_gather_as_the_headline():
    line = run_bench("--gpus", "2", "--config", "3", "--gather", "p2p", "--steps", "5", "--warmup", "2", "--shared-gpu", "--verify-gather",
                     "--no-cpu-baseline")
    assert line["config"]["gather"] == "p2p" and line["gather_verified"] is True
    assert line["value"] == pytest.approx(line["gather_modes"]["p2p"]["value"], rel=1e-9) and line["config"]["gather_bytes_per_rank"] == (1 << 19) * 56


@pytest.mark.gpu
def test_config3_headline_is_the_sharded_step():
    line = run_bench("--gpus", "2", "--config", "3", "--steps", "5", "--warmup", "2", "--shared-gpu", "--no-cpu-baseline")
    assert line["config"]["gather"] == "none" and line["config"]["gather_bytes_per_rank"] == 0
    assert line["value"] == pytest.approx(line["gather_modes"]["none"]["value"], rel=1e-9) and line["gather_us_per_step"] < line["compute_us_per_step"]


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
