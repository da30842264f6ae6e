#!/usr/bin/env python3
"""bench.py — FK+Jacobian evals/sec, Franka Panda 7-DoF, batch = 65 536 per GPU (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: a single `drm_fk_jacobian` launch through the
C ABI producing pos[B,3], quat[B,4], lin_jac[B,3,7], ang_jac[B,3,7] from q[B,7] (what the reference's
`compute_endeffector_jacobian` computes, robot_model.py:626-667), inputs resident in HBM.

Timing: W untimed steps, then EXACTLY K steps bracketed by barrier + torch.cuda.synchronize() on both
sides, max over ranks.  The K launches are captured once into a hipGraph (one launch per step, same
stream, no fusion / skipping) and replayed, so the host does not throttle a ~6 us kernel; `--no-graph`
launches them eagerly.  HIP events on the launch stream around the timed region give the average
duration of a launch for the roofline object.

Multi-GPU: the batch shards by rows with no data-path collective (every sample is independent), so
each rank runs its own 65 536-row shard ("scaling": "weak"); `--gather` adds the optional RCCL
all-gather of the outputs to the timed step.

The JSON line also carries
  "roofline":     algorithmic bytes (224 B/eval) / average launch duration vs the 8 TB/s HBM peak
  "cpu_baseline": the oracle's fp32 C restatement of the reference (oracle/, "port") timed on this
                  box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)
EE_LINK = {"panda_no_gripper": "panda_virtual_ee_link", "iiwa7": "iiwa_link_ee"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU per step")
    ap.add_argument("--robot", default="panda_no_gripper", choices=sorted(EE_LINK))
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--gather", action="store_true", help="all-gather the outputs over RCCL inside every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work budget of the cpu_baseline leg")
    return ap.parse_args()


def cpu_baseline(spec, link_idx, q_host, seconds):
    """Oracle (CPU port of the reference algorithm, fp32, OpenMP over samples) on a bounded sample."""
    import numpy as np

    from oracle import Oracle  # checker / baseline only — never on the product path
    orc = Oracle(spec)
    cores = Oracle.max_threads()
    q = np.ascontiguousarray(q_host, np.float32)
    orc.fk_jacobian(q[:1024], link_idx, np.float32)  # page in / spin up the thread pool
    reps, t0 = 0, time.perf_counter()
    while True:  # bounded by wall time, not by a pass count guessed from one (possibly cold) pass
        orc.fk_jacobian(q, link_idx, np.float32)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 100000:
            break
    evals = reps * q.shape[0]
    return {"value": evals / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d passes over the same %d-sample batch (%.1f s wall on all host cores), fp32 C restatement of the "
                      "reference algorithm (oracle/drm_oracle.c), OpenMP over samples" % (reps, q.shape[0], dt)}


def recorded_traffic(batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json), if this
    batch size was profiled; PMC counters cannot be collected from inside the timed run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)["per_batch"].get(str(batch))
            if rec:
                return rec["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU compute path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)  # "nccl" IS RCCL on ROCm
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import contextlib
    import io

    from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel, robot_description_folder
    with contextlib.redirect_stdout(io.StringIO()):
        model = DifferentiableRobotModel(os.path.join(robot_description_folder, args.robot + ".urdf"), device=device)
    link = EE_LINK[args.robot]
    n, B, K, W = model._n_dofs, args.batch, args.steps, args.warmup

    # synthetic inputs, resident in HBM: q ~ U(lower, upper) per joint (data_utils.py:49-67 distribution)
    lim = model.get_joint_limits()
    lo = torch.tensor([j["lower"] for j in lim], device=device)
    hi = torch.tensor([j["upper"] for j in lim], device=device)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    q = (lo + (hi - lo) * torch.rand(B, n, device=device, generator=gen)).contiguous()

    plan = model.plan_fk_and_jacobian(q, link)
    gathered = None
    if args.gather and world > 1:
        gathered = [torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=device) for t in plan.outputs()]

    def step():
        plan.launch()
        if gathered is not None:
            for out, loc in zip(gathered, plan.outputs()):
                dist.all_gather_into_tensor(out, loc)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(W):
        step()
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph and gathered is None:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(K):
                    plan.launch()
            graph.replay()  # one untimed replay (uploads the executable graph)
            torch.cuda.synchronize()
        except Exception as err:  # pragma: no cover - depends on the runtime
            if rank == 0:
                print("hipGraph capture failed (%s); launching eagerly" % err, file=sys.stderr)
            graph = None

    stream = torch.cuda.current_stream(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    if graph is not None:
        graph.replay()
    else:
        for _ in range(K):
            step()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()

    elapsed = torch.tensor([t1 - t0, ev0.elapsed_time(ev1) * 1e-3], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    wall, dev_time = float(elapsed[0]), float(elapsed[1])

    if rank == 0:
        bytes_per_eval = 4 * (n + 7 + 6 * n)        # q in; pos, quat, lin_jac, ang_jac out
        launch_s = dev_time / K                     # average duration of one launch, HIP events on the launch stream
        achieved = bytes_per_eval * B / launch_s / 1e9
        traffic, traffic_src = recorded_traffic(B) if args.robot == "panda_no_gripper" else (None, None)
        line = {
            "metric": "FK+Jacobian evals/sec, Panda 7-DoF, batch=65 536 @1/2/4/8 MI355X",
            "value": world * B * K / wall, "unit": "evals/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Franka Panda 7-DoF (panda_no_gripper), FK + end-effector geometric Jacobian to "
                                   "panda_virtual_ee_link, batch=%d per GPU, q~U(joint limits), inputs resident in HBM"
                                   % B if args.robot == "panda_no_gripper" else "%s FK+Jacobian batch=%d" % (args.robot, B),
                       "batch_per_gpu": B, "global_batch": world * B, "parallelism": "batch-sharded x%d" % world,
                       "launch": "hipGraph of K launches" if graph is not None else "eager launches",
                       "gather": bool(gathered is not None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_per_eval * B,
                         "kernel": "drm::fk_jacobian_arm_kernel<8, 7, true>", "bytes_per_eval": bytes_per_eval,
                         "launch_us": launch_s * 1e6,
                         "note": "algorithmic bytes / average launch duration (HIP events over the timed region, "
                                 "includes inter-launch gaps); a 65 536-sample launch is one wave per SIMD and "
                                 "latency-bound, see DESIGN.md §6 for B=2^20..2^22 (>=75% of peak)"},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is reported at N = 1 only
            line["cpu_baseline"] = cpu_baseline(model._spec, model._name_to_idx_map[link],
                                                q.cpu().numpy(), args.cpu_seconds)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
