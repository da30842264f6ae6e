#!/usr/bin/env python3
"""bench.py — FK+Jacobian evals/sec, Franka Panda 7-DoF, batch = 65 536 per GPU (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

--config metric (default)
    One "step" = one pass of the hot path over one batch: a single `drm_fk_jacobian` launch through the C ABI
    producing pos[B,3], quat[B,4], lin_jac[B,3,7], ang_jac[B,3,7] from q[B,7] (what the reference's
    `compute_endeffector_jacobian` computes, robot_model.py:626-667), inputs resident in HBM.  Weak scaling: every
    rank runs its own 65 536-row shard, no data-path collective (every sample is independent).
--config 3
    BASELINE.json configuration 3: Panda, GLOBAL batch 2^20 sharded by rows over the ranks (131 072 per GPU at N = 8),
    one step = FK(end effector) + RNEA inverse dynamics through `drm_fk_rnea` (one fused launch) + the exchange `--gather`
    names: none (default headline: the outputs stay sharded — the MPC / particle case, strong scaling of the compute), tau /
    all (`all_gather_into_tensor` of the torques / of tau + pos + quat, 28 / 56 B per row), root (gather to rank 0).  Every
    mode is timed in the same run (`gather_modes`) with the xGMI estimate beside it (`gather_model_us`): at N = 8 the shard's
    7.3 MB take ~48 us over one 153 GB/s link against ~5.5 us of compute, so `all` is predicted SLOWER than one GPU doing the
    whole batch (~30 us) — the number to quote for a sharded consumer is `none`.  The compute runs the robot's OWN kernel
    (model.specialize(): Panda's constants folded into the instruction stream, hipcc at run time; `--library-kernels` for the
    library's table-driven kernel: 7.1 / 42.6 us).

Timing: W untimed steps, then EXACTLY K steps bracketed by barrier + torch.cuda.synchronize() on both sides, max
over ranks.  For the metric config the K launches are captured once into a hipGraph (one launch per step, same
stream, no fusion / skipping) and replayed, so the host does not throttle a ~4 us kernel; the captured graph is
replayed untimed first — twice to upload the executable graph, then for ~50 ms so that the clocks have left their idle
state before a timed region of ~80 us (a cold box measured 14.6 us per step at --steps 20 without it, 5.1 us with) —
and `--no-graph` launches eagerly.
HIP events on the launch stream around the timed region give the average duration of a launch for the roofline
object.

The JSON line also carries
  "roofline":       algorithmic bytes (224 B/eval) / average launch duration vs the 8 TB/s HBM peak at the
                    metric batch (which is Infinity-Cache resident: 14.7 MB per launch replayed over the same buffers);
                    with fewer than 200 steps a "steady_state" entry adds the same launch measured over a region of 200
                    (a 20-step region carries one graph-launch latency, ~9 us, in its average)
  "roofline_large": the same kernel at batches whose per-launch traffic (0.94 GB / 3.8 GB) is far beyond the 256 MiB
                    Infinity Cache, i.e. genuine HBM streaming (N = 1 only)
  "cpu_baseline":   value = the UNMODIFIED reference itself (oracle/_ref/, staged from /root/reference by oracle/stage_ref.py;
                    kind "reference") timed on this box's host cores in this run on the same joint states: tensor-only
                    (its Python quaternion loop stubbed) on all 65 536 rows, one thread, min of 3.  Under "reference": the
                    same with all threads, its public compute_endeffector_jacobian on 4 096 rows (linear extrapolation
                    stated) and the HIP path's deviation from the reference's own outputs; under "port": the oracle's fp32 C
                    restatement of the algorithm (oracle/, OpenMP over samples, all host cores) on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this stack needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid argument` otherwise);
# the images export it already — keep it for every rank this script spawns
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s achievable)
EE_LINK = {"panda_no_gripper": "panda_virtual_ee_link", "iiwa7": "iiwa_link_ee"}
CONFIG3_GLOBAL_BATCH = 1 << 20


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="metric", choices=["metric", "3"])
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU per step (metric config)")
    ap.add_argument("--robot", default="panda_no_gripper", choices=sorted(EE_LINK))
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--gather", nargs="?", const="all", default=None, choices=["none", "all", "root", "tau", "p2p"],
                    help="what a step exchanges over RCCL.  metric config: `--gather` = all-gather the outputs inside every step "
                         "(default: nothing).  --config 3: which mode the headline `value` is timed with — none (default: the "
                         "outputs stay sharded, the MPC / particle case), all (every rank gets tau | pos | quat of every row), root "
                         "(rank 0 only), tau (all-gather of the torques only), p2p (as all, ONE-SIDED: the fused kernel's epilogue writes "
                         "every tile into the peers' gathered arrays over xGMI while the walk runs — drm_fk_rnea_put, "
                         "distributed.PeerGather; no collective launch, no second pass); every mode is timed and reported under gather_modes")
    ap.add_argument("--library-kernels", action="store_true",
                    help="--config 3: time the library's table-driven kernels instead of the robot's own constant-folded ones "
                         "(model.specialize(), the default when hipcc is on the machine)")
    ap.add_argument("--no-large", action="store_true", help="skip the roofline_large legs (2^22, 2^24 samples)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--configs", default="fast", choices=["off", "fast", "full"],
                    help="the `configs` legs (BASELINE.json configurations 2-5 + eager API overhead; N = 1 only).  fast (default): the "
                         "GPU launch times and roofline fractions only (~10 s).  full: also the unmodified reference timed beside every "
                         "leg and two rocprofv3 --pmc passes per leg for its HBM traffic (~70 s; what tools/profile_round.sh records "
                         "under profiles/).  off: none")
    ap.add_argument("--no-configs", dest="configs", action="store_const", const="off", help="same as --configs off")
    ap.add_argument("--detail", default=os.path.join(ROOT, "profiles", "bench_last.json"),
                    help="where rank 0 writes the FULL record of the run (every leg, note and reference timing); stdout carries "
                         "ONE compact JSON line (< 4 KB) with the contract fields only")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure roofline.traffic (N = 1 only; ~30 s)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work budget of the cpu_baseline leg")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 path: nccl (= RCCL over xGMI, the product path) or gloo "
                         "(collectives staged through the host; what lets N ranks share ONE GPU, see --shared-gpu)")
    ap.add_argument("--shared-gpu", action="store_true", default=os.environ.get("DRM_BENCH_SHARED_GPU") == "1",
                    help="TEST MODE (also DRM_BENCH_SHARED_GPU=1): the N ranks share the visible GPU(s) round-robin instead of "
                         "owning one each, so that every line of the multi-rank path runs on a 1-GPU box.  Implies "
                         "--backend gloo (RCCL refuses two ranks on one device); the JSON line says so and is not a scaling number")
    ap.add_argument("--verify-gather", action="store_true",
                    help="after the timed region, rank 0 recomputes ALL ranks' rows in one single-rank launch and compares the "
                         "gathered buffer with it bit for bit (gather_verified in the JSON line)")
    args = ap.parse_args(argv)
    if args.shared_gpu:
        args.backend = "gloo"
    if args.config != "3":
        args.gather = args.gather not in (None, "none")     # the metric config: a switch
    return args


def respawn_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: become N ranks (one per GPU) by
    re-executing under torch.distributed.run.  Fails loudly when the node has fewer than N GPUs."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not (args.shared_gpu and have >= 1):
        sys.exit("bench.py: --gpus %d requested but this node exposes %d HIP device(s); refusing to report a "
                 "%d-GPU number from fewer GPUs" % (args.gpus, have, args.gpus))
    from differentiable_robot_model_amd.distributed import torchrun_command
    cmd = torchrun_command(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    sys.exit(subprocess.call(cmd))


def reference_cpu(robot, link, q_host, qd_host=None, qdd_host=None, gpu_outputs=None, public_rows=4096):
    """The UNMODIFIED reference, timed on THIS box's host cores in THIS run: oracle/ref_timing.py runs in its own
    interpreter (the bench process never imports the reference) on the SAME joint states the GPU leg used.  The reference
    is pure Python; oracle/stage_ref.py stages it from /root/reference into the git-ignored oracle/_ref/, which ships to
    the GPU box with the snapshot.  `gpu_outputs` (dict of host arrays of the HIP path on the same rows): the largest
    deviation from the reference's own outputs is reported next to the timings."""
    import tempfile

    import numpy as np
    script = os.path.join(ROOT, "oracle", "ref_timing.py")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, script, "--robot", robot, "--link", link, "--public-rows", str(public_rows),
               "--q", os.path.join(tmp, "q.npy"), "--out-npz", os.path.join(tmp, "ref.npz")]
        np.save(os.path.join(tmp, "q.npy"), np.ascontiguousarray(q_host, np.float32))
        if qd_host is not None and qdd_host is not None:
            np.save(os.path.join(tmp, "qd.npy"), np.ascontiguousarray(qd_host, np.float32))
            np.save(os.path.join(tmp, "qdd.npy"), np.ascontiguousarray(qdd_host, np.float32))
            cmd += ["--qd", os.path.join(tmp, "qd.npy"), "--qdd", os.path.join(tmp, "qdd.npy")]
        try:
            env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")   # host cores only
            done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
            rec = json.loads(done.stdout.decode().strip().splitlines()[-1])
        except Exception as err:  # the baseline leg must never take the GPU numbers down with it
            return {"kind": "reference", "error": "%s: %s" % (type(err).__name__, err)}
        if gpu_outputs and "error" not in rec and os.path.exists(os.path.join(tmp, "ref.npz")):
            ref = np.load(os.path.join(tmp, "ref.npz"))
            rows, dev = rec.get("outputs_npz_rows", 0), {}
            for key, got in gpu_outputs.items():
                if key in ref.files:
                    want, got = ref[key], np.asarray(got)[:rows]
                    if key == "quat":   # sign of the whole quaternion: the reference's own branch rule decides it
                        dev["quat_sign_flips"] = int((np.sum(want * got, -1) < 0).sum())
                    scale = 1.0 + np.abs(want) if key == "tau" else 1.0
                    dev[key] = float((np.abs(got - want) / scale).max())
            rec["gpu_vs_reference_max_abs"] = dev
            rec["gpu_vs_reference_note"] = ("HIP path vs the reference's own fp32 outputs on the first %d rows of this "
                                            "run's batch (tau relative to 1 + |tau|)" % rows)
    return rec


def cpu_baseline(spec, link_idx, q_host, seconds):
    """Oracle (CPU port of the reference algorithm, fp32, OpenMP over samples) on a bounded sample: median pass."""
    import numpy as np

    from oracle import Oracle  # checker / baseline only — never on the product path
    orc = Oracle(spec)
    cores = Oracle.max_threads()
    q = np.ascontiguousarray(q_host, np.float32)
    for _ in range(3):
        orc.fk_jacobian(q, link_idx, np.float32)  # page in / spin up the thread pool
    times, t_start = [], time.perf_counter()
    while True:  # bounded by wall time, not by a pass count guessed from one (possibly cold) pass
        t0 = time.perf_counter()
        orc.fk_jacobian(q, link_idx, np.float32)
        t1 = time.perf_counter()
        times.append(t1 - t0)
        if t1 - t_start >= seconds or len(times) >= 100000:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": q.shape[0] / med, "unit": "evals/s", "cores": cores, "kind": "port",
            "best": q.shape[0] / times[0], "mean": q.shape[0] * len(times) / sum(times),
            "sample": "%d passes over the same %d-sample batch (%.1f s wall on all host cores; value = the median pass, "
                      "the host is shared so single passes swing), fp32 C restatement of the reference algorithm "
                      "(oracle/drm_oracle.c), OpenMP over samples" % (len(times), q.shape[0], sum(times))}


def recorded_traffic(batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json, newest round first),
    if this batch size was profiled; PMC counters cannot be collected from inside the timed run."""
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


def measured_traffic(args, batch):
    """HBM bytes per launch of the metric kernel, MEASURED now: two child runs of this same script (same robot and batch, no CPU
    legs) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes, kernel tracing only,
    as /opt/skills/guides/MI355X_MICROARCH.md prescribes), averaged over the dispatches of the kernel: FETCH_SIZE (KiB, x2: the
    gfx950 correction) + WRITE_SIZE (KiB).  None (with the reason) when rocprofv3 is missing or a pass fails — the caller then
    falls back to the recorded figure under profiles/."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    got = {}
    env = dict(os.environ, TMPDIR="/tmp", DRM_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="drm_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
               os.path.abspath(__file__), "--no-cpu-baseline", "--no-large", "--no-traffic", "--configs", "off", "--detail", os.devnull, "--steps", "50", "--warmup", "5",
               "--batch", str(batch), "--robot", args.robot]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90, check=True)
            vals = []
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row.get("Counter_Name") == counter and "fk_jacobian" in row.get("Kernel_Name", ""):
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "no %s rows for the metric kernel in the rocprofv3 output" % counter
            got[counter] = (sum(vals) / len(vals), len(vals))
        except (subprocess.SubprocessError, OSError, ValueError, KeyError) as e:
            return None, "%s pass failed: %s" % (counter, type(e).__name__)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fetch, write = got["FETCH_SIZE"][0] * 1024.0 * 2.0, got["WRITE_SIZE"][0] * 1024.0
    return {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
            "dispatches": min(got["FETCH_SIZE"][1], got["WRITE_SIZE"][1])}, None


def sample_q(model, B, device, seed):
    """q ~ U(lower, upper) per joint (data_utils.py:49-67 distribution), resident in HBM."""
    import torch
    lim = model.get_joint_limits()
    lo = torch.tensor([j["lower"] for j in lim], device=device)
    hi = torch.tensor([j["upper"] for j in lim], device=device)
    gen = torch.Generator(device=device).manual_seed(seed)
    return (lo + (hi - lo) * torch.rand(B, len(lim), device=device, generator=gen)).contiguous(), gen


def timed_graph_region(launch, K, stream, barrier, use_graph=True, warm_replays=2):
    """EXACTLY K launches between barrier + synchronize on both sides, twice: once timed by the host's clock (wall seconds),
    once more with HIP events on the launch stream around them (device seconds); returns (wall, device, whether a hipGraph was
    used)."""
    import torch
    graph = None
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures (multi-rank runs)
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(K):
                    launch()
            for _ in range(warm_replays):   # untimed: uploads the executable graph
                graph.replay()
            torch.cuda.synchronize()
            # untimed: ~50 ms of the same replays so that the clocks have left their idle state before a timed region that is
            # ~80 us long at the driver's --steps 20 (measured on a box that had idled through the imports: 14.6 us per step
            # without this, 5.1 us with the GPU warm).  Local to the rank: the graph holds no collective.
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.05:
                for _ in range(10):
                    graph.replay()
                torch.cuda.synchronize()
        except Exception as err:  # pragma: no cover - depends on the runtime
            print("hipGraph capture failed (%s); launching eagerly" % err, file=sys.stderr)
            graph = None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream); ev1.record(stream)   # untimed: an event's first record creates it (~10 us on this stack)

    def k_steps():
        if graph is not None:
            graph.replay()
        else:
            for _ in range(K):
                launch()

    # (1) the wall-clock bracket: nothing but the K steps between the two synchronizes — recording the two events from the
    # host inside it costs ~6 us of an ~95 us region at the driver's --steps 20 (tools/probe_region.py --events: 4.95 -> 4.65 us per step)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_steps()
    torch.cuda.synchronize()
    t1 = time.perf_counter()    # this rank's K steps are done; the caller takes the MAX over ranks
    barrier()                   # (closing barrier of the bracket: its own latency, ~50 us over 8 GPUs, is not a step)
    # (2) the same K steps once more, same bracket, with HIP events on the launch stream around them: the device-side duration
    # (the roofline's launch time).  A graph launch carries ~9 us of its own on the device (4.2 us per step at --steps 20
    # against 3.8 at 200; submitting the timed pass behind two untimed ones, so that the device never idles, does not change
    # it — measured).  Events recorded as NODES of the graph do not bracket the kernels on this stack (0.2 us per step).
    barrier()
    torch.cuda.synchronize()
    ev0.record(stream)
    k_steps()
    ev1.record(stream)
    while not ev1.query():      # poll instead of sleeping in the driver
        pass
    torch.cuda.synchronize()
    barrier()
    return t1 - t0, ev0.elapsed_time(ev1) * 1e-3, graph is not None


COMPACT_LIMIT = 4096      # bytes of the stdout line (round 5's 21 KB line was not readable by the round driver)


def _num(x, digits=6):
    """Floats of the compact line: `digits` significant digits; non-finite values become null (strict JSON)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, x))


def compact_line(full):
    """The ONE line stdout carries: the driver's contract fields, `roofline` and `cpu_baseline` of the dominant kernel, and one
    flat number per BASELINE configuration — no prose, no nesting beyond one level, < COMPACT_LIMIT bytes.  Everything else of the
    run (`full`) goes to --detail (profiles/bench_last.json) and to stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "ranks_seen")
    out = {k: _num(full.get(k), 9) for k in keep}
    cfg = full.get("config", {})
    out["config"] = {k: cfg.get(k) for k in ("workload", "batch_per_gpu", "global_batch", "parallelism", "launch", "gather",
                                             "gather_bytes_per_rank") if k in cfg}
    roof = full.get("roofline", {})
    out["roofline"] = {k: _num(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic",
                                                      "algorithmic_bytes_per_launch", "traffic_over_algorithmic", "kernel",
                                                      "launch_us") if k in roof}
    out["roofline"]["traffic_measured_in_this_run"] = bool(roof.get("traffic_measured_in_this_run", False))
    if "steady_state" in roof:          # the same launch in a region of 200 (a 20-step region carries one graph-launch latency)
        out["roofline"]["steady_state_launch_us"] = _num(roof["steady_state"]["launch_us"])
        out["roofline"]["steady_state_frac"] = _num(roof["steady_state"]["frac"])
    large = [r for r in full.get("roofline_large", []) if "frac" in r]
    if large:                           # genuine HBM streaming: per-launch traffic far beyond the Infinity Cache
        out["roofline_large"] = {str(r["batch"]): {"launch_us": _num(r["launch_us"]), "frac": _num(r["frac"])} for r in large}
    cpu = full.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {k: _num(cpu.get(k)) for k in ("value", "unit", "cores", "kind", "sample", "gpu_over_cpu") if k in cpu}
        if len(out["cpu_baseline"].get("sample") or "") > 300:
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:297] + "..."
        if isinstance(cpu.get("port"), dict):
            out["cpu_baseline"]["port_value"] = _num(cpu["port"].get("value"))
            out["cpu_baseline"]["port_cores"] = cpu["port"].get("cores")
        ref = cpu.get("reference") if isinstance(cpu.get("reference"), dict) else {}
        dev = ref.get("gpu_vs_reference_max_abs") or full.get("gpu_vs_reference_max_abs")
        if dev:
            out["gpu_vs_reference_max_abs"] = {k: _num(v, 3) for k, v in dev.items()}
        if "error" in ref:
            out["cpu_baseline"]["reference_error"] = str(ref["error"])[:160]
    configs = full.get("configs")
    if isinstance(configs, dict) and "legs" in configs:
        short = {"config2": "c2", "config3_shard": "c3_shard", "config3_whole": "c3_whole", "config4": "c4", "config5": "c5"}
        frac, us, own = {}, {}, {}
        for leg in configs["legs"]:
            key = short.get(leg["name"], leg["name"])
            frac[key], us[key] = _num(leg["roofline"]["frac"], 4), _num(leg["roofline"]["launch_us"], 4)
            if "own_kernel" in leg:
                own[key] = leg.get("own_kernel_path", bool(leg["own_kernel"]))
            if "fk_mse_roofline" in leg:
                frac[key + "_fk_mse"] = _num(leg["fk_mse_roofline"]["frac"], 4)
                us[key + "_fk_mse"] = _num(leg["fk_mse_roofline"]["launch_us"], 4)
            for k_src, k_dst in (("graph_step_us", "_graph_step"), ("graph_step_fused_us", "_graph_step_fused"), ("eager_step_us", "_eager_step")):
                if k_src in leg:
                    us[key + k_dst] = _num(leg[k_src], 4)
        dyn = configs.get("learn_dynamics_step") or {}
        for k_src, k_dst in (("graph_step_us", "dyn_graph_step"), ("graph_step_fused_adam_us", "dyn_graph_step_fused_adam"),
                             ("without_table_links_us", "dyn_graph_step_fused_adam_no_abi13")):
            if k_src in dyn:
                us[k_dst] = _num(dyn[k_src], 4)
        out["configs_frac"], out["configs_launch_us"], out["configs_own_kernel"] = frac, us, own
        eager = configs.get("api_eager_us_per_call") or {}
        out["api_eager_us_per_call"] = {k.replace("compute_", ""): _num(v["us_per_call"], 4) for k, v in eager.items()
                                        if isinstance(v, dict) and "us_per_call" in v}
    elif isinstance(configs, dict) and "error" in configs:
        out["configs_error"] = str(configs["error"])[:200]
    for k in ("own_kernel", "compute_us_per_step", "step_us_with_gather", "gather_us_per_step", "gather_verified"):   # --config 3 / --gather
        if full.get(k) is not None:
            out[k] = _num(full[k])
    if "gather_modes" in full:
        out["gather_step_us"] = {m: _num(v["step_us_device"], 4) for m, v in full["gather_modes"].items()}
        out["gather_model_us"] = {m: _num(v["gather_model_us"], 4) for m, v in full["gather_modes"].items()}
    dist_ = full.get("distributed") or {}
    if dist_.get("backend"):
        out["distributed"] = {k: dist_[k] for k in ("backend", "shared_gpu", "devices_used") if k in dist_}
    return out


def emit(full, args):
    """Rank 0: the full record to --detail and stderr, the compact line — the only thing on stdout — last."""
    compact = compact_line(full)
    if args.detail and args.detail != os.devnull:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
            with open(args.detail, "w") as f:
                json.dump(full, f, indent=1)
            compact["detail"] = os.path.relpath(args.detail, ROOT) if args.detail.startswith(ROOT) else args.detail
        except OSError as err:       # (a read-only checkout: the detail still goes to stderr)
            compact["detail"] = "stderr (%s)" % type(err).__name__
    print("bench.py detail: " + json.dumps(full), file=sys.stderr)
    text = json.dumps(compact, allow_nan=False)
    if len(text) > COMPACT_LIMIT:      # never let the line outgrow what the driver reads: drop the optional groups, largest first
        for key in ("configs_launch_us", "api_eager_us_per_call", "gather_model_us", "roofline_large", "configs_own_kernel",
                    "gpu_vs_reference_max_abs"):
            compact.pop(key, None)
            text = json.dumps(compact, allow_nan=False)
            if len(text) <= COMPACT_LIMIT:
                break
    sys.stderr.flush()
    print(text, flush=True)


def main():
    args = parse_args()
    respawn_if_needed(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with --nproc-per-node equal to --gpus" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU compute path)")
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and not args.shared_gpu:
        sys.exit("bench.py: rank %d has no GPU (node exposes %d)" % (local_rank, n_dev))
    dev_index = local_rank % n_dev if args.shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)  # "nccl" IS RCCL on ROCm
            probe = torch.ones(1, device=device)
        else:
            dist.init_process_group(backend="gloo")
            probe = torch.ones(1)
        dist.all_reduce(probe)                                     # every rank answers before anything is timed
        ranks_seen = int(probe.item())
        if ranks_seen != world or dist.get_world_size() != world:
            sys.exit("bench.py: %s sees %d ranks, expected %d" % (args.backend, ranks_seen, world))

    import contextlib
    import io

    from differentiable_robot_model_amd.distributed import shard_bounds
    from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel, robot_description_folder
    with contextlib.redirect_stdout(io.StringIO()):
        model = DifferentiableRobotModel(os.path.join(robot_description_folder, args.robot + ".urdf"), device=device)
    link = EE_LINK[args.robot]
    n, K, W = model._n_dofs, args.steps, args.warmup
    stream = torch.cuda.current_stream(device)

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(vals):
        t = torch.tensor(vals, device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    if args.config == "3":
        line = run_config3(args, model, link, device, world, rank, ranks_seen, stream, barrier, reduce_max, shard_bounds)
    else:
        line = run_metric(args, model, link, device, world, rank, ranks_seen, stream, barrier, reduce_max)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:          # LAST: nothing of this process follows the compact line on stdout (the process group is already down)
        emit(line, args)


def run_metric(args, model, link, device, world, rank, ranks_seen, stream, barrier, reduce_max):
    import torch

    from differentiable_robot_model_amd.distributed import all_gather_flat
    n, B, K, W = model._n_dofs, args.batch, args.steps, args.warmup
    q, _ = sample_q(model, B, device, 1234 + rank)
    plan = model.plan_fk_and_jacobian(q, link)
    gathered = None
    if args.gather and world > 1:
        gathered = [torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=device) for t in plan.outputs()]

    def step():
        plan.launch()
        if gathered is not None:
            for out, loc in zip(gathered, plan.outputs()):
                all_gather_flat(out, loc)

    for _ in range(W):
        step()
    torch.cuda.synchronize()
    wall, dev_time, graphed = timed_graph_region(step, K, stream, barrier, use_graph=not args.no_graph and gathered is None)
    wall, dev_time = reduce_max([wall, dev_time])

    verified, vs_whole = None, None
    if args.verify_gather and gathered is not None:
        step()
        torch.cuda.synchronize()
        ok = 1
        if rank == 0:
            # every rank's q again (same seeds).  The gather is held BIT FOR BIT to launches of the shard size (what the ranks
            # ran: the C ABI picks a kernel form by launch size, and two forms may differ in the last bit of a row), and the
            # same rows as ONE single-rank launch over all of them are reported beside it (bit-equal or not, max |difference|).
            qs = [sample_q(model, B, device, 1234 + r)[0] for r in range(world)]
            for r, qr in enumerate(qs):
                shard = model.plan_fk_and_jacobian(qr, link)
                shard.launch()
                torch.cuda.synchronize()
                for got, want in zip(gathered, shard.outputs()):
                    ok &= int(torch.equal(got[r * B:(r + 1) * B], want))
            whole = model.plan_fk_and_jacobian(torch.cat(qs), link)
            whole.launch()
            torch.cuda.synchronize()
            vs_whole = {"bit_equal": all(torch.equal(g, w_) for g, w_ in zip(gathered, whole.outputs())),
                        "max_abs": max(float((g - w_).abs().max()) for g, w_ in zip(gathered, whole.outputs()))}
        verified = bool(ok)
    bytes_per_eval = 4 * (n + 7 + 6 * n)            # q in; pos, quat, lin_jac, ang_jac out
    launch_s = dev_time / K                         # average duration of one launch, HIP events on the launch stream
    achieved = bytes_per_eval * B / launch_s / 1e9
    traffic, traffic_src = recorded_traffic(B) if args.robot == "panda_no_gripper" else (None, None)
    measured, why_not = (None, "not requested")
    if rank == 0 and world == 1 and not args.no_traffic and os.environ.get("DRM_BENCH_CHILD") != "1":
        measured, why_not = measured_traffic(args, B)     # (after the timed region: the child runs own the GPU meanwhile)
    line = {
        "metric": "FK+Jacobian evals/sec, Panda 7-DoF, batch=65 536 @1/2/4/8 MI355X",
        "value": world * B * K / wall, "unit": "evals/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen, "gather_verified": verified,
        "gather_vs_whole_launch": vs_whole, "distributed": test_mode_fields(args, world),
        "config": {"workload": "Franka Panda 7-DoF (panda_no_gripper), FK + end-effector geometric Jacobian to "
                               "panda_virtual_ee_link, batch=%d per GPU, q~U(joint limits), inputs resident in HBM"
                               % B if args.robot == "panda_no_gripper" else "%s FK+Jacobian batch=%d" % (args.robot, B),
                   "batch_per_gpu": B, "global_batch": world * B, "parallelism": "batch-sharded x%d" % world,
                   "launch": "hipGraph of K launches (replayed twice untimed first)" if graphed else "eager launches",
                   "gather": bool(gathered is not None),
                   "gather_bytes_per_rank": B * 4 * (7 + 6 * n) if gathered is not None else 0},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured["bytes"] if measured else traffic,
                     "traffic_unit": "bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                     "traffic_source": "two rocprofv3 --kernel-trace --pmc child passes of this command (FETCH_SIZE, WRITE_SIZE), "
                                       "run right after the timed region" if measured else traffic_src,
                     "traffic_measured_in_this_run": bool(measured),
                     "traffic_detail": measured if measured else {"fallback_reason": why_not},
                     "traffic_note": ("MEASURED in this run (separate counter passes, kernel tracing only)" if measured else
                                      "RECORDED figure: rocprofv3 --pmc passes of this same command, committed under profiles/"),
                     "traffic_over_algorithmic": (measured["bytes"] if measured else traffic) / (bytes_per_eval * B)
                     if (measured or traffic) else None,
                     "algorithmic_bytes_per_launch": bytes_per_eval * B,
                     "kernel": "drm::fk_jacobian_arm_kernel<8, 7, true, 1, false>", "bytes_per_eval": bytes_per_eval,
                     "launch_us": launch_s * 1e6,
                     "note": "algorithmic bytes / average launch duration: HIP events on the launch stream around a second pass of "
                             "the same K steps in the same bracket (the wall-clock pass, `ms_per_step`, carries no event calls); "
                             "includes inter-launch gaps and the ~9 us a graph launch costs on the device, i.e. 0.45 us per step "
                             "at --steps 20.  At this batch the 14.7 MB of a launch are replayed over the same buffers "
                             "and stay in the 256 MiB Infinity Cache: `achieved` is an effective rate against the HBM peak, "
                             "bounded by launch floor (1.66 us) + write-through drain of 12.8 MB (a launch that only moves "
                             "the bytes takes 2.9 us = 0.63, profiles/r02_metric_lab.txt); genuine HBM streaming is in "
                             "roofline_large"},
    }
    if graphed and gathered is None:
        line["fixed_overhead_us"] = {
            "value": 9.0, "note": "one hipGraph launch carries ~9 us between its start and its first kernel on this stack; a region "
            "of K steps pays it once: ms_per_step = kernel + gaps + 9 us / K (0.45 us per step at the driver's --steps 20, 0.045 "
            "at 200); `roofline.steady_state` is the same launch in a region of 200"}
    if graphed and K < 200 and gathered is None:
        # a short timed region (the round driver passes --steps 20: ~80 us of kernels) carries the latency of one graph
        # launch, ~9 us between the start event and the first kernel, in its average; the same launches in a region of 200
        # (its own graph, its own events; not part of the timed region above) give the kernel's steady-state duration
        _, dev200, _ = timed_graph_region(step, 200, stream, barrier, use_graph=True)
        (dev200,) = reduce_max([dev200])
        line["roofline"]["steady_state"] = {
            "steps": 200, "launch_us": dev200 / 200 * 1e6, "achieved": bytes_per_eval * B / (dev200 / 200) / 1e9,
            "frac": bytes_per_eval * B / (dev200 / 200) / 1e9 / HBM_PEAK_GBS,
            "note": "same launch, a separate region of 200 (HIP events): what `launch_us` converges to as --steps grows"}
    if rank == 0 and world == 1 and not args.no_large and args.robot == "panda_no_gripper":
        line["roofline_large"] = roofline_large(model, link, device, stream, bytes_per_eval)
    if not args.no_cpu_baseline and world == 1:   # the CPU leg is reported at N = 1 only
        line["cpu_baseline"] = cpu_baseline(model._spec, model._name_to_idx_map[link], q.cpu().numpy(), args.cpu_seconds)
        plan.launch()
        torch.cuda.synchronize()
        names = ("pos", "quat", "lin_jac", "ang_jac")
        ref = reference_cpu(
            args.robot, link, q.cpu().numpy(), gpu_outputs={k: t.cpu().numpy() for k, t in zip(names, plan.outputs())})
        if "one_thread" in ref:
            # THE CPU baseline (round 5: the top-level value): the UNMODIFIED reference's vectorised math on ONE thread (its
            # many-thread runs are slower on [B, 3]-sized ops).  The OpenMP port of the algorithm (oracle/, all host cores), which
            # swings 10x between passes on a shared host, is reported beside it under `port`.
            one = ref["one_thread"]["tensor_only"]
            port = line["cpu_baseline"]
            line["cpu_baseline"] = {
                "value": one["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "reference",
                "sample": "compute_endeffector_jacobian of the unmodified reference (oracle/_ref, staged from /root/reference) on all "
                          "%d rows of this run's batch, get_quaternion stubbed (tensor-only: its per-row Python loop is timed "
                          "separately under reference.public_api), one thread, min of %d" % (one["rows"], ref.get("reps", 3)),
                "gpu_over_cpu": line["value"] / one["evals_per_s"], "port": port, "reference": ref}
        else:      # (the reference leg failed: the port stays the value, the error is reported)
            line["cpu_baseline"]["reference"] = ref
    if rank == 0 and world == 1 and args.configs != "off" and os.environ.get("DRM_BENCH_CHILD") != "1":
        del plan
        torch.cuda.empty_cache()
        from bench_configs import run_config_legs
        full = args.configs == "full"
        try:
            line["configs"] = run_config_legs(device, with_reference=full and not args.no_cpu_baseline,
                                              with_traffic=full and not args.no_traffic)
            line["configs"]["mode"] = args.configs
        except Exception as err:   # the extra legs must never take the metric line down with them
            line["configs"] = {"error": "%s: %s" % (type(err).__name__, err)}
    return line


def roofline_large(model, link, device, stream, bytes_per_eval):
    """The metric kernel at batches whose per-launch traffic is far beyond the Infinity Cache (256 MiB): every launch
    streams its inputs from and its outputs to HBM.  Same measurement as the metric leg (hipGraph, HIP events)."""
    import torch
    out = []
    for B, K, reps in ((1 << 22, 50, 5), (1 << 24, 20, 5)):   # 5 timed regions of ~7-13 ms of streaming each
        need = B * bytes_per_eval * 1.1
        free, _ = torch.cuda.mem_get_info(device)
        if free < need:
            out.append({"batch": B, "skipped": "needs %.1f GB of HBM, %.1f GB free" % (need / 1e9, free / 1e9)})
            continue
        q, _ = sample_q(model, B, device, 99)
        plan = model.plan_fk_and_jacobian(q, link)
        for _ in range(3):
            plan.launch()
        torch.cuda.synchronize()
        # HBM streaming rates on this pool swing by +-12 % from one timed region to the next (same process, same buffers),
        # so one region says little: `achieved` is the MEDIAN of `reps` regions, the spread is reported alongside
        times = sorted(timed_graph_region(plan.launch, K, stream, lambda: None, warm_replays=1)[1] / K for _ in range(reps))
        launch_s = times[len(times) // 2]
        achieved = bytes_per_eval * B / launch_s / 1e9
        out.append({"batch": B, "steps": K, "regions": reps, "launch_us": launch_s * 1e6,
                    "launch_us_min_max": [times[0] * 1e6, times[-1] * 1e6], "achieved": achieved,
                    "achieved_best": bytes_per_eval * B / times[0] / 1e9, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "bytes_per_launch": bytes_per_eval * B,
                    "evals_per_s": B / launch_s,
                    "kernel": "drm::fk_jacobian_arm_kernel<8, 7, true, 4, true> (outputs streamed past the Infinity Cache)"})
        del plan, q
        torch.cuda.empty_cache()
    return out


def config3_inputs(model, rows, device, seed):
    """q ~ U(limits), qd ~ U(+-0.2 vmax), qdd ~ U(+-0.4 vmax): the distribution of data_utils.py:70-98, resident in HBM."""
    import torch
    n = model._n_dofs
    q, gen = sample_q(model, rows, device, seed)
    vmax = torch.tensor([j["velocity"] for j in model.get_joint_limits()], device=device)
    qd = ((torch.rand(rows, n, device=device, generator=gen) * 2 - 1) * 0.2 * vmax).contiguous()
    qdd = ((torch.rand(rows, n, device=device, generator=gen) * 2 - 1) * 0.4 * vmax).contiguous()
    return q, qd, qdd


def test_mode_fields(args, world):
    import torch
    out = {"backend": args.backend if world > 1 else None}
    if args.shared_gpu and world > 1:
        out["shared_gpu"] = True
        out["devices_used"] = min(world, torch.cuda.device_count())
        out["note"] = ("TEST MODE: %d ranks share %d device(s), collectives staged through the host by gloo — this line "
                       "exercises the multi-rank code path and is NOT a scaling measurement" % (world, out["devices_used"]))
    return out


def run_config3(args, model, link, device, world, rank, ranks_seen, stream, barrier, reduce_max, shard_bounds):
    """BASELINE configuration 3: global batch 2^20 sharded by rows, FK(EE) + RNEA per shard (one fused launch), and what a step
    exchanges afterwards: nothing (the outputs stay sharded: the headline), the torques, everything to rank 0, or everything to
    every rank — all four timed in this run (gather_modes)."""
    import torch

    from differentiable_robot_model_amd.distributed import all_gather_flat
    n, K, W = model._n_dofs, args.steps, args.warmup
    G = CONFIG3_GLOBAL_BATCH
    lo, hi = shard_bounds(G, world, rank)
    rows = hi - lo
    q, qd, qdd = config3_inputs(model, rows, device, 4321 + rank)
    # round 6: a constant model runs its OWN kernels by default (the streaming walk with this arm's constants folded into the
    # instruction stream; the code objects ship next to the library, nothing is compiled on the call path) — the model below is
    # DifferentiableRobotModel(urdf, device) and nothing else.  --library-kernels: the library's table-driven kernels (own_kernels
    # = "off").  A checkout without the shipped code objects compiles them now (model.specialize()); the line says which happened.
    own_kernel, own_why = False, "--library-kernels"
    if args.library_kernels:
        model.own_kernels = "off"
    # tau | pos | quat of a shard live in ONE allocation (three contiguous blocks), so the collective sends the kernel's own
    # output buffer: no packing kernel between the launch and the all-gather
    width = n + 3 + 4                                                    # 56 B per row
    flat = torch.empty(rows * width, device=device)
    outs = (flat[:rows * n].view(rows, n), flat[rows * n:rows * (n + 3)].view(rows, 3), flat[rows * (n + 3):].view(rows, 4))
    plan = model.plan_fk_and_inverse_dynamics(q, qd, qdd, link, outputs=outs)
    if not args.library_kernels:
        from differentiable_robot_model_amd import specialize as sp
        fused = lambda: (getattr(model._dynamics_walk().program, "_special", None) or {}).get(sp.SPECIAL_FK_RNEA_ARM)
        own_kernel, own_why = ("default", None) if fused() else (False, "no shipped code object")
        if not own_kernel:
            try:
                model.specialize()
                plan = model.plan_fk_and_inverse_dynamics(q, qd, qdd, link, outputs=outs)
                own_kernel, own_why = ("specialize()", None) if fused() else (False, "specialize() attached nothing")
            except Exception as err:       # noqa: BLE001
                own_kernel, own_why = False, str(err)[:200]
    gathered = torch.empty(world * rows * width, device=device) if world > 1 and G % world == 0 else None
    if world > 1 and gathered is None:
        sys.exit("bench.py --config 3: 2^20 rows do not split evenly over %d ranks" % world)

    def compute():
        plan.launch()

    from differentiable_robot_model_amd.distributed import gather_flat, gather_model_us
    gathered_tau = torch.empty(world * rows * n, device=device) if world > 1 else None
    headline = args.gather or "none"

    # the one-sided gather (round 6): every rank owns the gathered arrays of the whole batch, the handles are exchanged ONCE, and
    # the launch itself writes this rank's rows into its own arrays and into the peers' (drm_fk_rnea_put)
    peer, plan_put = None, None
    if world > 1:
        from differentiable_robot_model_amd.distributed import PeerGather
        try:
            peer = PeerGather(G, n, device, mode="all")
            assert (peer.lo, peer.hi) == (lo, hi)
            plan_put = model.plan_fk_and_inverse_dynamics(q, qd, qdd, link, outputs=peer.outputs(), put=peer.put())
        except Exception as err:       # noqa: BLE001  (no IPC on this stack: the collective modes stay)
            peer, plan_put, p2p_why = None, None, "%s: %s" % (type(err).__name__, str(err)[:160])
            print("bench.py: one-sided gather unavailable (%s)" % p2p_why, file=sys.stderr)

    def exchange(mode):
        if world <= 1 or mode == "none":
            return
        if mode == "all":
            all_gather_flat(gathered, flat)               # rank r's blocks at gathered[r * rows * width:]
        elif mode == "tau":
            all_gather_flat(gathered_tau, flat[:rows * n])
        else:
            gather_flat(gathered, flat, dst=0)

    def step_of(mode):
        if mode == "p2p":
            return plan_put.launch          # (compute and exchange are ONE launch)

        def step():
            compute()
            exchange(mode)
        return step

    step = step_of(headline)
    for _ in range(W):
        step()
    torch.cuda.synchronize()
    # the compute of a step alone (graph of K launches), then the full step of every exchange mode (compute + collective)
    _, dev_compute, _ = timed_graph_region(compute, K, stream, barrier, use_graph=not args.no_graph)
    modes = {}
    if headline == "p2p" and plan_put is None:
        sys.exit("bench.py --gather p2p: the one-sided gather needs IPC-mapped peer buffers (%s)" % ("one rank" if world == 1 else "see stderr"))
    for mode in (["none"] if world == 1 else ["none", "tau", "root", "all"] + (["p2p"] if plan_put is not None else [])):
        fn = step_of(mode)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        # (a step with a collective is launched eagerly: RCCL under stream capture has never run on this stack — no multi-GPU
        # node, and RCCL refuses two ranks on one device — so there is no switch for it)
        w_, d_, graphed = timed_graph_region(fn, K, stream, barrier, use_graph=mode in ("none", "p2p") and not args.no_graph)
        w_, d_ = reduce_max([w_, d_])
        modes[mode] = {"ms_per_step": w_ / K * 1e3, "value": G * K / w_, "step_us_device": d_ / K * 1e6, "hipgraph": bool(graphed),
                       "gather_bytes_per_rank": 0 if mode == "none" else rows * 4 * (n if mode == "tau" else width),
                       "gather_model_us": gather_model_us(mode, rows * width * 4, rows * n * 4, world)}
        if mode == "p2p":      # the puts ride on the walk: a step should take max(compute, one block over one link), not their sum
            modes[mode]["one_sided"] = True
    wall, dev_time = modes[headline]["ms_per_step"] * K * 1e-3, modes[headline]["step_us_device"] * K * 1e-6
    (dev_compute,) = reduce_max([dev_compute])
    verified, vs_whole = None, None
    if args.verify_gather and world > 1:
        # rank 0 rebuilds EVERY rank's inputs (same seeds), runs them as ONE single-rank launch over all 2^20 rows and holds
        # the gathered buffers to it bit for bit: rank r's tau | pos | quat blocks at gathered[r * rows * width:] (all-gather,
        # then the gather to rank 0 into the zeroed buffer), its torques at gathered_tau[r * rows * n:]
        step_of("all")()
        step_of("tau")()
        torch.cuda.synchronize()
        ok = 1
        if rank == 0:
            # The gathered buffers are held BIT FOR BIT to launches of the SHARD size (what the ranks ran: the C ABI picks a kernel
            # form by launch size — 131 072-row shards take the latency form, 2^20 rows the streaming form — and two forms may
            # differ in the last bit of a row); the same rows as ONE launch over all of them are reported beside it.
            parts = [config3_inputs(model, rows, device, 4321 + r) for r in range(world)]
            outs = []
            for p_ in parts:
                shard = model.plan_fk_and_inverse_dynamics(p_[0], p_[1], p_[2], link)
                shard.launch()
                torch.cuda.synchronize()
                outs.append([t.clone() for t in shard.outputs()])
            tau_w, pos_w, quat_w = (torch.cat([o[i] for o in outs]) for i in range(3))
            qa, qda, qdda = (torch.cat([p[i] for p in parts]) for i in range(3))
            whole = model.plan_fk_and_inverse_dynamics(qa, qda, qdda, link)
            whole.launch()
            torch.cuda.synchronize()
            pairs = list(zip((tau_w, pos_w, quat_w), whole.outputs()))
            vs_whole = {"bit_equal": all(torch.equal(a, b.reshape(a.shape)) for a, b in pairs),
                        "max_abs": max(float((a - b.reshape(a.shape)).abs().max()) for a, b in pairs),
                        "tau_max_rel": float(((pairs[0][0] - pairs[0][1]).abs() / pairs[0][1].abs().clamp_min(1.0)).max())}
            def blocks_ok():
                good = 1
                for r in range(world):
                    blk = gathered[r * rows * width:(r + 1) * rows * width]
                    sl = slice(r * rows, (r + 1) * rows)
                    good &= int(torch.equal(blk[:rows * n].view(rows, n), tau_w[sl]))
                    good &= int(torch.equal(blk[rows * n:rows * (n + 3)].view(rows, 3), pos_w[sl].reshape(rows, 3)))
                    good &= int(torch.equal(blk[rows * (n + 3):].view(rows, 4), quat_w[sl].reshape(rows, 4)))
                return good
            ok &= blocks_ok()
            ok &= int(torch.equal(gathered_tau.view(world * rows, n), tau_w))
            gathered.zero_()
        step_of("root")()                                   # (every rank takes part; only rank 0 receives)
        torch.cuda.synchronize()
        if rank == 0:
            ok &= blocks_ok()
        if plan_put is not None:
            # the one-sided gather: every rank's arrays hold ALL rows after its peers' launches have completed (synchronize + barrier)
            for t in peer.gathered():
                t.zero_()
            torch.cuda.synchronize()
            barrier()
            plan_put.launch()
            torch.cuda.synchronize()
            barrier()
            if rank == 0:
                for got, want in zip(peer.gathered(), (tau_w, pos_w, quat_w)):
                    ok &= int(torch.equal(got, want.reshape(got.shape)))
        verified = bool(ok)
    bytes_per_eval = 4 * (3 * n + n + 7)                                 # q qd qdd in; tau pos quat out = 140 B
    launch_s = dev_compute / K
    achieved = bytes_per_eval * rows / launch_s / 1e9
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # the UNMODIFIED reference on this box's host cores in this run, on a bounded sample of rank 0's shard (its RNEA runs
        # ~5e4 evals/s on one core): compute_endeffector_jacobian / compute_inverse_dynamics timings + the HIP path's deviation
        # from the reference's own tau / pos / quat on the first rows
        sample = min(rows, 32768)
        compute()
        torch.cuda.synchronize()
        got = {k: t[:sample].cpu().numpy() for k, t in zip(("tau", "pos", "quat"), plan.outputs())}
        cpu = reference_cpu(args.robot, link, q[:sample].cpu().numpy(), qd[:sample].cpu().numpy(), qdd[:sample].cpu().numpy(),
                            gpu_outputs=got, public_rows=2048)
        if "one_thread" in cpu and "inverse_dynamics" in cpu["one_thread"]:
            cpu = {"value": cpu["one_thread"]["inverse_dynamics"]["evals_per_s"], "unit": "evals/s", "cores": 1, "kind": "reference",
                   "sample": "compute_inverse_dynamics of the unmodified reference on the first %d rows of rank 0's shard, one "
                             "thread, min of 3 (FK not included: its tensor part is in `reference`)" % sample, "reference": cpu}
    line = {
        "metric": "FK + RNEA evals/sec, Panda 7-DoF, global batch 2^20 sharded over the GPUs (BASELINE.json configuration 3)",
        "value": G * K / wall, "unit": "evals/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen, "gather_verified": verified,
        "gather_vs_whole_launch": vs_whole, "distributed": test_mode_fields(args, world),
        "config": {"workload": "Franka Panda 7-DoF, FK(panda_virtual_ee_link) + RNEA inverse dynamics (gravity, damping), "
                               "global batch %d = %d rows per GPU, q~U(limits), qd~U(+-0.2 vmax), qdd~U(+-0.4 vmax); "
                               "one fused drm_fk_rnea launch per step; exchange mode of the headline value: %s" % (G, rows, headline),
                   "batch_per_gpu": rows, "global_batch": G, "parallelism": "batch-sharded x%d" % world,
                   "launch": "hipGraph of K steps" if modes[headline]["hipgraph"] else "eager steps (kernel + collective)",
                   "gather": headline if world > 1 else "none",
                   "gather_bytes_per_rank": modes[headline]["gather_bytes_per_rank"]},
        "own_kernel": own_kernel, "own_kernel_unavailable": own_why,
        "compute_us_per_step": dev_compute / K * 1e6, "step_us_with_gather": dev_time / K * 1e6,
        "gather_us_per_step": max(0.0, (dev_time - dev_compute) / K * 1e6) if world > 1 else 0.0,
        "gather_modes": modes,
        "gather_modes_note": "the same K steps timed once per exchange mode (barrier + synchronize bracket, max over ranks): none = "
                             "outputs stay sharded on their GPUs (the headline), tau / all = all_gather_into_tensor of the torques / of "
                             "tau | pos | quat, root = gather to rank 0; gather_model_us = one block over one 153 GB/s xGMI link "
                             "(distributed.gather_model_us) — what the exchange should add once it runs over RCCL",
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None,
                     "kernel": ("drm_fk_rnea_arm_static (this robot's constants folded in, csrc/drm_arm_stream.hpp)"
                                if own_kernel and rows >= 1024 * 128 else
                                "drm::fk_rnea_arm2_kernel<8, 7, 7> (two samples per lane)" if rows > 1024 * 64
                                else "drm::fk_rnea_arm_kernel<8, 7, 7>"), "bytes_per_eval": bytes_per_eval,
                     "launch_us": launch_s * 1e6,
                     "note": "the fused kernel alone (hipGraph of K launches, HIP events); RNEA sits at the vector-FP32 / HBM "
                             "ridge (2.6 kflop per 140 B), see DESIGN.md"},
    }
    if peer is not None:
        line["one_sided_gather"] = {"in_kernel": bool(own_kernel) and rows >= 1024 * 128,
                                    "note": "drm_fk_rnea_put: the arm's own fused kernel stores every tile to the peers' gathered arrays "
                                            "(IPC-mapped) from its epilogue; launches below 131 072 rows / walks without that kernel "
                                            "compute first and copy on the stream"}
        peer.close()
    if cpu is not None:
        line["cpu_baseline"] = cpu
        ref = cpu.get("reference", cpu)
        if "gpu_vs_reference_max_abs" in ref:
            line["gpu_vs_reference_max_abs"] = ref["gpu_vs_reference_max_abs"]
    return line


if __name__ == "__main__":
    main()
