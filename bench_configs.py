"""bench.py's `configs` legs: every configuration BASELINE.json names beside the metric, measured in the SAME run as the
metric line (N = 1, rank 0, after the timed region), each with its roofline fraction against SURVEY.md §8(d)'s algorithmic
bytes, the UNMODIFIED reference timed on this box's host cores beside it (oracle/ref_timing.py --jobs, its own interpreter:
one thread = the CPU baseline of record, and up to 16 threads) and the HIP path's deviation from the reference's own outputs.

  config 2   Kuka iiwa 7-DoF, 65 536 rows, FK + end-effector Jacobian          (reference robot_model.py:626-667)   224 B/eval
  config 3   Franka Panda, FK(EE) + RNEA: one GPU's shard of the 8-GPU run (131 072 rows) and the whole 2^20 batch, ONE fused
             launch                                                           (robot_model.py:305-375 + 223-248)    140 B/eval
  config 4   Allegro hand, 65 536 rows, FK to the four fingertips through compute_forward_kinematics_links
                                                                               (robot_model.py:197-248)              176 B/eval
  config 5   iiwa, learnable trans + rot_angles of iiwa_link_1, 16 384 rows: FK + backward kernels, and the whole training step
             (forward, MSE, backward, Adam) as a replayed hipGraph   (examples/learn_kinematics_of_iiwa.py:25-61)    96 B/eval
  api_eager  host time of one eager call of the three public methods at 65 536 rows (what a drop-in caller pays per call; the
             metric line itself replays prepared launches from a hipGraph)

Launch times: hipGraph of K launches, HIP events on the launch stream, median of `reps` regions (the same method as the metric
line's `roofline.steady_state`).  Input distributions: SURVEY.md §8(d).
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0
FP32_VECTOR_PEAK_TFLOPS = 157.3          # packed FMA, SURVEY.md §8(d)
ALLEGRO_TIPS = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]
# measured bytes-only floors (a kernel that moves the same bytes and computes nothing), where one was recorded under profiles/
IO_FLOOR_US = {"config2": (2.9, "profiles/r02_metric_lab.txt (224 B/eval at 65 536 rows)"),
               "config3_whole": (17.7, "profiles/r03_io_floors_2p20.txt (140 B/eval at 2^20 rows)"),
               "config4": (2.17, "profiles/r03_metric_lab.txt (176 B/eval at 65 536 rows)")}


def graph_launch_us(fn, K, reps=5):
    """Median and minimum duration (us) of one launch: K launches captured into a hipGraph, HIP events around a replay."""
    import torch
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t_warm = time.perf_counter()      # ~20 ms of the same replays first: the clocks leave their idle state (as bench.py's metric leg does)
    while time.perf_counter() - t_warm < 0.02:
        g.replay()
        torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e) / K * 1e3)
    times.sort()
    return times[len(times) // 2], times[0]


def roofline(bytes_per_eval, rows, launch_us, kernel, flops_per_eval=None):
    achieved = bytes_per_eval * rows / launch_us / 1e3
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "bytes_per_eval": bytes_per_eval, "algorithmic_bytes_per_launch": bytes_per_eval * rows, "launch_us": launch_us,
           "kernel": kernel, "traffic": None}
    if flops_per_eval:
        tf = flops_per_eval * rows / launch_us / 1e6
        out["vector_fp32"] = {"achieved": tf, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_VECTOR_PEAK_TFLOPS,
                              "flops_per_eval": flops_per_eval}
    return out


def load(name, device):
    import contextlib
    import io

    from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel, robot_description_folder
    with contextlib.redirect_stdout(io.StringIO()):
        return DifferentiableRobotModel(os.path.join(robot_description_folder, name + ".urdf"), device=device)


def uniform_q(model, rows, device, seed):
    import torch
    lim = model.get_joint_limits()
    lo = torch.tensor([j["lower"] for j in lim], device=device)
    hi = torch.tensor([j["upper"] for j in lim], device=device)
    gen = torch.Generator(device=device).manual_seed(seed)
    return (lo + (hi - lo) * torch.rand(rows, len(lim), device=device, generator=gen)).contiguous(), gen


def deviation(ref, name, got, tau_relative=True):
    """max |HIP - reference| per output on the rows the reference produced (tau relative to 1 + |tau|)."""
    import numpy as np
    dev = {}
    for key, val in got.items():
        want = ref["%s/%s" % (name, key)]
        val = np.asarray(val)[:want.shape[0]]
        if key == "quat":
            dev["quat_sign_flips"] = int((np.sum(want * val, -1) < 0).sum())
        scale = 1.0 + np.abs(want) if (key == "tau" and tau_relative) else 1.0
        dev[key] = float((np.abs(val - want) / scale).max())
    return dev


def api_eager(model_panda, device, rows=65536, calls=300):
    """Host time per eager call of the public API (wall clock over `calls` back-to-back calls, one synchronize at the end: the
    kernels take 3-6 us, the host side of a call is what bounds the loop)."""
    import torch
    q, gen = uniform_q(model_panda, rows, device, 7)
    qd = torch.rand(rows, 7, device=device, generator=gen) - 0.5
    qdd = torch.rand(rows, 7, device=device, generator=gen) - 0.5
    link = "panda_virtual_ee_link"
    out = {"rows": rows, "calls": calls}
    for name, fn in (("compute_forward_kinematics", lambda: model_panda.compute_forward_kinematics(q, link)),
                     ("compute_endeffector_jacobian", lambda: model_panda.compute_endeffector_jacobian(q, link)),
                     ("compute_inverse_dynamics", lambda: model_panda.compute_inverse_dynamics(q, qd, qdd))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(calls):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / calls * 1e6)
        out[name] = {"us_per_call": best, "evals_per_s": rows / best * 1e6}
    # a LEARNED model (the Panda's link 4 with a PositiveScalar mass and an SPD inertia matrix) where no graph is built: the prepared
    # call rebuilds the walk table from the parameter tensors in front of the launch (two launches per call)
    try:
        from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, SymmPosDef3DInertiaMatrixNet
        learned = load("panda_no_gripper", device)
        learned.make_link_param_learnable("panda_link4", "mass", PositiveScalar())
        learned.make_link_param_learnable("panda_link4", "inertia_mat", SymmPosDef3DInertiaMatrixNet())
        with torch.no_grad():
            fn = lambda: learned.compute_inverse_dynamics(q, qd, qdd)
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(calls):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / calls * 1e6)
        out["learned_model_inverse_dynamics"] = {"us_per_call": best, "evals_per_s": rows / best * 1e6}
    except Exception as err:   # pragma: no cover - depends on the runtime
        out["learned_model_error"] = repr(err)[:160]
    from differentiable_robot_model_amd import backend
    out["host_path"] = "csrc/drm_hostcall.so (C++)" if backend.hostcall() is not None else "Python + ctypes"
    out["note"] = ("eager public-API calls: tensor_check in Python, then input checks, ONE output allocation and the C-ABI call — in "
                   "csrc/drm_hostcall.so (a torch extension without device code) when it is built, in Python + ctypes otherwise — and the "
                   "HIP launch; the metric line replays a prepared launch (plan_fk_and_jacobian) from a hipGraph instead")
    return out


def learn_dynamics_step(device, B=256):
    """Not a BASELINE configuration; the reference's other learning workload (examples/learn_dynamics_iiwa.py:49-96, the L4DC notebook's
    batch): iiwa, PositiveScalar masses, free centres of mass and inertia matrices of the seven links, MSE on the torques, Adam — the
    whole step replayed from a hipGraph, with torch's default Adam and with its fused one (us per step, HIP events).  ABI 13's table
    kernels (drm_walk_table_links) evaluate the parameter modules; `without_table_links_us`: the modules' torch kernels + cat instead."""
    import torch
    from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, UnconstrainedTensor
    out = {"workload": "iiwa7, 21 parameter tensors (mass / com / inertia_mat of seven links), batch %d, forward + MSE + backward + Adam, "
                       "replayed from a hipGraph" % B, "batch": B}
    try:
        for key, links_path, fused in (("graph_step_us", True, None), ("graph_step_fused_adam_us", True, True),
                                       ("without_table_links_us", False, True)):
            torch.manual_seed(0)
            truth, m = load("iiwa7", device), load("iiwa7", device)
            m._table_links = links_path
            for k in range(1, 8):
                m.make_link_param_learnable("iiwa_link_%d" % k, "mass", PositiveScalar())
                m.make_link_param_learnable("iiwa_link_%d" % k, "com", UnconstrainedTensor(1, 3))
                m.make_link_param_learnable("iiwa_link_%d" % k, "inertia_mat", UnconstrainedTensor(3, 3))
            q, _ = uniform_q(m, B, device, 11)
            qd, qdd = torch.rand(B, 7, device=device) - 0.5, torch.rand(B, 7, device=device) - 0.5
            with torch.no_grad():
                want = truth.compute_inverse_dynamics(q, qd, qdd)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=fused)

            def step():
                loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(q, qd, qdd), want)
                loss.backward()
                opt.step()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    opt.zero_grad(set_to_none=True)
                    step()
            torch.cuda.current_stream().wait_stream(side)
            opt.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            times = []
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(50):
                    graph.replay()
                e.record()
                torch.cuda.synchronize()
                times.append(s.elapsed_time(e) / 50 * 1e3)
            out[key] = sorted(times)[len(times) // 2]
            del graph
        out["steps_per_s_fused_adam"] = 1e6 / out["graph_step_fused_adam_us"]
    except Exception as err:   # pragma: no cover - depends on the runtime
        out["error"] = repr(err)[:200]
    return out


def run_reference(jobs, arrays, reps=2, timeout=240):
    """oracle/ref_timing.py --jobs in its own interpreter on the host cores only; returns (record, outputs npz dict)."""
    import numpy as np
    script = os.path.join(ROOT, "oracle", "ref_timing.py")
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "arrays.npz"), **arrays)
        with open(os.path.join(tmp, "jobs.json"), "w") as f:
            json.dump(jobs, f)
        cmd = [sys.executable, script, "--jobs", os.path.join(tmp, "jobs.json"), "--arrays", os.path.join(tmp, "arrays.npz"),
               "--out-npz", os.path.join(tmp, "out.npz"), "--reps", str(reps)]
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        try:
            done = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env)
            rec = json.loads(done.stdout.decode().strip().splitlines()[-1])
            outs = dict(np.load(os.path.join(tmp, "out.npz"))) if os.path.exists(os.path.join(tmp, "out.npz")) else {}
        except Exception as err:   # the baseline leg must never take the GPU numbers down with it
            return {"kind": "reference", "error": "%s: %s" % (type(err).__name__, err)}, {}
    return rec, outs


# ---------------------------------------------------------------------------------------------------- HBM traffic of every leg
TRAFFIC_LEGS = ("config2", "config3_shard", "config3_whole", "config4", "config5", "config5_fk_mse")


def leg_launcher(name, device):
    """One leg's launch as a closure (the same robots, batches, seeds and entry points run_config_legs times) — what the
    `--pmc-child NAME` mode of this file runs under rocprofv3 so that the counters of a pass belong to ONE leg."""
    import torch

    from differentiable_robot_model_amd import backend
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    if name == "config2":
        iiwa = load("iiwa7", device)
        q, _ = uniform_q(iiwa, 65536, device, 2002)
        return iiwa.plan_fk_and_jacobian(q, "iiwa_link_ee").launch
    if name in ("config3_shard", "config3_whole"):
        panda = load("panda_no_gripper", device)      # (its own kernels by default: the code objects ship with the library)
        rows = 131072 if name == "config3_shard" else 1 << 20
        vmax = torch.tensor([j["velocity"] for j in panda.get_joint_limits()], device=device)
        q, gen = uniform_q(panda, rows, device, 4321)
        qd = ((torch.rand(rows, 7, device=device, generator=gen) * 2 - 1) * 0.2 * vmax).contiguous()
        qdd = ((torch.rand(rows, 7, device=device, generator=gen) * 2 - 1) * 0.4 * vmax).contiguous()
        return panda.plan_fk_and_inverse_dynamics(q, qd, qdd, "panda_virtual_ee_link").launch
    if name == "config4":
        hand = load("allegro_left", device)
        q, _ = uniform_q(hand, 65536, device, 4004)
        return lambda: hand.compute_forward_kinematics_links(q, ALLEGRO_TIPS)
    if name in ("config5", "config5_fk_mse"):
        torch.manual_seed(0)
        iiwa, learn = load("iiwa7", device), load("iiwa7", device)
        learn.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(dim1=1, dim2=3))
        learn.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(dim1=1, dim2=3))
        B = 16384
        q, _ = uniform_q(iiwa, B, device, 5005)
        with torch.no_grad():
            want, _ = iiwa.compute_forward_kinematics(q, "iiwa_link_ee")
        ee = learn._name_to_idx_map["iiwa_link_ee"]
        dw = learn._get_walk(("fk", (ee,)), targets=[ee])
        ops_f = learn._ops_f(dw).detach()
        gpos = torch.randn(B, 1, 3, device=device)
        mask = learn._kinematic_param_mask(dw)
        if name == "config5_fk_mse":
            return lambda: backend.fk_mse(dw.program, ops_f, dw.ops_i, q, want, 7, mask, False)

        def both():
            backend.fk(dw.program, ops_f, dw.ops_i, q, 1, 7)
            backend.fk_backward(dw.program, ops_f, dw.ops_i, q, gpos, 1, 7, mask, False)
        return both
    raise ValueError(name)


PMC_CALLS = 12


def pmc_child(name):
    import torch
    device = torch.device("cuda", 0)
    fn = leg_launcher(name, device)
    torch.cuda.synchronize()
    print("PMC_LEG_BEGIN", flush=True)
    with torch.no_grad():
        for _ in range(PMC_CALLS):
            fn()
    torch.cuda.synchronize()


def leg_kernel(row_name):
    """Is a dispatch of the counter CSV one of a leg's own launches?  Everything this library launches (drm:: kernels, the robots'
    own *_static kernels) and the runtime's fill kernel behind hipMemsetAsync (the backward scratch), minus what building a model
    launches once (the link-row / walk-table builders)."""
    if any(t in row_name for t in ("link_rows", "walk_table")):
        return False
    return any(t in row_name for t in ("drm", "_static", "fillBuffer"))


def measured_leg_traffic(name, timeout=120):
    """HBM bytes per launch of one leg, MEASURED: two runs of `bench_configs.py --pmc-child NAME` under `rocprofv3 --kernel-trace
    --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes, kernel tracing only, MI355X_MICROARCH.md), summed over the leg's kernels
    (memsets and reductions of the backward included) and divided by the number of calls: FETCH_SIZE (KiB) x 2 (the gfx950
    correction) + WRITE_SIZE (KiB)."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    env = dict(os.environ, TMPDIR="/tmp", DRM_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    got, kernels = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="drm_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", name]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            rows_of = {}
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        kn = row.get("Kernel_Name", "")
                        if row.get("Counter_Name") == counter and leg_kernel(kn):
                            rows_of.setdefault(kn.split("(")[0][:80], []).append((int(row.get("Dispatch_Id", 0)), float(row["Counter_Value"])))
            # a kernel's dispatches beyond a multiple of PMC_CALLS are the child's setup (the target of config 5's loss, a memset
            # behind an allocation): they come first and are dropped
            total, per = 0.0, {}
            for kn, rows in rows_of.items():
                rows.sort()
                keep = rows[len(rows) % PMC_CALLS:]
                if keep:
                    total += sum(v for _, v in keep)
                    per[kn] = len(keep)
            if not per:
                return None, "no %s rows for the leg's kernels in the rocprofv3 output" % counter
            got[counter] = total / PMC_CALLS
            kernels = per
        except (subprocess.SubprocessError, OSError, ValueError, KeyError) as e:
            return None, "%s pass failed: %s" % (counter, type(e).__name__)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fetch, write = got["FETCH_SIZE"] * 1024.0 * 2.0, got["WRITE_SIZE"] * 1024.0
    return {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "calls": PMC_CALLS,
            "dispatches_per_call": {k: v / PMC_CALLS for k, v in kernels.items()}}, None


def fill_traffic(legs):
    """roofline.traffic of every leg from counters collected NOW (N = 1, after the timed regions; ~8 s per leg and counter)."""
    for leg in legs:
        for key, child in (("roofline", leg["name"]), ("fk_mse_roofline", leg["name"] + "_fk_mse")):
            if key not in leg or child not in TRAFFIC_LEGS:
                continue
            got, why = measured_leg_traffic(child)
            r = leg[key]
            if got is None:
                r["traffic_unmeasured"] = why
                continue
            r["traffic"] = got["bytes"]
            r["traffic_fetch_bytes"], r["traffic_write_bytes"] = got["fetch_bytes"], got["write_bytes"]
            r["traffic_over_algorithmic"] = got["bytes"] / r["algorithmic_bytes_per_launch"]
            r["traffic_dispatches_per_call"] = got["dispatches_per_call"]
            r["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of bench_configs.py "
                                   "--pmc-child %s (FETCH x 2 + WRITE, KiB), summed over the leg's kernels, per call" % child)


def run_config_legs(device, with_reference=True, with_traffic=True):
    import numpy as np
    import torch

    from differentiable_robot_model_amd import backend
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    legs, jobs, arrays, gpu_out = [], [], {}, {}
    host = lambda t: t.detach().cpu().numpy()

    # ------------------------------------------------------------------ config 2: iiwa FK + Jacobian, 65 536 rows
    iiwa = load("iiwa7", device)
    B = 65536
    q, _ = uniform_q(iiwa, B, device, 2002)
    plan = iiwa.plan_fk_and_jacobian(q, "iiwa_link_ee")
    us, us_min = graph_launch_us(plan.launch, 100)
    legs.append({"config": 2, "name": "config2", "workload": "Kuka iiwa 7-DoF, FK + end-effector Jacobian to iiwa_link_ee, batch %d, "
                 "q~U(joint limits)" % B, "batch": B, "launch_us": us, "launch_us_min": us_min, "evals_per_s": B / us * 1e6,
                 "roofline": roofline(224, B, us, "drm::fk_jacobian_arm_kernel<8, 7, true, 1, false>"),
                 "io_floor_us": IO_FLOOR_US["config2"][0], "io_floor_source": IO_FLOOR_US["config2"][1]})
    arrays["c2_q"] = host(q)
    jobs.append({"name": "config2", "robot": "iiwa7", "kind": "jacobian", "link": "iiwa_link_ee", "q": "c2_q", "public_rows": 1024})
    gpu_out["config2"] = {k: host(t) for k, t in zip(("pos", "quat", "lin_jac", "ang_jac"), plan.outputs())}
    del plan

    # ------------------------------------------------------------------ config 3: Panda FK(EE) + RNEA, shard and whole batch
    from differentiable_robot_model_amd import specialize as sp
    link = "panda_virtual_ee_link"
    # round 6: the drop-in path IS the benchmarked path.  A constant model attaches its OWN kernels by itself on first use — Panda's
    # constants folded into the instruction stream of the streaming walk; the code objects ship next to the library (csrc/special_cache/,
    # built by __graft_entry__.build()), nothing is compiled here.  `panda` below is DifferentiableRobotModel(urdf, device) and
    # nothing else.  The library's table-driven kernels (own_kernels = "off") are timed beside it.
    panda = load("panda_no_gripper", device)
    own = load("panda_no_gripper", device)
    panda.own_kernels = "off"
    probe_q = torch.zeros(64, 7, device=device)
    own.plan_fk_and_inverse_dynamics(probe_q, probe_q.clone(), probe_q.clone(), link)
    own_ok = bool((getattr(own._dynamics_walk().program, "_special", None) or {}).get(sp.SPECIAL_FK_RNEA_ARM))
    own_path, own_why = "default" if own_ok else None, None
    if not own_ok:       # (a checkout whose special_cache was not built: compile now, and say so)
        try:
            own_ok, own_path = bool(own.specialize()), "specialize()"
            own.plan_fk_and_inverse_dynamics(probe_q, probe_q.clone(), probe_q.clone(), link)
        except Exception as err:       # noqa: BLE001  (no hipcc on this machine either: the library's kernels are the path)
            own_ok, own_why = False, str(err)[:200]
    vmax = torch.tensor([j["velocity"] for j in panda.get_joint_limits()], device=device)
    for rows, label, K in ((131072, "config3_shard", 100), (1 << 20, "config3_whole", 30)):
        q, gen = uniform_q(panda, rows, device, 4321)
        qd = ((torch.rand(rows, 7, device=device, generator=gen) * 2 - 1) * 0.2 * vmax).contiguous()
        qdd = ((torch.rand(rows, 7, device=device, generator=gen) * 2 - 1) * 0.4 * vmax).contiguous()
        plan = panda.plan_fk_and_inverse_dynamics(q, qd, qdd, link)
        us_lib, us_lib_min = graph_launch_us(plan.launch, K)
        us, us_min, kernel = us_lib, us_lib_min, "drm::fk_rnea_arm2_kernel<8, 7, 7> (library, table in LDS)"
        if own_ok:
            plan = own.plan_fk_and_inverse_dynamics(q, qd, qdd, link)
            us, us_min = graph_launch_us(plan.launch, K)
            kernel = "drm_fk_rnea_arm_static (this robot's constants folded in; csrc/drm_arm_stream.hpp)"
        leg = {"config": 3, "name": label, "batch": rows, "launch_us": us, "launch_us_min": us_min, "evals_per_s": rows / us * 1e6,
               "workload": "Franka Panda 7-DoF, FK(%s) + RNEA inverse dynamics (gravity, damping), %d rows%s, q~U(limits), "
                           "qd~U(+-0.2 vmax), qdd~U(+-0.4 vmax); ONE fused drm_fk_rnea launch" %
                           (link, rows, " = one GPU's shard of the 2^20 batch at N = 8" if rows == 131072 else " = the whole batch on one GPU"),
               "own_kernel": own_ok, "own_kernel_path": own_path, "own_kernel_unavailable": own_why,
               "library_kernel_launch_us": us_lib, "library_kernel_launch_us_min": us_lib_min,
               "roofline": roofline(140, rows, us, kernel, flops_per_eval=2600 + 830)}
        if label in IO_FLOOR_US:
            leg["io_floor_us"], leg["io_floor_source"] = IO_FLOOR_US[label]
        legs.append(leg)
        if rows == 131072:
            ref_rows = 32768     # the reference's RNEA runs ~5e4 evals/s on one core: a bounded sample of the shard
            arrays.update(c3_q=host(q[:ref_rows]), c3_qd=host(qd[:ref_rows]), c3_qdd=host(qdd[:ref_rows]))
            jobs.append({"name": "config3", "robot": "panda_no_gripper", "kind": "fk_id", "link": link, "q": "c3_q", "qd": "c3_qd",
                         "qdd": "c3_qdd", "public_rows": 1024})
            gpu_out["config3"] = {k: host(t) for k, t in zip(("tau", "pos", "quat"), plan.outputs())}
        del plan, q, qd, qdd
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ config 4: Allegro, four fingertips, 65 536 rows
    hand = load("allegro_left", device)
    hand.own_kernels = "off"
    B = 65536
    q, _ = uniform_q(hand, B, device, 4004)
    with torch.no_grad():
        us_lib, us_lib_min = graph_launch_us(lambda: hand.compute_forward_kinematics_links(q, ALLEGRO_TIPS), 100)
    us, us_min, kernel4, own4, own4_path, own4_why = us_lib, us_lib_min, "drm::fk_fan_chain_kernel<8, 5, 4> (library, link-major outputs)", False, None, None
    try:      # the hand's own fan-out kernel (every finger's constants folded in, scalar chain walk; csrc/drm_arm_static.hpp): what a
        # plain model runs by default — the code object of this set of fingertips ships with the library
        hand_own = load("allegro_left", device)
        with torch.no_grad():
            hand_own.compute_forward_kinematics_links(q, ALLEGRO_TIPS)
        tips_idx = sorted(hand_own._name_to_idx_map[t] for t in ALLEGRO_TIPS)
        own4_path = "default" if hand_own._fan_own(tips_idx) else None
        if own4_path is None:
            hand_own.specialize()
            own4_path = "specialize()" if hand_own._fan_own(tips_idx) else None
        if own4_path:
            hand, own4 = hand_own, True
            with torch.no_grad():
                us, us_min = graph_launch_us(lambda: hand.compute_forward_kinematics_links(q, ALLEGRO_TIPS), 100)
            kernel4 = "drm_fk_fan_links_static (the hand's constants folded in; csrc/drm_arm_static.hpp)"
    except Exception as err:       # noqa: BLE001  (no hipcc: the library's kernel is the path)
        own4_why = str(err)[:200]
    with torch.no_grad():
        poses = hand.compute_forward_kinematics_links(q, ALLEGRO_TIPS)
    legs.append({"config": 4, "name": "config4", "workload": "Allegro hand 16-DoF (allegro_left), FK to the four fingertips through "
                 "compute_forward_kinematics_links (one launch, one wavefront per finger), batch %d, q~U(joint limits)" % B,
                 "batch": B, "launch_us": us, "launch_us_min": us_min, "evals_per_s": B / us * 1e6,
                 "own_kernel": own4, "own_kernel_path": own4_path, "own_kernel_unavailable": own4_why, "library_kernel_launch_us": us_lib,
                 "roofline": roofline(176, B, us, kernel4),
                 "io_floor_us": IO_FLOOR_US["config4"][0], "io_floor_source": IO_FLOOR_US["config4"][1]})
    ref_rows = 16384             # the reference walks all 21 links once per fingertip: ~4e4 evals/s on one core
    arrays["c4_q"] = host(q[:ref_rows])
    jobs.append({"name": "config4", "robot": "allegro_left", "kind": "fk_links", "links": ALLEGRO_TIPS, "q": "c4_q", "public_rows": 256})
    gpu_out["config4"] = {"pos": np.stack([host(poses[t][0]) for t in ALLEGRO_TIPS], 1),
                          "quat": np.stack([host(poses[t][1]) for t in ALLEGRO_TIPS], 1)}
    del poses

    # ------------------------------------------------------------------ config 5: iiwa, learnable link offset, 16 384 rows
    B = 16384
    torch.manual_seed(0)
    learn = load("iiwa7", device)
    learn.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(dim1=1, dim2=3))
    learn.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(dim1=1, dim2=3))
    body = learn._bodies[learn._name_to_idx_map["iiwa_link_1"]]
    init = {p: getattr(body, p).param.detach().clone() for p in ("trans", "rot_angles")}
    q, _ = uniform_q(iiwa, B, device, 5005)
    with torch.no_grad():
        want, _ = iiwa.compute_forward_kinematics(q, "iiwa_link_ee")
    ee = learn._name_to_idx_map["iiwa_link_ee"]
    dw = learn._get_walk(("fk", (ee,)), targets=[ee])
    ops_f = learn._ops_f(dw).detach()
    gpos = torch.randn(B, 1, 3, device=device)
    mask = learn._kinematic_param_mask(dw)
    us_f, _ = graph_launch_us(lambda: backend.fk(dw.program, ops_f, dw.ops_i, q, 1, 7), 50)
    us_b, _ = graph_launch_us(lambda: backend.fk_backward(dw.program, ops_f, dw.ops_i, q, gpos, 1, 7, mask, False), 50)
    leg = {"config": 5, "name": "config5", "batch": B, "workload": "Kuka iiwa, iiwa_link_1.trans and .rot_angles learnable "
           "(UnconstrainedTensor(1, 3), N(0, 0.1^2) under seed 0), batch %d, FK(EE position) + MSE + backward" % B,
           "fk_launch_us": us_f, "fk_backward_launch_us": us_b, "launch_us": us_f + us_b, "evals_per_s": B / (us_f + us_b) * 1e6,
           "roofline": roofline(96, B, us_f + us_b, "drm_fk (chain_fk_kernel) + drm_fk_backward (incl. its scratch memset / reduction)"),
           "note": "1.57 MB per step: launch-bound at this batch, not bandwidth-bound"}
    # the whole training step of the example (forward, loss, backward, Adam) eager and as ONE replayed hipGraph
    opt = torch.optim.Adam(learn.parameters(), lr=1e-3, capturable=True)

    def train_step():
        pos, _ = learn.compute_forward_kinematics(q, "iiwa_link_ee")
        loss = torch.nn.functional.mse_loss(pos, want)
        loss.backward()
        opt.step()
        return loss

    def eager_step():
        opt.zero_grad(set_to_none=True)
        train_step()

    for _ in range(5):
        eager_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eager_step()
    torch.cuda.synchronize()
    leg["eager_step_us"] = (time.perf_counter() - t0) / 50 * 1e6
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                train_step()
        torch.cuda.current_stream().wait_stream(side)
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            train_step()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                graph.replay()
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e) / 50 * 1e3)
        leg["graph_step_us"] = sorted(times)[len(times) // 2]
        leg["graph_step_evals_per_s"] = B / leg["graph_step_us"] * 1e6
        del graph
    except Exception as err:   # pragma: no cover - depends on the runtime
        leg["graph_step_error"] = repr(err)[:200]
    # the same step with the loss as ONE node (fk_mse_loss -> drm_fk_mse: forward kinematics, MSE and the gradients in one pass)
    try:
        with torch.no_grad():
            for p in ("trans", "rot_angles"):
                getattr(body, p).param.copy_(init[p])
        opt2 = torch.optim.Adam(learn.parameters(), lr=1e-3, capturable=True, fused=True)

        def fused_step():
            loss = learn.fk_mse_loss(q, "iiwa_link_ee", want)
            loss.backward()
            opt2.step()
            return loss

        us_m, _ = graph_launch_us(lambda: backend.fk_mse(dw.program, ops_f, dw.ops_i, q, want, 7, mask, False), 50)
        leg["fk_mse_launch_us"] = us_m
        # ABI 12: the same from the links' parameter tensors to their gradients (table build + table backward inside the two launches)
        links, base, sel = learn._learnable_plan(dw)
        pieces = [p.detach() for p in learn._learnable_pieces(links)]
        leg["fk_mse_links_launch_us"], _ = graph_launch_us(
            lambda: backend.fk_mse_links(dw.program, base, dw.ops_i, sel, dw.gsign, pieces, q, want, 7, mask, False), 50)
        leg["fk_mse_roofline"] = roofline(96, B, us_m, "drm_fk_mse: fk_backward_arm_kernel<8, 7, true> + fk_backward_reduce_kernel")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                opt2.zero_grad(set_to_none=True)
                fused_step()
        torch.cuda.current_stream().wait_stream(side)
        opt2.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fused_step()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                graph.replay()
            e.record()
            torch.cuda.synchronize()
            times.append(s.elapsed_time(e) / 50 * 1e3)
        leg["graph_step_fused_us"] = sorted(times)[len(times) // 2]
        leg["graph_step_fused_evals_per_s"] = B / leg["graph_step_fused_us"] * 1e6
        leg["graph_step_fused_note"] = ("the same training step with model.fk_mse_loss (ONE node: drm_fk_mse_links, two launches) and fused Adam, "
                                        "replayed from a hipGraph")
        del graph
    except Exception as err:   # pragma: no cover - depends on the runtime
        leg["graph_step_fused_error"] = repr(err)[:200]
    # gradients of the loss at the initial parameters (what the reference's autograd is asked for below)
    with torch.no_grad():
        for p in ("trans", "rot_angles"):
            getattr(body, p).param.copy_(init[p])
    learn.zero_grad(set_to_none=True)
    pos, _ = learn.compute_forward_kinematics(q, "iiwa_link_ee")
    loss = torch.nn.functional.mse_loss(pos, want)
    loss.backward()
    gpu_out["config5"] = {"loss": host(loss).reshape(1), "grad_trans": host(body.trans.param.grad),
                          "grad_rot_angles": host(body.rot_angles.param.grad)}
    arrays.update(c5_q=host(q), c5_want=host(want), c5_trans=host(init["trans"]), c5_rot=host(init["rot_angles"]))
    jobs.append({"name": "config5", "robot": "iiwa7", "kind": "learn_kinematics", "link": "iiwa_link_ee", "learn_link": "iiwa_link_1",
                 "q": "c5_q", "want": "c5_want", "init": {"trans": "c5_trans", "rot_angles": "c5_rot"}, "public_rows": 1024})
    legs.append(leg)

    # ------------------------------------------------------------------ eager public-API overhead
    eager = api_eager(panda, device)
    learn_dyn = learn_dynamics_step(device)

    # ------------------------------------------------------------------ the reference beside every leg (one subprocess)
    ref_meta = None
    if with_reference:
        rec, outs = run_reference(jobs, arrays)
        ref_meta = {k: v for k, v in rec.items() if k != "jobs"}
        by_name = {"config3_shard": "config3", "config3_whole": "config3"}   # (one reference job: a sample of the shard's rows)
        for leg in legs:
            job = rec.get("jobs", {}).get(by_name.get(leg["name"], leg["name"]))
            if job is None:
                continue
            leg["reference"] = job
            one = job["one_thread"]["tensor_only"]
            leg["cpu_baseline_of_record"] = {"evals_per_s": one["evals_per_s"], "threads": 1, "rows": one["rows"],
                                             "what": "unmodified reference, tensor-only, one thread"}
            speed = leg.get("graph_step_fused_evals_per_s", leg.get("graph_step_evals_per_s", leg["evals_per_s"]))
            leg["speedup_vs_reference_one_thread"] = speed / one["evals_per_s"]
            name = by_name.get(leg["name"], leg["name"])
            if outs and name in gpu_out and leg["name"] != "config3_whole":
                tol = dict(tau_relative=True)
                leg["gpu_vs_reference_max_abs"] = deviation(outs, name, gpu_out[name], **tol)
                leg["gpu_vs_reference_rows"] = job["outputs_rows"]
    if with_traffic and os.environ.get("DRM_BENCH_CHILD") != "1":
        torch.cuda.synchronize()
        fill_traffic(legs)
    return {"legs": legs, "api_eager_us_per_call": eager, "learn_dynamics_step": learn_dyn, "reference_run": ref_meta,
            "method": "launch_us: hipGraph of 30-100 launches, HIP events on the launch stream, median of 5 replays; reference: "
                      "oracle/ref_timing.py --jobs (unmodified reference, own interpreter, host cores only) in this run"}


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--pmc-child":
        sys.path.insert(0, ROOT)
        pmc_child(sys.argv[2])
    else:
        sys.exit("usage: bench_configs.py --pmc-child LEG   (a child mode of bench.py; run bench.py for the configs legs)")
