#!/usr/bin/env python3
"""TEST / BASELINE INFRASTRUCTURE ONLY — times the UNMODIFIED reference (staged by oracle/stage_ref.py into
oracle/_ref/) on this machine's host cores.  Run as a SUBPROCESS of bench.py's cpu_baseline leg (its own interpreter:
the bench process never imports the reference, and torch's thread settings here do not leak into it).

    python oracle/ref_timing.py --q q.npy [--qd qd.npy --qdd qdd.npy] --robot panda_no_gripper --link panda_virtual_ee_link \
        --public-rows 4096 --out-npz ref_out.npz

Prints ONE JSON object:
  (A) the full public API: `compute_endeffector_jacobian(q, link)` (robot_model.py:626-667) — which runs
      `compute_forward_kinematics` and with it the per-sample Python loop of `CoordinateTransform.get_quaternion`
      (spatial_vector_algebra.py:108-136) — on the first `--public-rows` rows (the loop is O(B): ~80 us per sample, so
      65 536 rows take ~7 s per call; the extrapolation to the full batch is linear and stated in the output);
  (B) "tensor only": the same call on ALL rows with `get_quaternion` replaced by a stub that returns zeros, i.e. the
      reference's vectorised math without its Python-loop artefact;
  each with every host core torch will use and with 1 thread, min of `--reps` after one warm-up call.
  With --qd/--qdd: `compute_inverse_dynamics` (robot_model.py:305-375) on all rows as well (BASELINE configuration 3).
--jobs jobs.json --arrays arrays.npz (bench.py's "configs" legs, BASELINE.json configurations 2-5): a list of jobs
  {"name", "robot", "kind": "jacobian" | "fk_id" | "fk_links" | "learn_kinematics", "link"/"links", array keys, "public_rows"};
  every job is timed like the above (public API on a bounded number of rows AND tensor-only on all rows, all threads AND one
  thread) and its outputs on the first rows go to --out-npz as "<name>/<key>" for the caller's deviation report.
--out-npz stores the reference's outputs of (A) (pos, quat, lin_jac, ang_jac on the first public rows, tau if asked) so the
caller can report the GPU path's deviation from the reference in the same run.
"""
import argparse
import json
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
SHIM = os.path.join(os.path.dirname(HERE), "tests", "golden", "_shim")   # xml.etree stand-in for urdf_parser_py (absent here)

URDF = {"panda_no_gripper": "panda_description/urdf/panda_no_gripper.urdf",
        "iiwa7": "kuka_iiwa/urdf/iiwa7.urdf",
        "allegro_left": "allegro/urdf/allegro_hand_description_left.urdf"}
MAX_THREADS_PROBED = 16     # "all threads" legs: torch's default pool (every core of a 128-core host) is SLOWER than one thread on
                            # these [B, 3]-sized ops (round 3: 2.2e5 vs 1.24e6 evals/s); the many-thread figure is taken at <= 16


def best_of(fn, reps):
    fn()                                  # warm-up (allocator, lazy inits)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return min(times)


def run_jobs(args):
    """bench.py's `configs` legs: every BASELINE.json configuration through the unmodified reference on this host."""
    import contextlib
    import io

    import numpy as np
    import torch

    import differentiable_robot_model.robot_model as rm
    import differentiable_robot_model.spatial_vector_algebra as sva
    from differentiable_robot_model.rigid_body_params import UnconstrainedTensor
    with open(args.jobs) as f:
        jobs = json.load(f)
    arrays = np.load(args.arrays)
    real_quat = sva.CoordinateTransform.get_quaternion

    def stub_quat(self):
        return torch.zeros(self._rot.shape[0], 4)

    def load(robot):
        with contextlib.redirect_stdout(io.StringIO()):
            return rm.DifferentiableRobotModel(os.path.join(REF, "diff_robot_data", URDF[robot]), device="cpu")

    all_threads = torch.get_num_threads()
    many = min(all_threads, MAX_THREADS_PROBED)
    out = {"kind": "reference", "where": "this run", "host_cores": os.cpu_count(), "torch": torch.__version__,
           "threads_default": all_threads, "threads_many": many, "jobs": {}}
    saved = {}
    for job in jobs:
        name, kind = job["name"], job["kind"]
        t = lambda key: torch.from_numpy(arrays[job[key]].astype(np.float32))
        q = t("q")
        B, rows = q.shape[0], min(int(job.get("public_rows", 1024)), q.shape[0])
        model = load(job["robot"])
        rec = {"robot": job["robot"], "kind": kind, "batch": B}
        if kind == "jacobian":
            call = lambda x: model.compute_endeffector_jacobian(x, job["link"])
            what = "compute_endeffector_jacobian"
        elif kind == "fk_links":
            call = lambda x: [model.compute_forward_kinematics(x, link) for link in job["links"]]
            what = "compute_forward_kinematics, once per link (%d links)" % len(job["links"])
        elif kind == "fk_id":
            qd, qdd = t("qd"), t("qdd")
            call = None
            what = "compute_inverse_dynamics + compute_forward_kinematics"
        elif kind == "learn_kinematics":
            want = t("want")
            what = "learn_kinematics_of_iiwa.py step: zero_grad, compute_forward_kinematics, MSE, backward, Adam step"
        else:
            raise ValueError(kind)
        rec["what"] = what

        def fk_id(n_rows):
            tau = model.compute_inverse_dynamics(q[:n_rows], qd[:n_rows], qdd[:n_rows], include_gravity=True, use_damping=True)
            pos, quat = model.compute_forward_kinematics(q[:n_rows], job["link"])
            return tau, pos, quat

        if kind == "learn_kinematics":
            inits = {p: torch.from_numpy(arrays[job["init"][p]].astype(np.float32)) for p in ("trans", "rot_angles")}
            for p in ("trans", "rot_angles"):
                model.make_link_param_learnable(job["learn_link"], p, UnconstrainedTensor(dim1=1, dim2=3, init_tensor=inits[p].clone()))
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)

            def train_step(n_rows, apply=True):
                opt.zero_grad()
                pos, _ = model.compute_forward_kinematics(q[:n_rows], job["link"])
                loss = torch.nn.functional.mse_loss(pos, want[:n_rows])
                loss.backward()
                if apply:
                    opt.step()
                return loss

            def reset():
                body = model._bodies[model._name_to_idx_map[job["learn_link"]]]
                with torch.no_grad():
                    for p in ("trans", "rot_angles"):
                        getattr(body, p).param.copy_(inits[p])

        for label, threads in (("one_thread", 1), ("many_threads", many)):
            torch.set_num_threads(threads)
            leg = {"threads": threads}
            for mode, n_rows, reps in (("public_api", rows, 1), ("tensor_only", B, args.reps)):
                sva.CoordinateTransform.get_quaternion = real_quat if mode == "public_api" else stub_quat
                if kind == "learn_kinematics":
                    secs = best_of(lambda: train_step(n_rows), reps)
                    reset()
                else:
                    with torch.no_grad():
                        secs = best_of((lambda: fk_id(n_rows)) if kind == "fk_id" else (lambda: call(q[:n_rows])), reps)
                leg[mode] = {"rows": n_rows, "seconds": secs, "evals_per_s": n_rows / secs}
            rec[label] = leg
        sva.CoordinateTransform.get_quaternion = real_quat
        torch.set_num_threads(all_threads)
        rec["note"] = ("public_api = the reference's method as shipped (per-sample Python quaternion loop, linear in rows) on the "
                       "first `rows` rows; tensor_only = the same call on ALL rows with get_quaternion stubbed to zeros")
        # outputs on the first rows (the real quaternion path) for the caller's deviation report
        if kind == "jacobian":
            with torch.no_grad():
                lin, ang = model.compute_endeffector_jacobian(q[:rows], job["link"])
                pos, quat = model.compute_forward_kinematics(q[:rows], job["link"])
            outs = dict(pos=pos, quat=quat, lin_jac=lin, ang_jac=ang)
        elif kind == "fk_links":
            with torch.no_grad():
                pq = [model.compute_forward_kinematics(q[:rows], link) for link in job["links"]]
            outs = dict(pos=torch.stack([p for p, _ in pq], 1), quat=torch.stack([r for _, r in pq], 1))
        elif kind == "fk_id":
            with torch.no_grad():
                tau, pos, quat = fk_id(rows)
            outs = dict(tau=tau, pos=pos, quat=quat)
        else:   # gradients of the whole-batch loss (the quaternion does not enter it: stubbed, so all rows in ~0.1 s)
            sva.CoordinateTransform.get_quaternion = stub_quat
            reset()
            loss = train_step(B, apply=False)
            sva.CoordinateTransform.get_quaternion = real_quat
            body = model._bodies[model._name_to_idx_map[job["learn_link"]]]
            outs = dict(loss=loss.detach().reshape(1), grad_trans=body.trans.param.grad, grad_rot_angles=body.rot_angles.param.grad)
        for k, v in outs.items():
            saved["%s/%s" % (name, k)] = v.detach().numpy()
        rec["outputs_rows"] = B if kind == "learn_kinematics" else rows
        out["jobs"][name] = rec
    if args.out_npz:
        np.savez(args.out_npz, **saved)
    print(json.dumps(out))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--q")
    ap.add_argument("--jobs")
    ap.add_argument("--arrays")
    ap.add_argument("--qd")
    ap.add_argument("--qdd")
    ap.add_argument("--robot", default="panda_no_gripper", choices=sorted(URDF))
    ap.add_argument("--link", default="panda_virtual_ee_link")
    ap.add_argument("--public-rows", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out-npz")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "differentiable_robot_model")):
        print(json.dumps({"error": "oracle/_ref/ is not staged (run oracle/stage_ref.py where /root/reference exists)"}))
        return 0
    warnings.filterwarnings("ignore")
    import numpy as np
    import torch
    try:
        import urdf_parser_py.urdf  # noqa: F401
    except ImportError:
        sys.path.insert(0, SHIM)
    sys.path.insert(0, REF)
    import contextlib
    import io

    import differentiable_robot_model.robot_model as rm          # the reference itself
    import differentiable_robot_model.spatial_vector_algebra as sva
    assert os.path.realpath(rm.__file__).startswith(os.path.realpath(REF)), rm.__file__
    if args.jobs:
        return run_jobs(args)

    q = torch.from_numpy(np.load(args.q).astype(np.float32))
    with contextlib.redirect_stdout(io.StringIO()):
        model = rm.DifferentiableRobotModel(os.path.join(REF, "diff_robot_data", URDF[args.robot]), device="cpu")
    B, rows = q.shape[0], min(args.public_rows, q.shape[0])
    all_threads = torch.get_num_threads()
    out = {"kind": "reference", "where": "this run", "robot": args.robot, "link": args.link, "batch": B,
           "host_cores": os.cpu_count(), "torch": torch.__version__, "reps": args.reps, "threads_all": all_threads,
           "source": "unmodified reference staged from /root/reference by oracle/stage_ref.py (oracle/_ref/, git-ignored)"}

    real_quat = sva.CoordinateTransform.get_quaternion

    def stub_quat(self):                                        # (B): the O(B) Python loop removed, nothing else
        return torch.zeros(self._rot.shape[0], 4)

    def jac(x):
        with torch.no_grad():
            return model.compute_endeffector_jacobian(x, args.link)

    res = {}
    for label, threads in (("all_threads", all_threads), ("one_thread", 1)):
        torch.set_num_threads(threads)
        sva.CoordinateTransform.get_quaternion = real_quat
        t_pub = best_of(lambda: jac(q[:rows]), max(1, args.reps - 1))
        sva.CoordinateTransform.get_quaternion = stub_quat
        t_ten = best_of(lambda: jac(q), args.reps)
        res[label] = {"threads": threads,
                      "public_api": {"rows": rows, "seconds": t_pub, "evals_per_s": rows / t_pub,
                                     "extrapolated_seconds_at_batch": t_pub * B / rows,
                                     "note": "compute_endeffector_jacobian incl. the per-sample Python quaternion loop; "
                                             "linear in rows, so evals/s at the full batch is the same figure"},
                      "tensor_only": {"rows": B, "seconds": t_ten, "evals_per_s": B / t_ten,
                                      "note": "get_quaternion stubbed (returns zeros), everything else unmodified"}}
        if args.qd and args.qdd:
            qd = torch.from_numpy(np.load(args.qd).astype(np.float32))
            qdd = torch.from_numpy(np.load(args.qdd).astype(np.float32))
            with torch.no_grad():
                t_id = best_of(lambda: model.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True),
                               args.reps)
            res[label]["inverse_dynamics"] = {"rows": B, "seconds": t_id, "evals_per_s": B / t_id}
    out.update(res)
    sva.CoordinateTransform.get_quaternion = real_quat
    torch.set_num_threads(all_threads)
    if args.out_npz:
        with torch.no_grad():
            lin, ang = model.compute_endeffector_jacobian(q[:rows], args.link)
            pos, quat = model.compute_forward_kinematics(q[:rows], args.link)
            save = dict(pos=pos.numpy(), quat=quat.numpy(), lin_jac=lin.numpy(), ang_jac=ang.numpy())
            if args.qd and args.qdd:
                save["tau"] = model.compute_inverse_dynamics(q[:rows], qd[:rows], qdd[:rows], include_gravity=True,
                                                             use_damping=True).numpy()
        np.savez(args.out_npz, **save)
        out["outputs_npz_rows"] = rows
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
