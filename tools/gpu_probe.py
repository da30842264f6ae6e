#!/usr/bin/env python3
"""Development probe run on the GPU box: eager kernel timings; provides `load` / `sample` to the other tools.  (Parity
against the oracle lives in tests/ only: nothing outside tests/, smoke() and bench.py's cpu_baseline touches oracle/.)"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import differentiable_robot_model_amd as drm  # noqa: E402
from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel, robot_description_folder  # noqa: E402


def load(name, device="cuda"):
    with contextlib.redirect_stdout(io.StringIO()):
        return DifferentiableRobotModel(os.path.join(robot_description_folder, name + ".urdf"), device=device)


def sample(model, B, seed=0):
    lim = model.get_joint_limits()
    lo = torch.tensor([j["lower"] for j in lim]); hi = torch.tensor([j["upper"] for j in lim])
    g = torch.Generator().manual_seed(seed)
    n = model._n_dofs
    return (lo + (hi - lo) * torch.rand(B, n, generator=g), torch.rand(B, n, generator=g) * 2 - 1,
            torch.rand(B, n, generator=g) * 4 - 2)


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def timing():
    from differentiable_robot_model_amd import backend
    m = load("panda_no_gripper")
    link = "panda_virtual_ee_link"
    print("device:", torch.cuda.get_device_name(0))
    for B in (65536, 1 << 20, 1 << 22):
        q, qd, qdd = (t.cuda() for t in sample(m, B))
        m.compute_fk_and_jacobian(q, link)
        dw = m._walks[("chain", m._name_to_idx_map[link])]
        ops_f = m._ops_f(dw)
        us = timeit(lambda: backend.fk_jacobian(dw.program, ops_f, dw.ops_i, q, 7))
        us_api = timeit(lambda: m.compute_fk_and_jacobian(q, link))
        print("fk_jac  B=%8d  %9.1f us/call (api %9.1f)  %7.2f Mevals/s  %6.1f GB/s algorithmic" %
              (B, us, us_api, B / us, B * 224 / us / 1e3), flush=True)
        m.compute_inverse_dynamics(q, qd, qdd)
        dt = m._get_walk(("tree",), whole_tree=True)
        us = timeit(lambda: backend.rnea(dt.program, m._ops_f(dt), dt.ops_i, q, qd, qdd, True, True, 7))
        print("rnea    B=%8d  %9.1f us/call  %7.2f Mevals/s  %6.1f GB/s algorithmic" %
              (B, us, B / us, B * 112 / us / 1e3), flush=True)
        us = timeit(lambda: m.compute_forward_kinematics(q, link))
        print("fk      B=%8d  %9.1f us/call  %7.2f Mevals/s  %6.1f GB/s algorithmic" %
              (B, us, B / us, B * 56 / us / 1e3), flush=True)
    m = load("allegro_left")
    q, _, _ = (t.cuda() for t in sample(m, 65536))
    tips = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]
    idx = [m._name_to_idx_map[t] for t in tips]
    us = timeit(lambda: m._fk_targets(q, idx))
    print("allegro 4-tip fk B=65536 %9.1f us/call  %6.1f GB/s algorithmic" % (us, 65536 * 176 / us / 1e3))


if __name__ == "__main__":
    t0 = time.time()
    import __graft_entry__
    __graft_entry__.smoke()
    timing()
    print("probe done in %.1fs" % (time.time() - t0))
