#!/usr/bin/env python3
"""Host-side cost of one eager API call (small batch: the kernel is ~3 us, the rest is Python + ctypes + allocation)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample

m = load("panda_no_gripper"); link = "panda_virtual_ee_link"
for B in (64, 4096):
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    calls = {
        "compute_forward_kinematics": lambda: m.compute_forward_kinematics(q, link),
        "compute_endeffector_jacobian": lambda: m.compute_endeffector_jacobian(q, link),
        "compute_inverse_dynamics": lambda: m.compute_inverse_dynamics(q, qd, qdd),
        "compute_lagrangian_inertia_matrix": lambda: m.compute_lagrangian_inertia_matrix(q),
        "compute_forward_dynamics": lambda: m.compute_forward_dynamics(q, qd, qdd),
        "plan.launch (prepared)": m.plan_fk_and_jacobian(q, link).launch,
    }
    for rep in range(2):      # the first pass only warms the host (clocks, allocator, code paths); the second is reported
        for name, fn in calls.items():
            for _ in range(50): fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2000): fn()
            torch.cuda.synchronize()
            if rep:
                print("B=%5d  %-36s %7.1f us per call" % (B, name, (time.perf_counter() - t0) / 2000 * 1e6))
