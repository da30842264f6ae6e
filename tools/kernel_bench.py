#!/usr/bin/env python3
"""Launch the metric kernel (Panda FK+Jacobian) back to back at a few batch sizes — for rocprofv3 runs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample  # noqa: E402
from differentiable_robot_model_amd import backend  # noqa: E402

m = load("panda_no_gripper")
link = "panda_virtual_ee_link"
sizes = [int(a) for a in sys.argv[1:]] or [64, 65536, 1 << 20]
for B in sizes:
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    m.compute_fk_and_jacobian(q, link)
    dw = m._walks[("chain", m._name_to_idx_map[link])]
    ops_f = m._ops_f(dw)
    for _ in range(30):
        backend.fk_jacobian(dw.program, ops_f, dw.ops_i, q, 7)
    m.compute_inverse_dynamics(q, qd, qdd)
    dt = m._dynamics_walk()
    for _ in range(30):
        backend.rnea(dt.program, m._ops_f(dt), dt.ops_i, q, qd, qdd, True, True, 7)
    torch.cuda.synchronize()
print("done")
