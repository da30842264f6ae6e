#!/usr/bin/env python3
"""drm_rnea_backward of an arm WITH learnable link parameters: the library's table-driven kernel against the arm's own kernel for
that set of learnable blocks (constant blocks folded in, csrc/drm_arm_static.hpp) — hipGraph of K launches, HIP events, median of 5.

    python tools/ab_learnable_arm.py [robot]        (default iiwa7)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench_configs import graph_launch_us  # noqa: E402
from differentiable_robot_model_amd import backend  # noqa: E402
from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, UnconstrainedTensor  # noqa: E402
from helpers import load_model, sample_states  # noqa: E402

robot = sys.argv[1] if len(sys.argv) > 1 else "iiwa7"
prefix = {"iiwa7": "iiwa_link_%d", "panda_no_gripper": "panda_link%d"}[robot]
PLANS = {"one link (mass, com, inertia_mat)": [(prefix % 4, p) for p in ("mass", "com", "inertia_mat")],
         "all 7 links (mass, com, inertia_mat)": [(prefix % k, p) for k in range(1, 8) for p in ("mass", "com", "inertia_mat")],
         "one link, trans": [(prefix % 3, "trans")]}
shapes = {"com": (1, 3), "inertia_mat": (3, 3), "trans": (1, 3)}


def model(plan, own):
    torch.manual_seed(0)
    m = load_model(robot, "cuda")
    for link, pname in plan:
        m.make_link_param_learnable(link, pname, PositiveScalar() if pname == "mass" else UnconstrainedTensor(dim1=shapes[pname][0], dim2=shapes[pname][1]))
    if own:
        m.specialize()
    else:
        m.own_kernels = "off"
    return m


mc = load_model(robot)
for B in (65536, 1 << 20):
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, B, seed=1, vel=0.5, acc=1.0))
    gt = torch.randn(B, 7, device="cuda")
    for name, plan in PLANS.items():
        for inputs in (True, False):
            row = []
            for own in (False, True):
                m = model(plan, own)
                dw = m._dynamics_walk()
                ops_f, mask = m._ops_f(dw).detach(), m._learnable_op_mask(dw)
                us, _ = graph_launch_us(lambda: backend.rnea_backward(dw.program, ops_f, dw.ops_i, q, qd, qdd, gt, True, True, 7, mask, inputs), 20 if B > 100000 else 50)
                row.append(us)
            print("%-18s B=%8d  %-40s %-22s library %7.2f us   own %7.2f us   x%.2f" % (
                robot, B, name, "+ input gradients" if inputs else "parameters only", row[0], row[1], row[0] / row[1]), flush=True)
