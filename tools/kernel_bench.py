#!/usr/bin/env python3
"""Launch a group of kernels a few times — the workload of the rocprofv3 counter runs (tools/profile_round.sh).

    kernel_bench.py hot [B ...]          Panda FK + Jacobian (the metric kernel), RNEA, the fused FK + RNEA launch
    kernel_bench.py configs              BASELINE configurations 2-5 at their sizes (iiwa FK + Jacobian 65 536; Panda fused FK +
                                         RNEA 131 072 and 2^20; Allegro four fingertips 65 536; iiwa FK + backward / fused MSE 16 384)
    kernel_bench.py secondary [B]        Allegro fingertips, Panda mass matrix / forward dynamics / RNEA backward
    kernel_bench.py hand [B]             the Allegro hand's FK, dynamics and RNEA backward kernels
    kernel_bench.py dynamics ROBOT [B]   RNEA, mass matrix, forward dynamics of one robot
    kernel_bench.py backward ROBOT [B]   its inverse-dynamics backward (input gradients)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample  # noqa: E402
from differentiable_robot_model_amd import backend  # noqa: E402

TIPS = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]


def hot(sizes):
    m = load("panda_no_gripper"); link = "panda_virtual_ee_link"
    for B in sizes or [64, 65536, 1 << 20]:
        q, qd, qdd = (t.cuda() for t in sample(m, B))
        plans = [m.plan_fk_and_jacobian(q, link), m.plan_inverse_dynamics(q, qd, qdd), m.plan_fk_and_inverse_dynamics(q, qd, qdd, link)]
        for plan in plans:
            for _ in range(30):
                plan.launch()
        torch.cuda.synchronize()


def configs():
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    iiwa = load("iiwa7")
    q = sample(iiwa, 65536)[0].cuda()
    plan = iiwa.plan_fk_and_jacobian(q, "iiwa_link_ee")
    for _ in range(30):
        plan.launch()
    panda = load("panda_no_gripper")
    for B in (131072, 1 << 20):
        q, qd, qdd = (t.cuda() for t in sample(panda, B))
        plan = panda.plan_fk_and_inverse_dynamics(q, qd, qdd, "panda_virtual_ee_link")
        for _ in range(20):
            plan.launch()
    hand = load("allegro_left")
    q = sample(hand, 65536)[0].cuda()
    with torch.no_grad():
        for _ in range(30):
            hand.compute_forward_kinematics_links(q, TIPS)
    learn = load("iiwa7")
    learn.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3))
    learn.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(1, 3))
    q = sample(iiwa, 16384)[0].cuda()
    with torch.no_grad():
        want, _ = iiwa.compute_forward_kinematics(q, "iiwa_link_ee")
    for _ in range(20):
        learn.zero_grad()
        torch.nn.functional.mse_loss(learn.compute_forward_kinematics(q, "iiwa_link_ee")[0], want).backward()
        learn.zero_grad()
        learn.fk_mse_loss(q, "iiwa_link_ee", want).backward()
    torch.cuda.synchronize()


def secondary(B):
    ma = load("allegro_left")
    qa = sample(ma, B)[0].cuda()
    tips = [ma._name_to_idx_map[t] for t in TIPS]
    m = load("panda_no_gripper")
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    for _ in range(10):
        ma._fk_targets(qa, tips)
        m.compute_lagrangian_inertia_matrix(q)
        m.compute_forward_dynamics(q, qd, qdd)
    qg = q.clone().requires_grad_(True)
    for _ in range(5):
        m.compute_inverse_dynamics(qg, qd, qdd).sum().backward()
    torch.cuda.synchronize()


def hand(B):
    ma = load("allegro_left")
    qa, qda, qdda = (t.cuda() for t in sample(ma, B))
    tips = [ma._name_to_idx_map[t] for t in TIPS]
    for _ in range(5):
        ma._fk_targets(qa, tips)
        ma.compute_inverse_dynamics(qa, qda, qdda)
        ma.compute_lagrangian_inertia_matrix(qa)
        ma.compute_forward_dynamics(qa, qda, qdda)
    full = ma._get_walk(("tree",), whole_tree=True)
    of = ma._ops_f(full)
    gt = torch.randn(B, ma._n_dofs, device="cuda")
    for _ in range(3):   # the RNEA backward fanned out over the fingers: input gradients, then with one learnable link
        backend.rnea_backward(full.program, of, full.ops_i, qa, qda, qdda, gt, True, True, ma._n_dofs, 0, True)
        backend.rnea_backward(full.program, of, full.ops_i, qa, qda, qdda, gt, True, True, ma._n_dofs, 1 << 5, True)
    torch.cuda.synchronize()


def dynamics(robot, B):
    m = load(robot)
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    for _ in range(4):
        m.compute_inverse_dynamics(q, qd, qdd)
        m.compute_lagrangian_inertia_matrix(q)
        m.compute_forward_dynamics(q, qd, qdd)
    torch.cuda.synchronize()


def backward(robot, B):
    m = load(robot)
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    dw = m._dynamics_walk()
    of, gt = m._ops_f(dw), torch.randn(B, m._n_dofs, device="cuda")
    for _ in range(4):
        backend.rnea_backward(dw.program, of, dw.ops_i, q, qd, qdd, gt, True, True, m._n_dofs, 0, True)
    torch.cuda.synchronize()


def main(argv):
    what, rest = (argv[0] if argv else "hot"), argv[1:]
    if what == "hot":
        hot([int(a) for a in rest])
    elif what == "configs":
        configs()
    elif what in ("secondary", "hand"):
        {"secondary": secondary, "hand": hand}[what](int(rest[0]) if rest else 65536)
    elif what in ("dynamics", "backward"):
        {"dynamics": dynamics, "backward": backward}[what](rest[0] if rest else "panda", int(rest[1]) if len(rest) > 1 else 1 << 18)
    else:
        sys.exit(__doc__)
    print("done")


if __name__ == "__main__":
    main(sys.argv[1:])
