"""Second derivatives (`create_graph=True`) through the hand-written backward kernels: autograd._GradLaunch makes a
first-order gradient launch a differentiable node whose derivatives are directional differences of first-order launches.
Held to the UNMODIFIED reference's own double backward (tests/golden/golden_hvp.npz, written by make_golden_hvp.py from torch
autograd on the reference's CPU path): forward kinematics (robot_model.py:197-248), the end-effector Jacobian
(robot_model.py:626-667) and inverse dynamics (robot_model.py:305-375) — a 7-DoF arm (arm kernels), a hand (tree kernels)
and an arm with a gripper (arm + hand kernels).

Tolerance: 2e-3 of the result's scale — fp32 differences (h = 4e-2 and one Richardson step for the joint angles, a step of the
input's own size for qd / qdd) carry ~1e-4 of the first-order gradient's scale as noise (autograd._GradLaunch: error model);
the reference's own second derivatives are exact to fp32 rounding.
"""
import os

import numpy as np
import pytest
import torch

from helpers import load_model

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_hvp.npz")
ROBOTS = ["iiwa7", "panda_no_gripper", "allegro_left", "panda"]
TOL = 2e-3


def close(got, want, what):
    got, want = got.detach().cpu().numpy(), np.asarray(want)
    err = float(np.abs(got - want).max())
    assert err <= TOL * max(1.0, float(np.abs(want).max())), (what, err, float(np.abs(want).max()))


def case(g, robot, tag, n_x, n_w):
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    k = lambda name: g["%s/%s/%s" % (robot, tag, name)]
    ws = [dev(k("w%d" % i)).requires_grad_(True) for i in range(n_w)]
    return ws, [k("g%d" % i) for i in range(n_x)], [k("hvp%d" % i) for i in range(n_x)], [k("dsdw%d" % i) for i in range(n_w)]


def second(outputs, ws, xs, vs):
    L = sum((w * o).sum() for w, o in zip(ws, outputs))
    g = torch.autograd.grad(L, xs, create_graph=True)
    s = sum((v * gi).sum() for v, gi in zip(vs, g))
    h = torch.autograd.grad(s, list(xs) + list(ws))
    return g, h[:len(xs)], h[len(xs):]


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ROBOTS)
def test_second_derivatives_vs_the_reference_autograd(robot):
    g = np.load(GOLDEN)
    m = load_model(robot, "cuda", reference_compat=True)
    link = str(g[robot + "/link"])
    dev = lambda name: torch.from_numpy(np.ascontiguousarray(g["%s/%s" % (robot, name)])).cuda()
    q, qd, qdd = (dev(k).requires_grad_(True) for k in ("q", "qd", "qdd"))
    vq, vqd, vqdd = dev("vq"), dev("vqd"), dev("vqdd")
    # forward kinematics: pos and quat of the end link
    ws, g_ref, h_ref, dw_ref = case(g, robot, "fk", 1, 2)
    pos, quat = m.compute_forward_kinematics(q, link)
    got_g, got_h, got_dw = second((pos, quat), ws, (q,), (vq,))
    close(got_g[0], g_ref[0], "fk gradient")
    close(got_h[0], h_ref[0], "fk (d2L/dq2) v")
    for a, b, name in zip(got_dw, dw_ref, ("pos", "quat")):
        close(a, b, "fk J v, " + name)
    # the end-effector Jacobian
    ws, g_ref, h_ref, dw_ref = case(g, robot, "jac", 1, 2)
    lin, ang = m.compute_endeffector_jacobian(q, link)
    got_g, got_h, got_dw = second((lin, ang), ws, (q,), (vq,))
    close(got_g[0], g_ref[0], "jacobian gradient")
    close(got_h[0], h_ref[0], "jacobian (d2L/dq2) v")
    for a, b, name in zip(got_dw, dw_ref, ("lin", "ang")):
        close(a, b, "jacobian dJ v, " + name)
    # inverse dynamics: all three inputs
    ws, g_ref, h_ref, dw_ref = case(g, robot, "id", 3, 1)
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    got_g, got_h, got_dw = second((tau,), ws, (q, qd, qdd), (vq, vqd, vqdd))
    for i, name in enumerate(("q", "qd", "qdd")):
        close(got_g[i], g_ref[i], "inverse dynamics gradient, " + name)
        close(got_h[i], h_ref[i], "inverse dynamics Hessian-vector product, " + name)
    close(got_dw[0], dw_ref[0], "inverse dynamics J v")


@pytest.mark.gpu
def test_gradient_penalty_and_a_hessian_row_by_row():
    """The two things people do with create_graph=True: a penalty on the norm of an input gradient, backpropagated; and a
    full Hessian of a scalar of the end-effector position, here against its closed form for a chain of revolute joints,
    d2p/dq_i dq_j = z_i x (z_j x (p - p_j)) for i <= j."""
    m = load_model("iiwa7", "cuda")
    torch.manual_seed(3)
    B, n = 5, m._n_dofs
    q = (torch.rand(B, n, device="cuda") * 2 - 1).requires_grad_(True)
    w = torch.randn(B, 3, device="cuda")
    pos, _ = m.compute_forward_kinematics(q, "iiwa_link_ee")
    (g,) = torch.autograd.grad((w * pos).sum(), q, create_graph=True)
    penalty = (g.norm(dim=1) - 1.0).pow(2).mean()
    penalty.backward(retain_graph=True)
    assert q.grad is not None and torch.isfinite(q.grad).all() and float(q.grad.abs().max()) > 0
    # Hessian of w . p(q), one row per backward of a gradient component
    rows = []
    for i in range(n):
        (r,) = torch.autograd.grad(g[:, i].sum(), q, retain_graph=True)
        rows.append(r)
    H = torch.stack(rows, dim=1)                         # [B, n, n]
    assert float((H - H.transpose(1, 2)).abs().max()) <= 2e-3
    # closed form from the frames of the joints (all-links FK gives p_k and the rotation's z column)
    with torch.no_grad():
        poses = m.compute_forward_kinematics_all_links(q.detach())
    names = [b.name for b in m._bodies]
    joints = [names[i] for i in m._spec.controlled]
    P, Z = [], []
    for name in joints:
        p, quat = poses[name]
        x, y, z, ww = quat.unbind(-1)
        zcol = torch.stack([2 * (x * z + y * ww), 2 * (y * z - x * ww), 1 - 2 * (x * x + y * y)], dim=-1)
        sign = float(m._spec.axis_sign[m._name_to_idx_map[name]]) or 1.0
        axis = int(m._spec.axis_idx[m._name_to_idx_map[name]])
        assert axis == 2                                  # (iiwa: every joint about its local z)
        P.append(p); Z.append(sign * zcol)
    pe = pos.detach()
    Href = torch.zeros(B, n, n, device="cuda")
    for i in range(n):
        for j in range(i, n):
            v = torch.cross(Z[i], torch.cross(Z[j], pe - P[j], dim=-1), dim=-1)
            Href[:, i, j] = Href[:, j, i] = (w * v).sum(-1)
    assert float((H - Href).abs().max()) <= 2e-3 * max(1.0, float(Href.abs().max()))


@pytest.mark.gpu
def test_fast_hard_accelerating_states():
    """|qd| up to 50 rad/s, |qdd| up to 100 rad/s^2 (torques ~1e4 N m; golden_hvp.npz "panda_no_gripper_fast"): every input of
    the inverse-dynamics node is differenced with a step of its own size (autograd._GradLaunch), so the second derivatives
    keep the 2e-3 of the small-state cases instead of drowning in the rounding of 1e4-sized first-order launches."""
    g = np.load(GOLDEN)
    robot = "panda_no_gripper_fast"
    m = load_model("panda_no_gripper", "cuda")
    dev = lambda name: torch.from_numpy(np.ascontiguousarray(g["%s/%s" % (robot, name)])).cuda()
    q, qd, qdd = (dev(k).requires_grad_(True) for k in ("q", "qd", "qdd"))
    assert float(qd.abs().max()) > 30 and float(qdd.abs().max()) > 60
    ws, g_ref, h_ref, dw_ref = case(g, robot, "id", 3, 1)
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    got_g, got_h, got_dw = second((tau,), ws, (q, qd, qdd), (dev("vq"), dev("vqd"), dev("vqdd")))
    for i, name in enumerate(("q", "qd", "qdd")):
        close(got_g[i], g_ref[i], "gradient, " + name)
        close(got_h[i], h_ref[i], "Hessian-vector product, " + name)
    close(got_dw[0], dw_ref[0], "J v")


@pytest.mark.gpu
def test_create_graph_with_learnable_parameters_is_first_order_and_says_so():
    """Trainers that always pass create_graph=True (MAML-style inner loops, gradient penalties on the joint state) keep working
    on a model with learnable link parameters: the parameter gradients they get are the first-order ones; only a graph that
    really differentiates THROUGH a parameter gradient raises (second derivatives exist for the joint-state inputs and the
    output cotangents, INTEGRATION.md 'Second derivatives')."""
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    torch.manual_seed(0)
    m = load_model("iiwa7", "cuda")
    m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3))
    m.make_link_param_learnable("iiwa_link_3", "mass", UnconstrainedTensor(1, 1, init_tensor=torch.tensor([[3.0]])))
    params = list(m.parameters())
    q = (torch.rand(64, m._n_dofs, device="cuda") - 0.5).requires_grad_(True)
    qd = torch.rand(64, m._n_dofs, device="cuda") - 0.5

    def loss_of():
        pos, _ = m.compute_forward_kinematics(q, "iiwa_link_ee")
        lin, _ = m.compute_endeffector_jacobian(q, "iiwa_link_ee")
        tau = m.compute_inverse_dynamics(q, qd, qd)
        return pos.square().sum() + lin.square().sum() + 1e-2 * tau.square().sum()

    plain = torch.autograd.grad(loss_of(), params + [q])
    graph = torch.autograd.grad(loss_of(), params + [q], create_graph=True)
    for a, b in zip(plain, graph):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-6)
    # asking for the joint-state gradient alone (a gradient penalty) never touches the parameter path
    (gq,) = torch.autograd.grad(loss_of(), q, create_graph=True)
    gq.square().sum().backward()
    assert q.grad is not None and torch.isfinite(q.grad).all()
    # differentiating through a PARAMETER gradient is what does not exist: it raises (the table kernels' once-differentiable
    # backward, or autograd._FirstOrderOnly behind it) instead of contributing zero
    with pytest.raises((NotImplementedError, RuntimeError), match="Second derivatives|differentiate twice"):
        graph[0].sum().backward()
