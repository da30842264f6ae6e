"""Properties the domain offers, on the device="cpu" model (libdrm_cpu.so — the kernels' per-sample arithmetic) for every robot of
the package, sliding joints and skew axes included (reference_compat=False: the joint models the URDF states):

  * tau(q, qd, qdd) = H(q) qdd + nle(q, qd)                       (robot_model.py:305-450: the three methods agree)
  * forward dynamics inverts inverse dynamics (in torque space)    (robot_model.py:487-624 against 305-375)
  * H is symmetric positive definite
  * the linear Jacobian is the derivative of the link's position   (robot_model.py:626-667 against 223-248)
  * quaternions have unit norm and the pose of a link does not depend on the other links asked for in the same call
"""
import numpy as np
import pytest
import torch

from helpers import ALL_ROBOTS, load_model, sample_states


@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_dynamics_identities(robot, compat, cpu_library):
    m = load_model(robot, "cpu", reference_compat=compat)
    q, qd, qdd = (torch.from_numpy(a) for a in sample_states(m, 40, seed=11))
    for grav, damp in ((True, True), (False, False)):
        tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=grav, use_damping=damp)
        H = m.compute_lagrangian_inertia_matrix(q)
        nle = m.compute_non_linear_effects(q, qd, include_gravity=grav, use_damping=damp)
        want = torch.einsum("bij,bj->bi", H, qdd) + nle
        scale = 1.0 + tau.abs().max()
        assert float((tau - want).abs().max()) <= 2e-5 * float(scale), (robot, compat, grav)
        # (compared in TORQUE space: a hand's inertia matrix is ill-conditioned — light distal links — and fp32 accelerations of the
        # Allegro hand differ by percents between two correct evaluations, the torques they produce by 1e-4)
        back = m.compute_forward_dynamics(q, qd, tau, include_gravity=grav, use_damping=damp)
        again = m.compute_inverse_dynamics(q, qd, back, include_gravity=grav, use_damping=damp)
        assert float((again - tau).abs().max()) <= 1e-3 * float(scale), (robot, compat, grav)
    H = m.compute_lagrangian_inertia_matrix(q).double()
    assert torch.equal(H, H.transpose(1, 2))
    assert float(torch.linalg.eigvalsh(H).min()) > 0.0, robot


@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_the_jacobian_is_the_derivative_of_the_pose(robot, cpu_library):
    m = load_model(robot, "cpu", reference_compat=False)
    n = m._n_dofs
    q = torch.from_numpy(sample_states(m, 16, seed=3)[0])
    g = torch.Generator().manual_seed(2)
    dq = torch.rand(16, n, generator=g) - 0.5
    h = 1e-2
    for body in list(m._bodies)[1::3]:
        lin, ang = m.compute_endeffector_jacobian(q, body.name)
        pp, qp = m.compute_forward_kinematics(q + h * dq, body.name)
        pm, qm = m.compute_forward_kinematics(q - h * dq, body.name)
        fd = (pp - pm) / (2 * h)
        an = torch.einsum("bij,bj->bi", lin, dq)
        assert float((fd - an).abs().max()) < 2e-3 * (1.0 + float(an.abs().max())), (robot, body.name)   # (O(h^2) + fp32 / h)
        assert float((qp.norm(dim=1) - 1.0).abs().max()) < 1e-5
        # angular part: the rotation between the two poses is ~ 2 h (ang_jac . dq)
        w = torch.einsum("bij,bj->bi", ang, dq)
        conj = qm * torch.tensor([-1.0, -1.0, -1.0, 1.0])
        x1, y1, z1, w1 = qp.unbind(1); x2, y2, z2, w2 = conj.unbind(1)
        vec = torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                           w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], 1)
        sgn = torch.sign(w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2).unsqueeze(1)
        rot = 2.0 * vec * sgn / (2 * h)       # small-angle rotation vector per unit step
        assert float((rot - w).abs().max()) < 5e-3 * (1.0 + float(w.abs().max())), (robot, body.name)


@pytest.mark.parametrize("robot", ["panda", "allegro_left", "fetch", "iiwa7_allegro"])
def test_a_links_pose_does_not_depend_on_the_call_it_comes_from(robot, cpu_library):
    m = load_model(robot, "cpu")
    q = torch.from_numpy(sample_states(m, 70, seed=9)[0])
    poses = m.compute_forward_kinematics_all_links(q)
    for name in list(poses)[::4]:
        pos, quat = m.compute_forward_kinematics(q, name)
        assert float((poses[name][0] - pos).abs().max()) < 2e-6 and float((poses[name][1] - quat).abs().max()) < 2e-6, (robot, name)
