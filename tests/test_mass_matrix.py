"""Joint-space inertia matrix (K6, csrc/drm_crba.hip + drm_sample.hpp::crba_walk).

CPU (not gpu): the oracle's restatement of the reference's n + 1 inverse-dynamics construction
(robot_model.py:402-450) against matrices recorded from the UNMODIFIED reference (tests/golden/golden_mass.npz,
made by tests/golden/make_golden_mass.py), and the kernel arithmetic (host emulation) against the fp64 oracle for
every shipped robot.  GPU (-m gpu): the real kernel through the public API, same checks + properties at full size.

Tolerances: the reference builds H by subtracting two fp32 inverse-dynamics results that both carry the gravity
torques (tens of N m), so its own H is only symmetric to ~2e-5 (measured, make_golden_mass.py); against the
reference we therefore hold atol 5e-5, against the fp64 oracle the usual tau tolerance (2e-5 abs / rel).
"""
import ctypes

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.flatten import build_walk
from helpers import ALL_ROBOTS, GOLDEN_ROBOTS, TOL_TAU, load_golden, load_golden_mass, load_model, sample_states
from oracle import Oracle
from test_host_emu import _ptr, emu, host_walk  # noqa: F401  (emu is a fixture)

TOL_H_REF = dict(atol=5e-5, rtol=2e-5)


@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_oracle_mass_matrix_vs_reference(robot, links):
    g, gm = load_golden(robot), load_golden_mass()
    m = load_model(robot)
    q = g["fast_q"]
    for tag, grav, damp in (("g1_d1", True, True), ("g0_d0", False, False)):
        H32 = Oracle(m._spec).mass_matrix(q, grav, damp, np.float32)
        H64 = Oracle(m._spec).mass_matrix(q.astype(np.float64), grav, damp, np.float64)
        ref = gm["%s/H_%s" % (robot, tag)]
        assert np.allclose(H32, ref, **TOL_H_REF), np.abs(H32 - ref).max()
        assert np.allclose(H64, ref, **TOL_H_REF), np.abs(H64 - ref).max()
    # the flags cancel out of the construction (what lets the kernel drop them)
    a = Oracle(m._spec).mass_matrix(q.astype(np.float64), True, True, np.float64)
    b = Oracle(m._spec).mass_matrix(q.astype(np.float64), False, False, np.float64)
    assert np.abs(a - b).max() < 1e-9


@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_emu_crba_vs_oracle(emu, robot):
    m = load_model(robot)
    n, B = m._n_dofs, 19
    q, _, _ = sample_states(m, B, seed=41)
    prog = build_walk(m._spec, whole_tree=True)
    walk, keep = host_walk(m, prog)
    H = np.full((B, n, n), np.nan, np.float32)
    assert emu.emu_crba(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(H)) == 0
    ref = Oracle(m._spec).mass_matrix(q.astype(np.float64), False, False, np.float64)
    assert np.allclose(H, ref, **TOL_TAU), (robot, np.abs(H - ref).max())
    assert np.array_equal(H, H.transpose(0, 2, 1)), "CRBA fills both triangles with the same value"
    assert np.linalg.eigvalsh(H.astype(np.float64)).min() > 0


@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "fetch_arm_no_gripper"])
def test_emu_crba_arm_chain_vs_oracle(emu, robot):
    """crba_chain (the arithmetic of crba_arm_kernel<8, 7>) against the fp64 oracle."""
    m = load_model(robot)
    n, B = m._n_dofs, 37
    q, _, _ = sample_states(m, B, seed=43)
    q[2, 5] = 4.0e5   # fp64 sincos fallback
    prog = build_walk(m._spec, whole_tree=True)
    walk, keep = host_walk(m, prog)
    H = np.full((B, n, n), np.nan, np.float32)
    assert emu.emu_crba_arm(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(H)) == 0
    ref = Oracle(m._spec).mass_matrix(q.astype(np.float64), False, False, np.float64)
    assert np.allclose(H, ref, **TOL_TAU), (robot, np.abs(H - ref).max())
    assert np.array_equal(H, H.transpose(0, 2, 1))


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("robot", ALL_ROBOTS)
@pytest.mark.parametrize("B", [1, 64, 130])
def test_gpu_crba_vs_oracle(robot, B):
    m = load_model(robot, "cuda")
    q, _, _ = sample_states(m, B, seed=50 + B)
    H = m.compute_lagrangian_inertia_matrix(torch.from_numpy(q).cuda()).cpu().numpy()
    ref = Oracle(m._spec).mass_matrix(q.astype(np.float64), False, False, np.float64)
    assert H.shape == (B, m._n_dofs, m._n_dofs)
    assert np.allclose(H, ref, **TOL_TAU), (robot, np.abs(H - ref).max())
    assert np.array_equal(H, H.transpose(0, 2, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("robot,compat", [("panda", True), ("panda", False), ("jaco", True), ("trifinger_edu", True), ("allegro_left", True)])
def test_gpu_crba_throughput_form_of_the_small_shapes(robot, compat):
    """Launches of >= 2 048 tiles of a small arm + hand / hand (<= 12 ops) take the one-wavefront-per-tile walk over the shape's tree
    (csrc/drm_static.hpp crba_shape_body; sliding fingers included); smaller launches the wavefront-per-sub-chain kernels.  Every
    row of a 131 072 + 70 row launch against the fp64 oracle, symmetric, and equal to rounding to the same rows in small launches
    (the Allegro hand, 16 ops, keeps its kernel at every size: the two must agree there bit for bit)."""
    mc, m = load_model(robot, reference_compat=compat), load_model(robot, "cuda", reference_compat=compat)
    m.own_kernels = "off"       # (the library's two forms; the Panda and the Jaco ship their own inertia-matrix kernel, round 6)
    B = 2048 * 64 + 70
    q, _, _ = sample_states(mc, B, seed=3)
    dq = torch.from_numpy(q).cuda()
    H = m.compute_lagrangian_inertia_matrix(dq)
    small = torch.cat([m.compute_lagrangian_inertia_matrix(dq[a:a + 8192]) for a in range(0, B, 8192)])
    ref = Oracle(mc._spec).mass_matrix(q[:16384].astype(np.float64), False, False, np.float64)
    assert np.allclose(H[:16384].cpu().numpy(), ref, **TOL_TAU), (robot, np.abs(H[:16384].cpu().numpy() - ref).max())
    assert torch.equal(H, H.transpose(1, 2))
    if robot == "allegro_left":
        assert torch.equal(H, small)
    else:
        assert float((H - small).abs().max()) <= 2e-5 * float(small.abs().max())
        assert not torch.equal(H[:64], small[:64]) or robot == "trifinger_edu", "the large launch took the other kernel"


@pytest.mark.gpu
@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_gpu_crba_vs_reference_golden(robot, links):
    g, gm = load_golden(robot), load_golden_mass()
    m = load_model(robot, "cuda")
    q = torch.from_numpy(g["fast_q"]).cuda()
    for tag, grav, damp in (("g1_d1", True, True), ("g0_d0", False, False)):
        H = m.compute_lagrangian_inertia_matrix(q, include_gravity=grav, use_damping=damp).cpu().numpy()
        assert np.allclose(H, gm["%s/H_%s" % (robot, tag)], **TOL_H_REF)
    H1 = m.compute_lagrangian_inertia_matrix(q[0])          # unbatched call (tensor_check strips the batch dim)
    assert tuple(H1.shape) == (m._n_dofs, m._n_dofs)


@pytest.mark.gpu
def test_gpu_crba_full_size_consistent_with_rnea():
    """H qdd = ID(q, 0, qdd) - ID(q, 0, 0): ties the CRBA kernel to the RNEA kernel at batch 65 536."""
    m = load_model("panda_no_gripper", "cuda")
    B = 65536
    q, _, qdd = sample_states(m, B, seed=7)
    qt, at = torch.from_numpy(q).cuda(), torch.from_numpy(qdd).cuda()
    H = m.compute_lagrangian_inertia_matrix(qt)
    zero = torch.zeros_like(qt)
    lhs = torch.einsum("bij,bj->bi", H, at)
    rhs = m.compute_inverse_dynamics(qt, zero, at, include_gravity=False, use_damping=False)
    assert (lhs - rhs).abs().max().item() < 5e-5
    assert torch.equal(H, H.transpose(1, 2))
    assert torch.linalg.eigvalsh(H[:4096].double()).min().item() > 0


# ---------------------------------------------------------------------------------------------------------------
# Gradients of a loss on H: n passes of the RNEA backward (column j: qdd = e_j, grad_tau = dL/dH[:, :, j]) against
# torch autograd through the reference's n + 1 inverse-dynamics construction (tests/golden/golden_grad_mass.npz,
# made by tests/golden/make_golden_grad_mass.py).
# ---------------------------------------------------------------------------------------------------------------
H_GRAD_CASES = ["iiwa7", "panda_no_gripper", "trifinger_edu", "fetch", "jaco", "panda", "iiwa7_allegro"]
H_GRAD_RTOL = 2e-3


def load_golden_grad_mass():
    import os
    from helpers import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "golden_grad_mass.npz"), allow_pickle=False)


@pytest.mark.parametrize("case", H_GRAD_CASES)
def test_emu_mass_matrix_backward_vs_reference_autograd(emu, case):
    check_emu_mass_matrix_backward_vs_reference_autograd(emu, load_golden_grad_mass(), case)


def check_emu_mass_matrix_backward_vs_reference_autograd(emu, g, case):
    from test_forward_dynamics import grad_close, learnable_model_fd
    from test_rnea_backward import dynamic_param_mask
    m, params = learnable_model_fd(g, case)
    q = np.ascontiguousarray(g[case + "/q"])
    B, n = q.shape
    prog = build_walk(m._spec, whole_tree=True)
    table = m._link_table()
    ops_f_t = (table.reshape(-1)[torch.from_numpy(prog.gather.reshape(-1))]
               * torch.from_numpy(prog.gsign.reshape(-1))).reshape(prog.capacity, 32)
    ops_f = np.ascontiguousarray(ops_f_t.detach().numpy(), np.float32)
    walk, _keep = host_walk(m, prog)
    walk.ops_f = ops_f.ctypes.data
    H = np.full((B, n, n), np.nan, np.float32)
    assert emu.emu_crba(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(H)) == 0
    assert np.allclose(H, g[case + "/H"], **TOL_H_REF), np.abs(H - g[case + "/H"]).max()
    G = (2.0 * g[case + "/weight"] * (H - g[case + "/want"]) / H.size).astype(np.float32)
    zero = np.zeros((B, n), np.float32)
    gq_sum, gops_sum = np.zeros((B, n), np.float64), np.zeros((prog.capacity, 32), np.float64)
    mask = dynamic_param_mask(m, prog)
    for j in range(n):
        unit = np.zeros((B, n), np.float32); unit[:, j] = 1.0
        gt = np.ascontiguousarray(G[:, :, j])
        gq, gqd, gqdd = (np.full((B, n), np.nan, np.float32) for _ in range(3))
        gops = np.full((prog.capacity, 32), np.nan, np.float32)
        assert emu.emu_rnea_backward(ctypes.byref(walk), _ptr(q), _ptr(zero), _ptr(unit), ctypes.c_int64(B), 0, _ptr(gt),
                                     ctypes.c_uint64(mask), _ptr(gq), _ptr(gqd), _ptr(gqdd), _ptr(gops)) == 0
        gq_sum += gq; gops_sum += gops
    assert grad_close(gq_sum, g[case + "/grad_q"], H_GRAD_RTOL), np.abs(gq_sum - g[case + "/grad_q"]).max()
    m.zero_grad()
    ops_f_t.backward(torch.from_numpy(gops_sum.astype(np.float32)))
    for key, p in params.items():
        ref = g["%s/grad/%s" % (case, key)]
        assert grad_close(p.grad.numpy(), ref, H_GRAD_RTOL), (case, key, p.grad.numpy().reshape(-1), ref.reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("case", H_GRAD_CASES)
def test_gpu_mass_matrix_backward_vs_reference_autograd(case):
    check_gpu_mass_matrix_backward_vs_reference_autograd(load_golden_grad_mass(), case)


def check_gpu_mass_matrix_backward_vs_reference_autograd(g, case, device="cuda"):
    from test_forward_dynamics import grad_close, learnable_model_fd
    m, params = learnable_model_fd(g, case, device)
    q = torch.from_numpy(g[case + "/q"].copy()).to(device).requires_grad_(True)
    want, weight = (torch.from_numpy(g[case + "/" + k].copy()).to(device) for k in ("want", "weight"))
    H = m.compute_lagrangian_inertia_matrix(q)
    loss = (weight * (H - want) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - float(g[case + "/loss"])) <= 1e-3 * max(1e-6, float(g[case + "/loss"]))
    assert grad_close(q.grad.cpu().numpy(), g[case + "/grad_q"], H_GRAD_RTOL)
    for key, p in params.items():
        assert grad_close(p.grad.cpu().numpy(), g["%s/grad/%s" % (case, key)], H_GRAD_RTOL), (case, key)
