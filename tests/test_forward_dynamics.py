"""Forward dynamics (K8, csrc/drm_forward_dynamics.hip: the articulated-body recursion for robots with a long segment,
composite-rigid-body H + RNEA bias torques + leaf-to-root L^T D L solve for 7-DoF arms and hands).

CPU (not gpu): the oracle's line-by-line restatement of the reference's articulated-body recursion
(robot_model.py:487-624) against accelerations recorded from the UNMODIFIED reference (tests/golden/golden_fd.npz,
made by tests/golden/make_golden_fd.py) and against the RNEA oracle (ABA inverts RNEA to 1e-10 in fp64); the kernel
arithmetic (host emulation) against the fp64 oracle for every shipped robot.  GPU (-m gpu): the real kernel.

Tolerance: forward dynamics amplifies fp32 rounding by cond(H) (up to ~1e6 for the Jaco's gram-scale finger links,
where |qdd| reaches 1e6 rad/s^2); the reference's own fp32 result is 1.5e-4 (relative to 1 + |qdd|) away from an fp64
evaluation there.  Accelerations are therefore compared as |d qdd| <= tol * (1 + |qdd|) with tol = 1e-3 against
fp64 and against the reference (hands, grippers, mobile bases with gram-scale links), 1e-4 for the arms.  The robots
that are badly conditioned (Fetch, Jaco, an arm carrying a hand: sub-tree masses spread over more than two decades) run
the articulated-body kernel, whose error is that of the reference's own fp32 recursion (3e-5 .. 2e-4 against fp64).
"""
import ctypes

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.flatten import build_walk
from helpers import ALL_ROBOTS, GOLDEN_DIR, GOLDEN_ROBOTS, load_golden, load_model, sample_states
from oracle import Oracle
from test_host_emu import _ptr, emu, host_walk  # noqa: F401  (emu is a fixture)

import os

TOL = 1e-3
TOL_ARMS = 1e-4
ARMS = ("panda_no_gripper", "iiwa7", "2link_robot", "panda")


def tol_of(robot):
    # one bound for every robot that carries gram-scale links, a 7-DoF arm with a 16-DoF hand (cond(H) ~ 1e8) included:
    # H is factorised L^T D L from the distal joints inwards, like the reference's recursion (drm_sample.hpp ltdl_solve)
    return TOL_ARMS if robot in ARMS else TOL
FLAGS = ((1, 0), (1, 1), (0, 0))


def load_golden_fd():
    return np.load(os.path.join(GOLDEN_DIR, "golden_fd.npz"), allow_pickle=False)


def rel_err(a, ref):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    return float((np.abs(a - ref) / (1.0 + np.abs(ref))).max())


@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_oracle_aba_vs_reference(robot, links):
    g, gf = load_golden(robot), load_golden_fd()
    orc = Oracle(load_model(robot)._spec)
    for grav, damp in FLAGS:
        ref = gf["%s/qdd_g%d_d%d" % (robot, grav, damp)]
        a32 = orc.forward_dynamics(g["fast_q"], g["fast_qd"], gf[robot + "/f"], grav, damp, np.float32)
        assert rel_err(a32, ref) < 5e-5, (robot, grav, damp, rel_err(a32, ref))


@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_oracle_aba_inverts_oracle_rnea(robot):
    m = load_model(robot)
    q, qd, qdd = (a.astype(np.float64) for a in sample_states(m, 6, seed=2))
    orc = Oracle(m._spec)
    for grav, damp in FLAGS:
        tau = orc.rnea(q, qd, qdd, grav, damp, np.float64)
        assert np.abs(orc.forward_dynamics(q, qd, tau, grav, damp, np.float64) - qdd).max() < 1e-8


@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_emu_forward_dynamics_vs_oracle(emu, robot):
    m = load_model(robot)
    n, B = m._n_dofs, 23
    q, qd, _ = sample_states(m, B, seed=61)
    f = np.random.default_rng(3).uniform(-1, 1, (B, n)).astype(np.float32)
    prog = build_walk(m._spec, whole_tree=True)
    walk, keep = host_walk(m, prog)
    orc = Oracle(m._spec)
    for grav, damp in FLAGS:
        out = np.full((B, n), np.nan, np.float32)
        assert emu.emu_forward_dynamics(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(f), ctypes.c_int64(B),
                                        grav | (damp << 1), _ptr(out)) == 0
        ref = orc.forward_dynamics(q.astype(np.float64), qd.astype(np.float64), f.astype(np.float64), grav, damp, np.float64)
        assert rel_err(out, ref) < tol_of(robot), (robot, grav, damp, rel_err(out, ref))


# ---------------------------------------------------------------------------------------------- GPU
def _supported(m):
    return True  # every shipped robot fits (lower triangle of H in LDS); the limit is n ~ 30


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ALL_ROBOTS)
@pytest.mark.parametrize("B", [1, 64, 100])
def test_gpu_forward_dynamics_vs_oracle(robot, B):
    m = load_model(robot, "cuda")
    q, qd, _ = sample_states(m, B, seed=70 + B)
    f = np.random.default_rng(B).uniform(-1, 1, (B, m._n_dofs)).astype(np.float32)
    tq, tqd, tf = (torch.from_numpy(a).cuda() for a in (q, qd, f))
    if not _supported(m):
        with pytest.raises(RuntimeError, match="does not fit"):
            m.compute_forward_dynamics(tq, tqd, tf)
        return
    orc = Oracle(m._spec)
    for grav, damp in FLAGS:
        f_before = tf.clone()
        out = m.compute_forward_dynamics(tq, tqd, tf, include_gravity=bool(grav), use_damping=bool(damp)).cpu().numpy()
        assert torch.equal(tf, f_before), "the caller's torques are not modified"
        ref = orc.forward_dynamics(q.astype(np.float64), qd.astype(np.float64), f.astype(np.float64), grav, damp, np.float64)
        assert rel_err(out, ref) < tol_of(robot), (robot, grav, damp, rel_err(out, ref))


@pytest.mark.gpu
@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_gpu_forward_dynamics_vs_reference_golden(robot, links):
    g, gf = load_golden(robot), load_golden_fd()
    m = load_model(robot, "cuda")
    if not _supported(m):
        pytest.skip("inertia matrix tile does not fit in LDS (documented limit)")
    q, qd, f = (torch.from_numpy(a.copy()).cuda() for a in (g["fast_q"], g["fast_qd"], gf[robot + "/f"]))
    for grav, damp in FLAGS:
        out = m.compute_forward_dynamics(q, qd, f, include_gravity=bool(grav), use_damping=bool(damp)).cpu().numpy()
        ref = gf["%s/qdd_g%d_d%d" % (robot, grav, damp)]
        assert rel_err(out, ref) < tol_of(robot), (robot, grav, damp, rel_err(out, ref))


@pytest.mark.gpu
def test_gpu_forward_dynamics_inverts_inverse_dynamics_full_size():
    """qdd -> tau = ID(q, qd, qdd) -> FD(q, qd, tau) == qdd at batch 65 536 (ties K8 to the RNEA kernel)."""
    m = load_model("panda_no_gripper", "cuda")
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(m, 65536, seed=9))
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    back = m.compute_forward_dynamics(q, qd, tau, include_gravity=True, use_damping=True)
    assert ((back - qdd).abs() / (1 + qdd.abs())).max().item() < 5e-4


# ---------------------------------------------------------------------------------------------------------------
# Gradients of a forward-dynamics loss (examples/learn_forward_dynamics_iiwa.py): implicit differentiation
# (lambda = H^-1 g, then the RNEA backward with grad_tau = lambda, negated) against torch autograd through the
# reference's articulated-body recursion (tests/golden/golden_grad_fd.npz, made by tests/golden/make_golden_grad_fd.py).
# Tolerance: relative to the largest entry of each gradient tensor; the gradient carries cond(H) twice (two solves).
# ---------------------------------------------------------------------------------------------------------------
FD_GRAD_CASES = ["iiwa7", "panda_no_gripper", "trifinger_edu", "fetch", "jaco", "panda", "iiwa7_allegro"]
FD_GRAD_RTOL = {"iiwa7": 2e-3, "panda_no_gripper": 2e-3, "trifinger_edu": 1e-2, "fetch": 1e-2, "jaco": 1e-2, "panda": 2e-3,
                "iiwa7_allegro": 1e-2}


# forward value of the PERTURBED robot (the learnable link's trans / com / inertia start from random values, which leaves a
# gram-scale finger or a gripper behind a badly placed link): two fp32 evaluations of that system — the reference's and
# the kernel's — differ by up to 5e-3 relative on single entries of the 6-9-row fixtures and up to 1.6e-2 on the worst of the 192
# rows of golden_tiles_grad_fd.npz; neither is the yardstick, the gradients below are held to FD_GRAD_RTOL of their largest
# entry all the same
FD_GRAD_FWD_TOL = {"fetch": 2e-2, "jaco": 1e-2, "iiwa7_allegro": 2e-2}


def load_golden_grad_fd():
    return np.load(os.path.join(GOLDEN_DIR, "golden_grad_fd.npz"), allow_pickle=False)


def learnable_model_fd(g, case, device="cpu"):
    """test_rnea_backward.learnable_model with the inertia parametrisation of make_golden_grad_fd.py."""
    from differentiable_robot_model_amd.rigid_body_params import SymmPosDef3DInertiaMatrixNet
    from test_rnea_backward import parametrization
    m = load_model(case, device)
    params = {}
    for key in g[case + "/keys"]:
        link, pname, tensor_name = str(key).split("/")
        mod = SymmPosDef3DInertiaMatrixNet() if (pname == "inertia_mat" and case != "iiwa7") else parametrization(pname)
        m.make_link_param_learnable(link, pname, mod)
        p = dict(mod.named_parameters())[tensor_name]
        with torch.no_grad():
            p.copy_(torch.from_numpy(g["%s/init/%s" % (case, key)].copy()).reshape(p.shape).to(p.device))
        params[str(key)] = p
    return m, params


def grad_close(a, b, rtol):
    a = np.asarray(a, np.float64).reshape(-1); b = np.asarray(b, np.float64).reshape(-1)
    return np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-12)


@pytest.mark.parametrize("case", FD_GRAD_CASES)
def test_emu_forward_dynamics_backward_vs_reference_autograd(emu, case):
    check_emu_forward_dynamics_backward_vs_reference_autograd(emu, load_golden_grad_fd(), case)


def check_emu_forward_dynamics_backward_vs_reference_autograd(emu, g, case):
    from test_rnea_backward import dynamic_param_mask
    m, params = learnable_model_fd(g, case)
    q, qd, f = (np.ascontiguousarray(g["%s/%s" % (case, k)]) for k in ("q", "qd", "f"))
    want = g[case + "/want"]
    prog = build_walk(m._spec, whole_tree=True)
    table = m._link_table()
    ops_f_t = (table.reshape(-1)[torch.from_numpy(prog.gather.reshape(-1))]
               * torch.from_numpy(prog.gsign.reshape(-1))).reshape(prog.capacity, 32)
    ops_f = np.ascontiguousarray(ops_f_t.detach().numpy(), np.float32)
    walk, _keep = host_walk(m, prog)
    walk.ops_f = ops_f.ctypes.data
    B, n = q.shape
    qdd = np.zeros((B, n), np.float32)
    assert emu.emu_forward_dynamics(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(f), ctypes.c_int64(B), 3, _ptr(qdd)) == 0
    assert rel_err(qdd, g[case + "/qdd"]) <= FD_GRAD_FWD_TOL.get(case, tol_of(case))
    gqdd = np.ascontiguousarray(2.0 * (qdd - want) / (B * n), np.float32)
    lam, zero = np.zeros((B, n), np.float32), np.zeros((B, n), np.float32)
    assert emu.emu_forward_dynamics(ctypes.byref(walk), _ptr(q), _ptr(zero), _ptr(gqdd), ctypes.c_int64(B), 0, _ptr(lam)) == 0
    gq, gqd, gx = (np.full((B, n), np.nan, np.float32) for _ in range(3))
    gops = np.full((prog.capacity, 32), np.nan, np.float32)
    assert emu.emu_rnea_backward(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(lam),
                                 ctypes.c_uint64(dynamic_param_mask(m, prog)), _ptr(gq), _ptr(gqd), _ptr(gx), _ptr(gops)) == 0
    rtol = FD_GRAD_RTOL[case]
    assert grad_close(-gq, g[case + "/grad_q"], rtol), np.abs(-gq - g[case + "/grad_q"]).max()
    assert grad_close(-gqd, g[case + "/grad_qd"], rtol)
    assert grad_close(lam, g[case + "/grad_f"], rtol)
    m.zero_grad()
    ops_f_t.backward(torch.from_numpy(-gops))
    for key, p in params.items():
        ref = g["%s/grad/%s" % (case, key)]
        assert grad_close(p.grad.numpy(), ref, rtol), (case, key, p.grad.numpy().reshape(-1), ref.reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("case", FD_GRAD_CASES)
def test_gpu_forward_dynamics_backward_vs_reference_autograd(case):
    check_gpu_forward_dynamics_backward_vs_reference_autograd(load_golden_grad_fd(), case)


def check_gpu_forward_dynamics_backward_vs_reference_autograd(g, case, device="cuda"):
    m, params = learnable_model_fd(g, case, device)
    q, qd, f = (torch.from_numpy(g["%s/%s" % (case, k)].copy()).to(device).requires_grad_(True) for k in ("q", "qd", "f"))
    want = torch.from_numpy(g[case + "/want"].copy()).to(device)
    qdd = m.compute_forward_dynamics(q, qd, f, include_gravity=True, use_damping=True)
    loss = torch.nn.functional.mse_loss(qdd, want)
    loss.backward()
    rtol = FD_GRAD_RTOL[case]
    assert abs(loss.item() - float(g[case + "/loss"])) <= 2 * FD_GRAD_FWD_TOL.get(case, tol_of(case)) * max(1.0, float(g[case + "/loss"]))
    assert grad_close(q.grad.cpu().numpy(), g[case + "/grad_q"], rtol)
    assert grad_close(qd.grad.cpu().numpy(), g[case + "/grad_qd"], rtol)
    assert grad_close(f.grad.cpu().numpy(), g[case + "/grad_f"], rtol)
    for key, p in params.items():
        assert grad_close(p.grad.cpu().numpy(), g["%s/grad/%s" % (case, key)], rtol), (case, key)


@pytest.mark.gpu
def test_gpu_learn_forward_dynamics_loop_lowers_the_loss():
    """examples/learn_forward_dynamics_iiwa.py:55-90 in miniature: mass, com and inertia of iiwa_link_1 learned from
    accelerations of the ground-truth model."""
    from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, UnconstrainedTensor
    torch.manual_seed(0)
    gt = load_model("iiwa7", "cuda")
    m = load_model("iiwa7", "cuda")
    m.make_link_param_learnable("iiwa_link_1", "mass", PositiveScalar())
    m.make_link_param_learnable("iiwa_link_1", "com", UnconstrainedTensor(dim1=1, dim2=3))
    m.make_link_param_learnable("iiwa_link_1", "inertia_mat", UnconstrainedTensor(dim1=3, dim2=3))
    q, qd, f = (torch.from_numpy(a).cuda() for a in sample_states(gt, 256, seed=3))
    with torch.no_grad():
        want = gt.compute_forward_dynamics(q, qd, f, include_gravity=True, use_damping=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(m.compute_forward_dynamics(q, qd, f, include_gravity=True, use_damping=True), want)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


@pytest.mark.gpu
def test_gpu_forward_dynamics_old_is_the_same_linear_system():
    """compute_forward_dynamics_old (robot_model.py:452-485, dead upstream: torch.solve) = H^-1 (f - nle), damping on by default."""
    m = load_model("iiwa7", "cuda")
    q, qd, f = (torch.from_numpy(a).cuda() for a in sample_states(m, 100, seed=5))
    old = m.compute_forward_dynamics_old(q, qd, f)
    assert torch.equal(old, m.compute_forward_dynamics(q, qd, f, include_gravity=True, use_damping=True))
    H = m.compute_lagrangian_inertia_matrix(q)
    nle = m.compute_non_linear_effects(q, qd)
    assert torch.allclose(torch.einsum("bij,bj->bi", H, old) + nle, f, atol=2e-4, rtol=2e-4)


# ---------------------------------------------------------------------------------------------- articulated-body walk
LONG_SEGMENT_ROBOTS = ["fetch", "jaco", "iiwa7_allegro", "panda"]


@pytest.mark.parametrize("robot", LONG_SEGMENT_ROBOTS)
def test_emu_articulated_body_walk_has_the_accuracy_of_the_reference_recursion(emu, robot):
    """drm_tree.hpp aba_tree_walk in fp32 against the fp64 oracle: no worse than the reference's own recursion evaluated in
    fp32 (the oracle's fp32 build), on the whole-tree walk and on the folded walk the API launches."""
    from test_host_emu import folded_host_walk
    m = load_model(robot)
    n, B = m._n_dofs, 120
    q, qd, _ = sample_states(m, B, seed=61)
    f = np.random.default_rng(3).uniform(-1, 1, (B, n)).astype(np.float32)
    orc = Oracle(m._spec)
    args = (q.astype(np.float64), qd.astype(np.float64), f.astype(np.float64), 1, 1)
    ref, aba32 = orc.forward_dynamics(*args, np.float64), orc.forward_dynamics(*args, np.float32)
    prog = build_walk(m._spec, whole_tree=True)
    assert max(prog.seg_begin[s + 1] - prog.seg_begin[s] for s in range(prog.n_segments)) > 6  # (what selects the walk)
    walk, keep = host_walk(m, prog)
    fprog = build_walk(m._spec, whole_tree=True, drop_folded=True)
    fwalk, fkeep = folded_host_walk(m, fprog)
    for w in (walk, fwalk):
        out = np.full((B, n), np.nan, np.float32)
        assert emu.emu_forward_dynamics(ctypes.byref(w), _ptr(q), _ptr(qd), _ptr(f), ctypes.c_int64(B), 3, _ptr(out)) == 0
        assert rel_err(out, ref) < max(3e-4, 3.0 * rel_err(aba32, ref)), (robot, rel_err(out, ref), rel_err(aba32, ref))


@pytest.mark.gpu
@pytest.mark.parametrize("robot", LONG_SEGMENT_ROBOTS)
def test_gpu_articulated_body_kernel_vs_fp64_oracle(robot):
    m = load_model(robot, "cuda")
    n, B = m._n_dofs, 200
    q, qd, _ = sample_states(m, B, seed=61)
    f = np.random.default_rng(3).uniform(-1, 1, (B, n)).astype(np.float32)
    tq, tqd, tf = (torch.from_numpy(a).cuda() for a in (q, qd, f))
    args = (q.astype(np.float64), qd.astype(np.float64), f.astype(np.float64), 1, 1)
    orc = Oracle(m._spec)
    ref, aba32 = orc.forward_dynamics(*args, np.float64), orc.forward_dynamics(*args, np.float32)
    got = rel_err(m.compute_forward_dynamics(tq, tqd, tf, True, True).cpu().numpy(), ref)
    assert got < max(3e-4, 3.0 * rel_err(aba32, ref)), (robot, got, rel_err(aba32, ref))
