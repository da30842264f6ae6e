"""Forward dynamics (K8, csrc/drm_forward_dynamics.hip: composite-rigid-body H + RNEA bias torques + Cholesky).

CPU (not gpu): the oracle's line-by-line restatement of the reference's articulated-body recursion
(robot_model.py:487-624) against accelerations recorded from the UNMODIFIED reference (tests/golden/golden_fd.npz,
made by tests/golden/make_golden_fd.py) and against the RNEA oracle (ABA inverts RNEA to 1e-10 in fp64); the kernel
arithmetic (host emulation) against the fp64 oracle for every shipped robot.  GPU (-m gpu): the real kernel.

Tolerance: forward dynamics amplifies fp32 rounding by cond(H) (up to ~1e6 for the Jaco's gram-scale finger links,
where |qdd| reaches 1e6 rad/s^2); the reference's own fp32 result is 1.5e-4 (relative to 1 + |qdd|) away from an fp64
evaluation there.  Accelerations are therefore compared as |d qdd| <= tol * (1 + |qdd|) with tol = 2e-3 against
fp64 and against the reference (hands, grippers, mobile bases with gram-scale links), 1e-4 for the arms (observed:
<= 5e-5 arms, <= 8e-4 hands).
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

TOL = 2e-3
TOL_ARMS = 1e-4
ARMS = ("panda_no_gripper", "iiwa7", "2link_robot", "panda")
# a 7-DoF arm carrying a 16-DoF hand: cond(H) ~ 1e8 (kilogram links above gram links).  A dense factorisation of H
# loses cond(H) * eps where the articulated-body recursion divides joint by joint; held to 1e-2
TOL_BY_ROBOT = {"iiwa7_allegro": 1e-2}


def tol_of(robot):
    return TOL_BY_ROBOT.get(robot, TOL_ARMS if robot in ARMS else TOL)
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
