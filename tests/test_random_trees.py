"""Random kinematic trees through the loop-structured dynamics walks: RNEA, the mass matrix and forward dynamics of robots
nobody shipped — random branching, revolute / prismatic / fixed joints about axis-aligned and skew axes, random frames and
inertias — against the fp64 oracle.  What it is after: the branch-point bookkeeping of the walks (save slots, accumulators,
segments hanging off a fixed root) in shapes the 13 shipped robots do not have, in particular the articulated-body walk
(drm_tree.hpp aba_tree_walk) and the short-segment forms.

CPU: the host emulation of the kernel arithmetic (tests/host_emu).  GPU (-m gpu): the kernels through the API."""
import contextlib
import ctypes
import io
import os

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.flatten import KIND_PRISMATIC, build_walk
from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel
from helpers import TOL_TAU, sample_states
from oracle import Oracle
from test_host_emu import _ptr, emu, host_walk  # noqa: F401  (emu is a fixture)

SEEDS = list(range(int(os.environ.get("DRM_FUZZ_SEEDS", "10"))))


def tree_urdf(seed: int) -> str:
    rng = np.random.default_rng(77000 + seed)
    n_links = int(rng.integers(3, 22))
    chainy = rng.random()              # how often a link hangs off the newest link (1: a chain, 0: a bush)
    fixed_root = rng.random() < 0.4    # sub-trees behind fixed joints at the root: independent segments
    out = ['<?xml version="1.0"?>', '<robot name="tree%d">' % seed, '  <link name="base"/>']
    names = ["base"]
    axes_aligned = ["1 0 0", "0 1 0", "0 0 1", "-1 0 0", "0 -1 0", "0 0 -1"]
    movable = 0
    for i in range(n_links):
        name = "l%d" % i
        parent = names[-1] if rng.random() < chainy else names[int(rng.integers(len(names)))]
        if fixed_root and rng.random() < 0.3:
            parent = "base"
        A = rng.standard_normal((3, 3)) * 0.03
        I = A @ A.T + np.eye(3) * 0.002
        m, c = 0.05 + rng.random() * 0.8, rng.standard_normal(3) * 0.04
        out.append('  <link name="%s"><inertial><origin xyz="%.5f %.5f %.5f" rpy="0 0 0"/><mass value="%.5f"/>'
                   '<inertia ixx="%.6f" ixy="%.6f" ixz="%.6f" iyy="%.6f" iyz="%.6f" izz="%.6f"/></inertial></link>'
                   % (name, c[0], c[1], c[2], m, I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))
        xyz, rpy = rng.standard_normal(3) * 0.08, rng.standard_normal(3) * 0.7
        u = rng.random()
        kind = "fixed" if (u < 0.2 or (parent == "base" and fixed_root)) else ("prismatic" if u < 0.35 else "revolute")
        if kind == "fixed":
            out.append('  <joint name="j%d" type="fixed"><parent link="%s"/><child link="%s"/>'
                       '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/></joint>' % (i, parent, name, *xyz, *rpy))
        else:
            movable += 1
            if rng.random() < 0.6:
                axis = axes_aligned[int(rng.integers(6))]
            else:
                a = rng.standard_normal(3)
                axis = "%.6f %.6f %.6f" % tuple(a / np.linalg.norm(a))
            lim = (-0.3, 0.3) if kind == "prismatic" else (-2.5, 2.5)
            out.append('  <joint name="j%d" type="%s"><parent link="%s"/><child link="%s"/>'
                       '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/><axis xyz="%s"/>'
                       '<limit effort="10" lower="%.2f" upper="%.2f" velocity="3"/><dynamics damping="%.3f"/></joint>'
                       % (i, kind, parent, name, *xyz, *rpy, axis, lim[0], lim[1], rng.random() * 0.2))
        names.append(name)
    if movable == 0:   # at least one joint that moves
        out.append('  <link name="tail"><inertial><origin xyz="0 0 0.02" rpy="0 0 0"/><mass value="0.3"/>'
                   '<inertia ixx="0.002" ixy="0" ixz="0" iyy="0.002" iyz="0" izz="0.002"/></inertial></link>')
        out.append('  <joint name="jt" type="revolute"><parent link="%s"/><child link="tail"/><origin xyz="0 0 0.1" rpy="0 0 0"/>'
                   '<axis xyz="0 1 0"/><limit effort="10" lower="-2" upper="2" velocity="3"/></joint>' % names[-1])
    out.append("</robot>")
    return "\n".join(out)


def tree_model(tmp_path, seed, device="cpu"):
    path = os.path.join(str(tmp_path), "tree%d.urdf" % seed)
    with open(path, "w") as f:
        f.write(tree_urdf(seed))
    with contextlib.redirect_stdout(io.StringIO()):
        return DifferentiableRobotModel(path, device=device)


def rel(a, ref):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    return float((np.abs(a - ref) / (1.0 + np.abs(ref))).max())


def test_the_generator_covers_the_shapes_it_is_meant_to(tmp_path):
    seen = dict(segments=0, slots=0, prismatic=0, long=0, short=0)
    for seed in range(40):
        m = tree_model(tmp_path, seed)
        prog = build_walk(m._spec, whole_tree=True)
        seen["segments"] += prog.n_segments > 1
        seen["slots"] += prog.n_slots > 0
        seen["prismatic"] += bool(np.any(np.asarray(m._spec.kind) == KIND_PRISMATIC))
        longest = max(prog.seg_begin[s + 1] - prog.seg_begin[s] for s in range(prog.n_segments))
        seen["long"] += longest > 6
        seen["short"] += longest <= 6
    assert all(v >= 4 for v in seen.values()), seen


@pytest.mark.parametrize("seed", SEEDS)
def test_emu_random_tree_dynamics_vs_oracle(emu, tmp_path, seed):
    m = tree_model(tmp_path, seed)
    n, B = m._n_dofs, 9
    q, qd, qdd = sample_states(m, B, seed=seed)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    orc = Oracle(m._spec)
    prog = build_walk(m._spec, whole_tree=True)
    walk, _keep = host_walk(m, prog)
    tau = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(tau)) == 0
    assert np.allclose(tau, orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU), seed
    H = np.full((B, n, n), np.nan, np.float32)
    assert emu.emu_crba(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(H)) == 0
    assert np.allclose(H, orc.mass_matrix(q64, False, False, np.float64), **TOL_TAU), seed
    for flags in (3, 0):
        acc = np.full((B, n), np.nan, np.float32)
        assert emu.emu_forward_dynamics(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(acc)) == 0
        ref = orc.forward_dynamics(q64, qd64, qdd64, flags & 1, flags >> 1, np.float64)
        assert rel(acc, ref) < 1e-3, (seed, flags, rel(acc, ref))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_random_tree_dynamics_vs_oracle(tmp_path, seed):
    mc, m = tree_model(tmp_path, seed), tree_model(tmp_path, seed, "cuda")
    n = m._n_dofs
    B = int(np.random.default_rng(seed).choice([1, 37, 64, 200]))
    q, qd, qdd = sample_states(mc, B, seed=seed)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    orc = Oracle(mc._spec)
    dev = lambda a: torch.from_numpy(a).cuda()
    tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=True, use_damping=True)
    assert np.allclose(tau.cpu().numpy(), orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU), seed
    H = m.compute_lagrangian_inertia_matrix(dev(q))
    assert np.allclose(H.cpu().numpy(), orc.mass_matrix(q64, False, False, np.float64), **TOL_TAU), seed
    acc = m.compute_forward_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=True, use_damping=True)
    ref = orc.forward_dynamics(q64, qd64, qdd64, True, True, np.float64)
    assert rel(acc.cpu().numpy(), ref) < 1e-3, (seed, B, rel(acc.cpu().numpy(), ref))


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["iiwa7_allegro", "fetch", "panda"])
def test_gpu_persistent_kernels_walk_every_tile_of_a_large_batch(robot):
    """Inverse dynamics, the mass matrix and forward dynamics of robots with a long segment run on a PERSISTENT grid: at
    150 001 samples (2 344 tiles, the last one ragged) every block loops over several tiles and re-uses its slice of scratch.
    Rows from the start, the middle and the ragged end of that launch must be bit-identical to the same rows launched on their
    own (a sample's arithmetic does not depend on its tile), and a few of them are checked against the fp64 oracle."""
    from helpers import load_model
    mc, m = load_model(robot), load_model(robot, "cuda")
    B, n = 150001, m._n_dofs
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, B, seed=5))
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    H = m.compute_lagrangian_inertia_matrix(q)
    acc = m.compute_forward_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    from differentiable_robot_model_amd.flatten import SHAPE_ARM_HAND
    arm_hand = bool(m._dynamics_walk().program.shape & SHAPE_ARM_HAND)
    for lo in (0, 70000, 131072 - 65, B - 130):
        sl = slice(lo, lo + 130)
        alone = m.compute_inverse_dynamics(q[sl], qd[sl], qdd[sl], include_gravity=True, use_damping=True)
        if arm_hand:
            # inverse dynamics of an arm that carries a hand: full 64-row tiles run the straight-line kernel (drm_arm_hand.hip),
            # only the ragged tail of a launch the persistent loop kernel — the same row is bit-identical wherever the SAME kernel
            # computes it, and agrees to rounding between the two kernels
            assert np.allclose(tau[sl].cpu().numpy(), alone.cpu().numpy(), **TOL_TAU), lo
            if lo % 64 == 0 and lo + 128 <= B - B % 64:
                assert torch.equal(tau[lo:lo + 128], alone[:128]), lo
        else:
            assert torch.equal(tau[sl], alone), lo
        H_alone = m.compute_lagrangian_inertia_matrix(q[sl])
        if arm_hand:
            assert np.allclose(H[sl].cpu().numpy(), H_alone.cpu().numpy(), **TOL_TAU), lo
            if lo % 64 == 0 and lo + 128 <= B - B % 64:
                assert torch.equal(H[lo:lo + 128], H_alone[:128]), lo
        else:
            assert torch.equal(H[sl], H_alone), lo
        acc_alone = m.compute_forward_dynamics(q[sl], qd[sl], qdd[sl], include_gravity=True, use_damping=True)
        if arm_hand:   # (as for inverse dynamics above: full tiles and the ragged tail run different kernels on these robots)
            assert rel(acc[sl].cpu().numpy(), acc_alone.cpu().numpy()) < 1e-3, lo
            if lo % 64 == 0 and lo + 128 <= B - B % 64:
                assert torch.equal(acc[lo:lo + 128], acc_alone[:128]), lo
        else:
            assert torch.equal(acc[sl], acc_alone), lo
    rows = [0, 63, 64, 99999, B - 1]
    orc = Oracle(mc._spec)
    q64, qd64, qdd64 = (t[rows].cpu().numpy().astype(np.float64) for t in (q, qd, qdd))
    assert np.allclose(tau[rows].cpu().numpy(), orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU)
    assert np.allclose(H[rows].cpu().numpy(), orc.mass_matrix(q64, False, False, np.float64), **TOL_TAU)
    assert rel(acc[rows].cpu().numpy(), orc.forward_dynamics(q64, qd64, qdd64, True, True, np.float64)) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_random_tree_fk_and_jacobian_to_every_link_vs_oracle(tmp_path, seed):
    """The root-to-link walk of ANY link is a serial chain, so on these random trees (revolute / prismatic / fixed joints in any
    order, skew axes as two ops, arbitrary DoF numbering, chains of 1 .. 20+ ops) `compute_fk_and_jacobian` runs the
    straight-line chain kernels (drm_chain_kernels.hip, capacities 4 / 8 / 12 / 16) for full tiles and the loop kernel for the
    ragged tail and for longer chains: every link, every row against the fp64 oracle; FK alone (drm_fk) must return the
    same pose."""
    from differentiable_robot_model_amd.flatten import SHAPE_SERIAL_CHAIN
    from helpers import TOL_JAC, TOL_POS, TOL_QUAT, quat_close
    mc, m = tree_model(tmp_path, seed), tree_model(tmp_path, seed, "cuda")
    B = 64 * 3 + 5
    q, _, _ = sample_states(mc, B, seed=seed)
    q64 = q.astype(np.float64)
    orc = Oracle(mc._spec)
    dq = torch.from_numpy(q).cuda()
    took_chain_kernel = 0
    for idx, body in enumerate(mc._bodies):
        if idx == 0:
            continue
        prog = build_walk(mc._spec, targets=[idx])
        assert prog.shape & SHAPE_SERIAL_CHAIN, (seed, body.name)
        took_chain_kernel += prog.capacity in (4, 8, 12, 16)
        pos, quat, lin, ang = m.compute_fk_and_jacobian(dq, body.name)
        rp, rq, rl, ra = orc.fk_jacobian(q64, idx, np.float64)
        scale = max(1.0, float(np.abs(rp).max()))
        assert np.abs(pos.cpu().numpy() - rp).max() <= TOL_POS["atol"] * scale, (seed, body.name)
        assert quat_close(quat.cpu().numpy(), rq, 2 * TOL_QUAT["atol"])[0], (seed, body.name)
        assert np.abs(lin.cpu().numpy() - rl).max() <= TOL_JAC["atol"] * scale and np.abs(ang.cpu().numpy() - ra).max() <= TOL_JAC["atol"], (seed, body.name)
        p2, r2 = m.compute_forward_kinematics(dq, body.name)
        assert np.abs(p2.cpu().numpy() - rp).max() <= TOL_POS["atol"] * scale and quat_close(r2.cpu().numpy(), rq, 2 * TOL_QUAT["atol"])[0]
    assert took_chain_kernel > 0
