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
import warnings

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
        return DifferentiableRobotModel(path, device=device, reference_compat=False)


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


def fk_every_link_in_walk_order(emu, m, seed):
    """All links' poses in walk order through the emulated many-target walk (fanned out where flatten.fk_fan_partition finds a
    hub — each run with NaN-poisoned save slots of its own) against the oracle; returns whether the walk fanned out."""
    from differentiable_robot_model_amd.flatten import SHAPE_FK_FAN
    from helpers import quat_close
    B = 7
    q, _, _ = sample_states(m, B, seed=seed)
    targets = [i for i in m._spec.preorder() if i != 0]
    prog = build_walk(m._spec, targets=targets)
    fan = bool(prog.shape & SHAPE_FK_FAN)
    if fan:
        P, sb = prog.prefix_end, list(prog.seg_begin)
        assert sb[0] == P and sb[-1] == prog.n_ops and 2 <= len(sb) - 1 <= 4 and all(a < b for a, b in zip(sb, sb[1:]))
        assert (len(sb) - 1) * P + prog.n_ops - P <= 2 * prog.n_ops
    walk, _keep = host_walk(m, prog)
    pos = np.full((B, len(targets), 3), np.nan, np.float32); quat = np.full((B, len(targets), 4), np.nan, np.float32)
    assert emu.emu_fk(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), len(targets), _ptr(pos), _ptr(quat)) == 0
    op, oq = Oracle(m._spec).fk(q.astype(np.float64), targets, np.float64)
    assert np.abs(pos - op).max() <= 2e-6 * max(1.0, float(np.abs(op).max())), seed
    assert quat_close(quat, oq, 2e-6)[0], seed
    return fan


def test_emu_every_link_in_walk_order_on_random_trees_and_generated_hands(emu, tmp_path):
    fanned = sum(fk_every_link_in_walk_order(emu, tree_model(tmp_path, seed), seed) for seed in SEEDS)
    for P, K, L in ARM_HAND_SHAPES[::3]:
        fanned += fk_every_link_in_walk_order(emu, arm_hand_model(tmp_path, P, K, L, 100 * P + 10 * K + L, "cpu"), P)
    assert fanned >= 3


@pytest.mark.gpu
def test_gpu_every_link_in_one_launch_on_random_trees_and_generated_hands(tmp_path):
    """compute_forward_kinematics_all_links (one many-target launch: grouped, or fanned out behind a hub) against the oracle,
    a ragged batch."""
    from helpers import TOL_POS, TOL_QUAT, quat_close
    models = [(tree_model(tmp_path, seed), tree_model(tmp_path, seed, "cuda")) for seed in SEEDS]
    models += [(arm_hand_model(tmp_path, P, K, L, 100 * P + 10 * K + L, "cpu"), arm_hand_model(tmp_path, P, K, L, 100 * P + 10 * K + L, "cuda"))
               for P, K, L in ARM_HAND_SHAPES[::3]]
    for mc, m in models:
        B = 64 * 2 + 11
        q, _, _ = sample_states(mc, B, seed=5)
        poses = m.compute_forward_kinematics_all_links(torch.from_numpy(q).cuda())
        targets = list(range(1, len(mc._bodies)))
        op, oq = Oracle(mc._spec).fk(q.astype(np.float64), targets, np.float64)
        for k, idx in enumerate(targets):
            pos, quat = poses[mc._bodies[idx].name]
            assert np.abs(pos.cpu().numpy() - op[:, k]).max() <= TOL_POS["atol"] * max(1.0, float(np.abs(op).max()))
            assert quat_close(quat.cpu().numpy(), oq[:, k], 2 * TOL_QUAT["atol"])[0]


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
    m.own_kernels = "off"       # (the library's persistent kernels are what this test walks; Fetch and the Panda ship their own, round 6)
    B, n = 150001, m._n_dofs
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, B, seed=5))
    tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    H = m.compute_lagrangian_inertia_matrix(q)
    acc = m.compute_forward_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    from differentiable_robot_model_amd.flatten import SHAPE_ARM_HAND
    arm_hand = bool(m._dynamics_walk().program.shape & SHAPE_ARM_HAND)
    tree_form = arm_hand and m._dynamics_walk().program.n_ops <= 12 and B // 64 >= 2048
    if tree_form:
        part = m.compute_lagrangian_inertia_matrix(q[64:64 + 2048 * 64])
        assert torch.equal(H[64:64 + 2048 * 64], part)
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
            # (small shapes: launches of >= 2 048 tiles take the one-wavefront-per-tile walk, drm_static.hpp crba_shape_body — the
            # rows of the big launch are then compared, bit for bit, with another launch of that size below)
            if lo % 64 == 0 and lo + 128 <= B - B % 64 and not tree_form:
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


# ------------------------------------------------------------------ arms that carry a hand nobody shipped
def arm_hand_urdf(P: int, K: int, L: int, seed: int) -> str:
    """An arm of P moving joints (one of them sliding when seed is odd) with a fixed flange, carrying K fingers of L joints each
    (revolute or, every third one, prismatic) behind fixed knuckle plates and ending in fixed fingertips: after the fixed
    joints are folded the dynamics walk is DRM_WALK_ARM_HAND (P, K, L).  Axis-aligned axes (the straight-line kernels' walks
    hold one op per joint), random frames and inertias."""
    rng = np.random.default_rng(91000 + 131 * seed + 17 * P + 5 * K + L)
    out = ['<?xml version="1.0"?>', '<robot name="arm%d_%d_%d_%d">' % (P, K, L, seed), '  <link name="base"/>']
    axes = ["1 0 0", "0 1 0", "0 0 1", "-1 0 0", "0 -1 0", "0 0 -1"]
    count = [0]

    def link(name, mass_scale=1.0):
        A = rng.standard_normal((3, 3)) * 0.03
        I = (A @ A.T + np.eye(3) * 0.002) * mass_scale
        m, c = (0.05 + rng.random() * 0.8) * mass_scale, rng.standard_normal(3) * 0.04
        out.append('  <link name="%s"><inertial><origin xyz="%.5f %.5f %.5f" rpy="0 0 0"/><mass value="%.5f"/>'
                   '<inertia ixx="%.6f" ixy="%.6f" ixz="%.6f" iyy="%.6f" iyz="%.6f" izz="%.6f"/></inertial></link>'
                   % (name, c[0], c[1], c[2], m, I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))

    def joint(parent, child, kind):
        xyz, rpy = rng.standard_normal(3) * 0.08, rng.standard_normal(3) * 0.7
        count[0] += 1
        if kind == "fixed":
            out.append('  <joint name="j%d" type="fixed"><parent link="%s"/><child link="%s"/>'
                       '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/></joint>' % (count[0], parent, child, *xyz, *rpy))
            return
        lim = (-0.3, 0.3) if kind == "prismatic" else (-2.5, 2.5)
        out.append('  <joint name="j%d" type="%s"><parent link="%s"/><child link="%s"/>'
                   '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/><axis xyz="%s"/>'
                   '<limit effort="10" lower="%.2f" upper="%.2f" velocity="3"/><dynamics damping="%.3f"/></joint>'
                   % (count[0], kind, parent, child, *xyz, *rpy, axes[int(rng.integers(6))], lim[0], lim[1], rng.random() * 0.2))

    parent = "base"
    slide = int(rng.integers(P)) if seed % 2 else -1
    for k in range(P):
        link("a%d" % k)
        joint(parent, "a%d" % k, "prismatic" if k == slide else "revolute")
        parent = "a%d" % k
    link("flange")
    joint(parent, "flange", "fixed")
    for j in range(K):
        link("k%d" % j, 0.2)
        joint("flange", "k%d" % j, "fixed")
        parent = "k%d" % j
        for i in range(L):
            name = "f%d_%d" % (j, i)
            link(name, 0.2)
            joint(parent, name, "prismatic" if (j + i + seed) % 3 == 0 else "revolute")
            parent = name
        link("tip%d" % j, 0.05)
        joint(parent, "tip%d" % j, "fixed")
    out.append("</robot>")
    return "\n".join(out)


ARM_HAND_SHAPES = [(P, K, L) for P in (5, 6, 7, 8, 9) for K, L in ((2, 1), (3, 1), (2, 2), (3, 2), (2, 3), (4, 3), (3, 4), (4, 4))]


def arm_hand_model(tmp_path, P, K, L, seed, device):
    path = os.path.join(str(tmp_path), "arm_%d_%d_%d_%d.urdf" % (P, K, L, seed))
    with open(path, "w") as f:
        f.write(arm_hand_urdf(P, K, L, seed))
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return DifferentiableRobotModel(path, device=device, reference_compat=False)


def test_generated_arms_with_hands_have_the_shape_they_are_built_for(tmp_path):
    from differentiable_robot_model_amd.flatten import SHAPE_ARM_HAND
    for P, K, L in ARM_HAND_SHAPES:
        m = arm_hand_model(tmp_path, P, K, L, P + K + L, "cpu")
        sh = build_walk(m._spec, whole_tree=True, drop_folded=True).shape & 0xffffffff
        assert sh & SHAPE_ARM_HAND and ((sh >> 24) & 0xf, ((sh >> 28) & 3) + 1, ((sh >> 30) & 3) + 1) == (P, K, L), (P, K, L)
        assert m._n_dofs == P + K * L


@pytest.mark.gpu
@pytest.mark.parametrize("P,K,L", ARM_HAND_SHAPES)
def test_gpu_generated_arm_with_hand_vs_oracle(tmp_path, P, K, L):
    """Every compiled (P, L) of the arm + hand kernels (drm_arm_hand.hip: 5 .. 9 prefix ops, sub-chains of 1 .. 4 ops, 2 .. 4 of
    them) on a robot generated for the shape — sliding joints in the arm and in the fingers, fixed flange / knuckles / tips
    folded on the host: inverse dynamics, the mass matrix, forward dynamics and the gradient of a torque loss against the fp64
    oracle (the gradient against central differences of it), full tiles + a ragged tail."""
    seed = P + K + L
    mc, m = arm_hand_model(tmp_path, P, K, L, seed, "cpu"), arm_hand_model(tmp_path, P, K, L, seed, "cuda")
    from differentiable_robot_model_amd.flatten import SHAPE_ARM_HAND
    assert m._dynamics_walk().program.shape & SHAPE_ARM_HAND
    n, B = m._n_dofs, 64 * 3 + 7
    q, qd, qdd = sample_states(mc, B, seed=seed)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    orc = Oracle(mc._spec)
    dev = lambda a: torch.from_numpy(a).cuda()
    for grav, damp in ((True, True), (False, False)):
        tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=grav, use_damping=damp)
        assert np.allclose(tau.cpu().numpy(), orc.rnea(q64, qd64, qdd64, grav, damp, np.float64), **TOL_TAU), (P, K, L, grav)
    H = m.compute_lagrangian_inertia_matrix(dev(q))
    assert np.allclose(H.cpu().numpy(), orc.mass_matrix(q64, False, False, np.float64), **TOL_TAU), (P, K, L)
    assert torch.equal(H, H.transpose(1, 2))
    tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=True, use_damping=True)
    acc = m.compute_forward_dynamics(dev(q), dev(qd), tau, include_gravity=True, use_damping=True)
    ref = orc.forward_dynamics(q64, qd64, tau.cpu().numpy().astype(np.float64), True, True, np.float64)
    assert rel(acc.cpu().numpy(), ref) < 2e-3, (P, K, L, rel(acc.cpu().numpy(), ref))
    # reverse mode: d <g, tau> / dq along a random direction against central differences of the fp64 oracle
    g = np.random.default_rng(seed).standard_normal((B, n)).astype(np.float32)
    tq = dev(q).requires_grad_(True)
    (m.compute_inverse_dynamics(tq, dev(qd), dev(qdd), include_gravity=True, use_damping=True) * dev(g)).sum().backward()
    v = np.random.default_rng(seed + 1).standard_normal((B, n))
    e = 1e-6
    fd = ((orc.rnea(q64 + e * v, qd64, qdd64, True, True, np.float64) - orc.rnea(q64 - e * v, qd64, qdd64, True, True, np.float64))
          / (2 * e) * g).sum(axis=1)
    mine = (tq.grad.cpu().numpy().astype(np.float64) * v).sum(axis=1)
    assert np.abs(mine - fd).max() <= 2e-3 * max(1.0, float(np.abs(fd).max())), (P, K, L, float(np.abs(mine - fd).max()))


# ---------------------------------------------------------------------------------------------- bushy trees: five branch points
BUSHY_SEEDS = [137, 225, 231, 292]      # (of seeds 100 .. 399: the trees whose whole-tree walk has five branch points)


def _bushy_gradients(m, seed, device):
    """(gradients of a weighted torque sum w.r.t. q / qd / qdd, gradient of a position sum of the last link w.r.t. q)"""
    q, qd, qdd = (torch.from_numpy(a).to(device) for a in sample_states(m, 70, seed=seed))
    w = torch.from_numpy(np.random.default_rng(seed).standard_normal((70, m._n_dofs)).astype(np.float32)).to(device)
    xs = [t.clone().requires_grad_(True) for t in (q, qd, qdd)]
    (m.compute_inverse_dynamics(*xs, include_gravity=True, use_damping=True) * w).sum().backward()
    x = q.clone().requires_grad_(True)
    poses = m.compute_forward_kinematics_all_links(x)
    sum(p[0].sum() for p in poses.values()).backward()
    return [t.grad.cpu() for t in xs] + [x.grad.cpu()], (q, qd, qdd, w)


@pytest.mark.parametrize("seed", BUSHY_SEEDS)
def test_cpu_reverse_mode_on_trees_with_five_branch_points(tmp_path, seed, cpu_library):
    """The backward walks take up to six branch points (the slots the control word's source field addresses); until round 4 the
    limit was four and these trees were refused.  Reverse-mode inverse dynamics on the CPU build against a central difference of
    the fp64 oracle along a random direction."""
    m = tree_model(tmp_path, seed, "cpu")
    dw = m._dynamics_walk()
    assert dw.program.n_slots == 5 and dw.program.backward_ok
    grads, (q, qd, qdd, w) = _bushy_gradients(m, seed, "cpu")
    orc = Oracle(m._spec)
    q64, qd64, qdd64 = (t.numpy().astype(np.float64) for t in (q, qd, qdd))
    dirs = [np.random.default_rng(seed + 1 + i).standard_normal(q64.shape) for i in range(3)]
    f = lambda a, b, c: float((orc.rnea(a, b, c, True, True, np.float64) * w.numpy()).sum())
    h = 1e-5
    fd = (f(q64 + h * dirs[0], qd64 + h * dirs[1], qdd64 + h * dirs[2]) - f(q64 - h * dirs[0], qd64 - h * dirs[1], qdd64 - h * dirs[2])) / (2 * h)
    an = sum(float((g.numpy() * d).sum()) for g, d in zip(grads[:3], dirs))
    assert abs(an - fd) <= 2e-3 * (1 + abs(fd)), (seed, an, fd)
    assert torch.isfinite(grads[3]).all() and float(grads[3].abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", BUSHY_SEEDS[:2])
def test_gpu_reverse_mode_on_trees_with_five_branch_points(tmp_path, seed, cpu_library):
    """The same gradients from the kernels (rnea_backward_kernel / fk_backward_kernel with five slots in LDS) against the CPU build
    of the same walks."""
    host, _ = _bushy_gradients(tree_model(tmp_path, seed, "cpu"), seed, "cpu")
    dev_, _ = _bushy_gradients(tree_model(tmp_path, seed, "cuda"), seed, "cuda")
    for a, b in zip(dev_, host):
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), seed
