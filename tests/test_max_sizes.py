"""Large robots: synthetic serial chains of 22, 30 and 45 joints (walks of 23, 31 and 46 links; none of the shipped robots'
chains comes close).  Every forward kernel family against the fp64 oracle — they loop over the links, so the only limit is
the LDS their per-link records need —, the backward kernels (<= 64 links) against the host emulation.
CPU (not gpu): the kernel arithmetic (host emulation) on the same robots.
"""
import contextlib
import ctypes
import io
import os

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd import DifferentiableRobotModel
from differentiable_robot_model_amd.flatten import UnsupportedRobotError, build_walk
from helpers import TOL_JAC, TOL_POS, TOL_QUAT, TOL_TAU, quat_close, sample_states
from oracle import Oracle
from test_host_emu import _ptr, emu, host_walk  # noqa: F401  (emu is a fixture)

SIZES = [22, 30, 45]


def chain_urdf(n_joints: int) -> str:
    """A serial arm of n revolute joints about +-x / +-y / +-z with tilted, offset joint frames and full inertias."""
    rng = np.random.default_rng(1000 + n_joints)
    axes = ["1 0 0", "0 1 0", "0 0 1", "-1 0 0", "0 -1 0", "0 0 -1"]
    out = ['<?xml version="1.0"?>', '<robot name="chain%d">' % n_joints, '  <link name="base"/>']
    parent = "base"
    for i in range(n_joints + 1):
        name = "link%d" % i if i < n_joints else "tip"
        A = rng.standard_normal((3, 3)) * 0.02
        I = A @ A.T + np.eye(3) * 0.003
        m, c = (0.2 + rng.random() * 0.5, rng.standard_normal(3) * 0.03) if i < n_joints else (0.05, np.zeros(3))
        out.append('  <link name="%s"><inertial><origin xyz="%.5f %.5f %.5f" rpy="0 0 0"/><mass value="%.5f"/>'
                   '<inertia ixx="%.6f" ixy="%.6f" ixz="%.6f" iyy="%.6f" iyz="%.6f" izz="%.6f"/></inertial></link>'
                   % (name, c[0], c[1], c[2], m, I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))
        xyz, rpy = rng.standard_normal(3) * 0.05 + np.array([0, 0, 0.08]), rng.standard_normal(3) * 0.6
        if i < n_joints:
            out.append('  <joint name="j%d" type="revolute"><parent link="%s"/><child link="%s"/>'
                       '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/><axis xyz="%s"/>'
                       '<limit effort="10" lower="-2.5" upper="2.5" velocity="3"/><dynamics damping="%.3f"/></joint>'
                       % (i, parent, name, xyz[0], xyz[1], xyz[2], rpy[0], rpy[1], rpy[2], axes[int(rng.integers(6))],
                          rng.random() * 0.2))
        else:
            out.append('  <joint name="jtip" type="fixed"><parent link="%s"/><child link="tip"/>'
                       '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/></joint>' % (parent, xyz[0], xyz[1], xyz[2], rpy[0], rpy[1], rpy[2]))
        parent = name
    out.append("</robot>")
    return "\n".join(out)


def chain_model(tmp_path, n_joints, device="cpu"):
    path = os.path.join(str(tmp_path), "chain%d.urdf" % n_joints)
    with open(path, "w") as f:
        f.write(chain_urdf(n_joints))
    with contextlib.redirect_stdout(io.StringIO()):
        return DifferentiableRobotModel(path, device=device, reference_compat=False)


def test_walk_capacity_follows_the_robot(tmp_path):
    m = chain_model(tmp_path, 45)          # 45 moving links + tip = 46 ops (the root is not an op)
    tree = build_walk(m._spec, whole_tree=True)
    assert tree.n_ops == 46 and tree.capacity == 48 and tree.backward_ok
    assert build_walk(chain_model(tmp_path, 64)._spec, whole_tree=True).backward_ok is False   # 65 links: forward only


@pytest.mark.parametrize("n", SIZES)
def test_emu_long_chain_vs_oracle(emu, tmp_path, n):
    m = chain_model(tmp_path, n)
    orc, B = Oracle(m._spec), 5
    q, qd, qdd = sample_states(m, B, seed=n)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    tip = len(m._bodies) - 1
    prog = build_walk(m._spec, targets=[tip])
    assert prog.capacity == (n + 1 + 3) // 4 * 4
    walk, _keep = host_walk(m, prog)
    pos, quat = np.zeros((B, 3), np.float32), np.zeros((B, 4), np.float32)
    lin, ang = np.zeros((B, 3, n), np.float32), np.zeros((B, 3, n), np.float32)
    assert emu.emu_fk_jacobian(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(pos), _ptr(quat), _ptr(lin), _ptr(ang)) == 0
    rp, rq, rl, ra = orc.fk_jacobian(q64, tip, np.float64)
    assert np.abs(pos - rp).max() <= 4 * TOL_POS["atol"] and np.abs(lin - rl).max() <= 4 * TOL_JAC["atol"]
    assert np.abs(ang - ra).max() <= 4 * TOL_JAC["atol"] and quat_close(quat, rq, 4 * TOL_QUAT["atol"])[0]
    tree = build_walk(m._spec, whole_tree=True)
    twalk, _keep2 = host_walk(m, tree)
    tau = np.zeros((B, n), np.float32)
    assert emu.emu_rnea(ctypes.byref(twalk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(tau)) == 0
    assert np.allclose(tau, orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU)
    H = np.zeros((B, n, n), np.float32)
    assert emu.emu_crba(ctypes.byref(twalk), _ptr(q), ctypes.c_int64(B), _ptr(H)) == 0
    assert np.allclose(H, orc.mass_matrix(q64, False, False, np.float64), **TOL_TAU)
    acc = np.zeros((B, n), np.float32)
    assert emu.emu_forward_dynamics(ctypes.byref(twalk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(acc)) == 0
    ref = orc.forward_dynamics(q64, qd64, qdd64, True, True, np.float64)
    assert (np.abs(acc - ref) / (1.0 + np.abs(ref))).max() <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("B", [64, 130])
def test_gpu_long_chain_forward_kernels_vs_oracle(tmp_path, n, B):
    m = chain_model(tmp_path, n, "cuda")
    orc = Oracle(m._spec)
    q, qd, qdd = sample_states(m, B, seed=n + B)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    dev = lambda a: torch.from_numpy(a).cuda()
    L = len(m._bodies)
    poses = m.compute_forward_kinematics_all_links(dev(q))
    op, oq = orc.fk(q64, list(range(L)), np.float64)
    for i, body in enumerate(m._bodies):
        p, r = poses[body.name]
        assert np.abs(p.cpu().numpy() - op[:, i]).max() <= 4 * TOL_POS["atol"], body.name
        assert quat_close(r.cpu().numpy(), oq[:, i], 4 * TOL_QUAT["atol"])[0], body.name
    pos, quat, lin, ang = m.compute_fk_and_jacobian(dev(q), "tip")
    rp, rq, rl, ra = orc.fk_jacobian(q64, L - 1, np.float64)
    assert np.abs(pos.cpu().numpy() - rp).max() <= 4 * TOL_POS["atol"]
    assert np.abs(lin.cpu().numpy() - rl).max() <= 4 * TOL_JAC["atol"] and np.abs(ang.cpu().numpy() - ra).max() <= 4 * TOL_JAC["atol"]
    # (rounding accumulates along a chain: the tolerances of the 8-link robots, times 4 here as for the poses above)
    tol = dict(atol=4 * TOL_TAU["atol"], rtol=4 * TOL_TAU["rtol"])
    tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd))
    assert np.allclose(tau.cpu().numpy(), orc.rnea(q64, qd64, qdd64, True, True, np.float64), **tol)
    H = m.compute_lagrangian_inertia_matrix(dev(q))
    assert np.allclose(H.cpu().numpy(), orc.mass_matrix(q64, False, False, np.float64), **tol)
    acc = m.compute_forward_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=True, use_damping=True)
    ref = orc.forward_dynamics(q64, qd64, qdd64, True, True, np.float64)
    # cond(H) of a long chain of light links is 1e5 .. 1e6; the articulated-body walk never forms H and stays at 1e-4 (the
    # host emulation gives the same figure; the reference's recursion evaluated in fp32 is at 2e-4 .. 1e-3 here)
    assert (np.abs(acc.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max() <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES)
def test_gpu_long_chain_backward_kernels_vs_emu(emu, tmp_path, n):
    from differentiable_robot_model_amd import backend
    mc, m = chain_model(tmp_path, n), chain_model(tmp_path, n, "cuda")
    B, tip = 70, len(mc._bodies) - 1
    q, qd, qdd = sample_states(mc, B, seed=3 * n)
    rng = np.random.default_rng(n)
    dev = lambda a: torch.from_numpy(a).cuda()
    # Jacobian + position gradients through the chain walk (JAC form of K5)
    prog = build_walk(mc._spec, targets=[tip])
    walk, _keep = host_walk(mc, prog)
    gpos, glin, gang = (rng.standard_normal(s).astype(np.float32) for s in ((B, 3), (B, 3, n), (B, 3, n)))
    mask = (1 << (prog.n_ops - 1)) | (1 << 3)
    gq = np.full((B, n), np.nan, np.float32); gops = np.full((prog.capacity, 32), np.nan, np.float32)
    assert emu.emu_fk_jacobian_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(gpos), _ptr(glin), _ptr(gang),
                                        ctypes.c_uint64(mask), _ptr(gq), _ptr(gops)) == 0
    dw = m._get_walk(("chain", tip), targets=[tip])
    got_q, got_ops = backend.fk_jacobian_backward(dw.program, m._ops_f(dw), dw.ops_i, dev(q), dev(gpos), dev(glin), dev(gang),
                                                  n, mask, True)
    assert np.allclose(got_q.cpu().numpy(), gq, atol=1e-4, rtol=1e-4)
    assert np.abs(got_ops.cpu().numpy() - gops).max() <= 2e-4 * max(np.abs(gops).max(), 1e-6)
    # positions only (K5) and the RNEA backward (K7) over the whole tree
    got_q, _ = backend.fk_backward(dw.program, m._ops_f(dw), dw.ops_i, dev(q), dev(gpos).reshape(B, 1, 3), 1, n, 0, True)
    assert emu.emu_fk_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), 1, _ptr(gpos), ctypes.c_uint64(0), _ptr(gq), None) == 0
    assert np.allclose(got_q.cpu().numpy(), gq, atol=1e-4, rtol=1e-4)
    tree = build_walk(mc._spec, whole_tree=True)
    twalk, _keep2 = host_walk(mc, tree)
    gtau = rng.standard_normal((B, n)).astype(np.float32)
    eq, eqd, eqdd = (np.full((B, n), np.nan, np.float32) for _ in range(3))
    eops = np.full((tree.capacity, 32), np.nan, np.float32)
    tmask = (1 << 2) | (1 << (tree.n_ops - 2))
    assert emu.emu_rnea_backward(ctypes.byref(twalk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(gtau),
                                 ctypes.c_uint64(tmask), _ptr(eq), _ptr(eqd), _ptr(eqdd), _ptr(eops)) == 0
    dt = m._get_walk(("tree",), whole_tree=True)
    gin, gops_t = backend.rnea_backward(dt.program, m._ops_f(dt), dt.ops_i, dev(q), dev(qd), dev(qdd), dev(gtau), True, True,
                                        n, tmask, True)
    for got, ref in zip(gin, (eq, eqd, eqdd)):
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1.0)
    assert np.abs(gops_t.cpu().numpy() - eops).max() <= 5e-4 * max(np.abs(eops).max(), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("DRM_FUZZ_SEEDS", "8"))))
def test_gpu_random_robot_backward_kernels_vs_emu(emu, seed):
    """K5 (positions of a random set of targets, and the JAC form on a random chain) and K7 on a random shipped robot at a
    random batch size, random learnable ops, against the host emulation of the same sweeps."""
    from differentiable_robot_model_amd import backend
    from helpers import ALL_ROBOTS, load_model
    rng = np.random.default_rng(4242 + seed)
    robot = ALL_ROBOTS[int(rng.integers(len(ALL_ROBOTS)))]
    mc, m = load_model(robot), load_model(robot, "cuda")
    n, L = mc._n_dofs, len(mc._bodies)
    B = int(rng.choice([rng.integers(1, 130), rng.integers(130, 1500)]))
    q, qd, qdd = sample_states(mc, B, seed=seed)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    # ---- K5, several targets
    T = int(rng.integers(1, min(4, L - 1) + 1))
    targets = sorted(int(t) for t in rng.choice(np.arange(1, L), size=T, replace=False))
    prog = build_walk(mc._spec, targets=targets)
    if prog.slots_unique:
        walk, _k = host_walk(mc, prog)
        gpos = rng.standard_normal((B, T, 3)).astype(np.float32)
        mask = int(rng.integers(0, 1 << prog.n_ops))
        gq = np.full((B, n), np.nan, np.float32); gops = np.full((prog.capacity, 32), np.nan, np.float32)
        assert emu.emu_fk_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), T, _ptr(gpos), ctypes.c_uint64(mask),
                                   _ptr(gq), _ptr(gops) if mask else None) == 0
        dw = m._get_walk(("fk", tuple(targets)), targets=targets)
        got_q, got_ops = backend.fk_backward(dw.program, m._ops_f(dw), dw.ops_i, dev(q), dev(gpos), T, n, mask, True)
        assert np.abs(got_q.cpu().numpy() - gq).max() <= 1e-4 * max(1.0, np.abs(gq).max()), (robot, B, targets)
        if mask:
            assert np.abs(got_ops.cpu().numpy() - gops).max() <= 3e-4 * max(np.abs(gops).max(), 1e-6) * max(1.0, B / 256)
    # ---- K5, JAC form on the chain to a random link
    link = int(rng.integers(1, L))
    cprog = build_walk(mc._spec, targets=[link])
    cwalk, _k2 = host_walk(mc, cprog)
    gp, gl, ga = (rng.standard_normal(s).astype(np.float32) for s in ((B, 3), (B, 3, n), (B, 3, n)))
    gq = np.full((B, n), np.nan, np.float32)
    assert emu.emu_fk_jacobian_backward(ctypes.byref(cwalk), _ptr(q), ctypes.c_int64(B), _ptr(gp), _ptr(gl), _ptr(ga),
                                        ctypes.c_uint64(0), _ptr(gq), None) == 0
    dc = m._get_walk(("chain", link), targets=[link])
    got_q, _ = backend.fk_jacobian_backward(dc.program, m._ops_f(dc), dc.ops_i, dev(q), dev(gp), dev(gl), dev(ga), n, 0, True)
    assert np.abs(got_q.cpu().numpy() - gq).max() <= 1e-4 * max(1.0, np.abs(gq).max()), (robot, B, link)
    # ---- K7 over the whole tree
    tree = build_walk(mc._spec, whole_tree=True)
    if tree.slots_unique:
        twalk, _k3 = host_walk(mc, tree)
        gtau = rng.standard_normal((B, n)).astype(np.float32)
        tmask = int(rng.integers(0, 1 << min(tree.n_ops, 63)))
        eq, eqd, eqdd = (np.full((B, n), np.nan, np.float32) for _ in range(3))
        eops = np.full((tree.capacity, 32), np.nan, np.float32)
        assert emu.emu_rnea_backward(ctypes.byref(twalk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(gtau),
                                     ctypes.c_uint64(tmask), _ptr(eq), _ptr(eqd), _ptr(eqdd), _ptr(eops) if tmask else None) == 0
        dt = m._get_walk(("tree",), whole_tree=True)
        gin, gops_t = backend.rnea_backward(dt.program, m._ops_f(dt), dt.ops_i, dev(q), dev(qd), dev(qdd), dev(gtau), True, True,
                                            n, tmask, True)
        for got, ref in zip(gin, (eq, eqd, eqdd)):
            assert np.abs(got.cpu().numpy() - ref).max() <= 3e-4 * max(np.abs(ref).max(), 1.0), (robot, B)
        if tmask:
            assert np.abs(gops_t.cpu().numpy() - eops).max() <= 1e-3 * max(np.abs(eops).max(), 1e-6) * max(1.0, B / 256)


# ---------------------------------------------------------------------------------------------------------------------
# A learnable link BEYOND op 32 (the backward kernels' param_mask is 64 bits wide: ABI 7; a 32-bit mask silently aliased
# op k to op k - 32).  Yardstick: central differences of the fp64 oracle on the perturbed robot description.
# ---------------------------------------------------------------------------------------------------------------------
def _chain45_losses(spec, tip, q64, qd64, qdd64, Wp, Wt):
    orc = Oracle(spec)
    pos, _ = orc.fk(q64, [tip], np.float64)
    tau = orc.rnea(q64, qd64, qdd64, True, True, np.float64)
    return float((Wp * pos[:, 0]).sum()), float((Wt * tau).sum())


def _chain45_fd(spec, link, field, tip, args, Wp, Wt, h=1e-3):
    import dataclasses
    arr = np.asarray(getattr(spec, field), np.float32)
    row = arr[link].reshape(-1)
    g = np.zeros((2, row.size))
    for e in range(row.size):
        vals = []
        for sgn in (+1, -1):
            a2 = arr.copy()
            a2.reshape(arr.shape[0], -1)[link, e] = np.float32(row[e] + sgn * h)
            vals.append((float(a2.reshape(arr.shape[0], -1)[link, e]),
                         _chain45_losses(dataclasses.replace(spec, **{field: a2}), tip, *args, Wp, Wt)))
        for j in range(2):
            g[j, e] = (vals[0][1][j] - vals[1][1][j]) / (vals[0][0] - vals[1][0])
    return g


@pytest.mark.gpu
def test_gpu_learnable_link_beyond_op_32_vs_oracle_differences(tmp_path):
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedScalar, UnconstrainedTensor
    m = chain_model(tmp_path, 45, "cuda")
    spec, tip, n, B = m._spec, len(m._bodies) - 1, 45, 6
    link = m._name_to_idx_map["link40"]
    t_mod = UnconstrainedTensor(1, 3, init_tensor=torch.from_numpy(np.asarray(spec.trans[link], np.float32).reshape(1, 3).copy()))
    m_mod = UnconstrainedScalar(init_val=float(spec.mass[link]))
    m.make_link_param_learnable("link40", "trans", t_mod)
    m.make_link_param_learnable("link40", "mass", m_mod)
    dw = m._dynamics_walk()
    mask = m._learnable_op_mask(dw)
    assert mask and mask >> 32, "the learnable link must sit beyond op 32 for this test to mean anything (mask %x)" % mask
    q, qd, qdd = sample_states(m, B, seed=77)
    rng = np.random.default_rng(5)
    Wp, Wt = rng.standard_normal((B, 3)), rng.standard_normal((B, n))
    args = [a.astype(np.float64) for a in (q, qd, qdd)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    pos, _ = m.compute_forward_kinematics(dev(q), "tip")
    loss_p = (dev(Wp) * pos).sum()
    loss_p.backward()
    g_trans_pos = t_mod.param.grad.cpu().numpy().reshape(-1).copy()
    m.zero_grad()
    tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=True, use_damping=True)
    loss_t = (dev(Wt) * tau).sum()
    loss_t.backward()
    g_trans_tau = t_mod.param.grad.cpu().numpy().reshape(-1).copy()
    g_mass_tau = m_mod.param.grad.cpu().numpy().reshape(-1).copy()
    want = _chain45_losses(spec, tip, *args, Wp, Wt)
    assert abs(loss_p.item() - want[0]) <= 2e-4 * max(1.0, abs(want[0])) and abs(loss_t.item() - want[1]) <= 2e-3 * max(1.0, abs(want[1]))
    fd_t = _chain45_fd(spec, link, "trans", tip, args, Wp, Wt)
    fd_m = _chain45_fd(spec, link, "mass", tip, args, Wp, Wt)
    close = lambda got, ref: np.abs(got - ref).max() <= 5e-3 * max(np.abs(ref).max(), 5e-3)
    assert np.abs(fd_t[0]).max() > 1e-2 and np.abs(fd_m[1]).max() > 1e-2          # the gradients are not trivially zero
    assert close(g_trans_pos, fd_t[0]), (g_trans_pos, fd_t[0])
    assert close(g_trans_tau, fd_t[1]), (g_trans_tau, fd_t[1])
    assert close(g_mass_tau, fd_m[1]), (g_mass_tau, fd_m[1])


def test_emu_learnable_op_beyond_32_gets_its_own_gradient(emu, tmp_path):
    """CPU: the host emulation of K5 with ONLY bit 40 of the mask set writes the constant gradient of op 40 and nothing
    into op 8 (= 40 - 32, where a 32-bit shift would have landed) or any other row."""
    m = chain_model(tmp_path, 45)
    spec, tip, n, B = m._spec, len(m._bodies) - 1, 45, 4
    prog = build_walk(spec, targets=[tip])
    walk, _keep = host_walk(m, prog)
    k = 40
    q, qd, qdd = sample_states(m, B, seed=78)
    rng = np.random.default_rng(6)
    gpos = rng.standard_normal((B, 1, 3)).astype(np.float32)
    gq = np.full((B, n), np.nan, np.float32); gops = np.full((prog.capacity, 32), np.nan, np.float32)
    assert emu.emu_fk_backward(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), 1, _ptr(gpos), ctypes.c_uint64(1 << k),
                               _ptr(gq), _ptr(gops)) == 0
    assert np.abs(gops[k]).max() > 1e-3 and np.abs(gops[k - 32]).max() == 0.0
    rows = [r for r in range(prog.capacity) if r != k]
    assert np.abs(gops[rows]).max() == 0.0


def comb_urdf(n_teeth: int) -> str:
    """A spine of n_teeth revolute joints, every spine link carrying a two-joint side branch: n_teeth branch points, all of them open
    while the deepest tooth is walked."""
    rng = np.random.default_rng(77)
    axes = ["1 0 0", "0 1 0", "0 0 1", "-1 0 0", "0 -1 0", "0 0 -1"]
    out = ['<?xml version="1.0"?>', '<robot name="comb%d">' % n_teeth, '  <link name="base"/>']

    def link(name):
        out.append('  <link name="%s"><inertial><origin xyz="0.01 0.0 0.02" rpy="0 0 0"/><mass value="0.3"/>'
                   '<inertia ixx="0.003" ixy="0" ixz="0" iyy="0.003" iyz="0" izz="0.002"/></inertial></link>' % name)

    def joint(name, parent, child):
        xyz, rpy = rng.standard_normal(3) * 0.05 + np.array([0, 0, 0.08]), rng.standard_normal(3) * 0.5
        out.append('  <joint name="%s" type="revolute"><parent link="%s"/><child link="%s"/><origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/>'
                   '<axis xyz="%s"/><limit effort="10" lower="-2" upper="2" velocity="3"/></joint>'
                   % (name, parent, child, xyz[0], xyz[1], xyz[2], rpy[0], rpy[1], rpy[2], axes[int(rng.integers(6))]))
    parent = "base"
    for i in range(n_teeth):
        link("spine%d" % i); joint("js%d" % i, parent, "spine%d" % i)
        link("tooth%da" % i); joint("jt%da" % i, "spine%d" % i, "tooth%da" % i)
        link("tooth%db" % i); joint("jt%db" % i, "tooth%da" % i, "tooth%db" % i)
        parent = "spine%d" % i
    link("tip"); joint("jtip", parent, "tip")
    out.append("</robot>")
    return "\n".join(out)


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_gradients_of_every_links_pose_on_a_tree_with_more_branch_points_than_the_backward_walk_takes(tmp_path, device):
    """The backward kernels walk trees with up to 6 branch points open at once (the packed control word addresses six save slots); a
    many-target FK call that needs gradients on a wider tree goes target by target over root -> link chains instead of refusing —
    same poses as the one-launch forward, gradients against central differences of it."""
    path = os.path.join(str(tmp_path), "comb.urdf")
    with open(path, "w") as f:
        f.write(comb_urdf(9))
    with contextlib.redirect_stdout(io.StringIO()):
        m = DifferentiableRobotModel(path, device=device)
    n = m._n_dofs
    assert n == 28
    g = torch.Generator().manual_seed(3)
    q = ((torch.rand(5, n, generator=g) - 0.5) * 2.0).to(device)
    with torch.no_grad():
        want = m.compute_forward_kinematics_all_links(q)
    leaves = [i for i, b in enumerate(m._bodies) if b.name.endswith("b") or b.name == "tip"]
    walk = m._get_walk(("fk", tuple(sorted(leaves))), targets=sorted(leaves))
    assert not walk.program.backward_ok and walk.program.n_slots > 6       # (the case this test is about)
    x = q.clone().requires_grad_(True)
    got = m.compute_forward_kinematics_all_links(x)
    w = {name: torch.randn(5, 3, generator=g).to(device) for name in got}
    for name in got:
        assert torch.allclose(got[name][0].detach(), want[name][0], atol=2e-6) and torch.allclose(got[name][1].detach(), want[name][1], atol=2e-6)
    loss = sum((w[name] * got[name][0]).sum() for name in got)
    (grad,) = torch.autograd.grad(loss, x)
    h = 2e-3
    with torch.no_grad():
        num = torch.zeros_like(q)
        for d in range(n):
            e = torch.zeros(n, device=device); e[d] = h
            hi, lo = m.compute_forward_kinematics_all_links(q + e), m.compute_forward_kinematics_all_links(q - e)
            num[:, d] = sum((w[name] * (hi[name][0] - lo[name][0])).sum(dim=1) for name in got) / (2 * h)
    assert float((grad - num).abs().max()) <= 5e-3 * max(1.0, float(num.abs().max()))
