"""Parity tests proper (-m gpu): the HIP path, called through the public API -> ctypes -> C ABI, against
  * the fp64 oracle on seeded inputs (every shipped robot, ragged / tiny / multi-tile batches),
  * the committed golden fixtures of the reference (its test matrix and batch shapes),
  * size-independent properties at BASELINE.json's full sizes (65 536 and 2^20).
Tolerances (fp32 path; the reference's own fp32 noise floor is ~2e-7): helpers.TOL_*.
"""
import numpy as np
import pytest
import torch

from helpers import (ALL_ROBOTS, GOLDEN_ROBOTS, REFERENCE_BATCH_SHAPES, REFERENCE_TEST_MATRIX, TOL_JAC, TOL_POS,
                     TOL_QUAT, TOL_TAU, load_golden, load_model, max_err, quat_close, sample_states)
from oracle import Oracle

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_library():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    from differentiable_robot_model_amd import backend
    lib = backend.load_library()            # raises if the in-tree .so is missing: no silent fallback
    assert backend.LIB_PATH.endswith("csrc/libdrm_hip.so") and lib.drm_abi_version() == backend.ABI_VERSION


# ------------------------------------------------------------------ vs the fp64 oracle
@pytest.mark.parametrize("robot", ALL_ROBOTS)
@pytest.mark.parametrize("B", [1, 63, 64, 65, 257])
def test_fk_jacobian_rnea_vs_oracle(robot, B):
    m = load_model(robot, "cuda")
    orc = Oracle(m._spec)
    L = len(m._bodies)
    q, qd, qdd = sample_states(m, B, seed=100 + B)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    # all-links FK (branching walks, save slots, target un-permutation)
    poses = m.compute_forward_kinematics_all_links(dev(q))
    op, oq = orc.fk(q64, list(range(L)), np.float64)
    for i, body in enumerate(m._bodies):
        p, r = poses[body.name]
        assert max_err(host(p), op[:, i]) <= TOL_POS["atol"], (robot, body.name)
        ok, _ = quat_close(host(r), oq[:, i], TOL_QUAT["atol"])
        assert ok, (robot, body.name)
    # FK + Jacobian of a few links (first moving link, a middle one, the last one)
    for link in sorted({1, L // 2, L - 1}):
        name = m._bodies[link].name
        pos, quat, lin, ang = m.compute_fk_and_jacobian(dev(q), name)
        rp, rq, rl, ra = orc.fk_jacobian(q64, link, np.float64)
        assert max_err(host(pos), rp) <= TOL_POS["atol"]
        assert max_err(host(lin), rl) <= TOL_JAC["atol"] and max_err(host(ang), ra) <= TOL_JAC["atol"]
        ok, _ = quat_close(host(quat), rq, TOL_QUAT["atol"])
        assert ok
        lin2, ang2 = m.compute_endeffector_jacobian(dev(q), name)
        assert torch.equal(lin2, lin) and torch.equal(ang2, ang)
    # RNEA, all flag combinations + the qdd = 0 entry point
    for grav in (False, True):
        for damp in (False, True):
            tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=grav, use_damping=damp)
            ref = orc.rnea(q64, qd64, qdd64, grav, damp, np.float64)
            assert np.allclose(host(tau), ref, **TOL_TAU), (robot, grav, damp, np.abs(host(tau) - ref).max())
    nle = m.compute_non_linear_effects(dev(q), dev(qd))
    assert np.allclose(host(nle), orc.rnea(q64, qd64, np.zeros_like(q64), True, True, np.float64), **TOL_TAU)


# ------------------------------------------------------------------ vs the reference's golden fixtures
@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
@pytest.mark.parametrize("tag", ["slow", "fast"])
def test_against_reference_golden(robot, links, tag):
    g = load_golden(robot)
    m = load_model(robot, "cuda")
    q, qd, qdd = (dev(g[tag + k]) for k in ("_q", "_qd", "_qdd"))
    for link in links:
        for recursive in (False, True):
            pos, quat = m.compute_forward_kinematics(q, link, recursive=recursive)
            assert max_err(host(pos), g["%s_pos_%s" % (tag, link)]) <= TOL_POS["atol"]
            ok, _ = quat_close(host(quat), g["%s_quat_%s" % (tag, link)], TOL_QUAT["atol"])
            assert ok
        lin, ang = m.compute_endeffector_jacobian(q, link)
        assert max_err(host(lin), g["%s_lin_%s" % (tag, link)]) <= TOL_JAC["atol"]
        assert max_err(host(ang), g["%s_ang_%s" % (tag, link)]) <= TOL_JAC["atol"]
    for grav in (0, 1):
        for damp in (0, 1):
            tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=bool(grav), use_damping=bool(damp))
            assert np.allclose(host(tau), g["%s_tau_g%d_d%d" % (tag, grav, damp)], **TOL_TAU)


@pytest.mark.parametrize("robot,links", REFERENCE_TEST_MATRIX)
@pytest.mark.parametrize("shape", REFERENCE_BATCH_SHAPES)
def test_reference_batch_shapes(robot, links, shape):
    """The reference's batch shapes (), (1,), (3,), (6,), (7,) (test_kinematics_dynamics.py:54-61)."""
    g = load_golden(robot)
    m = load_model(robot, "cuda")
    n = m._n_dofs
    rows = shape[0] if shape else 1
    sl = (lambda a: a[0]) if not shape else (lambda a: a[:rows])
    q, qd, qdd = (dev(sl(g["slow" + k])) for k in ("_q", "_qd", "_qdd"))
    link = links[0]
    pos, quat = m.compute_forward_kinematics(q, link)
    lin, ang = m.compute_endeffector_jacobian(q, link)
    tau = m.compute_inverse_dynamics(q, qd, qdd)
    assert tuple(pos.shape) == shape + (3,) and tuple(quat.shape) == shape + (4,)
    assert tuple(lin.shape) == shape + (3, n) and tuple(ang.shape) == shape + (3, n) and tuple(tau.shape) == shape + (n,)
    assert max_err(host(pos), sl(g["slow_pos_" + link])) <= TOL_POS["atol"]
    assert max_err(host(lin), sl(g["slow_lin_" + link])) <= TOL_JAC["atol"]
    assert np.allclose(host(tau), sl(g["slow_tau_g1_d1"]), **TOL_TAU)
    if not shape:
        assert max_err(host(pos), g["slow_pos_unbatched"]) <= TOL_POS["atol"]
    allp = m.compute_forward_kinematics_all_links(q)     # dict results keep the batch dim
    assert tuple(allp[link][0].shape) == (rows, 3)


# ------------------------------------------------------------------ edge cases of the boundary
def test_unaligned_and_noncontiguous_inputs():
    m = load_model("panda_no_gripper", "cuda")
    orc = Oracle(m._spec)
    q, qd, qdd = sample_states(m, 130, seed=3)
    big = dev(np.concatenate([np.zeros((1, 7), np.float32), q]))
    view = big[1:]                                    # contiguous but only 4-byte aligned
    assert view.data_ptr() % 16 != 0
    pos, quat, lin, ang = m.compute_fk_and_jacobian(view, "panda_virtual_ee_link")
    rp, rq, rl, ra = orc.fk_jacobian(q.astype(np.float64), 8, np.float64)
    assert max_err(host(pos), rp) <= TOL_POS["atol"] and max_err(host(lin), rl) <= TOL_JAC["atol"]
    wide = dev(np.concatenate([q, q], axis=1))[:, :7]  # non-contiguous rows
    assert not wide.is_contiguous()
    tau = m.compute_inverse_dynamics(wide, dev(qd), dev(qdd))
    assert np.allclose(host(tau), orc.rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64),
                                           True, True, np.float64), **TOL_TAU)
    pos64, _ = m.compute_forward_kinematics(dev(q).double(), "panda_virtual_ee_link")  # dtype is normalised
    assert pos64.dtype == torch.float32 and max_err(host(pos64), rp) <= TOL_POS["atol"]


def test_errors_on_gpu_model():
    m = load_model("panda_no_gripper", "cuda")
    with pytest.raises(AssertionError):                # CPU tensor into a GPU model (robot_model.py:38-40)
        m.compute_forward_kinematics(torch.zeros(2, 7), "panda_virtual_ee_link")
    with pytest.raises(AssertionError):
        m.compute_endeffector_jacobian(torch.zeros(2, 6).cuda(), "panda_virtual_ee_link")
    with pytest.raises(KeyError):
        m.compute_forward_kinematics(torch.zeros(2, 7).cuda(), "nope")
    empty = m.compute_inverse_dynamics(torch.zeros(0, 7).cuda(), torch.zeros(0, 7).cuda(), torch.zeros(0, 7).cuda())
    assert tuple(empty.shape) == (0, 7)
    root_pos, root_quat = m.compute_forward_kinematics(torch.rand(5, 7).cuda(), "panda_link0")  # the root link
    assert torch.equal(root_pos, torch.zeros(5, 3).cuda()) and torch.equal(root_quat[:, 3], torch.ones(5).cuda())
    lin, ang = m.compute_endeffector_jacobian(torch.rand(5, 7).cuda(), "panda_link0")
    assert not lin.any() and not ang.any()


def test_plan_and_hipgraph_replay_match_direct_call():
    m = load_model("panda_no_gripper", "cuda")
    q, _, _ = sample_states(m, 4096, seed=9)
    qd = dev(q)
    ref = m.compute_fk_and_jacobian(qd, "panda_virtual_ee_link")
    plan = m.plan_fk_and_jacobian(qd, "panda_virtual_ee_link")
    plan.launch()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(plan.outputs(), ref))
    for t in plan.outputs():
        t.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        plan.launch()
    for t in plan.outputs():
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(plan.outputs(), ref))


def test_inverse_dynamics_plan_and_hipgraph_replay_match_direct_call():
    m = load_model("iiwa7", "cuda")
    q, qd, qdd = (dev(a) for a in sample_states(m, 3000, seed=12))
    for acc in (qdd, None):
        ref = m.compute_inverse_dynamics(q, qd, qdd) if acc is not None else m.compute_non_linear_effects(q, qd)
        plan = m.plan_inverse_dynamics(q, qd, acc)
        plan.launch()
        torch.cuda.synchronize()
        assert torch.equal(plan.tau, ref)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            plan.launch()
        plan.tau.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(plan.tau, ref)
        # new inputs written into the plan's buffers are what the replay sees
        plan.q.mul_(0.5)
        graph.replay()
        torch.cuda.synchronize()
        want = m.compute_inverse_dynamics(plan.q, qd, qdd) if acc is not None else m.compute_non_linear_effects(plan.q, qd)
        assert torch.equal(plan.tau, want)
        plan.q.mul_(2.0)


# ------------------------------------------------------------------ full-size, size-independent properties
@pytest.mark.parametrize("B", [65536, 1 << 20])
def test_full_size_properties(B):
    m = load_model("panda_no_gripper", "cuda")
    ee = "panda_virtual_ee_link"
    lim = m.get_joint_limits()
    lo = torch.tensor([j["lower"] for j in lim]).cuda(); hi = torch.tensor([j["upper"] for j in lim]).cuda()
    gen = torch.Generator(device="cuda").manual_seed(B)
    q = lo + (hi - lo) * torch.rand(B, 7, device="cuda", generator=gen)
    qd = torch.rand(B, 7, device="cuda", generator=gen) * 2 - 1
    a1 = torch.rand(B, 7, device="cuda", generator=gen) * 4 - 2
    a2 = torch.rand(B, 7, device="cuda", generator=gen) * 4 - 2
    pos, quat, lin, ang = m.compute_fk_and_jacobian(q, ee)
    # determinism: a second launch is bit-identical
    pos2, quat2, lin2, ang2 = m.compute_fk_and_jacobian(q, ee)
    assert torch.equal(pos, pos2) and torch.equal(quat, quat2) and torch.equal(lin, lin2) and torch.equal(ang, ang2)
    # unit quaternions and unit joint axes; last joint's own column of lin_jac: z x (p_e - p_7) with p_e - p_7 || z
    assert (quat.norm(dim=1) - 1).abs().max() < 1e-5
    assert (ang.norm(dim=1) - 1).abs().max() < 1e-5
    assert lin[:, :, 6].abs().max() < 1e-5
    # joint 1 of the Panda turns about the world z axis through the origin: shifting q1 by d rotates the pose
    d = 0.37
    qs = q.clone(); qs[:, 0] += d
    pos_s, _, lin_s, _ = m.compute_fk_and_jacobian(qs, ee)
    c, s = np.cos(d), np.sin(d)
    rot = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32).cuda()
    assert (pos_s - pos @ rot.T).abs().max() < 5e-6
    assert (lin_s - torch.einsum("ij,bjk->bik", rot, lin)).abs().max() < 5e-6
    # column 0 of the linear Jacobian is d pos / d q1 = z x pos
    assert (lin[:, 0, 0] + pos[:, 1]).abs().max() < 5e-6 and (lin[:, 1, 0] - pos[:, 0]).abs().max() < 5e-6
    # RNEA is affine in qdd: tau(a1 + a2) - tau(0) = (tau(a1) - tau(0)) + (tau(a2) - tau(0))
    t0 = m.compute_non_linear_effects(q, qd)
    t1 = m.compute_inverse_dynamics(q, qd, a1); t2 = m.compute_inverse_dynamics(q, qd, a2)
    t12 = m.compute_inverse_dynamics(q, qd, a1 + a2)
    scale = 1 + t12.abs().max().item()
    assert ((t12 - t0) - ((t1 - t0) + (t2 - t0))).abs().max() / scale < 2e-5
    # shard consistency: rows computed in two halves equal the single launch (what multi-GPU sharding relies on)
    half = B // 2
    pa = m.compute_fk_and_jacobian(q[:half].contiguous(), ee); pb = m.compute_fk_and_jacobian(q[half:].contiguous(), ee)
    assert torch.equal(torch.cat([pa[2], pb[2]]), lin) and torch.equal(torch.cat([pa[0], pb[0]]), pos)
    # against the oracle: EVERY row at the metric's batch (65 536), 16 384 random rows of the 2^20 batch; quaternions with
    # the reference's sign wherever the rotation is further than 1e-5 from a case boundary (helpers.quat_close)
    idx = torch.arange(B) if B <= 65536 else torch.randperm(B, generator=torch.Generator().manual_seed(1))[:16384]
    orc = Oracle(m._spec)
    sel = idx.cuda()
    rp, rq, rl, ra = orc.fk_jacobian(host(q[sel]).astype(np.float64), 8, np.float64)
    assert max_err(host(pos[sel]), rp) <= TOL_POS["atol"] and max_err(host(lin[sel]), rl) <= TOL_JAC["atol"]
    assert max_err(host(ang[sel]), ra) <= TOL_JAC["atol"]
    ok, flips = quat_close(host(quat[sel]), rq, TOL_QUAT["atol"])
    assert ok, flips
    rt = orc.rnea(host(q[sel]).astype(np.float64), host(qd[sel]).astype(np.float64),
                  host(a1[sel]).astype(np.float64), True, True, np.float64)
    assert np.allclose(host(t1[sel]), rt, **TOL_TAU)


def test_allegro_fingertips_full_batch():
    """Config 4: Allegro 16-DoF branching tree, batch 65 536, FK to the four fingertips."""
    m = load_model("allegro_left", "cuda")
    tips = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]
    idx = [m._name_to_idx_map[t] for t in tips]
    q, _, _ = sample_states(m, 65536, seed=4)
    pos, quat = m._fk_targets(dev(q), idx)
    assert tuple(pos.shape) == (65536, 4, 3) and tuple(quat.shape) == (65536, 4, 4)
    # EVERY one of the 65 536 rows against the fp64 oracle (as the metric and config 2 do)
    rp, rq = Oracle(m._spec).fk(q.astype(np.float64), idx, np.float64)
    assert max_err(host(pos), rp) <= TOL_POS["atol"]
    ok, _ = quat_close(host(quat), rq, TOL_QUAT["atol"])
    assert ok
    # a finger's tip does not move when another finger's joints move
    q2 = q.copy(); q2[:, 4:8] += 0.3
    pos2, _ = m._fk_targets(dev(q2), idx)
    moved = (host(pos2) - host(pos)).reshape(65536, 4, 3)
    still = [t for t in range(4) if np.abs(moved[:, t]).max() == 0.0]
    assert len(still) == 3


@pytest.mark.gpu
@pytest.mark.parametrize("robot,tips", [("allegro_left", ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]),
                                        ("trifinger_edu", ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"])])
def test_fanout_fk_two_samples_per_lane(robot, tips):
    """From 2^18 samples on the fan-out FK walks pairs of tiles with two samples per lane (fk_fan_chain2_kernel), an odd tile
    with one, a ragged tail through the loop kernel: rows from every part against the fp64 oracle, and against the same rows
    launched as a small batch (the one-sample kernels; same poses to rounding)."""
    m = load_model(robot, "cuda")
    idx = [m._name_to_idx_map[t] for t in tips]
    B = (1 << 18) + 64 * 3 + 5
    q, _, _ = sample_states(m, B, seed=44)
    pos, quat = m._fk_targets(dev(q), idx)
    rows = np.r_[0:130, 70000:70130, (1 << 18) - 70:B]
    rp, rq = Oracle(m._spec).fk(q[rows].astype(np.float64), idx, np.float64)
    assert max_err(host(pos)[rows], rp) <= TOL_POS["atol"]
    ok, _ = quat_close(host(quat)[rows], rq, TOL_QUAT["atol"])
    assert ok
    ps, qs = m._fk_targets(dev(np.ascontiguousarray(q[rows])), idx)
    assert max_err(host(pos)[rows], host(ps)) <= 1e-6 and max_err(host(quat)[rows], host(qs)) <= 1e-6


def test_update_kinematic_state_exposes_body_pose_and_velocity():
    """robot_model.py:139-195: after update_kinematic_state every body carries its world pose and its body-frame
    spatial velocity; here they are evaluated lazily from the recorded (q, qd)."""
    m = load_model("panda_no_gripper", "cuda")
    fresh = m._bodies[3].pose
    assert torch.equal(fresh.rotation()[0], torch.eye(3).cuda()) and not fresh.translation().any()
    q, qd, _ = sample_states(m, 33, seed=12)
    m.update_kinematic_state(dev(q), dev(qd))
    orc = Oracle(m._spec)
    R, p = orc.fk_all_poses(q.astype(np.float64), np.float64)
    for i in (1, 4, 8):
        pose = m._bodies[i].pose
        assert max_err(host(pose.translation()), p[:, i]) <= TOL_POS["atol"]
        assert max_err(host(pose.rotation()), R[:, i]) <= 2e-6
        # body-frame velocity: v = R^T J_lin qd, w = R^T J_ang qd (oracle Jacobians)
        _, _, lin, ang = orc.fk_jacobian(q.astype(np.float64), i, np.float64)
        v_ref = np.einsum("bji,bj->bi", R[:, i], np.einsum("bij,bj->bi", lin, qd.astype(np.float64)))
        w_ref = np.einsum("bji,bj->bi", R[:, i], np.einsum("bij,bj->bi", ang, qd.astype(np.float64)))
        vel = m._bodies[i].vel
        assert max_err(host(vel.lin), v_ref) <= 5e-6 and max_err(host(vel.ang), w_ref) <= 5e-6


@pytest.mark.parametrize("robot,tips", [
    ("allegro_left", ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]),
    ("allegro_left", ["link_15.0_tip", "link_3.0_tip"]),
    ("trifinger_edu", ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"]),
])
@pytest.mark.parametrize("B", [1, 63, 64, 200])
def test_fanout_fk_equals_merged_walk_and_oracle(robot, tips, B):
    """Hands: the fingertip chains are disjoint, so `_fk_targets` plans one wavefront per chain (drm_fk_fanout); the
    merged multi-target walk (drm_fk) and the oracle must agree with it."""
    from differentiable_robot_model_amd import backend
    m = load_model(robot, "cuda")
    idx = [m._name_to_idx_map[t] for t in tips]
    q, _, _ = sample_states(m, B, seed=300 + B)
    pos, quat = m._fk_targets(dev(q), idx)
    dw = m._get_walk(("fk", tuple(idx)), targets=idx)
    assert m._fanout_chains(idx, dw) is not None, "these targets are expected to take the fan-out plan"
    pm, qm = backend.fk(dw.program, m._ops_f(dw), dw.ops_i, dev(q), len(idx), m._n_dofs)   # merged walk
    assert max_err(host(pos), host(pm)) <= 1e-6 and max_err(host(quat), host(qm)) <= 1e-6
    rp, rq = Oracle(m._spec).fk(q.astype(np.float64), idx, np.float64)
    assert max_err(host(pos), rp) <= TOL_POS["atol"]
    ok, _ = quat_close(host(quat), rq, TOL_QUAT["atol"])
    assert ok
    # a parameter becomes learnable afterwards: the cached fan-out plan (folded chain walks) is dropped, the same poses come
    # back through the walks that keep every link an op, and a loss on them reaches the parameter
    if robot == "trifinger_edu" and B == 64:
        from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
        body = m._bodies[m._name_to_idx_map["finger_middle_link_0"]]
        m.make_link_param_learnable(body.name, "trans", UnconstrainedTensor(1, 3, init_tensor=body.trans().detach().reshape(1, 3).clone()))
        pos_l, quat_l = m._fk_targets(dev(q), idx)
        assert max_err(host(pos_l), rp) <= TOL_POS["atol"]
        pos_l.sum().backward()
        (p,) = list(m.parameters())
        assert p.grad is not None and float(p.grad.abs().max()) > 0
    # arms keep the merged / chain kernels
    arm = load_model("panda_no_gripper", "cuda")
    two = [arm._name_to_idx_map[n] for n in ("panda_link4", "panda_virtual_ee_link")]
    assert arm._fanout_chains(two, arm._get_walk(("fk", tuple(two)), targets=two)) is None


# ------------------------------------------------------------------ random robots x batch sizes
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DRM_FUZZ_SEEDS", "16"))))
def test_random_robot_and_batch_size_vs_oracle(seed):
    """Every forward call of a random robot at a random (mostly ragged) batch size against the fp64 oracle — partial
    tiles, unaligned tails of the arm kernels and multi-wave blocks in one sweep."""
    rng = np.random.default_rng(9000 + seed)
    robot = ALL_ROBOTS[int(rng.integers(len(ALL_ROBOTS)))]
    B = int(rng.choice([rng.integers(1, 200), rng.integers(200, 3000), 64 * rng.integers(1, 40) + rng.integers(0, 2)]))
    m = load_model(robot, "cuda")
    orc = Oracle(m._spec)
    q, qd, qdd = sample_states(m, B, seed=seed)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    link = int(rng.integers(1, len(m._bodies)))
    pos, quat, lin, ang = m.compute_fk_and_jacobian(dev(q), m._bodies[link].name)
    rp, rq, rl, ra = orc.fk_jacobian(q64, link, np.float64)
    assert max_err(host(pos), rp) <= TOL_POS["atol"] and quat_close(host(quat), rq, TOL_QUAT["atol"])[0], (robot, B, link)
    assert max_err(host(lin), rl) <= TOL_JAC["atol"] and max_err(host(ang), ra) <= TOL_JAC["atol"], (robot, B, link)
    p2, r2 = m.compute_forward_kinematics(dev(q), m._bodies[link].name)
    assert max_err(host(p2), rp) <= TOL_POS["atol"] and quat_close(host(r2), rq, TOL_QUAT["atol"])[0]
    tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd))
    assert np.allclose(host(tau), orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU), (robot, B)
    H = m.compute_lagrangian_inertia_matrix(dev(q))
    assert np.allclose(host(H), orc.mass_matrix(q64, False, False, np.float64), **TOL_TAU), (robot, B)


# ------------------------------------------------------------------ fused FK + inverse dynamics (drm_fk_rnea)
@pytest.mark.parametrize("robot,link", [("panda_no_gripper", "panda_virtual_ee_link"), ("iiwa7", "iiwa_link_ee"),
                                        ("panda_no_gripper", "panda_link4"), ("allegro_left", "link_15.0_tip"),
                                        ("fetch_arm_no_gripper", "virtual_ee_link")])
@pytest.mark.parametrize("B", [1, 64, 257, 4096])
def test_fk_and_inverse_dynamics_one_call(robot, link, B):
    """tau, pos, quat of drm_fk_rnea (one fused launch for a serial 7-DoF arm whose last link is the target, the two
    walks back to back otherwise) are BIT-identical to the separate calls and agree with the fp64 oracle."""
    m = load_model(robot, "cuda")
    q, qd, qdd = sample_states(m, B, seed=700 + B)
    for grav, damp in ((True, True), (False, False)):
        tau, pos, quat = m.compute_fk_and_inverse_dynamics(dev(q), dev(qd), dev(qdd), link, grav, damp)
        t2 = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=grav, use_damping=damp)
        p2, r2 = m.compute_forward_kinematics(dev(q), link)
        assert torch.equal(tau, t2) and torch.equal(pos, p2) and torch.equal(quat, r2), (robot, link, B)
    orc = Oracle(m._spec)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    tau, pos, quat = m.compute_fk_and_inverse_dynamics(dev(q), dev(qd), dev(qdd), link)
    rp, rq = orc.fk(q64, [m._name_to_idx_map[link]], np.float64)
    assert max_err(host(pos), rp[:, 0]) <= TOL_POS["atol"] and quat_close(host(quat), rq[:, 0], TOL_QUAT["atol"])[0]
    assert np.allclose(host(tau), orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU)


def test_fk_and_inverse_dynamics_plan_under_hipgraph_config3_shard():
    """BASELINE configuration 3, one GPU's shard of 2^20 (131 072 rows) and the full 2^20: the fused plan replayed from a
    hipGraph against the separate calls (bit-exact) and EVERY row against the oracle."""
    ee = "panda_virtual_ee_link"
    orc = Oracle(load_model("panda_no_gripper")._spec)
    for B, mode in ((131072, "off"), (131072, None), (1 << 20, "off"), (1 << 20, None)):
        # "off": the library's kernels — the fused launch against the separate calls bit for bit.  None: the model as it comes (round
        # 6: the arm's own constant-folded kernels, shipped with the library) — a few ulp between the fused and the separate kernels
        m = load_model("panda_no_gripper", "cuda")
        m.own_kernels = mode
        q, qd, qdd = (dev(a) for a in sample_states(m, B, seed=B, vel=0.4, acc=0.8))
        plan = m.plan_fk_and_inverse_dynamics(q, qd, qdd, ee)
        graph = torch.cuda.CUDAGraph()
        plan.launch()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph):
            plan.launch()
        for t in plan.outputs():
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        tau2 = m.compute_inverse_dynamics(q, qd, qdd)
        p2, r2 = m.compute_forward_kinematics(q, ee)
        if mode == "off":
            assert torch.equal(plan.tau, tau2) and torch.equal(plan.pos, p2) and torch.equal(plan.quat, r2)
        else:
            assert float(((plan.tau - tau2).abs() / tau2.abs().clamp_min(1.0)).max()) <= 2e-5
            assert float((plan.pos - p2).abs().max()) <= 1e-6 and float((plan.quat - r2).abs().max()) <= 1e-6
        for lo in range(0, B, 1 << 17):      # EVERY row against the fp64 oracle, 131 072 at a time
            sl = slice(lo, lo + (1 << 17))
            q64, qd64, qdd64 = (host(a[sl]).astype(np.float64) for a in (q, qd, qdd))
            rp, rq = orc.fk(q64, [8], np.float64)
            assert max_err(host(plan.pos[sl]), rp[:, 0]) <= TOL_POS["atol"]
            assert quat_close(host(plan.quat[sl]), rq[:, 0], TOL_QUAT["atol"])[0]
            assert np.allclose(host(plan.tau[sl]), orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU)


def test_eight_shards_against_one_launch():
    """What `bench.py --verify-gather` meets on an 8-GPU node, rehearsed on one: the rows of a global batch computed as 8
    launches of the shard size against ONE launch over all of them.  The C ABI picks a kernel form by launch size — 65 536-row
    shards of the metric take the register-resident-table form and 524 288 rows the plain one (csrc/drm_arm_kernels.hip
    launch_fk_jacobian_arm); 131 072-row shards of configuration 3 take the latency form of fk_rnea_arm2_kernel and 2^20 rows
    the streaming form (csrc/drm_arm_dynamics.hip launch_fk_rnea_arm) — so this is the test that says whether the two agree bit
    for bit.  Asserted: a shard launched twice is bit-identical (what the gather check in bench.py relies on), and shards and
    whole agree within a few ulp (what `gather_vs_whole_launch` reports); bit-equality of the forms is printed, not required."""
    m = load_model("panda_no_gripper", "cuda")
    ee = "panda_virtual_ee_link"
    # metric: 8 x 65 536 against 524 288
    q = dev(sample_states(m, 8 * 65536, seed=11)[0])
    whole = m.plan_fk_and_jacobian(q, ee)
    whole.launch()
    torch.cuda.synchronize()
    names = ("pos", "quat", "lin_jac", "ang_jac")
    report = {}
    for r in range(8):
        sl = slice(r * 65536, (r + 1) * 65536)
        shard = m.plan_fk_and_jacobian(q[sl].contiguous(), ee)
        shard.launch()
        torch.cuda.synchronize()
        first = [t.clone() for t in shard.outputs()]
        shard.launch()
        torch.cuda.synchronize()
        for name, a, b, w in zip(names, first, shard.outputs(), whole.outputs()):
            assert torch.equal(a, b), name
            d = float((a - w[sl]).abs().max())
            report["metric " + name] = max(report.get("metric " + name, 0.0), d)
            if name == "quat":                      # (a last-bit difference of R on a case boundary may pick the other sign)
                assert quat_close(host(a), host(w[sl]), 2e-6)[0], (name, r, d)
            else:
                assert d <= 2e-6, (name, r, d)      # (entries are O(1): a few ulp)
    # configuration 3: 8 x 131 072 against 2^20
    q, qd, qdd = (dev(a) for a in sample_states(m, 1 << 20, seed=12, vel=0.4, acc=0.8))
    whole = m.plan_fk_and_inverse_dynamics(q, qd, qdd, ee)
    whole.launch()
    torch.cuda.synchronize()
    for r in range(8):
        sl = slice(r << 17, (r + 1) << 17)
        shard = m.plan_fk_and_inverse_dynamics(q[sl].contiguous(), qd[sl].contiguous(), qdd[sl].contiguous(), ee)
        shard.launch()
        torch.cuda.synchronize()
        first = [t.clone() for t in shard.outputs()]
        shard.launch()
        torch.cuda.synchronize()
        for name, a, b, w in zip(("tau", "pos", "quat"), first, shard.outputs(), whole.outputs()):
            assert torch.equal(a, b), name
            w = w[sl].reshape(a.shape)
            d = float(((a - w).abs() / w.abs().clamp_min(1.0)).max())
            report["config3 " + name] = max(report.get("config3 " + name, 0.0), d)
            if name == "quat":
                assert quat_close(host(a), host(w), 2e-6)[0], (name, r, d)
            else:
                assert d <= (2e-5 if name == "tau" else 2e-6), (name, r, d)
    print("shards vs one launch, max deviation:", {k: ("bit-equal" if v == 0.0 else "%.2e" % v) for k, v in report.items()})


@pytest.mark.parametrize("robot,link", [("panda_no_gripper", "panda_virtual_ee_link"), ("iiwa7", "iiwa_link_ee")])
def test_two_samples_per_lane_kernels_every_row_vs_oracle(robot, link):
    """Launches of more than 1 024 tiles take the two-samples-per-lane kernels (rnea_arm2_kernel / fk_rnea_arm2_kernel: a
    wavefront owns 128 rows).  B = 1 027 tiles + 17 rows: 513 pairs through them, the odd tile through the one-sample
    kernel, the ragged tail through the loop kernel — EVERY row against the fp64 oracle, on the folded 7-link table (the API)
    and on the full 8-link table (drm_rnea on the unfolded walk), fused and separate."""
    import ctypes
    from differentiable_robot_model_amd import backend
    m = load_model(robot, "cuda")
    B, n = 1027 * 64 + 17, 7
    q, qd, qdd = sample_states(m, B, seed=4242, vel=0.6, acc=1.2)
    q[40000, 3] = 1.5e5          # one sample of a pair takes the fp64 argument reduction: wave-uniform fallback
    orc = Oracle(m._spec)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    ee = m._name_to_idx_map[link]
    rp, rq = orc.fk(q64, [ee], np.float64)
    for grav, damp in ((True, True), (False, False)):
        ref = orc.rnea(q64, qd64, qdd64, grav, damp, np.float64)
        tau, pos, quat = m.compute_fk_and_inverse_dynamics(dev(q), dev(qd), dev(qdd), link, grav, damp)
        t2 = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=grav, use_damping=damp)
        assert torch.equal(tau, t2)
        assert np.allclose(host(tau), ref, **TOL_TAU)
        assert max_err(host(pos), rp[:, 0]) <= TOL_POS["atol"] and quat_close(host(quat), rq[:, 0], TOL_QUAT["atol"])[0]
        p2, r2 = m.compute_forward_kinematics(dev(q), link)
        # bit for bit the separate call — except in the 128-row tile of the sample with the huge angle: the large-argument
        # sincos fallback is wave-uniform, and a wave is 128 rows in the fused launch but 64 in compute_forward_kinematics
        same = torch.ones(B, dtype=torch.bool, device="cuda")
        same[40000 // 128 * 128:(40000 // 128 + 1) * 128] = False
        assert torch.equal(pos[same], p2[same]) and torch.equal(quat[same], r2[same])
        assert (pos - p2).abs().max().item() <= 1e-6 and (quat - r2).abs().max().item() <= 1e-6
        # the unfolded walk: all 8 links are ops (LINKS = 8 instantiation)
        dt = m._get_walk(("tree",), whole_tree=True)
        assert dt.program.n_ops == 8
        walk = backend._walk_struct(dt.program, m._ops_f(dt), dt.ops_i, n)
        out = torch.empty(B, n, device="cuda")
        dq, dqd, dqdd = dev(q), dev(qd), dev(qdd)
        lib = backend.load_library()
        scratch = torch.empty(max(1, lib.drm_rnea_scratch_floats(ctypes.byref(walk), B)), device="cuda")   # the ragged tail's records
        backend._check(lib.drm_rnea(ctypes.byref(walk), dq.data_ptr(), dqd.data_ptr(), dqdd.data_ptr(), B,
                                    (1 if grav else 0) | (2 if damp else 0), out.data_ptr(), scratch.data_ptr(),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        assert np.allclose(host(out), ref, **TOL_TAU)
    # non-linear effects (qdd = NULL)
    nle = m.compute_non_linear_effects(dev(q), dev(qd))
    assert np.allclose(host(nle), orc.rnea(q64, qd64, np.zeros_like(q64), True, True, np.float64), **TOL_TAU)


@pytest.mark.parametrize("robot,compat", [("panda", True), ("panda", False), ("jaco", True), ("iiwa7_allegro", True)])
def test_inverse_dynamics_of_an_arm_that_carries_a_hand(robot, compat):
    """drm_arm_hand.hip: the straight-line kernel for P prefix ops + K sub-chains of L ops (Panda with gripper — its fingers as
    the reference models them and as the prismatic joints they are —, Jaco, iiwa7 + Allegro): full tiles + a ragged tail,
    every row against the fp64 oracle, all four flag combinations; the non-linear effects (qdd = NULL) too."""
    from differentiable_robot_model_amd.flatten import SHAPE_ARM_HAND
    m = load_model(robot, "cuda", reference_compat=compat)
    assert m._dynamics_walk().program.shape & SHAPE_ARM_HAND
    B = 64 * 37 + 11
    q, qd, qdd = sample_states(m, B, seed=91)
    orc = Oracle(m._spec)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    for grav in (True, False):
        for damp in (True, False):
            tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=grav, use_damping=damp)
            assert np.allclose(host(tau), orc.rnea(q64, qd64, qdd64, grav, damp, np.float64), **TOL_TAU), (robot, grav, damp)
    nle = m.compute_non_linear_effects(dev(q), dev(qd))
    assert np.allclose(host(nle), orc.rnea(q64, qd64, np.zeros_like(q64), True, True, np.float64), **TOL_TAU)
    # rows of full tiles are the same whether or not a tail follows them
    t_full = m.compute_inverse_dynamics(dev(q[:64 * 37]), dev(qd[:64 * 37]), dev(qdd[:64 * 37]))
    assert torch.equal(t_full, m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd))[:64 * 37])
    # forward dynamics of the same shapes (forward_dynamics_arm_hand_kernel: the articulated-body recursion, sub-chains visited
    # twice instead of stored), on torques that produce accelerations of order one (tau = ID(q, qd, qdd)): every row against the
    # fp64 oracle and against the loop-structured kernel (the walk without its shape bit), both flag settings
    import ctypes
    from differentiable_robot_model_amd import backend
    lib = backend.load_library()
    dw = m._dynamics_walk()
    generic = backend._walk_struct_build(dw.program, m._ops_f(dw), dw.ops_i, m._n_dofs)   # (a private copy, not the cached struct)
    generic.shape &= ~SHAPE_ARM_HAND
    dq, dqd = dev(q), dev(qd)
    worst = 0.0
    for grav, damp in ((True, True), (False, False)):
        tau = m.compute_inverse_dynamics(dq, dqd, dev(qdd), include_gravity=grav, use_damping=damp)
        acc = m.compute_forward_dynamics(dq, dqd, tau, include_gravity=grav, use_damping=damp)
        ref = orc.forward_dynamics(q64, qd64, host(tau).astype(np.float64), grav, damp, np.float64)
        err = float((np.abs(host(acc) - ref) / (1.0 + np.abs(ref))).max())
        loop = torch.empty_like(acc)
        scratch = torch.empty(max(1, lib.drm_forward_dynamics_scratch_floats(ctypes.byref(generic), B)), device="cuda")
        backend._check(lib.drm_forward_dynamics(ctypes.byref(generic), dq.data_ptr(), dqd.data_ptr(), tau.data_ptr(), B,
                                                (1 if grav else 0) | (2 if damp else 0), loop.data_ptr(), scratch.data_ptr(),
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        err_loop = float((np.abs(host(loop) - ref) / (1.0 + np.abs(ref))).max())
        worst = max(worst, err)
        print("forward dynamics %s grav=%d damp=%d: straight-line %.2e, loop kernel %.2e (rel. to 1 + |qdd|, fp64 oracle)" % (robot, grav, damp, err, err_loop))
        # cond(H) of the arm with a 16-DoF hand is ~1e8: both fp32 recursions sit at 1e-3 there (tests/test_forward_dynamics.py)
        assert err <= max(1e-3, 2.0 * err_loop), (robot, grav, damp, err, err_loop)
    # the joint-space inertia matrix of the same shapes (crba_arm_hand_kernel: column forces of a sub-chain carried together up
    # the prefix, nothing parked): every row against the fp64 oracle, symmetric bit for bit, the full tiles identical whether
    # or not a tail follows, and against the loop-structured kernel
    H = m.compute_lagrangian_inertia_matrix(dq)
    ref = orc.mass_matrix(q64, dtype=np.float64)
    TOL_H = dict(atol=5e-5, rtol=2e-5)        # tests/test_mass_matrix.py
    assert np.allclose(host(H), ref, **TOL_H), (robot, float(np.abs(host(H) - ref).max()))
    assert torch.equal(H, H.transpose(1, 2))
    assert torch.equal(H[:64 * 37], m.compute_lagrangian_inertia_matrix(dq[:64 * 37]))
    n = m._n_dofs
    loop = torch.empty_like(H)
    scratch = torch.empty(max(1, lib.drm_crba_scratch_floats(ctypes.byref(generic), B)), device="cuda")
    backend._check(lib.drm_crba(ctypes.byref(generic), dq.data_ptr(), B, loop.data_ptr(), scratch.data_ptr(),
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert np.allclose(host(loop), ref, **TOL_H)
    print("mass matrix %s: straight-line %.2e, loop kernel %.2e (max abs, fp64 oracle)" % (robot, float(np.abs(host(H) - ref).max()), float(np.abs(host(loop) - ref).max())))


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["allegro_left", "trifinger_edu"])
def test_inverse_dynamics_of_a_hand_two_samples_per_lane(robot):
    """rnea_fingers2_kernel (DRM_WALK_FINGERS: a finger per wavefront, two samples per lane, full 128-row tiles; the rest of a
    launch through the loop kernel): every row against the fp64 oracle, all flag combinations, the non-linear effects, misaligned
    views, and the rows of full tiles identical whether or not a tail follows."""
    from differentiable_robot_model_amd.flatten import SHAPE_FINGERS
    m = load_model(robot, "cuda")
    assert m._dynamics_walk().program.shape & SHAPE_FINGERS
    B = 128 * 7 + 77
    q, qd, qdd = sample_states(m, B, seed=83)
    orc = Oracle(m._spec)
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    for grav in (True, False):
        for damp in (True, False):
            tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=grav, use_damping=damp)
            assert np.allclose(host(tau), orc.rnea(q64, qd64, qdd64, grav, damp, np.float64), **TOL_TAU), (robot, grav, damp)
    nle = m.compute_non_linear_effects(dev(q), dev(qd))
    assert np.allclose(host(nle), orc.rnea(q64, qd64, np.zeros_like(q64), True, True, np.float64), **TOL_TAU)
    full = m.compute_inverse_dynamics(dev(q[:128 * 7]), dev(qd[:128 * 7]), dev(qdd[:128 * 7]))
    assert torch.equal(full, m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd))[:128 * 7])
    view = m.compute_inverse_dynamics(dev(q)[1:], dev(qd)[1:], dev(qdd)[1:])      # rows start 4 n bytes into the buffers
    assert np.allclose(host(view), orc.rnea(q64[1:], qd64[1:], qdd64[1:], True, True, np.float64), **TOL_TAU)
    # forward dynamics of the same shape (forward_dynamics_fingers_kernel: bias torques, the finger's block of H and its L^T D L
    # solve per wavefront) on torques that give accelerations of order one: every row against the fp64 oracle, both flag
    # settings, aligned and not, full tiles identical with and without a tail
    # (the Allegro's fingertip inertias make this system badly conditioned for physically consistent torques: the reference's own
    # recursion in fp32 is 8e-2 off the fp64 result here — the kernels are held to twice that, or 1e-3)
    for grav, damp in ((True, True), (False, False)):
        tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd), include_gravity=grav, use_damping=damp)
        ref = orc.forward_dynamics(q64, qd64, host(tau).astype(np.float64), grav, damp, np.float64)
        ref32 = orc.forward_dynamics(q, qd, host(tau), grav, damp, np.float32)
        floor = float((np.abs(ref32 - ref) / (1.0 + np.abs(ref))).max())
        for off in (0, 1):
            acc = m.compute_forward_dynamics(dev(q)[off:], dev(qd)[off:], tau[off:], include_gravity=grav, use_damping=damp)
            err = float((np.abs(host(acc) - ref[off:]) / (1.0 + np.abs(ref[off:]))).max())
            assert err <= max(1e-3, 2.0 * floor), (robot, grav, off, err, floor)
    tau = m.compute_inverse_dynamics(dev(q), dev(qd), dev(qdd))
    whole = m.compute_forward_dynamics(dev(q), dev(qd), tau)
    assert torch.equal(whole[:64 * 14], m.compute_forward_dynamics(dev(q[:64 * 14]), dev(qd[:64 * 14]), tau[:64 * 14]))


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["allegro_left", "trifinger_edu"])
def test_inverse_dynamics_backward_of_a_hand(robot):
    """rnea_backward_fingers_kernel (a finger per wavefront on the arm form of the adjoint walk) against the loop-structured
    backward kernel on the same walk without its shape bit — the kernel the reference-autograd goldens hold (their batches are
    below one tile): input gradients and the constant gradients of three ops, full tiles + ragged tail, with and without qdd,
    aligned and misaligned; and through autograd against central differences of the fp64 oracle."""
    import copy
    from differentiable_robot_model_amd import backend
    from differentiable_robot_model_amd.flatten import SHAPE_FINGERS
    m = load_model(robot, "cuda")
    dw = m._dynamics_walk()
    assert dw.program.shape & SHAPE_FINGERS
    B, n = 64 * 11 + 9, m._n_dofs
    q, qd, qdd = (dev(a) for a in sample_states(m, B, seed=29))
    gtau = torch.randn(B, n, device="cuda", generator=torch.Generator("cuda").manual_seed(8))
    of = m._ops_f(dw)
    mask = (1 << 0) | (1 << (n // 2)) | (1 << (n - 1))
    generic = copy.copy(dw.program)
    generic.shape = dw.program.shape & ~SHAPE_FINGERS & 0xffffff
    for use_qdd in (True, False):
        for grav, damp in ((True, True), (False, False)):
            for off in (0, 1):      # off = 1: rows start 4 n bytes into the buffers (no 16-byte accesses)
                a = [t[off:] for t in (q, qd, qdd, gtau)]
                got = backend.rnea_backward(dw.program, of, dw.ops_i, a[0], a[1], a[2] if use_qdd else None, a[3], grav, damp, n, mask, True)
                ref = backend.rnea_backward(generic, of, dw.ops_i, a[0], a[1], a[2] if use_qdd else None, a[3], grav, damp, n, mask, True)
                for x, y, name in zip(got[0], ref[0], ("grad_q", "grad_qd", "grad_qdd")):
                    scale = max(1.0, float(y.abs().max()))
                    assert float((x - y).abs().max()) <= 5e-5 * scale, (robot, name, use_qdd, grav, off, float((x - y).abs().max()))
                scale = max(1.0, float(ref[1].abs().max()))
                assert float((got[1] - ref[1]).abs().max()) <= 2e-4 * scale, (robot, "grad_ops_f", float((got[1] - ref[1]).abs().max()))
                assert float(got[1].abs().max()) > 0
    orc = Oracle(m._spec)
    tq = q.clone().requires_grad_(True)
    (m.compute_inverse_dynamics(tq, qd, qdd, include_gravity=True, use_damping=True) * gtau).sum().backward()
    v = np.random.default_rng(0).standard_normal((B, n))
    q64, qd64, qdd64 = (host(t).astype(np.float64) for t in (q, qd, qdd))
    e = 1e-6
    fd = ((orc.rnea(q64 + e * v, qd64, qdd64, True, True, np.float64) - orc.rnea(q64 - e * v, qd64, qdd64, True, True, np.float64))
          / (2 * e) * host(gtau)).sum(axis=1)
    mine = (host(tq.grad).astype(np.float64) * v).sum(axis=1)
    assert np.abs(mine - fd).max() <= 2e-3 * max(1.0, float(np.abs(fd).max())), float(np.abs(mine - fd).max())


@pytest.mark.gpu
@pytest.mark.parametrize("robot,compat", [("panda", True), ("panda", False), ("jaco", True), ("iiwa7_allegro", True)])
def test_inverse_dynamics_backward_of_an_arm_that_carries_a_hand(robot, compat):
    """rnea_backward_arm_hand_kernel (full tiles of an arm + hand walk; nothing stored per link) against the loop-structured
    backward kernel on the same walk without its shape bit — the kernel the reference-autograd goldens hold (their batches are
    below one tile): input gradients and the constant gradients of three ops (a prefix link, the palm's parent op, a fingertip),
    full tiles + ragged tail, with and without qdd; and through the API against a float64 finite difference of the oracle."""
    import ctypes
    from differentiable_robot_model_amd import backend
    from differentiable_robot_model_amd.flatten import SHAPE_ARM_HAND
    m = load_model(robot, "cuda", reference_compat=compat)
    dw = m._dynamics_walk()
    assert dw.program.shape & SHAPE_ARM_HAND
    B, n = 64 * 9 + 5, m._n_dofs
    q, qd, qdd = (dev(a) for a in sample_states(m, B, seed=97))
    gtau = torch.randn(B, n, device="cuda", generator=torch.Generator("cuda").manual_seed(4))
    of = m._ops_f(dw)
    P = (dw.program.shape >> 24) & 0xf
    mask = (1 << 1) | (1 << (P - 1)) | (1 << (dw.program.n_ops - 1))

    import copy
    generic = copy.copy(dw.program)     # the same walk, shape bit cleared: the loop kernel takes every row
    generic.shape = dw.program.shape & ~SHAPE_ARM_HAND & 0xffffff
    for use_qdd in (True, False):
        for grav, damp in ((True, True), (False, False)):
            got = backend.rnea_backward(dw.program, of, dw.ops_i, q, qd, qdd if use_qdd else None, gtau, grav, damp, n, mask, True)
            ref = backend.rnea_backward(generic, of, dw.ops_i, q, qd, qdd if use_qdd else None, gtau, grav, damp, n, mask, True)
            for a, b, name in zip(got[0], ref[0], ("grad_q", "grad_qd", "grad_qdd")):
                scale = max(1.0, float(b.abs().max()))
                assert float((a - b).abs().max()) <= 5e-5 * scale, (robot, name, use_qdd, grav, float((a - b).abs().max()), scale)
            scale = max(1.0, float(ref[1].abs().max()))
            assert float((got[1] - ref[1]).abs().max()) <= 2e-4 * scale, (robot, "grad_ops_f", float((got[1] - ref[1]).abs().max()), scale)
            assert float(got[1].abs().max()) > 0
    # through autograd, against central differences of the fp64 oracle along a random direction
    orc = Oracle(m._spec)
    tq = q.clone().requires_grad_(True)
    tau = m.compute_inverse_dynamics(tq, qd, qdd, include_gravity=True, use_damping=True)
    (tau * gtau).sum().backward()
    v = np.random.default_rng(0).standard_normal((B, n))
    q64, qd64, qdd64 = (host(t).astype(np.float64) for t in (q, qd, qdd))
    e = 1e-6
    fd = ((orc.rnea(q64 + e * v, qd64, qdd64, True, True, np.float64) - orc.rnea(q64 - e * v, qd64, qdd64, True, True, np.float64))
          / (2 * e) * host(gtau)).sum(axis=1)
    mine = (host(tq.grad).astype(np.float64) * v).sum(axis=1)
    assert np.abs(mine - fd).max() <= 2e-3 * max(1.0, float(np.abs(fd).max())), float(np.abs(mine - fd).max())


def test_all_links_fk_beyond_the_infinity_cache():
    """compute_forward_kinematics_all_links of a 29-link robot at 2^19 samples: 411 MB of poses, so the many-target kernel
    (fk_tree_links_kernel: link-major outputs, a link's 64 poses of a tile one run) takes its `nt` store path; rows from the start,
    the middle and the ragged end against the fp64 oracle, and every row of the launch identical to the same rows launched as a
    small batch."""
    m = load_model("iiwa7_allegro", "cuda")
    B = (1 << 19) + 37
    q, _, _ = sample_states(m, B, seed=61)
    dq = dev(q)
    poses = m.compute_forward_kinematics_all_links(dq)
    rows = np.r_[0:70, 262100:262200, B - 80:B]
    small = m.compute_forward_kinematics_all_links(dev(np.ascontiguousarray(q[rows])))
    names = [b.name for b in m._bodies]
    op, oq = Oracle(m._spec).fk(q[rows].astype(np.float64), list(range(len(names))), np.float64)
    for i, name in enumerate(names):
        p, r = poses[name]
        assert torch.equal(p[rows], small[name][0]) and torch.equal(r[rows], small[name][1]), name
        assert max_err(host(p[rows]), op[:, i]) <= TOL_POS["atol"], name
        ok, _ = quat_close(host(r[rows]), oq[:, i], TOL_QUAT["atol"])
        assert ok, name


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["panda", "allegro_left", "iiwa7_allegro", "fetch", "trifinger_edu"])
@pytest.mark.parametrize("B", [1, 63, 64 * 5, 64 * 3 + 2, 4099])
def test_link_major_fk_is_the_sample_major_fk_transposed(robot, B):
    """drm_fk_links (pos [T, B, 3], quat [T, B, 4]; fanned out over wavefronts where the walk splits behind a hub) against drm_fk
    on the same walk (pos [B, T, 3]: the grouped / whole-tile kernels) — the same poses bit for bit, any batch size (B not a
    multiple of 4 takes the 4-byte position stores), and each link's arrays contiguous."""
    from differentiable_robot_model_amd import backend
    m = load_model(robot, "cuda")
    q, _, _ = sample_states(m, B, seed=B)
    dq = dev(q)
    idx = [i for i in m._spec.preorder() if i != 0]
    dw = m._get_walk(("fk", tuple(idx)), targets=idx)
    p1, r1 = backend.fk(dw.program, m._ops_f(dw), dw.ops_i, dq, len(idx), m._n_dofs)
    p2, r2 = backend.fk_links(dw.program, m._ops_f(dw), dw.ops_i, dq, len(idx), m._n_dofs)
    assert p2.shape == (len(idx), B, 3) and r2.shape == (len(idx), B, 4)
    assert torch.equal(p2.permute(1, 0, 2), p1) and torch.equal(r2.permute(1, 0, 2), r1)
    poses = m.compute_forward_kinematics_all_links(dq)
    for k, i in enumerate(idx):
        p, r = poses[m._bodies[i].name]
        assert p.is_contiguous() and r.is_contiguous() and torch.equal(p, p2[k]) and torch.equal(r, r2[k])


@pytest.mark.gpu
@pytest.mark.parametrize("robot,tips", [("allegro_left", ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]),
                                        ("trifinger_edu", ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"])])
@pytest.mark.parametrize("B", [1, 63, 64 * 4, 64 * 3 + 2, 64 * 2 + 8, 4100])
def test_fingertips_in_one_call_link_major(robot, tips, B):
    """compute_forward_kinematics_links (BASELINE configuration 4 as one API call): a wavefront per fingertip chain writing its own
    link's contiguous arrays (drm_fk_fanout_links; the ragged tail and batches that are not a multiple of 4 through single-target
    launches into the same arrays) — equal to compute_forward_kinematics called per link (bit for bit where the same kernels run) and to the oracle."""
    m = load_model(robot, "cuda")
    q, _, _ = sample_states(m, B, seed=7 * B)
    dq = dev(q)
    idx = [m._name_to_idx_map[t] for t in tips]
    rp, rq = Oracle(m._spec).fk(q.astype(np.float64), idx, np.float64)
    # the library's kernels (own_kernels = "off"): bit for bit; then the model as it comes (round 6: a hand whose fan-out kernel ships
    # with the library — the Allegro's four fingertips — runs its own, constants folded in: same poses to rounding)
    for mode in ("off", None):
        m.own_kernels = mode
        got = m.compute_forward_kinematics_links(dq, tips)
        own = mode is None and m._fan_own(sorted(idx)) is not None
        for k, name in enumerate(tips):
            p, r = got[name]
            assert p.shape == (B, 3) and r.shape == (B, 4) and p.is_contiguous() and r.is_contiguous()
            p1, r1 = m.compute_forward_kinematics(dq, name)
            if B % 4 == 0 and not own:
                assert torch.equal(p, p1) and torch.equal(r, r1), name
            else:   # (a link's array is then not 16-byte aligned: the loop kernel writes it, same poses to rounding)
                assert max_err(host(p), host(p1)) <= 1e-6 and max_err(host(r), host(r1)) <= 1e-6, name
            assert max_err(host(p), rp[:, k]) <= TOL_POS["atol"]
            assert quat_close(host(r), rq[:, k], TOL_QUAT["atol"])[0]
    # under autograd the same call builds the graph (sample-major walk behind it)
    tq = dq.clone().requires_grad_(True)
    out = m.compute_forward_kinematics_links(tq, tips)
    sum(p.sum() + r.sum() for p, r in out.values()).backward()
    assert tq.grad is not None and torch.isfinite(tq.grad).all()
    for name in tips:
        assert max_err(host(out[name][0].detach()), host(got[name][0])) <= 1e-6


@pytest.mark.gpu
def test_config2_iiwa_fk_jacobian_full_batch_vs_oracle():
    """BASELINE configuration 2 at full size: KUKA iiwa 7-DoF, batch 65 536, FK + end-effector Jacobian — EVERY row against
    the fp64 oracle (the oracle does 65 536 rows in well under a second)."""
    m = load_model("iiwa7", "cuda")
    q, _, _ = sample_states(m, 65536, seed=22)
    pos, quat, lin, ang = m.compute_fk_and_jacobian(dev(q), "iiwa_link_ee")
    rp, rq, rl, ra = Oracle(m._spec).fk_jacobian(q.astype(np.float64), m._name_to_idx_map["iiwa_link_ee"], np.float64)
    assert max_err(host(pos), rp) <= TOL_POS["atol"]
    assert max_err(host(lin), rl) <= TOL_JAC["atol"] and max_err(host(ang), ra) <= TOL_JAC["atol"]
    ok, flips = quat_close(host(quat), rq, TOL_QUAT["atol"])
    assert ok and flips <= 2, flips       # the sign may only differ on a branch boundary of sva.py:117-128


def test_quaternion_on_branch_boundaries_gpu():
    """The kernels' quaternion on the case boundaries of the reference's get_quaternion (sva.py:117-128): the reference's
    value and SIGN wherever the rotation is more than 1e-5 from a boundary (tests/golden/golden_quat_branches.npz); the
    batch is padded to full tiles so that the arm kernel takes it, and run once more ragged through the loop-structured one."""
    from test_oracle_golden import check_quaternion_branches

    def fk(robot, q, link):
        m = load_model(robot, "cuda")
        reps = -(-64 // q.shape[0])
        full = np.tile(q, (reps, 1))[:64 * (reps * q.shape[0] // 64) or 64]
        pos, quat = m.compute_forward_kinematics(dev(np.ascontiguousarray(full)), link)
        p_r, q_r = m.compute_forward_kinematics(dev(np.ascontiguousarray(q[:33])), link)     # ragged: tree kernel
        assert max_err(host(p_r), host(pos)[:33]) <= 1e-6
        return host(pos)[:q.shape[0]], host(quat)[:q.shape[0]]
    check_quaternion_branches(fk)


# ------------------------------------------------------------------ folded fixed leaf links (robot_model._dynamics_walk)
@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "allegro_left", "trifinger_edu", "fetch"])
def test_folded_dynamics_walk_matches_the_full_walk(robot):
    """Without learnable parameters the dynamics calls (and their backward kernels) run a walk without the links behind
    fixed leaf joints, whose inertia is folded into the parents' rows once on the host.  Same torques, inertia matrix,
    accelerations and input gradients as the full walk within the parity tolerances — from fewer ops."""
    from differentiable_robot_model_amd import backend
    m = load_model(robot, "cuda")
    B, n = 300, m._n_dofs
    q, qd, qdd = (dev(a) for a in sample_states(m, B, seed=77))
    folded = m._dynamics_walk()
    full = m._get_walk(("tree",), whole_tree=True)
    assert folded.folded and folded.program.n_ops < full.program.n_ops
    f = dev(np.random.default_rng(5).uniform(-1, 1, (B, n)).astype(np.float32))   # (as in test_forward_dynamics.py)
    qg, qdg, qddg = (t.clone().requires_grad_(True) for t in (q, qd, qdd))
    tau = m.compute_inverse_dynamics(qg, qdg, qddg)
    H = m.compute_lagrangian_inertia_matrix(q)
    acc = m.compute_forward_dynamics(q, qd, f, include_gravity=True, use_damping=True)
    gtau = dev(np.random.default_rng(6).standard_normal((B, n)).astype(np.float32))
    tau.backward(gtau)
    of = m._ops_f(full)
    tau_full = backend.rnea(full.program, of, full.ops_i, q, qd, qdd, True, True, n)
    H_full = backend.crba(full.program, of, full.ops_i, q, n)
    acc_full = backend.forward_dynamics(full.program, of, full.ops_i, q, qd, f, True, True, n)
    gin, _ = backend.rnea_backward(full.program, of, full.ops_i, q, qd, qdd, gtau, True, True, n, 0, True)
    assert torch.allclose(tau.detach(), tau_full, **TOL_TAU) and torch.allclose(H, H_full, **TOL_TAU)
    # (two fp32 solves of the same system, each held to 1e-3 of the fp64 oracle in tests/test_forward_dynamics.py)
    assert ((acc - acc_full).abs() / (1 + acc_full.abs())).max().item() < 2e-3
    for got, ref in zip((qg.grad, qdg.grad, qddg.grad), gin):
        assert (got - ref).abs().max().item() <= 1e-3 * max(ref.abs().max().item(), 1e-6)
    # a link with learnable parameters stays an op of its own (its gradients are its own); the other leaves still fold
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    from differentiable_robot_model_amd.flatten import foldable_links
    last = len(m._bodies) - 1
    assert foldable_links(m._spec)[last]
    m.make_link_param_learnable(m._bodies[last].name, "trans",
                                UnconstrainedTensor(1, 3, init_tensor=m._bodies[last].trans().detach().reshape(1, 3).clone()))
    dw = m._dynamics_walk()
    assert last in [int(i) for i in dw.program.links]
    assert dw.program.n_ops == full.program.n_ops - int(foldable_links(m._spec, keep=[last]).sum())
    tq = q.clone().requires_grad_(True)
    tau_l = m.compute_inverse_dynamics(tq, qd, qdd)
    assert torch.allclose(tau_l.detach(), tau_full, **TOL_TAU)
    tau_l.backward(gtau)
    assert (tq.grad - gin[0]).abs().max().item() <= 1e-3 * max(gin[0].abs().max().item(), 1e-6)
    p = m._bodies[last].trans.param
    _, gops = backend.rnea_backward(full.program, of, full.ops_i, q, qd, qdd, gtau, True, True, n, m._learnable_op_mask(full), False)
    assert p.grad is not None and torch.isfinite(p.grad).all()
    # dL/dtrans of the learnable link through the folded walk == through the full walk (the op's trans entries of grad_ops_f)
    k = full.program.op_of_link[last]
    perm_free = [int(full.program.gather[k, c]) % 32 for c in (7, 9, 11)]   # where the op row's trans entries come from
    assert sorted(perm_free) == [9, 10, 11]
    ref_t = torch.zeros(3, device="cuda")
    for c in (7, 9, 11):
        ref_t[int(full.program.gather[k, c]) % 32 - 9] += gops[k, c] * float(full.program.gsign[k, c])
    assert torch.allclose(p.grad.reshape(-1), ref_t, rtol=1e-3, atol=1e-5 * max(1.0, float(ref_t.abs().max())))


def test_fused_plan_writes_into_caller_owned_output_blocks():
    """plan_fk_and_inverse_dynamics(outputs=...): tau | pos | quat as three contiguous blocks of ONE buffer (what bench.py
    --config 3 hands to the all-gather as it stands) — same values as the plan's own buffers."""
    m = load_model("panda_no_gripper", "cuda")
    B, n = 4096, m._n_dofs
    q, qd, qdd = (dev(a) for a in sample_states(m, B, seed=91))
    flat = torch.full((B * (n + 7),), float("nan"), device="cuda")
    outs = (flat[:B * n].view(B, n), flat[B * n:B * (n + 3)].view(B, 3), flat[B * (n + 3):].view(B, 4))
    plan = m.plan_fk_and_inverse_dynamics(q, qd, qdd, "panda_virtual_ee_link", outputs=outs)
    plan.launch()
    ref = m.plan_fk_and_inverse_dynamics(q, qd, qdd, "panda_virtual_ee_link")
    ref.launch()
    torch.cuda.synchronize()
    assert torch.equal(plan.tau, ref.tau) and torch.equal(plan.pos, ref.pos) and torch.equal(plan.quat, ref.quat)
    assert not torch.isnan(flat).any()
    with pytest.raises(ValueError):
        m.plan_fk_and_inverse_dynamics(q, qd, qdd, "panda_virtual_ee_link", outputs=(outs[0], outs[1], outs[2][:, :3]))
    # blocks that do not start on a 16-byte boundary are refused (the plan's scratch is sized for aligned pointers):
    # a ragged batch (B % 64 != 0) whose tau block starts one float into the buffer
    Br = 4099
    qr, qdr, qddr = (dev(a) for a in sample_states(m, Br, seed=92))
    odd = torch.empty(Br * (n + 7) + 1, device="cuda")[1:]
    bad = (odd[:Br * n].view(Br, n), odd[Br * n:Br * (n + 3)].view(Br, 3), odd[Br * (n + 3):].view(Br, 4))
    with pytest.raises(ValueError, match="16-byte"):
        m.plan_fk_and_inverse_dynamics(qr, qdr, qddr, "panda_virtual_ee_link", outputs=bad)


# ------------------------------------------------------------------ bench.py: the JSON line the round driver reads
@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields(tmp_path):
    """`python bench.py --gpus 1 --steps K --warmup W` prints ONE compact JSON line (< 4 KB, strict JSON: round 5's 21 KB line was
    not readable by the round driver) with the driver's fields, the roofline object of the dominant kernel, the CPU baseline and one
    flat number per BASELINE configuration; the full record of the run goes to --detail."""
    import json
    import os
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    detail = str(tmp_path / "bench_detail.json")
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-large", "--cpu-seconds", "1",
                        "--detail", detail], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    assert len(lines[0]) < 4096, len(lines[0])

    def strict(name):
        raise ValueError("non-finite constant %s in the bench line" % name)
    d = json.loads(lines[0], parse_constant=strict)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "ranks_seen", "configs_frac"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["unit"] == "evals/s" and d["dtype"] == "f32"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"] and d["config"]["batch_per_gpu"] == 65536
    assert abs(d["value"] - 65536 * 20 / (d["ms_per_step"] * 1e-3 * 20)) <= 1e-5 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-5
    assert abs(roof["achieved"] - 224 * 65536 / (roof["launch_us"] * 1e-6) / 1e9) <= 1e-4 * roof["achieved"]
    assert roof["launch_us"] * 1e-3 <= d["ms_per_step"] * 1.001          # a launch is not longer than a step
    assert 0.2 < roof["frac"] < 1.0 and 0.2 < roof["steady_state_frac"] < 1.0
    # HBM traffic of the kernel, measured by the run itself (two rocprofv3 --pmc child passes) when rocprofv3 is there: within a
    # few percent of the algorithmic bytes (nothing is read twice); otherwise the recorded figure, and the line says which
    assert roof["traffic"] is not None and roof["algorithmic_bytes_per_launch"] == 224 * 65536
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "reference" and cpu["unit"] == "evals/s" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["sample"]
    assert cpu["port_value"] > 0 and cpu["port_cores"] >= 1
    dev = d["gpu_vs_reference_max_abs"]
    assert dev["pos"] <= 2e-6 and dev["lin_jac"] <= 2e-6 and dev["ang_jac"] <= 2e-6 and dev["quat_sign_flips"] == 0
    assert set(d["configs_frac"]) == {"c2", "c3_shard", "c3_whole", "c4", "c5", "c5_fk_mse"}
    assert all(0 < v < 1.0 for v in d["configs_frac"].values())
    # round 6: the drop-in path IS the benchmarked path — a constant model picks its own (shipped) kernels up by itself
    own_off = os.environ.get("DRM_SPECIALIZE") == "0"      # (the suite run on the library's kernels: nothing is attached)
    assert own_off or d["configs_own_kernel"] == {"c3_shard": "default", "c3_whole": "default", "c4": "default"}, d["configs_own_kernel"]
    assert set(d["api_eager_us_per_call"]) == {"forward_kinematics", "endeffector_jacobian", "inverse_dynamics", "learned_model_inverse_dynamics"}

    # ---- the full record (--detail): what the compact line was cut from
    with open(detail) as f:
        full = json.load(f)
    assert abs(full["value"] - d["value"]) <= 1e-6 * d["value"] and full["roofline"]["steady_state"]["steps"] == 200
    roof = full["roofline"]
    if roof["traffic_measured_in_this_run"]:
        assert 0.9 <= roof["traffic_over_algorithmic"] <= 1.2, roof["traffic_detail"]
        assert roof["traffic_detail"]["dispatches"] >= 50
    else:
        assert roof["traffic_detail"]["fallback_reason"] and roof["traffic_source"].startswith("profiles/")
    # the CPU baseline: the UNMODIFIED reference timed in this run (one thread, tensor-only) is the value; the OpenMP port of the
    # algorithm (oracle/) beside it; the HIP outputs against the reference's own on the same rows
    cpu = full["cpu_baseline"]
    assert cpu["port"]["kind"] == "port" and cpu["port"]["value"] > 0 and cpu["port"]["cores"] >= 1
    assert cpu["reference"]["gpu_vs_reference_max_abs"]["pos"] <= 2e-6
    legs = {leg["name"]: leg for leg in full["configs"]["legs"]}
    assert set(legs) == {"config2", "config3_shard", "config3_whole", "config4", "config5"}
    for name, leg in legs.items():
        roof = leg["roofline"]
        assert roof["launch_us"] > 0 and 0 < roof["frac"] < 1.0, name
    c3 = legs["config3_whole"]
    assert own_off or (c3["own_kernel"] is True and c3["launch_us"] < c3["library_kernel_launch_us"])
    assert own_off or (legs["config4"]["own_kernel"] is True and legs["config4"]["launch_us"] < legs["config4"]["library_kernel_launch_us"])
