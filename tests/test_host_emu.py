"""Kernel arithmetic + walk encoding on the CPU (not gpu): csrc/drm_sample.hpp compiled with g++
(tests/host_emu/host_emu.cpp) against the fp64 oracle, for every shipped robot.

This covers what can be checked without a GPU: the per-sample math of the HIP kernels, the axis
canonicalisation, the depth-first walk programs (slots, padding) and the constant tables.  The
tile I/O and launches are covered by the `-m gpu` tests.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.backend import DrmWalk
from differentiable_robot_model_amd.flatten import OPI_PERM, build_walk, fold_link_table, foldable_links
from helpers import ALL_ROBOTS, load_model, max_err, quat_close, sample_states
from oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))


ROCM_CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# g++ AND the ROCm clang (the device compiler's front end: a clang-only quirk in the shared header once
# produced wrong sin/cos on the GPU while the g++ build was fine)
COMPILERS = ["g++"] + (["clang++"] if os.path.exists(ROCM_CLANG) else [])


@pytest.fixture(scope="module", params=COMPILERS)
def emu(request):
    cxx = request.param
    src = os.path.join(HERE, "host_emu", "host_emu.cpp")
    lib = os.path.join(HERE, "host_emu", "libdrm_host_emu%s.so" % ("" if cxx == "g++" else "_clang"))
    csrc = os.path.join(HERE, "..", "differentiable-robot-model_amd", "csrc")
    deps = [src, os.path.join(csrc, "drm_sample.hpp"), os.path.join(csrc, "drm_tree.hpp"), os.path.join(csrc, "drm_host_loops.hpp"),
            os.path.join(HERE, "..", "include", "drm_hip.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
        # -O1: the straight-line templates take minutes at -O2/-O3 and the checks here are about arithmetic and
        # front-end semantics (the clang quirk above reproduces at every optimisation level), not about speed
        tmp = "%s.%d.tmp" % (lib, os.getpid())   # (pytest-xdist workers may all find the library stale at once)
        subprocess.check_call([cxx if cxx == "g++" else ROCM_CLANG, "-O1", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=fast", "-mfma", "-w", "-o", tmp, src])
        os.replace(tmp, lib)
    return ctypes.CDLL(lib)


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def host_walk(model, prog):
    table = model._link_table().detach().cpu().numpy().reshape(-1)
    ops_f = np.ascontiguousarray((table[prog.gather.reshape(-1)] * prog.gsign.reshape(-1)).reshape(prog.capacity, 32),
                                 np.float32)
    perm = int(prog.ops_i[prog.n_ops - 1, OPI_PERM]) if prog.n_ops else 2
    from differentiable_robot_model_amd.backend import fill_walk_struct
    walk = fill_walk_struct(DrmWalk, prog, ops_f.ctypes.data, prog.ops_i_dev.ctypes.data, model._n_dofs, perm)
    return walk, ops_f


@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_fk_all_links(emu, robot):
    m = load_model(robot)
    L, n, B = len(m._bodies), m._n_dofs, 33
    q, _, _ = sample_states(m, B, seed=1)
    targets = list(range(1, L))
    prog = build_walk(m._spec, targets=targets)
    walk, keep = host_walk(m, prog)
    pos = np.zeros((B, len(targets), 3), np.float32); quat = np.zeros((B, len(targets), 4), np.float32)
    assert emu.emu_fk(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), len(targets), _ptr(pos), _ptr(quat)) == 0
    op, oq = Oracle(m._spec).fk(q.astype(np.float64), targets, np.float64)
    assert max_err(pos, op) < 1e-6
    ok, flips = quat_close(quat, oq, 1e-6)
    assert ok and flips == 0


@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_fk_all_links_in_walk_order_fanned_out_where_the_tree_has_a_hub(emu, robot):
    """compute_forward_kinematics_all_links asks for the links in walk order; a hand or an arm that carries one then splits
    behind its hub (flatten.fk_fan_partition, DRM_WALK_FK_FAN) and every wavefront walks the shared part plus its run with
    nothing from the others — the emulation gives each run save slots of its own, poisoned with NaN."""
    from differentiable_robot_model_amd.flatten import SHAPE_FK_FAN
    m = load_model(robot)
    B = 17
    q, _, _ = sample_states(m, B, seed=2)
    targets = [i for i in m._spec.preorder() if i != 0]
    prog = build_walk(m._spec, targets=targets)
    if robot in ("allegro_left", "iiwa7_allegro", "trifinger_edu"):
        assert prog.shape & SHAPE_FK_FAN and len(prog.seg_begin) >= 3
    if prog.shape & SHAPE_FK_FAN:
        assert prog.seg_begin[0] == prog.prefix_end and prog.seg_begin[-1] == prog.n_ops
        assert all(a < b for a, b in zip(prog.seg_begin, prog.seg_begin[1:]))
    walk, keep = host_walk(m, prog)
    pos = np.full((B, len(targets), 3), np.nan, np.float32); quat = np.full((B, len(targets), 4), np.nan, np.float32)
    assert emu.emu_fk(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), len(targets), _ptr(pos), _ptr(quat)) == 0
    op, oq = Oracle(m._spec).fk(q.astype(np.float64), targets, np.float64)
    assert max_err(pos, op) < 1e-6
    ok, flips = quat_close(quat, oq, 1e-6)
    assert ok and flips == 0


@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_jacobian_every_link(emu, robot):
    m = load_model(robot)
    L, n, B = len(m._bodies), m._n_dofs, 17
    q, _, _ = sample_states(m, B, seed=2)
    orc = Oracle(m._spec)
    for link in range(0, L):
        prog = build_walk(m._spec, targets=[link] if link else [])
        walk, keep = host_walk(m, prog)
        pos = np.zeros((B, 3), np.float32); quat = np.zeros((B, 4), np.float32)
        lin = np.full((B, 3, n), np.nan, np.float32); ang = np.full((B, 3, n), np.nan, np.float32)
        assert emu.emu_fk_jacobian(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(pos), _ptr(quat),
                                   _ptr(lin), _ptr(ang)) == 0
        op, oq, ol, oa = orc.fk_jacobian(q.astype(np.float64), link, np.float64)
        assert max_err(pos, op) < 1e-6 and max_err(lin, ol) < 1e-6 and max_err(ang, oa) < 1e-6, (robot, link)
        ok, _ = quat_close(quat, oq, 1e-6)
        assert ok, (robot, link)


@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "fetch_arm_no_gripper"])
def test_arm_chain_packed_arithmetic(emu, robot):
    """The packed-FP32 chain arithmetic of fk_jacobian_arm_kernel<8, 7> (pairs, two-joint sincos, fp64 fallback
    for huge angles) against the fp64 oracle."""
    m = load_model(robot)
    L, n = len(m._bodies), m._n_dofs
    prog = build_walk(m._spec, targets=[L - 1])
    assert prog.shape & 1 and prog.capacity == 8 and n == 7
    walk, keep = host_walk(m, prog)
    q, _, _ = sample_states(m, 257, seed=21)
    q[5] *= 100.0          # beyond any joint range, still on the fp32 reduction
    q[6, 3] = 2.5e5        # one huge angle: the whole sample takes the fp64 reduction
    q[7] = 0.0
    B = q.shape[0]
    pos = np.zeros((B, 3), np.float32); quat = np.zeros((B, 4), np.float32)
    lin = np.full((B, 3, n), np.nan, np.float32); ang = np.full((B, 3, n), np.nan, np.float32)
    assert emu.emu_fk_jacobian_arm(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(pos), _ptr(quat), _ptr(lin),
                                   _ptr(ang)) == 0
    op, oq, ol, oa = Oracle(m._spec).fk_jacobian(q.astype(np.float64), L - 1, np.float64)
    assert max_err(pos, op) < 2e-6 and max_err(lin, ol) < 2e-6 and max_err(ang, oa) < 1e-6
    ok, _ = quat_close(quat, oq, 1e-6)
    assert ok
    # walks that skip DoF columns (a finger of a hand) are not arm-shaped
    assert not build_walk(load_model("allegro_left")._spec, targets=[10]).shape & 1


@pytest.mark.parametrize("robot", ALL_ROBOTS)
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_rnea(emu, robot, flags):
    m = load_model(robot)
    n, B = m._n_dofs, 29
    q, qd, qdd = sample_states(m, B, seed=3)
    prog = build_walk(m._spec, whole_tree=True)
    walk, keep = host_walk(m, prog)
    tau = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(tau)) == 0
    ot = Oracle(m._spec).rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64),
                              bool(flags & 1), bool(flags & 2), np.float64)
    assert np.allclose(tau, ot, atol=2e-5, rtol=2e-5), (robot, np.abs(tau - ot).max())
    # qdd = NULL == zero joint accelerations (compute_non_linear_effects)
    tau0 = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), None, ctypes.c_int64(B), flags, _ptr(tau0)) == 0
    o0 = Oracle(m._spec).rnea(q.astype(np.float64), qd.astype(np.float64), np.zeros_like(q, np.float64),
                              bool(flags & 1), bool(flags & 2), np.float64)
    assert np.allclose(tau0, o0, atol=2e-5, rtol=2e-5)
    # robots whose segments are short (the fingers of a hand) take the register-parked form of the walk: the same steps,
    # unrolled (the host compilers contract a few FMAs differently in the two forms, hence not bit for bit here)
    if max(b - a for a, b in zip(prog.seg_begin, prog.seg_begin[1:])) <= 6:
        tau_s = np.full((B, n), np.nan, np.float32)
        assert emu.emu_rnea_short(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(tau_s)) == 0
        assert np.allclose(tau_s, tau, atol=1e-6, rtol=1e-6) and np.allclose(tau_s, ot, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "fetch_arm_no_gripper", "fetch_arm_no_gripper_small_damping"])
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_arm_chain_rnea(emu, robot, flags):
    """rnea_chain (the arithmetic of rnea_arm_kernel<8, 7>) against the fp64 oracle."""
    m = load_model(robot)
    n, B = m._n_dofs, 65
    q, qd, qdd = sample_states(m, B, seed=31)
    q[3, 2] = -3.0e5  # fp64 reduction fallback
    prog = build_walk(m._spec, whole_tree=True)
    assert prog.shape & 1 and prog.capacity == 8 and n == 7
    walk, keep = host_walk(m, prog)
    tau = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea_arm(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(tau)) == 0
    ot = Oracle(m._spec).rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64),
                              bool(flags & 1), bool(flags & 2), np.float64)
    assert np.allclose(tau, ot, atol=2e-5, rtol=2e-5), (robot, np.abs(tau - ot).max())
    tau0 = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea_arm(ctypes.byref(walk), _ptr(q), _ptr(qd), None, ctypes.c_int64(B), flags, _ptr(tau0)) == 0
    o0 = Oracle(m._spec).rnea(q.astype(np.float64), qd.astype(np.float64), np.zeros_like(q, np.float64),
                              bool(flags & 1), bool(flags & 2), np.float64)
    assert np.allclose(tau0, o0, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "fetch_arm_no_gripper_small_damping"])
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
@pytest.mark.parametrize("folded", [False, True])
def test_arm_chain_rnea_two_samples_per_lane(emu, robot, flags, folded):
    """rnea_chain2_trig (the arithmetic of rnea_arm2_kernel: every quantity a pair over two samples) against the fp64 oracle
    and against the one-sample form; on the full 8-link table and on the 7-link table with the fixed tail folded."""
    m = load_model(robot)
    n, B = m._n_dofs, 67
    q, qd, qdd = sample_states(m, B, seed=37)
    q[4, 1] = 2.0e5  # fp64 reduction fallback for one sample of a pair
    if folded:
        prog = build_walk(m._spec, whole_tree=True, drop_folded=True)
        assert prog.n_ops == 7 and prog.capacity == 8
        walk, keep = folded_host_walk(m, prog)
    else:
        prog = build_walk(m._spec, whole_tree=True)
        walk, keep = host_walk(m, prog)
    tau = np.full((B, n), np.nan, np.float32); tau1 = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea_arm2(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(tau)) == 0
    assert emu.emu_rnea_arm(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(tau1)) == 0
    ot = Oracle(m._spec).rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64),
                              bool(flags & 1), bool(flags & 2), np.float64)
    assert np.allclose(tau, ot, atol=2e-5, rtol=2e-5), (robot, np.abs(tau - ot).max())
    assert np.allclose(tau, tau1, atol=1e-5, rtol=1e-5)
    tau0 = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea_arm2(ctypes.byref(walk), _ptr(q), _ptr(qd), None, ctypes.c_int64(B), flags, _ptr(tau0)) == 0
    o0 = Oracle(m._spec).rnea(q.astype(np.float64), qd.astype(np.float64), np.zeros_like(q, np.float64),
                              bool(flags & 1), bool(flags & 2), np.float64)
    assert np.allclose(tau0, o0, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("robot,shape", [("allegro_left", (4, 4)), ("trifinger_edu", (3, 3))])
@pytest.mark.parametrize("flags", [0, 3])
def test_fingers_two_samples_per_lane(emu, robot, shape, flags):
    """flatten.fingers_shape finds (K, L) of a hand's folded dynamics walk (K serial chains of L revolute ops off the root, op k
    = DoF k); the arithmetic of rnea_fingers2_kernel (the arm recursion per finger, two samples per lane, all body forces in
    registers) against the fp64 oracle and the loop form, with and without qdd, odd batch."""
    from differentiable_robot_model_amd.flatten import SHAPE_FINGERS
    m = load_model(robot)
    prog = build_walk(m._spec, whole_tree=True, drop_folded=True)
    sh = prog.shape & 0xffffffff
    assert sh & SHAPE_FINGERS and (((sh >> 28) & 3) + 1, ((sh >> 30) & 3) + 1) == shape
    walk, keep = folded_host_walk(m, prog)
    n, B = m._n_dofs, 21
    q, qd, qdd = sample_states(m, B, seed=71)
    q[3, 1] = 3.1e5
    orc = Oracle(m._spec)
    for acc in (qdd, None):
        tau = np.full((B, n), np.nan, np.float32); tau_loop = np.full((B, n), np.nan, np.float32)
        assert emu.emu_rnea_fingers(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(acc) if acc is not None else None, ctypes.c_int64(B),
                                    flags, _ptr(tau)) == 0
        assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(acc) if acc is not None else None, ctypes.c_int64(B), flags,
                            _ptr(tau_loop)) == 0
        ref = orc.rnea(q.astype(np.float64), qd.astype(np.float64), (acc if acc is not None else np.zeros_like(q)).astype(np.float64),
                       bool(flags & 1), bool(flags & 2), np.float64)
        assert np.allclose(tau, ref, atol=2e-5, rtol=2e-5), (robot, float(np.abs(tau - ref).max()))
        assert np.allclose(tau, tau_loop, atol=1e-5, rtol=1e-5)
    for other in ("panda", "iiwa7", "fetch"):
        assert not (build_walk(load_model(other)._spec, whole_tree=True, drop_folded=True).shape & SHAPE_FINGERS), other


def arm_hand_case(case):
    """"robot[:sliding-fingers][+kept link]" -> (model, folded dynamics walk, its host struct).  ":sliding-fingers": the
    gripper's prismatic joints modelled as such (reference_compat=False), not as the reference does; "+link": that link keeps
    an op of its own, as it does when it has learnable parameters (the flange / palm stays in the prefix)."""
    name, _, kept = case.partition("+")
    m = load_model(name.split(":")[0], reference_compat=":" not in name)
    fold = foldable_links(m._spec, keep=[m._name_to_idx_map[kept]] if kept else [])
    prog = build_walk(m._spec, whole_tree=True, drop_folded=True, fold=fold)
    walk, keep = folded_host_walk(m, prog, fold)
    return m, prog, walk, keep


ARM_HAND_CASES = ["panda", "jaco", "iiwa7_allegro", "panda:sliding-fingers", "panda+panda_hand", "jaco+j2n6s300_link_base",
                  "iiwa7_allegro+palm_link"]
ARM_HAND_SHAPES = [(7, 2, 1), (6, 3, 2), (7, 4, 4), (7, 2, 1), (9, 3, 1), (7, 3, 2), (8, 4, 4)]


@pytest.mark.parametrize("robot,shape", list(zip(ARM_HAND_CASES, ARM_HAND_SHAPES)))
@pytest.mark.parametrize("flags", [0, 3])
def test_arm_that_carries_a_hand(emu, robot, shape, flags):
    """flatten.arm_hand_shape finds (P, K, L) of the folded dynamics walk; rnea_arm_hand (the arithmetic of rnea_arm_hand_kernel:
    sub-chains swept forward and backward while the palm's motion is live, only the prefix forces parked) against the fp64
    oracle and against the loop form of the walk."""
    from differentiable_robot_model_amd.flatten import SHAPE_ARM_HAND
    m, prog, walk, keep = arm_hand_case(robot)
    n, B = m._n_dofs, 23
    q, qd, qdd = sample_states(m, B, seed=41)
    q[5, 2] = -2.5e5
    sh = prog.shape & 0xffffffff
    assert sh & SHAPE_ARM_HAND and ((sh >> 24) & 0xf, ((sh >> 28) & 3) + 1, ((sh >> 30) & 3) + 1) == shape
    tau = np.full((B, n), np.nan, np.float32); tau_loop = np.full((B, n), np.nan, np.float32)
    assert emu.emu_rnea_arm_hand(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(tau)) == 0
    assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(tau_loop)) == 0
    ot = Oracle(m._spec).rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), bool(flags & 1), bool(flags & 2), np.float64)
    assert np.allclose(tau, ot, atol=2e-5, rtol=2e-5), (robot, np.abs(tau - ot).max())
    assert np.allclose(tau, tau_loop, atol=1e-5, rtol=1e-5)
    # shapes that are NOT an arm with a hand
    for other in ("panda_no_gripper", "allegro_left", "fetch"):
        mo = load_model(other)
        assert not (build_walk(mo._spec, whole_tree=True, drop_folded=True).shape & SHAPE_ARM_HAND), other


@pytest.mark.parametrize("robot", ARM_HAND_CASES)
@pytest.mark.parametrize("flags", [0, 3])
def test_forward_dynamics_of_an_arm_that_carries_a_hand(emu, robot, flags):
    """aba_arm_hand (the arithmetic of forward_dynamics_arm_hand_kernel: the articulated-body recursion with the sub-chains
    visited twice instead of stored) against the fp64 oracle and against the loop form of the recursion."""
    m, prog, walk, keep = arm_hand_case(robot)
    n, B = m._n_dofs, 17
    q, qd, f = sample_states(m, B, seed=43)
    acc = np.full((B, n), np.nan, np.float32); acc_loop = np.full((B, n), np.nan, np.float32)
    assert emu.emu_forward_dynamics_arm_hand(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(f), ctypes.c_int64(B), flags, _ptr(acc)) == 0
    assert emu.emu_forward_dynamics(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(f), ctypes.c_int64(B), flags, _ptr(acc_loop)) == 0
    ref = Oracle(m._spec).forward_dynamics(q.astype(np.float64), qd.astype(np.float64), f.astype(np.float64), bool(flags & 1), bool(flags & 2), np.float64)
    rel = lambda a, b: float((np.abs(a - b) / (1.0 + np.abs(b))).max())
    assert rel(acc, ref) < 1e-3, (robot, rel(acc, ref))
    assert rel(acc, acc_loop) < 1e-3, (robot, rel(acc, acc_loop))


@pytest.mark.parametrize("robot", ARM_HAND_CASES)
@pytest.mark.parametrize("flags", [0, 3])
def test_inverse_dynamics_backward_of_an_arm_that_carries_a_hand(emu, robot, flags):
    """rnea_backward_arm_hand (the arithmetic of rnea_backward_arm_hand_kernel: nothing stored per link, the palm's motion and
    force adjoint put away once, every sub-chain adding its force and motion adjoint to the sums the prefix walks back with)
    against the loop form of the adjoint walk (itself held to the reference's autograd by the golden gradient tests): input
    gradients and the constant gradients of a learnable prefix link, palm and fingertip."""
    m, prog, walk, keep = arm_hand_case(robot)
    n, B = m._n_dofs, 13
    q, qd, qdd = sample_states(m, B, seed=53)
    gtau = np.random.default_rng(5).standard_normal((B, n)).astype(np.float32)
    mask = (1 << 1) | (1 << (((prog.shape >> 24) & 0xf) - 1)) | (1 << (prog.n_ops - 1))
    outs = []
    for fn in (emu.emu_rnea_backward_arm_hand, emu.emu_rnea_backward):
        gq, gqd, gqdd = (np.full((B, n), np.nan, np.float32) for _ in range(3))
        gops = np.zeros((prog.capacity, 32), np.float32)
        assert fn(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), flags, _ptr(gtau), ctypes.c_uint64(mask),
                  _ptr(gq), _ptr(gqd), _ptr(gqdd), _ptr(gops)) == 0
        outs.append((gq, gqd, gqdd, gops))
    for a, b, name in zip(outs[0], outs[1], ("grad_q", "grad_qd", "grad_qdd", "grad_ops_f")):
        scale = max(1.0, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 2e-5 * scale, (robot, name, float(np.abs(a - b).max()), scale)
    assert np.abs(outs[0][3]).max() > 0
    # without qdd (the non-linear effects' gradient)
    g0 = [np.full((B, n), np.nan, np.float32) for _ in range(3)]
    g1 = [np.full((B, n), np.nan, np.float32) for _ in range(3)]
    for fn, gs in ((emu.emu_rnea_backward_arm_hand, g0), (emu.emu_rnea_backward, g1)):
        assert fn(ctypes.byref(walk), _ptr(q), _ptr(qd), None, ctypes.c_int64(B), flags, _ptr(gtau), ctypes.c_uint64(0),
                  _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), None) == 0
    assert np.abs(g0[0] - g1[0]).max() <= 2e-5 * max(1.0, float(np.abs(g1[0]).max()))


@pytest.mark.parametrize("robot", ARM_HAND_CASES)
def test_mass_matrix_of_an_arm_that_carries_a_hand(emu, robot):
    """crba_arm_hand (the arithmetic of crba_arm_hand_kernel: a sub-chain's column forces carried together up the prefix, nothing
    parked) against the fp64 oracle and the loop form of the composite-rigid-body walk; structural zeros stay zero."""
    m, prog, walk, keep = arm_hand_case(robot)
    n, B = m._n_dofs, 11
    q, _, _ = sample_states(m, B, seed=47)
    H = np.full((B, n, n), np.nan, np.float32); H_loop = np.full((B, n, n), np.nan, np.float32)
    assert emu.emu_crba_arm_hand(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(H)) == 0
    assert emu.emu_crba(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(H_loop)) == 0
    ref = Oracle(m._spec).mass_matrix(q.astype(np.float64), False, False, np.float64)
    assert np.allclose(H, ref, atol=2e-5, rtol=2e-5), (robot, np.abs(H - ref).max())
    assert np.allclose(H, H_loop, atol=1e-5, rtol=1e-5)
    assert np.array_equal(H, np.swapaxes(H, 1, 2))


def test_sincos_large_arguments(emu):
    """The kernels' branch-free sincos keeps fp32 accuracy far outside any joint range."""
    m = load_model("2link_robot")
    prog = build_walk(m._spec, targets=[3])
    walk, keep = host_walk(m, prog)
    q = np.array([[1e4, -3e5], [123456.7, 1e6], [-7e6, 2.5e7], [0.0, -0.0], [np.pi, -np.pi / 2]], np.float32)
    B = q.shape[0]
    pos = np.zeros((B, 1, 3), np.float32); quat = np.zeros((B, 1, 4), np.float32)
    assert emu.emu_fk(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), 1, _ptr(pos), _ptr(quat)) == 0
    op, oq = Oracle(m._spec).fk(q.astype(np.float64), [3], np.float64)
    assert max_err(pos, op) < 2e-6


def test_link_rows_and_their_derivative_vs_torch(emu):
    """link_row / link_row_backward (the fused parameter kernels) against the torch construction of the same rows
    (robot_model._link_rows, which mirrors rigid_body.py:138-143 and spatial_vector_algebra.py:321-327) and its
    autograd."""
    import torch
    m = load_model("iiwa7")
    links = [1, 4, 8]
    params = m._link_params(links).detach().clone()
    params += 0.1 * torch.randn(params.shape, generator=torch.Generator().manual_seed(0))   # generic rpy / com / inertia
    p_np = np.ascontiguousarray(params.numpy(), np.float32)
    rows = np.zeros((len(links), 32), np.float32)
    assert emu.emu_link_rows(_ptr(p_np), len(links), _ptr(rows)) == 0

    def torch_rows(p):   # the same map written with torch ops
        rpy, trans, mass, com, inertia, damp = p[:, 0:3], p[:, 3:6], p[:, 6:7], p[:, 7:10], p[:, 10:19].reshape(-1, 3, 3), p[:, 19:20]
        c, s = torch.cos(rpy), torch.sin(rpy)
        one, zero = torch.ones(len(p)), torch.zeros(len(p))
        mat = lambda r: torch.stack([torch.stack(x, dim=-1) for x in r], dim=-2)
        Rx = mat([[one, zero, zero], [zero, c[:, 0], -s[:, 0]], [zero, s[:, 0], c[:, 0]]])
        Ry = mat([[c[:, 1], zero, s[:, 1]], [zero, one, zero], [-s[:, 1], zero, c[:, 1]]])
        Rz = mat([[c[:, 2], -s[:, 2], zero], [s[:, 2], c[:, 2], zero], [zero, zero, one]])
        S = mat([[zero, -com[:, 2], com[:, 1]], [com[:, 2], zero, -com[:, 0]], [-com[:, 1], com[:, 0], zero]])
        Io = inertia + mass.reshape(-1, 1, 1) * (S @ S.transpose(-2, -1))
        return torch.cat([((Rz @ Ry) @ Rx).reshape(-1, 9), trans, mass, com * mass, Io.reshape(-1, 9), damp,
                          torch.zeros(len(p), 6)], dim=1)

    pt = params.clone().double().requires_grad_(True)
    ref = torch_rows(pt.float() if False else pt)
    assert np.abs(rows - ref.detach().numpy()).max() < 1e-6
    g = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    g[:, 26:] = 0
    ref.backward(g)
    g_np = np.ascontiguousarray(g.numpy(), np.float32)
    gp = np.zeros_like(p_np)
    assert emu.emu_link_rows_backward(_ptr(p_np), _ptr(g_np), len(links), _ptr(gp)) == 0
    assert np.abs(gp - pt.grad.numpy()).max() < 2e-6 * max(1.0, np.abs(pt.grad.numpy()).max())


def test_quaternion_on_branch_boundaries(emu):
    """The select-based quaternion of the kernels on the case boundaries of the reference's algorithm
    (tests/golden/golden_quat_branches.npz): the reference's sign wherever the margin exceeds 1e-5."""
    from test_oracle_golden import check_quaternion_branches

    def fk(robot, q, link):
        m = load_model(robot)
        prog = build_walk(m._spec, targets=[m._name_to_idx_map[link]])
        walk, _keep = host_walk(m, prog)
        B = q.shape[0]
        pos = np.zeros((B, 3), np.float32); quat = np.zeros((B, 4), np.float32)
        lin = np.zeros((B, 3, 7), np.float32); ang = np.zeros((B, 3, 7), np.float32)
        qc = np.ascontiguousarray(q, np.float32)
        assert emu.emu_fk_jacobian_arm(ctypes.byref(walk), _ptr(qc), ctypes.c_int64(B), _ptr(pos), _ptr(quat), _ptr(lin), _ptr(ang)) == 0
        p2 = np.zeros((B, 1, 3), np.float32); q2 = np.zeros((B, 1, 4), np.float32)
        assert emu.emu_fk(ctypes.byref(walk), _ptr(qc), ctypes.c_int64(B), 1, _ptr(p2), _ptr(q2)) == 0
        assert np.abs(p2[:, 0] - pos).max() < 1e-6          # the loop-structured walk agrees with the arm chain
        return pos, quat
    check_quaternion_branches(fk)


# ---------------------------------------------------------------------------------------------- folded fixed links
def folded_host_walk(model, prog, fold=None):
    """host_walk on the folded link table (flatten.fold_link_table; robot_model._folded_table on the device)."""
    L1 = len(model._bodies) + 1
    base = fold_link_table(model._spec, model._link_table().detach().cpu().numpy()[:L1], fold)
    table = model._with_virtual_rows(torch.from_numpy(base.astype(np.float32))).numpy().reshape(-1)
    ops_f = np.ascontiguousarray((table[prog.gather.reshape(-1)] * prog.gsign.reshape(-1)).reshape(prog.capacity, 32),
                                 np.float32)
    from differentiable_robot_model_amd.backend import fill_walk_struct
    walk = fill_walk_struct(DrmWalk, prog, ops_f.ctypes.data, prog.ops_i_dev.ctypes.data, model._n_dofs, 2)
    return walk, ops_f


@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "allegro_left", "trifinger_edu", "panda", "fetch"])
def test_folding_fixed_links_into_their_parents_leaves_the_dynamics_unchanged(emu, robot):
    """flatten.foldable_links / fold_link_table: a dynamics walk without the links behind fixed joints (tool frames and
    fingertips, and the flanges / palms / mounting plates between moving joints), on a table whose rows carry their inertia
    (the nearest moving ancestor's) and their transforms (the links below), gives the torques, inertia matrix and
    accelerations of the full walk (and of the oracle, which knows nothing about folding)."""
    m = load_model(robot)
    fold = foldable_links(m._spec)
    assert fold.any()
    full = build_walk(m._spec, whole_tree=True)
    short = build_walk(m._spec, whole_tree=True, drop_folded=True)
    assert short.n_ops == full.n_ops - int(fold[1:].sum())
    B, n = 19, m._n_dofs
    q, qd, qdd = sample_states(m, B, seed=31)
    res = []
    for walk, _keep in (host_walk(m, full), folded_host_walk(m, short)):
        tau = np.zeros((B, n), np.float32); H = np.zeros((B, n, n), np.float32); acc = np.zeros((B, n), np.float32)
        assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(tau)) == 0
        assert emu.emu_crba(ctypes.byref(walk), _ptr(q), ctypes.c_int64(B), _ptr(H)) == 0
        assert emu.emu_forward_dynamics(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(acc)) == 0
        res.append((tau, H, acc))
    (t0, H0, a0), (t1, H1, a1) = res
    assert np.allclose(t1, t0, rtol=2e-5, atol=2e-5) and np.allclose(H1, H0, rtol=2e-5, atol=2e-5)
    assert (np.abs(a1 - a0) / (1 + np.abs(a0))).max() < 2e-3
    orc = Oracle(m._spec)
    assert np.allclose(t1, orc.rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), True, True, np.float64),
                       rtol=2e-5, atol=2e-5)
    # the chain walk to a folded link on the folded table (drm_fk_rnea's second walk): it steps over the folded links whose
    # transforms the rows below them carry, keeps the folded links at its end — the same pose as the full chain on the plain
    # table, the target's own inertia gone
    tip = int(np.nonzero(fold)[0][-1])
    chain0 = build_walk(m._spec, targets=[tip])
    chain = build_walk(m._spec, targets=[tip], fold=fold)
    assert chain.n_ops <= chain0.n_ops
    w1, ops_f = folded_host_walk(m, chain, fold)
    w0, _ops_f0 = host_walk(m, chain0)
    assert not ops_f[chain.n_ops - 1, 12:25].any()
    poses = []
    for w in (w0, w1):
        pos = np.zeros((B, 1, 3), np.float32); quat = np.zeros((B, 1, 4), np.float32)
        assert emu.emu_fk(ctypes.byref(w), _ptr(q), ctypes.c_int64(B), 1, _ptr(pos), _ptr(quat)) == 0
        poses.append((pos, quat))
    assert max_err(poses[0][0], poses[1][0]) < 2e-6 and quat_close(poses[0][1][:, 0], poses[1][1][:, 0], 2e-6)[0]
    # ... and so for every link of the robot as the target
    for link in range(1, m._spec.n_links):
        poses = []
        plain, stepped = build_walk(m._spec, targets=[link]), build_walk(m._spec, targets=[link], fold=fold)   # (own the tables)
        for w, _k in (host_walk(m, plain), folded_host_walk(m, stepped, fold)):
            pos = np.zeros((4, 1, 3), np.float32); quat = np.zeros((4, 1, 4), np.float32)
            assert emu.emu_fk(ctypes.byref(w), _ptr(q), ctypes.c_int64(4), 1, _ptr(pos), _ptr(quat)) == 0
            poses.append((pos, quat))
        assert max_err(poses[0][0], poses[1][0]) < 2e-6, m._spec.link_names[link]
        assert quat_close(poses[0][1][:, 0], poses[1][1][:, 0], 2e-6)[0], m._spec.link_names[link]


def test_folding_keeps_learnable_links_and_what_would_fold_into_them(emu):
    """foldable_links(keep=...): a link with learnable parameters stays an op of its own, and so does a fixed leaf whose fold
    target is such a link; everything else still folds, and the dynamics are unchanged."""
    m = load_model("allegro_left")
    names = m._name_to_idx_map
    tip, tip_parent = names["link_3.0_tip"], names["link_3.0"]
    assert m._spec.parent[tip] == tip_parent
    everything = foldable_links(m._spec)
    assert everything[tip] and everything.sum() >= 4
    keep_tip = foldable_links(m._spec, keep=[tip])
    assert not keep_tip[tip] and keep_tip.sum() == everything.sum() - 1
    keep_parent = foldable_links(m._spec, keep=[tip_parent])          # the tip would fold INTO a learnable link: it stays
    assert not keep_parent[tip] and not keep_parent[tip_parent] and keep_parent.sum() == everything.sum() - 1
    B, n = 11, m._n_dofs
    q, qd, qdd = sample_states(m, B, seed=8)
    full = build_walk(m._spec, whole_tree=True)
    walk, _k = host_walk(m, full)
    ref = np.zeros((B, n), np.float32)
    assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(ref)) == 0
    for fold in (keep_tip, keep_parent):
        short = build_walk(m._spec, whole_tree=True, drop_folded=True, fold=fold)
        assert short.n_ops == full.n_ops - int(fold.sum())
        swalk, _k2 = folded_host_walk(m, short, fold)
        tau = np.zeros((B, n), np.float32)
        assert emu.emu_rnea(ctypes.byref(swalk), _ptr(q), _ptr(qd), _ptr(qdd), ctypes.c_int64(B), 3, _ptr(tau)) == 0
        assert np.allclose(tau, ref, rtol=2e-5, atol=2e-5)
