"""Per-robot straight-line dynamics kernels for arbitrary trees (specialize.py, csrc/drm_static.hpp): the tree of a robot's
whole-tree walk as compile-time constants, compiled at run time (hipcc, cached) and launched through drm_walk.special[] by the
C ABI.  What the reference does with the same Python loop for every robot (robot_model.py:173-193, 262-301).

CPU (not gpu): the tree decoded from the control words, the generated translation unit, and that hipcc builds it.
GPU (-m gpu): inverse dynamics of Fetch (both joint models) and of random trees through their OWN kernels against the fp64 oracle
and against the loop kernels; ragged batches, misaligned slices.
"""
import contextlib
import os

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd import specialize as sp
from differentiable_robot_model_amd.flatten import build_walk
from helpers import TOL_TAU, load_model, sample_states
from oracle import Oracle
from test_host_emu import emu  # noqa: F401  (fixture)
from test_random_trees import tree_model

needs_hipcc = pytest.mark.skipif(sp.hipcc() is None, reason="hipcc not on this machine")


@pytest.fixture(autouse=True)
def _own_kernels_as_shipped(monkeypatch):
    """This file tests the own-kernel machinery itself: DRM_SPECIALIZE in the caller's environment (e.g. =0 to run every OTHER suite on
    the library's kernels) does not reach it; the tests of that switch set it themselves."""
    monkeypatch.delenv("DRM_SPECIALIZE", raising=False)


def library_only(m):
    """A model that keeps to the library's kernels (round 6: by default a model attaches whatever own kernel is already built —
    shipped, or left in the run-time cache by an earlier test): the other side of every own-vs-library comparison below."""
    m.own_kernels = "off"
    return m


def folded_walk(m):
    key = m._fold_key()
    fold = m._fold_masks[key]
    return build_walk(m._spec, whole_tree=True, drop_folded=True, fold=fold) if fold.any() else build_walk(m._spec, whole_tree=True)


def test_the_tree_of_a_walk_comes_from_its_control_words():
    m = load_model("fetch", reference_compat=False)
    t = sp.walk_tree(folded_walk(m))
    # wheels, torso (slides) off the root; head pan -> tilt and the arm off the torso; two sliding fingers off the wrist
    assert t["parent"] == [-1, -1, -1, 2, 3, 2, 5, 6, 7, 8, 9, 10, 11, 11]
    assert t["dof"] == list(range(14)) and [k for k, p in enumerate(t["prismatic"]) if p] == [2, 12, 13]
    assert not any(sp.walk_tree(folded_walk(load_model("fetch")))["prismatic"])          # (reference_compat: every joint turns)
    for robot in ("jaco", "allegro_left", "trifinger_edu", "iiwa7_allegro"):
        mm = load_model(robot)
        prog = folded_walk(mm)
        tt = sp.walk_tree(prog)
        links = [int(x) for x in prog.links[:prog.n_ops]]
        for k, par in enumerate(tt["parent"]):     # the parent op carries the nearest moving ancestor's link
            anc = int(mm._spec.parent[links[k]])
            while anc > 0 and anc not in links:
                anc = int(mm._spec.parent[anc])
            assert (links[par] if par >= 0 else 0) == (anc if anc in links else 0), (robot, k)


@needs_hipcc
def test_generated_source_builds_and_is_cached(tmp_path, monkeypatch):
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(tmp_path))
    m = load_model("fetch", reference_compat=False)
    src = sp.source(sp.walk_tree(folded_walk(m)), m._n_dofs)
    assert "N = 14, NDOF = 14" in src and "drm_rnea_static" in src and "drm_crba_static" in src and "drm_fd_static" in src
    assert "drm_rnea_backward_static" not in src                              # (the reverse-mode kernel needs the table's pitch)
    assert "drm_rnea_backward_static" in sp.source(sp.walk_tree(folded_walk(m)), m._n_dofs, 16)
    path = sp.build(src)
    assert os.path.getsize(path) > 1000 and path.startswith(str(tmp_path))
    stamp = os.path.getmtime(path)
    assert sp.build(src) == path and os.path.getmtime(path) == stamp          # second call: the cache
    assert sp.build(src.replace("N = 14", "N = 14 ")) != path                  # another robot: another code object


# ------------------------------------------------------------------------------------------------ CPU: the walks themselves
def host_harness(tmp_path, tree, n_dofs, cxx="g++"):
    """The straight-line walks of csrc/drm_static.hpp instantiated on `tree` for the HOST (tests/host_emu/static_emu.hpp), from the
    same `struct Robot` text the device code object is built from."""
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    cpp, lib = os.path.join(str(tmp_path), "static_host.cpp"), os.path.join(str(tmp_path), "libstatic_host.so")
    with open(cpp, "w") as f:
        f.write(sp.host_source(tree, n_dofs, os.path.join(here, "host_emu", "static_emu.hpp")))
    subprocess.check_call([cxx, "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast", "-mfma", "-w", "-o", lib, cpp])
    return ctypes.CDLL(lib)


def _static_vs_loops(tmp_path, emu, m, seed):
    import ctypes
    from differentiable_robot_model_amd import backend
    from test_host_emu import _ptr
    dw = m._dynamics_walk()
    prog, n = dw.program, m._n_dofs
    tree = sp.walk_tree(prog)
    from test_host_emu import ROCM_CLANG
    st = host_harness(tmp_path, tree, n, ROCM_CLANG if "_clang" in emu._name else "g++")     # (the compiler the emulation was built with)
    assert st.static_n_ops() == prog.n_ops
    ops_f = m._ops_f(dw).detach().contiguous()
    walk = backend._walk_struct_build(prog, ops_f, dw.ops_i, n)
    of, B = ctypes.c_void_p(ops_f.data_ptr()), 9
    q, qd, qdd = sample_states(m, B, seed=seed)
    gtau = np.random.default_rng(seed).standard_normal((B, n)).astype(np.float32)
    cb = ctypes.c_int64(B)
    # reverse mode: the same arithmetic in the same order as the loop walk -> the same bits from g++; clang forms its fused
    # multiply-adds per inlining context (as the device compiler does), so there the two agree to rounding
    same_bits = "_clang" not in emu._name
    mask = (1 << prog.n_ops) - 1 if prog.backward_ok else 0
    if prog.backward_ok:
        for flags, with_qdd in ((3, True), (0, True), (3, False)):
            a = [np.full((B, n), np.nan, np.float32) for _ in range(3)] + [np.zeros((prog.capacity, 32), np.float32)]
            b = [np.full((B, n), np.nan, np.float32) for _ in range(3)] + [np.zeros((prog.capacity, 32), np.float32)]
            dd = _ptr(qdd) if with_qdd else None
            assert emu.emu_rnea_backward(ctypes.byref(walk), _ptr(q), _ptr(qd), dd, cb, flags, _ptr(gtau), ctypes.c_uint64(mask),
                                         *[_ptr(x) for x in a]) == 0
            assert st.static_rnea_backward(of, prog.capacity, _ptr(q), _ptr(qd), dd, cb, flags, _ptr(gtau), ctypes.c_uint64(mask),
                                           *[_ptr(x) for x in b]) == 0
            for x, y, name in zip(a, b, ("grad_q", "grad_qd", "grad_qdd", "grad_ops_f")):
                err = float(np.abs(x - y).max())
                assert np.array_equal(x, y) if same_bits else err <= 2e-5 * max(1.0, float(np.abs(x).max())), (name, flags, with_qdd, err)
            assert np.abs(a[3]).max() > 0 and np.isfinite(a[0]).all()
    # forward walks: the loop forms to rounding
    ta, tb = np.zeros((B, n), np.float32), np.zeros((B, n), np.float32)
    assert emu.emu_rnea(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), cb, 3, _ptr(ta)) == 0
    assert st.static_rnea(of, _ptr(q), _ptr(qd), _ptr(qdd), cb, 3, _ptr(tb)) == 0
    assert np.allclose(ta, tb, **TOL_TAU)
    Ha, Hb = np.zeros((B, n, n), np.float32), np.zeros((B, n, n), np.float32)
    assert emu.emu_crba(ctypes.byref(walk), _ptr(q), cb, _ptr(Ha)) == 0 and st.static_crba(of, _ptr(q), cb, _ptr(Hb)) == 0
    assert np.allclose(Ha, Hb, **TOL_TAU)
    aa, ab = np.zeros((B, n), np.float32), np.zeros((B, n), np.float32)
    assert emu.emu_forward_dynamics(ctypes.byref(walk), _ptr(q), _ptr(qd), _ptr(qdd), cb, 3, _ptr(aa)) == 0
    assert st.static_fd(of, _ptr(q), _ptr(qd), _ptr(qdd), cb, 3, _ptr(ab)) == 0
    assert float((np.abs(aa - ab) / (1.0 + np.abs(aa))).max()) < 1e-3


@pytest.mark.parametrize("compat", [True, False])
def test_static_walks_of_fetch_on_the_host(tmp_path, emu, compat):
    """Every straight-line walk of csrc/drm_static.hpp, instantiated on Fetch's tree and compiled for the host, against the loop
    walks (csrc/drm_host_loops.hpp): reverse-mode inverse dynamics bit for bit (input gradients and the constant gradients of
    every op), inverse dynamics / inertia matrix / forward dynamics to rounding."""
    _static_vs_loops(tmp_path, emu, load_model("fetch", reference_compat=compat), 5)


@pytest.mark.parametrize("seed", [0, 3, 6])
def test_static_walks_of_random_trees_on_the_host(tmp_path, emu, seed):
    m = tree_model(tmp_path, seed)
    if m._dynamics_walk().program.n_ops > sp.MAX_STATIC_OPS:
        pytest.skip("more ops than the straight-line form is built for")
    _static_vs_loops(tmp_path, emu, m, seed)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("compat", [True, False])
def test_gpu_fetch_inverse_dynamics_through_its_own_kernel(compat):
    mc = load_model("fetch", reference_compat=compat)
    loop, own = library_only(load_model("fetch", "cuda", reference_compat=compat)), load_model("fetch", "cuda", reference_compat=compat)
    assert own.specialize() is True and own._dynamics_walk().program._special
    n = own._n_dofs
    orc = Oracle(mc._spec)
    for B in (64, 200, 4096 + 37):
        q, qd, qdd = sample_states(mc, B, seed=B)
        dq, dqd, dqdd = (torch.from_numpy(a).cuda() for a in (q, qd, qdd))
        for grav, damp in ((True, True), (False, False)):
            tau = own.compute_inverse_dynamics(dq, dqd, dqdd, include_gravity=grav, use_damping=damp)
            ref = orc.rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), grav, damp, np.float64)
            assert np.allclose(tau.cpu().numpy(), ref, **TOL_TAU), (B, grav, damp)
            other = loop.compute_inverse_dynamics(dq, dqd, dqdd, include_gravity=grav, use_damping=damp)
            assert np.allclose(tau.cpu().numpy(), other.cpu().numpy(), **TOL_TAU)
        H = own.compute_lagrangian_inertia_matrix(dq)
        Href = orc.mass_matrix(q.astype(np.float64), False, False, np.float64)
        assert np.allclose(H.cpu().numpy(), Href, **TOL_TAU), B
        assert np.allclose(H.cpu().numpy(), loop.compute_lagrangian_inertia_matrix(dq).cpu().numpy(), **TOL_TAU)
        assert torch.equal(H, H.transpose(1, 2))            # (both triangles from the same slot)
        for grav, damp in ((True, True), (False, False)):     # forward dynamics: the articulated-body recursion on torques that
            f = loop.compute_inverse_dynamics(dq, dqd, dqdd, include_gravity=grav, use_damping=damp)   # give accelerations of order one
            acc = own.compute_forward_dynamics(dq, dqd, f, include_gravity=grav, use_damping=damp)
            ref = orc.forward_dynamics(q.astype(np.float64), qd.astype(np.float64), f.cpu().numpy().astype(np.float64), grav, damp, np.float64)
            err = float((np.abs(acc.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max())
            other = loop.compute_forward_dynamics(dq, dqd, f, include_gravity=grav, use_damping=damp)
            err_loop = float((np.abs(other.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max())
            assert err < max(1e-3, 2.0 * err_loop), (B, grav, damp, err, err_loop)
        nle = own.compute_non_linear_effects(dq, dqd)      # qdd = NULL
        assert np.allclose(nle.cpu().numpy(), orc.rnea(q.astype(np.float64), qd.astype(np.float64), np.zeros_like(q, np.float64),
                                                       True, True, np.float64), **TOL_TAU)
    # the C ABI itself with misaligned row slices (the special kernel takes any alignment; no scratch for full tiles)
    import ctypes
    from differentiable_robot_model_amd import backend
    lib = backend.load_library()
    dw = own._dynamics_walk()
    walk = backend._walk_struct(dw.program, own._ops_f(dw), dw.ops_i, n)
    assert walk.special[0] and lib.drm_rnea_scratch_floats(ctypes.byref(walk), ctypes.c_int64(128)) == 0
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, 129, seed=9))
    out = torch.zeros(129, n, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert q[1:].data_ptr() & 15
    assert lib.drm_rnea(ctypes.byref(walk), q[1:].data_ptr(), qd[1:].data_ptr(), qdd[1:].data_ptr(), 128, 3, out[1:].data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    ref = loop.compute_inverse_dynamics(q[1:].clone(), qd[1:].clone(), qdd[1:].clone())
    assert np.allclose(out[1:].cpu().numpy(), ref.cpu().numpy(), **TOL_TAU)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("seed", __import__("test_random_trees").SEEDS)
def test_gpu_random_trees_through_their_own_kernels(tmp_path, seed):
    mc, m = tree_model(tmp_path, seed), tree_model(tmp_path, seed, "cuda")
    try:
        took = m.specialize()
    except sp.SpecializeError as err:       # (a tree of more ops than the straight-line form is built for)
        pytest.skip(str(err))
    if not took:
        pytest.skip("this tree already has a compiled straight-line shape")
    B = 192 + seed
    q, qd, qdd = sample_states(mc, B, seed=seed)
    tau = m.compute_inverse_dynamics(*(torch.from_numpy(a).cuda() for a in (q, qd, qdd)), include_gravity=True, use_damping=True)
    ref = Oracle(mc._spec).rnea(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), True, True, np.float64)
    assert np.allclose(tau.cpu().numpy(), ref, **TOL_TAU), seed
    H = m.compute_lagrangian_inertia_matrix(torch.from_numpy(q).cuda())
    assert np.allclose(H.cpu().numpy(), Oracle(mc._spec).mass_matrix(q.astype(np.float64), False, False, np.float64), **TOL_TAU), seed
    acc = m.compute_forward_dynamics(*(torch.from_numpy(a).cuda() for a in (q, qd, qdd)), include_gravity=True, use_damping=True)
    ref = Oracle(mc._spec).forward_dynamics(q.astype(np.float64), qd.astype(np.float64), qdd.astype(np.float64), True, True, np.float64)
    assert float((np.abs(acc.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max()) < 1e-3, seed
    # reverse mode through the tree's own kernel against the loop kernel (input gradients)
    if m._dynamics_walk().program._special.get(sp.SPECIAL_RNEA_BACKWARD):
        loop = library_only(tree_model(tmp_path, seed, "cuda"))
        grads = []
        for mm in (loop, m):
            xs = [torch.from_numpy(a).cuda().requires_grad_(True) for a in (q, qd, qdd)]
            mm.compute_inverse_dynamics(*xs).pow(2).mean().backward()
            grads.append([x.grad for x in xs])
        for a, b in zip(*grads):
            assert float((a - b).abs().max()) <= 2e-4 * max(1e-6, float(b.abs().max())), seed


def _specialized_backward(m):
    assert m.specialize() is True
    dw = m._dynamics_walk()
    assert dw.program._special.get(sp.SPECIAL_RNEA_BACKWARD), "the walk carries its own reverse-mode kernel"


@pytest.mark.gpu
@needs_hipcc
def test_gpu_fetch_rnea_backward_through_its_own_kernel_vs_reference_autograd():
    """Three full tiles of Fetch through drm_rnea_backward_static (every row by the per-robot kernel) against the gradients torch
    autograd produced through the UNMODIFIED reference — input gradients and the gradients of the learnable link parameters
    (tests/golden/golden_tiles_grad_dyn.npz); and the 7-19-row fixture, whose rows all take the ragged-tail path."""
    import test_golden_tiles as tl
    import test_rnea_backward as rbt
    rbt.check_gpu_backward_vs_reference_autograd(tl.tiles("grad_dyn"), "fetch", prepare=_specialized_backward)
    rbt.check_gpu_backward_vs_reference_autograd(rbt.load_golden_dyn(), "fetch", prepare=_specialized_backward)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("compat", [True, False])
def test_gpu_fetch_rnea_backward_own_kernel_vs_loop_kernel(compat):
    """Ragged batches, several learnable links (one of them sliding when the URDF's joint types are honoured), with and without
    qdd / input gradients: the per-robot kernel against the loop kernel on the same walk."""
    import test_rnea_backward as rbt
    from differentiable_robot_model_amd import backend
    mc = load_model("fetch", reference_compat=compat)
    loop, own = library_only(load_model("fetch", "cuda", reference_compat=compat)), load_model("fetch", "cuda", reference_compat=compat)
    for m in (loop, own):
        for link, pname in (("shoulder_lift_link", "mass"), ("shoulder_lift_link", "com"), ("torso_lift_link", "trans"),
                            ("r_gripper_finger_link", "inertia_mat"), ("wrist_roll_link", "rot_angles"), ("head_tilt_link", "joint_damping")):
            m.make_link_param_learnable(link, pname, rbt.parametrization(pname))
    own.load_state_dict(loop.state_dict())
    _specialized_backward(own)
    for B in (64, 200, 2048 + 5):
        q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, B, seed=B))
        want = torch.randn(B, own._n_dofs, device="cuda", generator=torch.Generator("cuda").manual_seed(B))
        grads = []
        for m in (loop, own):
            m.zero_grad()
            xs = [t.clone().requires_grad_(True) for t in (q, qd, qdd)]
            torch.nn.functional.mse_loss(m.compute_inverse_dynamics(*xs), want).backward()
            grads.append([x.grad for x in xs] + [p.grad.clone() for p in m.parameters()])
        for a, b in zip(*grads):
            scale = max(1e-6, float(b.abs().max()))
            assert float((a - b).abs().max()) <= 2e-4 * scale, (B, float((a - b).abs().max()), scale)
    # parameter gradients only (no graph to q), and without qdd (the non-linear effects)
    q, qd, _ = (torch.from_numpy(a).cuda() for a in sample_states(mc, 130, seed=1))
    outs = []
    for m in (loop, own):
        m.zero_grad()
        m.compute_non_linear_effects(q, qd).pow(2).mean().backward()
        outs.append([p.grad.clone() for p in m.parameters()])
    for a, b in zip(*outs):
        assert float((a - b).abs().max()) <= 2e-4 * max(1e-6, float(b.abs().max()))


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("robot", ["panda", "jaco"])
def test_gpu_tuned_choice_between_shape_kernels_and_own_kernels(robot, tmp_path, monkeypatch):
    """specialize(tune=True): an arm that carries a hand builds its own kernels too and keeps, per entry point, whichever is faster
    on this device; whatever was kept, every entry point still meets the fp64 oracle and the untuned model.  (Its own run-time cache:
    the record this machine's measurement leaves must not decide what LATER tests' models attach — a kernel that wins by a few per cent
    in the shipped record may lose here.)"""
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(tmp_path / "runtime"))
    mc, plain, tuned = load_model(robot), library_only(load_model(robot, "cuda")), load_model(robot, "cuda")
    report = tuned.specialize(tune=True)
    assert set(report) == set(sp.KERNELS.values())
    kept = tuned._dynamics_walk().program._special
    for kind, name in sp.KERNELS.items():
        assert report[name]["kept"] == (kind in kept) and report[name]["own_us"] > 0 and report[name]["library_us"] > 0
        if report[name]["kept"]:
            assert report[name]["own_us"] < report[name]["library_us"]
    orc = Oracle(mc._spec)
    B = 64 * 5 + 9
    q, qd, qdd = sample_states(mc, B, seed=11)
    dq, dqd, dqdd = (torch.from_numpy(a).cuda() for a in (q, qd, qdd))
    f64 = lambda a: a.astype(np.float64)
    assert np.allclose(tuned.compute_inverse_dynamics(dq, dqd, dqdd).cpu().numpy(), orc.rnea(f64(q), f64(qd), f64(qdd), True, True, np.float64), **TOL_TAU)
    assert np.allclose(tuned.compute_lagrangian_inertia_matrix(dq).cpu().numpy(), orc.mass_matrix(f64(q), False, False, np.float64), **TOL_TAU)
    ref = orc.forward_dynamics(f64(q), f64(qd), f64(qdd), True, True, np.float64)
    assert float((np.abs(tuned.compute_forward_dynamics(dq, dqd, dqdd).cpu().numpy() - ref) / (1.0 + np.abs(ref))).max()) < 1e-3
    grads = []
    for m in (plain, tuned):
        xs = [t.clone().requires_grad_(True) for t in (dq, dqd, dqdd)]
        m.compute_inverse_dynamics(*xs).pow(2).mean().backward()
        grads.append([x.grad for x in xs])
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 2e-4 * max(1e-6, float(b.abs().max()))
    assert load_model("panda_no_gripper", "cuda").specialize(tune=True) == {}       # (plain 7-DoF arms keep their kernels)


@contextlib.contextmanager
def resident_blocks_clamped(n):
    """DRM_MAX_RESIDENT_BLOCKS (csrc/drm_tree_dev.hpp clamp_resident): the grid of every persistent kernel held to n blocks, so a
    small batch walks several tiles per wavefront."""
    before = os.environ.get("DRM_MAX_RESIDENT_BLOCKS")
    os.environ["DRM_MAX_RESIDENT_BLOCKS"] = str(n)
    try:
        yield
    finally:
        if before is None:
            os.environ.pop("DRM_MAX_RESIDENT_BLOCKS", None)
        else:
            os.environ["DRM_MAX_RESIDENT_BLOCKS"] = before


def _fetch_pair(compat, learnable=False):
    import test_rnea_backward as rbt
    mc = load_model("fetch", reference_compat=compat)
    loop, own = library_only(load_model("fetch", "cuda", reference_compat=compat)), load_model("fetch", "cuda", reference_compat=compat)
    if learnable:
        for m in (loop, own):
            for link, pname in (("shoulder_lift_link", "mass"), ("torso_lift_link", "trans"), ("wrist_roll_link", "rot_angles")):
                m.make_link_param_learnable(link, pname, rbt.parametrization(pname))
        own.load_state_dict(loop.state_dict())
        _specialized_backward(own)
    else:
        assert own.specialize() is True
    return mc, loop, own


def _own_vs_loop_every_entry_point(mc, loop, own, B, seed, oracle_rows=512):
    """Inverse dynamics (with and without qdd), forward dynamics and the input + parameter gradients of the per-robot kernels against
    the loop kernels on EVERY row; the first and the last `oracle_rows` rows also against the fp64 oracle (models without learnable links)."""
    q, qd, qdd = sample_states(mc, B, seed=seed)
    dq, dqd, dqdd = (torch.from_numpy(a).cuda() for a in (q, qd, qdd))
    orc = Oracle(mc._spec)
    f64 = lambda a: a.astype(np.float64)
    ends = np.r_[0:oracle_rows, B - oracle_rows:B]
    tau = own.compute_inverse_dynamics(dq, dqd, dqdd)
    ref = loop.compute_inverse_dynamics(dq, dqd, dqdd)
    tau, ref = tau.detach(), ref.detach()
    assert np.allclose(tau.cpu().numpy(), ref.cpu().numpy(), **TOL_TAU), B
    if not list(own.parameters()):      # (learnable links carry freshly drawn parameters, not the URDF's the oracle was built from)
        assert np.allclose(tau.cpu().numpy()[ends], orc.rnea(f64(q[ends]), f64(qd[ends]), f64(qdd[ends]), True, True, np.float64), **TOL_TAU)
    nle = own.compute_non_linear_effects(dq, dqd)
    assert np.allclose(nle.detach().cpu().numpy(), loop.compute_non_linear_effects(dq, dqd).detach().cpu().numpy(), **TOL_TAU), B
    acc = own.compute_forward_dynamics(dq, dqd, ref, include_gravity=True, use_damping=True).detach()
    other = loop.compute_forward_dynamics(dq, dqd, ref, include_gravity=True, use_damping=True).detach()
    err = float(((acc - other).abs() / (1.0 + other.abs())).max())
    err_loop = float(((other - dqdd).abs() / (1.0 + dqdd.abs())).max())          # (forward dynamics undoes inverse dynamics)
    assert err <= max(1e-3, 2.0 * err_loop), (B, err, err_loop)
    want = torch.randn(B, own._n_dofs, device="cuda", generator=torch.Generator("cuda").manual_seed(seed))
    grads = []
    for m in (loop, own):
        m.zero_grad()
        xs = [t.clone().requires_grad_(True) for t in (dq, dqd, dqdd)]
        torch.nn.functional.mse_loss(m.compute_inverse_dynamics(*xs), want).backward()
        grads.append([x.grad for x in xs] + [p.grad.clone() for p in m.parameters()])
    assert len(grads[0]) == len(grads[1])
    for a, b in zip(*grads):
        scale = max(1e-6, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 3e-4 * scale, (B, float((a - b).abs().max()), scale)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("learnable", [False, True])
def test_gpu_own_kernels_walk_several_tiles_per_wavefront(learnable):
    """the persistent loops of the per-robot kernels — the next tile's rows staged in LDS by global_load_lds_dword
    (rows_to_stage / rows_from_stage), the partial parameter sums a wavefront carries across its tiles — only run when a launch
    has more tiles than resident wavefronts (B above ~65 k rows).  Here the grid is clamped to 3 blocks, so 11 tiles + a ragged
    tail make every wavefront walk 3-4 tiles: rnea, non-linear effects, forward dynamics and the reverse mode of Fetch against
    the loop kernels on every row."""
    mc, loop, own = _fetch_pair(False, learnable)
    with resident_blocks_clamped(3):
        _own_vs_loop_every_entry_point(mc, loop, own, 11 * 64 + 29, seed=5, oracle_rows=128)
    with resident_blocks_clamped(1):       # one wavefront walks every tile
        _own_vs_loop_every_entry_point(mc, loop, own, 5 * 64, seed=6, oracle_rows=64)


@pytest.mark.gpu
@needs_hipcc
def test_gpu_own_kernels_at_a_batch_beyond_the_resident_grid():
    """The same at the size where it happens by itself: 2^18 + 77 rows of Fetch (4 097 tiles against at most 2 048 resident
    wavefronts), no clamp."""
    mc, loop, own = _fetch_pair(True, learnable=True)
    _own_vs_loop_every_entry_point(mc, loop, own, (1 << 18) + 77, seed=7)


# ------------------------------------------------------------------------------------------------ serial 7-DoF arms: constants folded
def test_arm_source_writes_the_table_as_exact_literals():
    """(CPU) The generated translation unit of an arm: every float of the walk table as a hex-float literal that reads back
    bit for bit, exact zeros as the plain 0.0f the compiler folds; rows >= links from the target chain's table."""
    import re
    m = load_model("panda_no_gripper")
    dw = m._dynamics_walk()
    assert sp.arm_qualifies(dw.program, m._n_dofs) and dw.program.n_ops == 7
    table = m._ops_f(dw).detach().numpy()
    src = sp.arm_source(table, 7, False)
    assert "drm_rnea_arm_static" in src and "arm2_stream_body<8, NJ, 7, false, false>" in src
    body = src[src.index("ROBOT_OPS[8 * DRM_OPF_STRIDE] = {") + 33:src.index("};")]
    lits = [t.strip() for t in body.replace("\n", " ").split(",") if t.strip()]
    assert len(lits) == 256
    back = np.array([float.fromhex(t[:-1]) if t != "0.0f" else 0.0 for t in lits], np.float32).reshape(8, 32)
    assert np.array_equal(back.view(np.uint32) & 0x7fffffff, table.view(np.uint32) & 0x7fffffff) and np.array_equal(back, table)
    assert (back == 0).sum() > 100                       # (Panda: most of F, half of t and of the inertia rows are exact zeros)
    cw = m._get_walk(("chain", 8, "folded", dw.fold_key), targets=[8], folded=True, fold_key=dw.fold_key)
    fused = sp.arm_source(np.concatenate([table[:7], m._ops_f(cw).detach().numpy()[7:]]), 7, True)
    assert "drm_fk_rnea_arm_static" in fused and "float *pos, float *quat" in fused and "<8, NJ, 7, true, false>" in fused
    with pytest.raises(sp.SpecializeError):
        sp.arm_source(table[:7], 7, False)


@needs_hipcc
def test_arm_source_builds(tmp_path, monkeypatch):
    """(CPU) hipcc turns the arm's translation unit into a code object with both kernels' symbols; a machine without hipcc finds
    the same robot's code object in an exported cache by its source key."""
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(tmp_path))
    monkeypatch.setenv("DRM_SHIPPED_CACHE", str(tmp_path / "nothing_shipped"))
    m = load_model("iiwa7")
    dw = m._dynamics_walk()
    src = sp.arm_source(m._ops_f(dw).detach().numpy(), dw.program.n_ops, False)
    path = sp.build(src, sp.ARM_FLAGS)
    blob = open(path, "rb").read()
    assert b"drm_rnea_arm_static" in blob and os.path.getsize(path) > 10000
    shipped = tmp_path / "shipped"
    assert sp.export_cache(str(shipped)) == [os.path.basename(path)]
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(shipped))
    monkeypatch.setattr(sp, "hipcc", lambda: None)
    assert sp.build(src, sp.ARM_FLAGS) == str(shipped / os.path.basename(path))
    with pytest.raises(sp.SpecializeError, match="export_cache"):
        sp.build(src + " ", sp.ARM_FLAGS)


def _build_in_child(args):
    cache, src = args
    os.environ["DRM_SPECIAL_CACHE"] = cache
    os.environ["DRM_SHIPPED_CACHE"] = os.path.join(cache, "nothing_shipped")
    from differentiable_robot_model_amd import specialize as child_sp
    path = child_sp.build(src, child_sp.ARM_FLAGS)
    return path, os.path.getsize(path)


@needs_hipcc
def test_ranks_of_a_node_build_the_same_robot_at_once(tmp_path):
    """(CPU) Four processes build the same translation unit into one cache at the same time — the ranks of a node under
    DRM_SPECIALIZE=1: every one returns the same complete code object (per-process temporary names, atomic publish), and nothing but
    that code object is left in the cache."""
    import multiprocessing as mp
    m = load_model("iiwa7")
    dw = m._dynamics_walk()
    src = sp.arm_source(m._ops_f(dw).detach().numpy(), dw.program.n_ops, False)
    with mp.get_context("spawn").Pool(4) as pool:
        got = pool.map(_build_in_child, [(str(tmp_path), src)] * 4)
    # (ONE published name; every process saw a complete code object — they may differ by a few bytes of embedded source path)
    assert len({p for p, _ in got}) == 1 and all(n > 10000 for _, n in got)
    assert sorted(os.listdir(str(tmp_path))) == [os.path.basename(got[0][0])]


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("robot,link", [("panda_no_gripper", "panda_virtual_ee_link"), ("iiwa7", "iiwa_link_ee")])
def test_gpu_arm_kernels_with_folded_constants_every_row_vs_oracle(robot, link):
    """model.specialize() on a serial 7-DoF arm: inverse dynamics and the fused FK + RNEA launch through the robot's own
    constant-folded streaming kernels (csrc/drm_arm_stream.hpp, launches of >= 1 024 pairs of 64-row tiles).  B = 2 051 pairs + one
    odd tile + 21 rows: the pairs through the robot's kernel (two waves walk a second tile: the staged-rows path), the odd tile
    and the tail through the library's — EVERY row against the fp64 oracle; against the library's kernels (unspecialised model)
    to a few ulp; plans under a hipGraph; without qdd; a learnable parameter drops the kernels."""
    mc, plain, own = load_model(robot), library_only(load_model(robot, "cuda")), load_model(robot, "cuda")
    assert own.specialize() is True
    tree = own._dynamics_walk().program
    assert tree._special.get(sp.SPECIAL_RNEA_ARM)
    B, n = (2 * 1024 + 3) * 128 + 64 + 21, 7
    q, qd, qdd = sample_states(mc, B, seed=77, vel=0.6, acc=1.2)
    dq, dqd, dqdd = (torch.from_numpy(a).cuda() for a in (q, qd, qdd))
    orc = Oracle(mc._spec)
    f64 = lambda a: a.astype(np.float64)
    ee = mc._name_to_idx_map[link]
    rp, rq = orc.fk(f64(q), [ee], np.float64)
    from helpers import TOL_POS, TOL_QUAT, max_err, quat_close
    for grav, damp in ((True, True), (False, False)):
        ref = orc.rnea(f64(q), f64(qd), f64(qdd), grav, damp, np.float64)
        tau = own.compute_inverse_dynamics(dq, dqd, dqdd, include_gravity=grav, use_damping=damp)
        assert np.allclose(tau.cpu().numpy(), ref, **TOL_TAU), (grav, damp)
        lib = plain.compute_inverse_dynamics(dq, dqd, dqdd, include_gravity=grav, use_damping=damp)
        assert float(((tau - lib).abs() / lib.abs().clamp_min(1.0)).max()) <= 2e-5
        t2, pos, quat = own.compute_fk_and_inverse_dynamics(dq, dqd, dqdd, link, grav, damp)
        assert tree._special.get(sp.SPECIAL_FK_RNEA_ARM)          # (built on the first fused call for this link)
        assert np.allclose(t2.cpu().numpy(), ref, **TOL_TAU)
        assert max_err(pos.cpu().numpy(), rp[:, 0]) <= TOL_POS["atol"] and quat_close(quat.cpu().numpy(), rq[:, 0], TOL_QUAT["atol"])[0]
        assert float((t2 - tau).abs().max()) <= 2e-5 * float(tau.abs().max())
    nle = own.compute_non_linear_effects(dq, dqd)
    assert np.allclose(nle.cpu().numpy(), orc.rnea(f64(q), f64(qd), np.zeros_like(q, np.float64), True, True, np.float64), **TOL_TAU)
    # prepared launches replayed from a hipGraph; 8 shards of 131 072 rows against one launch of 2^20 (bench.py --verify-gather)
    Bg = 1 << 20
    dq, dqd, dqdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, Bg, seed=78, vel=0.4, acc=0.8))
    plan = own.plan_fk_and_inverse_dynamics(dq, dqd, dqdd, link)
    plan.launch()
    torch.cuda.synchronize()
    first = [t.clone() for t in plan.outputs()]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.launch()
    for t in plan.outputs():
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(first, plan.outputs()):
        assert torch.equal(a, b)
    for r in range(8):
        sl = slice(r << 17, (r + 1) << 17)
        shard = own.plan_fk_and_inverse_dynamics(dq[sl].contiguous(), dqd[sl].contiguous(), dqdd[sl].contiguous(), link)
        shard.launch()
        torch.cuda.synchronize()
        for a, b in zip(shard.outputs(), first):
            assert torch.equal(a, b[sl].reshape(a.shape))          # (one kernel at both sizes: bit for bit)
    sel = slice(0, 1 << 15)
    assert np.allclose(first[0][sel].cpu().numpy(), orc.rnea(*(f64(t[sel].cpu().numpy()) for t in (dq, dqd, dqdd)), True, True, np.float64), **TOL_TAU)
    # a learnable parameter: the constant-folded kernels no longer describe the model
    from differentiable_robot_model_amd.rigid_body_params import PositiveScalar
    own.make_link_param_learnable(mc.get_link_names()[3], "mass", PositiveScalar())
    assert not any(k >= 4 for dw in own._walks.values() for k in (getattr(dw.program, "_special", None) or {}))
    t3 = own.compute_inverse_dynamics(dq[:262144], dqd[:262144], dqdd[:262144])
    assert not torch.allclose(t3, first[0][:262144])            # (another mass: other torques — through the library's kernels)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7"])
def test_gpu_arm_inertia_matrix_forward_dynamics_and_input_gradients_with_folded_constants(robot):
    """model.specialize() on a serial 7-DoF arm also builds the constant-folded inertia-matrix, forward-dynamics and reverse-mode
    (input gradients) kernels (csrc/drm_arm_static.hpp): every row against the fp64 oracle and against the library's kernels, full
    tiles + a ragged tail; gradients against the unspecialised model's (whose kernels meet the reference's autograd elsewhere)."""
    mc, plain, own = load_model(robot), library_only(load_model(robot, "cuda")), load_model(robot, "cuda")
    assert own.specialize() is True
    special = own._dynamics_walk().program._special
    assert all(special.get(k) for k in sp.ARM_DYNAMICS)
    orc = Oracle(mc._spec)
    f64 = lambda a: a.astype(np.float64)
    for B in (64, 64 * 37 + 11, 70000):
        q, qd, qdd = sample_states(mc, B, seed=B)
        dq, dqd, dqdd = (torch.from_numpy(a).cuda() for a in (q, qd, qdd))
        H = own.compute_lagrangian_inertia_matrix(dq)
        assert np.allclose(H.cpu().numpy(), orc.mass_matrix(f64(q), False, False, np.float64), **TOL_TAU), B
        Hl = plain.compute_lagrangian_inertia_matrix(dq)
        assert float((H - Hl).abs().max()) <= 2e-5 * float(Hl.abs().max())
        for grav, damp in ((True, True), (False, False)):
            f = plain.compute_inverse_dynamics(dq, dqd, dqdd, include_gravity=grav, use_damping=damp)
            acc = own.compute_forward_dynamics(dq, dqd, f, include_gravity=grav, use_damping=damp)
            ref = orc.forward_dynamics(f64(q), f64(qd), f64(f.cpu().numpy()), grav, damp, np.float64)
            err = float((np.abs(acc.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max())
            other = plain.compute_forward_dynamics(dq, dqd, f, include_gravity=grav, use_damping=damp)
            err_lib = float((np.abs(other.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max())
            assert err < max(1e-3, 2.0 * err_lib), (B, grav, damp, err, err_lib)
        want = torch.randn(B, 7, device="cuda", generator=torch.Generator("cuda").manual_seed(B))
        grads = []
        for m in (plain, own):
            xs = [t.clone().requires_grad_(True) for t in (dq, dqd, dqdd)]
            torch.nn.functional.mse_loss(m.compute_inverse_dynamics(*xs), want).backward()
            grads.append([x.grad for x in xs])
        for a, b in zip(*grads):
            scale = max(1e-9, float(b.abs().max()))
            assert float((a - b).abs().max()) <= 2e-4 * scale, (B, float((a - b).abs().max()), scale)
        # without qdd: the non-linear effects' gradients
        grads = []
        for m in (plain, own):
            xs = [t.clone().requires_grad_(True) for t in (dq, dqd)]
            m.compute_non_linear_effects(*xs).pow(2).mean().backward()
            grads.append([x.grad for x in xs])
        for a, b in zip(*grads):
            assert float((a - b).abs().max()) <= 2e-4 * max(1e-9, float(b.abs().max()))
    # round 6: launches of >= 1 024 pairs of tiles run forward dynamics TWO samples per lane (drm_fd_arm2_static): 1 027 pairs + an
    # odd tile + a ragged tail — the pairs through that kernel, the odd tile through the one-sample own kernel, the tail through the
    # library's — every row against the fp64 oracle and the library's kernel
    assert special.get(sp.SPECIAL_FD_ARM2)
    B = 1027 * 128 + 64 + 9
    q, qd, qdd = sample_states(mc, B, seed=41)
    dq, dqd, dqdd = (torch.from_numpy(a).cuda() for a in (q, qd, qdd))
    for grav, damp in ((True, True), (False, False)):
        f = plain.compute_inverse_dynamics(dq, dqd, dqdd, include_gravity=grav, use_damping=damp)
        acc = own.compute_forward_dynamics(dq, dqd, f, include_gravity=grav, use_damping=damp)
        ref = orc.forward_dynamics(f64(q), f64(qd), f64(f.cpu().numpy()), grav, damp, np.float64)
        err = float((np.abs(acc.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max())
        other = plain.compute_forward_dynamics(dq, dqd, f, include_gravity=grav, use_damping=damp)
        err_lib = float((np.abs(other.cpu().numpy() - ref) / (1.0 + np.abs(ref))).max())
        assert err < max(1e-3, 2.0 * err_lib), (B, grav, damp, err, err_lib)
        assert float(((acc - other).abs() / (1.0 + other.abs())).max()) < 2e-3
    # ... and the input gradients of inverse dynamics (drm_rnea_backward_arm2_static), with and without qdd, both flag settings
    assert special.get(sp.SPECIAL_RNEA_BACKWARD_ARM2)
    want = torch.randn(B, 7, device="cuda", generator=torch.Generator("cuda").manual_seed(7))
    for grav, damp, with_qdd in ((True, True, True), (False, False, True), (True, True, False)):
        grads = []
        for m in (plain, own):
            xs = [t.clone().requires_grad_(True) for t in ((dq, dqd, dqdd) if with_qdd else (dq, dqd))]
            tau = m.compute_inverse_dynamics(*xs, include_gravity=grav, use_damping=damp) if with_qdd else \
                m.compute_non_linear_effects(*xs, include_gravity=grav, use_damping=damp)
            (tau * want).sum().backward()
            grads.append([x.grad for x in xs])
        for a, b in zip(*grads):
            scale = max(1e-9, float(b.abs().max()))
            assert float((a - b).abs().max()) <= 2e-4 * scale, (grav, damp, with_qdd, float((a - b).abs().max()), scale)


def test_fan_source_describes_the_chains():
    """(CPU) The fan-out FK call of the Allegro's four fingertips as a generated translation unit: per chain its ops' DoF columns,
    the 16-byte load of four consecutive columns, the table rows as exact literals."""
    h = load_model("allegro_left")
    tips = [h._name_to_idx_map[t] for t in ("link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip")]
    merged = h._get_walk(("fk", tuple(tips)), targets=tips)
    fan = h._fanout_chains(tips, merged)
    chains = [sp.fan_chain(w.program, h._n_dofs) for w in fan]
    assert sorted(c["dof"] for c in chains) == [[0, 1, 2, 3, -1], [4, 5, 6, 7, -1], [8, 9, 10, 11, -1], [12, 13, 14, 15, -1]]
    assert all(c["q4"] and c["moving"] == 4 and c["used"] == 5 and c["perm"] == 2 for c in chains)
    src = sp.fan_source(chains, [h._ops_f(w).detach().numpy() for w in fan], h._n_dofs)
    assert src.count("struct Chain") == 4 and "__launch_bounds__(256) drm_fk_fan_links_static" in src
    assert "case 3: drm::fk_fan_links_static_wave<drm::Chain3>(q, pos, quat, B, 3, st + 3 * 3 * drm::WAVE)" in src
    m = load_model("panda_no_gripper")                    # (an arm's chain qualifies as a chain; one chain is not a fan)
    assert sp.fan_chain(m._chain_walk(8).program, 7)["dof"] == [0, 1, 2, 3, 4, 5, 6, -1]
    with pytest.raises(sp.SpecializeError):
        sp.fan_source(chains[:1], [h._ops_f(fan[0]).detach().numpy()], h._n_dofs)


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("robot,links", [("allegro_left", ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]),
                                         ("trifinger_edu", ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"]),
                                         ("iiwa7_allegro", ["link_3.0_tip", "link_15.0_tip"])])
def test_gpu_fan_out_fk_with_folded_constants(robot, links):
    """BASELINE configuration 4's call (compute_forward_kinematics_links of a hand's fingertips) through the hand's own fan-out
    kernel: every row against the fp64 oracle and against the library's kernel; full tiles, a ragged tail, a batch that is not a
    multiple of 4 (the library's path)."""
    from helpers import TOL_POS, TOL_QUAT, max_err, quat_close
    mc, plain, own = load_model(robot), library_only(load_model(robot, "cuda")), load_model(robot, "cuda")
    own.specialize()
    orc = Oracle(mc._spec)
    idx = [mc._name_to_idx_map[n] for n in links]
    for B in (64, 64 * 9 + 20, 65536, 1001):
        q = sample_states(mc, B, seed=B)[0]
        dq = torch.from_numpy(q).cuda()
        got, lib = own.compute_forward_kinematics_links(dq, links), plain.compute_forward_kinematics_links(dq, links)
        rp, rq = orc.fk(q.astype(np.float64), idx, np.float64)
        for t, name in enumerate(links):
            assert max_err(got[name][0].cpu().numpy(), rp[:, t]) <= TOL_POS["atol"], (B, name)
            assert quat_close(got[name][1].cpu().numpy(), rq[:, t], TOL_QUAT["atol"])[0], (B, name)
            assert float((got[name][0] - lib[name][0]).abs().max()) <= 1e-6
    order = sorted(idx)          # (_fk_links launches the targets in walk order)
    fan = own._fanout_chains(order, own._get_walk(("fk", tuple(order)), targets=order))
    if fan is not None and all(sp.fan_chain(w.program, own._n_dofs) for w in fan):
        assert own._fan_own(order)          # ONE kernel, kept with the plan of this ordered set of targets ...
        assert not any((getattr(w.program, "_special", None) or {}).get(sp.SPECIAL_FK_FAN_LINKS) for w in fan)   # ... not on shared walks
        assert plain._fan_own(order) is None


@pytest.mark.gpu
@needs_hipcc
def test_gpu_fan_out_kernel_belongs_to_its_ordered_set_of_targets():
    """the chain walk of a fingertip is shared by every set of targets that names it, the constant-folded fan-out
    kernel is not — four tips, then non-prefix subsets and back, on ONE specialised model: every call against the library's kernel
    and the fp64 oracle."""
    from helpers import TOL_POS, max_err
    tips = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]
    mc, plain, own = load_model("allegro_left"), library_only(load_model("allegro_left", "cuda")), load_model("allegro_left", "cuda")
    own.specialize()
    orc = Oracle(mc._spec)
    q = sample_states(mc, 4096, seed=4)[0]
    dq = torch.from_numpy(q).cuda()
    seen = set()
    for names in (tips, [tips[1], tips[2]], [tips[2], tips[3]], [tips[1], tips[3]], [tips[0], tips[1]], tips, [tips[3], tips[1]], tips[1:]):
        got, lib = own.compute_forward_kinematics_links(dq, names), plain.compute_forward_kinematics_links(dq, names)
        idx = [mc._name_to_idx_map[n] for n in names]
        rp, _ = orc.fk(q.astype(np.float64), idx, np.float64)
        for t, name in enumerate(names):
            assert max_err(got[name][0].cpu().numpy(), rp[:, t]) <= TOL_POS["atol"], (names, name)
            assert float((got[name][0] - lib[name][0]).abs().max()) <= 1e-6 and float((got[name][1] - lib[name][1]).abs().max()) <= 1e-5
        order = sorted(idx)
        assert own._fan_own(order), names          # (every one of these sets has a kernel of its own)
        seen.add(own._fan_own(order))
    assert len(seen) == 6          # six distinct ordered sets -> six kernels ([3, 1] and [1, 3] launch in walk order: the same set)


@pytest.mark.gpu
@needs_hipcc
def test_gpu_models_attach_built_kernels_by_default_and_only_those(tmp_path, monkeypatch, caplog):
    """Round 6: the drop-in path is the fast path.  `DifferentiableRobotModel(urdf, device="cuda")` and nothing else runs the robot's
    OWN kernels whenever their code objects are already built (shipped next to the library / the run-time cache) and never compiles
    on a call path; with nothing built it keeps the library's kernels and says so once; own_kernels = "off" / DRM_SPECIALIZE=0 switch
    it off.  Same numbers to a few ulp either way."""
    import logging
    link = "panda_virtual_ee_link"
    shipped = tmp_path / "shipped"
    monkeypatch.setenv("DRM_SHIPPED_CACHE", str(shipped))
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(tmp_path / "runtime"))
    monkeypatch.delenv("DRM_SPECIALIZE", raising=False)
    mc = load_model("panda_no_gripper")
    B = 1024 * 128 + 64
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, B, seed=5, vel=0.5, acc=1.0))
    special = lambda m: {k: v for k, v in (getattr(m._dynamics_walk().program, "_special", None) or {}).items() if v}
    # (1) nothing built anywhere: the library's kernels, one INFO line per kind of kernel, no compiler run (the caches stay empty)
    cold = load_model("panda_no_gripper", "cuda")
    with caplog.at_level(logging.INFO, logger="differentiable_robot_model_amd"):
        tau_lib = cold.compute_inverse_dynamics(q, qd, qdd)
        cold.compute_inverse_dynamics(q, qd, qdd)
        fused_lib = cold.compute_fk_and_inverse_dynamics(q, qd, qdd, link)
    assert special(cold) == {}
    assert not (tmp_path / "runtime").exists() or os.listdir(str(tmp_path / "runtime")) == []
    said = [r.getMessage() for r in caplog.records if "no pre-built own kernel" in r.getMessage()]
    assert 1 <= len(said) <= 3 and len(set(said)) == len(said), said
    # (2) the package's build step ships them: a NEW plain model picks them up by itself
    assert len(sp.prebuild_shipped(("panda_no_gripper",))) == 3 + 8          # (+ the reverse-mode kernels of 8 sets of learnable links)
    monkeypatch.setattr(sp, "hipcc", lambda: None)          # (from here on a compile would raise: nothing below needs one)
    warm = load_model("panda_no_gripper", "cuda")
    tau_own = warm.compute_inverse_dynamics(q, qd, qdd)
    assert set(special(warm)) >= {sp.SPECIAL_RNEA_ARM, sp.SPECIAL_CRBA_ARM, sp.SPECIAL_FD_ARM, sp.SPECIAL_RNEA_BACKWARD_ARM}
    fused_own = warm.compute_fk_and_inverse_dynamics(q, qd, qdd, link)
    assert special(warm).get(sp.SPECIAL_FK_RNEA_ARM)
    assert float(((tau_own - tau_lib).abs() / tau_lib.abs().clamp_min(1.0)).max()) <= 2e-5 and not torch.equal(tau_own, tau_lib)
    for a, b in zip(fused_own, fused_lib):
        assert float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) <= 2e-5
    # (3) the off switches: per model, and DRM_SPECIALIZE=0
    warm.own_kernels = "off"
    assert torch.equal(warm.compute_inverse_dynamics(q, qd, qdd), tau_lib) and special(warm) == {}
    warm.own_kernels = None
    assert torch.equal(warm.compute_inverse_dynamics(q, qd, qdd), tau_own)
    monkeypatch.setenv("DRM_SPECIALIZE", "0")
    off = load_model("panda_no_gripper", "cuda")
    assert torch.equal(off.compute_inverse_dynamics(q, qd, qdd), tau_lib) and special(off) == {}
    # (4) a learnable parameter: constant-folded kernels never attach
    monkeypatch.delenv("DRM_SPECIALIZE")
    from differentiable_robot_model_amd.rigid_body_params import PositiveScalar
    learn = load_model("panda_no_gripper", "cuda")
    learn.make_link_param_learnable("panda_link3", "mass", PositiveScalar())
    learn.compute_inverse_dynamics(q[:4096], qd[:4096], qdd[:4096])
    # ... only the reverse-mode kernel of THIS set of learnable blocks, which reads them from the table (shipped for single links)
    assert set(special(learn)) <= {sp.SPECIAL_RNEA_BACKWARD_ARM_PARAM}


# ------------------------------------------------------------------ arms WITH learnable link parameters (round 6)
def _learnable_iiwa(device, what):
    """iiwa7 with (link, parameter) pairs learnable, seeded; `what` in the table below."""
    from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, UnconstrainedTensor
    torch.manual_seed(3)
    m = load_model("iiwa7", device)
    plans = {
        "all_dynamic": [("iiwa_link_%d" % k, p) for k in range(1, 8) for p in ("mass", "com", "inertia_mat")],
        "one_mass": [("iiwa_link_4", "mass")],
        "mixed": [("iiwa_link_2", "trans"), ("iiwa_link_5", "mass"), ("iiwa_link_5", "com"), ("iiwa_link_6", "rot_angles"), ("iiwa_link_3", "joint_damping")],
    }
    shapes = {"com": (1, 3), "inertia_mat": (3, 3), "trans": (1, 3), "rot_angles": (1, 3), "joint_damping": (1, 1)}
    for link, pname in plans[what]:
        par = PositiveScalar() if pname == "mass" else UnconstrainedTensor(dim1=shapes[pname][0], dim2=shapes[pname][1])
        m.make_link_param_learnable(link, pname, par)
    return m


def test_learnable_arm_source_folds_the_constant_blocks():
    """(CPU) The translation unit of an arm with learnable parameters: the learnable blocks are zeros in the literal table (a
    parameter update does not change the source key), the masks tell kinematic from dynamic blocks, other shapes are refused."""
    m = _learnable_iiwa("cpu", "mixed")
    dw = m._dynamics_walk()
    kin, dyn = m._learnable_block_masks(dw)
    links = [int(x) for x in dw.program.links[:dw.program.n_ops]]
    op = {name: links.index(m._name_to_idx_map[name]) for name in ("iiwa_link_2", "iiwa_link_3", "iiwa_link_5", "iiwa_link_6")}
    assert kin == (1 << op["iiwa_link_2"]) | (1 << op["iiwa_link_6"]) and dyn == (1 << op["iiwa_link_5"]) | (1 << op["iiwa_link_3"])
    assert kin | dyn == m._learnable_op_mask(dw)
    table = m._ops_f(dw).detach().numpy()
    src = sp.arm_param_backward_source(table, dw.program.n_ops, kin, dyn)
    assert "KIN = 0x%xu, DYN = 0x%xu" % (kin, dyn) in src and "drm_rnea_backward_arm_param_static" in src
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.123)
    assert sp.arm_param_backward_source(m._ops_f(dw).detach().numpy(), dw.program.n_ops, kin, dyn) == src     # (same key after a step)
    assert sp.arm_param_backward_source(table, dw.program.n_ops, kin, dyn | 1) != src
    with pytest.raises(sp.SpecializeError):
        sp.arm_param_backward_source(table, dw.program.n_ops, 0, 0)
    with pytest.raises(sp.SpecializeError):
        sp.arm_param_backward_source(table, dw.program.n_ops, 0, 1 << 8)
    assert sp.learnable_arm_source("iiwa7", ["iiwa_link_4"], ("mass", "com", "inertia_mat")).count("static constexpr uint32_t KIN = 0x0u") == 1


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("what", ["all_dynamic", "one_mass", "mixed"])
def test_gpu_learnable_arm_gradients_through_its_own_kernel(what):
    """drm_rnea_backward of an arm with learnable link parameters through the arm's own kernel for that set of learnable blocks
    (constant blocks folded into the instruction stream, live sums per lane; csrc/drm_arm_static.hpp) against the library's
    table-driven kernel: parameter gradients (sums over the batch) and input gradients, with and without input gradients wanted,
    full tiles + a ragged tail, a batch beyond the persistent grid; then after an optimiser step (same kernel, new table)."""
    lib, own = library_only(_learnable_iiwa("cuda", what)), _learnable_iiwa("cuda", what)
    own.specialize()
    dw = own._dynamics_walk()
    assert (dw.program._special or {}).get(sp.SPECIAL_RNEA_BACKWARD_ARM_PARAM) and dw.program._special_mask == own._learnable_op_mask(dw)
    assert not (getattr(lib._dynamics_walk().program, "_special", None) or {})
    mc = load_model("iiwa7")
    for B in (64 * 37 + 11, 1 << 17, 64 * 5000):
        q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, B, seed=B % 1000, vel=0.8, acc=1.5))
        want = torch.randn(B, 7, device="cuda", generator=torch.Generator("cuda").manual_seed(B % 977))
        for with_inputs in (True, False):
            res = []
            for m in (lib, own):
                m.zero_grad()
                xs = [t.clone().requires_grad_(with_inputs) for t in (q, qd, qdd)]
                tau = m.compute_inverse_dynamics(*xs)
                torch.nn.functional.mse_loss(tau, want).backward()
                res.append(([p.grad.clone() for p in m.parameters()], [x.grad for x in xs] if with_inputs else []))
            scale = max(float(b.abs().max()) for b in res[0][0])      # (a base link's mass gradient is 1e-6 of the largest: fp32 noise of the sums)
            for a, b in zip(res[1][0], res[0][0]):
                assert float((a - b).abs().max()) <= 3e-4 * float(b.abs().max()) + 1e-7 * scale, (what, B, with_inputs)
            for a, b in zip(res[1][1], res[0][1]):
                assert float((a - b).abs().max()) <= 2e-4 * max(1e-9, float(b.abs().max())), (what, B)
    # a step of the optimiser changes the table, not the kernel
    handle = dw.program._special[sp.SPECIAL_RNEA_BACKWARD_ARM_PARAM]
    for m in (lib, own):
        with torch.no_grad():
            for i, p in enumerate(m.parameters()):
                p.add_(0.01 * (i + 1))
    res = []
    for m in (lib, own):
        m.zero_grad()
        torch.nn.functional.mse_loss(m.compute_inverse_dynamics(q, qd, qdd), want).backward()
        res.append([p.grad.clone() for p in m.parameters()])
    assert own._dynamics_walk().program._special[sp.SPECIAL_RNEA_BACKWARD_ARM_PARAM] == handle
    scale = max(float(b.abs().max()) for b in res[0])
    for a, b in zip(res[1], res[0]):
        assert float((a - b).abs().max()) <= 3e-4 * float(b.abs().max()) + 1e-7 * scale, what


@pytest.mark.gpu
def test_gpu_learn_dynamics_example_runs_its_shipped_kernel_by_default():
    """examples/learn_dynamics_iiwa.py's model — mass, com, inertia_mat of the seven moving links learnable — picks the shipped
    reverse-mode kernel of that set up by itself (no specialize(), no compiler on the call path)."""
    if not os.path.isdir(sp.SHIPPED_CACHE) or not os.listdir(sp.SHIPPED_CACHE):
        pytest.skip("the package's code objects were not built (python __graft_entry__.py build)")
    m = _learnable_iiwa("cuda", "all_dynamic")
    mc = load_model("iiwa7")
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(mc, 4096, seed=1))
    m.compute_inverse_dynamics(q, qd, qdd).pow(2).mean().backward()
    prog = m._dynamics_walk().program
    assert (prog._special or {}).get(sp.SPECIAL_RNEA_BACKWARD_ARM_PARAM), getattr(m, "_own_kernel_missed", None)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


# ------------------------------------------------------------------ non-finite rows through the robots' own kernels
NON_FINITE = [("q", 3, float("nan")), ("q", 6, float("nan")), ("qd", 2, float("inf")), ("qdd", 5, float("-inf")), ("q", 0, 1e30),
              ("qd", 1, 1e30), ("qdd", 6, float("nan")), ("q", 1, float("inf")), ("q", 2, -3e9)]


def _entry_points(robot):
    """{name: fn(model, q, qd, qdd) -> tuple of [B, ...] tensors} of every entry point a robot's own kernels serve."""
    def grads(m, q, qd, qdd):
        xs = [t.clone().requires_grad_(True) for t in (q, qd, qdd)]
        m.compute_inverse_dynamics(*xs).sum().backward()
        return tuple(x.grad for x in xs)
    if robot == "allegro_left":
        tips = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]

        def fan(m, q, qd, qdd):
            out = m.compute_forward_kinematics_links(q, tips)
            return tuple(out[t][0] for t in tips) + tuple(out[t][1] for t in tips)
        return {"fk_links": fan}
    points = {"inverse_dynamics": lambda m, q, qd, qdd: (m.compute_inverse_dynamics(q, qd, qdd),),
              "inertia_matrix": lambda m, q, qd, qdd: (m.compute_lagrangian_inertia_matrix(q),),
              "forward_dynamics": lambda m, q, qd, qdd: (m.compute_forward_dynamics(q, qd, qdd),),
              "input_gradients": grads}
    if robot != "fetch":
        link = {"panda_no_gripper": "panda_virtual_ee_link", "iiwa7": "iiwa_link_ee"}[robot]
        points["fk_and_inverse_dynamics"] = lambda m, q, qd, qdd: m.compute_fk_and_inverse_dynamics(q, qd, qdd, link)
    return points


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("robot", ["panda_no_gripper", "iiwa7", "allegro_left", "fetch"])
def test_gpu_own_kernels_on_rows_with_non_finite_inputs(robot):
    """The robots' own kernels are built with -ffinite-math-only so that products with the robot's zeros fold away.  What that may
    NOT change: a row with a NaN / +-Inf / out-of-domain (|q| > 1e9) input gives a NON-FINITE value for every output that depends on
    that input — as the reference does (spatial_vector_algebra.py:14-53 propagates NaN), as libdrm_cpu.so (IEEE arithmetic) does.
    Held here, per output element of every poisoned row, against libdrm_cpu.so:
      * own non-finite  =>  the library's is non-finite too (nothing is made up);
      * own finite      =>  the library's value when that is finite too (a huge but finite velocity); else the output does not depend
                            on the poisoned input at all: the library gives the same value (and own's)
                            whatever finite number stands in its place — e.g. the end-effector position of an arm whose last joint
                            turns about the tool axis, the wheel torque of a base whose wheel angle is NaN.  The library's NaN there
                            comes from NaN x 0 of a term the own kernel does not carry.
    Clean rows of the same launch are untouched (every tile holds 64 rows: a poisoned lane must not leak into its neighbours)."""
    mc, own = load_model(robot), load_model(robot, "cuda")
    own.specialize()
    n = mc._n_dofs
    B = 1024 * 128 + 64 * 3 + 7 if robot in ("panda_no_gripper", "iiwa7") else 64 * 40 + 9      # (>= 1 024 tile pairs: the arms' streaming kernels)
    q, qd, qdd = sample_states(mc, B, seed=13, vel=0.5, acc=1.0)
    rows = [5 + 67 * i for i in range(len(NON_FINITE))]
    clean = np.ones(B, bool)
    clean[rows] = False
    for r, (which, col, val) in zip(rows, NON_FINITE):
        {"q": q, "qd": qd, "qdd": qdd}[which][r, col % n] = val
    # the poisoned rows again with two different finite stand-ins (the library, on the host)
    stand = []
    for v in (0.3, -0.7):
        a = [x[rows].copy() for x in (q, qd, qdd)]
        for i, (which, col, _) in enumerate(NON_FINITE):
            a[("q", "qd", "qdd").index(which)][i, col % n] = v
        stand.append([torch.from_numpy(x) for x in a])
    dev = lambda arrs: [torch.from_numpy(x).cuda() for x in arrs]
    host = lambda arrs: [torch.from_numpy(x) for x in arrs]
    for name, fn in _entry_points(robot).items():
        small = name in ("inertia_matrix", "forward_dynamics", "input_gradients") and B > 8192
        cut = (lambda x: x[:4096]) if small else (lambda x: x)
        got = [t.detach().cpu().numpy() for t in fn(own, *[cut(x) for x in dev((q, qd, qdd))])]
        lib = [t.detach().numpy() for t in fn(mc, *[cut(x) for x in host((q, qd, qdd))])]
        s1, s2 = ([t.detach().numpy() for t in fn(mc, *st)] for st in stand)
        for k, (o, l, a, b) in enumerate(zip(got, lib, s1, s2)):
            o, l = o.reshape(o.shape[0], -1), l.reshape(l.shape[0], -1)
            a, b = a.reshape(len(rows), -1), b.reshape(len(rows), -1)
            ok = clean[:o.shape[0]]
            assert np.isfinite(o[ok]).all() and np.allclose(o[ok], l[ok], rtol=2e-4, atol=2e-4), (name, k)
            for i, r in enumerate(rows):
                fin = np.isfinite(o[r])
                assert not np.isfinite(l[r][~fin]).any(), (name, k, NON_FINITE[i], "own kernel non-finite where the library is finite")
                both = fin & np.isfinite(l[r])          # a huge but finite input (1e30 rad/s): finite on both sides, the same number
                big = max(1.0, float(np.abs(l[r][both]).max())) if both.any() else 1.0
                assert (np.abs(o[r][both] - l[r][both]) <= 1e-3 * big).all(), (name, k, NON_FINITE[i])
                alone = fin & ~np.isfinite(l[r])        # finite here, non-finite in the library: must not depend on the poisoned input
                scale = 1.0 + np.abs(a[i][alone])
                assert (np.abs(a[i][alone] - b[i][alone]) <= 2e-4 * scale).all(), (name, k, NON_FINITE[i], "finite output that depends on the poisoned input")
                assert (np.abs(o[r][alone] - a[i][alone]) <= 2e-4 * scale).all(), (name, k, NON_FINITE[i])


def test_sincos_domain_on_the_host(cpu_library):
    """(CPU) sin / cos of a joint angle beyond 1e9 rad (or +-Inf, NaN) are NaN on every path — the quadrant no longer fits an int
    there — by integer tests on the bit pattern (csrc/drm_sample.hpp sincos_f): poses of such rows are NaN, their neighbours' are not."""
    m = load_model("panda_no_gripper")
    q = torch.from_numpy(sample_states(m, 8, seed=1)[0])
    for r, v in enumerate((1e30, -3e9, float("inf"), float("nan"), 9.9e8)):
        q[r, 2] = v
    pos, quat = m.compute_forward_kinematics(q, "panda_virtual_ee_link")
    assert torch.isnan(pos[:4]).all() and torch.isnan(quat[:4]).all()
    assert torch.isfinite(pos[4:]).all() and torch.isfinite(quat[4:]).all()
    ref = load_model("panda_no_gripper").compute_forward_kinematics(torch.remainder(q[4:5].double(), 2 * np.pi).float(), "panda_virtual_ee_link")[0]
    assert float((pos[4:5] - ref).abs().max()) < 2e-3        # (9.9e8 rad: inside the domain; fp32 spacing there is 64 rad ... of the INPUT)


@needs_hipcc
def test_shipped_cache_serves_a_machine_without_hipcc(tmp_path, monkeypatch):
    """(CPU) `__graft_entry__.build()` pre-builds the shipped robots' own kernels next to the library (csrc/special_cache/); `build`
    finds them there — by source key — when neither the run-time cache nor hipcc has them."""
    monkeypatch.setenv("DRM_SHIPPED_CACHE", str(tmp_path / "shipped"))
    names = sp.prebuild_shipped(("iiwa7",))
    assert len(names) == 3 + 8 and all(os.path.exists(os.path.join(sp.shipped_cache(), n)) for n in names)
    m = load_model("iiwa7")
    dw = m._dynamics_walk()
    src = sp.arm_source(m._ops_f(dw).detach().numpy(), dw.program.n_ops, False)
    runtime = tmp_path / "runtime"
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(runtime))           # (an empty run-time cache)
    assert sp.build(src, sp.ARM_FLAGS).startswith(sp.shipped_cache()) and os.listdir(str(runtime)) == []
    monkeypatch.setattr(sp, "hipcc", lambda: None)
    assert sp.build(src, sp.ARM_FLAGS).startswith(sp.shipped_cache())


def test_specialize_needs_a_device_model():
    """(CPU) per-robot kernels are HIP code objects: specialize() needs a model on a HIP device."""
    m = load_model("panda_no_gripper")
    with pytest.raises(RuntimeError):
        m.specialize()


def test_tuning_records_limit_what_a_model_attaches(tmp_path, monkeypatch):
    """(CPU) `specialize.tune` leaves drm_special_<source key>.tuned.json next to the code object it measured; `tuned_kinds` reads the
    run-time cache's record before the shipped one; the records shipped with the package name code objects that `prebuild_shipped`
    builds from the current sources (a stale record would silently switch the robot's own kernels off)."""
    import json
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(tmp_path / "runtime"))
    monkeypatch.setenv("DRM_SHIPPED_CACHE", str(tmp_path / "shipped"))
    os.makedirs(sp.shipped_cache(), exist_ok=True)
    path = os.path.join(sp.cache_dir(), "drm_special_0123456789abcdef0123_deadbeef.hsaco")
    assert sp._tuned_name(path) == "drm_special_0123456789abcdef0123.tuned.json" and sp.tuned_kinds(path) is None
    with open(os.path.join(sp.shipped_cache(), sp._tuned_name(path)), "w") as f:
        json.dump({"kept": [1, 3]}, f)
    assert sp.tuned_kinds(path) == {1, 3}
    with open(os.path.join(sp.cache_dir(), sp._tuned_name(path)), "w") as f:
        json.dump({"kept": []}, f)
    assert sp.tuned_kinds(path) == set()                 # (this machine's own measurement wins)
    monkeypatch.delenv("DRM_SHIPPED_CACHE")
    monkeypatch.delenv("DRM_SPECIAL_CACHE")
    records = [f for f in os.listdir(sp.SHIPPED_CACHE) if f.endswith(".tuned.json")]
    assert len(records) == len(sp.SHIPPED_TREES)
    if sp.hipcc() is not None:       # every shipped record belongs to a code object the current sources generate
        monkeypatch.setenv("DRM_SHIPPED_CACHE", str(tmp_path / "rebuilt"))
        built = sp.prebuild_shipped(robots=(), trees=sp.SHIPPED_TREES)
        assert sorted(sp._tuned_name(b) for b in built) == sorted(records)


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["fetch", "panda", "jaco"])
def test_gpu_shipped_tree_robots_run_their_tuned_entry_points_by_default(robot, tmp_path, monkeypatch):
    """Robots that are not plain arms (round 6): the whole-tree kernels of Fetch, the Panda with its gripper and the Jaco ship with the
    package together with a tuning record measured on an MI355X (tools/tune_shipped.py, profiles/r06_tune_shipped.txt); a plain
    DifferentiableRobotModel(urdf, device="cuda") attaches exactly the entry points the record keeps — and the four entry points agree
    with the library's kernels."""
    import json
    monkeypatch.setenv("DRM_SPECIAL_CACHE", str(tmp_path / "runtime"))     # (an empty run-time cache: the SHIPPED record decides)
    own, plain = load_model(robot, "cuda"), library_only(load_model(robot, "cuda"))
    special = own._dynamics_walk().program._special
    path = own._dynamics_walk().program._special_path
    assert path.startswith(sp.shipped_cache()) or path.startswith(sp.cache_dir())
    with open(os.path.join(sp.SHIPPED_CACHE, sp._tuned_name(path))) as f:
        kept = set(json.load(f)["kept"])
    assert kept and {k for k in special if k in sp.KERNELS} == kept
    assert not (getattr(plain._dynamics_walk().program, "_special", None) or {})
    B = 64 * 9 + 5
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(own, B, seed=3))
    close = lambda a, b: float((a - b).abs().max()) <= 2e-4 * max(1e-6, float(b.abs().max()))
    tau = plain.compute_inverse_dynamics(q, qd, qdd)
    assert close(own.compute_inverse_dynamics(q, qd, qdd), tau)
    assert close(own.compute_lagrangian_inertia_matrix(q), plain.compute_lagrangian_inertia_matrix(q))
    a, b = own.compute_forward_dynamics(q, qd, tau), plain.compute_forward_dynamics(q, qd, tau)
    assert float(((a - b).abs() / (1.0 + b.abs())).max()) < 2e-3        # (the solve amplifies rounding: as the other FD comparisons)
    grads = []
    for m in (own, plain):
        x = q.clone().requires_grad_(True)
        m.compute_inverse_dynamics(x, qd, qdd).pow(2).mean().backward()
        grads.append(x.grad)
    assert close(*grads)
