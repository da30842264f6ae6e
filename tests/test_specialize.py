"""Per-robot straight-line dynamics kernels for arbitrary trees (specialize.py, csrc/drm_static.hpp): the tree of a robot's
whole-tree walk as compile-time constants, compiled at run time (hipcc, cached) and launched through drm_walk.special[] by the
C ABI.  What the reference does with the same Python loop for every robot (robot_model.py:173-193, 262-301).

CPU (not gpu): the tree decoded from the control words, the generated translation unit, and that hipcc builds it.
GPU (-m gpu): inverse dynamics of Fetch (both joint models) and of random trees through their OWN kernels against the fp64 oracle
and against the loop kernels; ragged batches, misaligned slices.
"""
import os

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd import specialize as sp
from differentiable_robot_model_amd.flatten import build_walk
from helpers import TOL_TAU, load_model, sample_states
from oracle import Oracle
from test_random_trees import tree_model

needs_hipcc = pytest.mark.skipif(sp.hipcc() is None, reason="hipcc not on this machine")


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
    path = sp.build(src)
    assert os.path.getsize(path) > 1000 and path.startswith(str(tmp_path))
    stamp = os.path.getmtime(path)
    assert sp.build(src) == path and os.path.getmtime(path) == stamp          # second call: the cache
    assert sp.build(src.replace("N = 14", "N = 14 ")) != path                  # another robot: another code object


@pytest.mark.gpu
@needs_hipcc
@pytest.mark.parametrize("compat", [True, False])
def test_gpu_fetch_inverse_dynamics_through_its_own_kernel(compat):
    mc = load_model("fetch", reference_compat=compat)
    loop, own = load_model("fetch", "cuda", reference_compat=compat), load_model("fetch", "cuda", reference_compat=compat)
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


def test_robots_with_a_compiled_shape_keep_it():
    """(CPU) specialize() is for trees the library has no straight-line kernel for; it needs a device model."""
    m = load_model("panda_no_gripper")
    with pytest.raises(RuntimeError):
        m.specialize()
