"""Host-side logic (not gpu): URDF ingest, constant tables, walk programs, tensor_check, learnable-parameter
plumbing, error behaviour, and the C-ABI library's exported symbols."""
import contextlib
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd import backend
from differentiable_robot_model_amd.flatten import (OPF_F, OPF_IO, OPF_MCOM, OPF_T, OPI_DOF, OPI_OUT, OPI_SAVE,
                                                    opf_fij, opf_ti,
                                                    OPI_SRC, SRC_PREV, SRC_ROOT, UnsupportedRobotError, build_walk)
from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel
from differentiable_robot_model_amd.urdf_utils import parse_urdf
from helpers import ALL_ROBOTS, GOLDEN_ROBOTS, load_golden, load_model, urdf_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ URDF ingest == reference's loader
@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_urdf_parameters_bit_identical_to_reference(robot, links):
    g = load_golden(robot)
    m = load_model(robot)
    assert [b.name for b in m._bodies] == list(g["link_names"])
    assert m._controlled_joints == list(g["controlled_joints"])
    assert list(m._spec.parent) == list(g["parent"])
    s = m._spec
    for mine, ref in ((s.rpy, "rot_angles"), (s.trans, "trans"), (s.axis, "joint_axis"), (s.damping, "joint_damping"),
                      (s.mass, "mass"), (s.com, "com"), (s.inertia, "inertia_mat")):
        assert np.array_equal(mine, g[ref]), ref
    lim = m.get_joint_limits()
    for k in ("lower", "upper", "velocity", "effort"):
        assert [j[k] for j in lim] == list(g["limit_" + k])
    # R_fixed of the constant table == the reference's (z_rot(yaw) @ y_rot(pitch)) @ x_rot(roll), bit for bit
    table = m._link_table().numpy()
    assert np.array_equal(table[:-1, OPF_F:OPF_F + 9], g["R_fixed"])
    assert np.array_equal(table[:-1, OPF_T:OPF_T + 3], g["trans"])


def test_missing_inertial_and_dynamics_defaults():
    m = load_model("2link_robot")     # root link "base" has no <inertial> (urdf_utils.py:114-124)
    assert float(m._bodies[0].inertia.mass()) == 1.0
    assert torch.equal(m._bodies[0].inertia.inertia_mat()[0], torch.eye(3))
    p = load_model("panda_no_gripper")  # joints without <dynamics>: damping 0 (urdf_utils.py:65-72)
    assert all(float(b.joint_damping()) == 0.0 for b in p._bodies[1:])
    assert p._bodies[0].joint_damping() is None


def test_lenient_xml_and_bad_robots(tmp_path):
    bad = tmp_path / "unbound.urdf"
    bad.write_text('<robot name="r"><link name="a"/><gazebo><sensor:camera name="c"/></gazebo></robot>')
    assert parse_urdf(str(bad)).links[0].name == "a"
    skew = tmp_path / "skew.urdf"
    skew.write_text('<robot name="r"><link name="a"/><link name="b"/><joint name="j" type="revolute"><parent link="a"/>'
                    '<child link="b"/><axis xyz="0 0.7071 0.7071"/><limit lower="-1" upper="1" effort="1" velocity="1"/>'
                    '</joint></robot>')
    with pytest.raises(UnsupportedRobotError, match="reference_compat=False"):   # the reference cannot model this axis
        DifferentiableRobotModel(str(skew), device="cpu")                          # (rigid_body.py:149-154): the default refuses ...
    ok = DifferentiableRobotModel(str(skew), device="cpu", reference_compat=False)  # ... the opt-in models it: the joint becomes two ops
    assert ok._spec.skew.tolist() == [False, True] and build_walk(ok._spec, whole_tree=True).n_ops == 2
    floating = tmp_path / "floating.urdf"
    floating.write_text('<robot name="r"><link name="a"/><link name="b"/><joint name="j" type="floating"><parent link="a"/>'
                        '<child link="b"/><limit lower="-1" upper="1" effort="1" velocity="1"/></joint></robot>')
    # upstream counts EVERY joint that is not `fixed` as one revolute DoF about its <axis> (robot_model.py:122-126, rigid_body.py:
    # 130-157), `floating` and `planar` ones included: the default (reference_compat=True) mirrors that — the same numbers as the
    # same URDF with type="revolute" — and reference_compat=False, which models joints as the URDF says, refuses what it cannot model
    with pytest.raises(UnsupportedRobotError, match="reference_compat=False"):
        DifferentiableRobotModel(str(floating), device="cpu", reference_compat=False)
    body = ('<robot name="r"><link name="a"/><link name="b"/><link name="c"/>'
            '<joint name="j" type="%s"><parent link="a"/><child link="b"/><origin xyz="0.1 0 0.2" rpy="0.3 0 0"/><axis xyz="0 1 0"/>'
            '<limit lower="-1" upper="1" effort="1" velocity="1"/></joint>'
            '<joint name="k" type="%s"><parent link="b"/><child link="c"/><origin xyz="0 0.3 0"/><axis xyz="0 0 -1"/>'
            '<limit lower="-1" upper="1" effort="1" velocity="1"/></joint></robot>')
    models = []
    for n, kinds in enumerate((("floating", "planar"), ("revolute", "revolute"))):
        path = tmp_path / ("exotic%d.urdf" % n)
        path.write_text(body % kinds)
        with contextlib.redirect_stdout(io.StringIO()):
            models.append(DifferentiableRobotModel(str(path), device="cpu"))
    assert models[0]._n_dofs == 2 and models[0]._controlled_joints == models[1]._controlled_joints
    q, qd, qdd = (torch.rand(9, 2, generator=torch.Generator().manual_seed(k)) - 0.5 for k in range(3))
    for a, b in zip(models[0].compute_forward_kinematics(q, "c"), models[1].compute_forward_kinematics(q, "c")):
        assert torch.equal(a, b)
    assert torch.equal(models[0].compute_inverse_dynamics(q, qd, qdd), models[1].compute_inverse_dynamics(q, qd, qdd))
    order = tmp_path / "order.urdf"
    order.write_text('<robot name="r"><link name="a"/><link name="c"/><link name="b"/>'
                     '<joint name="j1" type="fixed"><parent link="a"/><child link="b"/></joint>'
                     '<joint name="j2" type="fixed"><parent link="b"/><child link="c"/></joint></robot>')
    with pytest.raises(UnsupportedRobotError):
        DifferentiableRobotModel(str(order), device="cpu")


# ------------------------------------------------------------------ walk programs
@pytest.mark.parametrize("robot", ALL_ROBOTS)
def test_walk_program_invariants(robot):
    m = load_model(robot)
    spec = m._spec
    L = spec.n_links
    prog = build_walk(spec, whole_tree=True)
    assert prog.n_ops == L - 1 and prog.capacity % 4 == 0 and prog.n_ops <= prog.capacity < prog.n_ops + 8
    assert sorted(prog.links.tolist()) == list(range(1, L))
    assert prog.dof_mask == (1 << spec.n_dofs) - 1
    saved = {}
    for k in range(prog.n_ops):
        link = int(prog.links[k]); src = int(prog.ops_i[k, OPI_SRC]); par = int(spec.parent[link])
        if src == SRC_ROOT:
            assert par == 0
        elif src == SRC_PREV:
            assert k > 0 and int(prog.links[k - 1]) == par
        else:
            assert saved[src] == par          # the slot currently holds the parent's state
        if prog.ops_i[k, OPI_SAVE] >= 0:
            saved[int(prog.ops_i[k, OPI_SAVE])] = link
        assert int(prog.ops_i[k, OPI_DOF]) == int(spec.dof[link])
    assert np.all(prog.ops_i[prog.n_ops:, OPI_DOF] == -1) and np.all(prog.ops_i[prog.n_ops:, OPI_SRC] == SRC_PREV)
    assert np.array_equal(prog.ops_i_dev, prog.ops_i.T)
    # chain walk of the deepest link: exactly its ancestors, in order, outputs on the last op
    deepest = max(range(L), key=lambda i: len(spec.chain_to(i)))
    chain = build_walk(spec, targets=[deepest])
    assert chain.links.tolist() == spec.chain_to(deepest) and chain.n_slots == 0
    assert int(chain.ops_i[chain.n_ops - 1, OPI_OUT]) == 0
    # a serial chain of up to 16 ops also carries its DoF columns and prismatic bits as plain struct fields (DRM_WALK_CHAIN_DOFS,
    # ABI 8: launch arguments of the chain kernels) — the same facts as the per-op columns, and they reach struct drm_walk
    from differentiable_robot_model_amd.flatten import KIND_PRISMATIC, SHAPE_CHAIN_DOFS, SHAPE_SERIAL_CHAIN
    for link in range(1, L):
        c = build_walk(spec, targets=[link])
        assert bool(c.shape & SHAPE_CHAIN_DOFS) == (bool(c.shape & SHAPE_SERIAL_CHAIN) and c.n_ops <= 16)
        if not c.shape & SHAPE_CHAIN_DOFS:
            assert not any(c.chain_dof1) and c.chain_prismatic == 0
            continue
        for k in range(16):
            dof = int(c.ops_i[k, OPI_DOF]) if k < c.n_ops else -1
            assert c.chain_dof1[k] == dof + 1
            slides = dof >= 0 and int(spec.kind[int(c.links[k])]) == KIND_PRISMATIC
            assert bool((c.chain_prismatic >> k) & 1) == slides
        w = backend.fill_walk_struct(backend.DrmWalk, c, 0, 0, spec.n_dofs, 2)
        assert list(w.chain_dof1) == list(c.chain_dof1) and w.chain_prismatic == c.chain_prismatic
    assert not prog.shape & SHAPE_CHAIN_DOFS or (prog.shape & SHAPE_SERIAL_CHAIN and prog.n_ops <= 16)


def test_axis_canonicalisation_is_a_pure_reindexing():
    m = load_model("allegro_left")   # has x, y and -z joints
    spec = m._spec
    assert set(int(a) for a in spec.axis_idx[spec.dof >= 0]) == {0, 1, 2}
    prog = build_walk(spec, whole_tree=True)
    table = m._link_table().numpy().reshape(-1)
    ops_f = table[prog.gather.reshape(-1)].reshape(prog.capacity, 32)
    for k in range(prog.n_ops):
        link = int(prog.links[k])
        raw = table[link * 32:(link + 1) * 32]
        # op rows: F and t interleaved in the FT block (first 12 floats), the rest as in the link table
        f_cols = [opf_fij(i, j) for i in range(3) for j in range(3)]
        t_cols = [opf_ti(i) for i in range(3)]
        assert sorted(f_cols + t_cols) == list(range(12))
        assert sorted(ops_f[k, f_cols].tolist()) == sorted(raw[OPF_F:OPF_F + 9].tolist())
        assert sorted(ops_f[k, t_cols].tolist()) == sorted(raw[OPF_T:OPF_T + 3].tolist())
        for a, b in ((OPF_MCOM, 3), (OPF_IO, 9)):
            assert sorted(ops_f[k, a:a + b].tolist()) == sorted(raw[a:a + b].tolist())
    # the signs only ever flip entries (exact), and only for negative axes
    assert set(np.unique(prog.gsign).tolist()) <= {-1.0, 1.0}
    ident = ops_f[prog.n_ops:]
    eye = np.zeros(32, np.float32)
    eye[[opf_fij(0, 0), opf_fij(1, 1), opf_fij(2, 2)]] = 1.0
    assert np.array_equal(ident, np.tile(eye, (len(ident), 1)))


def test_capacity_limit():
    m = load_model("panda_no_gripper")
    assert build_walk(m._spec, targets=[8]).capacity == 8
    assert build_walk(m._spec, targets=[2]).capacity == 4


# ------------------------------------------------------------------ API behaviour without a GPU
def test_neither_library_stands_in_for_the_other():
    """The library is chosen by the device of the tensors and by nothing else: without libdrm_hip.so a HIP device raises even
    though libdrm_cpu.so is there, and a CPU model without libdrm_cpu.so raises even though libdrm_hip.so is there."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import importlib, sys, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        backend = importlib.import_module("differentiable-robot-model_amd.backend")
        kind, missing = sys.argv[1:3]
        try:
            backend.library_for(torch.device(kind))
        except backend.NativeLibraryError as err:
            assert missing in str(err), err
            print("raised")
        assert backend._libs == {}, "nothing else was loaded in its place"
    """) % (ROOT, os.path.join(ROOT, "tests"))
    for kind, var, missing in (("cuda", "DRM_HIP_LIBRARY", "no CPU fallback"), ("cpu", "DRM_CPU_LIBRARY", "host build of the library not found")):
        env = dict(os.environ, **{var: "/nonexistent/lib.so"})
        out = subprocess.run([sys.executable, "-c", code, kind, missing], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "raised" in out.stdout, out.stderr[-2000:]


def test_tensor_check_and_errors():
    m = load_model("panda_no_gripper", device="cpu")
    with pytest.raises(AssertionError):                       # ndim must be 1 or 2 (robot_model.py:42-43)
        m.compute_forward_kinematics(torch.zeros(2, 3, 7), "panda_virtual_ee_link")
    with pytest.raises(AssertionError):                       # batch mismatch (robot_model.py:45-48)
        m.compute_inverse_dynamics(torch.zeros(3, 7), torch.zeros(4, 7), torch.zeros(3, 7))
    with pytest.raises(AssertionError):                       # wrong DoF count (robot_model.py:151-154)
        m.compute_inverse_dynamics(torch.zeros(3, 6), torch.zeros(3, 6), torch.zeros(3, 6))
    with pytest.raises(KeyError):                             # unknown link
        m.compute_forward_kinematics(torch.zeros(3, 7), "no_such_link")
    with pytest.raises(AttributeError):                       # robot_model.py:676-679
        m.make_link_param_learnable("panda_link1", "colour", torch.nn.Identity())
    assert m.get_link_names()[0] == "panda_link0" and len(m.get_joint_limits()) == 7


def test_learnable_parameters_reach_the_constant_table():
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    torch.manual_seed(0)
    m = load_model("iiwa7", device="cpu")
    before = m._link_table().clone()
    m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(dim1=1, dim2=3))
    m.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(dim1=1, dim2=3))
    names = [n for n, _ in m.named_parameters()]
    assert len(names) == 2 and all("_bodies.1." in n for n in names)
    table = m._link_table()
    assert table.requires_grad and not torch.equal(table[1], before[1]) and torch.equal(table[2:], before[2:])
    table[:, :12].sum().backward()
    grads = [p.grad for p in m.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    m.freeze_learnable_link_param("iiwa_link_1", "trans")
    assert sum(p.requires_grad for p in m.parameters()) == 1
    m.unfreeze_learnable_link_param("iiwa_link_1", "trans")
    assert sum(p.requires_grad for p in m.parameters()) == 2
    assert "_bodies.1.trans.param" in m.state_dict()


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "drm_hip.h")).read()
    declared = set(re.findall(r"\b(drm_[a-z_]+)\s*\(", header))
    assert declared == set(backend.EXPORTS), declared ^ set(backend.EXPORTS)
    assert os.path.exists(backend.LIB_PATH), "run `python __graft_entry__.py build` first"
    lib = ctypes.CDLL(backend.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.drm_abi_version() == backend.ABI_VERSION == int(re.search(r"#define DRM_ABI_VERSION (\d+)", header).group(1))


def test_abi_argument_errors_need_no_gpu():
    lib = backend.load_library()
    assert lib.drm_fk_jacobian(None, None, 1, None, None, None, None, None) == -1
    assert b"NULL" in lib.drm_last_error()
    w = backend.DrmWalk(1, 1, 9, 8, 7, 0, 0, 2, 0)   # n_ops > capacity
    assert lib.drm_rnea(ctypes.byref(w), None, None, None, 1, 0, None, None, None) == -1
    w = backend.DrmWalk(1, 1, 8, 10, 7, 0, 0, 2, 0)  # capacity is not a multiple of 4
    assert lib.drm_fk(ctypes.byref(w), None, 1, 1, None, None, None) == -1
    w = backend.DrmWalk(1, 1, 8, 8, 7, 17, 0, 2, 0)  # more save slots than the kernels have
    assert lib.drm_fk(ctypes.byref(w), None, 1, 1, None, None, None) == -2
    # struct drm_walk: 48 bytes of scalars + n_segments + seg_begin[9] + seg_dof_lo[8] + seg_dof_cnt[8] + prefix_end + seg_leaf_begin[9]
    assert ctypes.sizeof(backend.DrmWalk) == 48 + 4 * (1 + 9 + 8 + 8 + 1 + 9) + 16 + 4 + 4 + 16 * 8    # (+ chain_dof1, chain_prismatic, reserved0: ABI 8; special[4]: ABI 9; special[12]: ABI 10; special[16]: ABI 11)


def test_rnea_backward_scratch_covers_the_fanned_out_launch():
    """drm_rnea_backward_scratch_floats: one row of partial sums (capacity x 32 floats) per wavefront — and the launch that
    fans a hand's fingers out over wavefronts writes one per tile AND segment (up to 8), capped at 2048: a query sized for
    one row per tile let that launch write past the scratch buffer (caught as an intermittent GPU fault at B = 700)."""
    lib = backend.load_library()
    lib.drm_rnea_backward_scratch_floats.restype = ctypes.c_int64
    lib.drm_rnea_backward_scratch_floats.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    cap, n = 20, 16
    for B in (1, 64, 700, 65536, 1 << 20):
        tiles = (B + 63) // 64
        rows = min(tiles * 8, 2048)
        assert lib.drm_rnea_backward_scratch_floats(B, cap, n, 1) >= rows * cap * 32, B


# ---------------------------------------------------------------------------------------------- scratch of the persistent kernels
SCRATCH_QUERIES = ("drm_rnea_scratch_floats", "drm_crba_scratch_floats", "drm_forward_dynamics_scratch_floats")




def test_scratch_queries_answer_zero_without_a_device():
    """The scratch of the persistent dynamics kernels is sized by what the DEVICE holds at once (occupancy x CUs): on a box
    without one the queries return 0 and say why; they do not crash."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    from differentiable_robot_model_amd.flatten import build_walk
    from helpers import load_model
    from test_host_emu import host_walk
    lib = backend.load_library()
    m = load_model("iiwa7_allegro")
    walk, _keep = host_walk(m, build_walk(m._spec, whole_tree=True))
    for name in SCRATCH_QUERIES:
        assert getattr(lib, name)(ctypes.byref(walk), ctypes.c_int64(1000)) == 0
        assert lib.drm_last_error() != b""


@pytest.mark.gpu
def test_gpu_scratch_of_the_persistent_kernels():
    """Robots with a long segment (an arm carrying a hand) need caller-owned scratch wherever their rows go through the loop
    kernels (inverse dynamics, the mass matrix, forward dynamics): the queries say how much, do NOT grow with the batch once the
    grid is full (persistent blocks own a slice each), and a launch without scratch is refused; 7-DoF arms and hands need none."""
    import torch
    from helpers import load_model, sample_states
    lib = backend.load_library()
    m = load_model("iiwa7_allegro", "cuda")
    dw = m._dynamics_walk()
    of = m._ops_f(dw)
    walk = backend._walk_struct(dw.program, of, dw.ops_i, m._n_dofs)
    for name in SCRATCH_QUERIES:
        small, big, bigger = (int(getattr(lib, name)(ctypes.byref(walk), ctypes.c_int64(B))) for B in (64, 1 << 20, 1 << 22))
        # an arm that carries a hand: full ALIGNED 64-row tiles run the straight-line kernels (drm_arm_hand.hip, no scratch); the
        # query cannot see the caller's pointers, so it sizes for min(tiles, 64) tiles of the loop kernel — the ragged tail, and a
        # call with misaligned pointers (ABI 9: served on at most 64 blocks instead of refused)
        assert 0 < small < big == bigger == 64 * small, (name, small, big, bigger)
        assert int(getattr(lib, name)(ctypes.byref(walk), ctypes.c_int64((1 << 20) + 7))) == big
    # ... the same walk without its shape bit takes the persistent loop kernels for every row: their scratch does NOT grow with
    # the batch once the grid is full (a persistent block owns a slice)
    generic = backend._walk_struct_build(dw.program, of, dw.ops_i, m._n_dofs)   # (a private copy: _walk_struct hands out a cached struct)
    generic.shape &= ~4
    for name in SCRATCH_QUERIES:
        small, big, bigger = (int(getattr(lib, name)(ctypes.byref(generic), ctypes.c_int64(B))) for B in (64, 1 << 20, 1 << 22))
        assert 0 < small < big == bigger < (1 << 28), (name, small, big, bigger)   # < 1 GiB whatever the batch
    B, n = 130, m._n_dofs
    q, qd, qdd = (torch.from_numpy(a).cuda() for a in sample_states(m, B, seed=3))
    out, H = torch.empty(B, n, device="cuda"), torch.empty(B, n, n, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.drm_rnea(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), B, 3, out.data_ptr(), None, st) == -1
    assert b"scratch" in lib.drm_last_error()          # (the two rows behind the two full tiles)
    # ... and a misaligned view (a row slice starts 4 n bytes in) is SERVED, by the loop kernel, with the scratch the query asks for:
    # the same torques as the aligned call to rounding
    need = int(lib.drm_rnea_scratch_floats(ctypes.byref(walk), ctypes.c_int64(B - 1)))
    scr = torch.empty(need, device="cuda")
    ref = m.compute_inverse_dynamics(q[1:].clone(), qd[1:].clone(), qdd[1:].clone())
    assert q[1:].data_ptr() & 15
    assert lib.drm_rnea(ctypes.byref(walk), q[1:].data_ptr(), qd[1:].data_ptr(), qdd[1:].data_ptr(), B - 1, 3, out[1:].data_ptr(),
                        scr.data_ptr(), st) == 0, lib.drm_last_error()
    torch.cuda.synchronize()
    assert torch.allclose(out[1:], ref, rtol=2e-5, atol=2e-5)
    assert lib.drm_crba(ctypes.byref(walk), q.data_ptr(), B, H.data_ptr(), None, st) == -1
    assert b"scratch" in lib.drm_last_error()
    assert lib.drm_forward_dynamics(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), B, 3, out.data_ptr(), None, st) == -1
    assert b"scratch" in lib.drm_last_error()
    torch.cuda.synchronize()
    for robot in ("panda_no_gripper", "allegro_left"):
        r = load_model(robot, "cuda")
        rw = r._dynamics_walk()
        w = backend._walk_struct(rw.program, r._ops_f(rw), rw.ops_i, r._n_dofs)
        for name in SCRATCH_QUERIES:     # (for a caller that guarantees aligned pointers — this package's bindings do)
            fn = getattr(lib, name + "_aligned")
            fn.restype, fn.argtypes = ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64]
            assert fn(ctypes.byref(w), ctypes.c_int64(1 << 20)) == 0, (robot, name)
    fn = lib.drm_rnea_scratch_floats_aligned
    assert fn(ctypes.byref(walk), ctypes.c_int64(1 << 20)) == 0 and fn(ctypes.byref(walk), ctypes.c_int64((1 << 20) + 7)) > 0


def test_departures_from_the_reference_are_announced():
    """The constructor's defaults are the reference's (device CPU, every non-fixed joint an axis-aligned revolute one); a robot
    with a prismatic joint says once at construction that it is modelled upstream's way and names the opt-in;
    reference_compat=False, and robots without such joints, stay silent."""
    import contextlib
    import io
    import warnings
    from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel, robot_description_folder
    path = lambda r: os.path.join(robot_description_folder, r + ".urdf")
    from differentiable_robot_model_amd import robot_model as rm
    rm._WARNED_URDFS.discard(os.path.abspath(path("panda")))     # (once per URDF and process: forget earlier tests' models)
    with warnings.catch_warnings(record=True) as seen, contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("always")
        DifferentiableRobotModel(path("panda"))                                  # (the defaults: the reference's joint model)
        DifferentiableRobotModel(path("panda"), device="cpu", reference_compat=False)
        DifferentiableRobotModel(path("iiwa7"), device="cpu")
    texts = [str(w.message) for w in seen if "reference_compat" in str(w.message)]
    assert len(texts) == 1 and "panda_leftfinger" in texts[0] and "robot_model.py:122-126" in texts[0]
    assert "REVOLUTE" in texts[0] and "reference_compat=False" in texts[0]


@pytest.mark.gpu
def test_gpu_plans_refuse_buffers_they_would_have_to_copy():
    """A prepared launch works on the caller's own q / qd / qdd (an MPC loop overwrites them and replays); a misaligned row
    slice or a non-fp32 tensor would have to be copied, which would silently detach the plan from the caller's buffer."""
    import torch
    from helpers import load_model
    m = load_model("panda_no_gripper", "cuda")
    q = torch.zeros(130, 7, device="cuda")
    plan = m.plan_fk_and_jacobian(q, "panda_virtual_ee_link")
    assert plan.q.data_ptr() == q.data_ptr()
    with pytest.raises(ValueError, match="16-byte aligned"):
        m.plan_fk_and_jacobian(q[1:], "panda_virtual_ee_link")
    with pytest.raises(ValueError, match="own buffer"):
        m.plan_inverse_dynamics(q.double(), q, q)
    # the eager API takes the same slice (the binding clones it)
    pos, quat, lin, ang = m.compute_fk_and_jacobian(q[1:], "panda_virtual_ee_link")
    assert torch.equal(pos, plan_out(m, q)[0][1:])


def plan_out(m, q):
    return m.compute_fk_and_jacobian(q, "panda_virtual_ee_link")


def test_fan_partition_refuses_a_walk_that_writes_a_save_slot_twice():
    """The fan-out FK kernels share the save slots between wavefronts that never synchronise: a walk in which TWO ops write the
    same slot (legal in sequence: a slot is reused once a sub-tree is done) must not be partitioned, wherever the two writes sit
    — checked inside fk_fan_partition itself, not left to the caller's `slots_unique` gate."""
    from differentiable_robot_model_amd.flatten import OPI_SAVE, fk_fan_partition
    m = load_model("allegro_left")
    idx = [m._name_to_idx_map[t] for t in ("link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip")]
    prog = build_walk(m._spec, targets=idx)
    parent_op = [int(p) for p in prog.parent_op] if hasattr(prog, "parent_op") else None
    if parent_op is None:
        link_to_op = {int(l): k for k, l in enumerate(prog.links[:prog.n_ops])}
        parent_op = [link_to_op.get(int(m._spec.parent[int(l)]), -1) for l in prog.links[:prog.n_ops]]
    ops = [list(map(int, row)) for row in prog.ops_i[:prog.n_ops]]
    assert fk_fan_partition(ops, parent_op, prog.n_ops) is not None
    # the same walk with one more op writing an already used slot, inside ONE run (the per-run set comparison cannot see it)
    used = [o[OPI_SAVE] for o in ops if o[OPI_SAVE] >= 0]
    spare = next(k for k in range(prog.n_ops) if ops[k][OPI_SAVE] < 0)
    twice = [list(o) for o in ops]
    twice[spare][OPI_SAVE] = used[0] if used else 0
    if not used:
        other = next(k for k in range(prog.n_ops) if k != spare)
        twice[other][OPI_SAVE] = 0
    assert fk_fan_partition(twice, parent_op, prog.n_ops) is None


def test_joint_model_warning_is_given_once_per_urdf():
    import warnings
    from differentiable_robot_model_amd import robot_model as rm
    rm._WARNED_URDFS.discard(os.path.abspath(urdf_path("panda")))
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        for _ in range(3):
            load_model("panda", reference_compat=True)       # (the default: upstream's joint model, announced once)
        load_model("panda", reference_compat=False)
    mine = [w for w in seen if "modelled as REVOLUTE joints, as the reference does" in str(w.message)]
    assert len(mine) == 1 and mine[0].filename.endswith("helpers.py")      # (the caller's frame, not robot_model.py)
