"""Models on device="cpu" — the reference's default device (robot_model.py:100-104; its own suite constructs every model there,
tests/test_kinematics_dynamics.py:133-137) — compute through csrc/libdrm_cpu.so, the HOST build of the C ABI of
include/drm_hip.h (csrc/drm_cpu.cpp: the kernels' own per-sample headers compiled by g++).

Same fixtures, same checks and the same bars as the kernels' tests (the bodies are shared with the -m gpu tests): forward
outputs and torch-autograd gradients of the UNMODIFIED reference.  Plus what is particular to the host build: it exports the
whole ABI, its batch sums do not depend on the thread count, and it is never what a HIP tensor reaches.
"""
import ctypes
import importlib

import numpy as np
import pytest
import torch

from helpers import GOLDEN_ROBOTS, load_model
import test_fk_backward as fkb
import test_forward_dynamics as fdt
import test_golden_tiles as tl
import test_golden_wide as wide
import test_mass_matrix as mmt
import test_rnea_backward as rbt

backend = importlib.import_module("differentiable_robot_model_amd.backend")


def test_host_build_exports_the_whole_abi(cpu_library):
    lib = ctypes.CDLL(backend.CPU_LIB_PATH)
    for sym in backend.EXPORTS:
        getattr(lib, sym)
    assert lib.drm_abi_version() == backend.ABI_VERSION and lib.drm_walk_sizeof() == ctypes.sizeof(backend.DrmWalk)
    # per-robot code objects are HIP kernels
    out = ctypes.c_void_p()
    assert cpu_library.drm_special_load(b"/nonexistent", b"k", ctypes.byref(out)) == -2
    assert b"HIP" in cpu_library.drm_last_error()


@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_cpu_forward_vs_reference_tiles(robot, links, cpu_library):
    wide.check_gpu_vs_reference(tl.tiles("fwd"), robot, links, device="cpu")


@pytest.mark.parametrize("case", fkb.CASES)
def test_cpu_fk_backward_vs_reference_autograd(case, cpu_library):
    fkb.check_gpu_backward_vs_reference_autograd(fkb.load_golden_grad(), case, device="cpu")


@pytest.mark.parametrize("case", rbt.CASES)
def test_cpu_rnea_backward_vs_reference_autograd(case, cpu_library):
    rbt.check_gpu_backward_vs_reference_autograd(rbt.load_golden_dyn(), case, device="cpu")


@pytest.mark.parametrize("case", mmt.H_GRAD_CASES)
def test_cpu_mass_matrix_backward_vs_reference_autograd(case, cpu_library):
    mmt.check_gpu_mass_matrix_backward_vs_reference_autograd(mmt.load_golden_grad_mass(), case, device="cpu")


@pytest.mark.parametrize("case", fdt.FD_GRAD_CASES)
def test_cpu_forward_dynamics_backward_vs_reference_autograd(case, cpu_library):
    fdt.check_gpu_forward_dynamics_backward_vs_reference_autograd(fdt.load_golden_grad_fd(), case, device="cpu")


@pytest.mark.parametrize("case", ["iiwa7", "panda_no_gripper"])
def test_cpu_fk_backward_full_tiles_and_fused_step(case, cpu_library):
    """192 rows: the reference's gradients through compute_forward_kinematics, and fk_mse_loss (drm_fk_mse of the host build: any
    batch size) against the same numbers."""
    g = tl.tiles("grad")
    fkb.check_gpu_backward_vs_reference_autograd(g, case, device="cpu")
    targets = [str(t) for t in g[case + "/targets"]]
    m = fkb.learnable_model(g, case, "cpu")
    q = torch.from_numpy(g[case + "/q"].copy()).requires_grad_(True)
    want = torch.from_numpy(g["%s/want/%s" % (case, targets[0])].copy())
    loss = m.fk_mse_loss(q, targets[0], want)
    assert type(loss.grad_fn).__name__.startswith("_FkMse")
    loss.backward()
    assert abs(loss.item() - float(g[case + "/loss"])) < 1e-6
    assert fkb.close(q.grad.numpy(), g[case + "/grad_q"])
    for link in g[case + "/learnable"]:
        body = m._bodies[m._name_to_idx_map[str(link)]]
        for pname in ("trans", "rot_angles"):
            assert fkb.close(getattr(body, pname).param.grad.numpy(), g["%s/grad/%s/%s" % (case, link, pname)]), (case, link, pname)


def test_cpu_batch_sums_do_not_depend_on_the_thread_count(cpu_library):
    """grad_ops_f is one partial per 256 rows added in chunk order: bit-identical with 1 thread and with 7."""
    g = tl.tiles("grad_dyn")
    case = "panda"
    out = []
    for threads in (1, 7, 1):
        cpu_library.drm_cpu_set_threads(threads)
        m, params = rbt.learnable_model(g, case, "cpu")
        rep = 5                                                     # 960 rows: four chunks
        q, qd, qdd = (torch.from_numpy(np.tile(g["%s/%s" % (case, k)], (rep, 1))).requires_grad_(True) for k in ("q", "qd", "qdd"))
        tau = m.compute_inverse_dynamics(q, qd, qdd)
        tau.pow(2).mean().backward()
        out.append([p.grad.clone() for p in params.values()] + [q.grad.clone(), tau.detach().clone()])
    cpu_library.drm_cpu_set_threads(torch.get_num_threads())
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    for a, b in zip(out[0], out[2]):
        assert torch.equal(a, b)


def test_cpu_empty_and_single_row_batches(cpu_library):
    m = load_model("panda", "cpu")
    n = m._n_dofs
    link = "panda_virtual_ee_link"
    for B in (0, 1):
        q = torch.zeros(B, n)
        pos, quat = m.compute_forward_kinematics(q, link)
        lin, ang = m.compute_endeffector_jacobian(q, link)
        tau = m.compute_inverse_dynamics(q, q, q)
        H = m.compute_lagrangian_inertia_matrix(q)
        acc = m.compute_forward_dynamics(q, q, q)
        assert pos.shape == (B, 3) and quat.shape == (B, 4) and lin.shape == (B, 3, n) and ang.shape == (B, 3, n)
        assert tau.shape == (B, n) and H.shape == (B, n, n) and acc.shape == (B, n)
        assert all(t.device.type == "cpu" for t in (pos, quat, lin, ang, tau, H, acc))


def test_cpu_all_links_and_fused_call_match_the_single_calls(cpu_library):
    m = load_model("iiwa7_allegro", "cpu")
    g = torch.Generator().manual_seed(3)
    q, qd, qdd = (torch.rand(300, m._n_dofs, generator=g) - 0.5 for _ in range(3))
    poses = m.compute_forward_kinematics_all_links(q)
    for name in list(poses)[::5]:
        pos, quat = m.compute_forward_kinematics(q, name)
        # (the walk of one link folds the fixed joints along its chain differently from the walk of all links: rounding)
        assert (poses[name][0] - pos).abs().max() < 2e-6 and (poses[name][1] - quat).abs().max() < 2e-6
    m = load_model("panda_no_gripper", "cpu")
    q, qd, qdd = (torch.rand(300, 7, generator=g) - 0.5 for _ in range(3))
    tau, pos, quat = m.compute_fk_and_inverse_dynamics(q, qd, qdd, "panda_virtual_ee_link")
    assert torch.equal(tau, m.compute_inverse_dynamics(q, qd, qdd))
    p2, q2 = m.compute_forward_kinematics(q, "panda_virtual_ee_link")
    assert torch.equal(pos, p2) and torch.equal(quat, q2)


def test_the_library_follows_the_tensors_device(cpu_library):
    """libdrm_cpu.so is what CPU tensors reach and nothing else: the choice is made from the tensor's device alone, and a model
    does not take tensors from another device."""
    assert backend.library_for(torch.device("cpu")) is cpu_library
    with pytest.raises(RuntimeError, match="HIP device or on the CPU"):
        backend.library_for(torch.device("meta"))
    m = load_model("panda", "cpu")

    class TableOnHipDevice(object):      # (no HIP device on the CPU test box: what the check reads of a tensor)
        is_cuda, device = True, torch.device("cuda", 0)

    with pytest.raises(RuntimeError, match="model's tables are on cuda:0"):
        backend._lib_of(torch.zeros(2, m._n_dofs), "q", TableOnHipDevice())
    with pytest.raises(RuntimeError, match="HIP device"):
        m.specialize()


def test_host_work_in_cxx_is_the_python_host_path(cpu_library, hostcall_module, monkeypatch):
    """csrc/drm_hostcall.so does per call what backend.fk / fk_jacobian / rnea do in Python (checks, one output allocation, the C
    ABI call): the same bits from both, for conforming inputs; inputs the kernels do not take as they are (float64, a column
    slice, a misaligned row slice) come back NOT_CONFORMING from C++ and are converted by the Python path; empty batches; errors
    of the C ABI surface as the same exceptions."""
    m = load_model("panda", "cpu")
    n, link = m._n_dofs, "panda_virtual_ee_link"
    g = torch.Generator().manual_seed(5)
    q, qd, qdd = (torch.rand(130, n, generator=g) - 0.5 for _ in range(3))

    def results(*args):
        return [m.compute_forward_kinematics(args[0], link), m.compute_endeffector_jacobian(args[0], link),
                (m.compute_inverse_dynamics(*args),), (m.compute_non_linear_effects(args[0], args[1]),),
                (m.compute_lagrangian_inertia_matrix(args[0]),), (m.compute_forward_dynamics(*args),)]

    fast = results(q, qd, qdd)
    monkeypatch.setattr(backend, "_hostcall", None)
    assert backend.hostcall() is None and m.compute_forward_kinematics.__module__.startswith(backend.__name__.rsplit(".", 1)[0])
    slow = results(q, qd, qdd)
    monkeypatch.setattr(backend, "_hostcall", hostcall_module)
    for a, b in zip(fast, slow):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # not conforming: served by the Python path, same numbers to rounding of the conversion
    wide = torch.zeros(131, n + 3)
    wide[:130, :n] = q
    for variant in (q.double(), wide[:130, :n], torch.cat([torch.zeros(1, n), q])[1:]):
        assert hostcall_module.fk_jacobian(0, 0, variant, n, 0)[4] == hostcall_module.NOT_CONFORMING
        lin, ang = m.compute_endeffector_jacobian(variant, link)
        assert torch.equal(lin, fast[1][0]) and torch.equal(ang, fast[1][1])
        assert torch.equal(m.compute_inverse_dynamics(variant, qd, qdd), fast[2][0])
    # empty batches and the 1-D convenience shape of tensor_check
    e = torch.zeros(0, n)
    assert m.compute_endeffector_jacobian(e, link)[0].shape == (0, 3, n) and m.compute_inverse_dynamics(e, e, e).shape == (0, n)
    assert torch.equal(m.compute_forward_kinematics(q[3], link)[0], fast[0][0][3])
    # a C-ABI error comes back as the exception the Python path raises
    dw = m._dynamics_walk()
    bad = backend._walk_struct_build(dw.program, m._ops_f(dw), dw.ops_i, n)
    bad.n_dofs = 999
    tau, rc = hostcall_module.rnea(backend._fn_addr(cpu_library, "drm_rnea"), backend._fn_addr(cpu_library, "drm_rnea_scratch_floats_aligned"),
                                   ctypes.addressof(bad), torch.zeros(2, 999), torch.zeros(2, 999), None, 999, 3, 0)
    assert rc == -2
    with pytest.raises(backend.KernelUnsupported):
        backend._check(rc, cpu_library)


def test_prepared_eager_calls_are_the_python_path(cpu_library, hostcall_module, monkeypatch):
    """Round 6: repeat calls of a constant model's hot public methods go through ONE C++ call — tensor_check
    (robot_model.py:25-84), the asserts, the allocation and the C-ABI launch in csrc/drm_hostcall.cpp FastCall.  Same bits as the
    Python path for 2-D and 1-D inputs; everything the prepared call does not take as it is falls back to the Python method, which
    converts / differentiates / words the reference's errors; a learnable parameter retires the prepared calls."""
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    monkeypatch.setattr(backend, "_hostcall", hostcall_module)
    m, ref = load_model("panda_no_gripper", "cpu"), load_model("panda_no_gripper", "cpu")
    n, link = m._n_dofs, "panda_virtual_ee_link"
    g = torch.Generator().manual_seed(7)
    q, qd, qdd = (torch.rand(70, n, generator=g) - 0.5 for _ in range(3))
    monkeypatch.setattr(backend, "_hostcall", None)         # `ref` never prepares a call: the Python host path, every time
    want = [ref.compute_forward_kinematics(q, link), ref.compute_endeffector_jacobian(q, link), ref.compute_fk_and_jacobian(q, link),
            (ref.compute_inverse_dynamics(q, qd, qdd),), (ref.compute_inverse_dynamics(q, qd, qdd, include_gravity=False, use_damping=False),),
            (ref.compute_non_linear_effects(q, qd),), (ref.compute_lagrangian_inertia_matrix(q),),
            (ref.compute_forward_dynamics(q, qd, qdd, use_damping=True),), ref.compute_fk_and_inverse_dynamics(q, qd, qdd, link)]
    monkeypatch.setattr(backend, "_hostcall", hostcall_module)

    def results(a, b, c):
        return [m.compute_forward_kinematics(a, link), m.compute_endeffector_jacobian(a, link), m.compute_fk_and_jacobian(a, link),
                (m.compute_inverse_dynamics(a, b, c),), (m.compute_inverse_dynamics(a, b, c, include_gravity=False, use_damping=False),),
                (m.compute_non_linear_effects(a, b),), (m.compute_lagrangian_inertia_matrix(a),),
                (m.compute_forward_dynamics(a, b, c, use_damping=True),), m.compute_fk_and_inverse_dynamics(a, b, c, link)]

    first = results(q, qd, qdd)
    assert set(m._fast_fk) == {link} and set(m._fast_jac) == {link} and m._fast_id is not None and not ref._fast_fk
    assert m._fast_crba is not None and m._fast_fd is not None and set(m._fast_fkid) == {link}
    for rep in range(3):          # (the first call prepares, the rest run prepared)
        for a, b in zip(results(q, qd, qdd), want):
            assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))
    # the prepared calls really serve: the Python method underneath is not reached
    monkeypatch.setattr(type(m), "_compute_endeffector_jacobian", lambda *a, **k: pytest.fail("the prepared call should have served this"))
    monkeypatch.setattr(type(m), "_compute_inverse_dynamics", lambda *a, **k: pytest.fail("the prepared call should have served this"))
    for name in ("_compute_lagrangian_inertia_matrix", "_compute_forward_dynamics", "_compute_fk_and_inverse_dynamics"):
        monkeypatch.setattr(type(m), name, lambda *a, **k: pytest.fail("the prepared call should have served this"))
    assert torch.equal(m.compute_lagrangian_inertia_matrix(q), want[6][0]) and torch.equal(m.compute_forward_dynamics(q, qd, qdd, use_damping=True), want[7][0])
    assert all(torch.equal(x, y) for x, y in zip(m.compute_fk_and_inverse_dynamics(q, qd, qdd, link), want[8]))
    assert m.compute_lagrangian_inertia_matrix(q[4]).shape == (n, n) and torch.equal(m.compute_fk_and_inverse_dynamics(q[4], qd[4], qdd[4], link)[1], want[8][1][4])
    assert torch.equal(m.compute_endeffector_jacobian(q, link)[0], want[1][0]) and torch.equal(m.compute_inverse_dynamics(q, qd, qdd), want[3][0])
    one = m.compute_endeffector_jacobian(q[4], link)          # 1-D in (a 16-byte aligned row), the batch dimension stripped (tensor_check)
    assert one[0].shape == (3, n) and torch.equal(one[0], want[1][0][4]) and torch.equal(one[1], want[1][1][4])
    assert torch.equal(m.compute_inverse_dynamics(q[4], qd[4], qdd[4]), want[3][0][4])
    assert m.compute_inverse_dynamics(torch.zeros(0, n), torch.zeros(0, n), torch.zeros(0, n)).shape == (0, n)
    monkeypatch.undo()
    monkeypatch.setattr(backend, "_hostcall", hostcall_module)
    # not taken as it is -> the Python method: another dtype, a strided view, gradients wanted, a 1-D / 2-D mix, a wrong width
    assert torch.equal(m.compute_endeffector_jacobian(q.double(), link)[0], want[1][0])
    wide = torch.zeros(70, n + 1)
    wide[:, :n] = q
    assert torch.equal(m.compute_forward_kinematics(wide[:, :n], link)[0], want[0][0])
    qg = q.clone().requires_grad_(True)
    m.compute_forward_kinematics(qg, link)[0].sum().backward()
    assert qg.grad is not None and float(qg.grad.abs().sum()) > 0
    with torch.no_grad():
        assert torch.equal(m.compute_forward_kinematics(qg, link)[0], want[0][0])       # (no graph wanted: prepared call, same bits)
    with pytest.raises(AssertionError, match="Batch size mismatch"):
        m.compute_inverse_dynamics(q, qd[0], qdd)
    with pytest.raises(AssertionError):
        m.compute_endeffector_jacobian(q[:, :5], link)
    with pytest.raises(AssertionError, match="ndim of 1 or 2"):
        m.compute_forward_kinematics(q.reshape(2, 35, n), link)
    with pytest.raises(KeyError):
        m.compute_forward_kinematics(q, "no_such_link")
    # a learnable parameter: the prepared calls snapshot constants and are retired
    m.make_link_param_learnable("panda_link2", "trans", UnconstrainedTensor(dim1=1, dim2=3))
    assert not m._fast_fk and not m._fast_jac and m._fast_id is None and m._fast_crba is None and m._fast_fd is None and not m._fast_fkid
    pos = m.compute_forward_kinematics(q, link)[0]
    assert pos.requires_grad and not m._fast_fk


def test_cpu_fingertips_in_one_call(cpu_library):
    """compute_forward_kinematics_links (drm_fk_fanout_links of the host build: a chain walk per fingertip, link-major outputs) gives
    what the per-link calls give."""
    for robot, tips, n in (("allegro_left", ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"], 16),
                           ("trifinger_edu", ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"], 9)):
        m = load_model(robot, "cpu")
        q = torch.rand(300, n, generator=torch.Generator().manual_seed(1)) - 0.5
        out = m.compute_forward_kinematics_links(q, tips)
        pairs = list(out.values()) if isinstance(out, dict) else list(out)
        for tip, (pos, quat) in zip(tips, pairs):
            p2, q2 = m.compute_forward_kinematics(q, tip)
            assert torch.equal(pos, p2) and torch.equal(quat, q2), (robot, tip)
