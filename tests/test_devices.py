"""Device smoke of the method surface under a process-wide default device — the reference's tests/test_devices.py:26-47: every
public method of a model on "cuda" and on "cpu", with the process default set to the GPU and to the CPU
(torch.set_default_tensor_type(torch.cuda.FloatTensor) there; torch.set_default_device here, its successor).

A package that builds a host tensor with a bare torch.tensor(...) / torch.zeros(...) lands it on the DEFAULT device, not the
model's; a user whose process defaults to the GPU then gets device-mismatch errors (or silent copies).  CPU leg: the default
device "meta" stands in for "a device that is not the model's" — anything the package creates without saying where would become
a meta tensor and fail.  GPU leg (-m gpu): the reference's matrix {model on cuda, cpu} x {default cuda, cpu}, models built UNDER
the default, results against the fp64 oracle."""
import contextlib
import io

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel
from helpers import TOL_JAC, TOL_POS, TOL_QUAT, TOL_TAU, max_err, quat_close, urdf_path
from oracle import Oracle


@contextlib.contextmanager
def default_device(name):
    before = torch.get_default_device()
    torch.set_default_device(name)
    try:
        yield
    finally:
        torch.set_default_device(before)


def every_method(model_device, robot="2link_robot", ee="endEffector", B=5):
    """Build the model and run the reference's list of methods (test_devices.py:38-47) + the learnable-parameter path; inputs
    are created on the model's device explicitly, as the reference's test does.  Returns what the calls returned, on the host."""
    with contextlib.redirect_stdout(io.StringIO()):
        m = DifferentiableRobotModel(urdf_path(robot), device=model_device)
    assert m._device.type == torch.device(model_device).type
    n = m._n_dofs
    g = torch.Generator().manual_seed(3)
    q, qd, qdd = (torch.rand([B, n], generator=g, device="cpu").to(m._device) for _ in range(3))
    out = {"q": q, "qd": qd, "qdd": qdd}
    m.update_kinematic_state(q, qd)
    out["pos"], out["quat"] = m.compute_forward_kinematics(q, ee)
    out["tau"] = m.compute_inverse_dynamics(q, qd, qdd)
    out["nle"] = m.compute_non_linear_effects(q, qd)
    out["H"] = m.compute_lagrangian_inertia_matrix(q)
    out["acc"] = m.compute_forward_dynamics(q, qd, qdd)
    out["lin_jac"], out["ang_jac"] = m.compute_endeffector_jacobian(q, ee)
    links = m.compute_forward_kinematics_all_links(q)
    assert set(links) == set(m.get_link_names())
    for t in out.values():
        assert t.device.type == m._device.type
    # one learnable link, forward + backward (robot_model.py:669-713)
    from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor
    name = m.get_link_names()[1]
    with torch.device(m._device):      # (the user's parametrisation lives where the user's model does)
        param = UnconstrainedTensor(dim1=1, dim2=3, init_std=0.1)
    m.make_link_param_learnable(name, "trans", param)
    pos, _ = m.compute_forward_kinematics(q, ee)
    pos.square().sum().backward()
    grads = [p.grad for p in m.parameters()]
    assert len(grads) == 1 and grads[0] is not None and grads[0].device.type == m._device.type
    assert bool(torch.isfinite(grads[0]).all())
    return m, {k: v.detach().cpu().numpy() for k, v in out.items()}


def against_oracle(m_spec, r):
    orc = Oracle(m_spec)
    q, qd, qdd = (r[k].astype(np.float64) for k in ("q", "qd", "qdd"))
    ee = len(m_spec.link_names) - 1
    rp, rq = orc.fk(q, [ee], np.float64)
    assert max_err(r["pos"], rp[:, 0]) <= TOL_POS["atol"] and quat_close(r["quat"], rq[:, 0], TOL_QUAT["atol"])[0]
    assert np.allclose(r["tau"], orc.rnea(q, qd, qdd, True, True, np.float64), **TOL_TAU)
    assert np.allclose(r["nle"], orc.rnea(q, qd, np.zeros_like(q), True, True, np.float64), **TOL_TAU)
    _, _, lin, ang = orc.fk_jacobian(q, ee, np.float64)
    assert np.allclose(r["lin_jac"], lin, **TOL_JAC) and np.allclose(r["ang_jac"], ang, **TOL_JAC)
    assert np.allclose(r["H"], orc.mass_matrix(q, dtype=np.float64), rtol=1e-4, atol=1e-5)
    assert np.allclose(r["acc"], orc.forward_dynamics(q, qd, qdd, dtype=np.float64), rtol=2e-3, atol=2e-3)


def test_cpu_model_does_not_lean_on_the_default_device():
    """A device="cpu" model built and used while the process default device is NOT the CPU (here: "meta")."""
    with default_device("meta"):
        m, r = every_method("cpu")
    against_oracle(m._spec, r)
    with default_device("cpu"):
        m2, r2 = every_method("cpu")
    for k in r:
        assert np.array_equal(r[k], r2[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("default", ["cuda", "cpu"])
@pytest.mark.parametrize("model_device", ["cuda", "cpu"])
def test_robot_model_under_default_device(model_device, default):
    """/root/reference/tests/test_devices.py:26-47 — model on cuda / cpu x process default cuda / cpu."""
    with default_device(default):
        m, r = every_method(model_device)
        if model_device == "cuda":
            m7, r7 = every_method("cuda", "panda_no_gripper", "panda_virtual_ee_link", B=70)
    against_oracle(m._spec, r)
    if model_device == "cuda":
        against_oracle(m7._spec, r7)
