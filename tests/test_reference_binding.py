"""INTEGRATION.md stub B, executed: differentiable-robot-model_amd/reference_binding.py binds the three hot methods to the C
ABI from nothing but the per-link dicts `URDFRobotModel.get_body_parameters_from_urdf` returns (the reference's format).

CPU (not gpu): the binding's own table construction against the package's (same constants, same walks), and — where
/root/reference exists — that the REFERENCE's loader output binds to bit-identical tables.  GPU: drm_fk / drm_fk_jacobian /
drm_rnea called through the binding against the fp64 oracle.
"""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

from differentiable_robot_model_amd.reference_binding import HipBinding, bind_reference_model, link_table
from differentiable_robot_model_amd.urdf_utils import URDFRobotModel
from helpers import TOL_JAC, TOL_POS, TOL_QUAT, TOL_TAU, load_model, quat_close, sample_states, urdf_path
from oracle import Oracle

ROBOTS = ["panda_no_gripper", "iiwa7", "allegro_left", "trifinger_edu"]


def dicts_of(robot, device="cpu"):
    um = URDFRobotModel(urdf_path(robot), device=torch.device(device))
    params = [um.get_body_parameters_from_urdf(i, link) for i, link in enumerate(um.robot.links)]
    parents = [None] + [um.get_name_of_parent_body(link.name) for link in um.robot.links[1:]]
    return params, parents


@pytest.mark.parametrize("robot", ROBOTS)
def test_binding_tables_equal_the_package_model(robot):
    params, parents = dicts_of(robot)
    b = HipBinding(params, parents, "cpu")
    m = load_model(robot)
    assert torch.equal(b.table, m._link_table())
    assert b.spec.parent.tolist() == m._spec.parent.tolist() and b.spec.dof.tolist() == m._spec.dof.tolist()
    b2 = bind_reference_model(m)                       # through a model's own loader (robot_model.py:107-137)
    assert torch.equal(b2.table, b.table)


@pytest.mark.skipif(not os.path.isdir("/root/reference/differentiable_robot_model"), reason="needs /root/reference (build container)")
def test_binding_takes_the_references_own_loader_output():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_import
    rm = ref_import.import_reference()
    path = os.path.join(ref_import.reference_data_dir(), "panda_description/urdf/panda_no_gripper.urdf")
    with contextlib.redirect_stdout(io.StringIO()):
        ref_model = rm.DifferentiableRobotModel(path)
    b = bind_reference_model(ref_model)
    assert torch.equal(b.table, load_model("panda_no_gripper")._link_table())


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ROBOTS)
def test_gpu_binding_calls_the_c_abi(robot):
    params, parents = dicts_of(robot, "cuda")
    b = HipBinding(params, parents, "cuda")
    m = load_model(robot)
    orc = Oracle(m._spec)
    q, qd, qdd = sample_states(m, 193, seed=77)
    dev = lambda a: torch.from_numpy(a).cuda()
    q64, qd64, qdd64 = (a.astype(np.float64) for a in (q, qd, qdd))
    link = m._bodies[-1].name
    idx = m._name_to_idx_map[link]
    pos, quat = b.fk(dev(q), link)
    lin, ang = b.jacobian(dev(q), link)
    tau = b.inverse_dynamics(dev(q), dev(qd), dev(qdd))
    rp, rq, rl, ra = orc.fk_jacobian(q64, idx, np.float64)
    assert np.abs(pos.cpu().numpy() - rp).max() <= TOL_POS["atol"] and quat_close(quat.cpu().numpy(), rq, TOL_QUAT["atol"])[0]
    assert np.abs(lin.cpu().numpy() - rl).max() <= TOL_JAC["atol"] and np.abs(ang.cpu().numpy() - ra).max() <= TOL_JAC["atol"]
    assert np.allclose(tau.cpu().numpy(), orc.rnea(q64, qd64, qdd64, True, True, np.float64), **TOL_TAU)
    every = b.fk_all_links(dev(q))                                            # -> drm_fk_links
    assert set(every) == {body.name for body in m._bodies}
    ap, aq = orc.fk(q64, list(range(len(m._bodies))), np.float64)
    for i, body in enumerate(m._bodies):
        p, r = every[body.name]
        assert p.is_contiguous() and np.abs(p.cpu().numpy() - ap[:, i]).max() <= TOL_POS["atol"], body.name
        assert quat_close(r.cpu().numpy(), aq[:, i], TOL_QUAT["atol"])[0], body.name
    root_pos, root_quat = b.fk(dev(q), m._bodies[0].name)
    assert not root_pos.any() and torch.equal(root_quat[:, 3], torch.ones(193, device="cuda"))
