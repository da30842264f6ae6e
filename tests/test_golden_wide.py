"""tests/golden/golden_wide.npz: 48 joint states per robot (the per-robot fixtures hold 7), q over the whole joint range,
through every entry point of the hot path of the UNMODIFIED reference (tests/golden/make_golden_wide.py).

CPU: the oracle (fp32 and fp64 builds) against it — the wider pin of the restatement.  GPU (-m gpu): the kernels, through
the API, against the same numbers."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR, GOLDEN_ROBOTS, TOL_TAU, load_model, max_err, quat_close
from oracle import Oracle

WIDE = np.load(os.path.join(GOLDEN_DIR, "golden_wide.npz"), allow_pickle=False)
F32 = dict(pos=1e-6, quat=1e-6, jac=1e-6)       # the oracle against the reference (test_oracle_golden.py)
TOL_H_REF = dict(atol=5e-5, rtol=2e-5)          # (test_mass_matrix.py)
GPU = dict(pos=2e-6, quat=2e-6, jac=2e-6)       # the kernels (helpers.TOL_*)
ARMS = ("panda_no_gripper", "iiwa7", "2link_robot", "panda", "fetch_arm_no_gripper", "fetch_arm_no_gripper_small_damping")


def rel(a, ref):
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    return float((np.abs(a - ref) / (1.0 + np.abs(ref))).max())


def test_the_fixture_covers_every_golden_robot():
    for robot, links in GOLDEN_ROBOTS:
        assert WIDE[robot + "/q"].shape[0] == 48 and list(WIDE[robot + "/links"]) == links


@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_vs_reference_wide(robot, links, dtype):
    check_oracle_vs_reference(WIDE, robot, links, dtype)


def check_oracle_vs_reference(WIDE, robot, links, dtype):
    m = load_model(robot)
    orc = Oracle(m._spec)
    q, qd, qdd, f = (WIDE["%s/%s" % (robot, k)].astype(dtype) for k in ("q", "qd", "qdd", "f"))
    for link in links:
        pos, quat, lin, ang = orc.fk_jacobian(q, m._name_to_idx_map[link], dtype)
        assert max_err(pos, WIDE["%s/pos_%s" % (robot, link)]) < F32["pos"]
        ok, flips = quat_close(quat, WIDE["%s/quat_%s" % (robot, link)], F32["quat"])
        assert ok and flips == 0
        assert max_err(lin, WIDE["%s/lin_%s" % (robot, link)]) < F32["jac"] and max_err(ang, WIDE["%s/ang_%s" % (robot, link)]) < F32["jac"]
    for g, d in ((1, 1), (0, 0)):
        tau = orc.rnea(q, qd, qdd, bool(g), bool(d), dtype)
        assert np.allclose(tau, WIDE["%s/tau_g%d_d%d" % (robot, g, d)], atol=2e-5, rtol=2e-5), (robot, g, d)
        acc = orc.forward_dynamics(q, qd, f, g, d, dtype)
        # two fp32 evaluations of the same recursion (the reference's and, for float32, the oracle's) differ by the
        # conditioning of the robot: 1e-3 covers the arm carrying a hand, the arms are at 1e-5
        assert rel(acc, WIDE["%s/acc_g%d_d%d" % (robot, g, d)]) < (1e-4 if robot in ARMS else 1e-3), (robot, g, d)
    H = orc.mass_matrix(q, False, False, dtype)
    assert np.allclose(H, WIDE[robot + "/H"], **TOL_H_REF), np.abs(H - WIDE[robot + "/H"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_gpu_vs_reference_wide(robot, links):
    check_gpu_vs_reference(WIDE, robot, links)


@pytest.mark.parametrize("robot,links", GOLDEN_ROBOTS)
def test_cpu_model_vs_reference_wide(robot, links, cpu_library):
    """A model on device="cpu" (the reference's default): the host build of the C ABI (libdrm_cpu.so), same bars as the kernels."""
    check_gpu_vs_reference(WIDE, robot, links, device="cpu")


def check_gpu_vs_reference(WIDE, robot, links, repeat=1, device="cuda"):
    """`repeat` > 1: the fixture's rows tiled that many times into ONE launch (so that a 192-row fixture reaches kernels that
    only engage beyond a launch size, e.g. the two-samples-per-lane arm kernels past 1 024 tiles); every copy must meet the
    reference on its own."""
    m = load_model(robot, device)
    q, qd, qdd, f = (torch.from_numpy(np.tile(WIDE["%s/%s" % (robot, k)], (repeat, 1))).to(device) for k in ("q", "qd", "qdd", "f"))
    if repeat > 1:
        WIDE = {k: (np.tile(WIDE[k], (repeat,) + (1,) * (WIDE[k].ndim - 1)) if WIDE[k].dtype == np.float32 else WIDE[k])
                for k in WIDE.files if k.startswith(robot + "/")}
    host = lambda t: t.cpu().numpy()
    for link in links:
        pos, quat = m.compute_forward_kinematics(q, link)
        lin, ang = m.compute_endeffector_jacobian(q, link)
        assert max_err(host(pos), WIDE["%s/pos_%s" % (robot, link)]) < GPU["pos"]
        assert quat_close(host(quat), WIDE["%s/quat_%s" % (robot, link)], GPU["quat"])[0]
        assert max_err(host(lin), WIDE["%s/lin_%s" % (robot, link)]) < GPU["jac"] and max_err(host(ang), WIDE["%s/ang_%s" % (robot, link)]) < GPU["jac"]
    for g, d in ((1, 1), (0, 0)):
        tau = m.compute_inverse_dynamics(q, qd, qdd, include_gravity=bool(g), use_damping=bool(d))
        assert np.allclose(host(tau), WIDE["%s/tau_g%d_d%d" % (robot, g, d)], **TOL_TAU), (robot, g, d)
        acc = m.compute_forward_dynamics(q, qd, f, include_gravity=bool(g), use_damping=bool(d))
        assert rel(host(acc), WIDE["%s/acc_g%d_d%d" % (robot, g, d)]) < (1e-4 if robot in ARMS else 1e-3), (robot, g, d)
    H = m.compute_lagrangian_inertia_matrix(q)
    assert np.allclose(host(H), WIDE[robot + "/H"], **TOL_H_REF), (robot, np.abs(host(H) - WIDE[robot + "/H"]).max())
