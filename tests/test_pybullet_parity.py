"""The reference's REAL oracle is pybullet 3.0.8 (/root/reference/tests/test_kinematics_dynamics.py:233-511, tests/requirements.txt:2),
which is not installable in this image (no network): every other test here pins parity to the reference's CPU torch path instead.
This module is the hook that closes that last indirection wherever pybullet IS importable — it is skipped as a whole otherwise:
the same robots, links and sampling as the reference's suite, the same tolerances (1e-5 abs on poses and Jacobians, 1e-5 on
torques at small velocities), the models of THIS package on the CPU (libdrm_cpu.so) and, under -m gpu, on the HIP device."""
import numpy as np
import pytest
import torch

p = pytest.importorskip("pybullet", reason="pybullet (the reference's oracle, tests/requirements.txt:2) is not installed")

from helpers import REFERENCE_TEST_MATRIX, load_model, urdf_path     # noqa: E402

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


class Bullet(object):
    """One robot in a DIRECT-mode client, set up as the reference does (test_kinematics_dynamics.py:118-159): fixed base, gravity
    -9.81 z, joint damping taken from the model, velocity limits lifted, no linear / angular damping."""

    def __init__(self, robot, model):
        self.cid = p.connect(p.DIRECT)
        self.rid = p.loadURDF(urdf_path(robot), basePosition=[0, 0, 0], useFixedBase=True,
                              flags=p.URDF_USE_INERTIA_FROM_FILE, physicsClientId=self.cid)
        p.setGravity(0, 0, -9.81, physicsClientId=self.cid)
        self.joints = [j - 1 for j in model._controlled_joints]      # (pybullet numbers links from -1 = base)
        damping = model.get_joint_damping() if hasattr(model, "get_joint_damping") else None
        for link in range(p.getNumJoints(self.rid, physicsClientId=self.cid)):
            p.changeDynamics(self.rid, link, linearDamping=0.0, angularDamping=0.0, maxJointVelocity=200, physicsClientId=self.cid)
        self.damping = damping

    def set(self, q, qd):
        for j, a, v in zip(self.joints, q, qd):
            p.resetJointState(self.rid, j, targetValue=float(a), targetVelocity=float(v), physicsClientId=self.cid)

    def close(self):
        p.disconnect(self.cid)


def samples(model, B, seed=0):
    """q ~ U(limits), qd ~ U(+-0.01 vmax), qdd ~ 10 U(+-0.01 vmax) (test_kinematics_dynamics.py:162-190)."""
    rng = np.random.default_rng(seed)
    lim = model.get_joint_limits()
    lo, hi = (np.asarray([j[k] for j in lim]) for k in ("lower", "upper"))
    vmax = 0.01 * np.asarray([j["velocity"] for j in lim])
    q = rng.uniform(lo, hi, (B, len(lim)))
    qd = rng.uniform(-vmax, vmax, (B, len(lim)))
    qdd = 10.0 * rng.uniform(-vmax, vmax, (B, len(lim)))
    return q, qd, qdd


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("robot,links", REFERENCE_TEST_MATRIX)
def test_against_pybullet(robot, links, device):
    model = load_model(robot, device)
    sim = Bullet(robot, model)
    try:
        B, n = 7, model._n_dofs
        q, qd, qdd = samples(model, B)
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=model._device)
        tau = model.compute_inverse_dynamics(t(q), t(qd), t(qdd), include_gravity=True, use_damping=False).cpu().numpy()
        H = model.compute_lagrangian_inertia_matrix(t(q)).cpu().numpy()
        for name in links:
            idx = model._name_to_idx_map[name] - 1
            pos, quat = (x.cpu().numpy() for x in model.compute_forward_kinematics(t(q), name))
            lin, ang = (x.cpu().numpy() for x in model.compute_endeffector_jacobian(t(q), name))
            for b in range(B):
                sim.set(q[b], qd[b])
                state = p.getLinkState(sim.rid, idx, physicsClientId=sim.cid)
                want_q = np.asarray(state[5])
                assert np.abs(pos[b] - np.asarray(state[4])).max() <= 1e-5, (robot, name)
                assert min(np.abs(quat[b] - want_q).max(), np.abs(quat[b] + want_q).max()) <= 1e-5, (robot, name)
                jl, ja = p.calculateJacobian(sim.rid, idx, [0, 0, 0], list(q[b]), [0.0] * n, [0.0] * n, physicsClientId=sim.cid)
                assert np.abs(lin[b] - np.asarray(jl)).max() <= 1e-5 and np.abs(ang[b] - np.asarray(ja)).max() <= 1e-5, (robot, name)
        for b in range(B):
            sim.set(q[b], qd[b])
            want = np.asarray(p.calculateInverseDynamics(sim.rid, list(q[b]), list(qd[b]), list(qdd[b]), physicsClientId=sim.cid))
            assert np.abs(tau[b] - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), robot
            Hb = np.asarray(p.calculateMassMatrix(sim.rid, list(q[b]), physicsClientId=sim.cid))
            assert np.abs(H[b] - Hb).max() <= 1e-4 * max(1.0, np.abs(Hb).max()), robot
    finally:
        sim.close()
