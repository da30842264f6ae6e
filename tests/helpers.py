"""Shared test helpers (robot list, model construction, sampling, tolerances)."""
import contextlib
import io
import os

import numpy as np
import torch

import differentiable_robot_model_amd as drm
from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel, robot_description_folder

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# the reference's test matrix (reference tests/test_kinematics_dynamics.py:19-52): (robot, test links)
REFERENCE_TEST_MATRIX = [
    ("fetch_arm_no_gripper_small_damping", ["virtual_ee_link"]),
    ("2link_robot", ["endEffector"]),
    ("iiwa7", ["iiwa_link_ee"]),
    ("panda_no_gripper", ["panda_virtual_ee_link"]),
    ("allegro_left_small_damping", ["link_11.0_tip", "link_7.0_tip", "link_3.0_tip", "link_15.0_tip"]),
    ("trifinger_edu", ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"]),
    ("jaco_clean", ["j2n6s300_link_ee"]),
]
EXTRA_ROBOTS = [
    ("allegro_left", ["link_3.0_tip", "link_15.0_tip"]),
    ("fetch_arm_no_gripper", ["virtual_ee_link"]),
    ("iiwa7_allegro", ["link_3.0_tip", "link_15.0_tip"]),
    ("panda", ["panda_hand", "panda_rightfinger"]),
    ("jaco", ["j2n6s300_end_effector"]),
]
GOLDEN_ROBOTS = REFERENCE_TEST_MATRIX + EXTRA_ROBOTS
ALL_ROBOTS = sorted(f[:-5] for f in os.listdir(robot_description_folder) if f.endswith(".urdf"))
# batch shapes of the reference's tests (test_kinematics_dynamics.py:54-61)
REFERENCE_BATCH_SHAPES = [tuple(), (1,), (3,), (6,), (7,)]

# fp32 tolerances of the path: the reference's own fp32 noise floor is ~2e-7
TOL_POS = dict(atol=2e-6, rtol=0)       # metres
TOL_QUAT = dict(atol=2e-6, rtol=0)
TOL_JAC = dict(atol=2e-6, rtol=0)
TOL_TAU = dict(atol=2e-5, rtol=2e-5)    # N m


def urdf_path(name):
    return os.path.join(robot_description_folder, name + ".urdf")


def load_model(name, device="cpu", reference_compat=True):
    """Models are loaded the way the reference models their joints by default (prismatic = revolute, robot_model.py:122-126):
    most tests compare with the reference's goldens or with the oracle's restatement of it.  The tests of the correct
    prismatic / general-axis models pass reference_compat=False."""
    with contextlib.redirect_stdout(io.StringIO()):
        return DifferentiableRobotModel(urdf_path(name), device=device, reference_compat=reference_compat)


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, "golden_%s.npz" % name), allow_pickle=False)


def sample_states(model, B, seed=0, vel=1.0, acc=2.0):
    """q ~ U(lower, upper), qd ~ U(+-vel), qdd ~ U(+-acc) as float32 numpy arrays."""
    lim = model.get_joint_limits()
    lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
    rng = np.random.default_rng(seed)
    n = len(lim)
    q = (lo + (hi - lo) * rng.random((B, n))).astype(np.float32)
    qd = ((rng.random((B, n)) * 2 - 1) * vel).astype(np.float32)
    qdd = ((rng.random((B, n)) * 2 - 1) * acc).astype(np.float32)
    return q, qd, qdd


def quat_branch_margin(b):
    """Distance of a rotation (given as the reference quaternion b = xyzw) from the nearest case boundary that the
    reference's get_quaternion actually evaluates for it (spatial_vector_algebra.py:117-128): tr R = 0 first; in the
    else-branch the diagonal comparisons R11 > R00 and R22 > R_ii.  From b alone: tr R = 4 w^2 - 1, R00 = 1 - 2 (y^2 + z^2), ..."""
    b = np.asarray(b, np.float64)
    b = b / np.linalg.norm(b, axis=-1, keepdims=True)
    x, y, z, w = (b[..., i] for i in range(4))
    tr = 4.0 * w * w - 1.0
    d = np.stack([1 - 2 * (y * y + z * z), 1 - 2 * (x * x + z * z), 1 - 2 * (x * x + y * y)], -1)
    m01 = np.abs(d[..., 1] - d[..., 0])
    dii = np.maximum(d[..., 0], d[..., 1])
    m2 = np.abs(d[..., 2] - dii)
    return np.where(tr > 0, np.abs(tr), np.minimum(np.abs(tr), np.minimum(m01, m2)))


def quat_close(a, b, atol, margin=1e-5):
    """Quaternions agree WITH THE REFERENCE'S SIGN (the sign is part of the reference's output: its case rule makes one
    component positive) on every row whose rotation is further than `margin` from a case boundary of
    spatial_vector_algebra.py:117-128; on rows within the margin (fp32 rounding of R is ~3e-7, so either case is a correct
    answer there) the global sign may differ.  Returns (ok, number of sign flips) — flips can only be rows within the margin."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    near = (quat_branch_margin(b) <= margin)[..., None]
    sgn = np.sign((a * b).sum(-1, keepdims=True))
    sgn[sgn == 0] = 1.0
    sgn = np.where(near, sgn, 1.0)
    return np.abs(a * sgn - b).max() <= atol, int((sgn < 0).sum())


def max_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def load_golden_grad():
    return np.load(os.path.join(GOLDEN_DIR, "golden_grad.npz"), allow_pickle=False)


def load_golden_mass():
    return np.load(os.path.join(GOLDEN_DIR, "golden_mass.npz"), allow_pickle=False)
