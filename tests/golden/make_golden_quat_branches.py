#!/usr/bin/env python3
"""Generate tests/golden/golden_quat_branches.npz: the quaternion of compute_forward_kinematics ON the branch boundaries of
the reference's get_quaternion (spatial_vector_algebra.py:108-136), recorded from the UNMODIFIED reference.

The algorithm switches case on  t = tr(R) + 1 > 1  and, otherwise, on which diagonal entry is largest (R11 > R00, then
R22 > R_ii); a select-based kernel may legitimately take the other case only when rounding moves R across a boundary, and
the two cases give the same rotation with (possibly) the opposite sign.  For Panda and iiwa7 this script finds joint
configurations (bisection on one joint, fp64 oracle) whose end-effector rotation sits at a signed distance
delta in {+-1e-3, +-1e-4, +-3e-5, +-1e-5, +-3e-6, +-1e-6} from each boundary and stores q, the boundary kind, delta (re-measured
in fp64 at the float32 q) and the reference's (pos, quat).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_quat_branches.py
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import ref_import  # noqa: E402

ROBOTS = [("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf", "panda_virtual_ee_link"),
          ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", "iiwa_link_ee")]
DELTAS = [1e-3, 1e-4, 3e-5, 1e-5, 3e-6, 1e-6]


def boundary_value(R, kind):
    """Signed distance to a case boundary: > 0 on one side, < 0 on the other."""
    if kind == "trace":      # t > 1  <=>  tr R > 0
        return R[0, 0] + R[1, 1] + R[2, 2]
    if kind == "yx":         # R11 > R00 (only consulted when tr R <= 0)
        return R[1, 1] - R[0, 0]
    if kind == "zx":         # R22 > R00 (when R11 <= R00)
        return R[2, 2] - R[0, 0]
    return R[2, 2] - R[1, 1]  # "zy": R22 > R11 (when R11 > R00)


def applicable(R, kind):
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if kind == "trace":
        return True
    if tr > -0.05:            # the diagonal tests only matter in the t <= 1 region, stay clear of the trace boundary
        return False
    if kind == "yx":
        return R[2, 2] < min(R[0, 0], R[1, 1]) - 0.05      # and away from the second test
    if kind == "zx":
        return R[1, 1] < R[0, 0] - 0.05
    return R[1, 1] > R[0, 0] + 0.05


def main():
    from helpers import load_model
    from oracle import Oracle
    rm = ref_import.import_reference()
    torch.set_num_threads(1)
    out = {}
    for name, rel, link in ROBOTS:
        m = load_model(name)
        orc = Oracle(m._spec)
        idx = m._name_to_idx_map[link]
        lim = m.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])

        def rot(qv):
            R, _ = orc.fk_all_poses(np.asarray(qv, np.float64).reshape(1, -1), np.float64)
            return R[0, idx]

        rng = np.random.default_rng(2024)
        qs, kinds, deltas = [], [], []
        for kind in ("trace", "yx", "zx", "zy"):
            for delta in DELTAS:
                for sign in (+1.0, -1.0):
                    target = sign * delta
                    for _attempt in range(4000):
                        q0 = lo + (hi - lo) * rng.random(len(lim))
                        j = int(rng.integers(len(lim)))
                        a, b = lo[j], hi[j]
                        f = lambda x: boundary_value(rot(np.concatenate([q0[:j], [x], q0[j + 1:]])), kind) - target
                        xs = np.linspace(a, b, 9)
                        fs = [f(x) for x in xs]
                        br = [(xs[i], xs[i + 1]) for i in range(8) if fs[i] * fs[i + 1] < 0]
                        if not br:
                            continue
                        x0, x1 = br[0]
                        for _ in range(80):
                            xm = 0.5 * (x0 + x1)
                            if f(x0) * f(xm) <= 0:
                                x1 = xm
                            else:
                                x0 = xm
                        q = q0.copy(); q[j] = 0.5 * (x0 + x1)
                        q32 = q.astype(np.float32)
                        R = rot(q32.astype(np.float64))
                        got = boundary_value(R, kind)
                        if not applicable(R, kind) or abs(got - target) > 0.3 * delta or got * target <= 0:
                            continue
                        qs.append(q32); kinds.append(kind); deltas.append(got)
                        break
                    else:
                        raise SystemExit("no configuration for %s %s %g" % (name, kind, target))
        q = torch.tensor(np.stack(qs))
        path = os.path.join(ref_import.reference_data_dir(), rel)
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(path)
        pos, quat = model.compute_forward_kinematics(q, link)
        out[name + "/q"] = q.numpy(); out[name + "/kind"] = np.array(kinds); out[name + "/delta"] = np.asarray(deltas)
        out[name + "/pos"] = pos.detach().numpy(); out[name + "/quat"] = quat.detach().numpy()
        out[name + "/link"] = np.array(link)
        print(name, len(qs), "configurations; |delta| from %.1e to %.1e" % (np.abs(deltas).min(), np.abs(deltas).max()))
    np.savez_compressed(os.path.join(HERE, "golden_quat_branches.npz"), **out)


if __name__ == "__main__":
    main()
