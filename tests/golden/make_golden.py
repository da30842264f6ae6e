#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

The reference stores no golden vectors: its tests compare live against pybullet
3.0.8 (reference tests/test_kinematics_dynamics.py:233-377), which is not
installable here.  These fixtures pin the oracle (and through it the HIP
kernels) to the reference's own CPU torch path instead.

For every robot of the reference's test matrix (tests/test_kinematics_dynamics.py:19-52)
plus the remaining shipped URDFs, with the reference's seeding and sampling
(`random.seed(0); np.random.seed(0); torch.manual_seed(0)`, q ~ U(lower, upper),
qd ~ U(+-0.01 v_lim), qdd = 10 U(+-0.01 v_lim); test_kinematics_dynamics.py:162-190)
and a second, faster state set (qd ~ U(+-1), qdd ~ U(+-2)) it records:
  inputs, FK (pos, quat) and Jacobians of the test links, RNEA torques for the
  four gravity/damping flag combinations, the per-link parameters the
  reference's URDF loader produced, and its R_fixed matrices.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
"""
import contextlib
import io
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

# (fixture name, reference urdf path, test links) — first 7 rows = the reference's test matrix
ROBOTS = [
    ("fetch_arm_no_gripper_small_damping", "fetch_description/urdf/fetch_arm_no_gripper_small_damping.urdf",
     ["virtual_ee_link"]),
    ("2link_robot", "2link_robot.urdf", ["endEffector"]),
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", ["iiwa_link_ee"]),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf", ["panda_virtual_ee_link"]),
    ("allegro_left_small_damping", "allegro/urdf/allegro_hand_description_left_small_damping.urdf",
     ["link_11.0_tip", "link_7.0_tip", "link_3.0_tip", "link_15.0_tip"]),
    ("trifinger_edu", "trifinger_edu_description/trifinger_edu.urdf",
     ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"]),
    ("jaco_clean", "kinova_description/urdf/jaco_clean.urdf", ["j2n6s300_link_ee"]),
    # not in the reference's test matrix, shipped in diff_robot_data/
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf", ["link_3.0_tip", "link_15.0_tip"]),
    ("fetch_arm_no_gripper", "fetch_description/urdf/fetch_arm_no_gripper.urdf", ["virtual_ee_link"]),
    ("iiwa7_allegro", "kuka_iiwa/urdf/iiwa7_allegro.urdf", ["link_3.0_tip", "link_15.0_tip"]),
    ("panda", "panda_description/urdf/panda.urdf", ["panda_hand", "panda_rightfinger"]),
    ("jaco", "kinova_description/urdf/jaco.urdf", ["j2n6s300_end_effector"]),
]
BATCH = 7


def sample(model, fast):
    lim = model.get_joint_limits()
    lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
    vl = np.asarray([0.01 * j["velocity"] for j in lim])
    n = len(lim)
    q = np.random.uniform(low=lo, high=hi, size=(BATCH, n))
    if fast:
        qd = np.random.uniform(-1.0, 1.0, size=(BATCH, n)); qdd = np.random.uniform(-2.0, 2.0, size=(BATCH, n))
    else:
        qd = np.random.uniform(low=-vl, high=vl, size=(BATCH, n))
        qdd = 10.0 * np.random.uniform(low=-vl, high=vl, size=(BATCH, n))
    # the reference's tests go numpy float64 -> python lists -> torch.Tensor (float32)
    return tuple(torch.Tensor(a.tolist()) for a in (q, qd, qdd))


def main():
    rm = ref_import.import_reference()
    from differentiable_robot_model.spatial_vector_algebra import x_rot, y_rot, z_rot
    torch.set_num_threads(1)
    for name, rel, links in ROBOTS:
        path = os.path.join(ref_import.reference_data_dir(), rel)
        out = {}
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(path)
        L = len(model._bodies)
        out["link_names"] = np.array([b.name for b in model._bodies])
        out["test_links"] = np.array(links)
        out["controlled_joints"] = np.asarray(model._controlled_joints, np.int32)
        parent = [-1] + [model._name_to_idx_map[model._urdf_model.get_name_of_parent_body(b.name)]
                         for b in model._bodies[1:]]
        out["parent"] = np.asarray(parent, np.int32)
        out["rot_angles"] = np.stack([b.rot_angles().reshape(3).numpy() for b in model._bodies])
        out["trans"] = np.stack([b.trans().reshape(3).numpy() for b in model._bodies])
        out["joint_axis"] = np.stack([b.joint_axis.reshape(3).numpy() for b in model._bodies])
        out["joint_damping"] = np.asarray(
            [0.0 if b.joint_damping() is None else float(b.joint_damping()) for b in model._bodies], np.float32)
        out["mass"] = np.stack([b.inertia.mass().reshape(1).numpy() for b in model._bodies])[:, 0]
        out["com"] = np.stack([b.inertia.com().reshape(3).numpy() for b in model._bodies])
        out["inertia_mat"] = np.stack([b.inertia.inertia_mat().reshape(9).numpy() for b in model._bodies])
        fixed = []
        for b in model._bodies:  # rigid_body.py:138-143
            r = b.rot_angles()
            fixed.append(((z_rot(r[0, 2]) @ y_rot(r[0, 1])) @ x_rot(r[0, 0])).reshape(9).numpy())
        out["R_fixed"] = np.stack(fixed)
        lim = model.get_joint_limits()
        for k in ("lower", "upper", "velocity", "effort"):
            out["limit_" + k] = np.asarray([j[k] for j in lim], np.float64)

        for tag, fast in (("slow", False), ("fast", True)):
            random.seed(0); np.random.seed(0); torch.manual_seed(0)  # test_kinematics_dynamics.py:162-164
            q, qd, qdd = sample(model, fast)
            out[tag + "_q"], out[tag + "_qd"], out[tag + "_qdd"] = q.numpy(), qd.numpy(), qdd.numpy()
            for link in links:
                with contextlib.redirect_stdout(io.StringIO()):
                    fresh = rm.DifferentiableRobotModel(path)  # recursive FK needs a fresh model
                with torch.no_grad():
                    pos, quat = model.compute_forward_kinematics(q, link)
                    lin, ang = model.compute_endeffector_jacobian(q, link)
                    pos_r, quat_r = fresh.compute_forward_kinematics(q, link, recursive=True)
                out["%s_pos_%s" % (tag, link)] = pos.numpy(); out["%s_quat_%s" % (tag, link)] = quat.numpy()
                out["%s_lin_%s" % (tag, link)] = lin.numpy(); out["%s_ang_%s" % (tag, link)] = ang.numpy()
                out["%s_posrec_%s" % (tag, link)] = pos_r.numpy(); out["%s_quatrec_%s" % (tag, link)] = quat_r.numpy()
            for g in (0, 1):
                for d in (0, 1):
                    with torch.no_grad():
                        tau = model.compute_inverse_dynamics(q, qd, qdd, include_gravity=bool(g), use_damping=bool(d))
                    out["%s_tau_g%d_d%d" % (tag, g, d)] = tau.numpy()
            # unbatched call (batch shape ()) — tensor_check strips the batch dim, robot_model.py:52-61
            with torch.no_grad():
                p1, r1 = model.compute_forward_kinematics(q[0], links[0])
            out[tag + "_pos_unbatched"], out[tag + "_quat_unbatched"] = p1.numpy(), r1.numpy()
        np.savez_compressed(os.path.join(HERE, "golden_%s.npz" % name), **out)
        print("%-40s L=%2d n=%2d  %d arrays" % (name, L, model._n_dofs, len(out)))


if __name__ == "__main__":
    main()
