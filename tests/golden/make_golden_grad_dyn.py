#!/usr/bin/env python3
"""Generate tests/golden/golden_grad_dyn.npz: gradients of an inverse-dynamics loss through the UNMODIFIED reference
(torch autograd on its CPU path) for learnable dynamics / kinematics link parameters and for q, qd, qdd.

Cases mirror the reference's dynamics-learning example (examples/learn_dynamics_iiwa.py:49-96: learnable `mass`
(PositiveScalar), `inertia_mat` (UnconstrainedTensor 3x3) and `trans` of iiwa_link_1, loss on the predicted torques)
and add `com`, `rot_angles`, `joint_damping`, and branching trees.  Only links with a moving joint get learnable
kinematic parameters.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_grad_dyn.py
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

# (case, reference urdf, {link: [parameter names]}, batch)
CASES = [
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf",
     {"iiwa_link_1": ["mass", "inertia_mat", "trans"], "iiwa_link_4": ["com", "rot_angles", "joint_damping"]}, 48),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf",
     {"panda_link3": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "panda_link7": ["mass", "com"]}, 21),
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf",
     {"link_1.0": ["mass", "com", "trans"], "link_14.0": ["inertia_mat", "rot_angles", "joint_damping"],
      "link_15.0": ["mass", "trans"]}, 19),
    ("trifinger_edu", "trifinger_edu_description/trifinger_edu.urdf",
     {"finger_middle_link_120": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"]}, 9),
    # round 3: the robots with ONE long segment (an arm carrying a gripper / a hand, a mobile manipulator) — they run the
    # persistent backward kernels with HBM-parked records (PARK_HBM), which the four robots above never reach.  The
    # reference turns every non-fixed joint into a revolute one (robot_model.py:122-126): the fixtures of fetch / panda hold
    # THAT behaviour (the package's `reference_compat=True`)
    ("fetch", "fetch_description/urdf/fetch.urdf",
     {"shoulder_lift_link": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"],
      "torso_lift_link": ["mass", "trans"], "r_gripper_finger_link": ["mass", "com"]}, 11),
    ("jaco", "kinova_description/urdf/jaco.urdf",
     {"j2n6s300_link_3": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"],
      "j2n6s300_link_finger_tip_2": ["mass", "com", "trans"]}, 9),
    ("panda", "panda_description/urdf/panda.urdf",
     {"panda_link4": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"],
      "panda_leftfinger": ["mass", "com", "trans"]}, 10),
    ("iiwa7_allegro", "kuka_iiwa/urdf/iiwa7_allegro.urdf",
     {"iiwa_link_3": ["mass", "com", "inertia_mat", "trans", "rot_angles"],
      "link_13.0": ["mass", "com", "trans", "joint_damping"]}, 7),
]


def parametrization(rbp, pname):
    if pname == "mass":
        return rbp.PositiveScalar()
    if pname == "joint_damping":
        return rbp.UnconstrainedScalar()
    if pname == "inertia_mat":
        return rbp.UnconstrainedTensor(dim1=3, dim2=3)
    return rbp.UnconstrainedTensor(dim1=1, dim2=3)


def generate(cases):
    """The fixture dictionary for `cases` (also driven by make_golden_tiles.py with 192-row batches)."""
    rm = ref_import.import_reference()
    import differentiable_robot_model.rigid_body_params as rbp
    torch.set_num_threads(1)
    out = {}
    for name, rel, learn, B in cases:
        torch.manual_seed(0)
        np.random.seed(0)
        path = os.path.join(ref_import.reference_data_dir(), rel)
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(path)
            gt = rm.DifferentiableRobotModel(path)
        for link, pnames in learn.items():
            for pname in pnames:
                model.make_link_param_learnable(link, pname, parametrization(rbp, pname))
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        n = len(lim)
        mk = lambda a: torch.tensor(a, dtype=torch.float32, requires_grad=True)
        q = mk(np.random.uniform(lo, hi, size=(B, n)))
        qd = mk(np.random.uniform(-1.0, 1.0, size=(B, n)))
        qdd = mk(np.random.uniform(-2.0, 2.0, size=(B, n)))
        with torch.no_grad():
            want = gt.compute_inverse_dynamics(q.detach(), qd.detach(), qdd.detach(), include_gravity=True, use_damping=True)
        tau = model.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
        loss = torch.nn.functional.mse_loss(tau, want)
        loss.backward()
        out[name + "/q"], out[name + "/qd"], out[name + "/qdd"] = q.detach().numpy(), qd.detach().numpy(), qdd.detach().numpy()
        out[name + "/grad_q"], out[name + "/grad_qd"], out[name + "/grad_qdd"] = q.grad.numpy(), qd.grad.numpy(), qdd.grad.numpy()
        out[name + "/tau"], out[name + "/want"] = tau.detach().numpy(), want.numpy()
        out[name + "/loss"] = np.asarray(loss.item(), np.float64)
        keys = []
        for link, pnames in learn.items():
            body = model._bodies[model._name_to_idx_map[link]]
            for pname in pnames:
                mod = getattr(body if pname in ("trans", "rot_angles", "joint_damping") else body.inertia, pname)
                for k, p in mod.named_parameters():
                    key = "%s/%s/%s" % (link, pname, k)
                    out["%s/init/%s" % (name, key)] = p.detach().numpy()
                    out["%s/grad/%s" % (name, key)] = p.grad.numpy()
                    keys.append(key)
        out[name + "/keys"] = np.array(keys)
        print("%-20s B=%3d loss=%.5f  %d parameter tensors" % (name, B, loss.item(), len(keys)))
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "golden_grad_dyn.npz"), **generate(CASES))


if __name__ == "__main__":
    main()
