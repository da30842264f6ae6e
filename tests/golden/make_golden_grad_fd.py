#!/usr/bin/env python3
"""Generate tests/golden/golden_grad_fd.npz: gradients of a forward-dynamics loss through the UNMODIFIED reference
(torch autograd through its articulated-body recursion, robot_model.py:487-624, on the CPU path) for learnable link
parameters and for q, qd, f.

The first case is the reference's own example (examples/learn_forward_dynamics_iiwa.py:55-90: learnable `mass`
(PositiveScalar), `com` and `inertia_mat` of iiwa_link_1, include_gravity=True, use_damping=True, loss on the
predicted accelerations); the others add kinematic parameters, damping and a branching tree, with `inertia_mat` through
the reference's SymmPosDef3DInertiaMatrixNet (a physical, symmetric positive-definite inertia): for a NON-symmetric inertia matrix (an UnconstrainedTensor on a link whose
off-diagonal terms matter) the reference's articulated-body recursion and its own RNEA stop describing the same robot
(ID(FD(f)) - f = 57 N m in a probe on panda_link3), so there is no single answer to be faithful to.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_grad_fd.py
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
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", {"iiwa_link_1": ["mass", "com", "inertia_mat"]}, 40),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf",
     {"panda_link3": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "panda_link6": ["mass", "joint_damping"]}, 21),
    ("trifinger_edu", "trifinger_edu_description/trifinger_edu.urdf",
     {"finger_middle_link_120": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"]}, 9),
    # round 3: robots with one long segment — forward dynamics runs the articulated-body kernel there and its gradient the
    # implicit differentiation through it (lambda = H^-1 g by the same kernel, then the RNEA backward); see make_golden_grad_dyn.py
    ("fetch", "fetch_description/urdf/fetch.urdf",
     {"shoulder_lift_link": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"],
      "r_gripper_finger_link": ["mass", "com"]}, 8),
    ("jaco", "kinova_description/urdf/jaco.urdf",
     {"j2n6s300_link_3": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"],
      "j2n6s300_link_finger_tip_2": ["mass", "com"]}, 8),
    ("panda", "panda_description/urdf/panda.urdf",
     {"panda_link4": ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"],
      "panda_leftfinger": ["mass", "com"]}, 9),
    ("iiwa7_allegro", "kuka_iiwa/urdf/iiwa7_allegro.urdf",
     {"iiwa_link_3": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "link_13.0": ["mass", "com", "joint_damping"]}, 6),
]


WELL_SCALED = ("fetch", "jaco", "panda", "iiwa7_allegro")


def parametrization(rbp, pname, case):
    if pname == "inertia_mat" and case != "iiwa7":
        return rbp.SymmPosDef3DInertiaMatrixNet()
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
                model.make_link_param_learnable(link, pname, parametrization(rbp, pname, name))
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        n = len(lim)
        mk = lambda a: torch.tensor(a, dtype=torch.float32, requires_grad=True)
        q = mk(np.random.uniform(lo, hi, size=(B, n)))
        qd = mk(np.random.uniform(-1.0, 1.0, size=(B, n)))
        if name in WELL_SCALED:
            # torques that produce accelerations of order one: the inverse dynamics of the ground-truth model for
            # qdd ~ U(-2, 2) (random torques of +-2 Nm throw a 10 g fingertip link to 1e5 rad/s^2, where an fp32 gradient —
            # the reference's included — is rounding noise); the loss comes from the learnable model's random parameters
            with torch.no_grad():
                qdd0 = torch.tensor(np.random.uniform(-2.0, 2.0, size=(B, n)), dtype=torch.float32)
                f0 = gt.compute_inverse_dynamics(q.detach(), qd.detach(), qdd0, include_gravity=True, use_damping=True)
            f = mk(f0.numpy().copy())
        else:
            f = mk(np.random.uniform(-2.0, 2.0, size=(B, n)))
        with torch.no_grad():
            # the reference subtracts the damping torques from its `f` argument in place (robot_model.py:515-521)
            want = gt.compute_forward_dynamics(q.detach(), qd.detach(), f.detach().clone(), include_gravity=True, use_damping=True)
        f_in = f.clone()  # keeps the leaf `f` intact under the in-place update
        qdd = model.compute_forward_dynamics(q, qd, f_in, include_gravity=True, use_damping=True)
        loss = torch.nn.functional.mse_loss(qdd, want)
        loss.backward()
        out[name + "/q"], out[name + "/qd"], out[name + "/f"] = q.detach().numpy(), qd.detach().numpy(), f.detach().numpy()
        out[name + "/grad_q"], out[name + "/grad_qd"], out[name + "/grad_f"] = q.grad.numpy(), qd.grad.numpy(), f.grad.numpy()
        out[name + "/qdd"], out[name + "/want"] = qdd.detach().numpy(), want.numpy()
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
        print("%-20s B=%3d loss=%.5f  %d parameter tensors  |grad_q|max %.3g" % (name, B, loss.item(), len(keys), np.abs(q.grad.numpy()).max()))
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "golden_grad_fd.npz"), **generate(CASES))


if __name__ == "__main__":
    main()
