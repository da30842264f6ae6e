#!/usr/bin/env python3
"""Generate tests/golden/golden_grad_mass.npz: gradients of a loss on the joint-space inertia matrix through the
UNMODIFIED reference (torch autograd through its n + 1 inverse-dynamics passes, robot_model.py:402-450, CPU path)
for learnable link parameters and for q.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_grad_mass.py
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
from make_golden_grad_fd import parametrization  # noqa: E402  (SymmPosDef inertia except for the iiwa example case)

CASES = [
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", {"iiwa_link_1": ["mass", "com", "inertia_mat"], "iiwa_link_5": ["trans", "rot_angles"]}, 12),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf",
     {"panda_link3": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "panda_link6": ["mass"]}, 9),
    ("trifinger_edu", "trifinger_edu_description/trifinger_edu.urdf",
     {"finger_middle_link_120": ["mass", "com", "inertia_mat", "trans", "rot_angles"]}, 5),
    # round 3: robots with one long segment (persistent kernels, HBM-parked records); see make_golden_grad_dyn.py
    ("fetch", "fetch_description/urdf/fetch.urdf",
     {"shoulder_lift_link": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "r_gripper_finger_link": ["mass"]}, 4),
    ("jaco", "kinova_description/urdf/jaco.urdf",
     {"j2n6s300_link_3": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "j2n6s300_link_finger_tip_2": ["mass", "com"]}, 4),
    ("panda", "panda_description/urdf/panda.urdf",
     {"panda_link4": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "panda_leftfinger": ["mass", "com"]}, 5),
    ("iiwa7_allegro", "kuka_iiwa/urdf/iiwa7_allegro.urdf",
     {"iiwa_link_3": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "link_13.0": ["mass", "com"]}, 3),
]


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
        q = torch.tensor(np.random.uniform(lo, hi, size=(B, n)), dtype=torch.float32, requires_grad=True)
        weight = torch.tensor(np.random.uniform(0.5, 1.5, size=(B, n, n)), dtype=torch.float32)
        with torch.no_grad():
            want = gt.compute_lagrangian_inertia_matrix(q.detach())
        H = model.compute_lagrangian_inertia_matrix(q)
        loss = (weight * (H - want) ** 2).mean()   # a non-symmetric weight: both triangles carry their own gradient
        loss.backward()
        out[name + "/q"], out[name + "/grad_q"] = q.detach().numpy(), q.grad.numpy()
        out[name + "/H"], out[name + "/want"], out[name + "/weight"] = H.detach().numpy(), want.numpy(), weight.numpy()
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
        print("%-20s B=%3d loss=%.5g  %d parameter tensors  |grad_q|max %.3g" % (name, B, loss.item(), len(keys), np.abs(q.grad.numpy()).max()))
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "golden_grad_mass.npz"), **generate(CASES))


if __name__ == "__main__":
    main()
