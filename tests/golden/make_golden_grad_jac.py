#!/usr/bin/env python3
"""Generate tests/golden/golden_grad_jac.npz: gradients of a loss on the end-effector Jacobian (and position) through
the UNMODIFIED reference (torch autograd through compute_endeffector_jacobian, robot_model.py:626-667, and
compute_forward_kinematics on its CPU path) for q and for learnable `trans` / `rot_angles`.

Only links with a moving joint get learnable kinematic parameters.
Run in the build container only (needs /root/reference):   python tests/golden/make_golden_grad_jac.py
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

# (case, reference urdf, end-effector link, {link: [parameter names]}, batch)
CASES = [
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", "iiwa_link_ee", {"iiwa_link_1": ["trans", "rot_angles"], "iiwa_link_5": ["trans"]}, 33),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf", "panda_virtual_ee_link",
     {"panda_link3": ["trans", "rot_angles"], "panda_link7": ["rot_angles"]}, 17),
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf", "link_15.0_tip",
     {"link_13.0": ["trans", "rot_angles"], "link_1.0": ["trans"]}, 11),
]


def main():
    rm = ref_import.import_reference()
    import differentiable_robot_model.rigid_body_params as rbp
    torch.set_num_threads(1)
    out = {}
    for name, rel, ee, learn, B in CASES:
        torch.manual_seed(0)
        np.random.seed(0)
        path = os.path.join(ref_import.reference_data_dir(), rel)
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(path)
        for link, pnames in learn.items():
            for pname in pnames:
                model.make_link_param_learnable(link, pname, rbp.UnconstrainedTensor(dim1=1, dim2=3))
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        n = len(lim)
        q = torch.tensor(np.random.uniform(lo, hi, size=(B, n)), dtype=torch.float32, requires_grad=True)
        w_lin = torch.tensor(np.random.uniform(-1, 1, size=(B, 3, n)), dtype=torch.float32)
        w_ang = torch.tensor(np.random.uniform(-1, 1, size=(B, 3, n)), dtype=torch.float32)
        w_pos = torch.tensor(np.random.uniform(-1, 1, size=(B, 3)), dtype=torch.float32)
        lin, ang = model.compute_endeffector_jacobian(q, ee)
        pos, _ = model.compute_forward_kinematics(q, ee)
        # linear + quadratic terms so that the gradients depend on the values
        loss = (w_lin * lin).sum() + (w_ang * ang).sum() + (lin ** 2).sum() + 0.5 * (ang ** 2 * w_ang).sum() + (w_pos * pos).sum()
        loss.backward()
        out[name + "/q"], out[name + "/grad_q"] = q.detach().numpy(), q.grad.numpy()
        out[name + "/w_lin"], out[name + "/w_ang"], out[name + "/w_pos"] = w_lin.numpy(), w_ang.numpy(), w_pos.numpy()
        out[name + "/lin"], out[name + "/ang"], out[name + "/pos"] = lin.detach().numpy(), ang.detach().numpy(), pos.detach().numpy()
        out[name + "/loss"] = np.asarray(loss.item(), np.float64)
        out[name + "/ee"] = np.array(ee)
        keys = []
        for link, pnames in learn.items():
            body = model._bodies[model._name_to_idx_map[link]]
            for pname in pnames:
                for k, p in getattr(body, pname).named_parameters():
                    key = "%s/%s/%s" % (link, pname, k)
                    out["%s/init/%s" % (name, key)] = p.detach().numpy()
                    if p.grad is None:
                        print("  no gradient reaches", name, key, "(off the chain)")
                    out["%s/grad/%s" % (name, key)] = p.grad.numpy() if p.grad is not None else np.zeros_like(p.detach().numpy())
                    keys.append(key)
        out[name + "/keys"] = np.array(keys)
        print("%-20s B=%3d loss=%.5g  %d parameter tensors  |grad_q|max %.3g" % (name, B, loss.item(), len(keys), np.abs(q.grad.numpy()).max()))
    np.savez_compressed(os.path.join(HERE, "golden_grad_jac.npz"), **out)


if __name__ == "__main__":
    main()
