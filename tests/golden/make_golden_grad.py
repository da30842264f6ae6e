#!/usr/bin/env python3
"""Generate tests/golden/golden_grad.npz: gradients of an FK position loss through the UNMODIFIED reference
(torch autograd on its CPU path) for learnable `trans` / `rot_angles` link parameters and for q.

Cases mirror the reference's kinematics-learning example (examples/learn_kinematics_of_iiwa.py:25-61:
`make_link_param_learnable(link, "trans" | "rot_angles", UnconstrainedTensor(1, 3))`, MSE loss on the
end-effector position) on a chain (iiwa = BASELINE config 5, Panda) and on branching trees with x / y / z joint
axes (Allegro, TriFinger).  Only links with a moving joint are made learnable: the reference ignores learnable
parameters of fixed-joint links.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_grad.py
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

# (case, reference urdf, target links, learnable links, batch)
CASES = [
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", ["iiwa_link_ee"], ["iiwa_link_1"], 64),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf", ["panda_virtual_ee_link"],
     ["panda_link4", "panda_link7"], 33),
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf",
     ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"], ["link_1.0", "link_13.0", "link_15.0"], 40),
    ("trifinger_edu", "trifinger_edu_description/trifinger_edu.urdf",
     ["finger_tip_link_0", "finger_tip_link_120", "finger_tip_link_240"], ["finger_middle_link_120"], 17),
]


def generate(cases):
    """The fixture dictionary for `cases` (also driven by make_golden_tiles.py with 192-row batches)."""
    rm = ref_import.import_reference()
    from differentiable_robot_model.rigid_body_params import UnconstrainedTensor
    torch.set_num_threads(1)
    out = {}
    for name, rel, targets, learn, B in cases:
        torch.manual_seed(0)
        np.random.seed(0)
        path = os.path.join(ref_import.reference_data_dir(), rel)
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(path)
            gt = rm.DifferentiableRobotModel(path)
        for link in learn:  # rigid_body_params.py:46-56: init N(0, 0.1^2)
            model.make_link_param_learnable(link, "trans", UnconstrainedTensor(dim1=1, dim2=3))
            model.make_link_param_learnable(link, "rot_angles", UnconstrainedTensor(dim1=1, dim2=3))
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        q = torch.tensor(np.random.uniform(lo, hi, size=(B, len(lim))), dtype=torch.float32, requires_grad=True)
        loss = 0.0
        for t in targets:
            with torch.no_grad():
                want, _ = gt.compute_forward_kinematics(q.detach(), t)
            pos, _ = model.compute_forward_kinematics(q, t)
            out["%s/pos/%s" % (name, t)] = pos.detach().numpy()
            out["%s/want/%s" % (name, t)] = want.numpy()
            loss = loss + torch.nn.functional.mse_loss(pos, want)
        loss.backward()
        out[name + "/q"] = q.detach().numpy()
        out[name + "/grad_q"] = q.grad.numpy()
        out[name + "/loss"] = np.asarray(loss.item(), np.float64)
        out[name + "/targets"] = np.array(targets)
        out[name + "/learnable"] = np.array(learn)
        for link in learn:
            body = model._bodies[model._name_to_idx_map[link]]
            for pname in ("trans", "rot_angles"):
                p = getattr(body, pname).param
                out["%s/init/%s/%s" % (name, link, pname)] = p.detach().numpy()
                out["%s/grad/%s/%s" % (name, link, pname)] = p.grad.numpy()
        print("%-20s B=%3d loss=%.6f |grad_q|max=%.3e" % (name, B, loss.item(), q.grad.abs().max().item()))
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "golden_grad.npz"), **generate(CASES))


if __name__ == "__main__":
    main()
