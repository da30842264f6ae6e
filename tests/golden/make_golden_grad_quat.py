#!/usr/bin/env python3
"""Generate tests/golden/golden_grad_quat.npz: gradients of losses that read the QUATERNION of compute_forward_kinematics,
through the UNMODIFIED reference (torch autograd on its CPU path).

The reference's get_quaternion (spatial_vector_algebra.py:108-136) assembles the quaternion from entries of the rotation
matrix inside autograd (only the 0.5 / math.sqrt(t) normalisation is a Python float), so an orientation loss has gradients
with respect to q and to learnable `trans` / `rot_angles`.  Cases: a pure orientation loss and a pose loss (position +
orientation) on a chain (iiwa7) and on a tree (Allegro, two fingertips; x / y / z joint axes), plus BASELINE configuration 5
at its FULL size — iiwa7, `iiwa_link_1.trans` and `.rot_angles` learnable, batch 16 384, FK(EE position) MSE loss — of which
only the seed of q, the loss and the six summed parameter-gradient scalars are stored.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_grad_quat.py
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

CASES = [
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", ["iiwa_link_ee"], ["iiwa_link_1", "iiwa_link_5"], 48),
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf", ["link_3.0_tip", "link_15.0_tip"],
     ["link_1.0", "link_13.0"], 31),
]
CONFIG5_BATCH, CONFIG5_SEED = 16384, 555


def build(rm, rel, learn):
    from differentiable_robot_model.rigid_body_params import UnconstrainedTensor
    path = os.path.join(ref_import.reference_data_dir(), rel)
    with contextlib.redirect_stdout(io.StringIO()):
        model = rm.DifferentiableRobotModel(path)
        gt = rm.DifferentiableRobotModel(path)
    for link in learn:  # rigid_body_params.py:46-56: init N(0, 0.1^2)
        model.make_link_param_learnable(link, "trans", UnconstrainedTensor(dim1=1, dim2=3))
        model.make_link_param_learnable(link, "rot_angles", UnconstrainedTensor(dim1=1, dim2=3))
    return model, gt


def main():
    rm = ref_import.import_reference()
    torch.set_num_threads(1)
    out = {}
    for name, rel, targets, learn, B in CASES:
        for mode in ("quat", "pose"):
            torch.manual_seed(0)
            np.random.seed(0)
            model, gt = build(rm, rel, learn)
            lim = model.get_joint_limits()
            lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
            q = torch.tensor(np.random.uniform(lo, hi, size=(B, len(lim))), dtype=torch.float32, requires_grad=True)
            loss = 0.0
            key = "%s/%s" % (name, mode)
            for t in targets:
                with torch.no_grad():
                    wp, wq = gt.compute_forward_kinematics(q.detach(), t)
                pos, quat = model.compute_forward_kinematics(q, t)
                out["%s/want_pos/%s" % (key, t)] = wp.numpy(); out["%s/want_quat/%s" % (key, t)] = wq.numpy()
                out["%s/quat/%s" % (key, t)] = quat.detach().numpy()
                loss = loss + torch.nn.functional.mse_loss(quat, wq)
                if mode == "pose":
                    loss = loss + torch.nn.functional.mse_loss(pos, wp)
            loss.backward()
            out[key + "/q"] = q.detach().numpy()
            out[key + "/grad_q"] = q.grad.numpy()
            out[key + "/loss"] = np.asarray(loss.item(), np.float64)
            out[key + "/targets"] = np.array(targets)
            out[key + "/learnable"] = np.array(learn)
            for link in learn:
                body = model._bodies[model._name_to_idx_map[link]]
                for pname in ("trans", "rot_angles"):
                    p = getattr(body, pname).param
                    out["%s/init/%s/%s" % (key, link, pname)] = p.detach().numpy()
                    out["%s/grad/%s/%s" % (key, link, pname)] = p.grad.numpy()
            print("%-24s B=%3d loss=%.6f |grad_q|max=%.3e" % (key, B, loss.item(), q.grad.abs().max().item()))
    # ---- BASELINE configuration 5 at full size: only scalars are stored, q is regenerated from the seed by the test
    torch.manual_seed(0)
    model, gt = build(rm, "kuka_iiwa/urdf/iiwa7.urdf", ["iiwa_link_1"])
    lim = model.get_joint_limits()
    lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
    qn = (lo + (hi - lo) * np.random.default_rng(CONFIG5_SEED).random((CONFIG5_BATCH, len(lim)))).astype(np.float32)
    q = torch.tensor(qn, requires_grad=True)
    with torch.no_grad():
        want, _ = gt.compute_forward_kinematics(q.detach(), "iiwa_link_ee")
    pos, _ = model.compute_forward_kinematics(q, "iiwa_link_ee")
    loss = torch.nn.functional.mse_loss(pos, want)
    loss.backward()
    body = model._bodies[model._name_to_idx_map["iiwa_link_1"]]
    out["config5/seed"] = np.asarray(CONFIG5_SEED); out["config5/batch"] = np.asarray(CONFIG5_BATCH)
    out["config5/loss"] = np.asarray(loss.item(), np.float64)
    out["config5/q_checksum"] = np.asarray(float(qn.astype(np.float64).sum()), np.float64)
    out["config5/grad_q_abs_sum"] = np.asarray(float(q.grad.double().abs().sum()), np.float64)
    out["config5/pos_sum"] = pos.detach().double().sum(0).numpy()
    for pname in ("trans", "rot_angles"):
        p = getattr(body, pname).param
        out["config5/init/" + pname] = p.detach().numpy()
        out["config5/grad/" + pname] = p.grad.numpy()
    print("config5 B=%d loss=%.6f" % (CONFIG5_BATCH, loss.item()), out["config5/grad/trans"], out["config5/grad/rot_angles"])
    np.savez_compressed(os.path.join(HERE, "golden_grad_quat.npz"), **out)


if __name__ == "__main__":
    main()
