#!/usr/bin/env python3
"""Generate tests/golden/golden_hvp.npz: SECOND derivatives through the UNMODIFIED reference (torch autograd with
create_graph=True on its CPU path) — what a gradient penalty or a Hessian-vector product asks of the library.

For every case, with random weights w on the outputs and a random direction v on the inputs:
    L = sum(w * outputs(x));   g = dL/dx (create_graph);   s = sum(v * g);
    hvp = ds/dx  = (d2L/dx2) v       and       dsdw = ds/dw = J(x) v
for   fk   x = q,            outputs = pos, quat of the end link            (robot_model.py:197-248)
      jac  x = q,            outputs = lin, ang Jacobians of the end link   (robot_model.py:626-667)
      id   x = (q, qd, qdd), outputs = tau                                  (robot_model.py:305-375)

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_hvp.py
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

# (case, reference urdf, end link, batch)
CASES = [
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", "iiwa_link_ee", 6),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf", "panda_virtual_ee_link", 6),
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf", "link_15.0_tip", 5),
    ("panda", "panda_description/urdf/panda.urdf", "panda_leftfinger", 5),       # (the package's reference_compat=True)
    # a FAST, hard-accelerating state (|qd| up to 50 rad/s, |qdd| up to 100 rad/s^2: torques ~1e4 N m): the second derivatives
    # of inverse dynamics where a fixed small difference step would drown them in the rounding of the first-order launches
    ("panda_no_gripper_fast", "panda_description/urdf/panda_no_gripper.urdf", "panda_virtual_ee_link", 6, 50.0),
]


def main():
    rm = ref_import.import_reference()
    torch.set_num_threads(1)
    out = {}
    for name, rel, link, B, *fast in CASES:
        vel = fast[0] if fast else 1.0
        torch.manual_seed(0)
        np.random.seed(0)
        path = os.path.join(ref_import.reference_data_dir(), rel)
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(path)
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        n = len(lim)
        mk = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, requires_grad=True)
        rnd = lambda *shape: np.random.uniform(-1.0, 1.0, size=shape)
        q = mk(np.random.uniform(lo, hi, size=(B, n)))
        qd = mk(vel * rnd(B, n))
        qdd = mk(2.0 * vel * rnd(B, n))
        vq, vqd, vqdd = (torch.tensor(rnd(B, n), dtype=torch.float32) for _ in range(3))
        out[name + "/link"] = np.array(link)
        for k, t in (("q", q), ("qd", qd), ("qdd", qdd), ("vq", vq), ("vqd", vqd), ("vqdd", vqdd)):
            out["%s/%s" % (name, k)] = t.detach().numpy()

        def second(outputs, weights, xs, vs, tag):
            L = sum((w * o).sum() for w, o in zip(weights, outputs))
            g = torch.autograd.grad(L, xs, create_graph=True)
            s = sum((v * gi).sum() for v, gi in zip(vs, g))
            h = torch.autograd.grad(s, list(xs) + list(weights), allow_unused=True)
            for i, gi in enumerate(g):
                out["%s/%s/g%d" % (name, tag, i)] = gi.detach().numpy()
            for i in range(len(xs)):
                out["%s/%s/hvp%d" % (name, tag, i)] = h[i].numpy()
            for i, w in enumerate(weights):
                out["%s/%s/w%d" % (name, tag, i)] = w.detach().numpy()
                out["%s/%s/dsdw%d" % (name, tag, i)] = h[len(xs) + i].numpy()
            return float(s)

        pos, quat = model.compute_forward_kinematics(q, link)
        s_fk = second((pos, quat), (mk(rnd(B, 3)), mk(rnd(B, 4))), (q,), (vq,), "fk")
        lin, ang = model.compute_endeffector_jacobian(q, link)
        s_jac = second((lin, ang), (mk(rnd(B, 3, n)), mk(rnd(B, 3, n))), (q,), (vq,), "jac")
        tau = model.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
        s_id = second((tau,), (mk(rnd(B, n)),), (q, qd, qdd), (vq, vqd, vqdd), "id")
        print("%-18s B=%d  v.g: fk %.4f  jac %.4f  id %.4f" % (name, B, s_fk, s_jac, s_id))
    np.savez_compressed(os.path.join(HERE, "golden_hvp.npz"), **out)


if __name__ == "__main__":
    main()
