#!/usr/bin/env python3
"""Generate tests/golden/golden_fd.npz: joint accelerations from the UNMODIFIED reference's articulated-body
`compute_forward_dynamics` (robot_model.py:487-624) on its CPU path, for the robots and joint states of the golden
fixtures and seeded random joint torques (the reference's own test, tests/test_kinematics_dynamics.py:411-511, compares
against a pybullet simulation step, which is not installable here).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_fd.py
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
from make_golden import ROBOTS  # noqa: E402


def main():
    rm = ref_import.import_reference()
    torch.set_num_threads(1)
    out = {}
    for name, rel, _links in ROBOTS:
        g = np.load(os.path.join(HERE, "golden_%s.npz" % name))
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(os.path.join(ref_import.reference_data_dir(), rel))
        q, qd = torch.from_numpy(g["fast_q"]), torch.from_numpy(g["fast_qd"])
        rng = np.random.default_rng(7)
        f = torch.from_numpy(rng.uniform(-1.0, 1.0, size=q.shape).astype(np.float32))
        out[name + "/f"] = f.numpy().copy()
        for grav, damp in ((1, 0), (1, 1), (0, 0)):
            with torch.no_grad():  # the reference subtracts the damping torques from its input IN PLACE: pass a copy
                qdd = model.compute_forward_dynamics(q, qd, f.clone(), include_gravity=bool(grav), use_damping=bool(damp))
            out["%s/qdd_g%d_d%d" % (name, grav, damp)] = qdd.numpy()
        print("%-40s n=%2d |qdd| max %.3g" % (name, q.shape[1], np.abs(out[name + "/qdd_g1_d0"]).max()))
    np.savez_compressed(os.path.join(HERE, "golden_fd.npz"), **out)


if __name__ == "__main__":
    main()
