#!/usr/bin/env python3
"""Generate tests/golden/golden_mass.npz: joint-space inertia matrices from the UNMODIFIED reference
(`compute_lagrangian_inertia_matrix`, robot_model.py:402-450 = n + 1 inverse-dynamics passes) on its CPU path, for the
robots and joint states of the golden fixtures (the reference's own test, tests/test_kinematics_dynamics.py:379-409,
compares against pybullet's calculateMassMatrix, which is not installable here).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_mass.py
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
        q = torch.from_numpy(g["fast_q"])
        with torch.no_grad():
            out[name + "/H_g1_d1"] = model.compute_lagrangian_inertia_matrix(q, include_gravity=True, use_damping=True).numpy()
            out[name + "/H_g0_d0"] = model.compute_lagrangian_inertia_matrix(q, include_gravity=False, use_damping=False).numpy()
        print("%-40s n=%2d |H - H^T| max %.2e" % (name, q.shape[1],
              np.abs(out[name + "/H_g1_d1"] - out[name + "/H_g1_d1"].transpose(0, 2, 1)).max()))
    np.savez_compressed(os.path.join(HERE, "golden_mass.npz"), **out)


if __name__ == "__main__":
    main()
