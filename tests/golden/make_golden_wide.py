#!/usr/bin/env python3
"""Generate tests/golden/golden_wide.npz: a WIDER batch than the per-robot fixtures (48 joint states per robot instead of
7, another seed, q over the whole joint range, qd ~ U(+-1), qdd ~ U(+-2), f ~ U(+-1)) through every entry point of the hot
path of the UNMODIFIED reference on its CPU path: FK + Jacobian of the test links, inverse dynamics with and without
gravity / damping, the joint-space inertia matrix and forward dynamics.  float32 in, float32 out, stored as they come.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_wide.py
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

BATCH = 48


def generate(batch, seed_base):
    rm = ref_import.import_reference()
    torch.set_num_threads(1)
    out = {}
    for idx, (name, rel, links) in enumerate(ROBOTS):
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(os.path.join(ref_import.reference_data_dir(), rel))
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        n = len(lim)
        rng = np.random.default_rng(seed_base + idx)
        q = torch.from_numpy((lo + (hi - lo) * rng.random((batch, n))).astype(np.float32))
        qd = torch.from_numpy(rng.uniform(-1.0, 1.0, (batch, n)).astype(np.float32))
        qdd = torch.from_numpy(rng.uniform(-2.0, 2.0, (batch, n)).astype(np.float32))
        f = torch.from_numpy(rng.uniform(-1.0, 1.0, (batch, n)).astype(np.float32))
        for k, v in (("q", q), ("qd", qd), ("qdd", qdd), ("f", f)):
            out["%s/%s" % (name, k)] = v.numpy().copy()
        out[name + "/links"] = np.array(links)
        with torch.no_grad():
            for link in links:
                pos, quat = model.compute_forward_kinematics(q, link)
                lin, ang = model.compute_endeffector_jacobian(q, link)
                for k, v in (("pos", pos), ("quat", quat), ("lin", lin), ("ang", ang)):
                    out["%s/%s_%s" % (name, k, link)] = v.numpy()
            for g, d in ((1, 1), (0, 0)):
                out["%s/tau_g%d_d%d" % (name, g, d)] = model.compute_inverse_dynamics(
                    q, qd, qdd, include_gravity=bool(g), use_damping=bool(d)).numpy()
                # (the reference subtracts the damping torques from its input IN PLACE: pass a copy)
                out["%s/acc_g%d_d%d" % (name, g, d)] = model.compute_forward_dynamics(
                    q, qd, f.clone(), include_gravity=bool(g), use_damping=bool(d)).numpy()
            out[name + "/H"] = model.compute_lagrangian_inertia_matrix(q).numpy()
        print("%-40s n=%2d" % (name, n), flush=True)
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "golden_wide.npz"), **generate(BATCH, 9000))


if __name__ == "__main__":
    main()
