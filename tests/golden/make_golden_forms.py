#!/usr/bin/env python3
"""Generate tests/golden/golden_forms.npz: the reference's OWN parameter modules in the loop (ABI 13's forms).

The UNMODIFIED reference on its CPU path, learnable links whose mass / joint damping are `PositiveScalar`s and whose inertia matrix is
one of the l[6] modules (rigid_body_params.py:26-43 PositiveScalar, :252-339 CovParameterized3DInertiaMatrixNet, :342-384
SymmPosDef3DInertiaMatrixNet, :387-404 Symm3DInertiaMatrixNet), centres of mass / trans free tensors — what
examples/learn_dynamics_iiwa.py:49-96 and the L4DC notebook train.  Per case: the RAW parameters of every module (so that this
package's modules start at the same point), joint states, random output weights w, and

    tau = compute_inverse_dynamics(q, qd, qdd, gravity, damping)          (robot_model.py:305-375)
    pos = compute_forward_kinematics(q, end link)[0]                      (robot_model.py:223-248)
    L   = sum(w_tau * tau) + sum(w_pos * pos)
    dL / d (every raw parameter), dL / d (q, qd, qdd)                      torch autograd through the modules

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_forms.py
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

# (case, reference urdf, end link, {link: {parameter: module}}, batch)
CASES = [
    ("iiwa7_spd", "kuka_iiwa/urdf/iiwa7.urdf", "iiwa_link_ee",
     {"iiwa_link_%d" % k: {"mass": "PositiveScalar", "com": "UnconstrainedTensor", "inertia_mat": "SymmPosDef3DInertiaMatrixNet"}
      for k in range(1, 8)}, 70),
    ("iiwa7_cov", "kuka_iiwa/urdf/iiwa7.urdf", "iiwa_link_ee",
     {"iiwa_link_2": {"mass": "PositiveScalar", "inertia_mat": "CovParameterized3DInertiaMatrixNet", "joint_damping": "PositiveScalar"},
      "iiwa_link_5": {"inertia_mat": "CovParameterized3DInertiaMatrixNet", "com": "UnconstrainedTensor", "trans": "UnconstrainedTensor"},
      "iiwa_link_7": {"mass": "PositiveScalar"}}, 33),
    ("panda_symm", "panda_description/urdf/panda_no_gripper.urdf", "panda_virtual_ee_link",
     {"panda_link4": {"mass": "PositiveScalar", "inertia_mat": "Symm3DInertiaMatrixNet", "rot_angles": "UnconstrainedTensor"},
      "panda_link6": {"inertia_mat": "SymmPosDef3DInertiaMatrixNet", "joint_damping": "PositiveScalar"}}, 64),
    ("allegro_mixed", "allegro/urdf/allegro_hand_description_left.urdf", "link_15.0_tip",
     {"link_13.0": {"mass": "PositiveScalar", "inertia_mat": "CovParameterized3DInertiaMatrixNet"},
      "link_2.0": {"inertia_mat": "Symm3DInertiaMatrixNet", "com": "UnconstrainedTensor"},
      "link_9.0": {"inertia_mat": "UnconstrainedTensor", "mass": "UnconstrainedTensor"}}, 40),
]
SHAPES = {"mass": (1, 1), "joint_damping": (1, 1), "com": (1, 3), "trans": (1, 3), "rot_angles": (1, 3), "inertia_mat": (3, 3)}


def make_module(rbp, pname, kind, k):
    """A module of the reference, deterministically initialised (the fixture carries the raw values)."""
    if kind == "PositiveScalar":
        return rbp.PositiveScalar(min_val=0.01 * (1 + k % 3), init_param_std=0.7)
    if kind == "UnconstrainedTensor":
        return rbp.UnconstrainedTensor(dim1=SHAPES[pname][0], dim2=SHAPES[pname][1], init_std=0.1)
    if kind == "Symm3DInertiaMatrixNet":
        return rbp.Symm3DInertiaMatrixNet(init_param_std=0.1)
    if kind == "SymmPosDef3DInertiaMatrixNet":
        return rbp.SymmPosDef3DInertiaMatrixNet(bias=1e-3 * (1 + k % 2), init_param_std=0.3)
    if kind == "CovParameterized3DInertiaMatrixNet":
        return rbp.CovParameterized3DInertiaMatrixNet(bias=1e-3 * (1 + k % 2), init_param_std=0.3)
    raise KeyError(kind)


def main():
    rm = ref_import.import_reference()
    import differentiable_robot_model.rigid_body_params as rbp
    torch.set_num_threads(1)
    out = {}
    for name, rel, link, learn, B in CASES:
        torch.manual_seed(0)
        np.random.seed(0)
        path = os.path.join(ref_import.reference_data_dir(), rel)
        with contextlib.redirect_stdout(io.StringIO()):
            model = rm.DifferentiableRobotModel(path)
        keys, kinds, consts, params = [], [], [], []
        k = 0
        for lname, pieces in learn.items():
            for pname, kind in pieces.items():
                module = make_module(rbp, pname, kind, k)
                k += 1
                model.make_link_param_learnable(lname, pname, module)
                (raw,) = list(module.parameters())
                keys.append("%s/%s" % (lname, pname))
                kinds.append(kind)
                consts.append(float(getattr(module, "_min_val", getattr(module, "spd_3d_inertia_mat_diag_bias",
                                                                        getattr(module, "spd_3d_cov_inertia_mat_diag_bias", 0.0)))))
                params.append(raw)
                out["%s/raw/%s/%s" % (name, lname, pname)] = raw.detach().numpy().copy()
                holder = model._bodies[model._name_to_idx_map[lname]]
                holder = holder if pname in ("trans", "rot_angles", "joint_damping") else holder.inertia
                out["%s/value/%s/%s" % (name, lname, pname)] = getattr(holder, pname)().detach().numpy().copy()
        out[name + "/keys"], out[name + "/kinds"], out[name + "/consts"] = np.array(keys), np.array(kinds), np.array(consts, np.float64)
        out[name + "/link"] = np.array(link)
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        n = len(lim)
        mk = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, requires_grad=True)
        rnd = lambda *shape: np.random.uniform(-1.0, 1.0, size=shape)
        q, qd, qdd = mk(np.random.uniform(lo, hi, size=(B, n))), mk(rnd(B, n)), mk(2.0 * rnd(B, n))
        w_tau, w_pos = torch.tensor(rnd(B, n), dtype=torch.float32), torch.tensor(rnd(B, 3), dtype=torch.float32)
        tau = model.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
        pos, _ = model.compute_forward_kinematics(q, link)
        L = (w_tau * tau).sum() + (w_pos * pos).sum()
        grads = torch.autograd.grad(L, params + [q, qd, qdd], allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(t) for g, t in zip(grads, params + [q, qd, qdd])]
        for key, g in zip(keys, grads):
            out["%s/grad/%s" % (name, key)] = g.numpy()
        for tag, t in (("q", q), ("qd", qd), ("qdd", qdd), ("w_tau", w_tau), ("w_pos", w_pos), ("tau", tau), ("pos", pos)):
            out["%s/%s" % (name, tag)] = t.detach().numpy()
        for tag, g in zip(("gq", "gqd", "gqdd"), grads[len(keys):]):
            out["%s/%s" % (name, tag)] = g.numpy()
        print("%-14s B=%d  %d parameter modules  L = %.5f  max |grad| %.3e" % (name, B, len(keys), float(L),
                                                                                max(float(g.abs().max()) for g in grads[:len(keys)])))
    np.savez_compressed(os.path.join(HERE, "golden_forms.npz"), **out)


if __name__ == "__main__":
    main()
