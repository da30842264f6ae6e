#!/usr/bin/env python3
"""Generate tests/golden/golden_hvp_dyn.npz: SECOND derivatives with respect to LEARNABLE LINK PARAMETERS (and the joint state) through
the UNMODIFIED reference — torch autograd with create_graph=True on its CPU path (robot_model.py:305-450, 487-624 with the
parametrisations of rigid_body_params.py; robot_model.py:669-713).  What a meta-learning inner loop or a curvature-aware
optimiser asks of a model with learnable parameters.

For every case, with learnable parameters theta, inputs x, random output weights w and random directions v:
    L = sum(w * outputs(x; theta));   g = dL/d(theta, x) (create_graph);   s = sum(v * g);
    hvp = ds/d(theta, x) = (d2L/d(theta, x)2) v       and       dsdw = ds/dw = J v
for   id    outputs = tau(q, qd, qdd)          x = (q, qd, qdd)         compute_inverse_dynamics      (robot_model.py:305-375)
      mass  outputs = H(q)                     x = (q,)                 compute_lagrangian_inertia_matrix (402-450)
      fd    outputs = qdd(q, qd, f)            x = (q, qd, f)           compute_forward_dynamics      (487-624)
      fk    outputs = pos, quat of the end link, Jacobians   x = (q,)   compute_forward_kinematics / compute_endeffector_jacobian

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_hvp_dyn.py
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

# (case, reference urdf, end link, {link: [parameter names]}, batch)
CASES = [
    ("iiwa7", "kuka_iiwa/urdf/iiwa7.urdf", "iiwa_link_ee",
     {"iiwa_link_2": ["mass", "com", "inertia_mat"], "iiwa_link_4": ["trans", "rot_angles"], "iiwa_link_6": ["mass", "joint_damping"]}, 6),
    ("panda_no_gripper", "panda_description/urdf/panda_no_gripper.urdf", "panda_virtual_ee_link",
     {"panda_link3": ["mass", "com", "inertia_mat", "trans", "rot_angles"], "panda_link7": ["com"]}, 5),
    ("allegro_left", "allegro/urdf/allegro_hand_description_left.urdf", "link_15.0_tip",
     {"link_13.0": ["mass", "com", "trans"], "link_2.0": ["inertia_mat", "rot_angles"]}, 4),
]
SHAPES = {"mass": (1, 1), "joint_damping": (1, 1), "com": (1, 3), "trans": (1, 3), "rot_angles": (1, 3), "inertia_mat": (3, 3)}


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
        keys, params = [], []
        for lname, pnames in learn.items():
            body = model._bodies[model._name_to_idx_map[lname]]
            for pname in pnames:
                # UnconstrainedTensor parametrisations started AT the URDF's value (+ a small offset): the learnable model is a
                # physical one, and the fixture carries the initial values so that the package's model starts at the same point
                holder = body if pname in ("trans", "rot_angles", "joint_damping") else body.inertia
                now = getattr(holder, pname)()
                now = torch.zeros(1) if now is None else now
                init = now.detach().reshape(SHAPES[pname]).clone() + 0.02 * torch.randn(SHAPES[pname])
                if pname == "inertia_mat":
                    init = 0.5 * (init + init.t())
                model.make_link_param_learnable(lname, pname, rbp.UnconstrainedTensor(dim1=SHAPES[pname][0], dim2=SHAPES[pname][1],
                                                                                     init_tensor=init.clone()))
                key = "%s/%s" % (lname, pname)
                keys.append(key)
                out["%s/init/%s" % (name, key)] = init.numpy()
                holder = body if pname in ("trans", "rot_angles", "joint_damping") else body.inertia
                params.append(getattr(holder, pname).param)
        out[name + "/keys"], out[name + "/link"] = np.array(keys), np.array(link)
        lim = model.get_joint_limits()
        lo = np.asarray([j["lower"] for j in lim]); hi = np.asarray([j["upper"] for j in lim])
        n = len(lim)
        mk = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, requires_grad=True)
        rnd = lambda *shape: np.random.uniform(-1.0, 1.0, size=shape)
        q, qd, qdd = mk(np.random.uniform(lo, hi, size=(B, n))), mk(rnd(B, n)), mk(2.0 * rnd(B, n))
        with torch.no_grad():
            f0 = model.compute_inverse_dynamics(q.detach(), qd.detach(), qdd.detach(), include_gravity=True, use_damping=True)
        f = mk(f0.numpy())
        vp = [torch.tensor(rnd(*p.shape), dtype=torch.float32) for p in params]
        # directions on an inertia matrix are SYMMETRIC (the package's inertia-matrix / forward-dynamics kernels read an inertia
        # matrix as the symmetric matrix it physically is; the gradients themselves cover all nine entries)
        vp = [0.5 * (v + v.t()) if key.endswith("inertia_mat") else v for key, v in zip(keys, vp)]
        for k, t in (("q", q), ("qd", qd), ("qdd", qdd), ("f", f)):
            out["%s/%s" % (name, k)] = t.detach().numpy()
        for key, v in zip(keys, vp):
            out["%s/vp/%s" % (name, key)] = v.numpy()

        def second(tag, outputs, xs):
            ws = [mk(rnd(*o.shape)) for o in outputs]
            vx = [torch.tensor(rnd(*x.shape), dtype=torch.float32) for x in xs]
            L = sum((w * o).sum() for w, o in zip(ws, outputs))
            g = torch.autograd.grad(L, params + list(xs), create_graph=True, allow_unused=True)
            g = [gi if gi is not None else torch.zeros_like(t) for gi, t in zip(g, params + list(xs))]
            s = sum((v * gi).sum() for v, gi in zip(vp + vx, g))
            h = torch.autograd.grad(s, params + list(xs) + ws, allow_unused=True)
            h = [hi if hi is not None else torch.zeros_like(t) for hi, t in zip(h, params + list(xs) + ws)]
            for i, key in enumerate(keys):
                out["%s/%s/gp/%s" % (name, tag, key)] = g[i].detach().numpy()
                out["%s/%s/hp/%s" % (name, tag, key)] = h[i].numpy()
            for j in range(len(xs)):
                out["%s/%s/gx%d" % (name, tag, j)] = g[len(keys) + j].detach().numpy()
                out["%s/%s/hx%d" % (name, tag, j)] = h[len(keys) + j].numpy()
                out["%s/%s/vx%d" % (name, tag, j)] = vx[j].numpy()
            for j, w in enumerate(ws):
                out["%s/%s/w%d" % (name, tag, j)] = w.detach().numpy()
                out["%s/%s/dsdw%d" % (name, tag, j)] = h[len(keys) + len(xs) + j].numpy()
            return float(s)

        s_id = second("id", (model.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True),), (q, qd, qdd))
        s_m = second("mass", (model.compute_lagrangian_inertia_matrix(q),), (q,))
        s_fd = second("fd", (model.compute_forward_dynamics(q, qd, f.clone(), include_gravity=True, use_damping=True),), (q, qd, f))
        pos, quat = model.compute_forward_kinematics(q, link)
        lin, ang = model.compute_endeffector_jacobian(q, link)
        s_fk = second("fk", (pos, quat, lin, ang), (q,))
        print("%-18s B=%d  %d parameter tensors   v.g: id %.4f  mass %.4f  fd %.4f  fk %.4f" % (name, B, len(keys), s_id, s_m, s_fd, s_fk))
    np.savez_compressed(os.path.join(HERE, "golden_hvp_dyn.npz"), **out)


if __name__ == "__main__":
    main()
