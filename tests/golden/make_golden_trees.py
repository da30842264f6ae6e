#!/usr/bin/env python3
"""Generate tests/golden/golden_trees.npz: RANDOM kinematic trees — robots nobody shipped — through the UNMODIFIED reference on
its CPU path (the random-tree tests had no reference side).  Twenty trees of 4 .. 18 links with random
branching, sub-trees behind fixed joints at the root, fixed joints in between, revolute AND prismatic joints about +-x / y / z (what
the reference models: every non-fixed joint an axis-aligned revolute one, robot_model.py:122-126), random frames, masses, centres of
mass, inertias and dampings.  70 joint states per tree (one 64-row tile + a ragged tail): FK of every link, the Jacobian of the last
link, inverse dynamics with and without gravity / damping, the joint-space inertia matrix, forward dynamics — and the reference's
AUTOGRAD through five scalar losses on them (input gradients: q for FK, the Jacobian and the inertia matrix; q, qd, qdd for inverse
dynamics; q, qd, f for forward dynamics).  The URDF text travels in the file.  float32 in, float32 out, stored as they come.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_trees.py
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

N_TREES, BATCH = 20, 70
AXES = ["1 0 0", "0 1 0", "0 0 1", "-1 0 0", "0 -1 0", "0 0 -1"]


def tree_urdf(seed):
    """A random tree the reference accepts (links in parent-before-child order, axis-aligned joint axes)."""
    rng = np.random.default_rng(91000 + seed)
    n_links = int(rng.integers(4, 19))
    chainy = rng.random()
    fixed_root = rng.random() < 0.4
    out = ['<?xml version="1.0"?>', '<robot name="gtree%d">' % seed, '  <link name="base"/>']
    names, movable = ["base"], 0
    for i in range(n_links):
        name = "l%d" % i
        parent = names[-1] if rng.random() < chainy else names[int(rng.integers(len(names)))]
        if fixed_root and rng.random() < 0.3:
            parent = "base"
        A = rng.standard_normal((3, 3)) * 0.03
        I = A @ A.T + np.eye(3) * 0.002
        m, c = 0.05 + rng.random() * 0.8, rng.standard_normal(3) * 0.04
        out.append('  <link name="%s"><inertial><origin xyz="%.5f %.5f %.5f" rpy="0 0 0"/><mass value="%.5f"/>'
                   '<inertia ixx="%.6f" ixy="%.6f" ixz="%.6f" iyy="%.6f" iyz="%.6f" izz="%.6f"/></inertial></link>'
                   % (name, c[0], c[1], c[2], m, I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))
        xyz, rpy = rng.standard_normal(3) * 0.08, rng.standard_normal(3) * 0.7
        u = rng.random()
        last = i == n_links - 1 and movable == 0
        kind = "fixed" if (not last and (u < 0.2 or (parent == "base" and fixed_root))) else ("prismatic" if u < 0.35 else "revolute")
        if kind == "fixed":
            out.append('  <joint name="j%d" type="fixed"><parent link="%s"/><child link="%s"/>'
                       '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/></joint>' % (i, parent, name, *xyz, *rpy))
        else:
            movable += 1
            out.append('  <joint name="j%d" type="%s"><parent link="%s"/><child link="%s"/>'
                       '<origin xyz="%.5f %.5f %.5f" rpy="%.5f %.5f %.5f"/><axis xyz="%s"/>'
                       '<limit effort="10" lower="-2.5" upper="2.5" velocity="3"/><dynamics damping="%.3f"/></joint>'
                       % (i, kind, parent, name, *xyz, *rpy, AXES[int(rng.integers(6))], rng.random() * 0.2))
        names.append(name)
    out.append("</robot>")
    return "\n".join(out) + "\n"


def parametrization(rbp, pname):
    if pname == "mass":
        return rbp.PositiveScalar()
    if pname == "joint_damping":
        return rbp.UnconstrainedScalar()
    if pname == "inertia_mat":
        return rbp.UnconstrainedTensor(dim1=3, dim2=3)
    return rbp.UnconstrainedTensor(dim1=1, dim2=3)


def learnable_links(model):
    """Which parameters a tree's learnable run opens: all six of its FIRST moving link, mass / com / trans of its LAST moving link
    (only links with a moving joint get learnable kinematic parameters: SURVEY.md Appendix B, Q2)."""
    moving = [model._bodies[i].name for i in model._controlled_joints]
    learn = {moving[0]: ["mass", "com", "inertia_mat", "trans", "rot_angles", "joint_damping"]}
    if len(moving) > 1:
        learn[moving[-1]] = ["mass", "com", "trans"]
    return learn


def generate():
    rm = ref_import.import_reference()
    import differentiable_robot_model.rigid_body_params as rbp
    torch.set_num_threads(1)
    out = {"n_trees": np.array(N_TREES)}
    with tempfile.TemporaryDirectory() as tmp:
        for t in range(N_TREES):
            text = tree_urdf(t)
            path = os.path.join(tmp, "gtree%d.urdf" % t)
            with open(path, "w") as f:
                f.write(text)
            with contextlib.redirect_stdout(io.StringIO()):
                model = rm.DifferentiableRobotModel(path)
            n = model._n_dofs
            rng = np.random.default_rng(5000 + t)
            q = torch.from_numpy(rng.uniform(-2.5, 2.5, (BATCH, n)).astype(np.float32))
            qd = torch.from_numpy(rng.uniform(-1.0, 1.0, (BATCH, n)).astype(np.float32))
            qdd = torch.from_numpy(rng.uniform(-2.0, 2.0, (BATCH, n)).astype(np.float32))
            f = torch.from_numpy(rng.uniform(-1.0, 1.0, (BATCH, n)).astype(np.float32))
            key = "t%d/" % t
            out[key + "urdf"] = np.array(text)
            for k, v in (("q", q), ("qd", qd), ("qdd", qdd), ("f", f)):
                out[key + k] = v.numpy().copy()
            names = model.get_link_names()
            out[key + "links"] = np.array(names)
            with torch.no_grad():
                poses = model.compute_forward_kinematics_all_links(q)
                # (the root's pose comes back as one row, robot_model.py:197-221)
                out[key + "pos"] = np.stack([np.broadcast_to(poses[nm][0].reshape(-1, 3).numpy(), (BATCH, 3)) for nm in names], 1)
                out[key + "quat"] = np.stack([np.broadcast_to(poses[nm][1].reshape(-1, 4).numpy(), (BATCH, 4)) for nm in names], 1)
                lin, ang = model.compute_endeffector_jacobian(q, names[-1])
                out[key + "lin"], out[key + "ang"] = lin.numpy(), ang.numpy()
                for g, d in ((1, 1), (0, 0)):
                    out[key + "tau_g%d_d%d" % (g, d)] = model.compute_inverse_dynamics(
                        q, qd, qdd, include_gravity=bool(g), use_damping=bool(d)).numpy()
                out[key + "acc"] = model.compute_forward_dynamics(q, qd, f.clone(), include_gravity=True, use_damping=True).numpy()
                out[key + "H"] = model.compute_lagrangian_inertia_matrix(q).numpy()
            # the reference's autograd through the same entry points: input gradients of five scalar losses (weights w_* fixed
            # pseudo-random tensors, so every output entry carries its own cotangent)
            w3 = torch.from_numpy(rng.standard_normal((BATCH, 3)).astype(np.float32))
            wj = torch.from_numpy(rng.standard_normal((BATCH, 3, n)).astype(np.float32))
            wn = torch.from_numpy(rng.standard_normal((BATCH, n)).astype(np.float32))
            wh = torch.from_numpy(rng.standard_normal((BATCH, n, n)).astype(np.float32))
            for k, v in (("w3", w3), ("wj", wj), ("wn", wn), ("wh", wh)):
                out[key + k] = v.numpy().copy()
            def grads(fn, *xs):
                xs = [x.clone().requires_grad_(True) for x in xs]
                y = fn(*xs)
                if y.requires_grad:      # (a link behind fixed joints only does not depend on q: zero gradients)
                    y.backward()
                return [x.grad.numpy() if x.grad is not None else np.zeros(tuple(x.shape), np.float32) for x in xs]
            (out[key + "g_fk_q"],) = grads(lambda a: (model.compute_forward_kinematics(a, names[-1])[0] * w3).sum(), q)
            (out[key + "g_jac_q"],) = grads(lambda a: sum((j * wj).sum() for j in model.compute_endeffector_jacobian(a, names[-1])), q)
            out[key + "g_id_q"], out[key + "g_id_qd"], out[key + "g_id_qdd"] = grads(
                lambda a, b, c: (model.compute_inverse_dynamics(a, b, c, include_gravity=True, use_damping=True) * wn).sum(), q, qd, qdd)
            (out[key + "g_h_q"],) = grads(lambda a: (model.compute_lagrangian_inertia_matrix(a) * wh).sum(), q)
            out[key + "g_fd_q"], out[key + "g_fd_qd"], out[key + "g_fd_f"] = grads(
                lambda a, b, c: (model.compute_forward_dynamics(a, b, c.clone(), include_gravity=True, use_damping=True) * wn).sum(), q, qd, f)
            # ... and PARAMETER gradients: a second model of the same tree with learnable link parameters (freshly drawn under
            # seed t), the inverse-dynamics loss of the reference's learning example against the untouched tree's torques, and an
            # FK loss on the last link; gradients of both with respect to every parameter tensor, and the ID loss's input gradients
            torch.manual_seed(t)
            with contextlib.redirect_stdout(io.StringIO()):
                learn_model = rm.DifferentiableRobotModel(path)
            learn = learnable_links(learn_model)
            for link, pnames in learn.items():
                for pname in pnames:
                    learn_model.make_link_param_learnable(link, pname, parametrization(rbp, pname))
            ql, qdl, qddl = (x.clone().requires_grad_(True) for x in (q, qd, qdd))
            tau_l = learn_model.compute_inverse_dynamics(ql, qdl, qddl, include_gravity=True, use_damping=True)
            loss = torch.nn.functional.mse_loss(tau_l, torch.from_numpy(out[key + "tau_g1_d1"]))
            loss.backward()
            out[key + "p_loss"] = np.asarray(loss.item(), np.float64)
            out[key + "p_tau"] = tau_l.detach().numpy()
            out[key + "p_g_q"], out[key + "p_g_qd"], out[key + "p_g_qdd"] = ql.grad.numpy(), qdl.grad.numpy(), qddl.grad.numpy()
            keys = []
            for link, pnames in learn.items():
                body = learn_model._bodies[learn_model._name_to_idx_map[link]]
                for pname in pnames:
                    mod = getattr(body if pname in ("trans", "rot_angles", "joint_damping") else body.inertia, pname)
                    for k, p_ in mod.named_parameters():
                        pk = "%s/%s/%s" % (link, pname, k)
                        out[key + "p_init/" + pk] = p_.detach().numpy().copy()
                        out[key + "p_grad_id/" + pk] = p_.grad.numpy().copy()
                        keys.append(pk)
            out[key + "p_keys"] = np.array(keys)
            learn_model.zero_grad()
            fk_loss = (learn_model.compute_forward_kinematics(q, names[-1])[0] * w3).sum()
            if fk_loss.requires_grad:
                fk_loss.backward()
            for link, pnames in learn.items():
                body = learn_model._bodies[learn_model._name_to_idx_map[link]]
                for pname in pnames:
                    if pname not in ("trans", "rot_angles"):
                        continue
                    for k, p_ in getattr(body, pname).named_parameters():
                        out[key + "p_grad_fk/%s/%s/%s" % (link, pname, k)] = (p_.grad.numpy().copy() if p_.grad is not None
                                                                           else np.zeros(tuple(p_.shape), np.float32))
            print("tree %2d: %2d links, %2d DoF, %d learnable tensors" % (t, len(names), n, len(keys)), flush=True)
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "golden_trees.npz"), **generate())


if __name__ == "__main__":
    main()
