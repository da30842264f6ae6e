#!/usr/bin/env python3
"""Soak test of the LEARNABLE paths (GPU box): random robots x random sets of learnable link parameters (1 .. 12 links; trans,
rot_angles, mass, com, inertia_mat, joint_damping) x random batch sizes — the gradients of a loss on FK, the Jacobian, inverse
dynamics and (7-DoF arms) fk_mse_loss with respect to every parameter and to q, the HIP path (default own kernels and library kernels)
against the host build of the same ABI.  A wide net for the table kernels (drm_walk_table / _backward, drm_fk_mse_links) and the
reverse-mode kernels' parameter sums; parity itself lives in tests/."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_model, sample_states  # noqa: E402
from differentiable_robot_model_amd.rigid_body_params import (CovParameterized3DInertiaMatrixNet, PositiveScalar,  # noqa: E402
                                                               Symm3DInertiaMatrixNet, SymmPosDef3DInertiaMatrixNet, UnconstrainedTensor)

GPU = "cuda" if torch.cuda.is_available() else "cpu"      # (without a device the same checks run host against host: a dry run of the tool)
ROBOTS = ["panda_no_gripper", "iiwa7", "allegro_left", "panda", "fetch", "jaco", "trifinger_edu", "2link_robot"]
SHAPES = {"trans": (1, 3), "rot_angles": (1, 3), "mass": (1, 1), "com": (1, 3), "inertia_mat": (3, 3), "joint_damping": (1, 1)}
SIZES = [1, 63, 64, 65, 128, 192, 1000, 4096, 4160]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n_models, n_checks, worst = time.time(), 0, 0, {}
while time.time() - t0 < budget:
    robot = ROBOTS[rng.integers(len(ROBOTS))]
    probe = load_model(robot)
    names = [b.name for b in probe._bodies[1:]]
    chosen = [names[i] for i in rng.choice(len(names), size=min(len(names), int(rng.integers(1, 13))), replace=False)]
    picks = [(link, p) for link in chosen for p in SHAPES if rng.random() < 0.4] or [(chosen[0], "trans")]
    seed = int(rng.integers(1 << 30))
    models = []
    # the module of every learnable piece (round 6, ABI 13: the kernels know PositiveScalar and the l[6] inertia-matrix modules; the
    # reference model evaluates them with torch ops and packs their outputs: DRM_TABLE_LINKS=0's path on the host)
    kinds = {(link, p): int(rng.integers(4)) for link, p in picks}
    def module(link, p):
        k = kinds[(link, p)]
        if p in ("mass", "joint_damping") and k >= 2:
            return PositiveScalar(min_val=0.05 * k)
        if p == "inertia_mat" and k >= 1:
            return (Symm3DInertiaMatrixNet(init_param_std=0.05), SymmPosDef3DInertiaMatrixNet(bias=1e-3, init_param_std=0.1),
                    CovParameterized3DInertiaMatrixNet(bias=1e-3, init_param_std=0.1))[k - 1]
        return UnconstrainedTensor(*SHAPES[p])
    for dev, own, links_path in (("cpu", None, False), (GPU, None, True), (GPU, "off", True), ("cpu", None, True)):
        m = load_model(robot, dev)
        m._table_links = links_path
        torch.manual_seed(seed)
        for link, p in picks:
            m.make_link_param_learnable(link, p, module(link, p))
        if own:
            m.own_kernels = own
        models.append(m)
    with torch.no_grad():       # the same parameter values everywhere
        for ps in zip(*(m.parameters() for m in models)):
            for other in ps[1:]:
                other.copy_(ps[0].to(other.device))
    n_models += 1
    cpu = models[0]
    link = list(cpu._name_to_idx_map)[int(rng.integers(1, len(cpu._name_to_idx_map)))]
    for _ in range(3):
        B = int(SIZES[rng.integers(len(SIZES))])
        q, qd, qdd = (torch.from_numpy(a) for a in sample_states(cpu, B, seed=int(rng.integers(1 << 30))))
        want = torch.randn(B, 3, generator=torch.Generator().manual_seed(1)) * 0.2
        arm = cpu._n_dofs == 7 and B % 64 == 0 and len(cpu._bodies) <= 10
        results = []
        for m in models:
            dev = m._device.type
            x = q.to(dev).clone().requires_grad_(True)
            m.zero_grad()
            pos, _ = m.compute_forward_kinematics(x, link)
            lin, ang = m.compute_endeffector_jacobian(x, link)
            tau = m.compute_inverse_dynamics(x, qd.to(dev), qdd.to(dev))
            loss = pos.pow(2).mean() + lin.pow(2).mean() + ang.pow(2).mean() + 1e-3 * tau.pow(2).mean()
            if arm:
                last = list(m._name_to_idx_map)[-1]
                loss = loss + m.fk_mse_loss(x, last, want.to(dev))
            loss.backward()
            results.append([loss.detach().cpu(), x.grad.cpu()] + [p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu() for p in m.parameters()])
        # the learned model where no graph is built (round 6: prepared calls that rebuild the table per launch): the second and third
        # call of every method against the first (the Python path), bit for bit, on every model
        with torch.no_grad():
            for m in models:
                dev = m._device.type
                x, xd, xdd = q.to(dev), qd.to(dev), qdd.to(dev)
                for name, fn in (("fk", lambda: torch.cat(m.compute_forward_kinematics(x, link), dim=1)),
                                 ("jac", lambda: torch.cat(m.compute_endeffector_jacobian(x, link), dim=1)),
                                 ("id", lambda: m.compute_inverse_dynamics(x, xd, xdd)),
                                 ("links", lambda: torch.cat([torch.cat(v, dim=1) for v in m.compute_forward_kinematics_all_links(x).values()], dim=1))):
                    first = fn()
                    for _ in range(2):
                        again = fn()
                        n_checks += 1
                        if not torch.equal(first, again):
                            print("FAIL no_grad repeat call %s differs from the first  robot %s link %s B %d learnable %s dev %s" % (name, robot, link, B, picks, dev))
                            sys.exit(1)
        ref = results[0]
        pnames = [n.replace("_bodies.", "").replace(".param", "") for n, _ in cpu.named_parameters()]
        for tag, got in (("own", results[1]), ("lib", results[2]), ("host links", results[3])):
            for k, (a, b) in enumerate(zip(got, ref)):
                what = ("loss", "grad q")[k] if k < 2 else "grad " + pnames[k - 2].split(".")[-1]
                # (a gradient that cancels analytically — a rotation-invariant loss term with respect to a frame's angles, a mass at a
                # joint's origin — is fp32 noise of the size of its summands: the floor follows the loss)
                err = float((a - b).abs().max()) / max(1e-6, 1e-4 * float(ref[0].abs()), float(b.abs().max()),
                                                       1e-4 * max(float(r.abs().max()) for r in ref[2:]))
                n_checks += 1
                worst[what + " " + tag] = max(worst.get(what + " " + tag, 0.0), err)
                if not err <= 5e-3:
                    print("FAIL %s (%s) %s err %.3e  robot %s link %s B %d learnable %s arm %s" % (what, pnames[k - 2] if k >= 2 else "", tag, err, robot, link, B, picks, arm))
                    np.set_printoptions(precision=6, linewidth=200)
                    for kk in range(len(ref)):
                        print("  %-28s cpu %s\n  %-28s own %s\n  %-28s lib %s" % ((["loss", "q"] + pnames)[kk], ref[kk].reshape(-1)[:9].numpy(), "", results[1][kk].reshape(-1)[:9].numpy(), "", results[2][kk].reshape(-1)[:9].numpy()))
                    print("  parameters:", [(n, p.detach().reshape(-1).numpy().round(4).tolist()) for n, p in cpu.named_parameters()])
                    sys.exit(1)
print("soak (learnable): %d checks over %d models in %.0f s, all within 5e-3 of libdrm_cpu (relative to the largest entry); worst:" %
      (n_checks, n_models, time.time() - t0))
for k in sorted(worst):
    print("  %-22s %.2e" % (k, worst[k]))
