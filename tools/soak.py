#!/usr/bin/env python3
"""Soak test (GPU box): random robots x random batch sizes x misaligned slices x every public entry point, the HIP path (default own
kernels AND library kernels) against the host build of the same ABI (libdrm_cpu.so, itself pinned to the oracle and the reference in
tests/).  Not a parity test — tests/ hold those — but a wide net for kernel-selection bugs (tile tails, size switches, alignment)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_model, sample_states  # noqa: E402

ROBOTS = ["panda_no_gripper", "iiwa7", "allegro_left", "panda", "fetch", "jaco", "trifinger_edu", "2link_robot", "fetch_arm_no_gripper",
          "iiwa7_allegro"]
SIZES = [1, 3, 63, 64, 65, 127, 128, 129, 1000, 4096, 4097, 65536 + 64, 131072, 131072 + 64 + 7, 2048 * 64, 2048 * 64 + 5]
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n_checks, worst = time.time(), 0, {}


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float(((a - b).abs() / (1.0 + b.abs())).max()) if a.numel() else 0.0


def note(what, err, tol, ctx):
    global n_checks
    n_checks += 1
    worst[what] = max(worst.get(what, 0.0), err)
    if not err <= tol:
        print("FAIL %s err %.3e > %.1e  %s" % (what, err, tol, ctx))
        sys.exit(1)


models = {}
while time.time() - t0 < budget:
    robot = ROBOTS[rng.integers(len(ROBOTS))]
    if robot not in models:
        cpu, own, lib = load_model(robot), load_model(robot, "cuda"), load_model(robot, "cuda")
        lib.own_kernels = "off"
        models[robot] = (cpu, own, lib)
    cpu, own, lib = models[robot]
    B = int(SIZES[rng.integers(len(SIZES))]) if rng.random() < 0.7 else int(rng.integers(1, 9000))
    off = int(rng.integers(0, 2))                      # a misaligned view (one row in)
    q, qd, qdd = (torch.from_numpy(a) for a in sample_states(cpu, B + off, seed=int(rng.integers(1 << 30))))
    names = list(cpu._name_to_idx_map)
    link = names[int(rng.integers(1, len(names)))]
    ctx = (robot, B, off, link)
    dq, dqd, dqdd = (t.cuda()[off:] for t in (q, qd, qdd))
    q, qd, qdd = q[off:], qd[off:], qdd[off:]
    big = B > 20000
    rp, rq = cpu.compute_forward_kinematics(q, link)
    rl, ra = cpu.compute_endeffector_jacobian(q, link)
    rt = cpu.compute_inverse_dynamics(q, qd, qdd)
    rH = cpu.compute_lagrangian_inertia_matrix(q) if not big else None
    rF = cpu.compute_forward_dynamics(q, qd, rt) if not big else None
    for tag, m in (("own", own), ("lib", lib)):
        p, qq = m.compute_forward_kinematics(dq, link)
        note("fk pos " + tag, rel(p, rp), 2e-5, ctx)
        sgn = torch.sign((qq.cpu() * rq).sum(-1, keepdim=True))
        note("fk quat " + tag, rel(qq.cpu() * sgn, rq), 5e-5, ctx)
        lin, ang = m.compute_endeffector_jacobian(dq, link)
        note("jac " + tag, max(rel(lin, rl), rel(ang, ra)), 2e-5, ctx)
        note("id " + tag, rel(m.compute_inverse_dynamics(dq, dqd, dqdd), rt), 3e-4, ctx)
        if not big:
            note("crba " + tag, rel(m.compute_lagrangian_inertia_matrix(dq), rH), 3e-4, ctx)
            note("fd " + tag, rel(m.compute_forward_dynamics(dq, dqd, rt.cuda()), rF), 2e-2, ctx)
    if B <= 4097:       # gradients of a scalar of the torques and of the pose
        def grads(m, a, b, c):
            xs = [t.clone().requires_grad_(True) for t in (a, b, c)]
            (m.compute_inverse_dynamics(*xs).pow(2).mean() + m.compute_forward_kinematics(xs[0], link)[0].pow(2).mean()).backward()
            return [x.grad for x in xs]
        ref = grads(cpu, q, qd, qdd)
        for tag, m in (("own", own), ("lib", lib)):
            got = grads(m, dq, dqd, dqdd)
            scale = max(1e-9, max(float(r.abs().max()) for r in ref))
            note("grads " + tag, max(float((g.cpu() - r).abs().max()) for g, r in zip(got, ref)) / scale, 2e-3, ctx)
print("soak: %d checks in %.0f s over %d robots, all within tolerance; worst relative deviations from libdrm_cpu:" % (n_checks, time.time() - t0, len(models)))
for k in sorted(worst):
    print("  %-12s %.2e" % (k, worst[k]))
