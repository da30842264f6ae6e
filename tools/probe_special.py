#!/usr/bin/env python3
"""Fetch (a tree with no compiled straight-line shape): inverse dynamics through the loop kernels and through the robot's own
kernel (model.specialize(): csrc/drm_static.hpp built for its tree), hipGraph of 20 launches, best of 5."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from ab_rnea import graph_time

FORCE = "--force" in sys.argv       # robots with a compiled shape: their shape's kernels ("loop kernels" column) against their own
for robot in [a for a in sys.argv[1:] if not a.startswith("--")] or ["fetch"]:
    loop, own = load(robot), load(robot)
    t0 = time.perf_counter()
    took = own.specialize(force=FORCE)
    print("%s: specialize() -> %s in %.2f s" % (robot, took, time.perf_counter() - t0))
    for B in (65536, 1 << 20):
        q, qd, qdd = (t.cuda() for t in sample(loop, B))
        a = graph_time(lambda: loop.compute_inverse_dynamics(q, qd, qdd), launches=20, reps=5)
        b = graph_time(lambda: own.compute_inverse_dynamics(q, qd, qdd), launches=20, reps=5)
        n = loop._n_dofs
        print("%-10s B=%8d  inverse dynamics: loop kernels %8.2f us   own straight-line kernel %8.2f us  (%.2fx, %.0f GB/s of %d B/eval)"
              % (robot, B, a, b, a / b, B * 16 * n / b / 1e3, 16 * n))
        a = graph_time(lambda: loop.compute_lagrangian_inertia_matrix(q), launches=10, reps=5)
        b = graph_time(lambda: own.compute_lagrangian_inertia_matrix(q), launches=10, reps=5)
        fa = graph_time(lambda: loop.compute_forward_dynamics(q, qd, qdd), launches=10, reps=5)
        fb = graph_time(lambda: own.compute_forward_dynamics(q, qd, qdd), launches=10, reps=5)
        print("%-10s B=%8d  forward dynamics: loop kernels %8.2f us   own straight-line kernel %8.2f us  (%.2fx)" % (robot, B, fa, fb, fa / fb))
        print("%-10s B=%8d  mass matrix:      loop kernels %8.2f us   own straight-line kernel %8.2f us  (%.2fx, %.0f GB/s of %d B/eval)"
              % (robot, B, a, b, a / b, B * 4 * (n + n * n) / b / 1e3, 4 * (n + n * n)))
        from differentiable_robot_model_amd import backend
        gt = torch.randn(B, n, device="cuda")
        times = []
        for m in (loop, own):
            dw = m._dynamics_walk()
            ops_f = m._ops_f(dw)
            for mask, want_q in ((0, True), (1 << 5, True)):
                times.append(graph_time(lambda: backend.rnea_backward(dw.program, ops_f, dw.ops_i, q, qd, qdd, gt, True, True, n, mask, want_q),
                                        launches=10, reps=5))
        print("%-10s B=%8d  inverse dynamics, reverse mode (input gradients): loop kernels %8.2f us   own straight-line kernel %8.2f us  (%.2fx, %.0f GB/s of %d B/eval)"
              % (robot, B, times[0], times[2], times[0] / times[2], B * 28 * n / times[2] / 1e3, 28 * n))
        print("%-10s B=%8d  ... + the constants of one learnable link:        loop kernels %8.2f us   own straight-line kernel %8.2f us  (%.2fx)"
              % (robot, B, times[1], times[3], times[1] / times[3]))
