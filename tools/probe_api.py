#!/usr/bin/env python3
"""Whole-API timings (hipGraph) of the secondary entry points: all-links FK, non-linear effects, the fused FK + Jacobian, FK of
one link, the Jacobian to a link in the middle of a tree.   usage: probe_api.py [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample


def graph_time(fn, launches=20, reps=3):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / launches * 1e3)
    return best


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for robot in ("panda_no_gripper", "panda", "allegro_left", "iiwa7_allegro", "fetch"):
    m = load(robot)
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    names = [b.name for b in m._bodies]
    ee, mid = names[-1], names[len(names) // 2]
    L = len(names)
    t_all = graph_time(lambda: m.compute_forward_kinematics_all_links(q), launches=5)
    t_fk = graph_time(lambda: m.compute_forward_kinematics(q, ee))
    t_mid = graph_time(lambda: m.compute_endeffector_jacobian(q, mid))
    t_nle = graph_time(lambda: m.compute_non_linear_effects(q, qd))
    t_fused = graph_time(lambda: m.compute_fk_and_jacobian(q, ee))
    print("%-18s B=%d links=%2d  FK all links %8.1f us (%5.0f GB/s)  FK(%s) %6.1f  Jacobian(%s) %6.1f  non-linear effects %6.1f  FK+Jacobian(ee) %6.1f" % (
        robot, B, L, t_all, B * (4 * m._n_dofs + 28 * L) / t_all / 1e3, ee[:10], t_fk, mid[:10], t_mid, t_nle, t_fused), flush=True)
