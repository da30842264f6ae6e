#!/usr/bin/env python3
"""Whole-API timings (hipGraph of launches) of the robots that do NOT take the 7-DoF arm kernels: which kernel shapes matter."""
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
ONLY_FD = len(sys.argv) > 2 and sys.argv[2] == "fd"   # probe_robots.py B fd: the forward-dynamics column only
for robot in ("panda_no_gripper", "panda", "fetch", "fetch_arm_no_gripper", "jaco", "iiwa7_allegro", "trifinger_edu"):
    m = load(robot)
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    ee = m._bodies[-1].name
    dw = m._dynamics_walk()
    t_fkj = t_id = t_h = float("nan")
    if len(sys.argv) > 2 and sys.argv[2] == "crba":
        print("%-22s CRBA %7.1f us" % (robot, graph_time(lambda: m.compute_lagrangian_inertia_matrix(q))), flush=True)
        continue
    if not ONLY_FD:
        t_fkj = graph_time(lambda: m.compute_endeffector_jacobian(q, ee))
        t_id = graph_time(lambda: m.compute_inverse_dynamics(q, qd, qdd))
        t_h = graph_time(lambda: m.compute_lagrangian_inertia_matrix(q))
    t_fd = graph_time(lambda: m.compute_forward_dynamics(q, qd, qdd))
    t_bwd = float("nan")
    if not ONLY_FD and dw.program.backward_ok:
        from differentiable_robot_model_amd import backend
        of, gt = m._ops_f(dw), torch.randn(B, m._n_dofs, device="cuda")
        t_bwd = graph_time(lambda: backend.rnea_backward(dw.program, of, dw.ops_i, q, qd, qdd, gt, True, True, m._n_dofs, 0, True), launches=5)
    print("%-22s n=%2d ops=%2d segs=%d  FK+Jac(%s) %7.1f  RNEA %7.1f  CRBA %7.1f  FD %7.1f  RNEA backward (input gradients) %7.1f us" % (
        robot, m._n_dofs, dw.program.n_ops, dw.program.n_segments, ee[:14], t_fkj, t_id, t_h, t_fd, t_bwd), flush=True)
