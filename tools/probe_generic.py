#!/usr/bin/env python3
"""Panda dynamics through the loop-form (tree) kernels instead of the arm specialisations — A/B probe."""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from differentiable_robot_model_amd import backend


def graph_time(fn, launches=50, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / launches * 1e3)
    return best


m = load("panda_no_gripper")
lib = backend.load_library()
for B in [int(a) for a in sys.argv[1:]] or [65536, 1 << 20]:
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    m.compute_inverse_dynamics(q[:64], qd[:64], qdd[:64])
    dt = m._get_walk(("tree",), whole_tree=True); of = m._ops_f(dt)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for generic in (False, True):
        walk = backend._walk_struct(dt.program, of, dt.ops_i, 7)
        if generic:
            walk.shape &= ~1
        tau = torch.empty(B, 7, device="cuda"); Hm = torch.empty(B, 7, 7, device="cuda"); acc = torch.empty(B, 7, device="cuda")
        scr1 = torch.empty(max(1, int(lib.drm_rnea_scratch_floats(ctypes.byref(walk), B))), device="cuda")
        t1 = graph_time(lambda: backend._check(lib.drm_rnea(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), B, 3, tau.data_ptr(), scr1.data_ptr(), st())))
        scr2 = torch.empty(max(1, int(lib.drm_crba_scratch_floats(ctypes.byref(walk), B))), device="cuda")
        t2 = graph_time(lambda: backend._check(lib.drm_crba(ctypes.byref(walk), q.data_ptr(), B, Hm.data_ptr(), scr2.data_ptr(), st())))
        scr = torch.empty(max(1, int(lib.drm_forward_dynamics_scratch_floats(ctypes.byref(walk), B))), device="cuda")
        t3 = graph_time(lambda: backend._check(lib.drm_forward_dynamics(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), B, 1, acc.data_ptr(), scr.data_ptr(), st())))
        print("B=%8d %s  rnea %8.2f  crba %8.2f  fd %8.2f us" % (B, "tree" if generic else "arm ", t1, t2, t3))

# RNEA backward: arm kernel vs the loop-structured kernel on the same robot
import dataclasses
for B in [65536, 1 << 20]:
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    gtau = torch.randn(B, 7, device="cuda")
    dt = m._get_walk(("tree",), whole_tree=True); of = m._ops_f(dt)
    for generic in (False, True):
        prog = dataclasses.replace(dt.program, shape=dt.program.shape & ~1) if generic else dt.program
        for mask in (0, 0b10):
            us = graph_time(lambda: backend.rnea_backward(prog, of, dt.ops_i, q, qd, qdd, gtau, True, True, 7, mask, True), launches=20)
            print("B=%8d rnea bwd %s mask=%d  %8.2f us" % (B, "tree" if generic else "arm ", mask, us))
    df = m._dynamics_walk(); off = m._ops_f(df)
    us = graph_time(lambda: backend.rnea_backward(df.program, off, df.ops_i, q, qd, qdd, gtau, True, True, 7, 0, True), launches=20)
    print("B=%8d rnea bwd arm, folded walk (7 links), input gradients only  %8.2f us" % (B, us))
