#!/usr/bin/env python3
"""Launch RNEA, the mass matrix and forward dynamics of one robot a few times — for rocprofv3 counter runs.
   usage: kernel_bench4.py ROBOT [B]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
m = load(robot)
q, qd, qdd = (t.cuda() for t in sample(m, B))
for _ in range(4):
    tau = m.compute_inverse_dynamics(q, qd, qdd)
    H = m.compute_lagrangian_inertia_matrix(q)
    a = m.compute_forward_dynamics(q, qd, qdd)
torch.cuda.synchronize()
print("done")
