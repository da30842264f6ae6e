#!/usr/bin/env python3
"""Launch the inverse-dynamics backward kernels (input gradients only) of one robot a few times — for rocprofv3 counter runs.
   usage: kernel_bench5.py ROBOT [B]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from differentiable_robot_model_amd import backend
robot = sys.argv[1] if len(sys.argv) > 1 else "panda"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
m = load(robot)
q, qd, qdd = (t.cuda() for t in sample(m, B))
dw = m._dynamics_walk()
of, gt = m._ops_f(dw), torch.randn(B, m._n_dofs, device="cuda")
for _ in range(4):
    backend.rnea_backward(dw.program, of, dw.ops_i, q, qd, qdd, gt, True, True, m._n_dofs, 0, True)
torch.cuda.synchronize()
print("done")
