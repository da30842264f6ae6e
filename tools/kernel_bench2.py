#!/usr/bin/env python3
"""Launch the secondary kernels a few times (Allegro 4-tip FK, Panda CRBA, forward dynamics, RNEA backward) — for
rocprofv3 counter runs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ma = load("allegro_left")
qa = sample(ma, B)[0].cuda()
tips = [ma._name_to_idx_map[t] for t in ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]]
m = load("panda_no_gripper")
q, qd, qdd = (t.cuda() for t in sample(m, B))
for _ in range(10):
    ma._fk_targets(qa, tips)
    H = m.compute_lagrangian_inertia_matrix(q)
    a = m.compute_forward_dynamics(q, qd, qdd)
qg = q.clone().requires_grad_(True)
for _ in range(5):
    m.compute_inverse_dynamics(qg, qd, qdd).sum().backward()
torch.cuda.synchronize()
print("done")
