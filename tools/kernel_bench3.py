#!/usr/bin/env python3
"""Launch the loop-structured (tree) kernels on the Allegro hand a few times — for rocprofv3 counter runs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ma = load("allegro_left")
qa, qda, qdda = (t.cuda() for t in sample(ma, B))
tips = [ma._name_to_idx_map[t] for t in ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]]
for _ in range(5):
    ma._fk_targets(qa, tips)
    tau = ma.compute_inverse_dynamics(qa, qda, qdda)
    H = ma.compute_lagrangian_inertia_matrix(qa)
    a = ma.compute_forward_dynamics(qa, qda, qdda)
torch.cuda.synchronize()
print("done")
# the loop-structured RNEA backward (fanned out over the fingers): input gradients, then with one learnable link
from differentiable_robot_model_amd import backend
full = ma._get_walk(("tree",), whole_tree=True)
of = ma._ops_f(full)
gt = torch.randn(B, ma._n_dofs, device="cuda")
for _ in range(3):
    backend.rnea_backward(full.program, of, full.ops_i, qa, qda, qdda, gt, True, True, ma._n_dofs, 0, True)
    backend.rnea_backward(full.program, of, full.ops_i, qa, qda, qdda, gt, True, True, ma._n_dofs, 1 << 5, True)
torch.cuda.synchronize()
print("done backward")
