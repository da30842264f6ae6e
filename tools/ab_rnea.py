#!/usr/bin/env python3
"""A/B timing of the arm dynamics kernels of whatever library DRM_HIP_LIBRARY names (hipGraph, HIP events, best of 7):
RNEA at 65 536 / 2^20, the fused FK + RNEA launch at 131 072 (config-3 shard) / 2^20, forward dynamics and CRBA at 2^20."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample


def graph_time(fn, launches=100, reps=5):
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


def main():
    m = load("panda_no_gripper"); link = "panda_virtual_ee_link"
    tag = os.path.basename(os.environ.get("DRM_HIP_LIBRARY", "libdrm_hip.so"))
    for B, mode in ((65536, "off"), (65536, None), (131072, "off"), (131072, None), (1 << 20, "off"), (1 << 20, None)):
        m.own_kernels = mode          # round 6: "off" = the library's kernels, None = the default (the arm's own kernels)
        tag = "library" if mode == "off" else "default (own kernels)"
        q, qd, qdd = (t.cuda() for t in sample(m, B))
        p_id = m.plan_inverse_dynamics(q, qd, qdd)
        p_fu = m.plan_fk_and_inverse_dynamics(q, qd, qdd, link)
        t_id = graph_time(p_id.launch, launches=100, reps=7)
        t_fu = graph_time(p_fu.launch, launches=100, reps=7)
        line = "%-22s B=%8d  rnea %8.2f us   fk+rnea %8.2f us" % (tag, B, t_id, t_fu)
        if B == 1 << 20:
            tau = p_id.tau if hasattr(p_id, "tau") else None
            t_fd = graph_time(lambda: m.compute_forward_dynamics(q, qd, qdd), launches=20, reps=5)
            t_h = graph_time(lambda: m.compute_lagrangian_inertia_matrix(q), launches=20, reps=5)
            line += "   fwd dyn (API) %8.2f us   mass matrix (API) %8.2f us" % (t_fd, t_h)
        print(line, flush=True)


if __name__ == "__main__":
    main()
