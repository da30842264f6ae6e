#!/usr/bin/env python3
"""The learnable-RNEA training step of the reference's L4DC experiments (experiments/l4dc-sim-exps.ipynb,
batch 256, ~14 it/s unconstrained / ~11 it/s constrained on an unknown CPU): iiwa7, mass / com / inertia_mat of all seven
moving links learnable, loss on the predicted torques, Adam.  Eager API step and the same step replayed as a hipGraph."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample  # noqa: E402
from differentiable_robot_model_amd.rigid_body_params import (PositiveScalar, SymmPosDef3DInertiaMatrixNet,  # noqa: E402
                                                               UnconstrainedTensor)


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for constrained in (False, True):
    for B in (256, 16384) + (() if constrained else (1 << 20,)):
        torch.manual_seed(0)
        gt, m = load("iiwa7"), load("iiwa7")
        for k in range(1, 8):
            link = "iiwa_link_%d" % k
            m.make_link_param_learnable(link, "mass", PositiveScalar())
            m.make_link_param_learnable(link, "com", UnconstrainedTensor(1, 3))
            m.make_link_param_learnable(link, "inertia_mat",
                                        SymmPosDef3DInertiaMatrixNet() if constrained else UnconstrainedTensor(3, 3))
        q, qd, qdd = (t.cuda() for t in sample(m, B))
        with torch.no_grad():
            want = gt.compute_inverse_dynamics(q, qd, qdd)
        params0 = [p.detach().clone() for p in m.parameters()]
        res = {}
        for fused in (False, True):      # torch's default (foreach) Adam as in the reference's example, and its fused one
            with torch.no_grad():
                for p, p0 in zip(m.parameters(), params0):
                    p.copy_(p0)
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=True if fused else None)

            def train_step():
                loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(q, qd, qdd), want)
                loss.backward()
                opt.step()
                return loss

            def eager():
                opt.zero_grad(set_to_none=True)
                train_step()

            eager_us = timeit(eager, iters=100)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    opt.zero_grad(set_to_none=True)
                    train_step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                static_loss = train_step()
            graph.replay(); torch.cuda.synchronize()
            l0 = static_loss.item()
            res[fused] = (eager_us, timeit(graph.replay), l0, static_loss.item())
        (eager_us, graph_us, l0, l1), (eager_f, graph_f, _, _) = res[False], res[True]
        print("learnable-RNEA train step, iiwa7, 21 parameter tensors (%s inertia), batch %6d:  eager %8.1f us = %7.0f it/s   "
              "hipGraph %7.1f us = %7.0f it/s   loss %.4g -> %.4g   | fused Adam: eager %7.1f us   hipGraph %6.1f us = %7.0f it/s"
              % ("SPD" if constrained else "unconstrained", B, eager_us, 1e6 / eager_us, graph_us, 1e6 / graph_us, l0, l1,
                 eager_f, graph_f, 1e6 / graph_f))
