#!/usr/bin/env python3
"""BASELINE config 5 timing: iiwa, learnable trans + rot_angles of iiwa_link_1, batch 16 384, FK + backward."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample  # noqa: E402
from differentiable_robot_model_amd import backend  # noqa: E402
from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor  # noqa: E402


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3, (time.perf_counter() - t0) / iters * 1e6


for B in (16384, 1 << 20):
    torch.manual_seed(0)
    m = load("iiwa7")
    gt = load("iiwa7")
    m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3))
    m.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(1, 3))
    q = sample(m, B)[0].cuda()
    with torch.no_grad():
        want, _ = gt.compute_forward_kinematics(q, "iiwa_link_ee")

    def step():
        m.zero_grad()
        pos, _ = m.compute_forward_kinematics(q, "iiwa_link_ee")
        loss = torch.nn.functional.mse_loss(pos, want)
        loss.backward()

    dev_us, wall_us = timeit(step)
    print("config5 API step  B=%8d  %9.1f us (device events) %9.1f us (wall)  %8.3f Mevals/s" %
          (B, dev_us, wall_us, B / wall_us))
    # the same training step (forward, loss, backward, Adam) captured once into a hipGraph and replayed: the host
    # leaves the loop, what remains is the kernels
    for fused in (False, True):
        try:
            state = {k: v.detach().clone() for k, v in m.state_dict().items()}
            # default implementation (foreach) as in the reference's examples, then the fused one
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, **({"fused": True} if fused else {}))

            def train_step():
                pos, _ = m.compute_forward_kinematics(q, "iiwa_link_ee")
                loss = torch.nn.functional.mse_loss(pos, want)
                loss.backward()
                opt.step()
                return loss

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
            g_us, g_wall = timeit(graph.replay, iters=200)
            print("config5 hipGraph-captured training step (fwd + loss + bwd + %s)  %9.1f us (device) %9.1f us (wall)  "
                  "%8.3f Mevals/s   loss %.6f -> %.6f" % ("fused Adam" if fused else "Adam", g_us, g_wall, B / g_wall, l0,
                                                          static_loss.item()))
            m.load_state_dict(state)
        except Exception as err:  # pragma: no cover - depends on the runtime
            print("graph capture of the training step failed:", repr(err)[:300])
    # kernels alone: forward FK + backward, constants fixed
    dw = m._get_walk(("fk", (m._name_to_idx_map["iiwa_link_ee"],)), targets=[m._name_to_idx_map["iiwa_link_ee"]])
    ops_f = m._ops_f(dw).detach()
    gpos = torch.randn(B, 1, 3, device="cuda")
    mask = m._kinematic_param_mask(dw)
    def graph_us(fn, launches=50):
        """launches captured into one hipGraph: kernel time without the host (an eager call costs 12-19 us of Python + ctypes,
        more than these kernels take at 16 384 samples)"""
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(launches):
                fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / launches * 1e3)
        return best

    us_f = graph_us(lambda: backend.fk(dw.program, ops_f, dw.ops_i, q, 1, 7))
    us_b = graph_us(lambda: backend.fk_backward(dw.program, ops_f, dw.ops_i, q, gpos, 1, 7, mask, False))
    us_bq = graph_us(lambda: backend.fk_backward(dw.program, ops_f, dw.ops_i, q, gpos, 1, 7, mask, True))
    print("  kernels (hipGraph of 50 launches each, incl. their scratch memsets / reductions): fk %.1f us, backward(params) %.1f us, "
          "backward(params + grad_q) %.1f us -> %.1f GB/s of 96 B/eval" % (us_f, us_b, us_bq, B * 96 / (us_f + us_bq) / 1e3))
