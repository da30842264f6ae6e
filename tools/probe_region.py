#!/usr/bin/env python3
"""The driver's K = 20 timed region (barrier-free, one rank): a replayed hipGraph of K launches against K direct launches enqueued by
a C++ loop (drm_hostcall.repeat_fk_jacobian) against K launches from a Python loop — wall time between two synchronizes."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from differentiable_robot_model_amd import backend

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = load("panda_no_gripper")
q = sample(m, 65536)[0].cuda()
plan = m.plan_fk_and_jacobian(q, "panda_virtual_ee_link")
lib = backend.load_library()
hc = backend.hostcall()
fn = backend._fn_addr(lib, "drm_fk_jacobian")
walk = ctypes.addressof(plan._walk)
args = (fn, walk, plan.q.data_ptr(), plan.batch, plan.pos.data_ptr(), plan.quat.data_ptr(), plan.lin.data_ptr(), plan.ang.data_ptr())
stream = torch.cuda.current_stream()
g = torch.cuda.CUDAGraph()
plan.launch(); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(K):
        plan.launch()
modes = {"hipGraph replay": lambda: g.replay(),
         "C++ loop of direct launches": lambda: hc.repeat_fk_jacobian(*args, torch.cuda.current_stream().cuda_stream, K),
         "Python loop of plan.launch()": lambda: [plan.launch() for _ in range(K)]}
side = torch.cuda.Stream()
for where in ("default stream", "a created stream"):
  ctx = torch.cuda.stream(side) if where != "default stream" else torch.cuda.stream(torch.cuda.default_stream())
  with ctx:
   if where != "default stream":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            plan.launch()
    modes["hipGraph replay"] = lambda: g.replay()
   # the same K launches as a SHORT graph followed by the rest: the doorbell rings after the first packets are written
   for head in (1, 2, 4):
    if head < K:
        gh, gt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gh):
            for _ in range(head):
                plan.launch()
        with torch.cuda.graph(gt):
            for _ in range(K - head):
                plan.launch()
        modes["two graphs: %d + %d" % (head, K - head)] = (lambda a, b: (lambda: (a.replay(), b.replay())))(gh, gt)
   for name, fn_ in modes.items():
    name = name + " / " + where
    for _ in range(5):
        fn_()
    torch.cuda.synchronize()
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.05:
        fn_(); torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn_(); torch.cuda.synchronize(); t1 = time.perf_counter()
        ts.append((t1 - t0) / K * 1e6)
    ts.sort()
    print("K=%d  %-48s median %.2f us per step   p10 %.2f   min %.2f" % (K, name, ts[len(ts) // 2], ts[len(ts) // 10], ts[0]))
