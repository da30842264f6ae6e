#!/usr/bin/env python3
"""What does the host's way of waiting cost the driver's K = 20 region?  The same replayed hipGraph of 20 metric launches, bracketed by
torch.cuda.synchronize() as bench.py does, with the device's schedule flag at its default (auto), spin and blocking-sync."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
m = load("panda_no_gripper")
q = sample(m, 65536)[0].cuda()
plan = m.plan_fk_and_jacobian(q, "panda_virtual_ee_link")
plan.launch(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        plan.launch()
for name, flag in (("auto (default)", 0), ("spin", 1), ("yield", 2), ("blocking sync", 4), ("auto again", 0)):
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag))
    for _ in range(20):
        g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e6)
    ts.sort()
    print("%-16s hipSetDeviceFlags rc %d   us per step: median %.3f  best %.3f  p90 %.3f" % (name, rc, ts[100], ts[0], ts[180]))
