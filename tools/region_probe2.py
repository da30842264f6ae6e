#!/usr/bin/env python3
"""Development probe: HIP events recorded as NODES of the graph that holds the K launches (hipGraphAddEventRecordNode through
torch's stream capture) against events recorded around graph.replay() from the host.   usage: region_probe2.py [K]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = load("panda_no_gripper"); link = "panda_virtual_ee_link"
q = sample(m, 65536)[0].cuda()
plan = m.plan_fk_and_jacobian(q, link)
for _ in range(5): plan.launch()
torch.cuda.synchronize()
s = torch.cuda.current_stream()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(s); b.record(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        a.record(torch.cuda.current_stream())
        for _ in range(K): plan.launch()
        b.record(torch.cuda.current_stream())
    inside = True
except Exception as err:
    print("capture with event nodes failed:", err); inside = False
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(K): plan.launch()
c, d = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
c.record(s); d.record(s); torch.cuda.synchronize()
t_end = time.perf_counter() + 0.05
while time.perf_counter() < t_end:
    for _ in range(10): g2.replay()
    torch.cuda.synchronize()
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); c.record(s); g2.replay(); d.record(s)
    while not d.query(): pass
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = "K=%d  host-recorded events %.2f us/step   wall %.2f us/step" % (K, c.elapsed_time(d) * 1e3 / K, (t1 - t0) * 1e6 / K)
    if inside:
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
        try:
            out += "   |  event NODES %.2f us/step   wall %.2f us/step" % (a.elapsed_time(b) * 1e3 / K, (t1 - t0) * 1e6 / K)
        except Exception as err:
            out += "   |  elapsed_time of event nodes failed: %s" % err
    print(out)
