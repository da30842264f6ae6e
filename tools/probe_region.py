#!/usr/bin/env python3
"""ONE tool for the experiments on the timed region (round 5: region_probe.py / region_probe2.py folded in).
    probe_region.py [K]            the driver's K-step region issued as a graph / a C++ loop / a Python loop (below)
    probe_region.py [K] --events   HIP events recorded as NODES of the graph against events recorded around graph.replay()

The driver's K = 20 timed region (barrier-free, one rank): a replayed hipGraph of K launches against K direct launches enqueued by
a C++ loop (drm_hostcall.repeat_fk_jacobian) against K launches from a Python loop — wall time between two synchronizes."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from differentiable_robot_model_amd import backend

ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(ARGS[0]) if ARGS else 20

def event_nodes():
    """HIP events recorded as NODES of the graph that holds the K launches (hipGraphAddEventRecordNode through torch's stream
    capture) against events recorded around graph.replay() from the host — what is behind bench.py's two-pass timing."""
    import time
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


if "--events" in sys.argv:
    event_nodes()
    sys.exit(0)

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
