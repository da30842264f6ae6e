import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from bench_configs import load, uniform_q
dev = torch.device("cuda", 0)
m = load("panda_no_gripper", dev)
q, _ = uniform_q(m, 65536, dev, 1)
plan = m.plan_fk_and_jacobian(q, "panda_virtual_ee_link")
for _ in range(50): plan.launch()
torch.cuda.synchronize()
def wall(fn, reps=15):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    out.sort(); return out[0], out[len(out)//2]
for K in (20, 200):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K): plan.launch()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    a = wall(g.replay)
    b = wall(lambda: plan.launch_many(K))
    def py():
        for _ in range(K): plan.launch()
    c = wall(py)
    print("K=%3d  graph %.2f / %.2f us per step (min / median)   C++ loop %.2f / %.2f   Python loop %.2f / %.2f" % (K, a[0]/K*1e6, a[1]/K*1e6, b[0]/K*1e6, b[1]/K*1e6, c[0]/K*1e6, c[1]/K*1e6))
