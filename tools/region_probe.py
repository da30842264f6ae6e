import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import contextlib, io
from differentiable_robot_model_amd.robot_model import DifferentiableRobotModel, robot_description_folder
import bench
with contextlib.redirect_stdout(io.StringIO()):
    m = DifferentiableRobotModel(os.path.join(robot_description_folder, "panda_no_gripper.urdf"), device="cuda:0")
q, _ = bench.sample_q(m, 65536, torch.device("cuda:0"), 1)
plan = m.plan_fk_and_jacobian(q, "panda_virtual_ee_link")
stream = torch.cuda.current_stream()
K = 20
for _ in range(5): plan.launch()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(K): plan.launch()
g.replay(); g.replay(); torch.cuda.synchronize()
def region(mode):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream); ev1.record(stream); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "events":
        ev0.record(stream); g.replay(); ev1.record(stream)
        while not ev1.query(): pass
    elif mode == "streamquery":
        g.replay()
        while not stream.query(): pass
    elif mode == "eager":
        for _ in range(K): plan.launch()
        while not stream.query(): pass
    elif mode == "sync":
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6
for mode in ("events", "streamquery", "sync", "eager"):
    ts = sorted(region(mode) for _ in range(15))
    print("%-12s K=%d wall us: min %.1f med %.1f  -> us/step %.2f" % (mode, K, ts[0], ts[7], ts[7] / K))
