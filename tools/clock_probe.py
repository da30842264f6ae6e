#!/usr/bin/env python3
"""What shader clock does the GPU run at while the metric kernel is replayed back to back? (rocm-smi while a
hipGraph of FK+Jacobian launches loops for a few seconds)"""
import os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
m = load("panda_no_gripper")
q = sample(m, 65536)[0].cuda()
plan = m.plan_fk_and_jacobian(q, "panda_virtual_ee_link")
plan.launch(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(1000):
        plan.launch()
stop = False
def spin():
    while not stop:
        g.replay(); torch.cuda.synchronize()
t = threading.Thread(target=spin); t.start()
time.sleep(1.5)
for _ in range(3):
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    print("\n".join(l for l in out.splitlines() if "sclk" in l or "mclk" in l or "fclk" in l or "Power" in l))
    time.sleep(0.7)
stop = True; t.join()
