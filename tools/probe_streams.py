#!/usr/bin/env python3
"""The metric launch (Panda FK + Jacobian, 65 536 rows) as INDEPENDENT batches in flight on S streams: a hipGraph of K launches
dealt round-robin to S captured side streams, each with its own inputs and outputs — what a caller with several independent
batches (particle sets, parallel MPC problems) gets, against the one-after-the-other replay of bench.py (S = 1)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = load("panda_no_gripper")
K = 240
for S in (1, 2, 3, 4, 8):
    plans = [m.plan_fk_and_jacobian(sample(m, B)[0].cuda(), "panda_virtual_ee_link") for _ in range(S)]
    main = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in range(S)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        for p in plans:
            p.launch()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main):
            for s in side:
                s.wait_stream(main)
            for k in range(K):
                with torch.cuda.stream(side[k % S]):
                    plans[k % S].launch()
            for s in side:
                main.wait_stream(s)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main); g.replay(); b.record(main); b.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / K)
    print("B=%d  S=%d streams: %.2f us per launch  %.2f Gevals/s  %.0f GB/s of 224 B/eval" % (B, S, best, B / best / 1e3, B * 224 / best / 1e3))
