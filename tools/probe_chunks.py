#!/usr/bin/env python3
"""2^24 rows of the metric kernel as ONE launch against 2 / 4 / 8 back-to-back launches over slices of the same buffers
(if 0.84 of HBM holds per chunk, the fall-off beyond 4 GB per launch is an allocation-span effect and
chunking is the fix).  hipGraph of K steps, HIP events, median of 5."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_configs import graph_launch_us, load, uniform_q  # noqa: E402

dev = torch.device("cuda", 0)
m = load("panda_no_gripper", dev)
link = "panda_virtual_ee_link"
for B in (1 << 22, 1 << 24):
    q, _ = uniform_q(m, B, dev, 3)
    whole = m.plan_fk_and_jacobian(q, link)
    us, _ = graph_launch_us(whole.launch, 10)
    print("B = 2^%d   one launch            %8.1f us   %.3f of 8 TB/s" % (B.bit_length() - 1, us, 224 * B / us / 1e3 / 8000), flush=True)
    for parts in (2, 4, 8):
        rows = B // parts
        plans = [m.plan_fk_and_jacobian(q[i * rows:(i + 1) * rows], link) for i in range(parts)]

        def step():
            for p in plans:
                p.launch()
        us, _ = graph_launch_us(step, 10)
        print("B = 2^%d   %d launches of 2^%d  %8.1f us   %.3f of 8 TB/s" % (B.bit_length() - 1, parts, rows.bit_length() - 1, us, 224 * B / us / 1e3 / 8000), flush=True)
        del plans
    del whole, q
    torch.cuda.empty_cache()
