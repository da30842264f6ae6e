import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from oracle import Oracle
for name in ("panda_no_gripper", "iiwa7", "fetch_arm_no_gripper"):
    m = load(name); L = len(m._bodies)
    q = sample(m, 128, seed=3)[0]
    link = m._bodies[L - 1].name
    pos, quat, lin, ang = m.compute_fk_and_jacobian(q.cuda(), link)       # arm kernel (full tiles)
    p2, q2, l2, a2 = m.compute_fk_and_jacobian(q[:63].cuda(), link)       # generic kernel (ragged tile)
    op, oq, ol, oa = Oracle(m._spec).fk_jacobian(q.numpy().astype(np.float64), L - 1, np.float64)
    e = lambda a, b: float(np.abs(a.cpu().numpy() - b).max())
    print(name, "arm vs oracle: pos %.2e lin %.2e ang %.2e | generic vs oracle: pos %.2e lin %.2e" %
          (e(pos, op), e(lin, ol), e(ang, oa), e(p2, op[:63]), e(l2, ol[:63])))
    bad = np.abs(pos.cpu().numpy() - op).max(axis=1)
    print("  rows with pos error > 1e-5:", np.nonzero(bad > 1e-5)[0][:20], " q of first bad row:",
          q[np.nonzero(bad > 1e-5)[0][0]].numpy() if (bad > 1e-5).any() else None)
