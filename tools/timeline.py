#!/usr/bin/env python3
"""Per-wavefront timeline of the hot kernels (development; needs the -DDRM_TIMELINE variant library):

    tools/build_variants.sh timeline "-DDRM_TIMELINE"
    DRM_HIP_LIBRARY=tools/variants/libdrm_timeline.so python tools/timeline.py

Every wavefront stamps the 100 MHz real-time counter (one clock for the whole chip) at: 0 first instruction, 1 inputs + table
landed, 2 / 3 kernel-specific points of the arithmetic, 4 last store issued, 5 last store acknowledged.  Printed per launch:
when the first / median / last wave reaches each point (us after the first wave of the launch started) and the median time a
wave spends in each phase — i.e. how much of a launch is dispatch skew, load latency, issue time and store drain."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample  # noqa: E402
from differentiable_robot_model_amd import backend  # noqa: E402

SLOTS, WAVES = 8, 1 << 16
TICK_US = 0.01
TIPS = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]


def read(unit, n_waves):
    lib = backend.load_library()
    fn = getattr(lib, "drm_tl_read_" + unit)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    buf = np.zeros((WAVES, SLOTS), np.uint64)
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data, buf.nbytes, 1) == 0
    return buf[:n_waves]


def graph_us(fn, K=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / K * 1e3)
    return best, g


def report(name, unit, fn, n_waves, slots, labels):
    us, g = graph_us(fn)
    read(unit, n_waves)               # clear
    g.replay()                        # the buffer keeps the stamps of the LAST launch of the replayed graph
    rec = read(unit, n_waves).astype(np.int64)
    t0 = rec[:, 0].min()
    rel = (rec[:, :6] - t0) * TICK_US
    print("%s: %d waves, %.2f us per launch (graph of 50, instrumented build)" % (name, n_waves, us))
    print("   point                          first    p10    median    p90     last   (us after the launch's first wave started)")
    for s in slots:
        col = rel[:, s]
        print("   %d %-26s %7.2f %7.2f %7.2f %7.2f %7.2f" % (s, labels[s], col.min(), np.percentile(col, 10), np.median(col),
                                                             np.percentile(col, 90), col.max()))
    prev = slots[0]
    for s in slots[1:]:
        d = rel[:, s] - rel[:, prev]
        print("   phase %d -> %d: median %.2f us  (p10 %.2f, p90 %.2f)" % (prev, s, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
        prev = s
    hw = rec[:, 7]
    xcc = (hw >> 32) & 0xf
    simd = (hw >> 4) & 0x3
    cu = (hw >> 8) & 0xf
    se = (hw >> 13) & 0x7
    slot = hw & 0xf
    print("   placement: %d XCCs, waves per (xcc, se, cu, simd): max %d; wave slots used %s; start by xcc (median us): %s" % (
        len(set(xcc.tolist())), max(np.unique(((xcc * 8 + se) * 16 + cu) * 4 + simd, return_counts=True)[1]),
        sorted(set(slot.tolist())), " ".join("%.2f" % np.median(rel[xcc == x, 0]) for x in sorted(set(xcc.tolist())))))
    print(flush=True)


def main():
    m = load("panda_no_gripper"); link = "panda_virtual_ee_link"
    L = {0: "first instruction", 1: "inputs + table landed", 2: "first stores out (ang_jac)", 3: "chain done", 4: "last store issued",
         5: "last store acknowledged"}
    for B in (64, 65536, 131072):
        q, qd, qdd = (t.cuda() for t in sample(m, B))
        plan = m.plan_fk_and_jacobian(q, link)
        report("metric kernel B=%d" % B, "arm", plan.launch, B // 64, [0, 1, 2, 3, 4, 5], L)
    Lr = dict(L); Lr[2] = "forward sweep done"; Lr[3] = "backward sweep done"
    for B in (64, 32768, 65536):
        q, qd, qdd = (t.cuda() for t in sample(m, B))
        plan = m.plan_inverse_dynamics(q, qd, qdd)
        report("rnea_arm_kernel B=%d" % B, "dyn", plan.launch, B // 64, [0, 1, 2, 3, 4, 5], Lr)
    h = load("allegro_left")
    Lf = dict(L); Lf[3] = "chain done"
    for B in (256, 65536):
        q = sample(h, B)[0].cuda()
        with torch.no_grad():
            report("fk_fan_chain_kernel (Allegro, 4 tips, link-major) B=%d" % B, "chain",
                   lambda: h.compute_forward_kinematics_links(q, TIPS), B // 64 * 4, [0, 1, 3, 4, 5], Lf)


if __name__ == "__main__":
    main()
