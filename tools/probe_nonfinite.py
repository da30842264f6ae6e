#!/usr/bin/env python3
"""Non-finite rows through every kernel family: the robots' OWN kernels (constants folded in, built with -ffinite-math-only) against
the library's kernels on the GPU and against libdrm_cpu.so (the host build of the same ABI, IEEE arithmetic) — which outputs of a
row with a NaN / +-Inf / 1e30 input are NaN / Inf / finite on each path.

    python tools/probe_nonfinite.py            (on a GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_model, sample_states  # noqa: E402

BAD = [("q", 3, float("nan")), ("q", 6, float("nan")), ("qd", 2, float("inf")), ("qdd", 5, float("-inf")), ("q", 0, 1e30),
       ("qd", 1, 1e30), ("qdd", 6, float("nan")), ("q", 1, float("inf"))]


def poisoned(m, B, seed):
    q, qd, qdd = sample_states(m, B, seed=seed)
    rows = []
    for i, (which, col, val) in enumerate(BAD):
        r = 5 + 67 * i                  # spread over tiles and lanes
        {"q": q, "qd": qd, "qdd": qdd}[which][r, col % q.shape[1]] = val
        rows.append(r)
    return q, qd, qdd, rows


def pattern(a):
    a = np.asarray(a).reshape(a.shape[0], -1)
    return "".join("N" if np.isnan(v) else ("I" if np.isinf(v) else ".") for v in a[0]) if a.shape[0] == 1 else None


def compare(name, outs, rows):
    """outs: {path: array [B, ...]}"""
    paths = list(outs)
    base = np.asarray(outs["cpu"])
    for p in paths:
        if p == "cpu":
            continue
        a = np.asarray(outs[p])
        good = np.ones(a.shape[0], bool)
        good[rows] = False
        fin = np.abs(a[good] - base[good]).max() / max(1e-9, np.abs(base[good]).max())
        print("%-26s %-8s clean rows: max rel diff vs cpu %.2e, non-finite in clean rows: %d" % (name, p, fin, int((~np.isfinite(a[good])).sum())))
    for i, r in enumerate(rows):
        pats = {p: pattern(np.asarray(outs[p])[r:r + 1]) for p in paths}
        same = len(set(pats.values())) == 1
        print("   row %4d %-3s[%d]=%-5s %s" % (r, BAD[i][0], BAD[i][1], BAD[i][2], "same " + pats["cpu"] if same else "DIFFER " + "  ".join("%s=%s" % kv for kv in pats.items())))


def main():
    dev = "cuda"
    for robot, link in (("panda_no_gripper", "panda_virtual_ee_link"), ("iiwa7", "iiwa_link_ee")):
        mc = load_model(robot)
        lib, own = load_model(robot, dev), load_model(robot, dev)
        lib.own_kernels = "off"
        own.specialize()
        B = 1024 * 128 + 64 * 3 + 7
        q, qd, qdd, rows = poisoned(mc, B, 1)
        t = lambda a: torch.from_numpy(a)
        res = {}
        for name, m, d in (("cpu", mc, "cpu"), ("library", lib, dev), ("own", own, dev)):
            a = [t(x).to(d) for x in (q, qd, qdd)]
            tau = m.compute_inverse_dynamics(*a)
            t2, pos, quat = m.compute_fk_and_inverse_dynamics(a[0], a[1], a[2], link)
            H = m.compute_lagrangian_inertia_matrix(a[0][:4096])
            acc = m.compute_forward_dynamics(a[0][:4096], a[1][:4096], a[2][:4096])
            xs = [x[:4096].clone().requires_grad_(True) for x in a]
            m.compute_inverse_dynamics(*xs).sum().backward()
            res[name] = dict(tau=tau.cpu().numpy(), fused_tau=t2.cpu().numpy(), fused_pos=pos.cpu().numpy(), fused_quat=quat.cpu().numpy(),
                             H=H.cpu().numpy(), fd=acc.cpu().numpy(), gq=xs[0].grad.cpu().numpy(), gqd=xs[1].grad.cpu().numpy(),
                             gqdd=xs[2].grad.cpu().numpy())
        print("=== %s" % robot)
        for key in res["cpu"]:
            compare(key, {p: res[p][key] for p in res}, rows)
    # the hand's fan-out FK
    tips = ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]
    mc = load_model("allegro_left")
    lib, own = load_model("allegro_left", dev), load_model("allegro_left", dev)
    lib.own_kernels = "off"
    own.specialize()
    q, _, _, rows = poisoned(mc, 4096, 2)
    res = {}
    for name, m, d in (("cpu", mc, "cpu"), ("library", lib, dev), ("own", own, dev)):
        out = m.compute_forward_kinematics_links(torch.from_numpy(q).to(d), tips)
        res[name] = {"pos_" + k: v[0].cpu().numpy() for k, v in out.items()}
        res[name].update({"quat_" + k: v[1].cpu().numpy() for k, v in out.items()})
    print("=== allegro_left fan-out")
    for key in res["cpu"]:
        compare(key, {p: res[p][key] for p in res}, rows)
    # a tree with its own straight-line kernels and a constant table (Fetch)
    mc = load_model("fetch")
    lib, own = load_model("fetch", dev), load_model("fetch", dev)
    lib.own_kernels = "off"
    own.specialize()
    q, qd, qdd, rows = poisoned(mc, 4096, 3)
    res = {}
    for name, m, d in (("cpu", mc, "cpu"), ("library", lib, dev), ("own", own, dev)):
        a = [torch.from_numpy(x).to(d) for x in (q, qd, qdd)]
        xs = [x.clone().requires_grad_(True) for x in a]
        m.compute_inverse_dynamics(*xs).sum().backward()
        res[name] = dict(tau=m.compute_inverse_dynamics(*a).cpu().numpy(), H=m.compute_lagrangian_inertia_matrix(a[0]).cpu().numpy(),
                         fd=m.compute_forward_dynamics(*a).cpu().numpy(), gq=xs[0].grad.cpu().numpy())
    print("=== fetch (own straight-line kernels, constant table)")
    for key in res["cpu"]:
        compare(key, {p: res[p][key] for p in res}, rows)


if __name__ == "__main__":
    main()
