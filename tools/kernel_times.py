#!/usr/bin/env python3
"""Kernel timings (hipGraph of 100 launches, HIP events): Panda FK+Jacobian, RNEA, CRBA, forward dynamics, RNEA
backward, FK; Allegro 4-tip FK and whole-tree dynamics.  Sizes above 2^22 time the metric kernel only."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample
from differentiable_robot_model_amd import backend


def graph_time(fn, launches=100, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / launches * 1e3)
    return best


# executed flops per evaluation from the ISA (tools/flop_count.py -> profiles/r05_flops.json); the robot's own kernels by default (round 6), DRM_SPECIALIZE=0: the library's
import json
try:
    FLOPS = json.load(open(os.path.join(ROOT, "profiles", "r05_flops.json")))["kernels"]
except OSError:
    FLOPS = {}
OWN = os.environ.get("DRM_SPECIALIZE") != "0"      # (round 6: the robots' own kernels are the default; DRM_SPECIALIZE=0: the library's)


def bound(what, B, us, nbytes):
    """' | HBM x.xx  VALU y.yy (N flop/eval executed) -> bound: ...': fractions of 8 TB/s and of 157.3 TFLOP/s vector FP32; a kernel is
    bound by whichever it uses more of (B < 131 072 rows of RNEA / FK + RNEA run the library's kernels either way)."""
    own = OWN and not (what in ("rnea", "fk+rnea") and B < 131072) and what != "fk_jacobian"
    if own:
        key = {"rnea": "rnea own", "fk+rnea": "fk+rnea own", "crba": "crba own", "fwd dyn": "fwd dyn own",
               "rnea bwd": "rnea bwd own (input gradients)"}.get(what)
    else:
        key = {"rnea": "rnea library (one sample per lane)" if B <= 65536 else "rnea library (two samples per lane)",
               "fk+rnea": "fk+rnea library (two samples per lane)", "crba": "crba library", "fwd dyn": "fwd dyn library",
               "rnea bwd": "rnea bwd library", "fk_jacobian": "fk_jacobian library"}.get(what)
    if key not in FLOPS:
        key = None
    if key is None:
        return ""
    f = FLOPS[key]["flops_per_eval"]
    hbm, valu = B * nbytes / us / 1e3 / 8000.0, B * f / us / 1e6 / 157.3
    return "  | HBM %.2f  VALU %.2f (%d flop/eval executed, %s) -> bound: %s" % (
        hbm, valu, f, "own kernel" if own else "library kernel", "HBM" if hbm >= valu else "vector FP32")


sizes = [int(a) for a in sys.argv[1:]] or [65536, 1 << 20, 1 << 22]
m = load("panda_no_gripper"); link = "panda_virtual_ee_link"
for B in sizes:
    q, qd, qdd = (t.cuda() for t in sample(m, B))
    plan = m.plan_fk_and_jacobian(q, link)
    us = graph_time(plan.launch)
    print("fk_jacobian panda B=%8d %9.2f us  %7.1f GB/s (224 B/eval)  %6.2f Gevals/s" % (B, us, B * 224 / us / 1e3, B / us / 1e3) + bound("fk_jacobian", B, us, 224))
    if B > (1 << 22):   # working set >> the 256 MB Infinity Cache: the metric kernel only
        del plan
        continue
    m.compute_inverse_dynamics(q[:64], qd[:64], qdd[:64])
    dt = m._get_walk(("tree",), whole_tree=True); of = m._ops_f(dt)          # the full walk (backward kernels)
    df = m._dynamics_walk()                              # forward dynamics: fixed leaf links folded into their parents
    tau = torch.empty(B, 7, device="cuda")
    lib = backend.load_library(); import ctypes
    walk = backend._walk_struct(df.program, m._ops_f(df), df.ops_i, 7)
    def rnea():
        backend._check(lib.drm_rnea(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), B, 3, tau.data_ptr(), None,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    us = graph_time(rnea)
    print("rnea        panda B=%8d %9.2f us  %7.1f GB/s (112 B/eval)  %6.2f Gevals/s  %5.1f TFLOP/s (2.6 kflop/eval)" %
          (B, us, B * 112 / us / 1e3, B / us / 1e3, B * 2.6e3 / us / 1e6) + bound("rnea", B, us, 112))
    Hm = torch.empty(B, 7, 7, device="cuda")
    def crba():
        backend._check(lib.drm_crba(ctypes.byref(walk), q.data_ptr(), B, Hm.data_ptr(), None,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    us = graph_time(crba)
    print("crba        panda B=%8d %9.2f us  %7.1f GB/s (224 B/eval)  %6.2f Gevals/s" % (B, us, B * 224 / us / 1e3, B / us / 1e3) + bound("crba", B, us, 224))
    acc = torch.empty(B, 7, device="cuda")
    def fd():
        backend._check(lib.drm_forward_dynamics(ctypes.byref(walk), q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), B, 1,
                                                acc.data_ptr(), None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    us = graph_time(fd)
    print("fwd dyn     panda B=%8d %9.2f us  %7.1f GB/s (112 B/eval)  %6.2f Gevals/s" % (B, us, B * 112 / us / 1e3, B / us / 1e3) + bound("fwd dyn", B, us, 112))
    gtau = torch.randn(B, 7, device="cuda")
    us = graph_time(lambda: backend.rnea_backward(dt.program, of, dt.ops_i, q, qd, qdd, gtau, True, True, 7, 0b10, True), launches=20)
    print("rnea bwd    panda B=%8d %9.2f us  %7.1f GB/s (196 B/eval)  %6.2f Gevals/s  (1 learnable link + input grads)" %
          (B, us, B * 196 / us / 1e3, B / us / 1e3))
    if B == sizes[0]:   # the walk the API runs for that case: the end-effector frame folded into link 7, link 2 learnable
        from differentiable_robot_model_amd.rigid_body_params import UnconstrainedScalar
        ml = load("panda_no_gripper")
        ml.make_link_param_learnable("panda_link2", "mass", UnconstrainedScalar(init_val=2.0))
    dl = ml._dynamics_walk(); ofl = ml._ops_f(dl).detach(); maskl = ml._learnable_op_mask(dl)
    us = graph_time(lambda: backend.rnea_backward(dl.program, ofl, dl.ops_i, q, qd, qdd, gtau, True, True, 7, maskl, True), launches=20)
    print("rnea bwd    panda B=%8d %9.2f us  the same on the %d-op walk the API builds (fixed tail folded)" % (B, us, dl.program.n_ops))
    dfz = m._dynamics_walk()
    us = graph_time(lambda: backend.rnea_backward(dfz.program, m._ops_f(dfz), dfz.ops_i, q, qd, qdd, gtau, True, True, 7, 0, True), launches=20)
    print("rnea bwd    panda B=%8d %9.2f us  input gradients only (%d-op walk), 196 B/eval" % (B, us, dfz.program.n_ops) + bound("rnea bwd", B, us, 196))
    idx = m._name_to_idx_map[link]
    df = m._get_walk(("fk", (idx,)), targets=[idx]); off = m._ops_f(df)
    pos = torch.empty(B, 1, 3, device="cuda"); quat = torch.empty(B, 1, 4, device="cuda")
    wf = backend._walk_struct(df.program, off, df.ops_i, 7)
    def fk():
        backend._check(lib.drm_fk(ctypes.byref(wf), q.data_ptr(), B, 1, pos.data_ptr(), quat.data_ptr(),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    us = graph_time(fk)
    print("fk          panda B=%8d %9.2f us  %7.1f GB/s (56 B/eval)" % (B, us, B * 56 / us / 1e3))
ma = load("allegro_left")
qa = sample(ma, 65536)[0].cuda()
tips = [ma._name_to_idx_map[t] for t in ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]]
da = ma._get_walk(("fk", tuple(tips)), targets=tips); ofa = ma._ops_f(da)
pa = torch.empty(65536, 4, 3, device="cuda"); ra = torch.empty(65536, 4, 4, device="cuda")
import ctypes
wa = backend._walk_struct(da.program, ofa, da.ops_i, 16)
def fka():
    backend._check(backend.load_library().drm_fk(ctypes.byref(wa), qa.data_ptr(), 65536, 4, pa.data_ptr(), ra.data_ptr(),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
us = graph_time(fka)
print("fk allegro 4 tips B=   65536 %9.2f us  %7.1f GB/s (176 B/eval)" % (us, 65536 * 176 / us / 1e3))
fan = ma._fanout_chains(tips, da)
if fan is not None:
    chains = [(c.program, ma._ops_f(c), c.ops_i) for c in fan]
    walks = (backend.DrmWalk * len(chains))(*[backend._walk_struct(p, f, i, 16) for p, f, i in chains])
    def fan_fk():
        backend._check(backend.load_library().drm_fk_fanout(walks, len(chains), qa.data_ptr(), 65536, pa.data_ptr(), ra.data_ptr(),
                                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    us = graph_time(fan_fk)
    print("fk allegro 4 tips B=   65536 %9.2f us  %7.1f GB/s (176 B/eval)   fan-out kernel (one wavefront per finger)" % (us, 65536 * 176 / us / 1e3))
    pl = torch.empty(4, 65536, 3, device="cuda"); rl = torch.empty(4, 65536, 4, device="cuda")
    def fan_fk_links():
        backend._check(backend.load_library().drm_fk_fanout_links(walks, len(chains), qa.data_ptr(), 65536, pl.data_ptr(), rl.data_ptr(),
                                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    us = graph_time(fan_fk_links)
    print("fk allegro 4 tips B=   65536 %9.2f us  %7.1f GB/s (176 B/eval)   fan-out kernel, link-major outputs (drm_fk_fanout_links)" % (us, 65536 * 176 / us / 1e3))
    us = graph_time(lambda: ma.compute_forward_kinematics_links(qa, ["link_3.0_tip", "link_7.0_tip", "link_11.0_tip", "link_15.0_tip"]))
    print("fk allegro 4 tips B=   65536 %9.2f us  the same through compute_forward_kinematics_links" % us)
# whole-tree dynamics of a branching robot through the generic (table-driven) kernels: Allegro, 21 links, 16 DoF
for B in [s for s in sizes if s <= (1 << 20)]:
    qa, qda, qdda = (t.cuda() for t in sample(ma, B))
    ma.compute_inverse_dynamics(qa[:64], qda[:64], qdda[:64])
    dta = ma._get_walk(("tree",), whole_tree=True); ofa = ma._ops_f(dta)
    dfa = ma._dynamics_walk()
    wt = backend._walk_struct(dfa.program, ma._ops_f(dfa), dfa.ops_i, 16)
    lib = backend.load_library()
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ta = torch.empty(B, 16, device="cuda"); Ha = torch.empty(B, 16, 16, device="cuda"); aa = torch.empty(B, 16, device="cuda")
    ga = torch.randn(B, 16, device="cuda")
    us = graph_time(lambda: backend._check(lib.drm_rnea(ctypes.byref(wt), qa.data_ptr(), qda.data_ptr(), qdda.data_ptr(), B, 3, ta.data_ptr(), None, st())), launches=20)
    print("rnea        allegro B=%8d %9.2f us  %7.1f GB/s (256 B/eval)  %6.2f Gevals/s" % (B, us, B * 256 / us / 1e3, B / us / 1e3))
    us = graph_time(lambda: backend._check(lib.drm_crba(ctypes.byref(wt), qa.data_ptr(), B, Ha.data_ptr(), None, st())), launches=20)
    print("crba        allegro B=%8d %9.2f us  %7.1f GB/s (1088 B/eval) %6.2f Gevals/s" % (B, us, B * 1088 / us / 1e3, B / us / 1e3))
    us = graph_time(lambda: backend._check(lib.drm_forward_dynamics(ctypes.byref(wt), qa.data_ptr(), qda.data_ptr(), qdda.data_ptr(), B, 1, aa.data_ptr(), None, st())), launches=20)
    print("fwd dyn     allegro B=%8d %9.2f us  %7.1f GB/s (256 B/eval)  %6.2f Gevals/s" % (B, us, B * 256 / us / 1e3, B / us / 1e3))
    us = graph_time(lambda: backend.rnea_backward(dta.program, ofa, dta.ops_i, qa, qda, qdda, ga, True, True, 16, 0b10, True), launches=10)
    print("rnea bwd    allegro B=%8d %9.2f us  %7.1f GB/s (448 B/eval)  %6.2f Gevals/s  (1 learnable link + input grads)" % (B, us, B * 448 / us / 1e3, B / us / 1e3))
    ofz = ma._ops_f(dfa)
    us = graph_time(lambda: backend.rnea_backward(dfa.program, ofz, dfa.ops_i, qa, qda, qdda, ga, True, True, 16, 0, True), launches=10)
    print("rnea bwd    allegro B=%8d %9.2f us  input gradients only (the %d-op walk the API builds: every fixed joint folded)" % (B, us, dfa.program.n_ops))
# BASELINE.json configs 2 and 3 as stated: iiwa7 FK + EE Jacobian at 65 536; Panda FK(EE) + RNEA on one GPU's shard of
# the 2^20 batch (131 072 rows), the two calls captured back to back
mi = load("iiwa7")
qi = sample(mi, 65536)[0].cuda()
plan_i = mi.plan_fk_and_jacobian(qi, "iiwa_link_ee")
us = graph_time(plan_i.launch)
print("config 2: fk_jacobian iiwa7 B=   65536 %9.2f us  %7.1f GB/s (224 B/eval)  %6.2f Gevals/s" % (us, 65536 * 224 / us / 1e3, 65536 / us / 1e3))
Bs = (1 << 20) // 8
qs, qds, qdds = (t.cuda() for t in sample(m, Bs))
plan_fk = m.plan_fk_and_jacobian(qs, link, want_pose=True)
idx = m._name_to_idx_map[link]
df = m._get_walk(("fk", (idx,)), targets=[idx]); off = m._ops_f(df)
pos_s = torch.empty(Bs, 1, 3, device="cuda"); quat_s = torch.empty(Bs, 1, 4, device="cuda")
wf = backend._walk_struct(df.program, off, df.ops_i, 7)
plan_id = m.plan_inverse_dynamics(qs, qds, qdds)
lib = backend.load_library()
def fk_then_rnea():
    backend._check(lib.drm_fk(ctypes.byref(wf), qs.data_ptr(), Bs, 1, pos_s.data_ptr(), quat_s.data_ptr(),
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    plan_id.launch()
us = graph_time(fk_then_rnea)
print("config 3: fk + rnea panda B=  %d (one GPU's shard of 2^20) %9.2f us  %7.1f GB/s (168 B/eval, two calls)  %6.2f Gevals/s"
      % (Bs, us, Bs * 168 / us / 1e3, Bs / us / 1e3))
for Bf in (Bs, 1 << 20):
    qf, qdf, qddf = (t.cuda() for t in sample(m, Bf))
    plan_f = m.plan_fk_and_inverse_dynamics(qf, qdf, qddf, link)
    us = graph_time(plan_f.launch)
    print("config 3: fk + rnea panda B=%8d, ONE fused launch (drm_fk_rnea) %9.2f us  %7.1f GB/s (140 B/eval)  %6.2f Gevals/s"
          % (Bf, us, Bf * 140 / us / 1e3, Bf / us / 1e3) + bound("fk+rnea", Bf, us, 140))
    plan_r = m.plan_inverse_dynamics(qf, qdf, qddf)
    us = graph_time(plan_r.launch)
    print("          rnea alone      B=%8d %9.2f us" % (Bf, us))
