#!/usr/bin/env python3
"""Where the host time of an eager public-API call goes (cProfile over 3 000 calls per method, Panda, 65 536 rows)."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_probe import load, sample

m = load("panda_no_gripper"); link = "panda_virtual_ee_link"
q, qd, qdd = (t.cuda() for t in sample(m, 65536))
calls = {"compute_forward_kinematics": lambda: m.compute_forward_kinematics(q, link),
         "compute_endeffector_jacobian": lambda: m.compute_endeffector_jacobian(q, link),
         "compute_inverse_dynamics": lambda: m.compute_inverse_dynamics(q, qd, qdd),
         "compute_forward_kinematics_all_links": lambda: m.compute_forward_kinematics_all_links(q),
         "compute_lagrangian_inertia_matrix": lambda: m.compute_lagrangian_inertia_matrix(q),
         "compute_forward_dynamics": lambda: m.compute_forward_dynamics(q, qd, qdd),
         "compute_fk_and_inverse_dynamics": lambda: m.compute_fk_and_inverse_dynamics(q, qd, qdd, link)}
N = 3000
for name, fn in calls.items():
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(N):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e6)
    print("%-38s %6.2f us per eager call (wall, %d calls, one synchronize at the end)" % (name, best, N))
# a LEARNED model (iiwa, mass / com / SPD inertia matrix of the seven links learnable) called where no graph is built: the prepared call
# rebuilds the walk table from the parameter tensors in front of every launch (FastCall.set_table); DRM_NO_HOSTCALL=1: the Python path
from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, SymmPosDef3DInertiaMatrixNet, UnconstrainedTensor
lm = load("iiwa7")
for k in range(1, 8):
    lm.make_link_param_learnable("iiwa_link_%d" % k, "mass", PositiveScalar())
    lm.make_link_param_learnable("iiwa_link_%d" % k, "com", UnconstrainedTensor(1, 3))
    lm.make_link_param_learnable("iiwa_link_%d" % k, "inertia_mat", SymmPosDef3DInertiaMatrixNet())
lq, lqd, lqdd = (t.cuda() for t in sample(lm, 65536))
learned = {"compute_forward_kinematics": lambda: lm.compute_forward_kinematics(lq, "iiwa_link_ee"),
           "compute_endeffector_jacobian": lambda: lm.compute_endeffector_jacobian(lq, "iiwa_link_ee"),
           "compute_inverse_dynamics": lambda: lm.compute_inverse_dynamics(lq, lqd, lqdd),
           "compute_lagrangian_inertia_matrix": lambda: lm.compute_lagrangian_inertia_matrix(lq),
           "compute_forward_dynamics": lambda: lm.compute_forward_dynamics(lq, lqd, lqdd)}
with torch.no_grad():
    for name, fn in learned.items():
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(N):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / N * 1e6)
        print("learned model, no_grad: %-34s %6.2f us per eager call" % (name, best))
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    for name, fn in calls.items():
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(N):
            fn()
        pr.disable()
        torch.cuda.synchronize()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
        print("==== %s (tottime, %d calls)" % (name, N))
        print("\n".join(l for l in s.getvalue().splitlines()[4:] if l.strip())[:3500])
