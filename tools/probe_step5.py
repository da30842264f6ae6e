#!/usr/bin/env python3
"""BASELINE configuration 5, the whole training step: WHICH kernels does one step launch?

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_step5 -- python tools/probe_step5.py [dyn] [graph] [B]
  python tools/probe_step5.py --read gpurun_out/prof_step5        (lists the kernels of the last steps, in order, with durations)

The step is the one bench_configs.py times: model.fk_mse_loss (ONE node) + loss.backward() + fused Adam, eagerly launched here so that
every kernel is a dispatch of its own in the trace.  A roctx-free way to find the step boundaries: the Adam kernel ends a step."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def read(folder):
    files = glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    ends = [i for i, n in enumerate(names) if "adam" in n.lower()]
    if len(ends) < 3:
        print("no Adam kernels in the trace (%d dispatches)" % len(rows))
        return
    for label, lo, hi in (("second-to-last step", ends[-3] + 1, ends[-2] + 1), ("last step", ends[-2] + 1, ends[-1] + 1)):
        print("== %s: %d kernels" % (label, hi - lo))
        t0 = int(rows[lo]["Start_Timestamp"])
        total = 0
        for r in rows[lo:hi]:
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            total += dur
            print("  +%8.2f us  %6.2f us  grid %-8s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur / 1e3,
                                                           r.get("Grid_Size_X", "?"), r["Kernel_Name"][:110]))
        print("  kernel time %.2f us, first start -> last end %.2f us" % (total / 1e3, (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        return read(sys.argv[2])
    import torch
    from gpu_probe import load, sample
    from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, UnconstrainedTensor
    dyn = "dyn" in sys.argv[1:]        # the learn-dynamics step (examples/learn_dynamics_iiwa.py) instead
    graph = "graph" in sys.argv[1:]    # ... and ten replays of the step captured into a hipGraph behind the eager steps
    args = [a for a in sys.argv[1:] if a not in ("dyn", "graph")]
    B = int(args[0]) if args else 16384
    torch.manual_seed(0)
    m, gt = load("iiwa7"), load("iiwa7")
    if dyn:
        for k in range(1, 8):
            link = "iiwa_link_%d" % k
            m.make_link_param_learnable(link, "mass", PositiveScalar())
            m.make_link_param_learnable(link, "com", UnconstrainedTensor(1, 3))
            m.make_link_param_learnable(link, "inertia_mat", UnconstrainedTensor(3, 3))
        q, qd, qdd = (t.cuda() for t in sample(m, B))
        with torch.no_grad():
            want = gt.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    else:
        m.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(1, 3))
        m.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(1, 3))
        q = sample(m, B)[0].cuda()
        with torch.no_grad():
            want, _ = gt.compute_forward_kinematics(q, "iiwa_link_ee")
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=True)
    def step():
        opt.zero_grad(set_to_none=True)
        if dyn:
            loss = torch.nn.functional.mse_loss(m.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True), want)
        else:
            loss = m.fk_mse_loss(q, "iiwa_link_ee", want)
        loss.backward()
        opt.step()
        return loss
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(12):
            loss = step()
            torch.cuda.synchronize()
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                loss = step()
            for _ in range(10):
                g.replay()
                torch.cuda.synchronize()
    print("loss %.6f" % float(loss))


if __name__ == "__main__":
    main()
