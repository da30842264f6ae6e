#!/usr/bin/env python3
"""Recover the mounting offset of a KUKA iiwa link from end-effector positions (BASELINE configuration 5; the workload of
the reference's examples/learn_kinematics_of_iiwa.py): `trans` and `rot_angles` of iiwa_link_1 are learnable, the loss is
the MSE of the end-effector position against a ground-truth model, Adam.  Everything runs on the MI355X.

The DEFAULT is the fast form of the same loop: the loss as ONE node (`model.fk_mse_loss`: forward kinematics, MSE and the
gradients in one pass, drm_fk_mse) and the whole step (zero_grad, loss, backward, Adam) replayed from a hipGraph — 34 us per step
at 16 384 rows.  `--reference-loop` runs the reference's loop literally (compute_forward_kinematics -> torch MSELoss ->
backward(): drm_fk + drm_fk_backward behind autograd), `--eager` without the graph: ~490 us per step, all of it host time
(profiles/r04_config5.txt; both forms are timed in bench.py's config 5 leg).

    python examples/learn_kinematics_of_iiwa.py [--batch 16384] [--epochs 300] [--reference-loop] [--eager]
"""
import argparse
import time

import _common  # noqa: F401  (sys.path)
import torch

from differentiable_robot_model_amd import DifferentiableKUKAiiwa
from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor


def run(batch=16384, epochs=300, lr=1e-2, use_graph=False, device="cuda", verbose=True, fused_loss=False):
    truth = DifferentiableKUKAiiwa(device=device)
    model = DifferentiableKUKAiiwa(device=device)
    model.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(dim1=1, dim2=3))
    model.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(dim1=1, dim2=3))
    q, _, _ = _common.sample_states(truth, batch, seed=1)
    with torch.no_grad():
        target, _ = truth.compute_forward_kinematics(q, "iiwa_link_ee")
    opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=use_graph)

    def step():
        if fused_loss:      # one node: FK, MSE and its gradients in one pass over q (drm_fk_mse)
            loss = model.fk_mse_loss(q, "iiwa_link_ee", target)
        else:               # the reference's loop, literally (examples/learn_kinematics_of_iiwa.py:47-55 upstream)
            pos, _ = model.compute_forward_kinematics(q, "iiwa_link_ee")
            loss = torch.nn.functional.mse_loss(pos, target)
        loss.backward()
        opt.step()
        return loss

    if use_graph:
        run_step = _common.graphed(step, opt)
    else:
        def run_step():
            opt.zero_grad(set_to_none=True)
            return step()
    history = []
    _common.sync(device)
    t0 = time.perf_counter()
    for epoch in range(epochs):
        loss = run_step()
        if epoch % max(1, epochs // 10) == 0 or epoch == epochs - 1:
            history.append(float(loss.detach()))
            if verbose:
                print("epoch %5d  loss %.3e" % (epoch, history[-1]))
    _common.sync(device)
    if verbose:
        print("%.1f us per step (%s, %s)" % ((time.perf_counter() - t0) / epochs * 1e6, "hipGraph" if use_graph else "eager",
                                             "fk_mse_loss" if fused_loss else "compute_forward_kinematics + MSELoss"))
        print("learned trans      ", model._bodies[model._name_to_idx_map["iiwa_link_1"]].trans().detach().cpu().numpy().ravel())
        print("ground-truth trans ", truth._bodies[truth._name_to_idx_map["iiwa_link_1"]].trans().detach().cpu().numpy().ravel())
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--reference-loop", action="store_true", help="compute_forward_kinematics + torch MSELoss instead of fk_mse_loss")
    ap.add_argument("--eager", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    a = ap.parse_args()
    run(a.batch, a.epochs, use_graph=not a.eager, fused_loss=not a.reference_loop)
