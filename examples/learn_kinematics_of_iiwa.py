#!/usr/bin/env python3
"""Recover the mounting offset of a KUKA iiwa link from end-effector positions (BASELINE configuration 5; the workload of
the reference's examples/learn_kinematics_of_iiwa.py): `trans` and `rot_angles` of iiwa_link_1 are learnable, the loss is
the MSE of the end-effector position against a ground-truth model, Adam.  Everything runs on the MI355X: FK through
drm_fk, its backward through drm_fk_backward, the table of the learnable link through drm_walk_table.

    python examples/learn_kinematics_of_iiwa.py [--batch 16384] [--epochs 300] [--graph]
"""
import argparse
import time

import _common  # noqa: F401  (sys.path)
import torch

from differentiable_robot_model_amd import DifferentiableKUKAiiwa
from differentiable_robot_model_amd.rigid_body_params import UnconstrainedTensor


def run(batch=16384, epochs=300, lr=1e-2, use_graph=False, device="cuda", verbose=True):
    truth = DifferentiableKUKAiiwa(device=device)
    model = DifferentiableKUKAiiwa(device=device)
    model.make_link_param_learnable("iiwa_link_1", "trans", UnconstrainedTensor(dim1=1, dim2=3))
    model.make_link_param_learnable("iiwa_link_1", "rot_angles", UnconstrainedTensor(dim1=1, dim2=3))
    q, _, _ = _common.sample_states(truth, batch, seed=1)
    with torch.no_grad():
        target, _ = truth.compute_forward_kinematics(q, "iiwa_link_ee")
    opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=use_graph)

    def step():
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
        print("%.1f us per step (%s)" % ((time.perf_counter() - t0) / epochs * 1e6, "hipGraph" if use_graph else "eager"))
        print("learned trans      ", model._bodies[model._name_to_idx_map["iiwa_link_1"]].trans().detach().cpu().numpy().ravel())
        print("ground-truth trans ", truth._bodies[truth._name_to_idx_map["iiwa_link_1"]].trans().detach().cpu().numpy().ravel())
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--graph", action="store_true", help="replay the training step as a hipGraph")
    a = ap.parse_args()
    run(a.batch, a.epochs, use_graph=a.graph)
