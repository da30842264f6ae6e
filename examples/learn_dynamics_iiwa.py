#!/usr/bin/env python3
"""System identification of a KUKA iiwa from inverse dynamics (the workload of the reference's
examples/learn_dynamics_iiwa.py and of its L4DC notebook): mass, centre of mass and inertia matrix of the seven moving
links are learnable, the loss is the MSE of the predicted joint torques against a ground-truth model.  Forward through
drm_rnea, backward through drm_rnea_backward (hand-written adjoint sweeps, deterministic batch reduction).

    python examples/learn_dynamics_iiwa.py [--batch 4096] [--epochs 300] [--spd] [--graph] [--fused-adam]

The parameter modules (PositiveScalar, the inertia-matrix modules) are evaluated and differentiated inside the table kernels (ABI 13,
drm_walk_table_links): a step is 12 kernels, five of them this package's; --graph --fused-adam replays it in ~45 us at batch 256.
"""
import argparse
import time

import _common  # noqa: F401
import torch

from differentiable_robot_model_amd import DifferentiableKUKAiiwa
from differentiable_robot_model_amd.rigid_body_params import PositiveScalar, SymmPosDef3DInertiaMatrixNet, UnconstrainedTensor


def run(batch=4096, epochs=300, lr=1e-3, spd=False, use_graph=False, device="cuda", verbose=True, fused_adam=False):
    torch.manual_seed(0)
    truth = DifferentiableKUKAiiwa(device=device)
    model = DifferentiableKUKAiiwa(device=device)
    for k in range(1, 8):
        link = "iiwa_link_%d" % k
        model.make_link_param_learnable(link, "mass", PositiveScalar())
        model.make_link_param_learnable(link, "com", UnconstrainedTensor(dim1=1, dim2=3))
        model.make_link_param_learnable(link, "inertia_mat", SymmPosDef3DInertiaMatrixNet() if spd else UnconstrainedTensor(dim1=3, dim2=3))
    q, qd, qdd = _common.sample_states(truth, batch, seed=2)
    with torch.no_grad():
        target = truth.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
    opt = torch.optim.Adam(model.parameters(), lr=lr, capturable=use_graph, fused=True if fused_adam else None)

    def step():
        tau = model.compute_inverse_dynamics(q, qd, qdd, include_gravity=True, use_damping=True)
        loss = torch.nn.functional.mse_loss(tau, target)
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
                print("epoch %5d  loss %.4e" % (epoch, history[-1]))
    _common.sync(device)
    if verbose:
        print("%.1f us per step = %.0f it/s (%s)" % ((time.perf_counter() - t0) / epochs * 1e6, epochs / (time.perf_counter() - t0),
                                                   "hipGraph" if use_graph else "eager"))
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--spd", action="store_true", help="symmetric positive definite inertia parametrisation")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--fused-adam", action="store_true", help="torch's fused Adam (two kernels) instead of its default foreach implementation")
    a = ap.parse_args()
    run(a.batch, a.epochs, spd=a.spd, use_graph=a.graph, fused_adam=a.fused_adam)
