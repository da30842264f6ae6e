#!/usr/bin/env python3
"""Learn link masses of a KUKA iiwa from observed joint accelerations (the workload of the reference's
examples/learn_forward_dynamics_iiwa.py): the loss is on compute_forward_dynamics' qdd.  Forward: one fused kernel (mass
matrix, bias torques, leaf-to-root L^T D L solve); backward by implicit differentiation: one more solve through the same
kernel for lambda = H^-1 dL/dqdd, then the RNEA adjoint with grad_tau = lambda.

    python examples/learn_forward_dynamics_iiwa.py [--batch 4096] [--epochs 300]
"""
import argparse

import _common  # noqa: F401
import torch

from differentiable_robot_model_amd import DifferentiableKUKAiiwa
from differentiable_robot_model_amd.rigid_body_params import PositiveScalar


def run(batch=4096, epochs=300, lr=3e-2, device="cuda", verbose=True):
    torch.manual_seed(0)
    truth = DifferentiableKUKAiiwa(device=device)
    model = DifferentiableKUKAiiwa(device=device)
    links = ["iiwa_link_%d" % k for k in (2, 4, 6)]
    for link in links:
        model.make_link_param_learnable(link, "mass", PositiveScalar(init_param=torch.tensor(2.0)))
    q, qd, _ = _common.sample_states(truth, batch, seed=3)
    torque = 5.0 * (2 * torch.rand(batch, 7, device=device) - 1)
    with torch.no_grad():
        target = truth.compute_forward_dynamics(q, qd, torque, include_gravity=True, use_damping=True)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    history = []
    for epoch in range(epochs):
        opt.zero_grad(set_to_none=True)
        qdd = model.compute_forward_dynamics(q, qd, torque, include_gravity=True, use_damping=True)
        loss = torch.nn.functional.mse_loss(qdd, target)
        loss.backward()
        opt.step()
        if epoch % max(1, epochs // 10) == 0 or epoch == epochs - 1:
            history.append(float(loss.detach()))
            if verbose:
                print("epoch %5d  loss %.4e" % (epoch, history[-1]))
    if verbose:
        for link in links:
            i = model._name_to_idx_map[link]
            print("%s mass: learned %.3f, ground truth %.3f" % (link, float(model._bodies[i].inertia.mass()), float(truth._bodies[i].inertia.mass())))
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=300)
    a = ap.parse_args()
    run(a.batch, a.epochs)
