"""Shared bits of the examples: the package on sys.path, joint-state sampling inside the URDF limits, a hipGraph'ed step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sample_states(model, n, seed=0, vel=0.2, acc=0.4):
    """q ~ U(limits), qd ~ U(+-vel * vmax), qdd ~ U(+-acc * vmax) on the model's device."""
    lim = model.get_joint_limits()
    dev = model._device
    lo = torch.tensor([j["lower"] for j in lim], device=dev)
    hi = torch.tensor([j["upper"] for j in lim], device=dev)
    vmax = torch.tensor([j["velocity"] for j in lim], device=dev)
    gen = torch.Generator(device=dev).manual_seed(seed)
    u = lambda: torch.rand(n, len(lim), device=dev, generator=gen)
    return lo + (hi - lo) * u(), vel * vmax * (2 * u() - 1), acc * vmax * (2 * u() - 1)


def graphed(step, optimizer, warmup=3):
    """Capture one training step (zero_grad + forward + backward + optimizer.step) into a hipGraph and return a callable
    that replays it.  The step must not allocate new parameters; Adam needs ``capturable=True``."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            optimizer.zero_grad(set_to_none=True)
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    optimizer.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = step()
    return lambda: (graph.replay(), loss)[1]


def sync(device):
    """Wait for the device's queued work (a HIP device; nothing to wait for on the CPU, where calls return when they are done)."""
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
