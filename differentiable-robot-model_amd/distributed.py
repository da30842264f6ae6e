"""Batch sharding over the GPUs of one node (one process per GPU, torch.distributed).

Every sample of the FK / Jacobian / RNEA path is independent and the per-robot
constants are < 4 KB, so the path shards trivially: rank r owns the contiguous
rows [lo_r, hi_r) of the batch, the walk tables are replicated, and NO collective
is needed during compute.  The only exchange the path can have is the optional
gather of the outputs (an MPC / particle batch that has to be re-assembled on
every rank or on rank 0); it is a single `all_gather_into_tensor` per output —
RCCL over xGMI with backend "nccl" on ROCm, gloo on CPU (tests).
"""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [lo, hi) of `rank`; the first (batch % world) ranks get one extra row."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(batch), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(t: torch.Tensor, world_size: int, rank: int) -> torch.Tensor:
    """The rows of a [B, ...] tensor that `rank` owns (a view, no copy)."""
    lo, hi = shard_bounds(t.shape[0], world_size, rank)
    return t[lo:hi]


def all_gather_rows(local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """Re-assemble a [B, ...] tensor from per-rank row shards produced with `shard_bounds`.

    Equal shards use one `all_gather_into_tensor`; ragged shards (B % world != 0) are padded by
    one row to the largest shard first.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], sizes[rank]))
    most = max(sizes)
    tail = tuple(local.shape[1:])
    send = local.contiguous()
    if send.shape[0] < most:
        pad = torch.zeros((most - send.shape[0],) + tail, dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    out = torch.empty((world * most,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if min(sizes) == most:
        return out
    out = out.reshape((world, most) + tail)
    return torch.cat([out[r, :sizes[r]] for r in range(world)], dim=0)


_host_stage = {}


def all_gather_flat(out: torch.Tensor, local: torch.Tensor, group=None) -> None:
    """`all_gather_into_tensor(out, local)` for equal-sized contiguous shards, whatever the backend: RCCL ("nccl") moves
    the device buffers themselves over xGMI; under gloo (CPU tests, or N ranks sharing one GPU in bench.py's test mode)
    device tensors are staged through pinned host buffers, because gloo's collectives run on the host."""
    if dist.get_backend(group) != "gloo" or not local.is_cuda:
        dist.all_gather_into_tensor(out, local, group=group)
        return
    key = (out.numel(), local.numel(), local.dtype)
    if key not in _host_stage:
        _host_stage[key] = (torch.empty(out.numel(), dtype=local.dtype).pin_memory(),
                            torch.empty(local.numel(), dtype=local.dtype).pin_memory())
    h_out, h_loc = _host_stage[key]
    h_loc.copy_(local.reshape(-1), non_blocking=False)
    dist.all_gather_into_tensor(h_out, h_loc, group=group)
    out.reshape(-1).copy_(h_out, non_blocking=False)


def gather_flat(out, local: torch.Tensor, dst: int = 0, group=None) -> None:
    """Gather equal-sized contiguous shards on rank `dst` only (`out`: [world * local.numel()] there, ignored elsewhere): the
    MPC / particle case in which ONE rank consumes the re-assembled batch — the other ranks only send, so the node moves
    (N - 1) shards instead of N (N - 1).  RCCL moves device buffers; gloo stages device tensors through the host (as above)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) != "gloo" or not local.is_cuda:
        parts = list(out.reshape(world, -1).unbind(0)) if rank == dst else None
        dist.gather(local.reshape(-1), parts, dst=dst, group=group)
        return
    h_loc = local.reshape(-1).cpu()
    parts = [torch.empty_like(h_loc) for _ in range(world)] if rank == dst else None
    dist.gather(h_loc, parts, dst=dst, group=group)
    if rank == dst:
        out.reshape(world, -1).copy_(torch.stack(parts))


def gather_model_us(mode: str, shard_bytes: int, tau_bytes: int, world: int, link_gbs: float = 153.0) -> float:
    """What one exchange of a config-3 step should cost over xGMI (a fully connected mesh, one ~153 GB/s link per peer, all
    links of a GPU busy in parallel): every receiving GPU takes one peer's block per link, so the time is ONE block over ONE
    link — (N - 1) blocks arrive over (N - 1) links at once.  `all` / `root`: the whole shard (tau | pos | quat, 56 B per row),
    `tau`: the torques only (28 B per row), `none`: the outputs stay sharded.  Collective launch latency (~10 us) not included."""
    if world <= 1 or mode == "none":
        return 0.0
    return {"all": shard_bytes, "root": shard_bytes, "tau": tau_bytes}[mode] / (link_gbs * 1e3)


def gather_outputs(outputs: Sequence[torch.Tensor], batch: int, group=None) -> List[torch.Tensor]:
    return [all_gather_rows(t, batch, group) for t in outputs]


def free_port() -> int:
    """A TCP port that is free on 127.0.0.1 right now (rendezvous of a single-node launch)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def torchrun_command(nproc: int, script: str, script_args: Sequence[str], port: int = 0) -> List[str]:
    """The single-node launch line that turns `script` into `nproc` ranks (one per GPU): what `bench.py --gpus N` executes
    when it is started without a rendezvous in its environment.  127.0.0.1, never the container hostname."""
    import sys
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(script_args)


def all_reduce_gradients(params, group=None, average: bool = False) -> None:
    """Sum (or average) the gradients of the learnable link parameters over the ranks of a batch-sharded training step.

    The backward kernels (drm_fk_backward / drm_rnea_backward) leave on every rank the parameter gradients of ITS rows:
    6 .. 12 floats per learnable link.  They are flattened into ONE buffer and reduced with ONE
    all_reduce (latency-bound, < 1 KB for the reference's learn_kinematics_of_iiwa.py workload), then scattered back, so
    every rank steps its optimiser on the gradient of the whole batch.  Parameters without a gradient contribute zeros
    (a rank whose shard is empty still takes part in the collective)."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    world = dist.get_world_size(group)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in params])
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= world
    off = 0
    for p in params:
        k = p.numel()
        g = flat[off:off + k].reshape(p.shape).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += k
