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


class PeerGather:
    """One-sided gather of the fused FK + inverse-dynamics launch's outputs over the GPUs of one node ("direct peer
    writes"): every rank owns ONE buffer holding the gathered arrays tau [G, n] | pos [G, 3] | quat [G, 4] of
    the global batch; the ranks exchange the buffers' IPC handles ONCE (hipIpcGetMemHandle / OpenMemHandle behind torch's CUDA-tensor
    sharing; HSA_ENABLE_IPC_MODE_LEGACY=0 on this stack) and every rank's launch then writes its own rows into its own buffer AND
    into the peers' (drm_fk_rnea_put: for a 7-DoF arm with its own fused kernel the stores leave from the kernel's epilogue, tile
    by tile, while the walk runs — no collective launch, no second pass over the data).

        pg = PeerGather(batch, n_dofs, device, mode="all")              # collective: every rank of the group calls it
        plan = model.plan_fk_and_inverse_dynamics(q, qd, qdd, link, outputs=pg.outputs(), put=pg.put())
        plan.launch(); torch.cuda.synchronize(); dist.barrier()         # one-sided: the consumer synchronises with the writers
        tau, pos, quat = pg.gathered()                                  # all G rows

    mode: "all" every rank receives tau | pos | quat of every row; "tau": the torques only; "root": rank 0 receives everything,
    the others only keep their own rows.  CPU tensors (the gloo tests): the buffers are shared-memory tensors and the host build of
    the ABI copies into them.  At most backend.MAX_PEERS + 1 = 9 ranks."""

    def __init__(self, batch: int, n_dofs: int, device, mode: str = "all", group=None):
        import pickle

        from torch.multiprocessing.reductions import reduce_tensor

        from . import backend
        if mode not in ("all", "tau", "root"):
            raise ValueError("mode must be all, tau or root")
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if self.world - 1 > backend.MAX_PEERS:
            raise ValueError("PeerGather serves up to %d ranks" % (backend.MAX_PEERS + 1))
        self.batch, self.n, self.mode, self.group = int(batch), int(n_dofs), mode, group
        self.lo, self.hi = self.bounds(self.batch, self.world, self.rank)
        device = torch.device(device)
        self.flat = torch.zeros(self._off(3), dtype=torch.float32, device=device)
        if device.type == "cpu":
            torch.multiprocessing.set_sharing_strategy("file_system")      # (handles that travel as plain data, not as file descriptors)
            self.flat.share_memory_()
        # torch's own tensor-sharing reducers (what torch.multiprocessing sends over its queues): an IPC handle for device memory, the
        # name of the shared-memory file for a host tensor — as bytes, so that they can travel through all_gather_object
        from multiprocessing.reduction import ForkingPickler
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(ForkingPickler.dumps(reduce_tensor(self.flat))), group=group)
        self._peers = {}
        for r, blob in enumerate(handles):
            if r != self.rank and self._sends_to(r):
                fn_r, args_r = pickle.loads(blob)
                self._peers[r] = fn_r(*args_r)          # the peer's buffer, mapped into this process (kept alive here)
        dist.barrier(group=group)                       # every handle is open before anybody writes

    @staticmethod
    def bounds(batch: int, world: int, rank: int) -> Tuple[int, int]:
        """Rows [lo, hi) of `rank`: contiguous shards whose first row is a multiple of 4, so that every shard's slice of every
        gathered array starts on a 16-byte boundary (what the kernels' 16-byte stores and the plans' output checks ask for); the
        last ranks of a ragged batch get fewer rows (shard_bounds, which balances to one row, does not align)."""
        per = ((int(batch) + world - 1) // world + 3) & ~3
        lo = min(int(batch), rank * per)
        return lo, min(int(batch), lo + per)

    def _off(self, k: int) -> int:
        """Start (in floats) of array k of the buffer: 0 tau, 1 pos, 2 quat, 3 = the end; every array on a 16-byte boundary."""
        sizes = (self.batch * self.n, self.batch * 3, self.batch * 4)
        return sum((x + 3) & ~3 for x in sizes[:k])

    def _sends_to(self, r: int) -> bool:
        return r == 0 if self.mode == "root" else True

    def _arrays(self, flat):
        G, n = self.batch, self.n
        return (flat[self._off(0):self._off(0) + G * n].view(G, n), flat[self._off(1):self._off(1) + G * 3].view(G, 3),
                flat[self._off(2):self._off(2) + G * 4].view(G, 4))

    def gathered(self):
        """(tau [G, n], pos [G, 3], quat [G, 4]) of this rank's buffer (complete once every writer has synchronised)."""
        return self._arrays(self.flat)

    def outputs(self):
        """This rank's own rows inside its buffer: what the launch writes locally (no copy of the own shard afterwards)."""
        return tuple(a[self.lo:self.hi] for a in self._arrays(self.flat))

    def put(self):
        """The DrmPut of this rank's launches: the peers' arrays (tau only in mode "tau") and this shard's first row."""
        from . import backend
        put = backend.DrmPut()
        put.row_offset = self.lo
        for k, (r, flat) in enumerate(sorted(self._peers.items())):
            tau, pos, quat = self._arrays(flat)
            put.tau[k] = tau.data_ptr()
            if self.mode != "tau":
                put.pos[k], put.quat[k] = pos.data_ptr(), quat.data_ptr()
        put.n_peers = len(self._peers)
        return put

    def close(self):
        """Unmap the peers' buffers (collective: the owners must outlive the mappings)."""
        dist.barrier(group=self.group)
        self._peers.clear()
        dist.barrier(group=self.group)


def gather_model_us(mode: str, shard_bytes: int, tau_bytes: int, world: int, link_gbs: float = 153.0) -> float:
    """What one exchange of a config-3 step should cost over xGMI (a fully connected mesh, one ~153 GB/s link per peer, all
    links of a GPU busy in parallel): every receiving GPU takes one peer's block per link, so the time is ONE block over ONE
    link — (N - 1) blocks arrive over (N - 1) links at once.  `all` / `root`: the whole shard (tau | pos | quat, 56 B per row),
    `tau`: the torques only (28 B per row), `none`: the outputs stay sharded.  Collective launch latency (~10 us) not included."""
    if world <= 1 or mode == "none":
        return 0.0
    return {"all": shard_bytes, "root": shard_bytes, "tau": tau_bytes, "p2p": shard_bytes, "p2p_tau": tau_bytes}[mode] / (link_gbs * 1e3)


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
