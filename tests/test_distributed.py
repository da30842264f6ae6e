"""The N>1 path on CPU (not gpu): row sharding + output gather with world_size-2 gloo processes.

The compute itself has no CPU path, so the per-rank "kernel" here is the oracle (checker standing in
for the device launch); what is under test is the partition / gather logic bench.py and multi-GPU
callers use: every row is computed exactly once, by the right rank, and re-assembled in order.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from differentiable_robot_model_amd.distributed import all_gather_rows, gather_outputs, shard_bounds, shard_rows


def test_shard_bounds_cover_the_batch_exactly():
    for batch in (0, 1, 7, 64, 65, 65536, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, batch, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import load_model, sample_states
        from oracle import Oracle
        m = load_model("panda_no_gripper")
        q, qd, qdd = sample_states(m, batch, seed=5)         # every rank builds the same global batch
        orc = Oracle(m._spec)
        ee = m._name_to_idx_map["panda_virtual_ee_link"]
        mine = shard_rows(torch.from_numpy(q), world, rank).numpy()
        lo, hi = shard_bounds(batch, world, rank)
        assert mine.shape[0] == hi - lo
        pos, quat, lin, ang = (torch.from_numpy(a) for a in orc.fk_jacobian(mine, ee, np.float32))
        tau = torch.from_numpy(orc.rnea(mine, qd[lo:hi], qdd[lo:hi], True, True, np.float32))
        full = gather_outputs([pos, quat, lin, ang, tau], batch)
        assert [tuple(t.shape) for t in full] == [(batch, 3), (batch, 4), (batch, 3, 7), (batch, 3, 7), (batch, 7)]
        rp, rq, rl, ra = orc.fk_jacobian(q, ee, np.float32)   # single-process result
        rt = orc.rnea(q, qd, qdd, True, True, np.float32)
        for got, want in zip(full, (rp, rq, rl, ra, rt)):
            assert np.array_equal(got.numpy(), want)
        # ragged integer payload: rows must come back in order
        ids = torch.arange(lo, hi, dtype=torch.int64).reshape(-1, 1)
        assert torch.equal(all_gather_rows(ids, batch).reshape(-1), torch.arange(batch))
        open(os.path.join(result_dir, "ok%d" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [64, 101])
def test_shard_and_gather_world_size_2(tmp_path, batch):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
