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
        # the flat gather bench.py --config 3 uses for its tau | pos | quat buffer: rank r's block at out[r * len:]
        from differentiable_robot_model_amd.distributed import all_gather_flat
        flat = torch.full((5,), float(rank)) + torch.arange(5.0) / 10
        out = torch.empty(world * 5)
        all_gather_flat(out, flat)
        want_all = torch.cat([torch.full((5,), float(r)) + torch.arange(5.0) / 10 for r in range(world)])
        assert torch.equal(out, want_all)
        # ... and the gather to ONE rank (bench.py --config 3 --gather root): only the destination receives
        from differentiable_robot_model_amd.distributed import gather_flat
        dst = world - 1
        got = torch.full((world * 5,), -1.0)
        gather_flat(got if rank == dst else None, flat, dst=dst)
        assert torch.equal(got, want_all) if rank == dst else bool((got == -1.0).all())
        open(os.path.join(result_dir, "ok%d" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [64, 101])
def test_shard_and_gather_world_size_2(tmp_path, batch):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), batch, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _peer_worker(rank, world, port, batch, mode, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from differentiable_robot_model_amd.distributed import PeerGather
        from helpers import load_model, sample_states
        torch.set_num_threads(1)
        m = load_model("panda_no_gripper", "cpu")          # (libdrm_cpu.so: the host build of drm_fk_rnea_put copies into the peers' arrays)
        link = "panda_virtual_ee_link"
        q, qd, qdd = (torch.from_numpy(a) for a in sample_states(m, batch, seed=9))      # every rank builds the same global batch
        pg = PeerGather(batch, m._n_dofs, "cpu", mode=mode)
        lo, hi = pg.lo, pg.hi
        spans = [PeerGather.bounds(batch, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == batch and all(a[1] == b[0] and b[0] % 4 == 0 for a, b in zip(spans, spans[1:]))
        assert pg.put().n_peers == (world - 1 if mode != "root" else (0 if rank == 0 else 1)) and pg.put().row_offset == lo
        plan = m.plan_fk_and_inverse_dynamics(q[lo:hi].clone(), qd[lo:hi].clone(), qdd[lo:hi].clone(), link, outputs=pg.outputs(), put=pg.put())
        for _ in range(2):          # (a second launch overwrites the same rows: nothing accumulates)
            plan.launch()
        dist.barrier()              # one-sided puts: the consumer synchronises with the writers
        tau, pos, quat = pg.gathered()
        want = m.compute_fk_and_inverse_dynamics(q, qd, qdd, link)
        rows = slice(0, batch) if (mode != "root" or rank == 0) else slice(lo, hi)       # (root: the others keep their own rows only)
        assert torch.equal(tau[rows], want[0][rows])
        if mode == "tau" :
            assert torch.equal(pos[lo:hi], want[1][lo:hi]) and torch.equal(quat[lo:hi], want[2][lo:hi])
            other = [r for r in range(world) if r != rank][0]
            olo, ohi = PeerGather.bounds(batch, world, other)
            assert ohi == olo or float(pos[olo:ohi].abs().sum()) == 0.0                   # (the peers sent their torques only)
        else:
            assert torch.equal(pos[rows], want[1][rows]) and torch.equal(quat[rows], want[2][rows])
        pg.close()
        open(os.path.join(result_dir, "ok%d" % rank), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch,mode", [(2, 256, "all"), (3, 130, "all"), (2, 131, "tau"), (3, 64, "root")])
def test_one_sided_gather_world_size_n(tmp_path, cpu_library, world, batch, mode):
    """distributed.PeerGather + drm_fk_rnea_put (ABI 11) with world_size-N gloo processes on the CPU: every rank's launch writes its
    rows into its own gathered buffer and into the peers' (shared-memory tensors here, IPC-mapped device memory on a GPU node);
    after one barrier every receiving rank holds all rows, bit for bit what ONE process computes for the whole batch."""
    mp.spawn(_peer_worker, args=(world, _free_port(), batch, mode, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok%d" % r for r in range(world)]


def test_torchrun_spawn_path(tmp_path):
    """`python bench.py --gpus N` without a rendezvous re-executes itself through distributed.torchrun_command; the same
    line must bring up N ranks that see each other (gloo here, RCCL on the GPU node)."""
    import subprocess
    import sys
    from differentiable_robot_model_amd.distributed import torchrun_command
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "procs", "spawn_probe.py")
    cmd = torchrun_command(2, probe, [str(tmp_path)])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "127.0.0.1" in cmd
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    subprocess.run(cmd, check=True, env=env, timeout=300)
    assert sorted(os.listdir(tmp_path)) == ["rank0", "rank1"]
    for r in (0, 1):
        world, seen, addr = open(os.path.join(str(tmp_path), "rank%d" % r)).read().split()
        assert (world, seen, addr) == ("2", "2", "127.0.0.1")


def test_bench_refuses_more_gpus_than_the_node_has():
    """`bench.py --gpus 2` on a box with fewer than two GPUs must fail loudly instead of reporting n_gpus = 1."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs")
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and '"n_gpus"' not in r.stdout
    # a rank count that disagrees with --gpus is refused as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "2", "--warmup", "1"], env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def _grad_worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from differentiable_robot_model_amd.distributed import all_reduce_gradients
        torch.manual_seed(0)
        # a "learnable link": trans [1,3] and rot_angles [1,3] (learn_kinematics_of_iiwa.py), loss = mean over the batch
        trans = torch.nn.Parameter(torch.randn(1, 3)); rot = torch.nn.Parameter(torch.randn(1, 3))
        frozen = torch.nn.Parameter(torch.randn(2), requires_grad=False)
        x = torch.randn(10, 3)
        lo, hi = shard_bounds(10, world, rank)
        loss = ((x[lo:hi] * trans).sum() + (x[lo:hi].pow(2) * rot).sum()) / 10.0     # this rank's share of the mean
        loss.backward()
        all_reduce_gradients([trans, rot, frozen])
        want_t = x.sum(0, keepdim=True) / 10.0
        want_r = x.pow(2).sum(0, keepdim=True) / 10.0
        assert torch.allclose(trans.grad, want_t, atol=1e-6) and torch.allclose(rot.grad, want_r, atol=1e-6)
        assert frozen.grad is None
        open(os.path.join(result_dir, "ok%d" % rank), "w").close()
    finally:
        dist.destroy_process_group()


def test_all_reduce_gradients_world_size_2(tmp_path):
    """the backward of a batch-sharded step adds ONE all-reduce of the parameter gradients."""
    mp.spawn(_grad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
