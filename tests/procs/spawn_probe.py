"""Rank body of tests/test_distributed.py::test_torchrun_spawn_path: launched through distributed.torchrun_command (the
line `bench.py --gpus N` re-executes itself with), initialises gloo from the torchrun environment and records what it sees."""
import os
import sys

import torch
import torch.distributed as dist

out_dir = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
probe = torch.ones(1)
dist.all_reduce(probe)
world, rank = dist.get_world_size(), dist.get_rank()
assert int(os.environ["WORLD_SIZE"]) == world and int(os.environ["RANK"]) == rank
with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
    f.write("%d %d %s" % (world, int(probe.item()), os.environ["MASTER_ADDR"]))
dist.destroy_process_group()
