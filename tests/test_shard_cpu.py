"""CPU, world_size 2 over gloo: the N > 1 path of the benchmark / experiments -- contiguous scene shards, no
collective inside a step, one gather at the end -- gives exactly the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from diffsdfsim_amd import scenes, sharding
    from diffsdfsim_amd.engine import BatchEngine
    from emu import emu
    n_total = 3
    spec = scenes.sphere_drop(n_total, seed=5, floor_dims=(4.0, 1.0, 4.0))
    lo, hi = sharding.shard_range(n_total, rank, world)
    E = BatchEngine(sharding.shard_spec(spec, lo, hi), backend=emu.EmuBackend(), maxc=32, max_pc=16)
    for _ in range(3):
        E.step()
    full = sharding.gather_scenes(torch.tensor(E.get("pose")), n_total, dist)
    if rank == 0:
        np.save(out, full.numpy())
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(tmp_path):
    from diffsdfsim_amd import scenes, sharding
    from diffsdfsim_amd.engine import BatchEngine
    from emu import emu
    assert [sharding.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, 29500 + os.getpid() % 500, out), nprocs=2, join=True)
    E = BatchEngine(scenes.sphere_drop(3, seed=5, floor_dims=(4.0, 1.0, 4.0)), backend=emu.EmuBackend(), maxc=32, max_pc=16)
    for _ in range(3):
        E.step()
    assert np.array_equal(np.load(out), E.get("pose"))
