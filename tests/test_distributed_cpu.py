"""N > 1 orchestration on CPU: world_size-2 gloo processes shard independent items, agree on the
max elapsed time and gather results in order (dlimgedit_amd/sharding.py, used by bench.py)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dlimgedit_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def fn(i):            # stand-in for encode+mask of image i: deterministic "mask"
            rng = np.random.default_rng(i)
            return (rng.integers(0, 2, (4, 5)) * 255).astype(np.uint8)
        local = sharding.run_sharded(n_items, fn, rank, world)
        elapsed = sharding.max_over_ranks(1.0 + rank)          # slowest rank defines the step time
        gathered = sharding.gather_results(local, n_items, root=0)
        # the collective form (what runs over RCCL on the GPUs): equal-sized shares, padded, gathered in item order
        import torch
        b_local = (n_items + world - 1) // world
        share = torch.zeros((b_local, 4, 5), dtype=torch.uint8)
        for slot, i in enumerate(sorted(local)):
            share[slot] = torch.from_numpy(local[i])
        collected = sharding.gather_device_masks(share, n_items)
        np.save(Path(out_dir) / f"collected_{rank}.npy", collected.numpy())
        np.save(Path(out_dir) / f"ranks_{rank}.npy", np.array([sharding.count_ranks()]))
        dist.barrier()
        np.save(Path(out_dir) / f"elapsed_{rank}.npy", np.array([elapsed]))
        np.save(Path(out_dir) / f"mine_{rank}.npy", np.array(sorted(local), dtype=np.int64))
        if rank == 0:
            np.save(Path(out_dir) / "gathered.npy", np.stack(gathered))
        else:
            assert gathered is None
    finally:
        dist.destroy_process_group()


def test_assign_is_a_partition():
    from dlimgedit_amd import sharding
    for n, w in [(64, 8), (5, 2), (3, 8), (0, 4)]:
        parts = sharding.assign(n, w)
        assert sorted(i for p in parts for i in p) == list(range(n))
        assert all(i % w == r for r, p in enumerate(parts) for i in p)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_gloo_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    world, n_items = 2, 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_items, str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "mine_0.npy").tolist() == [0, 2, 4]
    assert np.load(tmp_path / "mine_1.npy").tolist() == [1, 3]
    assert float(np.load(tmp_path / "elapsed_0.npy")[0]) == 2.0 == float(np.load(tmp_path / "elapsed_1.npy")[0])
    gathered = np.load(tmp_path / "gathered.npy")
    assert gathered.shape == (n_items, 4, 5)
    for i in range(n_items):
        rng = np.random.default_rng(i)
        assert np.array_equal(gathered[i], (rng.integers(0, 2, (4, 5)) * 255).astype(np.uint8))
    for r in range(world):      # gather_device_masks: every rank ends up with all masks, in item order
        assert np.array_equal(np.load(tmp_path / f"collected_{r}.npy"), gathered)
        assert int(np.load(tmp_path / f"ranks_{r}.npy")[0]) == world


def test_gather_detects_duplicates_and_gaps():
    from dlimgedit_amd import sharding
    with pytest.raises(RuntimeError, match="no rank"):
        sharding.gather_results({0: np.zeros(1)}, 2)
    assert len(sharding.gather_results({0: np.zeros(1), 1: np.ones(1)}, 2)) == 2
