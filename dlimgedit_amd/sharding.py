"""Multi-GPU orchestration of the SAM path: independent images, one process per GPU.

The reference has no multi-device support at all.  Images are independent units, so the
path shards without any data-path collective: image i goes to rank i mod G; the only
communication is (1) the barrier / max-over-ranks of the elapsed time for benchmarking and (2) an
optional gather of the finished masks to one rank (RCCL all_gather of u8 tensors when they live on
the device, object gather otherwise).  Works with any torch.distributed backend; the CPU tests run it
on gloo with world_size 2.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import numpy as np


def assign(n_items: int, world: int) -> List[List[int]]:
    """Static round-robin partition: item i -> rank i % world."""
    if world <= 0:
        raise ValueError("world size must be positive")
    return [list(range(r, n_items, world)) for r in range(world)]


def my_items(n_items: int, rank: int, world: int) -> List[int]:
    return assign(n_items, world)[rank]


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (elapsed time); identity without an initialised group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, device=None) -> List[float]:
    """Every rank's host scalar, in rank order, on every rank (a straggler shows as one slow entry where the MAX hides it)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    world = dist.get_world_size()
    mine = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    out = torch.empty(world, dtype=torch.float64, device=device or "cpu")
    dist.all_gather_into_tensor(out, mine)
    return [float(v) for v in out.cpu().tolist()]


def gather_results(local: Dict[int, np.ndarray], n_items: int, root: int = 0):
    """Collects per-item host arrays on `root` in item order (None elsewhere).  Every item must be
    produced by exactly one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        parts = [local]
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, local)
        if dist.get_rank() != root:
            return None
    merged: Dict[int, np.ndarray] = {}
    for part in parts:
        for k, v in part.items():
            if k in merged:
                raise RuntimeError(f"item {k} was produced by more than one rank")
            merged[k] = v
    missing = [i for i in range(n_items) if i not in merged]
    if missing:
        raise RuntimeError(f"items {missing[:5]} were produced by no rank")
    return [merged[i] for i in range(n_items)]


def gather_device_masks(mask_tensor, n_items: int = None):
    """Collective gather of equally sized u8 mask tensors that live on the device (the optional 'all masks on one
    device' mode of SURVEY.md §8e; RCCL over xGMI when the group's backend is nccl, 1 MiB per 1024x1024 mask).
    mask_tensor: torch.uint8 [B_local, H, W]; rank r holds items r, r + world, r + 2*world, ... in that order
    (`assign`), padded to the same B_local on every rank.  Returns [n_items, H, W] in ITEM order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    b_local = mask_tensor.shape[0]
    out = torch.empty((world * b_local,) + tuple(mask_tensor.shape[1:]), dtype=mask_tensor.dtype,
                      device=mask_tensor.device)
    dist.all_gather_into_tensor(out, mask_tensor.contiguous())
    # rank-major [r][b] -> item-major [b][r]: item index = b * world + r
    out = out.reshape((world, b_local) + tuple(mask_tensor.shape[1:])).transpose(0, 1)
    out = out.reshape((world * b_local,) + tuple(mask_tensor.shape[1:]))
    return out if n_items is None else out[:n_items]


def count_ranks(device=None) -> int:
    """Sum-all-reduce of ones: the number of ranks that really take part in the group's collectives."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def run_sharded(n_items: int, fn: Callable[[int], np.ndarray], rank: int, world: int) -> Dict[int, np.ndarray]:
    """Applies fn to this rank's share of the items."""
    return {i: fn(i) for i in my_items(n_items, rank, world)}
