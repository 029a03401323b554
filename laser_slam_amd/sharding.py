"""Multi-GPU layout of the ICP hot path.

One process per GPU.  Each ``icp_.compute`` call is independent of every other
(laser_slam/src/laser_track.cpp:496 keeps no state across calls), so scan pairs are sharded over the
ranks with NO data-path collective; the only communication is the metric reduction of the benchmark
(max elapsed time, summed unit count) through ``torch.distributed`` (RCCL on GPUs, gloo in CPU tests).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple


def pairs_of_rank(n_pairs: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: pair i -> rank i mod world (BASELINE config 3: 256 pairs over 8 GPUs)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_pairs, world))


def run_shard(pair_ids: Sequence[int], align_pair: Callable[[int], object]) -> List[Tuple[int, object]]:
    """Run this rank's pairs through ``align_pair`` (the HIP path in production)."""
    return [(i, align_pair(i)) for i in pair_ids]


def aggregate_throughput(local_units: int, local_elapsed_s: float, device=None):
    """(total units over all ranks, max elapsed over ranks, units/s).  No-op without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_units, local_elapsed_s, local_units / local_elapsed_s
    t = torch.tensor([local_elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return int(u.item()), float(t.item()), float(u.item() / t.item())


def gather_results(local: List[Tuple[int, object]]):
    """All ranks' (pair id, result) lists merged and sorted by pair id (control plane only)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sorted(local, key=lambda kv: kv[0])
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    merged = [kv for part in out for kv in part]
    return sorted(merged, key=lambda kv: kv[0])


def split_shard(n_points: int, rank: int, world: int):
    """Contiguous query shard of rank `rank` for the split-scan mode (BASELINE config 4): slice object."""
    # balanced: the first n % world ranks get one point more, so every rank owns >= 1 point whenever n >= world
    # (an empty shard would leave its rank out of the per-iteration collectives)
    if n_points < world:
        raise ValueError(f"split-scan mode needs at least one point per rank ({n_points} points, {world} ranks)")
    base, extra = divmod(n_points, world)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def init_split_comm(handle, device=None):
    """Give `handle` (laser_slam_amd.icp.IcpHandle) an RCCL communicator spanning the default
    torch.distributed group: rank 0 creates the unique id, everybody receives it by broadcast."""
    import torch
    import torch.distributed as dist
    from . import icp
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    uid = icp.comm_unique_id() if rank == 0 else bytes(128)
    if world > 1:
        t = torch.tensor(list(uid), dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0)
        uid = bytes(t.cpu().tolist())
    handle.comm_init(rank, world, uid)
    return rank, world
