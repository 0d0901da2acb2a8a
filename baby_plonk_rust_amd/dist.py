"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

MSM  shards by contiguous POINT RANGE: rank r keeps SRS[lo_r, hi_r) resident in its HBM and receives
     the matching scalar slice; each rank runs the whole Pippenger pipeline on its slice and produces ONE 144-byte
     projective partial.  Elliptic-curve addition is not an RCCL reduction operator, so the "reduce" is a single
     all-gather of 144 B per rank followed by world-1 complete additions on every rank (bp_g1_sum_partials).
NTT  shards by INDEPENDENT COLUMNS: polynomial j of a batch belongs to rank j mod world; no collective inside a
     transform.  all_gather_columns() is provided for callers that need every column everywhere afterwards.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import api


def shard_range(n, rank, world):
    """contiguous point range [lo, hi) of rank `rank`; the first n % world ranks get one extra point"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def column_owner(j, world):
    return j % world


def my_columns(n_columns, rank, world):
    return list(range(rank, n_columns, world))


def allgather_partials(partial144, device=None, group=None):
    """the one collective of the MSM path: every rank contributes 144 bytes, every rank receives world*144"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bytes(partial144)
    world = dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    mine = torch.frombuffer(bytearray(partial144), dtype=torch.uint8).to(dev)
    out = torch.empty(world * 144, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu().numpy().tobytes()


def combine_partials(partial144, device=None, group=None):
    """all-gather + world-1 complete additions + one affine normalisation -> 96-byte encoding, same on every rank"""
    return api.sum_partials(allgather_partials(partial144, device, group))


def msm_sharded(ctx, srs_handle_local, scalars_local, device_ptr=None, n=None, group=None):
    """sum over ALL ranks' (point, scalar) pairs; `srs_handle_local` is this rank's point-range shard"""
    part = ctx.msm_partial(srs_handle_local, scalars_local, device_ptr=device_ptr, n=n)
    return combine_partials(part, torch.device("cuda", ctx.device) if torch.cuda.is_available() else None, group)


def all_gather_columns(local_columns, n_columns, group=None):
    """local_columns: dict {column index: uint64 array [N, 4]} of this rank's finished columns -> list of all columns"""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local_columns[j] for j in range(n_columns)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gathered = [None] * world
    dist.all_gather_object(gathered, {j: np.asarray(v) for j, v in local_columns.items()}, group=group)
    merged = {}
    for d in gathered:
        merged.update(d)
    return [merged[j] for j in range(n_columns)]
