"""Multi-GPU sharding of a robot population: robots are independent, so the batch is cut
into contiguous ranges, one per rank (one process per GPU), with no exchange inside a tick.  The only
collective of the path is the all-gather of the per-tick joint torques: on GPUs it is `qrgpu_allgather_tau` of the C ABI (RCCL over
xGMI on a stream of the context's own, csrc/qrgpu_comm.hip); the launcher only has to hand rank 0's 128-byte communicator id to the
other ranks, which `exchange_comm_id` does over torch.distributed's CPU backend (gloo) -- no GPU collective library of the launcher's
is involved.  `allgather_torques` is the same exchange on torch tensors, used by the CPU tests (gloo).
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous [lo, hi) of rank `rank`; sizes differ by at most one robot."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


def interleave_types(n_total, n_types):
    """type_id of robot i for mixed batches (config 5: equal numbers of each type on every rank)."""
    return (np.arange(n_total) % n_types).astype(np.int32)


def allgather_torques(tau_local, n_total, group=None):
    """tau_local: torch tensor [12, n_local] (SoA) on this rank.  Returns [12, n_total] with every rank's
    columns in rank order.  Uneven shards are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    nmax = max(sizes)
    pad = torch.zeros((12, nmax), dtype=tau_local.dtype, device=tau_local.device)
    pad[:, :tau_local.shape[1]] = tau_local
    out = torch.empty((world * 12, nmax), dtype=tau_local.dtype, device=tau_local.device)      # rank-major concatenation along dim 0
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view(world, 12, nmax)
    return torch.cat([out[r, :, :sizes[r]] for r in range(world)], dim=1)


def exchange_comm_id(rank, make_id, group=None):
    """Rank 0 calls make_id() (qrgpu.comm_unique_id: a 128-byte ncclUniqueId blob) and every rank returns that blob
    (broadcast over the already initialised torch.distributed group; gloo is enough: this is host data)."""
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    blob = bytes(box[0])
    if len(blob) != 128:
        raise ValueError("communicator id must be 128 bytes, got %d" % len(blob))
    return blob


def split_gathered(tau_all, n_total, world):
    """tau_all [world][12][nmax] as qrgpu_allgather_tau leaves it (every rank padded to the largest shard) -> [12][n_total]."""
    sizes = shard_sizes(n_total, world)
    if hasattr(tau_all, "cpu"):
        import torch
        return torch.cat([tau_all[r, :, :sizes[r]] for r in range(world)], dim=1)
    return np.concatenate([tau_all[r][:, :sizes[r]] for r in range(world)], axis=1)
