"""Multi-GPU sharding of independent audio streams.

Streams never exchange data during compute, so the path shards by stream with no
data-path collective; the single collective is the gather of finished columns
to one rank (north_star: "a single RCCL gather over xGMI to collect finished
columns").  One process per GPU; torch.distributed backend "nccl" is RCCL on ROCm,
"gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def stream_shard(rank, world, total_streams):
    """Contiguous block of streams owned by `rank`: (first, count).  512 streams / 8 ranks -> 64 each;
    a remainder goes to the lowest ranks."""
    base, rem = divmod(total_streams, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def gather_columns(local, dst=0, group=None):
    """Gather every rank's finished columns [S_local, C, R(,4)] to `dst`, in stream order.
    Returns the concatenated tensor on dst, None elsewhere.  All ranks must hold the same S_local."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    return torch.cat(bufs, dim=0) if rank == dst else None


def gather_columns_into(local, out_list, dst=0, group=None):
    """Same, into preallocated per-rank buffers (out_list on dst, None elsewhere): no allocation in the timed path."""
    dist.gather(local, out_list if dist.get_rank(group) == dst else None, dst=dst, group=group)


def comm_setup(engine, rank, world, group=None):
    """Create libemspec's own RCCL communicator on `engine` (emspec_comm_init): rank 0 draws the 128-byte id and
    torch.distributed carries it to the other ranks - the data path itself never goes through torch."""
    import emspec
    box = [emspec.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0], rank, world)
