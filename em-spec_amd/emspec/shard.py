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


def root_light_counts(world, total_streams, root, root_count):
    """Shard sizes when `root` takes root_count streams and the other ranks split the rest as evenly as they can (the
    remainder goes to the lowest of them).  The root also receives and expands every other rank's columns, so an equal
    split makes it the slowest rank; bench.py tries a few of these splits and keeps the fastest."""
    if world == 1:
        return [total_streams]
    root_count = max(0, min(int(root_count), total_streams))
    base, rem = divmod(total_streams - root_count, world - 1)
    counts, k = [], 0
    for r in range(world):
        if r == root:
            counts.append(root_count)
        else:
            counts.append(base + (1 if k < rem else 0))
            k += 1
    return counts


def trial_splits(world, total_streams, chunks):
    """The stream splits bench.py times at N > 1 (three steps each, the fastest is kept), heaviest root first: equal shards,
    then the other ranks 1/32, 1/16, 3/32 and 1/8 heavier than an equal share, the root taking what is left (it also
    receives every other rank's columns).  A split that would leave the root fewer streams than the step has chunks is
    left out.  512 streams on 8 ranks: roots of 64, 50, 36, 22 and 8 streams."""
    S = total_streams // world
    out = []
    for other in sorted({S, S + S // 32, S + S // 16, S + 3 * S // 32, S + S // 8}):
        root_count = total_streams - (world - 1) * other
        if root_count < max(1, chunks):
            continue
        out.append(root_light_counts(world, total_streams, 0, root_count))
    return out


def job_device_bytes(counts, rank, L, C, R, nbuf, nch, packed, wire_bound, root=0):
    """Device memory one rank's job holds for shard sizes `counts` (bench.py's Job): the samples, the float32 dB columns,
    nbuf palette-index buffers and, on the root, the gather's destination - per chunk either the ranks' expanded blocks or
    (packed) a directory + every rank's wire image at its bound (wire_bound(columns, rows) -> bytes).  The CPU tests hold the
    world-8 plans to a stated budget; the MI355X has 288 GB."""
    Sl = counts[rank]
    total = Sl * L * 4 + Sl * C * R * 4 + nbuf * Sl * C * R
    if rank == root and len(counts) > 1:
        world = len(counts)
        chunk = lambda Sx: [(Sx * i // nch, Sx * (i + 1) // nch) for i in range(nch)]
        per_rank = [chunk(c) for c in counts]
        for ci in range(nch):
            if packed:
                total += 256 * (world + 1) + sum(wire_bound((pr[ci][1] - pr[ci][0]) * C, R) for pr in per_rank)
            else:
                total += sum(pr[ci][1] - pr[ci][0] for pr in per_rank) * C * R
    return total


def first_streams(counts):
    """Index of each rank's first stream for shard sizes `counts`."""
    out, acc = [], 0
    for c in counts:
        out.append(acc)
        acc += c
    return out


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


def gather_columns_into(local, out_list, dst=0, group=None, uneven=False):
    """Same, into preallocated per-rank buffers (out_list on dst, None elsewhere): no allocation in the timed path.
    uneven=True: the ranks' shards differ in size (every rank must pass the same flag), so the gather is posted as
    point-to-point transfers; a rank with an empty shard takes no part."""
    rank = dist.get_rank(group)
    if not uneven:
        dist.gather(local, out_list if rank == dst else None, dst=dst, group=group)
        return
    if rank == dst:
        reqs = [dist.irecv(buf, src=r, group=group) for r, buf in enumerate(out_list) if r != dst and buf.numel()]
        if local.numel():
            out_list[dst].copy_(local)
        for q in reqs:
            q.wait()
    elif local.numel():
        dist.send(local, dst=dst, group=group)


def comm_setup(engine, rank, world, group=None):
    """Create libemspec's own RCCL communicator on `engine` (emspec_comm_init): rank 0 draws the 128-byte id and
    torch.distributed carries it to the other ranks - the data path itself never goes through torch."""
    import emspec
    box = [emspec.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0], rank, world)
