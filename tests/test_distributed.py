"""CPU, world_size 2 (gloo): streams shard across ranks with no data-path collective and the
gather returns the columns in stream order.  The per-rank compute is the CPU oracle here
(the GPU engine cannot run in this container); what is under test is the N>1 plumbing
that bench.py uses on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_streams, out_path):
    for p in (ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle as O
    from emspec import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, hop, frames = 1024, 256, 12
    first, count = shard.stream_shard(rank, world, total_streams)
    pcm = synth.streams(count, n + hop * (frames - 1), first=first)
    cfg = O.make_cfg(n, hop, True)
    _, _, idx = O.batch_f32(cfg, pcm, want=("index",), threads=1)
    got = shard.gather_columns(torch.from_numpy(idx), dst=0)
    if rank == 0:
        np.save(out_path, got.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges():
    sys.path.insert(0, os.path.join(ROOT, "em-spec_amd"))
    from emspec import shard
    assert [shard.stream_shard(r, 8, 512) for r in range(8)] == [(64 * r, 64) for r in range(8)]
    parts = [shard.stream_shard(r, 3, 8) for r in range(3)]
    assert parts == [(0, 3), (3, 3), (6, 2)]
    assert sum(c for _, c in parts) == 8


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process(tmp_path):
    import oracle as O
    from emspec import synth
    out = str(tmp_path / "gathered.npy")
    total = 4
    mp.spawn(_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    got = np.load(out)
    n, hop, frames = 1024, 256, 12
    pcm = synth.streams(total, n + hop * (frames - 1))
    _, _, idx = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("index",), threads=1)
    assert got.shape == idx.shape
    assert np.array_equal(got, idx)
