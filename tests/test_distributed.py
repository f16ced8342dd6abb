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


def test_root_light_split():
    """A lighter shard for the collecting rank (bench.py's N>1 split): the counts cover every stream exactly once."""
    sys.path.insert(0, os.path.join(ROOT, "em-spec_amd"))
    from emspec import shard
    assert shard.root_light_counts(8, 512, 0, 64) == [64] * 8
    assert shard.root_light_counts(8, 512, 0, 36) == [36] + [68] * 7
    assert shard.root_light_counts(8, 512, 0, 8) == [8] + [72] * 7
    assert shard.root_light_counts(4, 256, 0, 50) == [50, 69, 69, 68]
    assert shard.root_light_counts(4, 256, 2, 40) == [72, 72, 40, 72]
    assert shard.root_light_counts(1, 64, 0, 3) == [64]
    for world, total, rc in ((2, 128, 56), (3, 10, 1), (8, 512, 22)):
        counts = shard.root_light_counts(world, total, 0, rc)
        assert sum(counts) == total and counts[0] == rc and max(counts[1:]) - min(counts[1:]) <= 1
        firsts = shard.first_streams(counts)
        assert firsts[0] == 0 and all(firsts[r + 1] == firsts[r] + counts[r] for r in range(world - 1))


def _uneven_worker(rank, world, port, counts, out_path):
    for p in (ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import oracle as O
    from emspec import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, hop, frames = 1024, 256, 12
    first = shard.first_streams(counts)[rank]
    pcm = synth.streams(counts[rank], n + hop * (frames - 1), first=first)
    _, _, idx = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("index",), threads=1)
    local = torch.from_numpy(idx)
    bufs = [torch.empty((c,) + tuple(local.shape[1:]), dtype=torch.uint8) for c in counts] if rank == 0 else None
    for _ in range(2):                                    # twice: the second gather reuses the buffers, as bench.py does
        shard.gather_columns_into(local, bufs, dst=0, uneven=True)
    if rank == 0:
        np.save(out_path, torch.cat(bufs, dim=0).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_uneven_gather_matches_single_process(tmp_path):
    """Shards of different size (1 stream on the collecting rank, 3 on the other) arrive in stream order."""
    import oracle as O
    from emspec import synth
    out = str(tmp_path / "gathered_uneven.npy")
    counts = [1, 3]
    mp.spawn(_uneven_worker, args=(2, _free_port(), counts, out), nprocs=2, join=True)
    got = np.load(out)
    n, hop, frames = 1024, 256, 12
    pcm = synth.streams(sum(counts), n + hop * (frames - 1))
    _, _, idx = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("index",), threads=1)
    assert got.shape == idx.shape and np.array_equal(got, idx)


def _plain_bench(args, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.timeout(300)
def test_plain_bench_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command form) starts the two ranks itself as a
    child `python -m torch.distributed.run`, relays rank 0's ONE JSON line to stdout (everything else to stderr) and exits
    with the child's code.  --dry-run-ranks keeps the ranks off the GPU: rendezvous + one reduction over gloo."""
    import json
    r = _plain_bench(["--gpus", "2", "--steps", "7", "--backend", "gloo", "--dry-run-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout                       # nothing but the line on stdout
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["steps"] == 7 and len(set(b["pids"])) == 2 and os.getpid() not in b["pids"]
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    # a failing rank's exit code ends the parent non-zero
    r = _plain_bench(["--gpus", "2", "--backend", "gloo", "--dry-run-ranks", "--hang-rank", "1"])
    assert r.returncode != 0


def test_plain_bench_invocation_refuses_more_ranks_than_gpus():
    """--gpus N over RCCL needs N devices (RCCL refuses two ranks on one): a JSON error line and a non-zero exit, not a hang."""
    import json
    import torch as _t
    if _t.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the refusal cannot be provoked")
    r = _plain_bench(["--gpus", "2"], timeout=120)
    assert r.returncode != 0
    b = json.loads(r.stdout.splitlines()[0])
    assert "error" in b and b["n_gpus"] == 2 and b["value"] is None


@pytest.mark.timeout(300)
def test_plain_bench_invocation_starts_eight_ranks():
    """The driver's N = 8 command form (`python bench.py --gpus 8`), rehearsed on the CPU: eight ranks meet over gloo, rank 0's
    ONE line reaches stdout.  (RCCL with 8 ranks is the driver's run on an 8-GPU node: rccl.h:220.)"""
    import json
    r = _plain_bench(["--gpus", "8", "--steps", "3", "--backend", "gloo", "--dry-run-ranks"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    b = json.loads(lines[0])
    assert b["n_gpus"] == 8 and len(set(b["pids"])) == 8


@pytest.mark.timeout(300)
def test_failed_job_leaves_one_json_error_line():
    """A rank that dies before rank 0 printed its line: the parent exits non-zero AND leaves one parseable line with the exit
    code and the end of the ranks' output (the driver's record of a failed N = 8 run then carries a diagnosis)."""
    import json
    r = _plain_bench(["--gpus", "2", "--backend", "gloo", "--dry-run-ranks", "--hang-rank", "1", "--hang-at-step", "0"])
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["rc"] == r.returncode and b["value"] is None
    assert any("injected failure of rank 1" in x for x in b["rank_errors"]) and 0 < len(b["stderr_tail"]) <= 2048
    assert '"rank": 1' in r.stderr                      # the rank's own one-line record came first


def test_world8_split_plans_stay_within_the_memory_budget():
    """BASELINE configs[3] (512 streams on 8 GPUs): the five stream splits bench.py times at N = 8 (emspec.shard.trial_splits),
    the chunking of every rank and the root's gather buffers (expanded and packed form) - every rank's device memory stays
    under 48 GB (of the MI355X's 288 GB), the splits cover all 512 streams, the root never has fewer streams than chunks.
    The trials' wall clock is bounded separately at run time: --trial-budget-s (60 s) + the 2-step warm-up."""
    import emspec
    from emspec import shard
    world, S, L, n, hop, R, nbuf, nch = 8, 64, 1 << 22, 4096, 256, 1024, 2, 2
    C = (L - n) // hop + 1
    wire_bound = emspec.wire_bound          # (a host function of the library: no device needed)
    plans = shard.trial_splits(world, world * S, nch)
    assert [p[0] for p in plans] == [64, 50, 36, 22, 8] and len(plans) <= 5
    for counts in plans:
        assert sum(counts) == world * S and len(counts) == world and counts[0] >= nch
        assert max(counts[1:]) - min(counts[1:]) <= 1
        firsts = shard.first_streams(counts)
        assert firsts[0] == 0 and all(firsts[r + 1] == firsts[r] + counts[r] for r in range(world - 1))
        for packed in (True, False):
            worst = max(shard.job_device_bytes(counts, r, L, C, R, nbuf, nch, packed, wire_bound) for r in range(world))
            assert worst < 48e9, (counts, packed, worst)
    # the modelled default when no trial fits the budget is one of the planned splits
    assert shard.root_light_counts(world, world * S, 0, max(nch, world * S - (world - 1) * (S + S // 8))) == plans[-1]
