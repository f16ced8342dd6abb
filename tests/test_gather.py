"""Multi-GPU gather of finished columns: the wire image (pack / unpack kernels vs the numpy restatement
oracle/wire_ref.py, byte for byte) and the RCCL path itself on one rank (communicator creation, size all-gather,
self send/recv, root-side expand).  More ranks need more GPUs than a box has; the N > 1 control flow is covered with
gloo on the CPU in tests/test_distributed.py."""
import os

import numpy as np
import pytest
import torch

import emspec
import oracle as O
from palette import palette_close
import wire_ref as W
from emspec import synth


def _columns(rng, columns, rows, density):
    idx = (rng.random((columns, rows)) < density) * rng.integers(1, 256, (columns, rows))
    return idx.astype(np.uint8)


@pytest.mark.parametrize("columns,rows,density", [(1, 64, 0.5), (7, 68, 0.3), (300, 256, 0.07), (1025, 1024, 0.06),
                                                  (5000, 1024, 0.0), (40, 1024, 1.0), (3, 4096, 0.2), (2049, 100, 0.5)])
def test_wire_ref_round_trip(columns, rows, density):
    """CPU: the numpy restatement is its own inverse and respects the size bound."""
    idx = _columns(np.random.default_rng(columns * 7 + rows), columns, rows, density)
    w = W.pack(idx)
    assert w.size <= W.bound(columns, rows)
    assert np.array_equal(W.unpack(w, columns, rows), idx)
    assert w.size == W.fixed_bytes(columns, rows) + ((int((idx != 0).sum()) + 15) // 16) * 16


def test_wire_bound_matches_ref():
    for columns, rows in [(1, 64), (1000, 1024), (12345, 68), (7, 4096)]:
        assert emspec.wire_bound(columns, rows) == W.bound(columns, rows)
    assert emspec.wire_bound(10, 1023) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("columns,rows,density", [(1, 64, 0.5), (7, 68, 0.3), (300, 256, 0.07), (1025, 1024, 0.06),
                                                  (5000, 1024, 0.0), (40, 1024, 1.0), (3, 4096, 0.2), (2049, 100, 0.5),
                                                  (70000, 1024, 0.05)])
def test_pack_kernels_match_wire_ref(columns, rows, density):
    """GPU pack == numpy pack (every byte); GPU unpack of it == the columns."""
    idx = _columns(np.random.default_rng(columns + rows), columns, rows, density)
    ref = W.pack(idx)
    dev = torch.device("cuda", 0)
    with emspec.Engine(rows=rows) as e:
        t = torch.from_numpy(idx).to(dev)
        wire = torch.full((emspec.wire_bound(columns, rows),), 0xAB, dtype=torch.uint8, device=dev)
        nbytes = e.wire_pack(t, wire)
        assert nbytes == ref.size
        got = wire[:nbytes].cpu().numpy()
        assert np.array_equal(got, ref)        # header, offsets, masks, payload AND its zero pad, over a buffer that held 0xAB
        back = torch.full_like(t, 0xCD)
        e.wire_unpack(wire, nbytes, back)
        torch.cuda.synchronize()
        assert np.array_equal(back.cpu().numpy(), idx)
        # a mismatching header is refused
        with emspec.Engine(rows=rows * 2 if rows * 2 <= 4096 else rows // 2) as other:
            with pytest.raises(emspec.EmspecError) as ei:
                other.wire_unpack(wire, nbytes, torch.empty((columns, other.rows), dtype=torch.uint8, device=dev))
            assert ei.value.code == emspec.ERR_INVALID_ARG


@pytest.mark.gpu
def test_rccl_gather_one_rank_loopback():
    """The RCCL path on a single GPU: ncclCommInitRank(world 1), the size all-gather, a self send/recv of the packed
    image and the root-side expand, on real finished columns (batch output vs the oracle's index columns)."""
    n, hop, S, frames = 4096, 256, 3, 300
    pcm = synth.streams(S, n + hop * (frames - 1))
    dev = torch.device("cuda", 0)
    with emspec.Engine() as e:
        e.comm_init(emspec.comm_unique_id(), 0, 1)
        assert (e.comm_rank, e.comm_world) == (0, 1)
        idx = torch.empty((S, frames, e.rows), dtype=torch.uint8, device=dev)
        e.batch_device(torch.from_numpy(pcm).to(dev), n, hop, True, index=idx)
        out = torch.full((1, S, frames, e.rows), 0xEE, dtype=torch.uint8, device=dev)
        sent = e.gather_columns(idx, root=0, out=out, loopback=True)
        torch.cuda.synchronize()
        assert np.array_equal(out[0].cpu().numpy(), idx.cpu().numpy())
        assert 32 < sent < S * frames * e.rows // 2, sent              # the image is much smaller than the raw columns
        # without the loopback flag the root's own columns are a device copy and nothing goes on the wire
        out.fill_(0)
        assert e.gather_columns(idx, root=0, out=out) == 0
        torch.cuda.synchronize()
        assert np.array_equal(out[0].cpu().numpy(), idx.cpu().numpy())
        with pytest.raises(emspec.EmspecError) as ei:
            e.comm_init(emspec.comm_unique_id(), 0, 1)
        assert ei.value.code == emspec.ERR_STATE
        # a gathered buffer smaller than the announced shards is refused before anything is written
        small = torch.full((S * frames * e.rows - 1,), 0x5A, dtype=torch.uint8, device=dev)
        with pytest.raises(emspec.EmspecError) as ei:
            e.gather_columns(idx, root=0, out=small, loopback=True)
        assert ei.value.code == emspec.ERR_INVALID_ARG
        torch.cuda.synchronize()
        assert bool((small == 0x5A).all())
        # EMSPEC_GATHER_PACKED: the root keeps the images packed (directory + images) and expands on demand
        for lb in (True, False):
            cap = 256 * 2 + emspec.wire_bound(S * frames, e.rows)
            packed = torch.full((cap,), 0x77, dtype=torch.uint8, device=dev)
            sent_p = e.gather_columns(idx, root=0, out=packed, loopback=lb, packed=True)
            off, nb, cols = e.gather_packed_layout(0)
            assert (off, cols) == (256, S * frames) and nb == sent == sent_p
            torch.cuda.synchronize()
            d = packed[:32].cpu().numpy().view(np.uint64)
            assert tuple(int(v) for v in d[:3]) == (off, nb, cols)
            back = torch.full((S, frames, e.rows), 0xCD, dtype=torch.uint8, device=dev)
            e.wire_unpack(packed[off:], nb, back)
            torch.cuda.synchronize()
            assert torch.equal(back, idx)
        with pytest.raises(emspec.EmspecError) as ei:   # too small for directory + image
            e.gather_columns(idx, root=0, out=packed[:nb], loopback=True, packed=True)
        assert ei.value.code == emspec.ERR_INVALID_ARG
        _, _, oidx = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("index",))
        palette_close(out[0].cpu().numpy(), oidx)
    with emspec.Engine() as e2:
        with pytest.raises(emspec.EmspecError) as ei:
            e2.gather_columns(torch.zeros((4, 1024), dtype=torch.uint8, device=dev), out=torch.zeros((1, 4, 1024), dtype=torch.uint8, device=dev))
        assert ei.value.code == emspec.ERR_STATE


@pytest.mark.gpu
def test_gather_failure_before_the_exchange_is_collective(monkeypatch):
    """A rank that fails before the size exchange (out of memory, bad arguments) still enters the all-gather with an error
    mark, so every rank returns from the same call (round 3 returned on the failing rank alone and left its peers blocked in
    the collective).  One-rank RCCL form: the injected failure travels through the all-gather, the call returns the rank's
    own error, nothing is written, and the communicator is still usable afterwards."""
    n, hop, S, frames = 4096, 256, 2, 40
    pcm = synth.streams(S, n + hop * (frames - 1))
    dev = torch.device("cuda", 0)
    with emspec.Engine(diag=True) as e:
        e.comm_set_timeout(30.0)
        e.comm_init(emspec.comm_unique_id(), 0, 1)
        idx = torch.empty((S, frames, e.rows), dtype=torch.uint8, device=dev)
        e.batch_device(torch.from_numpy(pcm).to(dev), n, hop, True, index=idx)
        out = torch.full((1, S, frames, e.rows), 0xEE, dtype=torch.uint8, device=dev)
        monkeypatch.setenv("EMSPEC_GATHER_FAIL_RANK", "0")
        with pytest.raises(emspec.EmspecError) as ei:
            e.gather_columns(idx, root=0, out=out, loopback=True)
        assert ei.value.code == emspec.ERR_OOM and "injected" in str(ei.value)
        torch.cuda.synchronize()
        assert bool((out == 0xEE).all())
        # bad arguments on one rank take the same road (they used to return before the collective)
        monkeypatch.delenv("EMSPEC_GATHER_FAIL_RANK")
        with pytest.raises(emspec.EmspecError) as ei:
            e.gather_columns(idx, root=3, out=out, loopback=True)
        assert ei.value.code == emspec.ERR_INVALID_ARG
        # ... and the communicator survived both
        assert e.gather_columns(idx, root=0, out=out, loopback=True) > 32
        torch.cuda.synchronize()
        assert torch.equal(out[0], idx)
        assert e.comm_world == 1


# ---------------------------------------------------------------------------------------------------------------------
# world > 1 on ONE GPU.  RCCL refuses two ranks on one device and a box has one GPU, so until round 4 the library's
# multi-rank logic - uneven shards laid out in rank order on the root, the directory of the packed form, several images
# expanded, the error mark travelling through the size exchange, the timeout - had only ever run with world = 1.  The
# diagnostic build has an in-process stand-in for the three RCCL calls (EMSPEC_COMM_MOCK=1, emspec_comm.cpp: a barrier
# among the ranks' threads, the size pairs through host memory, send/recv as device-to-device copies); everything else
# is the product code.  One thread per rank, one engine each.

def _run_ranks(world, fn):
    import threading
    errs, res = [None] * world, [None] * world

    def body(r):
        try:
            res[r] = fn(r)
        except BaseException as ex:     # noqa: BLE001 - reported to the main thread
            errs[r] = ex
    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank thread is stuck in the collective"
    return res, errs


@pytest.mark.gpu
@pytest.mark.parametrize("world,packed", [(2, False), (4, False), (4, True), (8, True), (8, False)])
def test_gather_many_ranks_on_one_gpu(world, packed, monkeypatch):
    """Uneven shards (1-4 streams per rank), gathered to rank 0: expanding form == the ranks' own columns one after the other in
    rank order; packed form: directory + images, every image expands to its rank's columns."""
    monkeypatch.setenv("EMSPEC_COMM_MOCK", "1")
    n, hop, frames = 4096, 256, 48
    counts = [3, 1, 2, 4, 2, 1, 3, 2][:world]
    L = n + hop * (frames - 1)
    pcm_all = synth.streams(sum(counts), L)
    cid = emspec.comm_unique_id()
    dev = torch.device("cuda", 0)
    rows = 1024
    total_cols = sum(counts) * frames

    def rank(r):
        first, S = sum(counts[:r]), counts[r]
        with emspec.Engine(diag=True) as e:
            e.comm_set_timeout(60.0)
            e.comm_init(cid, r, world)
            assert (e.comm_rank, e.comm_world) == (r, world)
            st = torch.cuda.Stream(device=dev)
            x = torch.from_numpy(pcm_all[first:first + S]).to(dev)
            idx = torch.empty((S, frames, rows), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            e.batch_device(x, n, hop, True, index=idx, stream=st)
            out = None
            if r == 0:
                cap = (256 * (world + 1) + sum(emspec.wire_bound(c * frames, rows) for c in counts)) if packed else total_cols * rows
                out = torch.full((cap,), 0xA5, dtype=torch.uint8, device=dev)
            sent = e.gather_columns(idx, root=0, out=out, stream=st, packed=packed)
            st.synchronize()
            got = None
            if r == 0 and not packed:
                got = out.view(total_cols, rows).cpu().numpy()
            if r == 0 and packed:
                blocks = []
                for q in range(world):
                    off, nb, cols = e.gather_packed_layout(q)
                    assert cols == counts[q] * frames and off % 256 == 0 and nb > 32
                    back = torch.empty((cols, rows), dtype=torch.uint8, device=dev)
                    e.wire_unpack(out[off:], nb, back, stream=st)
                    st.synchronize()
                    blocks.append(back.cpu().numpy())
                got = np.concatenate(blocks)
            return idx.cpu().numpy().reshape(-1, rows), got, sent

    res, errs = _run_ranks(world, rank)
    assert not any(errs), errs
    want = np.concatenate([res[r][0] for r in range(world)])
    assert np.array_equal(res[0][1], want)
    assert all(res[r][2] > 32 for r in range(1, world)) and (res[0][2] == 0 or packed)


@pytest.mark.gpu
def test_gather_failure_on_one_of_four_ranks_ends_the_call_everywhere(monkeypatch):
    """world = 4, rank 2 fails before the size exchange (injected): rank 2 returns its own error, ranks 0, 1, 3 return
    EMSPEC_ERR_COMM from the SAME call - nobody is left in the collective - nothing is written on the root, and the next
    gather of the same communicator works."""
    monkeypatch.setenv("EMSPEC_COMM_MOCK", "1")
    world, n, hop, frames, rows = 4, 4096, 256, 24, 1024
    pcm_all = synth.streams(world, n + hop * (frames - 1))
    cid = emspec.comm_unique_id()
    dev = torch.device("cuda", 0)
    import threading
    gate = threading.Barrier(world)

    def rank(r):
        with emspec.Engine(diag=True) as e:
            e.comm_set_timeout(60.0)
            e.comm_init(cid, r, world)
            st = torch.cuda.Stream(device=dev)
            idx = torch.empty((1, frames, rows), dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            e.batch_device(torch.from_numpy(pcm_all[r:r + 1]).to(dev), n, hop, True, index=idx, stream=st)
            out = torch.full((world * frames * rows,), 0x3C, dtype=torch.uint8, device=dev) if r == 0 else None
            gate.wait()
            if r == 0:
                os.environ["EMSPEC_GATHER_FAIL_RANK"] = "2"
            gate.wait()
            code = 0
            try:
                e.gather_columns(idx, root=0, out=out, stream=st)
            except emspec.EmspecError as ex:
                code = ex.code
            gate.wait()
            if r == 0:
                os.environ.pop("EMSPEC_GATHER_FAIL_RANK")
                st.synchronize()
                assert bool((out == 0x3C).all())
            gate.wait()
            e.gather_columns(idx, root=0, out=out, stream=st)      # the communicator survived
            st.synchronize()
            return code, (out.view(world, frames, rows).cpu().numpy() if r == 0 else idx.cpu().numpy())

    res, errs = _run_ranks(world, rank)
    assert not any(errs), errs
    assert [c for c, _ in res] == [emspec.ERR_COMM, emspec.ERR_COMM, emspec.ERR_OOM, emspec.ERR_COMM]
    assert np.array_equal(res[0][1], np.concatenate([res[r][1] if r else res[0][1][0:1] for r in range(world)]))


@pytest.mark.gpu
def test_gather_missing_rank_times_out(monkeypatch):
    """world = 3 but rank 2 never calls the gather: ranks 0 and 1 return EMSPEC_ERR_COMM when the communicator's timeout
    expires (2 s here) instead of waiting for ever, and the communicator is gone afterwards (EMSPEC_ERR_STATE)."""
    monkeypatch.setenv("EMSPEC_COMM_MOCK", "1")
    world, rows = 3, 1024
    cid = emspec.comm_unique_id()
    dev = torch.device("cuda", 0)
    import time

    def rank(r):
        with emspec.Engine(diag=True) as e:
            e.comm_set_timeout(2.0)
            e.comm_init(cid, r, world)
            if r == 2:
                return None
            idx = torch.zeros((4, rows), dtype=torch.uint8, device=dev)
            out = torch.zeros((world * 4 * rows,), dtype=torch.uint8, device=dev) if r == 0 else None
            t0 = time.time()
            with pytest.raises(emspec.EmspecError) as ei:
                e.gather_columns(idx, root=0, out=out)
            assert ei.value.code == emspec.ERR_COMM and "timeout" in str(ei.value)
            with pytest.raises(emspec.EmspecError) as ei2:
                e.gather_columns(idx, root=0, out=out)
            assert ei2.value.code == emspec.ERR_STATE
            return time.time() - t0

    res, errs = _run_ranks(world, rank)
    assert not any(errs), errs
    assert 1.5 < res[0] < 30 and 1.5 < res[1] < 30
