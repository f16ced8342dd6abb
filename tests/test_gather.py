"""Multi-GPU gather of finished columns: the wire image (pack / unpack kernels vs the numpy restatement
oracle/wire_ref.py, byte for byte) and the RCCL path itself on one rank (communicator creation, size all-gather,
self send/recv, root-side expand).  More ranks need more GPUs than a box has; the N > 1 control flow is covered with
gloo on the CPU in tests/test_distributed.py."""
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
        pay_end = W.fixed_bytes(columns, rows) + int((idx != 0).sum())
        assert np.array_equal(got[:pay_end], ref[:pay_end])        # header, offsets, masks, payload (the pad bytes are don't-care)
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
