"""-m gpu: the host-buffer entry points (emspec_batch / emspec_batch_packed) - the three-stage pipeline over chunks of
streams (H2D | kernels | D2H on three HIP streams, three staging sets; from pageable memory with a second host thread for the
copies out) must return exactly what the device-resident call returns, in both arithmetic modes and with the display post-process, and the packed form's images must expand (on the
host, emspec_wire_unpack_host) to the same palette indices."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "em-spec_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import emspec  # noqa: E402
import oracle as O  # noqa: E402
import wire_ref as W  # noqa: E402
from emspec import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _pinned(a):
    p = emspec.PinnedArray(a.shape, a.dtype)
    p.array[...] = a
    return p


def _batch_pinned(e, pcm, n, hop, want=("db", "index")):
    """emspec_batch with EVERY host buffer page-locked (the condition for the pipelined path); returns copies."""
    import ctypes as C
    S, L = pcm.shape
    Cn = emspec.num_columns(L, n, hop)
    pin = _pinned(pcm)
    pdb = emspec.PinnedArray((S, Cn, e.rows), np.float32) if "db" in want else None
    pix = emspec.PinnedArray((S, Cn, e.rows), np.uint8) if "index" in want else None
    try:
        if pdb is not None:
            pdb.array[...] = -1.0
        if pix is not None:
            pix.array[...] = 255
        out = emspec.Out(pdb.array.ctypes.data if pdb else None, None, pix.array.ctypes.data if pix else None)
        e._chk(e._lib.emspec_batch(e._h, C.c_void_p(pin.array.ctypes.data), S, L, n, hop, 1, C.byref(out)))
        return {"db": pdb.array.copy() if pdb else None, "index": pix.array.copy() if pix else None}
    finally:
        pin.close()
        if pdb:
            pdb.close()
        if pix:
            pix.close()


@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("S", [2, 23])
def test_pageable_batch_equals_pinned_batch(mode, S):
    """Ordinary numpy arrays (pageable): the pipeline's copies out run on a second host thread (round 6).  23 streams = 12 chunks
    through three staging sets, 2 streams = fewer chunks than sets; dB, RGBA and palette index out at once; twice.  EXACT: bytes equal to the page-locked call's; FAST: equal but for its +-1 cells."""
    n, hop = 4096, 256
    L = n + hop * 149 + 5
    pcm = synth.streams(S, L)
    with emspec.Engine(mode=emspec.MODE_EXACT if mode == "exact" else emspec.MODE_FAST) as e:
        ref = _batch_pinned(e, pcm, n, hop)
        lut = emspec.make_colormap(0.7)
        e.set_colormap(lut)
        for _ in range(2):
            got = e.batch(pcm, n, hop, True, want=("db", "rgba", "index"))
            if mode == "exact":
                assert np.array_equal(got["index"], ref["index"])
                assert np.array_equal(got["db"].view(np.uint32), ref["db"].view(np.uint32))
            else:
                d = np.abs(got["index"].astype(int) - ref["index"].astype(int))
                assert d.max() <= 1 and np.mean(d != 0) < 1e-4
                assert np.max(np.abs(got["db"] - ref["db"])) < 1e-3
            assert np.array_equal(got["rgba"], np.asarray(lut).reshape(256, 4)[got["index"]])


@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("S,pinned", [(1, True), (1, False), (3, False)])
def test_few_long_streams_are_pipelined_by_runs_of_columns(mode, S, pinned):
    """Fewer streams than the pipeline has stages (BASELINE configs[1] is ONE stream): a stream's columns are cut into runs of
    >= 16,384, each computed as a batch of its own from the frames that reach it (D halo frames either side, their columns left
    behind).  33,100 columns: two runs per stream; the seam must not show - EXACT: bytes equal to the device-resident call's on
    the whole stream (dB bits and palette index); FAST: its +-1 cells."""
    import torch
    n, hop = 4096, 256
    L = n + hop * 33099 + 100
    one = synth.streams(1, L)[0]
    pcm = np.stack([one * (1.0 - 0.2 * s) for s in range(S)]).astype(np.float32)
    Cn = emspec.num_columns(L, n, hop)
    assert Cn == 33100
    with emspec.Engine(mode=emspec.MODE_EXACT if mode == "exact" else emspec.MODE_FAST) as e:
        if pinned:
            got = _batch_pinned(e, pcm, n, hop)
        else:
            got = e.batch(pcm, n, hop, True, want=("db", "index"))
        x = torch.from_numpy(pcm).cuda()
        db = torch.empty((S, Cn, e.rows), dtype=torch.float32, device="cuda")
        ix = torch.empty((S, Cn, e.rows), dtype=torch.uint8, device="cuda")
        e.batch_device(x, n, hop, True, db=db, index=ix)
        torch.cuda.synchronize()
        rdb, rix = db.cpu().numpy(), ix.cpu().numpy()
    if mode == "exact":
        assert np.array_equal(got["index"], rix)
        assert np.array_equal(got["db"].view(np.uint32), rdb.view(np.uint32))
    else:
        d = np.abs(got["index"].astype(np.int16) - rix.astype(np.int16))
        assert d.max() <= 1 and np.mean(d != 0) < 1e-4
        assert np.max(np.abs(got["db"] - rdb)) < 1e-3
        seam = Cn // 2
        assert np.mean(d[:, seam - 20:seam + 20] != 0) < 1e-3      # no more of them around the seam than anywhere


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_pipelined_batch_equals_device_resident_batch(mode):
    """23 streams from pinned buffers: 12 chunks of two streams through three staging sets (every set is reused three
    times, the last chunk is short).  EXACT mode: bytes equal to the bit model; FAST mode: palette index equal to the
    device-resident call's except for its documented +-1 cells, dB within a few ulp of it."""
    import torch
    n, hop, S = 4096, 256, 23
    L = n + hop * 299 + 17
    pcm = synth.streams(S, L)
    Cn = emspec.num_columns(L, n, hop)
    with emspec.Engine(mode=emspec.MODE_EXACT if mode == "exact" else emspec.MODE_FAST) as e:
        got = _batch_pinned(e, pcm, n, hop)
        gdb, gix = got["db"], got["index"]
        x = torch.from_numpy(pcm).cuda()
        db = torch.empty((S, Cn, e.rows), dtype=torch.float32, device="cuda")
        ix = torch.empty((S, Cn, e.rows), dtype=torch.uint8, device="cuda")
        e.batch_device(x, n, hop, True, db=db, index=ix)
        torch.cuda.synchronize()
        rdb, rix = db.cpu().numpy(), ix.cpu().numpy()
    if mode == "exact":
        assert np.array_equal(gix, rix) and np.array_equal(gdb.view(np.uint32), rdb.view(np.uint32))
        odb, _, oix, _ = O.batch_exact(O.make_cfg(n, hop, True), pcm[20:], want=("db", "index"))
        assert np.array_equal(gix[20:], oix) and np.array_equal(gdb[20:].view(np.uint32), odb.view(np.uint32))
    else:
        d = np.abs(gix.astype(int) - rix.astype(int))
        assert d.max() <= 1 and np.mean(d != 0) < 1e-4
        assert np.max(np.abs(gdb - rdb)) < 1e-3


def test_packed_batch_images_expand_to_the_plain_columns():
    """emspec_batch_packed: one lossless wire image per stream, tightly packed in stream order; every image expands on the
    host (emspec_wire_unpack_host) to the palette indices emspec_batch returns; the images are a fraction of the raw bytes;
    a buffer that cannot hold them is refused with a message, nothing is written past it."""
    n, hop, S = 4096, 256, 11
    L = n + hop * 399
    pcm = synth.streams(S, L)
    Cn = emspec.num_columns(L, n, hop)
    with emspec.Engine(mode=emspec.MODE_EXACT) as e:
        ref = e.batch(pcm, n, hop, True, want=("index",))["index"]
        pin = _pinned(pcm)
        pw = emspec.PinnedArray((S * emspec.wire_bound(Cn, e.rows),), np.uint8)
        try:
            wire, offs = e.batch_packed(pin.array, n, hop, True, wire=pw.array)
            assert offs[0] == 0 and np.all(np.diff(offs) > 0) and np.all(offs % 16 == 0)
            assert offs[-1] < 0.5 * ref.size                      # packed: well under the raw bytes on this input
            for s in range(S):
                got = emspec.wire_unpack_host(wire[offs[s]:offs[s + 1]], Cn, e.rows)
                assert np.array_equal(got, ref[s]), s
                # every byte of the image is specified (the payload's pad is zero): equal to the numpy restatement
                assert np.array_equal(wire[offs[s]:offs[s + 1]], W.pack(ref[s])), s
            first = wire[:offs[-1]].copy()
            pw.array[...] = 0x5A                                  # whatever the buffer (and the staging) held before
            wire, offs2 = e.batch_packed(pin.array, n, hop, True, wire=pw.array)
            assert np.array_equal(offs2, offs) and np.array_equal(wire[:offs[-1]], first)
            # pageable buffers work too (slower), and so does a single stream
            w2, o2 = e.batch_packed(pcm[:1], n, hop, True)
            assert np.array_equal(emspec.wire_unpack_host(w2[:o2[1]], Cn, e.rows), ref[0])
            small = np.full(int(offs[3]) + 64, 0xAB, np.uint8)
            with pytest.raises(emspec.EmspecError) as ei:
                e.batch_packed(pin.array, n, hop, True, wire=small[:int(offs[3])])
            assert "too small" in str(ei.value) and np.all(small[int(offs[3]):] == 0xAB)
        finally:
            pin.close(); pw.close()
    # the fast mode's images: equal to its own plain columns up to the mode's +-1 cells (two runs of float accumulates)
    with emspec.Engine() as f:
        ref = f.batch(pcm[:3], n, hop, True, want=("index",))["index"]
        wire, offs = f.batch_packed(pcm[:3], n, hop, True)
        for s in range(3):
            got = emspec.wire_unpack_host(wire[offs[s]:offs[s + 1]], Cn, f.rows)
            d = np.abs(got.astype(int) - ref[s].astype(int))
            assert d.max() <= 1 and np.mean(d != 0) < 1e-4


def test_pipelined_batch_with_display_postprocess_and_other_sizes():
    """The per-engine workspaces of the display post-process and of the N = 16384 EXACT records path are used by one chunk at
    a time (every kernel runs on the engine's compute stream): pinned batches through the pipeline equal the one-chunk result."""
    n, hop, S = 1024, 256, 9
    L = n + hop * 499
    pcm = synth.streams(S, L)
    with emspec.Engine() as e:
        e.set_display(smoothing=0.5, agc_strength=0.7)
        one = np.concatenate([e.batch(pcm[s:s + 1], n, hop, True, want=("db",))["db"] for s in range(S)])
        got = _batch_pinned(e, pcm, n, hop, want=("db",))["db"]
    assert np.max(np.abs(got - one)) < 2e-3
    n, hop, S = 16384, 512, 5
    L = n + hop * 59
    pcm = synth.streams(S, L)
    with emspec.Engine(mode=emspec.MODE_EXACT) as x:
        got = _batch_pinned(x, pcm, n, hop)
    odb, _, oix, _ = O.batch_exact(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
    assert np.array_equal(got["index"], oix) and np.array_equal(got["db"].view(np.uint32), odb.view(np.uint32))


@pytest.mark.parametrize("rows,n,hop", [(68, 1024, 256), (100, 2048, 128), (512, 4096, 512)])
def test_packed_batch_other_row_counts_and_sizes(rows, n, hop):
    """The wire image's generic form (rows a multiple of 4, not of 32) and other FFT sizes through emspec_batch_packed: every
    stream's image expands on the host to the columns emspec_batch returns (EXACT mode: the same bytes)."""
    S, frames = 7, 120
    L = n + hop * (frames - 1) + 5
    pcm = synth.streams(S, L)
    with emspec.Engine(mode=emspec.MODE_EXACT, rows=rows) as e:
        ref = e.batch(pcm, n, hop, True, want=("index",))["index"]
        wire, offs = e.batch_packed(pcm, n, hop, True)
        for s in range(S):
            assert np.array_equal(emspec.wire_unpack_host(wire[offs[s]:offs[s + 1]], frames, rows), ref[s]), (rows, s)


@pytest.mark.parametrize("columns", [401, 1021, 7])
def test_packed_batch_images_start_on_16_byte_boundaries_for_any_column_count(columns):
    """an image is 32 + 4 C (1 + R/32) + pad16(payload) bytes - a multiple of 4 only - so with C = 401 or 1021 (rows
    1024) the images used to start 4 mod 16.  Every image now STARTS on a 16-byte boundary (the header's promise); the <= 12
    bytes of slack inside a slot are not part of the image, and both unpackers take the slot as it is."""
    n, hop, S = 1024, 256, 5
    L = n + hop * (columns - 1)
    pcm = synth.streams(S, L)
    with emspec.Engine(mode=emspec.MODE_EXACT) as e:
        assert emspec.num_columns(L, n, hop) == columns
        ref = e.batch(pcm, n, hop, True, want=("index",))["index"]
        for pinned in (True, False):
            pin = _pinned(pcm) if pinned else None
            pw = emspec.PinnedArray((S * emspec.wire_bound(columns, e.rows),), np.uint8) if pinned else None
            try:
                if pinned:
                    pw.array[...] = 0xEE
                wire, offs = e.batch_packed(pin.array if pinned else pcm, n, hop, True, wire=pw.array if pinned else None)
                assert offs[0] == 0 and np.all(offs % 16 == 0) and np.all(np.diff(offs) > 0)
                assert offs[-1] <= S * emspec.wire_bound(columns, e.rows)
                for s in range(S):
                    image = W.pack(ref[s])
                    slot = wire[offs[s]:offs[s + 1]]
                    assert 0 <= slot.size - image.size <= 12
                    assert np.array_equal(slot[:image.size], image), s
                    assert np.array_equal(emspec.wire_unpack_host(slot, columns, e.rows), ref[s])
            finally:
                if pinned:
                    pin.close(); pw.close()
