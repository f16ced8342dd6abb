"""GPU parity of the LIVE multi-stream entry points (include/emspec.h: emspec_columns, emspec_columns_flush,
emspec_push_samples_multi, emspec_reset_stream): S streams advance by one frame / one block of samples per call, ONE launch.
BASELINE configs[2] ("64 concurrent 48 kHz streams") in the form the renderer calls it (north_star: per-frame
computeSpectrogramColumn).  Everything is checked against the ORACLE's batch columns of the same streams (FAST mode:
8.7e-4 dB, the float32 sums are taken in arrival order; EXACT mode: equal bytes), never against the engine's own batch
call.  The reference implementation is unavailable (private source): parity with it stays UNPINNED.
"""
import numpy as np
import pytest

import emspec
import oracle as O
from emspec import synth

pytestmark = pytest.mark.gpu

TOL_DB = 8.7e-4


def _oracle(n, hop, reassign, pcm, exact, want=("db", "rgba")):
    cfg = O.make_cfg(n, hop, reassign)
    if exact:
        db, rgba, _, _ = O.batch_exact(cfg, pcm, want=want)
    else:
        db, rgba, _ = O.batch_f32(cfg, pcm, want=want)
    return db, rgba


def _same_db(got, want, exact):
    if exact:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        assert np.max(np.abs(got - want)) < TOL_DB, float(np.max(np.abs(got - want)))


def _same_rgba(got, want, exact):
    if exact:
        assert np.array_equal(got, want)
    else:
        assert np.mean(got != want) < 1e-3


@pytest.mark.parametrize("exact", [False, True], ids=["fast", "exact"])
def test_64_streams_200_hops_equal_the_oracle_batch(exact):
    """configs[2] live: 64 streams x 200 frames of N = 4096 / hop 256, one emspec_columns call per hop, then the flush."""
    S, n, hop, frames = 64, 4096, 256, 200
    pcm = synth.streams(S, n + hop * (frames - 1))
    odb, orgba = _oracle(n, hop, True, pcm, exact)
    D = emspec.latency_columns(n, hop, True)
    got_db = np.empty((S, frames, 1024), np.float32)
    got_rgba = np.empty((S, frames, 1024, 4), np.uint8)
    with emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST) as e:
        for j in range(frames):
            db, rgba, cols = e.columns(pcm[:, j * hop:j * hop + n], hop, True, want_rgba=True)
            assert e.live_streams == S
            assert np.all(cols == (j - D if j >= D else -1))
            if j >= D:
                got_db[:, j - D], got_rgba[:, j - D] = db, rgba
            else:   # the empty column: every cell at the floor of the dB map, palette index 0
                assert np.all(db == db[0, 0]) and np.all(rgba == rgba[0, 0])
        for i in range(D):
            db, rgba, cols = e.columns_flush(want_rgba=True)
            assert np.all(cols == frames - D + i)
            got_db[:, frames - D + i], got_rgba[:, frames - D + i] = db, rgba
        with pytest.raises(emspec.EmspecError) as ei:
            e.columns_flush()
        assert ei.value.code == emspec.ERR_STATE
    _same_db(got_db, odb, exact)
    _same_rgba(got_rgba, orgba, exact)


@pytest.mark.parametrize("exact", [False, True], ids=["fast", "exact"])
@pytest.mark.parametrize("n,hop,reassign,S,frames", [(1024, 256, False, 5, 40), (16384, 512, True, 3, 40), (2048, 300, True, 7, 50),
                                                     (8192, 512, True, 2, 30), (512, 64, True, 9, 60)])
def test_columns_other_shapes(n, hop, reassign, S, frames, exact):
    pcm = synth.streams(S, n + hop * (frames - 1))
    odb, _ = _oracle(n, hop, reassign, pcm, exact, want=("db",))
    D = emspec.latency_columns(n, hop, reassign)
    got = np.empty((S, frames, 1024), np.float32)
    with emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST) as e:
        for j in range(frames):
            db, _, cols = e.columns(pcm[:, j * hop:j * hop + n], hop, reassign)
            if j >= D:
                got[:, j - D] = db
        for i in range(D):
            db, _, cols = e.columns_flush()
            got[:, frames - D + i] = db
    _same_db(got, odb, exact)


@pytest.mark.parametrize("exact", [False, True], ids=["fast", "exact"])
@pytest.mark.parametrize("n,hop,reassign,S,block", [(4096, 256, True, 64, 256), (4096, 256, True, 6, 128), (4096, 256, True, 4, 20000),
                                                    (1024, 256, False, 5, 777), (16384, 512, True, 3, 5000), (2048, 128, True, 8, 2048)])
def test_push_samples_multi_matches_oracle(n, hop, reassign, S, block, exact):
    """emspec_push_samples_multi with blocks of any length (one hop per call = the live case; a worklet's 128 samples; blocks
    longer than the staging block: several launches per call), then the flush, against the oracle's batch columns."""
    frames = 70
    L = n + hop * (frames - 1)
    pcm = synth.streams(S, L)
    odb, orgba = _oracle(n, hop, reassign, pcm, exact)
    D = emspec.latency_columns(n, hop, reassign)
    got_db = np.empty((S, frames, 1024), np.float32)
    got_rgba = np.empty((S, frames, 1024, 4), np.uint8)
    nxt = 0
    with emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST) as e:
        for a in range(0, L, block):
            cnt = min(block, L - a)
            k = e.push_columns_multi(cnt, n, hop, reassign)
            db, rgba, counts, firsts = e.push_samples_multi(pcm, n, hop, reassign, want_rgba=True, count=cnt, offset=a)
            assert np.all(counts == k) and db.shape[1] == k
            if k:
                assert np.all(firsts == nxt)
                got_db[:, nxt:nxt + k], got_rgba[:, nxt:nxt + k] = db, rgba
                nxt += k
            else:
                assert np.all(firsts == -1)
        assert nxt == max(frames - D, 0)
        for i in range(min(D, frames)):
            db, rgba, cols = e.columns_flush(want_rgba=True)
            assert np.all(cols == nxt)
            got_db[:, nxt], got_rgba[:, nxt] = db, rgba
            nxt += 1
    _same_db(got_db, odb, exact)
    _same_rgba(got_rgba, orgba, exact)


@pytest.mark.parametrize("exact", [False, True], ids=["fast", "exact"])
@pytest.mark.parametrize("form", ["frames", "samples"])
def test_reset_of_one_stream_leaves_the_others_alone(form, exact):
    """emspec_reset_stream(e, s) mid-session: stream s restarts from column 0 on new audio while the other streams continue."""
    S, n, hop, frames, cut, who = 6, 4096, 256, 60, 23, 2
    pcm = synth.streams(S, n + hop * (frames - 1))
    fresh = synth.streams(S + 1, n + hop * (frames - cut - 1))[S]          # the restarted stream's new audio
    odb, _ = _oracle(n, hop, True, pcm, exact, want=("db",))
    ndb, _ = _oracle(n, hop, True, fresh[None], exact, want=("db",))
    D = emspec.latency_columns(n, hop, True)
    cols_seen = [dict() for _ in range(S)]
    new_seen = {}
    with emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST) as e:
        if form == "frames":
            for j in range(frames):
                if j == cut:
                    e.reset_stream(who)
                fr = pcm[:, j * hop:j * hop + n].copy()
                if j >= cut:
                    fr[who] = fresh[(j - cut) * hop:(j - cut) * hop + n]
                db, _, cols = e.columns(fr, hop, True)
                for s in range(S):
                    if cols[s] >= 0:
                        (new_seen if (s == who and j >= cut) else cols_seen[s])[int(cols[s])] = db[s].copy()
                    if s == who and j >= cut:
                        assert cols[s] == (j - cut - D if j - cut >= D else -1)
        else:
            # stream `who` is cut at a hop boundary: before it the first cut*hop + n - hop samples... feed hop-sized blocks
            L = pcm.shape[1]
            feed = pcm.copy()
            pos, blk = 0, hop
            cut_at = n + hop * (cut - 1)                                         # samples fed when frame cut-1 is complete
            while pos < L:
                cnt = min(blk, L - pos)
                if pos == cut_at:
                    e.reset_stream(who)
                    feed[who, pos:pos + fresh.size] = fresh[:L - pos]
                db, _, counts, firsts = e.push_samples_multi(feed, n, hop, True, count=cnt, offset=pos)
                for s in range(S):
                    for i in range(int(counts[s])):
                        (new_seen if (s == who and pos >= cut_at) else cols_seen[s])[int(firsts[s]) + i] = db[s, i].copy()
                pos += cnt
        for s in range(S):
            if s == who:
                continue
            want = frames - D
            assert sorted(cols_seen[s]) == list(range(want))
            _same_db(np.stack([cols_seen[s][c] for c in range(want)]), odb[s, :want], exact)
    # the restarted stream: its old columns up to the reset, then the new audio's columns from 0
    before = sorted(cols_seen[who])
    assert before == list(range(len(before))) and len(before) >= cut - D
    _same_db(np.stack([cols_seen[who][c] for c in before]), odb[who, :len(before)], exact)
    got_new = sorted(new_seen)
    assert got_new == list(range(len(got_new))) and len(got_new) >= 10
    _same_db(np.stack([new_seen[c] for c in got_new]), ndb[0, :len(got_new)], exact)


def test_pinned_buffers_are_used_in_place_and_give_the_same_columns():
    """Page-locked frames / outputs (emspec_host_alloc) are read and written by the kernel in place; same bytes as staged."""
    S, n, hop, frames = 8, 4096, 256, 30
    pcm = synth.streams(S, n + hop * (frames - 1))
    pin_in = emspec.PinnedArray((S, n), np.float32)
    pin_db = emspec.PinnedArray((S, 1024), np.float32)
    pin_rgba = emspec.PinnedArray((S, 1024, 4), np.uint8)
    with emspec.Engine(mode=emspec.MODE_EXACT) as a, emspec.Engine(mode=emspec.MODE_EXACT) as b:
        for j in range(frames):
            fr = pcm[:, j * hop:j * hop + n]
            pin_in.array[:] = fr
            _, _, c1 = a.columns(pin_in.array, hop, True, want_rgba=True, db=pin_db.array, rgba=pin_rgba.array)
            db2, rgba2, c2 = b.columns(fr, hop, True, want_rgba=True)
            assert np.array_equal(c1, c2)
            assert np.array_equal(pin_db.array.view(np.uint32), db2.view(np.uint32)) and np.array_equal(pin_rgba.array, rgba2)
    for p in (pin_in, pin_db, pin_rgba):
        p.close()


@pytest.mark.parametrize("exact", [False, True], ids=["fast", "exact"])
def test_live_display_postprocess_matches_sequential_restatement(exact):
    """AGC + temporal smoothing (emspec_set_display) inside the live calls, per stream and in time order, against the numpy
    restatement applied to the oracle's raw columns."""
    S, n, hop, frames = 5, 4096, 256, 60
    pcm = synth.streams(S, n + hop * (frames - 1))
    cfg = O.make_cfg(n, hop, True)
    odb, _ = _oracle(n, hop, True, pcm, exact, want=("db",))
    want = O.postprocess(odb, 0.6, 0.8, cfg)[0]
    D = emspec.latency_columns(n, hop, True)
    got = np.empty_like(odb)
    with emspec.Engine(mode=emspec.MODE_EXACT if exact else emspec.MODE_FAST) as e:
        e.set_display(0.6, 0.8)
        nxt = 0
        for a in range(0, pcm.shape[1], 3 * hop):
            cnt = min(3 * hop, pcm.shape[1] - a)
            db, _, counts, firsts = e.push_samples_multi(pcm, n, hop, True, count=cnt, offset=a)
            k = int(counts[0])
            assert np.all(counts == k)
            got[:, nxt:nxt + k] = db[:, :k]
            nxt += k
        for _ in range(D):
            db, _, cols = e.columns_flush()
            got[:, nxt] = db
            nxt += 1
        assert nxt == frames
    assert np.max(np.abs(got - want)) < 2e-3, float(np.max(np.abs(got - want)))


def test_live_display_postprocess_into_an_oversized_pinned_block():
    """The post-process keeps the raw columns in a device block of its own stride (what the call completes, not the caller's
    max_columns): page-locked outputs of 40 columns per stream, three completed per call, written in place."""
    S, n, hop, frames = 3, 2048, 256, 40
    pcm = synth.streams(S, n + hop * (frames - 1))
    cfg = O.make_cfg(n, hop, True)
    odb, _ = _oracle(n, hop, True, pcm, False, want=("db",))
    want = O.postprocess(odb, 0.5, 0.7, cfg)[0]
    D = emspec.latency_columns(n, hop, True)
    pin = emspec.PinnedArray((S, 40, 1024), np.float32)
    pin_rgba = emspec.PinnedArray((S, 40, 1024, 4), np.uint8)
    got = np.empty_like(odb)
    with emspec.Engine() as e:
        e.set_display(0.5, 0.7)
        nxt = 0
        for a in range(0, pcm.shape[1], 3 * hop):
            cnt = min(3 * hop, pcm.shape[1] - a)
            pin.array[...] = -1.0
            db, rgba, counts, firsts = e.push_samples_multi(pcm, n, hop, True, want_rgba=True, db=pin.array, rgba=pin_rgba.array,
                                                            count=cnt, offset=a)
            k = int(counts[0])
            assert np.all(counts == k) and k <= 3
            assert np.all(pin.array[:, k:] == -1.0)          # nothing beyond the completed columns is touched
            got[:, nxt:nxt + k] = pin.array[:, :k]
            nxt += k
        assert nxt == frames - D
    assert np.max(np.abs(got[:, :nxt] - want[:, :nxt])) < 2e-3


def test_live_session_guards():
    S, n, hop = 3, 1024, 256
    pcm = synth.streams(S, n + hop * 20)
    with emspec.Engine() as e:
        with pytest.raises(emspec.EmspecError) as ei:
            e.reset_stream(0)                                   # no session yet
        assert ei.value.code == emspec.ERR_STATE
        e.columns(pcm[:, :n], hop, True)
        for bad in (lambda: e.columns(pcm[:2, :n], hop, True),            # other stream count
                    lambda: e.columns(pcm[:, :n], 128, True),             # other hop
                    lambda: e.push_samples_multi(pcm[:, :512], n, hop, True)):   # other feeding form
            with pytest.raises(emspec.EmspecError) as ei:
                bad()
            assert ei.value.code == emspec.ERR_STATE
        with pytest.raises(emspec.EmspecError) as ei:
            e.reset_stream(S)
        assert ei.value.code == emspec.ERR_INVALID_ARG
        with pytest.raises(emspec.EmspecError) as ei:
            e.set_row_edges_hz(None)                            # a column is pending
        assert ei.value.code == emspec.ERR_STATE
        e.reset()
        assert e.live_streams == 0
        # the single-stream calls keep their own state beside a live session
        e.push_samples_multi(pcm[:, :n + hop], n, hop, True)
        db1, c1 = e.column(pcm[0, :n], hop, True)
        assert c1 == -1 and e.live_streams == S
        # an output too small for what the block completes is rejected before any state changes
        small = np.empty((S, 1, 1024), np.float32)
        with pytest.raises(emspec.EmspecError) as ei:
            e.push_samples_multi(pcm[:, n + hop:n + 12 * hop], n, hop, True, db=small)
        assert ei.value.code == emspec.ERR_INVALID_ARG
        db, _, counts, firsts = e.push_samples_multi(pcm[:, n + hop:n + 12 * hop], n, hop, True)
        assert np.all(counts == 11) and np.all(firsts == 0)
        # a flush puts every stream with pending columns at its end: feeding again is refused until those streams restart
        _, _, cols = e.columns_flush()
        assert np.all(cols == 11)
        with pytest.raises(emspec.EmspecError) as ei:
            e.push_samples_multi(pcm[:, :hop], n, hop, True)
        assert ei.value.code == emspec.ERR_STATE and "flushed" in str(ei.value)
        for s in range(S):
            e.reset_stream(s)
        db, _, counts, firsts = e.push_samples_multi(pcm[:, :n + 4 * hop], n, hop, True)
        assert np.all(counts == 3) and np.all(firsts == 0)
        # the same rule for the single-stream calls
        e.flush()
        with pytest.raises(emspec.EmspecError) as ei:
            e.column(pcm[0, :n], hop, True)
        assert ei.value.code == emspec.ERR_STATE


def test_live_session_of_4000_hops_equals_the_batch_bits():
    """configs[2]'s 64 streams through the live entry for 2^20 samples each (4,081 hops: every sample ring wraps 63 times, every
    column ring 85 times), EXACT mode, page-locked blocks: every column's dB bits equal the BATCH call's on the same audio - the
    size-independent property (live == batch) at a length the CPU oracle does not reach in seconds."""
    import torch
    S, n, hop, L = 64, 4096, 256, 1 << 20
    base = synth.streams(8, L)
    pcm = np.ascontiguousarray(np.stack([np.roll(base[s % 8], 1013 * s) * np.float32(1.0 - 0.01 * (s % 11)) for s in range(S)]))
    Cn = emspec.num_columns(L, n, hop)
    D = emspec.latency_columns(n, hop, True)
    dev = torch.device("cuda", 0)
    with emspec.Engine(mode=emspec.MODE_EXACT) as ref:
        db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
        ref.batch_device(torch.from_numpy(pcm).to(dev), n, hop, True, db=db)
        torch.cuda.synchronize()
        want = db.cpu().numpy().view(np.uint32)
        del db
    blk = emspec.PinnedArray((S, hop), np.float32)
    out = emspec.PinnedArray((S, 1, 1024), np.float32)
    bad = seen = 0
    try:
        with emspec.Engine(mode=emspec.MODE_EXACT) as e:
            e.push_samples_multi(pcm[:, :n - hop].copy(), n, hop, True, want_db=False)
            for j in range(Cn):
                blk.array[:] = pcm[:, n - hop + j * hop:n + j * hop]
                _, _, counts, firsts = e.push_samples_multi(blk.array, n, hop, True, db=out.array)
                if j >= D:
                    assert np.all(counts == 1) and np.all(firsts == j - D)
                    bad += int(not np.array_equal(out.array.view(np.uint32)[:, 0], want[:, j - D]))
                    seen += 1
            for i in range(D):
                dbf, _, cols = e.columns_flush()
                assert np.all(cols == Cn - D + i)
                bad += int(not np.array_equal(dbf.view(np.uint32), want[:, Cn - D + i]))
                seen += 1
    finally:
        blk.close(); out.close()
    assert seen == Cn and bad == 0, (seen, bad)

