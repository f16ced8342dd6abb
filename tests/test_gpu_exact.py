"""GPU parity of the EXACT mode (EMSPEC_MODE_EXACT) through the C ABI.

north_star asks for results that match the reference JS path "exactly on reassigned integer (time,freq) bin
indices"; JavaScript arithmetic is binary64.  The reference itself is unavailable (private source, parity UNPINNED),
so the bar here is:
  (a) bit-identical to the binary64 bit model oracle/emspec_exact.c: per-bin power (float64), column, row, fixed-point
      energy; finished dB (float32 bits), palette index and RGBA bytes - array_equal, no tolerance;
  (b) (column,row) equal on EVERY bin to two independent float64 implementations of the three-window method
      (oracle/emspec_oracle.c:eo_frames_f64 and the committed numpy goldens) - asserted mismatch rate <= 1e-6;
  (c) identical bytes run to run and identical between the batch call and the streaming calls (integer accumulation is
      order-independent);
  (d) finished columns against the float64 method: >= 99.99 % of the cells above -60 dB within 8.7e-4 dB.
"""
import glob
import os

import numpy as np
import pytest

import emspec
import oracle as O
from emspec import synth

pytestmark = pytest.mark.gpu

@pytest.mark.parametrize("n,hop,frames", [(16384, 512, 10), (1024, 256, 48), (8192, 512, 12), (4096, 256, 40)])
def test_exact_without_power_floor_runs_the_generic_core(n, hop, frames):
    """power_floor = 0 takes a plan off the branch-free per-bin core (its short reciprocal needs 64 P >= 64e-200): every
    EXACT kernel then runs its generic core on the log axis - the template instances the usual plans no longer reach since
    round 4 (exact_frames_kernel<.., false>, exact_frames16384_kernel<false>, exact_fused4096_kernel<.., false, ..>).
    Dump and finished columns against the bit model, as for the fast plans."""
    pcm = _pcm(n, hop, frames, S=2)
    pcm[1, : n // 2] = 0.0                                   # exact zeros: bins with P = 0 pass a zero floor
    with emspec.Engine(mode=emspec.MODE_EXACT, power_floor=0.0) as e:
        cfg = O.make_cfg(n, hop, True, power_floor=0.0)
        pw, col, row, q = e.parity_dump_exact(pcm, n, hop, True, 0, frames)
        for s_ in range(2):
            opw, ocol, orow, oq = O.frames_exact(cfg, pcm[s_], 0, frames)
            assert np.array_equal(row[s_], orow) and np.array_equal(col[s_], ocol) and np.array_equal(q[s_], oq)
            assert np.array_equal(pw[s_].view(np.uint64), opw.view(np.uint64))
        out = e.batch(pcm, n, hop, True, want=("db", "index"))
        odb, _, oidx, _ = O.batch_exact(cfg, pcm, want=("db", "index"))
        assert np.array_equal(out["index"], oidx)
        assert np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32))


GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.fixture(scope="module")
def xengine():
    import torch   # torch's HIP runtime first (tests/conftest.py:_torch_first)
    if torch.cuda.is_available():
        torch.cuda.init()
    e = emspec.Engine(mode=emspec.MODE_EXACT)
    yield e
    e.close()


def _pcm(n, hop, frames, S=2, extra=37):
    return synth.streams(S, n + hop * (frames - 1) + extra)


DUMP_CASES = [(1024, 256), (4096, 256), (16384, 512), (2048, 128), (8192, 1024), (256, 64), (512, 512), (8192, 512),
              (4096, 300), (16384, 2048)]


@pytest.mark.parametrize("n,hop", DUMP_CASES)
@pytest.mark.parametrize("reassign", [True, False])
def test_exact_dump_equals_bit_model(xengine, n, hop, reassign):
    frames = 10 if n < 16384 else 6
    pcm = _pcm(n, hop, frames)
    pw, col, row, q = xengine.parity_dump_exact(pcm, n, hop, reassign, 0, frames)
    cfg = O.make_cfg(n, hop, reassign)
    for s in range(pcm.shape[0]):
        opw, ocol, orow, oq = O.frames_exact(cfg, pcm[s], 0, frames)
        assert np.array_equal(col[s], ocol), f"column indices differ in {np.sum(col[s] != ocol)} bins"
        assert np.array_equal(row[s], orow), f"row indices differ in {np.sum(row[s] != orow)} bins"
        assert np.array_equal(pw[s], opw), f"power differs in {np.sum(pw[s] != opw)} bins (max rel {np.max(np.abs(pw[s] - opw) / np.maximum(opw, 1e-300)):.2e})"
        assert np.array_equal(q[s], oq), f"fixed-point energy differs in {np.sum(q[s] != oq)} bins"


@pytest.mark.parametrize("n,hop,frames", [(1024, 256, 64), (4096, 256, 48), (16384, 512, 12), (2048, 128, 64), (8192, 512, 24)])
def test_exact_indices_equal_float64_method(xengine, n, hop, frames, record_property):
    """(column,row) of every bin against the independent float64 three-window method (explicitly windowed FFTs, shares
    no code with the exact path): the mismatch rate is asserted <= 1e-6 (measured: 0)."""
    pcm = _pcm(n, hop, frames, S=2)
    pw, col, row, _ = xengine.parity_dump_exact(pcm, n, hop, True, 0, frames)
    cfg = O.make_cfg(n, hop, True)
    bad = tot = 0
    for s in range(2):
        p64, _, _, c64, r64 = O.frames_f64(cfg, pcm[s], 0, frames)
        bad += int(np.sum(col[s] != c64)) + int(np.sum(row[s] != r64))
        tot += 2 * c64.size
        strong = p64 >= p64.max(axis=1, keepdims=True) * 1e-12
        rel = np.abs(pw[s] - p64) / np.maximum(p64, 1e-300)
        assert rel[strong].max() < 1e-9, rel[strong].max()
    record_property("index_mismatches_vs_float64", bad)
    print(f"N={n}: {bad} index mismatches in {tot} (column,row) values vs the float64 method")
    assert bad <= 1e-6 * tot


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_exact_dump_matches_golden(xengine, path):
    """The committed numpy float64 goldens (tests/golden/make_golden.py): every (column,row) equal, power to 1e-9."""
    g = np.load(path)
    n, hop, f0, fr, re = int(g["n"]), int(g["hop"]), int(g["frame0"]), int(g["frames"]), bool(g["reassign"])
    pw, col, row, _ = xengine.parity_dump_exact(g["pcm"], n, hop, re, f0, fr)
    valid = g["row"] >= 0
    assert np.array_equal(row[0][valid], g["row"][valid])
    assert np.array_equal(col[0][valid], g["col"][valid])
    # bins the golden drops (row -1: gated or off the axis) are dropped here too
    assert np.array_equal(row[0] >= 0, valid)
    gp = g["power"]
    strong = gp >= gp.max(axis=1, keepdims=True) * 1e-12
    assert (np.abs(pw[0] - gp) / np.maximum(gp, 1e-300))[strong].max() < 1e-9


BATCH_CASES = [(4096, 256, 1, 200, True), (4096, 256, 3, 90, True), (16384, 512, 2, 60, True), (1024, 256, 2, 120, False),
               (1024, 256, 2, 120, True), (2048, 128, 2, 100, True), (8192, 512, 2, 50, True), (4096, 1024, 2, 40, True),
               (512, 128, 2, 64, True)]


@pytest.mark.parametrize("n,hop,S,frames,reassign", BATCH_CASES)
def test_exact_batch_equals_bit_model_and_is_reproducible(xengine, n, hop, S, frames, reassign):
    """Finished columns: float32 dB BITS, palette index and RGBA equal to the bit model's; a second run gives the
    same bytes (integer accumulation: the arrival order of the bins does not matter)."""
    pcm = _pcm(n, hop, frames, S=S, extra=11)
    out = xengine.batch(pcm, n, hop, reassign, want=("db", "rgba", "index"))
    odb, orgba, oidx, _ = O.batch_exact(O.make_cfg(n, hop, reassign), pcm)
    assert np.array_equal(out["index"], oidx), f"{np.sum(out['index'] != oidx)} palette indices differ"
    assert np.array_equal(out["rgba"], orgba)
    assert np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32)), \
        f"{np.sum(out['db'].view(np.uint32) != odb.view(np.uint32))} dB cells differ, max {np.max(np.abs(out['db'] - odb)):.3e}"
    again = xengine.batch(pcm, n, hop, reassign, want=("db", "rgba", "index"))
    for k in ("db", "rgba", "index"):
        assert np.array_equal(out[k].view(np.uint8), again[k].view(np.uint8)), f"{k} differs between two runs"


def _columns_f64(cfg, pcm, frames):
    """finished dB columns of one stream by the float64 three-window method, float64 histogram"""
    p64, _, _, c64, r64 = O.frames_f64(cfg, pcm, 0, frames)
    hist = np.zeros((frames, cfg.rows))
    ok = (r64 >= 0) & (c64 >= 0) & (c64 < frames)
    np.add.at(hist, (c64[ok], r64[ok]), p64[ok])
    scale = 32.0 / (3.0 * cfg.n * cfg.n) * cfg.gain * cfg.gain
    return 10.0 * np.log10(hist * scale + 1e-20)


@pytest.mark.parametrize("n,hop,frames", [(4096, 256, 160), (16384, 512, 48), (1024, 256, 200)])
def test_exact_columns_vs_float64_method(xengine, n, hop, frames, record_property):
    """>= 99.99 % of the cells above -60 dB within 8.7e-4 dB (= 1e-4 relative on magnitude) of the float64 method
    (measured: 100 %, max error ~1e-6 dB from the float32 output format)."""
    pcm = _pcm(n, hop, frames, S=1, extra=0)
    db = xengine.batch(pcm, n, hop, True, want=("db",))["db"][0]
    ref = _columns_f64(O.make_cfg(n, hop, True), pcm[0], frames)
    loud = ref > -60.0
    err = np.abs(db.astype(np.float64) - ref)[loud]
    frac = float(np.mean(err < 8.7e-4))
    record_property("cells_within_8.7e-4_dB", frac)
    print(f"N={n}: {loud.sum()} cells above -60 dB, {frac * 100:.4f} % within 8.7e-4 dB, max {err.max():.2e} dB")
    assert frac >= 0.9999


@pytest.mark.parametrize("n,hop,reassign", [(4096, 256, True), (1024, 256, False), (16384, 512, True), (2048, 300, True)])
def test_exact_streaming_equals_batch_bytes(n, hop, reassign):
    """computeSpectrogramColumn (emspec_column, frame by frame + flush) and emspec_push_samples (ragged blocks) in EXACT
    mode produce the SAME bytes as the batch call and the bit model: the u64 ring is order-independent."""
    frames = 40 if n < 16384 else 24
    pcm = synth.streams(1, n + hop * (frames - 1))[0]
    odb, orgba, _, _ = O.batch_exact(O.make_cfg(n, hop, reassign), pcm[None])
    D = emspec.latency_columns(n, hop, reassign)
    with emspec.Engine(mode=emspec.MODE_EXACT) as e:
        got = {}
        for j in range(frames):
            db, rgba, c = e.column(pcm[j * hop:j * hop + n], hop, reassign, want_rgba=True)
            assert c == (j - D if j >= D else -1)
            if c >= 0:
                got[c] = (db, rgba)
        for _ in range(D):
            db, rgba, c = e.flush(want_rgba=True)
            got[c] = (db, rgba)
        assert sorted(got) == list(range(frames))
        for c in range(frames):
            assert np.array_equal(got[c][0].view(np.uint32), odb[0, c].view(np.uint32)), f"column {c} dB differs"
            assert np.array_equal(got[c][1], orgba[0, c])
        e.reset()
        rng = np.random.default_rng(5)
        pos, cols = 0, []
        while pos < pcm.size:
            blk = int(rng.integers(1, 3 * n))
            db, first = e.push_samples(pcm[pos:pos + blk], n, hop, reassign)
            if db.shape[0]:
                assert first == len(cols)
                cols.extend(db)
            pos += blk
        for _ in range(D):
            db, c = e.flush()
            cols.append(db)
        assert len(cols) == frames
        assert np.array_equal(np.stack(cols).view(np.uint32), odb[0].view(np.uint32))


@pytest.mark.parametrize("n,hop,frames", [(4096, 256, 40), (16384, 512, 40)])
def test_exact_custom_axis_and_settings(xengine, n, hop, frames):
    """A warped frequency axis (emspec_set_row_edges_hz; BASELINE configs[4]: 'FFT 16384 ... + low-end log-freq rebinning')
    and non-default display settings in EXACT mode."""
    pcm = _pcm(n, hop, frames, S=1)
    edges = emspec.warped_edges_hz(1024, 20.0, 24000.0, 2.0, 1.6)
    with emspec.Engine(mode=emspec.MODE_EXACT, gain=3.5, db_range=58.0, gate_db=-65.0) as e:
        e.set_row_edges_hz(edges)
        O.set_custom_edges_hz(edges)
        try:
            cfg = O.make_cfg(n, hop, True, gain=3.5, db_range=58.0, gate_db=-65.0)
            pw, col, row, q = e.parity_dump_exact(pcm, n, hop, True, 0, frames)
            opw, ocol, orow, oq = O.frames_exact(cfg, pcm[0], 0, frames)
            assert np.array_equal(row[0], orow) and np.array_equal(col[0], ocol) and np.array_equal(q[0], oq)
            out = e.batch(pcm, n, hop, True, want=("db", "index"))
            odb, _, oidx, _ = O.batch_exact(cfg, pcm, want=("db", "index"))
            assert np.array_equal(out["index"], oidx)
            assert np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32))
        finally:
            O.set_custom_edges_hz(None)


@pytest.mark.parametrize("n,hop,frames,S", [(4096, 256, 150, 3), (4096, 512, 90, 2), (16384, 512, 60, 2)])
def test_exact_axes_the_row_split_does_not_serve(n, hop, frames, S):
    """Round 5 keeps the ring's sparse low rows in global memory (fused kernel at N = 4096, walking scatter elsewhere) when at
    most 6 % of the bins fall there.  A LINEAR frequency axis puts 43 % of the bins below that row: the fused kernel with the
    parked ring (round 4's) / the 16-column tile scatter serve it instead - same bytes as the bit model; and on the default
    axis the diagnostic switch EMSPEC_EXACT_PARKED=1 runs the parked kernel beside the row-split one: identical bytes."""
    import os
    pcm = _pcm(n, hop, frames, S=S)
    edges = np.linspace(40.0, 23000.0, 1025).astype(np.float32)
    with emspec.Engine(mode=emspec.MODE_EXACT) as e:
        e.set_row_edges_hz(edges)
        O.set_custom_edges_hz(edges)
        try:
            out = e.batch(pcm, n, hop, True, want=("db", "index"))
            odb, _, oidx, _ = O.batch_exact(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
        finally:
            O.set_custom_edges_hz(None)
    assert np.array_equal(out["index"], oidx) and np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32))
    if n == 4096:
        with emspec.Engine(mode=emspec.MODE_EXACT, diag=True) as d:
            a = d.batch(pcm, n, hop, True, want=("db", "index"))
            os.environ["EMSPEC_EXACT_PARKED"] = "1"
            try:
                b = d.batch(pcm, n, hop, True, want=("db", "index"))
            finally:
                os.environ.pop("EMSPEC_EXACT_PARKED", None)
        assert np.array_equal(a["index"], b["index"]) and np.array_equal(a["db"].view(np.uint32), b["db"].view(np.uint32))


def test_exact_batch_from_pinned_buffers(xengine):
    """Host-buffer batch from page-locked memory (the fast mode pipelines stream chunks on two HIP streams there; the
    exact mode has one record workspace per engine and must not): > 96 MB of staging, bytes equal to the bit model."""
    n, hop, S = 4096, 256, 6
    L = n + hop * 1500
    pcm = synth.streams(S, L)
    pin = emspec.PinnedArray(pcm.shape, np.float32)
    pin.array[...] = pcm
    Cn = emspec.num_columns(L, n, hop)
    pout = emspec.PinnedArray((S, Cn, xengine.rows), np.float32)
    out = xengine.batch(pin.array, n, hop, True, want=("db",), db_out=pout.array)["db"]
    odb, _, _, _ = O.batch_exact(O.make_cfg(n, hop, True), pcm, want=("db",))
    assert np.array_equal(out.view(np.uint32), odb.view(np.uint32))
    pin.close(); pout.close()


def test_exact_mode_guards(xengine, engine):
    pcm = _pcm(1024, 256, 4, S=1)
    with pytest.raises(emspec.EmspecError) as ei:
        xengine.parity_dump(pcm, 1024, 256, True, 0, 4)
    assert ei.value.code == emspec.ERR_STATE
    with pytest.raises(emspec.EmspecError) as ei:
        engine.parity_dump_exact(pcm, 1024, 256, True, 0, 4)
    assert ei.value.code == emspec.ERR_STATE
    # emspec_uses_fused answers for EXACT engines since ABI 2: one kernel at N = 4096 / 2048 / 1024 (round 6) when the u64 ring fits
    # LDS beside the planes - whole, or with its sparse low rows in the L2 scratch - records otherwise
    assert xengine.fused(4096, 256, True) and xengine.fused(4096, 1024, False) and not xengine.fused(4096, 128, True)
    assert not xengine.fused(16384, 512, True) and not xengine.fused(8192, 512, True)
    assert xengine.fused(1024, 256, True) and xengine.fused(2048, 256, True) and xengine.fused(2048, 128, True)
    assert xengine.fused(1024, 64, True) and not xengine.fused(1024, 32, True)
    assert xengine.mode == emspec.MODE_EXACT and engine.mode == emspec.MODE_FAST
    with pytest.raises(emspec.EmspecError):
        emspec.Engine(mode=7)


def test_exact_silence_and_extremes(xengine):
    """Silence, a full-scale square wave and an over-range input: no NaN, cells stay consistent with the bit model."""
    n, hop = 4096, 256
    L = n + hop * 20
    t = np.arange(L)
    sig = np.stack([np.zeros(L, np.float32), np.sign(np.sin(2 * np.pi * 997.0 * t / 48000.0)).astype(np.float32),
                    (4.0 * np.sin(2 * np.pi * 5000.0 * t / 48000.0)).astype(np.float32)])
    out = xengine.batch(sig, n, hop, True, want=("db", "index"))
    odb, _, oidx, _ = O.batch_exact(O.make_cfg(n, hop, True), sig, want=("db", "index"))
    assert np.isfinite(out["db"]).all()
    assert np.array_equal(out["index"], oidx) and np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32))
    assert out["index"][0].max() == 0


def test_exact_nan_and_inf_samples(xengine):
    """A NaN and an Inf sample poison the frames that contain them (every bin of such a frame fails the power gate and is
    dropped); the other frames are untouched, the output stays finite and equal to the bit model's."""
    n, hop, frames = 4096, 256, 48
    pcm = synth.streams(2, n + hop * (frames - 1))
    pcm[0, 5000] = np.nan
    pcm[1, 9000] = np.inf
    out = xengine.batch(pcm, n, hop, True, want=("db", "index"))
    odb, _, oidx, _ = O.batch_exact(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
    assert np.isfinite(out["db"]).all()
    assert np.array_equal(out["index"], oidx) and np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32))
    pw, col, row, q = xengine.parity_dump_exact(pcm, n, hop, True, 0, frames)
    opw, ocol, orow, oq = O.frames_exact(O.make_cfg(n, hop, True), pcm[0], 0, frames)
    assert np.array_equal(row[0], orow) and np.array_equal(col[0], ocol) and np.array_equal(q[0], oq)
    assert np.array_equal(pw[0], opw, equal_nan=True)
    poisoned = [j for j in range(frames) if j * hop <= 5000 < j * hop + n]
    assert poisoned and np.all(row[0][poisoned] == -1) and np.all(q[0][poisoned] == 0)


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the one-kernel path of the mode at N = 4096 (exact_fused.hip.inc: u64 ring resident in LDS, the binary64 planes
# laid over its first 64 KB, branch-free per-bin core with a 7-instruction reciprocal).  Everything above already runs on
# it at N = 4096; the cases below aim at what is new in it.

def test_short_reciprocal64_equals_ieee_division(diag_engine):
    """recip_normal64 (v_rcp_f64 + the two Newton steps + the residual correction of hipcc's own division expansion,
    without its scaling / special-case instructions) == 1.0 / d bit for bit on the range the per-bin gate guarantees."""
    import ctypes as C
    lib = emspec.load(diag=True)
    lib.emspec_debug_recip64.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(20264)
    mant = np.concatenate([rng.integers(0, 1 << 52, 3_000_000, dtype=np.uint64),
                           np.array([0, 1, 2, 3, (1 << 52) - 1, (1 << 52) - 2, 1 << 51, (1 << 51) - 1, (1 << 51) + 1,
                                     0x6A09E667F3BCD, 0x6A09E667F3BCC, 0x6A09E667F3BCE], np.uint64)])    # sqrt(2) - 1 and neighbours
    expo = rng.integers(1023 - 699, 1023 + 999, mant.size, dtype=np.uint64)
    expo[-12:] = 1023
    vals = ((expo << np.uint64(52)) | mant).view(np.float64)
    # every binade edge of the range and the gate's own ends at the default configuration
    extra = np.concatenate([np.ldexp(1.0, np.arange(-699, 1000)), np.nextafter(np.ldexp(1.0, np.arange(-699, 1000)), 0),
                            np.nextafter(np.ldexp(1.0, np.arange(-699, 999)), np.inf), [64e-8, 2.0 ** 35]])
    vals = np.ascontiguousarray(np.concatenate([vals, extra]))
    a = np.empty_like(vals)
    b = np.empty_like(vals)
    assert lib.emspec_debug_recip64(diag_engine._h, vals.ctypes.data, vals.size, a.ctypes.data, b.ctypes.data) == 0
    bad = a.view(np.uint64) != b.view(np.uint64)
    assert not bad.any(), f"{bad.sum()} of {vals.size} reciprocals differ, first d = {vals[bad][0]!r}"
    # ... and the GPU's division is the IEEE one (what the CPU bit model computes)
    assert np.array_equal(b.view(np.uint64), (1.0 / vals).view(np.uint64))


FUSED_CASES = [  # hop, frames, S, reassign, rows, seglen (0: the launcher's plan), engine settings
    (256, 300, 2, True, 1024, 64, {}),          # several segments per stream: halo frames, ring restarts
    (256, 129, 1, True, 1024, 65, {}),          # odd segment length, a last segment of one column
    (256, 5, 2, True, 1024, 0, {}),             # fewer frames than the reassignment reach
    (256, 1, 1, True, 1024, 0, {}),             # one frame
    (512, 150, 2, True, 1024, 0, {}),           # D = 4: 10 ring slots (80 KB), the planes cover most of the ring
    (1024, 90, 2, True, 1024, 70, {}),          # D = 2: the ring (48 KB) is smaller than the planes
    (300, 70, 1, True, 1024, 0, {}),            # a hop that is no power of two (D = 7)
    (256, 120, 2, False, 1024, 64, {}),         # reassignment off: D = 0, three slots, the branchy per-bin core
    (256, 90, 2, True, 512, 0, {}),             # fewer rows
    (256, 90, 1, True, 64, 0, {}),
    (256, 90, 2, True, 1024, 0, {"power_floor": 0.0}),   # no power floor: the generic core (recip_normal64's range is not guaranteed)
    (256, 60, 1, True, 1024, 0, {"gain": 3.5, "db_range": 58.0, "gate_db": -65.0}),
]


@pytest.mark.parametrize("hop,frames,S,reassign,rows,seglen,kw", FUSED_CASES)
def test_exact_fused_kernel_shapes(hop, frames, S, reassign, rows, seglen, kw, monkeypatch):
    """exact_fused4096_kernel over its shape space: bytes equal to the binary64 bit model (dB bits, palette index, RGBA)."""
    n = 4096
    if seglen:
        monkeypatch.setenv("EMSPEC_SEGLEN", str(seglen))
    pcm = _pcm(n, hop, frames, S=S, extra=5)
    with emspec.Engine(mode=emspec.MODE_EXACT, diag=True, rows=rows, **kw) as e:
        out = e.batch(pcm, n, hop, reassign, want=("db", "rgba", "index"))
    odb, orgba, oidx, _ = O.batch_exact(O.make_cfg(n, hop, reassign, rows=rows, **kw), pcm)
    assert out["db"].shape == odb.shape == (S, frames, rows)
    assert np.array_equal(out["index"], oidx), f"{np.sum(out['index'] != oidx)} palette indices differ"
    assert np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32)), \
        f"{np.sum(out['db'].view(np.uint32) != odb.view(np.uint32))} dB cells differ"
    assert np.array_equal(out["rgba"], orgba)


SMALL_FUSED_CASES = [  # n, hop, frames, S, reassign, rows, seglen, engine settings
    (1024, 256, 300, 2, True, 1024, 64, {}),      # D = 2, four frames per team and half-iteration: 12 ring slots, low rows in the L2 scratch
    (1024, 256, 131, 1, True, 1024, 33, {}),      # odd segment length, blocks that straddle the segment's end
    (1024, 256, 3, 2, True, 1024, 0, {}),         # fewer frames than one block
    (1024, 256, 1, 1, True, 1024, 0, {}),
    (1024, 128, 200, 2, True, 1024, 0, {}),       # D = 4
    (1024, 64, 150, 1, True, 1024, 0, {}),        # D = 8: 24 slots, 604 low rows - more low quads per block than the team has threads
    (1024, 96, 150, 2, True, 1024, 40, {}),       # D = 6: 20 slots, 520 low rows
    (1024, 100, 90, 1, True, 1024, 0, {}),        # a hop that is no power of two
    (1024, 256, 120, 2, False, 1024, 48, {}),     # reassignment off: D = 0, the branchy per-bin core
    (1024, 256, 90, 2, True, 512, 0, {}),         # the whole ring in LDS (no low rows)
    (1024, 256, 90, 1, True, 64, 0, {}),
    (1024, 256, 90, 2, True, 1024, 0, {"power_floor": 0.0}),
    (2048, 256, 300, 2, True, 1024, 64, {}),      # D = 4, two frames per team and half-iteration
    (2048, 128, 140, 1, True, 1024, 37, {}),      # D = 8: 20 slots
    (2048, 512, 5, 2, True, 1024, 0, {}),
    (2048, 300, 80, 1, True, 1024, 0, {}),
    (2048, 256, 100, 2, False, 1024, 0, {}),
    (2048, 256, 90, 1, True, 256, 0, {"gain": 3.5, "db_range": 58.0, "gate_db": -65.0}),
]


@pytest.mark.parametrize("n,hop,frames,S,reassign,rows,seglen,kw", SMALL_FUSED_CASES)
def test_exact_fused_small_sizes(n, hop, frames, S, reassign, rows, seglen, kw, monkeypatch):
    """The one-kernel EXACT path at N = 1024 and 2048 (round 6: exact_fused4096_lr_kernel<.., S>, the 4096-point carrier network with
    its first S stages skipped = 2 / 4 frames per team and half-iteration): the engine reports a fused kernel for the shape, and
    the bytes equal the binary64 bit model (dB bits, palette index, RGBA) - and the records path's (EMSPEC_EXACT_RECORDS=1)."""
    if seglen:
        monkeypatch.setenv("EMSPEC_SEGLEN", str(seglen))
    pcm = _pcm(n, hop, frames, S=S, extra=5)
    with emspec.Engine(mode=emspec.MODE_EXACT, diag=True, rows=rows, **kw) as e:
        assert e.fused(n, hop, reassign)
        out = e.batch(pcm, n, hop, reassign, want=("db", "rgba", "index"))
        e.device_status()
    odb, orgba, oidx, _ = O.batch_exact(O.make_cfg(n, hop, reassign, rows=rows, **kw), pcm)
    assert out["db"].shape == odb.shape == (S, frames, rows)
    assert np.array_equal(out["index"], oidx), f"{np.sum(out['index'] != oidx)} palette indices differ"
    assert np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32)), \
        f"{np.sum(out['db'].view(np.uint32) != odb.view(np.uint32))} dB cells differ"
    assert np.array_equal(out["rgba"], orgba)


def test_exact_fused_custom_axis_uses_generic_core(monkeypatch):
    """A warped (not log-spaced) axis: the fused kernel runs its generic per-bin core (binary search of the edge table)."""
    n, hop, frames = 4096, 256, 80
    monkeypatch.setenv("EMSPEC_SEGLEN", "64")
    pcm = _pcm(n, hop, frames, S=1)
    edges = emspec.warped_edges_hz(1024, 20.0, 24000.0, 2.0, 1.6)
    with emspec.Engine(mode=emspec.MODE_EXACT, diag=True) as e:
        e.set_row_edges_hz(edges)
        O.set_custom_edges_hz(edges)
        try:
            out = e.batch(pcm, n, hop, True, want=("db", "index"))
            odb, _, oidx, _ = O.batch_exact(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
        finally:
            O.set_custom_edges_hz(None)
    assert np.array_equal(out["index"], oidx) and np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32))


def test_exact_full_size_spot_checks_against_bit_model(xengine):
    """BASELINE configs[2] at full size in EXACT mode (64 streams x 2^22 samples, 1,047,616 columns, one launch of the fused
    kernel: 256 workgroups of 4,093 columns): random (stream, column) cells, the first and last columns of a stream and the
    columns on both sides of every kind of segment boundary against the bit model evaluated on the slice of audio that can
    reach them (column c depends on frames c-8 .. c+8 only) - array_equal on the dB bits and the palette index; a second
    run gives the same bytes."""
    import torch
    n, hop, D = 4096, 256, 8
    S, L = 64, 1 << 22
    base = synth.streams(4, L)
    rng = np.random.default_rng(4242)
    pcm = np.stack([np.roll(base[s % 4], 1237 * s) * (0.5 + 0.5 * ((s * 7) % 5) / 4) for s in range(S)]).astype(np.float32)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(pcm).to(dev)
    Cn = (L - n) // hop + 1
    db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
    idx = torch.empty((S, Cn, 1024), dtype=torch.uint8, device=dev)
    xengine.batch_device(x, n, hop, True, db=db, index=idx)
    torch.cuda.synchronize()
    cfg = O.make_cfg(n, hop, True)
    seg = -(-Cn // 4)      # 64 streams on 256 CUs: four segments per stream
    cols = [0, 1, 7, 8, 9, Cn - 1, Cn - 2, Cn - 9] + [k * seg + d for k in (1, 2, 3) for d in (-9, -1, 0, 1, 8)] + \
        list(rng.integers(20, Cn - 20, 10))
    for c in cols:
        s = int(rng.integers(0, S))
        f0, f1 = max(0, c - D), min(Cn - 1, c + D)
        odb, _, oidx, _ = O.batch_exact(cfg, pcm[s, f0 * hop:f1 * hop + n][None], want=("db", "index"), threads=1)
        assert np.array_equal(db[s, c].cpu().numpy().view(np.uint32), odb[0, c - f0].view(np.uint32)), f"dB bits differ at stream {s} column {c}"
        assert np.array_equal(idx[s, c].cpu().numpy(), oidx[0, c - f0]), f"index differs at stream {s} column {c}"
    chk = (int(idx.sum(dtype=torch.int64).item()), float(db.double().sum().item()))
    db.zero_(); idx.zero_()
    xengine.batch_device(x, n, hop, True, db=db, index=idx)
    torch.cuda.synchronize()
    assert chk == (int(idx.sum(dtype=torch.int64).item()), float(db.double().sum().item()))


def test_exact_n16384_full_size_spot_checks_against_bit_model(xengine):
    """BASELINE configs[4] as named, in EXACT mode: 64 streams x 2^22 samples, FFT 16384, hop 512 (522,304 columns).  This
    shape runs the two-kernel records path with the streams in chunks of the record workspace (12 B per bin: ~0.8 GB per
    stream, so a 4 GiB workspace takes five streams at a time) - the spot checks sit on both sides of stream-chunk
    boundaries, on the first / last columns of streams, on both sides of scatter-tile boundaries and at random cells, each
    against the binary64 bit model evaluated on the slice of audio that can reach the column (frames c-16 .. c+16):
    array_equal on the dB bits and the palette index; a second run gives the same bytes."""
    import torch
    n, hop, D = 16384, 512, 16
    S, L = 64, 1 << 22
    base = synth.streams(4, L)
    rng = np.random.default_rng(1638)
    pcm = np.stack([np.roll(base[s % 4], 2311 * s) * (0.4 + 0.6 * ((s * 5) % 7) / 6) for s in range(S)]).astype(np.float32)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(pcm).to(dev)
    Cn = (L - n) // hop + 1
    assert S * Cn == 522304
    db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
    idx = torch.empty((S, Cn, 1024), dtype=torch.uint8, device=dev)
    assert not xengine.fused(n, hop, True)
    xengine.batch_device(x, n, hop, True, db=db, index=idx)
    torch.cuda.synchronize()
    cfg = O.make_cfg(n, hop, True)
    # (stream, column): every stream next to a possible chunk boundary (chunks of 1..8 streams all put one between two of
    # these), the ends of streams, multiples of 16 and 32 columns +- 1 (scatter tiles), random cells
    cells = [(s, c) for s in (0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 15, 16, 31, 32, 63) for c in (0, Cn - 1)]
    cells += [(int(rng.integers(0, S)), c) for c in (15, 16, 17, 31, 32, 33, 4095, 4096, 4097, Cn - 17, Cn - 16)]
    cells += [(int(rng.integers(0, S)), int(rng.integers(40, Cn - 40))) for _ in range(12)]
    for s, c in cells:
        f0, f1 = max(0, c - D), min(Cn - 1, c + D)
        odb, _, oidx, _ = O.batch_exact(cfg, pcm[s, f0 * hop:f1 * hop + n][None], want=("db", "index"), threads=1)
        assert np.array_equal(db[s, c].cpu().numpy().view(np.uint32), odb[0, c - f0].view(np.uint32)), f"dB bits differ at stream {s} column {c}"
        assert np.array_equal(idx[s, c].cpu().numpy(), oidx[0, c - f0]), f"index differs at stream {s} column {c}"
    chk = (int(idx.sum(dtype=torch.int64).item()), float(db.double().sum().item()))
    db.zero_(); idx.zero_()
    xengine.batch_device(x, n, hop, True, db=db, index=idx)
    torch.cuda.synchronize()
    assert chk == (int(idx.sum(dtype=torch.int64).item()), float(db.double().sum().item()))


@pytest.mark.parametrize("n,hop,S,frames", [(16384, 512, 5, 70), (8192, 512, 4, 90), (4096, 128, 3, 120)])
def test_exact_record_path_stream_chunks(n, hop, S, frames, monkeypatch):
    """The shapes that still run as two kernels with per-bin records in HBM (N != 4096, or N = 4096 at a hop whose u64 ring does
    not fit in LDS) process their streams in chunks so that the record workspace stays bounded; with a forced 64 MB budget the
    batch below takes several chunks (one or two streams each) - bytes equal to the bit model across the chunk boundaries."""
    monkeypatch.setenv("EMSPEC_RECORD_BUDGET_MB", "64")
    pcm = _pcm(n, hop, frames, S=S, extra=3)
    with emspec.Engine(mode=emspec.MODE_EXACT, diag=True) as e:
        assert not e.fused(n, hop, True)
        out = e.batch(pcm, n, hop, True, want=("db", "index"))
    odb, _, oidx, _ = O.batch_exact(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
    assert np.array_equal(out["index"], oidx) and np.array_equal(out["db"].view(np.uint32), odb.view(np.uint32))
