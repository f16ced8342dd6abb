"""GPU parity, second tier (VERDICT r01 "close the parity gaps that can be closed"):

 (a) the HIP parity dump against every committed golden vector (tests/golden/*.npz: numpy three-window float64,
     generator committed) - not only against this repo's C oracle;
 (b) the streaming entry points (emspec_column's one-launch path, emspec_push_samples, emspec_column_flush) against
     the ORACLE's batch output, not against the engine's own batch call;
 (d) finished dB columns against the float64 three-window method scattered with float64 indices, with a stated
     statistical bound (float32 vs float64 cannot agree on a bin that sits on a cell edge);
 (e) the low-end warp of the frequency axis (emspec_set_row_edges_hz) at FFT 16384 (BASELINE configs[4]:
     "low-end log-freq rebinning").
The reference implementation itself is unavailable (private source): parity with it stays UNPINNED.
"""
import glob
import os

import numpy as np
import pytest

import emspec
import oracle as O
from emspec import synth

pytestmark = pytest.mark.gpu

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


# float32-vs-float64 (row, column) mismatch rates measured on the committed vectors, x 2
# (measured r03: n1024_h256 3.34e-4 / 0; n1024_h256_off 0 / 0; n16384_h512 0 / 4.43e-4; n4096_h256 0 / 0; a zero gets room for one bin of the vector)
GOLDEN_BOUNDS = {"n1024_h256": (6.7e-4, 3.4e-4), "n1024_h256_off": (3.4e-4, 3.4e-4), "n16384_h512": (6.4e-5, 8.9e-4), "n4096_h256": (1.3e-4, 1.3e-4)}


def test_golden_vectors_exist():
    assert len(GOLD) >= 4


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_hip_dump_matches_golden(engine, path, record_property):
    """HIP per-bin dump vs the golden float64 vectors: power within 1e-4 (relative) on every bin within 60 dB of its
    frame's maximum; integer (column,row) equal except for bins whose continuous coordinate sits on a cell edge - the
    measured mismatch rate is asserted (<= 2x the rate measured on these vectors, GOLDEN_BOUNDS) and recorded."""
    g = np.load(path)
    n, hop, f0, fr, re = int(g["n"]), int(g["hop"]), int(g["frame0"]), int(g["frames"]), bool(g["reassign"])
    pw, col, row = engine.parity_dump(g["pcm"], n, hop, re, f0, fr)
    pw, col, row = pw[0], col[0], row[0]
    gp = g["power"]
    strong = gp >= gp.max(axis=1, keepdims=True) * 1e-6
    rel = np.abs(pw - gp) / np.maximum(gp, 1e-300)
    assert rel[strong].max() < 1e-4, rel[strong].max()
    valid = g["row"] >= 0
    row_rate = float(np.mean(row[valid] != g["row"][valid]))
    col_rate = float(np.mean(col[valid] != g["col"][valid]))
    record_property("row_mismatch_rate_vs_float64", row_rate)
    record_property("col_mismatch_rate_vs_float64", col_rate)
    print(f"{os.path.basename(path)}: max rel power err (strong bins) {rel[strong].max():.2e}, "
          f"row mismatch {row_rate:.2e}, col mismatch {col_rate:.2e} of {int(valid.sum())} bins")
    print(f"MEASURED golden {os.path.basename(path)[:-4]}: row {row_rate:.3e} col {col_rate:.3e}")
    rb, cb = GOLDEN_BOUNDS[os.path.basename(path)[:-4]]     # 2 x measured on these fixed vectors
    assert row_rate <= rb and col_rate <= cb, (row_rate, col_rate)
    # and the bit model says exactly what the HIP path says (the golden pins the oracle, the oracle pins the kernels)
    opw, ocol, orow = O.frames_f32(O.make_cfg(n, hop, re), g["pcm"], f0, fr)
    assert np.array_equal(col, ocol) and np.array_equal(row, orow) and np.array_equal(pw, opw)


@pytest.mark.parametrize("n,hop,reassign", [(4096, 256, True), (1024, 256, False), (16384, 512, True), (2048, 300, True)])
def test_column_call_matches_oracle(n, hop, reassign):
    """emspec_column (the renderer's computeSpectrogramColumn; default one-launch path, dB + RGBA) frame by frame and
    emspec_column_flush against the ORACLE's batch columns."""
    frames = 40
    pcm = synth.streams(1, n + hop * (frames - 1))[0]
    cfg = O.make_cfg(n, hop, reassign)
    odb, orgba, _ = O.batch_f32(cfg, pcm[None], want=("db", "rgba"))
    D = emspec.latency_columns(n, hop, reassign)
    cols, rg = {}, {}
    with emspec.Engine() as e:
        for j in range(frames):
            db, rgba, c = e.column(pcm[j * hop:j * hop + n], hop, reassign, want_rgba=True)
            assert c == (j - D if j >= D else -1)
            if c >= 0:
                cols[c], rg[c] = db, rgba
        for _ in range(min(D, frames)):
            db, rgba, c = e.flush(want_rgba=True)
            cols[c], rg[c] = db, rgba
    got = np.stack([cols[c] for c in range(frames)])
    assert np.max(np.abs(got - odb[0])) < 8.7e-4
    assert np.mean(np.stack([rg[c] for c in range(frames)]) != orgba[0]) < 1e-3


@pytest.mark.parametrize("n,hop,reassign,block", [(4096, 256, True, 128), (4096, 256, True, 5000), (1024, 256, False, 777),
                                                  (16384, 512, True, 20000), (2048, 128, True, 2048)])
def test_push_samples_matches_oracle(n, hop, reassign, block):
    """emspec_push_samples fed with blocks of any length (a WebAudio worklet's 128 samples; blocks longer than a frame),
    then emspec_column_flush, against the ORACLE's batch columns."""
    frames = 70
    pcm = synth.streams(1, n + hop * (frames - 1))[0]
    cfg = O.make_cfg(n, hop, reassign)
    odb, _, _ = O.batch_f32(cfg, pcm[None], want=("db",))
    D = emspec.latency_columns(n, hop, reassign)
    out, nxt = [], 0
    with emspec.Engine() as e:
        for a in range(0, pcm.size, block):
            db, first = e.push_samples(pcm[a:a + block], n, hop, reassign)
            if len(db):
                assert first == nxt
                out.append(db)
                nxt += len(db)
        assert nxt == max(frames - D, 0)
        for _ in range(min(D, frames)):
            db, c = e.flush()
            assert c == nxt
            out.append(db[None])
            nxt += 1
    got = np.concatenate(out)
    assert got.shape == odb[0].shape
    assert np.max(np.abs(got - odb[0])) < 8.7e-4


# measured r03: 4096: 100 % / 1.05e-6 dB; 16384: 99.7845 % / 1.68e-6 dB; 1024: 100 % / 1.04e-6 dB -> disagreeing share and median x 2
COLUMN_BOUNDS = {4096: (0.9995, 2.2e-6), 16384: (0.9956, 3.4e-6), 1024: (0.9995, 2.2e-6)}


@pytest.mark.parametrize("n,hop", [(4096, 256), (16384, 512), (1024, 256)])
def test_finished_columns_vs_float64_three_window(engine, n, hop, record_property):
    """Finished dB columns of the HIP path against the float64 textbook method (three explicitly windowed DFTs,
    float64 reassignment, float64 indices, float64 scatter and dB): the independent end-to-end check.
    float32 and float64 legitimately disagree on a bin whose coordinate sits on a cell edge (that bin's energy then
    lands in the neighbouring cell), and cells near the -80 dB display floor carry float32 FFT rounding noise.  Bound
    asserted: among cells the float64 method puts above -60 dB, the share that disagrees by more than 0.01 dB and the
    median error stay within 2x what was measured (COLUMN_BOUNDS); the rates are recorded.  (The EXACT mode closes this
    gap: tests/test_gpu_exact.py.)"""
    frames = 48
    pcm = synth.streams(1, n + hop * (frames - 1))
    cfg = O.make_cfg(n, hop, True)
    p64, _, _, c64, r64 = O.frames_f64(cfg, pcm[0], 0, frames)
    R = cfg.rows
    hist = np.zeros((frames, R), np.float64)
    ok = (r64 >= 0) & (c64 >= 0) & (c64 < frames)
    np.add.at(hist, (c64[ok], r64[ok]), p64[ok])
    scale = 32.0 / (3.0 * float(n) ** 2)
    db64 = 10.0 * np.log10(hist * scale + 1e-20)
    got = engine.batch(pcm, n, hop, True, want=("db",))["db"][0].astype(np.float64)
    err = np.abs(got - db64)
    strong = db64 > -60.0
    agree = float(np.mean(err[strong] < 1e-2))
    med = float(np.median(err[strong]))
    record_property("cells_above_-60dB_within_0.01dB", agree)
    record_property("median_abs_err_dB", med)
    print(f"N={n}: {int(strong.sum())} cells above -60 dB, {agree:.4%} within 0.01 dB of float64, median |err| {med:.2e} dB, "
          f"max {err[strong].max():.3f} dB")
    print(f"MEASURED columns_vs_float64 N={n}: agree {agree:.6f} median {med:.3e}")
    amin, mmax = COLUMN_BOUNDS[n]      # disagreeing share and median error: 2 x measured
    assert agree >= amin and med <= mmax, (agree, med)


@pytest.mark.parametrize("boost,zoom", [(1.5, 1.0), (2.0, 1.6)])
def test_low_end_rebinning_at_n16384(boost, zoom):
    """BASELINE configs[4]: FFT 16384, hop 512, reassignment ON + low-end log-frequency rebinning.  The warped axis
    (emspec_warped_edges_hz -> emspec_set_row_edges_hz: arbitrary monotone edges, binary-search row lookup) through
    the fused N = 16384 kernel, against the oracle with the same edge table; and back to the log axis."""
    n, hop, frames, S = 16384, 512, 80, 2
    pcm = synth.streams(S, n + hop * (frames - 1))
    with emspec.Engine() as e:
        edges = emspec.warped_edges_hz(e.rows, 20.0, 24000.0, low_end_boost=boost, freq_scale=zoom)
        assert np.all(np.diff(edges) > 0)
        e.set_row_edges_hz(edges)
        assert e.fused(n, hop, True)
        assert np.array_equal(e.row_edges_hz(), edges)
        out = e.batch(pcm, n, hop, True, want=("db", "index"))
        pw, col, row = e.parity_dump(pcm[:1], n, hop, True, 0, 6)
        O.set_custom_edges_hz(edges)
        try:
            cfg = O.make_cfg(n, hop, True)
            odb, _, oidx = O.batch_f32(cfg, pcm, want=("db", "index"))
            opw, ocol, orow = O.frames_f32(cfg, pcm[0], 0, 6)
        finally:
            O.set_custom_edges_hz(None)
        assert np.array_equal(row[0], orow) and np.array_equal(col[0], ocol) and np.array_equal(pw[0], opw)
        assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
        assert np.mean(out["index"] != oidx) < 1e-3
        # the warp moved energy: the low rows are wider in Hz than on the log axis
        e.set_row_edges_hz(None)
        back = e.batch(pcm, n, hop, True, want=("db",))["db"]
        olog, _, _ = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("db",))
        assert np.max(np.abs(back - olog)) < 8.7e-4
        assert np.max(np.abs(back - out["db"])) > 1.0


def test_pinned_batch_over_96mb_with_display_postprocess():
    """a host-buffer batch from page-locked memory larger than the 96 MB chunk budget used to run its
    chunks on two HIP streams that shared the display post-process workspaces.  With set_display active it must
    equal the oracle's sequential post-process."""
    n, hop, S = 4096, 256, 6
    L = n + hop * 3999                      # 4000 columns x 1024 rows x 4 B = 16 MB of dB per stream + 4 MB pcm
    pcm = synth.streams(S, L)
    with emspec.Engine() as e:
        e.set_display(smoothing=0.6, agc_strength=0.5)
        pin_in = emspec.PinnedArray((S, L), np.float32)
        pin_out = emspec.PinnedArray((S, 4000, e.rows), np.float32)
        try:
            pin_in.array[:] = pcm
            out = e.batch(pin_in.array, n, hop, True, want=("db",), db_out=pin_out.array)["db"].copy()
        finally:
            pin_in.close()
            pin_out.close()
        e.set_display(0.0, 0.0)
        raw = e.batch(pcm, n, hop, True, want=("db",))["db"]
    cfg = O.make_cfg(n, hop, True)
    ref, _, _ = O.postprocess(raw, 0.6, 0.5, cfg)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 2e-3
