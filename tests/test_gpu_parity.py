"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Integer (column,row) indices must be EXACT against the float32 bit model;
power within 1e-4 relative against the float64 three-window oracle for bins
within 60 dB of their frame's maximum (below that float32 rounding noise of
any FFT dominates).  The reference implementation itself is unavailable
(private source) — see DESIGN.md "Parity statement".
"""
import numpy as np
import pytest

import oracle as O
from palette import palette_close
from emspec import synth

pytestmark = pytest.mark.gpu

CASES = [(1024, 256), (4096, 256), (16384, 512), (2048, 128), (8192, 1024), (256, 64), (512, 512)]


# measured float32-vs-float64 index mismatch rates of test_dump_vs_float64_oracle (row, column), x 2
# (measured r03: 1024: 0 / 0 of 3,946 bins; 4096: 0 / 5.66e-4 of 15,912; 16384: 1.57e-5 / 2.20e-4 of 63,675; a zero is given room for one bin)
F64_DUMP_BOUNDS = {1024: (2.6e-4, 2.6e-4), 4096: (6.3e-5, 1.14e-3), 16384: (3.2e-5, 4.4e-4)}


def _pcm(n, hop, frames, S=2):
    return synth.streams(S, n + hop * (frames - 1) + 37)


@pytest.mark.parametrize("n,hop", CASES)
@pytest.mark.parametrize("reassign", [True, False])
def test_dump_matches_bit_model(engine, n, hop, reassign):
    frames = 12
    pcm = _pcm(n, hop, frames)
    pw, col, row = engine.parity_dump(pcm, n, hop, reassign, 0, frames)
    cfg = O.make_cfg(n, hop, reassign)
    for s in range(pcm.shape[0]):
        opw, ocol, orow = O.frames_f32(cfg, pcm[s], 0, frames)
        assert np.array_equal(col[s], ocol), f"column indices differ: {np.sum(col[s] != ocol)}"
        assert np.array_equal(row[s], orow), f"row indices differ: {np.sum(row[s] != orow)}"
        # same operation order -> power is bit-identical as well
        assert np.array_equal(pw[s], opw), f"power differs in {np.sum(pw[s] != opw)} bins"


@pytest.mark.parametrize("n,hop", [(1024, 256), (4096, 256), (16384, 512)])
def test_dump_vs_float64_oracle(engine, n, hop):
    frames = 8
    pcm = _pcm(n, hop, frames, S=1)
    pw, col, row = engine.parity_dump(pcm, n, hop, True, 0, frames)
    cfg = O.make_cfg(n, hop, True)
    p64, that, khat, c64, r64 = O.frames_f64(cfg, pcm[0], 0, frames)
    strong = p64 >= p64.max(axis=1, keepdims=True) * 1e-6
    rel = np.abs(pw[0] - p64) / np.maximum(p64, 1e-300)
    assert rel[strong].max() < 1e-4, rel[strong].max()
    # float32 vs float64 may legitimately disagree on a bin whose continuous
    # coordinate sits on a cell edge; it must be rare
    valid = r64 >= 0
    rr, cr = float(np.mean(row[0][valid] != r64[valid])), float(np.mean(col[0][valid] != c64[valid]))
    print(f"MEASURED dump_vs_float64 N={n}: row mismatch {rr:.3e}, col mismatch {cr:.3e} of {int(valid.sum())} bins")
    # bounds = 2 x the rates measured on this (deterministic) input, so that drift is caught
    rb, cb = F64_DUMP_BOUNDS[n]
    assert rr <= rb and cr <= cb, (rr, cr)


def test_tables_match_oracle(engine):
    for n in (1024, 4096, 16384):
        tw, eb = engine.tables(n)
        otw, oeb = O.tables(O.make_cfg(n, 256, True))
        assert np.array_equal(tw, otw)
        assert np.array_equal(eb, oeb)


@pytest.mark.parametrize("n,hop,reassign", [(4096, 256, True), (1024, 256, False), (1024, 256, True), (16384, 512, True)])
def test_batch_columns_match_oracle(engine, n, hop, reassign):
    frames = 40 if n < 16384 else 36
    pcm = _pcm(n, hop, frames, S=2)
    out = engine.batch(pcm, n, hop, reassign, want=("db", "rgba", "index"))
    cfg = O.make_cfg(n, hop, reassign)
    odb, orgba, oidx = O.batch_f32(cfg, pcm)
    assert out["db"].shape == odb.shape
    # 1e-4 relative on magnitude == 8.7e-4 dB
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    palette_close(out["index"], oidx)
    lut = O.default_lut()
    assert np.array_equal(out["rgba"], lut[out["index"]])


def test_streaming_equals_batch(engine):
    n, hop = 4096, 256
    frames = 30
    pcm = _pcm(n, hop, frames, S=1)[0]
    pcm = pcm[: n + hop * (frames - 1)]
    ref = engine.batch(pcm[None], n, hop, True, want=("db",))["db"][0]
    engine.reset()
    D = 8
    cols = {}
    for j in range(frames):
        db, c = engine.column(pcm[j * hop:j * hop + n], hop, True)
        if j < D:
            assert c == -1 and np.all(db == db[0])
        else:
            assert c == j - D
            cols[c] = db
    for _ in range(D):
        db, c = engine.flush()
        cols[c] = db
    with pytest.raises(Exception):
        engine.flush()
    engine.reset()
    got = np.stack([cols[c] for c in range(frames)])
    assert np.max(np.abs(got - ref)) < 8.7e-4


@pytest.mark.parametrize("n,hop", [(4096, 256), (4096, 512), (4096, 1024), (8192, 512), (8192, 1024),
                                   (2048, 128), (2048, 256), (2048, 200), (2048, 2048), (1024, 256), (1024, 128), (1024, 100),
                                   (1024, 1024), (4096, 230), (4096, 300), (4096, 2048), (4096, 4096)])
@pytest.mark.parametrize("frames,S,reassign", [(203, 3, True), (64, 1, True), (5, 2, True), (130, 2, False), (1, 1, True)])
def test_fused_segments_match_oracle(engine, frames, S, reassign, n, hop):
    """Fused LDS-ring kernels (N=4096 at hop 256/512/1024, N=8192 at hop 512/1024): several segments
    per stream, odd column counts, fewer columns than the reassignment reach, reassign off."""
    assert engine.fused(n, hop, reassign)
    pcm = _pcm(n, hop, frames, S=S)
    out = engine.batch(pcm, n, hop, reassign, want=("db", "rgba", "index"))
    cfg = O.make_cfg(n, hop, reassign)
    odb, orgba, oidx = O.batch_f32(cfg, pcm)
    assert out["db"].shape == odb.shape == (S, frames, 1024)
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    palette_close(out["index"], oidx)
    assert np.array_equal(out["rgba"], O.default_lut()[out["index"]])


@pytest.mark.parametrize("hop,frames,S,reassign,rows", [(512, 150, 2, True, 1024), (512, 40, 1, True, 1024), (512, 9, 2, True, 1024),
                                                         (512, 1, 1, True, 1024), (1024, 90, 2, True, 1024), (2048, 70, 1, True, 1024),
                                                         (16384, 5, 2, True, 1024), (700, 60, 1, True, 1024), (512, 130, 2, False, 1024),
                                                         (100, 200, 1, False, 1024), (512, 80, 2, True, 256), (384, 40, 1, True, 512)])
def test_fused_n16384_matches_oracle(hop, frames, S, reassign, rows):
    """BASELINE configs[4] shape (FFT 16384): the fused walking kernel whose column ring is parked in registers while
    the FFT owns the LDS.  Several segments per stream (the launcher cuts >= 128-column segments), fewer frames than
    the reassignment reach (2D = 32), one frame, other hops / row counts, reassignment off (D = 0: one ring slot)."""
    import emspec
    n = 16384
    pcm = _pcm(n, hop, frames, S=S)
    eng = emspec.Engine(rows=rows)
    try:
        assert eng.fused(n, hop, reassign)
        out = eng.batch(pcm, n, hop, reassign, want=("db", "rgba", "index"))
    finally:
        eng.close()
    odb, orgba, oidx = O.batch_f32(O.make_cfg(n, hop, reassign, rows=rows), pcm)
    assert out["db"].shape == odb.shape == (S, frames, rows)
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    palette_close(out["index"], oidx)
    assert np.array_equal(out["rgba"], O.default_lut()[out["index"]])


@pytest.mark.parametrize("hop,seglen,frames", [(512, 130, 600), (512, 33, 200), (1024, 65, 300)])
def test_fused_n16384_short_segments(hop, seglen, frames, monkeypatch):
    """Same kernel with many short segments per stream (each recomputes a 2D-frame halo), odd segment lengths."""
    import emspec
    n = 16384
    monkeypatch.setenv("EMSPEC_SEGLEN", str(seglen))
    pcm = _pcm(n, hop, frames, S=2)
    eng = emspec.Engine(diag=True)      # EMSPEC_SEGLEN is a switch of the diagnostic build (same kernel sources)
    try:
        out = eng.batch(pcm, n, hop, True, want=("db", "index"))
    finally:
        eng.close()
    odb, _, oidx = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    palette_close(out["index"], oidx)


@pytest.mark.parametrize("n,hop,frames,S", [(4096, 256, 2300, 2), (4096, 512, 700, 3), (8192, 512, 900, 2), (2048, 128, 1500, 2),
                                            (1024, 256, 2000, 2), (16384, 512, 700, 2), (4096, 256, 130, 1)])
def test_shared_device_segment_plan(n, hop, frames, S, monkeypatch):
    """When the engine shares its GPU with a collective (communicator, world > 1) a launch cuts the last quarter of every
    stream into quarter-length segments and dispatches the long ones first (SegPlan, fused.hip.inc).  EMSPEC_SHARED=1
    (diagnostic build) selects that plan on one GPU: results must not depend on how a stream is cut."""
    import emspec
    monkeypatch.setenv("EMSPEC_SHARED", "1")
    pcm = _pcm(n, hop, frames, S=S)
    eng = emspec.Engine(diag=True)
    try:
        assert eng.fused(n, hop, True)
        out = eng.batch(pcm, n, hop, True, want=("db", "index"))
    finally:
        eng.close()
    odb, _, oidx = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    palette_close(out["index"], oidx)


def test_generic_records_path_still_serves_n16384_small_hop(engine):
    """N = 16384 at hop 256 needs 65 ring slots: not fused, stays on per-bin records + walk/tile scatter."""
    n, hop, frames = 16384, 256, 80
    assert not engine.fused(n, hop, True)
    pcm = _pcm(n, hop, frames, S=1)
    out = engine.batch(pcm, n, hop, True, want=("db",))
    odb, _, _ = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("db",))
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4


@pytest.mark.parametrize("n,hop,seglen", [(4096, 512, 66), (4096, 512, 250), (4096, 1024, 64), (4096, 1024, 130),
                                          (8192, 512, 65), (8192, 512, 131), (8192, 1024, 64), (8192, 1024, 99),
                                          (2048, 128, 67), (2048, 256, 130), (2048, 777, 64), (1024, 256, 65), (1024, 128, 101),
                                          (1024, 333, 64), (4096, 384, 70), (4096, 3000, 64)])
def test_fused_other_shapes_short_segments(n, hop, seglen, monkeypatch):
    """The other builds of the fused kernels with many short segments per stream (segment boundaries
    recompute a 2D-frame halo; odd segment lengths; smaller rings)."""
    import emspec
    monkeypatch.setenv("EMSPEC_SEGLEN", str(seglen))
    frames, S = (400 if n == 8192 else 700), 2
    pcm = _pcm(n, hop, frames, S=S)
    eng = emspec.Engine(diag=True)      # EMSPEC_SEGLEN is a switch of the diagnostic build (same kernel sources)
    try:
        assert eng.fused(n, hop, True)
        out = eng.batch(pcm, n, hop, True, want=("db", "index"))
    finally:
        eng.close()
    odb, _, oidx = O.batch_f32(O.make_cfg(n, hop, True), pcm, want=("db", "index"))
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    palette_close(out["index"], oidx)


def test_hinted_row_lookup_equals_binary_search(diag_engine):
    """The fused kernels find the log-frequency row with a log2 hint + exact table compares.
    Sweep every table edge with its float32 neighbours (+-1, +-2 ulp), row midpoints, values
    just outside the table, zeros, negatives, inf and NaN: must equal the binary search."""
    import ctypes as C
    import emspec
    engine = diag_engine
    lib = emspec.load(diag=True)
    f = lib.emspec_debug_row_lookup
    f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    for n in (1024, 4096, 16384):
        _, eb = engine.tables(n)
        vals = [eb]
        for k in (1, 2):
            up, dn = eb.copy(), eb.copy()
            for _ in range(k):
                up = np.nextafter(up, np.float32(np.inf)); dn = np.nextafter(dn, np.float32(-np.inf))
            vals += [up, dn]
        vals.append((0.5 * (eb[:-1].astype(np.float64) + eb[1:])).astype(np.float32))
        rng = np.random.default_rng(n)
        vals.append(np.exp(rng.uniform(np.log(eb[0] * 0.5), np.log(eb[-1] * 2.0), 200000)).astype(np.float32))
        vals.append(np.array([0.0, -0.0, -1.0, 1e-30, 1e30, np.inf, -np.inf, np.nan], np.float32))
        kh = np.ascontiguousarray(np.concatenate(vals), np.float32)
        a = np.empty(kh.size, np.int32); b = np.empty(kh.size, np.int32)
        assert f(engine._h, n, kh.ctypes.data, kh.size, a.ctypes.data, b.ctypes.data) == 0
        assert np.array_equal(a, b), f"n={n}: {np.sum(a != b)} mismatches"
        ref = np.searchsorted(eb, kh, side="right") - 1
        ok = (kh >= eb[0]) & (kh < eb[-1])
        assert np.array_equal(b[ok], ref[ok]) and np.all(b[~ok] == -1)


def test_error_paths(engine):
    import emspec
    with pytest.raises(emspec.EmspecError) as ei:
        engine.batch(np.zeros((1, 5000), np.float32), 3000, 256)       # unsupported fft size
    assert ei.value.code == emspec.ERR_INVALID_ARG
    with pytest.raises(emspec.EmspecError):
        engine.batch(np.zeros((1, 100), np.float32), 4096, 256)        # shorter than one frame
    engine.reset()
    engine.column(np.zeros(1024, np.float32), 256, True)
    with pytest.raises(emspec.EmspecError) as ei:
        engine.column(np.zeros(4096, np.float32), 256, True)           # shape change mid-stream
    assert ei.value.code == emspec.ERR_STATE
    engine.reset()


def test_full_size_properties(engine):
    """BASELINE config 2 at full size (1 stream x 2^22 samples -> 16,369 columns), checked through
    size-independent properties: (i) linearity of the energy histogram in the input power
    (x -> 2x adds 20 log10 2 dB to every non-empty cell), (ii) time-shift covariance (delaying the
    input by k hops shifts the columns by k), (iii) the column total tracks the frame energy."""
    import torch
    n, hop = 4096, 256
    L = 1 << 22
    pcm = synth.stream(7, L)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(pcm)[None].to(dev)
    Cn = (L - n) // hop + 1
    db1 = torch.empty((1, Cn, 1024), dtype=torch.float32, device=dev)
    db2 = torch.empty_like(db1)
    engine.batch_device(x, n, hop, True, db=db1)
    engine.batch_device((0.5 * x).contiguous(), n, hop, True, db=db2)
    torch.cuda.synchronize()
    assert Cn == 16369
    live = db1 > -100      # well above the power floor (-142 dB), whose gate is not scale-invariant
    d = (db1 - db2)[live]
    assert torch.all(torch.abs(d - 20 * np.log10(2.0)) < 2e-3)        # scaling by 1/2 is exact in float32
    k = 5
    xs = torch.zeros_like(x)
    xs[0, k * hop:] = x[0, :L - k * hop]
    db3 = torch.empty_like(db1)
    engine.batch_device(xs, n, hop, True, db=db3)
    torch.cuda.synchronize()
    a = db1[0, 40:Cn - 40 - k]
    b = db3[0, 40 + k:Cn - 40]
    assert torch.max(torch.abs(a - b)) < 1e-3
    # energy conservation: sum over rows of a column's linear energy ~ Parseval of its frames (loose, in dB)
    e_cols = torch.pow(10.0, db1[0] / 10).sum(dim=1).cpu().numpy()
    assert np.isfinite(e_cols).all() and e_cols.max() < 10.0 and np.median(e_cols) > 1e-6


def test_node_addon_on_gpu():
    """The renderer-side call through the N-API addon, when node exists on the box."""
    import os
    import shutil
    import subprocess
    js = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "em-spec_amd", "js")
    if shutil.which("node") is None or not os.path.exists(os.path.join(js, "emspec.node")):
        pytest.skip("node or the built addon is not available on this box")
    r = subprocess.run(["node", "test_emspec.js"], cwd=js, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "node addon ok" in r.stdout


@pytest.mark.parametrize("rows,n,hop,reassign", [(512, 4096, 256, True), (256, 4096, 256, True), (64, 4096, 256, False),
                                                  (512, 1024, 256, True), (2048, 2048, 512, True), (1024, 8192, 1000, True),
                                                  (1024, 1024, 1024, True), (1024, 256, 1, True)])
def test_other_grids_and_hops(rows, n, hop, reassign):
    """Row counts other than 1024 (fused ring and tile sizes change), a non-power-of-two hop,
    hop = N (no overlap), hop = 1, and a non-default frequency range / display mapping."""
    import emspec
    frames = 37 if hop >= 64 else 300
    pcm = synth.streams(2, n + hop * (frames - 1) + min(3, hop - 1))
    kw = dict(rows=rows, fmin_hz=35.0, fmax_hz=18000.0, gain=3.5, db_range=58.0, gate_db=-65.0)
    with emspec.Engine(**kw) as e:
        out = e.batch(pcm, n, hop, reassign, want=("db", "index"))
        pw, col, row = e.parity_dump(pcm, n, hop, reassign, 0, min(frames, 6))
    cfg = O.make_cfg(n, hop, reassign, **kw)
    odb, _, oidx = O.batch_f32(cfg, pcm, want=("db", "index"))
    assert out["db"].shape == odb.shape == (2, frames, rows)
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    palette_close(out["index"], oidx)
    for s in range(2):
        opw, ocol, orow = O.frames_f32(cfg, pcm[s], 0, min(frames, 6))
        assert np.array_equal(col[s], ocol) and np.array_equal(row[s], orow) and np.array_equal(pw[s], opw)


def test_custom_colormap_and_silence(engine):
    lut = np.zeros((256, 4), np.uint8)
    lut[:, 1] = np.arange(256)
    lut[:, 3] = 255
    engine.set_colormap(lut)
    try:
        n, hop = 4096, 256
        pcm = synth.streams(1, n + hop * 20)
        out = engine.batch(pcm, n, hop, True, want=("rgba", "index"))
        assert np.array_equal(out["rgba"], lut[out["index"]])
        z = engine.batch(np.zeros((1, n + hop * 5), np.float32), n, hop, True, want=("db", "index"))
        assert np.all(z["index"] == 0) and np.allclose(z["db"], -200.0, atol=1e-3)      # silence: every bin gated
    finally:
        engine.set_colormap(O.default_lut())


def test_full_batch_spot_checks_against_oracle(engine):
    """BASELINE config 3 at full size (64 streams x 2^22 samples, 1,047,616 columns): random
    (stream, column) cells of the full run are checked against the oracle evaluated on the local
    slice of audio that can reach them (column c depends on frames c-8 .. c+8 only), including
    segment boundaries of the fused kernel and the first/last columns of a stream."""
    import torch
    n, hop, D = 4096, 256, 8
    S, L = 64, 1 << 22
    base = synth.streams(4, L)
    rng = np.random.default_rng(99)
    pcm = np.stack([np.roll(base[s % 4], 1237 * s) * (0.5 + 0.5 * ((s * 7) % 5) / 4) for s in range(S)]).astype(np.float32)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(pcm).to(dev)
    Cn = (L - n) // hop + 1
    db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
    idx = torch.empty((S, Cn, 1024), dtype=torch.uint8, device=dev)
    engine.batch_device(x, n, hop, True, db=db, index=idx)
    torch.cuda.synchronize()
    cfg = O.make_cfg(n, hop, True)
    cols = [0, 1, 7, 8, 9, Cn - 1, Cn - 2, Cn - 9, 1023, 1024, 1025, 511, 512] + list(rng.integers(20, Cn - 20, 12))
    worst, bad_cells = 0.0, 0
    for c in cols:
        s = int(rng.integers(0, S))
        f0 = max(0, c - D)
        f1 = min(Cn - 1, c + D)
        seg = pcm[s, f0 * hop:f1 * hop + n]
        odb, _, oidx = O.batch_f32(cfg, seg[None], want=("db", "index"), threads=1)
        got = db[s, c].cpu().numpy()
        ref = odb[0, c - f0]
        worst = max(worst, float(np.max(np.abs(got - ref))))
        d = np.abs(idx[s, c].cpu().numpy().astype(int) - oidx[0, c - f0].astype(int))
        assert d.max() <= 1
        bad_cells += int(np.count_nonzero(d))
    print(f"MEASURED palette +-1 cells in the {len(cols)} spot-checked columns: {bad_cells} of {len(cols) * 1024}")
    assert bad_cells <= max(8, len(cols) * 1024 // 1000)
    assert worst < 8.7e-4, worst
    # identical streams give identical columns (no cross-stream leakage in the batch)
    xb = x.clone()
    xb[5] = x[9]
    db2 = torch.empty((S, 256, 1024), dtype=torch.float32, device=dev)
    engine.batch_device(xb[:, :n + hop * 255].contiguous(), n, hop, True, db=db2)
    torch.cuda.synchronize()
    # (float sums are order-dependent in the last bits, so equal streams agree to ~1e-5 dB, not bitwise)
    assert float(torch.max(torch.abs(db2[5] - db2[9]))) < 2e-4


def test_full_batch_n16384_spot_checks_against_oracle(engine):
    """BASELINE configs[4] at full size (64 streams x 2^22 samples, FFT 16384, hop 512: 522,304 columns, one launch of
    fused16384_kernel: 256 workgroups of 2,041 columns with the ring parked in registers): random (stream, column) cells,
    the ends of a stream and both sides of the segment boundaries against the oracle on the slice of audio that can
    reach them (column c depends on frames c-16 .. c+16)."""
    import torch
    n, hop, D = 16384, 512, 16
    S, L = 64, 1 << 22
    base = synth.streams(4, L)
    rng = np.random.default_rng(1616)
    pcm = np.stack([np.roll(base[s % 4], 2311 * s) * (0.5 + 0.5 * ((s * 3) % 5) / 4) for s in range(S)]).astype(np.float32)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(pcm).to(dev)
    Cn = (L - n) // hop + 1
    assert Cn == 8161 and engine.fused(n, hop, True)
    db = torch.empty((S, Cn, 1024), dtype=torch.float32, device=dev)
    idx = torch.empty((S, Cn, 1024), dtype=torch.uint8, device=dev)
    engine.batch_device(x, n, hop, True, db=db, index=idx)
    torch.cuda.synchronize()
    cfg = O.make_cfg(n, hop, True)
    seg = -(-Cn // 4)       # 64 streams on 256 CUs: four segments per stream
    cols = [0, 1, 15, 16, 17, Cn - 1, Cn - 2, Cn - 17] + [k * seg + d for k in (1, 2, 3) for d in (-17, -1, 0, 1, 16)] + \
        list(rng.integers(40, Cn - 40, 8))
    worst, bad_cells = 0.0, 0
    for c in cols:
        s = int(rng.integers(0, S))
        f0, f1 = max(0, c - D), min(Cn - 1, c + D)
        odb, _, oidx = O.batch_f32(cfg, pcm[s, f0 * hop:f1 * hop + n][None], want=("db", "index"), threads=1)
        worst = max(worst, float(np.max(np.abs(db[s, c].cpu().numpy() - odb[0, c - f0]))))
        d = np.abs(idx[s, c].cpu().numpy().astype(int) - oidx[0, c - f0].astype(int))
        assert d.max() <= 1, (s, c)
        bad_cells += int(np.count_nonzero(d))
    print(f"MEASURED N=16384 full size: max dB error {worst:.2e}, palette +-1 cells {bad_cells} of {len(cols) * 1024}")
    assert worst < 8.7e-4, worst
    assert bad_cells <= max(8, len(cols) * 1024 // 1000)
    engine.device_status()


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a plain C ABI: build a C program against include/emspec.h + libemspec.so and run it."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_driver")
    lib = os.path.join(root, "em-spec_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O1", os.path.join(root, "tests", "cdriver", "abi_driver.c"),
                           "-I", os.path.join(root, "include"), "-L", lib, "-lemspec", "-lm",
                           "-Wl,-rpath," + lib, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_driver ok" in r.stdout


@pytest.mark.parametrize("n,hop", [(4096, 256), (1024, 256)])
def test_custom_row_edges(n, hop):
    """An arbitrary monotone frequency axis (the hook for 'Low-End Boost' / 'Frequency Scale'):
    the kernels switch to the binary-search row lookup; results must still match the oracle exactly."""
    import emspec
    rows = 512
    u = np.arange(rows + 1) / rows
    edges = (30.0 * np.exp(np.log(16000.0 / 30.0) * u ** 1.9)).astype(np.float32)       # warped (low end boosted)
    edges[200:260] = np.linspace(edges[200], edges[260], 61)[:60].astype(np.float32)   # a linear stretch: not log-spaced
    frames = 30
    pcm = synth.streams(2, n + hop * (frames - 1))
    with emspec.Engine(rows=rows) as e:
        e.set_row_edges_hz(edges)
        assert np.array_equal(e.row_edges_hz(), edges)
        out = e.batch(pcm, n, hop, True, want=("db",))
        pw, col, row = e.parity_dump(pcm, n, hop, True, 0, 6)
        with pytest.raises(emspec.EmspecError):
            e.set_row_edges_hz(edges[::-1].copy())
        e.set_row_edges_hz(None)
        back = e.batch(pcm, n, hop, True, want=("db",))
    O.set_custom_edges_hz(edges)
    try:
        cfg = O.make_cfg(n, hop, True, rows=rows)
        odb, _, _ = O.batch_f32(cfg, pcm, want=("db",))
        for s in range(2):
            opw, ocol, orow = O.frames_f32(cfg, pcm[s], 0, 6)
            assert np.array_equal(row[s], orow) and np.array_equal(col[s], ocol)
    finally:
        O.set_custom_edges_hz(None)
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    odb_log, _, _ = O.batch_f32(O.make_cfg(n, hop, True, rows=rows), pcm, want=("db",))
    assert np.max(np.abs(back["db"] - odb_log)) < 8.7e-4


def test_no_spin_timeouts(diag_engine):
    """The decoupled-team fused kernel (diagnostic build only) bounds every wait; a timeout raises a device flag."""
    import ctypes as C
    import emspec
    engine = diag_engine
    lib = emspec.load(diag=True)
    lib.emspec_debug_fused_error.argtypes = [C.c_void_p]
    assert lib.emspec_debug_fused_error(engine._h) == 0


_VARIANT_CHILD = r"""
import os, sys
import numpy as np
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "em-spec_amd")]
import emspec
from emspec import synth
pcm = synth.streams(3, 4096 + 256 * 1499)
with emspec.Engine(diag=%(diag)r) as e:
    out = e.batch(pcm, 4096, 256, True, want=("db", "index"))
    if %(diag)r:
        import ctypes as C
        lib = emspec.load(diag=True)
        lib.emspec_debug_fused_error.argtypes = [C.c_void_p]
        assert lib.emspec_debug_fused_error(e._h) == 0, "a bounded spin timed out"
np.savez(sys.argv[1], db=out["db"], index=out["index"])
"""


@pytest.mark.parametrize("variant", ["r8", "pp3", "ppt", "r8t", "r16"])
def test_fused_ab_variants_match_the_product(variant, tmp_path):
    """The A/B variants of the N = 4096 kernel kept in libemspec_diag.so (lock step `r8` = the default of rounds 1-2,
    `pp3` = the product of rounds 3-4 with the spectrum reads at the start of role 1, its software-team-barrier form `ppt`, decoupled teams `r8t`, 512-thread radix-16 `r16`) compute the same
    columns as the product kernel: same arithmetic per bin, only the order of the float32 histogram sums differs, so dB
    agrees to a few ulp and the palette index may differ by one step on a handful of cells.  One child process per
    variant (the switch is read once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, diag, env in (("product", False, {}), (variant, True, {"EMSPEC_FUSED_VARIANT": variant})):
        f = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, "-c", _VARIANT_CHILD % dict(root=root, diag=diag), f],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(f)
    a, b = outs["product"], outs[variant]
    live = a["db"] > -100
    assert np.max(np.abs(a["db"][live] - b["db"][live])) < 1e-3
    diff = a["index"].astype(np.int16) - b["index"].astype(np.int16)
    assert np.max(np.abs(diff)) <= 1
    assert np.count_nonzero(diff) <= 1e-3 * diff.size
    print(f"MEASURED variant {variant}: max |dB difference| {np.max(np.abs(a['db'][live] - b['db'][live])):.2e}, "
          f"{np.count_nonzero(diff)} of {diff.size} palette indices differ by one step")


@pytest.mark.parametrize("smoothing,agc", [(0.6, 0.0), (0.0, 1.0), (0.85, 0.7)])
def test_display_postprocess(smoothing, agc):
    """Temporal smoothing + adaptive brightness over finished columns: batch (chunked IIR with
    warm-up, > 1 chunk) and the streaming call must both match the sequential numpy restatement."""
    import emspec
    n, hop = 1024, 256                 # generic path, cheap oracle; C spans several 1024-column chunks
    frames = 2600
    rng = np.random.default_rng(3)
    pcm = synth.streams(2, n + hop * (frames - 1))
    pcm *= (0.05 + 0.95 * (np.sin(np.arange(pcm.shape[1]) / 40000.0) ** 2)).astype(np.float32)   # loudness swells: AGC has work
    cfg = O.make_cfg(n, hop, True)
    raw, _, _ = O.batch_f32(cfg, pcm, want=("db",))
    want_db, want_idx, want_rgba = O.postprocess(raw, smoothing, agc, cfg)
    with emspec.Engine() as e:
        e.set_display(smoothing, agc)
        out = e.batch(pcm, n, hop, True, want=("db", "index", "rgba"))
        assert np.max(np.abs(out["db"] - want_db)) < 2e-3
        palette_close(out["index"], want_idx)
        assert np.array_equal(out["rgba"], O.default_lut()[out["index"]])
        only_idx = e.batch(pcm[:1], n, hop, True, want=("index",))["index"]       # no dB buffer from the caller
        assert np.array_equal(only_idx, out["index"][:1])
        # streaming: same law, state carried in the engine
        D = emspec.latency_columns(n, hop, True)
        got = []
        for j in range(300):
            db, c = e.column(pcm[0, j * hop:j * hop + n], hop, True)
            if c >= 0:
                got.append(db)
        got = np.stack(got)
        assert np.max(np.abs(got - want_db[0, :got.shape[0]])) < 2e-3
        e.reset()
        e.set_display(0.0, 0.0)
        plain = e.batch(pcm[:1, :n + hop * 50], n, hop, True, want=("db",))["db"]
        assert np.max(np.abs(plain[:, :40] - raw[:1, :40])) < 8.7e-4      # (later columns lack their future frames)
        with pytest.raises(emspec.EmspecError):
            e.set_display(0.99, 0.0)


@pytest.mark.parametrize("kind", ["sine", "impulses", "dc_plus_nyquist", "loud"])
def test_heavy_cell_contention(engine, kind):
    """Signals that pile many bins into few histogram cells (the exchange-based LDS accumulate is
    exercised at its worst: a stationary sine sends its whole main lobe and skirts of up to 17 frames
    into ONE cell; an impulse train puts all 2049 bins of 16 frames into one column)."""
    n, hop, frames = 4096, 256, 80
    L = n + hop * (frames - 1)
    t = np.arange(L)
    if kind == "sine":
        x = 0.9 * np.sin(2 * np.pi * 3000.0 * t / 48000.0)
    elif kind == "impulses":
        x = np.zeros(L)
        x[1000::4096] = 1.0
    elif kind == "dc_plus_nyquist":
        x = 0.5 + 0.4 * np.cos(np.pi * t)          # energy exactly at k = 0 and k = N/2: both are dropped by the axis
    else:
        x = 30.0 * np.sin(2 * np.pi * 440.0 * t / 48000.0) + 5.0 * np.sign(np.sin(2 * np.pi * 97.0 * t / 48000.0))   # far outside [-1,1]
    pcm = np.stack([x, 0.5 * x]).astype(np.float32)
    out = engine.batch(pcm, n, hop, True, want=("db", "index"))
    cfg = O.make_cfg(n, hop, True)
    odb, _, oidx = O.batch_f32(cfg, pcm, want=("db", "index"))
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    assert np.abs(out["index"].astype(int) - oidx.astype(int)).max() <= 1
    if kind == "sine":      # physical check: the peak cell carries the sine's energy, 20 log10(0.9) dB
        assert abs(out["db"][0, 40].max() - 20 * np.log10(0.9)) < 0.05


@pytest.mark.parametrize("rows", [2048, 4096])
def test_generic_path_at_4096_with_many_rows(rows):
    """N = 4096 / hop 256 with more rows than the fused ring holds: the records + tile-scatter path."""
    import emspec
    n, hop, frames = 4096, 256, 45
    pcm = synth.streams(1, n + hop * (frames - 1))
    with emspec.Engine(rows=rows) as e:
        assert not e.fused(n, hop, True)
        out = e.batch(pcm, n, hop, True, want=("db",))
    odb, _, _ = O.batch_f32(O.make_cfg(n, hop, True, rows=rows), pcm, want=("db",))
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4


# measured r03 (landing / row / column): 4096: 0 / 0 / 4.09e-4 of 31,757 bins; 16384: 0 / 7.86e-6 / 2.28e-4 of 127,261: x 2, a zero gets room for one bin
HIPFFT_BOUNDS = {4096: (3.2e-5, 3.2e-5, 8.2e-4), 16384: (7.9e-6, 1.6e-5, 4.6e-4)}


@pytest.mark.parametrize("n,hop", [(4096, 256), (16384, 512)])
def test_dump_vs_hipfft_three_window(engine, n, hop):
    """an on-device cross-check that shares nothing with the hand-written
    FFT — three explicitly windowed float64 rfft's through torch.fft (hipFFT/rocFFT on ROCm)."""
    import torch
    frames = 16
    pcm = _pcm(n, hop, frames, S=1)
    pw, col, row = engine.parity_dump(pcm, n, hop, True, 0, frames)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(pcm[0]).to(dev, torch.float64).unfold(0, n, hop)[:frames]
    i = torch.arange(n, device=dev, dtype=torch.float64)
    h = 0.5 - 0.5 * torch.cos(2 * np.pi * i / n)
    Xh = torch.fft.rfft(x * h)
    Xt = torch.fft.rfft(x * ((i - n // 2) * h))
    Xd = torch.fft.rfft(x * ((np.pi / n) * torch.sin(2 * np.pi * i / n)))
    P = (Xh.real ** 2 + Xh.imag ** 2)
    Ps = torch.where(P > 0, P, torch.ones_like(P))
    ts = (Xt * Xh.conj()).real / Ps
    ks = -(n / (2 * np.pi)) * (Xd * Xh.conj()).imag / Ps
    P, ts, ks = P.cpu().numpy(), ts.cpu().numpy(), ks.cpu().numpy()
    strong = P >= P.max(axis=1, keepdims=True) * 1e-6
    rel = np.abs(pw[0] - P) / np.maximum(P, 1e-300)
    assert rel[strong].max() < 1e-4, rel[strong].max()
    cfg = O.make_cfg(n, hop, True)
    _, eb = O.tables(cfg)
    D = -(-n // (2 * hop))
    cf = np.floor(ts / hop + 0.5)
    khat = np.arange(n // 2 + 1)[None, :] + ks
    valid = (P >= cfg.power_floor * (n / 4.0) ** 2) & (np.abs(cf) <= D) & (khat >= eb[0]) & (khat < eb[-1])
    hrow = np.searchsorted(eb.astype(np.float64), khat, side="right") - 1
    hcol = np.arange(frames)[:, None] + cf
    # both sides must agree on which bins land at all, up to rare edge cases
    both = valid & (row[0] >= 0)
    lr, rr, cr = float(np.mean((row[0] >= 0) != valid)), float(np.mean(row[0][both] != hrow[both])), float(np.mean(col[0][both] != hcol[both]))
    print(f"MEASURED dump_vs_hipfft N={n}: landing mismatch {lr:.3e}, row {rr:.3e}, col {cr:.3e} of {int(both.sum())} bins")
    lb, rb, cb = HIPFFT_BOUNDS[n]      # 2 x measured
    assert lr <= lb and rr <= rb and cr <= cb, (lr, rr, cr)


@pytest.mark.parametrize("n,hop,reassign", [(4096, 256, True), (1024, 256, False), (2048, 2048, True), (16384, 512, True),
                                            (512, 1, True)])
def test_sample_block_streaming_equals_batch(engine, n, hop, reassign):
    """blocks of arbitrary length through emspec_push_samples (device sample
    ring + pending-column ring, up to 64 frames per launch) give the batch columns."""
    frames = 300 if hop == 1 else 150 if n <= 4096 else 70
    rng = np.random.default_rng(n + hop)
    L = n + hop * (frames - 1) + min(hop - 1, 77)
    pcm = synth.streams(1, L)[0]
    out = engine.batch(pcm[None], n, hop, reassign, want=("db", "rgba"))
    ref, ref_rgba = out["db"][0], out["rgba"][0]
    engine.reset()
    import emspec
    D = emspec.latency_columns(n, hop, reassign)
    cols, rg, pos, nxt = [], [], 0, 0
    while pos < L:
        ln = int(min(rng.choice([1, 3, hop, hop + 1, n, 3 * n + 5, 70 * hop]), L - pos))
        db, rgba, first = engine.push_samples(pcm[pos:pos + ln], n, hop, reassign, want_rgba=True)
        pos += ln
        if len(db):
            assert first == nxt
            cols.append(db); rg.append(rgba); nxt += len(db)
        else:
            assert first == -1
    assert nxt == frames - D
    for _ in range(D):
        db, rgba, c = engine.flush(want_rgba=True)
        assert c == nxt
        cols.append(db[None]); rg.append(rgba[None]); nxt += 1
    with pytest.raises(Exception):
        engine.flush()
    with pytest.raises(Exception):   # one stream, one feeding mode
        engine.column(pcm[:n], hop, reassign)
    engine.reset()
    got = np.concatenate(cols)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) < 8.7e-4
    # colours can differ by one palette step where the dB value sits on a step edge
    assert np.mean(np.concatenate(rg) != ref_rgba) < 1e-3


def test_sample_block_streaming_rejects_small_output(engine):
    import ctypes as C
    import emspec
    lib = emspec.load()
    engine.reset()
    x = np.zeros(4096 + 256 * 40, np.float32)
    need = lib.emspec_push_columns(engine._h, x.size, 4096, 256, 1)
    assert need == 41 - 8
    db = np.empty((need - 1, engine.rows), np.float32)
    cnt, first = C.c_int64(), C.c_int64()
    rc = lib.emspec_push_samples(engine._h, C.c_void_p(x.ctypes.data), x.size, 4096, 256, 1, C.c_void_p(db.ctypes.data),
                                 None, engine.rows, need - 1, C.byref(cnt), C.byref(first))
    assert rc == emspec.ERR_INVALID_ARG
    # nothing was consumed: the same block still completes `need` columns
    assert lib.emspec_push_columns(engine._h, x.size, 4096, 256, 1) == need
    assert lib.emspec_push_columns(engine._h, 10, 3000, 256, 1) == -1
    engine.reset()


def test_nan_and_inf_samples_are_contained(engine):
    """A NaN or Inf sample poisons every bin of the frames that contain it (their power is not a finite
    number >= the floor, so those bins are dropped) and nothing else: outputs stay finite, frames that do not
    overlap the bad sample are unaffected, and the oracle agrees."""
    n, hop, frames = 4096, 256, 60
    pcm = _pcm(n, hop, frames, S=2)
    clean = engine.batch(pcm, n, hop, True, want=("db", "index"))
    bad = pcm.copy()
    bad[0, 7000] = np.nan
    bad[1, 9000] = np.inf
    out = engine.batch(bad, n, hop, True, want=("db", "index"))
    assert np.isfinite(out["db"]).all()
    cfg = O.make_cfg(n, hop, True)
    odb, _, oidx = O.batch_f32(cfg, bad, want=("db", "index"))
    assert np.max(np.abs(out["db"] - odb)) < 8.7e-4
    assert np.mean(out["index"] != oidx) < 1e-3
    for s, pos in ((0, 7000), (1, 9000)):
        touched = [j for j in range(frames) if j * hop <= pos < j * hop + n]
        # columns further than D from every poisoned frame equal the clean run (up to the order of the float sums)
        far = [c for c in range(frames) if all(abs(c - j) > 8 for j in touched)]
        assert far, "test needs some unaffected columns"
        assert np.max(np.abs(out["db"][s, far] - clean["db"][s, far])) < 8.7e-4
        # a column whose own frame and all neighbours within D are poisoned is empty
        dead = [c for c in range(frames) if all((c + d) in touched for d in range(-8, 9) if 0 <= c + d < frames)]
        if dead:
            assert np.all(out["index"][s, dead] == 0)


def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 control flow (stream sharding, chunked + double-buffered gather, max-over-ranks timing)
    rehearsed with two ranks on this one GPU: gloo carries the gather through host tensors, everything else is
    the code the 8-GPU run executes."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--streams", "6", "--log2-samples", "18", "--chunks", "3", "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # rank 0 prints ONE json line
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["steps"] == 3
    C = (2 ** 18 - 4096) // 256 + 1
    assert b["config"]["columns_per_step"] == 2 * 6 * C
    assert b["value"] > 0 and b["cpu_baseline"] is None


def test_bench_plain_invocation_starts_two_ranks():
    """The driver's own command form, `python bench.py --gpus 2 ...` with no launcher around it: bench.py starts its two
    ranks as a child `python -m torch.distributed.run` (the parent never touches the GPU), relays rank 0's one JSON line and
    exits 0.  Two ranks on this one GPU over gloo; at N = 8 the same code path runs over RCCL (rccl.h:220)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--streams", "6", "--log2-samples", "18", "--chunks", "3", "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]     # ONLY the line reaches stdout
    b = json.loads(lines[0])
    C = (2 ** 18 - 4096) // 256 + 1
    assert b["n_gpus"] == 2 and b["config"]["columns_per_step"] == 2 * 6 * C and b["value"] > 0
    # over RCCL the same invocation needs two devices: a JSON error line and a non-zero exit on this one-GPU box
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd[:-2], capture_output=True, text=True, timeout=120, env=env, cwd=root)
        assert r.returncode != 0 and "error" in json.loads(r.stdout.splitlines()[0])


def test_bench_two_rank_uneven_split_rehearsal():
    """The N > 1 stream split (the collecting rank takes a lighter shard; bench.py tries five splits and keeps the fastest)
    rehearsed with two ranks on this one GPU over gloo: the trial loop, the re-built buffers and the uneven point-to-point
    gather are the code the 8-GPU run executes; only the transport differs (libemspec's RCCL gather needs one GPU per rank)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
            "--streams", "32", "--log2-samples", "17", "--chunks", "2", "--backend", "gloo"]
    C = (2 ** 17 - 4096) // 256 + 1
    for extra, check in (([], "trials"), (["--root-streams", "20"], "pinned")):
        r = subprocess.run(base + extra, capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        b = json.loads(lines[0])
        counts = b["config"]["streams_per_rank"]
        assert sum(counts) == 64 and b["config"]["columns_per_step"] == 64 * C and b["value"] > 0
        if check == "trials":
            trials = b["gather"]["split_trials"]
            assert len(trials) == 5 and all(sum(t["streams_per_rank"]) == 64 and t["columns_per_s"] > 0 for t in trials)
            assert len(b["gather"]["kernel_ms_per_rank"]) == 2 and min(b["gather"]["kernel_ms_per_rank"]) > 0
            assert counts in [t["streams_per_rank"] for t in trials]
            assert counts == max(trials, key=lambda t: t["columns_per_s"])["streams_per_rank"]
        else:
            assert counts == [20, 44] and b["gather"]["split_trials"] is None
    # a spent trial budget: the remaining splits are skipped on every rank together and the modelled split is used
    r = subprocess.run(base + ["--trial-budget-s", "0"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    b = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert b["gather"]["split_trials"][-1]["streams_per_rank"] is None and sum(b["config"]["streams_per_rank"]) == 64
    assert b["config"]["streams_per_rank"][0] < 32 and b["value"] > 0


def test_bench_one_rank_with_process_group_rehearsal():
    """bench.py --gather dist-loopback: ONE rank, but everything the N > 1 run does around the gather is alive -
    torch.distributed's NCCL process group (the id broadcast, barriers and reductions go through it), libemspec's own
    RCCL communicator beside it in the same process, the chunked pipeline with the host synchronisation inside the
    gather - and the rank's own columns take the wire (pack, self send/recv, expand)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29539")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gather", "dist-loopback", "--no-cpu-baseline", "--no-configs",
           "--steps", "3", "--warmup", "1", "--streams", "8", "--log2-samples", "19"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 1 and b["gather"]["path"] == "lib" and b["gather"]["note"] is None
    assert 32 < b["gather"]["wire_bytes_per_column"] < 768 and b["value"] > 0
    # what the first N > 1 run must show: the rank count RCCL itself reports, every rank's kernel time, the split, and what
    # the root's expand costs
    g = b["gather"]
    assert g["rccl_world"] == 1 and g["streams_per_rank"] == [8] and len(g["kernel_ms_per_rank"]) == 1 and g["kernel_ms_per_rank"][0] > 0
    # (since round 4 the timed form keeps the images packed on the root; the expanding form runs beside it)
    assert g["root_expand_ms_standalone"] is not None and g["root_expand_ms_standalone"] >= 0 and g["root_keeps_images_packed"] is True
    assert g["expanding_columns_per_s"] > 0 and g["packed_columns_per_s"] is None and g["step_deadline_s"] >= 30.0
    # ... and the expanding form as the timed one (the root turns every image back into plain arrays)
    r = subprocess.run(cmd + ["--gather-expand"], capture_output=True, text=True, timeout=300, env=dict(env, MASTER_PORT="29541"), cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    p = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert p["gather"]["root_keeps_images_packed"] is False and 32 < p["gather"]["wire_bytes_per_column"] < 768 and p["value"] > 0
    assert p["gather"]["packed_columns_per_s"] > 0
    # a failing rank ends the job with a non-zero exit
    r = subprocess.run(cmd[:2] + ["--gather", "dist-loopback", "--streams", "2", "--log2-samples", "11", "--no-cpu-baseline", "--no-configs"],
                       capture_output=True, text=True, timeout=300, env=dict(env, MASTER_PORT="29543"), cwd=root)
    assert r.returncode != 0


def test_bench_watchdog_ends_a_job_whose_rank_stops():
    """A rank that stops making progress mid-run (here: rank 1 sleeps at timed step 1) must end the job, not hang it: rank 0
    blocks in the gather, its per-step watchdog thread fires after the deadline and exits 1, the launcher stops the rest.
    Two ranks on this one GPU over gloo (the control flow of the N > 1 run)."""
    import os, subprocess, sys, time as _t
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--streams", "4", "--log2-samples", "15", "--chunks", "2", "--backend", "gloo", "--root-streams", "4",
           "--hang-rank", "1", "--hang-at-step", "1", "--step-deadline-min", "6", "--step-deadline-factor", "1"]
    t0 = _t.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=root)
    took = _t.time() - t0
    assert r.returncode != 0, "a stuck rank must end the job with a non-zero exit"
    assert "watchdog" in r.stderr and took < 200, (took, r.stderr[-1500:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no bench line may be printed for a failed job"


def test_bench_device_synth_matches_definition():
    """bench.py generates its input on the device from the same counter-based definition as emspec/synth.py."""
    import os, sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    got = bench.synth_device(2, 50000, 7, torch.device("cuda", 0)).cpu().numpy()
    ref = synth.streams(2, 50000, first=7)
    assert np.max(np.abs(got - ref)) < 1e-6


@pytest.mark.gpu
def test_short_reciprocal_equals_ieee_division(diag_engine):
    """The specialised kernels compute 1/den with recip_normal (7 instructions: v_rcp_f32 + six fma, emspec_device.h)
    instead of hipcc's 11-instruction IEEE division.  Both are evaluated on the GPU for EVERY float of three binades
    (2^23 mantissas each: scaling by a power of two is exact, so a binade stands for all of them away from the range's
    ends), for the binades at both ends of the range the kernels guarantee (64 P between 2^-90 and 2^126: power floor
    >= 1e-27 checked by the launcher, upper power gate 1e36), and for random values across it: bit-identical."""
    import ctypes as C
    import emspec
    lib = emspec.load(diag=True)
    lib.emspec_debug_recip.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]

    def check(vals):
        vals = np.ascontiguousarray(vals, np.float32)
        a, b = np.empty_like(vals), np.empty_like(vals)
        assert lib.emspec_debug_recip(diag_engine._h, vals.ctypes.data, vals.size, a.ctypes.data, b.ctypes.data) == 0
        with np.errstate(all="ignore"):
            ref = (np.float32(1.0) / vals).astype(np.float32)
        assert np.array_equal(b.view(np.uint32), ref.view(np.uint32))          # the GPU's division is IEEE
        bad = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32))
        assert bad.size == 0, (vals[bad[:5]], a[bad[:5]], b[bad[:5]])

    mant = np.arange(1 << 23, dtype=np.uint32)
    for exp in (127, 127 + 20, 127 - 37, 127 - 90, 127 + 125):                   # biased exponents: 1, 2^20, 2^-37, 2^-90, 2^125
        check(((np.uint32(exp) << np.uint32(23)) | mant).view(np.float32))
    rng = np.random.default_rng(5)
    bits = (rng.integers(127 - 90, 127 + 126, size=1 << 22).astype(np.uint32) << np.uint32(23)) | rng.integers(0, 1 << 23, size=1 << 22).astype(np.uint32)
    check(bits.view(np.float32))
    # the largest argument the power gate lets through: 64 * 1e36
    check(np.array([64.0 * 1.0e36, np.nextafter(np.float32(64.0e36), np.float32(0))], np.float32))


def test_bench_four_rank_rehearsal():
    """`python bench.py --gpus 4` (the driver's command form at N = 4) with four ranks on this one GPU over gloo: the plain
    invocation starts its ranks, the split trials run for world 4 (the root lighter, the other three even), the gather collects
    four shards per chunk, ONE line comes back.  What the driver's N = 4 / 8 runs add is the transport (RCCL, rccl.h:220)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--streams", "16",
           "--log2-samples", "17", "--chunks", "2", "--backend", "gloo", "--trial-budget-s", "40"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    b = json.loads(lines[0])
    C = (2 ** 17 - 4096) // 256 + 1
    counts = b["config"]["streams_per_rank"]
    assert b["n_gpus"] == 4 and len(counts) == 4 and sum(counts) == 64 and b["config"]["columns_per_step"] == 64 * C and b["value"] > 0
    trials = [t for t in b["gather"]["split_trials"] if t.get("streams_per_rank")]
    assert len(trials) >= 2 and all(sum(t["streams_per_rank"]) == 64 and t["streams_per_rank"][0] <= 16 for t in trials)
    assert all(max(t["streams_per_rank"][1:]) - min(t["streams_per_rank"][1:]) <= 1 for t in trials)
    assert counts == max(trials, key=lambda t: t["columns_per_s"])["streams_per_rank"]

