"""ctypes loader for the CPU oracle (oracle/libemspec_oracle.so).

TEST INFRASTRUCTURE ONLY (see emspec_oracle.h).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
PARITY UNPINNED: no reference source or fixtures exist (/root/reference/README.md:73).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libemspec_oracle.so")


class EoCfg(C.Structure):
    _fields_ = [("n", C.c_int32), ("hop", C.c_int32), ("rows", C.c_int32), ("reassign", C.c_int32),
                ("sample_rate", C.c_float), ("fmin_hz", C.c_float), ("fmax_hz", C.c_float),
                ("gain", C.c_float), ("db_top", C.c_float), ("db_range", C.c_float),
                ("gate_db", C.c_float), ("power_floor", C.c_float)]


DEFAULTS = dict(rows=1024, sample_rate=48000.0, fmin_hz=20.0, fmax_hz=24000.0, gain=1.0,
                db_top=0.0, db_range=80.0, gate_db=-80.0, power_floor=1e-14)


def make_cfg(n, hop, reassign=True, **kw):
    d = dict(DEFAULTS)
    d.update(kw)
    return EoCfg(n=n, hop=hop, rows=int(d["rows"]), reassign=int(bool(reassign)),
                 sample_rate=d["sample_rate"], fmin_hz=d["fmin_hz"], fmax_hz=d["fmax_hz"], gain=d["gain"],
                 db_top=d["db_top"], db_range=d["db_range"], gate_db=d["gate_db"], power_floor=d["power_floor"])


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("emspec_oracle.c", "emspec_exact.c", "emspec_cpu_fast.c", "emspec_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libemspec_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.eo_max_threads.restype = C.c_int
        _lib.eo_exact_db.restype = C.c_double
        _lib.eo_exact_db.argtypes = [C.c_double]
        _lib.eo_exact_sum32.restype = C.c_float
        _lib.eo_exact_sum32.argtypes = [C.c_int64]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def num_columns(L, n, hop):
    return (L - n) // hop + 1 if L >= n else 0


def tables(cfg):
    tw = np.empty(cfg.n, np.float32)
    eb = np.empty(cfg.rows + 1, np.float32)
    assert lib().eo_tables(C.byref(cfg), _p(tw, C.c_float), _p(eb, C.c_float)) == 0
    return tw, eb


def set_custom_edges_hz(hz):
    if hz is None:
        lib().eo_set_custom_edges_hz(None, 0)
    else:
        hz = np.ascontiguousarray(hz, np.float32)
        lib().eo_set_custom_edges_hz(_p(hz, C.c_float), C.c_int(hz.size))


def default_lut():
    lut = np.empty((256, 4), np.uint8)
    lib().eo_default_lut(_p(lut, C.c_uint8))
    return lut


def frames_f32(cfg, pcm, frame0, nframes):
    pcm = np.ascontiguousarray(pcm, np.float32)
    K = cfg.n // 2 + 1
    pw = np.empty((nframes, K), np.float32)
    col = np.empty((nframes, K), np.int32)
    row = np.empty((nframes, K), np.int32)
    rc = lib().eo_frames_f32(C.byref(cfg), _p(pcm, C.c_float), C.c_int64(pcm.size), C.c_int64(frame0),
                             C.c_int64(nframes), _p(pw, C.c_float), _p(col, C.c_int32), _p(row, C.c_int32))
    assert rc == 0, rc
    return pw, col, row


def frames_f64(cfg, pcm, frame0, nframes):
    pcm = np.ascontiguousarray(pcm, np.float32)
    K = cfg.n // 2 + 1
    pw = np.empty((nframes, K), np.float64)
    that = np.empty((nframes, K), np.float64)
    khat = np.empty((nframes, K), np.float64)
    col = np.empty((nframes, K), np.int32)
    row = np.empty((nframes, K), np.int32)
    rc = lib().eo_frames_f64(C.byref(cfg), _p(pcm, C.c_float), C.c_int64(pcm.size), C.c_int64(frame0),
                             C.c_int64(nframes), _p(pw, C.c_double), _p(that, C.c_double),
                             _p(khat, C.c_double), _p(col, C.c_int32), _p(row, C.c_int32))
    assert rc == 0, rc
    return pw, that, khat, col, row


def hist_f32(cfg, pcm, threads=0):
    pcm = np.ascontiguousarray(pcm, np.float32)
    S, L = pcm.shape
    Cn = num_columns(L, cfg.n, cfg.hop)
    hist = np.empty((S, Cn, cfg.rows), np.float32)
    rc = lib().eo_hist_f32(C.byref(cfg), _p(pcm, C.c_float), C.c_int32(S), C.c_int64(L), _p(hist, C.c_float),
                           C.c_int32(threads))
    assert rc == 0, rc
    return hist


def batch_f32(cfg, pcm, lut=None, want=("db", "rgba", "index"), threads=0):
    pcm = np.ascontiguousarray(pcm, np.float32)
    S, L = pcm.shape
    Cn = num_columns(L, cfg.n, cfg.hop)
    db = np.empty((S, Cn, cfg.rows), np.float32) if "db" in want else None
    rgba = np.empty((S, Cn, cfg.rows, 4), np.uint8) if "rgba" in want else None
    idx = np.empty((S, Cn, cfg.rows), np.uint8) if "index" in want else None
    if lut is not None:
        lut = np.ascontiguousarray(lut, np.uint8)
    rc = lib().eo_batch_f32(C.byref(cfg), _p(pcm, C.c_float), C.c_int32(S), C.c_int64(L), _p(lut, C.c_uint8),
                            _p(db, C.c_float), _p(rgba, C.c_uint8), _p(idx, C.c_uint8), C.c_int32(threads))
    assert rc == 0, rc
    return db, rgba, idx


def edges64(cfg):
    e = np.empty(cfg.rows + 1, np.float64)
    assert lib().eo_edges64(C.byref(cfg), _p(e, C.c_double)) == 0
    return e


def frames_exact(cfg, pcm, frame0, nframes):
    """EXACT-mode bit model, one stream: (power float64, col, row, q int64), each [nframes][n/2+1]."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    K = cfg.n // 2 + 1
    pw = np.empty((nframes, K), np.float64)
    col = np.empty((nframes, K), np.int32)
    row = np.empty((nframes, K), np.int32)
    q = np.empty((nframes, K), np.int64)
    rc = lib().eo_frames_exact(C.byref(cfg), _p(pcm, C.c_float), C.c_int64(pcm.size), C.c_int64(frame0),
                               C.c_int64(nframes), _p(pw, C.c_double), _p(col, C.c_int32), _p(row, C.c_int32),
                               _p(q, C.c_int64))
    assert rc == 0, rc
    return pw, col, row, q


def batch_exact(cfg, pcm, lut=None, want=("db", "rgba", "index"), threads=0):
    """EXACT-mode bit model, whole pipeline: (db, rgba, index, hist int64 or None)."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    if pcm.ndim == 1:
        pcm = pcm[None]
    S, L = pcm.shape
    Cn = num_columns(L, cfg.n, cfg.hop)
    db = np.empty((S, Cn, cfg.rows), np.float32) if "db" in want else None
    rgba = np.empty((S, Cn, cfg.rows, 4), np.uint8) if "rgba" in want else None
    idx = np.empty((S, Cn, cfg.rows), np.uint8) if "index" in want else None
    hist = np.empty((S, Cn, cfg.rows), np.int64) if "hist" in want else None
    if lut is not None:
        lut = np.ascontiguousarray(lut, np.uint8)
    rc = lib().eo_batch_exact(C.byref(cfg), _p(pcm, C.c_float), C.c_int32(S), C.c_int64(L), _p(lut, C.c_uint8),
                              _p(db, C.c_float), _p(rgba, C.c_uint8), _p(idx, C.c_uint8), _p(hist, C.c_int64),
                              C.c_int32(threads))
    assert rc == 0, rc
    return db, rgba, idx, hist


def exact_db(x):
    return float(lib().eo_exact_db(float(x)))


def exact_sum32(v):
    """The int64 cell sum as the dB stage takes it in: binary32 from the two 32-bit halves (eo_exact_sum32)."""
    return float(lib().eo_exact_sum32(int(v)))


def max_threads():
    return lib().eo_max_threads()


def fast_batch(cfg, pcm, want=("db", "index"), threads=0):
    """The CPU port written for speed (emspec_cpu_fast.c): same pipeline, not bit-identical to the bit model."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    S, L = pcm.shape
    Cn = num_columns(L, cfg.n, cfg.hop)
    db = np.empty((S, Cn, cfg.rows), np.float32) if "db" in want else None
    idx = np.empty((S, Cn, cfg.rows), np.uint8) if "index" in want else None
    rc = lib().eo_fast_batch(C.byref(cfg), _p(pcm, C.c_float), C.c_int32(S), C.c_int64(L), _p(db, C.c_float),
                             _p(idx, C.c_uint8), C.c_int32(threads))
    assert rc == 0, rc
    return db, idx


def postprocess(db, smoothing, agc, cfg, lut=None):
    """numpy restatement of the display post-process (post.hip.inc): db [S][C][R]
    raw dB columns -> (db', index, rgba).  float32 arithmetic in the same order as the kernels."""
    f = np.float32
    db = np.asarray(db, f)
    S, Cn, R = db.shape
    out = np.empty_like(db)
    sm, ag, top = f(smoothing), f(agc), f(cfg.db_top)
    up, down, gmax = f(0.25), f(0.02), f(40.0)
    for s in range(S):
        m = db[s].max(axis=1)
        p = m[0]
        y = None
        for c in range(Cn):
            g = f(0.0)
            if ag > 0:
                d = f(m[c] - p)
                p = f(p + (up if d > 0 else down) * d)
                g = f(min(max(f(ag * f(top - p)), -gmax), gmax))
            x = (db[s, c] + g).astype(f)
            y = x if y is None else (sm * y + f(f(1.0) - sm) * x).astype(f)
            out[s, c] = y
    lo = f(cfg.db_top - cfg.db_range)
    inv = f(1.0 / float(cfg.db_range))
    v = np.clip((out - lo) * inv, f(0), f(1)).astype(f)
    v[out < f(cfg.gate_db)] = 0
    idx = (v * f(255.0) + f(0.5)).astype(np.int32).astype(np.uint8)
    lut = default_lut() if lut is None else lut
    return out, idx, lut[idx]
