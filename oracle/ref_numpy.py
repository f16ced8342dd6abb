"""Independent numpy formulation of the three-window reassignment method.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (no reference source exists,
/root/reference/README.md:73).  This file shares no code with
emspec_oracle.c: it uses np.fft.rfft on three explicitly windowed frames
(SURVEY.md §8a rows "Windows", "STFT x3", "Reassign") and is used to
  (i)  pin the C float64 oracle (tests/test_oracle.py), and
  (ii) generate the committed golden fixtures (tests/golden/make_golden.py).
"""
import numpy as np


def windows(n):
    i = np.arange(n, dtype=np.float64)
    h = 0.5 - 0.5 * np.cos(2 * np.pi * i / n)          # periodic Hann
    th = (i - n // 2) * h                               # time-weighted
    dh = (np.pi / n) * np.sin(2 * np.pi * i / n)        # d h / d n
    return h, th, dh


def edges_bins(n, rows, fs, fmin, fmax):
    r = np.arange(rows + 1, dtype=np.float64)
    return fmin * (fmax / fmin) ** (r / rows) * n / fs


def reassign_frames(pcm, n, hop, frame0, nframes, rows=1024, fs=48000.0, fmin=20.0, fmax=24000.0,
                    power_floor=1e-14, reassign=True):
    """Returns dict(power, that, khat, col, row), each [nframes, n/2+1]."""
    pcm = np.asarray(pcm, dtype=np.float64)
    h, th, dh = windows(n)
    K = n // 2 + 1
    D = -(-n // (2 * hop)) if reassign else 0
    e = edges_bins(n, rows, fs, fmin, fmax)
    pfloor = power_floor * (n / 4.0) ** 2
    out = {k: np.zeros((nframes, K)) for k in ("power", "that", "khat")}
    out["col"] = np.zeros((nframes, K), np.int32)
    out["row"] = np.full((nframes, K), -1, np.int32)
    kk = np.arange(K, dtype=np.float64)
    for f in range(nframes):
        j = frame0 + f
        x = pcm[j * hop:j * hop + n]
        Xh, Xt, Xd = np.fft.rfft(x * h), np.fft.rfft(x * th), np.fft.rfft(x * dh)
        P = np.abs(Xh) ** 2
        ok = (P >= pfloor) & (P > 0)
        Ps = np.where(ok, P, 1.0)
        ts = np.real(Xt * np.conj(Xh)) / Ps if reassign else np.zeros(K)
        ks = -(n / (2 * np.pi)) * np.imag(Xd * np.conj(Xh)) / Ps if reassign else np.zeros(K)
        that = j * hop + n // 2 + np.where(ok, ts, 0.0)
        khat = kk + np.where(ok, ks, 0.0)
        cf = np.floor(ts / hop + 0.5)
        inrange = ok & (np.abs(cf) <= D)
        col = np.where(inrange, j + cf, j).astype(np.int32)
        row = np.searchsorted(e, khat, side="right") - 1
        valid = inrange & (khat >= e[0]) & (khat < e[-1])
        out["power"][f], out["that"][f], out["khat"][f] = P, that, khat
        out["col"][f] = col
        out["row"][f] = np.where(valid, row, -1)
    return out
