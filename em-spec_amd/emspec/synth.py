"""Synthetic 48 kHz test/bench audio ([BUILD-DEFINED]).

Stream s uses seed 1000+s.  Counter-based splitmix64 -> uniforms, so the
signal does not depend on numpy's RNG implementation.  Signal = 8 sinusoids
(log-uniform 30 Hz..20 kHz, -40..0 dBFS) + one linear chirp (<= 4e4 Hz/s)
+ Gaussian noise at -60 dBFS + a unit click every 24000 samples, scaled into
[-1, 1], float32.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, count, offset=0):
    """count outputs of splitmix64 seeded with `seed`, starting at draw `offset`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform(seed, count, offset=0):
    return (splitmix64(seed, count, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def stream(s, L, fs=48000.0):
    """One synthetic stream of L float32 samples."""
    seed = 1000 + int(s)
    u = uniform(seed, 64)
    t = np.arange(L, dtype=np.float64) / fs
    x = np.zeros(L)
    for i in range(8):
        f = 30.0 * (20000.0 / 30.0) ** u[i]
        a = 10.0 ** (-40.0 * u[8 + i] / 20.0)
        x += a * np.sin(2 * np.pi * f * t + 2 * np.pi * u[16 + i])
    f0 = 200.0 + 4000.0 * u[24]
    rate = 4.0e4 * (0.25 + 0.75 * u[25])
    dur = L / fs
    # keep the chirp below 20 kHz: wrap its sweep period
    tt = np.mod(t, max(1e-3, min(dur, (20000.0 - f0) / rate)))
    x += 0.25 * np.sin(2 * np.pi * (f0 * tt + 0.5 * rate * tt * tt))
    # Gaussian noise at -60 dBFS via Box-Muller on splitmix uniforms
    un = uniform(seed ^ 0x5EED, 2 * ((L + 1) // 2) , offset=64)
    u1 = np.maximum(un[0::2], 1e-300)
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2 * np.pi * un[1::2])
    g2 = np.sqrt(-2.0 * np.log(u1)) * np.sin(2 * np.pi * un[1::2])
    noise = np.empty(2 * g.size)
    noise[0::2], noise[1::2] = g, g2
    x += 1e-3 * noise[:L]
    x[::24000] += 1.0
    x /= max(1.0, np.max(np.abs(x)))
    return x.astype(np.float32)


def streams(S, L, fs=48000.0, first=0):
    return np.stack([stream(first + s, L, fs) for s in range(S)])
