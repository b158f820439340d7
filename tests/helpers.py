"""Shared test helpers: deterministic synthetic frames and conversions."""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1


def splitmix64(seed: int, count: int) -> np.ndarray:
    """SplitMix64 stream (uint64) -- the generator named in SURVEY.md section 8d."""
    out = np.empty(count, dtype=np.uint64)
    x = seed & MASK64
    for i in range(count):
        x = (x + 0x9E3779B97F4A7C15) & MASK64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        out[i] = z ^ (z >> 31)
    return out


def uniform_frames(batch: int, n: int, bits: int, seed: int) -> np.ndarray:
    """[batch, n, 2] int64, i.i.d. uniform in [-2^(bits-1), 2^(bits-1)) (numpy PCG64, seeded)."""
    rng = np.random.default_rng(seed)
    lo, hi = -(1 << (bits - 1)), (1 << (bits - 1))
    return rng.integers(lo, hi, size=(batch, n, 2), dtype=np.int64)


def chirp_frame(n: int, asig: float = 255.0, fsig: float = 24.0, b: float = 0.95) -> np.ndarray:
    """The deterministic rounded chirp of math/test_fft_radix2.m:45-46,74-75 ([n, 2] int64).
    (The 50 dB awgn() of :62-66 has sigma ~ 0.003 and is erased by the round() of :74-75.)"""
    i = np.arange(n, dtype=np.float64)
    ph = (fsig * i + b * i * i / 2) * 2 * np.pi / n
    win = np.sin(i * np.pi / n)
    re = np.round(asig * np.cos(ph) * win)
    im = np.round(asig * np.sin(ph) * win)
    # Octave round() is half-away-from-zero; numpy is half-even.  Exact .5 cannot occur for these
    # irrational products except at i = 0 (value 0), so both agree.
    return np.stack([re, im], axis=-1).astype(np.int64)


def edge_frames(n: int, bits: int) -> np.ndarray:
    """The 8 edge-case frames of SURVEY.md section 8d scaled to `bits`-bit data: [8, n, 2] int64."""
    full = (1 << (bits - 1)) - 1
    amp = 1 << (bits - 2)
    f = np.zeros((8, n, 2), dtype=np.int64)
    f[1, 1, 0] = amp                                   # impulse at n=1
    f[2, :, 0] = amp                                   # DC
    f[3, 0::2, :] = full                               # alternating +/- full scale
    f[3, 1::2, :] = -full
    f[4] = uniform_frames(1, n, bits, 0xC0FFEE)[0]     # full-scale random (wraps)
    f[5, :, :] = -(1 << (bits - 1))                    # most negative value everywhere
    ch = chirp_frame(n)
    f[6] = ch * max(1, (amp // 256))                   # chirp scaled up
    i = np.arange(n)
    tone = (1 << (bits - 3))
    f[7, :, 0] = np.round(tone * np.cos(2 * np.pi * 129 * i / n)).astype(np.int64)
    f[7, :, 1] = np.round(tone * np.sin(2 * np.pi * 129 * i / n)).astype(np.int64)
    return f


def to_list(frame: np.ndarray):
    return [(int(a), int(b)) for a, b in frame]


def to_complex(a: np.ndarray) -> np.ndarray:
    return a[..., 0].astype(np.float64) + 1j * a[..., 1].astype(np.float64)
