"""Development aid (CPU, numpy): what would 16-bit storage of the registration spectra cost in accuracy?

Simulates the estimate's data path -- row FFT -> F (stored) -> column FFT, cross-power, inverse column FFT -> Q (stored)
-> inverse row FFT + arg-max, matrix-DFT refinement from Q -- in float32 arithmetic, with F and Q either kept as
float32 (today's kernels) or rounded to float16 at the two store points (F with one fixed scale, Q with one scale per
(frame, column)), and compares the shifts with the float64 oracle.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis_pipeline_amd import synth  # noqa: E402
from oracle import ref_path as O  # noqa: E402


def q16(a, scale):
    """complex64 array -> rounded through float16 at the given scale (broadcastable)"""
    s = (a * scale).astype(np.complex64)
    re = s.real.astype(np.float16).astype(np.float32)
    im = s.imag.astype(np.float16).astype(np.float32)
    return ((re + 1j * im) / scale).astype(np.complex64)


def estimate(ref, mov, half_f=False, half_q=False):
    ny, nx = ref.shape
    fr = np.fft.rfft(ref.astype(np.float32), axis=1).astype(np.complex64)
    fm = np.fft.rfft(mov.astype(np.float32), axis=1).astype(np.complex64)
    if half_f:
        sc = np.float32(32768.0 / (nx * 65535.0))          # |F| <= nx * 65535
        fm = q16(fm, sc)
    g = np.fft.fft(fr, axis=0).astype(np.complex64)
    f = np.fft.fft(fm, axis=0).astype(np.complex64)
    p = (g * np.conj(f) * np.float32(1.0 / (ny * nx))).astype(np.complex64)
    p[0, 0] = 0
    q = (np.fft.ifft(p, axis=0) * ny).astype(np.complex64)  # unnormalised inverse, as the kernel
    if half_q:
        m = np.maximum(np.abs(q.real).max(axis=0), np.abs(q.imag).max(axis=0))
        sc = np.where(m > 0, 16384.0 / np.maximum(m, 1e-30), 1.0).astype(np.float32)
        sc = np.exp2(np.floor(np.log2(sc))).astype(np.float32)   # power of two: scaling itself is exact
        q = q16(q, sc[None, :])
    c = np.fft.irfft(q, n=nx, axis=1) * nx
    k = int(np.argmax(c))
    py, px = divmod(k, nx)
    sy = py - ny if py > ny // 2 else py
    sx = px - nx if px > nx // 2 else px
    # refinement from Q: T[j][v] = sum_y Q[y][v] K(y'_j - y), then the x contraction (float64 here: the
    # question is what the stored values do, not the summation order)
    offs = np.arange(-O.UP_HALF, O.UP_HALF + 1) * O.UP_STEP
    fu = np.fft.fftfreq(ny) * ny
    yy = np.arange(ny)
    rfull = np.fft.fft(q.astype(np.complex128), axis=0) / ny      # back to R[u][v] for the oracle's formula
    ky = np.exp(2j * np.pi * np.outer(sy + offs, fu) / ny)
    t = ky @ rfull                                                 # (15, nx/2+1)
    fv = np.arange(nx // 2 + 1)
    w = np.where((fv == 0) | (fv == nx // 2), 1.0, 2.0)
    kx = np.exp(2j * np.pi * np.outer(fv, sx + offs) / nx)
    v = ((t * w[None, :]) @ kx).real
    kk = int(np.argmax(v))
    n = len(offs)
    iy, ix = divmod(kk, n)
    cy, cx = sy + offs[iy], sx + offs[ix]
    if 0 < iy < n - 1:
        cy += O.UP_STEP * O._parabola(v[iy - 1, ix], v[iy, ix], v[iy + 1, ix])
    if 0 < ix < n - 1:
        cx += O.UP_STEP * O._parabola(v[iy, ix - 1], v[iy, ix], v[iy, ix + 1])
    return cy, cx


def run(name, frames, ref):
    e32 = e16 = 0.0
    for mov in frames:
        o = O.phase_shift(ref, mov)
        a = estimate(ref, mov)
        b = estimate(ref, mov, True, True)
        e32 = max(e32, abs(a[0] - o[0]), abs(a[1] - o[1]))
        e16 = max(e16, abs(b[0] - o[0]), abs(b[1] - o[1]))
    print(f"{name:40s} max |shift - oracle|: f32 storage {e32:.2e} px, f16 storage {e16:.2e} px", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    t = 6
    st, _ = synth.make_stack(t, 1, 1, n, n, n_cells=300 * n * n // (2048 * 2048) + 10, integer_drift=False)
    raw = st[:, 0, 0].astype(np.float64)
    run(f"raw {n}^2 first", raw[1:], raw[0])
    run(f"raw {n}^2 mean", raw, raw.mean(axis=0))
    sub = np.maximum(raw - 500.0, 0.0)
    run(f"bg-subtracted {n}^2 first", sub[1:], sub[0])
    run(f"bg-subtracted {n}^2 mean", sub, sub.mean(axis=0))
    rng = np.random.default_rng(5)
    tex = rng.integers(0, 65536, size=(n, n)).astype(np.float64)
    run("white-noise texture, integer rolls", [np.roll(tex, (3, -5), (0, 1)), np.roll(tex, (-40, 17), (0, 1))], tex)
    dim = rng.poisson(3.0, size=(n, n)).astype(np.float64)
    dim[100:140, 200:260] += 12
    run("dim frame (values ~3), integer roll", [np.roll(dim, (2, 7), (0, 1))], dim)
    bright = np.full((n, n), 65000.0)
    bright[300:340, 100:180] -= 30000
    bright += rng.integers(0, 500, size=(n, n))
    run("near-saturated frame, integer roll", [np.roll(bright, (-6, 4), (0, 1))], bright)
