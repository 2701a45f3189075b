"""Seeded synthetic time-lapse stacks (BASELINE.md 'Configs -> concrete synthetic inputs').

Host (numpy) generator for tests and small runs.  Content: Poisson(100)+400 background,
Gaussian-blob cells, per-frame drift applied by a Fourier shift *before* quantisation,
disc label masks built from the same centres.  Seeds: 0 image, 1 drift, 2 labels.
"""
from __future__ import annotations

import numpy as np


def cell_layout(y: int, x: int, n_cells: int, seed: int = 0):
    rng = np.random.default_rng(seed)
    cy = rng.uniform(0.08 * y, 0.92 * y, n_cells)
    cx = rng.uniform(0.08 * x, 0.92 * x, n_cells)
    sig = rng.uniform(6.0, 12.0, n_cells) * min(1.0, min(y, x) / 512.0 + 0.25)
    amp = rng.uniform(2000.0, 20000.0, n_cells)
    return cy, cx, sig, amp


def clean_frame(y: int, x: int, n_cells: int, seed: int = 0) -> np.ndarray:
    """Noise-free float64 cell image (no background)."""
    cy, cx, sig, amp = cell_layout(y, x, n_cells, seed)
    img = np.zeros((y, x), dtype=np.float64)
    yy = np.arange(y)[:, None]
    xx = np.arange(x)[None, :]
    for i in range(n_cells):
        r = int(4 * sig[i]) + 1
        y0, y1 = max(0, int(cy[i]) - r), min(y, int(cy[i]) + r + 1)
        x0, x1 = max(0, int(cx[i]) - r), min(x, int(cx[i]) + r + 1)
        img[y0:y1, x0:x1] += amp[i] * np.exp(
            -((yy[y0:y1] - cy[i]) ** 2 + (xx[:, x0:x1] - cx[i]) ** 2) / (2 * sig[i] ** 2))
    return img


def fourier_shift(img: np.ndarray, dy: float, dx: float) -> np.ndarray:
    """out[y, x] = img[y - dy, x - dx] (periodic), exact for band-limited content."""
    y, x = img.shape
    fy = np.fft.fftfreq(y)[:, None]
    fx = np.fft.fftfreq(x)[None, :]
    ph = np.exp(-2j * np.pi * (fy * dy + fx * dx))
    return np.fft.ifft2(np.fft.fft2(img) * ph).real


def drift_walk(t: int, integer: bool, seed: int = 1) -> np.ndarray:
    """(T, 2) cumulative drift of every frame relative to frame 0; drift[0] == 0."""
    rng = np.random.default_rng(seed)
    if integer:
        steps = rng.integers(-3, 4, size=(t, 2)).astype(np.float64)
    else:
        steps = rng.normal(0.0, 0.7, size=(t, 2))
    steps[0] = 0
    return np.cumsum(steps, axis=0)


def make_stack(t: int, c: int, z: int, y: int, x: int, n_cells: int = 50, integer_drift: bool = True,
               seed: int = 0, drift_seed: int = 1):
    """Returns (stack u16 (T,C,Z,Y,X), drift (T,2)).  Channel ch is scaled by (1+0.25 ch);
    z-plane k is scaled by a defocus factor so that the MIP is well defined."""
    rng = np.random.default_rng(seed)
    base = clean_frame(y, x, n_cells, seed)
    drift = drift_walk(t, integer_drift, drift_seed)
    stack = np.empty((t, c, z, y, x), dtype=np.uint16)
    zf = 1.0 - 0.6 * np.abs(np.arange(z) - (z - 1) / 2.0) / max(1.0, (z - 1) / 2.0) if z > 1 else np.ones(1)
    for f in range(t):
        sh = fourier_shift(base, drift[f, 0], drift[f, 1])
        sh = np.maximum(sh, 0.0)
        for ch in range(c):
            for k in range(z):
                bgd = rng.poisson(100.0, size=(y, x)) + 400.0
                stack[f, ch, k] = np.clip(np.rint(bgd + sh * (1 + 0.25 * ch) * zf[k]), 0, 65535).astype(np.uint16)
    return stack, drift


def make_labels(y: int, x: int, n_cells: int = 50, seed: int = 0, label_seed: int = 2) -> np.ndarray:
    """Disc labels (radius 8..14 px scaled with image size) at the blob centres, ids shuffled, u16."""
    cy, cx, sig, _ = cell_layout(y, x, n_cells, seed)
    rng = np.random.default_rng(label_seed)
    ids = rng.permutation(n_cells) + 1
    lab = np.zeros((y, x), dtype=np.uint16)
    yy = np.arange(y)[:, None]
    xx = np.arange(x)[None, :]
    for i in range(n_cells):
        r = 1.2 * sig[i]
        ri = int(r) + 1
        y0, y1 = max(0, int(cy[i]) - ri), min(y, int(cy[i]) + ri + 1)
        x0, x1 = max(0, int(cx[i]) - ri), min(x, int(cx[i]) + ri + 1)
        m = (yy[y0:y1] - cy[i]) ** 2 + (xx[:, x0:x1] - cx[i]) ** 2 <= r * r
        sub = lab[y0:y1, x0:x1]
        sub[m] = ids[i]
    return lab


def make_dense_labels(y: int, x: int, grid: int = 71, seed: int = 2) -> np.ndarray:
    """C5 stress: grid x grid jittered discs (71^2 = 5041 labels at 2048^2), shuffled ids."""
    rng = np.random.default_rng(seed)
    n = grid * grid
    ids = rng.permutation(n) + 1
    py, px = y / grid, x / grid
    lab = np.zeros((y, x), dtype=np.uint16)
    yy = np.arange(y)[:, None]
    xx = np.arange(x)[None, :]
    k = 0
    for gy in range(grid):
        for gx in range(grid):
            cy = (gy + 0.5) * py + rng.uniform(-0.15, 0.15) * py
            cx = (gx + 0.5) * px + rng.uniform(-0.15, 0.15) * px
            r = rng.uniform(0.28, 0.42) * min(py, px)
            ri = int(r) + 1
            y0, y1 = max(0, int(cy) - ri), min(y, int(cy) + ri + 1)
            x0, x1 = max(0, int(cx) - ri), min(x, int(cx) + ri + 1)
            m = (yy[y0:y1] - cy) ** 2 + (xx[:, x0:x1] - cx) ** 2 <= r * r
            sub = lab[y0:y1, x0:x1]
            sub[m] = ids[k]
            k += 1
    return lab
