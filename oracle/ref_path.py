"""CPU ORACLE (test infrastructure, NOT product code).

numpy-only restatement of the reference's background-subtract -> register ->
per-label-measure path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the shipped
package (``imageanalysis_pipeline_amd``) never does.

Pinning status
-------------------------------------
* ``measure_labels*``  : follows ``pipeline/analysis.py:80,98,164-173`` and numpy's
  ``mean(where=)`` semantics exactly; cross-checked against
  ``skimage.measure.regionprops`` (tests/golden/make_golden.py).  PINNED.
* ``bg_manual``        : ``pipeline/utility.py:228-233``; fully determined by numpy.  PINNED
  (by construction: it *is* the reference arithmetic).
* ``bg_smo``           : ``pipeline/utility.py:252-264`` + the published SMO operator
  (Silberberg & Grecco).  The ``smo`` package is an un-vendored, unpinned dependency
  that is absent from this image -> PARITY UNPINNED.  Filters are cross-checked
  against ``scipy.ndimage`` only.
* ``gauss_blur``       : ``pipeline/utility.py:462`` (``cv2.GaussianBlur``; cv2 absent)
  -> PARITY UNPINNED (build-defined: fp32 taps, reflect-101, RNE, saturate).
* ``phase_shift`` / ``apply_shift`` : the reference calls ``pystackreg`` (absent,
  unpinned; ``pipeline/utility.py:306,340,351``).  north_star replaces TurboReg by FFT
  phase correlation -> PARITY UNPINNED versus pystackreg; the estimator is
  cross-checked against ``skimage.registration.phase_cross_correlation`` and the
  warp against ``scipy.ndimage.shift(order=3, mode='constant')``.

Conventions
-----------
``shift = phase_shift(ref, mov)`` is the (dy, dx) to *apply* to ``mov`` so that it
overlays ``ref`` -- the skimage convention -- and ``apply_shift(img, dy, dx)``
computes ``out[y, x] = img[y - dy, x - dx]`` (``scipy.ndimage.shift`` convention).
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------------------
# helpers shared by several stages
# --------------------------------------------------------------------------------------


def to_u16_trunc(a: np.ndarray) -> np.ndarray:
    """``a[a<0]=0; a.astype(np.uint16)`` as in utility.py:262-264 / 354-355, except that
    values above 65535 saturate (the reference wraps there: undefined-looking behaviour
    the build deliberately replaces; SURVEY.md section 7 'hard parts')."""
    a = np.where(a < 0, 0, a)
    a = np.minimum(a, 65535)
    return np.trunc(a).astype(np.uint16)


def _reflect_index(i: np.ndarray, n: int) -> np.ndarray:
    """scipy 'reflect' (half-sample symmetric): d c b a | a b c d | d c b a."""
    p = 2 * n
    i = np.mod(i, p)
    return np.where(i >= n, p - 1 - i, i)


def _reflect101_index(i: np.ndarray, n: int) -> np.ndarray:
    """OpenCV BORDER_REFLECT_101 / scipy 'mirror': d c b | a b c d | c b a."""
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


# --------------------------------------------------------------------------------------
# a1: SMO background subtraction  (utility.py:242-270)
# --------------------------------------------------------------------------------------


def _gauss_taps_scipy(sigma: float, truncate: float = 4.0) -> np.ndarray:
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 * x * x / (sigma * sigma))
    return w / w.sum()


def _corr1d_reflect(a: np.ndarray, w: np.ndarray, axis: int) -> np.ndarray:
    r = (len(w) - 1) // 2
    n = a.shape[axis]
    out = np.zeros(a.shape, dtype=np.float64)
    idx = np.arange(n)
    for k in range(-r, r + 1):
        out += w[k + r] * np.take(a, _reflect_index(idx + k, n), axis=axis)
    return out


def smo_image(img: np.ndarray, sigma: float = 0.0, size: int = 7) -> np.ndarray:
    """SMO (Silver Mountain Operator) value per pixel, float64.

    image.astype(float) -> [gaussian(sigma)] -> np.gradient -> unit gradient vectors
    (zero gradient -> zero vector) -> uniform_filter(size) per component -> norm.
    Boundary: scipy.ndimage default 'reflect' for both filters.
    """
    a = np.asarray(img).astype(np.float64)
    if sigma and sigma > 0:
        w = _gauss_taps_scipy(sigma)
        a = _corr1d_reflect(_corr1d_reflect(a, w, 0), w, 1)
    gy, gx = np.gradient(a)
    nrm = np.sqrt(gy * gy + gx * gx)
    ok = nrm > 0
    ny = np.zeros_like(a)
    nx = np.zeros_like(a)
    np.divide(gy, nrm, out=ny, where=ok)
    np.divide(gx, nrm, out=nx, where=ok)
    box = np.full(size, 1.0 / size)
    my = _corr1d_reflect(_corr1d_reflect(ny, box, 0), box, 1)
    mx = _corr1d_reflect(_corr1d_reflect(nx, box, 0), box, 1)
    return np.sqrt(my * my + mx * mx)


def smo_threshold(shape, sigma: float = 0.0, size: int = 7, seed: int = 0, q: float = 0.05) -> float:
    """5 % quantile of the SMO value of a normal-noise image of the given shape.
    The reference draws this image unseeded (utility.py:252); the build seeds it and
    passes the threshold explicitly."""
    rng = np.random.default_rng(seed)
    null = rng.standard_normal(size=shape)
    return float(np.quantile(smo_image(null, sigma, size), q))


def bg_smo(img: np.ndarray, threshold: float, sigma: float = 0.0, size: int = 7):
    """Returns (out_u16, median_times_2, n_bg, mask).  median_times_2 is an integer:
    np.median of an even count can be a half-integer."""
    s = smo_image(img, sigma, size)
    mask = s < threshold
    sel = np.asarray(img)[mask]
    if sel.size == 0:
        med = 0.0
    else:
        med = float(np.median(sel))
    out = to_u16_trunc(np.asarray(img).astype(np.float64) - med)
    return out, int(round(2 * med)), int(sel.size), mask


# --------------------------------------------------------------------------------------
# a2: manual background subtraction  (utility.py:213-233)
# --------------------------------------------------------------------------------------


def bg_manual(img: np.ndarray, r0: int, c0: int, r1: int, c1: int) -> tuple[np.ndarray, int]:
    """ROI = rows r0..r1, cols c0..c1 inclusive (skimage.draw.rectangle semantics);
    bg = int(rint(mean(float32 ROI))) ; subtract ; clip ; u16."""
    im = np.asarray(img).astype(np.float32)
    roi = im[r0:r1 + 1, c0:c1 + 1]
    # Reference: np.mean over a fancy-indexed float32 array (float32 pairwise sum).  The
    # exact integer mean is used here; they differ only when the float32 pairwise sum
    # rounds across a .5 tie (documented tolerance).
    s = int(np.asarray(img)[r0:r1 + 1, c0:c1 + 1].astype(np.int64).sum())
    n = roi.size
    bg = int(np.rint(s / n))
    out = im.astype(np.float64) - bg
    return to_u16_trunc(out), bg


# --------------------------------------------------------------------------------------
# a3: Gaussian blur  (utility.py:462, cv2.GaussianBlur semantics, build-defined)
# --------------------------------------------------------------------------------------


_CV_SMALL_GAUSS = {
    1: [1.0],
    3: [0.25, 0.5, 0.25],
    5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
    7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125],
}


def gauss_taps_cv(ksize: int, sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel(ksize, sigma, CV_32F) as OpenCV 4.1.2 computes it (the version the reference pins,
    environment.yml:15; modules/imgproc/src/smooth.dispatch.cpp, getGaussianKernel): sigma <= 0 ->
    0.3*((k-1)*0.5-1)+0.8, odd sizes up to 7 with sigma <= 0 from the fixed small_gaussian_tab; otherwise
    t_i = exp(scale2X * x_i * x_i) with scale2X = -0.5 / sigma^2 in double, stored as float, summed (the float values,
    in index order, in a double), and normalised by a multiplication with the reciprocal of the sum, rounded to float."""
    if sigma <= 0 and ksize in _CV_SMALL_GAUSS:
        return np.asarray(_CV_SMALL_GAUSS[ksize], dtype=np.float32)
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    scale2x = -0.5 / (float(sigma) * float(sigma))
    cf = np.empty(ksize, dtype=np.float32)
    total = 0.0
    for i in range(ksize):
        x = i - (ksize - 1) * 0.5
        cf[i] = np.float32(np.exp(scale2x * x * x))
        total += float(cf[i])
    inv = 1.0 / total
    return np.array([np.float32(float(c) * inv) for c in cf], dtype=np.float32)


def _sym_fir_f32(a: np.ndarray, w: np.ndarray, axis: int) -> np.ndarray:
    """fp32 symmetric FIR, reflect-101, summation order
    s = w0*x0; s += wi*(x[-i] + x[+i]) for i = 1..r  (unfused mul/add)."""
    r = (len(w) - 1) // 2
    n = a.shape[axis]
    idx = np.arange(n)
    s = (w[r] * np.take(a, idx, axis=axis)).astype(np.float32)
    for i in range(1, r + 1):
        lo = np.take(a, _reflect101_index(idx - i, n), axis=axis)
        hi = np.take(a, _reflect101_index(idx + i, n), axis=axis)
        s = (s + (w[r + i] * (lo + hi).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return s


def gauss_blur(img: np.ndarray, ksize=(15, 15), sigma: float = 5.0) -> np.ndarray:
    """ksize = (kx, ky) as in cv2; row (x) pass first, then column (y) pass; RNE + saturate."""
    kx, ky = ksize
    a = np.asarray(img).astype(np.float32)
    a = _sym_fir_f32(a, gauss_taps_cv(kx, sigma), axis=1)
    a = _sym_fir_f32(a, gauss_taps_cv(ky, sigma), axis=0)
    return np.clip(np.rint(a), 0, 65535).astype(np.uint16)


# --------------------------------------------------------------------------------------
# a4: translation estimate by FFT phase correlation
# --------------------------------------------------------------------------------------

UP_HALF = 7          # sub-pixel grid: offsets -0.7 .. +0.7 in steps of UP_STEP
UP_STEP = 0.1


def fft_window(n: int) -> int:
    """Largest power of two <= n, capped at 4096 (the estimate uses the centred crop)."""
    p = 1 << (int(n).bit_length() - 1)
    return min(p, 4096)


def _crop_center(a: np.ndarray) -> np.ndarray:
    y, x = a.shape
    py, px = fft_window(y), fft_window(x)
    oy, ox = (y - py) // 2, (x - px) // 2
    return a[oy:oy + py, ox:ox + px]


def cross_power(ref: np.ndarray, mov: np.ndarray, normalize: bool = False) -> np.ndarray:
    """Cross-power spectrum P = F conj(G) with the DC bin zeroed (a constant offset carries no
    shift information); normalize=True divides by |P| (phase correlation proper).

    The default is the un-normalised form -- what skimage 0.18's phase_cross_correlation
    computes (_phase_cross_correlation.py:204-207) -- because on photon-noise-limited frames
    the whitened spectrum is dominated by noise bins: measured sub-pixel error 0.1-1 px with
    normalisation versus ~1e-3 px without ('estimator choice')."""
    f = np.fft.fft2(_crop_center(np.asarray(ref, dtype=np.float64)))
    g = np.fft.fft2(_crop_center(np.asarray(mov, dtype=np.float64)))
    p = f * np.conj(g)
    if normalize:
        m = np.abs(p)
        r = np.zeros_like(p)
        np.divide(p, m, out=r, where=m > 0)
        p = r
    p[0, 0] = 0
    return p


def _upsampled(r: np.ndarray, ys: np.ndarray, xs: np.ndarray) -> np.ndarray:
    """Real part of sum_uv R[u,v] exp(2 pi i (u y/Ny + v x/Nx)) for fractional y, x
    (matrix-multiply DFT, skimage _upsampled_dft)."""
    ny, nx = r.shape
    fu = np.fft.fftfreq(ny) * ny  # signed integer frequencies
    fv = np.fft.fftfreq(nx) * nx
    ky = np.exp(2j * np.pi * np.outer(ys, fu) / ny)   # (len(ys), ny)
    kx = np.exp(2j * np.pi * np.outer(fv, xs) / nx)   # (nx, len(xs))
    return (ky @ r @ kx).real / (ny * nx)


def _parabola(vm: float, v0: float, vp: float) -> float:
    den = vm - 2.0 * v0 + vp
    if not den < 0:
        return 0.0
    d = 0.5 * (vm - vp) / den
    return float(min(0.5, max(-0.5, d)))


def phase_shift(ref: np.ndarray, mov: np.ndarray, upsample: bool = True, normalize: bool = False) -> tuple[float, float]:
    """(dy, dx) to apply to ``mov`` to register it onto ``ref``.

    coarse : argmax (first in row-major order) of the real inverse FFT of P, wrapped to
             the signed range (index > N/2 -> index - N), as skimage
             _phase_cross_correlation.py:204-220.
    fine   : matrix-DFT upsampling (skimage :232-250, Guizar-Sicairos) on a 0.1 px grid over
             +-0.7 px around the coarse peak, finished by a 1-D parabolic vertex per axis.
             (A 0.01 px grid, skimage's upsample_factor=100, cannot be resolved in float32:
             neighbouring grid values of the broad correlation peak differ by < 1e-6
             relative; the parabola reaches ~1e-3 px instead.)
    """
    r = cross_power(ref, mov, normalize)
    ny, nx = r.shape
    c = np.fft.ifft2(r).real
    k = int(np.argmax(c))
    py, px = divmod(k, nx)
    sy = py - ny if py > ny // 2 else py
    sx = px - nx if px > nx // 2 else px
    if not upsample:
        return float(sy), float(sx)
    cy, cx = float(sy), float(sx)
    offs = np.arange(-UP_HALF, UP_HALF + 1) * UP_STEP
    v = _upsampled(r, cy + offs, cx + offs)
    k = int(np.argmax(v))
    n = len(offs)
    iy, ix = divmod(k, n)
    cy, cx = cy + offs[iy], cx + offs[ix]
    if 0 < iy < n - 1:
        cy += UP_STEP * _parabola(v[iy - 1, ix], v[iy, ix], v[iy + 1, ix])
    if 0 < ix < n - 1:
        cx += UP_STEP * _parabola(v[iy, ix - 1], v[iy, ix], v[iy, ix + 1])
    return float(cy), float(cx)


def phase_shift_near_ties(ref: np.ndarray, mov: np.ndarray, rel: float = 1e-6) -> list:
    """Every (dy, dx) that ``phase_shift`` would return if one of its two arg-max picks (integer peak, grid point)
    fell on another candidate whose value lies within ``rel`` (relative to the spread of the surface) of the
    maximum.  The refinement is a discontinuous function of its input at such ties: the pick decides which three
    grid values each parabola goes through, and on an oblique ridge the neighbouring column's parabola has its
    vertex somewhere else.  A float32 pipeline cannot be asked to break such a tie the way float64 does; the
    randomised differential runs use this list to tell a tie from a miss (test infrastructure only)."""
    r = cross_power(ref, mov, False)
    ny, nx = r.shape
    c = np.fft.ifft2(r).real
    span = float(c.max() - c.min()) or 1.0
    out = []
    offs = np.arange(-UP_HALF, UP_HALF + 1) * UP_STEP
    n = len(offs)
    for k in np.flatnonzero(c.ravel() >= c.max() - rel * span):
        py, px = divmod(int(k), nx)
        sy = py - ny if py > ny // 2 else py
        sx = px - nx if px > nx // 2 else px
        v = _upsampled(r, sy + offs, sx + offs)
        vspan = float(v.max() - v.min()) or 1.0
        for g in np.flatnonzero(v.ravel() >= v.max() - rel * vspan):
            iy, ix = divmod(int(g), n)
            cy, cx = sy + offs[iy], sx + offs[ix]
            if 0 < iy < n - 1:
                cy += UP_STEP * _parabola(v[iy - 1, ix], v[iy, ix], v[iy + 1, ix])
            if 0 < ix < n - 1:
                cx += UP_STEP * _parabola(v[iy, ix - 1], v[iy, ix], v[iy, ix + 1])
            out.append((float(cy), float(cx)))
    return out


# --------------------------------------------------------------------------------------
# a5: apply shift (cubic B-spline, mirror prefilter, zero outside), utility.py:351-355
# --------------------------------------------------------------------------------------

_Z1 = math.sqrt(3.0) - 2.0


def _bspline3(t: float) -> float:
    t = abs(t)
    if t < 1:
        return 2.0 / 3.0 - t * t + 0.5 * t * t * t
    if t < 2:
        return (2.0 - t) ** 3 / 6.0
    return 0.0


def cardinal_taps(frac: float, radius: int = 24) -> np.ndarray:
    """taps h[k], k=-radius..radius+1, such that out(i0+frac) = sum_k h[k] s(i0+k):
    cubic B-spline evaluation convolved with the (infinite, here truncated) inverse
    B-spline filter g[m] = (-6 z1 / (1 - z1^2)) z1^|m|."""
    c0 = -6.0 * _Z1 / (1.0 - _Z1 * _Z1)
    ks = np.arange(-radius, radius + 2)
    h = np.zeros(len(ks), dtype=np.float64)
    for j in (-1, 0, 1, 2):
        w = _bspline3(frac - j)
        h += w * c0 * _Z1 ** np.abs(ks - j)
    return h


def _shift_axis(a: np.ndarray, d: float, axis: int) -> np.ndarray:
    n = a.shape[axis]
    i = np.arange(n)
    x = i - d                       # source coordinate of each output sample
    i0 = np.floor(x).astype(np.int64)
    frac = float(-d - math.floor(-d))  # identical for every sample of a pure translation
    inside = (x >= 0) & (x <= n - 1)
    radius = 24
    h = cardinal_taps(frac, radius)
    out = np.zeros(a.shape, dtype=np.float64)
    for t, k in enumerate(range(-radius, radius + 2)):
        src = _reflect101_index(i0 + k, n)
        out += h[t] * np.take(a, src, axis=axis)
    shape = [1] * a.ndim
    shape[axis] = n
    return out * inside.reshape(shape)


def apply_shift_f64(img: np.ndarray, dy: float, dx: float) -> np.ndarray:
    """float64 result of the cubic-spline shift (== scipy.ndimage.shift(order=3,
    mode='constant', cval=0) up to the 0.268^25 tap truncation)."""
    a = np.asarray(img).astype(np.float64)
    if float(dy) != 0.0:
        a = _shift_axis(a, float(dy), 0) if float(dy) != round(dy) else _shift_int_axis(a, int(round(dy)), 0)
    if float(dx) != 0.0:
        a = _shift_axis(a, float(dx), 1) if float(dx) != round(dx) else _shift_int_axis(a, int(round(dx)), 1)
    return a


def _shift_int_axis(a: np.ndarray, d: int, axis: int) -> np.ndarray:
    out = np.zeros_like(a)
    n = a.shape[axis]
    if abs(d) >= n:
        return out
    src = [slice(None)] * a.ndim
    dst = [slice(None)] * a.ndim
    if d >= 0:
        src[axis] = slice(0, n - d)
        dst[axis] = slice(d, n)
    else:
        src[axis] = slice(-d, n)
        dst[axis] = slice(0, n + d)
    out[tuple(dst)] = a[tuple(src)]
    return out


def apply_shift(img: np.ndarray, dy: float, dx: float) -> np.ndarray:
    """u16 plane -> u16 plane: warp, clip negatives, truncate (utility.py:354-355).
    Integer shifts are exact copies with zero fill."""
    return to_u16_trunc(apply_shift_f64(img, dy, dx))


# --------------------------------------------------------------------------------------
# im_reg stack logic (utility.py:332-413)
# --------------------------------------------------------------------------------------


def mip(vol: np.ndarray) -> np.ndarray:
    """np.amax over z (utility.py:335,341)."""
    return np.amax(vol, axis=0)


def im_reg_stack(stack: np.ndarray, reg_channel: int = 0, reg_ref: str = "mean", upsample: bool = True,
                 normalize: bool = False):
    """stack: u16 (T, C, Z, Y, X).  Returns (registered u16 stack, shifts (T, 2) f64).

    'first'   : ref = frame 0 (MIP over z)                       utility.py:332-355
    'mean'    : ref = float64 mean over T of the MIPs            utility.py:356-380
    'previous': ref = previous *registered, quantised* frame;    utility.py:381-413
                frame 0 is copied verbatim.
    One (dy, dx) per frame, applied to every channel and z-plane.
    """
    if reg_ref not in ("first", "previous", "mean"):
        raise ValueError(f"{reg_ref} is not valid. Please only put ['first', 'previous', 'mean']")
    t, c, z, y, x = stack.shape
    out = np.zeros_like(stack)
    shifts = np.zeros((t, 2), dtype=np.float64)
    mips = lambda s, f: mip(s[f, reg_channel])  # noqa: E731
    if reg_ref == "first":
        ref = mips(stack, 0).astype(np.float64)
    elif reg_ref == "mean":
        ref = np.mean(np.stack([mips(stack, f) for f in range(t)]), axis=0)
    for f in range(t):
        if reg_ref == "previous":
            if f == 0:
                out[0] = stack[0]
                continue
            ref = mips(out, f - 1).astype(np.float64)
        dy, dx = phase_shift(ref, mips(stack, f), upsample=upsample, normalize=normalize)
        shifts[f] = (dy, dx)
        for ch in range(c):
            for zz in range(z):
                out[f, ch, zz] = apply_shift(stack[f, ch, zz], dy, dx)
    return out, shifts


def chan_shift_stack(stack: np.ndarray, reg_channel: int = 0, upsample: bool = True):
    """utility.py:313-330: per frame, channel and z-plane ``sr.register_transform(ref_plane, plane)`` with
    ref_plane the same frame's and z's plane of the registration channel (no MIP).  Returns (shifted u16
    stack, shifts (T, C, Z, 2)); the registration channel's own planes come back unchanged (shift 0)."""
    t, c, z, y, x = stack.shape
    out = np.zeros_like(stack)
    shifts = np.zeros((t, c, z, 2), dtype=np.float64)
    for f in range(t):
        for ch in range(c):
            for zz in range(z):
                if ch == reg_channel:
                    out[f, ch, zz] = stack[f, ch, zz]
                    continue
                dy, dx = phase_shift(stack[f, reg_channel, zz].astype(np.float64), stack[f, ch, zz], upsample=upsample)
                shifts[f, ch, zz] = (dy, dx)
                out[f, ch, zz] = apply_shift(stack[f, ch, zz], dy, dx)
    return out, shifts


# --------------------------------------------------------------------------------------
# a7/a8/a9: per-label measurement  (analysis.py:80,98,164-173; experiments.py:200-201)
# --------------------------------------------------------------------------------------


def measure_labels_literal(img: np.ndarray, labels: np.ndarray):
    """The reference's own loop.  img: (C, [Z,] Y, X) u16; labels: ([Z,] Y, X).
    Returns dict with ids, area, sum (n, C) int64, mean (n, C) f64, centroid_rne (n, ndim) int,
    centroid (n, ndim) f64 -- axis order (z,) y, x.  (The reference tabulates the mean; the sum is the
    north-star's extra column, the exact integer the mean is the quotient of.)"""
    ids = list(np.unique(labels))[1:]            # analysis.py:80 drops the smallest value
    c = img.shape[0]
    nd = labels.ndim
    mean = np.zeros((len(ids), c), dtype=np.float64)
    tot = np.zeros((len(ids), c), dtype=np.int64)
    cen = np.zeros((len(ids), nd), dtype=np.float64)
    cen_r = np.zeros((len(ids), nd), dtype=np.int64)
    area = np.zeros(len(ids), dtype=np.int64)
    for i, obj in enumerate(ids):
        sel = labels == obj
        for ch in range(c):
            mean[i, ch] = np.nanmean(a=img[ch], where=sel)      # analysis.py:98
            tot[i, ch] = int(np.asarray(img[ch])[sel].astype(np.int64).sum())
        coords = np.where(sel)                                    # analysis.py:164/169
        area[i] = coords[0].size
        for d in range(nd):
            m = np.nanmean(coords[d])
            cen[i, d] = m
            cen_r[i, d] = round(m)                                # round-half-even on np.float64
    return {"ids": np.asarray(ids, dtype=np.int64), "area": area, "sum": tot, "mean": mean,
            "centroid": cen, "centroid_rne": cen_r}


def measure_labels(img: np.ndarray, labels: np.ndarray):
    """Vectorised (np.bincount) restatement; bit-identical to the literal loop because
    every sum is an exact integer below 2**53."""
    lab = np.asarray(labels).astype(np.int64).ravel()
    nb = int(lab.max()) + 1 if lab.size else 1
    cnt = np.bincount(lab, minlength=nb)
    present = np.nonzero(cnt)[0]
    ids = present[1:]
    c = img.shape[0]
    nd = labels.ndim
    mean = np.zeros((len(ids), c), dtype=np.float64)
    tot = np.zeros((len(ids), c), dtype=np.int64)
    for ch in range(c):
        s = np.bincount(lab, weights=np.asarray(img[ch]).astype(np.float64).ravel(), minlength=nb)
        mean[:, ch] = s[ids] / cnt[ids].astype(np.float64)
        tot[:, ch] = s[ids].astype(np.int64)                      # exact: the sums stay below 2**53
    grids = np.indices(labels.shape)
    cen = np.zeros((len(ids), nd), dtype=np.float64)
    for d in range(nd):
        s = np.bincount(lab, weights=grids[d].astype(np.float64).ravel(), minlength=nb)
        cen[:, d] = s[ids] / cnt[ids].astype(np.float64)
    cen_r = np.rint(cen).astype(np.int64)                         # rint == round-half-even
    return {"ids": ids.astype(np.int64), "area": cnt[ids].astype(np.int64), "sum": tot, "mean": mean,
            "centroid": cen, "centroid_rne": cen_r}


# --------------------------------------------------------------------------------------
# whole path on one FOV (what bench.py's cpu_baseline times)
# --------------------------------------------------------------------------------------


def run_path(stack: np.ndarray, labels: np.ndarray, threshold: float, reg_ref: str = "first",
             sigma: float = 0.0, size: int = 7):
    return run_path_opts(stack, labels, threshold, reg_ref, sigma, size, True)


def run_path_opts(stack: np.ndarray, labels: np.ndarray, threshold: float, reg_ref: str = "first",
                  sigma: float = 0.0, size: int = 7, upsample: bool = True):
    """stack (T,C,Z,Y,X) u16, labels (T,Z,Y,X) u16 -> (registered stack, shifts, tables)."""
    t, c, z, y, x = stack.shape
    bg = np.empty_like(stack)
    for f in range(t):
        for ch in range(c):
            for zz in range(z):
                bg[f, ch, zz] = bg_smo(stack[f, ch, zz], threshold, sigma, size)[0]
    reg, shifts = im_reg_stack(bg, 0, reg_ref, upsample=upsample)
    tables = []
    for f in range(t):
        if z == 1:
            tables.append(measure_labels(reg[f, :, 0], labels[f, 0]))
        else:
            tables.append(measure_labels(reg[f], labels[f]))
    return reg, shifts, tables


# ---------------------------------------------------------------- distance map / pixel table
def edt_sq(mask: np.ndarray) -> np.ndarray:
    """Squared Euclidean distance of every pixel to the nearest non-zero pixel of a 2-D mask, exact int64:
    the square of ``mahotas.distance(np.logical_not(mask), 'euclidean')`` (utility.py:871-876; mahotas is
    absent here: pinned against scipy.ndimage.distance_transform_edt in tests/golden).  Separable: column
    distances, then the exact minimum over columns of g[j]^2 + (x-j)^2."""
    m = np.asarray(mask) != 0
    y, x = m.shape
    inf = 30000
    g = np.full((y, x), inf, dtype=np.int64)
    d = np.full(x, inf, dtype=np.int64)
    for r in range(y):
        d = np.where(m[r], 0, np.minimum(d + 1, inf))
        g[r] = d
    d = np.full(x, inf, dtype=np.int64)
    for r in range(y - 1, -1, -1):
        d = np.minimum(g[r], np.minimum(d + 1, inf))
        g[r] = d
    jj = np.arange(x, dtype=np.int64)
    dx2 = (jj[:, None] - jj[None, :]) ** 2            # (x, j)
    out = np.empty((y, x), dtype=np.int64)
    for r in range(y):
        out[r] = (g[r][None, :] ** 2 + dx2).min(axis=1)
    return out


def pixel_table(cell_mask: np.ndarray, img: np.ndarray, ref_mask: np.ndarray, pix_size: float):
    """analysis.py:276-283 for one frame: (dmap int64, values (count, C)) of the pixels where cell_mask is
    set, in boolean-indexing order; dmap = round(distance * pix_size) with np.round (half to even)."""
    sel = np.asarray(cell_mask).astype(bool)
    dist = np.sqrt(edt_sq(ref_mask).astype(np.float64))
    dmap = np.round(dist[sel] * pix_size).astype(int)
    vals = np.stack([np.asarray(img[c])[sel] for c in range(img.shape[0])], axis=1)
    return dmap, vals


# ---------------------------------------------------------------- mask stitching
def label_overlap(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """cellpose.metrics._label_overlap: overlap[i, j] = #{a == i and b == j} (cellpose is absent from the
    image; this is its published definition, used by utility.py:718)."""
    a = np.asarray(a).ravel().astype(np.int64)
    b = np.asarray(b).ravel().astype(np.int64)
    na, nb = int(a.max()) + 1, int(b.max()) + 1
    return np.bincount(a * nb + b, minlength=na * nb).reshape(na, nb)


def iou_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """cellpose.metrics._intersection_over_union(masks_true=a, masks_pred=b)."""
    ov = label_overlap(a, b).astype(np.float64)
    n_pred = ov.sum(axis=0, keepdims=True)
    n_true = ov.sum(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = ov / (n_pred + n_true - ov)
    iou[np.isnan(iou)] = 0.0
    return iou


def label_mode(masks: np.ndarray) -> np.ndarray:
    """utility.py:707-711: scipy.stats.mode over axis 0 of the stack with zeros as nan (omitted), as int:
    the most frequent non-zero label per pixel, ties to the smaller label, 0 where never labelled."""
    m = np.asarray(masks)
    t = m.shape[0]
    flat = m.reshape(t, -1)
    out = np.zeros(flat.shape[1], dtype=np.int64)
    for k in range(flat.shape[1]):
        v = flat[:, k]
        v = v[v != 0]
        if v.size:
            vals, cnt = np.unique(v, return_counts=True)
            out[k] = vals[np.argmax(cnt)]
    return out.reshape(m.shape[1:])


def _stitch_pass(masks: np.ndarray, ref_of, stitch_threshold: float, new_ids: bool) -> np.ndarray:
    """The loop body shared by Cellpose's stitch3D (new_ids=True: unmatched cells get fresh ids; frame i is
    matched to frame i-1) and the second round of utility.py:717-737 (new_ids=False: unmatched cells are
    dropped; every frame is matched to the master mask)."""
    masks = masks.copy()
    mmax = int(masks[0].max())
    empty = 0
    rng = range(1, len(masks)) if new_ids else range(len(masks))
    for i in rng:
        iou = iou_matrix(masks[i], ref_of(masks, i))[1:, 1:]
        if not iou.size and empty == 0:
            mmax = int(masks[i].max())
        elif not iou.size and not empty == 0:
            icount = int(masks[i].max())
            istitch = np.arange(mmax + 1, mmax + icount + 1, 1, int)
            mmax += icount
            istitch = np.append(np.array(0), istitch)
            masks[i] = istitch[masks[i]]
        else:
            iou[iou < stitch_threshold] = 0.0
            iou[iou < iou.max(axis=0)] = 0.0
            istitch = iou.argmax(axis=1) + 1
            ino = np.nonzero(iou.max(axis=1) == 0.0)[0]
            if new_ids:
                istitch[ino] = np.arange(mmax + 1, mmax + len(ino) + 1, 1, int)
                mmax += len(ino)
            else:
                istitch[ino] = 0
            istitch = np.append(np.array(0), istitch)
            masks[i] = istitch[masks[i]]
            empty = 1
    return masks


def modif_stitch3D(masks: np.ndarray, stitch_threshold: float) -> np.ndarray:
    """utility.py:700-738 on a (T, Y, X) integer label stack.  The first round is Cellpose's stitch3D
    (un-vendored third party, version unpinned in environment.yml: restated from its published source,
    parity unpinned); the master mask and the second round are the reference's own lines 705-737."""
    thr = 1 - stitch_threshold
    m = _stitch_pass(np.asarray(masks).astype(np.int64), lambda mm, i: mm[i - 1], thr, new_ids=True)
    master = label_mode(m)
    return _stitch_pass(m, lambda mm, i: master, thr, new_ids=False)


# ---------------------------------------------------------------- compartment masks
def disk_offsets(r: int):
    """skimage.morphology.disk(r) as (dy, dx) offsets: dx^2 + dy^2 <= r^2."""
    return [(dy, dx) for dy in range(-r, r + 1) for dx in range(-r, r + 1) if dy * dy + dx * dx <= r * r]


def label_erode(mask: np.ndarray, r: int, iterations: int = 1) -> np.ndarray:
    """utility.py:776-801 for one 2-D label plane: sum over objects of cv2.erode(mask == obj ? obj : 0,
    disk(r)) (cv2 absent: erosion = minimum over the footprint, border value +inf, i.e. pixels outside the
    image never erode; pinned against scipy.ndimage.grey_erosion per object in tests/golden)."""
    m = np.asarray(mask).astype(np.int64)
    y, x = m.shape
    for _ in range(iterations):
        keep = m != 0
        for dy, dx in disk_offsets(r):
            if abs(dy) >= y or abs(dx) >= x:
                continue                                   # the whole shifted plane lies outside the image
            sh = np.full_like(m, -1)                       # -1 = outside the image: matches everything
            ys, yd = slice(max(dy, 0), y + min(dy, 0)), slice(max(-dy, 0), y + min(-dy, 0))
            xs, xd = slice(max(dx, 0), x + min(dx, 0)), slice(max(-dx, 0), x + min(-dx, 0))
            sh[yd, xd] = m[ys, xs]
            keep &= (sh == m) | (sh == -1)
        m = np.where(keep, m, 0)
    return m


def label_dilate(mask: np.ndarray, r: int, iterations: int = 1) -> np.ndarray:
    """utility.py:804-826 for one 2-D label plane: cv2.dilate(mask, disk(r)) = maximum over the footprint."""
    m = np.asarray(mask).astype(np.int64)
    y, x = m.shape
    for _ in range(iterations):
        out = m.copy()
        for dy, dx in disk_offsets(r):
            if abs(dy) >= y or abs(dx) >= x:
                continue
            ys, yd = slice(max(dy, 0), y + min(dy, 0)), slice(max(-dy, 0), y + min(-dy, 0))
            xs, xd = slice(max(dx, 0), x + min(dx, 0)), slice(max(-dx, 0), x + min(-dx, 0))
            out[yd, xd] = np.maximum(out[yd, xd], m[ys, xs])
        m = out
    return m


# ---------------------------------------------------------------------------------------------
# Track clean-up after stitching (experiments.py:195-221) and compartments (experiments.py:251-262)

def filter_mask_shapes(masks: np.ndarray, shape_threshold: float = 0.2) -> np.ndarray:
    """experiments.py:195-206, the reference's loop as written (one object at a time)."""
    new_mask = np.zeros(masks.shape)
    for obj in list(np.unique(masks))[1:]:
        temp = masks.copy()
        temp[temp != obj] = 0
        t = np.where(temp != 0)[0]
        f_lst, size_lst = np.unique(t, return_counts=True)
        mean_size = np.mean(size_lst)
        up = mean_size + mean_size * shape_threshold
        down = mean_size - mean_size * shape_threshold
        temp[f_lst[np.where((size_lst < down) | (size_lst > up))]] = 0
        new_mask += temp
    return new_mask


def renumber_masks(masks: np.ndarray) -> np.ndarray:
    """experiments.py:208-211."""
    out = masks.copy()
    for n, val in enumerate(list(np.unique(out))):
        out[out == val] = n
    return out


def trim_incomplete_tracks(masks: np.ndarray, frames: int) -> np.ndarray:
    """experiments.py:217-221."""
    out = masks.copy()
    for obj in list(np.unique(out))[1:]:
        if len(set(np.where(out == obj)[0])) != frames:
            out[out == obj] = 0
    return out


def compartment_masks(mask_stack: np.ndarray, rad_ero: int, rad_dil=None, dil_iteration=1, ero_iteration=1):
    """experiments.py:251-262 with this module's label_erode / label_dilate: (mb, cyto, full)."""
    full = mask_stack
    if rad_dil:
        full = np.stack([label_dilate(p, rad_dil, dil_iteration) for p in full.reshape(-1, *full.shape[-2:])]).reshape(full.shape)
    ero = np.stack([label_erode(p, rad_ero, ero_iteration) for p in full.reshape(-1, *full.shape[-2:])]).reshape(full.shape)
    return full - ero, ero, full
