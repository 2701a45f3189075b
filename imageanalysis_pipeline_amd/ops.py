"""Array-level host API over the C-ABI (include/iap.h).

torch tensors are only the buffer and stream carrier: every function takes CUDA(HIP)
tensors, passes ``data_ptr()`` + the current stream to libiap_hip.so and returns tensors.
There is deliberately no CPU implementation here; CPU tensors raise.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

REF_MODES = {"first": 0, "mean": 1, "previous": 2}
REG_UPSAMPLE = 1
REG_NORMALIZE = 2


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if not t.is_cuda:
            raise _lib.IapError("imageanalysis_pipeline_amd.ops needs device tensors: the HIP library is the "
                                "only implementation (no CPU fallback)")
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")


def _u16(t: torch.Tensor) -> None:
    if t.dtype != torch.uint16:
        raise TypeError(f"expected torch.uint16, got {t.dtype}")


def to_device_u16(a: np.ndarray, device="cuda") -> torch.Tensor:
    """numpy u16 array -> device tensor (torch.uint16)."""
    a = np.ascontiguousarray(a, dtype=np.uint16)
    return torch.from_numpy(a.view(np.int16)).to(device).view(torch.uint16)


def to_numpy_u16(t: torch.Tensor) -> np.ndarray:
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


# ----------------------------------------------------------------------------- measurement


def max_label(labels: torch.Tensor) -> int:
    _need_cuda(labels)
    _u16(labels)
    out = torch.empty(1, dtype=torch.int32, device=labels.device)
    lib = _lib.load()
    _lib.check(lib.iap_max_u16(labels.data_ptr(), labels.numel(), out.data_ptr(), _stream()), "iap_max_u16")
    return int(out.item())


PROVISIONAL_LABELS = 8192   # accumulator rows of a measurement whose label range is not known beforehand ...
PROVISIONAL_BYTES = 256 << 20   # ... as far as T frames of them fit this many bytes (never fewer than 256 rows)


def provisional_labels(t: int, c: int) -> int:
    """Rows of the provisional table of a T-frame, C-channel measurement: T x rows x (4+C) int64 stays within
    PROVISIONAL_BYTES (T=5000, C=3 would otherwise hold 2.3 GB for masks of a few hundred ids)."""
    return int(min(PROVISIONAL_LABELS, max(256, PROVISIONAL_BYTES // (max(t, 1) * (4 + c) * 8))))


def _measure_call(img, labels, L, acc, frame_max=None):
    t, c, z, y, x = img.shape
    lib = _lib.load()
    if frame_max is None:
        _lib.check(lib.iap_measure_labels_u16(img.data_ptr(), labels.data_ptr(), t, c, z, y, x, L, acc.data_ptr(),
                                              _stream()), "iap_measure_labels_u16")
    else:
        _lib.check(lib.iap_measure_labels_max_u16(img.data_ptr(), labels.data_ptr(), t, c, z, y, x, L, acc.data_ptr(),
                                                  frame_max.data_ptr(), _stream()), "iap_measure_labels_max_u16")


def measure_accumulate(img: torch.Tensor, labels: torch.Tensor, n_labels: int | None = None) -> torch.Tensor:
    """img u16 (T,C,Z,Y,X), labels u16 (T,Z,Y,X) -> exact integer accumulators
    int64 (T, L, 4+C): count, sum_z, sum_y, sum_x, sum_I[c]  (analysis.py:80-98,164-173).
    L = ``n_labels`` or, when that is not given, largest label + 1 (the reference's np.unique(mask) range): the
    measuring pass itself reports the largest label, so the label stack is not read a second time for it; only a
    stack with more ids than the provisional table holds (``provisional_labels``) is measured again with the range
    it turned out to need."""
    _need_cuda(img, labels)
    _u16(img)
    _u16(labels)
    if img.dim() != 5 or labels.dim() != 4:
        raise ValueError("img must be (T,C,Z,Y,X) and labels (T,Z,Y,X)")
    t, c, z, y, x = img.shape
    if tuple(labels.shape) != (t, z, y, x):
        raise ValueError(f"labels shape {tuple(labels.shape)} does not match image stack {(t, z, y, x)}")
    if n_labels is not None:
        acc = torch.empty((t, int(n_labels), 4 + c), dtype=torch.int64, device=img.device)
        _measure_call(img, labels, int(n_labels), acc)
        return acc
    L0 = provisional_labels(t, c)
    acc = torch.empty((t, L0, 4 + c), dtype=torch.int64, device=img.device)
    fmax = torch.empty(t, dtype=torch.int32, device=img.device)
    _measure_call(img, labels, L0, acc, fmax)
    L = int(fmax.max().item()) + 1
    if L > L0:
        acc = torch.empty((t, L, 4 + c), dtype=torch.int64, device=img.device)
        _measure_call(img, labels, L, acc)
        return acc
    return acc[:, :L].contiguous()


def measure_finalize(acc: torch.Tensor):
    """accumulators -> (area i64 (T,L), mean f64 (T,L,C), centroid f64 (T,L,3), centroid_rne i64 (T,L,3));
    centroid axis order is (z, y, x)."""
    _need_cuda(acc)
    t, L, f = acc.shape
    c = f - 4
    dev = acc.device
    area = torch.empty((t, L), dtype=torch.int64, device=dev)
    mean = torch.empty((t, L, c), dtype=torch.float64, device=dev)
    cen = torch.empty((t, L, 3), dtype=torch.float64, device=dev)
    cen_r = torch.empty((t, L, 3), dtype=torch.int64, device=dev)
    lib = _lib.load()
    _lib.check(lib.iap_measure_finalize(acc.data_ptr(), t, L, c, area.data_ptr(), mean.data_ptr(), cen.data_ptr(),
                                        cen_r.data_ptr(), _stream()), "iap_measure_finalize")
    return area, mean, cen, cen_r


def measure_labels(img: torch.Tensor, labels: torch.Tensor, n_labels: int | None = None):
    """regionprops-style tables per frame with the reference's label selection: the labels
    present in the frame minus the smallest one (analysis.py:80 ``np.unique(mask)[1:]``).

    Returns a list (one dict per frame) of numpy arrays: ids, area, sum (n,C; exact int64 intensity sums),
    mean (n,C), centroid (n,ndim) and centroid_rne (n,ndim) with axis order (z,) y, x."""
    acc = measure_accumulate(img, labels, n_labels)
    area, mean, cen, cen_r = (a.cpu().numpy() for a in measure_finalize(acc))
    sums = acc[:, :, 4:].cpu().numpy()
    z = img.shape[2]
    out = []
    for f in range(img.shape[0]):
        present = np.nonzero(area[f])[0]
        ids = present[1:]
        sl = slice(0, 3) if z > 1 else slice(1, 3)
        out.append({"ids": ids.astype(np.int64), "area": area[f, ids], "sum": sums[f, ids], "mean": mean[f, ids],
                    "centroid": cen[f, ids][:, sl], "centroid_rne": cen_r[f, ids][:, sl]})
    return out


# ----------------------------------------------------------------------------- background

_ws_cache: dict = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch tensor per device and stream (the library itself never allocates; calls
    on different streams may overlap, so they must not share scratch)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = None
        _ws_cache[key] = None
        ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def _planes(t: torch.Tensor):
    if t.dim() < 2:
        raise ValueError("need at least a (Y, X) plane")
    y, x = t.shape[-2:]
    n = t.numel() // (y * x)
    return n, y, x


def smo_image(planes: torch.Tensor, size: int = 7, sigma: float = 0.0) -> torch.Tensor:
    """SMO value (float32) of u16 or f32 planes (..., Y, X)  [smo.SMO.smo_image, utility.py:252]; ``sigma`` > 0:
    gaussian pre-smoothing first, as ``SMO(sigma=...)``."""
    _need_cuda(planes)
    if planes.dtype == torch.uint16:
        kind = 0
    elif planes.dtype == torch.float32:
        kind = 1
    else:
        raise TypeError("smo_image takes uint16 or float32 planes")
    n, y, x = _planes(planes)
    out = torch.empty(planes.shape, dtype=torch.float32, device=planes.device)
    lib = _lib.load()
    nb = lib.iap_smo_image_ws_bytes(n, y, x, float(sigma))
    ws = _workspace(nb, planes.device) if nb else None
    _lib.check(lib.iap_smo_image(planes.data_ptr(), kind, out.data_ptr(), n, y, x, float(sigma), int(size),
                                 ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                 _stream()), "iap_smo_image")
    return out


_thr_cache: dict = {}


def smo_threshold(shape, sigma: float = 0.0, size: int = 7, seed: int = 0, q: float = 0.05, device="cuda") -> float:
    """5 % quantile of the SMO value of a seeded normal-noise image (the null distribution the
    reference draws unseeded in ``SMO(shape=..., sigma=, size=)``, utility.py:252: the same operator, gaussian
    pre-smoothing included, applied to the noise image).  The SMO image comes from the HIP kernels; only the
    quantile selection of the copied-back values runs on the host."""
    key = (tuple(shape), float(sigma), int(size), int(seed), float(q))
    if key not in _thr_cache:
        rng = np.random.default_rng(seed)
        null = torch.from_numpy(rng.standard_normal(size=tuple(shape)).astype(np.float32)).to(device)
        s = smo_image(null, size, sigma).cpu().numpy().astype(np.float64)
        _thr_cache[key] = float(np.quantile(s, q))
    return _thr_cache[key]


def bg_smo(planes: torch.Tensor, threshold: float, sigma: float = 0.0, size: int = 7, out: torch.Tensor | None = None):
    """utility.py:255-264 on u16 planes (..., Y, X); returns (out, med2 int32 (n,), nbg int32 (n,)).
    ``out=planes`` runs in place (the reference overwrites its files)."""
    _need_cuda(planes)
    _u16(planes)
    n, y, x = _planes(planes)
    if out is None:
        out = torch.empty_like(planes)
    _need_cuda(out)
    _u16(out)
    if out.shape != planes.shape:
        raise ValueError("out must have the shape of planes")
    lib = _lib.load()
    nb = lib.iap_bg_smo_ws_bytes(n, y, x, float(sigma))
    ws = _workspace(nb, planes.device)
    med2 = torch.empty(n, dtype=torch.int32, device=planes.device)
    nbg = torch.empty(n, dtype=torch.int32, device=planes.device)
    _lib.check(lib.iap_bg_smo_u16(planes.data_ptr(), out.data_ptr(), n, y, x, float(sigma), int(size),
                                  float(threshold), med2.data_ptr(), nbg.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _stream()), "iap_bg_smo_u16")
    return out, med2, nbg


def bg_smo_levels(planes: torch.Tensor, threshold: float, sigma: float = 0.0, size: int = 7):
    """The background levels of ``bg_smo`` without the subtracted stack: (med2 int32 (n,) = 2 * median of the
    background pixels, nbg int32 (n,)).  ``im_reg(..., sub2=med2)`` subtracts them as it reads."""
    _need_cuda(planes)
    _u16(planes)
    n, y, x = _planes(planes)
    lib = _lib.load()
    ws = _workspace(lib.iap_bg_smo_ws_bytes(n, y, x, float(sigma)), planes.device)
    med2 = torch.empty(n, dtype=torch.int32, device=planes.device)
    nbg = torch.empty(n, dtype=torch.int32, device=planes.device)
    _lib.check(lib.iap_bg_smo_u16(planes.data_ptr(), None, n, y, x, float(sigma), int(size), float(threshold),
                                  med2.data_ptr(), nbg.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "iap_bg_smo_u16")
    return med2, nbg


def bg_manual(planes: torch.Tensor, roi, out: torch.Tensor | None = None):
    """utility.py:228-233: roi = (r0, c0, r1, c1) inclusive; returns (out, bg int32 (n,))."""
    _need_cuda(planes)
    _u16(planes)
    n, y, x = _planes(planes)
    if out is None:
        out = torch.empty_like(planes)
    r0, c0, r1, c1 = (int(v) for v in roi)
    ws = _workspace(n * 16, planes.device)
    bg = torch.empty(n, dtype=torch.int32, device=planes.device)
    lib = _lib.load()
    _lib.check(lib.iap_bg_const_u16(planes.data_ptr(), out.data_ptr(), n, y, x, r0, c0, r1, c1, bg.data_ptr(),
                                    ws.data_ptr(), ws.numel(), _stream()), "iap_bg_const_u16")
    return out, bg


def gauss_blur(planes: torch.Tensor, ksize=(15, 15), sigma: float = 5.0) -> torch.Tensor:
    """cv2.GaussianBlur(im, (kx, ky), sigma).astype(uint16) of utility.py:462 on u16 planes."""
    _need_cuda(planes)
    _u16(planes)
    if type(ksize) is not tuple:
        raise TypeError("The input 'blur_kernel' must be a tuple: (int,int)")
    n, y, x = _planes(planes)
    out = torch.empty_like(planes)
    lib = _lib.load()
    _lib.check(lib.iap_gauss_blur_u16(planes.data_ptr(), out.data_ptr(), n, y, x, int(ksize[0]), int(ksize[1]),
                                      float(sigma), _stream()), "iap_gauss_blur_u16")
    return out


# ----------------------------------------------------------------------------- registration

DEFAULT_REG_BATCH = 64   # fewest frames per launch group: C2 estimate 17.5 / 13.7 / 13.4 / 12.8 / 12.65 ms at 4 / 8 / 16 / 32 / 64 (round 2)
REG_GROUP_BYTES = 4 << 30   # ... and, with ``batch=None``, as many frames as this much workspace holds, never fewer than 64
                            # (25.5 MB per frame at 2048^2: 128-frame groups there, C2 estimate 11.18 -> 10.99 ms against
                            # 64-frame groups in three alternating runs of round 3; groups of 7 .. 47 frames, sized
                            # for the Infinity Cache, were slower (16.2 .. 11.0 ms) and so was one 500-frame group
                            # in round 2, whose column pass finds none of the row pass's output in the caches: 7.2
                            # against 5.9 ms; small planes: one group, fewer launches)


def _reg_batch(batch, frames, ws_bytes_of):
    """frames per registration launch group: the caller's, or what REG_GROUP_BYTES of workspace holds"""
    frames = max(1, int(frames))
    if batch is not None and int(batch) > 0:
        return max(1, min(int(batch), frames))
    b = min(frames, DEFAULT_REG_BATCH)
    while b < frames and ws_bytes_of(min(frames, 2 * b)) <= REG_GROUP_BYTES:
        b = min(frames, 2 * b)
    return b


def mip(stack: torch.Tensor) -> torch.Tensor:
    """np.amax over z (utility.py:335,341): u16 (n, Z, Y, X) -> (n, Y, X)."""
    _need_cuda(stack)
    _u16(stack)
    n, z, y, x = stack.shape
    out = torch.empty((n, y, x), dtype=torch.uint16, device=stack.device)
    lib = _lib.load()
    _lib.check(lib.iap_mip_u16(stack.data_ptr(), z * y * x, out.data_ptr(), n, z, y, x, _stream()), "iap_mip_u16")
    return out


def lowered_channel_mip(stack: torch.Tensor, channel: int, sub2: torch.Tensor) -> torch.Tensor:
    """The registration planes of a RAW stack u16 (T, C, Z, Y, X) whose background levels ``sub2`` (int32, one per
    plane, ``bg_smo_levels``) have not been subtracted yet: z-MIP (utility.py:335,341) of channel ``channel``'s
    planes read as max(v - level, 0) -> u16 (T, Y, X), the planes ``im_reg(bg_smo(stack)[0])`` estimates on.  Reads
    the channel in place (no copy of the strided view)."""
    _need_cuda(stack, sub2)
    _u16(stack)
    if stack.dim() != 5:
        raise ValueError("stack must be (T, C, Z, Y, X)")
    t, c, z, y, x = stack.shape
    if sub2.dtype != torch.int32 or sub2.numel() != t * c * z:
        raise ValueError("sub2 must be int32 with one value per plane (T*C*Z)")
    if not 0 <= channel < c:
        raise ValueError(f"channel {channel} not in 0..{c - 1}")
    out = torch.empty((t, y, x), dtype=torch.uint16, device=stack.device)
    if t == 0:
        return out
    lib = _lib.load()
    px = y * x
    _lib.check(lib.iap_mip_sub_u16(stack.data_ptr() + 2 * channel * z * px, c * z * px, out.data_ptr(), t, z, y, x,
                                   sub2.data_ptr() + 4 * channel * z, c * z, _stream()), "iap_mip_sub_u16")
    return out


def _reg_flags(upsample: bool, normalize: bool) -> int:
    return (REG_UPSAMPLE if upsample else 0) | (REG_NORMALIZE if normalize else 0)


def estimate_shifts(ref: torch.Tensor, mov: torch.Tensor, upsample: bool = True, normalize: bool = False,
                    batch: int | None = None) -> torch.Tensor:
    """(dy, dx) to apply to each moving plane so that it overlays ``ref`` (replaces
    StackReg.register, utility.py:340).  ref: (Y, X) u16 / int32-sums / f32; mov: (n, Y, X) u16.
    Returns a float64 (n, 2) device tensor."""
    _need_cuda(ref, mov)
    _u16(mov)
    kind = {torch.uint16: 0, torch.int32: 1, torch.float32: 2}.get(ref.dtype)
    if kind is None:
        raise TypeError("ref must be uint16, int32 (sum of planes) or float32")
    if mov.dim() == 2:
        mov = mov[None]
    n, y, x = mov.shape
    if tuple(ref.shape) != (y, x):
        raise ValueError("ref and mov plane shapes differ")
    lib = _lib.load()
    nb = lib.iap_reg_ws_bytes(y, x, 1, _reg_batch(batch, n, lambda b: lib.iap_reg_ws_bytes(y, x, 1, b)))
    if nb == 0:
        raise ValueError(f"registration needs planes of at least 64x64, got {y}x{x}")
    ws = _workspace(nb, mov.device)
    shifts = torch.empty((n, 2), dtype=torch.float64, device=mov.device)
    _lib.check(lib.iap_reg_estimate_u16(ref.data_ptr(), kind, mov.data_ptr(), y * x, n, y, x,
                                        _reg_flags(upsample, normalize), shifts.data_ptr(), ws.data_ptr(), ws.numel(),
                                        _stream()), "iap_reg_estimate_u16")
    return shifts


def estimate_pair_shifts(ref: torch.Tensor, mov: torch.Tensor, upsample: bool = True, normalize: bool = False,
                         batch: int | None = None) -> torch.Tensor:
    """(dy, dx) for n independent pairs: ref (n, Y, X) u16 against mov (n, Y, X) u16, plane by plane.  Both may
    be strided views along n (planes themselves contiguous).  Returns float64 (n, 2) on the device."""
    for a in (ref, mov):          # strided along n is fine here, so no _need_cuda (it insists on contiguity)
        if not a.is_cuda:
            raise _lib.IapError("imageanalysis_pipeline_amd.ops needs device tensors: the HIP library is the "
                                "only implementation (no CPU fallback)")
    _u16(ref)
    _u16(mov)
    if ref.shape != mov.shape or ref.dim() != 3:
        raise ValueError("ref and mov must both be (n, Y, X)")
    n, y, x = mov.shape
    for a in (ref, mov):
        if a.stride(2) != 1 or a.stride(1) != x or (n > 1 and a.stride(0) < y * x):
            raise ValueError("planes must be contiguous (a stride along n is allowed)")
    lib = _lib.load()
    nb = lib.iap_reg_pairs_ws_bytes(y, x, _reg_batch(batch, n, lambda b: lib.iap_reg_pairs_ws_bytes(y, x, b)))
    if nb == 0:
        raise ValueError(f"registration needs planes of at least 64x64, got {y}x{x}")
    ws = _workspace(nb, mov.device)
    shifts = torch.empty((n, 2), dtype=torch.float64, device=mov.device)
    _lib.check(lib.iap_reg_estimate_pairs_u16(ref.data_ptr(), ref.stride(0) if n > 1 else y * x, mov.data_ptr(),
                                              mov.stride(0) if n > 1 else y * x, n, y, x,
                                              _reg_flags(upsample, normalize), shifts.data_ptr(), ws.data_ptr(),
                                              ws.numel(), _stream()), "iap_reg_estimate_pairs_u16")
    return shifts


def chan_shift(stack: torch.Tensor, reg_channel: int, upsample: bool = True):
    """The channel-shift pass of utility.py:313-330 on a resident stack u16 (T, C, Z, Y, X): every plane of
    every other channel is registered against the same frame's and z's plane of ``reg_channel`` and warped
    (``sr.register_transform(ref_plane, plane)``); the registration channel is copied.  Returns
    (shifted stack, shifts f64 (T, C, Z, 2)).  Opt-in: the reference overwrites this pass's files in its
    reg_ref pass, so the shim's im_reg does not run it."""
    _need_cuda(stack)
    _u16(stack)
    if stack.dim() != 5:
        raise ValueError("stack must be (T, C, Z, Y, X)")
    stack = stack.contiguous()
    t, c, z, y, x = stack.shape
    if not 0 <= reg_channel < c:
        raise ValueError(f"reg_channel {reg_channel} not in 0..{c - 1}")
    shifts = torch.zeros((t, c, z, 2), dtype=torch.float64, device=stack.device)
    for ch in range(c):
        if ch == reg_channel:
            continue
        if z == 1:      # one strided run over the frames
            shifts[:, ch, 0] = estimate_pair_shifts(stack[:, reg_channel, 0], stack[:, ch, 0], upsample=upsample)
        else:           # the z planes of a frame are contiguous; frames are not evenly spaced with them
            for f in range(t):
                shifts[f, ch] = estimate_pair_shifts(stack[f, reg_channel], stack[f, ch], upsample=upsample)
    out = apply_shifts(stack.view(t * c * z, 1, y, x), shifts.view(t * c * z, 2).contiguous())
    return out.view(t, c, z, y, x), shifts


def apply_shifts(stack: torch.Tensor, shifts: torch.Tensor, sub2: torch.Tensor | None = None) -> torch.Tensor:
    """StackReg.transform + clip + astype(uint16) (utility.py:351-355): stack u16 (T, ..., Y, X),
    shifts f64 (T, 2); every plane of frame f is warped by shifts[f].  ``sub2`` (int32, one per plane): the
    stack is raw and its background levels are subtracted as the planes are read (see ``im_reg``)."""
    _need_cuda(stack, shifts)
    _u16(stack)
    if shifts.dtype != torch.float64 or shifts.shape != (stack.shape[0], 2):
        raise ValueError("shifts must be float64 of shape (T, 2)")
    t = stack.shape[0]
    y, x = stack.shape[-2:]
    ppf = stack.numel() // (t * y * x)
    out = torch.empty_like(stack)
    lib = _lib.load()
    if sub2 is not None:
        _need_cuda(sub2)
        if sub2.dtype != torch.int32 or sub2.numel() != t * ppf:
            raise ValueError("sub2 must be int32 with one value per plane")
        _lib.check(lib.iap_apply_shift_sub_u16(stack.data_ptr(), out.data_ptr(), t, ppf, y, x, shifts.data_ptr(),
                                               sub2.data_ptr(), _stream()), "iap_apply_shift_sub_u16")
        return out
    _lib.check(lib.iap_apply_shift_u16(stack.data_ptr(), out.data_ptr(), t, ppf, y, x, shifts.data_ptr(), _stream()),
               "iap_apply_shift_u16")
    return out


def im_reg(stack: torch.Tensor, reg_channel: int = 0, reg_ref: str = "mean", upsample: bool = True,
           normalize: bool = False, batch: int | None = None, out: torch.Tensor | None = None,
           sub2: torch.Tensor | None = None):
    """utility.py:274-430 (translation) on a resident stack u16 (T, C, Z, Y, X).
    Returns (registered stack, shifts f64 (T, 2)).  ``sub2`` (int32, one per plane: ``bg_smo_levels``'s med2):
    ``stack`` is the raw stack and the background is subtracted as the kernels read it; the result equals
    ``im_reg(bg_smo(stack)[0])`` bit for bit, without the pass that writes the subtracted stack."""
    _need_cuda(stack)
    _u16(stack)
    if stack.dim() != 5:
        raise ValueError("stack must be (T, C, Z, Y, X)")
    if reg_ref not in REF_MODES:
        raise ValueError(f"{reg_ref} is not valid. Please only put {list(REF_MODES)}")
    t, c, z, y, x = stack.shape
    lib = _lib.load()
    nb = lib.iap_reg_ws_bytes(y, x, z, _reg_batch(batch, t, lambda b: lib.iap_reg_ws_bytes(y, x, z, b)))
    if nb == 0:
        raise ValueError(f"registration needs planes of at least 64x64, got {y}x{x}")
    ws = _workspace(nb, stack.device)
    if out is None:
        out = torch.empty_like(stack)
    shifts = torch.zeros((t, 2), dtype=torch.float64, device=stack.device)
    if sub2 is not None:
        _need_cuda(sub2)
        if sub2.dtype != torch.int32 or sub2.numel() != t * c * z:
            raise ValueError("sub2 must be int32 with one value per plane (T*C*Z)")
        _lib.check(lib.iap_im_reg_sub_u16(stack.data_ptr(), out.data_ptr(), t, c, z, y, x, int(reg_channel),
                                          REF_MODES[reg_ref], _reg_flags(upsample, normalize), sub2.data_ptr(),
                                          shifts.data_ptr(), ws.data_ptr(), nb, _stream()),
                   "iap_im_reg_sub_u16")
        return out, shifts
    _lib.check(lib.iap_im_reg_u16(stack.data_ptr(), out.data_ptr(), t, c, z, y, x, int(reg_channel),
                                  REF_MODES[reg_ref], _reg_flags(upsample, normalize), shifts.data_ptr(),
                                  ws.data_ptr(), nb, _stream()), "iap_im_reg_u16")
    return out, shifts


# ---------------------------------------------------------------- the whole path on one resident FOV
def hot_path(stack: torch.Tensor, labels: torch.Tensor, threshold: float, reg_channel: int = 0, reg_ref: str = "mean",
             upsample: bool = True, batch: int | None = None, out: torch.Tensor | None = None,
             tblock: int | None = None, n_labels: int | None = None, sigma: float = 0.0, size: int = 7):
    """core.py:99-112 + 328-330 on a resident FOV: smo_bg_sub -> im_reg -> per-label tables.  ``stack`` u16
    (T, C, Z, Y, X) raw, ``labels`` u16 (T, Z, Y, X).  Returns (tables, shifts, levels): tables as
    ``measure_finalize``, shifts f64 (T, 2), levels int32 (T*C*Z,) = 2 x background of every plane.

    The background levels are subtracted as the registration kernels read the raw stack (the bits of
    ``bg_smo`` followed by ``im_reg``).  ``tblock``: the registered stack is produced and measured ``tblock`` frames
    at a time into one reusable buffer (for stacks whose registered copy does not fit beside them, e.g. a
    2048 x 2048 x 31 z x 2 ch x 200 frame stack); the estimate then runs on the z-MIPs of the lowered registration
    channel, which are the planes ``im_reg`` estimates on, so shifts and tables equal the whole-stack call.
    ``reg_ref='previous'`` needs the registered previous frame and is not offered block-wise."""
    _need_cuda(stack, labels)
    t, c, z, y, x = stack.shape
    levels, _ = bg_smo_levels(stack, threshold, sigma=sigma, size=size)
    if not tblock or tblock >= t:
        reg, shifts = im_reg(stack, reg_channel, reg_ref, upsample=upsample, batch=batch, out=out, sub2=levels)
        return measure_finalize(measure_accumulate(reg, labels, n_labels)), shifts, levels
    if reg_ref == "previous":
        raise ValueError("time blocks cannot run reg_ref='previous' (serial through the registered frames)")
    planes = lowered_channel_mip(stack, reg_channel, levels)                       # (T, Y, X)
    scratch = torch.empty_like(planes)
    _, shifts = im_reg(planes.view(t, 1, 1, y, x), 0, reg_ref, upsample=upsample, batch=batch,
                       out=scratch.view(t, 1, 1, y, x))
    del scratch
    lv = levels.view(t, c * z)
    if out is not None:
        # the block's warp writes min(tblock, T) x C x Z x Y x X pixels through out's pointer: refuse what cannot hold them
        if (not out.is_cuda or out.device != stack.device or out.dtype != torch.uint16 or not out.is_contiguous()
                or out.dim() != 5 or tuple(out.shape[1:]) != tuple(stack.shape[1:]) or out.shape[0] < min(tblock, t)):
            raise ValueError("hot_path(tblock=...): out must be a contiguous uint16 tensor on the stack's device, "
                             f"shaped (>= {min(tblock, t)},) + {tuple(stack.shape[1:])}")
    buf = out if out is not None else torch.empty((tblock,) + tuple(stack.shape[1:]), dtype=torch.uint16, device=stack.device)
    lib = _lib.load()

    def blocks(L, fmax):
        acc = torch.empty((t, L, 4 + c), dtype=torch.int64, device=stack.device)
        for t0 in range(0, t, tblock):
            t1 = min(t, t0 + tblock)
            _lib.check(lib.iap_apply_shift_sub_u16(stack[t0:t1].data_ptr(), buf.data_ptr(), t1 - t0, c * z, y, x,
                                                   shifts[t0:t1].data_ptr(), lv[t0:t1].data_ptr(), _stream()),
                       "iap_apply_shift_sub_u16")
            _measure_call(buf[: t1 - t0], labels[t0:t1], L, acc[t0:t1], None if fmax is None else fmax[t0:t1])
        return acc

    if n_labels is not None:
        return measure_finalize(blocks(int(n_labels), None)), shifts, levels
    fmax = torch.empty(t, dtype=torch.int32, device=stack.device)
    L0 = provisional_labels(t, c)
    acc = blocks(L0, fmax)
    L = int(fmax.max().item()) + 1
    # more ids than the provisional table holds: the blocks are warped and measured once more (as rare as such masks)
    acc = blocks(L, None) if L > L0 else acc[:, :L].contiguous()
    return measure_finalize(acc), shifts, levels


# ---------------------------------------------------------------- distance map / pixel table
def edt_sq(mask: torch.Tensor) -> torch.Tensor:
    """Squared Euclidean distance (int32, exact) of every pixel to the nearest non-zero pixel of ``mask``
    (..., Y, X) u16: the square of ``mahotas.distance(~mask, 'euclidean')`` of utility.py:871-876."""
    _need_cuda(mask)
    _u16(mask)
    n, y, x = _planes(mask)
    lib = _lib.load()
    ws = _workspace(lib.iap_edt_ws_bytes(n, y, x), mask.device)
    out = torch.empty(mask.shape, dtype=torch.int32, device=mask.device)
    _lib.check(lib.iap_edt_sq_u16(mask.data_ptr(), out.data_ptr(), n, y, x, ws.data_ptr(), ws.numel(), _stream()),
               "iap_edt_sq_u16")
    return out


def masked_pixels(cell_mask: torch.Tensor, img: torch.Tensor, sq: torch.Tensor | None = None, pix_size: float = 1.0):
    """analysis.py:276-283 for T frames: ``cell_mask`` (T, Y, X) u16, ``img`` (T, C, Y, X) u16, ``sq`` (T, Y, X)
    int32 from :func:`edt_sq` (optional).  Returns (counts int64 (T,), dmap int32 (total,), values u16 (C, total));
    the pixels of frame f occupy [counts[:f].sum(), counts[:f+1].sum()) in numpy boolean-indexing order."""
    _need_cuda(cell_mask, img)
    _u16(cell_mask)
    _u16(img)
    if cell_mask.ndim != 3 or img.ndim != 4 or img.shape[0] != cell_mask.shape[0] or img.shape[2:] != cell_mask.shape[1:]:
        raise ValueError("masked_pixels: cell_mask (T, Y, X) and img (T, C, Y, X) must agree")
    t, c, y, x = img.shape
    if sq is not None:
        _need_cuda(sq)
        if sq.dtype != torch.int32 or sq.shape != cell_mask.shape:
            raise ValueError("masked_pixels: sq must be int32 with the shape of cell_mask")
    lib = _lib.load()
    dev = cell_mask.device
    nch = lib.iap_mask_chunks(y, x)
    offs = torch.empty((t, nch), dtype=torch.int32, device=dev)
    totals = torch.empty(t, dtype=torch.int64, device=dev)
    _lib.check(lib.iap_mask_count_u16(cell_mask.data_ptr(), t, y, x, offs.data_ptr(), totals.data_ptr(), _stream()),
               "iap_mask_count_u16")
    base = torch.cumsum(totals, 0) - totals
    total = int(totals.sum().item())
    dmap = torch.empty(total, dtype=torch.int32, device=dev)
    vals = torch.empty((c, total), dtype=torch.uint16, device=dev)
    _lib.check(lib.iap_masked_gather_u16(cell_mask.data_ptr(), sq.data_ptr() if sq is not None else None, img.data_ptr(),
                                         c * y * x, y * x, c, t, y, x, offs.data_ptr(), base.data_ptr(), float(pix_size),
                                         total, dmap.data_ptr(), vals.data_ptr(), _stream()), "iap_masked_gather_u16")
    return totals, dmap, vals


# ---------------------------------------------------------------- mask stitching
def label_overlap(a: torch.Tensor, b: torch.Tensor, na_max: int | None = None, nb_max: int | None = None) -> torch.Tensor:
    """cellpose.metrics._label_overlap(a, b) for one pair of u16 label arrays of equal shape:
    int64 (na_max+1, nb_max+1) with overlap[i, j] = #{a == i and b == j}."""
    _need_cuda(a, b)
    _u16(a)
    _u16(b)
    if a.shape != b.shape:
        raise ValueError("label_overlap: shapes differ")
    na = max_label(a) if na_max is None else int(na_max)
    nb = max_label(b) if nb_max is None else int(nb_max)
    out = torch.empty((na + 1, nb + 1), dtype=torch.int32, device=a.device)
    _lib.check(_lib.load().iap_label_overlap_u16(a.data_ptr(), b.data_ptr(), 1, a.numel(), na, nb, out.data_ptr(),
                                                 _stream()), "iap_label_overlap_u16")
    return out.to(torch.int64)


def iou_from_overlap(overlap) -> np.ndarray:
    """cellpose.metrics._intersection_over_union from the overlap matrix (host, tiny)."""
    ov = overlap.cpu().numpy().astype(np.float64) if isinstance(overlap, torch.Tensor) else np.asarray(overlap, np.float64)
    n_pred = ov.sum(axis=0, keepdims=True)
    n_true = ov.sum(axis=1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = ov / (n_pred + n_true - ov)
    iou[np.isnan(iou)] = 0.0
    return iou


def relabel(masks: torch.Tensor, lut: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """out = lut[masks] for u16 labels; ``lut`` (L,) for all planes or (n, L) per leading plane of ``masks``.
    Every label present must be < L (checked)."""
    _need_cuda(masks, lut)
    _u16(masks)
    _u16(lut)
    if out is None:
        out = torch.empty_like(masks)
    if lut.ndim == 1:
        n, px, stride = 1, masks.numel(), 0
    else:
        n, px, stride = masks.shape[0], masks[0].numel(), lut.shape[1]
        if lut.shape[0] != n:
            raise ValueError("relabel: one table per leading plane expected")
    if max_label(masks) >= lut.shape[-1]:
        raise ValueError("relabel: a label exceeds the lookup table")
    _lib.check(_lib.load().iap_relabel_u16(masks.data_ptr(), out.data_ptr(), n, px, lut.data_ptr(), stride, _stream()),
               "iap_relabel_u16")
    return out


def _column_modes(cols: np.ndarray) -> np.ndarray:
    """Most frequent non-zero value of every column of ``cols`` (T, K) u16, ties to the smaller value, 0 for a column
    of zeros: all columns at once (sorted runs, their lengths, first longest run)."""
    t, k = cols.shape
    v = np.sort(cols, axis=0)
    start = np.ones((t, k), dtype=bool)
    start[1:] = v[1:] != v[:-1]
    run = np.cumsum(start, axis=0) - 1                       # run number of every element within its column
    col = np.broadcast_to(np.arange(k), (t, k))
    length = np.zeros((t, k), dtype=np.int32)
    np.add.at(length, (run, col), 1)
    value = np.zeros((t, k), dtype=np.uint16)
    value[run, col] = v
    length[value == 0] = 0                                   # the background run does not compete
    best = length.argmax(axis=0)                             # runs are in ascending value order: ties -> smaller label
    return value[best, np.arange(k)]


def label_mode(masks: torch.Tensor) -> torch.Tensor:
    """Master mask of utility.py:707-711: per pixel the most frequent non-zero label over the T leading
    planes of ``masks`` (ties: smaller label; 0 if never labelled)."""
    _need_cuda(masks)
    _u16(masks)
    t = masks.shape[0]
    px = masks[0].numel()
    out = torch.empty(masks.shape[1:], dtype=torch.uint16, device=masks.device)
    nov = torch.zeros(1, dtype=torch.int32, device=masks.device)
    _lib.check(_lib.load().iap_label_mode_u16(masks.data_ptr(), t, px, out.data_ptr(), nov.data_ptr(), _stream()),
               "iap_label_mode_u16")
    if int(nov.item()):  # pixels that saw more than 8 distinct labels: recount those few on the host
        flat = out.view(-1).view(torch.int16)       # torch indexes int16, not uint16
        idx = torch.nonzero(flat == -1).squeeze(1)
        cols = masks.reshape(t, px).view(torch.int16)[:, idx].cpu().numpy().view(np.uint16)
        flat[idx] = torch.from_numpy(_column_modes(cols).view(np.int16)).to(masks.device)
    return out


# ---------------------------------------------------------------- compartment masks
def _label_morph(name: str, masks: torch.Tensor, radius: int, iterations: int) -> torch.Tensor:
    _need_cuda(masks)
    _u16(masks)
    n, y, x = _planes(masks)
    out = torch.empty_like(masks)
    nbytes = masks.numel() * 2 if iterations > 1 else 0
    ws = _workspace(nbytes, masks.device) if nbytes else None
    fn = getattr(_lib.load(), name)
    _lib.check(fn(masks.data_ptr(), out.data_ptr(), n, y, x, int(radius), int(iterations),
                  ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream()), name)
    return out


def label_erode(masks: torch.Tensor, radius: int, iterations: int = 1) -> torch.Tensor:
    """utility.py:776-801 on u16 label planes (..., Y, X): label-preserving erosion with skimage's disk(radius)."""
    return _label_morph("iap_label_erode_u16", masks, radius, iterations)


def label_dilate(masks: torch.Tensor, radius: int, iterations: int = 1) -> torch.Tensor:
    """utility.py:804-826 on u16 label planes (..., Y, X): grey dilation (largest label) with disk(radius)."""
    return _label_morph("iap_label_dilate_u16", masks, radius, iterations)


def label_sub(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """experiments.py:259: a - b element-wise on u16 label stacks of one shape (membrane = full - eroded)."""
    _need_cuda(a, b)
    _u16(a)
    _u16(b)
    if a.shape != b.shape:
        raise ValueError("label_sub: shapes differ")
    a, b = a.contiguous(), b.contiguous()
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.load().iap_label_sub_u16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()),
               "iap_label_sub_u16")
    return out


def label_areas(masks: torch.Tensor, n_labels: int | None = None) -> torch.Tensor:
    """Pixel count per (frame, label) of a u16 label stack (T, [Z,] Y, X) -> int64 (T, L)
    (experiments.py:200-201: ``np.unique(t, return_counts=True)`` of one object at a time).  The label
    stack doubles as the one-channel image of the measuring kernel; only its counts are kept."""
    _need_cuda(masks)
    _u16(masks)
    if masks.dim() == 3:
        masks = masks[:, None]
    if masks.dim() != 4:
        raise ValueError("label_areas: masks must be (T, Y, X) or (T, Z, Y, X)")
    masks = masks.contiguous()
    return measure_accumulate(masks[:, None], masks, n_labels)[..., 0].contiguous()
