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


def measure_accumulate(img: torch.Tensor, labels: torch.Tensor, n_labels: int | None = None) -> torch.Tensor:
    """img u16 (T,C,Z,Y,X), labels u16 (T,Z,Y,X) -> exact integer accumulators
    int64 (T, L, 4+C): count, sum_z, sum_y, sum_x, sum_I[c]  (analysis.py:80-98,164-173)."""
    _need_cuda(img, labels)
    _u16(img)
    _u16(labels)
    if img.dim() != 5 or labels.dim() != 4:
        raise ValueError("img must be (T,C,Z,Y,X) and labels (T,Z,Y,X)")
    t, c, z, y, x = img.shape
    if tuple(labels.shape) != (t, z, y, x):
        raise ValueError(f"labels shape {tuple(labels.shape)} does not match image stack {(t, z, y, x)}")
    L = int(n_labels) if n_labels is not None else max_label(labels) + 1
    acc = torch.empty((t, L, 4 + c), dtype=torch.int64, device=img.device)
    lib = _lib.load()
    _lib.check(lib.iap_measure_labels_u16(img.data_ptr(), labels.data_ptr(), t, c, z, y, x, L, acc.data_ptr(),
                                          _stream()), "iap_measure_labels_u16")
    return acc


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

    Returns a list (one dict per frame) of numpy arrays: ids, area, mean (n,C), centroid
    (n,ndim) and centroid_rne (n,ndim) with axis order (z,) y, x."""
    acc = measure_accumulate(img, labels, n_labels)
    area, mean, cen, cen_r = (a.cpu().numpy() for a in measure_finalize(acc))
    z = img.shape[2]
    out = []
    for f in range(img.shape[0]):
        present = np.nonzero(area[f])[0]
        ids = present[1:]
        sl = slice(0, 3) if z > 1 else slice(1, 3)
        out.append({"ids": ids.astype(np.int64), "area": area[f, ids], "mean": mean[f, ids],
                    "centroid": cen[f, ids][:, sl], "centroid_rne": cen_r[f, ids][:, sl]})
    return out
