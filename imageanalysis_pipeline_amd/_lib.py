"""ctypes binding of libiap_hip.so (the C-ABI declared in include/iap.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``.  There is
no CPU fallback: a missing library raises at first use.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IAP_LIB") or os.path.join(_HERE, "libiap_hip.so")

_vp, _i32, _i64, _f32, _f64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/iap.h one to one
SIGNATURES = {
    "iap_version": (C.c_char_p, []),
    "iap_last_error": (C.c_char_p, []),
    "iap_smo_image_ws_bytes": (_sz, [_i64, _i32, _i32, _f32]),
    "iap_smo_image": (_i32, [_vp, _i32, _vp, _i64, _i32, _i32, _f32, _i32, _vp, _sz, _vp]),
    "iap_bg_smo_ws_bytes": (_sz, [_i64, _i32, _i32, _f32]),
    "iap_bg_smo_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _f32, _i32, _f32, _vp, _vp, _vp, _sz, _vp]),
    "iap_bg_const_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "iap_bg_const_static_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "iap_gauss_blur_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _f64, _vp]),
    "iap_mip_u16": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "iap_mip_sub_u16": (_i32, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "iap_reg_ws_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "iap_reg_estimate_u16": (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "iap_reg_pairs_ws_bytes": (_sz, [_i32, _i32, _i32]),
    "iap_reg_estimate_pairs_u16": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "iap_apply_shift_u16": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "iap_apply_shift_sub_u16": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "iap_im_reg_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "iap_im_reg_sub_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "iap_measure_labels_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "iap_measure_labels_max_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "iap_measure_finalize": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "iap_max_u16": (_i32, [_vp, _i64, _vp, _vp]),
    "iap_edt_ws_bytes": (_sz, [_i64, _i32, _i32]),
    "iap_edt_sq_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _sz, _vp]),
    "iap_mask_chunks": (_i32, [_i32, _i32]),
    "iap_mask_count_u16": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "iap_masked_gather_u16": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i64, _i32, _i32, _vp, _vp, C.c_double, _i64,
                                     _vp, _vp, _vp]),
    "iap_label_overlap_u16": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "iap_relabel_u16": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _vp]),
    "iap_label_mode_u16": (_i32, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "iap_label_erode_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "iap_label_dilate_u16": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "iap_label_sub_u16": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "iap_tiff_probe": (_i32, [C.c_char_p] + [C.POINTER(_i32)] * 6),
    "iap_tiff_read_u16": (_i32, [C.POINTER(C.c_char_p), _i64, _vp, _i32, _i32, _i32]),
    "iap_tiff_write_u16": (_i32, [C.POINTER(C.c_char_p), _i64, _vp, _i32, _i32, _i32]),
    "iap_prof_enable": (None, [_i32]),
    "iap_prof_select": (None, [_i32]),
    "iap_prof_kernel_count": (_i32, []),
    "iap_prof_kernel_name": (C.c_char_p, [_i32]),
    "iap_prof_collect": (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(_i64)]),
}

_lib = None


class IapError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library once; raise loudly when it is missing (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IapError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C imageanalysis_pipeline_amd/csrc`). The HIP library is the only implementation.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if include/iap.h and the .so drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().iap_last_error().decode(errors="replace")
        if rc < 0:
            raise ValueError(f"{what}: {msg} (code {rc})")
        raise IapError(f"{what}: HIP error {rc}: {msg}")
