"""The C-ABI's refusals, called through ctypes as a foreign binding would: every entry point of the hot path with one
argument at a time made invalid (null pointer, non-positive size, pointer off its alignment, workspace one byte short,
enum out of range, size outside what the kernels are built for).  Each call must come back with the documented
negative code and a message in ``iap_last_error`` -- before anything is launched -- and the library must work
afterwards.  (include/iap.h lines 28-32: IAP_E_ARG -1, IAP_E_ALIGN -2, IAP_E_WS -3, IAP_E_UNSUPPORTED -4.)
"""
import numpy as np
import pytest
import torch

from imageanalysis_pipeline_amd import _lib, ops

pytestmark = pytest.mark.gpu

E_ARG, E_ALIGN, E_WS, E_UNSUP = -1, -2, -3, -4
T, C, Y, X = 3, 2, 64, 128


@pytest.fixture(scope="module")
def bufs(dev):
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(3)
    b = {
        "img": torch.randint(300, 900, (T, C, 1, Y, X), device=dev, generator=g, dtype=torch.int32).to(torch.uint16),
        "lab": torch.zeros((T, 1, Y, X), dtype=torch.uint16, device=dev),
        "shifts": torch.zeros((T, 2), dtype=torch.float64, device=dev),
        "i32": torch.zeros(T * C, dtype=torch.int32, device=dev),
        "f32": torch.zeros((T * C, Y, X), dtype=torch.float32, device=dev),
        "acc": torch.zeros((T, 8, 4 + C), dtype=torch.int64, device=dev),
        "tab": torch.zeros((T, 8, 4), dtype=torch.float64, device=dev),
        "lut": torch.zeros(16, dtype=torch.uint16, device=dev),
    }
    b["out"] = torch.empty_like(b["img"])
    b["ws"] = torch.empty(max(lib.iap_reg_ws_bytes(Y, X, 1, T), lib.iap_bg_smo_ws_bytes(T * C, Y, X, 1.5),
                              lib.iap_edt_ws_bytes(T, Y, X), T * C * Y * X * 2) + 4096, dtype=torch.uint8, device=dev)
    return lib, b


def _cases(lib, b):
    p = lambda k: b[k].data_ptr()   # noqa: E731
    n = T * C
    regws = lib.iap_reg_ws_bytes(Y, X, 1, T)
    bgws = lib.iap_bg_smo_ws_bytes(n, Y, X, 0.0)
    bgws_s = lib.iap_bg_smo_ws_bytes(n, Y, X, 1.5)
    edtws = lib.iap_edt_ws_bytes(T, Y, X)
    # (name, valid argument list, [(argument index, bad value, expected code), ...])
    return [
        ("iap_bg_smo_u16", [p("img"), p("out"), n, Y, X, 0.0, 7, 0.05, p("i32"), p("i32"), p("ws"), bgws, None],
         [(0, None, E_ARG), (2, 0, E_ARG), (2, -5, E_ARG), (3, 0, E_ARG), (4, -1, E_ARG), (6, 8, E_ARG), (6, 65, E_ARG),
          (5, -1.0, E_ARG), (0, p("img") + 2, E_ALIGN), (1, p("out") + 6, E_ALIGN), (10, None, E_ARG),
          (2, 1 << 33, E_UNSUP)]),
        ("iap_bg_smo_u16", [p("img"), p("out"), n, Y, X, 1.5, 7, 0.05, p("i32"), p("i32"), p("ws"), bgws_s, None],
         [(11, bgws_s - 1, E_WS), (11, 0, E_WS)]),
        ("iap_bg_const_u16", [p("img"), p("out"), n, Y, X, 2, 3, 10, 20, p("i32"), p("ws"), n * 16, None],
         [(0, None, E_ARG), (5, -1, E_ARG), (7, Y, E_ARG), (8, X, E_ARG), (5, 11, E_ARG), (11, n * 16 - 1, E_WS),
          (10, None, E_WS), (2, 70000, E_WS)]),   # (the workspace check comes before the 65535-plane limit)
        ("iap_gauss_blur_u16", [p("img"), p("out"), n, Y, X, 5, 5, 1.0, None],
         [(1, p("img"), E_ARG), (5, 4, E_ARG), (6, 65, E_ARG), (5, -3, E_ARG), (0, p("img") + 2, E_ALIGN),
          (2, 70000, E_UNSUP)]),
        ("iap_smo_image", [p("img"), 0, p("f32"), n, Y, X, 0.0, 7, None, 0, None],
         [(1, 2, E_ARG), (7, 6, E_ARG), (2, None, E_ARG), (6, -0.5, E_ARG)]),
        ("iap_smo_image", [p("img"), 0, p("f32"), n, Y, X, 1.5, 7, p("ws"), lib.iap_smo_image_ws_bytes(n, Y, X, 1.5), None],
         [(8, None, E_WS), (9, 16, E_WS)]),
        ("iap_mip_u16", [p("img"), Y * X, p("out"), n, 1, Y, X, None],
         [(0, None, E_ARG), (3, 0, E_ARG), (4, 0, E_ARG)]),
        ("iap_reg_estimate_u16", [p("img"), 0, p("img"), C * Y * X, T, Y, X, 1, p("shifts"), p("ws"), regws, None],
         [(0, None, E_ARG), (2, None, E_ARG), (1, 3, E_ARG), (4, 0, E_ARG), (5, 32, E_UNSUP), (6, 48, E_UNSUP),
          (10, 1024, E_WS), (9, None, E_ARG)]),
        ("iap_apply_shift_u16", [p("img"), p("out"), T, C, Y, X, p("shifts"), None],
         [(0, None, E_ARG), (1, p("img"), E_ARG), (6, None, E_ARG), (2, 0, E_ARG), (3, 0, E_ARG), (4, 0, E_ARG)]),
        ("iap_im_reg_u16", [p("img"), p("out"), T, C, 1, Y, X, 0, 1, 1, p("shifts"), p("ws"), regws, None],
         [(0, None, E_ARG), (1, p("img"), E_ARG), (7, C, E_ARG), (7, -1, E_ARG), (8, 3, E_ARG), (8, -1, E_ARG),
          (2, 0, E_ARG), (5, 40, E_UNSUP), (12, 64, E_WS), (2, 70000, E_UNSUP)]),
        ("iap_measure_labels_u16", [p("img"), p("lab"), T, C, 1, Y, X, 8, p("acc"), None],
         [(0, None, E_ARG), (1, None, E_ARG), (8, None, E_ARG), (7, 0, E_ARG), (7, 70000, E_ARG), (2, 70000, E_UNSUP),
          (0, p("img") + 2, E_ALIGN), (8, p("acc") + 4, E_ALIGN)]),
        ("iap_measure_finalize", [p("acc"), T, 8, C, p("tab"), p("tab"), p("tab"), p("tab"), None],
         [(0, None, E_ARG), (4, None, E_ARG), (1, 0, E_ARG), (2, 0, E_ARG), (3, 0, E_ARG)]),
        ("iap_max_u16", [p("lab"), T * Y * X, p("i32"), None],
         [(0, None, E_ARG), (2, None, E_ARG), (1, 0, E_ARG), (0, p("lab") + 1, E_ALIGN)]),
        ("iap_edt_sq_u16", [p("lab"), p("f32"), T, Y, X, p("ws"), edtws, None],
         [(0, None, E_ARG), (2, 0, E_ARG), (3, 5000, E_UNSUP), (6, edtws - 1, E_WS), (5, None, E_WS)]),
        ("iap_relabel_u16", [p("lab"), p("out"), 1, T * Y * X, p("lut"), 0, None],
         [(0, None, E_ARG), (4, None, E_ARG), (3, 0, E_ARG), (5, -1, E_ARG), (2, 70000, E_UNSUP)]),
        ("iap_label_erode_u16", [p("lab"), p("out"), T, Y, X, 2, 1, None, 0, None],
         [(1, p("lab"), E_ARG), (5, 32, E_UNSUP), (5, -1, E_UNSUP), (6, 0, E_ARG), (2, 0, E_ARG)]),
        ("iap_label_dilate_u16", [p("lab"), p("out"), T, Y, X, 2, 2, p("ws"), T * Y * X * 2, None],
         [(8, T * Y * X * 2 - 1, E_WS), (7, None, E_WS), (7, p("lab"), E_WS)]),
    ]


def test_every_refusal_names_its_reason_and_launches_nothing(bufs):
    lib, b = bufs
    before = b["out"].clone()
    seen, wrong = 0, []
    for name, good, bad in _cases(lib, b):
        fn = getattr(lib, name)
        for idx, value, code in bad:
            args = list(good)
            args[idx] = value
            rc = fn(*args)
            msg = lib.iap_last_error().decode(errors="replace")
            if rc != code or not msg.startswith("iap_") or len(msg) < 20:
                wrong.append((name, idx, value, "expected", code, "got", rc, msg))
            seen += 1
    torch.cuda.synchronize()
    assert not wrong, wrong
    assert seen > 90
    assert torch.equal(before.view(torch.int16), b["out"].view(torch.int16)), "a refused call wrote to its output"


def test_the_valid_calls_of_the_table_succeed(bufs):
    """the same argument lists with nothing replaced are accepted (the refusals above are not side effects of a bad
    baseline), and the library still computes after a run of refused calls"""
    lib, b = bufs
    for name, good, _ in _cases(lib, b):
        rc = getattr(lib, name)(*good)
        assert rc == 0, (name, rc, lib.iap_last_error().decode(errors="replace"))
    torch.cuda.synchronize()
    d = b["img"]
    lv, _ = ops.bg_smo_levels(d, 0.05)
    assert int(lv.min()) > 0
    assert np.isfinite(ops.im_reg(d, 0, "first")[1].cpu().numpy()).all()
