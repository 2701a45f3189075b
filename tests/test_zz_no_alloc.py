"""GPU: iap_bg_const_static_u16, the one entry point that allocates nothing (scratch in a file-scope device array of
the library), against iap_bg_const_u16 (workspace variant) through the C-ABI: bit-exact outputs and levels, on two
calls in a row, the second with a smaller input (the shared array keeps the first call's contents)."""
import numpy as np
import pytest
import torch

from imageanalysis_pipeline_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _both(dev, img, roi):
    lib = _lib.load()
    n, y, x = img.shape
    d = ops.to_device_u16(img, dev)
    st = torch.cuda.current_stream().cuda_stream
    out_ws, bg_ws = ops.bg_manual(d, roi)                                   # iap_bg_const_u16 (workspace)
    out = torch.empty_like(d)
    bg = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.iap_bg_const_static_u16(d.data_ptr(), out.data_ptr(), n, y, x, *roi, bg.data_ptr(), st),
               "iap_bg_const_static_u16")
    torch.cuda.synchronize()
    assert torch.equal(bg, bg_ws)
    assert torch.equal(out.view(torch.int16), out_ws.view(torch.int16))
    return bg.cpu().numpy()


def test_static_scratch_entry_point_matches_workspace_variant(dev):
    rng = np.random.default_rng(21)
    big = rng.integers(0, 65536, size=(37, 96, 160)).astype(np.uint16)
    b1 = _both(dev, big, (5, 7, 70, 131))
    # second call, fewer and smaller planes, another ROI: nothing of the first call's sums may leak in
    small = rng.integers(0, 3000, size=(5, 40, 72)).astype(np.uint16)
    b2 = _both(dev, small, (0, 0, 39, 71))
    assert np.array_equal(b2, np.rint(small.reshape(5, -1).astype(np.float64).mean(axis=1)).astype(np.int32))
    assert b1.shape == (37,)
    # again the larger input: the array is reusable after a smaller call
    _both(dev, big, (5, 7, 70, 131))


def test_static_scratch_entry_point_limits(dev):
    lib = _lib.load()
    d = torch.zeros((1, 8, 8), dtype=torch.uint16, device=dev)
    o = torch.empty_like(d)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.iap_bg_const_static_u16(d.data_ptr(), o.data_ptr(), 4097, 8, 8, 0, 0, 7, 7, None, st)
    assert rc == -4 and b"static scratch" in lib.iap_last_error()           # IAP_E_UNSUPPORTED, clear message
    rc = lib.iap_bg_const_static_u16(d.data_ptr(), o.data_ptr(), 1, 8, 8, 0, 0, 8, 7, None, st)
    assert rc == -1                                                         # ROI outside the plane
