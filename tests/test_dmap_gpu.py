"""Distance map of the reference mask and the per-pixel table of masked pixels (SURVEY 8f rank 2;
Utility.apply_dmap utility.py:871-876, Analysis.pixel_distance analysis.py:276-283) through the C-ABI
against the oracle (pinned to scipy's exact EDT by tests/golden/edt_scipy.npz).  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from imageanalysis_pipeline_amd import ops
from oracle import ref_path as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_edt_matches_golden_and_oracle(dev):
    g = np.load(os.path.join(GOLDEN, "edt_scipy.npz"))
    sq = ops.edt_sq(ops.to_device_u16(g["masks"], dev)).cpu().numpy()
    assert np.array_equal(np.sqrt(sq.astype(np.float64)), g["dist"])
    rng = np.random.default_rng(0)
    m = np.zeros((2, 97, 131), np.uint16)                 # odd sizes, ragged last block
    m[0][rng.random((97, 131)) < 0.003] = 1
    m[1, 50:60, 100:131] = 2
    got = ops.edt_sq(ops.to_device_u16(m, dev)).cpu().numpy()
    for k in range(2):
        assert np.array_equal(got[k], O.edt_sq(m[k]))


def test_edt_full_size_properties(dev):
    """2048 x 2048: a single seed pixel gives the analytic map; a half-plane gives column distances."""
    y = x = 2048
    m = np.zeros((2, y, x), np.uint16)
    m[0, 700, 1300] = 1
    m[1, :, :512] = 1
    sq = ops.edt_sq(ops.to_device_u16(m, dev)).cpu().numpy()
    yy, xx = np.mgrid[0:y, 0:x]
    assert np.array_equal(sq[0], (yy - 700) ** 2 + (xx - 1300) ** 2)
    assert np.array_equal(sq[1], np.broadcast_to(np.maximum(np.arange(x) - 511, 0) ** 2, (y, x)))


def test_pixel_table_matches_oracle(dev):
    rng = np.random.default_rng(1)
    t, c, y, x = 3, 2, 120, 150
    ref = np.zeros((t, y, x), np.uint16)
    ref[:, 40:70, 60:90] = 1
    ref[2] = 0
    ref[2, 5, 5] = 1
    cell = (rng.random((t, y, x)) < 0.2).astype(np.uint16) * rng.integers(1, 400, size=(t, y, x)).astype(np.uint16)
    cell[1] = 0                                              # an empty frame
    img = rng.integers(0, 65536, size=(t, c, y, x), dtype=np.uint16)
    sq = ops.edt_sq(ops.to_device_u16(ref, dev))
    counts, dmap, vals = ops.masked_pixels(ops.to_device_u16(cell, dev), ops.to_device_u16(img, dev), sq, pix_size=0.65)
    counts = counts.cpu().numpy()
    dmap = dmap.cpu().numpy()
    vals = ops.to_numpy_u16(vals)
    pos = 0
    for f in range(t):
        ed, ev = O.pixel_table(cell[f], img[f], ref[f], 0.65)
        assert counts[f] == len(ed)
        assert np.array_equal(dmap[pos:pos + len(ed)], ed)
        assert np.array_equal(vals[:, pos:pos + len(ed)].T, ev)
        pos += len(ed)
    assert pos == len(dmap)


def test_masked_pixels_without_distance_and_all_selected(dev):
    t, c, y, x = 1, 1, 64, 96
    img = np.arange(y * x, dtype=np.uint16).reshape(t, c, y, x)
    cell = np.ones((t, y, x), np.uint16)
    counts, dmap, vals = ops.masked_pixels(ops.to_device_u16(cell, dev), ops.to_device_u16(img, dev))
    assert int(counts[0]) == y * x and not dmap.cpu().numpy().any()
    assert np.array_equal(ops.to_numpy_u16(vals)[0], img.reshape(-1))
    with pytest.raises(ValueError):
        ops.masked_pixels(ops.to_device_u16(cell, dev), ops.to_device_u16(img[:, :, :32], dev))
