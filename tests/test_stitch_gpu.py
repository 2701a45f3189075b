"""Mask-stitching primitives and Utility.modif_stitch3D (utility.py:700-738) through the
C-ABI against the oracle.  Integer work: bit-exact.  The Cellpose part of the algorithm is restated from its
published source (third party, absent): parity unpinned; the reference's own second round is literal."""
import numpy as np
import pytest
import torch

from imageanalysis_pipeline_amd import ops, synth
from imageanalysis_pipeline_amd.pipeline import Utility
from oracle import ref_path as O

pytestmark = pytest.mark.gpu


def _moving_stack(t, y, x, n, seed, drop=None):
    base = synth.make_labels(y, x, n, seed=seed)
    rng = np.random.default_rng(seed)
    out = []
    for f in range(t):
        perm = np.concatenate([[0], rng.permutation(np.arange(1, base.max() + 1))])
        fr = np.roll(perm[base], (f, 2 * f), (0, 1))
        if drop and f in drop:
            fr[fr == perm[drop[f]]] = 0                     # a cell missing in this frame
        out.append(fr)
    return np.stack(out).astype(np.uint16)


def test_label_overlap_and_iou(dev):
    rng = np.random.default_rng(0)
    a = synth.make_labels(130, 171, 12, seed=1).astype(np.uint16)        # px not a multiple of 8
    b = np.roll(synth.make_labels(130, 171, 9, seed=2), (3, -4), (0, 1)).astype(np.uint16)
    ov = ops.label_overlap(ops.to_device_u16(a, dev), ops.to_device_u16(b, dev))
    assert np.array_equal(ov.cpu().numpy(), O.label_overlap(a, b))
    assert np.array_equal(ops.iou_from_overlap(ov), O.iou_matrix(a, b))
    dense_a = rng.integers(0, 3000, size=(64, 64)).astype(np.uint16)     # every pixel its own pair, label 0 rare
    dense_b = rng.integers(0, 2000, size=(64, 64)).astype(np.uint16)
    ov = ops.label_overlap(ops.to_device_u16(dense_a, dev), ops.to_device_u16(dense_b, dev))
    assert np.array_equal(ov.cpu().numpy(), O.label_overlap(dense_a, dense_b))
    z = np.zeros((16, 16), np.uint16)                                     # no labels at all
    assert ops.label_overlap(ops.to_device_u16(z, dev), ops.to_device_u16(z, dev)).cpu().numpy().tolist() == [[256]]
    # labels above the stated bounds are ignored, not written out of range
    ov = ops.label_overlap(ops.to_device_u16(a, dev), ops.to_device_u16(b, dev), na_max=5, nb_max=4)
    assert np.array_equal(ov.cpu().numpy(), O.label_overlap(a, b)[:6, :5])


def test_relabel(dev):
    rng = np.random.default_rng(1)
    m = rng.integers(0, 500, size=(3, 37, 53)).astype(np.uint16)
    lut = rng.integers(0, 65536, size=500).astype(np.uint16)
    got = ops.relabel(ops.to_device_u16(m[0], dev), ops.to_device_u16(lut, dev))
    assert np.array_equal(ops.to_numpy_u16(got), lut[m[0]])
    m8 = rng.integers(0, 500, size=(3, 40, 56)).astype(np.uint16)
    luts = rng.integers(0, 65536, size=(3, 500)).astype(np.uint16)
    got = ops.relabel(ops.to_device_u16(m8, dev), ops.to_device_u16(luts, dev))
    assert np.array_equal(ops.to_numpy_u16(got), np.stack([luts[k][m8[k]] for k in range(3)]))
    with pytest.raises(ValueError):
        ops.relabel(ops.to_device_u16(m[0], dev), ops.to_device_u16(lut[:100], dev))


def test_label_mode_with_ties_and_overflow(dev):
    rng = np.random.default_rng(2)
    t, y, x = 11, 20, 24
    m = rng.integers(0, 4, size=(t, y, x)).astype(np.uint16)             # few labels: many ties
    m[:, 0, 0] = 0                                                       # never labelled
    m[:, 0, 1] = np.arange(1, t + 1)                                     # 11 distinct labels: overflow path
    m[:, 0, 2] = [7, 7, 9, 9, 5, 5, 0, 0, 0, 0, 0]                       # three-way tie -> 5
    got = ops.to_numpy_u16(ops.label_mode(ops.to_device_u16(m, dev)))
    exp = O.label_mode(m)
    assert np.array_equal(got, exp)
    assert got[0, 0] == 0 and got[0, 1] == 1 and got[0, 2] == 5


def test_modif_stitch3D_matches_oracle(dev):
    st = _moving_stack(6, 96, 128, 9, seed=3, drop={2: 4, 3: 4})
    got = Utility.modif_stitch3D(st, 0.75)
    exp = O.modif_stitch3D(st, 0.75)
    assert got.dtype == st.dtype and np.array_equal(got, exp)
    empty_first = st.copy()
    empty_first[0] = 0                                                    # the 'empty' branches of the loop
    assert np.array_equal(Utility.modif_stitch3D(empty_first, 0.75), O.modif_stitch3D(empty_first, 0.75))


def _tracks(seed=3, t=6, y=96, x=128):
    """discs that persist over time, some with an outlier area in one frame, some missing in frames"""
    rng = np.random.default_rng(seed)
    m = np.zeros((t, y, x), np.uint16)
    yy, xx = np.mgrid[:y, :x]
    for lab in range(1, 13):
        cy, cx = rng.integers(10, y - 10), rng.integers(10, x - 10)
        for f in range(t):
            if lab % 4 == 0 and f == lab % t:
                continue                                   # gap: an incomplete track
            r = 5 + (4 if (lab % 3 == 0 and f == 2) else 0)  # one swollen frame
            m[f][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = lab * 3   # non-consecutive ids
    return m


def test_filter_mask_shapes_matches_reference_loop(dev):
    from imageanalysis_pipeline_amd.pipeline import Utility
    m = _tracks()
    got = Utility.filter_mask_shapes(m, shape_threshold=0.2)
    exp = O.filter_mask_shapes(m, 0.2)
    assert got.dtype == exp.dtype and np.array_equal(got, exp)
    assert (got != m).any()                                 # the swollen frames were dropped


def test_renumber_and_trim_match_reference_loops(dev):
    from imageanalysis_pipeline_amd.pipeline import Utility
    m = _tracks(seed=5)
    assert np.array_equal(Utility.renumber_masks(m), O.renumber_masks(m))
    assert np.array_equal(Utility.trim_incomplete_tracks(m), O.trim_incomplete_tracks(m, m.shape[0]))
    nob = m.copy()
    nob[nob == 0] = 7                                       # no background: rank 0 goes to the smallest label
    assert np.array_equal(Utility.renumber_masks(nob), O.renumber_masks(nob))


def test_label_areas_are_bincounts(dev):
    from imageanalysis_pipeline_amd import ops
    m = _tracks(seed=8)
    a = ops.label_areas(ops.to_device_u16(m, dev)).cpu().numpy()
    for f in range(m.shape[0]):
        assert np.array_equal(a[f], np.bincount(m[f].ravel(), minlength=a.shape[1]))


@pytest.mark.parametrize("shape", [(61, 23), (33, 35), (20, 42)])
def test_odd_planes_take_the_scalar_paths(dev, shape):
    """frames of a stack whose planes are not a multiple of 8 pixels start off a 16-byte boundary: label range,
    overlap counting and relabelling read them one pixel at a time instead of refusing them (IAP_E_ALIGN until
    scripts/fuzz_stages.py met such a stack through modif_stitch3D)"""
    y, x = shape
    t = 4
    masks = np.stack([synth.make_labels(y, x, 6, seed=s) for s in range(t)])
    d = ops.to_device_u16(masks, dev)
    for f in range(t):
        assert ops.max_label(d[f]) == int(masks[f].max())
    ov = ops.label_overlap(d[1], d[2]).cpu().numpy()
    assert np.array_equal(ov, O.label_overlap(masks[1], masks[2]))
    lut = np.arange(int(masks.max()) + 1, dtype=np.uint16)[::-1].copy()
    got = ops.to_numpy_u16(ops.relabel(d[3], ops.to_device_u16(lut, dev)))
    assert np.array_equal(got, lut[masks[3]])
    assert np.array_equal(Utility.modif_stitch3D(masks.copy(), 0.5), O.modif_stitch3D(masks.copy(), 0.5))
