"""GPU parity: iap_measure_labels_u16 / iap_measure_finalize vs the oracle (bit-exact).
Reference semantics: analysis.py:80-98 (mean), :164-173 (centroid), experiments.py:200-201 (area)."""
import numpy as np
import pytest
import torch

from oracle import ref_path as O
from imageanalysis_pipeline_amd import synth

pytestmark = pytest.mark.gpu


def _check(img, lab, dev, n_labels=None):
    """img (T,C,Z,Y,X), lab (T,Z,Y,X) numpy u16."""
    from imageanalysis_pipeline_amd import ops
    got = ops.measure_labels(ops.to_device_u16(img, dev), ops.to_device_u16(lab, dev), n_labels)
    t, c, z = img.shape[:3]
    for f in range(t):
        if z == 1:
            exp = O.measure_labels(img[f, :, 0], lab[f, 0])
        else:
            exp = O.measure_labels(img[f], lab[f])
        g = got[f]
        assert np.array_equal(g["ids"], exp["ids"])
        assert np.array_equal(g["area"], exp["area"])
        assert g["sum"].dtype == np.int64 and np.array_equal(g["sum"], exp["sum"]), "per-label intensity sums"
        assert np.array_equal(g["mean"].view(np.uint64), exp["mean"].view(np.uint64)), "mean not bit-exact"
        assert np.array_equal(g["centroid"].view(np.uint64), exp["centroid"].view(np.uint64))
        assert np.array_equal(g["centroid_rne"], exp["centroid_rne"])


def test_discs_2d_multichannel(dev):
    rng = np.random.default_rng(0)
    t, c, y, x = 3, 3, 256, 384
    img = rng.integers(0, 65536, size=(t, c, 1, y, x)).astype(np.uint16)
    lab = np.stack([synth.make_labels(y, x, 40, seed=s)[None] for s in range(t)])
    _check(img, lab, dev)


def test_literal_loop_agrees(dev):
    """the reference's own per-label loop (nanmean(where=), np.where, round) on a small case"""
    from imageanalysis_pipeline_amd import ops
    rng = np.random.default_rng(1)
    img = rng.integers(0, 65536, size=(1, 2, 1, 96, 128)).astype(np.uint16)
    lab = synth.make_labels(96, 128, 12, seed=3)[None, None]
    got = ops.measure_labels(ops.to_device_u16(img, dev), ops.to_device_u16(lab, dev))[0]
    exp = O.measure_labels_literal(img[0, :, 0], lab[0, 0])
    assert np.array_equal(got["ids"], exp["ids"])
    assert np.array_equal(got["mean"].view(np.uint64), exp["mean"].view(np.uint64))
    assert np.array_equal(got["centroid_rne"], exp["centroid_rne"])
    assert np.array_equal(got["area"], exp["area"])
    assert np.array_equal(got["sum"], exp["sum"])


def test_volumetric_labels(dev):
    rng = np.random.default_rng(2)
    t, c, z, y, x = 2, 2, 5, 64, 96
    img = rng.integers(0, 65536, size=(t, c, z, y, x)).astype(np.uint16)
    base = synth.make_labels(y, x, 10, seed=5)
    lab = np.zeros((t, z, y, x), np.uint16)
    for k in range(z):
        lab[:, k] = np.where(rng.random((y, x)) < 0.8, base, 0)
    _check(img, lab, dev)


def test_rne_ties_and_no_background(dev):
    """centroid .5 ties (round-half-even) and a frame without label 0: unique()[1:] then drops
    the smallest *label* (analysis.py:80)."""
    lab = np.zeros((1, 1, 8, 8), np.uint16)
    lab[0, 0, :, :4] = 3          # x centroid 1.5 -> 2 ; y centroid 3.5 -> 4
    lab[0, 0, :, 4:] = 7          # x centroid 5.5 -> 6
    lab[0, 0, 0:2, 4:6] = 9       # y 0.5 -> 0, x 4.5 -> 4
    img = np.arange(64, dtype=np.uint16).reshape(1, 1, 1, 8, 8) * 1000
    _check(img, lab, dev)
    from imageanalysis_pipeline_amd import ops
    got = ops.measure_labels(ops.to_device_u16(img, dev), ops.to_device_u16(lab, dev))[0]
    assert list(got["ids"]) == [7, 9]          # 3 (the smallest) was dropped, as the reference does
    assert got["centroid_rne"].tolist() == [[4, 6], [0, 4]]


def test_odd_sizes_scalar_path(dev):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 65536, size=(2, 1, 1, 37, 53)).astype(np.uint16)
    lab = rng.integers(0, 6, size=(2, 1, 37, 53)).astype(np.uint16)
    _check(img, lab, dev)


def test_hash_overflow_random_labels(dev):
    """every pixel a random label in 0..4999: far more than 512 distinct labels per chunk, so the
    LDS table overflows into the direct-to-HBM path; also max values 65535."""
    rng = np.random.default_rng(4)
    img = np.full((1, 2, 1, 128, 256), 65535, np.uint16)
    img[0, 1] = rng.integers(0, 65536, size=(1, 128, 256))
    lab = rng.integers(0, 5000, size=(1, 1, 128, 256)).astype(np.uint16)
    _check(img, lab, dev)


def test_label_65535_and_five_channels(dev):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 65536, size=(1, 5, 1, 64, 64)).astype(np.uint16)
    lab = np.zeros((1, 1, 64, 64), np.uint16)
    lab[0, 0, 10:20, 10:20] = 65535
    lab[0, 0, 30:40, 5:50] = 1
    _check(img, lab, dev)


def test_dense_5000_labels_full_size(dev):
    """BASELINE config C5: 2048x2048 with 5041 labels; oracle = vectorised bincount form."""
    rng = np.random.default_rng(6)
    lab = synth.make_dense_labels(2048, 2048)[None, None]
    img = rng.integers(0, 65536, size=(1, 1, 1, 2048, 2048)).astype(np.uint16)
    _check(img, lab, dev)


def test_empty_frame_all_background(dev):
    from imageanalysis_pipeline_amd import ops
    img = np.ones((1, 1, 1, 32, 32), np.uint16)
    lab = np.zeros((1, 1, 32, 32), np.uint16)
    got = ops.measure_labels(ops.to_device_u16(img, dev), ops.to_device_u16(lab, dev))[0]
    assert got["ids"].size == 0


def test_bad_arguments_raise(dev):
    import torch
    from imageanalysis_pipeline_amd import ops
    img = torch.zeros((1, 1, 1, 8, 8), dtype=torch.uint16, device=dev)
    lab = torch.zeros((1, 1, 8, 16), dtype=torch.uint16, device=dev)
    with pytest.raises(ValueError):
        ops.measure_accumulate(img, lab)
    with pytest.raises(TypeError):
        ops.measure_accumulate(img.view(torch.int16), lab)


@pytest.mark.parametrize("provisional", [4, 64, 8192])
def test_label_range_found_by_the_measuring_pass(dev, provisional, monkeypatch):
    """n_labels not given: the pass reports the largest label itself; stacks with more ids than the provisional
    table are measured again -- same tables either way, and the same as with the range given"""
    from imageanalysis_pipeline_amd import ops
    monkeypatch.setattr(ops, "PROVISIONAL_LABELS", provisional)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 65536, size=(3, 2, 1, 64, 96)).astype(np.uint16)
    lab = np.stack([synth.make_labels(64, 96, 10, seed=s)[None] for s in range(3)])
    lab[2][lab[2] == 3] = 40          # one frame reaches further than the others
    lab[1] = 0                        # one frame has no label at all
    d, dl = ops.to_device_u16(img, dev), ops.to_device_u16(lab, dev)
    acc = ops.measure_accumulate(d, dl)
    assert acc.shape == (3, 41, 6)
    assert torch.equal(acc, ops.measure_accumulate(d, dl, 41))
    _check(img, lab, dev)


@pytest.mark.parametrize("yx", [(256, 512), (96, 1004)])
def test_dense_and_sparse_rounds_in_one_workgroup(dev, yx):
    """The measuring pass visits a wave's round in place when at least half its lanes hold a labelled vector and
    queues the labelled vectors of the other rounds in LDS: a frame whose upper part is densely labelled (every
    8-pixel vector carries labels, two labels meeting inside vectors), whose lower part holds a few discs and a
    sprinkle of single labelled pixels, takes both routes inside the same workgroups (16384-pixel chunks).
    X = 1004: vectors straddle rows."""
    y, x = yx
    rng = np.random.default_rng(11)
    t, c = 2, 3
    img = rng.integers(0, 65536, size=(t, c, 1, y, x)).astype(np.uint16)
    lab = np.zeros((t, 1, y, x), np.uint16)
    for f in range(t):
        cols = (np.arange(x) // 5 + 1 + 7 * f).astype(np.uint16)       # runs of 5: label borders inside vectors
        lab[f, 0, : y // 3] = cols[None, :]
        lab[f, 0, y // 3 :] = synth.make_labels(y - y // 3, x, 9, seed=20 + f) * 300
        iy = rng.integers(y // 3, y, size=40)
        ix = rng.integers(0, x, size=40)
        lab[f, 0, iy, ix] = rng.integers(1, 65536, size=40)             # lone pixels, labels up to 65535
    _check(img, lab, dev)
