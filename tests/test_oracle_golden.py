"""CPU: pins the oracle (oracle/ref_path.py) against the committed golden vectors in tests/golden/,
which were produced by scikit-image 0.18.3 / scipy (tests/golden/make_golden.py) -- the closest
executable ground truth available: the reference's own numeric dependencies (smo, pystackreg, cv2)
are absent from the image and the reference ships no tests or fixtures."""
import os

import numpy as np

from oracle import ref_path as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_measure_vs_regionprops():
    d = np.load(os.path.join(G, "regionprops.npz"))
    r = O.measure_labels(d["img"], d["labels"])
    # regionprops never reports label 0; the reference drops the smallest value (0 here)
    assert np.array_equal(r["ids"], d["ids"])
    assert np.array_equal(r["area"], d["area"])
    assert np.array_equal(r["centroid"], d["centroid"])           # coords.mean(0): same f64 arithmetic
    assert np.array_equal(r["mean"], d["mean"])
    lit = O.measure_labels_literal(d["img"], d["labels"])
    assert np.array_equal(lit["mean"], d["mean"]) and np.array_equal(lit["area"], d["area"])


def test_apply_shift_vs_scipy_ndimage():
    d = np.load(os.path.join(G, "ndimage_shift.npz"))
    for s, exp in zip(d["shifts"], d["out"]):
        got = O.apply_shift_f64(d["img"], s[0], s[1])
        assert np.abs(got - exp).max() < 1e-6
        # u16 result: truncation of a value within 1e-6 of scipy's can differ only at integers
        q = O.apply_shift(d["img"], s[0], s[1]).astype(np.int64)
        assert np.abs(q - np.trunc(np.clip(exp, 0, 65535))).max() <= 1


def test_smo_vs_scipy_filters():
    d = np.load(os.path.join(G, "smo_scipy.npz"))
    assert np.abs(O.smo_image(d["img"], 0.0, 7) - d["smo_s0_k7"]).max() < 1e-12
    assert np.abs(O.smo_image(d["img"], 1.5, 7) - d["smo_s15_k7"]).max() < 1e-9
    assert np.abs(O.smo_image(d["img"], 0.0, 5) - d["smo_s0_k5"]).max() < 1e-12


def test_shift_estimate_vs_skimage():
    d = np.load(os.path.join(G, "xcorr_skimage.npz"))
    for pair, c, f in zip(d["pairs"], d["coarse"], d["fine"]):
        assert np.array_equal(np.array(O.phase_shift(pair[0], pair[1], upsample=False)), c)
        got = np.array(O.phase_shift(pair[0], pair[1]))
        # skimage searches a 0.01 px grid, the oracle a 0.1 px grid + parabola: agree within 0.01 px
        assert np.abs(got - f).max() <= 0.0101, (got, f)


def test_bg_smo_integer_identity():
    """out == trunc(max(img - med, 0)) with the median the oracle reports (half-integers included)"""
    rng = np.random.default_rng(0)
    img = (rng.poisson(30, size=(40, 64)) + 100).astype(np.uint16)
    out, med2, n, mask = O.bg_smo(img, 0.08)
    assert n == int(mask.sum()) and n > 0
    assert med2 == int(round(2 * np.median(img[mask])))
    assert np.array_equal(out, (np.maximum(2 * img.astype(np.int64) - med2, 0) >> 1).astype(np.uint16))


def test_bg_manual_and_blur_known_answers():
    img = np.arange(48, dtype=np.uint16).reshape(6, 8) * 10
    out, bg = O.bg_manual(img, 1, 2, 2, 3)                # ROI values 100,110,180,190 -> mean 145
    assert bg == 145 and out[0, 0] == 0 and out[5, 7] == 470 - 145
    const = np.full((20, 20), 777, np.uint16)
    assert np.array_equal(O.gauss_blur(const, (15, 15), 5.0), const)   # normalised taps keep constants
    imp = np.zeros((21, 21), np.uint16)
    imp[10, 10] = 60000
    b = O.gauss_blur(imp, (5, 5), 1.0)
    k = O.gauss_taps_cv(5, 1.0).astype(np.float64)
    assert abs(int(b[10, 10]) - 60000 * k[2] * k[2]) <= 1 and b.sum() > 0
    assert np.array_equal(b, b.T)


def test_im_reg_stack_modes_recover_integer_drift():
    from imageanalysis_pipeline_amd import synth
    st, dr = synth.make_stack(4, 2, 1, 64, 64, n_cells=6, integer_drift=True, seed=2)
    for mode in ("first", "previous"):
        out, sh = O.im_reg_stack(st, 0, mode, upsample=False)
        if mode == "first":
            assert np.array_equal(sh, -dr)
    try:
        O.im_reg_stack(st, 0, "last")
    except ValueError:
        pass
    else:
        raise AssertionError("bad reg_ref must raise ValueError (utility.py:297-299)")


def test_edt_vs_scipy():
    """oracle edt_sq (separable exact integers) against scipy.ndimage.distance_transform_edt (golden)."""
    g = np.load(os.path.join(G, "edt_scipy.npz"))
    for m, d in zip(g["masks"], g["dist"]):
        sq = O.edt_sq(m)
        assert np.array_equal(np.sqrt(sq.astype(np.float64)), d)      # integer squares: sqrt is exact in f64


def test_pixel_table_order_and_rounding():
    ref = np.zeros((6, 8), np.uint16)
    ref[0, 0] = 1
    cell = np.zeros((6, 8), np.uint16)
    cell[1, 2] = 3
    cell[0, 5] = 9
    cell[4, 0] = 1
    img = np.arange(2 * 48, dtype=np.uint16).reshape(2, 6, 8)
    dmap, vals = O.pixel_table(cell, img, ref, pix_size=0.5)
    # row-major order: (0,5), (1,2), (4,0); distances 5, sqrt(5), 4 -> x0.5 -> 2.5, 1.118, 2.0 -> round-half-even
    assert dmap.tolist() == [2, 1, 2]
    assert vals.tolist() == [[5, 53], [10, 58], [32, 80]]


def test_label_morphology_vs_scipy():
    """oracle label_erode / label_dilate against scipy grey morphology with skimage's disk (golden)."""
    g = np.load(os.path.join(G, "morph_scipy.npz"))
    lab = g["labels"]
    for r in (1, 3, 6):
        assert np.array_equal(O.label_erode(lab, r), g[f"erode_{r}"])
        assert np.array_equal(O.label_dilate(lab, r), g[f"dilate_{r}"])
    # iterations = repeated application (cv2 iterations=)
    assert np.array_equal(O.label_erode(lab, 2, iterations=2), O.label_erode(O.label_erode(lab, 2), 2))


def test_gauss_taps_cv_small_kernel_tables():
    """cv2.getGaussianKernel: odd sizes up to 7 with sigma <= 0 come from OpenCV's fixed table, everything
    else from the normalised exponential (sigma <= 0 -> 0.3*((k-1)*0.5-1)+0.8)."""
    assert np.array_equal(O.gauss_taps_cv(3, 0.0), np.float32([0.25, 0.5, 0.25]))
    assert np.array_equal(O.gauss_taps_cv(5, -1.0), np.float32([0.0625, 0.25, 0.375, 0.25, 0.0625]))
    assert np.array_equal(O.gauss_taps_cv(7, 0.0), np.float32([0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]))
    assert np.array_equal(O.gauss_taps_cv(1, 0.0), np.float32([1.0]))
    w = O.gauss_taps_cv(9, 0.0)                       # no table: sigma = 0.3*(4-1)+0.8 = 1.7
    x = np.arange(9) - 4.0
    e = np.exp(-x * x / (2 * 1.7 * 1.7))
    assert np.allclose(w, e / e.sum(), rtol=1e-6) and abs(float(w.sum()) - 1.0) < 1e-6
    assert not np.array_equal(O.gauss_taps_cv(3, 0.8), O.gauss_taps_cv(3, 0.0))   # an explicit sigma is honoured
