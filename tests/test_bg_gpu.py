"""GPU parity of the background stage against the oracle.
Reference: utility.py:242-270 (smo_bg_sub), :213-233 (man_bg_sub arithmetic), :462 (blur_img)."""
import numpy as np
import pytest
import torch

from oracle import ref_path as O
from imageanalysis_pipeline_amd import synth

pytestmark = pytest.mark.gpu

SMO_TOL = 2e-6      # fp32 kernel vs float64 oracle on the SMO value (range 0..1)


def _frames(n, y, x, seed=0, cells=12):
    st, _ = synth.make_stack(n, 1, 1, y, x, n_cells=cells, integer_drift=True, seed=seed)
    return st[:, 0, 0]


@pytest.mark.parametrize("shape", [(64, 256), (136, 508), (300, 252), (40, 248), (130, 1024)])
def test_smo_image_fast_path(dev, shape):
    from imageanalysis_pipeline_amd import ops
    y, x = shape
    img = _frames(2, y, x, seed=shape[0])
    got = ops.smo_image(ops.to_device_u16(img, dev)).cpu().numpy()
    for i in range(2):
        exp = O.smo_image(img[i])
        assert np.abs(got[i] - exp).max() < SMO_TOL


@pytest.mark.parametrize("shape", [(37, 53), (9, 250), (64, 100), (7, 256)])
def test_smo_image_generic_path(dev, shape):
    from imageanalysis_pipeline_amd import ops
    rng = np.random.default_rng(1)
    img = (rng.poisson(100, size=(1,) + shape) + 400).astype(np.uint16)
    got = ops.smo_image(ops.to_device_u16(img, dev)).cpu().numpy()
    assert np.abs(got[0] - O.smo_image(img[0])).max() < SMO_TOL


def test_smo_image_float_input_and_size5(dev):
    from imageanalysis_pipeline_amd import ops
    rng = np.random.default_rng(2)
    a = rng.standard_normal(size=(1, 96, 256)).astype(np.float32)
    got = ops.smo_image(torch.from_numpy(a).to(dev)).cpu().numpy()
    assert np.abs(got[0] - O.smo_image(a[0])).max() < SMO_TOL
    got5 = ops.smo_image(torch.from_numpy(a).to(dev), size=5).cpu().numpy()   # generic kernel
    assert np.abs(got5[0] - O.smo_image(a[0], 0.0, 5)).max() < SMO_TOL


def test_smo_threshold_matches_oracle(dev):
    from imageanalysis_pipeline_amd import ops
    t = ops.smo_threshold((256, 256), seed=0, device=dev)
    e = O.smo_threshold((256, 256), seed=0)
    # same seeded normal image (float32-cast on the GPU side)
    assert abs(t - e) < 1e-5 and 0.0 < t < 0.2


def _check_bg(img, thr, dev, sigma=0.0, inplace=False):
    from imageanalysis_pipeline_amd import ops
    d = ops.to_device_u16(img, dev)
    out, med2, nbg = ops.bg_smo(d, thr, sigma=sigma, out=d if inplace else None)
    out = ops.to_numpy_u16(out)
    med2 = med2.cpu().numpy()
    nbg = nbg.cpu().numpy()
    for i in range(img.shape[0]):
        eo, em2, en, emask = O.bg_smo(img[i], thr, sigma=sigma)
        s = O.smo_image(img[i], sigma)
        borderline = int((np.abs(s - thr) < (SMO_TOL if sigma == 0 else 2e-5)).sum())
        assert abs(int(nbg[i]) - en) <= borderline, (nbg[i], en, borderline)
        assert med2[i] == em2, "median differs"
        assert np.array_equal(out[i], eo), "output not bit-exact"


def test_bg_smo_parity_and_inplace(dev):
    img = _frames(3, 256, 512, seed=3, cells=25)
    thr = O.smo_threshold((256, 512))
    _check_bg(img, thr, dev)
    _check_bg(img, thr, dev, inplace=True)


def test_bg_smo_known_answer_constant_plus_blobs(dev):
    """constant background 500 + isolated blobs: background pixels have zero gradient -> SMO 0 ->
    all selected; median = 500 exactly; blobs keep (value - 500)."""
    from imageanalysis_pipeline_amd import ops
    img = np.full((1, 128, 256), 500, np.uint16)
    img[0, 40:60, 100:130] = 3000
    img[0, 90:100, 10:40] = 65535
    out, med2, nbg = ops.bg_smo(ops.to_device_u16(img, dev), 0.05)
    out = ops.to_numpy_u16(out)
    assert int(med2[0]) == 1000
    assert out[0, 0, 0] == 0 and out[0, 50, 110] == 2500 and out[0, 95, 20] == 65035
    eo = O.bg_smo(img[0], 0.05)[0]
    assert np.array_equal(out[0], eo)


def test_bg_smo_half_integer_median_and_empty_mask(dev):
    from imageanalysis_pipeline_amd import ops
    # two flat halves 100 | 101 with equal area: median of the zero-gradient pixels = 100.5
    img = np.full((1, 64, 256), 100, np.uint16)
    img[0, :, 128:] = 101
    out, med2, nbg = ops.bg_smo(ops.to_device_u16(img, dev), 0.05)
    eo, em2, en, _ = O.bg_smo(img[0], 0.05)
    assert int(med2[0]) == em2
    assert np.array_equal(ops.to_numpy_u16(out)[0], eo)
    # threshold 0 -> nothing selected -> median 0 -> identity
    out0, med0, n0 = ops.bg_smo(ops.to_device_u16(img, dev), 0.0)
    assert int(n0[0]) == 0 and int(med0[0]) == 0
    assert np.array_equal(ops.to_numpy_u16(out0), img)


def test_bg_smo_wide_value_range_uses_global_fallback(dev):
    """background values spread over more than the 1024-value LDS window"""
    rng = np.random.default_rng(7)
    img = rng.integers(0, 65536, size=(1, 96, 256)).astype(np.uint16)
    _check_bg(img, 0.08, dev)


def test_bg_smo_odd_shape_generic(dev):
    rng = np.random.default_rng(8)
    img = (rng.poisson(50, size=(2, 45, 77)) + 300).astype(np.uint16)
    _check_bg(img, O.smo_threshold((45, 77)), dev)


def test_bg_smo_sigma(dev):
    img = _frames(1, 128, 256, seed=9)
    _check_bg(img, 0.05, dev, sigma=1.5)


def test_bg_smo_full_size_properties(dev):
    """2048x2048 (BASELINE size): against the oracle on one plane, plus the size-independent
    property out == max(in - med, 0) truncated for the median the kernel reports."""
    from imageanalysis_pipeline_amd import ops
    rng = np.random.default_rng(10)
    img = (rng.poisson(100, size=(2, 2048, 2048)) + 400).astype(np.uint16)
    img[:, 500:900, 700:1300] += 5000
    thr = 0.04
    out, med2, nbg = ops.bg_smo(ops.to_device_u16(img, dev), thr)
    out = ops.to_numpy_u16(out)
    med2 = med2.cpu().numpy()
    for i in range(2):
        exp = np.maximum(2 * img[i].astype(np.int64) - int(med2[i]), 0) >> 1
        assert np.array_equal(out[i], exp.astype(np.uint16))
    eo, em2, en, _ = O.bg_smo(img[0], thr)
    assert int(med2[0]) == em2 and np.array_equal(out[0], eo)


def test_bg_manual(dev):
    from imageanalysis_pipeline_amd import ops
    rng = np.random.default_rng(11)
    img = rng.integers(0, 3000, size=(3, 70, 90)).astype(np.uint16)
    img[1, 10:20, 10:20] = 65535
    roi = (5, 7, 33, 41)
    out, bg = ops.bg_manual(ops.to_device_u16(img, dev), roi)
    out = ops.to_numpy_u16(out)
    for i in range(3):
        eo, ebg = O.bg_manual(img[i], *roi)
        assert int(bg[i]) == ebg
        assert np.array_equal(out[i], eo)
    # exact .5 tie: ROI of two pixels 2 and 3 -> mean 2.5 -> rint -> 2 (half-even)
    tie = np.zeros((1, 8, 8), np.uint16)
    tie[0, 0, 0], tie[0, 0, 1] = 2, 3
    out, bg = ops.bg_manual(ops.to_device_u16(tie, dev), (0, 0, 0, 1))
    assert int(bg[0]) == 2 == O.bg_manual(tie[0], 0, 0, 0, 1)[1]
    with pytest.raises(ValueError):
        ops.bg_manual(ops.to_device_u16(tie, dev), (0, 0, 9, 1))


@pytest.mark.parametrize("k,sig", [((15, 15), 5.0), ((3, 7), 0.0), ((31, 5), 2.5), ((63, 63), 12.0), ((57, 61), 0.0),
                                   ((5, 1), 0.0), ((59, 27), 0.7), ((1, 21), 0.7), ((9, 9), 1.1)])
def test_gauss_blur_bit_exact(dev, k, sig):
    from imageanalysis_pipeline_amd import ops
    # (sigmas that are not float32 numbers included: the entry point takes a double, as cv2's sigmaX -- with a float
    # 0.7 one pixel in 23 000 rounded the other way, found by scripts/fuzz_stages.py)
    rng = np.random.default_rng(12)
    for shape in ((2, 75, 133), (1, 251, 93)):
        img = rng.integers(0, 65536, size=shape).astype(np.uint16)
        got = ops.to_numpy_u16(ops.gauss_blur(ops.to_device_u16(img, dev), k, sig))
        for i in range(shape[0]):
            assert np.array_equal(got[i], O.gauss_blur(img[i], k, sig))
    with pytest.raises(TypeError):
        ops.gauss_blur(ops.to_device_u16(img, dev), [15, 15], 5.0)


@pytest.mark.parametrize("r", [1, 2, 3, 4, 5, 6, 7])
def test_gauss_blur_marching_kernel_bit_exact(dev, r):
    """Square kernels of radius 1..7 on planes of even width take the marching kernel (128-column strips, 32-row steps,
    bands of up to 256 rows): planes narrower than a strip, a column past a strip, more rows than a band, fewer rows and
    columns than the kernel's reach (cv2's BORDER_REFLECT_101 folds more than once), sigma given and derived from the
    size."""
    from imageanalysis_pipeline_amd import ops
    rng = np.random.default_rng(100 + r)
    k = 2 * r + 1
    for shape, sig in (((2, 64, 64), 0.0), ((1, 301, 258), 1.3), ((3, 33, 130), 0.0), ((1, 5, 8), 2.0),
                       ((1, 2, 300), 0.7), ((1, 530, 16), 5.0)):
        img = rng.integers(0, 65536, size=shape).astype(np.uint16)
        if shape[1] > 300:
            img[:, :40] = 65535       # saturated and empty stretches: rounding at the ends of the range
            img[:, 40:80] = 0
        got = ops.to_numpy_u16(ops.gauss_blur(ops.to_device_u16(img, dev), (k, k), sig))
        for i in range(shape[0]):
            assert np.array_equal(got[i], O.gauss_blur(img[i], (k, k), sig)), (shape, sig, i)


def test_gauss_blur_marching_kernel_full_plane(dev):
    """The reference's default (15, 15), sigma 5 (experiments.py:337) on 2048 x 2048 planes: 16 strips x 8 bands."""
    from imageanalysis_pipeline_amd import ops
    img = _frames(2, 2048, 2048, seed=21)
    got = ops.to_numpy_u16(ops.gauss_blur(ops.to_device_u16(img, dev), (15, 15), 5.0))
    for i in range(2):
        assert np.array_equal(got[i], O.gauss_blur(img[i], (15, 15), 5.0))


def test_smo_image_and_threshold_with_sigma(dev):
    """SMO(sigma > 0): gaussian pre-smoothing (scipy 'reflect', truncate 4) ahead of the operator, for the
    image planes and for the null image the threshold comes from (utility.py:252)."""
    from imageanalysis_pipeline_amd import ops
    img = _frames(2, 128, 256, seed=13)
    got = ops.smo_image(ops.to_device_u16(img, dev), 7, sigma=1.5).cpu().numpy()
    for i in range(2):
        # the smoothed plane is kept in fp32: ~500 counts +- 3e-5, and the operator takes differences of
        # neighbours that differ by a count or so -> 1e-4 on the SMO value (2e-6 without the pre-filter)
        assert np.abs(got[i] - O.smo_image(img[i].astype(np.float64), 1.5, 7)).max() <= 3e-4
    t = ops.smo_threshold((128, 256), sigma=1.5, seed=0, device=dev)
    e = O.smo_threshold((128, 256), sigma=1.5, seed=0)
    assert abs(t - e) <= 2e-5 * e + 2e-6
    assert t != ops.smo_threshold((128, 256), sigma=0.0, seed=0, device=dev)


@pytest.mark.parametrize("sigma", [0.6, 3.0, 7.5, 9.0])
def test_smo_sigma_prefilter_paths(dev, sigma):
    """The gaussian pre-filter of SMO(sigma > 0) (scipy 'reflect', truncate 4): radii up to 31 go through the LDS tile
    kernel, larger ones (sigma 9: radius 36) through the two-pass form; planes smaller than the filter's reach and
    planes that are not a multiple of the tile included."""
    from imageanalysis_pipeline_amd import ops
    for y, x in ((50, 70), (200, 312), (19, 40)):
        img = _frames(1, y, x, seed=int(10 * sigma) + y)
        got = ops.smo_image(ops.to_device_u16(img, dev), 7, sigma=sigma).cpu().numpy()[0]
        exp = O.smo_image(img[0].astype(np.float64), sigma, 7)
        # as test_smo_image_and_threshold_with_sigma: fp32 smoothed plane, differences of neighbours a count apart
        assert np.quantile(np.abs(got - exp), 0.999) <= 3e-4, (sigma, y, x, np.abs(got - exp).max())


def test_bg_levels_over_several_batches(dev):
    """More planes than one batch of histograms holds (256): the histograms are zeroed once per call and handed back
    clean by the median kernel, so the levels of later batches must equal those of the same planes taken alone --
    planes of very different backgrounds follow each other, a plane of noise over the full u16 range (background
    outside every LDS window) sits at the start of each batch."""
    from imageanalysis_pipeline_amd import ops
    rng = np.random.default_rng(21)
    n, y, x = 700, 16, 256
    img = (rng.poisson(30, size=(n, y, x)) + rng.integers(100, 60000, size=(n, 1, 1))).clip(0, 65535).astype(np.uint16)
    img[::256] = rng.integers(0, 65536, size=(3, y, x)).astype(np.uint16)
    d = ops.to_device_u16(img, dev)
    med_all, nbg_all = ops.bg_smo_levels(d, 0.06)
    for lo, hi in ((0, 3), (255, 259), (511, 515), (690, 700)):
        med, nbg = ops.bg_smo_levels(d[lo:hi].contiguous(), 0.06)
        assert torch.equal(med, med_all[lo:hi]) and torch.equal(nbg, nbg_all[lo:hi])
    for i in (0, 256, 300, 512, 699):
        _, em2, en, _ = O.bg_smo(img[i], 0.06)
        assert int(med_all[i]) == em2
    # a second call on the same workspace (left clean by the first) gives the same levels
    med2, _ = ops.bg_smo_levels(d, 0.06)
    assert torch.equal(med2, med_all)
