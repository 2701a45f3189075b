"""GPU: the folder-level shim (imageanalysis_pipeline_amd.pipeline) end to end on a synthetic
experiment folder, read like the reference's own call sequence (core.py:105-112, 328-330):
smo_bg_sub -> im_reg -> extract_channelData -> extract_centroids."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import ref_path as O
from imageanalysis_pipeline_amd import synth, tiffio
from imageanalysis_pipeline_amd.pipeline import Analysis, Utility
from test_shim_cpu import make_exp

pytestmark = pytest.mark.gpu


def _write_stack(exp, stack, chans):
    for ci, ch in enumerate(chans):
        for f in range(stack.shape[0]):
            for k in range(stack.shape[2]):
                tiffio.imwrite(os.path.join(exp, "Images", "%s_f%04d_z%04d.tif" % (ch, f + 1, k + 1)), stack[f, ci, k])


def test_pre_process_and_analysis_sequence(dev, tmp_path):
    from imageanalysis_pipeline_amd import ops
    t, c, y, x = 4, 2, 128, 256
    chans = ("GFP", "RFP")
    exp, _ = make_exp(tmp_path, t=t, c=c, y=y, x=x, chans=chans)
    stack, _ = synth.make_stack(t, c, 1, y, x, n_cells=10, integer_drift=True, seed=5)
    _write_stack(exp, stack, chans)
    img = os.path.join(exp, "Images")

    # --- smo_bg_sub (utility.py:242-270): files overwritten in place, exp_prop updated
    prop = Utility.smo_bg_sub(imgFold_path=img, sigma=0.0, size=7)
    assert prop["img_preProcess"]["bg_sub"] == "Auto"
    assert prop["fct_inputs"]["smo_bg_sub"] == {"imgFold_path": img, "smo_sigma": 0.0, "smo_size": 7}
    thr = ops.smo_threshold((y, x), device=dev)
    bg = np.empty_like(stack)
    for f in range(t):
        for ci, ch in enumerate(chans):
            got = tiffio.imread(os.path.join(img, "%s_f%04d_z0001.tif" % (ch, f + 1)))
            bg[f, ci, 0] = O.bg_smo(stack[f, ci, 0], thr)[0]
            assert np.array_equal(got, bg[f, ci, 0])

    # --- im_reg (utility.py:274-430)
    prop = Utility.im_reg(imgFold_path=img, reg_channel="GFP", reg_mtd="translation", chan_shift=False,
                          reg_ref="first", reg_ow=False)
    assert prop["img_preProcess"]["reg"] is True and prop["channel_seg"] == {"Images_Registered": ["GFP"]}
    regdir = img + "_Registered"
    ereg, _ = O.im_reg_stack(bg, 0, "first")
    reg = np.empty_like(stack)
    for f in range(t):
        for ci, ch in enumerate(chans):
            reg[f, ci, 0] = tiffio.imread(os.path.join(regdir, "%s_f%04d_z0001.tif" % (ch, f + 1)))
    # the shim refines to sub-pixel shifts (upsample), so the warp interpolates even on integer drift.  Bar: 1 count
    # (float error of the fp32 warp + the shift difference times the local gradient stays
    # below one count, and truncating two values less than 1 apart gives integers at most 1 apart)
    assert np.abs(reg.astype(np.int64) - ereg.astype(np.int64)).max() <= 1
    # second call without reg_ow: skipped (utility.py:292), files untouched
    mt = os.path.getmtime(os.path.join(regdir, "GFP_f0001_z0001.tif"))
    Utility.im_reg(img, "GFP", "translation", False, "first", False)
    assert os.path.getmtime(os.path.join(regdir, "GFP_f0001_z0001.tif")) == mt

    # --- masks + analysis (analysis.py:26-190)
    mdir = os.path.join(exp, "Masks_CP")
    os.makedirs(mdir)
    lab = synth.make_labels(y, x, 10, seed=5)
    Utility.save_mask(np.stack([lab] * t), mdir, batch=list(range(t)), z_size=1, frames=t, chan_seg="GFP")
    prop = Utility.open_exp_prop(exp)
    prop["channel_seg"]["Masks_CP"] = ["GFP"]
    Utility.save_exp_prop(exp, prop)

    a = Analysis(exp, "GFP")
    a.extract_channelData(imgFold="Images_Registered", maskFold="Masks_CP")
    a.extract_centroids(maskFold="Masks_CP")
    df = a.df_analysis
    assert list(df.columns) == ["cell", "frames", "time", "mask_chan", "tag", "exp", "GFP", "RFP", "mask_ID", "y", "x"]
    # the reference's literal loops on the same files
    rows = []
    for f in range(t):
        m = O.measure_labels_literal(reg[f, :, 0], lab)
        for i, obj in enumerate(m["ids"]):
            rows.append((f"cond_s1_exp_GFP_cell{obj}", f + 1, m["mean"][i, 0], m["mean"][i, 1], int(obj),
                         int(m["centroid_rne"][i, 0]), int(m["centroid_rne"][i, 1])))
    assert len(rows) == len(df)
    for r, (_, d) in zip(rows, df.iterrows()):
        assert d["cell"] == r[0] and d["frames"] == r[1] and d["mask_ID"] == r[4]
        assert d["GFP"] == r[2] and d["RFP"] == r[3]                     # bit-exact float64
        assert d["y"] == r[5] and d["x"] == r[6]
    assert os.path.exists(os.path.join(exp, "df_analysis.csv"))
    # cached second call reads the csv back (analysis.py:108-112)
    b = Analysis(exp, "GFP")
    b.extract_channelData(imgFold="Images_Registered", maskFold="Masks_CP")
    assert len(b.df_analysis) == len(df)


def test_resident_planes_follow_the_files(dev, tmp_path, monkeypatch):
    """the shim keeps the planes it just wrote on the GPU (smo_bg_sub -> im_reg reads no file), but only while the
    files still are what it wrote: a plane replaced from outside is read again, and the result is that of a run
    without the residency"""
    import time
    from imageanalysis_pipeline_amd import pipeline as P
    t, c, y, x = 3, 2, 96, 128
    chans = ("GFP", "RFP")
    stack, _ = synth.make_stack(t, c, 1, y, x, n_cells=6, integer_drift=False, seed=9)
    other, _ = synth.make_stack(1, 1, 1, y, x, n_cells=6, integer_drift=False, seed=77)

    def run(resident_gb, root):
        monkeypatch.setenv("IAP_SHIM_RESIDENT_GB", resident_gb)
        P.RESIDENT.clear()
        exp, _ = make_exp(root, t=t, c=c, y=y, x=x, chans=chans)
        _write_stack(exp, stack, chans)
        img = os.path.join(exp, "Images")
        h0 = P.RESIDENT.hits
        Utility.smo_bg_sub(imgFold_path=img, sigma=0.0, size=7)
        Utility.im_reg(img, "GFP", "translation", False, "mean", False)
        served = P.RESIDENT.hits - h0
        first = [tiffio.imread(os.path.join(img + "_Registered", "%s_f%04d_z0001.tif" % (ch, f + 1))) for f in range(t) for ch in chans]
        time.sleep(0.05)
        tiffio.imwrite(os.path.join(img, "GFP_f0002_z0001.tif"), other[0, 0, 0])      # an external writer
        Utility.im_reg(img, "GFP", "translation", False, "mean", True)
        second = [tiffio.imread(os.path.join(img + "_Registered", "%s_f%04d_z0001.tif" % (ch, f + 1))) for f in range(t) for ch in chans]
        return served, first, second

    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    served, first, second = run("48", tmp_path / "a")
    served0, first0, second0 = run("0", tmp_path / "b")
    assert served == 1 and served0 == 0              # im_reg's read of Images came from the GPU, once
    assert all(np.array_equal(a, b) for a, b in zip(first, first0))
    assert all(np.array_equal(a, b) for a, b in zip(second, second0))
    assert not all(np.array_equal(a, b) for a, b in zip(first, second))   # the replaced plane did change the result


def test_man_bg_sub_and_blur(dev, tmp_path):
    exp, stack = make_exp(tmp_path, t=2, c=2, y=40, x=64)
    img = os.path.join(exp, "Images")
    prop = Utility.man_bg_sub(img, roi=((2, 3), (20, 30)))
    assert prop["img_preProcess"]["bg_sub"] == "Manual"
    got = tiffio.imread(os.path.join(img, "RFP_f0002_z0001.tif"))
    assert np.array_equal(got, O.bg_manual(stack[1, 1, 0], 2, 3, 20, 30)[0])
    cur = tiffio.imread(os.path.join(img, "GFP_f0001_z0001.tif"))
    prop, path = Utility.blur_img(img, (15, 15), 5, False)
    assert path.endswith("Images_Blured") and prop["img_preProcess"]["blur"] is True
    assert np.array_equal(tiffio.imread(os.path.join(path, "GFP_f0001_z0001.tif")), O.gauss_blur(cur, (15, 15), 5))
    assert Utility.man_bg_sub(img, roi=None)["img_preProcess"]["bg_sub"] == "Manual"   # 'skip': untouched


def test_pixel_distance_table(dev, tmp_path):
    """Analysis.pixel_distance (analysis.py:262-304) on files: the literal numpy restatement per frame."""
    t, c, y, x = 3, 2, 96, 128
    chans = ("GFP", "RFP")
    exp, _ = make_exp(tmp_path, t=t, c=c, y=y, x=x, chans=chans)
    rng = np.random.default_rng(11)
    stack = rng.integers(0, 65536, size=(t, c, 1, y, x), dtype=np.uint16)
    _write_stack(exp, stack, chans)
    prop = Utility.open_exp_prop(exp)
    pix = float(prop["metadata"].get("pixel_microns") or 0.65)
    prop["metadata"]["pixel_microns"] = pix
    Utility.save_exp_prop(exp, prop)
    wound = np.zeros((t, y, x), np.uint16)
    for f in range(t):
        wound[f, 30:50 - 3 * f, 40 + 5 * f:90] = 1                    # a closing wound
    cells = np.stack([synth.make_labels(y, x, 8, seed=f) for f in range(t)])
    # the reference masks under the names the reference itself gives them (utility.py:866: no _z token)
    os.makedirs(os.path.join(exp, "Masks_wound"))
    for f in range(t):
        tiffio.imwrite(os.path.join(exp, "Masks_wound", "mask_wound_f%04d.tif" % (f + 1)), wound[f])
    os.makedirs(os.path.join(exp, "Masks_CP"))
    Utility.save_mask(cells, os.path.join(exp, "Masks_CP"), batch=list(range(t)), z_size=1, frames=t, chan_seg="GFP")

    a = Analysis(exp, "GFP")
    a.pixel_distance(imgFold="Images", maskFold="Masks_CP", maskLabel="wound")
    df = a.df_pixel
    assert list(df.columns) == ["dmap", "GFP", "RFP", "time", "frame", "tag", "exp"]
    pos = 0
    for f in range(t):
        ed, ev = O.pixel_table(cells[f], stack[f, :, 0], wound[f], pix)
        part = df.iloc[pos:pos + len(ed)]
        assert np.array_equal(part["dmap"].to_numpy(), ed)
        assert np.array_equal(part[["GFP", "RFP"]].to_numpy(), ev.astype(float))
        assert (part["frame"] == f + 1).all() and (part["time"] == a.ts[f]).all()
        pos += len(ed)
    assert pos == len(df)
    assert os.path.exists(os.path.join(exp, "df_pixel.csv"))
    assert Utility.open_exp_prop(exp)["fct_inputs"]["pixel_distance"]["maskLabel"] == "wound"
    # apply_dmap keeps the reference's return shapes
    assert Utility.apply_dmap(wound[0], 1).shape == (y, x)
    assert np.array_equal(Utility.apply_dmap(wound, t)[1], np.sqrt(O.edt_sq(wound[1]).astype(float)))


def test_masks_class_columns(dev, tmp_path):
    """extract_channelData(maskFold='Masks_Class') adds the reference's cell_class / pos_cell columns
    (analysis.py:44-52, 63, 87-92)."""
    t, c, y, x = 2, 2, 64, 96
    chans = ("GFP", "RFP")
    exp, stack = make_exp(tmp_path, t=t, c=c, y=y, x=x, chans=chans)
    lab = synth.make_labels(y, x, 6, seed=3)
    ids = [int(v) for v in np.unique(lab)[1:]]
    cdir = os.path.join(exp, "Masks_CP")
    os.makedirs(cdir)
    Utility.save_mask(np.stack([lab] * t), cdir, batch=list(range(t)), z_size=1, frames=t, chan_seg="GFP")
    prop = Utility.open_exp_prop(exp)
    prop["channel_seg"] = {"Masks_Class": ["GFP", "RFP"]}
    prop["masks_process"]["classification"] = {"folder": "Masks_CP", "pos": ids[::2], "neg": ids[1::2]}
    Utility.save_exp_prop(exp, prop)
    a = Analysis(exp, "GFP")
    a.extract_channelData(imgFold="Images", maskFold="Masks_Class")
    df = a.df_analysis
    assert list(df.columns) == ["cell", "frames", "time", "mask_chan", "tag", "exp", "GFP", "RFP", "cell_class", "pos_cell"]
    assert (df["cell_class"] == "if cell is RFP").all()
    assert len(df) == t * len(ids)
    for _, d in df.iterrows():
        obj = int(d["cell"].rsplit("cell", 1)[1])
        assert d["pos_cell"] == (1 if obj in ids[::2] else 0)
        m = O.measure_labels_literal(stack[d["frames"] - 1, :, 0], lab)
        i = list(m["ids"]).index(obj)
        assert d["GFP"] == m["mean"][i, 0] and d["RFP"] == m["mean"][i, 1]


def test_smo_bg_sub_with_sigma(dev, tmp_path):
    """smo_bg_sub(sigma > 0) (utility.py:242, 252; kwarg from core.py:76): the null threshold and the planes go
    through the same gaussian pre-filter.  Threshold within fp32 of the oracle's; with the oracle's threshold
    the files equal the oracle's output."""
    from imageanalysis_pipeline_amd import ops
    t, c, y, x = 2, 1, 128, 256
    exp, _ = make_exp(tmp_path, t=t, c=c, y=y, x=x, chans=("GFP",))
    stack, _ = synth.make_stack(t, c, 1, y, x, n_cells=8, integer_drift=True, seed=9)
    _write_stack(exp, stack, ("GFP",))
    img = os.path.join(exp, "Images")
    sigma = 1.5
    thr = ops.smo_threshold((y, x), sigma=sigma, size=7, seed=0, device=dev)
    ethr = O.smo_threshold((y, x), sigma=sigma, size=7, seed=0)
    assert abs(thr - ethr) <= 2e-5 * ethr + 2e-6           # float32 noise image and filters vs float64
    prop = Utility.smo_bg_sub(imgFold_path=img, sigma=sigma, size=7)
    assert prop["fct_inputs"]["smo_bg_sub"]["smo_sigma"] == sigma
    for f in range(t):
        got = tiffio.imread(os.path.join(img, "GFP_f%04d_z0001.tif" % (f + 1)))
        eout, emed2, enbg, emask = O.bg_smo(stack[f, 0, 0], thr, sigma=sigma)
        # the mask may differ on pixels whose SMO value lies within the fp32 band of the threshold; the median
        # of ~5 % of 32k pixels does not move with a handful of them on this input
        assert np.array_equal(got, eout)


def test_staged_folder_io_round_trip(dev, tmp_path):
    """tiffio.read_planes_to_device / write_planes_from_device: pinned staging blocks, several blocks per call,
    a ragged last block (the shim's folder calls go through these)."""
    import torch
    rng = np.random.default_rng(31)
    planes = rng.integers(0, 65536, size=(23, 45, 70), dtype=np.uint16)
    paths = [str(tmp_path / ("GFP_f%04d_z0001.tif" % (i + 1))) for i in range(len(planes))]
    tiffio.write_planes(paths, planes)
    d = tiffio.read_planes_to_device(paths, dev, block_planes=5)
    torch.cuda.synchronize()
    assert d.dtype == torch.uint16 and d.is_cuda
    assert np.array_equal(d.view(torch.int16).cpu().numpy().view(np.uint16), planes)
    out = [str(tmp_path / ("out_f%04d_z0001.tif" % (i + 1))) for i in range(len(planes))]
    tiffio.write_planes_from_device(out, d, block_planes=4)
    assert np.array_equal(tiffio.read_planes(out), planes)
    one = tiffio.read_planes_to_device(paths[:1], dev)           # a single block, a single plane
    assert np.array_equal(one.view(torch.int16).cpu().numpy().view(np.uint16), planes[:1])
    with pytest.raises(ValueError):
        tiffio.read_planes_to_device([os.path.join(os.path.dirname(__file__), "golden", "tiff", "other_deflate.tif")], dev)
