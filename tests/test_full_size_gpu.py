"""BASELINE.json's headline configuration at its full size (C2: 500 x 4 x 2048 x 2048 u16 = 8.39 G pixels, 16.8 GB:
byte offsets beyond 4 GiB from frame 128 on, element offsets beyond 2^32 from frame 256 on), where the oracle cannot
follow: the last frames of the big call must equal a small call on the same frames bit for bit (`reg_ref='first'`: every
frame is registered against frame 0 on its own), and the size-independent checksums must hold over the whole stack.
"""
import numpy as np
import pytest
import torch

from imageanalysis_pipeline_amd import ops

pytestmark = pytest.mark.gpu


def test_c2_sized_stack_high_offsets(dev):
    import bench
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free device memory")
    t, c, z, y, x = 500, 4, 1, 2048, 2048
    stack, labels, _ = bench.make_device_stack(t, c, z, y, x, 300, dev, seed=5)
    thr = ops.smo_threshold((y, x), device=dev)
    out = torch.empty_like(stack)
    tables, shifts, levels = ops.hot_path(stack, labels, thr, reg_ref="first", out=out)
    torch.cuda.synchronize()
    # --- the frames at the far end (and two in the middle, either side of the 2^32-element mark) in a call of their own
    idx = [0, 255, 256] + list(range(t - 5, t))
    it = torch.tensor(idx, device=dev)
    take = lambda a: a.view(torch.int16)[it].contiguous().view(torch.uint16)   # noqa: E731  (torch indexes int16, not uint16)
    small, small_lab = take(stack), take(labels)
    small_out = torch.empty_like(small)
    tb2, sh2, lv2 = ops.hot_path(small, small_lab, thr, reg_ref="first", out=small_out)
    assert torch.equal(shifts[it], sh2)
    assert float(shifts[it].abs().max()) > 1.0                    # the drift walk has moved by then
    assert torch.equal(levels.view(t, c)[it], lv2.view(len(idx), c))
    assert torch.equal(take(out).view(torch.int16), small_out.view(torch.int16))
    for a, b in zip(tables, tb2):
        assert torch.equal(a[it], b)
    # --- checksums over the whole stack, computed by torch on the registered output
    area, mean, _, _ = tables
    nlab = ops.max_label(labels[0])
    assert area.shape == (t, nlab + 1)
    fg = int((labels[0].view(torch.int16) != 0).sum())
    assert torch.equal(area[:, 1:].sum(dim=1), torch.full((t,), fg, dtype=area.dtype, device=dev))
    assert torch.equal(area[:, 0], torch.full((t,), y * x - fg, dtype=area.dtype, device=dev))
    for f in (0, 127, 128, 300, t - 1):
        lab = labels[f, 0].view(torch.int16).to(torch.int64).view(-1)
        for ch in (0, c - 1):
            sums = torch.zeros(nlab + 1, dtype=torch.float64, device=dev)
            sums.index_add_(0, lab, (out[f, ch, 0].view(torch.int16).to(torch.int64) & 0xFFFF).view(-1).to(torch.float64))
            exp = sums[1:] / area[f, 1:].to(torch.float64)
            assert torch.equal(mean[f, 1:, ch], exp), (f, ch)
    # --- reg_ref='mean' over all 500 frames: the reference rebuilt by torch (float64 mean of the lowered registration
    # planes) and handed to the estimator gives the shifts of the far-end frames
    del tables, small, small_out, tb2
    tbm, shm, lvm = ops.hot_path(stack, labels, thr, reg_ref="mean", out=out)
    assert torch.equal(lvm, levels)
    med2 = lvm.view(t, c)[:, 0].to(torch.int64)
    lowered = lambda f: ((2 * (stack[f, 0, 0].view(torch.int16).to(torch.int64) & 0xFFFF) - med2[f]).clamp_(min=0) >> 1)   # noqa: E731
    acc = torch.zeros((y, x), dtype=torch.int64, device=dev)
    for f in range(t):
        acc += lowered(f)
    ref = (acc.to(torch.float64) / t).to(torch.float32)
    far = list(range(t - 4, t))
    mov = torch.stack([lowered(f).to(torch.int16) for f in far]).view(torch.uint16)     # values < 32768 here
    chk = ops.estimate_shifts(ref, mov)
    assert float((chk - shm[far]).abs().max()) < 2e-3, (chk, shm[far])
    assert torch.equal(tbm[0].sum(dim=1), torch.full((t,), y * x, dtype=tbm[0].dtype, device=dev))
    # every level is a background level of the synthetic planes (Poisson(100) + 400 -> 2 x ~500)
    lv = levels.cpu().numpy()
    assert lv.min() > 900 and lv.max() < 1100, (lv.min(), lv.max())
    assert np.isfinite(shifts.cpu().numpy()).all()


def test_c3_shaped_zstack_time_blocks_high_offsets(dev):
    """C3's plane geometry (2 channels x 31 z-planes of 2048^2, registered on the z-MIP, measured over z) over 40 frames
    = 10.4 G pixels (C3 itself is 200 frames, 104 GB: the bench runs it): processed 16 frames at a time, as C3 is, the
    frames beyond the 2^32-element mark equal a whole-stack call on those frames alone."""
    import bench
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 60 << 30:
        pytest.skip("needs 60 GB of free device memory")
    t, c, z, y, x = 40, 2, 31, 2048, 2048
    stack, labels, _ = bench.make_device_stack(t, c, z, y, x, 200, dev, seed=9)
    thr = ops.smo_threshold((y, x), device=dev)
    tables, shifts, levels = ops.hot_path(stack, labels, thr, reg_ref="first", tblock=16)
    torch.cuda.synchronize()
    idx = [0, 16, 17, t - 2, t - 1]          # frame 17 starts at element 4.4e9
    it = torch.tensor(idx, device=dev)
    take = lambda a: a.view(torch.int16)[it].contiguous().view(torch.uint16)   # noqa: E731
    tb2, sh2, lv2 = ops.hot_path(take(stack), take(labels), thr, reg_ref="first")
    assert torch.equal(shifts[it], sh2)
    assert torch.equal(levels.view(t, c * z)[it], lv2.view(len(idx), c * z))
    for a, b in zip(tables, tb2):
        assert torch.equal(a[it], b)
    area = tables[0]
    fg = int((labels[0].view(torch.int16) != 0).sum())
    assert torch.equal(area[:, 1:].sum(dim=1), torch.full((t,), fg, dtype=area.dtype, device=dev))
