"""Development aid: distance map + masked-pixel table timing against scipy's EDT (one plane on the host)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imageanalysis_pipeline_amd import ops, synth
dev = torch.device("cuda:0")
t, c, y, x = 32, 3, 2048, 2048
ref = np.zeros((t, y, x), np.uint16)
for f in range(t):
    ref[f, 700:1300 - 8 * f, 600 + 6 * f:1500] = 1          # a closing wound
cells = np.stack([synth.make_labels(y, x, 300, seed=1)] * t)
img = torch.randint(0, 60000, (t, c, y, x), dtype=torch.int32, device=dev).to(torch.uint16)
dref, dcell = ops.to_device_u16(ref, dev), ops.to_device_u16(cells, dev)
for name, fn in (("edt_sq", lambda: ops.edt_sq(dref)),):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); sq = fn(); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / t * 1e3:.3f} ms/plane")
ops.masked_pixels(dcell, img, sq, 0.65); torch.cuda.synchronize(); t0 = time.perf_counter()
cnt, dm, v = ops.masked_pixels(dcell, img, sq, 0.65); torch.cuda.synchronize()
print(f"masked_pixels: {(time.perf_counter() - t0) / t * 1e3:.3f} ms/frame, {int(cnt.sum())} rows")
import scipy.ndimage as ndi
t0 = time.perf_counter(); d = ndi.distance_transform_edt(ref[5] == 0); dt = time.perf_counter() - t0
print(f"scipy distance_transform_edt: {dt * 1e3:.1f} ms/plane; equal: {np.array_equal(np.sqrt(sq[5].cpu().numpy().astype(float)), d)}")
