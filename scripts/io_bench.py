"""Development aid: throughput of the batched native TIFF reader/writer (csrc/tiffio.hip) against the
per-file Python loop it replaces (utility.py:487-524), plus pinned-host -> device copy when a GPU is there."""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from imageanalysis_pipeline_amd import tiffio

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
root = sys.argv[2] if len(sys.argv) > 2 else tempfile.gettempdir()
d = tempfile.mkdtemp(dir=root)
try:
    rng = np.random.default_rng(0)
    planes = rng.integers(0, 65536, size=(n, 2048, 2048), dtype=np.uint16)
    paths = [os.path.join(d, f"GFP_s01_f{i + 1:04d}_z0001.tif") for i in range(n)]
    gb = planes.nbytes / 1e9
    t0 = time.perf_counter(); tiffio.write_planes(paths, planes); t1 = time.perf_counter()
    print(f"native write  {n} planes: {gb / (t1 - t0):6.2f} GB/s ({tiffio.IO_THREADS} threads)")
    t0 = time.perf_counter(); a = tiffio.read_planes(paths); t1 = time.perf_counter()
    assert np.array_equal(a, planes)
    print(f"native read   {n} planes: {gb / (t1 - t0):6.2f} GB/s")
    k = min(n, 32)
    t0 = time.perf_counter(); b = np.stack([tiffio.imread(p) for p in paths[:k]]); t1 = time.perf_counter()
    print(f"python loop   {k} planes: {b.nbytes / 1e9 / (t1 - t0):6.2f} GB/s (imread per file, tifffile={'yes' if tiffio._tf else 'no'})")
    import torch
    if torch.cuda.is_available():
        t0 = time.perf_counter()
        h = tiffio.read_planes(paths, pinned=True)
        dev = h.to("cuda", non_blocking=True); torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"native read into pinned memory + H2D: {gb / (t1 - t0):6.2f} GB/s end to end (read everything, then one copy)")
        del h, dev
        tiffio.read_planes_to_device(paths[:8], "cuda")            # warm the side stream / pinned allocator
        t0 = time.perf_counter()
        dev = tiffio.read_planes_to_device(paths, "cuda"); torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"staged read (pinned blocks, reads of block k+1 under the upload of block k): {gb / (t1 - t0):6.2f} GB/s folder -> device")
        assert np.array_equal(dev.view(torch.int16).cpu().numpy().view(np.uint16), planes)
        t0 = time.perf_counter()
        tiffio.write_planes_from_device(paths, dev)
        t1 = time.perf_counter()
        print(f"staged write (download of block k+1 under the file writes of block k): {gb / (t1 - t0):6.2f} GB/s device -> folder")
        assert np.array_equal(tiffio.read_planes(paths), planes)
finally:
    shutil.rmtree(d, ignore_errors=True)
