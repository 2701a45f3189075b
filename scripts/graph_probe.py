"""Development aid (GPU): is the library safe under stream capture, and what does replaying the captured launches buy?
The serial `reg_ref='previous'` chain (seven dependent launches per frame) and the whole hot path on a C1-sized stack,
eager against a replayed HIP graph (torch.cuda.CUDAGraph is only the capture/replay carrier).
    python scripts/graph_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from imageanalysis_pipeline_amd import ops  # noqa: E402


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    t, c, y, x = 100, 2, 512, 512
    stack, labels, _ = bench.make_device_stack(t, c, 1, y, x, 60, dev, seed=1)
    thr = ops.smo_threshold((y, x), device=dev)
    out = torch.empty_like(stack)
    nlab = ops.max_label(labels) + 1
    for mode in ("previous", "mean", "first"):
        def eager():
            return ops.hot_path(stack, labels, thr, reg_ref=mode, out=out, n_labels=nlab)
        ref = eager()
        torch.cuda.synchronize()
        ms_eager = timeit(eager)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # scratch of the capturing stream is allocated before the capture
            eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            res = ops.hot_path(stack, labels, thr, reg_ref=mode, out=out, n_labels=nlab)
        torch.cuda.synchronize()
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(res[1], ref[1]) and all(torch.equal(a, b) for a, b in zip(res[0], ref[0]))
        ms_graph = timeit(g.replay)
        print(f"{mode:9s} eager {ms_eager:7.3f} ms   graph replay {ms_graph:7.3f} ms   same results: {same}", flush=True)


if __name__ == "__main__":
    main()
