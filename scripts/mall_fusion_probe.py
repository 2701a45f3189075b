"""Development aid (GPU): does running bg levels -> warp on groups of planes small enough for the 256 MiB Infinity
Cache beat whole-stack bg levels followed by a whole-stack warp?  Same kernels, same total work; only the order differs.
    python scripts/mall_fusion_probe.py [T] [C]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis_pipeline_amd import ops, _lib  # noqa: E402


def main():
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    c = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    y = x = 2048
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    stack = (torch.randint(400, 700, (t, c, 1, y, x), device=dev, generator=g, dtype=torch.int32)).to(torch.uint16)
    out = torch.empty_like(stack)
    shifts = (torch.rand((t, 2), device=dev, generator=g, dtype=torch.float64) - 0.5) * 6
    lib = _lib.load()
    thr = 0.0455
    st = torch.cuda.current_stream().cuda_stream

    def whole():
        med2, _ = ops.bg_smo_levels(stack.view(-1, y, x), thr)
        _lib.check(lib.iap_apply_shift_sub_u16(stack.data_ptr(), out.data_ptr(), t, c, y, x, shifts.data_ptr(),
                                               med2.data_ptr(), st), "apply")
        return med2

    def grouped(gf):
        px = c * y * x * 2
        allm = []
        for f0 in range(0, t, gf):
            n = min(gf, t - f0)
            sub = stack[f0:f0 + n]
            med2, _ = ops.bg_smo_levels(sub.view(-1, y, x), thr)
            _lib.check(lib.iap_apply_shift_sub_u16(stack.data_ptr() + f0 * px, out.data_ptr() + f0 * px, n, c, y, x,
                                                   shifts.data_ptr() + 16 * f0, med2.data_ptr(), st), "apply")
            allm.append(med2)
        return torch.cat(allm)

    def timeit(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r

    ms, ref = timeit(whole)
    ref_out = out.clone()
    print(f"whole stack ({t} x {c} planes): bg levels + warp {ms:.3f} ms", flush=True)
    for gf in (3, 6, 12, 24):
        ms, m = timeit(lambda: grouped(gf))
        same = bool(torch.equal(m, ref)) and bool(torch.equal(out, ref_out))
        print(f"groups of {gf:2d} frames ({gf * c * 8.39:.0f} MB): {ms:.3f} ms  identical={same}", flush=True)


if __name__ == "__main__":
    main()
