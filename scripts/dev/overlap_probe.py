"""Probe: do the VALU-bound background pass and the memory-bound estimate overlap when launched on two streams?
(GPU box)  python scripts/dev/overlap_probe.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imageanalysis_pipeline_amd import ops

dev = torch.device("cuda:0")
Y = X = 2048
g = torch.Generator(device=dev); g.manual_seed(0)
def planes(n):
    a = (torch.randn((n, Y, X), device=dev, generator=g) * 10 + 400).clamp_(0, 65535).to(torch.int32)
    return a.to(torch.int16).view(torch.uint16)
other = planes(1500)
regp = planes(500)
ref = regp[0].clone()
quad = planes(512).view(128, 4, 1, Y, X)
shifts = (torch.rand((128, 2), device=dev, dtype=torch.float64) - 0.5) * 6
labels = torch.zeros((128, 1, Y, X), dtype=torch.int16, device=dev).view(torch.uint16)
thr = 0.05
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3

def bg():
    with torch.cuda.stream(sB):
        ops.bg_smo_levels(other, thr)
def est():
    with torch.cuda.stream(sA):
        ops.estimate_shifts(ref, regp)
def app():
    with torch.cuda.stream(sB):
        ops.apply_shifts(quad, shifts)
def est128():
    with torch.cuda.stream(sA):
        ops.estimate_shifts(ref, regp[:128])
def meas():
    with torch.cuda.stream(sA):
        ops.measure_accumulate(quad, labels, 512)
def both(a, b):
    return lambda: (a(), b())

print("bg 1500 planes        %.2f ms" % timed(bg))
print("estimate 500 frames   %.2f ms" % timed(est))
print("both                  %.2f ms" % timed(both(est, bg)))
print("both (bg first)       %.2f ms" % timed(both(bg, est)))
print("apply 512 planes      %.2f ms" % timed(app))
print("estimate 128 frames   %.2f ms" % timed(est128))
print("both                  %.2f ms" % timed(both(est128, app)))
print("measure 128 frames    %.2f ms" % timed(meas))
print("apply+measure         %.2f ms" % timed(both(meas, app)))
