"""Development aid: do bg-subtract and registration-estimate kernels overlap usefully on two streams?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from imageanalysis_pipeline_amd import ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
img = torch.from_numpy((rng.poisson(100, size=(8, 2048, 2048)) + 400).astype(np.int32)).to(dev).to(torch.uint16)
A = img.repeat(32, 1, 1).contiguous()          # 256 planes for bg
M = img.repeat(12, 1, 1).contiguous()          # 96 frames for the estimate
B = img.repeat(32, 1, 1).contiguous()          # 256 planes for apply
sh = torch.tensor([[1.3, -2.6]] * 256, dtype=torch.float64, device=dev)
outA = torch.empty_like(A)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def bg(): ops.bg_smo(A, 0.045, out=outA)
def est(): ops.estimate_shifts(M[0], M)
def app(): ops.apply_shifts(B, sh)
def run(fa, fb, concurrent):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        if concurrent:
            with torch.cuda.stream(s1): fa()
            with torch.cuda.stream(s2): fb()
        else:
            fa(); fb()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 * 1e3
for name, fa, fb in (("bg+est", bg, est), ("apply+est", app, est), ("bg+apply", bg, app)):
    run(fa, fb, False); run(fa, fb, True)
    print(f"{name}: serial {run(fa, fb, False):.2f} ms  two streams {run(fa, fb, True):.2f} ms")
