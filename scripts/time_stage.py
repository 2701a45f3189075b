"""Ad-hoc stage timing on the GPU box (development aid, not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageanalysis_pipeline_amd import ops

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randint(0, 60000, (n, 1, 2048, 2048), dtype=torch.int32, device=dev).to(torch.uint16)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for name, sh in (("zero", (0.0, 0.0)), ("int", (3.0, -5.0)), ("fracx", (0.0, 0.3)), ("fracy", (0.4, 0.0)), ("frac", (1.3, -2.6))):
    s = torch.tensor([sh] * n, dtype=torch.float64, device=dev)
    ms = timeit(lambda: ops.apply_shifts(x, s))
    print(f"apply {name:6s} {ms:8.3f} ms  {ms / n * 1e3:7.2f} us/plane  {n * 2048 * 2048 * 4 / ms / 1e9:8.1f} GB/s")
y = torch.empty_like(x)
ms = timeit(lambda: y.copy_(x))
print(f"torch copy   {ms:8.3f} ms  {n * 2048 * 2048 * 4 / ms / 1e9:8.1f} GB/s")

import numpy as np
rng = np.random.default_rng(0)
img = torch.from_numpy((rng.poisson(100, size=(8, 2048, 2048)) + 400).astype(np.int32)).to(dev).to(torch.uint16)
img = img.repeat(n // 8, 1, 1).contiguous()
out = torch.empty_like(img)
ms = timeit(lambda: ops.bg_smo(img, 0.045, out=out))
print(f"bg_smo       {ms:8.3f} ms  {ms / n * 1e3:7.2f} us/plane")
