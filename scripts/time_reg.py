import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
from imageanalysis_pipeline_amd import ops, _lib
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
up = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
x = torch.randint(0, 3000, (n, 2048, 2048), dtype=torch.int32, device=dev).to(torch.uint16)
lib = _lib.load()
ops.estimate_shifts(x[0], x, batch=batch, upsample=up); torch.cuda.synchronize()
lib.iap_prof_enable(1)
t0 = time.perf_counter()
for _ in range(3): ops.estimate_shifts(x[0], x, batch=batch, upsample=up)
torch.cuda.synchronize()
print(f"estimate n={n} batch={batch}: {(time.perf_counter() - t0) / 3 * 1e3 / n * 1e3:.1f} us/frame")
for k in range(lib.iap_prof_kernel_count()):
    ms = ctypes.c_double(); c = ctypes.c_int64()
    lib.iap_prof_collect(k, ctypes.byref(ms), ctypes.byref(c))
    if c.value: print(f"  {lib.iap_prof_kernel_name(k).decode():26s} {ms.value / 3 / n * 1e3:8.2f} us/frame  ({c.value // 3} launches)")
lib.iap_prof_enable(0)
