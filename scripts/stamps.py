import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis_pipeline_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libiap_hip.so", "libiap_dbg.so")
import torch
from imageanalysis_pipeline_amd import ops
dev = torch.device("cuda:0")
n = 256
x = torch.randint(0, 60000, (n, 1, 2048, 2048), dtype=torch.int32, device=dev).to(torch.uint16)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
for name, sh in (("zero", (0.0, 0.0)), ("frac", (1.3, -2.6))):
    s = torch.tensor([sh] * n, dtype=torch.float64, device=dev)
    ops.apply_shifts(x, s); lib.iap_debug_stamps(buf, 1)
    ops.apply_shifts(x, s); lib.iap_debug_stamps(buf, 1)
    nwg = 32 * 64 * n
    print(name, [round(v / nwg) for v in buf[:5]], "cycles per WG: prologue, stage, xpass, ypass, store")
