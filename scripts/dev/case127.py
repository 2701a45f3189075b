"""Development aid: the one channel-shift plane of the randomised run (seed 7, case 127) whose y shift differs from the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from imageanalysis_pipeline_amd import ops, synth
from oracle import ref_path as O
dev = torch.device("cuda:0")
t, c, z, y, x, regc, ncell, seed = 2, 3, 2, 122, 105, 1, 6, 347021399
rolls = [(-1, 1), (2, -2), (-2, -3)]
st, _ = synth.make_stack(t, c, z, y, x, n_cells=ncell, integer_drift=False, seed=seed)
for ch in range(c):
    st[:, ch] = np.roll(st[:, ch], rolls[ch], (-2, -1))
d = ops.to_device_u16(st, dev)
_, sh = ops.chan_shift(d, regc)
print("chan_shift", sh.cpu().numpy()[0, 0, 0], "oracle", O.phase_shift(st[0, regc, 0].astype(np.float64), st[0, 0, 0]))
ref, mov = d[0, regc, 0].contiguous(), d[0, 0, 0].contiguous()
print("pair alone", ops.estimate_pair_shifts(ref[None], mov[None]).cpu().numpy())
print("single-ref", ops.estimate_shifts(ref, mov[None]).cpu().numpy())
print("coarse    ", ops.estimate_shifts(ref, mov[None], upsample=False).cpu().numpy())
# the same planes padded into a 128-row frame (window 128 x 64): another Py
for pad in (6, 134):
    big = torch.zeros((2, y + pad, x), dtype=torch.uint16, device=dev)
    big[0, pad // 2: pad // 2 + y] = ref
    big[1, pad // 2: pad // 2 + y] = mov
    b = ops.to_numpy_u16(big)
    print("padded", y + pad, ops.estimate_shifts(big[0], big[1:2]).cpu().numpy(), "oracle", O.phase_shift(b[0].astype(np.float64), b[1]))
