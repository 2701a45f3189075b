"""Rehearsal of distributed.process_plate: N ranks (gloo) sharing this box's GPUs, 5 small FOVs.
Run:  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 scripts/plate_rehearsal.py
Every rank checks the gathered table against a single-process run of all FOVs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from imageanalysis_pipeline_amd import ops, synth, distributed as D

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", rank % torch.cuda.device_count())
torch.cuda.set_device(dev)
t, c, y, x, n_fov = 3, 2, 128, 256, 5

def load_fov(i):
    st, _ = synth.make_stack(t, c, 1, y, x, n_cells=8, integer_drift=True, seed=10 + i)
    lab = np.stack([synth.make_labels(y, x, 8, seed=10 + i)[None]] * t)
    return ops.to_device_u16(st, dev), ops.to_device_u16(lab, dev)

thr = ops.smo_threshold((y, x), device=dev)
rows = D.process_plate(load_fov, n_fov, thr, reg_ref="first")
parts = []
for i in range(n_fov):
    st, lab = load_fov(i)
    bg, _, _ = ops.bg_smo(st, thr, out=st)
    reg, _ = ops.im_reg(bg, 0, "first")
    parts.append(D.pack_rows(ops.measure_finalize(ops.measure_accumulate(reg, lab)), i).cpu())
ref = torch.cat(parts, 0)
ok = rows.shape == ref.shape and bool((rows.cpu() == ref).all())
print(f"rank {rank}/{world}: shard {list(D.shard_fovs(n_fov, rank, world))}, gathered {tuple(rows.shape)}, identical to single-process: {ok}")
dist.destroy_process_group()
sys.exit(0 if ok else 1)
