"""Rehearsal of distributed.process_fov_timeblocks: N ranks (gloo) sharing this box's GPUs, one FOV split
in time blocks, both shardable reference modes, 2-D and z-stack.
Run:  python -m torch.distributed.run --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29513 scripts/timeblock_rehearsal.py
Every rank checks rows and shifts against a single-process ops.im_reg run over all frames."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from imageanalysis_pipeline_amd import ops, synth, distributed as D

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", rank % torch.cuda.device_count())
torch.cuda.set_device(dev)
ok_all = True
# the third: fewer frames than ranks; the last: full-size planes, where the launch groups of the ranks and of the
# single process differ most (shifts do not depend on them: DESIGN.md section 2)
for (t, c, z, y, x) in ((7, 2, 1, 128, 256), (5, 2, 3, 128, 128), (2, 2, 1, 128, 128), (10, 1, 1, 2048, 2048)):
    st, _ = synth.make_stack(t, c, z, y, x, n_cells=8, integer_drift=False, seed=21)
    lab = np.stack([np.stack([synth.make_labels(y, x, 8, seed=21)] * z)] * t)
    thr = ops.smo_threshold((y, x), device=dev)

    def load_block(t0, t1):
        return ops.to_device_u16(st[t0:t1], dev), ops.to_device_u16(lab[t0:t1], dev)

    for mode in ("first", "mean"):
        rows, sh = D.process_fov_timeblocks(load_block, t, thr, reg_ref=mode)
        full, flab = load_block(0, t)
        bg, _, _ = ops.bg_smo(full, thr, out=full)
        reg, esh = ops.im_reg(bg, 0, mode)
        ref = D.pack_rows(ops.measure_finalize(ops.measure_accumulate(reg, flab)), 0).cpu()
        fr = D.shard_frames(t, rank, world)
        ok = rows.shape == ref.shape and bool((rows.cpu() == ref).all()) and bool((sh.cpu() == esh[fr.start:fr.stop].cpu()).all())
        ok_all &= ok
        print(f"rank {rank}/{world} z={z} {mode}: frames {list(fr)}, rows {tuple(rows.shape)}, identical to single-process: {ok}")
dist.destroy_process_group()
sys.exit(0 if ok_all else 1)
