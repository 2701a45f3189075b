#!/usr/bin/env python3
"""Headline benchmark: Mpx/s through background-subtract + register + per-label measure.

One "step" = one pass of the whole hot path over one resident field of view (ops.hot_path):
    iap_max_u16 (label range)  ->  iap_bg_smo_u16 (raw -> background level per plane)  ->
    iap_im_reg_sub_u16 (raw - level -> registered)  ->  iap_measure_labels_u16 + iap_measure_finalize (tables)
The background is subtracted as the registration kernels read the stack (same bits as writing the subtracted
stack first; `--two-pass` runs that form, and the default N=1 run reports its step time as `two_pass_ms_per_step`).
Workload at N=1: BASELINE.json configs[1] -- one FOV, 4 channels, 2048x2048 uint16, 500 frames with sub-pixel
drift ("c2").  `--workload c1|c3|c5` run the other single-GPU configurations (c3 in time blocks: its registered
copy does not fit beside the 104 GB stack).

Multi-GPU: `--gpus N` with no torch.distributed environment makes this process -- which has not touched a GPU --
start N ranks of itself with `python -m torch.distributed.run` and pass their line through; under the driver's own
launcher (WORLD_SIZE set) it is one of the ranks.  Every rank holds its own FOV (weak scaling, FOV sharding, no
data-path collective); the per-cell row tables are gathered to every rank with one RCCL all_gather inside the step.
`n_gpus` is the world size RCCL actually formed.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the dominant kernel (HIP
events round each of its launches in the timed steps, on the stream it is launched on, collected through
iap_prof_*; `traffic` from two rocprofv3 --pmc passes run right after the timed steps over the same library through
its native harness, scripts/pmc_traffic.py), `verified` (frames of the timed result checked against the oracle) and
`cpu_baseline` (the numpy oracle = restated reference CPU path, timed on this host on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X datasheet (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (T, C, Z, Y, X, n_cells, tblock)
    "c1": (100, 2, 1, 512, 512, 60, 0),        # BASELINE configs[0]
    "c2": (500, 4, 1, 2048, 2048, 300, 0),     # BASELINE configs[1]  (metric config)
    "c2small": (64, 4, 1, 2048, 2048, 300, 0),
    "c3": (200, 2, 31, 2048, 2048, 300, 20),   # configs[2]: 104 GB stack, registered + measured 20 frames at a time
    "c4fov": (300, 3, 1, 2048, 2048, 300, 0),  # one FOV of configs[3] (the plate is 64 of them, sharded over ranks)
    "c5": (64, 1, 1, 2048, 2048, 0, 0),        # configs[4]: 5041 labels per frame
    "c4": (300, 3, 1, 2048, 2048, 300, 0),     # configs[3] as the plate it is: 64 such FOVs sharded over the ranks (bench_plate)
}
# stage lines of the smoothing subsystem (north_star (1)): not the headline metric, one kernel each
SMOOTHING = {
    # name: (planes, Y, X, harness stage, dominant kernel)
    "blur": (2000, 2048, 2048, "blur", "gauss_march_kernel"),           # blur_img, the reference's default (15, 15), sigma 5
    "smosigma": (500, 2048, 2048, "smosigma", "gauss_reflect_tile_kernel"),   # SMO(sigma=1)'s gaussian pre-filter + levels
}


# algorithmic (compulsory) HBM bytes per stack pixel, SURVEY.md section 8(d)
def stage_bytes_per_px(c: int, two_pass: bool = False) -> dict:
    # bg: read the stack (2 B) and, in the two-pass form only, write the subtracted stack (2 B)
    return {"bg": 4.0 if two_pass else 2.0, "estimate": 2.0 / c, "apply": 4.0, "measure": 2.0 + 2.0 / c}


STAGE_OF_KERNEL = {
    "smo_march_kernel": "bg", "smo_generic_kernel": "bg", "median_kernel": "bg", "sub_half_kernel": "bg",
    "rows_r2c16_kernel": "estimate", "rows_r2c_kernel": "estimate", "cols_xpower_kernel": "estimate", "cols_ref_kernel": "estimate",
    "cols_fused_kernel": "estimate", "rows_c2r_argmax_kernel": "estimate", "upsample_tables_kernel": "estimate",
    "coarse_finish_kernel": "estimate", "upsample_partial_kernel": "estimate", "upsample_finish_kernel": "estimate",
    "mip_kernel": "estimate", "accum_planes_kernel": "estimate",
    "apply_shift_kernel": "apply", "measure_kernel": "measure", "measure_finalize_kernel": "measure",
    "max_u16_kernel": "measure",
}
HARNESS_OF_STAGE = {"apply": lambda t, c, z, y, x: ["apply", t, c * z, y, x],
                    "bg": lambda t, c, z, y, x: ["bg", t * c * z, y, x],
                    "measure": lambda t, c, z, y, x: ["measure", t, c, y, x],
                    "estimate": lambda t, c, z, y, x: ["estimate", t, y, x]}


def spawn_ranks(n: int) -> int:
    """This process has made no GPU call: start n ranks of this script under torch's launcher (one process per
    GPU, RCCL rendezvous on 127.0.0.1) and return their exit code.  Never an exec of a process that holds a GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def make_device_stack(t, c, z, y, x, n_cells, device, seed=0, subpixel=True, label_frames=None):
    """Synthetic stack generated on the device (torch is only the data generator here): Poisson(100)+400
    background, Gaussian-blob cells, cumulative random-walk drift applied by Fourier shift before
    quantisation.  Returns (stack u16 (T,C,Z,Y,X), labels u16 (T,Z,Y,X), drift)."""
    from imageanalysis_pipeline_amd import ops, synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if n_cells > 0:
        base = torch.from_numpy(synth.clean_frame(y, x, n_cells, seed)).to(device=device, dtype=torch.float32)
        lab2d = synth.make_labels(y, x, n_cells, seed)
    else:
        base = torch.zeros((y, x), device=device)
        lab2d = synth.make_dense_labels(y, x)
    drift = synth.drift_walk(t, not subpixel, 1)
    fb = torch.fft.rfft2(base)
    fy = torch.fft.fftfreq(y, device=device)[:, None]
    fx = torch.fft.rfftfreq(x, device=device)[None, :]
    stack = torch.empty((t, c, z, y, x), dtype=torch.uint16, device=device)
    zf = [1.0 - 0.6 * abs(k - (z - 1) / 2.0) / max(1.0, (z - 1) / 2.0) for k in range(z)] if z > 1 else [1.0]
    lam = torch.full((y, x), 100.0, device=device)
    for f in range(t):
        ph = torch.exp(-2j * np.pi * (fy * float(drift[f, 0]) + fx * float(drift[f, 1])))
        sh = torch.fft.irfft2(fb * ph, s=(y, x)).clamp_(min=0)
        for ch in range(c):
            for k in range(z):
                v = torch.poisson(lam, generator=g) + 400.0 + sh * ((1 + 0.25 * ch) * zf[k])
                stack[f, ch, k] = v.round_().clamp_(0, 65535).to(torch.int32).to(torch.uint16)
    lab = ops.to_device_u16(lab2d, device)
    labels = lab[None, None].expand(t, z, y, x).contiguous()
    return stack, labels, drift


class Path:
    """The hot path on one resident FOV."""

    def __init__(self, stack, labels, reg_ref="mean", batch=None, two_pass=False, tblock=0):
        from imageanalysis_pipeline_amd import ops
        self.ops = ops
        self.raw, self.labels = stack, labels
        self.bg = None
        self.reg_ref, self.batch, self.two_pass, self.tblock = reg_ref, batch, two_pass, tblock
        t, c, z, y, x = stack.shape
        self.reg = torch.empty((tblock if tblock else t, c, z, y, x), dtype=torch.uint16, device=stack.device)
        # the null threshold of SMO(shape, sigma, size) is a property of the plane shape, drawn once per FOV folder
        # in the reference too (utility.py:252, outside its per-file loop): not part of the step
        self.thr = ops.smo_threshold((y, x), device=stack.device)

    def step(self):
        ops = self.ops
        if self.two_pass:
            if self.bg is None:
                self.bg = torch.empty_like(self.raw)
            _, self.levels, _ = ops.bg_smo(self.raw, self.thr, out=self.bg)
            _, self.shifts = ops.im_reg(self.bg, 0, self.reg_ref, upsample=True, batch=self.batch, out=self.reg)
            self.tables = ops.measure_finalize(ops.measure_accumulate(self.reg, self.labels))
        else:
            self.tables, self.shifts, self.levels = ops.hot_path(self.raw, self.labels, self.thr, 0, self.reg_ref,
                                                                 batch=self.batch, out=self.reg, tblock=self.tblock or None)
        return self.tables


def collect_kernel_times():
    import ctypes
    from imageanalysis_pipeline_amd import _lib
    lib = _lib.load()
    out = {}
    for k in range(lib.iap_prof_kernel_count()):
        ms = ctypes.c_double()
        n = ctypes.c_int64()
        _lib.check(lib.iap_prof_collect(k, ctypes.byref(ms), ctypes.byref(n)), "iap_prof_collect")
        if n.value:
            out[lib.iap_prof_kernel_name(k).decode()] = (ms.value, n.value)
    return out


def stream_copy_ceiling(dev, nbytes=1 << 30):
    """Measured device copy rate (read + write bytes per second): the practical HBM ceiling that SURVEY 8(d)
    asks to report beside the 8 TB/s datasheet peak.  torch's copy kernel is only the yardstick here."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * 5 / (e0.elapsed_time(e1) / 1e3) / 1e9


def verify(path, frames=(0, 1)):
    """Ties the timed result to the oracle, stage by stage, on a few frames of the SAME device stack (the oracle
    cannot run 500 frames): background levels exact; shift within 0.01 px of the oracle's estimate against the
    same reference plane; warped pixels within 1 count of the oracle's warp at the same shift; tables bit-exact
    against the oracle's measurement of the registered frame.  Raises AssertionError on a miss."""
    from oracle import ref_path as O
    ops = path.ops
    raw, lab = path.raw, path.labels
    t, c, z, y, x = raw.shape
    frames = [f for f in frames if f < t]
    levels = path.levels.view(t, c, z).cpu().numpy()
    shifts = path.shifts.cpu().numpy()
    area, mean, cen, cen_r = (a.cpu().numpy() for a in path.tables)
    rep = {"frames": frames}
    # reference plane of the registration channel: the lowered MIP planes summed by torch (checker only)
    lv = ((path.levels.view(t, c, z)[:, 0] + 1) >> 1).clamp(0, 65535)
    if path.reg_ref == "mean":
        ref = torch.zeros((y, x), dtype=torch.float64, device=raw.device)
        for f in range(t):
            ref += (raw[f, 0].to(torch.int32) - lv[f].view(z, 1, 1)).clamp_(min=0).amax(dim=0).double()
        ref = (ref / t).cpu().numpy()
    elif path.reg_ref == "first":
        ref = (raw[0, 0].to(torch.int32) - lv[0].view(z, 1, 1)).clamp_(min=0).amax(dim=0).double().cpu().numpy()
    else:
        ref = None
    worst_shift, worst_px = 0.0, 0
    shift_checked = warp_checked = False
    for f in frames:
        host = ops.to_numpy_u16(raw[f])                                # (C, Z, Y, X)
        sub = np.empty_like(host)
        for ch in range(c):
            for k in range(z if z <= 2 else 1):                        # z-stacks: the first plane of each channel
                out, med2, _, _ = O.bg_smo(host[ch, k], path.thr)
                assert int(levels[f, ch, k]) == med2, f"background level of frame {f} channel {ch}: {levels[f, ch, k]} vs oracle {med2}"
                sub[ch, k] = out
        if ref is not None and not (path.reg_ref == "first" and f == 0):
            # the moving plane of the estimate: the registration channel lowered by the GPU's levels (checked exact
            # above on the planes the oracle recomputed), z-MIP for z-stacks (utility.py:341, 366)
            mov = (raw[f, 0].to(torch.int32) - lv[f].view(z, 1, 1)).clamp_(min=0).amax(dim=0).double().cpu().numpy()
            dy, dx = O.phase_shift(ref, mov)
            worst_shift = max(worst_shift, abs(dy - shifts[f, 0]), abs(dx - shifts[f, 1]))
            shift_checked = True
        if path.tblock:
            continue                                                    # the registered block buffer is reused
        warp_checked = True
        reg = ops.to_numpy_u16(path.reg[f])
        for ch in sorted({0, c - 1}):
            e = O.apply_shift(sub[ch, 0], shifts[f, 0], shifts[f, 1])
            worst_px = max(worst_px, int(np.abs(e.astype(np.int64) - reg[ch, 0].astype(np.int64)).max()))
        e = O.measure_labels(reg if z > 1 else reg[:, 0], ops.to_numpy_u16(lab[f]) if z > 1 else ops.to_numpy_u16(lab[f, 0]))
        ids = e["ids"]
        assert np.array_equal(np.nonzero(area[f])[0][1:], ids), f"label ids of frame {f}"
        assert np.array_equal(area[f, ids], e["area"]), f"areas of frame {f}"
        assert np.array_equal(mean[f, ids].view(np.uint64), e["mean"].view(np.uint64)), f"means of frame {f} are not bit-exact"
        sl = slice(0, 3) if z > 1 else slice(1, 3)
        assert np.array_equal(cen_r[f, ids][:, sl], e["centroid_rne"]), f"centroids of frame {f}"
    assert worst_shift <= 0.01, f"shift differs from the oracle by {worst_shift} px"
    assert worst_px <= 1, f"warped pixels differ from the oracle by {worst_px} counts"
    rep.update({"background_levels": "exact" if z <= 2 else "exact (first plane of each channel)",
                "max_shift_error_px": worst_shift if shift_checked else "not checked",
                "max_warp_error_counts": worst_px if warp_checked else "not checked (block buffer reused)",
                "tables": "bit-exact (ids, areas, f64 means, rounded centroids)" if warp_checked else "not checked (block buffer reused)",
                "oracle": "oracle/ref_path.py on the frames listed, same device stack as the timed steps"})
    return rep


def cpu_baseline(st, lab, thr, reg_ref="mean"):
    """The oracle (numpy restatement of the reference CPU path) on a bounded sample of the workload: the first
    frames of the very stack the GPU steps ran on (`st` (t, C, 1, Y, X), `lab` (t, 1, Y, X), host copies), the same
    null threshold, the same stages."""
    from oracle import ref_path as O
    budget_frames, c, _, y, x = st.shape
    t0 = time.perf_counter()
    O.run_path(st, lab, thr, reg_ref=reg_ref)
    dt = time.perf_counter() - t0
    px = st.size
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    # the measure stage above is the vectorised bincount restatement; the reference's literal
    # per-label loop (analysis.py:80-98, one full-frame compare per label and channel) is timed on its own,
    # on one frame, so that the baseline is not inflated by it
    t1 = time.perf_counter()
    O.measure_labels(st[0, :, 0], lab[0, 0])
    t2 = time.perf_counter()
    O.measure_labels_literal(st[0, :, 0], lab[0, 0])
    t3 = time.perf_counter()
    return {"value": px / dt / 1e6, "unit": "Mpx/s", "cores": 1, "kind": "port",
            "sample": f"frames 0..{budget_frames - 1} x {c} ch x {y}x{x} of the stack the GPU steps ran on, same threshold, numpy oracle "
                      f"(single process; numpy/pocketfft threads: default; host exposes {cores} cores), {dt:.1f} s",
            "measure_s_per_frame": {"vectorised_bincount": t2 - t1, "reference_literal_loop": t3 - t2}}


def measure_bytes_touched(labels, c):
    """HBM bytes the measuring pass must touch per step: every label, and the C image vectors (8 pixels, 16 bytes)
    of each run of 8 pixels that holds a label -- the kernel skips the image loads of all-background runs, so
    crediting it SURVEY 8(d)'s 2 + 2/C B/px on sparse masks would put its fraction above 1."""
    n = labels.numel()
    lab8 = labels.reshape(-1)[: n - n % 8].view(-1, 8)
    runs = int((lab8.view(torch.int16) != 0).any(dim=1).sum().item()) + (1 if n % 8 else 0)
    return 2.0 * n + 16.0 * c * runs


def stage_traffic(dims, stages=("bg", "estimate", "apply", "measure")):
    """Counter traffic (FETCH_SIZE x 2 + WRITE_SIZE) of every stage, summed over its kernels, for one pass of the
    benchmark's shape through the library's native harness.  {stage: bytes} or (None, reason)."""
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None, "running under a profiler: counter passes skipped"
    try:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import pmc_traffic
        out, per_kernel = {}, {}
        for st in stages:
            got = pmc_traffic.measure(HARNESS_OF_STAGE[st](*dims), timeout=150)
            out[st] = sum(v["hbm_bytes_per_harness_call"] for v in got.values())
            per_kernel.update(got)
        return (out, per_kernel), None
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"[:300]


def bench_plate(args, rank, world, dev, backend):
    """`--workload c4`: BASELINE configs[3], the 64-FOV plate (core.py:99-112, 327-337 are the loops it replaces).
    One step = the whole plate through distributed.process_plate: every rank runs the hot path on its block of
    FOVs, one variable-length gather of the per-cell rows per step.  A rank keeps `--resident-fovs` distinct
    synthetic FOVs in HBM and maps its block onto them cyclically (at 8 ranks a block is 8 FOVs = 60 GB and all can
    be resident; at 1 rank the 483 GB plate cannot, so the resident ones are each processed several times -- the
    work of 64 FOVs, inputs resident as the metric asks).  N = 1 also times the STREAMED form on a few FOVs: pinned
    host copies uploaded into two device buffers on a side stream while the previous FOV computes (PCIe-inclusive,
    reported beside `value`, never as it)."""
    import torch.distributed as dist
    from imageanalysis_pipeline_amd import distributed as D, ops
    t, c, z, y, x, n_cells, _ = WORKLOADS["c4"]
    if args.frames:
        t = args.frames
    n_fov = args.fovs
    mine = list(D.shard_fovs(n_fov, rank, world))
    nres = max(1, min(len(mine), args.resident_fovs))
    resident = [make_device_stack(t, c, z, y, x, n_cells, dev, seed=1000 + mine[k])[:2] for k in range(nres)] if mine else []
    thr = ops.smo_threshold((y, x), device=dev)

    def load(i, slot=None):
        return resident[(i - mine[0]) % nres]

    def step():
        return D.process_plate(load, n_fov, thr, reg_ref=args.reg_ref)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    px_fov = t * c * z * y * x
    value = px_fov * n_fov * args.steps / dt / 1e6
    bpp = sum(stage_bytes_per_px(c).values())
    out = None
    if rank == 0:
        out = {"metric": "Mpx/s through bg-sub+registration+mask measure", "value": value, "unit": "Mpx/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
               "config": {"workload": f"c4: plate of {n_fov} FOVs, T={t} C={c} {y}x{x} uint16 each, sub-pixel drift, {n_cells} labels/frame, "
                                      f"reg_ref={args.reg_ref}; FOVs block-sharded over {world} rank(s), {nres} distinct FOV(s) resident per rank",
                          "parallelism": f"fov-shard x{world}, one all_gather of the row tables per step"},
               "rows_gathered": int(rows.shape[0]), "ms_per_fov": dt / args.steps * 1e3 / max(len(mine), 1),
               "hbm_frac_end_to_end": value * 1e6 / world * bpp / HBM_PEAK, "algorithmic_bytes_per_px": bpp}
    if world == 1 and not args.no_verify:
        # FOV 0's rows equal a direct call of the hot path on it (bit for bit); the path itself against the oracle
        tables, _, _ = ops.hot_path(resident[0][0], resident[0][1], thr, 0, args.reg_ref)
        direct = D.pack_rows(tables, mine[0])
        got = rows[rows[:, 0] == float(mine[0])]
        assert got.shape == direct.shape and bool((got == direct).all()), "plate rows of FOV 0 differ from a direct call"
        pth = Path(resident[0][0], resident[0][1], args.reg_ref)
        pth.step()
        out["verified"] = verify(pth)
        out["verified"]["plate"] = "rows of the first FOV equal a direct ops.hot_path call on it, bit for bit"
        del pth
    if world == 1 and args.streamed_fovs > 0:
        # the streamed form: host-resident FOVs (pinned), two device slots, upload of FOV k+1 under the compute of FOV k
        ns = args.streamed_fovs
        host = [(st.cpu().pin_memory(), lab.cpu().pin_memory()) for st, lab in resident[: min(nres, 2)]]
        slots = [(torch.empty_like(resident[0][0]), torch.empty_like(resident[0][1])) for _ in range(2)]

        def upload(i, slot=0):
            hs, hl = host[i % len(host)]
            slots[slot][0].copy_(hs, non_blocking=True)
            slots[slot][1].copy_(hl, non_blocking=True)
            return slots[slot]

        D.process_plate(upload, 2, thr, reg_ref=args.reg_ref, prefetch=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        srows = D.process_plate(upload, ns, thr, reg_ref=args.reg_ref, prefetch=True)
        torch.cuda.synchronize()
        sdt = time.perf_counter() - t1
        t2 = time.perf_counter()
        D.process_plate(upload, ns, thr, reg_ref=args.reg_ref, prefetch=False)
        torch.cuda.synchronize()
        sdt_serial = time.perf_counter() - t2
        fov_bytes = resident[0][0].numel() * 2 + resident[0][1].numel() * 2
        out["streamed"] = {"fovs": ns, "ms_per_fov": sdt / ns * 1e3, "ms_per_fov_without_overlap": sdt_serial / ns * 1e3,
                           "host_to_device_GBps": fov_bytes * ns / sdt / 1e9, "Mpx_per_s": px_fov * ns / sdt / 1e6,
                           "rows": int(srows.shape[0]),
                           "note": "PCIe-inclusive: pinned host FOVs -> two device slots on a side stream, FOV k+1 uploading "
                                   "while FOV k computes (distributed.process_plate(prefetch=True)); the 483 GB plate does not "
                                   "fit one GPU, this is what a 1-GPU run of it costs per FOV"}
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def bench_smoothing(args):
    """`--workload blur | smosigma`: one stage of the smoothing subsystem on resident planes (utility.py:433-468 with
    the defaults of experiments.py:337; SMO's sigma, utility.py:252), checked against the oracle on one plane, with
    the kernel's counter traffic from the native harness.  One JSON line; not the headline metric."""
    from imageanalysis_pipeline_amd import ops
    from oracle import ref_path as O
    n, y, x, hstage, kname = SMOOTHING[args.workload]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    planes = (torch.poisson(torch.full((8, y, x), 100.0, device=dev), generator=g) + 400.0).to(torch.int32).to(torch.uint16)
    planes = planes.repeat((n + 7) // 8, 1, 1)[:n].contiguous()
    if args.workload == "blur":
        step = lambda: ops.gauss_blur(planes, (15, 15), 5.0)          # noqa: E731
        got = ops.to_numpy_u16(step()[3])
        assert np.array_equal(got, O.gauss_blur(ops.to_numpy_u16(planes[3]), (15, 15), 5.0)), "blur differs from the oracle"
        bpp, what = 4.0, "blur_img (15, 15) sigma 5: u16 -> u16, bit-exact vs the oracle on plane 3"
    else:
        thr = ops.smo_threshold((y, x), sigma=1.0, device=dev)
        step = lambda: ops.bg_smo_levels(planes, thr, sigma=1.0)       # noqa: E731
        lv = step()[0]
        e = O.bg_smo(ops.to_numpy_u16(planes[3]), thr, 1.0, 7)[1]
        assert abs(int(lv[3].item()) - e) <= 1, "level differs from the oracle"
        bpp, what = 2.0, "smo_bg_sub levels with SMO(sigma=1): gaussian pre-filter + SMO march + median, level of plane 3 within half a count of the oracle"
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    px = n * y * x
    out = {"metric": f"Mpx/s through {args.workload} (stage line, not the headline metric)", "value": px / ms / 1e3, "unit": "Mpx/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u16 planes, f32 arithmetic", "data": "synthetic",
           "config": {"workload": f"{args.workload}: {n} planes of {y}x{x} uint16; {what}"},
           "hbm_frac_stage": bpp * px / (ms / 1e3) / HBM_PEAK, "algorithmic_bytes_per_px": bpp}
    if not args.no_traffic:
        del planes
        torch.cuda.empty_cache()
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import pmc_traffic
            hn = min(n, 500)
            got = pmc_traffic.measure([hstage, hn, y, x], timeout=150)
            out["counter_traffic"] = {k: {"hbm_bytes_per_plane": v["hbm_bytes_per_harness_call"] / hn,
                                          "fetch_bytes_per_plane": v["fetch_bytes_per_launch"] * v["hbm_bytes_per_harness_call"] / max(v["hbm_bytes_per_launch"], 1) / hn}
                                      for k, v in got.items()}
            out["harness_ms_per_call"] = next(iter(got.values())).get("harness_ms_per_call_under_profiler")
        except Exception as e:  # noqa: BLE001
            out["counter_traffic_note"] = f"{type(e).__name__}: {e}"[:300]
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + sorted(SMOOTHING))
    ap.add_argument("--frames", type=int, default=0, help="override T")
    ap.add_argument("--tblock", type=int, default=-1, help="frames registered + measured at a time (0: whole stack)")
    ap.add_argument("--reg-ref", default="mean", choices=["first", "mean", "previous"],
                    help="reference frame of the registration; 'mean' is the reference's default (core.py:77) and what its notebook passes")
    ap.add_argument("--batch", type=int, default=0, help="frames per registration launch group (0: the library wrapper's choice, 128 at 2048^2)")
    ap.add_argument("--two-pass", action="store_true", help="write the background-subtracted stack, then register it")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-traffic", action="store_true")
    ap.add_argument("--stage-report", action="store_true", help="also print a per-kernel table to stderr")
    ap.add_argument("--fovs", type=int, default=64, help="c4: fields of view on the plate")
    ap.add_argument("--resident-fovs", type=int, default=4, help="c4: distinct synthetic FOVs a rank keeps in HBM")
    ap.add_argument("--streamed-fovs", type=int, default=8, help="c4, N=1: FOVs of the streamed (host -> device) form; 0: skip")
    args = ap.parse_args()

    if args.workload in SMOOTHING:
        return bench_smoothing(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))          # parent of the ranks: no GPU call has been made in this process

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library is the only implementation)")
    from imageanalysis_pipeline_amd import _lib, distributed
    # one rank per GPU.  IAP_BENCH_BACKEND=gloo is a rehearsal hook for boxes with fewer GPUs than
    # ranks (ranks then share devices and the table gather goes through host memory).
    backend = os.environ.get("IAP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({ndev} visible); one rank per GPU")
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()             # what the backend actually formed
        if rank == 0 and world != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but the launcher formed a world of {world}", file=sys.stderr)

    if args.workload == "c4":
        return bench_plate(args, rank, world, dev, backend)
    t, c, z, y, x, n_cells, tblock = WORKLOADS[args.workload]
    if args.frames:
        t = args.frames
    if args.tblock >= 0:
        tblock = args.tblock
    if tblock >= t:
        tblock = 0
    if args.two_pass or args.reg_ref == "previous":
        tblock = 0
    stack, labels, _ = make_device_stack(t, c, z, y, x, n_cells, dev, seed=rank)
    path = Path(stack, labels, args.reg_ref, args.batch or None, args.two_pass, tblock)
    lib = _lib.load()

    def one_step():
        tables = path.step()
        if world > 1:
            return distributed.gather_rows(distributed.pack_rows(tables, rank))
        return tables

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    timing = not args.no_kernel_timing
    ktimes_all = {}
    if timing:
        # One more untimed step with HIP events round every launch: the per-stage table, and which kernel is the
        # dominant one.  The timed steps then carry events round that kernel only (two per launch), so `value`
        # is not taxed by ~800 event records a step.
        lib.iap_prof_select(-1)
        lib.iap_prof_enable(1)
        one_step()
        torch.cuda.synchronize()
        ktimes_all = collect_kernel_times()
        lib.iap_prof_enable(0)
        if ktimes_all:
            dom_name = max(ktimes_all.items(), key=lambda kv: kv[1][0])[0]
            names = [lib.iap_prof_kernel_name(k).decode() for k in range(lib.iap_prof_kernel_count())]
            lib.iap_prof_select(names.index(dom_name))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if timing:
        lib.iap_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ktimes = collect_kernel_times() if timing else {}
    lib.iap_prof_enable(0)
    lib.iap_prof_select(-1)

    px_step = t * c * z * y * x
    value = px_step * world * args.steps / dt / 1e6
    bpp = stage_bytes_per_px(c, args.two_pass)
    total_bpp = sum(bpp.values())
    # what the stages must touch: SURVEY's figures, but the measuring pass only the labels and the labelled runs
    touched_bpp = dict(bpp)
    touched_bpp["measure"] = min(bpp["measure"], measure_bytes_touched(labels, c) / px_step)

    roof = None
    stages = {}
    if ktimes:
        for name, (ms, n) in ktimes_all.items():   # the profiled step before the timed region
            s = STAGE_OF_KERNEL.get(name)
            if s:
                stages.setdefault(s, 0.0)
                stages[s] += ms
        dom = max(ktimes.items(), key=lambda kv: kv[1][0])   # the timed steps: the dominant kernel only
        dname, (dms, dn) = dom
        dstage = STAGE_OF_KERNEL.get(dname, "bg")
        # algorithmic bytes of the dominant kernel's stage, per launch of that kernel
        alg_bytes_launch = bpp[dstage] * px_step * args.steps / dn
        avg_s = dms / dn / 1e3
        achieved = alg_bytes_launch / avg_s
        roof = {"bound": "hbm", "kernel": dname, "stage": dstage, "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": None,
                "launches": dn, "avg_launch_ms": dms / dn,
                "algorithmic_bytes_per_launch": alg_bytes_launch,
                "note": "achieved = stage algorithmic bytes (SURVEY 8d: %.3g B/px x stack px) / launches / mean launch "
                        "duration of the stage's dominant kernel (HIP events over the timed steps; the stage table "
                        "comes from one step profiled in full before them)" % bpp[dstage]}
        if args.stage_report and rank == 0:
            for name, (ms, n) in sorted(ktimes_all.items(), key=lambda kv: -kv[1][0]):
                print(f"  {name:28s} {ms:9.3f} ms/step  {n:6d} launches/step", file=sys.stderr)

    out = None
    if rank == 0:
        out = {
            "metric": "Mpx/s through bg-sub+registration+mask measure",
            "value": value, "unit": "Mpx/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": f"{args.workload}: one FOV per GPU, T={t} C={c} Z={z} {y}x{x} uint16, sub-pixel drift, "
                                   f"{n_cells or 5041} labels/frame, reg_ref={args.reg_ref}"
                                   + (f", registered and measured {tblock} frames at a time" if tblock else ""),
                       "stages": "bg_smo + im_reg(estimate+apply) + measure_labels (label range found by the measuring pass)", "parallelism": f"fov-shard x{world}",
                       "bg_subtraction": "written as a stack of its own, then registered" if args.two_pass else
                                         "applied on load by the registration kernels (same bits, no intermediate stack)"},
            "hbm_frac_end_to_end": value * 1e6 / world * total_bpp / HBM_PEAK,
            "algorithmic_bytes_per_px": total_bpp,
            "hbm_frac_bytes_touched": value * 1e6 / world * sum(touched_bpp.values()) / HBM_PEAK,
            "bytes_touched_per_px": sum(touched_bpp.values()),
            "accounting": "hbm_frac_end_to_end prices a pixel at SURVEY 8(d)'s bytes (bg counted as its read only when "
                          "subtracted on load); hbm_frac_bytes_touched charges the measuring pass only the labels and the "
                          "image vectors of labelled 8-pixel runs (it skips the rest); hbm_frac_counter_traffic is the sum of "
                          "the stages' rocprofv3 counter bytes (FETCH_SIZE x 2 + WRITE_SIZE, native harness, same shapes) "
                          "over the step time: how busy the memory system really is, scratch included",
        }
        if stages:
            out["stage_ms_per_step"] = dict(stages)
            out["stage_hbm_frac"] = {k: touched_bpp[k] * px_step / (v / 1e3) / HBM_PEAK for k, v in stages.items()}
            out["stage_bytes_per_px"] = touched_bpp
        if roof:
            out["roofline"] = roof
        if world == 1 and not args.no_verify:
            out["verified"] = verify(path)            # raises (non-zero exit, no line) when the result is wrong
        if world == 1 and not args.two_pass and not tblock:
            # the same path with the subtracted stack written out and read back: same outputs, one more pass
            path.two_pass = True
            path.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                path.step()
            torch.cuda.synchronize()
            out["two_pass_ms_per_step"] = (time.perf_counter() - t1) / args.steps * 1e3
            path.two_pass = False
            path.bg = None
        if world == 1:
            out["stream_copy_ceiling_GBps"] = stream_copy_ceiling(dev)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        sample = None
        if world == 1 and not args.no_cpu_baseline:
            # the CPU baseline's sample: the first frames of this very stack (about 100 Mpx: 10-30 s of numpy), for
            # z-stacks their first three planes; copied out before the stacks are freed
            zz = min(z, 3)
            nf = max(1, min(t, int(100e6 // (c * zz * y * x))))
            from imageanalysis_pipeline_amd import ops as _ops
            sample = (_ops.to_numpy_u16(stack[:nf, :, :zz]), _ops.to_numpy_u16(labels[:nf, :zz]), float(path.thr))
        if world == 1 and roof and not args.no_traffic:
            # free the stacks first: the harness allocates its own copy of the workload on the same GPU
            del path, stack, labels
            torch.cuda.empty_cache()
            dims = (t if not tblock else tblock, c, z, y, x)
            got, why = stage_traffic(dims)
            if got:
                per_stage, per_kernel = got
                scale = t / dims[0]                     # time blocks: the harness ran one block
                step_bytes = sum(per_stage.values()) * scale
                out["hbm_frac_counter_traffic"] = step_bytes / (out["ms_per_step"] / 1e3) / HBM_PEAK
                out["counter_traffic_bytes_per_step"] = {k: v * scale for k, v in per_stage.items()}
                tr = per_kernel.get(roof["kernel"])
                if tr:
                    roof["traffic"] = tr["hbm_bytes_per_launch"]
                    roof["traffic_detail"] = tr
                else:
                    roof["traffic_note"] = f"{roof['kernel']} not among the profiled kernels {sorted(per_kernel)}"
            else:
                roof["traffic_note"] = why
        if sample is not None:
            out["cpu_baseline"] = cpu_baseline(*sample, reg_ref=args.reg_ref)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
