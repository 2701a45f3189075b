"""Multi-GPU driver for the hot path: fields of view are independent (core.py:99-112, 327-337 are
plain loops), so they shard across ranks (one process per GPU) with NO data-path collective; the
per-cell row tables are assembled with ONE variable-length all_gather over RCCL (backend "nccl" on
ROCm; "gloo" on CPU for tests) -- the counterpart of the reference's ``pd.concat`` of per-FOV
DataFrames (core.py:336).

Row layout (float64 matrix, exact for every field: ids/areas < 2**53):
    fov, frame, label, area, mean[0..C-1], centroid z, y, x

A single large FOV can instead be split into time blocks (``process_fov_timeblocks``): background
subtraction, warping and measuring are per frame, and with ``reg_ref`` 'first' or 'mean' every frame is
registered against one plane, which is the path's only exchange: a broadcast of frame 0's plane (8 MB at
2048^2) or an all-reduce of the plane sums (16 MB).  'previous' is serial in time and does not shard.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

# Rehearsal switch: with it set, a process group of ONE rank still runs every collective of this module instead of
# taking the single-process shortcuts -- the only way to execute the RCCL branch (device tensors through
# all_gather / all_reduce / broadcast) on a box with one GPU (tests/test_distributed_gpu.py).
FORCE_COLLECTIVES = False


def _exchanging(group=None) -> bool:
    """True when this module's collectives have somebody to talk to (or are being rehearsed)."""
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


def shard_fovs(n_fov: int, rank: int, world: int) -> range:
    """Contiguous block distribution; the first (n_fov % world) ranks hold one extra FOV."""
    q, r = divmod(n_fov, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))


def pack_rows(tables, fov: int) -> torch.Tensor:
    """(area (T,L), mean (T,L,C), centroid (T,L,3), ...) device tensors -> rows of the labels present.
    Label 0 rows (background count) are dropped; the reference's 'drop the smallest label of a frame
    without background' quirk (analysis.py:80) is applied by the table builder, not here."""
    area, mean, cen = tables[0], tables[1], tables[2]
    idx = torch.nonzero(area > 0)
    idx = idx[idx[:, 1] > 0]
    f, l = idx[:, 0], idx[:, 1]
    n = idx.shape[0]
    head = torch.stack([torch.full((n,), float(fov), dtype=torch.float64, device=area.device),
                        f.double(), l.double(), area[f, l].double()], 1)
    return torch.cat([head, mean[f, l], cen[f, l]], 1)


def gather_rows(rows: torch.Tensor, group=None) -> torch.Tensor:
    """Variable-length all_gather: every rank receives the concatenation (in rank order) of all
    ranks' row matrices.  Two collectives: row counts (int64), then the padded payload."""
    if not _exchanging(group):
        return rows
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":   # gloo gathers host tensors (tests, rehearsals); RCCL takes device ones
        rows = rows.cpu()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    pad = torch.zeros((nmax, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def process_plate(load_fov, n_fov: int, threshold: float, reg_ref: str = "mean", reg_channel: int = 0, group=None,
                  prefetch: bool = False):
    """Runs bg_smo -> im_reg -> measure on this rank's shard of the plate and gathers the rows.
    ``load_fov(i)`` returns (stack u16 (T,C,Z,Y,X), labels u16 (T,Z,Y,X)) as device tensors.

    ``prefetch``: FOV k+1 is loaded on a side stream while FOV k computes (a plate that does not fit the GPU is
    streamed through two buffers: SURVEY 8e).  The loader is then called as ``load_fov(i, slot)`` with slot = 0 / 1
    alternating and may reuse one pair of buffers per slot: the side stream waits until the FOV that used the slot
    last has been measured before the loader is called again for it, and the compute stream waits for the load."""
    from . import ops
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    parts = []
    mine = list(shard_fovs(n_fov, rank, world))
    if prefetch and mine and torch.cuda.is_available():
        main = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        loaded = [torch.cuda.Event(), torch.cuda.Event()]     # slot's FOV is in its buffers
        free = [torch.cuda.Event(), torch.cuda.Event()]       # slot's FOV has been consumed

        def start_load(k):
            slot = k % 2
            with torch.cuda.stream(side):
                if k >= 2:
                    side.wait_event(free[slot])
                got = load_fov(mine[k], slot)
                loaded[slot].record(side)
            return got

        nxt = start_load(0)
        for k, i in enumerate(mine):
            stack, labels = nxt
            if k + 1 < len(mine):
                nxt = start_load(k + 1)
            main.wait_event(loaded[k % 2])
            tables, _, _ = ops.hot_path(stack, labels, threshold, reg_channel, reg_ref)
            free[k % 2].record(main)
            parts.append(pack_rows(tables, i))
    else:
        for i in mine:
            stack, labels = load_fov(i)
            tables, _, _ = ops.hot_path(stack, labels, threshold, reg_channel, reg_ref)   # bg levels, im_reg, measure
            parts.append(pack_rows(tables, i))
    if parts:
        rows = torch.cat(parts, 0)
    else:
        rows = None
    # ranks without FOVs still take part in the collective with an empty matrix of the right width
    width = torch.tensor([rows.shape[1] if rows is not None else 0], dtype=torch.int64,
                         device=parts[0].device if parts else ("cuda" if torch.cuda.is_available() else "cpu"))
    if _exchanging(group):
        if dist.get_backend(group) == "gloo":
            width = width.cpu()
        dist.all_reduce(width, op=dist.ReduceOp.MAX, group=group)
    if rows is None:
        rows = torch.zeros((0, int(width.item())), dtype=torch.float64, device=width.device)
    return gather_rows(rows, group)


def shard_frames(n_frames: int, rank: int, world: int) -> range:
    """Contiguous time blocks, as shard_fovs."""
    return shard_fovs(n_frames, rank, world)


def reference_plane(planes: torch.Tensor, t0: int, n_frames: int, reg_ref: str, group=None) -> torch.Tensor:
    """The plane every frame is registered against (utility.py:332-366), from this rank's (MIP) planes
    u16 (tb, Y, X) of the registration channel, frames t0 .. t0+tb-1 of n_frames.
    'first': frame 0's plane, broadcast by the rank that holds it -> u16 (Y, X).
    'mean':  sum over all frames (exact int64 sums, all-reduced) times float32(1/n_frames) -> f32 (Y, X),
             the very values iap_im_reg_u16 feeds its reference spectrum (sum as f32, scaled on load)."""
    if reg_ref not in ("first", "mean"):
        raise ValueError(f"time blocks cannot shard reg_ref={reg_ref!r} ('previous' is serial in time)")
    multi = _exchanging(group)
    host = multi and dist.get_backend(group) == "gloo"      # gloo moves host tensors (tests, rehearsals)
    dev = planes.device
    if reg_ref == "first":
        if not multi:
            return planes[0].clone()
        has0 = t0 == 0 and planes.shape[0] > 0
        buf = (planes[0].clone() if has0 else torch.zeros(planes.shape[1:], dtype=torch.uint16, device=dev)).view(torch.uint8)   # as bytes: a type every backend moves
        if host:
            buf = buf.cpu()
        # the rank holding frame 0: rank 0 under the block distribution whenever it holds a frame at all
        src = torch.tensor([dist.get_rank(group) if has0 else -1], dtype=torch.int64, device=buf.device)
        dist.all_reduce(src, op=dist.ReduceOp.MAX, group=group)
        dist.broadcast(buf, src=dist.get_global_rank(group, int(src.item())) if group is not None else int(src.item()),
                       group=group)
        return buf.to(dev).view(torch.uint16)
    # (int64: 32768 frames of saturated pixels would pass int32; the library's own 'mean' path sums in u32)
    total = planes.to(torch.int32).sum(dim=0, dtype=torch.int64) if planes.shape[0] else \
        torch.zeros(planes.shape[1:], dtype=torch.int64, device=dev)
    if multi:
        if host:
            total = total.cpu()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        total = total.to(dev)
    return total.to(torch.float32) * torch.tensor(1.0 / n_frames, dtype=torch.float32, device=dev)


def process_fov_timeblocks(load_block, n_frames: int, threshold: float, reg_ref: str = "first", reg_channel: int = 0,
                           fov: int = 0, group=None):
    """One FOV split into contiguous time blocks, one per rank: bg_smo -> reference exchange -> estimate ->
    apply -> measure on the block, rows gathered with global frame numbers.  ``load_block(t0, t1)`` returns
    (stack u16 (t1-t0, C, Z, Y, X), labels u16 (t1-t0, Z, Y, X)) as device tensors.  Every rank returns
    (rows of all frames, this block's shifts f64 (t1-t0, 2)); the arithmetic per frame is that of
    ops.im_reg, so the rows equal a single-process run."""
    from . import ops
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    frames = shard_frames(n_frames, rank, world)
    t0, t1 = frames.start, frames.stop
    stack, labels = load_block(t0, t1)                               # a rank past the last frame gets (0, C, Z, Y, X)
    tb, c, z, y, x = stack.shape
    bg = stack                                                       # raw: the levels are subtracted on load
    if tb:
        levels, _ = ops.bg_smo_levels(stack, threshold)
        planes = ops.lowered_channel_mip(stack, reg_channel, levels)  # (tb, Y, X): the planes the estimate runs on
    else:                                                            # still takes part in the exchange and the gather
        planes = torch.zeros((0, y, x), dtype=torch.uint16, device=stack.device)
    ref = reference_plane(planes, t0, n_frames, reg_ref, group)
    shifts = ops.estimate_shifts(ref, planes) if tb else torch.zeros((0, 2), dtype=torch.float64, device=bg.device)
    if reg_ref == "first" and t0 == 0 and tb:
        shifts[0] = 0.0                                              # frame 0 is the reference itself (as ops.im_reg)
    reg = ops.apply_shifts(stack, shifts, sub2=levels) if tb else bg
    if tb:
        rows = pack_rows(ops.measure_finalize(ops.measure_accumulate(reg, labels)), fov)
        rows[:, 1] += float(t0)
    else:
        rows = torch.zeros((0, 4 + c + 3), dtype=torch.float64, device=bg.device)
    return gather_rows(rows, group), shifts
