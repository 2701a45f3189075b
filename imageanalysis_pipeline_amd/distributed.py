"""Multi-GPU driver for the hot path: fields of view are independent (core.py:99-112, 327-337 are
plain loops), so they shard across ranks (one process per GPU) with NO data-path collective; the
per-cell row tables are assembled with ONE variable-length all_gather over RCCL (backend "nccl" on
ROCm; "gloo" on CPU for tests) -- the counterpart of the reference's ``pd.concat`` of per-FOV
DataFrames (core.py:336).

Row layout (float64 matrix, exact for every field: ids/areas < 2**53):
    fov, frame, label, area, mean[0..C-1], centroid z, y, x
"""
from __future__ import annotations

import torch
import torch.distributed as dist


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
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
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


def process_plate(load_fov, n_fov: int, threshold: float, reg_ref: str = "mean", reg_channel: int = 0, group=None):
    """Runs bg_smo -> im_reg -> measure on this rank's shard of the plate and gathers the rows.
    ``load_fov(i)`` returns (stack u16 (T,C,Z,Y,X), labels u16 (T,Z,Y,X)) as device tensors."""
    from . import ops
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    parts = []
    for i in shard_fovs(n_fov, rank, world):
        stack, labels = load_fov(i)
        bg, _, _ = ops.bg_smo(stack, threshold, out=stack)
        reg, _ = ops.im_reg(bg, reg_channel, reg_ref)
        acc = ops.measure_accumulate(reg, labels)
        parts.append(pack_rows(ops.measure_finalize(acc), i))
    if parts:
        rows = torch.cat(parts, 0)
    else:
        rows = None
    # ranks without FOVs still take part in the collective with an empty matrix of the right width
    width = torch.tensor([rows.shape[1] if rows is not None else 0], dtype=torch.int64,
                         device=parts[0].device if parts else ("cuda" if torch.cuda.is_available() else "cpu"))
    if dist.is_initialized() and world > 1:
        if dist.get_backend(group) == "gloo":
            width = width.cpu()
        dist.all_reduce(width, op=dist.ReduceOp.MAX, group=group)
    if rows is None:
        rows = torch.zeros((0, int(width.item())), dtype=torch.float64, device=width.device)
    return gather_rows(rows, group)
