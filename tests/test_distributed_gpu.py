"""GPU: the multi-GPU drivers of distributed.py (FOV sharding, core.py:99-112 / 327-337 loops; time blocks) run
on the card.  Single rank against the sequential ops path and the oracle on a C4-shaped (3-channel) plate, then two
ranks (gloo, both on this box's GPU) against the single-process result: the gathered table must be identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_path as O
from imageanalysis_pipeline_amd import distributed as D
from imageanalysis_pipeline_amd import ops, synth

pytestmark = pytest.mark.gpu

T, C, Y, X, NFOV, CELLS = 4, 3, 128, 256, 3, 8      # BASELINE configs[3] in small: a plate of 3-channel FOVs


def _fov(i, dev):
    st, _ = synth.make_stack(T, C, 1, Y, X, n_cells=CELLS, integer_drift=True, seed=20 + i)
    lab = np.stack([synth.make_labels(Y, X, CELLS, seed=20 + i)[None]] * T)
    return st, lab, ops.to_device_u16(st, dev), ops.to_device_u16(lab, dev)


def _sequential_rows(dev, thr, reg_ref, fovs):
    parts = []
    for i in fovs:
        _, _, st, lab = _fov(i, dev)
        bg, _, _ = ops.bg_smo(st, thr)
        reg, _ = ops.im_reg(bg, 0, reg_ref)
        parts.append(D.pack_rows(ops.measure_finalize(ops.measure_accumulate(reg, lab)), i))
    return torch.cat(parts, 0)


@pytest.mark.parametrize("reg_ref", ["mean", "first"])
def test_process_plate_single_rank_c4_shaped(dev, reg_ref):
    thr = O.smo_threshold((Y, X))
    rows = D.process_plate(lambda i: _fov(i, dev)[2:], NFOV, thr, reg_ref=reg_ref)
    ref = _sequential_rows(dev, thr, reg_ref, range(NFOV))
    assert rows.shape == ref.shape == (rows.shape[0], 4 + C + 3)
    assert torch.equal(rows, ref)                        # subtract-on-load == two passes, bit for bit
    # and against the oracle, FOV by FOV: ids and areas exact, centroids exact, means within 1e-4 relative
    # (each side warps with its own sub-pixel shifts, tests/test_path_gpu.py)
    rows = rows.cpu().numpy()
    for i in range(NFOV):
        st, lab, _, _ = _fov(i, dev)
        _, _, etab = O.run_path_opts(st, lab, thr, reg_ref=reg_ref, upsample=True)
        mine = rows[rows[:, 0] == i]
        pos = 0
        for f, e in enumerate(etab):
            part = mine[mine[:, 1] == f]
            assert np.array_equal(part[:, 2].astype(np.int64), e["ids"])
            assert np.array_equal(part[:, 3].astype(np.int64), e["area"])
            assert np.array_equal(np.rint(part[:, 4 + C + 1:]).astype(np.int64), e["centroid_rne"])
            assert (np.abs(part[:, 4:4 + C] - e["mean"]) <= 1e-4 * np.abs(e["mean"])).all()
            pos += len(part)
        assert pos == len(mine)


def test_process_plate_streamed_through_two_slots(dev):
    """prefetch=True: FOV k+1 is copied into the other of two reusable device slots on a side stream while FOV k
    computes (a plate that does not fit the GPU); the rows equal the plain loop's, bit for bit, although every
    slot is overwritten several times"""
    thr = O.smo_threshold((Y, X))
    nfov = 7
    host = [tuple(t.cpu().pin_memory() for t in _fov(i, dev)[2:]) for i in range(nfov)]
    slots = [(torch.empty((T, C, 1, Y, X), dtype=torch.uint16, device=dev), torch.empty((T, 1, Y, X), dtype=torch.uint16, device=dev))
             for _ in range(2)]
    seen = []

    def upload(i, slot):
        seen.append((i, slot))
        slots[slot][0].copy_(host[i][0], non_blocking=True)
        slots[slot][1].copy_(host[i][1], non_blocking=True)
        return slots[slot]

    rows = D.process_plate(upload, nfov, thr, reg_ref="mean", prefetch=True)
    assert seen == [(i, i % 2) for i in range(nfov)]
    ref = D.process_plate(lambda i: _fov(i, dev)[2:], nfov, thr, reg_ref="mean")
    assert torch.equal(rows, ref)


@pytest.mark.parametrize("reg_ref", ["mean", "first"])
def test_process_fov_timeblocks_single_rank(dev, reg_ref):
    thr = O.smo_threshold((Y, X))
    _, _, st, lab = _fov(1, dev)
    rows, shifts = D.process_fov_timeblocks(lambda t0, t1: (st[t0:t1].contiguous(), lab[t0:t1].contiguous()), T, thr,
                                            reg_ref=reg_ref, fov=1)
    bg, _, _ = ops.bg_smo(st, thr)
    reg, sh = ops.im_reg(bg, 0, reg_ref)
    ref = D.pack_rows(ops.measure_finalize(ops.measure_accumulate(reg, lab)), 1)
    assert torch.equal(shifts, sh) and torch.equal(rows, ref)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        _worker_body(rank, world, port, q)
    except BaseException as e:  # noqa: BLE001  (the parent must learn why a rank died instead of timing out)
        q.put((rank, f"{type(e).__name__}: {e}"))
        raise


def _worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # gloo: both ranks sit on the one GPU here
    try:
        dev = torch.device("cuda", rank % torch.cuda.device_count())
        torch.cuda.set_device(dev)
        thr = O.smo_threshold((Y, X))
        plate = D.process_plate(lambda i: _fov(i, dev)[2:], 5, thr, reg_ref="mean")
        _, _, st, lab = _fov(0, dev)
        tb_rows, tb_sh = D.process_fov_timeblocks(lambda t0, t1: (st[t0:t1].contiguous(), lab[t0:t1].contiguous()), T,
                                                  thr, reg_ref="mean", fov=0)
        q.put((rank, plate.cpu().numpy(), tb_rows.cpu().numpy(), tb_sh.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_process(dev):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=170) for _ in range(world)]
        for r in res:
            assert len(r) == 4, f"rank {r[0]} failed: {r[1]}"
        res.sort(key=lambda r: r[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:          # no rank outlives the test holding the GPU
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
    thr = O.smo_threshold((Y, X))
    ref = _sequential_rows(dev, thr, "mean", range(5)).cpu().numpy()
    _, _, st, lab = _fov(0, dev)
    bg, _, _ = ops.bg_smo(st, thr)
    reg, sh = ops.im_reg(bg, 0, "mean")
    tb_ref = D.pack_rows(ops.measure_finalize(ops.measure_accumulate(reg, lab)), 0).cpu().numpy()
    sh = sh.cpu().numpy()
    for rank, plate, tb_rows, tb_sh in res:
        assert np.array_equal(plate, ref)                    # rank order == FOV order, identical on every rank
        assert np.array_equal(tb_rows, tb_ref)               # time blocks: the rows of a single-process run
        fr = D.shard_frames(T, rank, world)
        assert np.array_equal(tb_sh, sh[fr.start:fr.stop])


def _rccl_worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        try:
            D.FORCE_COLLECTIVES = True      # one rank, but every collective runs: device tensors through RCCL
            thr = O.smo_threshold((Y, X))
            plate = D.process_plate(lambda i: _fov(i, dev)[2:], 3, thr, reg_ref="mean")
            _, _, st, lab = _fov(0, dev)
            out = [plate.cpu().numpy()]
            for reg_ref in ("mean", "first"):
                rows, sh = D.process_fov_timeblocks(lambda t0, t1: (st[t0:t1].contiguous(), lab[t0:t1].contiguous()),
                                                    T, thr, reg_ref=reg_ref, fov=0)
                out += [rows.cpu().numpy(), sh.cpu().numpy()]
            tt = torch.tensor([1.5], dtype=torch.float64, device=dev)     # bench.py's max-over-ranks of the step time
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.barrier()
            out.append(float(tt.item()))
            q.put(("ok", out))
        finally:
            dist.destroy_process_group()
    except BaseException as e:  # noqa: BLE001
        q.put(("failed", f"{type(e).__name__}: {e}"))
        raise


def test_rccl_branch_with_one_rank(dev):
    """The RCCL ("nccl") branch of the gathers and of the reference-plane exchange, executed for real: a one-rank
    process group with the single-process shortcuts switched off (distributed.FORCE_COLLECTIVES).  Counts, padded
    payload, width reduction, the int64 plane-sum all_reduce, the byte broadcast of frame 0 and bench.py's float64
    MAX all go through RCCL on device tensors; the results equal the plain single-process run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    try:
        status, out = q.get(timeout=170)
        assert status == "ok", out
        p.join(timeout=60)
        assert p.exitcode == 0
    finally:
        if p.is_alive():
            p.terminate()
            p.join(timeout=10)
    thr = O.smo_threshold((Y, X))
    assert np.array_equal(out[0], _sequential_rows(dev, thr, "mean", range(3)).cpu().numpy())
    _, _, st, lab = _fov(0, dev)
    bg, _, _ = ops.bg_smo(st, thr)
    for k, reg_ref in enumerate(("mean", "first")):
        reg, sh = ops.im_reg(bg, 0, reg_ref)
        ref = D.pack_rows(ops.measure_finalize(ops.measure_accumulate(reg, lab)), 0).cpu().numpy()
        assert np.array_equal(out[1 + 2 * k], ref) and np.array_equal(out[2 + 2 * k], sh.cpu().numpy())
    assert out[5] == 1.5
