"""CPU, world_size 2 over gloo: FOV sharding and the variable-length table gather (the only
collective of the path; core.py:336 pd.concat counterpart)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imageanalysis_pipeline_amd.distributed import gather_rows, pack_rows, reference_plane, shard_fovs, shard_frames


def test_shard_fovs_partitions_exactly():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_fovs(n, r, world)]
            assert got == list(range(n))
            sizes = [len(shard_fovs(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_pack_rows_layout():
    area = torch.tensor([[5, 0, 3], [0, 2, 0]])
    mean = torch.arange(12, dtype=torch.float64).reshape(2, 3, 2)
    cen = torch.arange(18, dtype=torch.float64).reshape(2, 3, 3) + 100
    rows = pack_rows((area, mean, cen), fov=4)
    # label 0 of frame 0 dropped; rows: (frame0,label2), (frame1,label1)
    assert rows.shape == (2, 4 + 2 + 3)
    assert rows[:, :4].tolist() == [[4, 0, 2, 3], [4, 1, 1, 2]]
    assert rows[0, 4:6].tolist() == mean[0, 2].tolist() and rows[1, 6:].tolist() == cen[1, 1].tolist()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_fov = 5
        mine = []
        for fov in shard_fovs(n_fov, rank, world):
            g = torch.Generator().manual_seed(fov)
            k = 3 + fov                                     # ragged: every FOV has another row count
            mine.append(torch.cat([torch.full((k, 1), float(fov), dtype=torch.float64),
                                   torch.rand((k, 8), generator=g, dtype=torch.float64)], 1))
        rows = torch.cat(mine, 0)
        allrows = gather_rows(rows)
        empty = gather_rows(torch.zeros((0, 9), dtype=torch.float64) if rank == 1 else rows)   # one rank empty
        q.put((rank, allrows.numpy(), empty.shape[0]))
    finally:
        dist.destroy_process_group()


def test_gather_rows_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = []
    for fov in range(5):
        g = torch.Generator().manual_seed(fov)
        k = 3 + fov
        exp.append(np.concatenate([np.full((k, 1), float(fov)), torch.rand((k, 8), generator=g, dtype=torch.float64).numpy()], 1))
    exp = np.concatenate(exp, 0)
    for rank, rows, n_empty in res:
        assert np.array_equal(rows, exp)                    # rank order == FOV order, identical on every rank
        assert n_empty == sum(3 + f for f in shard_fovs(5, 0, 2))


def _ref_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        t = 5
        planes = torch.from_numpy(rng.integers(0, 65536, size=(t, 6, 8)).astype(np.int32)).to(torch.uint16)
        fr = shard_frames(t, rank, world)
        mine = planes[fr.start:fr.stop]
        first = reference_plane(mine, fr.start, t, "first")
        mean = reference_plane(mine, fr.start, t, "mean")
        q.put((rank, first.view(torch.int16).numpy(), mean.numpy()))
    finally:
        dist.destroy_process_group()


def test_reference_plane_exchange_two_ranks():
    """time-block sharding: every rank ends up with frame 0's plane ('first') and with the scaled exact sum
    of all planes ('mean'), bit for bit what one process computes"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ref_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    rng = np.random.default_rng(3)
    planes = rng.integers(0, 65536, size=(5, 6, 8)).astype(np.int32)
    exp_mean = planes.sum(0).astype(np.float32) * np.float32(1.0 / 5)
    for _, first, mean in got:
        assert np.array_equal(first.view(np.uint16), planes[0].astype(np.uint16))
        assert np.array_equal(mean, exp_mean)
    one = reference_plane(torch.from_numpy(planes).to(torch.uint16), 0, 5, "mean")
    assert np.array_equal(one.numpy(), exp_mean)
    try:
        reference_plane(torch.zeros((1, 2, 2), dtype=torch.uint16), 0, 1, "previous")
        assert False
    except ValueError:
        pass


def test_provisional_table_is_bounded_by_bytes():
    """ops.provisional_labels: 8192 rows while T x rows x (4 + C) int64 fits 256 MB, never fewer than 256."""
    from imageanalysis_pipeline_amd import ops
    assert ops.provisional_labels(100, 2) == 8192
    assert ops.provisional_labels(500, 4) == 8192            # 262 MB would be over: 256 MB // (500 * 8 * 8) = 8388
    t, c = 5000, 3
    rows = ops.provisional_labels(t, c)
    assert 256 <= rows < 8192 and t * rows * (4 + c) * 8 <= ops.PROVISIONAL_BYTES
    assert ops.provisional_labels(10 ** 7, 4) == 256
