"""Minimal baseline-TIFF reader/writer for the planes the reference exchanges on disk:
single-page, uncompressed, 16-bit (or 8/32-bit) grayscale, as written by
``tifffile.imwrite(path, plane.astype(np.uint16))`` (utility.py:166, 264, 355, 616).

``imread`` / ``imwrite`` are the single-plane calls of the reference (tifffile when importable, a
pure-Python baseline reader otherwise).  ``read_planes`` / ``write_planes`` are the batched native
path (csrc/tiffio.hip): a thread pool moves whole plane sequences between files
and one host buffer.  ``read_planes_to_device`` / ``write_planes_from_device`` are what the shim's folder calls
use: two pinned staging buffers, the file I/O of block k + 1 overlapped with the PCIe copy of block k.
"""
from __future__ import annotations

import os
import struct

import numpy as np

try:  # pragma: no cover - depends on the environment
    import tifffile as _tf
except Exception:  # noqa: BLE001
    _tf = None

IO_THREADS = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))


def _path_array(paths):
    import ctypes as C
    arr = (C.c_char_p * len(paths))()
    arr[:] = [os.fsencode(p) for p in paths]
    return arr


def probe(path: str):
    """(Y, X, bits, sample_format, compression, samples) of the first directory of a classic TIFF."""
    import ctypes as C
    from . import _lib
    v = [C.c_int() for _ in range(6)]
    _lib.check(_lib.load().iap_tiff_probe(os.fsencode(path), *[C.byref(x) for x in v]), "iap_tiff_probe")
    return tuple(x.value for x in v)


def read_planes(paths, shape=None, pinned: bool = False, threads: int | None = None):
    """n uint16 planes of identical shape from n TIFF files -> (n, Y, X) array (a pinned torch tensor
    when ``pinned``).  Raises ValueError naming the file for anything but stripped, uncompressed,
    single-sample 16-bit unsigned classic TIFF (callers fall back to ``imread`` per file)."""
    from . import _lib
    paths = list(paths)
    if not paths:
        raise ValueError("read_planes: no files")
    if shape is None:
        y, x, bits, sfmt, comp, spp = probe(paths[0])
    else:
        y, x = shape
    lib = _lib.load()
    if pinned:
        import torch
        out = torch.empty((len(paths), y, x), dtype=torch.uint16, pin_memory=torch.cuda.is_available())
        ptr = out.data_ptr()
    else:
        out = np.empty((len(paths), y, x), dtype=np.uint16)
        ptr = out.ctypes.data
    _lib.check(lib.iap_tiff_read_u16(_path_array(paths), len(paths), ptr, y, x, threads or IO_THREADS),
               "iap_tiff_read_u16")
    return out


def write_planes(paths, planes, threads: int | None = None) -> None:
    """(n, Y, X) uint16 planes -> n little-endian single-strip TIFF files."""
    from . import _lib
    paths = list(paths)
    a = planes
    if not isinstance(a, np.ndarray):  # torch tensor on the host
        a = a.view(getattr(__import__("torch"), "int16")).numpy().view(np.uint16)
    a = np.ascontiguousarray(a, dtype=np.uint16)
    if a.ndim != 3 or a.shape[0] != len(paths):
        raise ValueError("write_planes: planes must be (n, Y, X) with one path per plane")
    _lib.check(_lib.load().iap_tiff_write_u16(_path_array(paths), len(paths), a.ctypes.data, a.shape[1], a.shape[2],
                                              threads or IO_THREADS), "iap_tiff_write_u16")


STAGE_BYTES = 256 << 20      # one pinned staging buffer (two are in use): a block of planes read / written at a time


def _blocks(n, y, x, block_planes):
    if not block_planes:
        block_planes = max(1, STAGE_BYTES // (2 * y * x))
    return [(i, min(n, i + block_planes)) for i in range(0, n, block_planes)]


def read_planes_to_device(paths, device, shape=None, threads: int | None = None, block_planes: int | None = None):
    """n plane files -> device tensor u16 (n, Y, X), staged through two pinned host buffers: a reader thread fills
    block k + 1 (the native reader runs outside the GIL) while block k goes to the GPU with a non-blocking copy on
    a side stream -- the loop of utility.py:487-524 (imread per file, np.stack, then one synchronous upload) with
    file reads and the PCIe copy overlapped.  Raises ValueError as read_planes does for files it cannot take."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    paths = list(paths)
    if not paths:
        raise ValueError("read_planes_to_device: no files")
    y, x = shape if shape is not None else probe(paths[0])[:2]
    n = len(paths)
    out = torch.empty((n, y, x), dtype=torch.uint16, device=device)
    blocks = _blocks(n, y, x, block_planes)
    nb = max(b - a for a, b in blocks)
    stage = [torch.empty((nb, y, x), dtype=torch.uint16, pin_memory=True) for _ in range(min(2, len(blocks)))]
    free = [None] * len(stage)                     # event after which a staging buffer may be overwritten
    side = torch.cuda.Stream(device=device)
    lib_threads = threads or IO_THREADS

    def fill(k):
        a, b = blocks[k]
        buf = stage[k % len(stage)]
        if free[k % len(stage)] is not None:
            free[k % len(stage)].synchronize()     # its previous upload has left the buffer
        from . import _lib
        _lib.check(_lib.load().iap_tiff_read_u16(_path_array(paths[a:b]), b - a, buf.data_ptr(), y, x, lib_threads),
                   "iap_tiff_read_u16")
        return buf

    with ThreadPoolExecutor(1) as pool:
        pending = pool.submit(fill, 0)
        for k, (a, b) in enumerate(blocks):
            buf = pending.result()
            if k + 1 < len(blocks):
                pending = pool.submit(fill, k + 1)
            with torch.cuda.stream(side):
                out[a:b].copy_(buf[: b - a], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            free[k % len(stage)] = ev
    torch.cuda.current_stream(device).wait_stream(side)
    for ev in free:                                 # the pinned buffers go away with this call
        if ev is not None:
            ev.synchronize()
    return out


def write_planes_from_device(paths, planes, threads: int | None = None, block_planes: int | None = None) -> None:
    """device tensor u16 (n, Y, X) -> n plane files, block by block through two pinned buffers: the download of
    block k + 1 runs on a side stream while a writer thread puts block k on disk (utility.py:264, 355 write
    file by file after the whole result is on the host)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from . import _lib
    paths = list(paths)
    if planes.dim() != 3 or planes.shape[0] != len(paths) or planes.dtype != torch.uint16 or not planes.is_cuda:
        raise ValueError("write_planes_from_device: planes must be a device uint16 tensor (n, Y, X), one path per plane")
    planes = planes.contiguous()
    n, y, x = planes.shape
    blocks = _blocks(n, y, x, block_planes)
    nb = max(b - a for a, b in blocks)
    stage = [torch.empty((nb, y, x), dtype=torch.uint16, pin_memory=True) for _ in range(min(2, len(blocks)))]
    side = torch.cuda.Stream(device=planes.device)
    side.wait_stream(torch.cuda.current_stream(planes.device))      # the planes are final on the caller's stream
    lib_threads = threads or IO_THREADS

    def download(k):
        a, b = blocks[k]
        buf = stage[k % len(stage)]
        with torch.cuda.stream(side):
            buf[: b - a].copy_(planes[a:b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def put(k, ev):
        a, b = blocks[k]
        ev.synchronize()
        buf = stage[k % len(stage)]
        _lib.check(_lib.load().iap_tiff_write_u16(_path_array(paths[a:b]), b - a, buf.data_ptr(), y, x, lib_threads),
                   "iap_tiff_write_u16")

    with ThreadPoolExecutor(1) as pool:
        writing = [None] * len(stage)
        ev = download(0)
        for k in range(len(blocks)):
            fut = pool.submit(put, k, ev)
            writing[k % len(stage)] = fut
            if k + 1 < len(blocks):
                slot = (k + 1) % len(stage)
                if writing[slot] is not None:
                    writing[slot].result()          # that buffer's block is on disk
                ev = download(k + 1)
        for fut in writing:
            if fut is not None:
                fut.result()


_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 16: ("Q", 8)}


def imread(path: str) -> np.ndarray:
    if _tf is not None:
        return _tf.imread(path)
    with open(path, "rb") as fh:
        buf = fh.read()
    bo = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if bo is None or struct.unpack(bo + "H", buf[2:4])[0] != 42:
        raise ValueError(f"{path}: not a classic TIFF file")
    (ifd,) = struct.unpack(bo + "I", buf[4:8])
    (n,) = struct.unpack(bo + "H", buf[ifd:ifd + 2])
    tags = {}
    for i in range(n):
        e = ifd + 2 + 12 * i
        tag, typ, cnt = struct.unpack(bo + "HHI", buf[e:e + 8])
        fmt, size = _TYPES.get(typ, ("B", 1))
        if typ == 5:
            continue
        total = size * cnt
        off = e + 8 if total <= 4 else struct.unpack(bo + "I", buf[e + 8:e + 12])[0]
        tags[tag] = struct.unpack(bo + fmt * cnt, buf[off:off + total])
    w, h = tags[256][0], tags[257][0]
    bits = tags.get(258, (1,))[0]
    if tags.get(259, (1,))[0] != 1:
        raise ValueError(f"{path}: compressed TIFF is not supported by the built-in reader (install tifffile)")
    if tags.get(277, (1,))[0] != 1:
        raise ValueError(f"{path}: only single-sample (grayscale) planes are supported")
    sfmt = tags.get(339, (1,))[0]
    dt = {(8, 1): "u1", (16, 1): "u2", (32, 1): "u4", (16, 2): "i2", (32, 3): "f4"}.get((bits, sfmt))
    if dt is None:
        raise ValueError(f"{path}: unsupported sample format bits={bits} format={sfmt}")
    offs, cnts = tags[273], tags[279]
    data = b"".join(buf[o:o + c] for o, c in zip(offs, cnts))
    return np.frombuffer(data, dtype=np.dtype(bo + dt), count=w * h).reshape(h, w).astype(np.dtype(dt))


def imwrite(path: str, a: np.ndarray) -> None:
    if _tf is not None:
        _tf.imwrite(path, a)
        return
    a = np.ascontiguousarray(a)
    if a.ndim != 2 or a.dtype not in (np.uint8, np.uint16, np.uint32, np.float32):
        raise ValueError("built-in TIFF writer handles 2-D uint8/uint16/uint32/float32 planes")
    h, w = a.shape
    data = a.astype(a.dtype.newbyteorder("<")).tobytes()
    sfmt = 3 if a.dtype == np.float32 else 1
    entries = [(256, 4, w), (257, 4, h), (258, 3, a.dtype.itemsize * 8), (259, 3, 1), (262, 3, 1),
               (273, 4, 8), (277, 3, 1), (278, 4, h), (279, 4, len(data)), (339, 3, sfmt)]
    ifd_off = 8 + len(data) + (len(data) & 1)
    out = bytearray(b"II" + struct.pack("<HI", 42, ifd_off) + data + (b"\0" if len(data) & 1 else b""))
    out += struct.pack("<H", len(entries))
    for tag, typ, val in entries:
        out += struct.pack("<HHI", tag, typ, 1) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
    out += struct.pack("<I", 0)
    with open(path, "wb") as fh:
        fh.write(bytes(out))
