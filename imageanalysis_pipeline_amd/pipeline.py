"""Host-side mirror of the reference's interface for the hot path: same names, arguments, return
values, side effects and error behaviour as ``pipeline/utility.py`` and ``pipeline/analysis.py``,
with the per-file numpy/third-party loops replaced by ONE resident-stack call into libiap_hip.so.

    Utility.smo_bg_sub / man_bg_sub / im_reg / blur_img   <- utility.py:242, 175, 274, 433
    Utility.load_stack / load_mask / save_mask            <- utility.py:487, 558, 598 (layout contract)
    Analysis.extract_channelData / extract_centroids      <- analysis.py:26, 114

Drop-in use: ``Experiments`` / ``Exp_Indiv`` inherit these static methods instead of
the reference's, e.g. ``from imageanalysis_pipeline_amd.pipeline import Utility`` in core.py.
File I/O, the metadata pickle and the table post-processing (``transfo_df``) stay host side.
Differences are limited to what the reference cannot do headless: ``man_bg_sub`` takes the ROI as
an argument instead of opening an OpenCV window, and ``smo_bg_sub`` seeds the SMO null image.
"""
from __future__ import annotations

import pickle
from os import listdir, mkdir, scandir, sep
from os.path import exists, isdir, join

import numpy as np
import pandas as pd
import torch

from . import ops
from .tiffio import imread, imwrite, read_planes, read_planes_to_device, write_planes, write_planes_from_device


def _device():
    if not torch.cuda.is_available():
        from ._lib import IapError
        raise IapError("imageanalysis_pipeline_amd needs an MI355X: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _read_u16_planes(paths):
    """n plane files -> (n, Y, X) uint16: the batched native reader (csrc/tiffio.hip), or the generic
    per-file reader when a file is not a stripped, uncompressed 16-bit unsigned classic TIFF."""
    try:
        return read_planes(paths)
    except ValueError:
        return np.stack([imread(p) for p in paths]).astype(np.uint16)


def _write_u16_planes(paths, planes):
    write_planes(paths, np.ascontiguousarray(planes, dtype=np.uint16))


class _ResidentPlanes:
    """Planes the shim has just read or written stay on the GPU, so that consecutive calls on the same folders
    (smo_bg_sub -> im_reg -> extract_channelData, core.py:99-112, 327-337) cost one folder read and two folder
    writes instead of three reads and two writes.  An entry belongs to one folder and remembers, per file, the
    size and modification time the file had when the shim last touched it: a request is served from the GPU only if
    every requested file still carries them (any external write, deletion or new file of that name drops the
    entry), and only whole planes the shim itself produced or read are kept.  Bounded by ``IAP_SHIM_RESIDENT_GB``
    (default 48; 0 switches it off): the oldest folders go first."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._entries = {}          # folder -> (index {path: row}, sig {path: (size, mtime_ns)}, planes (n, Y, X), device)
        self.hits = self.misses = 0

    @staticmethod
    def _limit():
        import os
        try:
            return float(os.environ.get("IAP_SHIM_RESIDENT_GB", "48")) * 1e9
        except ValueError:
            return 48e9

    @staticmethod
    def _sig(path):
        import os
        st = os.stat(path)
        return (st.st_size, st.st_mtime_ns)

    def get(self, paths, dev):
        if not paths or self._limit() <= 0:
            return None
        from os.path import dirname
        folder = dirname(paths[0])
        with self._lock:
            e = self._entries.get(folder)
            if e is None or e[3] != dev:
                self.misses += 1
                return None
            index, sig, planes, _ = e
            try:
                rows = [index[p] for p in paths]
                fresh = all(self._sig(p) == sig[p] for p in paths)
            except (KeyError, OSError):
                fresh = False
            if not fresh:
                del self._entries[folder]       # somebody else wrote there: forget the folder
                self.misses += 1
                return None
            self.hits += 1
            self._entries[folder] = self._entries.pop(folder)    # most recently used
            if rows == list(range(planes.shape[0])):
                return planes
            return planes.index_select(0, torch.tensor(rows, dtype=torch.int64, device=planes.device))

    def put(self, paths, planes):
        """planes (n, Y, X) device u16 = the current contents of the files ``paths`` (just read or just written)"""
        limit = self._limit()
        if not paths or limit <= 0:
            return
        from os.path import dirname
        folder = dirname(paths[0])
        nbytes = planes.numel() * 2
        with self._lock:
            self._entries.pop(folder, None)
            if nbytes > limit or any(dirname(p) != folder for p in paths):
                return
            try:
                sig = {p: self._sig(p) for p in paths}
            except OSError:
                return
            self._entries[folder] = ({p: i for i, p in enumerate(paths)}, sig, planes, planes.device)
            while sum(e[2].numel() * 2 for e in self._entries.values()) > limit and len(self._entries) > 1:
                self._entries.pop(next(iter(self._entries)))

    def drop(self, folder):
        """before planes served from here are modified in place"""
        with self._lock:
            self._entries.pop(folder, None)

    def clear(self):
        with self._lock:
            self._entries.clear()


RESIDENT = _ResidentPlanes()


def _planes_to_device(paths, dev):
    """n plane files -> device u16 (n, Y, X): served from the GPU when the shim itself read or wrote exactly these
    files last and nobody touched them since (_ResidentPlanes), else file reads into pinned blocks overlapped with
    their upload (tiffio.read_planes_to_device); files the batched reader does not take go through the generic one."""
    got = RESIDENT.get(paths, dev)
    if got is not None:
        return got
    try:
        planes = read_planes_to_device(paths, dev)
    except ValueError:
        planes = ops.to_device_u16(np.stack([imread(p) for p in paths]).astype(np.uint16), dev)
    RESIDENT.put(paths, planes)
    return planes


def _planes_from_device(paths, planes):
    """device u16 planes (any leading shape, one path per plane) -> files, downloads overlapped with the writes;
    the planes stay resident as the folder's contents"""
    flat = planes.reshape(-1, *planes.shape[-2:])
    write_planes_from_device(paths, flat)
    RESIDENT.put(paths, flat)


class Utility:
    # ------------------------------------------------------------------ settings (utility.py:688-696)
    @staticmethod
    def open_exp_prop(exp_path):
        with open(join(sep, exp_path + sep, "exp_properties.pickle"), "rb") as pickfile:
            return pickle.load(pickfile)

    @staticmethod
    def save_exp_prop(exp_path, exp_prop):
        with open(join(sep, exp_path + sep, "exp_properties.pickle"), "wb") as file:
            pickle.dump(exp_prop, file, protocol=pickle.HIGHEST_PROTOCOL)

    # ------------------------------------------------------------------ layout helpers
    @staticmethod
    def _plane_name(chan, f, z):
        return chan + "_f%04d" % (f + 1) + "_z%04d" % (z + 1) + ".tif"      # utility.py:158-166

    @staticmethod
    def _folder_index(fold_path):
        """{(prefix, frame, z): filename} built with ONE listdir (the reference re-lists and re-sorts
        the folder for every (channel, frame): utility.py:513-517).  Two name forms exist in the reference:
        ``<prefix>_fNNNN_zNNNN.tif`` (create_imseq / save_mask, utility.py:158-166, 612-618) and
        ``<prefix>_fNNNN.tif`` without a z token (the reference masks of ref_mask, utility.py:859, 866, which
        its own load_mask finds through the digit count of the last token, utility.py:588-594): z = 0 there."""
        idx = {}
        for im in sorted(listdir(fold_path)):
            if not im.endswith(".tif"):
                continue
            parts = im[:-4].split("_")
            try:
                if parts[-1][:1] == "z" and len(parts) >= 3 and parts[-2][:1] == "f":
                    z, f, npre = int(parts[-1][1:]), int(parts[-2][1:]), len(parts) - 2
                elif parts[-1][:1] == "f" and len(parts) >= 2:
                    z, f, npre = 1, int(parts[-1][1:]), len(parts) - 1
                else:
                    continue
            except (ValueError, IndexError):
                continue
            idx[("_".join(parts[:npre]), f - 1, z - 1)] = im
        return idx

    @staticmethod
    def load_stack(imgFold_path, channel_list=None, input_range=None):
        """utility.py:487-524: (C,T,Z,Y,X) -> squeeze -> channel-last when more than one channel."""
        exp_path = sep.join(imgFold_path.split(sep)[:-1])
        exp_para = Utility.open_exp_prop(exp_path=exp_path)["metadata"]
        if channel_list:
            chan_lst = [channel_list] if type(channel_list) == str else channel_list
        else:
            chan_lst = exp_para["channel_list"]
        frame_range = input_range if input_range else range(exp_para["t"])
        idx = Utility._folder_index(imgFold_path)
        names = [[[v for (p, f, z), v in sorted(idx.items(), key=lambda kv: kv[0][2])
                   if p.startswith(chan) and f == frame] for frame in frame_range] for chan in chan_lst]
        flat = [join(sep, imgFold_path + sep, n) for chan in names for fr in chan for n in fr]
        nz = {len(fr) for chan in names for fr in chan}
        exp_list = None
        if flat and len(nz) == 1:
            # batched native read (csrc/tiffio.hip): one pass of a thread pool into one buffer
            try:
                planes = read_planes(flat)
                exp_list = planes.reshape(len(chan_lst), len(frame_range), nz.pop(), *planes.shape[1:])
            except ValueError:
                exp_list = None  # compressed / tiled / other sample type: generic reader below
        if exp_list is None:
            exp_list = [[[imread(join(sep, imgFold_path + sep, n)) for n in fr] for fr in chan] for chan in names]
        if len(chan_lst) == 1:
            return np.squeeze(np.stack(exp_list))
        return np.moveaxis(np.squeeze(np.stack(exp_list)), [0], [-1])

    @staticmethod
    def load_mask(maskFold_path, channel_seg, mask_shape=None, input_range=None, do_log=True):
        """utility.py:558-595."""
        if not any(channel_seg in file for file in listdir(maskFold_path)):
            raise AttributeError(f"No masks was found for the selected segmented channel: {channel_seg}")
        exp_path = sep.join(maskFold_path.split(sep)[:-1])
        exp_para = Utility.open_exp_prop(exp_path=exp_path)["metadata"]
        if input_range:
            if not isinstance(input_range, list):
                raise TypeError("input range must be a list")
            frame_range = input_range
        else:
            frame_range = list(range(exp_para["t"]))
        if mask_shape and type(mask_shape) != str:
            raise TypeError("mask_shape must be a string")
        if do_log:
            print(f"---> Loading frames {frame_range} of masks from '{maskFold_path}'")
        prefix = "mask_%s%s" % (channel_seg, mask_shape if mask_shape else "")
        idx = Utility._folder_index(maskFold_path)
        exp_list = []
        for f in frame_range:
            for (p, ff, z), name in sorted(idx.items(), key=lambda kv: (kv[0][1], kv[0][2])):
                if ff == f and p == prefix:
                    exp_list.append(imread(join(sep, maskFold_path + sep, name)))
        return np.squeeze(np.stack(exp_list))

    @staticmethod
    def save_mask(mask_stack, save_dir, batch, z_size, frames, chan_seg, motif=None):
        """utility.py:598-618."""
        if isinstance(mask_stack, np.ndarray):
            mask_stack = [mask_stack] if (mask_stack.ndim == 2 or frames == 1) else [m for m in mask_stack]
        motif = "_" + motif if motif else ""
        for i, f in enumerate(batch):
            for z in range(z_size):
                name = f"mask_{chan_seg}{motif}" + "_f%04d" % (f + 1) + "_z%04d" % (z + 1) + ".tif"
                plane = mask_stack[i] if z_size == 1 else mask_stack[i][z, ...]
                imwrite(join(sep, save_dir + sep, name), plane.astype(np.uint16))

    @staticmethod
    def _read_folder_stack(fold_path, exp_para, prefix_of=lambda c: c, channels=None, dev=None):
        """Whole folder -> u16 (T, C, Z, Y, X) (planar, the on-disk layout): a host array, or with ``dev`` a device
        tensor filled through pinned staging blocks while the next block is still being read."""
        chans = channels if channels is not None else exp_para["channel_list"]
        t, z = exp_para["t"], exp_para["z"]
        idx = Utility._folder_index(fold_path)
        paths = []
        for f in range(t):
            for chan in chans:
                for k in range(z):
                    name = idx.get((prefix_of(chan), f, k))
                    if name is None:
                        raise FileNotFoundError(f"{fold_path}: missing plane {prefix_of(chan)} f={f + 1} z={k + 1}")
                    paths.append(join(sep, fold_path + sep, name))
        if dev is not None:
            planes = _planes_to_device(paths, dev)
        else:
            planes = _read_u16_planes(paths)
        return planes.reshape(t, len(chans), z, *planes.shape[1:])

    @staticmethod
    def _write_folder_stack(fold_path, stack, channels):
        """(T, C, Z, Y, X) u16, host array or device tensor -> one file per plane"""
        t, c, z = stack.shape[:3]
        paths = [join(sep, fold_path + sep, Utility._plane_name(chan, f, k))
                 for f in range(t) for chan in channels for k in range(z)]
        if isinstance(stack, torch.Tensor) and stack.is_cuda:
            _planes_from_device(paths, stack)
        else:
            _write_u16_planes(paths, np.asarray(stack).reshape(t * c * z, *stack.shape[3:]))

    # ------------------------------------------------------------------ background (utility.py:175-270)
    @staticmethod
    def smo_bg_sub(imgFold_path, sigma=0.0, size=7, seed=0):
        print(f"--> Applying 'Auto' background substraction on {imgFold_path} with: sigma={sigma} and size={size}")
        exp_path = sep.join(imgFold_path.split(sep)[:-1])
        exp_prop = Utility.open_exp_prop(exp_path=exp_path)
        exp_para = exp_prop["metadata"]
        dev = _device()
        # every .tif of the folder, whatever its channel (utility.py:255-256), in one resident batch
        names = [f for f in sorted(listdir(imgFold_path)) if f.endswith(".tif")]
        if names:
            paths = [join(sep, imgFold_path + sep, f) for f in names]
            d = _planes_to_device(paths, dev)
            RESIDENT.drop(imgFold_path)     # d is about to change under the folder's old signature
            thr = ops.smo_threshold(tuple(d.shape[-2:]), sigma=sigma, size=size, seed=seed, device=dev)
            out, _, _ = ops.bg_smo(d, thr, sigma=sigma, size=size, out=d)     # in place, as the reference
            _planes_from_device(paths, out)
        exp_prop["fct_inputs"]["smo_bg_sub"] = {"imgFold_path": imgFold_path, "smo_sigma": sigma, "smo_size": size}
        exp_prop["img_preProcess"]["bg_sub"] = "Auto"
        Utility.save_exp_prop(exp_path=exp_path, exp_prop=exp_prop)
        return exp_prop

    @staticmethod
    def man_bg_sub(imgFold_path, roi=None):
        """utility.py:175-239 without the OpenCV window: ``roi`` = ((row0, col0), (row1, col1)) in image
        pixels, inclusive (what draw.rectangle receives at utility.py:210); ``None`` = 'skip' pressed."""
        print(f"--> Applying 'Manual' background substraction on {imgFold_path}")
        exp_path = sep.join(imgFold_path.split(sep)[:-1])
        exp_prop = Utility.open_exp_prop(exp_path=exp_path)
        exp_para = exp_prop["metadata"]
        if roi:
            (r0, c0), (r1, c1) = roi
            r0, r1 = sorted((int(r0), int(r1)))
            c0, c1 = sorted((int(c0), int(c1)))
            dev = _device()
            stack = Utility._read_folder_stack(imgFold_path, exp_para, dev=dev)
            out, _ = ops.bg_manual(stack, (r0, c0, r1, c1))
            Utility._write_folder_stack(imgFold_path, out, exp_para["channel_list"])
            exp_prop["fct_inputs"]["man_bg_sub"] = {"imgFold_path": imgFold_path}
            exp_prop["img_preProcess"]["bg_sub"] = "Manual"
            Utility.save_exp_prop(exp_path=exp_path, exp_prop=exp_prop)
        return exp_prop

    # ------------------------------------------------------------------ registration (utility.py:274-430)
    @staticmethod
    def im_reg(imgFold_path, reg_channel, reg_mtd, chan_shift, reg_ref, reg_ow):
        exp_path = sep.join(imgFold_path.split(sep)[:-1])
        exp_prop = Utility.open_exp_prop(exp_path=exp_path)
        exp_para = exp_prop["metadata"]
        if exp_para["t"] == 1:
            print(f"--> {imgFold_path} CANNOT be registrated as it is NOT a time sequence")
            exp_prop["img_preProcess"]["reg"] = False
        else:
            exp_prop["img_preProcess"]["reg"] = True
            reg_im_path = imgFold_path + "_Registered"
            if not isdir(reg_im_path):
                mkdir(reg_im_path)
            if any(scandir(reg_im_path)) and not reg_ow:
                print(f"--> {imgFold_path} are already registered")
            else:
                ref_list = ["first", "previous", "mean"]
                if reg_ref not in ref_list:
                    raise ValueError(f"{reg_ref} is not valid. Please only put {ref_list}")
                mtd_list = ["translation", "rigid_body", "scaled_rotation", "affine", "bilinear"]
                if reg_mtd not in mtd_list:
                    raise ValueError(f"{reg_mtd} is not valid. Please only put {mtd_list}")
                if reg_mtd != "translation":
                    raise NotImplementedError(f"reg_mtd='{reg_mtd}': the MI355X path implements the reference's "
                                              "default 'translation' (core.py:77); other StackReg modes stay in the reference")
                print(f"--> Registering {imgFold_path} with method '{reg_mtd}' and reference '{reg_ref}'")
                # chan_shift: the reference's channel-shift pass writes files that the reg_ref pass then
                # overwrites from the raw folder (utility.py:313-330 vs 338-355), so it has no observable
                # effect on the output; it is therefore not run here.
                chans = exp_para["channel_list"]
                dev = _device()
                stack = Utility._read_folder_stack(imgFold_path, exp_para, dev=dev)
                out, shifts = ops.im_reg(stack, chans.index(reg_channel), reg_ref)
                Utility._write_folder_stack(reg_im_path, out, chans)
                exp_prop["fct_inputs"]["im_reg"] = {"imgFold_path": imgFold_path, "reg_channel": reg_channel,
                                                    "reg_mtd": reg_mtd, "reg_ref": reg_ref, "reg_ow": reg_ow}
                if "channel_seg" in exp_prop:
                    if "Images_Registered" in exp_prop["channel_seg"]:
                        if reg_channel not in exp_prop["channel_seg"]["Images_Registered"]:
                            exp_prop["channel_seg"]["Images_Registered"].append(reg_channel)
                    else:
                        exp_prop["channel_seg"]["Images_Registered"] = [reg_channel]
                else:
                    exp_prop["channel_seg"] = {"Images_Registered": [reg_channel]}
        Utility.save_exp_prop(exp_path=exp_path, exp_prop=exp_prop)
        return exp_prop

    # ------------------------------------------------------------------ blur (utility.py:433-468)
    @staticmethod
    def blur_img(imgFold_path, blur_kernel, blur_sigma, blur_ow):
        if type(blur_kernel) != tuple:
            raise TypeError("The input 'blur_kernel' must be a tuple: (int,int)")
        if not all(i % 2 != 0 for i in blur_kernel) and not all(i >= 3 for i in blur_kernel):
            print("The input 'blur_kernel' must contain 2 odd intengers greater or equal to 3")
        exp_path = sep.join(imgFold_path.split(sep)[:-1])
        exp_prop = Utility.open_exp_prop(exp_path=exp_path)
        blur_img_path = join(sep, exp_path + sep, "Images_Blured")
        if not isdir(blur_img_path):
            mkdir(blur_img_path)
        if any(scandir(blur_img_path)) and not blur_ow:
            print("-> Images are already blured")
        else:
            print(f"-> Bluring images using a kernel of {blur_kernel} and sigma of {blur_sigma}")
            names = [f for f in sorted(listdir(imgFold_path)) if f.endswith(".tif")]
            if names:
                dev = _device()
                planes = _planes_to_device([join(sep, imgFold_path + sep, f) for f in names], dev)
                _planes_from_device([join(sep, blur_img_path + sep, f) for f in names],
                                    ops.gauss_blur(planes, blur_kernel, blur_sigma))
            exp_prop["fct_inputs"]["blur_img"] = {"imgFold_path": imgFold_path, "blur_kernel": blur_kernel,
                                                  "blur_sigma": blur_sigma, "blur_ow": blur_ow}
            exp_prop["img_preProcess"]["blur"] = True
            Utility.save_exp_prop(exp_path=exp_path, exp_prop=exp_prop)
        return exp_prop, blur_img_path

    # ------------------------------------------------------------------ mask stitching (utility.py:700-738)
    @staticmethod
    def _stitch_pass(d, ref_of, thr, new_ids):
        """One round of IoU stitching on a device stack (T, Y, X) u16, in place: Cellpose's stitch3D loop
        (new_ids) or the master-mask round of utility.py:717-737.  Overlap counting and relabelling run on
        the GPU; the (labels x labels) IoU logic is the reference's numpy, on the host."""
        mmax = ops.max_label(d[0])
        empty = 0
        for i in (range(1, d.shape[0]) if new_ids else range(d.shape[0])):
            iou = ops.iou_from_overlap(ops.label_overlap(d[i], ref_of(i)))[1:, 1:]
            if not iou.size and empty == 0:
                mmax = ops.max_label(d[i])
                continue
            if not iou.size:
                icount = ops.max_label(d[i])
                istitch = np.arange(mmax + 1, mmax + icount + 1, 1, int)
                mmax += icount
            else:
                iou[iou < thr] = 0.0
                iou[iou < iou.max(axis=0)] = 0.0
                istitch = iou.argmax(axis=1) + 1
                ino = np.nonzero(iou.max(axis=1) == 0.0)[0]
                if new_ids:
                    istitch[ino] = np.arange(mmax + 1, mmax + len(ino) + 1, 1, int)
                    mmax += len(ino)
                else:
                    istitch[ino] = 0
                empty = 1
            istitch = np.append(np.array(0), istitch)
            if istitch.max() > 65535:
                raise ValueError("stitching produced more than 65535 labels (masks are stored as uint16)")
            ops.relabel(d[i], ops.to_device_u16(istitch.astype(np.uint16), d.device), out=d[i])

    @staticmethod
    def modif_stitch3D(masks, stitch_threshold):
        """utility.py:700-738: Cellpose stitch3D, master mask (per-pixel mode over time), second round."""
        thr = 1 - stitch_threshold
        d = ops.to_device_u16(np.ascontiguousarray(masks).astype(np.uint16), _device())
        Utility._stitch_pass(d, lambda i: d[i - 1], thr, new_ids=True)
        master = ops.label_mode(d)
        Utility._stitch_pass(d, lambda i: master, thr, new_ids=False)
        return ops.to_numpy_u16(d).astype(np.asarray(masks).dtype)

    # ------------------------------------------------------------------ channel shift (utility.py:313-330)
    @staticmethod
    def chan_shift_stack(stack, reg_channel):
        """The arithmetic of the reference's chan_shift pass on an in-memory stack (T, C, Z, Y, X) u16: every
        plane registered against the same frame's plane of channel index ``reg_channel`` and warped.  Opt-in:
        im_reg does not run it, because the reference's reg_ref pass re-reads the raw folder and overwrites
        its files (the pass never survives there).  Returns (stack, shifts (T, C, Z, 2))."""
        d = ops.to_device_u16(np.ascontiguousarray(stack).astype(np.uint16), _device())
        out, sh = ops.chan_shift(d, int(reg_channel))
        return ops.to_numpy_u16(out), sh.cpu().numpy()

    # ------------------------------------------------------------------ track clean-up (experiments.py:195-221)
    @staticmethod
    def filter_mask_shapes(masks, shape_threshold=0.2):
        """experiments.py:195-206: an object is removed from the frames where its area is more than
        shape_threshold away (relative) from its mean area over the frames that hold it."""
        d = ops.to_device_u16(np.ascontiguousarray(masks).astype(np.uint16), _device())
        area = ops.label_areas(d).cpu().numpy().astype(np.float64)            # (T, L)
        present = area > 0
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = area.sum(axis=0) / present.sum(axis=0)
        up, down = mean + mean * shape_threshold, mean - mean * shape_threshold
        drop = present & ((area < down) | (area > up))
        drop[:, 0] = False
        lut = np.broadcast_to(np.arange(area.shape[1], dtype=np.uint16), area.shape).copy()
        lut[drop] = 0
        out = ops.relabel(d, ops.to_device_u16(lut, d.device))
        return ops.to_numpy_u16(out).astype(np.float64)      # the reference accumulates into np.zeros(shape)

    @staticmethod
    def renumber_masks(masks):
        """experiments.py:208-211: labels replaced by their rank among the values present."""
        d = ops.to_device_u16(np.ascontiguousarray(masks).astype(np.uint16), _device())
        present = ops.label_areas(d).sum(dim=0).cpu().numpy() > 0
        lut = (np.cumsum(present) - 1).clip(min=0).astype(np.uint16)
        out = ops.relabel(d, ops.to_device_u16(lut, d.device))
        return ops.to_numpy_u16(out).astype(np.asarray(masks).dtype)

    @staticmethod
    def trim_incomplete_tracks(masks, frames=None):
        """experiments.py:217-221: objects that are not present in every frame are removed."""
        d = ops.to_device_u16(np.ascontiguousarray(masks).astype(np.uint16), _device())
        n_frames = np.asarray(masks).shape[0] if frames is None else int(frames)
        present = (ops.label_areas(d) > 0).sum(dim=0).cpu().numpy()
        lut = np.arange(present.shape[0], dtype=np.uint16)
        lut[(present != n_frames)] = 0
        out = ops.relabel(d, ops.to_device_u16(lut, d.device))
        return ops.to_numpy_u16(out).astype(np.asarray(masks).dtype)

    # ------------------------------------------------------------------ compartments (utility.py:776-826)
    @staticmethod
    def compartment_masks(mask_stack, exp_para, rad_ero=10, rad_dil=None, dil_iteration=1, ero_iteration=1):
        """experiments.py:251-262 (Exp_Indiv.cell_compart, the mask arithmetic): optional dilation, erosion,
        membrane = full - eroded.  Returns (mask_mb, mask_cyto, mask_full) as the reference saves them."""
        d = ops.to_device_u16(np.ascontiguousarray(mask_stack).astype(np.uint16), _device())
        if rad_dil:
            d = ops.label_dilate(d, int(rad_dil), int(dil_iteration))
        ero = ops.label_erode(d, int(rad_ero), int(ero_iteration))
        mb = ops.label_sub(d, ero)
        return ops.to_numpy_u16(mb), ops.to_numpy_u16(ero), ops.to_numpy_u16(d)

    @staticmethod
    def erode_mask(mask_stack, exp_para, rad_ero, **kwargs):
        """utility.py:776-801: per-object cv2.erode with disk(rad_ero), summed; kwargs: iterations."""
        d = ops.to_device_u16(np.ascontiguousarray(mask_stack).astype(np.uint16), _device())
        out = ops.label_erode(d, int(rad_ero), int(kwargs.get("iterations", 1)))
        return ops.to_numpy_u16(out).astype(np.float64)      # the reference accumulates into np.zeros(shape)

    @staticmethod
    def dilate_mask(mask_stack, exp_para, rad_dil, **kwargs):
        """utility.py:804-826: cv2.dilate of the label planes with disk(rad_dil); kwargs: iterations."""
        d = ops.to_device_u16(np.ascontiguousarray(mask_stack).astype(np.uint16), _device())
        out = ops.label_dilate(d, int(rad_dil), int(kwargs.get("iterations", 1)))
        return ops.to_numpy_u16(out).astype(np.float64)

    # ------------------------------------------------------------------ distance map (utility.py:829-876)
    @staticmethod
    def ref_mask(imgFold_path, maskLabel, ref_mask_ow=False):
        """utility.py:829-868, loading branch: the reference masks drawn earlier (`Masks_<label>`).  Drawing
        them (get_pre_wound GUI) is not part of this path and stays in the reference."""
        if type(maskLabel) != str:
            raise TypeError(f"Maskname cannot be of type {type(maskLabel)}. Only string is accepted")
        exp_path = sep.join(imgFold_path.split(sep)[:-1])
        mask_ref_path = join(sep, exp_path + sep, "Masks_" + maskLabel)
        if isdir(mask_ref_path) and any(scandir(mask_ref_path)) and not ref_mask_ow:
            print(f"-> Loading reference {'Masks_' + maskLabel} masks from Exp. {exp_path}")
            return Utility.load_mask(maskFold_path=mask_ref_path, channel_seg=maskLabel, do_log=False)
        raise NotImplementedError(f"no reference masks in {mask_ref_path}: draw them with the reference "
                                  "pipeline (Utility.ref_mask); only existing masks are used here")

    @staticmethod
    def apply_dmap(mask_stack, frames):
        """utility.py:871-876: mahotas.distance(~mask, 'euclidean') per frame -> float64 distances."""
        m = np.asarray(mask_stack)
        planes = m[None] if frames == 1 else m
        sq = ops.edt_sq(ops.to_device_u16((planes != 0).astype(np.uint16), _device()))
        dist = np.sqrt(sq.cpu().numpy().astype(np.float64))
        return dist[0] if frames == 1 else dist

    @staticmethod
    def transfo_df(df, channel_list, stim_time=None, start_baseline=0, posCont_time=None, signal_type="linear"):
        """Table post-processing (ratios, deltaF, outliers: utility.py:924-1046) is out of the hot path and
        stays in the reference; subclasses that inherit the reference's Utility override this no-op."""
        if signal_type not in ["linear", "ocsillatory"]:
            raise AttributeError(f"signal_type can only be {['linear', 'ocsillatory']}")
        return df


class Analysis(Utility):
    """analysis.py:9-190 for the two hot methods; constructor as Exp_Indiv (experiments.py:19-37) + Analysis."""

    def __init__(self, exp_path, channel_seg, interval=None, tag=None, tag_ow=False):
        self.exp_path = exp_path
        if exists(join(sep, self.exp_path + sep, "exp_properties.pickle")):
            self.exp_prop = Analysis.open_exp_prop(exp_path=self.exp_path)
            if "img_process" not in self.exp_prop:
                self.exp_prop["img_process"] = {}
        else:
            raise ValueError(f"{self.exp_path} has no preprocessed experiment. Please run the Experiments class first")
        exp_para = self.exp_prop["metadata"]
        self.channel_list = exp_para["channel_list"]
        self.channel_seg = channel_seg
        self.x_size, self.y_size, self.z_size = exp_para["x"], exp_para["y"], exp_para["z"]
        self.frames, self.chan_numb = exp_para["t"], exp_para["c"]
        if tag:
            if not isinstance(tag, dict):
                raise TypeError("tag needs to be a dictionary with 'exp_path' as key and 'tag' as value")
            self.tag = tag[exp_path]
        elif tag_ow:
            print("Tags were overwritten")
            self.tag = exp_path.split(sep)[-2]
        else:
            self.tag = exp_para["tag"]
        self.pixSize = exp_para.get("pixel_microns")
        self.interval = interval if interval else exp_para["interval_sec"]
        self.ts = [0.0] + list(np.round(a=np.linspace(self.interval, self.interval * (self.frames - 1), self.frames - 1) / 60,
                                        decimals=2))

    # -- helpers ---------------------------------------------------------------------------------
    def _exp_name(self):
        split_path = self.exp_path.split(sep)[-1].split("_")
        return "_".join([self.tag, split_path[0], split_path[-1]])

    def _mask_sets(self, maskFold, centroid=False):
        """(mask folder, channel, shape-suffix) triplets exactly as analysis.py:36-59 / 124-137."""
        if maskFold == "Masks_Compartment":
            chan_keys = self.exp_prop["channel_seg"]["Masks_Compartment"]
            shapes = ["cyto"] if centroid else ["_mb", "_cyto", "_full"]
            return [(join(sep, self.exp_path + sep, maskFold), chan, s) for chan in chan_keys for s in shapes]
        if maskFold == "Masks_Class":
            chan_keys = self.exp_prop["channel_seg"]["Masks_Class"]
            folder = join(sep, self.exp_path + sep, self.exp_prop["masks_process"]["classification"]["folder"])
            return [(folder, chan_keys[0], None)]
        chan_keys = self.exp_prop["channel_seg"][maskFold]
        return [(join(sep, self.exp_path + sep, maskFold), chan, None) for chan in chan_keys]

    def _label_stack(self, folder, chan, shape):
        """u16 (T, Z, Y, X) from mask_<chan>[<shape>]_fNNNN_zNNNN.tif (utility.py:590-593)."""
        prefix = "mask_%s%s" % (chan, shape if shape else "")
        if not any(chan in file for file in listdir(folder)):
            raise AttributeError(f"No masks was found for the selected segmented channel: {chan}")
        return Utility._read_folder_stack(folder, self.exp_prop["metadata"], prefix_of=lambda c: prefix,
                                          channels=[chan], dev=_device())[:, 0].contiguous()

    def _measure(self, img_stack, lab_stack):
        dev = _device()
        as_dev = lambda a: a if isinstance(a, torch.Tensor) else ops.to_device_u16(a, dev)   # noqa: E731
        img = as_dev(img_stack) if img_stack is not None else None
        lab = as_dev(lab_stack)
        if img is None:  # centroids only: a zero image channel keeps one code path
            img = torch.zeros((lab.shape[0], 1) + tuple(lab.shape[1:]), dtype=torch.uint16, device=dev)
        return ops.measure_labels(img, lab)

    # -- analysis.py:26-112 ----------------------------------------------------------------------
    def extract_channelData(self, imgFold, maskFold, stim_time=None, start_baseline=0, posCont_time=None,
                            channel_seg=None, df_ow=False, signal_type="linear"):
        fi = self.exp_prop["fct_inputs"]
        if "extract_channelData" not in fi or fi["extract_channelData"]["maskFold"] != maskFold or df_ow:
            print(f"---> Extracting channel data with {maskFold}")
            sets = self._mask_sets(maskFold)
            if maskFold == "Masks_Compartment":
                mask_keys = [f"{chan}{shape}" for chan in self.channel_list for shape in ["_mb", "_cyto", "_full"]]
            else:
                mask_keys = [chan for chan in self.channel_list]
            keys = ["cell", "frames", "time", "mask_chan", "tag", "exp"] + mask_keys
            is_class = maskFold == "Masks_Class"
            if is_class:                                                      # analysis.py:46-50, 63
                class_keys = self.exp_prop["channel_seg"]["Masks_Class"]
                pos_lst = self.exp_prop["masks_process"]["classification"]["pos"]
                neg_lst = self.exp_prop["masks_process"]["classification"]["neg"]
                keys += ["cell_class", "pos_cell"]
            exp_name = self._exp_name()
            img_stack = Utility._read_folder_stack(join(sep, self.exp_path + sep, imgFold), self.exp_prop["metadata"],
                                                   dev=_device())
            df = pd.DataFrame()
            by_chan = {}
            for folder, chan, shape in sets:
                by_chan.setdefault(chan, []).append((folder, shape))
            for chan, lst in by_chan.items():
                d = {k: [] for k in keys}
                tables = [self._measure(img_stack, self._label_stack(folder, chan, shape)) for folder, shape in lst]
                # analysis.py:74-98 appends row by row (frame, mask set, object; per row the channel columns in the order
                # "for chan: for name"); the same lists are filled a frame's table at a time
                cols_of = {name: [ci for ci, ch in enumerate(self.channel_list) if ch in name] for name in mask_keys}   # analysis.py:96-98, incl. its substring match
                for f, t in enumerate(self.ts):
                    for (folder, shape), tab in zip(lst, tables):
                        tb = tab[f]
                        ids = tb["ids"]
                        n = len(ids)
                        if not n:
                            continue
                        d["cell"].extend([f"{exp_name}_{chan}_cell{obj}" for obj in ids])
                        d["mask_chan"].extend([chan] * n)
                        d["tag"].extend([self.tag] * n)
                        d["exp"].extend([exp_name] * n)
                        d["frames"].extend([f + 1] * n)
                        d["time"].extend([t] * n)
                        if is_class:                                          # analysis.py:87-92, as written there:
                            for obj in ids:
                                d["cell_class"].append(f"if cell is {class_keys[1]}")
                                if obj in pos_lst:                            # an object in neither list gets no entry
                                    d["pos_cell"].append(1)
                                elif obj in neg_lst:
                                    d["pos_cell"].append(0)
                        for name, cols in cols_of.items():
                            if len(cols) == 1:
                                d[name].extend(tb["mean"][:, cols[0]].tolist())
                            elif cols:                                        # several channels match: per row, in channel order
                                d[name].extend(tb["mean"][:, cols].ravel().tolist())
                df = pd.concat([df, pd.DataFrame.from_dict(d)], ignore_index=True)
            self.df_analysis = type(self).transfo_df(df=df, channel_list=mask_keys, stim_time=stim_time,
                                                     start_baseline=start_baseline, posCont_time=posCont_time,
                                                     signal_type=signal_type)
            self.exp_prop["df_analysis"] = self.df_analysis
            fi["extract_channelData"] = {"imgFold": imgFold, "maskFold": maskFold, "channel_seg": channel_seg}
            Analysis.save_exp_prop(exp_path=self.exp_path, exp_prop=self.exp_prop)
            df.to_csv(join(sep, self.exp_path + sep, "df_analysis.csv"), index=False)
        else:
            print(f"---> Channel data are already extracted with {maskFold}")
            self.df_analysis = pd.read_csv(join(sep, self.exp_path + sep, "df_analysis.csv"))

    # -- analysis.py:114-190 ---------------------------------------------------------------------
    def extract_centroids(self, maskFold, channel_seg=None, df_ow=False):
        fi = self.exp_prop["fct_inputs"]
        if "extract_centroids" not in fi or fi["extract_centroids"]["maskFold"] != maskFold or df_ow:
            print(f"---> Extracting centroids for {maskFold}")
            axes_keys = ["y", "x"] if self.z_size == 1 else ["y", "x", "z"]
            keys = ["cell", "frames", "time", "mask_chan", "mask_ID"] + axes_keys
            exp_name = self._exp_name()
            df = pd.DataFrame()
            for folder, chan, shape in self._mask_sets(maskFold, centroid=True):
                d = {k: [] for k in keys}
                tab = self._measure(None, self._label_stack(folder, chan, shape))
                for f, t in enumerate(self.ts):          # analysis.py:153-173, a frame's table at a time
                    tb = tab[f]
                    ids = tb["ids"]
                    n = len(ids)
                    if not n:
                        continue
                    d["cell"].extend([f"{exp_name}_{chan}_cell{obj}" for obj in ids])
                    d["mask_chan"].extend([chan] * n)
                    d["frames"].extend([f + 1] * n)
                    d["time"].extend([t] * n)
                    d["mask_ID"].extend(ids.tolist())
                    c = tb["centroid_rne"].astype(np.int64)
                    if self.z_size == 1:
                        d["y"].extend(c[:, 0].tolist())
                        d["x"].extend(c[:, 1].tolist())
                    else:
                        d["z"].extend(c[:, 0].tolist())
                        d["y"].extend(c[:, 1].tolist())
                        d["x"].extend(c[:, 2].tolist())
                df = pd.concat([df, pd.DataFrame.from_dict(d)], ignore_index=True)
            if not hasattr(self, "df_analysis"):
                self.df_analysis = pd.DataFrame.from_dict(df)
            else:
                self.df_analysis = pd.merge(self.df_analysis, pd.DataFrame.from_dict(df),
                                            on=["cell", "frames", "time", "mask_chan"])
            self.exp_prop["df_analysis"] = self.df_analysis
            fi["extract_centroids"] = {"maskFold": maskFold, "channel_seg": channel_seg}
            Analysis.save_exp_prop(exp_path=self.exp_path, exp_prop=self.exp_prop)
            self.df_analysis.to_csv(join(sep, self.exp_path + sep, "df_analysis.csv"), index=False)
        else:
            print(f"---> Centroids are already extracted for {maskFold}")
            self.df_analysis = pd.read_csv(join(sep, self.exp_path + sep, "df_analysis.csv"))


    # -- analysis.py:262-304 ---------------------------------------------------------------------
    def pixel_distance(self, imgFold, maskFold, channel_seg=None, maskLabel="wound", pix_ana_ow=False,
                       ref_mask_ow=False):
        fi = self.exp_prop["fct_inputs"]
        if "pixel_distance" not in fi or pix_ana_ow or not exists(join(sep, self.exp_path + sep, "df_pixel.csv")):
            dev = _device()
            img_path = join(sep, self.exp_path + sep, imgFold)
            ref = np.asarray(Analysis.ref_mask(imgFold_path=img_path, maskLabel=maskLabel, ref_mask_ow=ref_mask_ow))
            chan_seg = channel_seg if channel_seg else self.channel_seg
            masks = np.asarray(Analysis.load_mask(maskFold_path=join(sep, self.exp_path + sep, maskFold),
                                                  channel_seg=chan_seg))
            t = self.frames
            ref = ref.reshape(t, *ref.shape[-2:])
            if masks.ndim != (2 if t == 1 else 3):
                raise ValueError("pixel_distance needs 2-D cell masks per frame (volumetric masks cannot index the "
                                 "maximum projection, analysis.py:281-282)")
            masks = masks.reshape(t, *masks.shape[-2:])
            exp_para = self.exp_prop["metadata"]
            stack = Utility._read_folder_stack(img_path, exp_para, dev=dev)                       # (T, C, Z, Y, X)
            if self.z_size > 1:                                                                   # np.amax over z
                img = torch.stack([ops.mip(stack[:, c]) for c in range(stack.shape[1])], 1)
            else:
                img = stack[:, :, 0].contiguous()
            sq = ops.edt_sq(ops.to_device_u16((ref != 0).astype(np.uint16), dev))
            counts, dmap, vals = ops.masked_pixels(ops.to_device_u16(masks.astype(np.uint16), dev), img.contiguous(),
                                                   sq, pix_size=float(self.pixSize))
            counts = counts.cpu().numpy()
            d = {"dmap": dmap.cpu().numpy().astype(int)}
            v = ops.to_numpy_u16(vals)
            for ci, chan in enumerate(self.channel_list):
                d[chan] = v[ci].astype(float)
            self.df_pixel = pd.DataFrame.from_dict(d)
            self.df_pixel["time"] = np.repeat(np.asarray(self.ts, dtype=float), counts)
            self.df_pixel["frame"] = np.repeat(np.arange(1, t + 1), counts)
            self.df_pixel["tag"] = self.tag
            self.df_pixel["exp"] = self._exp_name()
            self.exp_prop["df_pixel"] = self.df_pixel
            fi["pixel_distance"] = {"imgFold": imgFold, "maskFold": maskFold, "channel_seg": channel_seg,
                                    "maskLabel": maskLabel}
            Analysis.save_exp_prop(exp_path=self.exp_path, exp_prop=self.exp_prop)
            self.df_pixel.to_csv(join(sep, self.exp_path + sep, "df_pixel.csv"), index=False)
        else:
            print(f"---> {maskLabel} mask was already created for pixel distance and value already extracted")
            self.df_pixel = pd.read_csv(join(sep, self.exp_path + sep, "df_pixel.csv"))
