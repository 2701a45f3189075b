/*
 * iap.h -- C-ABI of libiap_hip.so: the MI355X (gfx950) hot path behind the reference's
 * background-subtract -> register -> per-label-measure stages.
 *
 * Every entry point takes raw DEVICE pointers, sizes and a hipStream_t passed as void*;
 * the caller owns every buffer (in the Python host: torch tensors); the library never
 * allocates: scratch comes from the `ws` argument (query the size with the matching
 * *_ws_bytes function).  All work is enqueued on `stream`; nothing synchronises with the
 * host.  Return value: 0 = ok, negative = invalid argument (IAP_E_*), positive = the
 * hipError_t of a failed runtime call.  iap_last_error() returns a thread-local message.
 *
 * Stack layout (reference utility.py:487-524 load_stack, planar on disk):
 *   images  u16 [T][C][Z][Y][X]   contiguous, X fastest
 *   labels  u16 [T][Z][Y][X]      (utility.py:616-618 save_mask writes uint16)
 *
 * Each declaration cites the reference code it replaces (paths under pipeline/).
 */
#ifndef IAP_H
#define IAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IAP_OK 0
#define IAP_E_ARG (-1)       /* null pointer / non-positive size / bad enum          */
#define IAP_E_ALIGN (-2)     /* pointer not 16-byte aligned                          */
#define IAP_E_WS (-3)        /* workspace too small                                  */
#define IAP_E_UNSUPPORTED (-4) /* size outside what the kernels are built for       */

/* library / build identification; iap_last_error: message of the last failing call on this thread */
const char* iap_version(void);
const char* iap_last_error(void);

/* ------------------------------------------------------------------ background ------- */

/* SMO operator value per pixel (float32) of n planes.  in_kind: 0 = u16 planes, 1 = f32 planes.
 * Replaces smo.SMO.smo_image as used by utility.py:252,260 (null-distribution threshold and
 * tests).  sigma > 0 adds the gaussian pre-smoothing (scipy 'reflect', truncate 4) that SMO(sigma=...)
 * applies (utility.py:242, 252; kwarg plumbed from core.py:76) and needs a workspace of
 * iap_smo_image_ws_bytes (0 at sigma == 0: ws may then be NULL). */
size_t iap_smo_image_ws_bytes(int64_t n, int Y, int X, float sigma);
int iap_smo_image(const void* in, int in_kind, float* out, int64_t n, int Y, int X,
                  float sigma, int size, void* ws, size_t ws_bytes, void* stream);

/* Workspace bytes for iap_bg_smo_u16 on n planes with the given sigma (the two float planes per batch
 * plane are only needed for sigma > 0). */
size_t iap_bg_smo_ws_bytes(int64_t n, int Y, int X, float sigma);

/* utility.py:255-264 smo_bg_sub hot loop, for n independent u16 planes:
 *   mask = SMO(plane) < threshold ; med = median(plane[mask]) ;
 *   out = trunc(max(plane - med, 0)) as u16.           in == out is allowed (in place).
 * med2_out (optional, device int32[n]) receives 2*median (half-integers are possible);
 * nbg_out (optional, device int32[n]) the number of background pixels.
 * out == NULL (med2_out required): levels only, nothing is written back -- for a caller that hands
 * med2_out to iap_im_reg_sub_u16, which subtracts as it reads. */
int iap_bg_smo_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X,
                   float sigma, int size, float threshold,
                   int32_t* med2_out, int32_t* nbg_out,
                   void* ws, size_t ws_bytes, void* stream);

/* utility.py:228-233 man_bg_sub arithmetic for n planes: bg = rint(mean(ROI)) per plane
 * (ROI rows r0..r1, cols c0..c1 inclusive, identical for all planes), out = max(in-bg,0).
 * bg_out (optional, device int32[n]).  ws: n * 16 bytes. */
int iap_bg_const_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X,
                     int r0, int c0, int r1, int c1, int32_t* bg_out,
                     void* ws, size_t ws_bytes, void* stream);

/* The same computation as iap_bg_const_u16 for callers that cannot allocate or carry a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * The ROI sums and levels live in a file-scope device array of the library (IAP_BG_CONST_STATIC_MAX_PLANES * 16
 * bytes = 64 KiB on every GPU the library runs on, held until the process ends); the call makes no hipMalloc /
 * hipFree and takes no workspace.  n above IAP_BG_CONST_STATIC_MAX_PLANES is refused with IAP_E_UNSUPPORTED.  Call
 * it with the stream's device current.  All calls on one GPU share that array: calls that overlap in time on one
 * GPU (different streams or threads) race; calls in stream order are safe. */
#define IAP_BG_CONST_STATIC_MAX_PLANES 4096
int iap_bg_const_static_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X,
                            int r0, int c0, int r1, int c1, int32_t* bg_out, void* stream);

/* utility.py:462 blur_img: cv2.GaussianBlur(im,(kx,ky),sigma).astype(uint16) semantics,
 * reflect-101 border, fp32 taps (OpenCV 4.1.2's getGaussianKernel, the reference's pin: environment.yml:15),
 * round-half-even + saturate.  kx, ky odd, <= 63.  sigma is a double, as cv2's sigmaX: the taps are worked out
 * in double from it. */
int iap_gauss_blur_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X,
                       int kx, int ky, double sigma, void* stream);

/* ------------------------------------------------------------------ registration ----- */

#define IAP_REF_FIRST 0     /* utility.py:332-355 */
#define IAP_REF_MEAN 1      /* utility.py:356-380 */
#define IAP_REF_PREVIOUS 2  /* utility.py:381-413 */

#define IAP_REG_UPSAMPLE 1   /* sub-pixel refinement (0.1 px matrix-DFT grid + parabola)   */
#define IAP_REG_NORMALIZE 2  /* phase-normalise the cross-power spectrum (default: plain)  */

/* np.amax(stack, axis=z) of utility.py:335,341: in u16 [n][Z][Y][X] -> out u16 [n][Y][X];
 * in_stride = elements between consecutive items of `in`. */
int iap_mip_u16(const uint16_t* in, int64_t in_stride, uint16_t* out, int64_t n, int Z, int Y, int X,
                void* stream);

/* The same over planes read as trunc(max(v - sub2/2, 0)) (see iap_im_reg_sub_u16): plane z of item i takes
 * sub2[i * sub_stride + z].  With Z == 1 this is the subtracted copy of one plane per item. */
int iap_mip_sub_u16(const uint16_t* in, int64_t in_stride, uint16_t* out, int64_t n, int Z, int Y, int X,
                    const int32_t* sub2, int64_t sub_stride, void* stream);

/* Workspace bytes for iap_reg_estimate_u16 / iap_im_reg_u16 processing `batch` frames per
 * launch group (batch >= 1).  Y, X are plane sizes; the FFT window is the centred
 * power-of-two crop (<= 4096) of each axis. */
size_t iap_reg_ws_bytes(int Y, int X, int Z, int batch);

/* Translation estimate (replaces pystackreg StackReg.register, utility.py:340,365,393) for n
 * moving planes against one reference plane (ref_kind 0: u16, 1: u32 sum-of-planes, 2: f32).
 * mov planes are `mov_stride` elements apart.  shifts_out: device f64 [n][2] = (dy, dx) to
 * APPLY to each moving plane (scipy.ndimage.shift convention). */
int iap_reg_estimate_u16(const void* ref, int ref_kind, const uint16_t* mov, int64_t mov_stride,
                         int64_t n, int Y, int X, int flags, double* shifts_out,
                         void* ws, size_t ws_bytes, void* stream);

/* Channel shift (utility.py:313-330, sr.register_transform(ref_plane_of_reg_channel, plane) per frame,
 * channel and z-plane): n independent (reference, moving) pairs, pair i = ref + i*ref_stride against
 * mov + i*mov_stride; shifts_out f64 [n][2] as above.  Workspace: iap_reg_pairs_ws_bytes (the plain
 * estimate's plus one reference spectrum per frame of the batch). */
size_t iap_reg_pairs_ws_bytes(int Y, int X, int batch);
int iap_reg_estimate_pairs_u16(const uint16_t* ref, int64_t ref_stride, const uint16_t* mov, int64_t mov_stride,
                               int64_t n, int Y, int X, int flags, double* shifts_out,
                               void* ws, size_t ws_bytes, void* stream);

/* StackReg.transform + clip + astype(uint16) of utility.py:351-355 fused over a stack:
 * n_frames frames of planes_per_frame planes each; frame f is warped by shifts[f] (device f64
 * [n][2]): out[y][x] = cubic-spline(in)(y-dy, x-dx), outside -> 0, negatives -> 0, truncate,
 * saturate at 65535.  Integer shifts are exact copies.  in != out. */
int iap_apply_shift_u16(const uint16_t* in, uint16_t* out, int64_t n_frames, int64_t planes_per_frame,
                        int Y, int X, const double* shifts, void* stream);

/* iap_apply_shift_u16 over planes read as trunc(max(v - sub2/2, 0)), sub2: device int32 [n_frames * planes_per_frame]
 * (see iap_im_reg_sub_u16; used where estimate and warp are separate calls, e.g. time-block sharding). */
int iap_apply_shift_sub_u16(const uint16_t* in, uint16_t* out, int64_t n_frames, int64_t planes_per_frame,
                            int Y, int X, const double* shifts, const int32_t* sub2, void* stream);

/* utility.py:274-430 im_reg (translation) on a resident stack u16 [T][C][Z][Y][X]:
 * estimate on channel reg_channel (z-MIP), reference per ref_mode, apply to every channel and
 * plane.  out: same shape (must not alias in).  shifts_out: device f64 [T][2].  With IAP_REF_FIRST frame 0
 * is the reference itself and gets the shift (0, 0) exactly (a copy), as IAP_REF_PREVIOUS copies frame 0. */
int iap_im_reg_u16(const uint16_t* in, uint16_t* out, int64_t T, int C, int Z, int Y, int X,
                   int reg_channel, int ref_mode, int flags, double* shifts_out,
                   void* ws, size_t ws_bytes, void* stream);

/* core.py:99-112 pre_process_all runs smo_bg_sub (utility.py:255-264) and then im_reg (utility.py:274-430) over
 * the same stack; with the stack resident the subtraction needs no pass of its own.  As iap_im_reg_u16, but `in`
 * is the RAW stack and every kernel that reads it (row FFT, MIP, plane sum, warp) sees
 * trunc(max(v - sub2[plane]/2, 0)), plane = (t*C + c)*Z + z: bit for bit what iap_bg_smo_u16 (sub2 = its
 * med2_out) or iap_bg_const_u16 (sub2 = 2 * its bg_out) would have stored first.  sub2: device int32 [T*C*Z]. */
int iap_im_reg_sub_u16(const uint16_t* in, uint16_t* out, int64_t T, int C, int Z, int Y, int X,
                       int reg_channel, int ref_mode, int flags, const int32_t* sub2, double* shifts_out,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ measurement ------ */

/* Number of u64 accumulator fields per (frame, label): count, sum_z, sum_y, sum_x, sum_I[C]. */
#define IAP_ACC_FIELDS(C) (4 + (C))

/* analysis.py:80-98 (mean intensity), :164-173 (centroid), experiments.py:200-201 (area):
 * one pass over image + label stack accumulating exact integer sums per (frame, label):
 *   acc u64 [T][L][4+C], zeroed by this call; labels >= L are ignored (pass L = max+1).
 * Label 0 only gets its count (it is the value np.unique(...)[1:] normally drops). */
int iap_measure_labels_u16(const uint16_t* img, const uint16_t* labels,
                           int64_t T, int C, int Z, int Y, int X, int L,
                           uint64_t* acc, void* stream);

/* The same pass, also reporting frame_max u32 [T] (zeroed by the call): the largest label value of every
 * frame, labels >= L included.  It lets a caller that does not know the label range (the reference takes it from
 * np.unique(mask), analysis.py:80) measure with a provisional L and repeat the call only when max(frame_max) >= L,
 * instead of reading the label stack once more beforehand (iap_max_u16). */
int iap_measure_labels_max_u16(const uint16_t* img, const uint16_t* labels,
                               int64_t T, int C, int Z, int Y, int X, int L,
                               uint64_t* acc, uint32_t* frame_max, void* stream);

/* Accumulators -> tables (device): area i64 [T][L]; mean f64 [T][L][C] = (f64)sum/(f64)count
 * (numpy nanmean(where=) semantics, bit-exact); centroid f64 [T][L][3] (z,y,x);
 * centroid_rne i64 [T][L][3] (round-half-even, Python round()).  Rows with count 0: all 0. */
int iap_measure_finalize(const uint64_t* acc, int64_t T, int L, int C,
                         int64_t* area, double* mean, double* centroid, int64_t* centroid_rne,
                         void* stream);

/* Max label value of a u16 label stack of n elements -> device u32 (zeroed by the call). */
int iap_max_u16(const uint16_t* labels, int64_t n, uint32_t* out, void* stream);

/* ------------------------------------------------------------------ distance map / pixel table */

/* SURVEY 8f rank 2.  Utility.apply_dmap (utility.py:871-876): mahotas.distance(~mask, 'euclidean') per
 * frame = Euclidean distance of every pixel to the nearest pixel of the reference (wound) mask.  Exact
 * squared distances as int32 for n planes; mask != 0 marks the reference region (distance 0).  A plane
 * without any mask pixel yields 30000^2 + ... (no finite answer exists).  ws: iap_edt_ws_bytes. */
size_t iap_edt_ws_bytes(int64_t n, int Y, int X);
int iap_edt_sq_u16(const uint16_t* mask, int32_t* sq_out, int64_t n, int Y, int X,
                   void* ws, size_t ws_bytes, void* stream);

/* Analysis.pixel_distance (analysis.py:276-283): per frame, for the pixels of a cell mask in numpy
 * boolean-indexing (row-major) order: d = round(sqrt(sq) * pix_size) (np.round, half to even) and the
 * pixel's value in every channel.  Two calls: iap_mask_count_u16 fills chunk_offsets
 * (n x iap_mask_chunks(Y, X) int32, exclusive prefix per frame) and totals (int64 per frame); the caller
 * turns totals into frame_base (exclusive prefix over frames, device int64[n]) and sizes the outputs:
 * out_d int32[total], out_v u16[C][total].  img: frame f, channel c at img + f*img_frame_stride +
 * c*img_chan_stride (elements).  sq may be NULL (out_d = 0) and out_d may be NULL. */
int iap_mask_chunks(int Y, int X);
int iap_mask_count_u16(const uint16_t* mask, int64_t n, int Y, int X, int32_t* chunk_offsets,
                       int64_t* totals, void* stream);
int iap_masked_gather_u16(const uint16_t* mask, const int32_t* sq, const uint16_t* img, int64_t img_frame_stride,
                          int64_t img_chan_stride, int C, int64_t n, int Y, int X, const int32_t* chunk_offsets,
                          const int64_t* frame_base, double pix_size, int64_t total, int32_t* out_d,
                          uint16_t* out_v, void* stream);

/* ------------------------------------------------------------------ mask stitching ------- */

/* SURVEY 8f rank 3, the array work of Utility.modif_stitch3D (utility.py:700-738).
 * iap_label_overlap_u16: cellpose.metrics._label_overlap for n plane pairs of px pixels:
 *   overlap[p][i][j] = #{a == i and b == j}, u32 [n][na_max+1][nb_max+1], zeroed by the call (labels
 *   above the bounds are ignored).  IoU = overlap / (row sum + column sum - overlap) is left to the host.
 * (Planes that are not 16-byte aligned -- frames of a stack whose planes are not a multiple of 8 pixels -- are
 * read pixel by pixel by iap_label_overlap_u16, iap_relabel_u16 and iap_max_u16 instead of being refused.)
 * iap_relabel_u16: out = lut[frame][in] (masks[i] = istitch[masks[i]], utility.py:724,735); lut_stride
 *   elements between the tables of consecutive frames (0: one table for all).
 * iap_label_mode_u16: master mask of utility.py:707-711: per pixel the most frequent non-zero label over T
 *   frames (ties: the smaller label; 0 when never labelled).  Pixels with more than 8 distinct labels get
 *   65535 and are counted in *n_overflow (device u32) for the caller to redo. */
int iap_label_overlap_u16(const uint16_t* a, const uint16_t* b, int64_t n, int64_t px, int na_max, int nb_max,
                          uint32_t* overlap, void* stream);
int iap_relabel_u16(const uint16_t* in, uint16_t* out, int64_t n, int64_t px, const uint16_t* lut,
                    int lut_stride, void* stream);
int iap_label_mode_u16(const uint16_t* masks, int T, int64_t px, uint16_t* out, uint32_t* n_overflow,
                       void* stream);

/* ------------------------------------------------------------------ compartment masks ---- */

/* SURVEY 8f rank 4.  Utility.erode_mask (utility.py:776-801): the sum over objects of
 * cv2.erode(mask == obj ? obj : 0, disk(r), iterations) = a pixel keeps its label iff the whole disk round
 * it (inside the image) carries that label; Utility.dilate_mask (utility.py:804-826): cv2.dilate of the
 * label image = largest label inside the disk.  disk(r): dx^2 + dy^2 <= r^2, r <= 31.  n planes of Y x X;
 * in != out; iterations > 1 need ws of n*Y*X*2 bytes (distinct from in and out). */
int iap_label_erode_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int radius, int iterations,
                        void* ws, size_t ws_bytes, void* stream);
int iap_label_dilate_u16(const uint16_t* in, uint16_t* out, int64_t n, int Y, int X, int radius, int iterations,
                         void* ws, size_t ws_bytes, void* stream);

/* Exp_Indiv.cell_compart (experiments.py:259): mask_mb = mask_stack - mask_erode, element-wise on n u16
 * values (the eroded mask is a subset of the full one with the same labels, so no value wraps). */
int iap_label_sub_u16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, void* stream);

/* ------------------------------------------------------------------ plane files (host) -- */

/* The step on either side of the hot path: the reference keeps every
 * (channel, frame, z) plane in its own baseline TIFF (tifffile.imwrite(plane.astype(uint16)),
 * utility.py:155-166, 264, 355, 616) and re-reads the stack file by file (utility.py:487-524
 * load_stack, :558-595 load_mask).  These calls move n planes of Y x X u16 between n files and
 * one contiguous HOST buffer (pinned for a following async copy) with a pool of `threads`
 * threads; no device work, no stream.  Read accepts classic TIFF, II or MM, one 16-bit unsigned
 * sample, uncompressed, stripped; anything else returns IAP_E_UNSUPPORTED naming the file (the
 * shim then uses its generic reader).  Write produces little-endian single-strip files. */
int iap_tiff_probe(const char* path, int* Y, int* X, int* bits, int* sample_format, int* compression, int* samples);
int iap_tiff_read_u16(const char* const* paths, int64_t n, uint16_t* dst, int Y, int X, int threads);
int iap_tiff_write_u16(const char* const* paths, int64_t n, const uint16_t* src, int Y, int X, int threads);

/* ------------------------------------------------------------------ instrumentation ---- */

/* Optional per-kernel timing with HIP events recorded on the launch stream (no counterpart in
 * the reference; used by bench.py for the roofline line).  iap_prof_enable(1) clears and starts
 * recording, iap_prof_enable(0) stops; iap_prof_collect synchronises the recorded events and
 * returns the summed duration and launch count of one kernel id (0 .. iap_prof_kernel_count()-1). */
void iap_prof_enable(int on);
/* Restrict the recording to one kernel id (-1: every kernel, the default): a timed run then carries two events per
 * launch of that kernel only. */
void iap_prof_select(int kernel_id);
int iap_prof_kernel_count(void);
const char* iap_prof_kernel_name(int kernel_id);
int iap_prof_collect(int kernel_id, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* IAP_H */
