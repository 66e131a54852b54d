/*
 * deepcharuco_amd.h -- C ABI of libdeepcharuco_amd.so (MI355X / gfx950 only).
 *
 * The reference (JunkyByte/deepcharuco) has no FFI: its boundary is the Python
 * function API of src/inference.py.  This header is what a binding for that
 * path links against; deepcharuco_amd/_lib.py binds it with ctypes and
 * deepcharuco_amd/inference.py rebuilds the reference's Python API on top
 * (see INTEGRATION.md).  Each entry point cites the reference interface it
 * replaces (paths relative to /root/reference/src).
 *
 * Conventions
 *   - every `d_` pointer is a DEVICE pointer owned by the caller (e.g. a
 *     torch tensor's data_ptr()); every `h_` pointer is host memory;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no
 *     entry point synchronises the device or allocates after create();
 *   - return value: 0 ok, <0 argument/shape error (DCX_E_*), >0 hipError_t;
 *   - no C++ exception crosses the boundary; handles are not thread-safe
 *     (use one handle per stream);
 *   - activations inside the library use the "C4" layout
 *     [N][C/4][H][W][4] float32 (channel-quad planar); NCHW only at the API.
 */
#ifndef DEEPCHARUCO_AMD_H
#define DEEPCHARUCO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCX_E_ARG      (-1)   /* null pointer / bad scalar */
#define DCX_E_SHAPE    (-2)   /* H or W below 8, patch not 24x24, capacity overflow ... */
#define DCX_E_WS       (-3)   /* workspace too small */
#define DCX_E_NIDS     (-4)   /* n_ids outside [1, 62] at create(); dust_bin outside [0, 255] at decode */

typedef struct dcx_detector dcx_detector;   /* dcModel  (models/net.py:9-99)        */
typedef struct dcx_refiner  dcx_refiner;    /* RefineNet (models/refinenet.py:9-115) */

const char* dcx_version(void);
/* human readable text for a return code (static storage) */
const char* dcx_error_string(int code);

/* ---- model lifetime: replaces load_models() inference.py:73-84 ----------------------
 * h_tensors: host float32 arrays in the order deepcharuco_amd.weights.state_dict_keys()
 * gives, i.e. for every conv in forward order: weight (OIHW), bias and -- where a
 * BatchNorm2d follows -- gamma, beta, running_mean, running_var.  The library packs the
 * weights into its MFMA operand layout, folds eval-mode BN (eps 1e-5) into a per-channel
 * (alpha, beta') pair exactly as ATen's CPU inference path does, and uploads them.
 * Detector: 12 convs / 10 BN = 64 tensors; RefineNet: 12 convs / 11 BN = 68 tensors.   */
int dcx_detector_create(dcx_detector** out, const float* const* h_tensors, int n_tensors, int n_ids);
int dcx_detector_destroy(dcx_detector* det);
int dcx_refiner_create(dcx_refiner** out, const float* const* h_tensors, int n_tensors);
int dcx_refiner_destroy(dcx_refiner* rf);

/* ---- workspace sizing (bytes of device scratch the forward calls need) -------------- */
size_t dcx_detector_workspace_bytes(const dcx_detector* det, int batch, int height, int width);
size_t dcx_refiner_workspace_bytes(const dcx_refiner* rf, int max_patches);

/* ---- colour conversion: cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) call at inference.py:40 ---------
 * 8-bit BGR frames (interleaved, row pitch / frame stride in BYTES) -> dense gray u8 [B][H][W].  OpenCV is third-party and not
 * vendored in the reference ("parity unpinned" for this one step); the reference pins opencv-contrib-python
 * >= 4.6, < 4.12 (requirements.txt:5).
 *   dcx_bgr2gray           OpenCV 4.x RGB2Gray<uchar> (imgproc/src/color.hpp: gray_shift = 15, BY15 / GY15 / RY15):
 *                              gray = (3735 B + 19235 G + 9798 R + 16384) >> 15
 *   dcx_bgr2gray_legacy14  the older 14-bit form (B2Y / G2Y / R2Y, yuv_shift; in 4.x only 16-bit images and YUV use it):
 *                              gray = (1868 B + 9617 G + 4899 R + 8192) >> 14
 * The two differ by one level on ~0.26 % of colour pixels and never on gray-replicated input.                          */
int dcx_bgr2gray(const uint8_t* d_bgr, long frame_stride, int pitch, int batch, int height, int width,
                 uint8_t* d_gray, void* stream);
int dcx_bgr2gray_legacy14(const uint8_t* d_bgr, long frame_stride, int pitch, int batch, int height, int width,
                          uint8_t* d_gray, void* stream);

/* ---- pre-processing ------------------------------------------------------------------
 * pre_bgr_image models/model_utils.py:46-50:  out = (float(g) - 128) / 255 (IEEE division).
 * d_gray [n] u8 -> d_out [n] f32.                                                        */
int dcx_pre_image(const uint8_t* d_gray, float* d_out, size_t n, void* stream);

/* ---- detector forward: dcModel.forward net.py:50-80 / infer_image net.py:82-99 -------
 * Input either u8 gray frames (d_frames_u8, row pitch in bytes, frame stride in bytes;
 * normalised on the fly exactly like dcx_pre_image) or already-normalised f32 images
 * (d_images_f32, dense [B][H][W]); exactly one of the two must be non-null.
 * Outputs (either may be null): logits in NCHW, loc [B][65][H/8][W/8], ids [B][n_ids+1][H/8][W/8].
 * The C4 logits stay in the workspace for dcx_detector_decode().                          */
int dcx_detector_forward(const dcx_detector* det,
                         const uint8_t* d_frames_u8, long frame_stride, int pitch,
                         const float* d_images_f32,
                         int batch, int height, int width,
                         void* d_ws, size_t ws_bytes,
                         float* d_loc_nchw, float* d_ids_nchw, void* stream);

/* ---- decode: pred_to_keypoints model_utils.py:81-88 (= pred_argmax :53-78 +
 *      label_to_keypoints :91-124), batched -----------------------------------------------
 * Per cell: loc = argmax_c loc_logits (first max), id = argmax_c ids_logits, id = dust_bin
 * where loc == 64; cells with id != dust_bin emit a row {x = 8*cx + loc%8, y = 8*cy + loc/8,
 * id, cell = cy*Wc + cx} in raster order per frame.  d_counts[b] = number of firing cells
 * (may exceed kmax; only the first kmax rows are stored).  d_rows: int32 [B][kmax][4].
 * dust_bin is the reference's `dust_bin_ids` argument: any value in [0, 255] (else DCX_E_NIDS) with the reference's
 * semantics `ids != dust_bin_ids` -- it normally equals n_ids but is not required to.
 * Optional dense maps d_loc_argmax / d_ids_argmax: int32 [B][Hc][Wc] (ids map is post-mask).
 * dcx_detector_decode reads the logits the preceding dcx_detector_forward left in d_ws and
 * uses 4 bytes per cell of scratch in it (arg-max of all cells in parallel, then one ordered
 * compaction per frame); dcx_pred_to_keypoints takes caller NCHW logits (the reference's own
 * signature) and needs no scratch (one workgroup per frame).                                */
int dcx_detector_decode(const dcx_detector* det, int batch, int height, int width,
                        void* d_ws, int dust_bin, int kmax,
                        int32_t* d_counts, int32_t* d_rows,
                        int32_t* d_loc_argmax, int32_t* d_ids_argmax, void* stream);
int dcx_pred_to_keypoints(const float* d_loc_nchw, const float* d_ids_nchw,
                          int batch, int n_loc, int n_ids1, int hc, int wc,
                          int dust_bin, int kmax,
                          int32_t* d_counts, int32_t* d_rows,
                          int32_t* d_loc_argmax, int32_t* d_ids_argmax, void* stream);
/* label_to_keypoints model_utils.py:91-124 on caller label maps (class indices, int64 [B][Hc][Wc], as pred_argmax returns them):
 * mask = ids != dust_bin, x = 8*ix + loc%8, y = 8*iy + loc/8, rows in torch.nonzero's raster order -- the second half of
 * dcx_pred_to_keypoints as its own entry, like in the reference.  d_codes: 4 bytes per cell of scratch; d_bad (nullable int32):
 * set to 1 when a label lies outside [0, 255].  A dust_bin outside [0, 255] equals no label: every cell fires.                                                                            */
int dcx_label_to_keypoints(const long long* d_loc, const long long* d_ids, int batch, int hc, int wc, int dust_bin, int kmax,
                           int32_t* d_counts, int32_t* d_rows, int32_t* d_codes, int32_t* d_bad, void* stream);

/* ---- patch table: compacts the per-frame rows of a batch into one patch list ----------
 * d_table int32 [B*kmax][4] = {frame, x, y, slot = frame*kmax + k}; d_total int32 [1] =
 * sum_b min(counts[b], kmax).  Device-side replacement for the host syncs at
 * model_utils.py:114 / inference.py:51.                                                  */
int dcx_build_patch_table(const int32_t* d_counts, const int32_t* d_rows, int batch, int kmax,
                          int32_t* d_table, int32_t* d_total, void* stream);

/* ---- extract_patches models/model_utils.py:19-36 ---------------------------------------
 * patch[p][i][j] = img[y-12+i][x-12+j], 0.0f outside the image (zero pad of the NORMALISED
 * image).  d_table rows {frame,x,y,slot}; patches beyond *d_total (if non-null) are skipped.
 * u8 variant normalises on the fly; f32 variant takes dense normalised images [B][H][W].   */
int dcx_extract_patches_u8(const uint8_t* d_frames, long frame_stride, int pitch, int height, int width,
                           const int32_t* d_table, const int32_t* d_total, int max_patches,
                           float* d_patches, void* stream);
int dcx_extract_patches_f32(const float* d_images, int height, int width,
                            const int32_t* d_table, const int32_t* d_total, int max_patches,
                            float* d_patches, void* stream);

/* ---- RefineNet: forward refinenet.py:49-83 + infer_patches :85-115 ---------------------
 * d_patches f32 [P][24][24] -> per patch flat argmax (first max) of the 64x64 heat-map:
 * d_corners int32 [P][2] = {col,row}; if d_table/d_xy given, d_xy[slot] = {(col-32)/8 + x,
 * (row-32)/8 + y} (float32, refinenet.py:114).  d_heat (nullable) f32 [P][64][64] receives
 * the raw heat-map.  Patches >= *d_total (if non-null) are skipped.                       */
int dcx_refiner_forward(const dcx_refiner* rf, const float* d_patches, int max_patches,
                        const int32_t* d_total, const int32_t* d_table,
                        void* d_ws, size_t ws_bytes,
                        int32_t* d_corners, float* d_xy, float* d_heat, void* stream);

/* ---- speedy_bargmax2d models/model_utils.py:39-43 --------------------------------------
 * d_x f32 [K][h][w] -> d_out int32 [K][2] = {col,row} of the first maximum.               */
int dcx_argmax2d(const float* d_x, int k, int h, int w, int32_t* d_out, void* stream);

/* ---- whole path for a batch: infer_image inference.py:32-70 without host syncs ----------
 * frames u8 -> detector -> per-cell arg-max + dust-bin rule -> ordered compaction -> RefineNet on every firing cell -> xy.
 *
 * Frames: DCX_PIX_GRAY8 (what inference.py:40 produces) or interleaved BGR (what infer_image is handed, inference.py:32; the
 * conversion of dcx_bgr2gray / dcx_bgr2gray_legacy14 happens in the first layer's load, the gray frame is never stored);
 * frame_stride and pitch in BYTES.
 *
 * Results -- the batch's CORNER POOL.  The reference refines every firing cell of a frame (inference.py:51-57, no cap), so
 * there is no per-frame capacity: frame b's corners occupy the pool slots [d_starts[b], d_starts[b] + d_counts[b]) in raster
 * order (torch.nonzero's, model_utils.py:112), whatever their number, as long as the BATCH fits: sum_b counts[b] <= pool.
 *   d_counts int32 [B]        firing cells of frame b (never truncated)
 *   d_starts int32 [B]        first pool slot of frame b; which frame comes first in the pool is unspecified
 *   d_rows   int32 [pool][4]  (x, y, id, cell = cy*Wc + cx)
 *   d_xy     f32   [pool][2]  RefineNet's corners_og (refinenet.py:114); required iff rf != NULL
 *   d_conf   f32   [pool][2]  nullable: soft-max probability of the winning loc class (65-way) and ids class (n_ids+1-way) of
 *                             the cell -- the "confidences" the docstring of pred_to_keypoints (model_utils.py:81-84) mentions;
 *                             the reference never thresholds on them and neither does this library
 * Slots >= pool are dropped: when sum(counts) > pool the caller re-runs with a larger pool (complete frames are still valid).
 * d_ws must hold dcx_pipeline_workspace_bytes().  rf may be null (detector + decode only).                                */
#define DCX_PIX_GRAY8          0
#define DCX_PIX_BGR8           1   /* OpenCV 4.x 15-bit constants, = dcx_bgr2gray          */
#define DCX_PIX_BGR8_LEGACY14  2   /* 14-bit constants,            = dcx_bgr2gray_legacy14 */
size_t dcx_pipeline_workspace_bytes(const dcx_detector* det, const dcx_refiner* rf,
                                    int batch, int height, int width, int pool);
int dcx_infer_batch(const dcx_detector* det, const dcx_refiner* rf,
                    const uint8_t* d_frames_u8, long frame_stride, int pitch, int pixel_format,
                    int batch, int height, int width, int dust_bin, int pool,
                    void* d_ws, size_t ws_bytes,
                    int32_t* d_counts, int32_t* d_starts, int32_t* d_rows, float* d_xy, float* d_conf, void* stream);

/* ---- stage-level entry point for kernel tests / roofline measurement -------------------
 * One 3x3 (or 1x1) convolution + bias [+ eval-BN + ReLU] [+ 2x2 max-pool] on C4 tensors
 * using the same MFMA kernel the networks use.  h_* are host arrays in PyTorch layout;
 * the call packs/uploads them (synchronously) and then enqueues the kernel on `stream`.
 * ups: input is read through a nearest x2 up-sampling.  d_in C4 [N][cin/4][Hin][Win][4],
 * d_out C4 [N][ceil(cout/4)][Hout][Wout][4].                                              */
int dcx_conv_layer(const float* d_in, int n, int cin, int hin, int win,
                   const float* h_weight_oihw, const float* h_bias,
                   const float* h_bn_gamma, const float* h_bn_beta,
                   const float* h_bn_mean, const float* h_bn_var,
                   int cout, int ksize, int pad, int ups, int pool, int relu,
                   float* d_out, void* stream);
/* NCHW <-> C4 layout converters (tests, API edges). c need not be a multiple of 4 (zero fill). */
int dcx_nchw_to_c4(const float* d_nchw, int n, int c, int h, int w, float* d_c4, void* stream);
int dcx_c4_to_nchw(const float* d_c4, int n, int c, int h, int w, float* d_nchw, void* stream);

/* ---- instrumentation: per-stage device time of the last dcx_infer_batch() ---------------
 * When enabled, the pipeline records hipEvents on its stream around each stage; after the
 * stream has been synchronised by the caller, dcx_last_timings() returns milliseconds:
 * [0] detector conv stack, [1] fused tail (1x1 heads + arg-max + ordered compaction into the corner pool), [2] RefineNet INCLUDING
 * the patch gather (its conv1a reads the 24x24 windows out of the frames since round 3), [3] total.  */
int dcx_set_timing(int enabled);
int dcx_get_timing(void);            /* 1 while per-stage timing is on (callers that capture hipGraphs must launch eagerly then) */
int dcx_last_timings(float* h_ms4);

/* name of the kernel instantiation the tile cost model selects for a launch shape (host only, no GPU needed);
 * epi: 0 = BN+ReLU, 1 = raw (1x1 heads), 2 = RefineNet head; "" when no instantiation fits */
const char* dcx_conv_pick_name(int n, int cin, int ho, int wo, int cout, int ks, int pool, int epi);
/* same for a layer whose input is read through a nearest x2 up-sampling (ho x wo = output size = 2 x the stored input) */
const char* dcx_conv_pick_name_ups(int n, int cin, int ho, int wo, int cout, int ks, int pool, int epi, int ups);

/* ---- kernel families / deterministic mode -------------------------------------------------
 * The kernel family of a layer (= the fp32 summation order of its outputs: direct implicit GEMM, 2-D Winograd F(2x2,3x3),
 * or phases x Winograd F(2x2,2x2) behind an up-sampling) depends on the LAYER only, never on the batch size or launch size:
 * a frame's results are bit-identical alone and inside any batch.  dcx_set_deterministic(1) (or
 * DCX_DETERMINISTIC=1 in the environment) forces the direct family everywhere -- every multiply-add of the layers as written,
 * the A/B reference of the Winograd families -- at about 0.5x the default throughput.  Process-global; callers that hold
 * captured hipGraphs must re-capture after switching (the Python layer does).                                          */
int dcx_set_deterministic(int enabled);
int dcx_get_deterministic(void);

/* ---- hand-off mode of the detector tail's fused compaction (csrc/dcx_tail.hip) --------------
 * 0 (default): the frame's last work item learns about the other work items' codes without fences -- write-through (sc1) stores
 * drained by s_waitcnt vmcnt(0), one relaxed agent-scope ticket, agent-scope (sc1) loads: gfx942 / gfx950 behaviour, soaked by
 * tests/test_gpu_parity.py::test_tail_handoff_is_never_stale.  1 (or DCX_TAIL_FENCE=1 in the environment): the release / acquire
 * pair the HIP memory model defines (__threadfence() on both sides, ~4x the kernel's time): the A/B for new ROCm drops.
 * Same results either way.  Process-global; captured hipGraphs keep the mode they were captured with.                  */
int dcx_set_tail_fence(int enabled);
int dcx_get_tail_fence(void);

/* ---- per-XCD item shares (speed only; the work items and their bits do not change) ---------
 * The convolution kernels are persistent grids whose workgroups walk one contiguous share of the item list per XCD (8 XCDs, each
 * with its own L2).  The XCDs of one chip do not run this load equally fast (3-6 % apart), and a launch ends with its slowest XCD:
 * dcx_calibrate_xcd measures them (`rounds` ~0.8 ms launches of the dominant kernel on a synthetic conv1b-sized layer; synchronous,
 * set-up code: GPU warm, outside timed regions and hipGraph capture) and sets the shares accordingly; dcx_set_xcd_weights sets
 * them by hand (8 relative speeds, NULL = equal; more than 25 % from equal is refused with DCX_E_ARG); dcx_get_xcd_weights returns
 * them (1.0 = an equal share).  Per device (the current one), process-global; hipGraphs keep the shares they were captured with. */
int dcx_calibrate_xcd(int rounds, float* w8_out, void* stream);
int dcx_set_xcd_weights(const float* w8);
int dcx_get_xcd_weights(float* w8);

/* ---- instrumentation: per-launch profile of the MFMA convolution kernel (roofline) ---------
 * While enabled, every launch of the convolution kernel is bracketed by two hipEvents on its
 * stream and recorded.  dcx_profile_enable(1) clears the record list.  After the caller has
 * synchronised the stream(s), dcx_profile_fetch() returns up to max_records records:
 * kernel_ids[i] (index into the instantiation table, see dcx_profile_kernel_name), n_images[i]
 * (images the grid covered), limited[i] (1 if a device-side patch count may have skipped some
 * of them), flops_per_image[i] = 2*cout*cin*ks*ks*Ho*Wo (algorithmic, un-padded), ms[i].      */
int dcx_profile_enable(int enabled);
int dcx_profile_enabled(void);       /* 1 while per-launch profiling is on */
int dcx_profile_count(void);
/* restrict recording to one kernel id (-1 = all): keeps the event overhead out of a timed region */
int dcx_profile_filter(int kernel_id);
int dcx_profile_sample(int every);   /* bracket only every `every`-th matching launch (1 = all): the two hipEvent records of a
                                        bracket keep the GPU idle for ~11 us, which a throughput measurement should not pay on
                                        every launch */
int dcx_profile_fetch(int* kernel_ids, int* n_images, int* limited, double* flops_per_image, float* ms,
                      int max_records);
/* kernel_id < 100: the convolution instantiation table; 100..103: the pipeline's other launches (detector conv1a, RefineNet
 * conv1a + patch gather, detector tail, refine finalize), recorded with flops_per_image = 0 while no filter is set */
const char* dcx_profile_kernel_name(int kernel_id);
/* effective shader clock (GHz) seen by workgroup 0 of each recorded launch: s_memtime ticks per
 * s_memrealtime (100 MHz) tick between its first and last instruction.  Same order as _fetch. */
int dcx_profile_clocks(float* ghz, int max_records);
/* raw 64 probe words of one recorded launch (kernel-tuning aid): [0..3] start/end {s_memtime, s_memrealtime},
 * then for the first 20 units of workgroup 0: s_memtime before the unit barrier, after it, after the MFMA loop */
int dcx_profile_probe_words(int record, unsigned long long* out64);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCHARUCO_AMD_H */
