// dcx_common.h -- internal declarations shared by the HIP translation units.
// gfx950 (MI355X, CDNA4) only: wave64, v_mfma_f32_32x32x2_f32, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/deepcharuco_amd.h"

#define DCX_CHECK_HIP(expr)                              \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return (int)_e;            \
    } while (0)

// ---------------------------------------------------------------------------------------
// MFMA convolution (dcx_conv_mfma.hip)

enum DcxEpilogue {
    DCX_EPI_BNRELU = 0,  // y = relu(fma(acc + bias, alpha, beta))
    DCX_EPI_RAW = 1,     // y = acc + bias                     (1x1 heads, "NO activ" net.py:74,77)
    DCX_EPI_HEAT = 2,    // BN+ReLU, then 1x1 conv to ONE channel + per-tile arg-max (RefineNet head)
};

struct DcxConvArgs {
    const float* in;       // C4 [N][in_cq_total][Hin][Win][4]
    const float* w;        // packed [ks*ks][cin/4][cout_pad][4]   (4 = cin % 4)
    const float* w_wino2;  // nullable; 3x3 + BN layers: F(2x2,3x3)-transformed weights [xi*4 + nu][cin/4][cout_pad][4] (dcx_conv_wino2h.h)
    const float* w_ups2w;  // nullable; 3x3 + BN layers read through a nearest x2 up-sampling: the four phases' pre-summed 2x2 kernels,
                           // F(2x2,2x2)-transformed: [phase][xi*3 + nu][cin/4][cout_pad][4] (dcx_conv_wino2p.h)
    const float* bias;     // [cout_pad]
    const float* alpha;    // [cout_pad]  gamma / sqrt(var + eps)
    const float* beta;     // [cout_pad]  bn_beta - mean * alpha
    float* out;            // C4 [N][out_cq_total][Hs][Ws][4]; Hs = Ho (or Ho/2 when pooled)
    const int* n_limit;    // optional device int: images n >= *n_limit are skipped
    unsigned long long* clk_probe;  // optional [4]: workgroup 0 stores {s_memtime, s_memrealtime} at start and end
    int probe_u0;                   // first unit of workgroup 0's 20-unit probe window (tuning aid, DCX_PROBE_U0)
    // DCX_EPI_HEAT only
    const float* head_w;   // [cout_pad] weights of the 1x1 conv to one channel
    float head_b;
    float* heat;           // nullable [N][Ho][Wo] raw heat-map
    float* part_val;       // [N][tiles] per-tile maximum
    int* part_idx;         // [N][tiles] flat index (y*Wo+x) of the first maximum in the tile
    int n;
    int n_hint;            // > 0 with n_limit: the number of images the launch is EXPECTED to process (n is only the capacity);
                           // used by the tile cost model, never by the kernel
    int in_cq_total, in_cq_off;
    int cin;               // multiple of 32
    int hin, win;          // physical input size
    int ups;               // 1: input is read through a nearest x2 up-sampling
    int pad;               // 0 or 1 (ks == 3), 0 (ks == 1)
    int ho, wo;            // convolution output size (before pooling)
    int out_cq_total, out_cq_off;
    int cout_pad;          // multiple of the kernel's COUT_TILE
    int cout_quads;        // valid output channel quads = ceil(cout / 4)
    int cout_real;         // un-padded output channels (profiling only)
    int tiles_x, tiles_y;
    int xcd_walk;          // set by the launcher: XCD-aware item walk
    const int* xcd_cum;    // set by the launcher: device table [9]; XCD x walks items [total * cum[x] >> 16, total * cum[x + 1] >> 16),
                           // cum[0] = 0, cum[8] = 65536.  Equal eighths by default; dcx_calibrate_xcd / dcx_set_xcd_weights re-weight the
                           // ranges by measured per-XCD speed (the XCDs of one chip differ by 3-6 % under this load).  Same items, same
                           // bits.  A table is immutable once a launch may read it (a new weight set gets a new table), so all
                           // workgroups of a launch -- and every replay of a captured graph -- see ONE partition.  (Through a pointer
                           // rather than nine kernel arguments: the unpooled kernels sit at the SGPR limit.)
    unsigned long long* xcd_stat;   // nullable [17]: calibration launches only -- [x] += s_memrealtime at the exit of every workgroup
                                    // of XCD x, [8] = start time of workgroup 0, [9 + x] += 1
    int ct_outer;          // set by the launcher (2-D Winograd kernels): work items ordered cout tile OUTERMOST (image inside), so that
                           // with the XCD-aware walk an XCD works on one or two cout tiles at a time -- for layers whose transformed
                           // weights (16 x cin x cout x 4 B: 4.2 MB for the fused 512-cout heads) do not fit an XCD's 4 MB L2 next to
                           // the activations.  Same items, same bits; only the order changes.
};

// Picks a tile configuration for (ho, wo, cout_pad, pool, epi, ks) and launches.
// Returns 0 / DCX_E_SHAPE / hipError_t.
int dcx_launch_conv_mfma(DcxConvArgs a, int ks, int pool, int epi, hipStream_t stream);
// Rounds cout up to what dcx_launch_conv_mfma needs for cout_pad.
int dcx_conv_cout_pad(int cout);
// Number of partial arg-max slots per image the DCX_EPI_HEAT launch will write for an (ho, wo) heat-map (size of part_*);
// ups: the layer reads its input through a x2 up-sampling (the phase variant then writes 4 slots per low-resolution tile).
int dcx_conv_heat_tiles(int ho, int wo, int ups);

// per-launch profiling of the pipeline's launches outside the convolution families (dcx_conv_mfma.hip keeps the record list;
// bench.py's step_breakdown): kernel ids >= 100, bracketed by dcx_api.hip
enum { DCX_PROF_CONV1A = 100, DCX_PROF_PATCHES = 101, DCX_PROF_TAIL = 102, DCX_PROF_FINALIZE = 103 };
int dcx_prof_begin(int kernel_id, int n, hipStream_t s);   // -> token, -1 when not recording
int dcx_prof_end(int token, hipStream_t s);

// ---------------------------------------------------------------------------------------
// everything else (dcx_misc.hip)

// pix: DCX_PIX_GRAY8 / DCX_PIX_BGR8 / DCX_PIX_BGR8_LEGACY14 (BGR -> gray in the load, the integer formula of dcx_bgr2gray);
// frame_stride / pitch in BYTES.  zero_words / n_zero (nullable): int32 words workgroup (0, 0) clears -- the pipeline's pool
// cursor and frame tickets, which the tail kernel nine launches later expects to be 0.
int dcx_launch_conv1_u8(const uint8_t* frames, long frame_stride, int pitch, int pix, int n, int h, int w, int pad,
                        const float* w9x64, const float* bias, const float* alpha, const float* beta,
                        float* out_c4, const int* n_limit, int32_t* zero_words, int n_zero, hipStream_t s);
int dcx_launch_conv1_f32(const float* images, long image_stride, int pitch, int n, int h, int w, int pad,
                         const float* w9x64, const float* bias, const float* alpha, const float* beta,
                         float* out_c4, const int* n_limit, hipStream_t s);

// strided logits accessor: value(b, c, cell) = p[b*sb + (c>>2)*sq + cell*sp + (c&3)*sc]
struct DcxLogitView {
    const float* p;
    long sb, sq, sp, sc;
};
int dcx_launch_decode(DcxLogitView loc, DcxLogitView ids, int batch, int n_loc, int n_ids1, int hc, int wc,
                      int dust_bin, int kmax, int32_t* counts, int32_t* rows,
                      int32_t* loc_argmax, int32_t* ids_argmax, int32_t* codes_scratch, hipStream_t s);
// The batch's corner pool, filled by the fused detector tail (dcx_tail.hip): frame b's firing cells occupy the pool slots
// [starts[b], starts[b] + counts[b]) in raster order; slots >= pool are dropped (the caller sees sum(counts) > pool).
struct DcxPoolOut {
    int32_t* tickets;      // [B]  per-frame arrival counters of the tail's work items, 0 at entry
    int32_t* cursor;       // [1]  0 at entry; ends as the number of firing cells of the batch (may exceed pool) = RefineNet's n_limit
    int32_t* counts;       // [B]  firing cells of frame b (never truncated)
    int32_t* starts;       // [B]  first pool slot of frame b
    int32_t* rows;         // [pool][4] = (x, y, id, cell)
    int32_t* table;        // nullable [pool][4] = (frame, x, y, slot): RefineNet's patch table
    float* conf;           // nullable [pool][2] = soft-max probability of the winning (loc, ids) class
    float* conf_cells;     // [B][cells][2] scratch, needed when conf is given
    int wc, pool;
};
// fused detector tail (dcx_tail.hip): 1x1 heads + per-cell arg-max + dust-bin rule -> packed codes (loc | id << 8) and, with
// `po`, the ordered compaction of every frame into the batch's corner pool (done by the frame's last work item)
int dcx_launch_tail(const float* act_c4_512, int batch, int cells, const float* w_loc, const float* b_loc,
                    const float* w_ids, const float* b_ids, int ids_cout_pad, int n_ids1, int dust_bin,
                    int32_t* codes, int32_t* loc_argmax, int32_t* ids_argmax, const DcxPoolOut* po, hipStream_t s);
// ordered compaction of packed codes into per-frame rows [B][kmax][4] (dcx_misc.hip; stand-alone decode entry points)
int dcx_launch_compact(const int32_t* codes, int batch, int hc, int wc, int dust_bin, int kmax, int32_t* counts,
                       int32_t* rows, hipStream_t s);
// RefineNet conv1a (pad 0, 24x24 -> 22x22) reading its patches straight out of the u8 frames through the patch table
// (extract_patches + pre_bgr_image + conv1a in one kernel: the patch tensor is never materialised)
int dcx_launch_conv1_patches_u8(const uint8_t* frames, long frame_stride, int pitch, int pix, int h, int w, const int32_t* table,
                                const int* total, int max_patches, int n_hint, const float* w9x64, const float* bias,
                                const float* alpha, const float* beta, float* out_c4, hipStream_t s);
int dcx_launch_refine_finalize(const float* part_val, const int* part_idx, int tiles, int wo,
                               int max_patches, const int* total, const int32_t* table,
                               int32_t* corners, float* xy, hipStream_t s);
