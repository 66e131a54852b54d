// dcx_tail.hip -- fused detector tail: convPb (256 -> 65, 1x1) | convDb (256 -> n_ids+1, 1x1) -> per-cell arg-max ->
// dust-bin substitution -> packed code, in ONE kernel.
//
// Reference: net.py:74,77 (the two raw 1x1 heads) + model_utils.py:53-78 (pred_argmax).  The pipeline used to run them as
// two implicit-GEMM launches that wrote C4 logits to HBM plus an arg-max kernel that read them back (3 launches, 79 us at
// bs=32 with the 1x1 GEMMs at 0.15 of the MFMA peak, 57 us at bs=1); here the logits never leave the accumulators.
// dcModel.forward (which must RETURN logits) keeps the separate kernels; both paths produce bit-identical logits because
// this kernel keeps the direct kernel's summation order (dcx_conv_mfma.h):
//     acc = 0;  for chunk c (16 channels) / s in 0..1 / j in 0..3:
//         acc = fmaf(w[16c+8s+j], x[16c+8s+j], acc);  acc = fmaf(w[16c+8s+4+j], x[16c+8s+4+j], acc);      logit = acc + bias
// (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain: lane half h supplies k = h; a lane's float4 holds channels
//  16c + 8s + 4h .. +3 and MFMA j consumes component j).
//
// Work item = 32*NT consecutive cells of one frame.  Four waves: waves 0..2 own loc couts 0..95 (65 valid), wave 3 owns
// the ids couts (one or two 32-row tiles); every wave runs K = 256 channels = 128 MFMAs per (m-tile, n-tile).  Operands
// come straight from L2 (weights 130 KB, shared by every item; activations are read once per wave) with a software
// prefetch of PF k-steps.  Epilogue: bias, in-lane arg-max over the lane's 16 couts (ascending, strict '>': first
// maximum wins as in torch.argmax), lane-half and cross-wave combination through LDS, dust-bin rule, one int per cell.
#include "dcx_common.h"

typedef float dcx_t_f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kTailPF = 2;   // prefetch distance in k-steps (measured at bs=32 with 32-cell items: 1 -> 30.8 us, 2 -> 29.4, 3 -> 30.7, 4 -> 31.9, 8 -> 37.1)

template <int NT, int IDS_TILES>
__global__ __launch_bounds__(256) void dcx_tail_kernel(const float4* __restrict__ act, int act_cq_total, int cells,
                                                        const float4* __restrict__ w_loc, const float* __restrict__ b_loc,
                                                        const float4* __restrict__ w_ids, const float* __restrict__ b_ids,
                                                        int ids_cout_pad, int n_ids1, int tiles_per_frame, int dust_bin,
                                                        int32_t* __restrict__ codes, int32_t* __restrict__ loc_argmax,
                                                        int32_t* __restrict__ ids_argmax, int32_t* __restrict__ zero_word) {
    constexpr int NPIX = 32 * NT;
    if (zero_word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;     // ticket of the compaction kernel that follows
    __shared__ float red_v[4 + 1][NPIX];     // [job: loc tile 0..2, ids tile 0..1][pixel]
    __shared__ int red_i[4 + 1][NPIX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int item = blockIdx.x;
    const int b = item / tiles_per_frame;
    const int cell0 = (item - b * tiles_per_frame) * NPIX;
    const bool is_ids = wave == 3;
    const int jobs = is_ids ? IDS_TILES : 1;
    const float4* wq = is_ids ? w_ids : w_loc;
    const int cout_pad = is_ids ? ids_cout_pad : 128;
    const int cq_base = is_ids ? 64 : 0;                    // activation channel quads 64..127 = the convDa half
    const int n_valid = is_ids ? n_ids1 : 65;
    const float* bias = is_ids ? b_ids : b_loc;

    // per-lane operand addresses (float4 units)
    size_t a_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int pix = min(cell0 + nt * 32 + l31, cells - 1);          // cells past the frame re-read the last cell; not stored
        a_off[nt] = ((size_t)b * act_cq_total + cq_base + half) * (size_t)cells + pix;     // + (4c + 2s) * cells
    }
    const size_t k_stride = (size_t)2 * cells;              // one k-step = 8 channels = 2 channel quads (the lane's half picks one)

    for (int job = 0; job < jobs; ++job) {
        const int m0 = is_ids ? job * 32 : wave * 32;        // first cout of this wave's 32-row tile
        const size_t w_lane = (size_t)half * cout_pad + (m0 + l31);     // + (4c + 2s) * cout_pad
        const size_t w_stride = (size_t)2 * cout_pad;
        dcx_t_f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        float4 aq[32 + kTailPF], bq[32 + kTailPF][NT];
#pragma unroll
        for (int st = 0; st < kTailPF; ++st) {
            aq[st] = wq[w_lane + st * w_stride];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[st][nt] = act[a_off[nt] + st * k_stride];
        }
#pragma unroll
        for (int st = 0; st < 32; ++st) {                    // st = 2 * chunk + s
            __builtin_amdgcn_sched_barrier(0);
            if (st + kTailPF < 32) {
                aq[st + kTailPF] = wq[w_lane + (st + kTailPF) * w_stride];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[st + kTailPF][nt] = act[a_off[nt] + (st + kTailPF) * k_stride];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float4 av = aq[st];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a1 = j == 0 ? av.x : j == 1 ? av.y : j == 2 ? av.z : av.w;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 bv = bq[st][nt];
                    const float b1 = j == 0 ? bv.x : j == 1 ? bv.y : j == 2 ? bv.z : bv.w;
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[nt], 0, 0, 0);
                }
            }
        }
        // ---- bias + arg-max over this lane's 16 couts (ascending cout order: strict '>' keeps the first maximum)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float best = -INFINITY;
            int besti = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + 8 * (r >> 2) + 4 * half + (r & 3);
                if (co < n_valid) {
                    const float v = acc[nt][r] + bias[co];
                    if (v > best || besti == 0x7fffffff) { best = v; besti = co; }
                }
            }
            const float ov = __shfl_xor(best, 32);
            const int oi = __shfl_xor(besti, 32);
            if (oi != 0x7fffffff && (besti == 0x7fffffff || ov > best || (ov == best && oi < besti))) { best = ov; besti = oi; }
            if (half == 0) {
                const int slot = is_ids ? 3 + job : wave;
                red_v[slot][nt * 32 + l31] = best;
                red_i[slot][nt * 32 + l31] = besti;
            }
        }
    }
    __syncthreads();
    if (tid < NPIX) {
        const int cell = cell0 + tid;
        if (cell < cells) {
            float lv = red_v[0][tid];
            int la = red_i[0][tid];
#pragma unroll
            for (int t = 1; t < 3; ++t) {        // tiles hold ascending cout ranges: strict '>' keeps the first maximum
                const int ti = red_i[t][tid];
                if (ti != 0x7fffffff && red_v[t][tid] > lv) { lv = red_v[t][tid]; la = ti; }
            }
            float iv = red_v[3][tid];
            int ia = red_i[3][tid];
            if (IDS_TILES > 1) {
                const int ti = red_i[4][tid];
                if (ti != 0x7fffffff && red_v[4][tid] > iv) { iv = red_v[4][tid]; ia = ti; }
            }
            if (la == 64) ia = dust_bin;         // where(loc_argmax == 64, dust_bin, ids_argmax)  model_utils.py:76
            const size_t o = (size_t)b * cells + cell;
            codes[o] = la | (ia << 8);
            if (loc_argmax) loc_argmax[o] = la;
            if (ids_argmax) ids_argmax[o] = ia;
        }
    }
}

}  // namespace

// act: C4 [B][act_cq_total = 128][cells][4] (convPa|convDa output); w_*: packed [cin/4 = 64][cout_pad][4]; codes [B][cells].
int dcx_launch_tail(const float* act, int batch, int cells, const float* w_loc, const float* b_loc, const float* w_ids,
                    const float* b_ids, int ids_cout_pad, int n_ids1, int dust_bin, int32_t* codes, int32_t* loc_argmax,
                    int32_t* ids_argmax, int32_t* zero_word, hipStream_t s) {
    if (!act || !w_loc || !b_loc || !w_ids || !b_ids || !codes) return DCX_E_ARG;
    if (batch <= 0 || cells <= 0 || n_ids1 < 2 || n_ids1 > 64 || ids_cout_pad < n_ids1) return DCX_E_SHAPE;
    if (dust_bin < 0 || dust_bin > 255) return DCX_E_NIDS;
    const bool two = n_ids1 > 32;
    // 32-cell work items: the kernel is latency-bound (operands straight from L2 / HBM into registers), so more, shorter
    // workgroups per CU win at every size (64-cell items: 41 vs 32 us at bs=32, 139 vs 119 at bs=128, 469 vs 446 at bs=128 640x480)
    const int npix = 32;
    const int tiles = (cells + npix - 1) / npix;
    const long items = (long)batch * tiles;
    if (items > 0x7fffffffL) return DCX_E_SHAPE;
    const float4* a4 = reinterpret_cast<const float4*>(act);
    const float4* wl = reinterpret_cast<const float4*>(w_loc);
    const float4* wi = reinterpret_cast<const float4*>(w_ids);
#define DCX_TAIL_LAUNCH(NT, IT)                                                                                          \
    hipLaunchKernelGGL((dcx_tail_kernel<NT, IT>), dim3((unsigned)items), dim3(256), 0, s, a4, 128, cells, wl, b_loc, wi, b_ids, \
                       ids_cout_pad, n_ids1, tiles, dust_bin, codes, loc_argmax, ids_argmax, zero_word)
    if (two) DCX_TAIL_LAUNCH(1, 2); else DCX_TAIL_LAUNCH(1, 1);
#undef DCX_TAIL_LAUNCH
    return (int)hipGetLastError();
}
