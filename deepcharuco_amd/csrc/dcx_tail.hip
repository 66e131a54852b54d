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
//
// Round 5: the ordered compaction is part of this kernel.  Every work item draws a ticket from its FRAME's counter after its
// codes are visible device-wide; the workgroup that draws a frame's last ticket compacts that frame's firing cells in raster
// order (torch.nonzero's order, model_utils.py:112) into the BATCH's corner pool: one atomicAdd on the pool cursor reserves
// `count` consecutive slots, so there is no per-frame capacity -- a frame may fire any number of cells as long as the batch
// fits the pool (the reference refines EVERY firing cell, inference.py:51-57).  rows[p] = (x, y, id, cell), table[p] =
// (frame, x, y, p) for RefineNet's patch gather; counts[b] / starts[b] say where frame b's corners are.  The placement of the
// frames in the pool depends on which frame finishes first (the host unpacks by starts[]); a frame's own rows never do.
// Optional (CONF): the soft-max probability of the winning loc / ids class of every cell (log-sum-exp over the 65 / n_ids+1
// logits while they are in the accumulators), delivered per corner -- the "confidences" model_utils.py:81-84 mentions.
#include "dcx_common.h"

int dcx_device_cu_count();   // dcx_conv_mfma.hip

#include <stdlib.h>
#include <string.h>

typedef float dcx_t_f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kTailPF = 2;   // prefetch distance in k-steps (measured at bs=32 with 32-cell items: 1 -> 30.8 us, 2 -> 29.4, 3 -> 30.7, 4 -> 31.9, 8 -> 37.1)
#ifndef DCX_TAIL_PF_SMALL
#define DCX_TAIL_PF_SMALL 8
#endif
constexpr int kTailPFSmall = DCX_TAIL_PF_SMALL;   // ... for launches of fewer work items than CUs (one frame): nothing else hides the latency of
                                                  // operands that come from the MALL, and a k-step is 256 matrix cycles

// Ordered compaction of ONE frame's packed codes into the batch's corner pool; all 256 threads of the workgroup call it.
// Super-chunks of 2,048 cells: eight coalesced loads per thread in flight at once, wave ballots, per-(chunk, wave) counts through
// LDS -- one L2 round trip and two barriers for a 320x240 frame (1,200 cells).
// The codes (and per-cell confidences) were stored write-through by work items on any XCD: they are read with agent-scope (sc1) loads.
__device__ __forceinline__ int dcx_ld_agent(const int32_t* p) {
    return __hip_atomic_load(const_cast<int32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool CONF>
__device__ __forceinline__ void dcx_tail_compact_frame(const int32_t* codes, int cells, int dust_bin, int b, const DcxPoolOut& po,
                                                       int* s_cnt /*[32]*/, int* s_bcast) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsuper = (cells + 2047) / 2048;
    const int32_t* fc = codes + (size_t)b * cells;
    const int idle = dust_bin << 8;                  // a code that does not fire
    int total = 0;
    if (nsuper > 1) {                                // big frames: the frame's count first (the pool slots are reserved before any row is written)
        int c = 0;
        for (int i = tid; i < cells; i += 256) c += ((dcx_ld_agent(fc + i) >> 8) != dust_bin) ? 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0) s_cnt[wave] = c;
        __syncthreads();
        total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    int run = 0, start = 0;
    for (int sc = 0; sc < nsuper; ++sc) {
        int code[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cell = sc * 2048 + c * 256 + tid;
            code[c] = cell < cells ? dcx_ld_agent(fc + cell) : idle;
        }
        unsigned long long m[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            m[c] = __ballot((code[c] >> 8) != dust_bin);
            if (lane == 0) s_cnt[c * 4 + wave] = __popcll(m[c]);
        }
        __syncthreads();
        int base[8], acc = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int w0 = s_cnt[c * 4], w1 = s_cnt[c * 4 + 1], w2 = s_cnt[c * 4 + 2], w3 = s_cnt[c * 4 + 3];
            base[c] = acc + (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
            acc += w0 + w1 + w2 + w3;
        }
        if (sc == 0) {
            if (nsuper == 1) total = acc;
            if (tid == 0) {
                const int st = atomicAdd(po.cursor, total);       // reserves [st, st + total) of the pool for this frame
                po.counts[b] = total;
                po.starts[b] = st;
                *s_bcast = st;
            }
            __syncthreads();
            start = *s_bcast;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if ((code[c] >> 8) != dust_bin) {
                const int pos = start + run + base[c] + __popcll(m[c] & ((1ull << lane) - 1ull));
                if (pos < po.pool) {
                    const int cell = sc * 2048 + c * 256 + tid;
                    const int la = code[c] & 255, ia = code[c] >> 8;
                    const int cy = cell / po.wc, cx = cell - cy * po.wc;
                    const int x = 8 * cx + (la & 7);              // xs = 8*ix + loc % 8    (model_utils.py:121)
                    const int y = 8 * cy + (la >> 3);             // ys = 8*iy + loc // 8   (model_utils.py:122)
                    reinterpret_cast<int4*>(po.rows)[pos] = make_int4(x, y, ia, cell);
                    if (po.table) reinterpret_cast<int4*>(po.table)[pos] = make_int4(b, x, y, pos);
                    if (CONF && po.conf) {
                        const unsigned long long cv = __hip_atomic_load(reinterpret_cast<unsigned long long*>(po.conf_cells) + (size_t)b * cells + cell,
                                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        reinterpret_cast<unsigned long long*>(po.conf)[pos] = cv;
                    }
                }
            }
        }
        run += acc;
        __syncthreads();                             // s_cnt is rewritten by the next super-chunk
    }
}

template <int NT, int IDS_TILES, bool CONF, int PF = kTailPF>
__global__ __launch_bounds__(256) void dcx_tail_kernel(const float4* __restrict__ act, int act_cq_total, int cells,
                                                        const float4* __restrict__ w_loc, const float* __restrict__ b_loc,
                                                        const float4* __restrict__ w_ids, const float* __restrict__ b_ids,
                                                        int ids_cout_pad, int n_ids1, int tiles_per_frame, int dust_bin,
                                                        int32_t* codes, int32_t* __restrict__ loc_argmax,
                                                        int32_t* __restrict__ ids_argmax, DcxPoolOut po, int fence) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "dcx_tail_kernel's fence-free hand-off counts on gfx942 / gfx950 behaviour (stores are tracked by vmcnt; sc1 stores / loads are write-through / L2-bypassing at agent scope): build with DCX_TAIL_FENCE semantics reviewed for any other ISA"
#endif
    constexpr int NPIX = 32 * NT;
    __shared__ float red_v[4 + 1][NPIX];     // [job: loc tile 0..2, ids tile 0..1][pixel]
    __shared__ int red_i[4 + 1][NPIX];
    __shared__ float red_s[CONF ? 4 + 1 : 1][NPIX];   // CONF: sum of exp(logit - red_v) over the job's valid couts
    __shared__ int s_cnt[32];
    __shared__ int s_bcast, s_last;
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
        float4 aq[32 + PF], bq[32 + PF][NT];
#pragma unroll
        for (int st = 0; st < PF; ++st) {
            aq[st] = wq[w_lane + st * w_stride];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[st][nt] = act[a_off[nt] + st * k_stride];
        }
#pragma unroll
        for (int st = 0; st < 32; ++st) {                    // st = 2 * chunk + s
            __builtin_amdgcn_sched_barrier(0);
            if (st + PF < 32) {
                aq[st + PF] = wq[w_lane + (st + PF) * w_stride];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bq[st + PF][nt] = act[a_off[nt] + (st + PF) * k_stride];
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
            float lv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + 8 * (r >> 2) + 4 * half + (r & 3);
                lv[r] = -INFINITY;
                if (co < n_valid) {
                    const float v = acc[nt][r] + bias[co];
                    lv[r] = v;
                    if (v > best || besti == 0x7fffffff) { best = v; besti = co; }
                }
            }
            float ssum = 0.f;
            if (CONF && besti != 0x7fffffff) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ssum += lv[r] == -INFINITY ? 0.f : expf(lv[r] - best);      // ascending cout order
            }
            const float ov = __shfl_xor(best, 32);
            const int oi = __shfl_xor(besti, 32);
            const float os = CONF ? __shfl_xor(ssum, 32) : 0.f;
            const float mine = best;
            if (oi != 0x7fffffff && (besti == 0x7fffffff || ov > best || (ov == best && oi < besti))) { best = ov; besti = oi; }
            if (half == 0) {
                const int slot = is_ids ? 3 + job : wave;
                red_v[slot][nt * 32 + l31] = best;
                red_i[slot][nt * 32 + l31] = besti;
                if (CONF) {      // both halves' sums re-based on the common maximum (lane half 0's term first)
                    const float a = ssum == 0.f ? 0.f : ssum * expf(mine - best);
                    const float c = os == 0.f ? 0.f : os * expf(ov - best);
                    red_s[slot][nt * 32 + l31] = a + c;
                }
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
            const size_t o = (size_t)b * cells + cell;
            if (CONF) {
                // softmax(logits)[arg-max] = 1 / sum_c exp(logit_c - max): the tiles' partial sums re-based on the overall maximum
                float sl = 0.f, si = 0.f;
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (red_i[t][tid] != 0x7fffffff) sl += red_s[t][tid] * expf(red_v[t][tid] - lv);
                si = red_s[3][tid] * expf(red_v[3][tid] - iv);
                if (IDS_TILES > 1 && red_i[4][tid] != 0x7fffffff) si += red_s[4][tid] * expf(red_v[4][tid] - iv);
                const float2 cf2 = make_float2(1.0f / sl, 1.0f / si);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(po.conf_cells) + o, *reinterpret_cast<const unsigned long long*>(&cf2),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (la == 64) ia = dust_bin;         // where(loc_argmax == 64, dust_bin, ids_argmax)  model_utils.py:76
            // write-through (sc1) store: the frame's last work item -- possibly on another XCD, whose L2 is not coherent with this
            // one -- reads the code without anybody paying for a release fence (MI355X_MICROARCH.md, inter-workgroup visibility)
            __hip_atomic_store(codes + o, la | (ia << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (loc_argmax) loc_argmax[o] = la;
            if (ids_argmax) ids_argmax[o] = ia;
        }
    }
    if (po.tickets == nullptr) return;
    // ---- the frame's last work item compacts the frame.  Hand-off without fences: the codes were stored write-through, every
    //      wave drains its stores, then ONE lane draws the frame's ticket (relaxed agent-scope atomic); the workgroup that draws the
    //      last one reads the frame's codes with agent-scope (sc1) loads, which bypass its L1.  (Round 5's first version had every
    //      thread of every work item run __threadfence() on both sides: 30 -> 136 us at bs=32.)
    //      INVARIANT the fast path rests on: `codes` and `po.conf_cells` are ONLY ever written with agent-scope atomic stores and read
    //      with agent-scope atomic loads (dcx_ld_agent) -- a plain store / load added to either side would go through a non-coherent
    //      L2 / L1 and break the hand-off silently.  `fence` != 0 (DCX_TAIL_FENCE=1 / dcx_set_tail_fence) selects the textbook
    //      release / acquire pair the HIP memory model defines instead: the A/B for new ROCm drops and other ISAs (~4x slower).
    if (fence) __threadfence();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&po.tickets[b], 1, fence ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tiles_per_frame - 1;
    __syncthreads();
    if (!s_last) return;
    if (fence) __threadfence();
    dcx_tail_compact_frame<CONF>(codes, cells, dust_bin, b, po, s_cnt, &s_bcast);
}

}  // namespace

// Hand-off mode of the fused compaction: 0 (default) = fence-free (write-through stores + one relaxed ticket + agent-scope loads),
// 1 = __threadfence() on both sides (release / acquire as the HIP memory model defines it).  DCX_TAIL_FENCE=1 or dcx_set_tail_fence.
static int g_tail_fence = -1;
static int dcx_tail_fence_mode() {
    if (g_tail_fence < 0) { const char* e = getenv("DCX_TAIL_FENCE"); g_tail_fence = (e && atoi(e)) ? 1 : 0; }
    return g_tail_fence;
}
extern "C" int dcx_set_tail_fence(int enabled) { g_tail_fence = enabled ? 1 : 0; return 0; }
extern "C" int dcx_get_tail_fence(void) { return dcx_tail_fence_mode(); }

// act: C4 [B][act_cq_total = 128][cells][4] (convPa|convDa output); w_*: packed [cin/4 = 64][cout_pad][4]; codes [B][cells].
// po (nullable): fused ordered compaction into the batch's corner pool; po->tickets[0..batch) and *po->cursor must be 0 at entry.
int dcx_launch_tail(const float* act, int batch, int cells, const float* w_loc, const float* b_loc, const float* w_ids,
                    const float* b_ids, int ids_cout_pad, int n_ids1, int dust_bin, int32_t* codes, int32_t* loc_argmax,
                    int32_t* ids_argmax, const DcxPoolOut* po_in, hipStream_t s) {
    if (!act || !w_loc || !b_loc || !w_ids || !b_ids || !codes) return DCX_E_ARG;
    if (batch <= 0 || cells <= 0 || n_ids1 < 2 || n_ids1 > 64 || ids_cout_pad < n_ids1) return DCX_E_SHAPE;
    if (dust_bin < 0 || dust_bin > 255) return DCX_E_NIDS;
    DcxPoolOut po;
    memset(&po, 0, sizeof(po));
    if (po_in != nullptr) {
        po = *po_in;
        if (!po.tickets || !po.cursor || !po.counts || !po.starts || !po.rows) return DCX_E_ARG;
        if (po.conf != nullptr && po.conf_cells == nullptr) return DCX_E_ARG;
        if (po.wc <= 0 || cells % po.wc != 0 || po.pool <= 0) return DCX_E_SHAPE;
    }
    const bool conf = po.conf != nullptr;
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
#define DCX_TAIL_LAUNCH_PF(NT, IT, CF, PF)                                                                               \
    hipLaunchKernelGGL((dcx_tail_kernel<NT, IT, CF, PF>), dim3((unsigned)items), dim3(256), 0, s, a4, 128, cells, wl, b_loc, wi, b_ids, \
                       ids_cout_pad, n_ids1, tiles, dust_bin, codes, loc_argmax, ids_argmax, po, dcx_tail_fence_mode())
    const bool small = items < dcx_device_cu_count();       // one frame: 38 work items
#define DCX_TAIL_LAUNCH(NT, IT, CF) do { if (small) DCX_TAIL_LAUNCH_PF(NT, IT, CF, kTailPFSmall); else DCX_TAIL_LAUNCH_PF(NT, IT, CF, kTailPF); } while (0)
    if (two) { if (conf) DCX_TAIL_LAUNCH(1, 2, true); else DCX_TAIL_LAUNCH(1, 2, false); }
    else     { if (conf) DCX_TAIL_LAUNCH(1, 1, true); else DCX_TAIL_LAUNCH(1, 1, false); }
#undef DCX_TAIL_LAUNCH
#undef DCX_TAIL_LAUNCH_PF
    return (int)hipGetLastError();
}
