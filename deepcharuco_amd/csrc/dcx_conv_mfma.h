// dcx_conv_mfma.h -- 3x3 / 1x1 convolution as implicit GEMM on the gfx950 fp32 matrix cores.
//
// Replaces every Conv2d(+BatchNorm2d+ReLU)(+MaxPool2d / UpsamplingNearest2d) of
//   dcModel.forward   /root/reference/src/models/net.py:60-77        and
//   RefineNet.forward /root/reference/src/models/refinenet.py:56-81
// except the two Cin=1 first layers (dcx_misc.hip).
//
// GEMM view (per image n):   D[cout][pixel] = sum_{tap, cin} W[tap][cin][cout] * X[pixel + tap][cin]
//   A operand = weights  (M = cout),  B operand = activations (N = pixel),  K = 9 * cin.
//   v_mfma_f32_32x32x2_f32: A lane l holds A[i = l&31][k = l>>5], B lane l holds B[k = l>>5][j = l&31],
//   D lane l holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31], r = 0..15.
//   => a lane owns ONE pixel and, per accumulator, four groups of 4 consecutive couts: exactly one
//      float4 of the C4 layout [N][C/4][H][W][4], so epilogue stores are 16 B per lane and 512 B
//      contiguous per half-wave, and the 2x2 max-pool is two v_max_f32_dpp quad_perm per value.
//
// Both operands are fetched as float4 = 4 consecutive cin of one cout / one pixel:
//   lane (half = l>>5) reads channel quad (2s + half) of an 8-channel group s, register j of the
//   float4 feeds MFMA j, so one ds_read_b128 + one buffer_load_dwordx4 per tile feed 4 MFMAs
//   (256 cycles).  The resulting summation order is specified (and restated bit-exactly by
//   oracle/conv_exact.c):
//     acc = 0;  for chunk c (16 cin) / tap (dy-major) / s in 0..1 / j in 0..3:
//                  acc = fmaf(w[8s+j],   x[8s+j],   acc);     (k = 0 half)
//                  acc = fmaf(w[8s+4+j], x[8s+4+j], acc);     (k = 1 half)
//   i.e. an exact sequential fp32 fmaf chain (MFMA f32 numerics, cdna_hip_programming.md section 3).
//
// Activations: the (TH+2)x(TW+2) input halo tile of a 16-channel chunk is staged in LDS as
//   sB[buf][cq][halo_pixel] float4 (double buffered) -- consecutive lanes read consecutive 16-B slots
//   (conflict-free ds_read_b128 without padding or swizzle).  Zero padding, the valid-conv offset and the
//   nearest x2 up-sampling of RefineNet are all resolved while staging (buffer loads: per-piece constant
//   voffset relative to the tile origin, hardware out-of-range -> 0.0f), so the MFMA loop is identical for
//   every layer.  Weights ([tap][cin/4][cout][4], <= 1.2 MB, L2 resident) are streamed from L2 into
//   registers two k-steps ahead of use (buffer loads: per-lane constant voffset, uniform soffset).
//
// Measured on gfx950 (tools/ubench/mfma_fill.hip): every VALU instruction issued between these MFMAs costs
//   its full ~8 cycles -- the f32 MFMA runs on the vector FMA datapath -- while memory, LDS, scalar
//   instructions and s_nop are free.  The k-loop therefore contains (almost) no VALU instruction, and the
//   epilogue is written for minimum VALU count (packed FMA, fused DPP max).  See DESIGN.md section 3.1.
#pragma once
#include "dcx_common.h"

typedef float dcx_f32x16 __attribute__((ext_vector_type(16)));
typedef float dcx_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int dcx_u32x4 __attribute__((ext_vector_type(4)));

#define DCX_CCH 16  // input channels per LDS chunk ("unit" of the software pipeline)

template <int WM_, int WN_, int MT_, int NT_, int TH_, int TW_, int KS_, bool POOL_, int EPI_>
struct DcxConvCfg {
    static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, TH = TH_, TW = TW_, KS = KS_;
    static constexpr bool POOL = POOL_;
    static constexpr int EPI = EPI_;
    static constexpr int NTHREADS = WM * WN * 64;
    static constexpr int COUT_TILE = WM * MT * 32;
    static constexpr int CAP = WN * NT * 32;          // pixels a workgroup can hold
    static constexpr int TILE_PIX = TH * TW;
    static constexpr int HH = TH + KS - 1, HW = TW + KS - 1;
    static constexpr int HALO = HH * HW;
    static constexpr int CQC = DCX_CCH / 4;           // channel quads per chunk
    static constexpr int LDS_FLOAT4 = CQC * HALO;     // one staging buffer
    static constexpr size_t LDS_BYTES = (size_t)2 * LDS_FLOAT4 * 16;   // double buffered
    static constexpr int STEPS = KS * KS * (DCX_CCH / 8);              // MFMA k-steps (8 channels of one tap) per unit
    static constexpr int ITER = (LDS_FLOAT4 + NTHREADS - 1) / NTHREADS;  // float4 pieces per thread per unit
    static constexpr int DA = 2;                                          // weights are fetched DA k-steps ahead
    static constexpr int DP = (STEPS - 1) < 3 ? (STEPS - 1) : 3;           // a staging piece is written DP steps after its load
    static constexpr int LOAD_STEPS = STEPS - DP;                          // steps that issue staging loads
    static constexpr int PPS = (ITER + LOAD_STEPS - 1) / LOAD_STEPS;       // pieces fetched per k-step
    static constexpr int NPAIR = 2 * MT * NT;                              // MFMA pairs per k-step
    // workgroups per CU the launch is sized for (registers: <= 168 VGPR+AGPR for 3 waves/SIMD)
    static constexpr int OCC_LDS = (LDS_BYTES * 3 <= 160 * 1024) ? 3 : ((LDS_BYTES * 2 <= 160 * 1024) ? 2 : 1);
    // 2 workgroups/CU measured best for 64-register accumulators (256 registers per lane); the big wave tiles
    // (128 accumulator registers) run one workgroup per CU with the whole 512-register file
    static constexpr int OCC = (MT * NT * 16 > 64) ? 1
                             : (MT * NT * 16 <= 16) ? (OCC_LDS > 3 ? 3 : OCC_LDS)   // small tiles: more workgroups hide the barriers
                             : (OCC_LDS > 2 ? 2 : OCC_LDS);
    static_assert(TILE_PIX <= CAP, "tile does not fit the wave layout");
    static_assert(!POOL || (TH % 2 == 0 && TW % 2 == 0), "pooled tiles must be even");
    static_assert(EPI != DCX_EPI_HEAT || (WM == 1 && !POOL), "heat epilogue needs all couts in one wave row");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
    static_assert(STEPS >= 2, "pipeline needs two k-steps per unit");
    static_assert(KS == 3 || KS == 1, "3x3 or 1x1");
};

__device__ __forceinline__ float4 dcx_f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// One v_max_f32.  fmaxf() makes hipcc emit an extra canonicalising v_max per operand (sNaN quieting);
// activations here are finite, and the epilogue runs with the matrix pipe idle, so every VALU counts.
__device__ __forceinline__ float dcx_vmax(float x, float y) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
typedef float dcx_f32x2 __attribute__((ext_vector_type(2)));
// one v_pk_add_f32 (hipcc scalarises float2 +/- into two v_add_f32; every VALU instruction in a k-loop costs matrix time)
__device__ __forceinline__ dcx_f32x2 dcx_pk_add(dcx_f32x2 x, dcx_f32x2 y) {
    dcx_f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ dcx_f32x2 dcx_pk_sub(dcx_f32x2 x, dcx_f32x2 y) {
    dcx_f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// y = x * al + be on 4 channels as two v_pk_fma_f32 (packed fp32: 2 results per VALU instruction)
__device__ __forceinline__ float4 dcx_fma4(float4 x, float4 al, float4 be) {
    const dcx_f32x2 lo = __builtin_elementwise_fma(dcx_f32x2{x.x, x.y}, dcx_f32x2{al.x, al.y}, dcx_f32x2{be.x, be.y});
    const dcx_f32x2 hi = __builtin_elementwise_fma(dcx_f32x2{x.z, x.w}, dcx_f32x2{al.z, al.w}, dcx_f32x2{be.z, be.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
// ReLU + 2x2 max-pool of one float4 as 12 VALU instructions (v_max_f32 with a DPP source = exchange + max in one)
// and NO s_nop: the "VALU write -> DPP read" hazard needs 2 wait states, which hipcc does not pad inside asm.  All 12
// statements are volatile (their order is kept) and each DPP reads a register written >= 3 instructions earlier
// (x, y, z, w round-robin), so the distance holds by construction.  quad_perm [1,0,3,2] = lane^1, [2,3,0,1] = lane^2.
#define DCX_VMAX0(x) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x))
#define DCX_MAX_DPP(x, PERM) asm volatile("v_max_f32_dpp %0, %0, %0 quad_perm:" PERM " row_mask:0xf bank_mask:0xf" : "+v"(x))
__device__ __forceinline__ float4 dcx_relu_quad_max(float4 v) {
    DCX_VMAX0(v.x); DCX_VMAX0(v.y); DCX_VMAX0(v.z); DCX_VMAX0(v.w);
    DCX_MAX_DPP(v.x, "[1,0,3,2]"); DCX_MAX_DPP(v.y, "[1,0,3,2]"); DCX_MAX_DPP(v.z, "[1,0,3,2]"); DCX_MAX_DPP(v.w, "[1,0,3,2]");
    DCX_MAX_DPP(v.x, "[2,3,0,1]"); DCX_MAX_DPP(v.y, "[2,3,0,1]"); DCX_MAX_DPP(v.z, "[2,3,0,1]"); DCX_MAX_DPP(v.w, "[2,3,0,1]");
    return v;
}

struct DcxItem {   // one work item = (image, cout tile, [phase,] spatial tile); all fields wave-uniform
    int n, ct, ty, tx;
    int ph;            // dcx_conv_wino2p.h only: output phase 2a + b
};

// Persistent, software-pipelined kernel.
//   work item  = (image n, cout tile, spatial tile);   unit = (work item, 16-channel chunk)
//   Each workgroup walks its items (blockIdx.x, +gridDim.x, ...) unit by unit.  While the MFMAs of
//   unit u run out of LDS buffer u&1, the halo tile of unit u+1 -- possibly the first chunk of the
//   NEXT work item -- is fetched one float4 "piece" per k-step into registers and written to buffer
//   (u+1)&1 one step later, so global-memory latency and the staging index math hide behind
//   16 MFMAs (1024 cycles) per step.  Only the first unit of a workgroup is staged synchronously.
//   One barrier per unit.  Weights (A) and LDS reads (B) are fetched one k-step ahead.
// First work item of XCD x's share: the equal eighth, moved by the weighted deviation in WHOLE CU-ROUNDS (32 items = one item for each
// of an XCD's 32 CUs), rounded to the nearest.  A launch lasts as long as its busiest CU, so a share that is not a multiple of 32 only
// adds a round to a few CUs: with near-equal weights or few items per XCD the deviation rounds to 0 and the split is exactly the equal
// one (a launch of 5 items per workgroup must not become one of 6 for three workgroups: measured -5.5 % on the whole step with
// unquantised shares); a 1.25 % weight moves a boundary of conv1b (2,400 items per XCD) by one round.
__device__ __forceinline__ int dcx_xcd_bound(int total, int x, int cum) {
    const int eq = (int)(((long)total * x) >> 3);
    const int dev = (int)(((long)total * (cum - (x << 13))) >> 16);
    return eq + ((dev + (dev >= 0 ? 16 : -16)) / 32) * 32;
}

template <class C>
__global__ __launch_bounds__(C::NTHREADS, C::OCC) void dcx_conv_mfma_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int WN = C::WN, MT = C::MT, NT = C::NT, TW = C::TW, KS = C::KS;
    constexpr int HW = C::HW, HALO = C::HALO, STEPS = C::STEPS, ITER = C::ITER, PPS = C::PPS;
    constexpr int LDSF = C::LDS_FLOAT4, CQC = C::CQC;
    constexpr bool P4 = C::POOL && MT == 1 && NT == 4;   // 2x2 pooling window held inside one lane

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    // ---- work list --------------------------------------------------------------------------
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    int n_eff = a.n;
    if (a.n_limit != nullptr) n_eff = min(n_eff, *a.n_limit);
    const int total = n_eff * n_ct * tiles;     // images are the slowest index: skipped ones are at the end
    // XCD-aware item walk: block b runs on XCD b % 8 (observed; used for speed only), so the blocks
    // of one XCD walk one contiguous eighth of the item list and share halos / repeated inputs through their L2
    int w = blockIdx.x, w_end = total, gstride = gridDim.x;
    if (a.xcd_walk && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7;
        const int lo = dcx_xcd_bound(total, x, a.xcd_cum[x]);           // equal eighths unless the launcher re-weighted the XCDs
        w_end = dcx_xcd_bound(total, x + 1, a.xcd_cum[x + 1]);
        gstride = gridDim.x >> 3;
        w = lo + (blockIdx.x >> 3);
    }
    if (w >= w_end) return;
    if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0) {
        a.clk_probe[0] = __builtin_amdgcn_s_memtime();
        a.clk_probe[1] = __builtin_amdgcn_s_memrealtime();
    }
    const int nch = a.cin / DCX_CCH;
    auto decode = [&](int wi) {
        DcxItem it;
        it.tx = wi % a.tiles_x; wi /= a.tiles_x;
        it.ty = wi % a.tiles_y; wi /= a.tiles_y;
        it.ph = 0;
        it.ct = wi % n_ct;
        it.n = wi / n_ct;
        return it;
    };
    auto pad_y = [&](const DcxItem&) { return a.pad; };
    auto pad_x = [&](const DcxItem&) { return a.pad; };

    const int hl = a.hin << a.ups, wl = a.win << a.ups;  // logical (up-sampled) input size

    // ---- per-lane pixel of every n-tile (tile-relative, identical for every work item) ---------
    int pixb[NT];      // halo-tile pixel index of the lane's output pixel (tap 0,0) + half*HALO
    int qys[NT], qxs[NT];
    bool qok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int q = (wn * NT + nt) * 32 + l31;
        int qy, qx;
        if (P4) {        // the lane's four n-tiles hold the four pixels of one 2x2 pooling window
            const int pq = wn * 32 + l31;
            const int py = pq / (TW / 2), px = pq - py * (TW / 2);
            qy = 2 * py + (nt >> 1);
            qx = 2 * px + (nt & 1);
            q = pq * 4;
        } else if (C::POOL) {   // lanes 4i..4i+3 hold one 2x2 pooling window
            const int pq = q >> 2, sub = q & 3;
            const int py = pq / (TW / 2), px = pq - py * (TW / 2);
            qy = 2 * py + (sub >> 1);
            qx = 2 * px + (sub & 1);
        } else {
            qy = q / TW;
            qx = q - qy * TW;
        }
        const bool ok = q < C::TILE_PIX;
        qok[nt] = ok;
        qys[nt] = qy;
        qxs[nt] = qx;
        pixb[nt] = half * HALO + (ok ? qy * HW + qx : 0);
    }

    // ---- operand fetch helpers -------------------------------------------------------------------
    // A (weights): address = uniform unit base + 32-bit (lane offset + step offset); lane reads float4
    // #(cq * cout_pad + cout) of tap `tap`, cq = chunk quad 2s + half.
    const unsigned w_lane_off = (unsigned)((half * a.cout_pad) + wm * (MT * 32) + l31) * 16u;
    const unsigned w_tap_stride = (unsigned)((a.cin >> 2) * a.cout_pad) * 16u;   // bytes between taps
    const unsigned w_s_stride = (unsigned)(2 * a.cout_pad) * 16u;                 // bytes between 8-channel groups
    // buffer addressing: voffset = per-lane constant, soffset = uniform (unit base + step offset) in an SGPR,
    // the m-tile stride (512 B) goes into the instruction's immediate field -> no VALU per load
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w), (short)0, (int)((unsigned)KS * KS * w_tap_stride), 0x00020000);
    auto unit_wbase = [&](const DcxItem& it, int c) {
        return (unsigned)((c * CQC) * a.cout_pad + it.ct * C::COUT_TILE) * 16u;
    };
    auto load_a = [&](unsigned wbase, int step, float4 (&dst)[MT]) {
        const int tap = step / (DCX_CCH / 8);
        const int s = step - tap * (DCX_CCH / 8);
        const unsigned soff = wbase + (unsigned)tap * w_tap_stride + (unsigned)s * w_s_stride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane_off + mt * 512, soff, 0);
            dst[mt] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    auto load_b = [&](int buf, int step, float4 (&dst)[NT]) {
        const int tap = step / (DCX_CCH / 8);
        const int s = step - tap * (DCX_CCH / 8);
        const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dst[nt] = sB[buf * LDSF + pixb[nt] + (2 * s) * HALO + dy * HW + dx];
    };
    // halo staging.  Piece `pi` of a unit = float4 #(tid + pi*NTHREADS) of the [cq][halo pixel] tile.  Its
    // (cq, hy, hx) never change, so they are computed once; per unit only the tile origin moves.
    // The load is a raw buffer load whose descriptor covers exactly this unit's 4 channel quads:
    // out-of-image taps (zero padding) get an out-of-range offset and the hardware returns 0.0f --
    // no branch, no select, so the whole unit body stays one basic block the scheduler can interleave.
    // Per piece (constant for the whole kernel): halo coordinates and the byte offset RELATIVE to the tile
    // origin, p_rel = ((cq*hin + ((hy-pad)>>ups) + pad) * win + ((hx-pad)>>ups) + pad) * 16; lanes past the
    // end of the tile get the out-of-range marker.  The uniform tile origin goes into the descriptor base.
    int p_hy[ITER], p_hx[ITER];
    unsigned p_rel[ITER];
#pragma unroll
    for (int pi = 0; pi < ITER; ++pi) {
        const int idx = tid + pi * C::NTHREADS;
        const int cq = idx / HALO;
        const int hp = idx - cq * HALO;
        p_hy[pi] = hp / HW;
        p_hx[pi] = hp - p_hy[pi] * HW;
        const int prow = ((p_hy[pi] - a.pad) >> a.ups) + a.pad, pcol = ((p_hx[pi] - a.pad) >> a.ups) + a.pad;
        p_rel[pi] = idx < LDSF ? (unsigned)((cq * a.hin + prow) * a.win + pcol) * 16u : 0x80000000u;
    }
    auto unit_rsrc = [&](const DcxItem& it, int c) {
        // element (row (ty*TH>>ups) - pad, col (tx*TW>>ups) - pad) of the unit's first channel quad; for tiles on the
        // top/left border this points before the tensor, but those lanes are masked and never dereferenced
        const long tile_off = (long)(((it.ty * C::TH) >> a.ups) - pad_y(it)) * a.win + (((it.tx * C::TW) >> a.ups) - pad_x(it));
        const float* base = a.in + (((size_t)it.n * a.in_cq_total + a.in_cq_off + (size_t)c * CQC) * (size_t)a.hin * a.win
                                    + tile_off) * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
    };
    // a tile is "interior" when its whole halo lies inside the (logical) image: no predicate needed at all
    auto tile_interior = [&](const DcxItem& it) {
        const int sy0 = it.ty * C::TH - pad_y(it), sx0 = it.tx * C::TW - pad_x(it);
        return sy0 >= 0 && sx0 >= 0 && sy0 + C::HH <= hl && sx0 + C::HW <= wl;
    };
    auto stage_off = [&](int sy0, int sx0, int pi) {   // general (border / overhanging tile) form
        const int ly = sy0 + p_hy[pi], lx = sx0 + p_hx[pi];
        const bool inb = (unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl;
        return inb ? p_rel[pi] : 0x80000000u;   // out of range -> the buffer load returns zeros
    };
    auto stage_fetch = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto stage_store = [&](int buf, int pi, const float4& v) {
        const int idx = tid + pi * C::NTHREADS;
        if (idx < LDSF) sB[buf * LDSF + idx] = v;
    };
    // same with the buffer base folded into one per-unit VGPR so each store is base + immediate
    auto stage_store_at = [&](float4* wbase_lds, int pi, const float4& v) {
        if (tid + pi * C::NTHREADS < LDSF) wbase_lds[pi * C::NTHREADS] = v;
    };

    // ---- epilogue constants: per-channel (bias, alpha, beta[, head weight]) staged in LDS once ---------
    // layout after the two halo buffers: sP[k][cout_pad/4] float4, k = 0 bias, 1 alpha, 2 beta, 3 head_w
    float4* sP = sB + 2 * LDSF;
    const int cq_pad = a.cout_pad >> 2;
    for (int i = tid; i < cq_pad; i += C::NTHREADS) {
        sP[i] = reinterpret_cast<const float4*>(a.bias)[i];
        if (C::EPI != DCX_EPI_RAW) {
            sP[cq_pad + i] = reinterpret_cast<const float4*>(a.alpha)[i];
            sP[2 * cq_pad + i] = reinterpret_cast<const float4*>(a.beta)[i];
        }
        if (C::EPI == DCX_EPI_HEAT) sP[3 * cq_pad + i] = reinterpret_cast<const float4*>(a.head_w)[i];
    }
    const int hs = C::POOL ? (a.ho >> 1) : a.ho, ws = C::POOL ? (a.wo >> 1) : a.wo;

    dcx_f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- prologue: first unit staged synchronously ---------------------------------------------
    DcxItem cur = decode(w);
    int c = 0;
    float4 a_c0[MT], a_c1[MT];      // weights of k-steps 0 and 1 of the unit about to run (carried across units)
    {
        const unsigned wb = unit_wbase(cur, 0);
        load_a(wb, 0, a_c0);
        load_a(wb, 1, a_c1);
        const __amdgpu_buffer_rsrc_t r0 = unit_rsrc(cur, 0);
        const int sy0 = cur.ty * C::TH - pad_y(cur), sx0 = cur.tx * C::TW - pad_x(cur);
        float4 v[ITER];
#pragma unroll
        for (int pi = 0; pi < ITER; ++pi) v[pi] = stage_fetch(r0, stage_off(sy0, sx0, pi));
#pragma unroll
        for (int pi = 0; pi < ITER; ++pi) stage_store(0, pi, v[pi]);
    }

    for (int u = 0;; ++u) {
        // ---- what comes after this unit (wave-uniform) ----------------------------------------------
        DcxItem nxt = cur;
        int cn = c + 1;
        bool has_next = true;
        if (cn == nch) {
            if (w + gstride < w_end) { nxt = decode(w + gstride); cn = 0; }
            else { has_next = false; cn = c; }   // nothing follows: harmlessly re-stage the current chunk
        }
        const int buf = u & 1;
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[4 + 3 * u] = __builtin_amdgcn_s_memtime();
        __syncthreads();   // unit u's tile is complete in sB[buf]; everyone is done reading sB[buf ^ 1]
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[5 + 3 * u] = __builtin_amdgcn_s_memtime();

        const __amdgpu_buffer_rsrc_t rs_n = unit_rsrc(nxt, cn);
        const int nsy0 = nxt.ty * C::TH - pad_y(nxt), nsx0 = nxt.tx * C::TW - pad_x(nxt);
        const unsigned wb_cur = unit_wbase(cur, c);
        const unsigned wb_nxt = unit_wbase(nxt, cn);
        // Software pipeline of one unit (everything below is ONE basic block, fully unrolled):
        //   weights  A(t+DA)  are requested DA = 2 k-steps ahead (loads return in order, so the weight
        //            loads queue behind the HBM-latency staging loads and need the extra distance),
        //   LDS      B(t+1)   one step ahead,
        //   staging  piece(t) of the NEXT unit is requested at step t and written to the other LDS
        //            buffer DP = 3 steps later.
        // A wave cannot issue past an MFMA that is waiting for the matrix pipe, so the non-MFMA work is
        // cut into short slots placed after every PAIR of MFMAs (each slot issues in the shadow of the
        // MFMA before it); sched_barrier(0) pins that order.
        float4 aq[STEPS + C::DA][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { aq[0][mt] = a_c0[mt]; aq[1][mt] = a_c1[mt]; }
        float4 bq[STEPS + 1][NT];
        load_b(buf, 0, bq[0]);
        float4* lds_w = sB + (buf ^ 1) * LDSF + tid;   // this thread's slot 0 in the buffer being filled
        float4 pv[STEPS][PPS];
        // staging offsets of the next unit's pieces: the per-piece constants, or (border / overhanging tiles, ONE uniform
        // branch per unit) the bounds-checked ones.  Nothing inside the k-loop is conditional: per-instruction branches on
        // "border tile" or "first chunk" cost ~500 cycles per unit (measured on the Winograd kernel, same structure).
        const bool n_interior = tile_interior(nxt);
        unsigned poff[ITER];
#pragma unroll
        for (int pi = 0; pi < ITER; ++pi) poff[pi] = p_rel[pi];
        if (!n_interior) {
#pragma unroll
            for (int pi = 0; pi < ITER; ++pi) poff[pi] = stage_off(nsy0, nsx0, pi);
        }
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            int pair = 0;
            auto mfma_pair = [&]() {      // MFMAs 2*pair, 2*pair+1 of the step in (j, mt, nt) order
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int f = 2 * pair + e;
                    const int j = f / (MT * NT), mt = (f / NT) % MT, nt = f % NT;
                    const float4 av4 = aq[step][mt], bv4 = bq[step][nt];
                    const float av = j == 0 ? av4.x : j == 1 ? av4.y : j == 2 ? av4.z : av4.w;
                    const float bv = j == 0 ? bv4.x : j == 1 ? bv4.y : j == 2 ? bv4.z : bv4.w;
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt][nt], 0, 0, 0);
                }
                ++pair;
                __builtin_amdgcn_sched_barrier(0);
            };
            __builtin_amdgcn_sched_barrier(0);
            // slot A: LDS operand of step+1
            if (step + 1 < STEPS) load_b(buf, step + 1, bq[step + 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (pair < C::NPAIR) mfma_pair();
            // slot B: weights of step+DA (of the next unit near the end)
            if (step + C::DA < STEPS) load_a(wb_cur, step + C::DA, aq[step + C::DA]);
            else load_a(wb_nxt, step + C::DA - STEPS, aq[step + C::DA]);
            __builtin_amdgcn_sched_barrier(0);
            if (pair < C::NPAIR) mfma_pair();
            // slot C: request this step's staging piece(s) of the next unit (address computed last step)
            if (step < C::LOAD_STEPS) {
#pragma unroll
                for (int k = 0; k < PPS; ++k) {
                    const int pi = step * PPS + k;
                    if (pi < ITER) pv[step][k] = stage_fetch(rs_n, poff[pi]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (pair < C::NPAIR) mfma_pair();
            // slot D: the piece(s) requested DP steps ago have landed: write them to the other LDS buffer
            if (step >= C::DP) {
#pragma unroll
                for (int k = 0; k < PPS; ++k) {
                    const int pi = (step - C::DP) * PPS + k;
                    if (pi < ITER) stage_store_at(lds_w, pi, pv[step - C::DP][k]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (pair < C::NPAIR) mfma_pair();
#pragma unroll
            for (int rest = 4; rest < C::NPAIR; ++rest) mfma_pair();
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { a_c0[mt] = aq[STEPS][mt]; a_c1[mt] = aq[STEPS + 1][mt]; }

        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[6 + 3 * u] = __builtin_amdgcn_s_memtime();
        if (c == nch - 1) {
            // ---- epilogue of work item `cur` ------------------------------------------------------
            const int oy0 = cur.ty * C::TH, ox0 = cur.tx * C::TW;
            const int n = cur.n;
            float hsum[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) hsum[nt] = 0.f;
            // output addressing: uniform 64-bit base of this wave's first channel quad + uniform plane
            // offset per (mt, g) + one 32-bit per-lane offset per n-tile
            const unsigned plane = (unsigned)(hs * ws);          // float4 per channel quad of one image
            const int cq_w0 = (cur.ct * C::COUT_TILE >> 2) + wm * MT * 8;
            char* obase = reinterpret_cast<char*>(a.out)
                        + ((size_t)n * a.out_cq_total + a.out_cq_off + cq_w0) * (size_t)plane * 16;
            unsigned lane_off[NT];
            bool pix_ok[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                int sy = oy0 + qys[nt], sx = ox0 + qxs[nt];
                bool ok = qok[nt] && sy < a.ho && sx < a.wo;
                if (C::POOL) {      // (sy, sx) = the window's top-left pixel; MaxPool2d(2,2) floors odd sizes: the whole window must exist
                    ok = ok && (sy | 1) < a.ho && (sx | 1) < a.wo && (P4 || (l31 & 3) == 0);
                    sy >>= 1; sx >>= 1;
                }
                pix_ok[nt] = ok;
                lane_off[nt] = ((unsigned)half * plane + (unsigned)(sy * ws + sx)) * 16u;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float4 bi[4], al[4], be[4], hw4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // batch the LDS reads of this m-tile's 4 channel quads
                    const int cq = (cur.ct * C::COUT_TILE >> 2) + (wm * MT + mt) * 8 + 2 * g + half;
                    if (C::EPI == DCX_EPI_RAW) bi[g] = sP[cq];
                    if (C::EPI != DCX_EPI_RAW) { al[g] = sP[cq_pad + cq]; be[g] = sP[2 * cq_pad + cq]; }
                    if (C::EPI == DCX_EPI_HEAT) hw4[g] = sP[3 * cq_pad + cq];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cq = (cur.ct * C::COUT_TILE >> 2) + (wm * MT + mt) * 8 + 2 * g + half;  // output channel quad
                    if (P4) {   // BN on the four window pixels, in-lane max, one ReLU, one store
                        float4 m;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            float4 v = make_float4(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1],
                                                   acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
                            v = dcx_fma4(v, al[g], be[g]);
                            if (nt == 0) m = v;
                            else { m.x = dcx_vmax(m.x, v.x); m.y = dcx_vmax(m.y, v.y);
                                   m.z = dcx_vmax(m.z, v.z); m.w = dcx_vmax(m.w, v.w); }
                        }
                        m.x = dcx_vmax(m.x, 0.f); m.y = dcx_vmax(m.y, 0.f);
                        m.z = dcx_vmax(m.z, 0.f); m.w = dcx_vmax(m.w, 0.f);
                        if (pix_ok[0] && cq < a.cout_quads)
                            *reinterpret_cast<float4*>(obase + (size_t)((unsigned)(mt * 8 + 2 * g) * plane * 16u) + lane_off[0]) = m;
                        continue;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        float4 v = make_float4(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1],
                                               acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
                        if (C::EPI == DCX_EPI_RAW) {
                            v.x += bi[g].x; v.y += bi[g].y; v.z += bi[g].z; v.w += bi[g].w;
                        } else {   // conv bias is folded into the BN shift: be = fma(bias, alpha, bn_beta - mean*alpha)
                            v = dcx_fma4(v, al[g], be[g]);
                            if (C::POOL) {
                                v = dcx_relu_quad_max(v);
                            } else {
                                v.x = dcx_vmax(v.x, 0.f); v.y = dcx_vmax(v.y, 0.f);
                                v.z = dcx_vmax(v.z, 0.f); v.w = dcx_vmax(v.w, 0.f);
                            }
                        }
                        if (C::EPI == DCX_EPI_HEAT) {
                            float h = hsum[nt];
                            h = fmaf(v.x, hw4[g].x, h); h = fmaf(v.y, hw4[g].y, h);
                            h = fmaf(v.z, hw4[g].z, h); h = fmaf(v.w, hw4[g].w, h);
                            hsum[nt] = h;
                            continue;
                        }
                        if (pix_ok[nt] && cq < a.cout_quads)
                            *reinterpret_cast<float4*>(obase + (size_t)((unsigned)(mt * 8 + 2 * g) * plane * 16u) + lane_off[nt]) = v;
                    }
                }
            }

            if (C::EPI == DCX_EPI_HEAT) {
                // 1x1 conv to one channel (refinenet.py:81 convPb) + arg-max of this tile (model_utils.py:39-43)
                float best = -INFINITY;
                int besti = 0x7fffffff;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float tot = hsum[nt] + __shfl_xor(hsum[nt], 32);   // the two half-waves hold disjoint couts
                    const float logit = tot + a.head_b;
                    int sy = oy0 + qys[nt], sx = ox0 + qxs[nt];
                    bool ok = qok[nt] && sy < a.ho && sx < a.wo;
                    if (ok) {
                        const int idx = sy * a.wo + sx;
                        if (a.heat != nullptr && half == 0) a.heat[((size_t)n * a.ho + sy) * a.wo + sx] = logit;
                        if (logit > best || (logit == best && idx < besti)) { best = logit; besti = idx; }
                    }
                }
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(best, off);
                    const int oi = __shfl_xor(besti, off);
                    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                }
                __syncthreads();   // all waves are done reading sB[buf]: reuse its head as reduction scratch
                float* red_v = reinterpret_cast<float*>(sB + buf * LDSF);
                int* red_i = reinterpret_cast<int*>(sB + buf * LDSF) + 16;
                if (lane == 0) { red_v[wave] = best; red_i[wave] = besti; }
                __syncthreads();
                if (tid == 0) {
                    for (int wv = 1; wv < C::WM * C::WN; ++wv) {
                        const float ov = red_v[wv];
                        const int oi = red_i[wv];
                        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                    }
                    const size_t slot = (size_t)n * tiles + cur.ty * a.tiles_x + cur.tx;
                    a.part_val[slot] = best;
                    a.part_idx[slot] = besti;
                }
            }
        }

        if (c == nch - 1) {   // the next unit starts a new work item
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        }

        if (!has_next) {
            if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0) {
                a.clk_probe[2] = __builtin_amdgcn_s_memtime();
                a.clk_probe[3] = __builtin_amdgcn_s_memrealtime();
            }
            break;
        }
        if (cn == 0) w += gstride;
        cur = nxt;
        c = cn;
    }
}

int dcx_device_cu_count();   // dcx_conv_mfma.hip: CUs of the CURRENT device (cached per device)
constexpr int DCX_MAX_DEVICES = 64;
int dcx_current_device();    // hipGetDevice() clamped to [0, DCX_MAX_DEVICES)
int dcx_occupancy_override();
int dcx_xcd_walk_enabled();   // DCX_XCD_WALK=0 keeps the flat item walk (A/B runs)
void dcx_fill_xcd_cum(DcxConvArgs& a);   // the current device's cumulative XCD weights (dcx_conv_mfma.hip)

template <class C>
static int dcx_conv_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    a.tiles_x = (a.wo + C::TW - 1) / C::TW;
    a.tiles_y = (a.ho + C::TH - 1) / C::TH;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0) return DCX_E_SHAPE;
    const long items = (long)a.n * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y;
    if (items <= 0 || items > 0x7fffffffL) return DCX_E_SHAPE;
    const int occ_env = dcx_occupancy_override();                      // tuning knob (DCX_OCC), 0 = default
    const long resident = (long)dcx_device_cu_count() * (occ_env > 0 && occ_env < C::OCC ? occ_env : C::OCC);   // persistent workgroups
    const long blocks = items < resident ? items : resident;
    a.xcd_walk = dcx_xcd_walk_enabled() && blocks == resident && (resident & 7) == 0 ? 1 : 0;
    dcx_fill_xcd_cum(a);
    static bool attr_set[DCX_MAX_DEVICES] = {};      // the attribute is per device (multi-GPU processes)
    const int dev_i = dcx_current_device();
    if (!attr_set[dev_i]) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_mfma_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C::LDS_BYTES + 8192)));
        attr_set[dev_i] = true;
    }
    const size_t lds = C::LDS_BYTES + (size_t)a.cout_pad * 16;   // halo double buffer + 4 per-channel parameter arrays
    if (lds > 160 * 1024) return DCX_E_SHAPE;
    hipLaunchKernelGGL((dcx_conv_mfma_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), lds, stream, a);
    return (int)hipGetLastError();
}
