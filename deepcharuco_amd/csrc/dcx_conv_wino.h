// dcx_conv_wino.h -- 3x3 convolution + BN + ReLU (+2x2 max-pool) with a 1-D Winograd F(2,3) transform along x,
// on the gfx950 fp32 matrix cores.  Same layers, same C4 layouts, same persistent software pipeline as
// dcx_conv_mfma.h (read that file first); what changes is the arithmetic:
//
//   for one output row y, one pair of output pixels (x0, x0+1), one kernel row ky and one input channel:
//       d0..d3 = in[y+ky-pad][x0-pad .. x0-pad+3]
//       v0 = d0 - d2    v1 = d1 + d2    v2 = d2 - d1    v3 = d1 - d3          (input transform, exact fp32 ops)
//       u0 = g0         u1 = ((g0+g1)+g2)*0.5f   u2 = ((g0-g1)+g2)*0.5f   u3 = g2   (weights, transformed on the host)
//       m_p += u_p * v_p                                                      (p = 0..3: FOUR products for TWO outputs
//                                                                              instead of six -> 1.5x fewer MFMAs)
//       out[x0] = (m0 + m1) + m2     out[x0+1] = (m1 - m2) - m3               (output transform)
//
// GEMM view: four independent GEMMs (one per position p), M = cout, N = output pairs, K = 3 * cin.  A lane owns one
// output PAIR and keeps the four partial sums m_p in four accumulators, so the output transform, BN, ReLU and the
// horizontal half of the 2x2 max-pool are in-lane.  The transformed input tile lives in LDS as
// sV[buf][p][cq][row][pair] float4.  Staging is two passes (every global load lane-contiguous): raw float4 -> sR, one
// extra barrier, then per (cq, row, pair) piece 4 ds_read_b128, 8 v_pk_add_f32, 4 ds_write_b128.  The k-loop is the same
// MFMA stream as in the direct kernel with "tap" = (ky, p): 12 weight planes instead of 9, and it is ONE basic block
// per unit (no per-instruction conditions: measured ~500 cycles per unit when they were there).
//
// Summation order (restated bit-exactly by oracle/conv_exact.c: dcx_conv_wino_exact):
//   m_p = 0;  for chunk c (16 cin) / ky / s in 0..1 / j in 0..3:
//                 m_p = fmaf(u_p[8s+j], v_p[8s+j], m_p);  m_p = fmaf(u_p[8s+4+j], v_p[8s+4+j], m_p);
//   then the output transform above, y = fmaf(out, alpha, beta2), ReLU, max-pool.
// F(2,3) in one dimension only: error growth is ~2x that of the direct sum (no large transform constants), far
// inside the 1e-4 logit margin of the parity policy (tests/test_gpu_parity.py).
#pragma once
#include "dcx_conv_mfma.h"

#include <type_traits>

// one v_pk_add_f32 (hipcc scalarises float2 +/- into two v_add_f32; every VALU instruction in the k-loop costs matrix time)
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

template <int WM_, int WN_, int TH_, int TW_, bool POOL_, int EPI_ = DCX_EPI_BNRELU>
struct DcxWinoCfg {
    static constexpr int WM = WM_, WN = WN_, TH = TH_, TW = TW_;
    static constexpr bool POOL = POOL_;
    static constexpr int EPI = EPI_;                   // DCX_EPI_BNRELU or DCX_EPI_HEAT (RefineNet head)
    static constexpr int NTHREADS = WM * WN * 64;
    static constexpr int COUT_TILE = WM * 32;
    static constexpr int PW = TW / 2;                  // output pairs per tile row
    static constexpr int CAP = WN * 32;                // output pairs a workgroup can hold
    static constexpr int TILE_PAIRS = TH * PW;
    static constexpr int HH = TH + 2;                  // input rows of the tile
    static constexpr int PLANE = HH * PW;              // float4 per (position, channel quad)
    static constexpr int CQC = DCX_CCH / 4;
    static constexpr int PIECES = CQC * PLANE;         // staging work items per unit: (cq, row, pair)
    static constexpr int ITER = (PIECES + NTHREADS - 1) / NTHREADS;   // transform pieces per thread per unit
    static constexpr int DUP = ITER * NTHREADS - PIECES;   // threads past the last piece redo pieces [PIECES-DUP, PIECES):
                                                           // same values written twice, no divergent branch in the k-loop
    static constexpr int PSTRIDE = PIECES;             // float4 between position planes
    static constexpr int LDS_FLOAT4 = 4 * PSTRIDE;     // one transformed buffer: 4 positions
    static constexpr int RW = TW + 8;                  // raw columns per staged row: [x0-4, x0+TW+4) although the conv needs
                                                       // [x0-1, x0+TW+1): with 4-pixel margins every group of 4 lanes maps to
                                                       // one ALIGNED 64-B block (misaligned groups cost ~4x address-path time)
    static constexpr int RAW = CQC * HH * RW;          // raw float4 of the tile (one 16-channel chunk)
    static constexpr int ITER_R = (RAW + NTHREADS - 1) / NTHREADS;   // raw float4 per thread per unit
    static constexpr int RAW_PAD = ITER_R * NTHREADS;  // raw buffer incl. padding slots (they receive zeros)
    static constexpr size_t LDS_BYTES = (size_t)(2 * LDS_FLOAT4 + RAW_PAD) * 16;   // 2 transformed buffers + 1 raw buffer
    static constexpr int STEPS = 3 * (DCX_CCH / 8);    // k-steps per unit: (ky, 8-channel group)
    static constexpr int DA = 2;                       // weights are requested DA k-steps ahead (3 and 4 measured the same)
    static constexpr int RAW_STORE_STEP = 3;           // raw loads are issued in k-step 0 and written to LDS in this step
    static constexpr int NPAIR = 8;                    // 16 MFMAs per k-step: 4 registers j x 4 positions p
    static constexpr int OCC = 2;
    static_assert(TW % 2 == 0, "tile width must be even");
    static_assert(TILE_PAIRS <= CAP, "tile does not fit the wave layout");
    static_assert(!POOL || (TH % 2 == 0 && TILE_PAIRS <= WN * 32), "pooled tiles must have an even height");
    static_assert(LDS_BYTES * 2 <= 150 * 1024, "LDS tile too large for two workgroups per CU");
    static_assert(ITER <= 2 && ITER_R <= 4 && STEPS == 6 && DUP <= NTHREADS, "k-loop schedule assumes <= 2 transform pieces and <= 4 raw float4 per thread");
    static_assert(EPI == DCX_EPI_BNRELU || (EPI == DCX_EPI_HEAT && !POOL && WM == 2), "unsupported epilogue");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, C::OCC) void dcx_conv_wino_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int WN = C::WN, PW = C::PW, PLANE = C::PLANE, PIECES = C::PIECES, PSTRIDE = C::PSTRIDE;
    constexpr int STEPS = C::STEPS, ITER = C::ITER, ITER_R = C::ITER_R, RW = C::RW, LDSF = C::LDS_FLOAT4, CQC = C::CQC;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    // ---- work list (same walk as the direct kernel) ------------------------------------------------
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    int n_eff = a.n;
    if (a.n_limit != nullptr) n_eff = min(n_eff, *a.n_limit);
    const int total = n_eff * n_ct * tiles;
    // XCD-aware item walk (see dcx_conv_wino2.h): block b runs on XCD b % 8 (observed; used for speed only), so the blocks
    // of one XCD walk one contiguous eighth of the item list and share halos / repeated inputs through their L2
    int w = blockIdx.x, w_end = total, gstride = gridDim.x;
    if (a.xcd_walk && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7;
        const int lo = (int)(((long)total * x) >> 3);
        w_end = (int)(((long)total * (x + 1)) >> 3);
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
        it.ct = wi % n_ct;
        it.n = wi / n_ct;
        return it;
    };
    const int hl = a.hin << a.ups, wl = a.win << a.ups;

    // ---- the lane's output pair (tile-relative) ---------------------------------------------------
    int qy, qxp;
    bool qok;
    if (C::POOL) {   // lanes i and i+16 hold vertically adjacent pairs: a 2x2 pooling window = one lane pair
        const int pq = wn * 16 + (l31 & 15);
        const int py = pq / PW, px = pq - py * PW;
        qy = 2 * py + (l31 >> 4);
        qxp = px;
        qok = pq < C::TILE_PAIRS / 2;
    } else {
        const int q = wn * 32 + l31;
        qy = q / PW;
        qxp = q - qy * PW;
        qok = q < C::TILE_PAIRS;
    }
    const int pixb = half * PLANE + (qok ? qy * PW + qxp : 0);

    // ---- operand fetch helpers -------------------------------------------------------------------
    // weights: [ky*4 + p][cin/4][cout_pad][4]
    const unsigned w_lane_off = (unsigned)((half * a.cout_pad) + wm * 32 + l31) * 16u;
    const unsigned w_tap_stride = (unsigned)((a.cin >> 2) * a.cout_pad) * 16u;
    const unsigned w_s_stride = (unsigned)(2 * a.cout_pad) * 16u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_wino), (short)0, (int)(12u * w_tap_stride), 0x00020000);
    auto unit_wbase = [&](const DcxItem& it, int c) {
        return (unsigned)((c * CQC) * a.cout_pad + it.ct * C::COUT_TILE) * 16u;
    };
    auto load_a1 = [&](unsigned wbase, int step, int p) {
        const int ky = step / (DCX_CCH / 8);
        const int s = step - ky * (DCX_CCH / 8);
        const unsigned soff = wbase + (unsigned)(ky * 4 + p) * w_tap_stride + (unsigned)s * w_s_stride;
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane_off, soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto load_a = [&](unsigned wbase, int step, float4 (&dst)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) dst[p] = load_a1(wbase, step, p);
    };
    auto load_b1 = [&](int buf, int step, int p) {
        const int ky = step / (DCX_CCH / 8);
        const int s = step - ky * (DCX_CCH / 8);
        return sB[buf * LDSF + p * PSTRIDE + (2 * s) * PLANE + pixb + ky * PW];
    };
    auto load_b = [&](int buf, int step, float4 (&dst)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) dst[p] = load_b1(buf, step, p);
    };
    // Staging is two passes through LDS so that every global load is lane-contiguous (16-B loads at a 32-B lane
    // stride cost ~4x the address-path time, measured):
    //   raw pass:   float4 #(tid + k*NTHREADS) of the raw [cq][row][RW] tile -> sR (exactly the direct kernel's staging:
    //               per-piece constant offset relative to the tile origin, hardware out-of-range -> 0.0f)
    //   transform:  piece (cq, row, pair) reads its four raw pixels 2*pair + e from sR, applies B^T d, writes the four
    //               position planes of the next unit's transformed buffer
    const int xs = 4 - a.pad;                   // raw column of the first input column the convolution needs
    int r_hy[ITER_R], r_hx[ITER_R];
    unsigned r_rel[ITER_R];
#pragma unroll
    for (int k = 0; k < ITER_R; ++k) {
        const int idx = tid + k * C::NTHREADS;
        const int cq = idx / (C::HH * RW);
        const int hp = idx - cq * (C::HH * RW);
        r_hy[k] = hp / RW;
        r_hx[k] = hp - r_hy[k] * RW;
        r_hx[k] -= xs;      // column relative to the tile's first needed input column (sx0): -3-.. +TW+4..
        const int prow = ((r_hy[k] - a.pad) >> a.ups) + a.pad, pcol = ((r_hx[k] - a.pad) >> a.ups) + a.pad;
        // the margin columns exist only to align the lane groups: they are never read back, so they are not fetched
        const bool needed = idx < C::RAW && r_hx[k] >= 0 && r_hx[k] < C::TW + 2;
        r_rel[k] = needed ? (unsigned)((cq * a.hin + prow) * a.win + pcol) * 16u : 0x80000000u;
    }
    float4* sR = sB + 2 * LDSF;                 // raw buffer (single: written in k-step 3, read in k-steps 4, 5)
    int t_src[ITER], t_dst[ITER];               // transform piece: raw index of its first pixel, index inside a position plane
#pragma unroll
    for (int pi = 0; pi < ITER; ++pi) {
        int idx = tid + pi * C::NTHREADS;
        if (idx >= PIECES) idx -= C::DUP;
        const int cq = idx / PLANE;
        const int hp = idx - cq * PLANE;
        const int row = hp / PW, pr = hp - row * PW;
        t_src[pi] = (cq * C::HH + row) * RW + 2 * pr + xs;
        t_dst[pi] = idx;
    }
    auto unit_rsrc = [&](const DcxItem& it, int c) {
        const long tile_off = (long)(((it.ty * C::TH) >> a.ups) - a.pad) * a.win + (((it.tx * C::TW) >> a.ups) - a.pad);
        const float* base = a.in + (((size_t)it.n * a.in_cq_total + a.in_cq_off + (size_t)c * CQC) * (size_t)a.hin * a.win
                                    + tile_off) * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
    };
    auto tile_interior = [&](const DcxItem& it) {
        const int sy0 = it.ty * C::TH - a.pad, sx0 = it.tx * C::TW - a.pad;
        return sy0 >= 0 && sx0 - xs >= 0 && sy0 + C::HH <= hl && sx0 - xs + RW <= wl;
    };
    auto stage_fetch = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // input transform of one piece, one position at a time: 2 v_pk_add_f32 + 1 ds_write_b128
    auto xform_p = [&](int p, const float4 (&d)[4]) {
        const float4& x = p == 0 ? d[0] : p == 1 ? d[1] : p == 2 ? d[2] : d[1];
        const float4& y = p == 0 ? d[2] : p == 1 ? d[2] : p == 2 ? d[1] : d[3];
        // packed fp32 (two lanes of the float4 per VALU instruction): v_pk_add_f32 with a neg modifier for the differences
        const dcx_f32x2 xl = {x.x, x.y}, xh = {x.z, x.w}, yl = {y.x, y.y}, yh = {y.z, y.w};
        const dcx_f32x2 lo = p == 1 ? dcx_pk_add(xl, yl) : dcx_pk_sub(xl, yl), hi = p == 1 ? dcx_pk_add(xh, yh) : dcx_pk_sub(xh, yh);
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    auto stage_store_p = [&](float4* vbuf, int pi, int p, const float4 (&d)[4]) {
        vbuf[t_dst[pi] + p * C::PSTRIDE] = xform_p(p, d);
    };

    // ---- epilogue constants in LDS: sP[0] alpha, sP[1] beta2 ---------------------------------------------
    float4* sP = sB + 2 * LDSF + C::RAW_PAD;
    const int cq_pad = a.cout_pad >> 2;
    for (int i = tid; i < cq_pad; i += C::NTHREADS) {
        sP[i] = reinterpret_cast<const float4*>(a.alpha)[i];
        sP[cq_pad + i] = reinterpret_cast<const float4*>(a.beta)[i];
        if (C::EPI == DCX_EPI_HEAT) sP[2 * cq_pad + i] = reinterpret_cast<const float4*>(a.head_w)[i];
    }
    const int hs = C::POOL ? (a.ho >> 1) : a.ho, ws = C::POOL ? (a.wo >> 1) : a.wo;

    dcx_f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- prologue: first unit staged synchronously ---------------------------------------------
    DcxItem cur = decode(w);
    int c = 0;
    float4 a_c[C::DA][4];          // weights of the first DA k-steps of the unit about to run (carried across units)
    {
        const unsigned wb = unit_wbase(cur, 0);
#pragma unroll
        for (int d = 0; d < C::DA; ++d) load_a(wb, d, a_c[d]);
        const __amdgpu_buffer_rsrc_t r0 = unit_rsrc(cur, 0);
        const int sy0 = cur.ty * C::TH - a.pad, sx0 = cur.tx * C::TW - a.pad;
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) {
            const int ly = sy0 + r_hy[k], lx = sx0 + r_hx[k];
            const bool inb = (unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl;
            sR[tid + k * C::NTHREADS] = stage_fetch(r0, inb ? r_rel[k] : 0x80000000u);
        }
        __syncthreads();
#pragma unroll
        for (int pi = 0; pi < ITER; ++pi) {
            float4 d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = sR[t_src[pi] + e];
#pragma unroll
            for (int p = 0; p < 4; ++p) stage_store_p(sB, pi, p, d);
        }
    }

    for (int u = 0;; ++u) {
        DcxItem nxt = cur;
        int cn = c + 1;
        bool has_next = true;
        if (cn == nch) {
            if (w + gstride < w_end) { nxt = decode(w + gstride); cn = 0; }
            else { has_next = false; cn = c; }
        }
        const int buf = u & 1;
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[4 + 3 * u] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[5 + 3 * u] = __builtin_amdgcn_s_memtime();

        const __amdgpu_buffer_rsrc_t rs_n = unit_rsrc(nxt, cn);
        const int nsy0 = nxt.ty * C::TH - a.pad, nsx0 = nxt.tx * C::TW - a.pad;
        const unsigned wb_cur = unit_wbase(cur, c);
        const unsigned wb_nxt = unit_wbase(nxt, cn);
        float4 aq[STEPS + C::DA][4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int d = 0; d < C::DA; ++d) aq[d][p] = a_c[d][p];
        float4 bq[STEPS + 1][4];
        load_b(buf, 0, bq[0]);
        float4* vnext = sB + (buf ^ 1) * LDSF;      // transformed buffer of the next unit
        float4 rv[ITER_R];                           // raw float4 in flight (k-step 0 -> RAW_STORE_STEP)
        float4 td[4];                                // the four raw pixels of the piece being transformed
        float4 tv;                                   // transformed position waiting for its LDS write
        const bool n_interior = tile_interior(nxt);
        // raw-load offsets of the next unit: the per-piece constants, or (border tiles, ONE uniform branch per unit) the
        // bounds-checked ones
        unsigned roff[ITER_R];
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) roff[k] = r_rel[k];
        if (!n_interior) {
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) {
                const int ly = nsy0 + r_hy[k], lx = nsx0 + r_hx[k];
                roff[k] = ((unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl) ? r_rel[k] : 0x80000000u;
            }
        }
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            int pair = 0;
            auto mfma_pair = [&]() {      // MFMAs 2*pair, 2*pair+1 of the step in (j, p) order
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int f = 2 * pair + e;
                    const int j = f >> 2, p = f & 3;
                    const float4 av4 = aq[step][p], bv4 = bq[step][p];
                    const float av = j == 0 ? av4.x : j == 1 ? av4.y : j == 2 ? av4.z : av4.w;
                    const float bv = j == 0 ? bv4.x : j == 1 ? bv4.y : j == 2 ? bv4.z : bv4.w;
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p], 0, 0, 0);
                }
                ++pair;
                __builtin_amdgcn_sched_barrier(0);
            };
            // The k-loop of a unit is ONE basic block (no condition inside it: per-instruction branches on "first chunk" or
            // "border tile" cost ~500 cycles per unit).  Eight slots, one before each MFMA pair (the pair before it covers
            // its issue time); a slot holds at most one LDS read, one weight load and one staging operation:
            //   slots 0..3: B(step+1, p = slot), A(step+DA, p = slot)
            //   k-step 0, slots 4..7: load raw float4 #(slot-4) of the next unit
            //   k-step 3, slots 4..7: raw float4 #(slot-4) -> sR;   then ONE extra barrier before k-step 4
            //   k-steps 4, 5 (transform piece 0, 1): slots 0..3 read raw pixel e = slot from sR, slots 4..7 transform and
            //               write position p = slot-4 into the next unit's buffer
            if (step == C::RAW_STORE_STEP + 1) __syncthreads();   // the raw tile is complete in sR
#pragma unroll
            for (int slot = 0; slot < 8; ++slot) {
                __builtin_amdgcn_sched_barrier(0);
                if (slot < 4) {
                    if (step + 1 < STEPS) bq[step + 1][slot] = load_b1(buf, step + 1, slot);
                    if (step + C::DA < STEPS) aq[step + C::DA][slot] = load_a1(wb_cur, step + C::DA, slot);
                    else aq[step + C::DA][slot] = load_a1(wb_nxt, step + C::DA - STEPS, slot);
                    if (step >= STEPS - ITER) td[slot] = sR[t_src[step >= STEPS - ITER ? step - (STEPS - ITER) : 0] + slot];
                } else {
                    const int e = slot - 4;
                    if (step == 0 && e < ITER_R) rv[e] = stage_fetch(rs_n, roff[e]);
                    if (step == C::RAW_STORE_STEP && e < ITER_R) sR[tid + e * C::NTHREADS] = rv[e];
                    if (step >= STEPS - ITER) {   // the write of a position is issued one slot after its two adds
                        const int tpi = step >= STEPS - ITER ? step - (STEPS - ITER) : 0;
                        if (e > 0) vnext[t_dst[tpi] + (e - 1) * C::PSTRIDE] = tv;
                        tv = xform_p(e, td);
                        if (e == 3) vnext[t_dst[tpi] + 3 * C::PSTRIDE] = tv;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_pair();
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int d = 0; d < C::DA; ++d) a_c[d][p] = aq[STEPS + d][p];

        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[6 + 3 * u] = __builtin_amdgcn_s_memtime();
        if (c == nch - 1) {
            // ---- epilogue: output transform, BN, ReLU (, pool), store ---------------------------------
            const int oy0 = cur.ty * C::TH, ox0 = cur.tx * C::TW;
            const unsigned plane = (unsigned)(hs * ws);
            const int cq_w0 = (cur.ct * C::COUT_TILE >> 2) + wm * 8;
            char* obase = reinterpret_cast<char*>(a.out)
                        + ((size_t)cur.n * a.out_cq_total + a.out_cq_off + cq_w0) * (size_t)plane * 16;
            const int sy = oy0 + qy, sx = ox0 + 2 * qxp;
            bool ok0, ok1;
            unsigned lane_off;
            if (C::POOL) {
                ok0 = qok && (l31 < 16) && sy < a.ho && sx < a.wo;
                ok1 = false;
                lane_off = ((unsigned)half * plane + (unsigned)((sy >> 1) * ws + (sx >> 1))) * 16u;
            } else {
                ok0 = qok && sy < a.ho && sx < a.wo;
                ok1 = ok0 && sx + 1 < a.wo;
                lane_off = ((unsigned)half * plane + (unsigned)(sy * ws + sx)) * 16u;
            }
            float4 al[4], be[4], hw4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cq = (cur.ct * C::COUT_TILE >> 2) + wm * 8 + 2 * g + half;
                al[g] = sP[cq];
                be[g] = sP[cq_pad + cq];
                if (C::EPI == DCX_EPI_HEAT) hw4[g] = sP[2 * cq_pad + cq];
            }
            float hsum0 = 0.f, hsum1 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cq = (cur.ct * C::COUT_TILE >> 2) + wm * 8 + 2 * g + half;
                float4 m[4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    m[p] = make_float4(acc[p][4 * g + 0], acc[p][4 * g + 1], acc[p][4 * g + 2], acc[p][4 * g + 3]);
                float4 y0 = make_float4((m[0].x + m[1].x) + m[2].x, (m[0].y + m[1].y) + m[2].y,
                                        (m[0].z + m[1].z) + m[2].z, (m[0].w + m[1].w) + m[2].w);
                float4 y1 = make_float4((m[1].x - m[2].x) - m[3].x, (m[1].y - m[2].y) - m[3].y,
                                        (m[1].z - m[2].z) - m[3].z, (m[1].w - m[2].w) - m[3].w);
                y0 = dcx_fma4(y0, al[g], be[g]);
                y1 = dcx_fma4(y1, al[g], be[g]);
                char* dst = obase + (size_t)((unsigned)(2 * g) * plane * 16u) + lane_off;
                if (C::POOL) {
                    float4 v = make_float4(dcx_vmax(y0.x, y1.x), dcx_vmax(y0.y, y1.y), dcx_vmax(y0.z, y1.z), dcx_vmax(y0.w, y1.w));
                    // vertical neighbour = lane ^ 16 (ds_swizzle: runs on the LDS crossbar, not on the vector ALU)
                    float4 o;
                    o.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v.x), 0x401F));
                    o.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v.y), 0x401F));
                    o.z = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v.z), 0x401F));
                    o.w = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v.w), 0x401F));
                    v.x = dcx_vmax(dcx_vmax(v.x, o.x), 0.f); v.y = dcx_vmax(dcx_vmax(v.y, o.y), 0.f);
                    v.z = dcx_vmax(dcx_vmax(v.z, o.z), 0.f); v.w = dcx_vmax(dcx_vmax(v.w, o.w), 0.f);
                    if (ok0 && cq < a.cout_quads) *reinterpret_cast<float4*>(dst) = v;
                } else {
                    y0.x = dcx_vmax(y0.x, 0.f); y0.y = dcx_vmax(y0.y, 0.f); y0.z = dcx_vmax(y0.z, 0.f); y0.w = dcx_vmax(y0.w, 0.f);
                    y1.x = dcx_vmax(y1.x, 0.f); y1.y = dcx_vmax(y1.y, 0.f); y1.z = dcx_vmax(y1.z, 0.f); y1.w = dcx_vmax(y1.w, 0.f);
                    if (C::EPI == DCX_EPI_HEAT) {   // 1x1 conv to one channel: this lane's 16 couts, g-major
                        hsum0 = fmaf(y0.x, hw4[g].x, hsum0); hsum0 = fmaf(y0.y, hw4[g].y, hsum0);
                        hsum0 = fmaf(y0.z, hw4[g].z, hsum0); hsum0 = fmaf(y0.w, hw4[g].w, hsum0);
                        hsum1 = fmaf(y1.x, hw4[g].x, hsum1); hsum1 = fmaf(y1.y, hw4[g].y, hsum1);
                        hsum1 = fmaf(y1.z, hw4[g].z, hsum1); hsum1 = fmaf(y1.w, hw4[g].w, hsum1);
                        continue;
                    }
                    if (cq < a.cout_quads) {
                        if (ok0) *reinterpret_cast<float4*>(dst) = y0;
                        if (ok1) *reinterpret_cast<float4*>(dst + 16) = y1;
                    }
                }
            }
            if (C::EPI == DCX_EPI_HEAT) {
                // refinenet.py:81 convPb + model_utils.py:39-43 arg-max.  The 64 couts of a pixel live in 4 places
                // (2 half-waves x 2 waves): logit = ((h[wm0,half0] + h[wm0,half1]) + (h[wm1,half0] + h[wm1,half1])) + bias
                const float t0 = hsum0 + __shfl_xor(hsum0, 32), t1 = hsum1 + __shfl_xor(hsum1, 32);
                __syncthreads();   // every wave is done reading sB[buf]: its head becomes reduction scratch
                float* scr = reinterpret_cast<float*>(sB + buf * LDSF);
                if (wm == 1 && half == 0) { scr[(wn * 32 + l31) * 2] = t0; scr[(wn * 32 + l31) * 2 + 1] = t1; }
                __syncthreads();
                float best = -INFINITY;
                int besti = 0x7fffffff;
                if (wm == 0) {
                    const float l0 = (t0 + scr[(wn * 32 + l31) * 2]) + a.head_b;
                    const float l1 = (t1 + scr[(wn * 32 + l31) * 2 + 1]) + a.head_b;
                    if (ok0) {
                        const int idx = sy * a.wo + sx;
                        if (a.heat != nullptr && half == 0) a.heat[((size_t)cur.n * a.ho + sy) * a.wo + sx] = l0;
                        best = l0; besti = idx;
                        if (ok1) {
                            if (a.heat != nullptr && half == 0) a.heat[((size_t)cur.n * a.ho + sy) * a.wo + sx + 1] = l1;
                            if (l1 > best) { best = l1; besti = idx + 1; }
                        }
                    }
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        const float ov = __shfl_xor(best, off);
                        const int oi = __shfl_xor(besti, off);
                        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                    }
                }
                float* red_v = scr + 256;
                int* red_i = reinterpret_cast<int*>(scr) + 256 + 16;
                if (wm == 0 && lane == 0) { red_v[wn] = best; red_i[wn] = besti; }
                __syncthreads();
                if (tid == 0) {
                    for (int wv = 1; wv < WN; ++wv) {
                        const float ov = red_v[wv];
                        const int oi = red_i[wv];
                        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                    }
                    a.part_val[(size_t)cur.n * tiles + cur.ty * a.tiles_x + cur.tx] = best;
                    a.part_idx[(size_t)cur.n * tiles + cur.ty * a.tiles_x + cur.tx] = besti;
                }
            }
        }

        if (c == nch - 1) {   // next unit starts a new work item
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
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

template <class C>
static int dcx_conv_wino_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    a.tiles_x = (a.wo + C::TW - 1) / C::TW;
    a.tiles_y = (a.ho + C::TH - 1) / C::TH;
    if (a.w_wino == nullptr || a.alpha == nullptr || a.beta == nullptr) return DCX_E_ARG;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0) return DCX_E_SHAPE;
    const long items = (long)a.n * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y;
    if (items <= 0 || items > 0x7fffffffL) return DCX_E_SHAPE;
    const int occ_env = dcx_occupancy_override();
    const long resident = (long)dcx_device_cu_count() * (occ_env > 0 && occ_env < C::OCC ? occ_env : C::OCC);
    const long blocks = items < resident ? items : resident;
    a.xcd_walk = dcx_xcd_walk_enabled() && blocks == resident && (resident & 7) == 0 ? 1 : 0;
    static bool attr_set[DCX_MAX_DEVICES] = {};      // the attribute is per device (multi-GPU processes)
    const int dev_i = dcx_current_device();
    if (!attr_set[dev_i]) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_wino_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(C::LDS_BYTES + 8192)));
        attr_set[dev_i] = true;
    }
    if (C::EPI == DCX_EPI_HEAT && (a.head_w == nullptr || a.part_val == nullptr || a.part_idx == nullptr)) return DCX_E_ARG;
    if (C::EPI != DCX_EPI_HEAT && a.out == nullptr) return DCX_E_ARG;
    const size_t lds = C::LDS_BYTES + (size_t)a.cout_pad * 12;   // transformed-tile double buffer + alpha, beta2, head weights
    if (lds > 160 * 1024) return DCX_E_SHAPE;
    hipLaunchKernelGGL((dcx_conv_wino_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), lds, stream, a);
    return (int)hipGetLastError();
}
