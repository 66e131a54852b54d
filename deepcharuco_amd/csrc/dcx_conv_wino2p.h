// dcx_conv_wino2p.h -- 3x3 convolution (pad 1) over a nearest-x2 UP-SAMPLED input + BN + ReLU (+ RefineNet head), computed
// on the LOW-RESOLUTION tensor as four 2x2-tap "phase" convolutions (dcx_conv_mfma.h, PH variant) EACH of which runs as a 2-D
// Winograd F(2x2, 2x2): 9 products per 2x2 output tile and channel instead of 16, i.e. 2.25 multiply-adds per output pixel
// where the layer as written needs 9 (the phase kernel: 4, the 2-D Winograd kernels on the up-sampled image: 4).
//
//   phase (a, b), output pixel (2y + a, 2x + b) = sum_{dy, dx in 0..1} Wp[a][b][dy][dx] * X[y + dy - (1 - a)][x + dx - (1 - b)]
//   (Wp: the 3x3 kernel's rows / columns pre-summed in fp32 exactly as pack_conv_ups2 does; X: low-resolution, zero outside).
//   For a 2x2 tile of low-resolution positions (y0.., x0..) of ONE phase:  d[r][s] = X[y0 - (1-a) + r][x0 - (1-b) + s], r, s = 0..2
//       t[0][s] = d[0][s] - d[1][s]   t[1][s] = d[1][s]   t[2][s] = d[2][s] - d[1][s]          (input transform, exact fp32 ops)
//       v[xi][0] = t[xi][0] - t[xi][1]   v[xi][1] = t[xi][1]   v[xi][2] = t[xi][2] - t[xi][1]
//       Wc[dy][0] = Wp[dy][0]   Wc[dy][1] = Wp[dy][0] + Wp[dy][1]   Wc[dy][2] = Wp[dy][1]       (weights, host, fp32, this order)
//       U[0][nu] = Wc[0][nu]    U[1][nu] = Wc[0][nu] + Wc[1][nu]    U[2][nu] = Wc[1][nu]
//       m[xi][nu] = sum over cin of U[xi][nu] * v[xi][nu]                                        (nine GEMMs: M = cout, N = tiles, K = cin)
//       y[i][j] = sum_{xi, nu} AT[i][xi] AT[j][nu] m[xi][nu],  AT = [[1, 1, 0], [0, 1, 1]]       (output transform)
//   All transform constants are 0 / +-1: the error growth is that of the direct sum.
//
// Kernel = dcx_conv_wino2h.h with 9 positions instead of 16 and the phase as a work-item dimension: workgroup = 4 waves = 64
// couts x 32 2x2-tiles (8x16 low-resolution pixels of one phase); wave wm owns couts 16 wm .. 16 wm + 15 for all 32 tiles and
// all 9 positions: 9 x 2 x 4 = 72 accumulator registers (AGPRs).  v_mfma_f32_16x16x4_f32, operands and summation order as there:
//       m = 0;  for chunk c (16 cin) / j in 0..3 / g in 0..3:  m = fmaf(u[16c + 4g + j], v[16c + 4g + j], m)
//   output transform: y_k = 0; for p = 3 xi + nu in 0..8: y_k = fmaf(T[k][p], m[p], y_k) (zero coefficients included: nine chained
//   v_mfma_f32_4x4x1 per accumulator register), then max(fmaf(y, alpha, beta2), 0).  Restated bit-exactly by
//   oracle/conv_exact.c: dcx_oracle_conv_ups2w_exact.
//
// Per unit (16 channels) a wave issues 9 positions x 8 MFMAs = 2,304 matrix cycles; staging = raw [4 cq][9][17] tile (3 float4
// per thread), mid barrier, then each thread transforms one (cq, tile, channel PAIR): 9 ds_read_b64, 12 v_pk_add_f32,
// 9 ds_write_b64.  LDS 47 KB + per-channel constants.
#pragma once
#include "dcx_conv_wino2h.h"

#ifndef DCX_W2P_DQ
#define DCX_W2P_DQ 3
#endif
#ifndef DCX_W2P_DQB
#define DCX_W2P_DQB 1
#endif
#ifndef DCX_W2P_E_STORE
#define DCX_W2P_E_STORE 6
#endif
#ifndef DCX_W2P_E_XFORM
#define DCX_W2P_E_XFORM 10
#endif
#ifndef DCX_W2P_OCC
#define DCX_W2P_OCC 3
#endif
#ifdef DCX_W2P_NUM_VGPR
#define DCX_W2P_ATTR __attribute__((amdgpu_num_vgpr(DCX_W2P_NUM_VGPR)))
#else
#define DCX_W2P_ATTR
#endif

// G_ > 1: a work item covers G_ whole low-resolution maps of TH x TW pixels (RefineNet's 8x8 maps: two per item fill the 32 tiles)
template <int TH_, int TW_, int EPI_ = DCX_EPI_BNRELU, int G_ = 1>
struct DcxWino2pCfg {
    static constexpr int TH = TH_, TW = TW_;           // LOW-RESOLUTION pixels of one phase
    static constexpr int EPI = EPI_;
    static constexpr int G = G_;
    static constexpr int NTHREADS = 256;
    static constexpr int COUT_TILE = 64;
    static constexpr int TY = TH / 2, TX = TW / 2;
    static constexpr int TPI = TY * TX;                    // 2x2 tiles per image
    static constexpr int NTILES = G * TPI;                 // <= 32
    static constexpr int HH = TH + 1, RW = TW + 1;
    static constexpr int CQC = DCX_CCH / 4;
    static constexpr int NP = 9;
    static constexpr int RAW = G * CQC * HH * RW;         // [image][cq][row][col]
    static constexpr int ITER_R = (RAW + NTHREADS - 1) / NTHREADS;
    // raw tile in LDS: row pitch RP = RW + 1, row hy shifted by (hy >> 1) & 1 slots: the transform's ds_read_b64 of a 32-lane
    // group (16 tiles x 2 channel pairs: two tile rows) then covers 32 different 8-byte slots; slot RP - 1 of row 0 is free
    static constexpr int RP = RW + 1;
    static constexpr int RAW_LDS = G * CQC * HH * RP;
    static constexpr int VPLANE = CQC * 32;                // float4 per position: [cq][tile]
    static constexpr int LDS_FLOAT4 = NP * VPLANE;         // one transformed buffer (18 KB)
    static constexpr size_t LDS_BYTES = (size_t)(2 * LDS_FLOAT4 + RAW_LDS) * 16;
    static constexpr int DQ = DCX_W2P_DQ;                  // weights: positions ahead
    static constexpr int DQB = DCX_W2P_DQB;                // transformed activations: positions ahead
    // staging schedule in events (two per position: 18 per unit, 128 matrix cycles apart)
    static constexpr int E_RAW_LOAD = 0;
    static constexpr int E_RAW_STORE = DCX_W2P_E_STORE;
    static constexpr int E_XFORM = DCX_W2P_E_XFORM;        // mid barrier before this event; 6 transform events follow
    static_assert(TH % 2 == 0 && TW % 2 == 0 && NTILES <= 32 && NTILES > 16 && ((G == 1 && TX == 8) || (G == 2 && TX == 4 && TY == 4)),
                  "tile: 17..32 2x2 tiles: 8 per row, or two 8x8 maps (the LDS row shift is built for these two shapes)");
    static_assert(G == 1 || EPI == DCX_EPI_BNRELU, "grouped maps: plain BN + ReLU layers only");
    static_assert(ITER_R <= 3 && E_RAW_STORE + ITER_R <= E_XFORM && E_XFORM % 2 == 0 && E_XFORM + 6 <= 2 * NP, "staging does not fit the schedule");
    static_assert(EPI == DCX_EPI_BNRELU || EPI == DCX_EPI_HEAT, "BN + ReLU, optionally followed by the RefineNet head");
};

template <class C>
__global__ __launch_bounds__(256, DCX_W2P_OCC) DCX_W2P_ATTR void dcx_conv_wino2p_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int TX = C::TX, RW = C::RW, RP = C::RP, ITER_R = C::ITER_R, LDSF = C::LDS_FLOAT4, CQC = C::CQC, VPLANE = C::VPLANE;
    constexpr int DQ = C::DQ, DQB = C::DQB, NP = C::NP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = 16-cout group (and the channel quad it transforms)
    const int g4 = lane >> 4, l15 = lane & 15;

    // ---- work list (persistent, XCD-aware walk: see the header comment); the four phases of a tile are neighbours --------
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    int n_eff = a.n;
    if (a.n_limit != nullptr) n_eff = min(n_eff, *a.n_limit);
    const int total = ((n_eff + C::G - 1) / C::G) * n_ct * tiles * 4;      // G > 1: one work item covers G images (tiles == 1)
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
        it.ph = wi & 3; wi >>= 2;
        it.tx = wi % a.tiles_x; wi /= a.tiles_x;
        it.ty = wi % a.tiles_y; wi /= a.tiles_y;
        it.ct = wi % n_ct;
        it.n = wi / n_ct;
        return it;
    };

    // ---- operand fetch -----------------------------------------------------------------------------------------
    // weights [phase][pos][cin/4][cout_pad][4]: lane (r = l15, g = g4) reads cout wm*16 + r, channel quad g of the chunk
    const unsigned w_lane_off = (unsigned)(g4 * a.cout_pad + wm * 16 + l15) * 16u;
    const unsigned w_pos_stride = (unsigned)((a.cin >> 2) * a.cout_pad) * 16u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_ups2w), (short)0, (int)(4u * NP * w_pos_stride), 0x00020000);
    auto unit_wbase = [&](const DcxItem& it, int c) {
        return (unsigned)((c * CQC) * a.cout_pad + it.ct * C::COUT_TILE) * 16u + (unsigned)(it.ph * NP) * w_pos_stride;
    };
    auto load_a = [&](unsigned wbase, int pos) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane_off, wbase + (unsigned)pos * w_pos_stride, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    const int tile_b = g4 * 32 + l15;
    auto load_b = [&](int buf, int pos, int tb) { return sB[buf * LDSF + pos * VPLANE + tile_b + tb * 16]; };

    // ---- staging: raw tile [cq][hy][hx] of the LOW-RESOLUTION tensor, origin (ty TH - (1-a), tx TW - (1-b)) -------------
    auto rowoff = [](int row) { return (row >> 1) & 1; };
    int r_hyx = 0, r_slot[ITER_R];          // r_hyx: (img << 9 | hy << 5 | hx) of the thread's raw pixels, 10 bits each
    unsigned r_rel[ITER_R];
    const unsigned in_img_stride = (unsigned)a.in_cq_total * (unsigned)(a.hin * a.win);   // float4 between images
#pragma unroll
    for (int k = 0; k < ITER_R; ++k) {
        const int idx = tid + k * C::NTHREADS;
        const int img = idx / (CQC * C::HH * RW);
        const int rem = idx - img * (CQC * C::HH * RW);
        const int cq = rem / (C::HH * RW);
        const int hp = rem - cq * (C::HH * RW);
        const int hy = hp / RW, hx = hp - hy * RW;
        r_hyx |= (img << 9 | hy << 5 | hx) << (10 * k);
        r_rel[k] = idx < C::RAW ? ((unsigned)img * in_img_stride + (unsigned)((cq * a.hin + hy) * a.win + hx)) * 16u : 0x80000000u;
        r_slot[k] = idx < C::RAW ? ((img * CQC + cq) * C::HH + hy) * RP + hx + rowoff(hy) : RP - 1;
    }
    float4* sR = sB + 2 * LDSF;
    // transform piece of this thread: channel pair hb of (cq = wave, tile); tiles past the end redo the last tile
    const int x_hb = lane & 1;
    const int x_tile = min(lane >> 1, C::NTILES - 1);                           // lanes 0..31: tiles 0..15, lanes 32..63: tiles 16..31
    const int x_img = x_tile / C::TPI, x_t = x_tile - x_img * C::TPI;
    const int x_ty = x_t / TX, x_tx = x_t - x_ty * TX;
    const int x_src = ((x_img * CQC + wm) * C::HH + 2 * x_ty) * RP + 2 * x_tx;
    // Relaxed atomic 8-byte loads: hipcc otherwise fuses neighbouring plain reads into ds_read2_b64, which costs 8 LDS cycles per
    // pair instead of 2 + 2 and is banked modulo 32 in 16-lane groups -- the raw tile's row shift is built for ds_read_b64 (32-lane
    // groups, 64 banks).  Round 3's counters: 15-16 % of this kernel's LDS cycles were bank conflicts, all of them from the fused
    // reads (tools/lds_sim.py reproduces the counters of every conv kernel).  An atomic load is never fused and, unlike a volatile
    // one, keeps its LDS address space (the volatile form compiles to flat_load).
    const unsigned long long* sR2 = reinterpret_cast<const unsigned long long*>(sR);
    auto ld2 = [&](int idx) {
        const unsigned long long v = __hip_atomic_load(sR2 + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return dcx_f32x2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
    };
    const int x_r0 = 2 * (x_src + rowoff(2 * x_ty)) + x_hb, x_r1 = 2 * (x_src + RP + rowoff(2 * x_ty + 1)) + x_hb,
              x_r2 = 2 * (x_src + 2 * RP + rowoff(2 * x_ty + 2)) + x_hb;      // float2 index of d[r][0]; d[r][s] at + 2 s
    const int x_dst = 2 * (wm * 32 + x_tile) + x_hb;                            // float2 index; + 2 * pos * VPLANE
    auto pad_y = [&](const DcxItem& it) { return 1 - (it.ph >> 1); };
    auto pad_x = [&](const DcxItem& it) { return 1 - (it.ph & 1); };
    auto unit_rsrc = [&](const DcxItem& it, int c) {
        const long tile_off = (long)(it.ty * C::TH - pad_y(it)) * a.win + (it.tx * C::TW - pad_x(it));
        const float* base = a.in + (((size_t)it.n * C::G * a.in_cq_total + a.in_cq_off + (size_t)c * CQC) * (size_t)a.hin * a.win
                                    + tile_off) * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
    };
    auto tile_interior = [&](const DcxItem& it) {
        if (C::G > 1) return false;        // whole maps: every tile touches the zero border (and the last group may be short)
        const int sy0 = it.ty * C::TH - pad_y(it), sx0 = it.tx * C::TW - pad_x(it);
        return sy0 >= 0 && sx0 >= 0 && sy0 + C::HH <= a.hin && sx0 + RW <= a.win;
    };
    auto stage_fetch = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // Transform as 7 events (x = 0..6):
    //   0: read d[0][*], d[1][*]     1: read d[2][*]     2: t0 = d0 - d1, v[0][0], v[0][2]     3: write v[0][*]; v[1][0], v[1][2]
    //   4: write v[1][*]; t2 = d2 - d1, v[2][0], v[2][2]                5: write v[2][*]       (6: spare)
    dcx_f32x2 xd0[3], xd1[3], xd2[3], xv0, xv2;
    auto xform_event = [&](float4* vbuf, int x) {
        dcx_f32x2* v2 = reinterpret_cast<dcx_f32x2*>(vbuf);
        if (x == 0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) { xd0[s] = ld2(x_r0 + 2 * s); xd1[s] = ld2(x_r1 + 2 * s); }
        } else if (x == 1) {
#pragma unroll
            for (int s = 0; s < 3; ++s) xd2[s] = ld2(x_r2 + 2 * s);
        } else if (x == 2) {
#pragma unroll
            for (int s = 0; s < 3; ++s) xd0[s] = dcx_pk_sub(xd0[s], xd1[s]);            // t[0][s]
            xv0 = dcx_pk_sub(xd0[0], xd0[1]); xv2 = dcx_pk_sub(xd0[2], xd0[1]);
        } else if (x == 3) {
            v2[x_dst + 2 * 0 * VPLANE] = xv0; v2[x_dst + 2 * 1 * VPLANE] = xd0[1]; v2[x_dst + 2 * 2 * VPLANE] = xv2;
            xv0 = dcx_pk_sub(xd1[0], xd1[1]); xv2 = dcx_pk_sub(xd1[2], xd1[1]);
        } else if (x == 4) {
            v2[x_dst + 2 * 3 * VPLANE] = xv0; v2[x_dst + 2 * 4 * VPLANE] = xd1[1]; v2[x_dst + 2 * 5 * VPLANE] = xv2;
#pragma unroll
            for (int s = 0; s < 3; ++s) xd2[s] = dcx_pk_sub(xd2[s], xd1[s]);            // t[2][s]
            xv0 = dcx_pk_sub(xd2[0], xd2[1]); xv2 = dcx_pk_sub(xd2[2], xd2[1]);
        } else if (x == 5) {
            v2[x_dst + 2 * 6 * VPLANE] = xv0; v2[x_dst + 2 * 7 * VPLANE] = xd2[1]; v2[x_dst + 2 * 8 * VPLANE] = xv2;
        }
    };

    // ---- epilogue constants in LDS: alpha, beta2, (head weights,) output-transform table ---------------------------
    float4* sP = sB + 2 * LDSF + C::RAW_LDS;
    const int cq_pad = a.cout_pad >> 2;
    constexpr int NPAR = C::EPI == DCX_EPI_HEAT ? 3 : 2;
    for (int i = tid; i < cq_pad; i += C::NTHREADS) {
        sP[i] = reinterpret_cast<const float4*>(a.alpha)[i];
        sP[cq_pad + i] = reinterpret_cast<const float4*>(a.beta)[i];
        if (C::EPI == DCX_EPI_HEAT) sP[2 * cq_pad + i] = reinterpret_cast<const float4*>(a.head_w)[i];
    }
    float* sT = reinterpret_cast<float*>(sP + NPAR * cq_pad);       // T[k][p], 16 floats per k (p = 9..15 unused)
    if (tid < 64) {
        const int k = tid >> 4, p = tid & 15, i = k >> 1, j = k & 1, xi = p / 3, nu = p - 3 * xi;
        const int ci = i == 0 ? (xi < 2 ? 1 : 0) : (xi > 0 ? 1 : 0);
        const int cj = j == 0 ? (nu < 2 ? 1 : 0) : (nu > 0 ? 1 : 0);
        sT[tid] = p < 9 ? (float)(ci * cj) : 0.f;
    }

    dcx_f32x4 acc[NP][2];       // acc[pos][tb], only ever defined by inline asm with an AGPR constraint (see the header comment)

    // ---- prologue: first unit staged synchronously ---------------------------------------------------------------
    DcxItem cur = decode(w);
    int c = 0;
    float4 a_c[DQ];
    {
        const unsigned wb = unit_wbase(cur, 0);
#pragma unroll
        for (int d = 0; d < DQ; ++d) a_c[d] = load_a(wb, d);
        const __amdgpu_buffer_rsrc_t r0 = unit_rsrc(cur, 0);
        const int sy0 = cur.ty * C::TH - pad_y(cur), sx0 = cur.tx * C::TW - pad_x(cur);
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) {
            const int ly = sy0 + ((r_hyx >> (10 * k + 5)) & 15), lx = sx0 + ((r_hyx >> (10 * k)) & 31);
            const bool inb = (unsigned)ly < (unsigned)a.hin && (unsigned)lx < (unsigned)a.win
                          && (C::G == 1 || cur.n * C::G + ((r_hyx >> (10 * k + 9)) & 1) < n_eff);
            sR[r_slot[k]] = stage_fetch(r0, inb ? r_rel[k] : 0x80000000u);
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 6; ++x) xform_event(sB, x);
    }

    int u = 0;
    auto run_unit = [&](auto zero_t) -> bool {
        constexpr bool ZERO = decltype(zero_t)::value;
        DcxItem nxt = cur;
        int cn = c + 1;
        bool has_next = true;
        if (cn == nch) {
            if (w + gstride < w_end) { nxt = decode(w + gstride); cn = 0; }
            else { has_next = false; cn = c; }
        }
        const int buf = u & 1;
#ifdef DCX_W2P_PROBES      // per-unit time stamps (tools/unit_probe.py); off by default: they cost SGPRs this kernel does not have to spare
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && (unsigned)(u - a.probe_u0) < 20u) a.clk_probe[4 + 3 * (u - a.probe_u0)] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef DCX_W2P_PROBES
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && (unsigned)(u - a.probe_u0) < 20u) a.clk_probe[5 + 3 * (u - a.probe_u0)] = __builtin_amdgcn_s_memtime();
#endif

        const __amdgpu_buffer_rsrc_t rs_n = unit_rsrc(nxt, cn);
        const int nsy0 = nxt.ty * C::TH - pad_y(nxt), nsx0 = nxt.tx * C::TW - pad_x(nxt);
        const unsigned wb_cur = unit_wbase(cur, c);
        const unsigned wb_nxt = unit_wbase(nxt, cn);
        float4 aq[NP + DQ], bq[NP][2];
#pragma unroll
        for (int d = 0; d < DQ; ++d) aq[d] = a_c[d];
#pragma unroll
        for (int d = 0; d < DQB; ++d) { bq[d][0] = load_b(buf, d, 0); bq[d][1] = load_b(buf, d, 1); }
        float4* vnext = sB + (buf ^ 1) * LDSF;
        float4 rv[ITER_R];
        const bool n_interior = tile_interior(nxt);
        unsigned roff[ITER_R];
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) roff[k] = r_rel[k];
        if (!n_interior) {
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) {
                const int ly = nsy0 + ((r_hyx >> (10 * k + 5)) & 15), lx = nsx0 + ((r_hyx >> (10 * k)) & 31);
                const bool inb = (unsigned)ly < (unsigned)a.hin && (unsigned)lx < (unsigned)a.win
                              && (C::G == 1 || nxt.n * C::G + ((r_hyx >> (10 * k + 9)) & 1) < n_eff);
                roff[k] = inb ? r_rel[k] : 0x80000000u;
            }
        }
        // one position = 8 MFMAs (tb 0 / 1 alternating, j = 0..3) in two slots of four; each slot is preceded by one staging
        // event; slot 0 also fetches the operands of position p + DQ / p + DQB
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p * 2 == C::E_XFORM) __syncthreads();       // the raw tile of the next unit is complete in sR
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                __builtin_amdgcn_sched_barrier(0);
                if (slot == 0) {
                    const int q = p + DQ, qb2 = p + DQB;
                    if (q < NP) aq[q] = load_a(wb_cur, q);
                    else aq[q] = load_a(wb_nxt, q - NP);
                    if (qb2 < NP) { bq[qb2][0] = load_b(buf, qb2, 0); bq[qb2][1] = load_b(buf, qb2, 1); }
                }
                {
                    const int e = p * 2 + slot;          // staging event 0 .. 17
                    if (e >= C::E_RAW_LOAD && e < C::E_RAW_LOAD + ITER_R) rv[e - C::E_RAW_LOAD] = stage_fetch(rs_n, roff[e - C::E_RAW_LOAD]);
                    if (e >= C::E_RAW_STORE && e < C::E_RAW_STORE + ITER_R) sR[r_slot[e - C::E_RAW_STORE]] = rv[e - C::E_RAW_STORE];
                    if (e >= C::E_XFORM && e < C::E_XFORM + 6) xform_event(vnext, e - C::E_XFORM);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const float4 aa = aq[p], b0 = bq[p][0], b1 = bq[p][1];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * slot + jj;
                        const float av = j == 0 ? aa.x : j == 1 ? aa.y : j == 2 ? aa.z : aa.w;
                        const float bv0 = j == 0 ? b0.x : j == 1 ? b0.y : j == 2 ? b0.z : b0.w;
                        const float bv1 = j == 0 ? b1.x : j == 1 ? b1.y : j == 2 ? b1.z : b1.w;
                        if (p == 0 && j == 0) asm volatile("s_nop 1");      // hazards: see dcx_conv_wino2h.h
                        if (ZERO && j == 0) {       // first touch of these two accumulators in this work item: C = 0
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[p][0]) : "v"(av), "v"(bv0));
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[p][1]) : "v"(av), "v"(bv1));
                        } else {
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p][0]) : "v"(av), "v"(bv0));
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p][1]) : "v"(av), "v"(bv1));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < DQ; ++d) a_c[d] = aq[NP + d];

#ifdef DCX_W2P_PROBES
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && (unsigned)(u - a.probe_u0) < 20u) a.clk_probe[6 + 3 * (u - a.probe_u0)] = __builtin_amdgcn_s_memtime();
#endif
        if (c == nch - 1 && (!ZERO || nch == 1)) {
            // ---- epilogue: output transform on the matrix cores, BN, ReLU, store at stride 2 (or the head) -----------------
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
            for (int p = 0; p < NP; ++p) { asm volatile("" : "+a"(acc[p][0])); asm volatile("" : "+a"(acc[p][1])); }
            const int pa = cur.ph >> 1, pb = cur.ph & 1;
            const unsigned plane = (unsigned)(a.ho * a.wo);
            int lne = lane;
            asm volatile("" : "+v"(lne));   // opaque copy: the epilogue's lane arithmetic must not be hoisted out of the unit loop (VGPR budget)
            const int g4e = lne >> 4, l15e = lne & 15;
            const int cq = (cur.ct * C::COUT_TILE >> 2) + wm * 4 + g4e;         // the lane's output channel quad
            char* obase = reinterpret_cast<char*>(a.out)
                        + ((size_t)cur.n * C::G * a.out_cq_total + a.out_cq_off + cq) * (size_t)plane * 16;
            float cf[NP];
            {
                const float4* tp = reinterpret_cast<const float4*>(sT + (lne & 3) * 16);
                const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
                cf[0] = t0.x; cf[1] = t0.y; cf[2] = t0.z; cf[3] = t0.w; cf[4] = t1.x; cf[5] = t1.y; cf[6] = t1.z; cf[7] = t1.w; cf[8] = t2.x;
            }
            const float4 al = sP[cq], be = sP[cq_pad + cq];
            const dcx_f32x2 al01 = {al.x, al.y}, al23 = {al.z, al.w}, be01 = {be.x, be.y}, be23 = {be.z, be.w};
            float hsum[2][4];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                // one tile block at a time (four interleaved chains: dependent accumulations stay >= 4 instructions apart)
                dcx_f32x4 e[2][4];        // e[tb][i][k]: output k of cout 4 * cq + i for the lane's tile tb * 16 + l15
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (p == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(e[tb][i]) : "v"(cf[0]), "a"(acc[0][tb][i]));
                        else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(e[tb][i]) : "v"(cf[p]), "a"(acc[p][tb][i]));
                    }
                }
                asm volatile("s_nop 7" : "+v"(e[tb][0]), "+v"(e[tb][1]), "+v"(e[tb][2]), "+v"(e[tb][3]));
                __builtin_amdgcn_sched_barrier(0);
                const int qt = tb * 16 + l15e;
                const int q_img = qt / C::TPI, q_t = qt - q_img * C::TPI;     // image inside the group (0 when G == 1)
                const int qty = q_t / TX, qtx = q_t - qty * TX;
                const int ly0 = cur.ty * C::TH + 2 * qty, lx0 = cur.tx * C::TW + 2 * qtx;     // low-resolution position of output k = 0
                const bool qok = qt < C::NTILES && cq < a.cout_quads && (C::G == 1 || cur.n * C::G + q_img < n_eff);
                const bool okr0 = qok && ly0 < a.hin, okr1 = qok && ly0 + 1 < a.hin;
                const bool okc0 = lx0 < a.win, okc1 = lx0 + 1 < a.win;
                if (C::EPI == DCX_EPI_HEAT) {
                    // head: h[k] = sum over the lane's four couts of relu(bn(e)) * head_w, an fmaf chain in cout order; nothing is stored
                    const float4 hw = cq < a.cout_quads ? sP[2 * cq_pad + cq] : make_float4(0.f, 0.f, 0.f, 0.f);
                    float h[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ai = i == 0 ? al.x : i == 1 ? al.y : i == 2 ? al.z : al.w;
                        const float bi = i == 0 ? be.x : i == 1 ? be.y : i == 2 ? be.z : be.w;
                        const float wi = i == 0 ? hw.x : i == 1 ? hw.y : i == 2 ? hw.z : hw.w;
#pragma unroll
                        for (int k = 0; k < 4; ++k) h[k] = fmaf(dcx_vmax(fmaf(e[tb][i][k], ai, bi), 0.f), wi, h[k]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) hsum[tb][k] = h[k];
                    continue;
                }
                dcx_f32x2 bn[4][2];     // [i][k / 2]
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kp = 0; kp < 2; ++kp) {
                        const dcx_f32x2 x = {e[tb][i][2 * kp], e[tb][i][2 * kp + 1]};
                        const dcx_f32x2 aa = i < 2 ? al01 : al23, bb = i < 2 ? be01 : be23;
                        if ((i & 1) == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(bn[i][kp]) : "v"(x), "v"(aa), "v"(bb));
                        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(bn[i][kp]) : "v"(x), "v"(aa), "v"(bb));
                    }
                float4 y[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    y[k] = make_float4(bn[0][k >> 1][k & 1], bn[1][k >> 1][k & 1], bn[2][k >> 1][k & 1], bn[3][k >> 1][k & 1]);
                    y[k].x = dcx_vmax(y[k].x, 0.f); y[k].y = dcx_vmax(y[k].y, 0.f);
                    y[k].z = dcx_vmax(y[k].z, 0.f); y[k].w = dcx_vmax(y[k].w, 0.f);
                }
                {
                    // output k = (i, j) of the tile is low-resolution pixel (ly0 + i, lx0 + j) -> pixel (2 (ly0 + i) + a, 2 (lx0 + j) + b)
                    char* dst = obase + (size_t)((unsigned)((2 * ly0 + pa) * a.wo + 2 * lx0 + pb) * 16u)
                              + (C::G > 1 ? (size_t)q_img * a.out_cq_total * plane * 16 : 0);
                    if (okr0 && okc0) *reinterpret_cast<float4*>(dst) = y[0];
                    if (okr0 && okc1) *reinterpret_cast<float4*>(dst + 32) = y[1];
                    if (okr1 && okc0) *reinterpret_cast<float4*>(dst + (size_t)a.wo * 32) = y[2];
                    if (okr1 && okc1) *reinterpret_cast<float4*>(dst + (size_t)a.wo * 32 + 32) = y[3];
                }
            }
            if (C::EPI == DCX_EPI_HEAT) {
                // logit = ((w0 + w1) + (w2 + w3)) + bias with w = the wave's sum over its four channel quads:
                // ((q0 + q1) + (q2 + q3)), each quad an fmaf chain over its four couts
                float t[2][4];
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v = hsum[tb][k] + __shfl_xor(hsum[tb][k], 16);
                        v = v + __shfl_xor(v, 32);
                        t[tb][k] = v;
                    }
                __syncthreads();                                   // every wave is done reading this unit's operands
                float* scr = reinterpret_cast<float*>(sB + buf * LDSF);    // [wave][tile 32][k 4]
                {
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    if ((ln >> 4) == 0) {
#pragma unroll
                        for (int tb = 0; tb < 2; ++tb)
                            *reinterpret_cast<float4*>(scr + ((wm * 32 + tb * 16 + ln) * 4)) = make_float4(t[tb][0], t[tb][1], t[tb][2], t[tb][3]);
                    }
                }
                __syncthreads();
                float best = -INFINITY;
                int besti = 0x7fffffff;
                if (wm == 0) {
                    int ln = lane;
                    asm volatile("" : "+v"(ln));       // opaque: keeps hipcc from hoisting this lane arithmetic out of the unit loop (VGPRs)
                    const int qt = ln & 31, kh = ln >> 5;            // lane: tile qt, outputs k = 2 kh, 2 kh + 1 (row i = kh)
                    const int qty = qt / TX, qtx = qt - qty * TX;
                    const int ly = cur.ty * C::TH + 2 * qty + kh, lx0 = cur.tx * C::TW + 2 * qtx;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int k = 2 * kh + jj;
                        const float lg = ((scr[(0 * 32 + qt) * 4 + k] + scr[(1 * 32 + qt) * 4 + k])
                                        + (scr[(2 * 32 + qt) * 4 + k] + scr[(3 * 32 + qt) * 4 + k])) + a.head_b;
                        const int lx = lx0 + jj;
                        if (qt < C::NTILES && ly < a.hin && lx < a.win) {
                            const int sy = 2 * ly + pa, sx = 2 * lx + pb;
                            const int idx = sy * a.wo + sx;
                            if (a.heat != nullptr) a.heat[((size_t)cur.n * a.ho + sy) * a.wo + sx] = lg;
                            if (lg > best || (lg == best && idx < besti)) { best = lg; besti = idx; }
                        }
                    }
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        const float ov = __shfl_xor(best, off);
                        const int oi = __shfl_xor(besti, off);
                        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                    }
                    if (lane == 0) {
                        const size_t slot = (size_t)cur.n * (tiles * 4) + (size_t)cur.ph * tiles + cur.ty * a.tiles_x + cur.tx;
                        a.part_val[slot] = best;
                        a.part_idx[slot] = besti;
                    }
                }
            }
        }

        if (!has_next) {
            if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0) {
                a.clk_probe[2] = __builtin_amdgcn_s_memtime();
                a.clk_probe[3] = __builtin_amdgcn_s_memrealtime();
            }
            return false;
        }
        if (cn == 0) w += gstride;
        cur = nxt;
        c = cn;
        ++u;
        return true;
    };
    // work items: first unit (C = 0), then the remaining nch - 1 units (the last one runs the epilogue); nch >= 2
    for (;;) {
        run_unit(std::true_type{});
        bool more = true;
        while (c != 0 && more) more = run_unit(std::false_type{});
        if (!more) break;
    }
}

template <class C>
static int dcx_conv_wino2p_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    // tiles are cut in the LOW-RESOLUTION image the kernel reads; ho x wo stays the (x2) output size
    a.tiles_x = (a.win + C::TW - 1) / C::TW;
    a.tiles_y = (a.hin + C::TH - 1) / C::TH;
    if (a.w_ups2w == nullptr || a.alpha == nullptr || a.beta == nullptr) return DCX_E_ARG;
    if (C::EPI == DCX_EPI_HEAT ? (a.head_w == nullptr || a.part_val == nullptr || a.part_idx == nullptr) : a.out == nullptr) return DCX_E_ARG;
    if (a.ups != 1 || a.pad != 1 || a.ho != 2 * a.hin || a.wo != 2 * a.win) return DCX_E_SHAPE;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0 || a.cin < 2 * DCX_CCH) return DCX_E_SHAPE;   // >= 2 units per work item
    if (C::EPI == DCX_EPI_HEAT && a.cout_pad != C::COUT_TILE) return DCX_E_SHAPE;     // the head sums over ONE cout tile
    if (C::G > 1 && (a.tiles_x != 1 || a.tiles_y != 1)) return DCX_E_SHAPE;                          // grouped: whole maps only
    const long items = (long)((a.n + C::G - 1) / C::G) * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y * 4;
    if (items <= 0 || items > 0x7fffffffL) return DCX_E_SHAPE;
    const size_t lds = C::LDS_BYTES + (size_t)a.cout_pad * (C::EPI == DCX_EPI_HEAT ? 12 : 8) + 256;
    if (DCX_W2P_OCC * lds > 160 * 1024) return DCX_E_SHAPE;
    const int occ_env = dcx_occupancy_override();
    const long resident = (long)(occ_env >= 1 && occ_env < DCX_W2P_OCC ? occ_env : DCX_W2P_OCC) * dcx_device_cu_count();
    const long blocks = items < resident ? items : resident;
    a.xcd_walk = dcx_xcd_walk_enabled() && blocks == resident && (resident & 7) == 0 ? 1 : 0;
    dcx_fill_xcd_cum(a);
    static bool attr_set[DCX_MAX_DEVICES] = {};
    const int dev_i = dcx_current_device();
    if (!attr_set[dev_i]) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_wino2p_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_set[dev_i] = true;
    }
    hipLaunchKernelGGL((dcx_conv_wino2p_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), lds, stream, a);
    return (int)hipGetLastError();
}
