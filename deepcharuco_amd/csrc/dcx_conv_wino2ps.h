// dcx_conv_wino2ps.h -- the phase x Winograd F(2x2,2x2) convolution of dcx_conv_wino2p.h (3x3 + BN + ReLU over a nearest-x2 up-sampled
// input) for launches that CANNOT fill the chip: RefineNet's conv4a / conv5a on the ~16 patches of ONE frame (the reference's own bs=1
// protocol, src/benchmark.py:37-53).  Same family, same summation orders, same bits (oracle/conv_exact.c:
// dcx_oracle_conv_ups2w_exact; tests: test_every_conv_instantiation_bit_exact) -- the shape of dcx_conv_wino2hs.h:
//
//   an item = (patch, cout tile of 16 CG, 8x8 low-resolution pixels = 16 2x2-tiles, phase); its 9 Winograd positions are SPLIT OVER
//   WAVES: workgroup = 3 position groups (xi = 0..2: positions 3 xi .. 3 xi + 2) x CG cout groups of 16 = 3 CG waves.  A wave's
//   chain per 16-channel unit is 3 positions x 4 MFMAs = 384 matrix cycles where dcx_conv_wino2p.h's wave issues 9 x 8 (2,304);
//   conv4a of 16 patches becomes 256 items of 32 couts (one per CU) instead of 64 items of 64 couts x two maps.
//   Raw tiles and weights are requested U - 1 units ahead (the operands of a bs=1 call come from the MALL, not from a warm L2).
//   The accumulators go through LDS once per item and the output transform y[k] = sum_p T[k][p] m[p], p = 0..8 ascending, zero
//   coefficients included, runs on the vector ALU as the same fmaf chain dcx_conv_wino2p.h runs on v_mfma_f32_4x4x1; then
//   max(fmaf(y, alpha, beta2), 0), stored at stride 2 (phase (a, b): low-resolution pixel (y, x) -> output pixel (2 y + a, 2 x + b)).
//   BN + ReLU layers only (the fused RefineNet head fills the chip at 16 patches and stays on dcx_conv_wino2p.h).
#pragma once
#include "dcx_conv_wino2p.h"

template <int CG_>
struct DcxWino2psCfg {
    static constexpr int CG = CG_;                          // cout groups (of 16) per workgroup
    static constexpr int TH = 8, TW = 8, TX = 4, NTILES = 16;   // LOW-RESOLUTION pixels of one phase
    static constexpr int HH = TH + 1, RW = TW + 1;
    static constexpr int NP = 9;
    static constexpr int NWAVES = 3 * CG, NTHREADS = 64 * NWAVES;
    static constexpr int COUT_TILE = 16 * CG;
    static constexpr int CQC = DCX_CCH / 4;
    static constexpr int RAW = CQC * HH * RW;               // 324 float4 per unit
    static constexpr int ITER_R = (RAW + NTHREADS - 1) / NTHREADS;
    static constexpr int RP = 12;                           // raw tile in LDS: the 8x8 layout of dcx_conv_wino2hs.h (conflict-free 16-lane groups)
    __host__ __device__ static constexpr int raw_slot(int cq, int hy, int hx) { return (cq * HH + hy) * RP + hx + ((hy >> 2) & 1); }
    static constexpr int RAW_LDS = CQC * HH * RP;
    static constexpr int VPLANE = CQC * 16;                 // float4 per position: [cq][tile]
    static constexpr int LDS_V = NP * VPLANE;               // one transformed buffer (9 KB)
    static constexpr int LDS_X = NP * (4 * CG) * 16;        // accumulator exchange: [pos][cout quad][tile] float4
    static constexpr size_t LDS_BYTES = (size_t)(2 * LDS_V + RAW_LDS + LDS_X) * 16;
#ifndef DCX_W2PS_U
#define DCX_W2PS_U 4
#endif
    static constexpr int U = DCX_W2PS_U;                    // units in flight (2..4)
    static_assert(CG == 1 || CG == 2 || CG == 4, "1, 2 or 4 cout groups");
    static_assert(U >= 2 && U <= 4, "ring depth");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, 1) void dcx_conv_wino2ps_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int CQC = C::CQC, RW = C::RW, HH = C::HH, ITER_R = C::ITER_R, LDSV = C::LDS_V, VPLANE = C::VPLANE, CG = C::CG, NP = C::NP;
    float4* const sR = sB + 2 * LDSV;
    float4* const sX = sR + C::RAW_LDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pgp = wv % 3;                                         // the wave's position group = Winograd row xi
    const int cg = wv / 3;                                          // the wave's cout group
    const int g4 = lane >> 4, l15 = lane & 15;

    // ---- work list (dcx_conv_wino2p.h: the four phases of a tile are neighbours) ---------------------------------------
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    int n_eff = a.n;
    if (a.n_limit != nullptr) n_eff = min(n_eff, *a.n_limit);
    const int total = n_eff * n_ct * tiles * 4;
    int w = blockIdx.x, w_end = total, gstride = gridDim.x;
    if (a.xcd_walk && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7;
        const int lo = dcx_xcd_bound(total, x, a.xcd_cum[x]);
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

    // ---- operands ----------------------------------------------------------------------------------------------------
    // weights [phase][pos][cin/4][cout_pad][4]: lane (r = l15, g = g4) reads cout ct * COUT_TILE + cg * 16 + r, channel quad g of the chunk
    const unsigned w_lane_off = (unsigned)(g4 * a.cout_pad + cg * 16 + l15) * 16u;
    const unsigned w_pos_stride = (unsigned)((a.cin >> 2) * a.cout_pad) * 16u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_ups2w), (short)0, (int)(4u * NP * w_pos_stride), 0x00020000);
    auto load_a = [&](unsigned wbase, int pp) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane_off, wbase + (unsigned)pp * w_pos_stride, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // transformed activations sV[buf][pos][cq][tile]: lane (n = l15, g = g4) reads tile n, channel quad g
    auto load_b = [&](int buf, int pp) { return sB[buf * LDSV + (3 * pgp + pp) * VPLANE + lane]; };

    // ---- staging: raw tile [cq][hy][hx] of the LOW-RESOLUTION tensor, origin (ty TH - (1 - a), tx TW - (1 - b)) ---------------
    int r_hyx[ITER_R], r_slot[ITER_R];
    unsigned r_rel[ITER_R];
#pragma unroll
    for (int k = 0; k < ITER_R; ++k) {
        const int idx = tid + k * C::NTHREADS;
        const int cq = idx / (HH * RW);
        const int hp = idx - cq * (HH * RW);
        const int hy = hp / RW, hx = hp - hy * RW;
        r_hyx[k] = hy << 16 | hx;
        r_rel[k] = idx < C::RAW ? (unsigned)((cq * a.hin + hy) * a.win + hx) * 16u : 0x80000000u;
        r_slot[k] = idx < C::RAW ? C::raw_slot(cq, hy, hx) : C::RP - 1;      // slot RP - 1 of row 0 is free: dump slot
    }
    auto stage_fetch = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };

    // ---- staging: input transform.  Threads 0 .. 191 = (tile, cq, xi): t[s] = d[0][s] - d[1][s] | d[1][s] | d[2][s] - d[1][s], then
    // v[xi][0] = t[0] - t[1], v[xi][1] = t[1], v[xi][2] = t[2] - t[1] -- the exact fp32 differences dcx_conv_wino2p.h forms ------------
    const bool x_on = tid < 192;
    const int x_tile = tid & 15, x_cq = (tid >> 4) & 3;
    const int x_xi = __builtin_amdgcn_readfirstlane(min(tid >> 6, 2));
    const int x_ty = x_tile >> 2, x_tx = x_tile & 3;
    const int x_ra = C::raw_slot(x_cq, 2 * x_ty + (x_xi == 2 ? 2 : x_xi == 1 ? 1 : 0), 2 * x_tx);     // row A: d0 | d1 | d2
    const int x_rb = C::raw_slot(x_cq, 2 * x_ty + 1, 2 * x_tx);                                        // row B: d1
    const int x_dst = (3 * x_xi) * VPLANE + x_cq * 16 + x_tile;       // + nu * VPLANE
    auto sub4 = [](const float4& x, const float4& y) {
        const dcx_f32x2 lo = dcx_pk_sub(dcx_f32x2{x.x, x.y}, dcx_f32x2{y.x, y.y}), hi = dcx_pk_sub(dcx_f32x2{x.z, x.w}, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    float4 xa[3], xb[3];
    auto xform_read = [&]() {
        if (x_on) {
#pragma unroll
            for (int s = 0; s < 3; ++s) { xa[s] = sR[x_ra + s]; if (x_xi != 1) xb[s] = sR[x_rb + s]; }
        }
    };
    auto xform_write = [&](float4* vbuf) {
        if (x_on) {
            float4 t[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) t[s] = x_xi != 1 ? sub4(xa[s], xb[s]) : xa[s];
            vbuf[x_dst] = sub4(t[0], t[1]);
            vbuf[x_dst + VPLANE] = t[1];
            vbuf[x_dst + 2 * VPLANE] = sub4(t[2], t[1]);
        }
    };

    // ---- epilogue roles: threads 0 .. 128 CG - 1 = (cout quad of the item, tile, output row i of the 2x2 tile) -----------------
    const bool e_on = tid < 128 * CG;
    const int e_i = tid & 1, e_tile = (tid >> 1) & 15, e_cq = tid >> 5;

    dcx_f32x4 acc[3];
    constexpr int U = C::U;
    float4 aq[U][3];                              // weights of unit u sit in aq[u % U]
    float4 rq[U][ITER_R];                         // raw float4 of unit u sit in rq[u % U] until they are stored to LDS during unit u - 1

    // ---- work items: prologue (first unit staged synchronously), nch units, epilogue; nothing is carried from one item to the next
    // (these launches give every item a CU of its own) -------------------------------------------------------------------------
    for (; w < w_end; w += gstride) {
        const DcxItem cur = decode(w);
        const int pa = cur.ph >> 1, pb = cur.ph & 1;
        const int sy0 = cur.ty * C::TH - (1 - pa), sx0 = cur.tx * C::TW - (1 - pb);
        unsigned roff[ITER_R];
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) {
            const int ly = sy0 + (r_hyx[k] >> 16), lx = sx0 + (r_hyx[k] & 0xffff);
            roff[k] = ((unsigned)ly < (unsigned)a.hin && (unsigned)lx < (unsigned)a.win) ? r_rel[k] : 0x80000000u;
        }
        auto unit_rsrc = [&](int c) {
            const long tile_off = (long)sy0 * a.win + sx0;
            const float* base = a.in + (((size_t)cur.n * a.in_cq_total + a.in_cq_off + (size_t)c * CQC) * (size_t)a.hin * a.win + tile_off) * 4;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
        };
        const unsigned wb0 = (unsigned)(cur.ct * C::COUT_TILE) * 16u + (unsigned)(cur.ph * NP + 3 * pgp) * w_pos_stride;
        const unsigned w_unit = (unsigned)(CQC * a.cout_pad) * 16u;        // bytes between the weights of consecutive units
        {
            float4 r0[ITER_R];
            const __amdgpu_buffer_rsrc_t rs0 = unit_rsrc(0);
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) r0[k] = stage_fetch(rs0, roff[k]);
#pragma unroll
            for (int i = 0; i < U - 1; ++i) {
                const __amdgpu_buffer_rsrc_t rs = unit_rsrc(i + 1 < nch ? i + 1 : nch - 1);
#pragma unroll
                for (int k = 0; k < ITER_R; ++k) rq[(i + 1) % U][k] = stage_fetch(rs, roff[k]);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) aq[i][pp] = load_a(wb0 + (unsigned)(i < nch ? i : nch - 1) * w_unit, pp);
            }
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) sR[r_slot[k]] = r0[k];
            __syncthreads();
            xform_read();
            xform_write(sB);
        }
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
                                                    : "=v"(acc[pp][0]), "=v"(acc[pp][1]), "=v"(acc[pp][2]), "=v"(acc[pp][3]));

        auto run_unit = [&](auto slot_t, int c) {
            constexpr int K = decltype(slot_t)::value;          // c % U
            const bool has_next = c + 1 < nch;
            const int buf = c & 1;
            __syncthreads();                     // transformed tile of this unit complete; raw tile free
            if (has_next) {
#pragma unroll
                for (int k = 0; k < ITER_R; ++k) sR[r_slot[k]] = rq[(K + 1) % U][k];
            }
            {   // requests for unit c + U - 1 (weights) / c + U (raw tile): clamped to the item's last unit (a harmless repeat)
                const int cw = c + U - 1 < nch ? c + U - 1 : nch - 1;
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) aq[(K + U - 1) % U][pp] = load_a(wb0 + (unsigned)cw * w_unit, pp);
                const __amdgpu_buffer_rsrc_t rs = unit_rsrc(c + U < nch ? c + U : nch - 1);
#pragma unroll
                for (int k = 0; k < ITER_R; ++k) rq[K][k] = stage_fetch(rs, roff[k]);      // (the slot of this unit's own raw tile: stored a unit ago)
            }
            float4 bq[3];
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) bq[pp] = load_b(buf, pp);
            float4* vnext = sB + (buf ^ 1) * LDSV;
            // MFMA j of position pp consumes component j of both operands; per accumulator the order is j = 0..3 (the family's
            // order); consecutive MFMAs of the wave rotate through the three accumulators
#define DCX_W2PS_MFMA(J0, J1)                                                                                               \
            _Pragma("unroll") for (int j = (J0); j < (J1); ++j) {                                                           \
                _Pragma("unroll") for (int pp = 0; pp < 3; ++pp) {                                                          \
                    const float4 aa = aq[K][pp], bb = bq[pp];                                                               \
                    const float av = j == 0 ? aa.x : j == 1 ? aa.y : j == 2 ? aa.z : aa.w;                                  \
                    const float bv = j == 0 ? bb.x : j == 1 ? bb.y : j == 2 ? bb.z : bb.w;                                  \
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[pp]) : "v"(av), "v"(bv));               \
                }                                                                                                           \
            }
            asm volatile("s_nop 1");
            DCX_W2PS_MFMA(0, 2)
            __syncthreads();                         // raw tile of the next unit complete
            if (has_next) xform_read();
            DCX_W2PS_MFMA(2, 4)
            if (has_next) xform_write(vnext);
        };
        for (int c0 = 0; c0 < nch; c0 += U) {
            run_unit(std::integral_constant<int, 0>{}, c0);
            if (U > 1 && c0 + 1 < nch) run_unit(std::integral_constant<int, 1 % U>{}, c0 + 1);
            if (U > 2 && c0 + 2 < nch) run_unit(std::integral_constant<int, 2 % U>{}, c0 + 2);
            if (U > 3 && c0 + 3 < nch) run_unit(std::integral_constant<int, 3 % U>{}, c0 + 3);
        }

        // ---- epilogue: accumulators -> LDS, output transform + BN + ReLU on the vector ALU, store at stride 2 ---------------------
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // MFMA result -> read by a non-MFMA instruction
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            asm volatile("" : "+v"(acc[pp]));
            sX[((3 * pgp + pp) * (4 * CG) + cg * 4 + g4) * 16 + l15] = make_float4(acc[pp][0], acc[pp][1], acc[pp][2], acc[pp][3]);
        }
        __syncthreads();
        if (e_on) {
            const int cq = (cur.ct * C::COUT_TILE >> 2) + e_cq;                // the thread's output channel quad
            const float4 al = reinterpret_cast<const float4*>(a.alpha)[cq], be = reinterpret_cast<const float4*>(a.beta)[cq];
            // T[k = 2 i + j][p = 3 xi + nu] = AT[i][xi] * AT[j][nu], AT = [[1, 1, 0], [0, 1, 1]]
            dcx_f32x2 y0a, y0b, y1a, y1b;        // output j = 0 / 1 of row i: couts (0, 1) and (2, 3) of the quad
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int xi = p / 3, nu = p - 3 * xi;
                const float4 m = sX[(p * (4 * CG) + e_cq) * 16 + e_tile];
                const float ci = e_i == 0 ? (xi < 2 ? 1.f : 0.f) : (xi > 0 ? 1.f : 0.f);
                const float t0 = ci * (nu < 2 ? 1.f : 0.f), t1 = ci * (nu > 0 ? 1.f : 0.f);
                const dcx_f32x2 tt0 = {t0, t0}, tt1 = {t1, t1}, mlo = {m.x, m.y}, mhi = {m.z, m.w}, z = {0.f, 0.f};
                y0a = __builtin_elementwise_fma(tt0, mlo, p == 0 ? z : y0a);
                y0b = __builtin_elementwise_fma(tt0, mhi, p == 0 ? z : y0b);
                y1a = __builtin_elementwise_fma(tt1, mlo, p == 0 ? z : y1a);
                y1b = __builtin_elementwise_fma(tt1, mhi, p == 0 ? z : y1b);
            }
            float4 y0 = dcx_fma4(make_float4(y0a.x, y0a.y, y0b.x, y0b.y), al, be);
            float4 y1 = dcx_fma4(make_float4(y1a.x, y1a.y, y1b.x, y1b.y), al, be);
            y0.x = dcx_vmax(y0.x, 0.f); y0.y = dcx_vmax(y0.y, 0.f); y0.z = dcx_vmax(y0.z, 0.f); y0.w = dcx_vmax(y0.w, 0.f);
            y1.x = dcx_vmax(y1.x, 0.f); y1.y = dcx_vmax(y1.y, 0.f); y1.z = dcx_vmax(y1.z, 0.f); y1.w = dcx_vmax(y1.w, 0.f);
            // output (i, j) of the tile is low-resolution pixel (ly, lx0 + j) -> pixel (2 ly + a, 2 (lx0 + j) + b)
            const int ly = cur.ty * C::TH + 2 * (e_tile >> 2) + e_i, lx0 = cur.tx * C::TW + 2 * (e_tile & 3);
            const size_t plane = (size_t)a.ho * a.wo;
            char* dst = reinterpret_cast<char*>(a.out) + ((size_t)cur.n * a.out_cq_total + a.out_cq_off + cq) * plane * 16
                      + (size_t)((2 * ly + pa) * a.wo + 2 * lx0 + pb) * 16;
            const bool ok = cq < a.cout_quads && ly < a.hin;
            if (ok && lx0 < a.win) *reinterpret_cast<float4*>(dst) = y0;
            if (ok && lx0 + 1 < a.win) *reinterpret_cast<float4*>(dst + 32) = y1;
        }
        __syncthreads();                         // sR / sV / sX are free for the next item
    }
    if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0) {
        a.clk_probe[2] = __builtin_amdgcn_s_memtime();
        a.clk_probe[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <class C>
static int dcx_conv_wino2ps_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    a.tiles_x = (a.win + C::TW - 1) / C::TW;          // tiles of the LOW-RESOLUTION map
    a.tiles_y = (a.hin + C::TH - 1) / C::TH;
    if (a.w_ups2w == nullptr || a.alpha == nullptr || a.beta == nullptr || a.out == nullptr) return DCX_E_ARG;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0 || a.cin < 2 * DCX_CCH) return DCX_E_SHAPE;
    if (a.ups != 1 || a.pad != 1 || a.ho != 2 * a.hin || a.wo != 2 * a.win) return DCX_E_SHAPE;
    const long items = (long)a.n * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y * 4;
    if (items <= 0 || items > 0x7fffffffL) return DCX_E_SHAPE;
    const long resident = (long)dcx_device_cu_count();
    const long blocks = items < resident ? items : resident;
    a.xcd_walk = dcx_xcd_walk_enabled() && blocks == resident && (resident & 7) == 0 ? 1 : 0;
    dcx_fill_xcd_cum(a);
    static bool attr_set[DCX_MAX_DEVICES] = {};
    const int dev_i = dcx_current_device();
    if (!attr_set[dev_i]) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_wino2ps_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set[dev_i] = true;
    }
    hipLaunchKernelGGL((dcx_conv_wino2ps_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}
