// dcx_conv_wino2hs.h -- the 2-D Winograd F(2x2,3x3) convolution of dcx_conv_wino2h.h for launches that CANNOT fill the chip
// (one frame, ~16 patches: the reference's own bs=1 protocol, src/benchmark.py:37-53).  Same family, same summation orders, same
// bits (oracle/conv_exact.c: dcx_oracle_conv_wino2h_exact; tests: test_every_conv_instantiation_bit_exact) -- another SHAPE:
//
//   dcx_conv_wino2h.h (TB = 1): workgroup = 4 waves = 64 couts x 16 tiles, every wave owns all 16 Winograd positions of its 16
//   couts -> a launch lasts as long as ONE wave's serial chain of 16 positions x 4 MFMAs per 16-channel unit (2,048 matrix cycles),
//   and with one workgroup per CU (these launches have fewer items than the chip has CUs) one in-order wave per SIMD pays for
//   every latency and every VALU instruction on top: ~3,200 cycles per unit (profiles/experiments/r05_bs1_small_launches.txt).
//
//   here: the 16 positions of an item are SPLIT OVER WAVES.  workgroup = 4 position groups (xi = 0..3: positions 4 xi .. 4 xi + 3)
//   x CG cout groups of 16 = 4 CG waves (CG = 4: 1,024 threads, 64 couts; CG = 2: 32 couts; CG = 1: 16 couts), item = 16 CG couts
//   x 16 tiles (8x8 output pixels).  A wave's chain per unit is 4 positions x 4 MFMAs = 512 matrix cycles; the CG waves that share
//   a SIMD interleave theirs, so with CG = 4 a SIMD carries the same 2,048 matrix cycles per unit as before but four waves hide each
//   other's latencies, and with CG = 1 / 2 a layer of 40 items (conv4a / conv4b of one frame) spreads over 160 / 80 CUs with a
//   chain a quarter / half as long.  The launcher picks the smallest CG whose items still fit the chip in one round.
//   The price: an item's output transform needs all 16 positions of a (cout, tile) -> the accumulators go through LDS once per
//   item (16 KB per cout group) and the transform y[k] = sum_p T[k][p] m[p] runs on the VECTOR ALU as the same sequential fmaf
//   chain over p = 0..15, zero coefficients included (dcx_conv_wino2h.h runs it on v_mfma_f32_4x4x1: the same chain, the same
//   bits -- the matrix pipe would need every position in one wave's registers); with CG < 4 every workgroup of a tile repeats
//   the tile's input transform (the CUs it runs on would otherwise idle).
//
// Per unit and thread (CG = 4): at most one raw float4 of a later unit's 4 x 10 x 10 tile, ONE position of the next unit's transform
// (4 ds_read_b128, 6 v_pk ops, 1 ds_write_b128), 4 weight loads, 4 ds_read_b128 of transformed tiles, 16 MFMAs.  Two barriers per
// unit as in dcx_conv_wino2h.h (raw tile complete / transformed tile complete).  OPERANDS AT LEAST ONE FULL UNIT AHEAD: a ring of U
// units (static register indices, the unit loop is unrolled U times) holds the raw tiles of units c + 1 .. c + U - 1 and the weights
// of c .. c + U - 2 -- at bs=1 nothing is warm (the input was written microseconds ago by other XCDs and comes back from the MALL,
// the weights were last read a call ago: ~1 us either way).  The first version requested a raw tile half a unit and the weights no
// unit ahead of their use and every 512-cycle unit waited for both (bs=1 protocol 2,800 -> 2,855 calls/s; with the ring -> 3,060;
// U = 2, 3, 4 measure the same: profiles/experiments/r06_bs1_split_positions.txt).  CG = 4 (128 registers per wave) keeps one weight set and refills a position
// pair's registers as soon as its MFMAs are issued.  Work items are independent: prologue, nch units, epilogue, nothing carried
// over (a cross-item pipeline was built and measured slower for these launches).  Layouts, item walk and zero padding by buffer
// range are dcx_conv_wino2h.h's.
#pragma once
#include "dcx_conv_wino2h.h"

template <bool POOL_, int CG_>
struct DcxWino2hsCfg {
    static constexpr bool POOL = POOL_;
    static constexpr int CG = CG_;                          // cout groups (of 16) per workgroup
    static constexpr int TH = 8, TW = 8, TX = 4, NTILES = 16;
    static constexpr int HH = TH + 2, RW = TW + 2;
    static constexpr int NWAVES = 4 * CG, NTHREADS = 64 * NWAVES;
    static constexpr int COUT_TILE = 16 * CG;
    static constexpr int CQC = DCX_CCH / 4;
    static constexpr int RAW = CQC * HH * RW;               // 400 float4 per unit
    static constexpr int ITER_R = (RAW + NTHREADS - 1) / NTHREADS;
    static constexpr int NUP = 4 / CG;                      // transform: positions (nu values) per thread
    // raw tile in LDS: row pitch 12, one slot of shift on every second row PAIR.  The transform reads float4 (row 2 ty + i, column
    // 2 tx + j) with lane = (cq, tile): with the hardware's ds_read_b128 lane groups every group then covers 16 different 16-byte slots
    // (tools/lds_sim.py; dcx_conv_wino2h.h's shift for the same tile -- (hy >> 2) & 1, built for its quarter-piece reads -- left a
    // third of these reads two-way conflicted: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 18 % on the 16-cout variant)
    static constexpr int RP = RW + 2;
    __host__ __device__ static constexpr int raw_slot(int cq, int hy, int hx) { return (cq * HH + hy) * RP + hx + ((hy >> 1) & 1); }
    static constexpr int RAW_LDS = CQC * HH * RP;
    static constexpr int VPLANE = CQC * 16;                 // float4 per position: [cq][tile]
    static constexpr int LDS_V = 16 * VPLANE;               // one transformed buffer (16 KB)
    static constexpr int LDS_X = 16 * (4 * CG) * 16;        // accumulator exchange: [pos][cout quad][tile] float4
    static constexpr size_t LDS_BYTES = (size_t)(2 * LDS_V + RAW_LDS + LDS_X) * 16;
    static constexpr int OCC = CG == 4 ? 1 : 2;             // workgroups per CU the launch is sized for (LDS; <= 4 waves per SIMD)
#ifndef DCX_W2HS_U
#define DCX_W2HS_U (CG == 4 ? 2 : 4)
#endif
    static constexpr int U = DCX_W2HS_U;                    // units in flight (1..4): raw tiles / weights requested U - 1 units ahead.  At bs=1 the
                                                            // operands come from the MALL (~1 us), not from a warm L2; what counts is a FULL
                                                            // unit of lead (U = 2 .. 4 measure the same)
    static_assert(CG == 1 || CG == 2 || CG == 4, "1, 2 or 4 cout groups");
    static_assert(U >= 2 && U <= 4, "ring depth");
    // CG = 4 (1,024 threads: 128 registers per wave): ONE set of weight registers -- the next unit's weights of a position pair are
    // requested into the pair's registers as soon as its MFMAs are issued (half a unit = 1,024 matrix cycles of lead)
    static constexpr bool HALFROT = CG == 4;
    static_assert(OCC * LDS_BYTES <= 160 * 1024, "LDS");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, C::OCC) void dcx_conv_wino2hs_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int CQC = C::CQC, RW = C::RW, HH = C::HH, ITER_R = C::ITER_R, LDSV = C::LDS_V, VPLANE = C::VPLANE, NUP = C::NUP, CG = C::CG;
    float4* const sR = sB + 2 * LDSV;
    float4* const sX = sR + C::RAW_LDS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pgp = wv & 3;                                         // the wave's position group = Winograd row xi
    const int cg = wv >> 2;                                         // the wave's cout group
    const int g4 = lane >> 4, l15 = lane & 15;

    // ---- work list (dcx_conv_wino2h.h) -----------------------------------------------------------------------------
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    int n_eff = a.n;
    if (a.n_limit != nullptr) n_eff = min(n_eff, *a.n_limit);
    const int total = n_eff * n_ct * tiles;
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
        it.tx = wi % a.tiles_x; wi /= a.tiles_x;
        it.ty = wi % a.tiles_y; wi /= a.tiles_y;
        it.ct = wi % n_ct;
        it.n = wi / n_ct;
        it.ph = 0;
        return it;
    };
    const int hl = a.hin << a.ups, wl = a.win << a.ups;

    // ---- operands ----------------------------------------------------------------------------------------------------
    // weights [pos][cin/4][cout_pad][4]: lane (r = l15, g = g4) reads cout ct * COUT_TILE + cg * 16 + r, channel quad g of the chunk
    const unsigned w_lane_off = (unsigned)(g4 * a.cout_pad + cg * 16 + l15) * 16u;
    const unsigned w_pos_stride = (unsigned)((a.cin >> 2) * a.cout_pad) * 16u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_wino2), (short)0, (int)(16u * w_pos_stride), 0x00020000);
    auto unit_wbase = [&](const DcxItem& it, int c) {
        return (unsigned)((c * CQC) * a.cout_pad + it.ct * C::COUT_TILE) * 16u + (unsigned)(4 * pgp) * w_pos_stride;
    };
    auto load_a = [&](unsigned wbase, int pp) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane_off, wbase + (unsigned)pp * w_pos_stride, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // transformed activations sV[buf][pos][cq][tile]: lane (n = l15, g = g4) reads tile n, channel quad g
    auto load_b = [&](int buf, int pp) { return sB[buf * LDSV + (4 * pgp + pp) * VPLANE + lane]; };

    // ---- staging: raw tile -----------------------------------------------------------------------------------------
    int r_hyx[ITER_R], r_slot[ITER_R];
    unsigned r_rel[ITER_R];
#pragma unroll
    for (int k = 0; k < ITER_R; ++k) {
        const int idx = tid + k * C::NTHREADS;
        const int cq = idx / (HH * RW);
        const int hp = idx - cq * (HH * RW);
        const int hy = hp / RW, hx = hp - hy * RW;
        r_hyx[k] = hy << 16 | hx;
        const int prow = ((hy - a.pad) >> a.ups) + a.pad, pcol = ((hx - a.pad) >> a.ups) + a.pad;
        r_rel[k] = idx < C::RAW ? (unsigned)((cq * a.hin + prow) * a.win + pcol) * 16u : 0x80000000u;
        r_slot[k] = idx < C::RAW ? C::raw_slot(cq, hy, hx) : C::RP - 1;      // slot RP - 1 of row 0 is free: dump slot
    }
    auto unit_rsrc = [&](const DcxItem& it, int c) {
        const long tile_off = (long)(((it.ty * C::TH) >> a.ups) - a.pad) * a.win + (((it.tx * C::TW) >> a.ups) - a.pad);
        const float* base = a.in + (((size_t)it.n * a.in_cq_total + a.in_cq_off + (size_t)c * CQC) * (size_t)a.hin * a.win + tile_off) * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
    };
    auto stage_fetch = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto raw_offsets = [&](const DcxItem& it, unsigned (&roff)[ITER_R]) {
        const int sy0 = it.ty * C::TH - a.pad, sx0 = it.tx * C::TW - a.pad;
        const bool interior = sy0 >= 0 && sx0 >= 0 && sy0 + HH <= hl && sx0 + RW <= wl;
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) {
            roff[k] = r_rel[k];
            if (!interior) {
                const int ly = sy0 + (r_hyx[k] >> 16), lx = sx0 + (r_hyx[k] & 0xffff);
                if (!((unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl)) roff[k] = 0x80000000u;
            }
        }
    };

    // ---- staging: input transform --------------------------------------------------------------------------------------
    // thread = (tile, cq, xi, nu group): row combination t[c] = d[ia][c] + sr * d[ib][c] (xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3),
    // then v[nu] over the columns the same way -- exact sums / differences, the values dcx_conv_wino2h.h forms.
    const int x_tile = tid & 15, x_cq = (tid >> 4) & 3;
    const int x_xi = __builtin_amdgcn_readfirstlane((tid >> 6) & 3), x_ng = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int x_ty = x_tile >> 2, x_tx = x_tile & 3;
    const int x_ia = x_xi == 0 ? 0 : x_xi == 2 ? 2 : 1, x_ib = x_xi == 2 ? 1 : x_xi == 3 ? 3 : 2;
    const float x_srf = x_xi == 1 ? 1.f : -1.f;
    const dcx_f32x2 x_sr = {x_srf, x_srf};
    // columns read: NUP = 4: 0..3;  NUP = 2: nu group 0 -> 0, 1, 2 (nu 0, 1), group 1 -> 1, 2, 3 (nu 2, 3);  NUP = 1: the two columns of nu
    constexpr int NC = NUP == 4 ? 4 : NUP == 2 ? 3 : 2;
    int x_col[NC];
    if (NUP == 4) { for (int i = 0; i < NC; ++i) x_col[i] = i; }
    else if (NUP == 2) { for (int i = 0; i < NC; ++i) x_col[i] = x_ng + i; }
    else { x_col[0] = x_ng == 0 ? 0 : x_ng == 2 ? 2 : 1; x_col[1] = x_ng == 2 ? 1 : x_ng == 3 ? 3 : 2; }
    int x_ra[NC], x_rb[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        x_ra[i] = C::raw_slot(x_cq, 2 * x_ty + x_ia, 2 * x_tx + x_col[i]);
        x_rb[i] = C::raw_slot(x_cq, 2 * x_ty + x_ib, 2 * x_tx + x_col[i]);
    }
    const float x_scf = (NUP == 1 && x_ng == 1) ? 1.f : -1.f;
    const dcx_f32x2 x_sc = {x_scf, x_scf};
    const int x_dst = (4 * x_xi + x_ng * NUP) * VPLANE + x_cq * 16 + x_tile;       // + local nu * VPLANE
    auto fmas = [](const float4& x, const dcx_f32x2 s, const float4& y) {          // y + s * x (s = +-1: exactly y +- x)
        const dcx_f32x2 lo = __builtin_elementwise_fma(dcx_f32x2{x.x, x.y}, s, dcx_f32x2{y.x, y.y});
        const dcx_f32x2 hi = __builtin_elementwise_fma(dcx_f32x2{x.z, x.w}, s, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    auto sub4 = [](const float4& x, const float4& y) {
        const dcx_f32x2 lo = dcx_pk_sub(dcx_f32x2{x.x, x.y}, dcx_f32x2{y.x, y.y}), hi = dcx_pk_sub(dcx_f32x2{x.z, x.w}, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    auto add4 = [](const float4& x, const float4& y) {
        const dcx_f32x2 lo = dcx_pk_add(dcx_f32x2{x.x, x.y}, dcx_f32x2{y.x, y.y}), hi = dcx_pk_add(dcx_f32x2{x.z, x.w}, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    float4 xa[NC], xb[NC];
    auto xform_read = [&]() {
#pragma unroll
        for (int i = 0; i < NC; ++i) { xa[i] = sR[x_ra[i]]; xb[i] = sR[x_rb[i]]; }
    };
    auto xform_write = [&](float4* vbuf) {
        float4 t[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) t[i] = fmas(xb[i], x_sr, xa[i]);
        if constexpr (NUP == 4) {
            vbuf[x_dst] = sub4(t[0], t[2]);
            vbuf[x_dst + VPLANE] = add4(t[1], t[2]);
            vbuf[x_dst + 2 * VPLANE] = sub4(t[2], t[1]);
            vbuf[x_dst + 3 * VPLANE] = sub4(t[1], t[3]);
        } else if constexpr (NUP == 2) {
            if (x_ng == 0) {            // columns 0, 1, 2: nu 0 = t0 - t2, nu 1 = t1 + t2
                vbuf[x_dst] = sub4(t[0], t[2]);
                vbuf[x_dst + VPLANE] = add4(t[1], t[2]);
            } else {                    // columns 1, 2, 3: nu 2 = t2 - t1, nu 3 = t1 - t3
                vbuf[x_dst] = sub4(t[1], t[0]);
                vbuf[x_dst + VPLANE] = sub4(t[0], t[2]);
            }
        } else {
            vbuf[x_dst] = fmas(t[1], x_sc, t[0]);       // nu 0: t0 - t2, 1: t1 + t2, 2: t2 - t1, 3: t1 - t3
        }
    };

    // ---- epilogue roles: thread = (cout quad of the item, tile, output pixel k = 2 i + j of the 2x2 tile) ---------------------
    const int e_k = lane & 3, e_tile = lane >> 2, e_cq = wv;        // NWAVES = 4 CG = cout quads per item
    const int hs = C::POOL ? (a.ho >> 1) : a.ho, ws = C::POOL ? (a.wo >> 1) : a.wo;

    dcx_f32x4 acc[4];
    constexpr int U = C::U;                       // units in flight: raw tiles and weights are requested U - 1 units ahead of their use
    constexpr int UW = C::HALFROT ? 1 : U;
    float4 aq[UW][4];                             // weights of unit u sit in aq[u % UW]
    float4 rq[U][ITER_R];                         // raw float4 of unit u sit in rq[u % U] until they are stored to LDS during unit u - 1
                                                  // (static indices: the unit loop is unrolled U times)

    // ---- work items: prologue (first unit staged synchronously), nch units, epilogue; nothing is carried from one item to the next
    // (these launches give every item a CU of its own) -------------------------------------------------------------------------
    for (; w < w_end; w += gstride) {
        const DcxItem cur = decode(w);
        unsigned roff[ITER_R];
        raw_offsets(cur, roff);
        const unsigned wb0 = unit_wbase(cur, 0);
        const unsigned w_unit = (unsigned)(CQC * a.cout_pad) * 16u;        // bytes between the weights of consecutive units
        {
            float4 r0[ITER_R];
            const __amdgpu_buffer_rsrc_t rs0 = unit_rsrc(cur, 0);
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) r0[k] = stage_fetch(rs0, roff[k]);
#pragma unroll
            for (int i = 0; i < U - 1; ++i) {
                const __amdgpu_buffer_rsrc_t rs = unit_rsrc(cur, i + 1 < nch ? i + 1 : nch - 1);
#pragma unroll
                for (int k = 0; k < ITER_R; ++k) rq[(i + 1) % U][k] = stage_fetch(rs, roff[k]);
                if (i < UW - 1 || (UW == 1 && i == 0)) {
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) aq[i % UW][pp] = load_a(wb0 + (unsigned)(i < nch ? i : nch - 1) * w_unit, pp);
                }
            }
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) sR[r_slot[k]] = r0[k];
            __syncthreads();
            xform_read();
            xform_write(sB);
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0"
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
                if (!C::HALFROT) {
                    const int cw = c + U - 1 < nch ? c + U - 1 : nch - 1;
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) aq[(K + U - 1) % UW][pp] = load_a(wb0 + (unsigned)cw * w_unit, pp);
                }
                const __amdgpu_buffer_rsrc_t rs = unit_rsrc(cur, c + U < nch ? c + U : nch - 1);
#pragma unroll
                for (int k = 0; k < ITER_R; ++k) rq[K][k] = stage_fetch(rs, roff[k]);      // (the slot of this unit's own raw tile: stored a unit ago)
            }
            float4 bq[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) bq[pp] = load_b(buf, pp);
            float4* vnext = sB + (buf ^ 1) * LDSV;
            // MFMA j of position pp consumes component j of both operands; per accumulator the order is j = 0..3 (the family's
            // order); consecutive MFMAs of the wave alternate between two accumulators
#define DCX_W2HS_MFMA_PAIR(P0)                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                 \
                _Pragma("unroll") for (int pp = (P0); pp < (P0) + 2; ++pp) {                                                \
                    const float4 aa = aq[K % UW][pp], bb = bq[pp];                                                               \
                    const float av = j == 0 ? aa.x : j == 1 ? aa.y : j == 2 ? aa.z : aa.w;                                  \
                    const float bv = j == 0 ? bb.x : j == 1 ? bb.y : j == 2 ? bb.z : bb.w;                                  \
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[pp]) : "v"(av), "v"(bv));               \
                }                                                                                                           \
            }
            asm volatile("s_nop 1");
            DCX_W2HS_MFMA_PAIR(0)
            const unsigned wb_n = wb0 + (unsigned)(has_next ? c + 1 : c) * w_unit;
            if (C::HALFROT) { aq[0][0] = load_a(wb_n, 0); aq[0][1] = load_a(wb_n, 1); }
            __syncthreads();                         // raw tile of the next unit complete
            if (has_next) xform_read();
            DCX_W2HS_MFMA_PAIR(2)
            if (C::HALFROT) { aq[0][2] = load_a(wb_n, 2); aq[0][3] = load_a(wb_n, 3); }
            if (has_next) xform_write(vnext);
        };
        for (int c0 = 0; c0 < nch; c0 += U) {
            run_unit(std::integral_constant<int, 0>{}, c0);
            if (U > 1 && c0 + 1 < nch) run_unit(std::integral_constant<int, 1 % U>{}, c0 + 1);
            if (U > 2 && c0 + 2 < nch) run_unit(std::integral_constant<int, 2 % U>{}, c0 + 2);
            if (U > 3 && c0 + 3 < nch) run_unit(std::integral_constant<int, 3 % U>{}, c0 + 3);
        }

        // ---- epilogue: accumulators -> LDS, output transform + BN + ReLU (+ pool) on the vector ALU, store -----------------
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // MFMA result -> read by a non-MFMA instruction
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            asm volatile("" : "+v"(acc[pp]));
            sX[((4 * pgp + pp) * (4 * CG) + cg * 4 + g4) * 16 + l15] = make_float4(acc[pp][0], acc[pp][1], acc[pp][2], acc[pp][3]);
        }
        const int cq = (cur.ct * C::COUT_TILE >> 2) + e_cq;                // the thread's output channel quad
        const float4 al = reinterpret_cast<const float4*>(a.alpha)[cq], be = reinterpret_cast<const float4*>(a.beta)[cq];
        __syncthreads();
        // T[k][p] = AT[i][xi] * AT[j][nu], AT = [[1,1,1,0],[0,1,-1,-1]]
        float e_ci[4], e_cj[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e_ci[q] = (e_k >> 1) == 0 ? (q < 3 ? 1.f : 0.f) : (q == 0 ? 0.f : q == 1 ? 1.f : -1.f);
            e_cj[q] = (e_k & 1) == 0 ? (q < 3 ? 1.f : 0.f) : (q == 0 ? 0.f : q == 1 ? 1.f : -1.f);
        }
        dcx_f32x2 y01, y23;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float4 m = sX[(p * (4 * CG) + e_cq) * 16 + e_tile];
            const float t = e_ci[p >> 2] * e_cj[p & 3];
            const dcx_f32x2 tt = {t, t};
            if (p == 0) {
                y01 = __builtin_elementwise_fma(tt, dcx_f32x2{m.x, m.y}, dcx_f32x2{0.f, 0.f});
                y23 = __builtin_elementwise_fma(tt, dcx_f32x2{m.z, m.w}, dcx_f32x2{0.f, 0.f});
            } else {
                y01 = __builtin_elementwise_fma(tt, dcx_f32x2{m.x, m.y}, y01);
                y23 = __builtin_elementwise_fma(tt, dcx_f32x2{m.z, m.w}, y23);
            }
        }
        float4 y = dcx_fma4(make_float4(y01.x, y01.y, y23.x, y23.y), al, be);
        const size_t plane = (size_t)hs * ws;
        char* obase = reinterpret_cast<char*>(a.out) + ((size_t)cur.n * a.out_cq_total + a.out_cq_off + cq) * plane * 16;
        const int oy = cur.ty * C::TH + 2 * (e_tile >> 2) + (e_k >> 1), ox = cur.tx * C::TW + 2 * (e_tile & 3) + (e_k & 1);
        const bool cok = cq < a.cout_quads;
        if (C::POOL) {
            y = dcx_relu_quad_max(y);          // lanes 4 t .. 4 t + 3 = the 2x2 window
            // MaxPool2d(2,2) floors: a window needs both of its rows and columns
            if (cok && e_k == 0 && oy + 1 < a.ho && ox + 1 < a.wo)
                *reinterpret_cast<float4*>(obase + (size_t)((oy >> 1) * ws + (ox >> 1)) * 16) = y;
        } else {
            y.x = dcx_vmax(y.x, 0.f); y.y = dcx_vmax(y.y, 0.f); y.z = dcx_vmax(y.z, 0.f); y.w = dcx_vmax(y.w, 0.f);
            if (cok && oy < a.ho && ox < a.wo) *reinterpret_cast<float4*>(obase + (size_t)(oy * ws + ox) * 16) = y;
        }
        __syncthreads();                         // sR / sV / sX are free for the next item
    }
    if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0) {
        a.clk_probe[2] = __builtin_amdgcn_s_memtime();
        a.clk_probe[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <class C>
static int dcx_conv_wino2hs_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    a.tiles_x = (a.wo + C::TW - 1) / C::TW;
    a.tiles_y = (a.ho + C::TH - 1) / C::TH;
    if (a.w_wino2 == nullptr || a.alpha == nullptr || a.beta == nullptr || a.out == nullptr) return DCX_E_ARG;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0 || a.cin < 2 * DCX_CCH) return DCX_E_SHAPE;
    const long items = (long)a.n * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y;
    if (items <= 0 || items > 0x7fffffffL) return DCX_E_SHAPE;
    const long resident = (long)C::OCC * dcx_device_cu_count();
    const long blocks = items < resident ? items : resident;
    a.xcd_walk = dcx_xcd_walk_enabled() && blocks == resident && (resident & 7) == 0 ? 1 : 0;
    dcx_fill_xcd_cum(a);
    a.ct_outer = 0;
    static bool attr_set[DCX_MAX_DEVICES] = {};
    const int dev_i = dcx_current_device();
    if (!attr_set[dev_i]) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_wino2hs_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set[dev_i] = true;
    }
    hipLaunchKernelGGL((dcx_conv_wino2hs_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}
