// dcx_conv_wino2.h -- 3x3 convolution + BN + ReLU (+2x2 max-pool) (+RefineNet head) with the 2-D Winograd transform
// F(2x2, 3x3) on the gfx950 fp32 matrix cores: 16 products per 2x2 output tile and input channel instead of 36, i.e.
// 4/9 of the direct convolution's MFMAs (the 1-D variant in dcx_conv_wino.h executes 2/3).  Read dcx_conv_mfma.h and
// dcx_conv_wino.h first; layouts, persistent work walk and staging scheme are the same.
//
//   per 2x2 output tile (rows 2ty, 2ty+1; columns 2tx, 2tx+1) and input channel, d = the 4x4 input window whose top-left
//   pixel is (2ty - pad, 2tx - pad):
//     rows     t[xi][c] :  t0 = d[0][c]-d[2][c]   t1 = d[1][c]+d[2][c]   t2 = d[2][c]-d[1][c]   t3 = d[1][c]-d[3][c]
//     columns  v[xi][nu]:  v0 = t[xi][0]-t[xi][2] v1 = t[xi][1]+t[xi][2] v2 = t[xi][2]-t[xi][1] v3 = t[xi][1]-t[xi][3]
//     weights  u = G g G^T, transformed on the host in fp32 (rows first, then columns; h1 = ((h0+h1)+h2)*0.5f ...)
//     m[xi][nu] += u[xi][nu] * v[xi][nu]                                   (16 accumulators = 256 registers per lane)
//     y[i][j] = 0;  for p = xi*4 + nu ascending:  y[i][j] = fmaf(AT[i][xi]*AT[j][nu], m[xi][nu], y[i][j]),
//               AT = [[1,1,1,0],[0,1,-1,-1]]  (a sequential fmaf chain over ALL 16 positions, zero coefficients included:
//               it runs on the matrix cores, see "Output transform" below)
//
// GEMM view: 16 independent GEMMs, M = cout, N = tiles, K = cin.  A wave owns 32 couts x 32 tiles x 16 positions; its
// 256 accumulator registers live in AGPRs (one workgroup of 4 waves per CU, the whole 512-register file per lane);
// workgroup = 64 couts x 64 tiles (16x16 or 8x32 output pixels).  Summation order of every m (restated bit-exactly by
// oracle/conv_exact.c: dcx_oracle_conv_wino2_exact):
//   m = 0;  for chunk c (16 cin) / s in 0..1 / j in 0..3:  m = fmaf(u[8s+j], v[8s+j], m);  m = fmaf(u[8s+4+j], v[8s+4+j], m)
//
// Unit = 16 channels = 2 k-steps x 16 positions x 4 MFMAs = 8192 matrix cycles.  Operands are prefetched at position
// granularity (weights from L2, transformed activations from LDS, DQ positions ahead); the next unit's raw tile is
// loaded early in the unit, written to sR in the middle, and transformed in the second half (per thread 16 ds_read_b128,
// 64 v_pk_add_f32, 16 ds_write_b128), one micro-step per MFMA pair.  The unit body is one basic block.
#pragma once
#include "dcx_conv_wino.h"

template <int TH_, int TW_, bool POOL_, int EPI_ = DCX_EPI_BNRELU, int G_ = 1>
struct DcxWino2Cfg {
    static constexpr int TH = TH_, TW = TW_;
    static constexpr int G = G_;                           // images per workgroup tile: G > 1 packs G whole small maps
                                                           // (<= TH x TW each, e.g. RefineNet's 8x8) into the 64 tiles
    static constexpr bool POOL = POOL_;
    static constexpr int EPI = EPI_;
    static constexpr int NTHREADS = 256;
    static constexpr int COUT_TILE = 64;
    static constexpr int TY = TH / 2, TX = TW / 2;         // 2x2 output tiles of the workgroup tile
    static constexpr int TPI = TY * TX;                    // 2x2 tiles per image region
    static constexpr int NTILES = G * TPI;                 // <= 64
    static constexpr int HH = TH + 2, RW = TW + 2;         // raw input rows / columns
    static constexpr int CQC = DCX_CCH / 4;
    static constexpr int RAW = G * CQC * HH * RW;          // [image][cq][row][col]
    static constexpr int ITER_R = (RAW + NTHREADS - 1) / NTHREADS;
    static constexpr int RAW_PAD = ITER_R * NTHREADS;
    static constexpr int VPLANE = CQC * 64;                // float4 per position: [cq][tile]
    static constexpr int LDS_FLOAT4 = 16 * VPLANE;         // one transformed buffer
    static constexpr size_t LDS_BYTES = (size_t)(2 * LDS_FLOAT4 + RAW_PAD) * 16;
    static constexpr int NQ = 2 * 16;                      // (k-step, position) pairs per unit
    // operand prefetch distances in positions (256 matrix cycles each).  Weights share the vector-memory return queue with
    // the raw staging loads, which come from HBM: loads return in order, so a weight load issued behind them arrives after
    // them -- its distance must cover HBM latency (measured: 4 positions cost 1,500 cycles per unit).  LDS reads do not.
    static constexpr int DQ = 8;                           // weights (A)
    static constexpr int DQB = 4;                          // transformed activations (B)
    // staging schedule in events (one per MFMA pair: 64 per unit, 128 matrix cycles apart)
    static constexpr int E_RAW_LOAD = 0;                   // raw float4 #k is requested at event k
    static constexpr int E_RAW_STORE = 22;                 // ... and written to sR at event 22 + k
    static constexpr int E_XFORM = 32;                     // mid barrier before this event; the transform follows
    static_assert(TH % 2 == 0 && TW % 2 == 0 && NTILES <= 64 && NTILES > 32, "tile must hold 33..64 2x2 tiles");
    static_assert(ITER_R <= 8 && E_RAW_STORE + ITER_R <= E_XFORM, "raw staging does not fit the schedule");
    static_assert(LDS_BYTES + 1536 <= 160 * 1024, "LDS tile too large");      // + per-launch epilogue constants (checked at launch)
    static_assert(EPI == DCX_EPI_BNRELU || (EPI == DCX_EPI_HEAT && !POOL), "unsupported epilogue");
    static_assert(G == 1 || (EPI == DCX_EPI_BNRELU && !POOL), "grouped tiles: plain BN + ReLU layers only");
};

template <class C>
__global__ __launch_bounds__(256, 1) void dcx_conv_wino2_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int TX = C::TX, RW = C::RW, ITER_R = C::ITER_R, LDSF = C::LDS_FLOAT4, CQC = C::CQC, VPLANE = C::VPLANE;
    constexpr int NQ = C::NQ, DQ = C::DQ, DQB = C::DQB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- work list ------------------------------------------------------------------------------
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    int n_eff = a.n;
    if (a.n_limit != nullptr) n_eff = min(n_eff, *a.n_limit);
    const int total = ((n_eff + C::G - 1) / C::G) * n_ct * tiles;      // G > 1: one work item covers G images (tiles == 1)
    // Work walk.  Flat: block b takes items b, b + grid, ...  XCD-aware (a.xcd_walk, grid a multiple of 8): block b runs on
    // XCD b % 8 (observed dispatch order, used for speed only -- any placement is correct), so XCD x's blocks walk the x-th
    // CONTIGUOUS eighth of the item list, slot b / 8 first, stride grid / 8: the 32 workgroups that share an L2 work on
    // neighbouring tiles of the same images (and on both cout tiles of a 128-cout layer) at the same time, so the 2-pixel
    // halos and the second read of the input are L2 hits instead of HBM / Infinity-Cache fetches (FETCH_SIZE A/B: profiles/).
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

    // ---- the lane's 2x2 output tile ---------------------------------------------------------------
    const int qt = wn * 32 + l31;
    const bool qok = qt < C::NTILES;
    const int q_img = qt / C::TPI, q_t = qt - q_img * C::TPI;     // image inside the group (0 when G == 1)
    const int qty = q_t / TX, qtx = q_t - qty * TX;
    const int tile_b = half * 64 + (qok ? qt : 0);        // + (pos * CQC + 2s) * 64  ->  sV index of the B operand

    // ---- operand fetch ---------------------------------------------------------------------------
    // weights: [pos][cin/4][cout_pad][4]
    const unsigned w_lane_off = (unsigned)((half * a.cout_pad) + wm * 32 + l31) * 16u;
    const unsigned w_pos_stride = (unsigned)((a.cin >> 2) * a.cout_pad) * 16u;
    const unsigned w_s_stride = (unsigned)(2 * a.cout_pad) * 16u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_wino2), (short)0, (int)(16u * w_pos_stride), 0x00020000);
    auto unit_wbase = [&](const DcxItem& it, int c) {
        return (unsigned)((c * CQC) * a.cout_pad + it.ct * C::COUT_TILE) * 16u;
    };
    auto load_a = [&](unsigned wbase, int q) {      // q = s * 16 + pos
        const unsigned soff = wbase + (unsigned)(q & 15) * w_pos_stride + (unsigned)(q >> 4) * w_s_stride;
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane_off, soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto load_b = [&](int buf, int q) {
        return sB[buf * LDSF + ((q & 15) * CQC + 2 * (q >> 4)) * 64 + tile_b];
    };

    // ---- staging ----------------------------------------------------------------------------------
    int r_hy[ITER_R], r_hx[ITER_R];
    unsigned r_rel[ITER_R];
    const unsigned in_img_stride = (unsigned)a.in_cq_total * (unsigned)(a.hin * a.win);   // float4 between images
#pragma unroll
    for (int k = 0; k < ITER_R; ++k) {
        const int idx = tid + k * C::NTHREADS;
        const int img = idx / (CQC * C::HH * RW);
        const int rem = idx - img * (CQC * C::HH * RW);
        const int cq = rem / (C::HH * RW);
        const int hp = rem - cq * (C::HH * RW);
        r_hy[k] = hp / RW;
        r_hx[k] = hp - r_hy[k] * RW;
        const int prow = ((r_hy[k] - a.pad) >> a.ups) + a.pad, pcol = ((r_hx[k] - a.pad) >> a.ups) + a.pad;
        r_rel[k] = idx < C::RAW ? ((unsigned)img * in_img_stride + (unsigned)((cq * a.hin + prow) * a.win + pcol)) * 16u : 0x80000000u;
        if (C::G > 1) {
            // a grouped tile always starts at pixel (0, 0) of its images: the zero-padding predicate is a per-piece constant
            const int ly = r_hy[k] - a.pad, lx = r_hx[k] - a.pad;
            if (!((unsigned)ly < (unsigned)(a.hin << a.ups) && (unsigned)lx < (unsigned)(a.win << a.ups))) r_rel[k] = 0x80000000u;
            r_hy[k] = img;      // what the per-unit check needs for grouped tiles: which image of the group the piece reads
        }
    }
    float4* sR = sB + 2 * LDSF;
    // transform piece of this thread: (cq, tile) = (tid / 64, tid % 64); tiles past the end redo the last tile
    const int x_cq = tid >> 6;
    const int x_tile = min(tid & 63, C::NTILES - 1);
    const int x_img = x_tile / C::TPI, x_t = x_tile - x_img * C::TPI;
    const int x_ty = x_t / TX, x_tx = x_t - x_ty * TX;
    const int x_src = ((x_img * CQC + x_cq) * C::HH + 2 * x_ty) * RW + 2 * x_tx;     // raw index of the window's top-left pixel
    const int x_dst = x_cq * 64 + x_tile;                            // + pos * VPLANE
    auto unit_rsrc = [&](const DcxItem& it, int c) {
        const long tile_off = (long)(((it.ty * C::TH) >> a.ups) - a.pad) * a.win + (((it.tx * C::TW) >> a.ups) - a.pad);
        const float* base = a.in + (((size_t)it.n * C::G * a.in_cq_total + a.in_cq_off + (size_t)c * CQC) * (size_t)a.hin * a.win
                                    + tile_off) * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
    };
    auto tile_interior = [&](const DcxItem& it) {
        if (C::G > 1) return (it.n + 1) * C::G <= a.n;     // all images of the group exist (spatial padding is folded into r_rel)
        const int sy0 = it.ty * C::TH - a.pad, sx0 = it.tx * C::TW - a.pad;
        return sy0 >= 0 && sx0 >= 0 && sy0 + C::HH <= hl && sx0 + RW <= wl;
    };
    auto stage_fetch = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // float4 a -/+ b as two packed adds (epilogue)
    auto sub4 = [](const float4& x, const float4& y) {
        const dcx_f32x2 lo = dcx_pk_sub(dcx_f32x2{x.x, x.y}, dcx_f32x2{y.x, y.y}), hi = dcx_pk_sub(dcx_f32x2{x.z, x.w}, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    auto add4 = [](const float4& x, const float4& y) {
        const dcx_f32x2 lo = dcx_pk_add(dcx_f32x2{x.x, x.y}, dcx_f32x2{y.x, y.y}), hi = dcx_pk_add(dcx_f32x2{x.z, x.w}, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    // Input transform of the thread's (cq, tile) piece as 21 events: state = raw rows r1, r2, one more row (r0, later r3),
    // the current t[4] and one finished position waiting for its LDS write (issued one event after its adds, so the
    // write never waits for the vector ALU).
    //   event 0: read rows 0 and 2 (8 ds_read_b128)      event 1: read row 1      event 8: also read row 3
    //   event 4 + ms (ms = xi*4 + nu = 0..15): at nu == 0 form t[xi] (8 packed adds), then position ms (2 packed adds)
    //   event 20: last write
    float4 xr1[4], xr2[4], xra[4], xt[4], xv;
    auto xform_read = [&](int which) {
#pragma unroll
        for (int cidx = 0; cidx < 4; ++cidx) {
            if (which == 0) { xra[cidx] = sR[x_src + 0 * RW + cidx]; xr2[cidx] = sR[x_src + 2 * RW + cidx]; }
            if (which == 1) xr1[cidx] = sR[x_src + 1 * RW + cidx];
            if (which == 3) xra[cidx] = sR[x_src + 3 * RW + cidx];
        }
    };
    auto xform_event = [&](float4* vbuf, int x) {      // x = 0 .. 20
        if (x == 0) xform_read(0);
        else if (x == 1) xform_read(1);
        else if (x >= 4) {                              // events 2, 3: idle -- the 12 reads (3 KB per lane group) need ~250 cycles
            const int ms = x - 4;
            if (ms > 0) vbuf[x_dst + (ms - 1) * VPLANE] = xv;      // position ms-1, computed one event ago
            if (ms < 16) {
                const int xi = ms >> 2, nu = ms & 3;
                if (nu == 0) {
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx)
                        xt[cidx] = xi == 0 ? sub4(xra[cidx], xr2[cidx]) : xi == 1 ? add4(xr1[cidx], xr2[cidx])
                                 : xi == 2 ? sub4(xr2[cidx], xr1[cidx]) : sub4(xr1[cidx], xra[cidx]);
                }
                xv = nu == 0 ? sub4(xt[0], xt[2]) : nu == 1 ? add4(xt[1], xt[2]) : nu == 2 ? sub4(xt[2], xt[1]) : sub4(xt[1], xt[3]);
                if (ms == 4) xform_read(3);          // row 0 is dead after t[0]; row 3 lands long before ms == 12
            }
        }
    };

    // ---- epilogue constants in LDS: alpha, beta2 (, head weights) ----------------------------------------
    float4* sP = sB + 2 * LDSF + C::RAW_PAD;
    const int cq_pad = a.cout_pad >> 2;
    for (int i = tid; i < cq_pad; i += C::NTHREADS) {
        sP[i] = reinterpret_cast<const float4*>(a.alpha)[i];
        sP[cq_pad + i] = reinterpret_cast<const float4*>(a.beta)[i];
        if (C::EPI == DCX_EPI_HEAT) sP[2 * cq_pad + i] = reinterpret_cast<const float4*>(a.head_w)[i];
    }
    // output-transform coefficients T[k = 2i + j][p = 4 xi + nu] = AT[i][xi] * AT[j][nu] as [k][p] floats (256 B)
    float* sT = reinterpret_cast<float*>(sP + 3 * cq_pad);
    if (tid < 64) {
        const int k = tid >> 4, p = tid & 15, i = k >> 1, j = k & 1, xi = p >> 2, nu = p & 3;
        const int ci = i == 0 ? (xi < 3 ? 1 : 0) : (xi == 0 ? 0 : xi == 1 ? 1 : -1);
        const int cj = j == 0 ? (nu < 3 ? 1 : 0) : (nu == 0 ? 0 : nu == 1 ? 1 : -1);
        sT[tid] = (float)(ci * cj);
    }
    const int hs = C::POOL ? (a.ho >> 1) : a.ho, ws = C::POOL ? (a.wo >> 1) : a.wo;

    // Accumulators are only ever defined by inline asm with an AGPR constraint: a C++ "acc = 0" makes the loop-carried
    // value a VGPR-class phi and hipcc then copies all 256 registers AGPR <-> VGPR around every unit.  Clearing = one MFMA
    // with zero operands and a zero C per accumulator (0*0 + 0, exact).
    dcx_f32x16 acc[16];
    auto clear_acc = [&]() {
        const float fz = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p)   // s_nop 1: `fz` may have been written by the VALU instruction just before (VALU -> MFMA
                                       // operand needs 2 wait states; hipcc pads nothing inside or around an asm statement)
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %1, 0" : "=a"(acc[p]) : "v"(fz));
    };
    clear_acc();

    // ---- prologue: first unit staged synchronously ---------------------------------------------
    DcxItem cur = decode(w);
    int c = 0;
    float4 a_c[DQ];
    {
        const unsigned wb = unit_wbase(cur, 0);
#pragma unroll
        for (int d = 0; d < DQ; ++d) a_c[d] = load_a(wb, d);
        const __amdgpu_buffer_rsrc_t r0 = unit_rsrc(cur, 0);
        const int sy0 = cur.ty * C::TH - a.pad, sx0 = cur.tx * C::TW - a.pad;
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) {
            const int ly = sy0 + r_hy[k], lx = sx0 + r_hx[k];
            const bool inb = C::G > 1 ? (cur.n * C::G + r_hy[k] < a.n)
                                      : ((unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl);
            sR[tid + k * C::NTHREADS] = stage_fetch(r0, inb ? r_rel[k] : 0x80000000u);
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 21; ++x) xform_event(sB, x);
    }

    int u = 0;
    // One unit.  `zero_t`: first unit of a work item -- the first MFMA on each accumulator takes C = 0.  The item loop below
    // runs the first unit and the remaining units from two separate copies of this body (a peeled loop, not a diamond).
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
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[4 + 3 * u] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[5 + 3 * u] = __builtin_amdgcn_s_memtime();

        const __amdgpu_buffer_rsrc_t rs_n = unit_rsrc(nxt, cn);
        const int nsy0 = nxt.ty * C::TH - a.pad, nsx0 = nxt.tx * C::TW - a.pad;
        const unsigned wb_cur = unit_wbase(cur, c);
        const unsigned wb_nxt = unit_wbase(nxt, cn);
        float4 aq[NQ + DQ], bq[NQ];
#pragma unroll
        for (int d = 0; d < DQ; ++d) aq[d] = a_c[d];
#pragma unroll
        for (int d = 0; d < DQB; ++d) bq[d] = load_b(buf, d);
        float4* vnext = sB + (buf ^ 1) * LDSF;
        float4 rv[ITER_R];
        const bool n_interior = tile_interior(nxt);
        unsigned roff[ITER_R];
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) roff[k] = r_rel[k];
        if (!n_interior) {
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) {
                const int ly = nsy0 + r_hy[k], lx = nsx0 + r_hx[k];
                const bool inb = C::G > 1 ? (nxt.n * C::G + r_hy[k] < a.n)
                                          : ((unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl);
                roff[k] = inb ? r_rel[k] : 0x80000000u;
            }
        }
        // Positions are processed in pairs (qa, qb = qa + 1): 8 MFMAs alternating between the two accumulators, one slot
        // (= one staging event) before each MFMA pair; slots 0 / 1 also fetch A, B of qa + DQ / qb + DQ.
#pragma unroll
        for (int qa = 0; qa < NQ; qa += 2) {
            const int qb = qa + 1;
            if (qa * 2 == C::E_XFORM) __syncthreads();      // the raw tile of the next unit is complete in sR
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                __builtin_amdgcn_sched_barrier(0);
                if (slot < 2) {
                    const int q = qa + slot + DQ, qb2 = qa + slot + DQB;
                    if (q < NQ) aq[q] = load_a(wb_cur, q);
                    else aq[q] = load_a(wb_nxt, q - NQ);
                    if (qb2 < NQ) bq[qb2] = load_b(buf, qb2);
                }
                {
                    const int e = qa * 2 + slot;          // staging event 0 .. 63
                    if (e >= C::E_RAW_LOAD && e < C::E_RAW_LOAD + ITER_R) rv[e - C::E_RAW_LOAD] = stage_fetch(rs_n, roff[e - C::E_RAW_LOAD]);
                    if (e >= C::E_RAW_STORE && e < C::E_RAW_STORE + ITER_R) sR[tid + (e - C::E_RAW_STORE) * C::NTHREADS] = rv[e - C::E_RAW_STORE];
                    if (e >= C::E_XFORM && e < C::E_XFORM + 21) xform_event(vnext, e - C::E_XFORM);
                }
                __builtin_amdgcn_sched_barrier(0);
                // MFMAs of this slot: register j = slot of qa, then of qb
                {
                    const int pa = qa & 15, pb = qb & 15;
                    const float4 aa = aq[qa], ba = bq[qa], ab = aq[qb], bb = bq[qb];
                    const float av0 = slot == 0 ? aa.x : slot == 1 ? aa.y : slot == 2 ? aa.z : aa.w;
                    const float bv0 = slot == 0 ? ba.x : slot == 1 ? ba.y : slot == 2 ? ba.z : ba.w;
                    const float av1 = slot == 0 ? ab.x : slot == 1 ? ab.y : slot == 2 ? ab.z : ab.w;
                    const float bv1 = slot == 0 ? bb.x : slot == 1 ? bb.y : slot == 2 ? bb.z : bb.w;
                    // inline asm pins the register classes: accumulators in AGPRs, operands in VGPRs.  With the builtin, hipcc
                    // treats the 512 registers as one pool under this kernel's pressure and shuffles accumulators through
                    // VGPRs and scratch inside the loop.  (Consecutive MFMAs alternate between two accumulators.)
                    // Hazards (hipcc pads nothing around an asm statement): the operands come straight from loads, which hipcc
                    // waits for; the only VALU-written candidates are the weight registers carried over from the previous
                    // unit (possible v_mov copies just before the loop) -> 2 wait states ahead of the unit's first MFMAs.
                    // C is always the previous D of the same accumulator (accumulate chain: no wait states).
                    // (A C = 0 variant of the first k-step chosen by a branch INSIDE the unit makes hipcc reconcile the two
                    //  variants by moving accumulators through VGPRs and scratch -- 535 spills; the peeled first unit below,
                    //  a separate copy of the whole unit body, does not.)
                    if (qa == 0 && slot == 0) asm volatile("s_nop 1");
                    if (ZERO && qa < 16 && slot == 0) {   // first touch of these two accumulators in this work item: C = 0
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "+a"(acc[pa]) : "v"(av0), "v"(bv0));
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "+a"(acc[pb]) : "v"(av1), "v"(bv1));
                    } else {
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[pa]) : "v"(av0), "v"(bv0));
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[pb]) : "v"(av1), "v"(bv1));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < DQ; ++d) a_c[d] = aq[NQ + d];

        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && u < 20) a.clk_probe[6 + 3 * u] = __builtin_amdgcn_s_memtime();
        if (!ZERO && c == nch - 1) {
            // ---- epilogue: 2-D output transform, BN, ReLU (, pool | head), store -----------------------
            // (the MFMAs are inline asm, so the compiler does not know the distance between the last 16-pass MFMA and the first
            //  instruction that reads an accumulator as srcB: 18 wait states required, the nops make 20 explicit)
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
            // re-define the accumulators here (empty asm, AGPR class): keeps every use of them inside this block
#pragma unroll
            for (int p = 0; p < 16; ++p) asm volatile("" : "+a"(acc[p]));
            const int oy0 = cur.ty * C::TH + 2 * qty, ox0 = cur.tx * C::TW + 2 * qtx;
            const unsigned plane = (unsigned)(hs * ws);
            const int cq_w0 = (cur.ct * C::COUT_TILE >> 2) + wm * 8;
            char* obase = reinterpret_cast<char*>(a.out)
                        + ((size_t)cur.n * C::G * a.out_cq_total + a.out_cq_off + cq_w0) * (size_t)plane * 16;
            const bool img_ok = C::G == 1 || cur.n * C::G + q_img < n_eff;
            const bool okr0 = qok && img_ok && oy0 < a.ho, okr1 = qok && img_ok && oy0 + 1 < a.ho;
            const bool okc0 = ox0 < a.wo, okc1 = ox0 + 1 < a.wo;
            const unsigned lane_off = (C::POOL ? ((unsigned)half * plane + (unsigned)((oy0 >> 1) * ws + (ox0 >> 1))) * 16u
                                               : ((unsigned)half * plane + (unsigned)(oy0 * ws + ox0)) * 16u)
                                    + (C::G > 1 ? (unsigned)q_img * (unsigned)a.out_cq_total * plane * 16u : 0u);
            float hsum[4] = {0.f, 0.f, 0.f, 0.f};
            // Output transform ON THE MATRIX CORES.  An fp32 MFMA occupies the vector ALU (tools/ubench/mfma_fill.hip,
            // mfma_overlap.hip: no VALU instruction of any wave overlaps it), so the 256 v_accvgpr_read + ~260 packed adds of
            // a VALU transform were pure serial time (~3,600 cycles per work item).  v_mfma_f32_4x4x1_16b_f32 (16 blocks of
            // 4 lanes, K = 1, 8 cycles) computes D_k(lane) = A(lane 4*(lane/4) + k) * B(lane) + C_k(lane): with B = one
            // accumulator register read STRAIGHT FROM ITS AGPR (srcB may be an AccVGPR) and A = the per-lane constant
            // T[k = lane % 4][p], sixteen of them chained over p give the lane its four outputs y[k] = sum_p T[k][p] m[p]
            // as an exact sequential fmaf chain (coefficients 0, +-1: every product is exact) -- no accumulator ever passes
            // through the vector ALU.  256 such MFMAs = 2,048 cycles per work item (probe: tools/ubench/mfma4x4_probe.hip;
            // dependent accumulations must be >= 4 instructions apart, hence the four interleaved chains of a cout quad).
            float cf[16];
            {
                const float4* tp = reinterpret_cast<const float4*>(sT + (lane & 3) * 16);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 t4 = tp[q4];
                    cf[4 * q4] = t4.x; cf[4 * q4 + 1] = t4.y; cf[4 * q4 + 2] = t4.z; cf[4 * q4 + 3] = t4.w;
                }
            }
#pragma unroll
            for (int gh = 0; gh < 2; ++gh) {     // two cout quads (8 accumulator registers, 8 chains) at a time
            dcx_f32x4 e[2][4];                    // e[g2][cc][k]: output k of cout 4*(2*gh+g2) + cc
#pragma unroll
            for (int p = 0; p < 16; ++p) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int r = 4 * (2 * gh + g2) + cc;
                        if (p == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(e[g2][cc]) : "v"(cf[0]), "a"(acc[0][r]));
                        else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(e[g2][cc]) : "v"(cf[p]), "a"(acc[p][r]));
                    }
                }
            }
            // 2-pass MFMA result -> VALU read needs wait states hipcc does not know about (asm); this statement also
            // (re)defines all eight results, so no read of them can be scheduled above it
            asm volatile("s_nop 7" : "+v"(e[0][0]), "+v"(e[0][1]), "+v"(e[0][2]), "+v"(e[0][3]),
                                     "+v"(e[1][0]), "+v"(e[1][1]), "+v"(e[1][2]), "+v"(e[1][3]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int g = 2 * gh + g2;
                const int cq = (cur.ct * C::COUT_TILE >> 2) + wm * 8 + 2 * g + half;
                const float4 al = sP[cq], be = sP[cq_pad + cq];
                // BN: an MFMA result holds the four outputs k of ONE cout, so y = fma(e, alpha[cout], beta2[cout]) is a
                // v_pk_fma_f32 over the pair (k, k+1) with alpha / beta2 broadcast by op_sel; ReLU (or the pooling max) then
                // writes the value where the float4 store over the quad's four couts wants it -- no register shuffling
                const dcx_f32x2 al01 = {al.x, al.y}, al23 = {al.z, al.w}, be01 = {be.x, be.y}, be23 = {be.z, be.w};
                dcx_f32x2 bn[4][2];     // [cc][k / 2]
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int kp = 0; kp < 2; ++kp) {
                        const dcx_f32x2 x = {e[g2][cc][2 * kp], e[g2][cc][2 * kp + 1]};
                        const dcx_f32x2 aa = cc < 2 ? al01 : al23, bb = cc < 2 ? be01 : be23;
                        if ((cc & 1) == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(bn[cc][kp]) : "v"(x), "v"(aa), "v"(bb));
                        else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(bn[cc][kp]) : "v"(x), "v"(aa), "v"(bb));
                    }
                float4 y[4];     // y[2*i + j] over the quad's four couts (un-ReLU'd; consumers below apply their max)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    y[k] = make_float4(bn[0][k >> 1][k & 1], bn[1][k >> 1][k & 1], bn[2][k >> 1][k & 1], bn[3][k >> 1][k & 1]);
                char* dst = obase + (size_t)((unsigned)(2 * g) * plane * 16u) + lane_off;
                if (C::POOL) {
                    float4 v;
                    v.x = dcx_vmax(dcx_vmax(dcx_vmax(y[0].x, y[1].x), dcx_vmax(y[2].x, y[3].x)), 0.f);
                    v.y = dcx_vmax(dcx_vmax(dcx_vmax(y[0].y, y[1].y), dcx_vmax(y[2].y, y[3].y)), 0.f);
                    v.z = dcx_vmax(dcx_vmax(dcx_vmax(y[0].z, y[1].z), dcx_vmax(y[2].z, y[3].z)), 0.f);
                    v.w = dcx_vmax(dcx_vmax(dcx_vmax(y[0].w, y[1].w), dcx_vmax(y[2].w, y[3].w)), 0.f);
                    if (okr0 && okc0 && cq < a.cout_quads) *reinterpret_cast<float4*>(dst) = v;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        y[k].x = dcx_vmax(y[k].x, 0.f); y[k].y = dcx_vmax(y[k].y, 0.f);
                        y[k].z = dcx_vmax(y[k].z, 0.f); y[k].w = dcx_vmax(y[k].w, 0.f);
                    }
                    if (C::EPI == DCX_EPI_HEAT) {
                        if (cq >= a.cout_quads) continue;   // padded couts contribute 0; the (uniform) branch also ends the basic
                                                            // block, which keeps hipcc from merging the two halves' register use
                        const float4 hw = sP[2 * cq_pad + cq];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float h = hsum[k];
                            h = fmaf(y[k].x, hw.x, h); h = fmaf(y[k].y, hw.y, h);
                            h = fmaf(y[k].z, hw.z, h); h = fmaf(y[k].w, hw.w, h);
                            hsum[k] = h;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (cq < a.cout_quads) {
                        if (okr0 && okc0) *reinterpret_cast<float4*>(dst) = y[0];
                        if (okr0 && okc1) *reinterpret_cast<float4*>(dst + 16) = y[1];
                        if (okr1 && okc0) *reinterpret_cast<float4*>(dst + (size_t)ws * 16) = y[2];
                        if (okr1 && okc1) *reinterpret_cast<float4*>(dst + (size_t)ws * 16 + 16) = y[3];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);    // keep the second half's AGPR reads out of the first half (register pressure)
            }   // gh
            if (C::EPI == DCX_EPI_HEAT) {
                // logit = ((h[wm0,half0] + h[wm0,half1]) + (h[wm1,half0] + h[wm1,half1])) + bias, as in the 1-D kernel
                float t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = hsum[k] + __shfl_xor(hsum[k], 32);
                __syncthreads();
                float* scr = reinterpret_cast<float*>(sB + buf * LDSF);
                if (wm == 1 && half == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) scr[(wn * 32 + l31) * 4 + k] = t[k];
                }
                __syncthreads();
                float best = -INFINITY;
                int besti = 0x7fffffff;
                if (wm == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lg = (t[k] + scr[(wn * 32 + l31) * 4 + k]) + a.head_b;
                        const int sy = oy0 + (k >> 1), sx = ox0 + (k & 1);
                        const bool ok = qok && sy < a.ho && sx < a.wo;
                        if (ok) {
                            const int idx = sy * a.wo + sx;
                            if (a.heat != nullptr && half == 0) a.heat[((size_t)cur.n * a.ho + sy) * a.wo + sx] = lg;
                            if (lg > best || (lg == best && idx < besti)) { best = lg; besti = idx; }
                        }
                    }
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        const float ov = __shfl_xor(best, off);
                        const int oi = __shfl_xor(besti, off);
                        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                    }
                }
                float* red_v = scr + 512;
                int* red_i = reinterpret_cast<int*>(scr) + 512 + 16;
                if (wm == 0 && lane == 0) { red_v[wn] = best; red_i[wn] = besti; }
                __syncthreads();
                if (tid == 0) {
                    const float ov = red_v[1];
                    const int oi = red_i[1];
                    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                    a.part_val[(size_t)cur.n * tiles + cur.ty * a.tiles_x + cur.tx] = best;
                    a.part_idx[(size_t)cur.n * tiles + cur.ty * a.tiles_x + cur.tx] = besti;
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
    // work items: first unit (C = 0), then the remaining nch-1 units (the last one runs the epilogue); nch >= 2
    for (;;) {
        run_unit(std::true_type{});
        bool more = true;
        while (c != 0 && more) more = run_unit(std::false_type{});
        if (!more) break;
    }
}

template <class C>
static int dcx_conv_wino2_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    a.tiles_x = (a.wo + C::TW - 1) / C::TW;
    a.tiles_y = (a.ho + C::TH - 1) / C::TH;
    if (a.w_wino2 == nullptr || a.alpha == nullptr || a.beta == nullptr) return DCX_E_ARG;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0 || a.cin < 2 * DCX_CCH) return DCX_E_SHAPE;   // >= 2 units per work item
    if (C::EPI == DCX_EPI_HEAT && (a.head_w == nullptr || a.part_val == nullptr || a.part_idx == nullptr)) return DCX_E_ARG;
    if (C::EPI != DCX_EPI_HEAT && a.out == nullptr) return DCX_E_ARG;
    if (C::G > 1 && (a.tiles_x != 1 || a.tiles_y != 1 || a.ups != 0 || a.hin + 2 > C::HH || a.win + 2 > C::RW)) return DCX_E_SHAPE;
    const long items = (long)((a.n + C::G - 1) / C::G) * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y;
    if (items <= 0 || items > 0x7fffffffL) return DCX_E_SHAPE;
    const long resident = (long)dcx_device_cu_count();      // one workgroup per CU
    const long blocks = items < resident ? items : resident;
    a.xcd_walk = dcx_xcd_walk_enabled() && blocks == resident && (resident & 7) == 0 ? 1 : 0;
    const size_t lds = C::LDS_BYTES + (size_t)a.cout_pad * 12 + 256;      // + alpha, beta2, head weights, output-transform table
    if (lds > 160 * 1024) return DCX_E_SHAPE;
    static bool attr_set[DCX_MAX_DEVICES] = {};      // the attribute is per device (multi-GPU processes)
    const int dev_i = dcx_current_device();
    if (!attr_set[dev_i]) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_wino2_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev_i] = true;
    }
    hipLaunchKernelGGL((dcx_conv_wino2_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), lds, stream, a);
    return (int)hipGetLastError();
}
