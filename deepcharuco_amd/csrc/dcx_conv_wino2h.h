// dcx_conv_wino2h.h -- 3x3 convolution + BN + ReLU (+2x2 max-pool) as the 2-D Winograd transform F(2x2, 3x3) on the gfx950
// fp32 matrix cores: 16 products per 2x2 output tile and input channel instead of 36, i.e. 4/9 of the direct convolution's
// MFMAs.  THE kernel family of every 3x3 + BN + ReLU layer that does not read through an up-sampling (those: dcx_conv_wino2p.h);
// the family is chosen from the layer shape alone (dcx_conv_mfma.hip: pick), so a frame's results do not depend on the batch it
// travels in.  Read dcx_conv_mfma.h first; layouts and the persistent work walk are the same.
//
//   per 2x2 output tile (rows 2ty, 2ty+1; columns 2tx, 2tx+1) and input channel, d = the 4x4 input window whose top-left
//   pixel is (2ty - pad, 2tx - pad):
//     rows     t[xi][c] :  t0 = d[0][c]-d[2][c]   t1 = d[1][c]+d[2][c]   t2 = d[2][c]-d[1][c]   t3 = d[1][c]-d[3][c]
//     columns  v[xi][nu]:  v0 = t[xi][0]-t[xi][2] v1 = t[xi][1]+t[xi][2] v2 = t[xi][2]-t[xi][1] v3 = t[xi][1]-t[xi][3]
//     weights  u = G g G^T, transformed on the host in fp32 (rows first, then columns; h1 = ((h0+h1)+h2)*0.5f ...)
//     m[xi][nu] += u[xi][nu] * v[xi][nu]
//     y[i][j] = 0;  for p = xi*4 + nu ascending:  y[i][j] = fmaf(AT[i][xi]*AT[j][nu], m[xi][nu], y[i][j]),
//               AT = [[1,1,1,0],[0,1,-1,-1]]  (a sequential fmaf chain over ALL 16 positions, zero coefficients included:
//               it runs on the matrix cores as sixteen chained v_mfma_f32_4x4x1 per accumulator register)
//
// GEMM view: 16 independent GEMMs (one per Winograd position), M = cout, N = 2x2 tiles, K = cin, on v_mfma_f32_16x16x4_f32.
//   workgroup = 4 waves = 64 couts x 32 2x2-tiles; wave wm owns couts 16 wm .. 16 wm + 15 for ALL 32 tiles (two 16-tile MFMA
//   blocks) and all 16 positions: 16 x 2 x 4 = 128 accumulator registers (AGPRs) -- so TWO workgroups share a CU: an fp32 MFMA
//   owns the vector ALU and a co-resident wave only ever runs while the other one stalls, which is exactly
//   what the barriers, LDS / L2 latencies, inter-unit bookkeeping and the epilogue's store tail need.
//   v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) B(4x16); lane l supplies A[l % 16][l / 16], B[l / 16][l % 16] and holds
//   D[4 (l / 16) + r][l % 16], r = 0..3.  With the C4 layouts a lane's float4 is channels 4g .. 4g+3 (g = l / 16) of its
//   cout (A: weights [pos][cin/4][cout_pad][4]) or of its tile (B: sV[pos][cq][tile]); MFMA j consumes component j, i.e.
//   channels {j, 4+j, 8+j, 12+j} of the 16-channel chunk.  One float4 per operand covers the whole chunk.
//   Summation order of every m (restated bit-exactly by oracle/conv_exact.c, dcx_oracle_conv_wino2h_exact):
//       m = 0;  for chunk c (16 cin) / j in 0..3 / g in 0..3:  m = fmaf(u[16c + 4g + j], v[16c + 4g + j], m)
//   It does not depend on the tile shape, the grouping, the batch size or the CU count.
//
// Tiles: 8x16 or 6x20 output pixels of one image, or (G = 2) two WHOLE 8x8 maps with their own zero borders (RefineNet after its
// pool).  Per unit (16 channels) a wave issues 16 positions x 8 MFMAs x 32 cycles = 4,096 matrix cycles; operands: one weight
// float4 (L2) and two activation float4 (LDS) per position.  Staging: the next unit's raw tile [img][cq][rows][cols] is loaded
// early in the unit (hardware out-of-range -> 0 for the padding), written to LDS in the middle, and after one extra barrier
// each thread transforms HALF a (cq, tile) piece (two of the four xi rows: 12 ds_read_b128, 32 v_pk_add_f32, 8 ds_write_b128),
// one event per MFMA group; the unit body is one basic block.  BN parameters of the lane's cout quad come straight from L2
// in the epilogue (requested ahead of the output-transform chain), so the cout count is not limited by LDS.
// hipcc notes (each a measured 2-10x slowdown, guarded by tests/test_host_logic.py::test_conv_kernels_do_not_spill): the MFMAs
// are inline asm with "+a" accumulators; accumulators are only ever defined by asm; the first unit of every work item is a
// PEELED copy of the unit body whose first MFMA on each accumulator takes C = 0; hazards hipcc does not pad around asm are
// padded by hand (2 wait states ahead of a unit's first MFMAs, 20 between the last MFMA and the epilogue's first read).
#pragma once
#include "dcx_conv_mfma.h"

#include <type_traits>

// schedule constants (overridable for sweeps: make EXTRA="-DDCX_W2H_E_STORE=11 -DDCX_W2H_E_XFORM=16")
#ifndef DCX_W2H_DQ
#define DCX_W2H_DQ 5
#endif
#ifndef DCX_W2H_DQB
#define DCX_W2H_DQB 2
#endif
// (round 6: 9 / 14 -> 13 / 18.  The raw tile of the next unit is requested at event 0 and stored 128 matrix cycles per event later:
//  at bs=1 -- three of a call's launches run this kernel at one workgroup per CU, operands from the MALL -- 1,150 cycles of lead left
//  every unit waiting: bs=1 protocol +2 %; bs=32 unchanged, 10,265 vs 10,293 fps over three alternating rounds; 16 / 20 spills)
#ifndef DCX_W2H_E_STORE
#define DCX_W2H_E_STORE 13
#endif
#ifndef DCX_W2H_E_XFORM
#define DCX_W2H_E_XFORM 18
#endif
// the same for the TB = 1 kernels (16-tile items: launches that cannot fill the chip -- one workgroup per CU, nothing hides a
// latency, and the operands come from the MALL / HBM rather than a warm L2): weights further ahead, the raw tile later.
// Measured on the reference's bs=1 protocol (tools/ab_bs1.sh, profiles/experiments/r05_bs1_small_launches.txt): 5 / 9 / 14 ->
// 10 / 14 / 18: +1.5 ... +2.5 %; 8 ... 14 positions ahead and stores at 14 ... 20 are all within the noise of each other
#ifndef DCX_W2H_DQ1
#define DCX_W2H_DQ1 10
#endif
#ifndef DCX_W2H_DQB1
#define DCX_W2H_DQB1 2
#endif
#ifndef DCX_W2H_E_STORE1
#define DCX_W2H_E_STORE1 14
#endif
#ifndef DCX_W2H_E_XFORM1
#define DCX_W2H_E_XFORM1 18
#endif
// The two workgroups of a CU take turns at wave priority: bit DCX_PRIO_FLIP of the 100 MHz real-time clock (12: every 41 us),
// inverted for the second-dispatched half of the grid.  Without it the SIMD arbiter prefers the OLDER workgroup, which finishes
// its items ~16 % earlier and leaves the other one alone at the end of the launch.  Measured at bs=32 (two runs each, same box):
// dominant kernel 0.750 -> 0.762 of peak, step 10,373 -> 10,461 fps; bit 11 / 13: +0.4 %; 0 = off.  The same rotation among the
// phase kernel's three workgroups gained nothing.  Speed only -- no effect on the bits.
#ifndef DCX_PRIO_FLIP
#define DCX_PRIO_FLIP 12
#endif
// TB = 1 work items issue their MFMAs position-pair-wise (see the unit body); 0 = one position at a time (the A/B of round 5)
#ifndef DCX_W2H_TB1_PAIRS
#define DCX_W2H_TB1_PAIRS 1
#endif

template <int TH_, int TW_, bool POOL_, int G_ = 1, int TB_ = 2>
struct DcxWino2hCfg {
    static constexpr int TH = TH_, TW = TW_;
    static constexpr int G = G_;                           // images per work item: G = 2 packs two whole small maps (<= TH x TW each)
    static constexpr bool POOL = POOL_;
    static constexpr int NTHREADS = 256;
    static constexpr int COUT_TILE = 64;
    static constexpr int TY = TH / 2, TX = TW / 2;
    static constexpr int TPI = TY * TX;                    // 2x2 tiles per image region
    static constexpr int NTILES = G * TPI;                 // <= 16 TB
    static constexpr int TB = TB_;                         // 16-tile MFMA blocks per wave: 2 (32 tiles, 128 accumulators), or 1 for
                                                           // launches that cannot fill the chip anyway (one frame, 16 patches): a
                                                           // work item's serial MFMA chain is half as long (same bits, more items)
    static constexpr int HH = TH + 2, RW = TW + 2;
    static constexpr int CQC = DCX_CCH / 4;
    static constexpr int RAW = G * CQC * HH * RW;          // [image][cq][row][col]
    static constexpr int ITER_R = (RAW + NTHREADS - 1) / NTHREADS;
    // Raw tile in LDS: row pitch RP, row hy shifted by ROWOFF(hy) slots.  The transform reads float4 (row 2 ty + i, col 2 tx + j)
    // in 16-lane groups (16 consecutive tiles of one cq); with the plain layout those land on few of the sixteen 16-B slots of a
    // 256-B bank row (4-way conflicts at TW = 16: 35 % of the kernel's LDS cycles).  Per tile shape:
    //   TW = 16 (8 tiles per row, a group = two tile rows): pitch 20, one slot of shift on every second row PAIR -> 16 distinct slots;
    //   TW = 20 (10 tiles per row): rounds 2-3 used pitch 23 + the same shift (2 lanes per slot at best: 17 % conflict cycles); see PLANAR below;
    //   8x8 maps, G = 2 (4 tiles per row, a group = one whole map): pitch 12, one slot of shift on every second PAIR of row pairs
    //           -> the four tile rows start at residues 0, 8, 1, 9: 16 distinct slots.
    //   TW = 20 since round 4: EVEN / ODD COLUMN PLANES.  Every lane of a transform read wants the same column parity (column
    //           2 tx + j), so inside a parity plane consecutive tiles sit on consecutive slots; plane row pitch 13 puts the three tile
    //           rows of a lane group 10 and 4 slots apart (mod 16): with the hardware's ds_read_b128 lane groups (tiles {0-3, 12-15,
    //           20-27} / {4-11, 16-19, 28-31}) all 16 slots of a group differ; odd plane at +108 (= 4 mod 8) keeps the raw tile's
    //           ds_write_b128 (8 consecutive columns) conflict-free.  tools/lds_sim.py: 17.0 % of the kernel's LDS cycles -> 2.2 %.
    static constexpr bool PLANAR = TW_ == 20;
    static constexpr int PRP = 13, PODD = 108, PCQ = 216;  // planar: plane row pitch, offset of the odd-column plane, slots per channel quad
    static constexpr int RP = TW_ == 16 ? RW + 2 : TW_ == 20 ? PRP : TW_ == 8 ? RW + 2 : RW + 1;
    static constexpr int ROW_SHIFT = TW_ == 8 ? 2 : 1;     // ROWOFF(hy) = (hy >> ROW_SHIFT) & 1
    static constexpr int RAW_LDS = PLANAR ? G * CQC * PCQ : G * CQC * HH * RP;      // slot RP - 1 of row 0 is free in every layout -> dump slot
    // LDS slot (float4 index inside the raw tile) of raw pixel (hy, hx) of channel quad cq of image img
    __host__ __device__ static constexpr int raw_slot(int img, int cq, int hy, int hx) {
        return PLANAR ? (img * CQC + cq) * PCQ + (hx & 1) * PODD + hy * PRP + (hx >> 1)
                      : ((img * CQC + cq) * HH + hy) * RP + hx + ((hy >> ROW_SHIFT) & 1);
    }
    // offset of column + j relative to an EVEN column's slot in the same row (the transform reads columns 2 tx .. 2 tx + 3)
    __host__ __device__ static constexpr int col_step(int j) { return PLANAR ? (j & 1) * PODD + (j >> 1) : j; }
    static_assert(!PLANAR || (HH * PRP <= PODD && PODD + HH * PRP <= PCQ && (RW + 1) / 2 < PRP), "planar raw tile does not fit its planes");
    static constexpr int VPLANE = CQC * 32;                // float4 per position: [cq][tile]
    static constexpr int LDS_FLOAT4 = 16 * VPLANE;         // one transformed buffer (32 KB)
    static constexpr size_t LDS_BYTES = (size_t)(2 * LDS_FLOAT4 + RAW_LDS) * 16 + 256;     // + output-transform table
    static constexpr int DQ = TB_ == 1 ? DCX_W2H_DQ1 : DCX_W2H_DQ;   // weights: positions ahead
    static constexpr int DQB = TB_ == 1 ? DCX_W2H_DQB1 : DCX_W2H_DQB;   // transformed activations: positions ahead
    // staging schedule in events (two per position: 32 per unit, 128 matrix cycles apart)
    static constexpr int E_RAW_LOAD = 0;
    static constexpr int E_RAW_STORE = TB_ == 1 ? DCX_W2H_E_STORE1 : DCX_W2H_E_STORE;
    static constexpr int E_XFORM = TB_ == 1 ? DCX_W2H_E_XFORM1 : DCX_W2H_E_XFORM;   // mid barrier before this event; XF_EVENTS transform events follow
    static constexpr int XF_EVENTS = TB_ == 1 ? 8 : 12;               // TB = 1: quarter-piece transform (see the kernel)
    static_assert(TH % 2 == 0 && TW % 2 == 0 && NTILES <= 16 * TB && NTILES > 8 * TB && (TB == 1 || TB == 2), "tile must hold 17..32 (TB = 1: 9..16) 2x2 tiles");
    static_assert(ITER_R <= 5 && E_RAW_STORE + ITER_R <= E_XFORM && E_XFORM + XF_EVENTS <= 32 && DQ <= 16, "staging does not fit the schedule");
    static_assert(G == 1 || (!POOL && TPI == 16), "grouped tiles: two plain 8x8 maps");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups must fit a CU's LDS");
};

template <class C>
__global__ __launch_bounds__(256, 2) void dcx_conv_wino2h_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int TX = C::TX, RW = C::RW, ITER_R = C::ITER_R, LDSF = C::LDS_FLOAT4, CQC = C::CQC, VPLANE = C::VPLANE;
    constexpr int DQ = C::DQ, DQB = C::DQB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = 16-cout group
    const int g4 = lane >> 4, l15 = lane & 15;                      // lane's channel quad of the chunk / row-or-column

    // ---- work list (persistent, XCD-aware walk: see the header comment) ------------------------------------------
    const int tiles = a.tiles_x * a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    int n_eff = a.n;
    if (a.n_limit != nullptr) n_eff = min(n_eff, *a.n_limit);
    const int total = ((n_eff + C::G - 1) / C::G) * n_ct * tiles;      // G > 1: one work item covers G images (tiles == 1)
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
    // calibration launches (dcx_calibrate_xcd) run the dominant kernel's instantiation only: the other instantiations carry no
    // code for it (the unpooled ones sit at the SGPR limit: one more live pointer spilled a VGPR)
    constexpr bool XSTAT = C::POOL && C::TW == 16 && C::TB == 2 && C::G == 1;
    if (XSTAT && a.xcd_stat != nullptr && blockIdx.x == 0 && tid == 0) a.xcd_stat[8] = __builtin_amdgcn_s_memrealtime();
#ifdef DCX_W2H_BLOCKTIMES      // tuning aid (tools/block_times.py): every workgroup's start / end time; overruns the launch's own probe slot
    if (a.clk_probe != nullptr && tid == 0) a.clk_probe[64 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    const int nch = a.cin / DCX_CCH;
    auto decode = [&](int wi) {
        DcxItem it;
        it.tx = wi % a.tiles_x; wi /= a.tiles_x;
        it.ty = wi % a.tiles_y; wi /= a.tiles_y;
        if (C::G == 1 && a.ct_outer) {
            it.n = wi % n_eff;
            it.ct = wi / n_eff;
        } else {
            it.ct = wi % n_ct;
            it.n = wi / n_ct;
        }
        it.ph = 0;
        return it;
    };
    const int hl = a.hin << a.ups, wl = a.win << a.ups;

    // ---- operand fetch -----------------------------------------------------------------------------------------
    // weights [pos][cin/4][cout_pad][4]: lane (r = l15, g = g4) reads cout wm*16 + r, channel quad g of the chunk
    const unsigned w_lane_off = (unsigned)(g4 * a.cout_pad + wm * 16 + l15) * 16u;
    const unsigned w_pos_stride = (unsigned)((a.cin >> 2) * a.cout_pad) * 16u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_wino2), (short)0, (int)(16u * w_pos_stride), 0x00020000);
    auto unit_wbase = [&](const DcxItem& it, int c) {
        return (unsigned)((c * CQC) * a.cout_pad + it.ct * C::COUT_TILE) * 16u;
    };
    auto load_a = [&](unsigned wbase, int pos) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane_off, wbase + (unsigned)pos * w_pos_stride, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // transformed activations sV[buf][pos][cq][tile]: lane (n = l15, g = g4) reads tile tb*16 + n, channel quad g
    const int tile_b = g4 * 32 + l15;
    auto load_b = [&](int buf, int pos, int tb) { return sB[buf * LDSF + pos * VPLANE + tile_b + tb * 16]; };

    // ---- staging: raw tile ---------------------------------------------------------------------------------------
    constexpr int RP = C::RP;
    int r_hyx[ITER_R], r_slot[ITER_R];       // (hy << 16 | hx) of the thread's raw pixels (G > 1: the image of the group) and their LDS slots
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
        r_hyx[k] = hy << 16 | hx;
        const int prow = ((hy - a.pad) >> a.ups) + a.pad, pcol = ((hx - a.pad) >> a.ups) + a.pad;
        r_rel[k] = idx < C::RAW ? ((unsigned)img * in_img_stride + (unsigned)((cq * a.hin + prow) * a.win + pcol)) * 16u : 0x80000000u;
        r_slot[k] = idx < C::RAW ? C::raw_slot(img, cq, hy, hx) : RP - 1;
        if (C::G > 1) {
            // a grouped tile always starts at pixel (0, 0) of its images: the zero-padding predicate is a per-piece constant
            const int ly = hy - a.pad, lx = hx - a.pad;
            if (!((unsigned)ly < (unsigned)(a.hin << a.ups) && (unsigned)lx < (unsigned)(a.win << a.ups))) r_rel[k] = 0x80000000u;
            r_hyx[k] = img;     // what the per-unit check needs: which image of the group the piece reads
        }
    }
    float4* sR = sB + 2 * LDSF;
    // transform piece of this thread.  TB = 2 (17..32 tiles): half h (xi rows 2h, 2h + 1) of (cq, tile); tiles past the end redo the
    // last tile.  TB = 1 (9..16 tiles): a QUARTER -- ONE xi row of (cq, tile), the row = the wave -- so that the 256 threads share
    // the 4 x 16 x 4 rows evenly instead of half of them repeating tile 15: 8 ds_read_b128, 16 packed ops, 4 ds_write_b128 per
    // thread and unit where the half-piece form takes 12 / 32 / 8 (the same fp32 operations on the same values: same bits).
    constexpr bool QP = C::TB == 1;
    const int x_h = __builtin_amdgcn_readfirstlane(tid >> 7);          // wave-uniform: waves 0, 1 -> xi 0, 1; waves 2, 3 -> xi 2, 3
    const int x_r = __builtin_amdgcn_readfirstlane(tid >> 6);          // QP: the wave's xi row
    const int x_cq = QP ? (tid >> 4) & 3 : (tid >> 5) & 3;
    const int x_tile = min(QP ? (tid & 15) : (tid & 31), C::NTILES - 1);
    const int x_img = x_tile / C::TPI, x_t = x_tile - x_img * C::TPI;
    const int x_ty = x_t / TX, x_tx = x_t - x_ty * TX;
    // rows of the half-piece, branch-free: first xi = row A - row B, second xi = row B + sgn * row C
    //   h = 0: xi 0 = d0 - d2 (A = 0, B = 2), xi 1 = d1 + d2 (C = 1, sgn = +1);  h = 1: xi 2 = d2 - d1 (A = 2, B = 1), xi 3 = d1 - d3 (C = 3, sgn = -1)
    // rows of the quarter-piece: xi = row A + sgn * row B:  xi 0 = d0 - d2, xi 1 = d1 + d2, xi 2 = d2 - d1, xi 3 = d1 - d3
    const int x_ia = QP ? (x_r == 0 ? 0 : x_r == 2 ? 2 : 1) : (x_h ? 2 : 0);
    const int x_ib = QP ? (x_r == 2 ? 1 : x_r == 3 ? 3 : 2) : (x_h ? 1 : 2);
    const int x_ic = x_h ? 3 : 1;
    // raw slots of the window's left pixel (column 2 tx, always even) in the rows this piece needs
    const int x_ra = C::raw_slot(x_img, x_cq, 2 * x_ty + x_ia, 2 * x_tx), x_rb = C::raw_slot(x_img, x_cq, 2 * x_ty + x_ib, 2 * x_tx),
              x_rc = C::raw_slot(x_img, x_cq, 2 * x_ty + x_ic, 2 * x_tx);
    const float x_sg = QP ? (x_r == 1 ? 1.f : -1.f) : (x_h ? -1.f : 1.f);
    const dcx_f32x2 x_sgn = {x_sg, x_sg};
    const int x_dst = QP ? (4 * x_r) * VPLANE + x_cq * 32 + x_tile
                         : (8 * x_h) * VPLANE + x_cq * 32 + x_tile;         // + local position * VPLANE
    auto unit_rsrc = [&](const DcxItem& it, int c) {
        const long tile_off = (long)(((it.ty * C::TH) >> a.ups) - a.pad) * a.win + (((it.tx * C::TW) >> a.ups) - a.pad);
        const float* base = a.in + (((size_t)it.n * C::G * a.in_cq_total + a.in_cq_off + (size_t)c * CQC) * (size_t)a.hin * a.win
                                    + tile_off) * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
    };
    auto tile_interior = [&](const DcxItem& it) {
        if (C::G > 1) return (it.n + 1) * C::G <= n_eff;   // all images of the group exist (spatial padding is folded into r_rel)
        const int sy0 = it.ty * C::TH - a.pad, sx0 = it.tx * C::TW - a.pad;
        return sy0 >= 0 && sx0 >= 0 && sy0 + C::HH <= hl && sx0 + RW <= wl;
    };
    auto stage_fetch = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
        const dcx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto sub4 = [](const float4& x, const float4& y) {
        const dcx_f32x2 lo = dcx_pk_sub(dcx_f32x2{x.x, x.y}, dcx_f32x2{y.x, y.y}), hi = dcx_pk_sub(dcx_f32x2{x.z, x.w}, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    auto add4 = [](const float4& x, const float4& y) {
        const dcx_f32x2 lo = dcx_pk_add(dcx_f32x2{x.x, x.y}, dcx_f32x2{y.x, y.y}), hi = dcx_pk_add(dcx_f32x2{x.z, x.w}, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    // Half-piece transform as 12 events (x = 0..11).  Rows needed: h = 0: xi 0 = d0 - d2, xi 1 = d1 + d2;  h = 1: xi 2 = d2 - d1, xi 3 = d1 - d3.
    //   event 0: read the two rows of the first xi (8 ds_read_b128)      event 1: read the third row
    //   event 3 + ms (ms = 0..7, local position = 4 * (xi - 2h) + nu): at nu == 0 form t (4 float4 ops), then the position
    //   (1 float4 op); its LDS write goes out one event later           event 11: last write
    // Quarter-piece (TB = 1): event 0 reads its two rows, events 3 .. 6 form the row's four positions, the last write is event 7.
    float4 xa[4], xb[4], xc[4], xt[4], xv;
    auto fma4s = [&](const float4& x, const float4& y) {      // y + sgn * x as two v_pk_fma_f32 (sgn = +-1: exactly y +- x)
        const dcx_f32x2 lo = __builtin_elementwise_fma(dcx_f32x2{x.x, x.y}, x_sgn, dcx_f32x2{y.x, y.y});
        const dcx_f32x2 hi = __builtin_elementwise_fma(dcx_f32x2{x.z, x.w}, x_sgn, dcx_f32x2{y.z, y.w});
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    };
    auto xform_event = [&](float4* vbuf, int x) {
        if (QP) {
            if (x == 0) {
#pragma unroll
                for (int cidx = 0; cidx < 4; ++cidx) { xa[cidx] = sR[x_ra + C::col_step(cidx)]; xb[cidx] = sR[x_rb + C::col_step(cidx)]; }
            } else if (x >= 3 && x <= 7) {
                const int nu = x - 3;
                if (nu > 0) vbuf[x_dst + (nu - 1) * VPLANE] = xv;
                if (nu == 0) {
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) xt[cidx] = fma4s(xb[cidx], xa[cidx]);      // row A + sgn * row B
                }
                if (nu < 4) xv = nu == 0 ? sub4(xt[0], xt[2]) : nu == 1 ? add4(xt[1], xt[2]) : nu == 2 ? sub4(xt[2], xt[1]) : sub4(xt[1], xt[3]);
            }
            return;
        }
        if (x == 0) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) { xa[cidx] = sR[x_ra + C::col_step(cidx)]; xb[cidx] = sR[x_rb + C::col_step(cidx)]; }
        } else if (x == 1) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) xc[cidx] = sR[x_rc + C::col_step(cidx)];
        } else if (x >= 3) {
            const int ms = x - 3;
            if (ms > 0) vbuf[x_dst + (ms - 1) * VPLANE] = xv;
            if (ms < 8) {
                const int xl = ms >> 2, nu = ms & 3;
                if (nu == 0) {
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) xt[cidx] = xl == 0 ? sub4(xa[cidx], xb[cidx]) : fma4s(xc[cidx], xb[cidx]);
                }
                xv = nu == 0 ? sub4(xt[0], xt[2]) : nu == 1 ? add4(xt[1], xt[2]) : nu == 2 ? sub4(xt[2], xt[1]) : sub4(xt[1], xt[3]);
            }
        }
    };

    // ---- output-transform table in LDS: T[k = 2i + j][p = 4 xi + nu] = AT[i][xi] * AT[j][nu] as [k][p] floats (256 B) ------
    float* sT = reinterpret_cast<float*>(sB + 2 * LDSF + C::RAW_LDS);
    if (tid < 64) {
        const int k = tid >> 4, p = tid & 15, i = k >> 1, j = k & 1, xi = p >> 2, nu = p & 3;
        const int ci = i == 0 ? (xi < 3 ? 1 : 0) : (xi == 0 ? 0 : xi == 1 ? 1 : -1);
        const int cj = j == 0 ? (nu < 3 ? 1 : 0) : (nu == 0 ? 0 : nu == 1 ? 1 : -1);
        sT[tid] = (float)(ci * cj);
    }
    const int hs = C::POOL ? (a.ho >> 1) : a.ho, ws = C::POOL ? (a.wo >> 1) : a.wo;

    // accumulators: acc[pos][tb], only ever defined by inline asm with an AGPR constraint (see the header comment)
    constexpr int TB = C::TB;
    dcx_f32x4 acc[16][TB];

    // ---- prologue: first unit staged synchronously ---------------------------------------------------------------
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
            const int ly = sy0 + (r_hyx[k] >> 16), lx = sx0 + (r_hyx[k] & 0xffff);
            const bool inb = C::G > 1 ? (cur.n * C::G + r_hyx[k] < n_eff)
                                      : ((unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl);
            sR[r_slot[k]] = stage_fetch(r0, inb ? r_rel[k] : 0x80000000u);
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 12; ++x) xform_event(sB, x);
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
#if DCX_PRIO_FLIP > 0
        if (C::TB == 2) {
            // the SIMD arbiter prefers the OLDER of a CU's two workgroups, which then finishes its items ~16 % earlier and leaves the
            // other one alone at the end: take turns instead -- priority follows a bit of the real-time clock,
            // inverted for the second-dispatched half of the grid.  Not in the TB = 1 kernels: their launches are the ones that
            // cannot fill the chip (bs=1, 16 patches: one workgroup per CU, nobody to take turns with), where the clock read and
            // its s_waitcnt at the head of every unit are ~110 exposed cycles per 3,400-cycle unit (tools/unit_probe.py, PROBE_B=1).
            const unsigned rt = (unsigned)__builtin_amdgcn_s_memrealtime();     // (read one unit ahead of its use -- no s_waitcnt at the
                                                                                 //  unit head -- measured in round 5: within the noise)
            const bool hi = (((rt >> DCX_PRIO_FLIP) & 1u) != 0u) != (blockIdx.x >= (gridDim.x >> 1));
            if (hi) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
#endif
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && (unsigned)(u - a.probe_u0) < 20u) a.clk_probe[4 + 3 * (u - a.probe_u0)] = __builtin_amdgcn_s_memtime();
        __syncthreads();
        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && (unsigned)(u - a.probe_u0) < 20u) a.clk_probe[5 + 3 * (u - a.probe_u0)] = __builtin_amdgcn_s_memtime();

        const __amdgpu_buffer_rsrc_t rs_n = unit_rsrc(nxt, cn);
        const int nsy0 = nxt.ty * C::TH - a.pad, nsx0 = nxt.tx * C::TW - a.pad;
        const unsigned wb_cur = unit_wbase(cur, c);
        const unsigned wb_nxt = unit_wbase(nxt, cn);
        float4 aq[16 + DQ], bq[16][TB];
#pragma unroll
        for (int d = 0; d < DQ; ++d) aq[d] = a_c[d];
#pragma unroll
        for (int d = 0; d < DQB; ++d)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) bq[d][tb] = load_b(buf, d, tb);
        float4* vnext = sB + (buf ^ 1) * LDSF;
        float4 rv[ITER_R];
        const bool n_interior = tile_interior(nxt);
        unsigned roff[ITER_R];
#pragma unroll
        for (int k = 0; k < ITER_R; ++k) roff[k] = r_rel[k];
        if (!n_interior) {
#pragma unroll
            for (int k = 0; k < ITER_R; ++k) {
                const int ly = nsy0 + (r_hyx[k] >> 16), lx = nsx0 + (r_hyx[k] & 0xffff);
                const bool inb = C::G > 1 ? (nxt.n * C::G + r_hyx[k] < n_eff)
                                          : ((unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl);
                roff[k] = inb ? r_rel[k] : 0x80000000u;
            }
        }
        // one position = 4 TB MFMAs (j = 0..3) in two slots of 2 TB; each slot is preceded by one staging event; the slot that
        // opens a position also fetches the operands of position p + DQ / p + DQB.
        // TB = 2: a slot's MFMAs alternate between the two tile blocks, so an accumulator is touched every other MFMA.
        // TB = 1: positions are taken in PAIRS -- slot s of a pair issues MFMA j = s of BOTH positions -- for the same reason:
        //   v_mfma_f32_16x16x4_f32 issues every 32 cycles but a dependent one (same accumulator) only after 40
        //   (MI355X_MICROARCH.md), and with one workgroup per CU (bs=1, 16 patches: every TB = 1 launch) nothing else fills the
        //   8-cycle bubble.  Round 5 until this change issued j = 0..3 of one position back to back: 2,560 instead of 2,048
        //   matrix cycles per unit.  The order of the MFMAs on ONE accumulator is unchanged, so the bits are.
        constexpr int PG = (TB == 1 && DCX_W2H_TB1_PAIRS) ? 2 : 1;       // positions per group
#pragma unroll
        for (int pg = 0; pg < 16; pg += PG) {
#pragma unroll
            for (int slot = 0; slot < 2 * PG; ++slot) {
                const int e = pg * 2 + slot;                // staging event 0 .. 31
                if (e == C::E_XFORM) __syncthreads();       // the raw tile of the next unit is complete in sR
                __builtin_amdgcn_sched_barrier(0);
                if ((slot & 1) == 0) {
                    const int p = pg + (slot >> 1);
                    const int q = p + DQ, qb2 = p + DQB;
                    if (q < 16) aq[q] = load_a(wb_cur, q);
                    else aq[q] = load_a(wb_nxt, q - 16);
                    if (qb2 < 16) {
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb) bq[qb2][tb] = load_b(buf, qb2, tb);
                    }
                }
                {
                    if (e >= C::E_RAW_LOAD && e < C::E_RAW_LOAD + ITER_R) rv[e - C::E_RAW_LOAD] = stage_fetch(rs_n, roff[e - C::E_RAW_LOAD]);
                    if (e >= C::E_RAW_STORE && e < C::E_RAW_STORE + ITER_R) sR[r_slot[e - C::E_RAW_STORE]] = rv[e - C::E_RAW_STORE];
                    if (e >= C::E_XFORM && e < C::E_XFORM + C::XF_EVENTS) xform_event(vnext, e - C::E_XFORM);
                }
                __builtin_amdgcn_sched_barrier(0);
                // (hazards: see the header comment -- operands come from loads hipcc waits for; VALU-written candidates only
                //  at the very start of a unit -> 2 wait states ahead of the first MFMA)
                if (e == 0) asm volatile("s_nop 1");
                if (PG == 2) {
                    const int j = slot;
#pragma unroll
                    for (int pi = 0; pi < 2; ++pi) {
                        const int p = pg + pi;
                        const float4 aa = aq[p], bb = bq[p][0];
                        const float av = j == 0 ? aa.x : j == 1 ? aa.y : j == 2 ? aa.z : aa.w;
                        const float bv = j == 0 ? bb.x : j == 1 ? bb.y : j == 2 ? bb.z : bb.w;
                        if (ZERO && j == 0)             // first touch of this accumulator in this work item: C = 0
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[p][0]) : "v"(av), "v"(bv));
                        else
                            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p][0]) : "v"(av), "v"(bv));
                    }
                } else {
                    const int p = pg;
                    const float4 aa = aq[p];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * slot + jj;
                        const float av = j == 0 ? aa.x : j == 1 ? aa.y : j == 2 ? aa.z : aa.w;
#pragma unroll
                        for (int tb = 0; tb < TB; ++tb) {
                            const float4 bb = bq[p][tb];
                            const float bv = j == 0 ? bb.x : j == 1 ? bb.y : j == 2 ? bb.z : bb.w;
                            if (ZERO && j == 0)         // first touch of this accumulator in this work item: C = 0
                                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc[p][tb]) : "v"(av), "v"(bv));
                            else
                                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[p][tb]) : "v"(av), "v"(bv));
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < DQ; ++d) a_c[d] = aq[16 + d];

        if (a.clk_probe != nullptr && blockIdx.x == 0 && tid == 0 && (unsigned)(u - a.probe_u0) < 20u) a.clk_probe[6 + 3 * (u - a.probe_u0)] = __builtin_amdgcn_s_memtime();
        if (c == nch - 1 && (!ZERO || nch == 1)) {
            // ---- epilogue: output transform on the matrix cores, BN, ReLU (, pool), store -----------------------------
            asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");     // 8-pass MFMA result -> read as srcB: make the distance explicit
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) asm volatile("" : "+a"(acc[p][tb]));
            const unsigned plane = (unsigned)(hs * ws);
            const int cq = (cur.ct * C::COUT_TILE >> 2) + wm * 4 + g4;          // the lane's output channel quad
            // BN parameters of the quad straight from L2 (arrays are padded to cout_pad); the ~2,100 cycles of the output
            // transform below cover the latency
            const float4 al = reinterpret_cast<const float4*>(a.alpha)[cq], be = reinterpret_cast<const float4*>(a.beta)[cq];
            char* obase = reinterpret_cast<char*>(a.out)
                        + ((size_t)cur.n * C::G * a.out_cq_total + a.out_cq_off + cq) * (size_t)plane * 16;
            float cf[16];
            {
                const float4* tp = reinterpret_cast<const float4*>(sT + (lane & 3) * 16);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 t4 = tp[q4];
                    cf[4 * q4] = t4.x; cf[4 * q4 + 1] = t4.y; cf[4 * q4 + 2] = t4.z; cf[4 * q4 + 3] = t4.w;
                }
            }
            dcx_f32x4 e[TB][4];       // e[tb][i][k]: output k of cout 4 * cq + i for the lane's tile tb * 16 + l15
#pragma unroll
            for (int p = 0; p < 16; ++p) {
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (p == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(e[tb][i]) : "v"(cf[0]), "a"(acc[0][tb][i]));
                        else asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(e[tb][i]) : "v"(cf[p]), "a"(acc[p][tb][i]));
                    }
                }
            }
            asm volatile("s_nop 7" : "+v"(e[0][0]), "+v"(e[0][1]), "+v"(e[0][2]), "+v"(e[0][3]));
            if (TB == 2) asm volatile("" : "+v"(e[TB - 1][0]), "+v"(e[TB - 1][1]), "+v"(e[TB - 1][2]), "+v"(e[TB - 1][3]));
            const dcx_f32x2 al01 = {al.x, al.y}, al23 = {al.z, al.w}, be01 = {be.x, be.y}, be23 = {be.z, be.w};
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int qt = tb * 16 + l15;
                const int q_img = qt / C::TPI, q_t = qt - q_img * C::TPI;     // image inside the group (0 when G == 1)
                const int qty = q_t / TX, qtx = q_t - qty * TX;
                const int oy0 = cur.ty * C::TH + 2 * qty, ox0 = cur.tx * C::TW + 2 * qtx;
                const bool qok = qt < C::NTILES && cq < a.cout_quads && (C::G == 1 || cur.n * C::G + q_img < n_eff);
                char* const obase_i = obase + (C::G > 1 ? (size_t)q_img * a.out_cq_total * (size_t)plane * 16 : (size_t)0);
                const bool okr0 = qok && oy0 < a.ho, okr1 = qok && oy0 + 1 < a.ho;
                const bool okc0 = ox0 < a.wo, okc1 = ox0 + 1 < a.wo;
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
                for (int k = 0; k < 4; ++k)
                    y[k] = make_float4(bn[0][k >> 1][k & 1], bn[1][k >> 1][k & 1], bn[2][k >> 1][k & 1], bn[3][k >> 1][k & 1]);
                if (C::POOL) {
                    float4 v;
                    v.x = dcx_vmax(dcx_vmax(dcx_vmax(y[0].x, y[1].x), dcx_vmax(y[2].x, y[3].x)), 0.f);
                    v.y = dcx_vmax(dcx_vmax(dcx_vmax(y[0].y, y[1].y), dcx_vmax(y[2].y, y[3].y)), 0.f);
                    v.z = dcx_vmax(dcx_vmax(dcx_vmax(y[0].z, y[1].z), dcx_vmax(y[2].z, y[3].z)), 0.f);
                    v.w = dcx_vmax(dcx_vmax(dcx_vmax(y[0].w, y[1].w), dcx_vmax(y[2].w, y[3].w)), 0.f);
                    char* dst = obase_i + (size_t)((unsigned)((oy0 >> 1) * ws + (ox0 >> 1)) * 16u);
                    if (okr1 && okc1) *reinterpret_cast<float4*>(dst) = v;       // MaxPool2d(2,2) floors: a window needs both rows and columns
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        y[k].x = dcx_vmax(y[k].x, 0.f); y[k].y = dcx_vmax(y[k].y, 0.f);
                        y[k].z = dcx_vmax(y[k].z, 0.f); y[k].w = dcx_vmax(y[k].w, 0.f);
                    }
                    char* dst = obase_i + (size_t)((unsigned)(oy0 * ws + ox0) * 16u);
                    if (okr0 && okc0) *reinterpret_cast<float4*>(dst) = y[0];
                    if (okr0 && okc1) *reinterpret_cast<float4*>(dst + 16) = y[1];
                    if (okr1 && okc0) *reinterpret_cast<float4*>(dst + (size_t)ws * 16) = y[2];
                    if (okr1 && okc1) *reinterpret_cast<float4*>(dst + (size_t)ws * 16 + 16) = y[3];
                }
            }
        }

        if (!has_next) {
            if (XSTAT && a.xcd_stat != nullptr && tid == 0) {       // calibration launches (dcx_calibrate_xcd): per-XCD sum of the workgroups' end times
                atomicAdd(&a.xcd_stat[blockIdx.x & 7], (unsigned long long)__builtin_amdgcn_s_memrealtime());
                atomicAdd(&a.xcd_stat[9 + (blockIdx.x & 7)], 1ull);
            }
#ifdef DCX_W2H_BLOCKTIMES
            if (a.clk_probe != nullptr && tid == 0) a.clk_probe[65 + 2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
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
static int dcx_conv_wino2h_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    a.tiles_x = (a.wo + C::TW - 1) / C::TW;
    a.tiles_y = (a.ho + C::TH - 1) / C::TH;
    if (a.w_wino2 == nullptr || a.alpha == nullptr || a.beta == nullptr || a.out == nullptr) return DCX_E_ARG;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0 || a.cin < 2 * DCX_CCH) return DCX_E_SHAPE;   // >= 2 units per work item
    if (C::G > 1 && (a.tiles_x != 1 || a.tiles_y != 1 || a.ups != 0)) return DCX_E_SHAPE;                   // grouped tiles: whole maps
    const long items = (long)((a.n + C::G - 1) / C::G) * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y;
    if (items <= 0 || items > 0x7fffffffL) return DCX_E_SHAPE;
    const int occ_env = dcx_occupancy_override();                        // tuning knob (DCX_OCC = 1: one workgroup per CU)
    const long resident = (occ_env == 1 ? 1L : 2L) * dcx_device_cu_count();
    const long blocks = items < resident ? items : resident;
    a.xcd_walk = dcx_xcd_walk_enabled() && blocks == resident && (resident & 7) == 0 ? 1 : 0;
    dcx_fill_xcd_cum(a);
    {   // cout tile outermost where the layer's transformed weights would otherwise thrash the XCDs' L2 (DCX_CT_OUTER=0/1 forces it)
        static int force = -2;
        if (force == -2) { const char* e = getenv("DCX_CT_OUTER"); force = e ? atoi(e) : -1; }
        const size_t w_bytes = (size_t)16 * a.cin * a.cout_pad * 4;
        a.ct_outer = force >= 0 ? force : (a.xcd_walk && a.n_limit == nullptr && a.cout_pad / C::COUT_TILE >= 4 && w_bytes > (size_t)(2u << 20)) ? 1 : 0;
    }
    static bool attr_set[DCX_MAX_DEVICES] = {};
    const int dev_i = dcx_current_device();
    if (!attr_set[dev_i]) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_wino2h_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_set[dev_i] = true;
    }
    hipLaunchKernelGGL((dcx_conv_wino2h_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}
