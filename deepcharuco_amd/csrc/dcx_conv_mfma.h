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
//      contiguous per half-wave, and the 2x2 max-pool is two DPP-style lane exchanges.
//
// Both operands are fetched as float4 = 4 consecutive cin of one cout / one pixel:
//   lane (half = l>>5) reads channel quad (2s + half) of an 8-channel group s, register j of the
//   float4 feeds MFMA j, so one ds_read_b128 + one global_load_dwordx4 per tile feed 4 MFMAs
//   (256 cycles).  The resulting summation order is specified (and restated bit-exactly by
//   oracle/conv_exact.c):
//     acc = 0;  for chunk c (32 cin) / tap (dy-major) / s in 0..3 / j in 0..3:
//                  acc = fmaf(w[8s+j],   x[8s+j],   acc);     (k = 0 half)
//                  acc = fmaf(w[8s+4+j], x[8s+4+j], acc);     (k = 1 half)
//   i.e. an exact sequential fp32 fmaf chain (MFMA f32 numerics, cdna_hip_programming.md section 3).
//
// Activations: the (TH+2)x(TW+2) input halo tile of a 32-channel chunk is staged in LDS as
//   sB[cq][halo_pixel] float4 -- consecutive lanes read consecutive 16-B slots (conflict-free
//   ds_read_b128 without padding or swizzle).  Zero padding, the valid-conv offset and the nearest
//   x2 up-sampling of RefineNet are all resolved while staging, so the MFMA loop is identical for
//   every layer.  Weights ([tap][cin/4][cout][4], <= 1.2 MB, L2 resident) are streamed straight
//   from L2 into registers one step ahead of use; the 4 waves of a workgroup share them through L1.
#pragma once
#include "dcx_common.h"

typedef float dcx_f32x16 __attribute__((ext_vector_type(16)));

#define DCX_CCH 32  // input channels staged per LDS chunk

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
    static constexpr int CQC = DCX_CCH / 4;
    static constexpr int LDS_FLOAT4 = CQC * HALO;
    static constexpr size_t LDS_BYTES = (size_t)LDS_FLOAT4 * 16;
    static_assert(TILE_PIX <= CAP, "tile does not fit the wave layout");
    static_assert(!POOL || (TH % 2 == 0 && TW % 2 == 0), "pooled tiles must be even");
    static_assert(EPI != DCX_EPI_HEAT || (WM == 1 && !POOL), "heat epilogue needs all couts in one wave row");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS tile too large");
};

__device__ __forceinline__ float4 dcx_f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <class C>
__global__ __launch_bounds__(C::NTHREADS, 2) void dcx_conv_mfma_kernel(const DcxConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 sB[];
    constexpr int WN = C::WN, MT = C::MT, NT = C::NT, TW = C::TW, KS = C::KS;
    constexpr int HW = C::HW, HALO = C::HALO;
    constexpr int STEPS = KS * KS * (DCX_CCH / 8);  // MFMA k-steps of 8 channels per chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n_ct = a.cout_pad / C::COUT_TILE;
    const int ct = bid % n_ct;
    const int n = bid / n_ct;
    if (a.n_limit != nullptr && n >= *a.n_limit) return;

    const int oy0 = ty * C::TH, ox0 = tx * C::TW;   // tile origin in conv-output coordinates
    const int hl = a.hin << a.ups, wl = a.win << a.ups;  // logical (up-sampled) input size

    // ---- per-lane pixel of every n-tile -------------------------------------------------
    int pixb[NT];      // halo-tile pixel index of the lane's output pixel (tap 0,0)
    int qys[NT], qxs[NT];
    bool qok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int q = (wn * NT + nt) * 32 + l31;
        int qy, qx;
        if (C::POOL) {   // lanes 4i..4i+3 hold one 2x2 pooling window
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

    // ---- weights: lane reads float4 #(cq * cout_pad + cout) --------------------------------
    const int cq_total_in = a.cin >> 2;
    const float4* wlane = reinterpret_cast<const float4*>(a.w)
                        + (size_t)half * a.cout_pad + ct * C::COUT_TILE + wm * (MT * 32) + l31;
    auto load_a = [&](int c0, int step, float4 (&dst)[MT]) {
        const int tap = step / (DCX_CCH / 8);
        const int s = step - tap * (DCX_CCH / 8);
        const size_t off = ((size_t)tap * cq_total_in + (c0 >> 2) + 2 * s) * (size_t)a.cout_pad;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dst[mt] = wlane[off + mt * 32];
    };

    dcx_f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    float4 a_cur[MT], a_nxt[MT];
    load_a(0, 0, a_cur);

    const float4* in4 = reinterpret_cast<const float4*>(a.in);
    const size_t in_img = ((size_t)n * a.in_cq_total + a.in_cq_off) * (size_t)a.hin * a.win;

    for (int c0 = 0; c0 < a.cin; c0 += DCX_CCH) {
        // ---- stage the halo tile of channels [c0, c0+32) ---------------------------------
        __syncthreads();   // everyone finished reading the previous chunk
        {
            constexpr int TOTAL = C::LDS_FLOAT4;
            constexpr int ITER = (TOTAL + C::NTHREADS - 1) / C::NTHREADS;
            float4 v[ITER];
            const size_t chunk_base = in_img + (size_t)(c0 >> 2) * a.hin * a.win;
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                // branch-free: always load from a clamped (valid) address, then select zero
                const int idx = min(tid + it * C::NTHREADS, TOTAL - 1);
                const int cq = idx / HALO;
                const int hp = idx - cq * HALO;
                const int hy = hp / HW;
                const int hx = hp - hy * HW;
                const int ly = oy0 - a.pad + hy, lx = ox0 - a.pad + hx;
                const bool inb = (unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl;
                const int cy = min(max(ly, 0), hl - 1) >> a.ups;
                const int cx = min(max(lx, 0), wl - 1) >> a.ups;
                const float4 t = in4[chunk_base + ((size_t)cq * a.hin + cy) * a.win + cx];
                v[it] = inb ? t : dcx_f4_zero();
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int idx = tid + it * C::NTHREADS;
                if (idx < TOTAL) sB[idx] = v[it];
            }
        }
        __syncthreads();

        // ---- 9 taps x 4 channel-octets, 4*MT*NT MFMAs each ----------------------------------
#pragma unroll
        for (int step = 0; step < STEPS; ++step) {
            const int tap = step / (DCX_CCH / 8);
            const int s = step - tap * (DCX_CCH / 8);
            const int dy = tap / KS, dx = tap - dy * KS;
            if (step + 1 < STEPS) {
                load_a(c0, step + 1, a_nxt);
            } else if (c0 + DCX_CCH < a.cin) {
                load_a(c0 + DCX_CCH, 0, a_nxt);
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = a_cur[mt];
            }
            float4 b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = sB[pixb[nt] + (2 * s) * HALO + dy * HW + dx];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float av[4] = {a_cur[mt].x, a_cur[mt].y, a_cur[mt].z, a_cur[mt].w};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float bv[4] = {b[nt].x, b[nt].y, b[nt].z, b[nt].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc[mt][nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    const float4* bias4 = reinterpret_cast<const float4*>(a.bias);
    const float4* alpha4 = reinterpret_cast<const float4*>(a.alpha);
    const float4* beta4 = reinterpret_cast<const float4*>(a.beta);
    const int hs = C::POOL ? (a.ho >> 1) : a.ho, ws = C::POOL ? (a.wo >> 1) : a.wo;
    float4* out4 = reinterpret_cast<float4*>(a.out);

    float hsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) hsum[nt] = 0.f;

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cq = (ct * C::COUT_TILE >> 2) + (wm * MT + mt) * 8 + 2 * g + half;  // output channel quad
            const float4 bi = bias4[cq];
            float4 al = dcx_f4_zero(), be = dcx_f4_zero(), hw4 = dcx_f4_zero();
            if (C::EPI != DCX_EPI_RAW) { al = alpha4[cq]; be = beta4[cq]; }
            if (C::EPI == DCX_EPI_HEAT) hw4 = reinterpret_cast<const float4*>(a.head_w)[cq];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float4 v = make_float4(acc[mt][nt][4 * g + 0], acc[mt][nt][4 * g + 1],
                                       acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
                v.x += bi.x; v.y += bi.y; v.z += bi.z; v.w += bi.w;
                if (C::EPI != DCX_EPI_RAW) {
                    v.x = fmaxf(fmaf(v.x, al.x, be.x), 0.f);
                    v.y = fmaxf(fmaf(v.y, al.y, be.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, al.z, be.z), 0.f);
                    v.w = fmaxf(fmaf(v.w, al.w, be.w), 0.f);
                }
                if (C::EPI == DCX_EPI_HEAT) {
                    float h = hsum[nt];
                    h = fmaf(v.x, hw4.x, h); h = fmaf(v.y, hw4.y, h);
                    h = fmaf(v.z, hw4.z, h); h = fmaf(v.w, hw4.w, h);
                    hsum[nt] = h;
                    continue;
                }
                int sy = oy0 + qys[nt], sx = ox0 + qxs[nt];
                bool ok = qok[nt] && sy < a.ho && sx < a.wo && cq < a.cout_quads;
                if (C::POOL) {
                    float4 o;
                    o.x = __shfl_xor(v.x, 1); o.y = __shfl_xor(v.y, 1); o.z = __shfl_xor(v.z, 1); o.w = __shfl_xor(v.w, 1);
                    v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
                    o.x = __shfl_xor(v.x, 2); o.y = __shfl_xor(v.y, 2); o.z = __shfl_xor(v.z, 2); o.w = __shfl_xor(v.w, 2);
                    v.x = fmaxf(v.x, o.x); v.y = fmaxf(v.y, o.y); v.z = fmaxf(v.z, o.z); v.w = fmaxf(v.w, o.w);
                    ok = ok && (l31 & 3) == 0;
                    sy >>= 1; sx >>= 1;
                }
                if (ok) {
                    const size_t o = (((size_t)n * a.out_cq_total + a.out_cq_off + cq) * hs + sy) * (size_t)ws + sx;
                    out4[o] = v;
                }
            }
        }
    }

    if (C::EPI == DCX_EPI_HEAT) {
        // 1x1 conv to one channel (refinenet.py:81 convPb) + arg-max of this tile (model_utils.py:39-43)
        float best = -INFINITY;
        int besti = 0x7fffffff;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float tot = hsum[nt] + __shfl_xor(hsum[nt], 32);   // both halves: couts 4*half+{0..3} interleaved
            const float logit = tot + a.head_b;
            const int sy = oy0 + qys[nt], sx = ox0 + qxs[nt];
            const bool ok = qok[nt] && sy < a.ho && sx < a.wo;
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
        __syncthreads();   // all waves are done with sB
        float* red_v = reinterpret_cast<float*>(sB);
        int* red_i = reinterpret_cast<int*>(sB) + 16;
        if (lane == 0) { red_v[wave] = best; red_i[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < C::WM * C::WN; ++w) {
                const float ov = red_v[w];
                const int oi = red_i[w];
                if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
            }
            const int tiles = a.tiles_x * a.tiles_y;
            a.part_val[(size_t)n * tiles + ty * a.tiles_x + tx] = best;
            a.part_idx[(size_t)n * tiles + ty * a.tiles_x + tx] = besti;
        }
    }
}

template <class C>
static int dcx_conv_launch_cfg(DcxConvArgs a, hipStream_t stream) {
    a.tiles_x = (a.wo + C::TW - 1) / C::TW;
    a.tiles_y = (a.ho + C::TH - 1) / C::TH;
    if (a.cout_pad % C::COUT_TILE != 0 || a.cin % DCX_CCH != 0) return DCX_E_SHAPE;
    const long blocks = (long)a.n * (a.cout_pad / C::COUT_TILE) * a.tiles_x * a.tiles_y;
    if (blocks <= 0 || blocks > 0x7fffffffL) return DCX_E_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        DCX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dcx_conv_mfma_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((dcx_conv_mfma_kernel<C>), dim3((unsigned)blocks), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    return (int)hipGetLastError();
}
