// EXPLORATORY prototype -- NOT part of the product, never on the measured path.
//
// conv1b of the detector (64 -> 64 channels, 3x3, pad 1, + BN + ReLU left out: the raw accumulators are compared) computed on
// the BF16 matrix pipe with every fp32 operand split into three bf16 terms:  a = a1 + a2 + a3 exactly (3 x 8 mantissa bits),
// same for the weights; six cross products  a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1  accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (the dropped terms a2b3, a3b2, a3b3 are <= 2^-24 relative).  The bf16 pipe is 16x faster than the fp32
// pipe (MI355X_MICROARCH.md: 2.5 PFLOP/s vs 157.3 TFLOP/s dense), so 6 bf16 MFMAs cost 6 x 32 = 192 matrix cycles where the fp32
// direct kernel spends 8 x 64 = 512 on the same 32x32x16 block: 2.67x fewer matrix cycles at (almost) fp32 accuracy.
//
// This program: (1) splits a seeded fp32 activation tensor and the weights, (2) runs a simple implicit-GEMM kernel (no LDS:
// operands straight from L1/L2, 64 couts x 64 pixels per wave, zero-padded input so the k-loop has no predicates), (3) compares
// with an fp64 evaluation of the same fp32 inputs next to a sequential fp32 fmaf chain in the direct kernel's order, (4) times it.
// It answers two questions only: what is the REAL accumulation error of the bf16 MFMA path, and is the rate in the right ballpark.
//   build: hipcc -O3 --offload-arch=gfx950 tools/ubench/bf16x3_conv.hip -o tools/ubench/bf16x3_conv      run: tools/ubench/bf16x3_conv [frames]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int C = 64;          // cin = cout
constexpr int H = 240, W = 320;
constexpr int HP = H + 2, WP = W + 2;     // zero-padded copy of the split activations

static inline uint16_t f2bf(float x) {    // round to nearest even
    uint32_t u; memcpy(&u, &x, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float x, uint16_t out[3]) {
    out[0] = f2bf(x); float r = x - bf2f(out[0]);
    out[1] = f2bf(r); r = r - bf2f(out[1]);
    out[2] = f2bf(r);
}

// device-side split of the activations: x fp32 NHWC [n][H][W][C] -> xs bf16 [n][HP][WP][C/16][3][16] (border = 0)
__device__ inline uint16_t d_f2bf(float x) {
    uint32_t u = __float_as_uint(x);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
__global__ void split_kernel(const float* __restrict__ x, uint16_t* __restrict__ xs, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one (pixel, channel)
    if (i >= (long)n * H * W * C) return;
    const int c = i % C; long p = i / C;
    const int xx = p % W; p /= W;
    const int yy = p % H; const int b = p / H;
    const float v = x[i];
    const uint16_t b1 = d_f2bf(v);
    float r = v - __uint_as_float((uint32_t)b1 << 16);
    const uint16_t b2 = d_f2bf(r);
    r = r - __uint_as_float((uint32_t)b2 << 16);
    const uint16_t b3 = d_f2bf(r);
    const long base = ((((long)b * HP + yy + 1) * WP + xx + 1) * (C / 16) + c / 16) * 48 + (c % 16);
    xs[base] = b1; xs[base + 16] = b2; xs[base + 32] = b3;
}

// ws: bf16 [tap][C/16][3][cout][16];  out fp32 [n][H][W][cout] raw accumulators
// workgroup = 4 waves = 64 couts x (4 rows x 64 pixels); wave = 64 couts x 64 pixels of one row: 2 x 2 tiles of 32x32
template <int TERMS>
__global__ __launch_bounds__(256) void conv_bf16x3_kernel(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ws,
                                                          float* __restrict__ out, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int tiles_x = W / 64;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % (H / 4);
    const int b = t / (H / 4);
    const int y = ty * 4 + wave, x0 = tx * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int ch = 0; ch < C / 16; ++ch) {
            bf16x8 a[2][3], bb[2][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)      // A: weights, row = cout
                    a[i][s] = *reinterpret_cast<const bf16x8*>(ws + ((((long)tap * (C / 16) + ch) * 3 + s) * C + i * 32 + l31) * 16 + half * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j) {    // B: activations, column = pixel (padded coordinates: + dy, + dx)
                    const long px = ((long)b * HP + y + dy) * WP + x0 + j * 32 + l31 + dx;
                    bb[j][s] = *reinterpret_cast<const bf16x8*>(xs + (px * (C / 16) + ch) * 48 + s * 16 + half * 8);
                }
            }
            // six (TERMS = 6) or eight cross products, small ones first (everything unrolled: operands must stay in registers)
            constexpr int ia[8] = {1, 2, 1, 0, 2, 0, 1, 0}, ib[8] = {2, 1, 1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int q = 8 - TERMS; q < 8; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ia[q]], bb[j][ib[q]], acc[i][j], 0, 0, 0);
        }
    }
    // D lane l holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) {
                const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(((long)b * H + y) * W + x0 + j * 32 + l31) * C + co] = acc[i][j][r];
            }
}


// The same computation with the activation halo tile of a 16-channel chunk staged in LDS (6 rows x 66 pixels x 3 splits x 16
// channels; pixel stride 112 B so that the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots), weights from L2.
// Still a prototype (single-buffered, two barriers per chunk, no software pipelining) -- three workgroups per CU cover the stalls.
constexpr int PXB = 112;                       // bytes per pixel in LDS: 96 payload + 16 pad
__global__ __launch_bounds__(256, 3) void conv_bf16x3_lds_kernel(const uint16_t* __restrict__ xs, const uint16_t* __restrict__ ws,
                                                                 float* __restrict__ out, int n) {
    __shared__ __attribute__((aligned(16))) unsigned char sX[6 * 66 * PXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int tiles_x = W / 64;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % (H / 4);
    const int b = t / (H / 4);
    const int y0 = ty * 4, x0 = tx * 64;       // padded coordinates of the halo tile's top-left pixel
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int ch = 0; ch < C / 16; ++ch) {
        __syncthreads();
        for (int idx = tid; idx < 6 * 66 * 6; idx += 256) {
            const int px = idx / 6, part = idx - px * 6;
            const int r = px / 66, c = px - r * 66;
            const uint4 v = *reinterpret_cast<const uint4*>(xs + ((((long)b * HP + y0 + r) * WP + x0 + c) * (C / 16) + ch) * 48 + part * 8);
            *reinterpret_cast<uint4*>(sX + px * PXB + part * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            bf16x8 a[2][3], bb[2][3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i][s] = *reinterpret_cast<const bf16x8*>(ws + ((((long)tap * (C / 16) + ch) * 3 + s) * C + i * 32 + l31) * 16 + half * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bb[j][s] = *reinterpret_cast<const bf16x8*>(sX + ((wave + dy) * 66 + j * 32 + l31 + dx) * PXB + s * 32 + half * 16);
            }
            constexpr int ia[6] = {1, 0, 2, 0, 1, 0}, ib[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ia[q]], bb[j][ib[q]], acc[i][j], 0, 0, 0);
        }
    }
    const int y = y0 + wave;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) {
                const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(((long)b * H + y) * W + x0 + j * 32 + l31) * C + co] = acc[i][j][r];
            }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    const long npx = (long)n * H * W;
    std::vector<float> x(npx * C), w((size_t)C * C * 9);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.0f; };
    auto gauss = [&]() { float s = 0.f; for (int i = 0; i < 6; ++i) s += rnd(); return (s - 3.0f) * 1.41421356f; };
    for (auto& v : x) { const float g = gauss() * 0.8f + 0.2f; v = g > 0.f ? g : 0.f; }          // post-ReLU-like
    for (auto& v : w) v = gauss() * sqrtf(2.0f / (C * 9));                                        // OIHW -> stored [co][ci][tap]
    // split + pack the weights: [tap][C/16][3][cout][16]
    std::vector<uint16_t> wsplit((size_t)9 * (C / 16) * 3 * C * 16);
    for (int co = 0; co < C; ++co)
        for (int ci = 0; ci < C; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                uint16_t s3[3]; split3(w[((size_t)co * C + ci) * 9 + tap], s3);
                for (int s = 0; s < 3; ++s)
                    wsplit[((((size_t)tap * (C / 16) + ci / 16) * 3 + s) * C + co) * 16 + ci % 16] = s3[s];
            }
    float *d_x, *d_out; uint16_t *d_xs, *d_ws;
    const size_t xs_elems = (size_t)n * HP * WP * (C / 16) * 48;
    CHECK(hipMalloc(&d_x, x.size() * 4)); CHECK(hipMalloc(&d_out, x.size() * 4));
    CHECK(hipMalloc(&d_xs, xs_elems * 2)); CHECK(hipMalloc(&d_ws, wsplit.size() * 2));
    CHECK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ws, wsplit.data(), wsplit.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_xs, 0, xs_elems * 2));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long tot = npx * C;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, d_x, d_xs, n);
    CHECK(hipDeviceSynchronize());
    const unsigned grid = (unsigned)(n * (H / 4) * (W / 64));
    std::vector<float> got(x.size());
    const double flop = 2.0 * npx * C * C * 9;
    for (int terms : {6, 8, 60}) {           // 60: six terms, LDS-staged kernel
        auto launch = [&]() {
            if (terms == 6) hipLaunchKernelGGL(conv_bf16x3_kernel<6>, dim3(grid), dim3(256), 0, 0, d_xs, d_ws, d_out, n);
            else if (terms == 8) hipLaunchKernelGGL(conv_bf16x3_kernel<8>, dim3(grid), dim3(256), 0, 0, d_xs, d_ws, d_out, n);
            else hipLaunchKernelGGL(conv_bf16x3_lds_kernel, dim3(grid), dim3(256), 0, 0, d_xs, d_ws, d_out, n);
        };
        for (int it = 0; it < 3; ++it) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        const int reps = 10;
        for (int it = 0; it < reps; ++it) launch();
        CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
        float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CHECK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
        // check a sample of outputs against fp64 and against the fp32 fmaf chain of the direct kernel (chunk / tap / s / j order)
        double max_bf = 0, sum_bf = 0, max_f32 = 0, sum_f32 = 0; long cnt = 0;
        for (long p = 0; p < npx; p += 97) {
            const int xx = p % W, yy = (p / W) % H, b = (int)(p / ((long)W * H));
            for (int co = 0; co < C; co += 5) {
                double ref = 0.0; float chain = 0.f;
                for (int c0 = 0; c0 < C; c0 += 16)
                    for (int tap = 0; tap < 9; ++tap) {
                        const int iy = yy + tap / 3 - 1, ix = xx + tap % 3 - 1;
                        const bool inb = iy >= 0 && iy < H && ix >= 0 && ix < W;
                        for (int s = 0; s < 2; ++s) for (int j = 0; j < 4; ++j) for (int k = 0; k < 2; ++k) {
                            const int ci = c0 + 8 * s + 4 * k + j;
                            const float xv = inb ? x[(((long)b * H + iy) * W + ix) * C + ci] : 0.f;
                            const float wv = w[((size_t)co * C + ci) * 9 + tap];
                            ref += (double)xv * (double)wv;
                            chain = fmaf(wv, xv, chain);
                        }
                    }
                const double ebf = fabs((double)got[p * C + co] - ref), ef = fabs((double)chain - ref);
                if (ebf > max_bf) max_bf = ebf;
                if (ef > max_f32) max_f32 = ef;
                sum_bf += ebf; sum_f32 += ef; ++cnt;
            }
        }
        printf("bf16x3 %d terms%s: %8.3f ms for %d frames  %7.1f TFLOP/s (fp32-equivalent, algorithmic)  | error vs fp64 over %ld outputs: "
               "bf16x3 max %.3e mean %.3e   fp32 fmaf chain (direct kernel's order) max %.3e mean %.3e\n",
               terms == 60 ? 6 : terms, terms == 60 ? " (LDS-staged)" : "", ms, n, flop / (ms * 1e-3) / 1e12, cnt, max_bf, sum_bf / cnt, max_f32, sum_f32 / cnt);
    }
    return 0;
}
