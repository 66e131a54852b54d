// dcx_misc.hip -- the HBM-bound kernels around the MFMA convolutions:
//   * Cin=1 first layers (conv1a of both nets) with the u8 -> f32 normalisation fused in,
//   * per-cell arg-max + dust-bin mask + 8x8 decode + ordered compaction,
//   * patch table / 24x24 patch gather / heat-map arg-max finalisation,
//   * NCHW <-> C4 converters, stand-alone pre_image and argmax2d.
#include "dcx_common.h"

#include <stdlib.h>

// pre_bgr_image /root/reference/src/models/model_utils.py:46-50: (float(g) - 128) / 255, IEEE division
// (hipcc's default f32 division is correctly rounded; tests check all 256 inputs bit-for-bit).
__device__ __forceinline__ float dcx_norm_u8(uint8_t g) { return ((float)g - 128.0f) / 255.0f; }
__device__ __forceinline__ float dcx_load_px(const uint8_t* p) { return dcx_norm_u8(*p); }
__device__ __forceinline__ float dcx_load_px(const float* p) { return *p; }

// Pixel formats of the u8 frames (DCX_PIX_*): gray, or interleaved BGR converted in the load with the integer formula of
// cv2.cvtColor(img, COLOR_BGR2GRAY) (dcx_bgr2gray_kernel below: 15-bit OpenCV 4.x constants; legacy 14-bit ones) -- the gray
// frame of inference.py:40 is never materialised.  BPP = bytes per pixel.
template <int PIX> struct DcxPix;
template <> struct DcxPix<DCX_PIX_GRAY8> {
    static constexpr int BPP = 1;
    static __device__ __forceinline__ float load(const uint8_t* p) { return dcx_norm_u8(*p); }
};
template <> struct DcxPix<DCX_PIX_BGR8> {
    static constexpr int BPP = 3;
    static __device__ __forceinline__ float load(const uint8_t* p) {
        const unsigned bb = p[0], gg = p[1], rr = p[2];
        return dcx_norm_u8((uint8_t)((bb * 3735u + gg * 19235u + rr * 9798u + (1u << 14)) >> 15));
    }
};
template <> struct DcxPix<DCX_PIX_BGR8_LEGACY14> {
    static constexpr int BPP = 3;
    static __device__ __forceinline__ float load(const uint8_t* p) {
        const unsigned bb = p[0], gg = p[1], rr = p[2];
        return dcx_norm_u8((uint8_t)((bb * 1868u + gg * 9617u + rr * 4899u + (1u << 13)) >> 14));
    }
};
struct DcxPixF32 {
    static constexpr int BPP = 4;
    static __device__ __forceinline__ float load(const uint8_t* p) { return *reinterpret_cast<const float*>(p); }
};

// ---------------------------------------------------------------------------------------
// conv1a (+bn1a+relu), Cin = 1 -> 64: net.py:23-24,60 (pad 1) and refinenet.py:21-22,56 (pad 0).
// One thread = one output pixel, all 64 output channels; acc = fmaf(w[tap], x[tap], acc) for
// tap = 0..8 (dy-major), then +bias, BN affine, ReLU.  Write-bound: 256 B per pixel, C4 layout,
// 16-B stores contiguous across lanes.
// CS = channel split: 1 -> a thread computes all 64 output channels of its pixel (big batches: the kernel is HBM-write bound);
// 4 -> blockIdx.z picks 16 of them (one frame: 1,200 waves of 1,000 instructions each leave the chip latency-bound -- four times
// as many waves, each a quarter as long: 13.6 -> ~6 us at 320x240, bs=1).  Same arithmetic per output, same bits.
template <typename PX, int CS>
__global__ __launch_bounds__(256) void dcx_conv1_kernel(const uint8_t* __restrict__ in, long image_stride, int pitch,
                                                          int h, int w, int pad,
                                                          const float* __restrict__ w9x64,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ alpha,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ out, int ho, int wo,
                                                          const int* __restrict__ n_limit, int n_images,
                                                          int32_t* __restrict__ zero_words, int n_zero) {
    __shared__ __attribute__((aligned(16))) float sw[9 * 64 + 3 * 64];
    if (zero_words != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)      // image_stride / pitch are in BYTES
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0;
    int n_end = n_images;
    if (n_limit != nullptr) n_end = min(n_end, *n_limit);
    if ((int)blockIdx.y >= n_end) return;
    for (int i = threadIdx.x; i < 9 * 64; i += 256) sw[i] = w9x64[i];
    if (threadIdx.x < 64) {
        sw[576 + threadIdx.x] = bias[threadIdx.x];
        sw[640 + threadIdx.x] = alpha[threadIdx.x];
        sw[704 + threadIdx.x] = beta[threadIdx.x];
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= ho * wo) return;
    const int oy = p / wo, ox = p - oy * wo;
    for (int n = blockIdx.y; n < n_end; n += gridDim.y) {     // gridDim.y is capped at 65535 images
    const uint8_t* img = in + (size_t)n * image_stride;
    float x[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iy = oy - pad + dy, ix = ox - pad + dx;
            const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
            const int cy = min(max(iy, 0), h - 1), cx = min(max(ix, 0), w - 1);
            const float v = PX::load(img + (size_t)cy * pitch + (size_t)cx * PX::BPP);
            x[dy * 3 + dx] = inb ? v : 0.0f;   // zero padding of the NORMALISED image
        }
    float4* out4 = reinterpret_cast<float4*>(out);
    const float4* sw4 = reinterpret_cast<const float4*>(sw);
    const int cq_lo = CS == 1 ? 0 : (int)blockIdx.z * (16 / CS);
#pragma unroll 4
    for (int cq = cq_lo; cq < cq_lo + 16 / CS; ++cq) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 wv = sw4[t * 16 + cq];
            acc.x = fmaf(wv.x, x[t], acc.x);
            acc.y = fmaf(wv.y, x[t], acc.y);
            acc.z = fmaf(wv.z, x[t], acc.z);
            acc.w = fmaf(wv.w, x[t], acc.w);
        }
        const float4 bi = sw4[144 + cq], al = sw4[160 + cq], be = sw4[176 + cq];
        float4 y;
        y.x = fmaxf(fmaf(acc.x + bi.x, al.x, be.x), 0.f);
        y.y = fmaxf(fmaf(acc.y + bi.y, al.y, be.y), 0.f);
        y.z = fmaxf(fmaf(acc.z + bi.z, al.z, be.z), 0.f);
        y.w = fmaxf(fmaf(acc.w + bi.w, al.w, be.w), 0.f);
        out4[((size_t)n * 16 + cq) * (size_t)(ho * wo) + p] = y;
    }
    }
}

// conv1a for launches of one or two frames (the reference's bs=1 protocol): workgroup = a 16x16 tile of output pixels x 16 of the 64
// channels (blockIdx.z).  The tile's 18x18 input pixels are decoded ONCE into LDS (BGR -> gray, normalise: three byte loads, an
// integer dot product and an IEEE division per pixel) -- in dcx_conv1_kernel<PX, 4> every thread decodes its nine taps itself, 36
// decodes per output pixel over the four channel quarters, which at bs=1 is half of the kernel's instructions (12.2 -> ~9 us for a
// 320x240 BGR frame).  Same taps, same fmaf order, same bits.
template <typename PX>
__global__ __launch_bounds__(256) void dcx_conv1_tile_kernel(const uint8_t* __restrict__ in, long image_stride, int pitch,
                                                               int h, int w, int pad,
                                                               const float* __restrict__ w9x64,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ alpha,
                                                               const float* __restrict__ beta,
                                                               float* __restrict__ out, int ho, int wo, int tiles_y,
                                                               int32_t* __restrict__ zero_words, int n_zero) {
    __shared__ __attribute__((aligned(16))) float sw[9 * 64 + 3 * 64];
    __shared__ float sp[18 * 18];
    if (zero_words != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0;
    const int tid = threadIdx.x;
    const int n = (int)blockIdx.y / tiles_y, ty = (int)blockIdx.y - n * tiles_y, tx = (int)blockIdx.x;
    for (int i = tid; i < 9 * 64; i += 256) sw[i] = w9x64[i];
    if (tid < 64) {
        sw[576 + tid] = bias[tid];
        sw[640 + tid] = alpha[tid];
        sw[704 + tid] = beta[tid];
    }
    const uint8_t* img = in + (size_t)n * image_stride;
    for (int e = tid; e < 18 * 18; e += 256) {
        const int i = e / 18, j = e - i * 18;
        const int iy = ty * 16 - pad + i, ix = tx * 16 - pad + j;
        const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
        const int cy = min(max(iy, 0), h - 1), cx = min(max(ix, 0), w - 1);
        const float v = PX::load(img + (size_t)cy * pitch + (size_t)cx * PX::BPP);
        sp[e] = inb ? v : 0.0f;       // zero padding of the NORMALISED image
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15;
    const int oy = ty * 16 + ly, ox = tx * 16 + lx;
    if (oy >= ho || ox >= wo) return;
    float x[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) x[dy * 3 + dx] = sp[(ly + dy) * 18 + lx + dx];
    float4* out4 = reinterpret_cast<float4*>(out);
    const float4* sw4 = reinterpret_cast<const float4*>(sw);
    const int p = oy * wo + ox;
    const int cq_lo = (int)blockIdx.z * 4;
#pragma unroll
    for (int cq = cq_lo; cq < cq_lo + 4; ++cq) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 wv = sw4[t * 16 + cq];
            acc.x = fmaf(wv.x, x[t], acc.x);
            acc.y = fmaf(wv.y, x[t], acc.y);
            acc.z = fmaf(wv.z, x[t], acc.z);
            acc.w = fmaf(wv.w, x[t], acc.w);
        }
        const float4 bi = sw4[144 + cq], al = sw4[160 + cq], be = sw4[176 + cq];
        float4 y;
        y.x = fmaxf(fmaf(acc.x + bi.x, al.x, be.x), 0.f);
        y.y = fmaxf(fmaf(acc.y + bi.y, al.y, be.y), 0.f);
        y.z = fmaxf(fmaf(acc.z + bi.z, al.z, be.z), 0.f);
        y.w = fmaxf(fmaf(acc.w + bi.w, al.w, be.w), 0.f);
        out4[((size_t)n * 16 + cq) * (size_t)(ho * wo) + p] = y;
    }
}

// RefineNet conv1a for the pipeline: output pixel (oy, ox) of patch p reads patch pixels (oy + dy, ox + dx), i.e. image pixels
// (y - 12 + oy + dy, x - 12 + ox + dx) of frame table[p].x around key-point (x, y) = table[p].(y, z), zero outside the image
// (model_utils.py:19-36 pads the NORMALISED image with 0) -- the values dcx_gather_kernel would have written, the arithmetic of
// dcx_conv1_kernel (taps 0..8 dy-major, + bias, BN affine, ReLU): bit-identical to gather + conv1a.
template <typename PX>
__global__ __launch_bounds__(256) void dcx_conv1_patches_kernel(const uint8_t* __restrict__ frames, long frame_stride, int pitch,
                                                                  int h, int w, const int32_t* __restrict__ table,
                                                                  const int* __restrict__ total, int max_patches,
                                                                  const float* __restrict__ w9x64, const float* __restrict__ bias,
                                                                  const float* __restrict__ alpha, const float* __restrict__ beta,
                                                                  float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float sw[9 * 64 + 3 * 64];
    __shared__ float sp[24 * 24];                     // the normalised, zero-padded 24x24 patch (what extract_patches returns)
    // the first patch's table entry is requested together with the live-patch count (the slot exists for every blockIdx.y; its
    // content is only used if the patch is live): one L2 / HBM latency instead of two at the head of a latency-bound kernel
    int4 t = reinterpret_cast<const int4*>(table)[blockIdx.y];
    const int n_end = min(max_patches, *total);
    if ((int)blockIdx.y >= n_end) return;
    const int tid = threadIdx.x;
    const int cq0 = blockIdx.x * 4;                    // this workgroup's 16 output channels (4 workgroups per patch: the kernel is
                                                       // latency-bound, not bandwidth-bound -- more, shorter workgroups)
    for (int i = tid; i < 9 * 64; i += 256) sw[i] = w9x64[i];
    if (tid < 64) {
        sw[576 + tid] = bias[tid];
        sw[640 + tid] = alpha[tid];
        sw[704 + tid] = beta[tid];
    }
    const float4* sw4 = reinterpret_cast<const float4*>(sw);
    float4* out4 = reinterpret_cast<float4*>(out);
    for (int n = blockIdx.y; n < n_end; n += gridDim.y) {     // gridDim.y is capped at 65535 patches
        if (n != (int)blockIdx.y) t = reinterpret_cast<const int4*>(table)[n];
        const uint8_t* img = frames + (size_t)t.x * frame_stride;
        __syncthreads();                                       // previous patch's readers are done with sp
        for (int e = tid; e < 576; e += 256) {
            const int i = e / 24, j = e - i * 24;
            const int iy = t.z - 12 + i, ix = t.y - 12 + j;
            const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
            const int cy = min(max(iy, 0), h - 1), cx = min(max(ix, 0), w - 1);
            const float v = PX::load(img + (size_t)cy * pitch + (size_t)cx * PX::BPP);
            sp[e] = inb ? v : 0.0f;
        }
        __syncthreads();
        for (int p = tid; p < 22 * 22; p += 256) {
            const int oy = p / 22, ox = p - oy * 22;
            float x[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) x[dy * 3 + dx] = sp[(oy + dy) * 24 + ox + dx];
#pragma unroll
            for (int cq = cq0; cq < cq0 + 4; ++cq) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const float4 wv = sw4[tp * 16 + cq];
                    acc.x = fmaf(wv.x, x[tp], acc.x);
                    acc.y = fmaf(wv.y, x[tp], acc.y);
                    acc.z = fmaf(wv.z, x[tp], acc.z);
                    acc.w = fmaf(wv.w, x[tp], acc.w);
                }
                const float4 bi = sw4[144 + cq], al = sw4[160 + cq], be = sw4[176 + cq];
                float4 y;
                y.x = fmaxf(fmaf(acc.x + bi.x, al.x, be.x), 0.f);
                y.y = fmaxf(fmaf(acc.y + bi.y, al.y, be.y), 0.f);
                y.z = fmaxf(fmaf(acc.z + bi.z, al.z, be.z), 0.f);
                y.w = fmaxf(fmaf(acc.w + bi.w, al.w, be.w), 0.f);
                out4[((size_t)n * 16 + cq) * (size_t)(22 * 22) + p] = y;
            }
        }
    }
}

int dcx_launch_conv1_patches_u8(const uint8_t* frames, long frame_stride, int pitch, int pix, int h, int w, const int32_t* table,
                                const int* total, int max_patches, int n_hint, const float* w9x64, const float* bias,
                                const float* alpha, const float* beta, float* out_c4, hipStream_t s) {
    if (!frames || !table || !total || !w9x64 || !bias || !alpha || !beta || !out_c4) return DCX_E_ARG;
    if (max_patches <= 0 || h <= 0 || w <= 0) return DCX_E_SHAPE;
    if (pix != DCX_PIX_GRAY8 && pix != DCX_PIX_BGR8 && pix != DCX_PIX_BGR8_LEGACY14) return DCX_E_ARG;
    // The grid covers the EXPECTED number of live patches (n_hint; the kernel strides over the rest): workgroups that only find
    // out that their slot is dead still cost a dispatch each, and at ~3 ns per workgroup 8,192 of them were the whole 24 us
    int gy = n_hint > 0 && n_hint < max_patches ? n_hint : max_patches;
    if (gy > 65535) gy = 65535;
    dim3 grid(4, (unsigned)gy);
#define DCX_P1(PX) hipLaunchKernelGGL((dcx_conv1_patches_kernel<PX>), grid, dim3(256), 0, s, frames, frame_stride, pitch, h, w, table, \
                                     total, max_patches, w9x64, bias, alpha, beta, out_c4)
    if (pix == DCX_PIX_GRAY8) DCX_P1(DcxPix<DCX_PIX_GRAY8>);
    else if (pix == DCX_PIX_BGR8) DCX_P1(DcxPix<DCX_PIX_BGR8>);
    else DCX_P1(DcxPix<DCX_PIX_BGR8_LEGACY14>);
#undef DCX_P1
    return (int)hipGetLastError();
}

template <typename PX>
static int launch_conv1(const uint8_t* in, long image_stride, int pitch, int n, int h, int w, int pad,
                        const float* w9x64, const float* bias, const float* alpha, const float* beta,
                        float* out, const int* n_limit, int32_t* zero_words, int n_zero, hipStream_t s) {
    if (!in || !w9x64 || !bias || !alpha || !beta || !out) return DCX_E_ARG;
    const int ho = h + 2 * pad - 2, wo = w + 2 * pad - 2;
    if (ho <= 0 || wo <= 0 || n <= 0) return DCX_E_SHAPE;
    const unsigned gx = (unsigned)((ho * wo + 255) / 256), gy = (unsigned)(n < 65535 ? n : 65535);
    // few pixels in the launch (one or two 320x240 frames; n_limit launches are sized for their capacity and stay unsplit):
    // split the 64 channels over four workgroups so that the chip has enough waves in flight
    // (DCX_CONV1_TILE=0: the round-5 form, every thread decoding its own nine taps -- A/B runs)
    static int tile = -1;
    if (tile < 0) { const char* e = getenv("DCX_CONV1_TILE"); tile = (e && !atoi(e)) ? 0 : 1; }
    if (n_limit == nullptr && (long)gx * gy < 1024 && tile) {
        const int tiles_x = (wo + 15) / 16, tiles_y = (ho + 15) / 16;
        hipLaunchKernelGGL((dcx_conv1_tile_kernel<PX>), dim3((unsigned)tiles_x, (unsigned)(tiles_y * n), 4), dim3(256), 0, s, in, image_stride,
                           pitch, h, w, pad, w9x64, bias, alpha, beta, out, ho, wo, tiles_y, zero_words, n_zero);
    } else if (n_limit == nullptr && (long)gx * gy < 1024)
        hipLaunchKernelGGL((dcx_conv1_kernel<PX, 4>), dim3(gx, gy, 4), dim3(256), 0, s, in, image_stride, pitch, h, w, pad,
                           w9x64, bias, alpha, beta, out, ho, wo, n_limit, n, zero_words, n_zero);
    else
        hipLaunchKernelGGL((dcx_conv1_kernel<PX, 1>), dim3(gx, gy), dim3(256), 0, s, in, image_stride, pitch, h, w, pad,
                           w9x64, bias, alpha, beta, out, ho, wo, n_limit, n, zero_words, n_zero);
    return (int)hipGetLastError();
}

int dcx_launch_conv1_u8(const uint8_t* frames, long frame_stride, int pitch, int pix, int n, int h, int w, int pad,
                        const float* w9x64, const float* bias, const float* alpha, const float* beta,
                        float* out_c4, const int* n_limit, int32_t* zero_words, int n_zero, hipStream_t s) {
    switch (pix) {
        case DCX_PIX_GRAY8:
            return launch_conv1<DcxPix<DCX_PIX_GRAY8>>(frames, frame_stride, pitch, n, h, w, pad, w9x64, bias, alpha, beta, out_c4, n_limit, zero_words, n_zero, s);
        case DCX_PIX_BGR8:
            return launch_conv1<DcxPix<DCX_PIX_BGR8>>(frames, frame_stride, pitch, n, h, w, pad, w9x64, bias, alpha, beta, out_c4, n_limit, zero_words, n_zero, s);
        case DCX_PIX_BGR8_LEGACY14:
            return launch_conv1<DcxPix<DCX_PIX_BGR8_LEGACY14>>(frames, frame_stride, pitch, n, h, w, pad, w9x64, bias, alpha, beta, out_c4, n_limit, zero_words, n_zero, s);
        default:
            return DCX_E_ARG;
    }
}
int dcx_launch_conv1_f32(const float* images, long image_stride, int pitch, int n, int h, int w, int pad,
                         const float* w9x64, const float* bias, const float* alpha, const float* beta,
                         float* out_c4, const int* n_limit, hipStream_t s) {
    // element strides -> bytes
    return launch_conv1<DcxPixF32>(reinterpret_cast<const uint8_t*>(images), image_stride * 4, pitch * 4, n, h, w, pad, w9x64, bias, alpha,
                                   beta, out_c4, n_limit, nullptr, 0, s);
}

// ---------------------------------------------------------------------------------------
// pred_argmax (model_utils.py:53-78) + label_to_keypoints (model_utils.py:91-124), batched.
// One workgroup per frame walks the cells in raster order, 256 at a time; firing cells are
// compacted in order with wave ballots + an LDS scan, so the row order equals torch.nonzero's.
__device__ __forceinline__ float dcx_logit(const DcxLogitView& v, int b, int c, int cell) {
    return v.p[(size_t)b * v.sb + (size_t)(c >> 2) * v.sq + (size_t)cell * v.sp + (size_t)(c & 3) * v.sc];
}

// per-cell 65-/17-way arg-max (model_utils.py:53-66: first maximum wins) with the dust-bin substitution
template <bool C4>
__device__ __forceinline__ void dcx_cell_argmax(const DcxLogitView& loc, const DcxLogitView& ids, int n_loc, int n_ids1,
                                                int cells, int dust_bin, int b, int cell, int& la, int& ia) {
    la = 0; ia = 0;
    if (C4) {
        // C4 logits [b][quad][cell][4]: one 16-B load per channel quad, lanes on consecutive cells.
        // Channels are visited in increasing order with a strict '>' so the first maximum wins
        // (torch.argmax); the zero-filled pad channels of the last quad are never looked at.
        const float4* lq = reinterpret_cast<const float4*>(loc.p) + (size_t)b * (loc.sb >> 2) + cell;
        const float4* iq = reinterpret_cast<const float4*>(ids.p) + (size_t)b * (ids.sb >> 2) + cell;
        float best = -INFINITY;
        for (int q = 0; 4 * q < n_loc; ++q) {
            const float4 v = lq[(size_t)q * cells];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * q + k < n_loc && (e[k] > best || (q == 0 && k == 0))) { best = e[k]; la = 4 * q + k; }
        }
        best = -INFINITY;
        for (int q = 0; 4 * q < n_ids1; ++q) {
            const float4 v = iq[(size_t)q * cells];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * q + k < n_ids1 && (e[k] > best || (q == 0 && k == 0))) { best = e[k]; ia = 4 * q + k; }
        }
    } else {
        float best = dcx_logit(loc, b, 0, cell);
        for (int c = 1; c < n_loc; ++c) {            // torch.argmax: first maximum wins
            const float v = dcx_logit(loc, b, c, cell);
            if (v > best) { best = v; la = c; }
        }
        best = dcx_logit(ids, b, 0, cell);
        for (int c = 1; c < n_ids1; ++c) {
            const float v = dcx_logit(ids, b, c, cell);
            if (v > best) { best = v; ia = c; }
        }
    }
    if (la == n_loc - 1) ia = dust_bin;          // where(loc_argmax == 64, dust_bin, ids_argmax)
}

// ordered compaction of one chunk of 256 cells of frame b (wave ballot + LDS scan); all 256 threads call it
__device__ __forceinline__ void dcx_compact_chunk(bool fire, int la, int ia, int cell, int wc, int kmax, int b,
                                                  int* wave_cnt, int* base_s, int32_t* __restrict__ rows) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long m = __ballot(fire);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = *base_s;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (fire) {
        const int pos = off + before;
        if (pos < kmax) {
            const int cy = cell / wc, cx = cell - cy * wc;
            int4 r;
            r.x = 8 * cx + (la & 7);                  // xs = 8*ix + loc % 8
            r.y = 8 * cy + (la >> 3);                 // ys = 8*iy + loc // 8
            r.z = ia;
            r.w = cell;
            reinterpret_cast<int4*>(rows)[(size_t)b * kmax + pos] = r;
        }
    }
    __syncthreads();
    if (tid == 0) *base_s += total;
    __syncthreads();
}

// single-kernel decode: one workgroup per frame (used when the caller has no scratch buffer)
template <bool C4>
__global__ __launch_bounds__(256) void dcx_decode_kernel(DcxLogitView loc, DcxLogitView ids, int n_loc, int n_ids1,
                                                           int hc, int wc, int dust_bin, int kmax,
                                                           int32_t* __restrict__ counts, int32_t* __restrict__ rows,
                                                           int32_t* __restrict__ loc_argmax,
                                                           int32_t* __restrict__ ids_argmax) {
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int cells = hc * wc;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < cells; c0 += 256) {
        const int cell = c0 + tid;
        bool fire = false;
        int la = 0, ia = 0;
        if (cell < cells) {
            dcx_cell_argmax<C4>(loc, ids, n_loc, n_ids1, cells, dust_bin, b, cell, la, ia);
            fire = ia != dust_bin;
            if (loc_argmax) loc_argmax[(size_t)b * cells + cell] = la;
            if (ids_argmax) ids_argmax[(size_t)b * cells + cell] = ia;
        }
        dcx_compact_chunk(fire, la, ia, cell, wc, kmax, b, wave_cnt, &base_s, rows);
    }
    if (tid == 0) counts[b] = base_s;
}

// two-phase decode (pipeline path): the arg-max of every cell of the batch in parallel (HBM-bound, 328 B per cell) ...
template <bool C4>
__global__ __launch_bounds__(256) void dcx_cell_argmax_kernel(DcxLogitView loc, DcxLogitView ids, int n_loc, int n_ids1,
                                                                int cells, int dust_bin, int32_t* __restrict__ codes,
                                                                int32_t* __restrict__ loc_argmax,
                                                                int32_t* __restrict__ ids_argmax) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    int la, ia;
    dcx_cell_argmax<C4>(loc, ids, n_loc, n_ids1, cells, dust_bin, b, cell, la, ia);
    codes[(size_t)b * cells + cell] = la | (ia << 8);
    if (loc_argmax) loc_argmax[(size_t)b * cells + cell] = la;
    if (ids_argmax) ids_argmax[(size_t)b * cells + cell] = ia;
}

// ... then the ordered compaction of each frame's firing cells (one workgroup per frame over 4 bytes per cell)
__global__ __launch_bounds__(256) void dcx_compact_kernel(const int32_t* __restrict__ codes, int hc, int wc, int dust_bin,
                                                            int kmax, int32_t* __restrict__ counts,
                                                            int32_t* __restrict__ rows) {
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int cells = hc * wc;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < cells; c0 += 256) {
        const int cell = c0 + tid;
        bool fire = false;
        int la = 0, ia = 0;
        if (cell < cells) {
            const int code = codes[(size_t)b * cells + cell];
            la = code & 255; ia = code >> 8;
            fire = ia != dust_bin;
        }
        dcx_compact_chunk(fire, la, ia, cell, wc, kmax, b, wave_cnt, &base_s, rows);
    }
    if (tid == 0) counts[b] = base_s;
}

int dcx_launch_decode(DcxLogitView loc, DcxLogitView ids, int batch, int n_loc, int n_ids1, int hc, int wc,
                      int dust_bin, int kmax, int32_t* counts, int32_t* rows,
                      int32_t* loc_argmax, int32_t* ids_argmax, int32_t* codes_scratch, hipStream_t s) {
    if (!loc.p || !ids.p || !counts || !rows) return DCX_E_ARG;
    if (batch <= 0 || hc <= 0 || wc <= 0 || kmax <= 0 || n_loc != 65 || n_ids1 < 2 || n_ids1 > 256) return DCX_E_SHAPE;
    // dust_bin is the reference's free parameter `dust_bin_ids` (model_utils.py:76,111): cells whose (masked) id equals it
    // do not fire.  Any value in [0, 255] has exactly the reference's semantics here (it need not equal n_ids); values
    // outside cannot be an id at all and would not survive the packed (loc | id << 8) code.
    if (dust_bin < 0 || dust_bin > 255) return DCX_E_NIDS;
    const bool c4 = loc.sc == 1 && loc.sp == 4 && ids.sc == 1 && ids.sp == 4 && loc.sq == 4L * hc * wc && ids.sq == 4L * hc * wc
                    && (loc.sb & 3) == 0 && (ids.sb & 3) == 0;
    if (codes_scratch != nullptr) {     // [batch][hc*wc] int32 of scratch: arg-max of all cells in parallel, then compaction
        const int cells = hc * wc;
        const dim3 grid((unsigned)((cells + 255) / 256), (unsigned)batch);
        if (c4)
            hipLaunchKernelGGL(dcx_cell_argmax_kernel<true>, grid, dim3(256), 0, s, loc, ids, n_loc, n_ids1, cells, dust_bin,
                               codes_scratch, loc_argmax, ids_argmax);
        else
            hipLaunchKernelGGL(dcx_cell_argmax_kernel<false>, grid, dim3(256), 0, s, loc, ids, n_loc, n_ids1, cells, dust_bin,
                               codes_scratch, loc_argmax, ids_argmax);
        hipLaunchKernelGGL(dcx_compact_kernel, dim3((unsigned)batch), dim3(256), 0, s, codes_scratch, hc, wc, dust_bin, kmax,
                           counts, rows);
        return (int)hipGetLastError();
    }
    if (c4)
        hipLaunchKernelGGL(dcx_decode_kernel<true>, dim3((unsigned)batch), dim3(256), 0, s, loc, ids, n_loc, n_ids1, hc, wc,
                           dust_bin, kmax, counts, rows, loc_argmax, ids_argmax);
    else
        hipLaunchKernelGGL(dcx_decode_kernel<false>, dim3((unsigned)batch), dim3(256), 0, s, loc, ids, n_loc, n_ids1, hc, wc,
                           dust_bin, kmax, counts, rows, loc_argmax, ids_argmax);
    return (int)hipGetLastError();
}

int dcx_launch_compact(const int32_t* codes, int batch, int hc, int wc, int dust_bin, int kmax, int32_t* counts,
                       int32_t* rows, hipStream_t s) {
    if (!codes || !counts || !rows) return DCX_E_ARG;
    if (batch <= 0 || hc <= 0 || wc <= 0 || kmax <= 0) return DCX_E_SHAPE;
    if (dust_bin < 0 || dust_bin > 256) return DCX_E_NIDS;          // 256: "no label is the dust bin" (label maps only)
    hipLaunchKernelGGL(dcx_compact_kernel, dim3((unsigned)batch), dim3(256), 0, s, codes, hc, wc, dust_bin, kmax, counts, rows);
    return (int)hipGetLastError();
}

extern "C" int dcx_pred_to_keypoints(const float* d_loc, const float* d_ids, int batch, int n_loc, int n_ids1,
                                     int hc, int wc, int dust_bin, int kmax, int32_t* d_counts, int32_t* d_rows,
                                     int32_t* d_loc_argmax, int32_t* d_ids_argmax, void* stream) {
    const long cells = (long)hc * wc;
    DcxLogitView lv{d_loc, (long)n_loc * cells, 4 * cells, 1, cells};
    DcxLogitView iv{d_ids, (long)n_ids1 * cells, 4 * cells, 1, cells};
    return dcx_launch_decode(lv, iv, batch, n_loc, n_ids1, hc, wc, dust_bin, kmax, d_counts, d_rows,
                             d_loc_argmax, d_ids_argmax, nullptr, (hipStream_t)stream);
}

// label_to_keypoints (model_utils.py:91-124) on caller label maps: loc / ids int64 [B][Hc][Wc] (class indices, as pred_argmax
// returns them or as the dataset's labels hold them) -> the same ordered rows as the decode of logits: mask = ids != dust_bin,
// x = 8 ix + loc % 8, y = 8 iy + loc / 8, in torch.nonzero's raster order.  The maps are packed to one code per cell
// (loc | id << 8, d_codes: 4 B per cell of scratch) and compacted by the kernel the pipeline uses.
__global__ __launch_bounds__(256) void dcx_pack_labels_kernel(const long long* __restrict__ loc, const long long* __restrict__ ids,
                                                              long n, int32_t* __restrict__ codes, int* __restrict__ bad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long l = loc[i], d = ids[i];
    if (l < 0 || l > 255 || d < 0 || d > 255) { if (bad) *bad = 1; codes[i] = 0; return; }
    codes[i] = (int32_t)l | ((int32_t)d << 8);
}

extern "C" int dcx_label_to_keypoints(const long long* d_loc, const long long* d_ids, int batch, int hc, int wc, int dust_bin,
                                      int kmax, int32_t* d_counts, int32_t* d_rows, int32_t* d_codes, int32_t* d_bad, void* stream) {
    if (!d_loc || !d_ids || !d_counts || !d_rows || !d_codes) return DCX_E_ARG;
    if (batch <= 0 || hc <= 0 || wc <= 0 || kmax <= 0) return DCX_E_SHAPE;
    // a dust_bin outside [0, 255] equals no (8-bit) label: `ids != dust_bin_ids` (model_utils.py:111) is then true everywhere
    if (dust_bin < 0 || dust_bin > 255) dust_bin = 256;
    const long n = (long)batch * hc * wc;
    hipLaunchKernelGGL(dcx_pack_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_loc, d_ids, n,
                       d_codes, d_bad);
    return dcx_launch_compact(d_codes, batch, hc, wc, dust_bin, kmax, d_counts, d_rows, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// patch table: exclusive scan of min(counts, kmax) over the frames of a batch (one workgroup).
__device__ __forceinline__ void dcx_patch_table_body(const int32_t* __restrict__ counts, const int32_t* __restrict__ rows,
                                                     int batch, int kmax, int32_t* __restrict__ table, int32_t* __restrict__ total,
                                                     int* s_cnt, int* s_start, int* wave_tot, int* carry) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < batch; b0 += 256) {
        const int b = b0 + tid;
        const int c = b < batch ? min(counts[b], kmax) : 0;
        // exclusive scan of the chunk's 256 counts: shuffle scan inside each wave, then the 4 wave totals
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = *carry;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        s_cnt[tid] = c;
        s_start[tid] = off + incl - c;
        __syncthreads();
        // all 256 threads copy the (frame, k) rows of the chunk
        const int nb = min(256, batch - b0);
        for (int idx = tid; idx < nb * kmax; idx += 256) {
            const int bl = idx / kmax, k = idx - bl * kmax;
            if (k < s_cnt[bl]) {
                const int bb = b0 + bl;
                const int4 r = reinterpret_cast<const int4*>(rows)[(size_t)bb * kmax + k];
                reinterpret_cast<int4*>(table)[s_start[bl] + k] = make_int4(bb, r.x, r.y, bb * kmax + k);
            }
        }
        __syncthreads();
        if (tid == 0) *carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
    if (tid == 0) *total = *carry;
}

__global__ __launch_bounds__(256) void dcx_patch_table_kernel(const int32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ rows, int batch, int kmax,
                                                                int32_t* __restrict__ table, int32_t* __restrict__ total) {
    __shared__ int s_cnt[256], s_start[256];
    __shared__ int wave_tot[4];
    __shared__ int carry;
    dcx_patch_table_body(counts, rows, batch, kmax, table, total, s_cnt, s_start, wave_tot, &carry);
}

extern "C" int dcx_build_patch_table(const int32_t* d_counts, const int32_t* d_rows, int batch, int kmax,
                                     int32_t* d_table, int32_t* d_total, void* stream) {
    if (!d_counts || !d_rows || !d_table || !d_total) return DCX_E_ARG;
    if (batch <= 0 || kmax <= 0) return DCX_E_SHAPE;
    hipLaunchKernelGGL(dcx_patch_table_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_counts, d_rows, batch,
                       kmax, d_table, d_total);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// extract_patches model_utils.py:19-36: patch[p][i][j] = img[y-12+i][x-12+j], zero outside.
template <typename TIn>
__global__ __launch_bounds__(192) void dcx_gather_kernel(const TIn* __restrict__ frames, long frame_stride, int pitch,
                                                           int h, int w, const int32_t* __restrict__ table,
                                                           const int32_t* __restrict__ total,
                                                           float* __restrict__ patches) {
    const int p = blockIdx.x;
    if (total != nullptr && p >= *total) return;
    const int4 t = reinterpret_cast<const int4*>(table)[p];
    const TIn* img = frames + (size_t)t.x * frame_stride;
    for (int e = threadIdx.x; e < 576; e += 192) {
        const int i = e / 24, j = e - i * 24;
        const int iy = t.z - 12 + i, ix = t.y - 12 + j;
        const bool inb = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
        const int cy = min(max(iy, 0), h - 1), cx = min(max(ix, 0), w - 1);
        const float v = dcx_load_px(img + (size_t)cy * pitch + cx);
        patches[(size_t)p * 576 + e] = inb ? v : 0.0f;
    }
}

extern "C" int dcx_extract_patches_u8(const uint8_t* d_frames, long frame_stride, int pitch, int height, int width,
                                      const int32_t* d_table, const int32_t* d_total, int max_patches,
                                      float* d_patches, void* stream) {
    if (!d_frames || !d_table || !d_patches) return DCX_E_ARG;
    if (max_patches <= 0 || height <= 0 || width <= 0) return DCX_E_SHAPE;
    hipLaunchKernelGGL((dcx_gather_kernel<uint8_t>), dim3((unsigned)max_patches), dim3(192), 0, (hipStream_t)stream,
                       d_frames, frame_stride, pitch, height, width, d_table, d_total, d_patches);
    return (int)hipGetLastError();
}

extern "C" int dcx_extract_patches_f32(const float* d_images, int height, int width, const int32_t* d_table,
                                       const int32_t* d_total, int max_patches, float* d_patches, void* stream) {
    if (!d_images || !d_table || !d_patches) return DCX_E_ARG;
    if (max_patches <= 0 || height <= 0 || width <= 0) return DCX_E_SHAPE;
    hipLaunchKernelGGL((dcx_gather_kernel<float>), dim3((unsigned)max_patches), dim3(192), 0, (hipStream_t)stream,
                       d_images, (long)height * width, width, height, width, d_table, d_total, d_patches);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// RefineNet.infer_patches refinenet.py:108-114: reduce the per-tile maxima of the heat-map to the
// first flat arg-max, corners = (col,row), corners_og = (corners - 32) / 8 + keypoint.
__global__ __launch_bounds__(64) void dcx_refine_finalize_kernel(const float* __restrict__ part_val,
                                                                   const int* __restrict__ part_idx, int tiles, int wo,
                                                                   int max_patches, const int* __restrict__ total,
                                                                   const int32_t* __restrict__ table,
                                                                   int32_t* __restrict__ corners,
                                                                   float* __restrict__ xy) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= max_patches) return;
    if (total != nullptr && p >= *total) return;
    float best = part_val[(size_t)p * tiles];
    int besti = part_idx[(size_t)p * tiles];
    for (int t = 1; t < tiles; ++t) {
        const float v = part_val[(size_t)p * tiles + t];
        const int i = part_idx[(size_t)p * tiles + t];
        if (v > best || (v == best && i < besti)) { best = v; besti = i; }
    }
    const int row = besti / wo, col = besti - row * wo;
    if (corners) { corners[2 * p] = col; corners[2 * p + 1] = row; }
    if (xy != nullptr && table != nullptr) {
        const int4 t = reinterpret_cast<const int4*>(table)[p];
        xy[2 * (size_t)t.w] = (float)(col - 32) / 8.0f + (float)t.y;
        xy[2 * (size_t)t.w + 1] = (float)(row - 32) / 8.0f + (float)t.z;
    }
}

int dcx_launch_refine_finalize(const float* part_val, const int* part_idx, int tiles, int wo, int max_patches,
                               const int* total, const int32_t* table, int32_t* corners, float* xy, hipStream_t s) {
    hipLaunchKernelGGL(dcx_refine_finalize_kernel, dim3((unsigned)((max_patches + 63) / 64)), dim3(64), 0, s, part_val,
                       part_idx, tiles, wo, max_patches, total, table, corners, xy);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// speedy_bargmax2d model_utils.py:39-43 for arbitrary (K,h,w): first flat maximum -> (col,row).
__global__ __launch_bounds__(256) void dcx_argmax2d_kernel(const float* __restrict__ x, int hw, int w,
                                                             int32_t* __restrict__ out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int k = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = x + (size_t)k * hw;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = tid; i < hw; i += 256) {
        const float v = p[i];
        if (v > best) { best = v; besti = i; }   // i increases per thread: first max kept
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int wv = 1; wv < 4; ++wv)
            if (sv[wv] > best || (sv[wv] == best && si[wv] < besti)) { best = sv[wv]; besti = si[wv]; }
        if (besti == 0x7fffffff) besti = 0;
        out[2 * k] = besti % w;
        out[2 * k + 1] = besti / w;
    }
}

extern "C" int dcx_argmax2d(const float* d_x, int k, int h, int w, int32_t* d_out, void* stream) {
    if (!d_x || !d_out) return DCX_E_ARG;
    if (k <= 0 || h <= 0 || w <= 0) return DCX_E_SHAPE;
    hipLaunchKernelGGL(dcx_argmax2d_kernel, dim3((unsigned)k), dim3(256), 0, (hipStream_t)stream, d_x, h * w, w, d_out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dcx_pre_image_kernel(const uint8_t* __restrict__ g, float* __restrict__ out,
                                                              size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = dcx_norm_u8(g[i]);
}

extern "C" int dcx_pre_image(const uint8_t* d_gray, float* d_out, size_t n, void* stream) {
    if (!d_gray || !d_out) return DCX_E_ARG;
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dcx_pre_image_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_gray, d_out, n);
    return (int)hipGetLastError();
}

// cv2.cvtColor(img, COLOR_BGR2GRAY) on 8-bit images (call site /root/reference/src/inference.py:40).  OpenCV is third-party and
// not vendored; the reference pins opencv-contrib-python >= 4.6, < 4.12 (requirements.txt:5; src/requirements.txt: 4.6.0.66).
// OpenCV 4.x, modules/imgproc/src/color.hpp + color_rgb.simd.hpp, RGB2Gray<uchar>: gray_shift = 15, RY15 / GY15 / BY15 =
// 9798 / 19235 / 3735 (= 0.299 / 0.587 / 0.114 x 32768, rounded; they sum to 32768):
//     gray = (3735 B + 19235 G + 9798 R + 16384) >> 15                                   <- dcx_bgr2gray (CB, CG, CR, SHIFT below)
// The 14-bit constants R2Y / G2Y / B2Y = 4899 / 9617 / 1868 (yuv_shift = 14), which older OpenCV generations also used for 8-bit
// gray, survive in 4.x only for 16-bit images and YUV:  gray = (1868 B + 9617 G + 4899 R + 8192) >> 14  <- dcx_bgr2gray_legacy14.
// The two differ by one gray level on ~0.26 % of colour pixels (never on B = G = R).  Integer arithmetic, so the device result
// equals the host restatement (deepcharuco_amd/imgproc.py, oracle bgr2gray) bit for bit.  One thread = 4 pixels (12 B in, 4 B
// out); the numpy version of this costs the host 150-300 us per 320x240 frame, more than half of a bs=1 call.
template <unsigned CB, unsigned CG, unsigned CR, int SHIFT>
__global__ __launch_bounds__(256) void dcx_bgr2gray_kernel(const uint8_t* __restrict__ bgr, long frame_stride, int pitch,
                                                             int h, int w, uint8_t* __restrict__ gray) {
    static_assert(CB + CG + CR == (1u << SHIFT), "the weights of a gray conversion sum to one");
    const int b = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= w) return;
    const uint8_t* src = bgr + (size_t)b * frame_stride + (size_t)y * pitch + (size_t)x0 * 3;
    uint8_t* dst = gray + ((size_t)b * h + y) * w + x0;
    const int n = min(4, w - x0);
    uint8_t g[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const unsigned bb = src[3 * i], gg = src[3 * i + 1], rr = src[3 * i + 2];
        g[i] = (uint8_t)((bb * CB + gg * CG + rr * CR + (1u << (SHIFT - 1))) >> SHIFT);
    }
    if (n == 4 && ((w & 3) == 0)) *reinterpret_cast<uchar4*>(dst) = make_uchar4(g[0], g[1], g[2], g[3]);
    else for (int i = 0; i < n; ++i) dst[i] = g[i];
}

template <unsigned CB, unsigned CG, unsigned CR, int SHIFT>
static int bgr2gray_launch(const uint8_t* d_bgr, long frame_stride, int pitch, int batch, int height, int width,
                           uint8_t* d_gray, void* stream) {
    if (!d_bgr || !d_gray) return DCX_E_ARG;
    if (batch <= 0 || height <= 0 || width <= 0 || batch > 65535 || height > 65535 || pitch < 3 * width) return DCX_E_SHAPE;
    const dim3 grid((unsigned)((width + 1023) / 1024), (unsigned)height, (unsigned)batch);
    hipLaunchKernelGGL((dcx_bgr2gray_kernel<CB, CG, CR, SHIFT>), grid, dim3(256), 0, (hipStream_t)stream, d_bgr, frame_stride,
                       pitch, height, width, d_gray);
    return (int)hipGetLastError();
}

extern "C" int dcx_bgr2gray(const uint8_t* d_bgr, long frame_stride, int pitch, int batch, int height, int width,
                            uint8_t* d_gray, void* stream) {
    return bgr2gray_launch<3735u, 19235u, 9798u, 15>(d_bgr, frame_stride, pitch, batch, height, width, d_gray, stream);
}

extern "C" int dcx_bgr2gray_legacy14(const uint8_t* d_bgr, long frame_stride, int pitch, int batch, int height, int width,
                                     uint8_t* d_gray, void* stream) {
    return bgr2gray_launch<1868u, 9617u, 4899u, 14>(d_bgr, frame_stride, pitch, batch, height, width, d_gray, stream);
}

// NCHW [n][c][hw] <-> C4 [n][ceil(c/4)][hw][4]
__global__ __launch_bounds__(256) void dcx_nchw_to_c4_kernel(const float* __restrict__ src, int c, int hw,
                                                               float* __restrict__ dst) {
    const int cq_n = (c + 3) >> 2;
    const int n = blockIdx.z, cq = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* s = src + ((size_t)n * c + 4 * cq) * hw + p;
    if (4 * cq + 0 < c) v.x = s[0];
    if (4 * cq + 1 < c) v.y = s[(size_t)hw];
    if (4 * cq + 2 < c) v.z = s[2 * (size_t)hw];
    if (4 * cq + 3 < c) v.w = s[3 * (size_t)hw];
    reinterpret_cast<float4*>(dst)[((size_t)n * cq_n + cq) * hw + p] = v;
}

__global__ __launch_bounds__(256) void dcx_c4_to_nchw_kernel(const float* __restrict__ src, int c, int hw,
                                                               float* __restrict__ dst) {
    const int cq_n = (c + 3) >> 2;
    const int n = blockIdx.z, cq = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const float4 v = reinterpret_cast<const float4*>(src)[((size_t)n * cq_n + cq) * hw + p];
    float* d = dst + ((size_t)n * c + 4 * cq) * hw + p;
    if (4 * cq + 0 < c) d[0] = v.x;
    if (4 * cq + 1 < c) d[(size_t)hw] = v.y;
    if (4 * cq + 2 < c) d[2 * (size_t)hw] = v.z;
    if (4 * cq + 3 < c) d[3 * (size_t)hw] = v.w;
}

extern "C" int dcx_nchw_to_c4(const float* d_nchw, int n, int c, int h, int w, float* d_c4, void* stream) {
    if (!d_nchw || !d_c4) return DCX_E_ARG;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || n > 65535) return DCX_E_SHAPE;
    dim3 grid((unsigned)((h * w + 255) / 256), (unsigned)((c + 3) / 4), (unsigned)n);
    hipLaunchKernelGGL(dcx_nchw_to_c4_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_nchw, c, h * w, d_c4);
    return (int)hipGetLastError();
}

extern "C" int dcx_c4_to_nchw(const float* d_c4, int n, int c, int h, int w, float* d_nchw, void* stream) {
    if (!d_nchw || !d_c4) return DCX_E_ARG;
    if (n <= 0 || c <= 0 || h <= 0 || w <= 0 || n > 65535) return DCX_E_SHAPE;
    dim3 grid((unsigned)((h * w + 255) / 256), (unsigned)((c + 3) / 4), (unsigned)n);
    hipLaunchKernelGGL(dcx_c4_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_c4, c, h * w, d_nchw);
    return (int)hipGetLastError();
}
