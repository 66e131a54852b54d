// Kernel-tuning / measurement aid (not part of the product): calibrates rocprofv3's FETCH_SIZE on gfx950 for the access
// patterns of the convolution kernels' input staging, on KNOWN byte counts.
//   rocprofv3 --pmc FETCH_SIZE -d out -o calib -- ./fetch_calib
// The tensor is C4 [CQ][H][W] float4 = 1.26 GB, far beyond the 256 MB Infinity Cache, every kernel touches each byte of
// its region once per "unique" count below:
//   calib_stream        : coalesced 16 B/lane stream over the whole tensor               (unique = fetched = 1.26 GB)
//   calib_tiles<18>     : the staging pattern of dcx_conv_wino2 (256 threads fetch a [4 cq][18][18] float4 halo tile with
//                         buffer_load_dwordx4, 288-B row segments at a 40,960-B pitch), tile origins 18 apart: DISJOINT
//                         tiles, so fetched = unique = known
//   calib_tiles<16>     : the same with tile origins 16 apart (the real kernel: 2-pixel halos shared by neighbours),
//                         requested = 1.27 x unique; what reaches HBM depends on the L2s
//   calib_tiles<32,15>  : disjoint tiles whose rows start 16 B before a 256-B boundary, exactly like the real kernel's
//                         (x0 = 16 tx - 1 pixels): the 288-B segments then straddle the same cache lines
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CQ = 16, H = 1920, W = 2560;

__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ t, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = t[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;
}

template <int STEP, int XOFF = 0>
__global__ __launch_bounds__(256) void calib_tiles(const float* __restrict__ t, int tiles_y, int tiles_x, float* out) {
    const int tid = threadIdx.x;
    const int items = (CQ / 4) * tiles_y * tiles_x;
    float s = 0.f;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {       // persistent walk like the conv kernels
        const int tx = item % tiles_x, ty = (item / tiles_x) % tiles_y, cg = item / (tiles_x * tiles_y);
        const float* base = t + (((size_t)cg * 4 * H + (size_t)ty * STEP) * W + (size_t)tx * STEP + XOFF) * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), (short)0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int idx = tid + k * 256;
            if (idx < 4 * 18 * 18) {
                const int cq = idx / 324, hp = idx - cq * 324, hy = hp / 18, hx = hp - hy * 18;
                const unsigned off = (unsigned)(((size_t)cq * H + hy) * W + hx) * 16u;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                s += __uint_as_float(v.x) + __uint_as_float(v.w);
            }
        }
    }
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const size_t n4 = (size_t)CQ * H * W;
    float4* d; float* o;
    if (hipMalloc(&d, n4 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&o, 16);
    (void)hipMemset(d, 0, n4 * 16);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_stream, dim3(2048), dim3(256), 0, 0, d, n4, o);
        const int ty18 = (H - 18) / 18 + 1, tx18 = (W - 18) / 18 + 1;
        hipLaunchKernelGGL(calib_tiles<18>, dim3(512), dim3(256), 0, 0, (const float*)d, ty18, tx18, o);
        const int ty16 = (H - 18) / 16 + 1, tx16 = (W - 18) / 16 + 1;
        hipLaunchKernelGGL(calib_tiles<16>, dim3(512), dim3(256), 0, 0, (const float*)d, ty16, tx16, o);
        const int ty32 = (H - 18) / 32 + 1, tx32 = (W - 15 - 18) / 32 + 1;
        hipLaunchKernelGGL((calib_tiles<32, 15>), dim3(512), dim3(256), 0, 0, (const float*)d, ty32, tx32, o);
        (void)hipDeviceSynchronize();
        if (rep == 0) {
            printf("known bytes: tiles<32,15> fetched = unique %zu\n", (size_t)(CQ / 4) * ty32 * tx32 * 4 * 324 * 16);
            printf("known bytes: stream %zu\n", n4 * 16);
            printf("known bytes: tiles<18> fetched = unique %zu\n", (size_t)(CQ / 4) * ty18 * tx18 * 4 * 324 * 16);
            printf("known bytes: tiles<16> requested %zu unique %zu\n", (size_t)(CQ / 4) * ty16 * tx16 * 4 * 324 * 16,
                   (size_t)CQ * ((size_t)(ty16 - 1) * 16 + 18) * ((size_t)(tx16 - 1) * 16 + 18) * 16);
        }
    }
    return 0;
}
