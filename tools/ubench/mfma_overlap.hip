// Kernel-tuning microbenchmark (not part of the product): does vector-ALU work of a SECOND wave on the same SIMD overlap
// with the fp32 MFMAs of the first one on gfx950?  One workgroup of 8 waves per CU (waves w and w+4 share SIMD w):
// waves 0..3 run a pure v_mfma_f32 stream (N MFMAs), waves 4..7 run a filler stream of a given kind until the MFMA waves
// are done.  Reported: cycles per MFMA seen by wave 0, and filler instructions retired per MFMA cycle by wave 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MF, int KIND>
__global__ __launch_bounds__(512, 1) void bench(float* out, const float* in, unsigned long long* res, int iters) {
    __shared__ float4 lds[2048];
    __shared__ volatile int done;
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 2048; i += 512) lds[i] = make_float4(i, 1, 2, 3);
    if (tid == 0) done = 0;
    __syncthreads();
    float a = in[tid & 255], b = in[(tid & 255) + 256];
    if (wave < 4) {
        f32x16 acc[4];
        f32x4 acc4[8];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                if (MF == 0) acc[p & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p & 3], 0, 0, 0);
                else acc4[p & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[p & 7], 0, 0, 0);
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
        out[blockIdx.x * 512 + tid] = s;
        if (tid == 0) { done = 1; if (blockIdx.x == 0) res[0] = t1 - t0; }
    } else {
        f32x2 x0 = {a, b}, x1 = {b, a}, x2 = {a, a}, x3 = {b, b};
        const f32x2 one = {1.f, 1.f};
        float y0 = a, y1 = b, y2 = a, y3 = b;
        f32x16 av; for (int r = 0; r < 16; ++r) av[r] = a;
        asm volatile("" : "+a"(av));
        float4 l = make_float4(0, 0, 0, 0);
        unsigned long long n = 0;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (KIND != 0) {
            while (!done) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (KIND == 1) {   // 4 independent v_pk_add_f32 chains
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x0) : "v"(one));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x1) : "v"(one));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x2) : "v"(one));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x3) : "v"(one));
                    }
                    if (KIND == 2) {   // 4 independent v_add_f32
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(y0) : "v"(a));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(y1) : "v"(a));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(y2) : "v"(a));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(y3) : "v"(a));
                    }
                    if (KIND == 3) {   // 4 ds_read_b128
                        for (int j = 0; j < 4; ++j) { l = lds[(tid + 64 * j + k) & 2047]; asm volatile("" :: "v"(l.x)); }
                    }
                    if (KIND == 4) {   // 4 v_accvgpr_read
                        float t;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(av[0]));  y0 += 0 * t;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(av[1]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(av[2]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(av[3]));
                    }
                    if (KIND == 5) {   // 4 v_mov
                        asm volatile("v_mov_b32 %0, %1" : "=v"(y0) : "v"(a));
                        asm volatile("v_mov_b32 %0, %1" : "=v"(y1) : "v"(a));
                        asm volatile("v_mov_b32 %0, %1" : "=v"(y2) : "v"(a));
                        asm volatile("v_mov_b32 %0, %1" : "=v"(y3) : "v"(a));
                    }
                    if (KIND == 6) {   // 4 v_pk_fma_f32
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(one));
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(one));
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(one));
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(one));
                    }
                }
                n += 64;
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 512 + tid] = x0.x + x1.x + x2.x + x3.x + y0 + y1 + y2 + y3 + l.x;
        if (tid == 256 && blockIdx.x == 0) { res[1] = n; res[2] = t1 - t0; }
    }
}

template <int MF, int KIND>
void run(const char* name, float* d_out, float* d_in, unsigned long long* d_c) {
    const int iters = 4000;
    (void)hipMemset(d_c, 0, 24);
    hipLaunchKernelGGL((bench<MF, KIND>), dim3(256), dim3(512), 0, 0, d_out, d_in, d_c, iters);
    hipLaunchKernelGGL((bench<MF, KIND>), dim3(256), dim3(512), 0, 0, d_out, d_in, d_c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c[3];
    (void)hipMemcpy(c, d_c, 24, hipMemcpyDeviceToHost);
    printf("%-12s filler %-18s: %7.2f cycles per MFMA (wave 0);  filler wave: %8.3f instr per 64 cycles, %6.2f cycles per instr\n",
           MF == 0 ? "32x32x2f32" : "16x16x4f32", name, (double)c[0] / (iters * 16.0),
           c[2] ? (double)c[1] / (double)c[2] * 64.0 : 0.0, c[1] ? (double)c[2] / (double)c[1] : 0.0);
}

int main() {
    float *d_out, *d_in; unsigned long long* d_c;
    (void)hipMalloc(&d_out, 256 * 512 * 4); (void)hipMalloc(&d_in, 65536 * 4); (void)hipMalloc(&d_c, 24);
    (void)hipMemset(d_in, 0, 65536 * 4);
    run<0, 0>("none (idle)", d_out, d_in, d_c);
    run<0, 1>("v_pk_add_f32", d_out, d_in, d_c);
    run<0, 2>("v_add_f32", d_out, d_in, d_c);
    run<0, 3>("ds_read_b128", d_out, d_in, d_c);
    run<0, 4>("v_accvgpr_read", d_out, d_in, d_c);
    run<0, 5>("v_mov_b32", d_out, d_in, d_c);
    run<0, 6>("v_pk_fma_f32", d_out, d_in, d_c);
    run<1, 0>("none (idle)", d_out, d_in, d_c);
    run<1, 1>("v_pk_add_f32", d_out, d_in, d_c);
    run<1, 2>("v_add_f32", d_out, d_in, d_c);
    run<1, 4>("v_accvgpr_read", d_out, d_in, d_c);
    return 0;
}
