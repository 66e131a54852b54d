// Kernel-tuning microbenchmark (not part of the product): cycles per v_mfma_f32_32x32x2_f32 for one wave
// per SIMD when K filler instructions of a given kind follow every MFMA pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void bench(float* out, const float* in, unsigned long long* cyc, int iters) {
    __shared__ float4 lds[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = in[tid], b = in[tid + 256];
    int x = tid, y = tid * 3;
    int sc = iters;
    float4 l = make_float4(0, 0, 0, 0);
    float4 g = make_float4(0, 0, 0, 0);
    const float4* gp = reinterpret_cast<const float4*>(in);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            __builtin_amdgcn_sched_barrier(0);
            acc[(2 * p) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(2 * p) & 3], 0, 0, 0);
            acc[(2 * p + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[(2 * p + 1) & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) { x = x * 3 + y; asm volatile("" : "+v"(x)); }                 // dependent VALU chain (v_mad)
                if (KIND == 1) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(y)); }               // simple VALU
                if (KIND == 2) { l = lds[(tid + k + p * 7 + it) & 1023]; asm volatile("" :: "v"(l.x)); }   // ds_read_b128 (waited)
                if (KIND == 3) { g = gp[(tid + 64 * k + p * 256) & 4095]; asm volatile("" :: "v"(g.x)); } // global_load (waited)
                if (KIND == 4) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc)); }           // SALU
                if (KIND == 5) { asm volatile("s_nop 0"); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s + x + y + l.x + g.x + sc;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int K>
void run(const char* name, float* d_out, float* d_in, unsigned long long* d_c) {
    const int iters = 2000;
    hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 0, 0, d_out, d_in, d_c, iters);
    hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 0, 0, d_out, d_in, d_c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c;
    (void)hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
    printf("%-22s K=%d per pair: %7.2f cycles per MFMA\n", name, K, (double)c / (iters * 16.0));
}

int main() {
    float *d_out, *d_in; unsigned long long* d_c;
    (void)hipMalloc(&d_out, 256 * 256 * 4); (void)hipMalloc(&d_in, 65536 * 4); (void)hipMalloc(&d_c, 8);
    (void)hipMemset(d_in, 0, 65536 * 4);
    run<5, 0>("none", d_out, d_in, d_c);
    run<0, 1>("valu dep chain", d_out, d_in, d_c); run<0, 2>("valu dep chain", d_out, d_in, d_c); run<0, 4>("valu dep chain", d_out, d_in, d_c); run<0, 8>("valu dep chain", d_out, d_in, d_c);
    run<1, 1>("valu simple", d_out, d_in, d_c); run<1, 2>("valu simple", d_out, d_in, d_c); run<1, 4>("valu simple", d_out, d_in, d_c); run<1, 8>("valu simple", d_out, d_in, d_c); run<1, 12>("valu simple", d_out, d_in, d_c);
    run<2, 1>("ds_read_b128+wait", d_out, d_in, d_c); run<2, 2>("ds_read_b128+wait", d_out, d_in, d_c);
    run<3, 1>("global_load+wait", d_out, d_in, d_c); run<3, 2>("global_load+wait", d_out, d_in, d_c);
    run<4, 1>("salu", d_out, d_in, d_c); run<4, 4>("salu", d_out, d_in, d_c); run<4, 8>("salu", d_out, d_in, d_c);
    run<5, 1>("s_nop", d_out, d_in, d_c); run<5, 4>("s_nop", d_out, d_in, d_c);
    return 0;
}
