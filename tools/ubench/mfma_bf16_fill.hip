// Kernel-tuning microbenchmark (not part of the product; round 4): does ordinary VALU work overlap with a BF16 MFMA
// stream on gfx950?  For the fp32 MFMAs it does not (mfma_fill.hip: every VALU instruction between two v_mfma_f32_32x32x2_f32 costs
// its full ~8 cycles) -- the question that decides whether a bf16x3 Winograd kernel can hide the three-way operand split.
// One wave per SIMD; cycles per MFMA when K filler instructions of a kind follow every MFMA pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MF, int KIND, int K>
__global__ __launch_bounds__(256, 1) void bench(float* out, const float* in, unsigned long long* cyc, int iters) {
    const int tid = threadIdx.x;
    f32x16 acc[4];
    f32x4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[tid + i]; b[i] = (__bf16)in[tid + 8 + i]; }
    int y = tid * 3;
    f32x2 pk = {in[tid], in[tid + 1]}, pk2 = {1.f, 2.f};
    float cv0 = in[tid + 2], cv1 = in[tid + 3];
    unsigned cvo = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            __builtin_amdgcn_sched_barrier(0);
            if (MF == 0) {
                acc[(2 * p) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[(2 * p) & 3], 0, 0, 0);
                acc[(2 * p + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[(2 * p + 1) & 3], 0, 0, 0);
            } else {
                acc4[(2 * p) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[(2 * p) & 7], 0, 0, 0);
                acc4[(2 * p + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc4[(2 * p + 1) & 7], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 1) { asm volatile("v_add_u32 %0, %0, 1" : "+v"(y)); }                                   // simple VALU
                if (KIND == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk) : "v"(pk2)); }                    // packed fp32 add (the transform's op)
                if (KIND == 3) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(cvo) : "v"(cv0), "v"(cv1)); }    // the split's conversion
                if (KIND == 5) { asm volatile("s_nop 0"); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    out[blockIdx.x * 256 + tid] = s + y + pk.x + pk.y + (float)cvo;
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MF, int KIND, int K>
void run(const char* name, float* d_out, float* d_in, unsigned long long* d_c) {
    const int iters = 2000;
    hipLaunchKernelGGL((bench<MF, KIND, K>), dim3(256), dim3(256), 0, 0, d_out, d_in, d_c, iters);
    hipLaunchKernelGGL((bench<MF, KIND, K>), dim3(256), dim3(256), 0, 0, d_out, d_in, d_c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long c;
    (void)hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
    printf("%-24s %-20s K=%2d per MFMA pair: %7.2f cycles per MFMA\n", MF == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", name, K,
           (double)c / (iters * 16.0));
}

template <int MF>
void sweep(float* d_out, float* d_in, unsigned long long* d_c) {
    run<MF, 5, 0>("none", d_out, d_in, d_c);
    run<MF, 1, 1>("v_add_u32", d_out, d_in, d_c); run<MF, 1, 2>("v_add_u32", d_out, d_in, d_c); run<MF, 1, 4>("v_add_u32", d_out, d_in, d_c);
    run<MF, 1, 8>("v_add_u32", d_out, d_in, d_c); run<MF, 1, 16>("v_add_u32", d_out, d_in, d_c);
    run<MF, 2, 2>("v_pk_add_f32", d_out, d_in, d_c); run<MF, 2, 4>("v_pk_add_f32", d_out, d_in, d_c); run<MF, 2, 8>("v_pk_add_f32", d_out, d_in, d_c);
    run<MF, 3, 2>("v_cvt_pk_bf16_f32", d_out, d_in, d_c); run<MF, 3, 4>("v_cvt_pk_bf16_f32", d_out, d_in, d_c); run<MF, 3, 8>("v_cvt_pk_bf16_f32", d_out, d_in, d_c);
    run<MF, 5, 4>("s_nop", d_out, d_in, d_c);
}

int main() {
    float *d_out, *d_in; unsigned long long* d_c;
    (void)hipMalloc(&d_out, 256 * 256 * 4); (void)hipMalloc(&d_in, 65536 * 4); (void)hipMalloc(&d_c, 8);
    (void)hipMemset(d_in, 0, 65536 * 4);
    sweep<0>(d_out, d_in, d_c);
    sweep<1>(d_out, d_in, d_c);
    return 0;
}
