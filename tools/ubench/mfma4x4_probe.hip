// Kernel-tuning probe (not part of the product): semantics and cost of v_mfma_f32_4x4x1_16b_f32 used as a per-lane
// linear transform out of the accumulator registers:  D_i(lane l) = sum_p A_p(lane 4*(l/4)+i) * B_p(lane l),
// with B read straight from AGPRs (no v_accvgpr_read).  Checks the lane mapping, back-to-back accumulate chains
// (1 chain vs 4 interleaved) and the cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64, 1) void probe(float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 100.f * r + l;      // m_p(lane) = 100 p + lane
    asm volatile("" : "+a"(acc));
    float coef[16];
    for (int p = 0; p < 16; ++p) coef[p] = ((l & 3) + 1) * ((p & 1) ? -1.f : 1.f) * (p % 3 == 2 ? 0.f : 1.f);   // T[i = l%4][p]
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // VALU-written operands (coef, v_accvgpr_write) -> MFMA
    // --- (a) one dependent chain with explicit wait states, (b) back to back, (c) one of 4 interleaved chains
    f32x4 d, d2, d3[4];
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0\n\ts_nop 7" : "=v"(d) : "v"(coef[0]), "a"(acc[0]));
#pragma unroll
    for (int p = 1; p < 16; ++p) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n\ts_nop 7" : "+v"(d) : "v"(coef[p]), "a"(acc[p]));
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(d2) : "v"(coef[0]), "a"(acc[0]));
#pragma unroll
    for (int p = 1; p < 16; ++p) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(d2) : "v"(coef[p]), "a"(acc[p]));
#pragma unroll
    for (int c = 0; c < 4; ++c) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(d3[c]) : "v"(coef[0]), "a"(acc[0]));
#pragma unroll
    for (int p = 1; p < 16; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(d3[c]) : "v"(coef[p]), "a"(acc[p]));
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    for (int i = 0; i < 4; ++i) { out[l * 4 + i] = d[i]; out[512 + l * 4 + i] = d2[i]; out[768 + l * 4 + i] = d3[3][i]; }
    // --- timing: 4 interleaved chains of 16, repeated
    f32x4 e[4];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 4; ++i) e[c][i] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(e[0]) : "v"(coef[p]), "a"(acc[p]));
            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(e[1]) : "v"(coef[p]), "a"(acc[(p + 1) & 15]));
            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(e[2]) : "v"(coef[p]), "a"(acc[(p + 2) & 15]));
            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(e[3]) : "v"(coef[p]), "a"(acc[(p + 3) & 15]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    f32x4 f;
    for (int i = 0; i < 4; ++i) f[i] = 0.f;
    unsigned long long t2 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; ++p) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(f) : "v"(coef[p]), "a"(acc[p]));
    }
    unsigned long long t3 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    out[256 + l] = e[0][0] + e[1][1] + e[2][2] + e[3][3] + f[0];
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
}

int main() {
    float* d_out; unsigned long long* d_c;
    (void)hipMalloc(&d_out, 1024 * 4); (void)hipMalloc(&d_c, 16);
    const int iters = 1000;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_c, iters);
    (void)hipDeviceSynchronize();
    float h[1024]; unsigned long long c[2];
    (void)hipMemcpy(h, d_out, 4096, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c, d_c, 16, hipMemcpyDeviceToHost);
    int bad_total = 0;
    const char* names[3] = {"chain with s_nop 7 between", "back-to-back dependent chain", "4 interleaved chains"};
    const int base[3] = {0, 512, 768};
    for (int v = 0; v < 3; ++v) {
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                float ref = 0.f;     // sequential fmaf chain in p order, coefficient of lane 4*(l/4)+i, data of lane l
                for (int p = 0; p < 16; ++p) {
                    const float cf = (i + 1) * ((p & 1) ? -1.f : 1.f) * (p % 3 == 2 ? 0.f : 1.f);
                    ref = fmaf(cf, 100.f * p + l, ref);
                }
                if (h[base[v] + l * 4 + i] != ref) { if (bad < 3) printf("  lane %d out %d: got %g expected %g\n", l, i, h[base[v] + l * 4 + i], ref); ++bad; }
            }
        printf("%-32s: %s (%d mismatches)\n", names[v], bad ? "WRONG" : "ok", bad);
        bad_total += bad;
    }
    int bad = bad_total;
    printf("4 interleaved chains: %.2f cycles per v_mfma_f32_4x4x1;  single dependent chain: %.2f\n",
           (double)c[0] / (iters * 64.0), (double)c[1] / (iters * 16.0));
    return bad != 0;
}
