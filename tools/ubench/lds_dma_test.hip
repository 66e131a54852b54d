// Semantics probe for `buffer_load_dwordx4 ... offen lds` on gfx950 (LDS-DMA): where do the 16 bytes of lane l land, and
// what do out-of-range lanes write?   build: hipcc -O3 --offload-arch=gfx950 lds_dma_test.hip -o lds_dma_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, float* out, int n) {
    extern __shared__ float4 s[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 512; i += 256) s[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), (short)0, n * 4, 0x00020000);
    unsigned lds_base = (unsigned)(size_t)(s) + (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6) * 1024u;
    unsigned keep;
    // lane l reads float4 #(255 - tid) (reversed), lanes with tid % 7 == 3 are out of range
    unsigned voff = (tid % 7 == 3) ? 0x80000000u : (unsigned)(255 - tid) * 16u;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_base) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    reinterpret_cast<float4*>(out)[tid] = s[tid];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n), o(n, -2.f);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *e;
    hipMalloc(&d, n * 4); hipMalloc(&e, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, e, n);
    hipMemcpy(o.data(), e, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        const float exp0 = (t % 7 == 3) ? 0.f : (float)((255 - t) * 4);
        for (int c = 0; c < 4; ++c) {
            const float exp = (t % 7 == 3) ? 0.f : exp0 + c;
            if (o[t * 4 + c] != exp) { if (bad < 8) printf("lane %d comp %d: got %g expected %g\n", t, c, o[t * 4 + c], exp); ++bad; }
        }
    }
    printf("mismatches: %d (0 = lane l's 16 B land at M0 + l*16 within its wave, out-of-range lanes write zeros)\n", bad);
    return 0;
}
