// issue_mix.hip -- does scalar work share issue bandwidth with vector work on gfx950?  Each wavefront runs ITER rounds of
// NV independent VALU instructions (a mix of cheap and SGPR-sourced ones, as in the DP kernels) interleaved with NS SALU
// instructions; 6 wavefronts per SIMD.  Prints cycles per round per SIMD (at 2.4 GHz nominal) for several (NV, NS).
// Bring-up tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 4096

template <int NV, int NS, int KIND>
__global__ void __launch_bounds__(256) k(float *out, int seed, float sgf) {
    float v[8], w[8];
    int s[8];
    for (int c = 0; c < 8; ++c) v[c] = 1.0f + 0.001f * (threadIdx.x + c + seed), w[c] = 0.5f + c, s[c] = seed + c;
    float sg = sgf;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < (NV > NS ? NV : NS); ++j) {
            if (j < NV) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(w[j & 7]));
                if (KIND == 1) asm volatile("v_fma_f32 %0, %2, %1, %0" : "+v"(v[j & 7]) : "v"(w[j & 7]), "s"(sg));
            }
            if (j < NS) asm volatile("s_add_i32 %0, %0, 3" : "+s"(s[j & 7]) : : "scc");
        }
    }
    float t = 0.f;
    int u = 0;
    for (int c = 0; c < 8; ++c) t += v[c], u += s[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t + u;
}

template <int NV, int NS, int KIND>
void run() {
    float *d;
    const int blocks = 256 * 6, threads = 256;  // 6 waves per SIMD
    (void)hipMalloc(&d, sizeof(float) * blocks * threads);
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<NV, NS, KIND>), dim3(blocks), dim3(threads), 0, 0, d, 1, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<NV, NS, KIND>), dim3(blocks), dim3(threads), 0, 0, d, 1, 0.5f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("launch failed: %s\n", hipGetErrorString(err));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    const double cyc = ms * 1e-3 * 2.4e9 / (6.0 * ITER);  // per round per wave-slot: 6 waves share the SIMD
    printf("kind %d  NV %2d NS %2d : %7.2f cycles per round per SIMD-wave  (%.2f per instruction of any kind)\n", KIND, NV, NS, cyc,
           cyc / (NV + NS));
    (void)hipFree(d);
}

int main() {
    run<16, 0, 0>(); run<16, 4, 0>(); run<16, 8, 0>(); run<16, 12, 0>(); run<16, 16, 0>(); run<0, 16, 0>();
    run<16, 0, 1>(); run<16, 4, 1>(); run<16, 8, 1>(); run<16, 16, 1>();
    return 0;
}
