// valu_rates.hip -- per-instruction issue cost on gfx950 for the ops the DP kernels use.
// Each test issues ITER x 8 independent copies of ONE instruction (inline asm, so nothing is folded) from 8
// waves per SIMD and reports cycles per wave-instruction per SIMD at the measured clock-agnostic 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 2048
#define CH 8

#define BODY(ASM)                                                                      \
    for (int i = 0; i < ITER; ++i) {                                                   \
        _Pragma("unroll") for (int c = 0; c < CH; ++c) { asm volatile(ASM : "+v"(v[c]) : "v"(w[c]), "s"(sg)); } \
    }

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int seed, float sgf) {
    float v[CH], w[CH];
    float sg = sgf;
    for (int c = 0; c < CH; ++c) v[c] = 1.0f + 0.001f * (threadIdx.x + c + seed), w[c] = 0.5f + c;
    if (OP == 0) BODY("v_fma_f32 %0, %0, %1, %1")
    if (OP == 1) BODY("v_mul_f32 %0, %0, %1")
    if (OP == 2) BODY("v_ldexp_f32 %0, %0, %1")
    if (OP == 3) BODY("v_frexp_exp_i32_f32 %0, %1")
    if (OP == 4) BODY("v_max3_f32 %0, %0, %1, %1")
    if (OP == 5) BODY("v_max_f32 %0, %0, %1")
    if (OP == 6) BODY("v_cndmask_b32 %0, %0, %1, vcc")
    if (OP == 7) BODY("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf")
    if (OP == 8) BODY("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
    if (OP == 9) BODY("v_exp_f32 %0, %1")
    if (OP == 10) BODY("v_log_f32 %0, %1")
    if (OP == 11) BODY("v_sub_u32 %0, %0, %1")
    if (OP == 12) BODY("v_fma_f32 %0, %2, %1, %0")
    if (OP == 13) BODY("v_mov_b32 %0, %1")
    if (OP == 14) BODY("v_mul_f32 %0, %2, %0")
    if (OP == 15) BODY("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char *name) {
    float *d;
    const int blocks = 256 * 8, threads = 256;  // 8 waves per SIMD resident
    (void)hipMalloc(&d, sizeof(float) * blocks * threads);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 2, 0.5f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * (threads / 64) * ITER * CH;
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / (256 * 4);
    printf("%-44s %8.3f ms  %6.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, 2.4e9 / per_simd_per_s);
    (void)hipFree(d);
}

int main() {
    run<0>("v_fma_f32 v,v,v,v");
    run<12>("v_fma_f32 v,s,v,v");
    run<1>("v_mul_f32");
    run<14>("v_mul_f32 v,s,v");
    run<11>("v_sub_u32");
    run<13>("v_mov_b32");
    run<2>("v_ldexp_f32");
    run<3>("v_frexp_exp_i32_f32");
    run<4>("v_max3_f32");
    run<5>("v_max_f32");
    run<6>("v_cndmask_b32");
    run<8>("v_mov_b32_dpp row_shr:1");
    run<7>("v_mov_b32_dpp wave_shl:1");
    run<15>("v_mov_b32_dpp wave_shr:1");
    run<9>("v_exp_f32");
    run<10>("v_log_f32");
    return 0;
}
