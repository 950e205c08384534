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

typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY2(ASM)                                                                     \
    for (int i = 0; i < ITER; ++i) {                                                   \
        _Pragma("unroll") for (int c = 0; c < CH; ++c) { asm volatile(ASM : "+v"(v2[c]) : "v"(w2[c]), "s"(sg2)); } \
    }

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int seed, float sgf) {
    float v[CH], w[CH];
    f2 v2[CH], w2[CH];
    float sg = sgf;
    f2 sg2 = {sgf, sgf + 1.0f};
    for (int c = 0; c < CH; ++c) v[c] = 1.0f + 0.001f * (threadIdx.x + c + seed), w[c] = 0.5f + c;
    for (int c = 0; c < CH; ++c) v2[c] = f2{v[c], v[c] + 1.f}, w2[c] = f2{w[c], w[c] + 2.f};
    if (OP == 20) BODY2("v_pk_fma_f32 %0, %0, %1, %1")
    if (OP == 21) BODY2("v_pk_mul_f32 %0, %0, %1")
    if (OP == 22) BODY2("v_pk_fma_f32 %0, %2, %1, %0")
    if (OP == 23) BODY2("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[1,0,1]")
    if (OP == 24) BODY2("v_pk_mul_f32 %0, %2, %0")
    if (OP == 25) BODY2("v_pk_add_f32 %0, %0, %1")
    if (OP == 26) BODY2("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]")
    if (OP == 30) BODY("v_lshl_add_u32 %0, %0, 23, %1")
    if (OP == 31) BODY("v_max3_i32 %0, %0, %1, %1")
    if (OP == 32) BODY("v_add3_u32 %0, %0, %1, %1")
    if (OP == 33) BODY("v_and_b32 %0, %0, %1")
    if (OP == 34) BODY("v_max_i32 %0, %0, %1")
    if (OP == 35) BODY("v_cmp_gt_i32 vcc, %0, %1")
    if (OP == 36) BODY("v_med3_i32 %0, %0, %1, %1")
    if (OP == 37) BODY("v_lshrrev_b32 %0, 23, %0")
    if (OP == 38) BODY("v_mad_u32_u24 %0, %0, %1, %1")
    if (OP == 39) BODY("v_fmac_f32 %0, %1, %1")
    if (OP == 40) BODY("v_add_f32 %0, %0, %1")
    if (OP == 41) BODY("v_perm_b32 %0, %0, %1, %1")
    if (OP == 42) BODY("v_max3_f32 %0, %2, %1, %0")
    if (OP == 43) BODY("v_sub_u32 %0, %2, %0")
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
    for (int c = 0; c < CH; ++c) s += v[c] + v2[c].x + v2[c].y;
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
    run<20>("v_pk_fma_f32 v,v,v,v");
    run<22>("v_pk_fma_f32 v,s,v,v");
    run<23>("v_pk_fma_f32 op_sel_hi broadcast");
    run<21>("v_pk_mul_f32");
    run<24>("v_pk_mul_f32 v,s,v");
    run<25>("v_pk_add_f32");
    run<26>("v_pk_mov_b32");
    run<30>("v_lshl_add_u32");
    run<31>("v_max3_i32");
    run<32>("v_add3_u32");
    run<33>("v_and_b32");
    run<34>("v_max_i32");
    run<35>("v_cmp_gt_i32");
    run<36>("v_med3_i32");
    run<37>("v_lshrrev_b32");
    run<38>("v_mad_u32_u24");
    run<39>("v_fmac_f32");
    run<40>("v_add_f32");
    run<41>("v_perm_b32");
    run<42>("v_max3_f32 v,s,v,v");
    run<43>("v_sub_u32 v,s,v");
    run<9>("v_exp_f32");
    run<10>("v_log_f32");
    return 0;
}
