// pk_step.hip -- how much of a forward step packs into v_pk_mul_f32 / v_pk_fma_f32 when the two slots of a lane are
// carried as one register pair (ext_vector_type(2) float) instead of two scalars.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only tools/pk_step.hip
// VALU instructions per step in the loop: 90.5 (30 packed) against 116.5 for the same arithmetic on scalars (replace
// the typedef by a struct of two floats) -- about -18 % issue cycles at the measured 4.8 / 4.07 cycles per packed /
// plain instruction -- on paper.  Built into k_dp_stair<2> (state carried as pairs, all parity tests green) the launch
// took 151 ms instead of 133 ms: see DESIGN.md 11.
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
struct Tr { float mm, sxm, sym, lxm, lym, msx, sxsx, sysx, msy, sysy, sxsy, mlx, lxlx, mly, lyly; };
__device__ __forceinline__ float dppf(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false)); }
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float scale2(int k) { const int t = k + 127; return __builtin_bit_cast(float, (t > 0 ? t : 0) << 23); }
__device__ __forceinline__ f2 sp(float t) { return f2{t, t}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
struct D2 { f2 m, sx, sy, lx, ly; int e0, e1; };
// pair-native step: io = d-2 -> d, p1 = d-1; U = shift_up(p1)
__device__ __forceinline__ void step(const Tr &t, D2 &io, const D2 &p1, f2 em, f2 exs, f2 exl, f2 eys, f2 eyl, uint64_t k0, uint64_t k1) {
    D2 U;
    U.m = f2{p1.m.y, dppf(p1.m.x)}, U.sx = f2{p1.sx.y, dppf(p1.sx.x)}, U.sy = f2{p1.sy.y, dppf(p1.sy.x)};
    U.lx = f2{p1.lx.y, dppf(p1.lx.x)}, U.ly = f2{p1.ly.y, dppf(p1.ly.x)};
    U.e0 = p1.e1, U.e1 = dppi(p1.e0);
    const int r0 = max(p1.e0, max(io.e0, U.e0)), r1 = max(p1.e1, max(io.e1, U.e1));
    const f2 fL{scale2(p1.e0 - r0), scale2(p1.e1 - r1)}, fM{scale2(io.e0 - r0), scale2(io.e1 - r1)}, fU{scale2(U.e0 - r0), scale2(U.e1 - r1)};
    f2 a, m, sx, sy, lx, ly;
    a = sp(t.mm) * io.m; a = fma2(sp(t.sxm), io.sx, a); a = fma2(sp(t.sym), io.sy, a); a = fma2(sp(t.lxm), io.lx, a); a = fma2(sp(t.lym), io.ly, a);
    m = (fM * em) * a;
    a = sp(t.msx) * p1.m; a = fma2(sp(t.sxsx), p1.sx, a); a = fma2(sp(t.sysx), p1.sy, a); sx = (fL * exs) * a;
    a = sp(t.mlx) * p1.m; a = fma2(sp(t.lxlx), p1.lx, a); lx = (fL * exl) * a;
    a = sp(t.msy) * U.m; a = fma2(sp(t.sysy), U.sy, a); a = fma2(sp(t.sxsy), U.sx, a); sy = (fU * eys) * a;
    a = sp(t.mly) * U.m; a = fma2(sp(t.lyly), U.ly, a); ly = (fU * eyl) * a;
    const float v0 = fmaxf(fmaxf(m.x, sx.x), fmaxf(fmaxf(sy.x, lx.x), ly.x)), v1 = fmaxf(fmaxf(m.y, sx.y), fmaxf(fmaxf(sy.y, lx.y), ly.y));
    const int b0 = __builtin_bit_cast(int, v0) & 0x7f800000, b1 = __builtin_bit_cast(int, v1) & 0x7f800000;
    const f2 inv{__builtin_bit_cast(float, 0x7e800000 - b0), __builtin_bit_cast(float, 0x7e800000 - b1)};
    m *= inv, sx *= inv, sy *= inv, lx *= inv, ly *= inv;
    int e0 = v0 > 0.f ? r0 + (b0 >> 23) - 126 : -(1 << 28), e1 = v1 > 0.f ? r1 + (b1 >> 23) - 126 : -(1 << 28);
    const bool in0 = __builtin_amdgcn_inverse_ballot_w64(k0), in1 = __builtin_amdgcn_inverse_ballot_w64(k1);
    io.m = f2{in0 ? m.x : 0.f, in1 ? m.y : 0.f}, io.sx = f2{in0 ? sx.x : 0.f, in1 ? sx.y : 0.f}, io.sy = f2{in0 ? sy.x : 0.f, in1 ? sy.y : 0.f};
    io.lx = f2{in0 ? lx.x : 0.f, in1 ? lx.y : 0.f}, io.ly = f2{in0 ? ly.x : 0.f, in1 ? ly.y : 0.f};
    io.e0 = in0 ? e0 : -(1 << 28), io.e1 = in1 ? e1 : -(1 << 28);
}
__global__ void k(Tr t, const f2 *in, f2 *out, const uint64_t *msk, int n) {
    D2 A, B;
    const int l = threadIdx.x;
    A.m = in[l], A.sx = in[l + 64], A.sy = in[l + 128], A.lx = in[l + 192], A.ly = in[l + 256], A.e0 = 0, A.e1 = 0;
    B = A;
    const f2 *em = in + 512;
    for (int i = 0; i < n; ++i) {
        const uint64_t k0 = __builtin_amdgcn_readfirstlane(msk[2 * i]), k1 = __builtin_amdgcn_readfirstlane(msk[2 * i + 1]);
        step(t, B, A, em[l], em[l + 64], em[l + 128], em[l + 192], em[l + 256], k0, k1);
        step(t, A, B, em[l + 1], em[l + 65], em[l + 129], em[l + 193], em[l + 257], k1, k0);
        em += 320;
    }
    out[l] = A.m + A.sx + A.sy + A.lx + A.ly + B.m + f2{(float)A.e0, (float)B.e1};
}
