// Microbenchmark (experiment): do an MFMA-issuing wave and a VALU-issuing wave on the same SIMD overlap?
// 8 waves per workgroup: waves 0-3 run the split-bf16 MFMA pattern, waves 4-7 a GELU-like VALU loop (pk fma, rcp, exp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 u) {
    const f32x2 x = u * 0.70710678118654752440f;
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 den = ax * 0.3275911f + 1.0f;
    const f32x2 tt = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 p = tt * -0.0779742014f + 0.151737503f;
    p = p * tt + 0.39572154f; p = p * tt + -0.574341196f; p = p * tt + 0.810336914f; p = p * tt + -0.151473053f;
    p = p * tt + 0.270560832f; p = p * tt + 0.175431661f; p = p * tt;
    const f32x2 ee = ax * ax * -1.44269504088896340736f;
    const f32x2 e = {__builtin_amdgcn_exp2f(ee[0]), __builtin_amdgcn_exp2f(ee[1])};
    const f32x2 er = 1.0f - p * e;
    const f32x2 half_u = u * 0.5f;
    const f32x2 sgn = {copysignf(er[0], x[0]), copysignf(er[1], x[1])};
    return half_u + half_u * sgn;
}
__device__ __forceinline__ float gelu_fast1(float u) {   // scalar (no packed f32 ops)
    const float x = u * 0.70710678118654752440f;
    const float ax = fabsf(x);
    const float tt = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f, 1.0f));
    float p = __builtin_fmaf(tt, -0.0779742014f, 0.151737503f);
    p = __builtin_fmaf(p, tt, 0.39572154f); p = __builtin_fmaf(p, tt, -0.574341196f); p = __builtin_fmaf(p, tt, 0.810336914f); p = __builtin_fmaf(p, tt, -0.151473053f);
    p = __builtin_fmaf(p, tt, 0.270560832f); p = __builtin_fmaf(p, tt, 0.175431661f); p = p * tt;
    const float e = __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
    const float er = __builtin_fmaf(-p, e, 1.0f);
    const float half_u = u * 0.5f;
    return __builtin_fmaf(half_u, copysignf(er, x), half_u);
}
// mask bit 0: MFMA waves active, bit 1: VALU waves active; mfma_waves: how many of the 8 waves do MFMA (4 or 8)
template <int SCALAR> __global__ __launch_bounds__(512) void k(float* out, int iters, int mask, int viters) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0.f;
    if (w < 4) {
        if (!(mask & 1)) return;
        u32x4 a[4][2], b[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) { a[i][p] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i}; b[i][p] = u32x4{0x3f003f00u, 0x3f003f00u + p, 0x3f003f00u, 0x3f003f00u}; }
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) { asm volatile("" : "+v"(a[i][p])); asm volatile("" : "+v"(b[i][p])); }
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j][pr == 1]), __builtin_bit_cast(bf16x8, a[i][pr == 0]), acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        if (!(mask & 2)) return;
        f32x2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = f32x2{0.01f * lane + i, 0.02f * lane - i};
        for (int it = 0; it < viters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { if (SCALAR == 2) { v[i][0] = __builtin_amdgcn_exp2f(v[i][0]) ; v[i][1] = __builtin_amdgcn_exp2f(v[i][1]); v[i][0] = __builtin_amdgcn_exp2f(v[i][0]) ; v[i][1] = __builtin_amdgcn_exp2f(v[i][1]); }
              else if (SCALAR == 3) { for (int q = 0; q < 8; ++q) { v[i][0] = __builtin_fmaf(v[i][0], 0.999f, 0.001f); v[i][1] = __builtin_fmaf(v[i][1], 1.001f, -0.001f); } }
              else if (SCALAR == 4) { for (int q = 0; q < 8; ++q) v[i] = v[i] * f32x2{0.999f, 1.001f} + f32x2{0.001f, -0.001f}; }
              else if (SCALAR == 5) { for (int q = 0; q < 8; ++q) { uint32_t x0 = __builtin_bit_cast(uint32_t, v[i][0]), x1 = __builtin_bit_cast(uint32_t, v[i][1]);   // integer add, 2 VGPR operands
                  asm volatile("v_add_u32_e32 %0, %0, %1\n\tv_add_u32_e32 %1, %1, %0" : "+v"(x0), "+v"(x1)); v[i][0] = __builtin_bit_cast(float, x0); v[i][1] = __builtin_bit_cast(float, x1); } }
              else if (SCALAR == 6) { for (int q = 0; q < 8; ++q) { asm volatile("v_fma_f32 %0, %0, %2, 0.5\n\tv_fma_f32 %1, %1, %2, 0.5" : "+v"(v[i][0]), "+v"(v[i][1]) : "s"(0.999f)); } }   // fma, ONE VGPR operand (+ one SGPR, one inline constant: gfx9 reads one scalar per VOP3)
              else if (SCALAR == 7) { for (int q = 0; q < 8; ++q) { asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %1, %1, %0, %1" : "+v"(v[i][0]), "+v"(v[i][1])); } }                               // fma, THREE VGPR operands
              else if (SCALAR == 8) { for (int q = 0; q < 8; ++q) { uint32_t h; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_fma_mix_f32 %1, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "=&v"(h), "+v"(v[i][0]) : "v"(v[i][1])); } }   // the split's pair
              else if (SCALAR == 9) { for (int q = 0; q < 8; ++q) { asm volatile("v_max3_f32 %0, %0, %1, %0\n\tv_max3_f32 %1, %1, %0, %1" : "+v"(v[i][0]), "+v"(v[i][1])); } }
              else if (SCALAR) { v[i][0] = gelu_fast1(v[i][0]) + 0.1f; v[i][1] = gelu_fast1(v[i][1]) + 0.1f; } else v[i] = gelu_fast2(v[i]) + 0.1f; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    }
    if (s == 123.456f) out[0] = s;
}
template <int SCALAR> float run(float* d, int iters, int mask, int viters) {
    hipLaunchKernelGGL(k<SCALAR>, dim3(256), dim3(512), 0, 0, d, iters, mask, viters);
    CHK(hipDeviceSynchronize());
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<SCALAR>, dim3(256), dim3(512), 0, 0, d, iters, mask, viters);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    return ms;
}
int main() {
    float* d; CHK(hipMalloc(&d, 64));
    const int iters = 2000;
    const float tm = run<0>(d, iters, 1, 0);
    printf("MFMA waves alone (1 per SIMD): %.3f ms  (%.0f TF)\n", tm, 256.0 * 4 * iters * 48 * 16384.0 / tm / 1e9);
    for (int viters : {2000}) {
        const float tv1 = run<1>(d, iters, 2, viters), tb1 = run<1>(d, iters, 3, viters);
        printf("scalar VALU iters %d: VALU alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f) -> overlap efficiency %.2f\n", viters, tv1, tb1, tm + tv1, fmaxf(tm, tv1), (tm + tv1 - tb1) / fminf(tm, tv1));
        { const float a = run<2>(d, iters, 2, viters * 4), b = run<2>(d, iters, 3, viters * 4); printf("v_exp_f32 only   x%d: alone %.3f both %.3f -> overlap %.2f\n", viters * 4, a, b, (tm + a - b) / fminf(tm, a)); }
        { const float a = run<3>(d, iters, 2, viters), b = run<3>(d, iters, 3, viters); printf("v_fma_f32 only   x%d: alone %.3f both %.3f -> overlap %.2f\n", viters, a, b, (tm + a - b) / fminf(tm, a)); }
        { const float a = run<4>(d, iters, 2, viters), b = run<4>(d, iters, 3, viters); printf("v_pk_fma_f32 only x%d: alone %.3f both %.3f -> overlap %.2f\n", viters, a, b, (tm + a - b) / fminf(tm, a)); }
        { const float a = run<5>(d, iters, 2, viters), b = run<5>(d, iters, 3, viters); printf("v_add_u32 (2 VGPR) x%d: alone %.3f both %.3f -> overlap %.2f\n", viters, a, b, (tm + a - b) / fminf(tm, a)); }
        { const float a = run<6>(d, iters, 2, viters), b = run<6>(d, iters, 3, viters); printf("v_fma_f32 1 VGPR + SGPR + const x%d: alone %.3f both %.3f -> overlap %.2f\n", viters, a, b, (tm + a - b) / fminf(tm, a)); }
        { const float a = run<7>(d, iters, 2, viters), b = run<7>(d, iters, 3, viters); printf("v_fma_f32 3 VGPR x%d: alone %.3f both %.3f -> overlap %.2f\n", viters, a, b, (tm + a - b) / fminf(tm, a)); }
        { const float a = run<8>(d, iters, 2, viters), b = run<8>(d, iters, 3, viters); printf("v_cvt_pk_f16_f32 + v_fma_mix_f32 x%d: alone %.3f both %.3f -> overlap %.2f\n", viters, a, b, (tm + a - b) / fminf(tm, a)); }
        { const float a = run<9>(d, iters, 2, viters), b = run<9>(d, iters, 3, viters); printf("v_max3_f32 x%d: alone %.3f both %.3f -> overlap %.2f\n", viters, a, b, (tm + a - b) / fminf(tm, a)); }
        const float tv = run<0>(d, iters, 2, viters), tb = run<0>(d, iters, 3, viters);
        printf("packed VALU iters %d: VALU alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f) -> overlap efficiency %.2f\n", viters, tv, tb, tm + tv, fmaxf(tm, tv), (tm + tv - tb) / fminf(tm, tv));
    }
    return 0;
}
