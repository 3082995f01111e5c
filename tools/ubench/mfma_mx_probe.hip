// Probe for a later round: operand layout, scale semantics and rate of the block-scaled matrix instruction
//   v_mfma_scale_f32_16x16x128_f8f6f4   with fp8 e4m3 (cbsz = blgp = 0) and fp6 e2m3 (cbsz = blgp = 2) operands.
// Hypothesis checked (from ck_tile's kAMLane = 16, kABKLane = 4, kABKPerLane = 32): lane (lr = lane & 15, g = lane >> 4) supplies
// row lr of A (column lr of B^T), the 32 CONSECUTIVE k in [32 g, 32 g + 32), element i in byte i (fp8) / bits [6 i, 6 i + 6) (fp6),
// and ITS OWN E8M0 scale (value 2^(byte - 127)) for exactly those 32 elements; C / D as for the other 16x16 shapes
// (column n = lane & 15, row m = 4 (lane >> 4) + r).  Then the rates against v_mfma_f32_16x16x32_f16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form mfma_mx_probe.hip -o mfma_mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// A, B: [16][128] fp32 holding values that are exact in the target format; SA, SB: [16][4] E8M0 bytes; C: [16][16]
template <int FMT>   // 0 = fp8 e4m3, 2 = fp6 e2m3
__global__ void probe(const float* A, const float* B, const int* SA, const int* SB, float* C) {
    const int lane = threadIdx.x, lr = lane & 15, g = lane >> 4;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    const float* ar = A + lr * 128 + 32 * g; const float* br = B + lr * 128 + 32 * g;
    if constexpr (FMT == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // dword i = elements 4 i .. 4 i + 3, low byte first
            int wa = 0, wb = 0;
            wa = __builtin_amdgcn_cvt_pk_fp8_f32(ar[4 * i], ar[4 * i + 1], wa, false); wa = __builtin_amdgcn_cvt_pk_fp8_f32(ar[4 * i + 2], ar[4 * i + 3], wa, true);
            wb = __builtin_amdgcn_cvt_pk_fp8_f32(br[4 * i], br[4 * i + 1], wb, false); wb = __builtin_amdgcn_cvt_pk_fp8_f32(br[4 * i + 2], br[4 * i + 3], wb, true);
            a[i] = wa; b[i] = wb;
        }
    } else {
        v16f a0, a1, b0, b1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0[i] = ar[i]; a1[i] = ar[16 + i]; b0[i] = br[i]; b1[i] = br[16 + i]; }
        const v6u pa = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, 1.0f), pb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(b0, b1, 1.0f);
#pragma unroll
        for (int i = 0; i < 6; ++i) { a[i] = (int)pa[i]; b[i] = (int)pb[i]; }
    }
    const int sa = SA[lr * 4 + g] * 0x01010101, sb = SB[lr * 4 + g] * 0x01010101;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, FMT, FMT, 0, sa, 0, sb);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + lr] = acc[r];
}

template <int KIND>   // 0: f16 16x16x32, 1: fp8 16x16x128, 2: fp6 16x16x128
__global__ __launch_bounds__(512) void spin(float* out, int iters, unsigned seed) {
    const unsigned h = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) * seed;
    v8i a = {(int)(h & 0x3b3b3b3b), (int)((h * 3u) & 0x3b3b3b3b), (int)((h * 5u) & 0x3b3b3b3b), (int)((h * 7u) & 0x3b3b3b3b), (int)((h * 11u) & 0x3b3b3b3b), (int)((h * 13u) & 0x3b3b3b3b), (int)((h * 17u) & 0x3b3b3b3b), (int)((h * 19u) & 0x3b3b3b3b)};
    v8i b = {(int)((h * 23u) & 0x3b3b3b3b), (int)((h * 29u) & 0x3b3b3b3b), (int)((h * 31u) & 0x3b3b3b3b), (int)((h * 37u) & 0x3b3b3b3b), (int)((h * 41u) & 0x3b3b3b3b), (int)((h * 43u) & 0x3b3b3b3b), (int)((h * 47u) & 0x3b3b3b3b), (int)((h * 53u) & 0x3b3b3b3b)};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), __builtin_bit_cast(f16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3)), acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], KIND == 1 ? 0 : 2, KIND == 1 ? 0 : 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

static std::vector<float> exact_values(int fmt) {   // a spread of magnitudes that are exact in the format
    std::vector<float> v;
    if (fmt == 0) for (int e = -3; e <= 3; ++e) for (int m = 0; m < 8; ++m) v.push_back(std::ldexp(1.0f + m / 8.0f, e));          // e4m3 normals
    else { for (int m = 0; m < 8; ++m) v.push_back(m / 8.0f); for (int e = 0; e <= 2; ++e) for (int m = 0; m < 8; ++m) v.push_back(std::ldexp(1.0f + m / 8.0f, e)); }   // e2m3: all 32 magnitudes
    return v;
}

template <int FMT> static void check(const char* name) {
    std::vector<float> vals = exact_values(FMT), A(16 * 128), B(16 * 128), C(256);
    std::vector<int> SA(64), SB(64);
    unsigned s = 12345u + FMT;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto& x : A) { x = vals[rnd() % vals.size()] * ((rnd() & 1) ? -1.f : 1.f); }
    for (auto& x : B) { x = vals[rnd() % vals.size()] * ((rnd() & 1) ? -1.f : 1.f); }
    for (int m = 0; m < 16; ++m) for (int g = 0; g < 4; ++g) { SA[m * 4 + g] = 127 + (m + g) % 5 - 2; SB[m * 4 + g] = 127 + (2 * m + 3 * g) % 4 - 1; }
    float *dA, *dB, *dC; int *dSA, *dSB;
    CHK(hipMalloc(&dA, A.size() * 4)); CHK(hipMalloc(&dB, B.size() * 4)); CHK(hipMalloc(&dC, 1024)); CHK(hipMalloc(&dSA, 256)); CHK(hipMalloc(&dSB, 256));
    CHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dSA, SA.data(), 256, hipMemcpyHostToDevice)); CHK(hipMemcpy(dSB, SB.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe<FMT>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    CHK(hipDeviceSynchronize()); CHK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    double worst = 0, worst_noscale = 0; int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double ref = 0, ref_ns = 0;
        for (int g = 0; g < 4; ++g) {
            double part = 0;
            for (int i = 0; i < 32; ++i) part += (double)A[m * 128 + 32 * g + i] * B[n * 128 + 32 * g + i];
            ref += part * std::ldexp(1.0, SA[m * 4 + g] - 127) * std::ldexp(1.0, SB[n * 4 + g] - 127); ref_ns += part;
        }
        const double e = std::fabs(C[m * 16 + n] - ref) / (std::fabs(ref) + 1.0);
        worst = std::fmax(worst, e); worst_noscale = std::fmax(worst_noscale, std::fabs(C[m * 16 + n] - ref_ns) / (std::fabs(ref_ns) + 1.0));
        if (e > 1e-5) ++bad;
    }
    printf("%-10s layout + per-lane E8M0 scale hypothesis: %s  (max rel err %.3g, mismatching entries %d / 256; against the UNSCALED sum: %.3g)\n",
           name, bad == 0 ? "CONFIRMED" : "REFUTED", worst, bad, worst_noscale);
    if (bad) { printf("  C[0][0..7] ="); for (int n = 0; n < 8; ++n) printf(" %.4f", C[n]); printf("\n"); }
}

template <int KIND> static void rate(const char* name, double flop_per_mfma) {
    float* out; CHK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 100000;
    hipLaunchKernelGGL(spin<KIND>, dim3(256), dim3(512), 0, 0, out, iters / 10, 12345u); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL(spin<KIND>, dim3(256), dim3(512), 0, 0, out, iters, 12345u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double mf = 256.0 * 8 * iters * 8;
    printf("%-22s %8.2f ms  %8.1f TFLOP/s  (%.2f ns per instruction per SIMD)\n", name, ms, mf * flop_per_mfma / (ms * 1e9), ms * 1e6 / (mf / 1024));
}

int main() {
    check<0>("fp8 e4m3"); check<2>("fp6 e2m3");
    rate<0>("f16 16x16x32", 16384.0); rate<1>("fp8 16x16x128 (scaled)", 65536.0); rate<2>("fp6 16x16x128 (scaled)", 65536.0);
    return 0;
}
