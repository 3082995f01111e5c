// Microbenchmark (experiment): MFMA issue rate of the fused tail's accumulate pattern with 8 waves per CU (2 per SIMD):
// 16 tiles of v_mfma_f32_16x16x32_bf16 x 3 dependent products  vs  4 tiles of v_mfma_f32_32x32x16_bf16 x 3 dependent products.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: 16x16x32 tile-major (3 dependent in a row), 1: 16x16x32 product-major, 2: 32x32x16 tile-major, 3: 32x32x16 product-major
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 a[4][2], b[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) { a[i][p] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + i}; b[i][p] = u32x4{0x3f003f00u, 0x3f003f00u + p, 0x3f003f00u, 0x3f003f00u}; }
    float s = 0.f;
    if constexpr (MODE < 2 || MODE == 4) {
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
            if constexpr (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 t1, t2;
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(t1) : "v"(b[j][0]), "v"(a[i][1]), "v"(acc[i][j]));
                        asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(t2) : "v"(b[j][1]), "v"(a[i][0]), "v"(t1));
                        asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %3\n\ts_nop 7" : "=&v"(acc[i][j]) : "v"(b[j][0]), "v"(a[i][0]), "v"(t2));
                    }
            } else if constexpr (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j][0]), __builtin_bit_cast(bf16x8, a[i][1]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j][1]), __builtin_bit_cast(bf16x8, a[i][0]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j][0]), __builtin_bit_cast(bf16x8, a[i][0]), acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[j][pr == 1]), __builtin_bit_cast(bf16x8, a[i][pr == 0]), acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {   // one iteration = 32 k = two k16 steps (same FLOPs as the 16x16 iteration)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) { asm volatile("" : "+v"(a[i][p])); asm volatile("" : "+v"(b[i][p])); }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[2 * ks + j][0]), __builtin_bit_cast(bf16x8, a[2 * ks + i][1]), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[2 * ks + j][1]), __builtin_bit_cast(bf16x8, a[2 * ks + i][0]), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[2 * ks + j][0]), __builtin_bit_cast(bf16x8, a[2 * ks + i][0]), acc[i][j], 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[2 * ks + j][pr == 1]), __builtin_bit_cast(bf16x8, a[2 * ks + i][pr == 0]), acc[i][j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 123.456f) out[0] = s;
}
template <int MODE> void run(float* d, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    CHK(hipDeviceSynchronize());
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    const double flop = 256.0 * 8 * iters * 48 * 16384.0;
    printf("%-34s %.3f ms  %.0f TF (%.1f%% of 2500)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 25);
}
int main() {
    float* d; CHK(hipMalloc(&d, 64));
    run<0>(d, "16x16x32 tile-major (dependent x3)"); run<1>(d, "16x16x32 product-major");
    run<2>(d, "32x32x16 tile-major (dependent x3)"); run<3>(d, "32x32x16 product-major");
    run<4>(d, "16x16x32 dep x3, dst != srcC (+nops)");
    run<0>(d, "16x16x32 tile-major (dependent x3)"); run<2>(d, "32x32x16 tile-major (dependent x3)");
    return 0;
}
