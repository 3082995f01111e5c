// What does the matrix pipe SUSTAIN?  A dense v_mfma_f32_16x16x32_bf16 spin (2 waves per SIMD, 8 independent accumulators
// per wave: the pipe never waits) for 2 / 20 / 200 ms, with operand data that toggles (pseudo-random bf16) or does not
// (zeros), while block 0 counts s_memtime ticks: ticks / wall time = the clock the shader engine actually ran at.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>   // 0: bf16 16x16x32, 1: f16 16x16x32, 2: bf16 32x32x16
__global__ __launch_bounds__(512) void spin(float* out, long long* ticks, int iters, unsigned seed) {
    const unsigned h = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) * seed;   // seed 0 -> all-zero operands
    u32x4 a = {h ^ 0x3f803f80u * (seed != 0), (h >> 3) * (seed != 0) | 0x3c003c00u * (seed != 0), (h * 7u) & 0x3fff3fffu, (h * 13u) & 0x3fff3fffu};
    u32x4 b = {(h * 3u) & 0x3fff3fffu, (h * 5u) & 0x3fff3fffu, (h * 11u) & 0x3fff3fffu, (h * 17u) & 0x3fff3fffu};
    if (seed == 0) { a = u32x4{0, 0, 0, 0}; b = a; }
    const long long t0 = clock64();
    if constexpr (KIND == 2) {
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
        if (s == 12345.678f) out[0] = s;
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[i], 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) out[0] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = clock64() - t0;
}

template <int KIND> void run(const char* name, int iters, unsigned seed, int waves_per_simd) {
    float* out; long long* ticks; CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&ticks, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(spin<KIND>, dim3(256), dim3(threads), 0, 0, out, ticks, iters / 10 + 1, seed);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(spin<KIND>, dim3(256), dim3(threads), 0, 0, out, ticks, iters, seed);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    long long t; CHK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double mfmas = 256.0 * (threads / 64) * iters * (KIND == 2 ? 4 : 8), flop = mfmas * 16384.0 * (KIND == 2 ? 2 : 1);
    printf("%-28s seed %u  %d waves/SIMD  %8.2f ms  s_memtime rate %7.1f MHz  %7.1f TFLOP/s  (%.2f MFMA cycles of this clock per MFMA per SIMD)\n", name, seed, waves_per_simd, ms,
           t / (ms * 1e3), flop / (ms * 1e9), (t / (double)iters) / ((KIND == 2 ? 4 : 8) * waves_per_simd));
    CHK(hipFree(out)); CHK(hipFree(ticks));
}

int main() {
    for (int iters : {20000, 200000, 2000000}) {
        run<0>("bf16 16x16x32", iters, 12345u, 2);
        run<0>("bf16 16x16x32 zeros", iters, 0u, 2);
    }
    run<1>("f16 16x16x32", 200000, 12345u, 2);
    run<2>("bf16 32x32x16", 200000, 12345u, 2);
    run<0>("bf16 16x16x32 1 wave/SIMD", 200000, 12345u, 1);
    run<0>("bf16 16x16x32 4 waves/SIMD", 100000, 12345u, 4);
    return 0;
}
