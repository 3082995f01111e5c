// Does it matter — at the board's power limit — WHICH operands consecutive MFMAs read?  (round 6: both big kernels are power-limited, LAB_NOTES.)
// A dense v_mfma_f32_16x16x32_f16 spin, 2 waves per SIMD, 8 independent accumulators, ~200 ms per variant; the figure is TFLOP/s at the clock the chip settles at.
//   V0  every MFMA reads the SAME random A and B registers                     (the round-2 "random" spin)
//   V1  every MFMA of a group of 8 reads its OWN A and B registers             (nothing shared between consecutive instructions)
//   V2  own A, shared B                                                        (what a loop over weight / key tiles with the activation fragment held does)
//   V3  the split-f16 triple per accumulator: (Ah, Bl), (Al, Bh), (Ah, Bh) with lo planes 2^-11 of the hi planes' magnitude
//   V4  as V3 but ordered by shared operand: (Ah, Bl), (Ah, Bh), (Al, Bh)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form mfma_operand_energy.hip -o mfma_operand_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32x4 rnd(unsigned h, unsigned k, unsigned expmask) {   // random f16 pairs with bounded exponents (|x| in [2^-8, 2) or, lo planes, [2^-19, 2^-10))
    u32x4 r;
    for (int i = 0; i < 4; ++i) { h = h * 1664525u + 1013904223u + k; r[i] = (h & 0x83ff83ffu) | expmask | ((h >> 7) & 0x1c001c00u); }
    return r;
}
#define MMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512) void spin(float* out, long long* ticks, int iters) {
    const unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u32x4 a[8], b[8], al[8], bl[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd(h, 2 * i + 1, 0x20002000u); b[i] = rnd(h, 2 * i + 2, 0x20002000u); al[i] = rnd(h, 2 * i + 17, 0x04000400u); bl[i] = rnd(h, 2 * i + 18, 0x04000400u); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) MMA(acc[i], a[0], b[0]);
        } else if constexpr (V == 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) MMA(acc[i], a[i], b[(i + r) & 7]);
        } else if constexpr (V == 2) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) MMA(acc[i], a[i], b[r]);
        } else if constexpr (V == 3) {      // the kernels' order: all (hi, lo), then all (lo, hi), then all (hi, hi) over a group of accumulators
#pragma unroll
            for (int i = 0; i < 8; ++i) MMA(acc[i], a[i], bl[i & 1]);
#pragma unroll
            for (int i = 0; i < 8; ++i) MMA(acc[i], al[i], b[i & 1]);
#pragma unroll
            for (int i = 0; i < 8; ++i) MMA(acc[i], a[i], b[i & 1]);
        } else {                            // per accumulator: consecutive instructions share one operand
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                MMA(acc[i], a[i], bl[i & 1]); MMA(acc[i + 1], a[i + 1], bl[i & 1]);
                MMA(acc[i], a[i], b[i & 1]); MMA(acc[i + 1], a[i + 1], b[i & 1]);
                MMA(acc[i], al[i], b[i & 1]); MMA(acc[i + 1], al[i + 1], b[i & 1]);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = clock64() - t0;
}

template <int V> void run(const char* name, int iters) {
    float* out; long long* ticks; CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&ticks, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(spin<V>, dim3(256), dim3(512), 0, 0, out, ticks, iters / 10 + 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(spin<V>, dim3(256), dim3(512), 0, 0, out, ticks, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    long long t; CHK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double mfmas = 256.0 * 8 * iters * 24.0, flop = mfmas * 16384.0;
    printf("%-64s %8.2f ms  clock %7.1f MHz  %7.1f TFLOP/s  (%.2f cycles per MFMA per SIMD)\n", name, ms, t / (ms * 1e3), flop / (ms * 1e9), (t / (double)iters) / (24.0 * 2));
    CHK(hipFree(out)); CHK(hipFree(ticks));
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("V0 same A, same B every MFMA", 70000);
        run<1>("V1 own A and own B every MFMA", 70000);
        run<2>("V2 own A, shared B", 70000);
        run<3>("V3 split triple, kernels' order (hl | lh | hh over the group)", 70000);
        run<4>("V4 split triple, consecutive MFMAs share an operand", 70000);
    }
    return 0;
}
