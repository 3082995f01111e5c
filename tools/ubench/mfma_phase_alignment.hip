// Does the chip clock differently when the waves of a workgroup alternate matrix and vector phases IN LOCKSTEP (a barrier per tile) or FREE-RUNNING?  (round 6: the attention
// without its tile barrier ran the same work at a 15 % higher clock and lower board power, LAB_NOTES calls h - n.)  8 waves per workgroup, 2 workgroups per CU (4 waves per SIMD),
// every wave loops over "tiles": 24 dependent-free f16 MFMAs (16x16x32), then NV VALU instructions (v_exp / v_fma mix), then optionally s_barrier.
//   A  MFMA only, no barrier          B  MFMA + VALU, no barrier (free-running)         C  MFMA + VALU, one s_barrier per tile (lockstep)
//   D  as C, but the waves 4-7 run the phases in the opposite order (VALU first): complementary pairing with a barrier
//   E  as B with a per-wave random extra delay once at the start (decorrelated on purpose)
// Reported: time, s_memtime clock, MFMA TFLOP/s.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form mfma_phase_alignment.hip -o mfma_phase_alignment
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define MMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512, 4) void spin(float* out, long long* ticks, int iters) {
    const unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u32x4 a[4], b[2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a[i][j] = ((h * (2 * i + 3 + j)) & 0x83ff83ffu) | 0x30003000u;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) b[i][j] = ((h * (7 * i + 5 + j)) & 0x83ff83ffu) | 0x30003000u;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (float)((h >> i) & 255);
    const int wave = threadIdx.x >> 6;
    if (V == 4) for (unsigned d = 0; d < (h >> 7) % 40u; ++d) __builtin_amdgcn_s_sleep(8);
    const long long t0 = clock64();
    auto mfma_phase = [&]() {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) MMA(acc[i], a[i], b[r & 1]);
    };
    auto valu_phase = [&]() {     // ~ the softmax of a 64-key tile per lane: 16 exp2, max / add / fma / convert mix, all dependent chains of length <= 4
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = x[i];
            v = __builtin_amdgcn_exp2f(v * 0.25f - 1.f);
            v = __builtin_fmaf(v, 0.999f, 0.01f);
            x[i] = __builtin_fmaxf(v, x[(i + 1) & 15] * 0.5f);
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) { x[i] += x[i + 1] * 0.3f; x[i + 1] = x[i] - x[i + 1]; }
    };
    for (int it = 0; it < iters; ++it) {
        if (V == 0) { mfma_phase(); mfma_phase(); }
        else if (V == 3 && wave >= 4) { valu_phase(); mfma_phase(); __builtin_amdgcn_s_barrier(); valu_phase(); mfma_phase(); __builtin_amdgcn_s_barrier(); }
        else {
            mfma_phase(); valu_phase();
            if (V == 2 || V == 3) __builtin_amdgcn_s_barrier();
            mfma_phase(); valu_phase();
            if (V == 2 || V == 3) __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = clock64() - t0;
}

template <int V> void run(const char* name, int iters) {
    float* out; long long* ticks; CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&ticks, 64));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(spin<V>, dim3(512), dim3(512), 0, 0, out, ticks, iters / 10 + 1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(spin<V>, dim3(512), dim3(512), 0, 0, out, ticks, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    long long t; CHK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double mfmas = 512.0 * 8 * iters * 48.0, flop = mfmas * 16384.0;
    printf("%-72s %8.2f ms  clock %7.1f MHz  %7.1f TFLOP/s of MFMA  (%.0f cycles per tile pair and wave)\n", name, ms, t / (ms * 1e3), flop / (ms * 1e9), t / (double)iters);
    CHK(hipFree(out)); CHK(hipFree(ticks));
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A  MFMA only (48 per iteration), no barrier", 40000);
        run<1>("B  MFMA + VALU phases, free-running (no barrier)", 40000);
        run<2>("C  MFMA + VALU phases, one s_barrier per tile (8 waves in lockstep)", 40000);
        run<3>("D  as C, waves 4-7 in the opposite phase order (complementary pairs)", 40000);
        run<4>("E  as B, every wave delayed by a random amount once at the start", 40000);
    }
    return 0;
}
