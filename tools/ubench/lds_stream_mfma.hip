// Microbenchmark for the tail-v2 core loop (experiment, not product):
// 8 waves per workgroup, each wave owns 16 keypoint rows (B operand, registers) and streams ALL weight fragments
// (A operand) of a [NT*16 x K] matrix through a 2-slab LDS ring filled by LDS-DMA; split-bf16: 3 MFMAs per fragment pair.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form lds_stream_mfma.hip -o lds_stream_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NT = 32;            // weight tiles (16 output units each) per chunk
constexpr int NKC = 16;           // 32-wide k chunks (K = 512)
constexpr int SLAB = NT * 2 * 1024;   // hi + lo planes: 64 KB
constexpr int THREADS = 512;

__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    float h[8], l[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = (float)(__bf16)a[i]; l[i] = a[i] - h[i]; h[4 + i] = (float)(__bf16)b[i]; l[4 + i] = b[i] - h[4 + i]; }
    hi[0] = pack2_bf16(h[0], h[1]); hi[1] = pack2_bf16(h[2], h[3]); hi[2] = pack2_bf16(h[4], h[5]); hi[3] = pack2_bf16(h[6], h[7]);
    lo[0] = pack2_bf16(l[0], l[1]); lo[1] = pack2_bf16(l[2], l[3]); lo[2] = pack2_bf16(l[4], l[5]); lo[3] = pack2_bf16(l[6], l[7]);
}
__device__ __forceinline__ f32x4 mma(f32x4 acc, u32x4 a, u32x4 b) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// W: [NKC][2 planes][NT][64 lanes][8 bf16]   X: [R][512] fp32   OUT: [R][512] fp32 (h = X W^T)
template <int MODE>   // 0 = full, 1 = no MFMA (reads kept live), 2 = no LDS reads (MFMA on constant frags), 3 = no DMA
__global__ __launch_bounds__(THREADS) void stream_kernel(const char* __restrict__ W, const float* __restrict__ X, float* __restrict__ OUT, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
    const long long row = (long long)blockIdx.x * 128 + w * 16 + lr;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto dma_slab = [&](int kc, int buf) {
        if (MODE == 3) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int piece = w * 8 + i;
            const char* src = W + ((long long)kc * 64 + piece) * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + buf * SLAB + piece * 1024), 16, 0, 0);
        }
    };
    auto load_act = [&](int kc, f32x4& a0, f32x4& a1) {
        const float* p = X + row * 512 + kc * 32 + g * 8;
        a0 = *reinterpret_cast<const f32x4*>(p); a1 = *reinterpret_cast<const f32x4*>(p + 4);
    };
    for (int rep = 0; rep < reps; ++rep) {
        f32x4 r0, r1;
        dma_slab(0, 0); dma_slab(1, 1);
        load_act(0, r0, r1);
        __syncthreads();
#pragma unroll 1
        for (int kc = 0; kc < NKC; ++kc) {
            u32x4 bh, bl;
            split8(r0, r1, bh, bl);
            load_act(kc + 1 < NKC ? kc + 1 : kc, r0, r1);
            const char* buf = smem + (kc & 1) * SLAB;
#pragma unroll
            for (int t0 = 0; t0 < NT; t0 += 4) {
                u32x4 ah[4], al[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == 2) { ah[j] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; al[j] = ah[j]; }
                    else {
                        ah[j] = *reinterpret_cast<const u32x4*>(buf + (t0 + j) * 1024 + lane * 16);
                        al[j] = *reinterpret_cast<const u32x4*>(buf + (NT + t0 + j) * 1024 + lane * 16);
                    }
                }
                if (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(ah[j]), "v"(al[j]));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t0 + j] = mma(acc[t0 + j], al[j], bh);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t0 + j] = mma(acc[t0 + j], ah[j], bl);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t0 + j] = mma(acc[t0 + j], ah[j], bh);
                }
            }
            __syncthreads();                       // everyone done with this slab; own DMA pieces of the next one have landed (vmcnt(0))
            if (kc + 2 < NKC) dma_slab(kc + 2, kc & 1);
        }
    }
    // h^T layout: acc[t][r] = h[row][t*16 + 4g + r]
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(OUT + row * 512 + t * 16 + g * 4) = acc[t];
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int MODE> float run(const char* dW, const float* dX, float* dO, int blocks, int reps, int iters) {
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLAB));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(stream_kernel<MODE>, dim3(blocks), dim3(THREADS), 2 * SLAB, 0, dW, dX, dO, reps);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(blocks), dim3(THREADS), 2 * SLAB, 0, dW, dX, dO, reps);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const int blocks = 512, R = blocks * 128, K = 512, N = 512;
    std::vector<float> Wf((size_t)N * K), X((size_t)R * K);
    srand(1);
    for (auto& v : Wf) v = (rand() / (float)RAND_MAX * 2 - 1) * 0.044f;
    for (auto& v : X) v = (rand() / (float)RAND_MAX * 2 - 1);
    std::vector<uint16_t> Wp((size_t)NKC * 2 * NT * 64 * 8);
    for (int kc = 0; kc < NKC; ++kc) for (int t = 0; t < NT; ++t) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
        const float v = Wf[(size_t)(t * 16 + (lane & 15)) * K + kc * 32 + (lane >> 4) * 8 + j];
        const uint16_t h = f2bf(v);
        Wp[((((size_t)kc * 2 + 0) * NT + t) * 64 + lane) * 8 + j] = h;
        Wp[((((size_t)kc * 2 + 1) * NT + t) * 64 + lane) * 8 + j] = f2bf(v - bf2f(h));
    }
    char* dW; float *dX, *dO;
    CHK(hipMalloc(&dW, Wp.size() * 2)); CHK(hipMalloc(&dX, X.size() * 4)); CHK(hipMalloc(&dO, (size_t)R * N * 4));
    CHK(hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    // correctness (reps = 1)
    run<0>(dW, dX, dO, blocks, 1, 1);
    std::vector<float> O((size_t)R * N); CHK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int rr = 0; rr < 64; ++rr) { const int r = (rr * 1031) % R;
        for (int n = 0; n < N; n += 7) { double s = 0; for (int k = 0; k < K; ++k) s += (double)X[(size_t)r * K + k] * Wf[(size_t)n * K + k];
            maxerr = fmax(maxerr, fabs(s - O[(size_t)r * N + n])); } }
    printf("max abs err vs fp64: %.3e\n", maxerr);
    const double flop = 2.0 * R * K * N * 3;   // issued MFMA flops (3 products)
    for (int reps : {1, 4}) {
        float t0 = run<0>(dW, dX, dO, blocks, reps, 10), t1 = run<1>(dW, dX, dO, blocks, reps, 10), t2 = run<2>(dW, dX, dO, blocks, reps, 10), t3 = run<3>(dW, dX, dO, blocks, reps, 10);
        printf("reps %d: full %.1f us (%.0f TF issued = %.1f%% of 2500)  noMFMA %.1f us  noLDSread %.1f us (%.0f TF)  noDMA %.1f us\n", reps, t0 * 1e3, flop * reps / t0 / 1e9, flop * reps / t0 / 1e9 / 25, t1 * 1e3, t2 * 1e3, flop * reps / t2 / 1e9, t3 * 1e3);
    }
    return 0;
}
