// Microkernel for a later round: the 512-long contraction of the fused tail's ffn.0,
//      H = X W^T     X [R][512] fp32 (keypoint rows), W [512][512] fp32,
// as   hi(x) hi(w)  on v_mfma_f32_16x16x32_f16            (hi = f16 rounding, 11 bits)
//    + hi(x) lo(w) + lo(x) hi(w)  on v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 e2m3 operands and one E8M0 scale per
//      (row, 32-element k-block)   (lo = value - hi, 2^-12 of the value; 3 mantissa bits are enough for it: tools/study_fp8_cross.py)
// i.e. 16 + 8 matrix instructions per 16x16 tile and K = 512 instead of the 48 of split-bf16 — at the same instruction time
// (tools/ubench/mfma_mx_probe.hip: 8.8 ns per instruction per SIMD for both).
// Same decomposition as lg_tail.hip phase A: workgroup = 64 rows, 8 waves, wave w owns hidden units {w + 8 j} x 16 (j = 0..3),
// transposed products (A operand = weights straight from L2 in fragment order, B operand = activations from LDS).
// Self-checking: fp64 host reference, error reported relative to sum |x||w|.  NOT part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize ffn0_f16_fp6.hip -o ffn0_f16_fp6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int K = 512, N = 512, TBM = 64, THREADS = 512;
// LDS: f16 plane as 8 K-stage tiles of [64 rows][128 B] (XOR-swizzled 16-byte slots, as lg_tail.hip), then two fp6 planes
// [64 rows][16 k-blocks][32 B] (24 B of packed e2m3 + the E8M0 scale in byte 24), 32-byte slots XOR-swizzled with the row
constexpr int TILE = TBM * 128, F16_BYTES = 8 * TILE, P6_BYTES = TBM * 16 * 32, LDS_BYTES = F16_BYTES + 2 * P6_BYTES;   // 128 KB

__device__ __forceinline__ int lds16(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int lds6(int row, int blk) { return row * 512 + ((blk ^ (row & 15)) << 5); }
__device__ __forceinline__ unsigned pack2h(float a, float b) { f16x2 v = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(unsigned, v); }

// 32 consecutive values -> f16 hi (4 chunks of 8), fp6 of hi and of lo = v - hi with their own power-of-two block scales.
// The values are pre-multiplied by the inverse scale (exact), so the conversion's own scale operand stays 1.0.
struct Split32 { u32x4 h16[4]; v6u h6, l6; int sh, sl; };
__device__ __forceinline__ int e8m0_for(float amax) {   // smallest E8M0 byte with amax / 2^(byte - 127) <= 7.5 (the e2m3 maximum)
    if (amax == 0.f) return 127;
    const int e = (int)((__builtin_bit_cast(unsigned, amax * (16.f / 15.f)) >> 23) & 0xFF) - 2;   // exponent of amax * 8 / 7.5, minus log2(4)
    return e < 1 ? 1 : e;
}
__device__ __forceinline__ Split32 split32(const float* v) {
    Split32 s;
    float h[32], l[32], ah = 0.f, al = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { h[i] = (float)(_Float16)v[i]; l[i] = v[i] - h[i]; ah = fmaxf(ah, fabsf(h[i])); al = fmaxf(al, fabsf(l[i])); }
#pragma unroll
    for (int c = 0; c < 4; ++c) s.h16[c] = u32x4{pack2h(h[8 * c], h[8 * c + 1]), pack2h(h[8 * c + 2], h[8 * c + 3]), pack2h(h[8 * c + 4], h[8 * c + 5]), pack2h(h[8 * c + 6], h[8 * c + 7])};
    s.sh = e8m0_for(ah); s.sl = e8m0_for(al);
    const float ih = __builtin_bit_cast(float, (unsigned)(254 - s.sh) << 23), il = __builtin_bit_cast(float, (unsigned)(254 - s.sl) << 23);   // 2^-(byte - 127)
    v16f a0, a1, b0, b1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[i] = h[i] * ih; a1[i] = h[16 + i] * ih; b0[i] = l[i] * il; b1[i] = l[16 + i] * il; }
    s.h6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, 1.0f);
    s.l6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(b0, b1, 1.0f);
    return s;
}

// ---- weight packing (device: the fp6 bit layout is whatever the conversion instruction writes; the matrix instruction reads the same)
// W16 [n-tile 32][k-chunk 16][lane 64][16 B]: f16 hi, lane (lr, g) <- W[nt*16 + lr][kc*32 + 8g .. +8]
// W6  [plane 2: hi, lo][n-tile 32][K-chunk 4][lane 64][32 B]: lane (lr, g) <- the 32 k of block 4c + g of row nt*16 + lr
__global__ __launch_bounds__(64) void pack_weights(const float* W, char* W16, char* W6) {
    const int nt = blockIdx.x, c = blockIdx.y, lane = threadIdx.x, lr = lane & 15, g = lane >> 4;
    const float* src = W + (long long)(nt * 16 + lr) * K + 128 * c + 32 * g;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = src[i];
    const Split32 s = split32(v);
    // this lane's 32 k are chunks kc = 4c + g, sub-chunks 0..3 (8 k each) -> they belong to lanes (lr, 0..3) of k-chunk 4c + g
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<u32x4*>(W16 + ((long long)(nt * 16 + 4 * c + g) * 64 + (q * 16 + lr)) * 16) = s.h16[q];
    unsigned* dh = reinterpret_cast<unsigned*>(W6 + ((long long)((0 * 32 + nt) * 4 + c) * 64 + lane) * 32);
    unsigned* dl = reinterpret_cast<unsigned*>(W6 + ((long long)((1 * 32 + nt) * 4 + c) * 64 + lane) * 32);
#pragma unroll
    for (int i = 0; i < 6; ++i) { dh[i] = s.h6[i]; dl[i] = s.l6[i]; }
    dh[6] = (unsigned)s.sh * 0x01010101u; dl[6] = (unsigned)s.sl * 0x01010101u; dh[7] = 0; dl[7] = 0;
}

__global__ __launch_bounds__(THREADS) void ffn0_kernel(const float* __restrict__ X, const char* __restrict__ W16, const char* __restrict__ W6, float* __restrict__ H, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* p6h = smem + F16_BYTES; char* p6l = p6h + P6_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const long long row0 = (long long)blockIdx.x * TBM;
    for (int rep = 0; rep < reps; ++rep) {
        // ---- activation tile -> LDS: thread = (row = tid >> 3, k-block = tid & 7) of each half of K
        {
            const int srow = tid >> 3, sblk = tid & 7;
#pragma unroll 1
            for (int hf = 0; hf < 2; ++hf) {
                const int blk = hf * 8 + sblk;
                const float* src = X + (row0 + srow) * K + 32 * blk;
                float v[32];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * i); v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3]; }
                const Split32 s = split32(v);
#pragma unroll
                for (int q = 0; q < 4; ++q)   // k = 32 blk + 8 q: K-stage (32 blk + 8 q) / 64, 16-byte slot ((32 blk + 8 q) % 64) / 8
                    *reinterpret_cast<u32x4*>(smem + (blk >> 1) * TILE + lds16(srow, (blk & 1) * 4 + q)) = s.h16[q];
                unsigned* dh = reinterpret_cast<unsigned*>(p6h + lds6(srow, blk)); unsigned* dl = reinterpret_cast<unsigned*>(p6l + lds6(srow, blk));
                *reinterpret_cast<u32x4*>(dh) = u32x4{s.h6[0], s.h6[1], s.h6[2], s.h6[3]}; *reinterpret_cast<u32x4*>(dh + 4) = u32x4{s.h6[4], s.h6[5], (unsigned)s.sh * 0x01010101u, 0u};
                *reinterpret_cast<u32x4*>(dl) = u32x4{s.l6[0], s.l6[1], s.l6[2], s.l6[3]}; *reinterpret_cast<u32x4*>(dl + 4) = u32x4{s.l6[4], s.l6[5], (unsigned)s.sl * 0x01010101u, 0u};
            }
        }
        __syncthreads();
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto w16frag = [&](int nt, int kc) { return *reinterpret_cast<const u32x4*>(W16 + ((long long)((w + 8 * nt) * 16 + kc) * 64 + lane) * 16); };
        auto w6frag = [&](int plane, int nt, int c, u32x4& lo, u32x4& hi) {
            const char* p = W6 + ((long long)((plane * 32 + (w + 8 * nt)) * 4 + c) * 64 + lane) * 32;
            lo = *reinterpret_cast<const u32x4*>(p); hi = *reinterpret_cast<const u32x4*>(p + 16);
        };
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            // main term: 4 k-chunks of 32 on the f16 path
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kc = 4 * c + q;
                u32x4 wf[4], af[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wf[nt] = w16frag(nt, kc);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const u32x4*>(smem + (kc >> 1) * TILE + lds16(mt * 16 + lr, (kc & 1) * 4 + g));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[nt]), __builtin_bit_cast(f16x8, af[mt]), acc[mt][nt], 0, 0, 0);
            }
            // cross terms: (weights lo) x (activations hi), then (weights hi) x (activations lo); K = 128 per instruction
#pragma unroll
            for (int term = 0; term < 2; ++term) {
                const char* ap = term == 0 ? p6h : p6l;
                v8i a6[4]; int sa[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const char* p = ap + lds6(mt * 16 + lr, 4 * c + g);
                    const u32x4 lo = *reinterpret_cast<const u32x4*>(p), hi = *reinterpret_cast<const u32x4*>(p + 16);
                    a6[mt] = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], 0, 0}; sa[mt] = (int)hi[2];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    u32x4 lo, hi;
                    w6frag(term == 0 ? 1 : 0, nt, c, lo, hi);
                    const v8i w6 = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], 0, 0};
                    const int sw = (int)hi[2];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w6, a6[mt], acc[mt][nt], 2, 2, 0, sw, 0, sa[mt]);
                }
            }
        }
        // C^T tiles: lane (lr, g) holds keypoint row lr, hidden units (w + 8 nt) * 16 + 4 g + r
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                *reinterpret_cast<f32x4*>(H + (row0 + mt * 16 + lr) * N + (w + 8 * nt) * 16 + 4 * g) = acc[mt][nt];
        __syncthreads();
    }
}

int main() {
    const int blocks = 1024, R = blocks * TBM;
    std::vector<float> X((size_t)R * K), W((size_t)N * K);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& x : X) x = 4.0f * rnd() * (1.0f + 3.0f * (rnd() > 0.45f));     // a residual-stream-like spread of magnitudes
    for (auto& x : W) x = 0.2f * rnd();
    float *dX, *dW, *dH; char *dW16, *dW6;
    CHK(hipMalloc(&dX, X.size() * 4)); CHK(hipMalloc(&dW, W.size() * 4)); CHK(hipMalloc(&dH, (size_t)R * N * 4));
    CHK(hipMalloc(&dW16, (size_t)N * K * 2)); CHK(hipMalloc(&dW6, (size_t)2 * 32 * 4 * 64 * 32));
    CHK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pack_weights, dim3(32, 4), dim3(64), 0, 0, dW, dW16, dW6);
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(ffn0_kernel, dim3(blocks), dim3(THREADS), LDS_BYTES, 0, dX, dW16, dW6, dH, 1);
    CHK(hipDeviceSynchronize());
    std::vector<float> Hh((size_t)64 * N);
    CHK(hipMemcpy(Hh.data(), dH, Hh.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_rel = 0;
    for (int r = 0; r < 64; ++r) for (int n = 0; n < N; ++n) {
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { ref += (double)X[(size_t)r * K + k] * W[(size_t)n * K + k]; mag += std::fabs((double)X[(size_t)r * K + k] * W[(size_t)n * K + k]); }
        worst = std::fmax(worst, std::fabs(Hh[(size_t)r * N + n] - ref)); worst_rel = std::fmax(worst_rel, std::fabs(Hh[(size_t)r * N + n] - ref) / mag);
    }
    printf("first 64 rows vs fp64: max |err| %.3g, max |err| / sum|x||w| %.3g   (split-bf16 x3 gives ~2e-6 here; a dropped cross term ~2e-4)\n", worst, worst_rel);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 20;
    CHK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(ffn0_kernel, dim3(blocks), dim3(THREADS), LDS_BYTES, 0, dX, dW16, dW6, dH, 1);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("%d rows: %.1f us per launch = %.1f algorithmic TFLOP/s (2 R K N);  lg_tail.hip phase A (split-bf16 x3, incl. prologue): ~35k cycles per 64-row workgroup\n", R, ms * 1e3, 2.0 * R * K * N / (ms * 1e9));
    return 0;
}
