// lightglue_amd — MFMA GEMM for every linear layer of the matcher (ref lightglue.py:165 Wqkv,
// :171 out_proj, :152-157 ffn, :204-205 to_qk/to_v, :227 to_out, :289 final_proj, :521 input_proj)
// and for the descriptor similarity matrix (ref :292).
//
// Tile: 128 x 128 outputs per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 each = 4 x 4 MFMA
// tiles of 16 x 16).  One stage = 128 bytes of contraction axis per row (64 x 16-bit or 32 x f32),
// staged global -> registers -> LDS (the fp32 -> operand conversion happens in the register hop),
// next stage's global loads in flight while the current stage is multiplied.
// LDS rows are 128 B, 16-byte slots XOR-swizzled (lg_common.h lds_off) => conflict-free b128 reads.
//
// PREC_F16X3 ("split f16"): x = hi + lo (both f16), three MFMAs per product (hi*hi + hi*lo + lo*hi), fp32
// accumulate — the gfx950 stand-in for the xf32 path that CDNA4 dropped; ~2^-22 relative operand
// error, which is what index parity with the fp32 reference needs.
#include "lg_kernels.h"

namespace lg {

constexpr int GBM = 128, GBN = 128, GTHREADS = 256;
constexpr int TILE_BYTES = 128 * 128;  // one operand part of one stage

template <int PREC> struct PT;
template <> struct PT<PREC_F32> { typedef TagF32 Tag; static constexpr int KE = 32, NPART = 1; };
template <> struct PT<PREC_BF16> { typedef TagBF16 Tag; static constexpr int KE = 64, NPART = 1; };
template <> struct PT<PREC_F16> { typedef TagF16 Tag; static constexpr int KE = 64, NPART = 1; };
template <> struct PT<PREC_F16X3> { typedef TagF16 Tag; static constexpr int KE = 64, NPART = 2; };

// Register image of one fp32-sourced operand tile slice owned by a thread: 4 chunks.
template <int PREC> struct F32Stage {
    static constexpr int NV = PT<PREC>::Tag::EPC / 4;  // float4 per chunk (1 for f32, 2 for 16-bit)
    f32x4 v[4][NV];
    __device__ __forceinline__ void load(const float* base, int ld, int k0, int tid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + GTHREADS * i, row = c >> 3, slot = c & 7;
            const float* p = base + (long long)row * ld + k0 + slot * PT<PREC>::Tag::EPC;
#pragma unroll
            for (int j = 0; j < NV; ++j) v[i][j] = *reinterpret_cast<const f32x4*>(p + 4 * j);
        }
    }
    __device__ __forceinline__ void store(char* part0, int tid) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + GTHREADS * i, row = c >> 3, slot = c & 7;
            const int off = lds_off<128>(row, slot);
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(part0 + off) = v[i][0];
            } else if constexpr (PT<PREC>::NPART == 2) {
                u32x4 hi, lo;
                split8<typename PT<PREC>::Tag>(v[i][0], v[i][1], hi, lo);
                *reinterpret_cast<u32x4*>(part0 + off) = hi;
                *reinterpret_cast<u32x4*>(part0 + TILE_BYTES + off) = lo;
            } else {
                *reinterpret_cast<u32x4*>(part0 + off) = pack8<typename PT<PREC>::Tag>(v[i][0], v[i][1]);
            }
        }
    }
};
// Register image of a pre-packed weight tile slice: 4 chunks per part.
template <int PREC> struct PackedStage {
    u32x4 v[PT<PREC>::NPART][4];
    __device__ __forceinline__ void load(const void* hi, const void* lo, int ldk /*elements*/, int k0, int tid) {
        constexpr int ES = sizeof(typename PT<PREC>::Tag::elem), EPC = PT<PREC>::Tag::EPC;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + GTHREADS * i, row = c >> 3, slot = c & 7;
            const long long e = (long long)row * ldk + k0 + slot * EPC;
            v[0][i] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(hi) + e * ES);
            if constexpr (PT<PREC>::NPART == 2) v[1][i] = *reinterpret_cast<const u32x4*>(static_cast<const char*>(lo) + e * ES);
        }
    }
    __device__ __forceinline__ void store(char* part0, int tid) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + GTHREADS * i, row = c >> 3, slot = c & 7;
            const int off = lds_off<128>(row, slot);
            *reinterpret_cast<u32x4*>(part0 + off) = v[0][i];
            if constexpr (PT<PREC>::NPART == 2) *reinterpret_cast<u32x4*>(part0 + TILE_BYTES + off) = v[1][i];
        }
    }
};

// One stage of MFMAs from LDS.  smA/smB point at part 0 of each operand.
template <int PREC>
__device__ __forceinline__ void compute_stage(f32x4 (&acc)[4][4], const char* smA, const char* smB, int wm, int wn, int lane) {
    typedef typename PT<PREC>::Tag Tag;
    const int lr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        u32x4 a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = *reinterpret_cast<const u32x4*>(smA + lds_off<128>(wm * 64 + t * 16 + lr, ks * 4 + g));
            b[t] = *reinterpret_cast<const u32x4*>(smB + lds_off<128>(wn * 64 + t * 16 + lr, ks * 4 + g));
        }
        if constexpr (PT<PREC>::NPART == 2) {
            u32x4 al[4], bl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                al[t] = *reinterpret_cast<const u32x4*>(smA + TILE_BYTES + lds_off<128>(wm * 64 + t * 16 + lr, ks * 4 + g));
                bl[t] = *reinterpret_cast<const u32x4*>(smB + TILE_BYTES + lds_off<128>(wn * 64 + t * 16 + lr, ks * 4 + g));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    mma_chunk<Tag>(acc[mt][nt], al[mt], b[nt]);
                    mma_chunk<Tag>(acc[mt][nt], a[mt], bl[nt]);
                    mma_chunk<Tag>(acc[mt][nt], a[mt], b[nt]);
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) mma_chunk<Tag>(acc[mt][nt], a[mt], b[nt]);
        }
    }
}

// Main loop.  A (and A2 for k >= K1) are fp32 rows of the tile; B is either packed weights
// (BF32 = false: Bhi/Blo, ldb elements) or fp32 rows (BF32 = true: Bhi as float*, ldb floats).
template <int PREC, bool BF32>
__device__ __forceinline__ void gemm_mainloop(f32x4 (&acc)[4][4], const float* A, int lda, const float* A2, int lda2,
                                              int K1, int K, const void* Bhi, const void* Blo, int ldb, char* smem) {
    constexpr int KE = PT<PREC>::KE, NPART = PT<PREC>::NPART;
    char* smA = smem;
    char* smB = smem + NPART * TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    F32Stage<PREC> ra;
    F32Stage<PREC> rbf;     // used when BF32
    PackedStage<PREC> rbp;  // used otherwise
    auto load_stage = [&](int k0) {
        if (k0 < K1) ra.load(A, lda, k0, tid); else ra.load(A2, lda2, k0 - K1, tid);
        if constexpr (BF32) rbf.load(static_cast<const float*>(Bhi), ldb, k0, tid);
        else rbp.load(Bhi, Blo, ldb, k0, tid);
    };
    load_stage(0);
    for (int k0 = 0; k0 < K; k0 += KE) {
        __syncthreads();  // everyone finished reading the previous stage
        ra.store(smA, tid);
        if constexpr (BF32) rbf.store(smB, tid); else rbp.store(smB, tid);
        __syncthreads();
        load_stage(k0 + KE < K ? k0 + KE : k0);  // in flight during the MFMAs below (clamped, never branched:
                                                 // a conditional load makes hipcc wait vmcnt(0) at the join)
        __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ahead of the MFMAs (hipcc would sink it)
        compute_stage<PREC>(acc, smA, smB, wm, wn, lane);
    }
}

template <int PREC, int EPI>
__global__ __launch_bounds__(GTHREADS) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware tile order (guide T1): workgroup id b runs on XCD b % 8, so give every XCD a contiguous
    // range of virtual ids, column tiles fastest -> the Nout/128 workgroups that re-read one 128-row A
    // tile run back-to-back on ONE XCD and hit its L2 instead of going to HBM Nout/128 times.
    const int ncol = a.Nout / GBN;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const TileLoc t = locate_tile(a.rs, v / ncol, GBM);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int n0 = (v % ncol) * GBN;
    const char* W = static_cast<const char*>(a.W);
    const char* Wlo = static_cast<const char*>(a.Wlo);
    const float* bias = a.bias;
    if (a.layer_of_pair) {
        const long long L = a.layer_of_pair[t.pair];
        constexpr int ES = sizeof(typename PT<PREC>::Tag::elem);
        W += L * a.w_layer_stride * ES;
        if (Wlo) Wlo += L * a.w_layer_stride * ES;
        if (bias) bias += L * a.b_layer_stride;
    }
    constexpr int ES = sizeof(typename PT<PREC>::Tag::elem);
    f32x4 acc[4][4];
    gemm_mainloop<PREC, false>(acc, a.A + (long long)t.grow0 * a.lda, a.lda,
                               a.A2 ? a.A2 + (long long)t.grow0 * a.lda2 : nullptr, a.lda2, a.K1, a.K,
                               W + (long long)n0 * a.K * ES, Wlo ? Wlo + (long long)n0 * a.K * ES : nullptr, a.K, smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, g = lane >> 4;
    {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = n0 + wn * 64 + nt * 16 + lr;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long row = t.grow0 + wm * 64 + mt * 16 + g * 4 + r;
                    float* p = a.out + row * a.ldo + col;
                    const float v = acc[mt][nt][r] + bv;
                    if constexpr (EPI == EPI_STORE) *p = v * a.out_scale; else *p += v;
                }
        }
    }
}

template <int PREC>
__global__ __launch_bounds__(GTHREADS) void sim_kernel(SimArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware order: all tiles of one pair run back to back on ONE XCD, whose L2 (4 MB) then holds the pair's two projected
    // descriptor sets (2 x n x 1 KB) for every tile that re-reads them — in the plain (x, y, pair) grid order the 8 XCDs each
    // pulled every pair's operands from HBM (round 2: 4.6x over-fetch, profiles/r02d_pmc_fetch.md)
    const int tx = a.rs.cap0 / GBM, ty = a.rs.cap1 / GBN;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v / (tx * ty), rem = v - pair * (tx * ty);
    const int a0 = (rem / ty) * GBM, b0 = (rem % ty) * GBN;
    if (a0 >= a.rs.len[2 * pair] || b0 >= a.rs.len[2 * pair + 1]) return;
    const long long rowA = seg_row_base(a.rs, 2 * pair) + a0, rowB = seg_row_base(a.rs, 2 * pair + 1) + b0;
    f32x4 acc[4][4];
    gemm_mainloop<PREC, true>(acc, a.X + rowA * a.ldx, a.ldx, nullptr, 0, a.K, a.K, a.X + rowB * a.ldx, nullptr, a.ldx, smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, g = lane >> 4;
    float* out = a.sim + (long long)pair * a.rs.cap0 * a.rs.cap1;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long row = a0 + wm * 64 + mt * 16 + g * 4 + r;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) out[row * a.rs.cap1 + b0 + wn * 64 + nt * 16 + lr] = acc[mt][nt][r];
        }
}

template <int PREC> static constexpr int smem_bytes() { return 2 * PT<PREC>::NPART * TILE_BYTES; }

template <int PREC, int EPI>
static hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    dim3 grid((R / GBM) * (a.Nout / GBN));
    auto kern = gemm_kernel<PREC, EPI>;
    constexpr int smem = smem_bytes<PREC>();
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(GTHREADS), smem, s, a);
    return hipGetLastError();
}

template <int PREC>
static hipError_t launch_prec(int epi, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case EPI_STORE: return launch_one<PREC, EPI_STORE>(a, s);
        case EPI_RESID: return launch_one<PREC, EPI_RESID>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemm(int prec, int epi, const GemmArgs& a, hipStream_t s) {
    if (a.Nout % GBN || a.K % 64 || a.K1 % 64 || a.rs.cap0 % GBM || a.rs.cap1 % GBM) return hipErrorInvalidValue;
    switch (prec) {
        case PREC_F32: return launch_prec<PREC_F32>(epi, a, s);
        case PREC_BF16: return launch_prec<PREC_BF16>(epi, a, s);
        case PREC_F16: return launch_prec<PREC_F16>(epi, a, s);
        case PREC_F16X3: return launch_prec<PREC_F16X3>(epi, a, s);
    }
    return hipErrorInvalidValue;
}

template <int PREC> static hipError_t launch_sim_prec(const SimArgs& a, hipStream_t s) {
    dim3 grid((a.rs.cap0 / GBM) * (a.rs.cap1 / GBN) * a.rs.B);
    auto kern = sim_kernel<PREC>;
    constexpr int smem = smem_bytes<PREC>();
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(GTHREADS), smem, s, a);
    return hipGetLastError();
}
hipError_t launch_sim(int prec, const SimArgs& a, hipStream_t s) {
    if (a.K % 64) return hipErrorInvalidValue;
    switch (prec) {
        case PREC_F32: return launch_sim_prec<PREC_F32>(a, s);
        case PREC_BF16: return launch_sim_prec<PREC_BF16>(a, s);
        case PREC_F16: return launch_sim_prec<PREC_F16>(a, s);
        case PREC_F16X3: return launch_sim_prec<PREC_F16X3>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace lg
