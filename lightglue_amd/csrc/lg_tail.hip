// lightglue_amd — fused block tail:  x <- x + ffn(cat[x, out_proj(ctx)])   in ONE kernel.
//   ref lightglue.py:171-172 (SelfBlock: out_proj, ffn) and :227-229 (CrossBlock: to_out, ffn), with
//   ffn = Linear(512,512) -> LayerNorm(512) -> GELU(erf) -> Linear(512,256)   (ref :152-157 / :187-192).
//
// Why fused: as separate kernels this chain moves ~23 KB of HBM traffic per keypoint (fp32 ctx/msg/h/g
// round trips, re-read per column tile) and is bandwidth-bound; fused it moves 3 KB (read ctx, read x,
// write x) and is bound by the matrix cores.
//
// Algebra: ffn.0([x ; Wo ctx + bo]) = [W1x | W1m Wo] [x ; ctx] + (b1 + W1m bo): the out_proj GEMM is folded
// into the first FFN matrix on the host (in double precision) — "Wcat" [512 x 512], "bcat" [512].
//
// Workgroup = 64 keypoint rows, 512 threads = 8 waves; wave w owns output columns [64w, 64w+64) of the
// hidden layer and [32w, 32w+32) of the output.  The A operand (activations) is shared by all waves
// through LDS; every wave needs a DIFFERENT slice of the weights, so weights are never staged in LDS:
// they are pre-packed on the host in MFMA-fragment order and each B fragment is one fully coalesced
// 1 KB wave load straight from L2 into registers (weights are ~2.3 MB per precision plane set and stay
// L2-resident).
//   phase A  h = [x ; ctx] Wcat^T + bcat      K = 512 streamed HBM -> regs -> (convert) -> LDS, double buffered
//   LN/GELU  two-pass row statistics across the 8 waves (LDS), exact erf GELU, all in registers
//   phase B  out = g W2^T + b2 (+ x)          g is written once to LDS in A-operand order (128 KB for split bf16)
//   epilogue out tile staged through LDS, residual add and store as full 1 KB rows
#include "lg_kernels.h"

namespace lg {

constexpr int TBM = 64, TTHREADS = 512;

template <int PREC> struct TT;
template <> struct TT<PREC_F32> { typedef TagF32 Tag; static constexpr int KE = 32, NPART = 1; };
template <> struct TT<PREC_BF16> { typedef TagBF16 Tag; static constexpr int KE = 64, NPART = 1; };
template <> struct TT<PREC_F16> { typedef TagF16 Tag; static constexpr int KE = 64, NPART = 1; };
template <> struct TT<PREC_BF16X3> { typedef TagBF16 Tag; static constexpr int KE = 64, NPART = 2; };

// LDS map (bytes):  [0, G_BYTES) g tiles  — aliased during phase A by the two staging buffers
//                   [G_BYTES, +RED_BYTES) cross-wave reduction scratch
template <int PREC> struct TL {
    static constexpr int STAGES = 512 / TT<PREC>::KE;               // K stages of the 512-long contractions
    static constexpr int TILE = TBM * 128;                          // one plane of one stage: 64 rows x 128 B
    static constexpr int G_PLANE = STAGES * TILE;                   // 64 KB (16-bit) / 128 KB (f32)
    static constexpr int G_BYTES = TT<PREC>::NPART * G_PLANE;
    static constexpr int RED_BYTES = 8 * TBM * 4;
    static constexpr int TOTAL = G_BYTES + RED_BYTES;
};

template <int PREC>
__device__ __forceinline__ void tail_mma(f32x4& acc, const u32x4* a, const u32x4* b) {
    typedef typename TT<PREC>::Tag Tag;
    if constexpr (TT<PREC>::NPART == 2) {
        mma_chunk<Tag>(acc, a[1], b[0]);   // lo * hi
        mma_chunk<Tag>(acc, a[0], b[1]);   // hi * lo
        mma_chunk<Tag>(acc, a[0], b[0]);   // hi * hi
    } else {
        mma_chunk<Tag>(acc, a[0], b[0]);
    }
}

template <int PREC>
__global__ __launch_bounds__(TTHREADS) void tail_kernel(TailArgs a) {
    typedef typename TT<PREC>::Tag Tag;
    constexpr int EPC = Tag::EPC, KE = TT<PREC>::KE, NPART = TT<PREC>::NPART;
    constexpr int STAGES = TL<PREC>::STAGES, TILE = TL<PREC>::TILE, G_PLANE = TL<PREC>::G_PLANE;
    constexpr int NKC = 2 * STAGES;          // 16-byte k-chunks per row (16 for 16-bit, 32 for f32)
    constexpr int NV = EPC / 4;              // float4 loads per staged chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + TL<PREC>::G_BYTES);

    const TileLoc t = locate_tile(a.rs, blockIdx.x, TBM);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;

    // ------------------------------------------------------------------ phase A
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int srow = tid >> 3, sslot = tid & 7;      // this thread's staged chunk: 64 rows x 8 slots
    f32x4 stg[NV];
    auto load_stage = [&](int s) {
        const int k0 = s * KE;
        const float* src = (k0 < 256 ? a.X : a.CTX) + (long long)(t.grow0 + srow) * 256 + (k0 & 255) + sslot * EPC;
#pragma unroll
        for (int j = 0; j < NV; ++j) stg[j] = *reinterpret_cast<const f32x4*>(src + 4 * j);
    };
    auto store_stage = [&](int s) {
        char* buf = smem + (s & 1) * NPART * TILE;
        const int off = lds_off<128>(srow, sslot);
        if constexpr (PREC == PREC_F32) {
            *reinterpret_cast<f32x4*>(buf + off) = stg[0];
        } else if constexpr (PREC == PREC_BF16X3) {
            u32x4 hi, lo;
            split8_bf16(stg[0], stg[1], hi, lo);
            *reinterpret_cast<u32x4*>(buf + off) = hi;
            *reinterpret_cast<u32x4*>(buf + TILE + off) = lo;
        } else {
            *reinterpret_cast<u32x4*>(buf + off) = pack8<Tag>(stg[0], stg[1]);
        }
    };
    // weight fragment: plane p, n-tile nt, k-chunk kc -> 64 lanes x 16 B contiguous
    auto wfrag = [&](const void* base, int p, long long plane_elems, int nt, int kc) -> u32x4 {
        const char* ptr = static_cast<const char*>(base) + (p ? plane_elems * (long long)sizeof(typename Tag::elem) : 0);
        return *reinterpret_cast<const u32x4*>(ptr + ((long long)(nt * NKC + kc) * 64 + lane) * 16);
    };
    const void* Wc = a.Wcat;
    const void* W2 = a.W2;

    u32x4 bf[2][4][NPART];   // double-buffered B fragments of one k-chunk: 4 n-tiles x planes
    auto load_b_A = [&](int buf, int kc) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < NPART; ++p) bf[buf][nt][p] = wfrag(Wc, p, 512LL * 512, w * 4 + nt, kc);
    };
    load_stage(0);
    load_b_A(0, 0);
#pragma unroll 1
    for (int s = 0; s < STAGES; ++s) {
        store_stage(s);
        __syncthreads();
        if (s + 1 < STAGES) load_stage(s + 1);
        const char* buf = smem + (s & 1) * NPART * TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kc = 2 * s + ks;
            if (kc + 1 < NKC) load_b_A((ks + 1) & 1, kc + 1);   // next chunk's weights in flight
            u32x4 af[4][NPART];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int p = 0; p < NPART; ++p)
                    af[mt][p] = *reinterpret_cast<const u32x4*>(buf + p * TILE + lds_off<128>(mt * 16 + lr, ks * 4 + g));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) tail_mma<PREC>(acc[mt][nt], af[mt], bf[ks & 1][nt]);
        }
    }
    // ------------------------------------------------------------------ bias + LayerNorm(512) + GELU
    {
        float bias[4], gam[4], bet[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = w * 64 + nt * 16 + lr;
            bias[nt] = a.bcat[col]; gam[nt] = a.gamma[col]; bet[nt] = a.beta[col];
        }
        float part[4][4];   // [mt][r] partial sums over this lane's 4 columns
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sacc = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) { acc[mt][nt][r] += bias[nt]; sacc += acc[mt][nt][r]; }
                part[mt][r] = sacc;
            }
        auto block_row_sum = [&](float (&p)[4][4]) {   // p[mt][r] -> sum over all 512 columns of row mt*16+4g+r
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = p[mt][r];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    p[mt][r] = v;
                }
            __syncthreads();   // previous users of `red` (and, first time, of the staging buffers) are done
            if (lr == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[w * TBM + mt * 16 + g * 4 + r] = p[mt][r];
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 8; ++ww) v += red[ww * TBM + mt * 16 + g * 4 + r];
                    p[mt][r] = v;
                }
        };
        block_row_sum(part);
        float mean[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mean[mt][r] = part[mt][r] * (1.f / 512.f);
                float sq = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) { const float d = acc[mt][nt][r] - mean[mt][r]; acc[mt][nt][r] = d; sq += d * d; }
                part[mt][r] = sq;
            }
        block_row_sum(part);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rstd = 1.f / sqrtf(part[mt][r] * (1.f / 512.f) + 1e-5f);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float u = acc[mt][nt][r] * rstd * gam[nt] + bet[nt];
                    acc[mt][nt][r] = 0.5f * u * (1.f + erff(u * 0.70710678118654752440f));
                }
            }
    }
    // ------------------------------------------------------------------ g -> LDS in A-operand order
    // all waves passed the second barrier of the last block_row_sum, i.e. nobody reads the staging buffers
    // any more, so the g tiles may overwrite them.
    if constexpr (EPC == 8) {
        char* tile0 = smem + w * TILE;   // this wave's 64 columns are exactly K-stage w of phase B
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
                    // even lanes write row rp, odd lanes row rp+1; each writes the (even col, odd col) pair
                    const bool odd = lr & 1;
                    const float mine = odd ? acc[mt][nt][rp + 1] : acc[mt][nt][rp];
                    const float give = odd ? acc[mt][nt][rp] : acc[mt][nt][rp + 1];
                    const float got = __shfl_xor(give, 1, 64);
                    const float c0 = odd ? got : mine, c1 = odd ? mine : got;   // values at (even col, odd col)
                    const int row = mt * 16 + g * 4 + rp + (odd ? 1 : 0);
                    const int col = nt * 16 + (lr & ~1);
                    const int off = lds_off<128>(row, col >> 3) + (col & 7) * 2;
                    if constexpr (PREC == PREC_BF16X3) {
                        const float h0 = bf16_round(c0), h1 = bf16_round(c1);
                        *reinterpret_cast<uint32_t*>(tile0 + off) = pack2_bf16(h0, h1);
                        *reinterpret_cast<uint32_t*>(tile0 + G_PLANE + off) = pack2_bf16(c0 - h0, c1 - h1);
                    } else {
                        *reinterpret_cast<uint32_t*>(tile0 + off) = pack2<Tag>(c0, c1);
                    }
                }
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = nt * 16 + lr, row = mt * 16 + g * 4 + r;
                    char* tile = smem + (2 * w + (c >> 5)) * TILE;
                    *reinterpret_cast<float*>(tile + lds_off<128>(row, (c & 31) >> 2) + (c & 3) * 4) = acc[mt][nt][r];
                }
    }
    __syncthreads();
    // ------------------------------------------------------------------ phase B: out[64 x 256], wave w -> columns [32w, 32w+32)
    f32x4 acc2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc2[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    u32x4 b2f[2][2][NPART];
    auto load_b_B = [&](int buf, int kc) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < NPART; ++p) b2f[buf][nt][p] = wfrag(W2, p, 256LL * 512, w * 2 + nt, kc);
    };
    load_b_B(0, 0);
#pragma unroll 1
    for (int st = 0; st < STAGES; ++st) {
        const char* tile = smem + st * TILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {     // register double buffer indexed by the compile-time ks
            const int kc = 2 * st + ks;
            if (kc + 1 < NKC) load_b_B((ks + 1) & 1, kc + 1);
            u32x4 af[4][NPART];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int p = 0; p < NPART; ++p)
                    af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * G_PLANE + lds_off<128>(mt * 16 + lr, ks * 4 + g));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) tail_mma<PREC>(acc2[mt][nt], af[mt], b2f[ks & 1][nt]);
        }
    }
    __syncthreads();   // g tiles are dead; reuse the region as a [64][256+4] fp32 output tile
    {
        float* ot = reinterpret_cast<float*>(smem);
        constexpr int OLD = 260;   // padded row stride (floats): rows 4g+r land on different banks
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = w * 32 + nt * 16 + lr;
            const float b2 = a.b2[col];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[(mt * 16 + g * 4 + r) * OLD + col] = acc2[mt][nt][r] + b2;
        }
        __syncthreads();
        const int qlen = a.rs.len[t.seg];
#pragma unroll
        for (int i = 0; i < 8; ++i) {     // 64 rows x 64 float4 = 4096 chunks, 8 per thread; a wave covers one full row
            const int c = tid + TTHREADS * i, row = c >> 6, c4 = c & 63;
            if (t.r0 + row < qlen) {
                float* xp = a.X + (long long)(t.grow0 + row) * 256 + c4 * 4;
                const f32x4 x = *reinterpret_cast<const f32x4*>(xp);
                const f32x4 d = *reinterpret_cast<const f32x4*>(ot + row * OLD + c4 * 4);
                *reinterpret_cast<f32x4*>(xp) = x + d;
            }
        }
    }
}

template <int PREC> static hipError_t launch_tail_prec(const TailArgs& a, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    auto kern = tail_kernel<PREC>;
    constexpr int smem = TL<PREC>::TOTAL > 64 * 260 * 4 ? TL<PREC>::TOTAL : 64 * 260 * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / TBM), dim3(TTHREADS), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_tail(int prec, const TailArgs& a, hipStream_t s) {
    switch (prec) {
        case PREC_F32: return launch_tail_prec<PREC_F32>(a, s);
        case PREC_BF16: return launch_tail_prec<PREC_BF16>(a, s);
        case PREC_F16: return launch_tail_prec<PREC_F16>(a, s);
        case PREC_BF16X3: return launch_tail_prec<PREC_BF16X3>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace lg
