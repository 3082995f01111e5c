// EXPERIMENT BUILDS ONLY (-DLG_EXPERIMENTS; not part of the product library).
// lightglue_amd — fused block tail, streaming variants:  x <- x + ffn(cat[x, out_proj(ctx)])   (see lg_tail.hip for
// the algebra, the reference lines and the weight packing; this file changes the workgroup decomposition only).
//
// Ablations of lg_tail.hip showed that the fused tail is NOT bound by the matrix cores but by the per-CU
// vector-memory path that streams the weight fragments L2 -> VGPR (~24 B/clk/CU): with 64 rows per workgroup every
// weight byte feeds too few MACs.  tailx_kernel<PREC, NW, MT> keeps the structure (activations through LDS, weight
// fragments pre-packed in MFMA order straight from L2, waves split the output columns) but
//   * streams the [x ; ctx] tile through a double-buffered LDS stage (one barrier per 64-wide K stage) instead of
//     keeping it resident, and runs GELU + ffn.3 in steps over a two-slot LDS ring of g, so LDS no longer caps the tile;
//   * <NW = 8, MT = 8>: 128 rows per workgroup — every weight fragment is reused by 8 row tiles instead of 4,
//     i.e. HALF the weight stream per keypoint (variant 2, default);
//   * <NW = 4, MT = 4>: 64 rows, 4 waves, < 70 KB LDS — two unsynchronised workgroups per CU (variant 1; measured
//     +-1 % vs lg_tail.hip: overlapping MFMA and VALU phases does not help a load-bound kernel).
// Wave w owns the hidden n-tiles {w + NW*j} and the output n-tiles {w + NW*n}.
#include "lg_kernels.h"

namespace lg {

template <int PREC> struct QT;
template <> struct QT<PREC_F32> { typedef TagF32 Tag; static constexpr int NPART = 1; };
template <> struct QT<PREC_BF16> { typedef TagBF16 Tag; static constexpr int NPART = 1; };
template <> struct QT<PREC_F16> { typedef TagF16 Tag; static constexpr int NPART = 1; };
template <> struct QT<PREC_BF16X3> { typedef TagBF16 Tag; static constexpr int NPART = 2; };

__device__ __forceinline__ f32x2 gelu4_fast2(f32x2 u) {   // same branch-free GELU as lg_tail.hip
    const f32x2 x = u * 0.70710678118654752440f;
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 den = ax * 0.3275911f + 1.0f;
    const f32x2 tt = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    f32x2 p = tt * -0.0779742014f + 0.151737503f;
    p = p * tt + 0.39572154f;
    p = p * tt + -0.574341196f;
    p = p * tt + 0.810336914f;
    p = p * tt + -0.151473053f;
    p = p * tt + 0.270560832f;
    p = p * tt + 0.175431661f;
    p = p * tt;
    const f32x2 ee = ax * ax * -1.44269504088896340736f;
    const f32x2 e = {__builtin_amdgcn_exp2f(ee[0]), __builtin_amdgcn_exp2f(ee[1])};
    const f32x2 er = 1.0f - p * e;
    const f32x2 half_u = u * 0.5f;
    const f32x2 sgn = {copysignf(er[0], x[0]), copysignf(er[1], x[1])};
    return half_u + half_u * sgn;
}

// N accumulators of one 16-row tile against one A fragment; product-major (hi*lo terms first)
template <int PREC, int N>
__device__ __forceinline__ void q_mma_row(f32x4* acc, const u32x4* a, const u32x4 (*b)[QT<PREC>::NPART]) {
    typedef typename QT<PREC>::Tag Tag;
    constexpr int NP = QT<PREC>::NPART;
#pragma unroll
    for (int pr = 0; pr < (NP == 2 ? 3 : 1); ++pr)
#pragma unroll
        for (int j = 0; j < N; ++j) mma_chunk<Tag>(acc[j], a[NP == 2 && pr == 0 ? 1 : 0], b[j][NP == 2 && pr == 1 ? 1 : 0]);
}

template <int PREC, int NW, int MT> struct TX {
    typedef typename QT<PREC>::Tag Tag;
    static constexpr int EPC = Tag::EPC, NPART = QT<PREC>::NPART;
    static constexpr int ROWS = MT * 16, THREADS = NW * 64;
    static constexpr int KE = 8 * EPC;                    // K elements per 128-byte row of a stage tile (64 / 32)
    static constexpr int NKC = 512 / (4 * EPC);           // 16-byte k-chunks per row of K = 512 (16 / 32)
    static constexpr int TILE = ROWS * 128;               // one plane of one stage tile
    static constexpr int STG_TILES = 64 / KE;             // phase A stages 64 K elements at a time (1 / 2 tiles)
    static constexpr int STG_PLANE = STG_TILES * TILE, STG_BUF = NPART * STG_PLANE;
    static constexpr int STEP_H = NW * 16;                // hidden units per phase-B step (one n-tile per wave)
    static constexpr int NSTEP = 512 / STEP_H;
    static constexpr int RING_TILES = STEP_H / KE, RING_PLANE = RING_TILES * TILE, SLOT = NPART * RING_PLANE;
    static constexpr int CPSS = 64 / (4 * EPC);           // k-chunks per phase-A super-stage (2 / 4)
    static constexpr int CPST = STEP_H / (4 * EPC);       // k-chunks per phase-B step
    static constexpr int OLD = 260;                       // padded fp32 output row stride
    static constexpr int OUT_BYTES = ROWS * OLD * 4;
    static constexpr int R0 = 2 * STG_BUF > 2 * SLOT ? 2 * STG_BUF : 2 * SLOT;
    static constexpr int REGION = R0 > OUT_BYTES ? R0 : OUT_BYTES;
    static constexpr int LDS = REGION + NW * ROWS * 4;
    static constexpr int NTA = 32 / NW, NGA = NTA / 4;    // phase-A n-tiles per wave, groups of 4 per k-chunk
    static constexpr int NTB = 16 / NW, NGB = NTB / 2;    // phase-B n-tiles per wave, groups of 2 per k-chunk
    static constexpr int CPT = STG_TILES * ROWS * 8 / THREADS;   // staged 16-byte chunks per thread per super-stage
    static constexpr bool B2_DOUBLE = !(NPART == 2 && NW == 4);  // (4 waves, split): registers allow one ffn.3 group only
};

template <int PREC, int NW, int MT>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void tailx_kernel(TailArgs a) {
    typedef TX<PREC, NW, MT> C;
    typedef typename C::Tag Tag;
    constexpr int EPC = C::EPC, NPART = C::NPART, NV = EPC / 4, KE = C::KE, NKC = C::NKC, TILE = C::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + C::REGION);

    const TileLoc t = locate_tile(a.rs, blockIdx.x, C::ROWS);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
    auto stamp = [&](int slot) {
        if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + slot] = clock64();
    };
    stamp(0);
    auto wfrag = [&](const void* base, int p, long long plane_elems, int nt, int kc) -> u32x4 {
        const char* ptr = static_cast<const char*>(base) + (p ? plane_elems * (long long)sizeof(typename Tag::elem) : 0);
        return *reinterpret_cast<const u32x4*>(ptr + ((long long)(nt * NKC + kc) * 64 + lane) * 16);
    };

    // ------------------------------------------------------------------ phase A: h = [x ; ctx] Wcat^T
    f32x4 acc[MT][C::NTA];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NTA; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 stg[C::CPT][NV];
    auto load_stage = [&](int S) {
        const int k0 = S * 64;
        const float* src0 = (k0 < 256 ? a.X : a.CTX) + (long long)t.grow0 * 256 + (k0 & 255);
#pragma unroll
        for (int i = 0; i < C::CPT; ++i) {
            const int c = tid + C::THREADS * i;
            const int tl = c / (C::ROWS * 8), cc = c % (C::ROWS * 8), row = cc >> 3, slot = cc & 7;
            const float* p = src0 + (long long)row * 256 + tl * KE + slot * EPC;
#pragma unroll
            for (int j = 0; j < NV; ++j) stg[i][j] = *reinterpret_cast<const f32x4*>(p + 4 * j);
        }
    };
    auto store_stage = [&](int S) {
        char* buf = smem + (S & 1) * C::STG_BUF;
#pragma unroll
        for (int i = 0; i < C::CPT; ++i) {
            const int c = tid + C::THREADS * i;
            const int tl = c / (C::ROWS * 8), cc = c % (C::ROWS * 8), row = cc >> 3, slot = cc & 7;
            char* tile = buf + tl * TILE;
            const int off = lds_off<128>(row, slot);
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(tile + off) = stg[i][0];
            } else if constexpr (PREC == PREC_BF16X3) {
                u32x4 hi, lo;
                split8_bf16(stg[i][0], stg[i][1], hi, lo);
                *reinterpret_cast<u32x4*>(tile + off) = hi;
                *reinterpret_cast<u32x4*>(tile + C::STG_PLANE + off) = lo;
            } else {
                *reinterpret_cast<u32x4*>(tile + off) = pack8<Tag>(stg[i][0], stg[i][1]);
            }
        }
    };
    // weight groups: group ga of k-chunk kc = this wave's n-tiles j = 4*ga .. 4*ga+3 (global n-tile w + NW*j); two register
    // buffers: the next group is in flight while the current one is multiplied.  No branch around a prefetch (clamped
    // instead) and sched_barrier(0) around it — see lg_tail.hip.
    u32x4 bh[2][4][NPART];
    auto load_bh = [&](u32x4 (&dst)[4][NPART], int kc, int ga) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(a.Wcat, p, 512LL * 512, w + NW * (4 * ga + j), kc);
    };
    load_stage(0);
    load_bh(bh[0], 0, 0);
#pragma unroll 1
    for (int S = 0; S < 8; ++S) {
        store_stage(S);
        __syncthreads();
        load_stage(S + 1 < 8 ? S + 1 : S);
        __builtin_amdgcn_sched_barrier(0);
        const char* buf = smem + (S & 1) * C::STG_BUF;
#pragma unroll
        for (int i = 0; i < C::CPSS; ++i) {
            const int kc = S * C::CPSS + i;
            const char* tile = buf + (i >> 1) * TILE;
            auto afrag = [&](u32x4 (&af)[NPART], int mt) {   // (re)read per 16-row tile: 8 live registers instead of 8*MT
#pragma unroll
                for (int p = 0; p < NPART; ++p)
                    af[p] = *reinterpret_cast<const u32x4*>(tile + p * C::STG_PLANE + lds_off<128>(mt * 16 + lr, (i & 1) * 4 + g));
            };
#pragma unroll
            for (int ga = 0; ga < C::NGA; ++ga) {
                const int cur = C::NGA == 2 ? ga : (i & 1);                 // compile-time after unrolling (CPSS is even)
                const bool last = ga + 1 == C::NGA;
                load_bh(bh[cur ^ 1], last ? (kc + 1 < NKC ? kc + 1 : kc) : kc, last ? 0 : ga + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    u32x4 af[NPART];
                    afrag(af, mt);
                    q_mma_row<PREC, 4>(&acc[mt][4 * ga], af, bh[cur]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    stamp(1);
    // ------------------------------------------------------------------ bias + LayerNorm(512)
    {
        float part[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[mt][r] = 0.f;
#pragma unroll
        for (int j = 0; j < C::NTA; ++j) {
            const float bv = a.bcat[(w + NW * j) * 16 + lr];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[mt][j][r] += bv; part[mt][r] += acc[mt][j][r]; }
        }
        auto block_row_sum = [&](float (&p)[MT][4]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[mt][r] = row16_sum(p[mt][r]);
            __syncthreads();
            if (lr == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[w * C::ROWS + mt * 16 + g * 4 + r] = p[mt][r];
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 v = *reinterpret_cast<const f32x4*>(red + mt * 16 + g * 4);
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) v += *reinterpret_cast<const f32x4*>(red + ww * C::ROWS + mt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[mt][r] = v[r];
                __builtin_amdgcn_sched_barrier(0);   // bound the live range: MT*NW 16-byte reads hoisted together would spill
            }
        };
        block_row_sum(part);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mean = part[mt][r] * (1.f / 512.f);
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < C::NTA; ++j) { const float d = acc[mt][j][r] - mean; acc[mt][j][r] = d; sq += d * d; }
                part[mt][r] = sq;
            }
        block_row_sum(part);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[mt][r] = __builtin_amdgcn_rsqf(part[mt][r] * (1.f / 512.f) + 1e-5f);
#pragma unroll
        for (int j = 0; j < C::NTA; ++j) {   // normalise + affine now; GELU happens at the tile's step
            const int col = (w + NW * j) * 16 + lr;
            const float gm = a.gamma[col], bt = a.beta[col];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][j][r] = acc[mt][j][r] * part[mt][r] * gm + bt;
        }
    }
    stamp(2);
    // ------------------------------------------------------------------ phase B: NSTEP steps of STEP_H hidden units
    // step j: this wave's n-tile j = hidden units [(w + NW*j)*16, +16) = columns [16w, 16w+16) of the step
    auto gelu_store = [&](int j) {
        char* slot = smem + (j & 1) * C::SLOT;
        const int hc = w * 16;                         // first column of this wave inside the step
        char* tile0 = slot + (hc / KE) * TILE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x2 v01 = gelu4_fast2(f32x2{acc[mt][j][0], acc[mt][j][1]});
            const f32x2 v23 = gelu4_fast2(f32x2{acc[mt][j][2], acc[mt][j][3]});
            const float gv[4] = {v01[0], v01[1], v23[0], v23[1]};
            if constexpr (EPC == 8) {
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {   // even lanes write row rp, odd lanes row rp+1, as (even col, odd col) pairs
                    const bool odd = lr & 1;
                    const float mine = odd ? gv[rp + 1] : gv[rp];
                    const float give = odd ? gv[rp] : gv[rp + 1];
                    const float got = dpp_xor1(give);
                    const float c0 = odd ? got : mine, c1 = odd ? mine : got;
                    const int row = mt * 16 + g * 4 + rp + (odd ? 1 : 0);
                    const int col = (hc % KE) + (lr & ~1);
                    const int off = lds_off<128>(row, col >> 3) + (col & 7) * 2;
                    if constexpr (PREC == PREC_BF16X3) {
                        const float h0 = bf16_round(c0), h1 = bf16_round(c1);
                        *reinterpret_cast<uint32_t*>(tile0 + off) = pack2_bf16(h0, h1);
                        *reinterpret_cast<uint32_t*>(tile0 + C::RING_PLANE + off) = pack2_bf16(c0 - h0, c1 - h1);
                    } else {
                        *reinterpret_cast<uint32_t*>(tile0 + off) = pack2<Tag>(c0, c1);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = (hc % KE) + lr, row = mt * 16 + g * 4 + r;
                    *reinterpret_cast<float*>(tile0 + lds_off<128>(row, c >> 2) + (c & 3) * 4) = gv[r];
                }
            }
        }
    };
    f32x4 acc2[MT][C::NTB];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NTB; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 b2h[C::B2_DOUBLE ? 2 : 1][2][NPART];   // groups of 2 out n-tiles: global n-tile w + NW*(2*gb + {0,1})
    auto load_b2h = [&](u32x4 (&dst)[2][NPART], int kc, int gb) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(a.W2, p, 256LL * 512, w + NW * (2 * gb + j), kc);
    };
    if constexpr (C::B2_DOUBLE) load_b2h(b2h[0], 0, 0);
    // the barriers of block_row_sum ordered every wave past phase A: the staging buffers are dead, the ring may reuse them
    gelu_store(0);
    __syncthreads();
    stamp(3);
#pragma unroll
    for (int j = 0; j < C::NSTEP; ++j) {
        const char* slot = smem + (j & 1) * C::SLOT;
#pragma unroll
        for (int i = 0; i < C::CPST; ++i) {
            const int kc = j * C::CPST + i;
            const char* tile = slot + (i >> 1) * TILE;
            auto afrag = [&](u32x4 (&af)[NPART], int mt) {
#pragma unroll
                for (int p = 0; p < NPART; ++p)
                    af[p] = *reinterpret_cast<const u32x4*>(tile + p * C::RING_PLANE + lds_off<128>(mt * 16 + lr, (i & 1) * 4 + g));
            };
#pragma unroll
            for (int gb = 0; gb < C::NGB; ++gb) {
                int cur = 0;
                if constexpr (C::B2_DOUBLE) {
                    cur = C::NGB == 2 ? gb : (i & 1);                       // compile-time after unrolling (CPST is even)
                    const bool last = gb + 1 == C::NGB;
                    load_b2h(b2h[cur ^ 1], last ? (kc + 1 < NKC ? kc + 1 : kc) : kc, last ? 0 : gb + 1);
                } else {
                    load_b2h(b2h[0], kc, gb);   // no register prefetch: the co-resident workgroup covers the L2 latency
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    u32x4 af[NPART];
                    afrag(af, mt);
                    q_mma_row<PREC, 2>(&acc2[mt][2 * gb], af, b2h[cur]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (j + 1 < C::NSTEP) {
            gelu_store(j + 1);     // other ring slot: every wave left it at the previous barrier
            __syncthreads();
        }
    }
    stamp(4);
    // ------------------------------------------------------------------ epilogue: + b2, + x, full-row stores
    const int qlen = a.rs.len[t.seg];
    constexpr int XPT = C::ROWS * 64 / C::THREADS;   // float4 per thread (16)
    f32x4 xres[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int c = tid + C::THREADS * i, row = c >> 6, c4 = c & 63;
        xres[i] = *reinterpret_cast<const f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + c4 * 4);
    }
    __syncthreads();   // ring is dead; reuse the region as a [ROWS][260] fp32 tile
    {
        float* ot = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int n = 0; n < C::NTB; ++n) {
            const int col = (w + NW * n) * 16 + lr;
            const float b2 = a.b2[col];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[(mt * 16 + g * 4 + r) * C::OLD + col] = acc2[mt][n][r] + b2;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int c = tid + C::THREADS * i, row = c >> 6, c4 = c & 63;
            if (t.r0 + row < qlen) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(ot + row * C::OLD + c4 * 4);
                *reinterpret_cast<f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + c4 * 4) = xres[i] + d;
            }
        }
    }
    stamp(5);
}

template <int PREC, int NW, int MT> static hipError_t launch_tailx_t(const TailArgs& a, hipStream_t s) {
    typedef TX<PREC, NW, MT> C;
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    auto kern = tailx_kernel<PREC, NW, MT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / C::ROWS), dim3(C::THREADS), C::LDS, s, a);
    return hipGetLastError();
}
template <int NW, int MT> static hipError_t launch_tailx_p(int prec, const TailArgs& a, hipStream_t s) {
    switch (prec) {
        case PREC_F32: return launch_tailx_t<PREC_F32, NW, MT>(a, s);
        case PREC_BF16: return launch_tailx_t<PREC_BF16, NW, MT>(a, s);
        case PREC_F16: return launch_tailx_t<PREC_F16, NW, MT>(a, s);
        case PREC_BF16X3: return launch_tailx_t<PREC_BF16X3, NW, MT>(a, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_tail4(int prec, const TailArgs& a, hipStream_t s) { return launch_tailx_p<4, 4>(prec, a, s); }
hipError_t launch_tail128(int prec, const TailArgs& a, hipStream_t s) { return launch_tailx_p<8, 8>(prec, a, s); }
// 32 rows per workgroup, 4 waves: twice the workgroups of the default — for grids that do not fill the chip (small batches)
hipError_t launch_tail32(int prec, const TailArgs& a, hipStream_t s) { return launch_tailx_p<4, 2>(prec, a, s); }

}  // namespace lg
