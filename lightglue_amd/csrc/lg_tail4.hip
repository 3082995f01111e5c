// lightglue_amd — fused block tail, 4-wave variant:  x <- x + ffn(cat[x, out_proj(ctx)])   (see lg_tail.hip for the
// algebra, reference lines and the weight packing; this file only changes the workgroup decomposition).
//
// lg_tail.hip runs ONE 8-wave workgroup per CU (133 KB of LDS): its MFMA phases and its VALU phases (LayerNorm,
// GELU, split + transposition of g) alternate in lock-step, so the matrix pipe idles ~60 % of the time.
// Here a workgroup is 4 waves (one per SIMD) x 64 rows and needs < 70 KB of LDS, so TWO workgroups share a CU and
// drift apart: while one is in a VALU phase the other one's MFMAs own the matrix pipe.
//   * phase A streams the [x ; ctx] tile through a double-buffered 2 x 16 KB LDS stage (one barrier per 64-wide
//     K stage) instead of keeping it resident;
//   * wave w owns the hidden n-tiles {w + 4j, j < 8} (128 accumulator registers); weight fragments are fetched in
//     half-chunks (4 n-tiles) on a two-deep register ring so that one half is in flight while the other is multiplied;
//   * phase B runs in 8 steps of 64 hidden units: GELU + split + transposition of ONE n-tile per wave into a
//     two-slot LDS ring, barrier, 2 k-chunks of MFMAs.
#include "lg_kernels.h"

namespace lg {

constexpr int QBM = 64, QTHREADS = 256;

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

template <int PREC>
__device__ __forceinline__ void q_mma(f32x4& acc, const u32x4* a, const u32x4* b) {
    typedef typename QT<PREC>::Tag Tag;
    if constexpr (QT<PREC>::NPART == 2) {
        mma_chunk<Tag>(acc, a[1], b[0]);
        mma_chunk<Tag>(acc, a[0], b[1]);
    }
    mma_chunk<Tag>(acc, a[0], b[0]);
}
// product-major over N accumulators of one 16-row tile: no back-to-back dependent MFMAs
template <int PREC, int N, int STRIDE>
__device__ __forceinline__ void q_mma_row(f32x4* acc, const u32x4* a, const u32x4 (*b)[QT<PREC>::NPART]) {
    typedef typename QT<PREC>::Tag Tag;
    constexpr int NP = QT<PREC>::NPART;
#pragma unroll
    for (int pr = 0; pr < (NP == 2 ? 3 : 1); ++pr)
#pragma unroll
        for (int j = 0; j < N; ++j) mma_chunk<Tag>(acc[j * STRIDE], a[NP == 2 && pr == 0 ? 1 : 0], b[j][NP == 2 && pr == 1 ? 1 : 0]);
}

template <int PREC>
__global__ __launch_bounds__(QTHREADS, 2) void tail4_kernel(TailArgs a) {
    typedef typename QT<PREC>::Tag Tag;
    constexpr int EPC = Tag::EPC, NPART = QT<PREC>::NPART;
    constexpr int KE = 8 * EPC;               // K elements per 128-byte stage row: 64 (16-bit) / 32 (f32)
    constexpr int STAGES = 512 / KE;          // 8 / 16
    constexpr int NKC = 2 * STAGES;           // 16-byte k-chunks per row: 16 / 32
    constexpr int NV = EPC / 4;
    constexpr int TILE = QBM * 128;           // one plane of one stage tile (8 KB)
    constexpr int STEP_TILES = 64 / KE;       // stage tiles per 64 hidden units: 1 / 2
    constexpr int PLANE = STEP_TILES * TILE;  // one plane of one ring slot / staging buffer
    constexpr int SLOT = NPART * PLANE;       // ring slot == staging buffer size (phase A stages 64 K elements at once for f32 too)
    constexpr int CPS = 64 / (4 * EPC);       // k-chunks per 64 hidden units: 2 / 4
    constexpr int OLD = 260;                  // padded fp32 output row stride
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + (2 * SLOT > QBM * OLD * 4 ? 2 * SLOT : QBM * OLD * 4));

    const TileLoc t = locate_tile(a.rs, blockIdx.x, QBM);
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
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: "super-stage" S = 64 K elements = STEP_TILES stage tiles; thread -> 2 x STEP_TILES chunks
    f32x4 stg[2 * STEP_TILES][NV];
    auto load_stage = [&](int S) {
        const int k0 = S * 64;
        const float* src0 = (k0 < 256 ? a.X : a.CTX) + (long long)t.grow0 * 256 + (k0 & 255);
#pragma unroll
        for (int i = 0; i < 2 * STEP_TILES; ++i) {
            const int c = tid + QTHREADS * i;                    // chunk id within the super-stage
            const int tl = c / 512, cc = c % 512, row = cc >> 3, slot = cc & 7;
            const float* p = src0 + (long long)row * 256 + tl * KE + slot * EPC;
#pragma unroll
            for (int j = 0; j < NV; ++j) stg[i][j] = *reinterpret_cast<const f32x4*>(p + 4 * j);
        }
    };
    auto store_stage = [&](int S) {
        char* buf = smem + (S & 1) * SLOT;
#pragma unroll
        for (int i = 0; i < 2 * STEP_TILES; ++i) {
            const int c = tid + QTHREADS * i;
            const int tl = c / 512, cc = c % 512, row = cc >> 3, slot = cc & 7;
            char* tile = buf + tl * TILE;
            const int off = lds_off<128>(row, slot);
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(tile + off) = stg[i][0];
            } else if constexpr (PREC == PREC_BF16X3) {
                u32x4 hi, lo;
                split8_bf16(stg[i][0], stg[i][1], hi, lo);
                *reinterpret_cast<u32x4*>(tile + off) = hi;
                *reinterpret_cast<u32x4*>(tile + PLANE + off) = lo;
            } else {
                *reinterpret_cast<u32x4*>(tile + off) = pack8<Tag>(stg[i][0], stg[i][1]);
            }
        }
    };
    // weight half-chunks: half hb of k-chunk kc = n-tiles j = 4*hb .. 4*hb+3 of this wave (global n-tile w + 4j)
    u32x4 bh[2][4][NPART];
    auto load_bh = [&](u32x4 (&dst)[4][NPART], int kc, int hb) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(a.Wcat, p, 512LL * 512, w + 4 * (4 * hb + j), kc);
    };
    load_stage(0);
    load_bh(bh[0], 0, 0);
    constexpr int NSS = 512 / 64;   // super-stages
#pragma unroll 1
    for (int S = 0; S < NSS; ++S) {
        store_stage(S);
        __syncthreads();
        load_stage(S + 1 < NSS ? S + 1 : S);   // clamped, never branched (hipcc's vmcnt counting)
        __builtin_amdgcn_sched_barrier(0);
        const char* buf = smem + (S & 1) * SLOT;
#pragma unroll
        for (int i = 0; i < CPS; ++i) {
            const int kc = S * CPS + i;
            const char* tile = buf + (i >> 1) * TILE;
            // A fragments are (re)read per 16-row tile right before use: 8 live registers instead of 32 — the
            // accumulators (128) + the weight ring (64) leave no room for more under the 256-register cap
            auto afrag = [&](u32x4 (&af)[NPART], int mt) {
#pragma unroll
                for (int p = 0; p < NPART; ++p)
                    af[p] = *reinterpret_cast<const u32x4*>(tile + p * PLANE + lds_off<128>(mt * 16 + lr, (i & 1) * 4 + g));
            };
            load_bh(bh[1], kc, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                u32x4 af[NPART];
                afrag(af, mt);
#pragma unroll
                for (int j = 0; j < 1; ++j) q_mma_row<PREC, 4, 1>(&acc[mt][0], af, bh[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_bh(bh[0], kc + 1 < NKC ? kc + 1 : kc, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                u32x4 af[NPART];
                afrag(af, mt);
#pragma unroll
                for (int j = 0; j < 1; ++j) q_mma_row<PREC, 4, 1>(&acc[mt][4], af, bh[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp(1);
    // ------------------------------------------------------------------ bias + LayerNorm(512) statistics
    {
        float part[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[mt][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float bv = a.bcat[(w + 4 * j) * 16 + lr];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[mt][j][r] += bv; part[mt][r] += acc[mt][j][r]; }
        }
        auto block_row_sum = [&](float (&p)[4][4]) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[mt][r] = row16_sum(p[mt][r]);
            __syncthreads();
            if (lr == 0) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[w * QBM + mt * 16 + g * 4 + r] = p[mt][r];
            }
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                f32x4 v = *reinterpret_cast<const f32x4*>(red + mt * 16 + g * 4);
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) v += *reinterpret_cast<const f32x4*>(red + ww * QBM + mt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[mt][r] = v[r];
            }
        };
        block_row_sum(part);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mean = part[mt][r] * (1.f / 512.f);
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = acc[mt][j][r] - mean; acc[mt][j][r] = d; sq += d * d; }
                part[mt][r] = sq;
            }
        block_row_sum(part);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[mt][r] = __builtin_amdgcn_rsqf(part[mt][r] * (1.f / 512.f) + 1e-5f);   // rstd
        // normalise + affine now (pre-GELU values stay in the accumulators until their step)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = (w + 4 * j) * 16 + lr;
            const float gm = a.gamma[col], bt = a.beta[col];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][j][r] = acc[mt][j][r] * part[mt][r] * gm + bt;
        }
    }
    stamp(2);
    // ------------------------------------------------------------------ phase B: 8 steps of 64 hidden units
    // step j: this wave's n-tile j = hidden units [(w+4j)*16, +16) = columns [16w, 16w+16) of the step's 64
    auto gelu_store = [&](int j) {
        char* slot = smem + (j & 1) * SLOT;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x2 v01 = gelu4_fast2(f32x2{acc[mt][j][0], acc[mt][j][1]});
            const f32x2 v23 = gelu4_fast2(f32x2{acc[mt][j][2], acc[mt][j][3]});
            const float gv[4] = {v01[0], v01[1], v23[0], v23[1]};
            if constexpr (EPC == 8) {
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
                    const bool odd = lr & 1;
                    const float mine = odd ? gv[rp + 1] : gv[rp];
                    const float give = odd ? gv[rp] : gv[rp + 1];
                    const float got = dpp_xor1(give);
                    const float c0 = odd ? got : mine, c1 = odd ? mine : got;
                    const int row = mt * 16 + g * 4 + rp + (odd ? 1 : 0);
                    const int col = w * 16 + (lr & ~1);
                    const int off = lds_off<128>(row, col >> 3) + (col & 7) * 2;
                    if constexpr (PREC == PREC_BF16X3) {
                        const float h0 = bf16_round(c0), h1 = bf16_round(c1);
                        *reinterpret_cast<uint32_t*>(slot + off) = pack2_bf16(h0, h1);
                        *reinterpret_cast<uint32_t*>(slot + PLANE + off) = pack2_bf16(c0 - h0, c1 - h1);
                    } else {
                        *reinterpret_cast<uint32_t*>(slot + off) = pack2<Tag>(c0, c1);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = w * 16 + lr, row = mt * 16 + g * 4 + r;     // 0..63 within the step; f32 tile = 32 columns
                    *reinterpret_cast<float*>(slot + (c >> 5) * TILE + lds_off<128>(row, (c & 31) >> 2) + (c & 3) * 4) = gv[r];
                }
            }
        }
    };
    f32x4 acc2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 b2h[NPART == 2 ? 1 : 2][2][NPART];   // half-chunks of ffn.3 fragments: out n-tiles w + 4*(2*hb + {0,1})
    auto load_b2h = [&](u32x4 (&dst)[2][NPART], int kc, int hb) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(a.W2, p, 256LL * 512, w + 4 * (2 * hb + j), kc);
    };
    load_b2h(b2h[0], 0, 0);
    // the barriers of block_row_sum ordered every wave past phase A: staging buffers are dead, the ring may reuse them
    gelu_store(0);
    __syncthreads();
    stamp(3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const char* slot = smem + (j & 1) * SLOT;
#pragma unroll
        for (int i = 0; i < CPS; ++i) {
            const int kc = j * CPS + i;
            const char* tile = slot + (i >> 1) * TILE;
            auto afrag = [&](u32x4 (&af)[NPART], int mt) {
#pragma unroll
                for (int p = 0; p < NPART; ++p)
                    af[p] = *reinterpret_cast<const u32x4*>(tile + p * PLANE + lds_off<128>(mt * 16 + lr, (i & 1) * 4 + g));
            };
            if constexpr (NPART == 2) {
                // split bf16: accumulators of both GEMMs (<= 112 + 64) leave room for ONE half-chunk of weights: no
                // register prefetch here — the co-resident workgroup's MFMAs cover the L2 latency
                if (i > 0 || j > 0) load_b2h(b2h[0], kc, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    u32x4 af[NPART];
                    afrag(af, mt);
#pragma unroll
                    for (int n = 0; n < 1; ++n) q_mma_row<PREC, 2, 1>(&acc2[mt][0], af, b2h[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
                load_b2h(b2h[0], kc, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    u32x4 af[NPART];
                    afrag(af, mt);
#pragma unroll
                    for (int n = 0; n < 2; ++n) q_mma<PREC>(acc2[mt][2 + n], af, b2h[0][n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                load_b2h(b2h[1], kc, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    u32x4 af[NPART];
                    afrag(af, mt);
#pragma unroll
                    for (int n = 0; n < 1; ++n) q_mma_row<PREC, 2, 1>(&acc2[mt][0], af, b2h[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
                load_b2h(b2h[0], kc + 1 < NKC ? kc + 1 : kc, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    u32x4 af[NPART];
                    afrag(af, mt);
#pragma unroll
                    for (int n = 0; n < 1; ++n) q_mma_row<PREC, 2, 1>(&acc2[mt][2], af, b2h[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (j < 7) {
            gelu_store(j + 1);     // other ring slot: every wave left it at the previous barrier
            __syncthreads();
        }
    }
    stamp(4);
    // ------------------------------------------------------------------ epilogue: + b2, + x, full-row stores
    const int qlen = a.rs.len[t.seg];
    f32x4 xres[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = tid + QTHREADS * i, row = c >> 6, c4 = c & 63;
        xres[i] = *reinterpret_cast<const f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + c4 * 4);
    }
    __syncthreads();   // ring is dead; reuse the region as a [64][260] fp32 tile
    {
        float* ot = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = (w + 4 * n) * 16 + lr;
            const float b2 = a.b2[col];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[(mt * 16 + g * 4 + r) * OLD + col] = acc2[mt][n][r] + b2;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = tid + QTHREADS * i, row = c >> 6, c4 = c & 63;
            if (t.r0 + row < qlen) {
                const f32x4 d = *reinterpret_cast<const f32x4*>(ot + row * OLD + c4 * 4);
                *reinterpret_cast<f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + c4 * 4) = xres[i] + d;
            }
        }
    }
    stamp(5);
}

template <int PREC> static hipError_t launch_tail4_prec(const TailArgs& a, hipStream_t s) {
    typedef typename QT<PREC>::Tag Tag;
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    constexpr int SLOT = QT<PREC>::NPART * (64 / (8 * Tag::EPC)) * QBM * 128;
    constexpr int region = 2 * SLOT > QBM * 260 * 4 ? 2 * SLOT : QBM * 260 * 4;
    constexpr int smem = region + 4 * QBM * 4;
    auto kern = tail4_kernel<PREC>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / QBM), dim3(QTHREADS), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_tail4(int prec, const TailArgs& a, hipStream_t s) {
    switch (prec) {
        case PREC_F32: return launch_tail4_prec<PREC_F32>(a, s);
        case PREC_BF16: return launch_tail4_prec<PREC_BF16>(a, s);
        case PREC_F16: return launch_tail4_prec<PREC_F16>(a, s);
        case PREC_BF16X3: return launch_tail4_prec<PREC_BF16X3>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace lg
