// lightglue_amd — the compute part of the attention input projections, shared by the standalone projection kernel
// (lg_proj.hip) and by the fused tail (lg_tail.hip), which runs the NEXT block's projection on the x tile it has
// just produced instead of writing it out and launching a second kernel.
// Precondition: the 64 x 256 activation tile sits in LDS at smA in operand precision ([NPART planes][STAGES][64][128 B],
// XOR-swizzled with lds_off<128>) and NO barrier has been executed since those writes (proj_compute issues its weight
// prefetch first and then synchronises).
#pragma once
#include "lg_kernels.h"

namespace lg {

constexpr int PBM = 64, PTHREADS = 512;
#ifndef LG_PROJ_PINGPONG
#define LG_PROJ_PINGPONG 0       // scheduling experiment (same arithmetic): waves 4..7 run both MFMA loops of the projection before both epilogues
#endif
#ifndef LG_PROJ_ABLATE_W
#define LG_PROJ_ABLATE_W 0       // timing ablation (wrong results): constant weight fragments in the projection's MFMA loop
#endif
#ifndef LG_PROJ_EPI_PREFETCH
#define LG_PROJ_EPI_PREFETCH 0   // scheduling experiment (same arithmetic): 1 = issue the epilogue's bias / rotary loads before the MFMA loop (+60 live VGPRs, no exposed L2 round trip at the head of the two epilogues)
#endif
#ifndef LG_ATTN_FOLD
#define LG_ATTN_FOLD 0   // experiment, see lg_attention.hip: q and k leave the projection pre-multiplied by the square root of the score scale
#endif

// Operand scheme of the projection.  NPART = weight planes, APART = activation planes.
// PREC_QKV_F16W2 is what the default precision ("bf16x3") uses for the q/k/v projections: the activation tile as ONE f16
// plane, the weights as split f16 (hi + lo), i.e. TWO MFMAs per product instead of the three of split-bf16.  q/k/v leave
// this kernel rounded to f16 for the attention anyway, so rounding x to f16 first costs nothing measurable (operand-rounding study,
// max |dscore| 2.3-3.0e-4 vs 2.2-2.5e-4 for split-bf16; a single f16 product would give 1.0-1.6e-3).
constexpr int PREC_QKV_F16W2 = 100;
template <int PREC> struct PJ;
template <> struct PJ<PREC_F32> { typedef TagF32 Tag; static constexpr int NPART = 1, APART = 1; };
template <> struct PJ<PREC_BF16> { typedef TagBF16 Tag; static constexpr int NPART = 1, APART = 1; };
template <> struct PJ<PREC_F16> { typedef TagF16 Tag; static constexpr int NPART = 1, APART = 1; };
template <> struct PJ<PREC_BF16X3> { typedef TagBF16 Tag; static constexpr int NPART = 2, APART = 2; };
template <> struct PJ<PREC_QKV_F16W2> { typedef TagF16 Tag; static constexpr int NPART = 2, APART = 1; };

// acc += product of one weight fragment set (NPART planes) and one activation fragment set (APART planes).  TRANSPOSED: the
// weights are the A operand.  split x split: hi*lo + lo*hi + hi*hi; split weights x single activation: lo*x + hi*x.
template <int PREC, bool TRANSPOSED>
__device__ __forceinline__ void pj_mma(f32x4& acc, const u32x4* wf, const u32x4* xf) {
    typedef typename PJ<PREC>::Tag Tag;
    auto mm = [&](const u32x4& wv, const u32x4& xv) { if constexpr (TRANSPOSED) mma_chunk<Tag>(acc, wv, xv); else mma_chunk<Tag>(acc, xv, wv); };
    if constexpr (PJ<PREC>::NPART == 2 && PJ<PREC>::APART == 2) { mm(wf[0], xf[1]); mm(wf[1], xf[0]); mm(wf[0], xf[0]); }
    else if constexpr (PJ<PREC>::NPART == 2) { mm(wf[1], xf[0]); mm(wf[0], xf[0]); }
    else mm(wf[0], xf[0]);
}

template <class T> __device__ __forceinline__ T pj_cvt(float x);
template <> __device__ __forceinline__ float pj_cvt<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t pj_cvt<bf16_t>(float x) { return (bf16_t)x; }
template <> __device__ __forceinline__ f16_t pj_cvt<f16_t>(float x) { return (f16_t)x; }

template <int PREC> struct PJL {   // LDS geometry of the activation tile
    typedef typename PJ<PREC>::Tag Tag;
    static constexpr int KE = 8 * Tag::EPC;            // K elements per 128-byte stage row (64 or 32)
    static constexpr int STAGES = 256 / KE;            // 4 (16-bit) or 8 (f32)
    static constexpr int TILE = PBM * 128;             // one plane of one stage
    static constexpr int A_PLANE = STAGES * TILE;      // 32 KB (16-bit) / 64 KB (f32)
    static constexpr int A_BYTES = PJ<PREC>::APART * A_PLANE;
};
// One pass of the projection: NTP n-tiles per wave (wave w owns the n-tiles w + 8*jj, jj = PASS*NTP + j, i.e. the 16 output
// columns w*16 + 128*jj ...).  The column group of a tile (q / k / v, or qk / v) is jj >> 1 — a compile-time constant — and decides
// the FORM of the tile's MFMAs:
//   q / k tiles: TRANSPOSED, C^T = W x^T (weights as the A operand): lane (lr, g) ends with keypoint row lr of the 16-row tile
//     and the 4 consecutive head channels 4g..4g+3 -> bias as one float4, the rotary pair (2j, 2j+1) sits in one lane (no
//     cross-lane traffic, ref :58-65), and the result leaves as ONE 8-byte (16-bit q/k) store per 16-row tile: lanes g = 0..3
//     cover 32 contiguous bytes of a [row][64] line.
//   v tiles: plain C = x W^T: lane (lr, g) holds channel lr of rows 4g..4g+3 -> one 8-byte store into the transposed
//     [head][64][R] layout (4 consecutive rows of one channel).
// Nothing is staged through LDS and no barrier follows the MFMA loop (the staged version spent 40 % of the kernel in its two
// epilogues: LDS write, barrier, LDS read, store, barrier).
#if LG_PROJ_PINGPONG   // experiment: PART 0 = the whole pass (as the product), 1 = its MFMA loop only, 2 = its epilogue only; accumulators owned by the caller
template <int PREC, class TA, int NTP, int NPASS, int PASS, int A_PLANE, int MT, int PART>
__device__ __forceinline__ void proj_pass(const ProjArgs& a, const TileLoc& t, const char* smA, u32x4 (&bf)[PJ<PREC>::NPART == 2 ? 2 : 4][NTP][PJ<PREC>::NPART],
                                          int stamp_base, f32x4 (&acc)[MT][NTP]) {
#else
template <int PREC, class TA, int NTP, int NPASS, int PASS, int A_PLANE, int MT>
__device__ __forceinline__ void proj_pass(const ProjArgs& a, const TileLoc& t, const char* smA, u32x4 (&bf)[PJ<PREC>::NPART == 2 ? 2 : 4][NTP][PJ<PREC>::NPART],
                                          int stamp_base) {
#endif
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int NPART = PJ<PREC>::NPART, APART = PJ<PREC>::APART;
    constexpr int STAGES = PJL<PREC>::STAGES, NKC = 2 * STAGES, TILE = MT * 16 * 128;   // one plane of one K stage: MT*16 rows x 128 B
    constexpr int NBUF = NPART == 2 ? 2 : 4;
    constexpr int N_QK = NTP == 3 ? 2 : 1;             // self: q, k, v groups of 256 columns; cross: qk, v
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const long long R = a.R;
    auto stamp = [&](int slot) {   // profiling tap (a.dbg == nullptr in production)
        if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + stamp_base + slot] = clock64();
    };
    auto wfrag = [&](int p, int nt, int kc) -> u32x4 {
#if LG_PROJ_ABLATE_W   // TIMING ABLATION ONLY (wrong results): no weight loads at all — is the loop bound by the L2 weight stream?
        return u32x4{0x3c003c00u + (unsigned)lane, 0x3c003c00u + (unsigned)(nt + kc), 0x3c003c00u + (unsigned)p, 0x3c003c00u};
#endif
        const char* ptr = static_cast<const char*>(a.W) + (p ? (long long)a.Nout * 256 * (long long)sizeof(typename Tag::elem) : 0);
        return *reinterpret_cast<const u32x4*>(ptr + ((long long)(nt * NKC + kc) * 64 + lane) * 16);
    };
    auto load_b = [&](u32x4 (&dst)[NTP][NPART], int pass, int kc) {
#pragma unroll
        for (int j = 0; j < NTP; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(p, w + 8 * (pass * NTP + j), kc);
    };
#if LG_PROJ_PINGPONG
    if constexpr (PART != 2) {
#else
    f32x4 acc[MT][NTP];
#endif
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if LG_PROJ_EPI_PREFETCH   // experiment: the epilogue's operands (bias, rotary tables) are fetched BEFORE the MFMA loop instead of after it
    typedef TA ta4 __attribute__((ext_vector_type(4)));
    f32x4 b4[NTP]; float bv[NTP];
    f32x2 c2[NTP][MT], s2[NTP][MT];
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int jj = PASS * NTP + j;
        const int col0 = (w + 8 * jj) * 16, d0 = col0 & 63;
        if ((jj >> 1) < N_QK) {
            b4[j] = *reinterpret_cast<const f32x4*>(a.bias + col0 + 4 * g);
            if constexpr (NTP == 3) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const long long row = t.grow0 + mt * 16 + lr;
                    c2[j][mt] = *reinterpret_cast<const f32x2*>(a.cosb + row * 32 + (d0 >> 1) + 2 * g);
                    s2[j][mt] = *reinterpret_cast<const f32x2*>(a.sinb + row * 32 + (d0 >> 1) + 2 * g);
                }
            }
        } else {
            bv[j] = a.bias[col0 + lr];
        }
    }
#endif
#pragma unroll 1
    for (int c0 = 0; c0 < NKC; c0 += NBUF) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) {
            const int kc = c0 + i;
            // prefetch NBUF-1 chunks ahead; past the end of a pass, start on the next pass's first chunks
            const int nk = kc + NBUF - 1;
            const int npass = nk < NKC ? PASS : (PASS + 1 < NPASS ? PASS + 1 : PASS);
            load_b(bf[(i + NBUF - 1) % NBUF], npass, nk < NKC ? nk : nk - NKC);
            __builtin_amdgcn_sched_barrier(0);
            const char* tile = smA + (kc >> 1) * TILE;
#pragma unroll
            for (int mh = 0; mh < MT; mh += 4) {   // activation fragments of (up to) 4 row tiles at a time (bounds the live registers at MT = 8)
            constexpr int MG = MT < 4 ? MT : 4;
            u32x4 afh[MG][APART];
#pragma unroll
            for (int mt = 0; mt < MG; ++mt)
#pragma unroll
                for (int p = 0; p < APART; ++p)
                    afh[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * A_PLANE + lds_off<128>((mh + mt) * 16 + lr, (kc & 1) * 4 + g));
#pragma unroll
            for (int mtl = 0; mtl < MG; ++mtl)
#pragma unroll
                for (int j = 0; j < NTP; ++j) {
                    const bool is_v = ((PASS * NTP + j) >> 1) >= N_QK;     // compile-time after unrolling
                    if (is_v) pj_mma<PREC, false>(acc[mh + mtl][j], bf[i][j], afh[mtl]);
                    else pj_mma<PREC, true>(acc[mh + mtl][j], bf[i][j], afh[mtl]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp(2 + 2 * PASS);
#if LG_PROJ_PINGPONG
    }
    if constexpr (PART == 1) return;
#endif
    // ---- epilogue of the pass, straight from the accumulators.  All loads (bias, rotary tables) are issued BEFORE the first
    // store: the compiler cannot prove that q/k/v do not alias the tables, so a load placed after a store stays there and
    // every (tile, 16-row tile) iteration would expose one full L2 round trip (measured: 14k cycles for 12 iterations).
#if !LG_PROJ_EPI_PREFETCH
    typedef TA ta4 __attribute__((ext_vector_type(4)));
    f32x4 b4[NTP]; float bv[NTP];
    f32x2 c2[NTP][MT], s2[NTP][MT];
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int jj = PASS * NTP + j;
        const int col0 = (w + 8 * jj) * 16, d0 = col0 & 63;
        if ((jj >> 1) < N_QK) {
            b4[j] = *reinterpret_cast<const f32x4*>(a.bias + col0 + 4 * g);
            if constexpr (NTP == 3) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const long long row = t.grow0 + mt * 16 + lr;
                    c2[j][mt] = *reinterpret_cast<const f32x2*>(a.cosb + row * 32 + (d0 >> 1) + 2 * g);
                    s2[j][mt] = *reinterpret_cast<const f32x2*>(a.sinb + row * 32 + (d0 >> 1) + 2 * g);
                }
            }
        } else {
            bv[j] = a.bias[col0 + lr];
        }
    }
#endif
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int jj = PASS * NTP + j;
        const int col0 = (w + 8 * jj) * 16;               // first output column of the tile
        const int group = jj >> 1, head = (col0 >> 6) & 3, d0 = col0 & 63;
        if (group < N_QK) {                               // q / k (or qk): transposed tile
            TA* base = static_cast<TA*>(group == 0 ? a.q : a.k);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const long long row = t.grow0 + mt * 16 + lr;
                f32x4 v = acc[mt][j] + b4[j];
                if constexpr (NTP == 3) {                 // SelfBlock: rotary, pairs (2f, 2f+1) with frequency f (ref :58-65)
                    const f32x2 c = c2[j][mt], sn = s2[j][mt];
                    const f32x4 u = v;
                    v[0] = u[0] * c[0] - u[1] * sn[0]; v[1] = u[1] * c[0] + u[0] * sn[0];
                    v[2] = u[2] * c[1] - u[3] * sn[1]; v[3] = u[3] * c[1] + u[2] * sn[1];
                }
#if LG_ATTN_FOLD
                v *= 0.42466090014400953f;                // sqrt(log2(e) / sqrt(64)): the attention's score scale, split over q and k (lg_attention.hip LG_ATTN_FOLD)
#endif
                ta4 o = {pj_cvt<TA>(v[0]), pj_cvt<TA>(v[1]), pj_cvt<TA>(v[2]), pj_cvt<TA>(v[3])};
                *reinterpret_cast<ta4*>(base + ((long long)head * R + row) * 64 + d0 + 4 * g) = o;
            }
        } else {                                          // v: plain tile -> transposed layout [head][64][R]
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ta4 o = {pj_cvt<TA>(acc[mt][j][0] + bv[j]), pj_cvt<TA>(acc[mt][j][1] + bv[j]), pj_cvt<TA>(acc[mt][j][2] + bv[j]), pj_cvt<TA>(acc[mt][j][3] + bv[j])};
                *reinterpret_cast<ta4*>(static_cast<TA*>(a.vt) + ((long long)head * 64 + d0 + lr) * R + t.grow0 + mt * 16 + 4 * g) = o;
            }
        }
    }
    stamp(3 + 2 * PASS);
}

// NTP = n-tiles per wave per pass: self (768 columns) 3 x 2 passes, cross (512 columns) 2 x 2.  A_PLANE = byte distance
// between the hi and lo planes of the activation tile in LDS; MT = 16-row tiles of the workgroup's row tile (4 or 8).
template <int PREC, class TA, int NTP, int NPASS, int A_PLANE = PJL<PREC>::A_PLANE, int MT = 4>
__device__ __forceinline__ void proj_compute(const ProjArgs& a, const TileLoc& t, const char* smA, int stamp_base) {
    static_assert(NPASS == 2, "two passes");
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int NPART = PJ<PREC>::NPART;
    constexpr int NKC = 2 * PJL<PREC>::STAGES;
    constexpr int NBUF = NPART == 2 ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32x4 bf[NBUF][NTP][NPART];
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
#pragma unroll
        for (int j = 0; j < NTP; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) {
                const char* ptr = static_cast<const char*>(a.W) + (p ? (long long)a.Nout * 256 * (long long)sizeof(typename Tag::elem) : 0);
                bf[i][j][p] = *reinterpret_cast<const u32x4*>(ptr + ((long long)((w + 8 * j) * NKC + i) * 64 + lane) * 16);
            }
    __syncthreads();   // the activation tile is complete
    if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + stamp_base + 1] = clock64();
#if LG_PROJ_PINGPONG
    // The two waves of a SIMD (w, w + 4) run the passes in DIFFERENT shapes: waves 0..3 as the product (MFMA loop, epilogue, MFMA loop,
    // epilogue), waves 4..7 both MFMA loops first and both epilogues after — so that an epilogue (VALU, loads, stores) of one wave
    // faces an MFMA loop of the other instead of the other's epilogue.  Same arithmetic per wave: bit-identical outputs.
    f32x4 acc0[MT][NTP], acc1[MT][NTP];
    if (w >= 4) {
        proj_pass<PREC, TA, NTP, NPASS, 0, A_PLANE, MT, 1>(a, t, smA, bf, stamp_base, acc0);
        proj_pass<PREC, TA, NTP, NPASS, 1, A_PLANE, MT, 1>(a, t, smA, bf, stamp_base, acc1);
        proj_pass<PREC, TA, NTP, NPASS, 0, A_PLANE, MT, 2>(a, t, smA, bf, stamp_base, acc0);
        proj_pass<PREC, TA, NTP, NPASS, 1, A_PLANE, MT, 2>(a, t, smA, bf, stamp_base, acc1);
    } else {
        proj_pass<PREC, TA, NTP, NPASS, 0, A_PLANE, MT, 0>(a, t, smA, bf, stamp_base, acc0);
        proj_pass<PREC, TA, NTP, NPASS, 1, A_PLANE, MT, 0>(a, t, smA, bf, stamp_base, acc0);
    }
#else
    proj_pass<PREC, TA, NTP, NPASS, 0, A_PLANE, MT>(a, t, smA, bf, stamp_base);
    proj_pass<PREC, TA, NTP, NPASS, 1, A_PLANE, MT>(a, t, smA, bf, stamp_base);
#endif
}

}  // namespace lg
