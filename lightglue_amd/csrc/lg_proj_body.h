// lightglue_amd — the compute part of the attention input projections, shared by the standalone projection kernel
// (lg_proj.hip) and by the fused tail (lg_tail.hip), which runs the NEXT block's projection on the x tile it has
// just produced instead of writing it out and launching a second kernel.
// Precondition: the 64 x 256 activation tile sits in LDS at smA in operand precision ([NPART planes][STAGES][64][128 B],
// XOR-swizzled with lds_off<128>), the tile's rotary tables (if any) at smCS, and NO barrier has been executed since
// those writes (proj_compute issues its weight prefetch first and then synchronises).
#pragma once
#include "lg_kernels.h"

namespace lg {

constexpr int PBM = 64, PTHREADS = 512;

template <int PREC> struct PJ;
template <> struct PJ<PREC_F32> { typedef TagF32 Tag; static constexpr int NPART = 1; };
template <> struct PJ<PREC_BF16> { typedef TagBF16 Tag; static constexpr int NPART = 1; };
template <> struct PJ<PREC_F16> { typedef TagF16 Tag; static constexpr int NPART = 1; };
template <> struct PJ<PREC_BF16X3> { typedef TagBF16 Tag; static constexpr int NPART = 2; };

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
    static constexpr int A_BYTES = PJ<PREC>::NPART * A_PLANE;
};
template <class TA, int NTP> struct PJO {             // LDS geometry of the output staging
    static constexpr int NHC = NTP * 2;                // 64-column head chunks per pass (6 or 4)
    static constexpr int LINE = 64 * (int)sizeof(TA) + 16;   // padded staging line (64 elements)
    static constexpr int HCB = 64 * LINE;              // staging bytes of one head chunk
    static constexpr int O_BYTES = NHC * HCB;
};
constexpr int PJ_CS_BYTES = 2 * 64 * 32 * 4;           // cos [64][32] + sin [64][32]

// NTP = n-tiles per wave per pass, NPASS passes: self (768 columns) 3 x 2 — or 2 x 3 when the staged outputs are
// fp32 and 3 x 2 would not fit LDS —, cross (512 columns) 2 x 2.
template <int PREC, class TA, int NTP, int NPASS>
__device__ __forceinline__ void proj_compute(const ProjArgs& a, const TileLoc& t, char* smA, char* smO, const float* smCS, int stamp_base) {
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int NPART = PJ<PREC>::NPART;
    constexpr int STAGES = PJL<PREC>::STAGES, NKC = 2 * STAGES, TILE = PJL<PREC>::TILE, A_PLANE = PJL<PREC>::A_PLANE;
    constexpr int NHC = PJO<TA, NTP>::NHC, LINE = PJO<TA, NTP>::LINE, HCB = PJO<TA, NTP>::HCB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, g = lane >> 4;
    const long long R = a.R;
    const bool rope = a.cosb != nullptr;
    auto stamp = [&](int slot) {   // profiling tap (a.dbg == nullptr in production)
        if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + stamp_base + slot] = clock64();
    };
    auto wfrag = [&](int p, int nt, int kc) -> u32x4 {
        const char* ptr = static_cast<const char*>(a.W) + (p ? (long long)a.Nout * 256 * (long long)sizeof(typename Tag::elem) : 0);
        return *reinterpret_cast<const u32x4*>(ptr + ((long long)(nt * NKC + kc) * 64 + lane) * 16);
    };
    constexpr int NBUF = NPART == 2 ? 2 : 4;
    u32x4 bf[NBUF][NTP][NPART];
    auto load_b = [&](u32x4 (&dst)[NTP][NPART], int pass, int kc) {
#pragma unroll
        for (int j = 0; j < NTP; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(p, w + 8 * (pass * NTP + j), kc);
    };
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load_b(bf[i], 0, i);
    __syncthreads();
    stamp(1);

#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        f32x4 acc[4][NTP];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NTP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int c0 = 0; c0 < NKC; c0 += NBUF) {
#pragma unroll
            for (int i = 0; i < NBUF; ++i) {
                const int kc = c0 + i;
                // prefetch NBUF-1 chunks ahead; past the end of a pass, start on the next pass's first chunks
                const int nk = kc + NBUF - 1;
                const int npass = nk < NKC ? pass : (pass + 1 < NPASS ? pass + 1 : pass);
                load_b(bf[(i + NBUF - 1) % NBUF], npass, nk < NKC ? nk : nk - NKC);
                __builtin_amdgcn_sched_barrier(0);
                const char* tile = smA + (kc >> 1) * TILE;
                u32x4 af[4][NPART];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int p = 0; p < NPART; ++p)
                        af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * A_PLANE + lds_off<128>(mt * 16 + lr, (kc & 1) * 4 + g));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int j = 0; j < NTP; ++j) {
                        if constexpr (NPART == 2) {
                            mma_chunk<Tag>(acc[mt][j], af[mt][1], bf[i][j][0]);
                            mma_chunk<Tag>(acc[mt][j], af[mt][0], bf[i][j][1]);
                        }
                        mma_chunk<Tag>(acc[mt][j], af[mt][0], bf[i][j][0]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(2 + 3 * pass);
        // ---- epilogue of the pass: bias, rotary, stage through LDS
        if (pass > 0) __syncthreads();    // the previous pass's staged tile has been fully read
#pragma unroll
        for (int j = 0; j < NTP; ++j) {
            const int nt = w + 8 * (pass * NTP + j);          // global n-tile
            const int col0 = nt * 16;                         // global output column of lane lr == 0
            const int group = col0 >> 8, d = (col0 & 63) + lr;
            const int hcl = (col0 >> 6) - pass * NHC;         // head chunk within this pass
            const float bv = a.bias[col0 + lr];
            char* hc = smO + hcl * HCB;
            if (group < a.n_qk_groups) {   // q / k (or qk): [row][64], written as (even, odd) column pairs
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[mt][j][r] + bv;
                        if (rope) {   // ref :58-65: pairs are adjacent columns = adjacent lanes
                            const int lrow = mt * 16 + g * 4 + r;
                            const float other = dpp_xor1(v[r]);
                            const float c = smCS[lrow * 32 + (d >> 1)], s = smCS[2048 + lrow * 32 + (d >> 1)];
                            v[r] = (d & 1) ? (v[r] * c + other * s) : (v[r] * c - other * s);
                        }
                    }
                    if constexpr (sizeof(TA) == 2) {
#pragma unroll
                        for (int rp = 0; rp < 4; rp += 2) {
                            const bool odd = lr & 1;
                            const float mine = odd ? v[rp + 1] : v[rp];
                            const float give = odd ? v[rp] : v[rp + 1];
                            const float got = dpp_xor1(give);
                            const float c0 = odd ? got : mine, c1 = odd ? mine : got;
                            const int row = mt * 16 + g * 4 + rp + (odd ? 1 : 0);
                            typedef TA ta2 __attribute__((ext_vector_type(2)));
                            ta2 o = {pj_cvt<TA>(c0), pj_cvt<TA>(c1)};
                            *reinterpret_cast<ta2*>(hc + row * LINE + (d & ~1) * 2) = o;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(hc + (mt * 16 + g * 4 + r) * LINE + d * 4) = v[r];
                    }
                }
            } else {                       // v: transposed [d][64 rows]; a lane holds 4 consecutive rows
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    typedef TA ta4 __attribute__((ext_vector_type(4)));
                    ta4 o = {pj_cvt<TA>(acc[mt][j][0] + bv), pj_cvt<TA>(acc[mt][j][1] + bv), pj_cvt<TA>(acc[mt][j][2] + bv), pj_cvt<TA>(acc[mt][j][3] + bv)};
                    *reinterpret_cast<ta4*>(hc + d * LINE + (mt * 16 + g * 4) * (int)sizeof(TA)) = o;
                }
            }
        }
        __syncthreads();
        stamp(3 + 3 * pass);
        // ---- cooperative store: every head chunk is 64 lines of 64 elements; 16 bytes per thread
        constexpr int PPL = 64 * (int)sizeof(TA) / 16;        // 16-byte pieces per line (8 or 16)
        constexpr int TOTALP = NHC * 64 * PPL;
#pragma unroll
        for (int i = 0; i < TOTALP / PTHREADS; ++i) {
            const int idx = tid + PTHREADS * i;
            const int piece = idx % PPL, line = (idx / PPL) & 63, hcl = idx / (PPL * 64);
            const int hcg = pass * NHC + hcl;                 // global head chunk: group = hcg / 4, head = hcg & 3
            const int group = hcg >> 2, head = hcg & 3;
            const u32x4 val = *reinterpret_cast<const u32x4*>(smO + hcl * HCB + line * LINE + piece * 16);
            TA* dst;
            if (group < a.n_qk_groups) dst = static_cast<TA*>(group == 0 ? a.q : a.k) + ((long long)head * R + t.grow0 + line) * 64;   // line = row
            else dst = static_cast<TA*>(a.vt) + ((long long)head * 64 + line) * R + t.grow0;                                            // line = d
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dst) + piece * 16) = val;
        }
        stamp(4 + 3 * pass);
    }
}

}  // namespace lg
