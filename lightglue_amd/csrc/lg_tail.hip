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
//   phase A  h = [x ; ctx] Wcat^T + bcat.  The whole 64 x 512 activation tile lives in LDS (128 KB as split
//            bf16): the x half is loaded/converted first, the ctx half is fetched while the x half is being
//            multiplied -> two barriers for the whole phase, weight fragments prefetched from L2 on a ring.
//            Wave w owns the n-tiles {w, w+8, w+16, w+24} (16 hidden units each).
//   LN       two-pass row statistics across the 8 waves (LDS scratch), in registers
//   phase B  out = g W2^T + b2 in 4 steps: step j multiplies K-stages 2j, 2j+1 of g (128 hidden units), then the
//            wave applies GELU to its next n-tile and writes it to LDS in A-operand order (one barrier per step).
//            (Running the two halves of the workgroup in opposite MFMA/GELU order to overlap the pipes was
//            measured and lost 8k cycles per workgroup; see DESIGN.md.)
//   epilogue out tile staged through LDS, residual add and store as full 1 KB rows
//   next     (optional, NEXT != 0) the new x tile is ALSO written to LDS in operand precision and the NEXT block's
//            q/k/v projection (lg_proj_body.h; SelfBlock -> this layer's CrossBlock, CrossBlock -> next layer's
//            SelfBlock) runs here: saves that kernel's launch, its x-tile read + conversion and one grid drain.
#include "lg_proj_body.h"

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

// GELU(u) = 0.5 u (1 + erf(u / sqrt 2)) with a branch-free erf:  erf(|x|) = 1 - exp(-x^2) * sum_{k=1..8} c_k t^k,
// t = 1 / (1 + 0.3275911 |x|)  (Abramowitz-Stegun 7.1.26 form, coefficients re-fitted to degree 8 here:
// max |erf error| 4.2e-7, max |GELU error| 9.7e-8 over [-10, 10] in fp32 — fp32 round-off class).  The ocml
// erff costs ~10x more instructions (two divergent ranges) and was a third of this kernel's time.
// Two values at a time so that the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32.
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 u) {
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
    const f32x2 er = 1.0f - p * e;                       // erf(|x|)
    const f32x2 half_u = u * 0.5f;
    const f32x2 sgn = {copysignf(er[0], x[0]), copysignf(er[1], x[1])};
    return half_u + half_u * sgn;
}

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

constexpr int OT_LD = 260;                       // padded row stride (floats) of the staged fp32 output tile
constexpr int OT_BYTES = TBM * OT_LD * 4;        // 66 560

// NEXT: 0 = plain tail, 1 = + SelfBlock projection of the next layer (768 columns, rotary), 2 = + CrossBlock
// projection (512 columns).  TA = element type of q/k/v (attention operand precision), only read when NEXT != 0.
template <int PREC, int NEXT, class TA>
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
    // optional per-phase shader-clock stamps (profiling tap; a.dbg == nullptr in production)
    auto stamp = [&](int slot) {
        if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + slot] = clock64();
    };
    stamp(0);
    // guide T5, static form: the second-dispatched half of an 8-wave workgroup loses issue arbitration to the older half on
    // every phase; one priority bump for it, no per-cluster flips (measured: tail -0.2 ... -1.3 %)
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);

    // ------------------------------------------------------------------ phase A
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight fragment: plane p, n-tile nt, k-chunk kc -> 64 lanes x 16 B contiguous
    auto wfrag = [&](const void* base, int p, long long plane_elems, int nt, int kc) -> u32x4 {
        const char* ptr = static_cast<const char*>(base) + (p ? plane_elems * (long long)sizeof(typename Tag::elem) : 0);
        return *reinterpret_cast<const u32x4*>(ptr + ((long long)(nt * NKC + kc) * 64 + lane) * 16);
    };
    const void* Wc = a.Wcat;
    const void* W2 = a.W2;

    // ---- activation tile -> LDS.  Half hf (0: x, 1: ctx) = 64 rows x 256 floats = STAGES/2 K-stage tiles.
    // Thread -> (row = tid >> 3, 16-byte-chunk slot = tid & 7) of every stage tile of the half.
    constexpr int HS = STAGES / 2;                    // stage tiles per half (4, or 8 for f32)
    const int srow = tid >> 3, sslot = tid & 7;
    f32x4 hreg[HS][NV];
    auto load_half = [&](int hf) {
        const float* src = (hf ? a.CTX : a.X) + (long long)(t.grow0 + srow) * 256 + sslot * EPC;
#pragma unroll
        for (int st = 0; st < HS; ++st)
#pragma unroll
            for (int j = 0; j < NV; ++j) hreg[st][j] = *reinterpret_cast<const f32x4*>(src + st * KE + 4 * j);
    };
    auto store_half = [&](int hf) {
        const int off = lds_off<128>(srow, sslot);
#pragma unroll
        for (int st = 0; st < HS; ++st) {
            char* tile = smem + (hf * HS + st) * TILE;   // planes: p * G_PLANE
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(tile + off) = hreg[st][0];
            } else if constexpr (PREC == PREC_BF16X3) {
                u32x4 hi, lo;
                split8_bf16(hreg[st][0], hreg[st][1], hi, lo);
                *reinterpret_cast<u32x4*>(tile + off) = hi;
                *reinterpret_cast<u32x4*>(tile + G_PLANE + off) = lo;
            } else {
                *reinterpret_cast<u32x4*>(tile + off) = pack8<Tag>(hreg[st][0], hreg[st][1]);
            }
        }
    };
    // NOTE no branch around prefetches anywhere in this kernel: hipcc counts s_waitcnt conservatively at a join,
    // so a conditional load forces vmcnt(0) right after it (guide §5 trap (c)); past-the-end prefetches are
    // clamped instead.  sched_barrier(0) pins the software pipeline (hipcc otherwise sinks a prefetch next to
    // its use to save registers, i.e. un-pipelines the loop).
    constexpr int NBUF = NPART == 2 ? 2 : 4;
    u32x4 bf[NBUF][4][NPART];   // ring of B fragments: this wave's 4 n-tiles x planes per k-chunk
    auto load_b_A = [&](u32x4 (&dst)[4][NPART], int kc) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[nt][p] = wfrag(Wc, p, 512LL * 512, w + 8 * nt, kc);
    };
    auto chunk_A = [&](int kc, const u32x4 (&b)[4][NPART]) {   // one k-chunk of MFMAs against the LDS-resident tile
        const char* tile = smem + (kc >> 1) * TILE;
        u32x4 af[4][NPART];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int p = 0; p < NPART; ++p)
                af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * G_PLANE + lds_off<128>(mt * 16 + lr, (kc & 1) * 4 + g));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) tail_mma<PREC>(acc[mt][nt], af[mt], b[nt]);
    };
    load_half(0);
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load_b_A(bf[i], i);
    store_half(0);
    __syncthreads();
    load_half(1);                       // ctx rows stream in while the x half is multiplied
    __builtin_amdgcn_sched_barrier(0);
    constexpr int HC = NKC / 2;         // k-chunks per half
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll 1
        for (int c0 = 0; c0 < HC; c0 += NBUF) {
#pragma unroll
            for (int i = 0; i < NBUF; ++i) {
                const int kc = hf * HC + c0 + i;
                load_b_A(bf[(i + NBUF - 1) % NBUF], kc + NBUF - 1 < NKC ? kc + NBUF - 1 : NKC - 1);
                __builtin_amdgcn_sched_barrier(0);
                chunk_A(kc, bf[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (hf == 0) {
            store_half(1);
            __syncthreads();
        }
    }
    stamp(1);
    // ------------------------------------------------------------------ bias + LayerNorm(512) + GELU
    {
        float bias[4], gam[4], bet[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = (w + 8 * nt) * 16 + lr;
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
                    p[mt][r] = row16_sum(p[mt][r]);
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
            for (int mt = 0; mt < 4; ++mt) {   // rows 4g..4g+3 of tile mt are contiguous: one 16-byte read per wave slot
                f32x4 v = *reinterpret_cast<const f32x4*>(red + mt * 16 + g * 4);
#pragma unroll
                for (int ww = 1; ww < 8; ++ww) v += *reinterpret_cast<const f32x4*>(red + ww * TBM + mt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) p[mt][r] = v[r];
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
                const float rstd = __builtin_amdgcn_rsqf(part[mt][r] * (1.f / 512.f) + 1e-5f);   // v_rsq_f32, 1 ulp
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][r] = acc[mt][nt][r] * rstd * gam[nt] + bet[nt];   // pre-GELU
            }
    }
    stamp(2);
    // ------------------------------------------------------------------ GELU + g -> LDS (A-operand order) + phase B
    // n-tile j of wave w = hidden units [(w + 8j)*16, +16) = K-stage 2j + (w >> 2), columns (w & 3)*16 + lr of it.
    auto gelu_store = [&](int j) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x2 v01 = gelu_fast2(f32x2{acc[mt][j][0], acc[mt][j][1]});
            f32x2 v23 = gelu_fast2(f32x2{acc[mt][j][2], acc[mt][j][3]});
            const float gv[4] = {v01[0], v01[1], v23[0], v23[1]};
            if constexpr (EPC == 8) {
                char* tile0 = smem + (2 * j + (w >> 2)) * TILE;
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
                    // even lanes write row rp, odd lanes row rp+1; each writes the (even col, odd col) pair
                    const bool odd = lr & 1;
                    const float mine = odd ? gv[rp + 1] : gv[rp];
                    const float give = odd ? gv[rp] : gv[rp + 1];
                    const float got = dpp_xor1(give);
                    const float c0 = odd ? got : mine, c1 = odd ? mine : got;   // values at (even col, odd col)
                    const int row = mt * 16 + g * 4 + rp + (odd ? 1 : 0);
                    const int col = (w & 3) * 16 + (lr & ~1);
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
                for (int r = 0; r < 4; ++r) {
                    const int hcol = (w + 8 * j) * 16 + lr, row = mt * 16 + g * 4 + r;   // f32: K-stage = 32 hidden units
                    char* tile = smem + (hcol >> 5) * TILE;
                    *reinterpret_cast<float*>(tile + lds_off<128>(row, (hcol & 31) >> 2) + (hcol & 3) * 4) = gv[r];
                }
            }
        }
    };
    // all waves passed the barriers of the last block_row_sum => the activation tile is dead, g may overwrite it
    f32x4 acc2[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc2[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    u32x4 b2f[4][2][NPART];   // ring, 3 k-chunks ahead
    auto load_b_B = [&](u32x4 (&dst)[2][NPART], int kc) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[nt][p] = wfrag(W2, p, 256LL * 512, w * 2 + nt, kc);
    };
    auto chunk_B = [&](int kc, const u32x4 (&b)[2][NPART]) {
        const char* tile = smem + (kc >> 1) * TILE;
        u32x4 af[4][NPART];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int p = 0; p < NPART; ++p)
                af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * G_PLANE + lds_off<128>(mt * 16 + lr, (kc & 1) * 4 + g));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) tail_mma<PREC>(acc2[mt][nt], af[mt], b[nt]);
    };
    load_b_B(b2f[0], 0); load_b_B(b2f[1], 1); load_b_B(b2f[2], 2);
    gelu_store(0);
    __syncthreads();
    stamp(3);
    constexpr int CPS = NKC / 4;   // k-chunks per step (4 for 16-bit: K-stages 2j, 2j+1; 8 for f32)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < CPS; ++i) {
            const int kc = j * CPS + i;
            load_b_B(b2f[(kc + 3) & 3], kc + 3 < NKC ? kc + 3 : NKC - 1);
            chunk_B(kc, b2f[kc & 3]);
        }
        if (j < 3) {
            gelu_store(j + 1);
            __syncthreads();
        }
    }
    stamp(4);
    // residual rows for the epilogue: issue the loads now so that they land during the output staging
    const int qlen = a.rs.len[t.seg];
    f32x4 xres[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + TTHREADS * i, row = c >> 6, c4 = c & 63;
        xres[i] = *reinterpret_cast<const f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + c4 * 4);
    }
    // NEXT: rotary tables of the tile for the next projection (registers now, LDS after the g tiles are dead)
    f32x4 ropec = {0.f, 0.f, 0.f, 0.f}, ropes = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NEXT == 1) {
        ropec = *reinterpret_cast<const f32x4*>(a.next.cosb + (long long)t.grow0 * 32 + tid * 4);
        ropes = *reinterpret_cast<const f32x4*>(a.next.sinb + (long long)t.grow0 * 32 + tid * 4);
    }
    char* smA = smem + OT_BYTES;                                   // next projection: activation tile (operand precision)
    float* smCS = reinterpret_cast<float*>(smA + PJL<PREC>::A_BYTES);   //                  rotary tables
    __syncthreads();   // g tiles are dead; reuse the region as a [64][256+4] fp32 output tile
    {
        float* ot = reinterpret_cast<float*>(smem);
        constexpr int OLD = OT_LD;   // padded row stride (floats): rows 4g+r land on different banks
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
#pragma unroll
        for (int i = 0; i < 8; ++i) {     // 64 rows x 64 float4 = 4096 chunks, 8 per thread; a wave covers one full row
            const int c = tid + TTHREADS * i, row = c >> 6, c4 = c & 63;
            const f32x4 d = *reinterpret_cast<const f32x4*>(ot + row * OLD + c4 * 4);
            const f32x4 xn = xres[i] + d;
            if (t.r0 + row < qlen) *reinterpret_cast<f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + c4 * 4) = xn;
            if constexpr (NEXT != 0) {   // the same 4 columns, in operand precision, into the projection's A tile
                static_assert(EPC == 8, "fused next projection: 16-bit operands only");
                const int col = c4 * 4;
                char* dst = smA + (col >> 6) * PJL<PREC>::TILE + lds_off<128>(row, (col & 63) >> 3) + (col & 7) * 2;
                if constexpr (PREC == PREC_BF16X3) {
                    const float h0 = bf16_round(xn[0]), h1 = bf16_round(xn[1]), h2 = bf16_round(xn[2]), h3 = bf16_round(xn[3]);
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2_bf16(h0, h1), pack2_bf16(h2, h3)};
                    *reinterpret_cast<u32x2*>(dst + PJL<PREC>::A_PLANE) = u32x2{pack2_bf16(xn[0] - h0, xn[1] - h1), pack2_bf16(xn[2] - h2, xn[3] - h3)};
                } else {
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<Tag>(xn[0], xn[1]), pack2<Tag>(xn[2], xn[3])};
                }
            }
        }
    }
    stamp(5);
    if constexpr (NEXT != 0) {
        if constexpr (NEXT == 1) {
            *reinterpret_cast<f32x4*>(smCS + tid * 4) = ropec;
            *reinterpret_cast<f32x4*>(smCS + 2048 + tid * 4) = ropes;
        }
        // staging of the projection outputs aliases the fp32 output tile: every thread is past its reads of it when it
        // reaches the barrier inside proj_compute, and the staging is first written after that barrier
        proj_compute<PREC, TA, NEXT == 1 ? 3 : 2, 2>(a.next, t, smA, smem, smCS, 8);
    }
}

template <int PREC, int NEXT, class TA> static hipError_t launch_tail_t(const TailArgs& a, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    auto kern = tail_kernel<PREC, NEXT, TA>;
    constexpr int base = TL<PREC>::TOTAL > OT_BYTES ? TL<PREC>::TOTAL : OT_BYTES;
    constexpr int fused = OT_BYTES + PJL<PREC>::A_BYTES + PJ_CS_BYTES;
    constexpr int smem = (NEXT != 0 && fused > base) ? fused : base;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / TBM), dim3(TTHREADS), smem, s, a);
    return hipGetLastError();
}
template <int PREC, class TA> static hipError_t launch_tail_next(const TailArgs& a, hipStream_t s) {
    if (!a.next.W) return launch_tail_t<PREC, 0, TA>(a, s);
    static_assert(PJO<TA, 3>::O_BYTES <= OT_BYTES, "projection staging must fit the dead output tile");
    if (a.next.Nout == 768 && a.next.cosb) return launch_tail_t<PREC, 1, TA>(a, s);
    if (a.next.Nout == 512 && !a.next.cosb) return launch_tail_t<PREC, 2, TA>(a, s);
    return hipErrorInvalidValue;
}

// the fused next projection exists for the 16-bit operand / attention combinations the engine runs by default
bool launch_tail_supports_next(int prec, int attn_prec) {
    return (prec == PREC_BF16X3 && attn_prec == PREC_F16) || (prec == PREC_BF16 && attn_prec == PREC_BF16) ||
           (prec == PREC_F16 && attn_prec == PREC_F16);
}

hipError_t launch_tail(int prec, int attn_prec, const TailArgs& a, hipStream_t s) {
    if (a.next.W && !launch_tail_supports_next(prec, attn_prec)) return hipErrorInvalidValue;
    switch (prec) {
        case PREC_F32: return launch_tail_t<PREC_F32, 0, float>(a, s);
        case PREC_BF16: return launch_tail_next<PREC_BF16, bf16_t>(a, s);
        case PREC_F16: return launch_tail_next<PREC_F16, f16_t>(a, s);
        case PREC_BF16X3: return launch_tail_next<PREC_BF16X3, f16_t>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace lg
