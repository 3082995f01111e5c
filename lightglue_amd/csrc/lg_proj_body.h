// lightglue_amd — the compute part of the attention input projections, shared by the standalone projection kernel
// (lg_proj.hip) and by the fused tail (lg_tail.hip), which runs the NEXT block's projection on the x tile it has
// just produced instead of writing it out and launching a second kernel.
// Precondition: the 64 x 256 activation tile sits in LDS at smA in operand precision ([NPART planes][STAGES][64][128 B],
// XOR-swizzled with pj_tile_off) and NO barrier has been executed since those writes (proj_compute issues its weight
// prefetch first and then synchronises).
#pragma once
#include "lg_kernels.h"

namespace lg {

constexpr int PBM = 64, PTHREADS = 512;
// q and k leave every projection pre-multiplied by the SQUARE ROOT of the attention's score scale, sqrt(log2(e) / sqrt(64)) — the
// reference's own CPU path splits its scale over both operands the same way (lightglue.py:215) — so that the attention kernels
// exponentiate the MFMA result directly (exp2, no per-score multiply; round 3 A/B: +1.6 % whole step, max |dscore| 4.2e-4 -> 2.8e-4).
constexpr float QK_PRESCALE = 0.42466090014400953f;

// Operand scheme of the projection.  NPART = weight planes, APART = activation planes, OPART = planes of q / k / v written.
//   PREC_F16X3      (default): split-f16 activations x split-f16 weights, THREE MFMAs per product, and q / k / v leave as split f16
//                   (hi + lo planes) for the split attention kernel — what holds the 1e-3 score bar when attention logits are sharp
//                   (recipe-D fixtures: one f16 plane anywhere on the q.k path costs 1e-2 in the scores).
//   PREC_QKV_F16W2  (precision f16x3 with attention_precision fp16, the fast opt-in): the activation tile as ONE f16 plane, the
//                   weights as split f16, TWO MFMAs per product; q / k / v rounded to one f16 plane (what the reference's own GPU
//                   path feeds its fp16 SDPA, lightglue.py:119).  Holds the bar only while attention is diffuse (recipes A-C).
constexpr int PREC_QKV_F16W2 = 100;
template <int PREC> struct PJ;
template <> struct PJ<PREC_F32> { typedef TagF32 Tag; static constexpr int NPART = 1, APART = 1, OPART = 1; };
template <> struct PJ<PREC_BF16> { typedef TagBF16 Tag; static constexpr int NPART = 1, APART = 1, OPART = 1; };
template <> struct PJ<PREC_F16> { typedef TagF16 Tag; static constexpr int NPART = 1, APART = 1, OPART = 1; };
template <> struct PJ<PREC_F16X3> { typedef TagF16 Tag; static constexpr int NPART = 2, APART = 2, OPART = 2; };
template <> struct PJ<PREC_QKV_F16W2> { typedef TagF16 Tag; static constexpr int NPART = 2, APART = 1, OPART = 1; };

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
// Which n-tile of the packed weights slot j of pass `pass` is for wave w, and of which kind.  Self (768 columns = q | k | v groups of
// 16 n-tiles, NTP = 3): pass p = the wave's PAIR of adjacent tiles of group p (q, then k) + one v tile; cross (512 columns = qk | v,
// NTP = 2): pass 0 = the qk pair, pass 1 = two v tiles.  A q / k pair is the two tiles 2w, 2w + 1 of its group = 32 consecutive head
// channels, and the host packs their 32 weight rows so that MFMA row 4g + r of tile e holds channel 8g + 4e + r (lg_engine.hip
// proj_row_permutation): in the transposed form lane (lr, g) then ends with 8 CONSECUTIVE channels 8g .. 8g + 7 of keypoint row lr —
// one 16-byte store per plane instead of two 8-byte ones (the projection epilogues were store-issue bound: 48 dwordx2 stores per wave).
// Keypoint ROWS are dealt the same way between pairs of 16-row tiles (MT >= 2): MFMA row slot i of m-tile 2q + e is keypoint row
// 32q + 8 (i >> 2) + 4e + (i & 3) of the workgroup's tile — a lane of a plain-form v tile (row slots 4g .. 4g + 3 of channel lr) then
// holds rows 8g .. 8g + 7 over the pair: one 16-byte store into the [head][64][R] layout.  A free choice (which LDS row a lane reads);
// the activation tile uses the swizzle of the attention's K tile (k-chunk slot ^ row bits 1, 3, 4), which is conflict-free for
// exactly this row pattern.
template <int MT> __device__ __forceinline__ int pj_row(int mt, int slot) {
    if constexpr (MT >= 2) return 32 * (mt >> 1) + 8 * (slot >> 2) + 4 * (mt & 1) + (slot & 3);
    else return slot;
}
__device__ __forceinline__ int pj_tile_off(int row, int slot16) { return row * 128 + ((slot16 ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1))) << 4); }
template <int NTP> __device__ __forceinline__ constexpr bool pj_is_v(int pass, int j) { return NTP == 3 ? j == 2 : pass == 1; }
template <int NTP> __device__ __forceinline__ int pj_tile(int w, int pass, int j) {
    if constexpr (NTP == 3) return j < 2 ? 16 * pass + 2 * w + j : 32 + w + 8 * pass;
    else return pass == 0 ? 2 * w + j : 16 + w + 8 * j;
}
// One pass of the projection.  The kind of a tile decides the FORM of its MFMAs:
//   q / k tiles: TRANSPOSED, C^T = W x^T (weights as the A operand): lane (lr, g) ends with keypoint row lr of the 16-row tile
//     and 4 consecutive head channels per tile -> bias as float4s, the rotary pair (2j, 2j+1) sits in one lane (no cross-lane
//     traffic, ref :58-65), and the pair's 8 channels leave as ONE 16-byte store per plane: lanes g = 0..3 cover 64 contiguous
//     bytes of a [row][64] line.
//   v tiles: plain C = x W^T: lane (lr, g) holds channel lr of row slots 4g..4g+3 = rows 8g..8g+7 over a pair of m-tiles (pj_row)
//     -> one 16-byte store per plane into the transposed [head][64][R] layout (8 consecutive rows of one channel).
// Nothing is staged through LDS and no barrier follows the MFMA loop (the staged version spent 40 % of the kernel in its two
// epilogues: LDS write, barrier, LDS read, store, barrier).
// the rotary rows of the tile (cos, sin: one float4 per 16-row tile and lane): the same 32 values for the q pass and the k pass
template <int MT> struct RopeRows { f32x4 c[MT], s[MT]; };
template <int MT>
__device__ __forceinline__ void proj_rope_load(const ProjArgs& a, const TileLoc& t, RopeRows<MT>& rr) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const int pcol = 32 * w;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long long row = t.grow0 + pj_row<MT>(mt, lr);
        rr.c[mt] = *reinterpret_cast<const f32x4*>(a.cosb + row * 32 + ((pcol & 63) >> 1) + 4 * g);
        rr.s[mt] = *reinterpret_cast<const f32x4*>(a.sinb + row * 32 + ((pcol & 63) >> 1) + 4 * g);
    }
}
// AHEAD (the fused tail's form): activation fragments read from LDS one chunk ahead of their MFMAs and the biases requested ahead of the loop.  The
// standalone kernel keeps the round-3 order (fragments and biases right where they are used): with AHEAD it measured 0.114 -> 0.118 ms per launch
// (profiles/r04g_ab_cfg2.log), the fused tail -1.5 %.
template <int PREC, class TA, int NTP, int NPASS, int PASS, int A_PLANE, int MT, bool AHEAD>
__device__ __forceinline__ void proj_pass(const ProjArgs& a, const TileLoc& t, const char* smA, u32x4 (&bf)[PJ<PREC>::NPART == 2 ? 2 : 4][NTP][PJ<PREC>::NPART],
                                          int stamp_base, const RopeRows<MT>& rr) {
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int NPART = PJ<PREC>::NPART, APART = PJ<PREC>::APART;
    constexpr int STAGES = PJL<PREC>::STAGES, NKC = 2 * STAGES, TILE = MT * 16 * 128;   // one plane of one K stage: MT*16 rows x 128 B
    constexpr int NBUF = NPART == 2 ? 2 : 4;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const long long R = a.R;
    auto stamp = [&](int slot) {   // profiling tap (a.dbg == nullptr in production)
        if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + stamp_base + slot] = clock64();
    };
    const __amdgpu_buffer_rsrc_t wrs = weight_rsrc(a.W);
    const int lane16 = lane * 16;
    auto wfrag = [&](int p, int nt, int kc) -> u32x4 {
        return weight_frag(wrs, lane16, (p ? a.Nout * 256 * (int)sizeof(typename Tag::elem) : 0) + (nt * NKC + kc) * 1024);
    };
    auto load_b = [&](u32x4 (&dst)[NTP][NPART], int pass, int kc) {
#pragma unroll
        for (int j = 0; j < NTP; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(p, pj_tile<NTP>(w, pass, j), kc);
    };
    constexpr bool HAS_PAIR = !pj_is_v<NTP>(PASS, 0);          // slots 0, 1 = a q / k pair
    constexpr bool ROPE = NTP == 3;                            // SelfBlock: rotary on q and k (ref :58-65)
    const int pcol = (NTP == 3 ? PASS * 256 : 0) + 32 * w;     // first packed column of the pair: [group][head][64]
    f32x4 pb[2]; f32x4 pc[MT], ps[MT]; float bv[NTP];
    auto load_operands = [&]() {                               // the epilogue's operands: biases (shared by every workgroup: L2 / L1 hits); rotary rows from `rr`
        if constexpr (HAS_PAIR) {
            pb[0] = *reinterpret_cast<const f32x4*>(a.bias + pcol + 8 * g); pb[1] = *reinterpret_cast<const f32x4*>(a.bias + pcol + 8 * g + 4);
            if constexpr (ROPE) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { pc[mt] = rr.c[mt]; ps[mt] = rr.s[mt]; }
            }
        }
#pragma unroll
        for (int j = 0; j < NTP; ++j)
            if (pj_is_v<NTP>(PASS, j)) bv[j] = a.bias[pj_tile<NTP>(w, PASS, j) * 16 + lr];
    };
    if constexpr (AHEAD) load_operands();
    f32x4 acc[MT][NTP];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // AHEAD: activation fragments are read from LDS one chunk ahead of their MFMAs (two register sets; round-4 ISA: read right in front of the
    // MFMAs they cost every chunk an exposed LDS round trip); a pass's first chunk reads its own
    static_assert(MT <= 4 && NBUF % 2 == 0, "row tiles per workgroup / ring depth");
    u32x4 afh[2][MT][APART];
    auto read_af = [&](u32x4 (&af)[MT][APART], int kc) {
        const char* tile = smA + (kc >> 1) * TILE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < APART; ++p)
                af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * A_PLANE + pj_tile_off(pj_row<MT>(mt, lr), (kc & 1) * 4 + g));
    };
    if constexpr (AHEAD) read_af(afh[0], 0);
#pragma unroll 1
    for (int c0 = 0; c0 < NKC; c0 += NBUF) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) {
            const int kc = c0 + i;
            // prefetch NBUF-1 chunks ahead; past the end of a pass, start on the next pass's first chunks
            const int nk = kc + NBUF - 1;
            const int npass = nk < NKC ? PASS : (PASS + 1 < NPASS ? PASS + 1 : PASS);
            load_b(bf[(i + NBUF - 1) % NBUF], npass, nk < NKC ? nk : nk - NKC);
            if constexpr (AHEAD) read_af(afh[(i + 1) & 1], kc + 1 < NKC ? kc + 1 : kc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!AHEAD) read_af(afh[i & 1], kc);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < NTP; ++j) {
                    if (pj_is_v<NTP>(PASS, j)) pj_mma<PREC, false>(acc[mt][j], bf[i][j], afh[i & 1][mt]);   // compile-time after unrolling
                    else pj_mma<PREC, true>(acc[mt][j], bf[i][j], afh[i & 1][mt]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp(2 + 2 * PASS);
    // ---- epilogue of the pass, straight from the accumulators.  All loads come BEFORE the first store: the compiler cannot prove that q / k / v do
    // not alias the tables, so a load placed after a store stays there (one exposed round trip per iteration)
    if constexpr (!AHEAD) load_operands();
    typedef TA ta4 __attribute__((ext_vector_type(4)));
    // range guard (LG_FLAG_CHECK_FINITE; workgroup-uniform switch): q / k / v values of live rows, tested right in front of their f16 split
    const bool range_on = a.range_flag != nullptr;
    const int live_rows = range_on ? a.rs.len[t.seg] - t.r0 : 0;
    bool out_of_range = false;
    auto bad4 = [](const f32x4& v) { return !(fabsf(v[0]) < 65504.f) | !(fabsf(v[1]) < 65504.f) | !(fabsf(v[2]) < 65504.f) | !(fabsf(v[3]) < 65504.f); };
    constexpr int OPART = PJ<PREC>::OPART;
    static_assert(OPART == 1 || sizeof(TA) == 2, "split q / k / v planes are f16");
    if constexpr (HAS_PAIR) {                                  // q / k (or qk) pair: 8 consecutive channels per lane and keypoint row
        TA* base = static_cast<TA*>((NTP == 3 && PASS == 1) ? a.k : a.q);
        const int head = (pcol >> 6) & 3, d0 = pcol & 63;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const long long row = t.grow0 + pj_row<MT>(mt, lr);
            f32x4 v0 = acc[mt][0] + pb[0], v1 = acc[mt][1] + pb[1];
            if constexpr (ROPE) {                              // pairs (2f, 2f+1) with frequency f
                const f32x4 c = pc[mt], sn = ps[mt];
                const f32x4 u0 = v0, u1 = v1;
                v0[0] = u0[0] * c[0] - u0[1] * sn[0]; v0[1] = u0[1] * c[0] + u0[0] * sn[0];
                v0[2] = u0[2] * c[1] - u0[3] * sn[1]; v0[3] = u0[3] * c[1] + u0[2] * sn[1];
                v1[0] = u1[0] * c[2] - u1[1] * sn[2]; v1[1] = u1[1] * c[2] + u1[0] * sn[2];
                v1[2] = u1[2] * c[3] - u1[3] * sn[3]; v1[3] = u1[3] * c[3] + u1[2] * sn[3];
            }
            v0 *= QK_PRESCALE; v1 *= QK_PRESCALE;
            if (range_on && pj_row<MT>(mt, lr) < live_rows) out_of_range |= bad4(v0) | bad4(v1);
            TA* dst = base + ((long long)head * R + row) * 64 + d0 + 8 * g;
            if constexpr (OPART == 2) {                        // hi plane + lo plane (f16 of the residual, exact subtraction in fp32)
                u32x4 hi, lo;
                split8_f16<true>(v0, v1, hi, lo);
                *reinterpret_cast<u32x4*>(dst) = hi;
                *reinterpret_cast<u32x4*>(dst + a.plane) = lo;
            } else if constexpr (sizeof(TA) == 2) {
                *reinterpret_cast<u32x4*>(dst) = pack8<Tag>(v0, v1);
            } else {
                *reinterpret_cast<f32x4*>(dst) = v0; *reinterpret_cast<f32x4*>(dst + 4) = v1;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NTP; ++j) {
        if (!pj_is_v<NTP>(PASS, j)) continue;                  // v: plain tile -> transposed layout [head][64][R]
        const int col0 = pj_tile<NTP>(w, PASS, j) * 16;        // first packed column of the tile
        const int head = (col0 >> 6) & 3, d0 = col0 & 63;
        TA* vrow = static_cast<TA*>(a.vt) + ((long long)head * 64 + d0 + lr) * R + t.grow0;
        if constexpr (MT >= 2) {                               // rows 8g .. 8g + 7 of a pair of m-tiles: 16-byte stores
#pragma unroll
            for (int q = 0; q < MT / 2; ++q) {
                const f32x4 v0 = acc[2 * q][j] + bv[j], v1 = acc[2 * q + 1][j] + bv[j];
                TA* dst = vrow + 32 * q + 8 * g;
                if (range_on) {   // rows 32 q + 8 g + 0..3 (v0) and + 4..7 (v1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        out_of_range |= (32 * q + 8 * g + r < live_rows) & !(fabsf(v0[r]) < 65504.f);
                        out_of_range |= (32 * q + 8 * g + 4 + r < live_rows) & !(fabsf(v1[r]) < 65504.f);
                    }
                }
                if constexpr (OPART == 2) {
                    u32x4 hi, lo;
                    split8_f16<true>(v0, v1, hi, lo);
                    *reinterpret_cast<u32x4*>(dst) = hi;
                    *reinterpret_cast<u32x4*>(dst + a.plane) = lo;
                } else if constexpr (sizeof(TA) == 2) {
                    *reinterpret_cast<u32x4*>(dst) = pack8<Tag>(v0, v1);
                } else {
                    *reinterpret_cast<f32x4*>(dst) = v0; *reinterpret_cast<f32x4*>(dst + 4) = v1;
                }
            }
        } else {
            const f32x4 v = acc[0][j] + bv[j];
            TA* dst = vrow + 4 * g;
            if (range_on) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out_of_range |= (4 * g + r < live_rows) & !(fabsf(v[r]) < 65504.f);
            }
            if constexpr (OPART == 2) {
                uint32_t h01, l01, h23, l23;
                split2_f16(v[0], v[1], h01, l01); split2_f16(v[2], v[3], h23, l23);
                *reinterpret_cast<u32x2*>(dst) = u32x2{h01, h23};
                *reinterpret_cast<u32x2*>(dst + a.plane) = u32x2{l01, l23};
            } else {
                ta4 o = {pj_cvt<TA>(v[0]), pj_cvt<TA>(v[1]), pj_cvt<TA>(v[2]), pj_cvt<TA>(v[3])};
                *reinterpret_cast<ta4*>(dst) = o;
            }
        }
    }
    if (range_on && out_of_range) a.range_flag[t.pair] = 1;
    stamp(3 + 2 * PASS);
}

// NTP = n-tiles per wave per pass: self (768 columns) 3 x 2 passes, cross (512 columns) 2 x 2.  A_PLANE = byte distance
// between the hi and lo planes of the activation tile in LDS; MT = 16-row tiles of the workgroup's row tile (4 or 8).
template <int PREC, class TA, int NTP, int NPASS, int A_PLANE = PJL<PREC>::A_PLANE, int MT = 4, bool AHEAD = false>
__device__ __forceinline__ void proj_compute(const ProjArgs& a, const TileLoc& t, const char* smA, int stamp_base, const RopeRows<MT>* preloaded = nullptr) {
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
                bf[i][j][p] = weight_frag(weight_rsrc(a.W), lane * 16, (p ? a.Nout * 256 * (int)sizeof(typename Tag::elem) : 0) + (pj_tile<NTP>(w, 0, j) * NKC + i) * 1024);
            }
    // the rotary rows are fetched ONCE (the q pass and the k pass rotate by the same rows) and AHEAD of the MFMA loops, behind the first weight
    // fragments in the in-order load queue; the standalone kernel requests them before its x tile (both cold, in flight together), the fused tail
    // has touched them into L2 a GELU step earlier.  Round 3 fetched them at the head of each pass's epilogue: two exposed round trips.
    RopeRows<MT> rr;
    if constexpr (NTP == 3) { if (preloaded) rr = *preloaded; else proj_rope_load<MT>(a, t, rr); }
    __syncthreads();   // the activation tile is complete
    if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + stamp_base + 1] = clock64();
    proj_pass<PREC, TA, NTP, NPASS, 0, A_PLANE, MT, AHEAD>(a, t, smA, bf, stamp_base, rr);
    proj_pass<PREC, TA, NTP, NPASS, 1, A_PLANE, MT, AHEAD>(a, t, smA, bf, stamp_base, rr);
}

// The final projection of the log assignment (ref :289-291) on a 64 x 256 activation tile in LDS (same layout and precondition as
// proj_compute): 256 output columns = 16 n-tiles, wave w owns tiles 2w, 2w + 1 = columns [32w, 32w + 32); TRANSPOSED form, so lane
// (lr, g) ends with keypoint row pj_row(mt, lr) and 4 consecutive columns per tile: one 16-byte fp32 store each, 64 contiguous bytes
// per row over g.  One pass, weight ring 3 chunks ahead (16 VGPRs per chunk).  Used by the standalone kernel (lg_proj.hip) and by the
// last tail (lg_tail.hip, NEXT == 3): the same arithmetic on the same operand planes, so both give bit-identical rows.
template <int PREC, int A_PLANE = PJL<PREC>::A_PLANE, int MT = 4>
__device__ __forceinline__ void final_compute(const FinalArgs& a, const TileLoc& t, const char* smA) {
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int NPART = PJ<PREC>::NPART, APART = PJ<PREC>::APART;
    constexpr int STAGES = PJL<PREC>::STAGES, NKC = 2 * STAGES, TILE = MT * 16 * 128;
    constexpr int NBUF = 4;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const int layer = a.layer_of_pair ? a.layer_of_pair[t.pair] : 0;
    const __amdgpu_buffer_rsrc_t wrs = weight_rsrc(static_cast<const char*>(a.W) + (long long)layer * a.w_layer_bytes);
    const float* bias = a.bias + (long long)layer * 256;
    const int lane16 = lane * 16;
    auto wfrag = [&](int p, int nt, int kc) -> u32x4 {
        return weight_frag(wrs, lane16, (p ? 256 * 256 * (int)sizeof(typename Tag::elem) : 0) + (nt * NKC + kc) * 1024);
    };
    auto load_b = [&](u32x4 (&dst)[2][NPART], int kc) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[j][p] = wfrag(p, 2 * w + j, kc);
    };
    u32x4 bf[NBUF][2][NPART];
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load_b(bf[i], i);
    f32x4 b4[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b4[j] = *reinterpret_cast<const f32x4*>(bias + (2 * w + j) * 16 + 4 * g);
    __syncthreads();   // the activation tile is complete
    f32x4 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int c0 = 0; c0 < NKC; c0 += NBUF) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) {
            const int kc = c0 + i;
            load_b(bf[(i + NBUF - 1) % NBUF], kc + NBUF - 1 < NKC ? kc + NBUF - 1 : NKC - 1);
            __builtin_amdgcn_sched_barrier(0);
            const char* tile = smA + (kc >> 1) * TILE;
            u32x4 af[MT][APART];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < APART; ++p)
                    af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * A_PLANE + pj_tile_off(pj_row<MT>(mt, lr), (kc & 1) * 4 + g));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j) pj_mma<PREC, true>(acc[mt][j], bf[i][j], af[mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (NPART == 2) {
        if (a.planes) {   // split-f16: the rows leave as the hi / lo f16 planes the similarity kernel multiplies (lg_sim.hip) — the same split, of the same fp32 values, that
            // sim_kernel used to redo per K stage in every workgroup; [2][R][256] f16 = the bytes of the fp32 rows.  4 consecutive columns per lane and tile: 8-byte stores
            f16_t* md = reinterpret_cast<f16_t*>(a.out);
            const long long plane = (long long)a.R * 256;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f16_t* dst = md + (t.grow0 + pj_row<MT>(mt, lr)) * 256LL + 32 * w + 4 * g;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 v = (acc[mt][j] + b4[j]) * a.scale;
                    uint32_t h01, l01, h23, l23;
                    split2_f16(v[0], v[1], h01, l01); split2_f16(v[2], v[3], h23, l23);
                    *reinterpret_cast<u32x2*>(dst + 16 * j) = u32x2{h01, h23};
                    *reinterpret_cast<u32x2*>(dst + 16 * j + plane) = u32x2{l01, l23};
                }
            }
            return;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float* dst = a.out + (t.grow0 + pj_row<MT>(mt, lr)) * 256LL + 32 * w + 4 * g;
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4*>(dst + 16 * j) = (acc[mt][j] + b4[j]) * a.scale;
    }
}

}  // namespace lg
