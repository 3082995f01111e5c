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
// Workgroup = 64 keypoint rows, 512 threads = 8 waves; wave w owns output columns {w + 8j} x 16 of the
// hidden layer and [32w, 32w+32) of the output.  The activations are shared by all waves through LDS; every
// wave needs a DIFFERENT slice of the weights, so weights are never staged in LDS: they are pre-packed on the
// host in MFMA-fragment order and each fragment is one fully coalesced 1 KB wave load straight from L2 into
// registers (the weights of a block stay L2-resident; measured L2 -> VGPR ceiling of this pattern: ~50 B/clk/CU).
//
// EVERY product is computed TRANSPOSED, C^T = W a^T (weight fragment = A operand, activation fragment = B operand;
// the two operand layouts of v_mfma_f32_16x16x32 are symmetric, so LDS tiles and packed weights are the same as for
// the plain form).  Lane (lr, g) of a 16x16 tile then holds keypoint ROW lr and the 4 CONSECUTIVE columns 4g..4g+3:
//   * a row's LayerNorm partial is 16 in-lane adds + 2 cross-lane steps (plain form: 16 values x 4 DPP steps),
//   * bias / gamma / beta arrive as one float4 per tile,
//   * GELU output, residual and the next projection's activation tile are 8/16-byte accesses of 4 consecutive
//     elements of one row: no lane-pair transposition, no staging of the output tile through LDS.
//   phase A  h = [x ; ctx] Wcat^T + bcat.  The whole 64 x 512 activation tile lives in LDS (128 KB as split
//            f16): the x half is loaded/converted first, the ctx half is fetched while the x half is being
//            multiplied -> two barriers for the whole phase, weight fragments prefetched from L2 on a ring.
//   LN       per-wave (mean, M2) over its 64 hidden units in registers, ONE exchange through LDS, merged with the
//            parallel-variance formula (exactly a two-pass variance; one barrier pair instead of two)
//   phase B  out = g W2^T + b2 in 4 steps: step j multiplies K-stages 2j, 2j+1 of g (128 hidden units) in 4 k-chunks; the GELU of
//            the wave's NEXT n-tile is dealt to those chunks by row tile, each piece as one VALU block in FRONT of its chunk's MFMA
//            run, written to LDS as 8-byte pieces (one barrier per step).  Closed sched_barrier windows pin the W2 ring 3 chunks and
//            the LDS fragment reads 1 chunk ahead (round 4: hipcc sinks both next to their use otherwise).
//   weights  every fragment stream is a raw buffer load: descriptor + constant per-lane offset + SCALAR byte offset (lg_common.h
//            weight_rsrc) — no VALU address arithmetic in the gaps between MFMA runs, which is where this kernel's time goes
//            (two waves share a fully paced matrix pipe: phase A reaches ~77 % of it with no memory instruction in the loop at all).
//   epilogue residual add and x store straight from the accumulators (16 bytes per lane, 64 contiguous bytes per row); optional
//            256 -> 1 heads on the new rows (token confidence, matchability as sigmoid and as the assignment's log-sigmoid terms)
//   next     (optional, NEXT != 0) the new x tile is ALSO written to LDS in operand precision — into K-stages 0..3 of
//            the g planes, which are dead once every wave has entered step 3, so no barrier is needed in front of it —
//            and the NEXT block's q/k/v projection (lg_proj_body.h; SelfBlock -> this layer's CrossBlock, CrossBlock ->
//            next layer's SelfBlock) runs here: saves that kernel's launch, its x-tile read + conversion and a grid drain.
//            NEXT == 3: the LAST tail of a fixed-depth forward runs the final projection of the log assignment instead.
#include "lg_proj_body.h"
#include <type_traits>

namespace lg {

constexpr int TTHREADS = 512;
#ifndef LG_TAIL_CTX_DMA
#define LG_TAIL_CTX_DMA 1   // ctx half of the activation tile by piecewise LDS-DMA under the x half's MFMAs (A/B switch; 0 = the round-4 register path)
#endif

template <int PREC> struct TT;
template <> struct TT<PREC_F32> { typedef TagF32 Tag; static constexpr int KE = 32, NPART = 1; };
template <> struct TT<PREC_BF16> { typedef TagBF16 Tag; static constexpr int KE = 64, NPART = 1; };
template <> struct TT<PREC_F16> { typedef TagF16 Tag; static constexpr int KE = 64, NPART = 1; };
template <> struct TT<PREC_F16X3> { typedef TagF16 Tag; static constexpr int KE = 64, NPART = 2; };

// LDS map (bytes):  [0, G_BYTES) g tiles  — aliased during phase A by the two staging buffers
//                   [G_BYTES, +RED_BYTES) cross-wave reduction scratch
template <int PREC, int MT = 4> struct TL {
    static constexpr int STAGES = 512 / TT<PREC>::KE;               // K stages of the 512-long contractions
    static constexpr int TILE = MT * 16 * 128;                      // one plane of one stage: MT*16 rows x 128 B
    static constexpr int G_PLANE = STAGES * TILE;                   // 64 KB (16-bit) / 128 KB (f32)
    static constexpr int G_BYTES = TT<PREC>::NPART * G_PLANE;
    static constexpr int RED_BYTES = 8 * MT * 16 * 8;               // (mean, M2) per row per wave
    static constexpr int TOTAL = G_BYTES + RED_BYTES;
};

// GELU(u) = 0.5 u (1 + erf(u / sqrt 2)) = 0.5 u + 0.5 |u| (1 - erfc(|u| / sqrt 2)), with erfc(a / sqrt 2) = exp2(-a Q(a)):
// Q = degree-7 weighted minimax fit of -log2(erfc(a / sqrt 2)) / a on [0, 7] (tools/fit_gelu.py; fit error of the GELU 9e-9, Q >= 1.15 for
// every a >= 0, so large |u| runs into exp2(-inf) = 0 and never into a positive exponent).  Evaluated in fp32: max |GELU error| 3.2e-7 at
// |u| = 4.6 (0.7 ulp of the result), 1.1e-7 for |u| < 2 — the same as the Abramowitz-Stegun form it replaces (3.0e-7 / 1.2e-7), for 15
// instead of 26 instructions per pair and ONE transcendental per value instead of two (no reciprocal).  The ocml erff costs ~10x more
// instructions (two divergent ranges).  Two values at a time so that the polynomial runs on v_pk_fma_f32 / v_pk_mul_f32.
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 u) {
    const f32x2 a = {fabsf(u[0]), fabsf(u[1])};
    f32x2 p = a * -1.902015583254979e-06f + 2.8056274459231645e-05f;   // -Q(a), Horner
    p = p * a + -0.0001314696710323915f;
    p = p * a + -0.00027208542451262474f;
    p = p * a + 0.00724543584510684f;
    p = p * a + -0.052627623081207275f;
    p = p * a + -0.4591621458530426f;
    p = p * a + -1.1511110067367554f;
    const f32x2 arg = p * a;
    const f32x2 e = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};   // erfc(|u| / sqrt 2)
    const f32x2 wgt = e * -0.5f + 0.5f;                                                  // 0.5 erf(|u| / sqrt 2)
    const f32x2 half_u = u * 0.5f;
    return a * wgt + half_u;
}
__device__ __forceinline__ f32x2 gelu_pair(f32x2 u) { return gelu_fast2(u); }

// acc (C^T tile) += w x^T for one k-chunk; split-f16: (w_hi x_lo) + (w_lo x_hi) + (w_hi x_hi)
template <int PREC>
__device__ __forceinline__ void tail_mma(f32x4& acc, const u32x4* wf, const u32x4* xf) {
    typedef typename TT<PREC>::Tag Tag;
    if constexpr (TT<PREC>::NPART == 2) {
        mma_chunk<Tag>(acc, wf[0], xf[1]);   // hi * lo
        mma_chunk<Tag>(acc, wf[1], xf[0]);   // lo * hi
        mma_chunk<Tag>(acc, wf[0], xf[0]);   // hi * hi
    } else {
        mma_chunk<Tag>(acc, wf[0], xf[0]);
    }
}

// NEXT: 0 = plain tail, 1 = + SelfBlock projection of the next layer (768 columns, rotary), 2 = + CrossBlock
// projection (512 columns), 3 = + the final projection of the log assignment (256 columns, fp32 out; the last block of a fixed-depth forward).  TA = element type of q/k/v (attention operand precision), only read when NEXT != 0.
// MT = 16-row tiles per workgroup: 4 (64 rows, the throughput shape) or 2 / 1 for under-filled grids (small batches): the
// same per-row arithmetic in the same order — outputs are bit-identical — on 2x / 4x as many workgroups; each of them streams
// the full weight set, so these shapes are L2-stream-bound per CU (~46k cycles) instead of matrix-bound.
// ASPLIT (PREC_F16X3 with NEXT != 0 only): the next projection feeds the SPLIT attention — the new x tile goes to LDS as hi + lo
// f16 planes, the projection runs three MFMAs per product and writes q / k / v as hi + lo planes (lg_proj_body.h PREC_F16X3);
// otherwise one f16 plane and two MFMAs per product (PREC_QKV_F16W2, attention precision fp16).
template <int PREC, int NEXT, class TA, int MT, bool ASPLIT>
__global__ __launch_bounds__(TTHREADS) void tail_kernel(TailArgs a) {
    constexpr int TBM = 16 * MT;
    typedef typename TT<PREC>::Tag Tag;
    constexpr int EPC = Tag::EPC, KE = TT<PREC>::KE, NPART = TT<PREC>::NPART;
    constexpr int STAGES = TL<PREC, MT>::STAGES, TILE = TL<PREC, MT>::TILE, G_PLANE = TL<PREC, MT>::G_PLANE;
    constexpr int NKC = 2 * STAGES;          // 16-byte k-chunks per row (16 for 16-bit, 32 for f32)
    constexpr int NV = EPC / 4;              // float4 loads per staged chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem + TL<PREC, MT>::G_BYTES);

    // workgroup -> row tile: round-robin with the residue rotated per group of 8 (rr_rotate, lg_common.h).  Round-3 A/B against the identity
    // map, one box (profiles/r03q_tail_tile_maps.md): cfg #2 +2.3 % (tail -3.7 %), cfg #3' +1.1 %, cfg #5 +4.4 % / +8.7 %.
    const TileLoc t = locate_tile(a.rs, rr_rotate(blockIdx.x, gridDim.x), TBM);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    // optional per-phase shader-clock stamps (profiling tap; a.dbg == nullptr in production)
    auto stamp = [&](int slot) {
        if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + slot] = clock64();
    };
    stamp(0);
    if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + 6] = wall_clock64();   // 100 MHz wall clock: occupancy timeline (tools/tail_wall.py)
    // guide T5, static form: the second-dispatched half of an 8-wave workgroup loses issue arbitration to the older half on
    // every phase; one priority bump for it, no per-cluster flips (measured: tail -0.2 ... -1.3 %)
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);

    // ------------------------------------------------------------------ phase A
    // the accumulators START at the folded bias (lane (lr, g) holds hidden units 4g .. 4g + 3 of its n-tiles for every row): the loads hide
    // under the x tile's, and the LayerNorm phase no longer opens with an exposed L2 round trip for them
    f32x4 acc[MT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bcat + (w + 8 * j) * 16 + 4 * g);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][j] = b4;
    }

    // weight fragment: plane p, n-tile nt, k-chunk kc -> 64 lanes x 16 B contiguous
    // by BUFFER load (lg_common.h weight_rsrc): descriptor + constant per-lane offset + scalar byte offset — no VALU address arithmetic between MFMA runs
    const __amdgpu_buffer_rsrc_t Wc = weight_rsrc(a.Wcat), W2 = weight_rsrc(a.W2);
    const int lane16 = lane * 16;
    auto wfrag = [&](__amdgpu_buffer_rsrc_t base, int p, long long plane_elems, int nt, int kc) -> u32x4 {
        return weight_frag(base, lane16, (p ? (int)(plane_elems * (long long)sizeof(typename Tag::elem)) : 0) + (nt * NKC + kc) * 1024);
    };

    // ---- activation tile -> LDS.  Half hf (0: x, 1: ctx) = TBM rows x 256 floats = STAGES/2 K-stage tiles.
    // A stage tile is TBM rows x 8 chunks of 16 bytes = 128 MT threads' worth, so the 512 threads cover 4 / MT stage tiles
    // per round: thread -> (stage = round * 4/MT + tid / (128 MT), row = (tid >> 3) % TBM, chunk slot = tid & 7).
    constexpr int HS = STAGES / 2;                    // stage tiles per half (4, or 8 for f32)
    constexpr int SPR = 4 / MT, ROUNDS = HS / SPR;    // stage tiles per round, rounds per half
    const int srow = (tid >> 3) & (TBM - 1), sslot = tid & 7, sst = tid / (128 * MT);
    f32x4 hregs[2][ROUNDS][NV];   // [0] x rows, [1] ctx rows
    auto load_half = [&](int hf) {
        const float* src = (hf ? a.CTX : a.X) + (long long)(t.grow0 + srow) * 256 + sslot * EPC;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i)
#pragma unroll
            for (int j = 0; j < NV; ++j) hregs[hf][i][j] = *reinterpret_cast<const f32x4*>(src + (i * SPR + sst) * KE + 4 * j);
    };
    auto store_half = [&](int hf) {
        const int off = lds_off<128>(srow, sslot);
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            char* tile = smem + (hf * HS + i * SPR + sst) * TILE;   // planes: p * G_PLANE
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(tile + off) = hregs[hf][i][0];
            } else if constexpr (NPART == 2) {
                u32x4 hi, lo;
                split8<Tag>(hregs[hf][i][0], hregs[hf][i][1], hi, lo);
                *reinterpret_cast<u32x4*>(tile + off) = hi;
                *reinterpret_cast<u32x4*>(tile + G_PLANE + off) = lo;
            } else {
                *reinterpret_cast<u32x4*>(tile + off) = pack8<Tag>(hregs[hf][i][0], hregs[hf][i][1]);
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
    // activation fragments of one k-chunk (4 row tiles x planes) from the LDS-resident tile
    auto read_af = [&](u32x4 (&af)[MT][NPART], int kc) {
        const char* tile = smem + (kc >> 1) * TILE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < NPART; ++p)
                af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * G_PLANE + lds_off<128>(mt * 16 + lr, (kc & 1) * 4 + g));
    };
    auto mma_A_rows = [&](const u32x4 (&af)[MT][NPART], const u32x4 (&b)[4][NPART], int mt0, int mt1) {
#pragma unroll
        for (int mt = mt0; mt < mt1; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) tail_mma<PREC>(acc[mt][nt], b[nt], af[mt]);
    };
    auto mma_A = [&](const u32x4 (&af)[MT][NPART], const u32x4 (&b)[4][NPART]) { mma_A_rows(af, b, 0, MT); };
    load_half(0);
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load_b_A(bf[i], i);
    __builtin_amdgcn_sched_barrier(0);
    store_half(0);
    __syncthreads();
    // ctx rows stream in while the x half is multiplied.  (What hipcc makes of it, round-4 ISA: it hoists the split arithmetic of store_half(1) up
    // to here and waits for these loads in front of the loop.  Forcing the conversion back behind the loop — registers laundered through an empty
    // asm — and / or requesting the ctx rows together with the x rows measured -1.4 % / +-0: the in-order load counter makes chunk 1's weight
    // fragments wait for the ctx rows either way, and the converted planes are 16 registers instead of 32 through the loop.  LAB_NOTES.md.)
    // CTX_DMA (round 5; split-f16, 64-row tiles): the ctx half is NOT requested here.  Stamps inside phase A (profiles/r05c_phaseA_diag.log) showed
    // what the round-4 form costs: x published 9.7k cycles after the kernel starts, then 3.7k more in front of the loop until the 64 KB of ctx rows have
    // arrived — a CU pulls HBM misses at ~10 B/clk whatever the other CUs do (profiles/r05d_stagger_diag.log) — while both MFMA loops run at 80 - 90 % of the
    // matrix pipe.  Now every wave moves ITS OWN 8 rows of ctx (8 KB) by LDS-DMA, one 1 KB piece per chunk of the x half, issued BEHIND that chunk's weight
    // fragments (placement: see the loop).  The
    // pieces land as fp32 in the wave's own eight 1 KB pieces of K-stages 4..7 (its rows of the hi and of the lo plane: exactly the bytes its operand rows
    // will occupy), so the conversion at the half boundary is wave-private and IN PLACE: read all 32 floats per lane, then write hi / lo — no staging
    // area, no staging registers through the loop (32 VGPRs), one barrier as before.  The DMA is the inline-asm form (lg_common.h lds_dma16): through the
    // builtin hipcc orders every later ds_read behind it with vmcnt(0) lgkmcnt(0) — a full round trip in front of every chunk (round-5 ISA).
    constexpr bool CTX_DMA = LG_TAIL_CTX_DMA && PREC == PREC_F16X3 && MT == 4;
    if constexpr (!CTX_DMA) load_half(1);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int HC = NKC / 2;         // k-chunks per half
    // piece j = (K-stage 4 + (j >> 1), row group j & 1): lane l fetches 16 bytes of row 8w + 4 (j & 1) + (l >> 4), columns 64 (j >> 1) + 4 (l & 15) ..
    const float* ctx_lane = a.CTX + (long long)(t.grow0 + 8 * w + (lane >> 4)) * 256 + (lane & 15) * 4;
    auto ctx_dma = [&](int s, int q) {
        if constexpr (CTX_DMA) lds_dma16(ctx_lane + q * 1024 + s * 64, smem + (HS + s) * TILE + q * G_PLANE + w * 1024);
    };
    auto ctx_convert_in_place = [&]() {       // lane l: row 8w + (l >> 3), eight floats 8 (l & 7) .. of each of the four K-stages
        const int r8 = lane >> 3, slot = lane & 7;
        const char* src = smem + HS * TILE + (r8 >> 2) * G_PLANE + w * 1024 + 256 * (r8 & 3) + 32 * slot;
        f32x4 st[4][2];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { st[s2][0] = *reinterpret_cast<const f32x4*>(src + s2 * TILE); st[s2][1] = *reinterpret_cast<const f32x4*>(src + s2 * TILE + 16); }
        __builtin_amdgcn_sched_barrier(0);    // every read of the wave's staging bytes is issued before the first write (LDS executes a wave's accesses in order)
        char* dst = smem + HS * TILE + lds_off<128>(8 * w + r8, slot);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            u32x4 hi, lo;
            split8<Tag>(st[s2][0], st[s2][1], hi, lo);
            *reinterpret_cast<u32x4*>(dst + s2 * TILE) = hi;
            *reinterpret_cast<u32x4*>(dst + s2 * TILE + G_PLANE) = lo;
        }
    };
    // The activation fragments of chunk kc + 1 are read from LDS BEFORE the MFMAs of chunk kc (two register sets): with
    // the wave index provably uniform (SGPR address parts) the kernel has the 32 VGPRs for it, and the ~200-cycle LDS round
    // trip at the head of every chunk — which both waves of a SIMD hit at the same time — disappears.  The prefetch stays
    // inside a half (the other half is not in LDS yet); a half's first chunk reads its own.
    u32x4 afr[2][MT][NPART];
    auto run_half = [&](auto HF) {            // two copies of the loop (no runtime branch around the DMA issue: hipcc merges wait states conservatively at joins)
        constexpr int hf = decltype(HF)::value;
        read_af(afr[0], hf * HC);
#pragma unroll 1
        for (int c0 = 0; c0 < HC; c0 += NBUF) {
#pragma unroll
            for (int i = 0; i < NBUF; ++i) {
                const int kc = hf * HC + c0 + i;
                load_b_A(bf[(i + NBUF - 1) % NBUF], kc + NBUF - 1 < NKC ? kc + NBUF - 1 : NKC - 1);
                read_af(afr[(i + 1) & 1], c0 + i + 1 < HC ? kc + 1 : kc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (CTX_DMA && hf == 0) {
                    // The piece is issued BEHIND the chunk's last fragment wait (all 8 fragments are consumed by the first row tile's MFMAs).  hipcc does not
                    // count the asm DMA: its waits vmcnt(15 - f) for fragment f of this chunk are one too strict from here on, i.e. they also cover the
                    // OLDEST operation behind that fragment — which with this placement is the piece of the PREVIOUS chunk (most of a chunk old), never a
                    // fragment or a piece that was just issued (any earlier placement makes every chunk wait for a round trip; ISA-checked).
                    mma_A_rows(afr[i & 1], bf[i], 0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    ctx_dma((c0 + i) >> 1, (c0 + i) & 1);          // HC = 8 chunks = 8 pieces
                    __builtin_amdgcn_sched_barrier(0);
                    mma_A_rows(afr[i & 1], bf[i], 1, MT);
                } else {
                    mma_A(afr[i & 1], bf[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    run_half(std::integral_constant<int, 0>{});
    if constexpr (CTX_DMA) {
        static_assert(!CTX_DMA || HC == 8, "one ctx piece per chunk of the x half");
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces have landed (the last one was issued a chunk ago, behind the fragments of chunk 8)
        __builtin_amdgcn_sched_barrier(0);
        ctx_convert_in_place();
    } else {
        store_half(1);
    }
    __syncthreads();
    run_half(std::integral_constant<int, 1>{});
    stamp(1);
    // ------------------------------------------------------------------ LayerNorm(512) (the bias is already in the accumulators)
    // acc[mt][nt][r] = h[row mt*16 + lr][hidden (w + 8 nt)*16 + 4g + r]
    {
        f32x4 gam[4], bet[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = (w + 8 * nt) * 16 + 4 * g;
            gam[nt] = *reinterpret_cast<const f32x4*>(a.gamma + col); bet[nt] = *reinterpret_cast<const f32x4*>(a.beta + col);
        }
        // this wave's 64 hidden units of row (mt, lr): local mean and M2 = sum (h - mean)^2, entirely in registers
        f32x2* red2 = reinterpret_cast<f32x2*>(red);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float sacc = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) sacc += (acc[mt][nt][0] + acc[mt][nt][1]) + (acc[mt][nt][2] + acc[mt][nt][3]);
            sacc = xor32_sum(xor16_sum(sacc));
            const float ml = sacc * (1.f / 64.f);
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[mt][nt][r] - ml; q += d * d; }
            q = xor32_sum(xor16_sum(q));
            if (g == 0) red2[(mt * 16 + lr) * 8 + w] = f32x2{ml, q};
        }
        __syncthreads();   // also: every wave is past its last read of the activation tile -> g may overwrite it below
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            // merge the 8 (count 64, mean, M2) triples of the row: mean = avg(mean_w), M2 = sum M2_w + 64 sum (mean_w - mean)^2
            const f32x4* pr = reinterpret_cast<const f32x4*>(red2 + (mt * 16 + lr) * 8);
            const f32x4 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
            const float mean = (((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) * 0.125f;
            float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]));
            float dm = 0.f;
            { float d;
              d = p0[0] - mean; dm += d * d; d = p0[2] - mean; dm += d * d; d = p1[0] - mean; dm += d * d; d = p1[2] - mean; dm += d * d;
              d = p2[0] - mean; dm += d * d; d = p2[2] - mean; dm += d * d; d = p3[0] - mean; dm += d * d; d = p3[2] - mean; dm += d * d; }
            m2 += 64.f * dm;
            const float rstd = __builtin_amdgcn_rsqf(m2 * (1.f / 512.f) + 1e-5f);   // v_rsq_f32, 1 ulp
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = (acc[mt][nt][r] - mean) * rstd * gam[nt][r] + bet[nt][r];   // pre-GELU
        }
    }
    stamp(2);
    // ------------------------------------------------------------------ GELU + g -> LDS (B-operand tiles) + phase B
    // n-tile j of wave w = hidden units [(w + 8j)*16, +16) = K-stage 2j + (w >> 2), columns (w & 3)*16 + 4g + r of it:
    // a lane's 4 values are 4 consecutive k of one row = one 8-byte piece per plane (f32: one 16-byte piece).
    auto gelu_store = [&](int j, int mt0 = 0, int mt1 = MT) {
#pragma unroll
        for (int mt = mt0; mt < mt1; ++mt) {
            const f32x2 v01 = gelu_pair(f32x2{acc[mt][j][0], acc[mt][j][1]});
            const f32x2 v23 = gelu_pair(f32x2{acc[mt][j][2], acc[mt][j][3]});
            const int row = mt * 16 + lr;
            if constexpr (EPC == 8) {
                char* dst = smem + (2 * j + (w >> 2)) * TILE + lds_off<128>(row, (w & 3) * 2 + (g >> 1)) + (g & 1) * 8;
                if constexpr (PREC == PREC_F16X3) {
                    uint32_t h01, l01, h23, l23;
                    split2_f16(v01[0], v01[1], h01, l01); split2_f16(v23[0], v23[1], h23, l23);
                    *reinterpret_cast<u32x2*>(dst) = u32x2{h01, h23};
                    *reinterpret_cast<u32x2*>(dst + G_PLANE) = u32x2{l01, l23};
                } else {
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<Tag>(v01[0], v01[1]), pack2<Tag>(v23[0], v23[1])};
                }
            } else {
                const int hcol = (w + 8 * j) * 16 + 4 * g;   // f32: K-stage = 32 hidden units
                *reinterpret_cast<f32x4*>(smem + (hcol >> 5) * TILE + lds_off<128>(row, (hcol & 31) >> 2)) = f32x4{v01[0], v01[1], v23[0], v23[1]};
            }
        }
    };
    // acc2[mt][nt][r] = out[row mt*16 + lr][column w*32 + nt*16 + 4g + r]
    f32x4 acc2[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc2[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    u32x4 b2f[4][2][NPART];   // ring, 3 k-chunks ahead
    auto load_b_B = [&](u32x4 (&dst)[2][NPART], int kc) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < NPART; ++p) dst[nt][p] = wfrag(W2, p, 256LL * 512, w * 2 + nt, kc);
    };
    // activation (g) fragments of one k-chunk, read one chunk AHEAD of their MFMAs inside a step (two register sets, as in phase A; round-4 ISA: read
    // right in front of the MFMAs they cost every chunk an exposed LDS round trip).  A step's first chunk reads its own: the tile it needs is
    // published by the barrier in front of it.
    u32x4 afB[2][MT][NPART];
    auto read_afB = [&](u32x4 (&af)[MT][NPART], int kc) {
        const char* tile = smem + (kc >> 1) * TILE;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < NPART; ++p)
                af[mt][p] = *reinterpret_cast<const u32x4*>(tile + p * G_PLANE + lds_off<128>(mt * 16 + lr, (kc & 1) * 4 + g));
    };
    auto mma_B = [&](const u32x4 (&af)[MT][NPART], const u32x4 (&b)[2][NPART]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) tail_mma<PREC>(acc2[mt][nt], b[nt], af[mt]);
    };
    load_b_B(b2f[0], 0); load_b_B(b2f[1], 1); load_b_B(b2f[2], 2);
    // The fused SelfBlock projection rotates q and k by the rotary rows of this tile (64 rows x 128 B per table, cold).  Touch every line now — behind
    // the first W2 fragments in the in-order load queue, a whole GELU step before anything younger is waited for — so that the projection's fetch
    // (lg_proj_body.h proj_rope_load) finds them in L2.  No branch around the load.
    float rope_pf = 0.f;
    if constexpr (NEXT == 1) rope_pf = (((tid >> 6) & 1) ? a.next.sinb : a.next.cosb)[(long long)(t.grow0 + (tid & 63)) * 32 + (tid >> 7) * 8];
    gelu_store(0);
    __syncthreads();
    stamp(3);
    constexpr int CPS = NKC / 4;   // k-chunks per step (4 for 16-bit: K-stages 2j, 2j+1; 8 for f32)
    const int qlen = a.rs.len[t.seg];
    f32x4 xres[MT][2];             // residual rows, same (row, 4 columns) per lane as acc2
    // the epilogue's operands — output bias, optional head weights (token confidence / matchability: this lane's 8 columns of each weight vector)
    // — are fetched with the residual rows, under the last step's MFMAs (round 3 issued them at the head of the epilogue: one exposed round trip)
    f32x4 b2v[2];
    const bool heads = a.head_w0 != nullptr;                         // workgroup-uniform
    f32x4 hw0[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, hw1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j == 3) {              // issue the residual loads so that they land under the last step's MFMAs
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    xres[mt][nt] = *reinterpret_cast<const f32x4*>(a.X + (long long)(t.grow0 + mt * 16 + lr) * 256 + w * 32 + nt * 16 + 4 * g);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b2v[nt] = *reinterpret_cast<const f32x4*>(a.b2 + w * 32 + nt * 16 + 4 * g);
            {   // NOT branched around (a conditional load makes hipcc wait vmcnt(0) at the join — here: for the residual rows just requested, in
                // front of the last step's MFMAs, in every CrossBlock tail of the adaptive path): absent heads read the output bias instead
                const float* w0p = heads ? a.head_w0 : a.b2;
                const float* w1p = a.head_w1 ? a.head_w1 : a.b2;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    hw0[nt] = *reinterpret_cast<const f32x4*>(w0p + w * 32 + nt * 16 + 4 * g);
                    hw1[nt] = *reinterpret_cast<const f32x4*>(w1p + w * 32 + nt * 16 + 4 * g);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CPS; ++i) {
            const int kc = j * CPS + i;
            // j and i are unrolled constants: a plain `if` is resolved at compile time (no join, no conservative wait)
            // The ring is PINNED where the source puts it (round 4).  Left to itself hipcc issued a chunk's W2 fragments ~1 chunk (last step: 0 - 7
            // MFMAs) ahead of their MFMAs instead of 3 (tools/isa_wait_distance.py); sched_barrier masks that let the GELU arithmetic float did
            // not hold the loads (the MFMAs moved instead), so the windows are closed and the GELU of the next n-tile is dealt to them by hand:
            // row tile i of n-tile j + 1 with chunk i.  cfg #2 +0.7 %, cfg #5' +2 % (profiles/r04e_*, r04f_*).
            if (i == 0) read_afB(afB[0], kc);
            __builtin_amdgcn_sched_barrier(0);
            if (kc + 3 < NKC) load_b_B(b2f[(kc + 3) & 3], kc + 3);
            if (i + 1 < CPS) read_afB(afB[(i + 1) & 1], kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            // the chunk's GELU piece as ONE block in FRONT of its MFMA run (round 4, call p: left in the same window hipcc sprinkles 3 - 5 VALU between every
            // two MFMAs; as a block behind the run +-0, in front of it tail -1.1 %: a wave's VALU block then meets its partner's MFMA run)
            if (j < 3) gelu_store(j + 1, i * MT / CPS, (i + 1) * MT / CPS);
            __builtin_amdgcn_sched_barrier(0);
            mma_B(afB[i & 1], b2f[kc & 3]);
        }
        if (j < 3) __syncthreads();
    }
    stamp(4);
    // ------------------------------------------------------------------ epilogue: + b2, + x, store; next block's activation tile
    // NEXT: the tile goes to K-stages 0..3 of the g planes (hi at 0, lo at G_PLANE): every wave is past the barrier that
    // ended step 2, so nobody reads those stages any more (step 3 reads stages 6, 7) — no barrier needed here.
    float hp0[MT], hp1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { hp0[mt] = 0.f; hp1[mt] = 0.f; }
    // range guard (LG_FLAG_CHECK_FINITE; workgroup-uniform switch): the new residual rows are what the next kernels split into f16 planes
    const bool range_on = a.range_flag != nullptr;
    bool out_of_range = false;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = w * 32 + nt * 16 + 4 * g;
        const f32x4 b2 = b2v[nt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + lr;
            const f32x4 xn = xres[mt][nt] + (acc2[mt][nt] + b2);
            if (t.r0 + row < qlen) {
                *reinterpret_cast<f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + col) = xn;
                if (range_on) out_of_range |= !(fabsf(xn[0]) < 65504.f) | !(fabsf(xn[1]) < 65504.f) | !(fabsf(xn[2]) < 65504.f) | !(fabsf(xn[3]) < 65504.f);
            }
            if (heads) {
                hp0[mt] += (xn[0] * hw0[nt][0] + xn[1] * hw0[nt][1]) + (xn[2] * hw0[nt][2] + xn[3] * hw0[nt][3]);
                hp1[mt] += (xn[0] * hw1[nt][0] + xn[1] * hw1[nt][1]) + (xn[2] * hw1[nt][2] + xn[3] * hw1[nt][3]);
            }
            if constexpr (NEXT != 0) {
                static_assert(EPC == 8, "fused next projection: 16-bit operands only");
                char* dst = smem + (col >> 6) * TILE + pj_tile_off(row, (col & 63) >> 3) + (col & 7) * 2;   // the projection's own swizzle (lg_proj_body.h)
                if constexpr (NPART == 2 && (ASPLIT || NEXT == 3)) {   // hi + lo planes of the new x tile (the final projection always takes both) (lo at G_PLANE: K-stages 0..3 of the lo g plane are just as dead)
                    uint32_t h01, l01, h23, l23;
                    split2_f16(xn[0], xn[1], h01, l01); split2_f16(xn[2], xn[3], h23, l23);
                    *reinterpret_cast<u32x2*>(dst) = u32x2{h01, h23};
                    *reinterpret_cast<u32x2*>(dst + G_PLANE) = u32x2{l01, l23};
                } else if constexpr (NPART == 2) {      // the projection takes ONE f16 plane (PREC_QKV_F16W2, lg_proj_body.h)
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2_f16(xn[0], xn[1]), pack2_f16(xn[2], xn[3])};
                } else {
                    *reinterpret_cast<u32x2*>(dst) = u32x2{pack2<Tag>(xn[0], xn[1]), pack2<Tag>(xn[2], xn[3])};
                }
            }
        }
    }
    if (range_on && out_of_range) a.range_flag[t.pair] = 1;   // (same value from every lane that sees one: no atomic needed)
    if (heads) {   // reduce over the 4 lane groups (the other column quarters of this wave), then over the 8 waves through LDS, in a fixed order
        f32x2* red2 = reinterpret_cast<f32x2*>(red);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float s0 = xor32_sum(xor16_sum(hp0[mt])), s1 = xor32_sum(xor16_sum(hp1[mt]));
            if (g == 0) red2[(mt * 16 + lr) * 8 + w] = f32x2{s0, s1};
        }
        __syncthreads();
        if (tid < TBM && t.r0 + tid < qlen) {
            const f32x4* pr = reinterpret_cast<const f32x4*>(red2 + tid * 8);
            const f32x4 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
            // sigmoid for the adaptive decisions; logsigmoid(z) / logsigmoid(-z) = the matchability terms of the log assignment (ref :268-276)
            auto emit = [&](float z, float* sig, float* ls, float* lsneg) {
                if (sig) sig[t.grow0 + tid] = 1.f / (1.f + expf(-z));
                if (ls || lsneg) {
                    const float sp = log1pf(expf(-fabsf(z)));
                    if (ls) ls[t.grow0 + tid] = fminf(z, 0.f) - sp;
                    if (lsneg) lsneg[t.grow0 + tid] = fminf(-z, 0.f) - sp;
                }
            };
            emit((((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) + a.head_b0[0], a.head_out0, a.head_ls0, a.head_lsneg0);
            if (a.head_w1) emit((((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]))) + a.head_b1[0], a.head_out1, a.head_ls1, a.head_lsneg1);
        }
    }
    stamp(5);
    asm volatile("" :: "v"(rope_pf));   // keeps the touch alive (and waits for it here at the latest)
    if constexpr (NEXT == 3) final_compute<PREC, G_PLANE, MT>(a.fin, t, smem);
    else if constexpr (NEXT != 0) proj_compute<(prec_is_split(PREC) ? (ASPLIT ? PREC_F16X3 : PREC_QKV_F16W2) : PREC), TA, NEXT == 1 ? 3 : 2, 2, G_PLANE, MT, true>(a.next, t, smem, 0);
    if (a.dbg && lane == 0)   // wall clock at the end + where the workgroup ran (HW_ID, XCC_ID)
        a.dbg[((long long)blockIdx.x * 8 + w) * 8 + 7] = (wall_clock64() & ((1LL << 44) - 1)) | ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFF00) << 40) | ((long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 44);
}

template <int PREC, int NEXT, class TA, int MT, bool ASPLIT> static hipError_t launch_tail_m(const TailArgs& a, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    auto kern = tail_kernel<PREC, NEXT, TA, MT, ASPLIT>;
    constexpr int smem = TL<PREC, MT>::TOTAL;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / (16 * MT)), dim3(TTHREADS), smem, s, a);
    return hipGetLastError();
}
template <int PREC, int NEXT, class TA, bool ASPLIT> static hipError_t launch_tail_t(const TailArgs& a, hipStream_t s) {
    if constexpr (PREC == PREC_F32) return launch_tail_m<PREC, NEXT, TA, 4, false>(a, s);   // (the exact mode keeps one shape)
    else {
        if (a.row_tiles == 1 && !a.dbg) return launch_tail_m<PREC, NEXT, TA, 1, ASPLIT>(a, s);
        if (a.row_tiles == 2 && !a.dbg) return launch_tail_m<PREC, NEXT, TA, 2, ASPLIT>(a, s);
        return launch_tail_m<PREC, NEXT, TA, 4, ASPLIT>(a, s);
    }
}
template <int PREC, class TA, bool ASPLIT> static hipError_t launch_tail_next(const TailArgs& a, hipStream_t s) {
    if (a.fin.W) return a.next.W ? hipErrorInvalidValue : launch_tail_t<PREC, 3, TA, false>(a, s);
    if (!a.next.W) return launch_tail_t<PREC, 0, TA, false>(a, s);
    if (ASPLIT && a.next.plane <= 0) return hipErrorInvalidValue;
    if (a.next.Nout == 768 && a.next.cosb) return launch_tail_t<PREC, 1, TA, ASPLIT>(a, s);
    if (a.next.Nout == 512 && !a.next.cosb) return launch_tail_t<PREC, 2, TA, ASPLIT>(a, s);
    return hipErrorInvalidValue;
}

// the fused next projection exists for the 16-bit operand / attention combinations the engine runs
bool launch_tail_supports_next(int prec, int attn_prec) {
    return (prec == PREC_F16X3 && (attn_prec == PREC_F16X3 || attn_prec == PREC_F16)) || (prec == PREC_BF16 && attn_prec == PREC_BF16) ||
           (prec == PREC_F16 && attn_prec == PREC_F16);
}

hipError_t launch_tail(int prec, int attn_prec, const TailArgs& a, hipStream_t s) {
    if ((a.next.W || a.fin.W) && !launch_tail_supports_next(prec, attn_prec)) return hipErrorInvalidValue;
    switch (prec) {
        case PREC_F32: return launch_tail_t<PREC_F32, 0, float, false>(a, s);
        case PREC_BF16: return launch_tail_next<PREC_BF16, bf16_t, false>(a, s);
        case PREC_F16: return launch_tail_next<PREC_F16, f16_t, false>(a, s);
        case PREC_F16X3: return attn_prec == PREC_F16X3 ? launch_tail_next<PREC_F16X3, f16_t, true>(a, s) : launch_tail_next<PREC_F16X3, f16_t, false>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace lg
