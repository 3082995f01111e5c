// lightglue_amd — flash-style attention for the matcher (ref lightglue.py:113-137 Attention,
// used by SelfBlock :170 and both directions of CrossBlock :211-214 / :216-223).
//   head_dim 64, 4 heads, non-causal, q_len != kv_len (cross), ragged per-segment lengths.
//
// Workgroup = 4 waves = 128 query rows of one (segment, head); K and V^T tiles of 64 keys are
// staged through LDS (register hop, next tile's loads in flight during the MFMAs) and shared by
// the 4 waves.  Each wave owns 32 query rows (two 16-row MFMA tiles).
//
// "Swapped" formulation (guide T12 idea, 16x16 tiles): S^T = K Q^T and O^T = V^T P^T, so that a
// query row lives in ONE lane column (lane & 15): the online-softmax max/sum run over a lane's own
// registers plus two cross-lane steps (lane groups g = 0..3), the rescale factor is a per-lane
// scalar, and P^T leaves the S^T accumulators already in B-operand order for the PV MFMA:
//   f32:    S^T acc[kt][qt][r] <-> key 16*kt + 4*g + r;  PV chunk kt: B element i <-> key 16*kt + 4*g + i
//   16-bit: the K rows feeding S^T tile kt are PERMUTED (a free choice: which K row a lane reads) so that the 8
//           probabilities a lane group packs for PV chunk tp are 8 CONSECUTIVE keys:
//           S^T acc[kt][qt][r] <-> key 32*(kt>>1) + 8*g + 4*(kt&1) + r,   PV chunk tp: B element j <-> key 32*tp + 8*g + j
// V is produced TRANSPOSED by the projection ([head][d][row]) so the matching A operand is ONE contiguous 16-byte
// LDS read per fragment in both cases (the natural order needed two 8-byte reads + register shuffles).
// The S matrix never touches HBM.  q and k arrive pre-multiplied by sqrt(log2(e) / sqrt(64)) (lg_proj_body.h QK_PRESCALE, the
// reference's CPU path splits its scale the same way, lightglue.py:215), so S is already the base-2 logit: softmax runs in fp32
// with bare exp2, and in the LDS-DMA kernels the score accumulators START at -m_run so that exp2 applies to the MFMA result.
#include "lg_kernels.h"

namespace lg {

constexpr int ABK = 64, ATHREADS = 256;   // query rows per workgroup = 64 * QT (QT 16-row query tiles per wave, 4 waves)

template <class Tag>
__device__ __forceinline__ u32x4 mask_tail(u32x4 v, int nvalid) {  // keep the first nvalid elements of the chunk
    constexpr int EPC = Tag::EPC;
    if (nvalid >= EPC) return v;
    if (nvalid <= 0) return u32x4{0u, 0u, 0u, 0u};
    if constexpr (EPC == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i >= nvalid) v[i] = 0u;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (2 * i >= nvalid) v[i] = 0u;
            else if (2 * i + 1 >= nvalid) v[i] &= 0xFFFFu;
        }
    }
    return v;
}

// K tile swizzle.  16-bit: a 16-lane ds_read_b128 group reads rows {8j + 4h + i : j, i = 0..3} (permuted key order), so
// the slot XOR is built from row bits 1, 3, 4 (8 values x 2 row parities = all 64 banks); f32 keeps lds_off<256>.
template <int ROWB> __device__ __forceinline__ int k_off(int row, int slot16) {
    if constexpr (ROWB == 128) return row * 128 + ((slot16 ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1))) << 4);
    else return lds_off<ROWB>(row, slot16);
}

template <class Tag, int QT>
__global__ __launch_bounds__(ATHREADS) void attn_kernel(AttnArgs a) {
    constexpr int ABM = 64 * QT;
    typedef typename Tag::elem T;
    constexpr int EPC = Tag::EPC;
    constexpr int ROWB = 64 * (int)sizeof(T);     // bytes per LDS tile row (128 or 256)
    constexpr int SLOTS = ROWB / 16;              // 16-byte slots per row
    constexpr int NC = 64 / (4 * EPC);            // chunks along a 64-long contraction (2 or 4)
    constexpr int NCT = 64 * SLOTS / ATHREADS;    // staged chunks per thread per tile (2 or 4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* smK = smem;                 // buffer b: K at smK + b * BUFB, V^T at smV + b * BUFB (BUFB = both tiles)
    char* smV = smem + 64 * ROWB;

    // XCD-aware order: the query tiles of one (segment, head) share its K/V (<= 1 MB at n = 4096), so keep
    // them consecutive on one XCD's L2; v = (row tile, head) with head fastest-but-one.
    const int ntile = gridDim.x >> 2;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int head = v / ntile;
    TileLoc t;
    if constexpr (QT <= 2) {
        t = locate_tile(a.rs, v - head * ntile, ABM);
    } else {   // 256-row tiles are laid out per segment (capacities are multiples of 128, not of 256)
        const int t0 = (a.rs.cap0 + ABM - 1) / ABM, t1 = (a.rs.cap1 + ABM - 1) / ABM, idx = v - head * ntile;
        t.pair = idx / (t0 + t1);
        const int rem = idx - t.pair * (t0 + t1);
        t.image = rem >= t0 ? 1 : 0;
        t.r0 = (rem - t.image * t0) * ABM;
        t.seg = 2 * t.pair + t.image;
        t.grow0 = seg_row_base(a.rs, t.seg) + t.r0;
    }
    const int qlen = a.rs.len[t.seg];
    if (t.r0 >= qlen) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int kvseg = a.cross ? (t.seg ^ 1) : t.seg;
    const int kvlen = a.rs.len[kvseg];
    const long long kvbase = seg_row_base(a.rs, kvseg);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
    const long long R = a.R;

    const T* Q = static_cast<const T*>(a.q);
    const T* Kp = static_cast<const T*>(a.cross ? a.q : a.k);
    const T* Vt = static_cast<const T*>(a.vt);

    if (kvlen == 0) {  // ref :114-115: empty key set -> zeros
        for (int i = tid; i < ABM * 16; i += ATHREADS) {
            const int row = i >> 4, c4 = i & 15;
            if (t.r0 + row < qlen) *reinterpret_cast<f32x4*>(a.ctx + (t.grow0 + row) * 256LL + head * 64 + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // Q fragments (B operand of S^T = K Q^T): lane supplies query column lr, k-slots of group g
    u32x4 qf[QT][NC];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const long long grow = min((long long)t.grow0 + wave * (16 * QT) + qt * 16 + lr, R - 1);   // a 256-row tile may overhang the last segment
#pragma unroll
        for (int c = 0; c < NC; ++c)
            qf[qt][c] = *reinterpret_cast<const u32x4*>(Q + ((long long)head * R + grow) * 64 + c * 4 * EPC + g * EPC);
    }

    f32x4 o[4][QT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -INFINITY; l_run[qt] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    u32x4 rk[NCT], rv[NCT];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            const int c = tid + ATHREADS * i, row = c / SLOTS, slot = c % SLOTS;
            rk[i] = *reinterpret_cast<const u32x4*>(Kp + ((long long)head * R + kvbase + kv0 + row) * 64 + slot * EPC);
            rv[i] = *reinterpret_cast<const u32x4*>(Vt + ((long long)head * 64 + row) * R + kvbase + kv0 + slot * EPC);
        }
    };
    constexpr int BUFB = 2 * 64 * ROWB;   // one K + V^T tile pair (a single buffer is used; see DESIGN.md for the double-buffered variants that lost)
    auto store_tile = [&](int buf, int kv0) {
        char* bK = smK + buf * BUFB; char* bV = smV + buf * BUFB;
        if (kv0 + ABK <= kvlen) {   // wave-uniform fast path: every key of the tile is live
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                const int c = tid + ATHREADS * i, row = c / SLOTS, slot = c % SLOTS;
                *reinterpret_cast<u32x4*>(bK + k_off<ROWB>(row, slot)) = rk[i];
                *reinterpret_cast<u32x4*>(bV + lds_off<ROWB>(row, slot)) = rv[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                const int c = tid + ATHREADS * i, row = c / SLOTS, slot = c % SLOTS;
                // rows/keys past the live length may hold anything (uninitialised workspace): zero them so
                // that 0-probabilities never multiply a NaN
                const u32x4 kz = (kv0 + row < kvlen) ? rk[i] : u32x4{0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(bK + k_off<ROWB>(row, slot)) = kz;
                *reinterpret_cast<u32x4*>(bV + lds_off<ROWB>(row, slot)) = mask_tail<Tag>(rv[i], kvlen - (kv0 + slot * EPC));
            }
        }
    };

    const int ntiles = (kvlen + ABK - 1) / ABK;
#ifdef LG_ATTN_TIMING   // profiling build only (tools/attn_timing.py): per-phase s_memtime sums, wave-uniform
    long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#define ATT_TICK(slot) do { const long long _n = clock64(); tacc[slot] += _n - tprev; tprev = _n; } while (0)
#else
#define ATT_TICK(slot) do { } while (0)
#endif
    // ---- S^T = K Q^T  (4 key tiles x QT query tiles) from the K tile in LDS buffer `buf`
    auto qk = [&](int buf, f32x4 (&s)[4][QT]) {
        const char* bK = smK + buf * BUFB;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int krow = EPC == 8 ? 32 * (kt >> 1) + 8 * (lr >> 2) + 4 * (kt & 1) + (lr & 3) : kt * 16 + lr;
                const u32x4 kf = *reinterpret_cast<const u32x4*>(bK + k_off<ROWB>(krow, c * 4 + g));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(s[kt][qt], kf, qf[qt][c]);
            }
        }
    };
    // ---- online softmax (fp32) of one tile's scores, per query column; s becomes the probabilities
    auto smax_rescale = [&](f32x4 (&s)[4][QT], int kv0) {
        if (kv0 + ABK > kvlen) {   // wave-uniform: only the last tile can hold dead keys
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + (EPC == 8 ? 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) : kt * 16 + g * 4) + r >= kvlen) s[kt][qt][r] = -INFINITY;
        }
        float m_new[QT];
        bool grew = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // 16 scores per lane and query tile: a depth-3 tree of v_max3
            const float t0 = vmax3(s[0][qt][0], s[0][qt][1], s[0][qt][2]), t1 = vmax3(s[0][qt][3], s[1][qt][0], s[1][qt][1]);
            const float t2 = vmax3(s[1][qt][2], s[1][qt][3], s[2][qt][0]), t3 = vmax3(s[2][qt][1], s[2][qt][2], s[2][qt][3]);
            const float t4 = vmax3(s[3][qt][0], s[3][qt][1], s[3][qt][2]);
            float mx = vmax2(vmax3(t0, t1, t2), vmax3(t3, t4, s[3][qt][3]));
            mx = xor32_max(xor16_max(mx));   // the 4 lane groups of a query column, no LDS round trip
            m_new[qt] = vmax2(m_run[qt], mx);   // finite: every tile holds >= 1 live key (q, k carry the score scale)
            grew = grew || (m_new[qt] > m_run[qt] + 8.f);
        }
        // Deferred rescale (guide T13): keep the stale running maximum while no row's maximum grew by more than 2^8
        // (base-2 units) — probabilities then reach at most 2^8 instead of 1, which costs nothing in the fp32 l / O
        // accumulators and nothing relative in the f16 / bf16 P operand.  Order is the textbook one: decide, rescale o
        // and l, THEN exponentiate this tile against the (possibly updated) maximum.  The first tile always takes the
        // branch (m_run = -inf); on most later tiles the O accumulators are not touched by the VALU at all.
        if (__any(grew)) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                // per ROW (see attn_dma_kernel): rows that did not grow keep their maximum, alpha = exp2(0) = 1 exactly
                const float m_upd = m_new[qt] > m_run[qt] + 8.f ? m_new[qt] : m_run[qt];
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_upd);
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
                m_run[qt] = m_upd;
            }
        }
    };
    auto sexp = [&](f32x4 (&s)[4][QT]) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // exp2(s - m) with SCALAR v_sub_f32 / v_add_f32 and the bare v_exp_f32 (arguments <= 8, flush-to-zero tail is
            // fine).  Packed f32 ops (v_pk_add_f32) are cheaper on their own but barely co-issue with another wave's MFMAs on
            // the SIMD (tools/ubench/mfma_valu_overlap.hip: 24 % overlap vs 79 % for scalar ops), and this kernel lives on that
            // overlap.  Two partial sums keep the add chain short.
            const float nm = m_run[qt];
            float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float p0 = __builtin_amdgcn_exp2f(s[kt][qt][0] - nm), p1 = __builtin_amdgcn_exp2f(s[kt][qt][1] - nm);
                const float p2 = __builtin_amdgcn_exp2f(s[kt][qt][2] - nm), p3 = __builtin_amdgcn_exp2f(s[kt][qt][3] - nm);
                s[kt][qt][0] = p0; s[kt][qt][1] = p1; s[kt][qt][2] = p2; s[kt][qt][3] = p3;
                rs0 += p0 + p2; rs1 += p1 + p3;
            }
            l_run[qt] += rs0 + rs1;
        }
    };
    // ---- O^T += V^T P^T with the V^T tile in LDS buffer `buf`
    auto pv = [&](int buf, f32x4 (&s)[4][QT]) {
        const char* bV = smV + buf * BUFB;
        if constexpr (EPC == 8) {
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                u32x4 pp[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) pp[qt] = pack8<Tag>(s[2 * tp][qt], s[2 * tp + 1][qt]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(bV + lds_off<ROWB>(dt * 16 + lr, 4 * tp + g));   // keys 32tp + 8g .. +7
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(o[dt][qt], vf, pp[qt]);
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(bV + lds_off<ROWB>(dt * 16 + lr, 4 * kt + g));
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(o[dt][qt], vf, __builtin_bit_cast(u32x4, s[kt][qt]));
                }
            }
        }
    };

    load_tile(0);
    {
        f32x4 s[4][QT];
        for (int tile = 0; tile < ntiles; ++tile) {
            const int kv0 = tile * ABK;
            __syncthreads();
            ATT_TICK(0);
            store_tile(0, kv0);
            __syncthreads();
            ATT_TICK(1);
            load_tile(tile + 1 < ntiles ? kv0 + ABK : kv0);   // clamped, not branched (keeps hipcc's vmcnt counting exact)
            __builtin_amdgcn_sched_barrier(0);                // and pinned ahead of the MFMAs
            qk(0, s);
            ATT_TICK(2);
            smax_rescale(s, kv0);
            sexp(s);
            ATT_TICK(4);
            pv(0, s);
            ATT_TICK(5);
        }
    }
#ifdef LG_ATTN_TIMING
    if (a.dbg && lane == 0) {
        long long* d = a.dbg + ((long long)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 6; ++i) d[i] = tacc[i];
        d[6] = ntiles; d[7] = 1;
    }
#endif
    // ---- normalise and store ctx[row][head*64 + d]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        l = xor32_sum(xor16_sum(l));
        const float inv = 1.f / l;
        const int qrow = t.r0 + wave * (16 * QT) + qt * 16 + lr;
        if (qrow < qlen) {
            float* dst = a.ctx + (t.grow0 + wave * (16 * QT) + qt * 16 + lr) * 256LL + head * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(dst + dt * 16) = o[dt][qt] * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the 16-bit, 32-rows-per-wave kernel (the default operating point).  Same mathematics, same
// operand layouts, same instruction order inside a tile, so its output is bit-identical to attn_kernel<Tag, 2>.
// What changes is how a K / V^T tile reaches LDS and how often the workgroup synchronises:
//   * `global_load_lds_dwordx4`: each wave moves 2 + 2 pieces of 1 KB (8 rows x 128 B) per tile straight from L2 into
//     LDS.  The DMA writes lane l's 16 bytes at piece base + 16 l, so the bank swizzle is applied on the SOURCE side:
//     lane l fetches the logical slot that belongs at physical slot l & 7 of row l >> 3.
//   * two tile buffers (32 KB per workgroup, 4 workgroups per CU) and ONE barrier per tile: "my pieces of tile t have
//     landed" (vmcnt) + barrier = tile t complete AND every wave done with tile t - 1, whose buffer then takes tile t + 1
//     while tile t is being multiplied.
//   * no staging registers -> 128 VGPRs -> 4 waves per SIMD (was 3): at N = M = 1024, B = 32 the 2048 workgroups of a
//     launch are exactly 2 full rounds of the chip instead of 2.67;
//   * LDS fragment addresses as two per-lane bases + immediates (the swizzle depends on the lane only);
//   * wave priority by progress (see the tile loop).
// A partial last tile is fixed up in LDS (V^T columns past the live length are zeroed: 0 x stale-NaN must not happen;
// K rows past it need nothing, their scores are overwritten with -inf).
template <class Tag>
__global__ __launch_bounds__(ATHREADS, 4) void attn_dma_kernel(AttnArgs a) {
    constexpr int QT = 2, ABM = 128, ROWB = 128, TILEB = 64 * ROWB, BUFB = 2 * TILEB;
    typedef typename Tag::elem T;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int ntile = gridDim.x >> 2;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int head = v / ntile;
    const TileLoc t = locate_tile(a.rs, v - head * ntile, ABM);
    const int qlen = a.rs.len[t.seg];
    if (t.r0 >= qlen) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int kvseg = a.cross ? (t.seg ^ 1) : t.seg;
    const int kvlen = a.rs.len[kvseg];
    const long long kvbase = seg_row_base(a.rs, kvseg);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const long long R = a.R;
    const T* Q = static_cast<const T*>(a.q);
    const T* Kp = static_cast<const T*>(a.cross ? a.q : a.k);
    const T* Vt = static_cast<const T*>(a.vt);

    if (kvlen == 0) {  // ref :114-115: empty key set -> zeros
        for (int i = tid; i < ABM * 16; i += ATHREADS) {
            const int row = i >> 4, c4 = i & 15;
            if (t.r0 + row < qlen) *reinterpret_cast<f32x4*>(a.ctx + (t.grow0 + row) * 256LL + head * 64 + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // DMA source offsets (elements), per lane: piece p = 2 * wave + i covers tile rows 8p .. 8p + 7
    const int prow = lane >> 3, pslot = lane & 7;
    int koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * wave + i) * 8 + prow;
        koff[i] = row * 64 + ((pslot ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1))) << 3);          // inverse of k_off<128>
        voff[i] = (pslot ^ ((row >> 1) & 7)) << 3;                                                      // inverse of lds_off<128>; + row * R below
    }
    const T* kseg = Kp + ((long long)head * R + kvbase) * 64;
    const T* vrow[2] = {Vt + ((long long)head * 64 + (2 * wave) * 8 + prow) * R + kvbase + voff[0],
                        Vt + ((long long)head * 64 + (2 * wave + 1) * 8 + prow) * R + kvbase + voff[1]};
    auto dma_tile = [&](int buf, int kv0) {
        char* bK = smem + buf * BUFB; char* bV = bK + TILEB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lds_dma16(kseg + (long long)kv0 * 64 + koff[i], bK + (2 * wave + i) * 1024);
            lds_dma16(vrow[i] + kv0, bV + (2 * wave + i) * 1024);
        }
    };
    dma_tile(0, 0);

    // LDS fragment offsets: the swizzle terms depend on the lane only (K: row bits 1, 3, 4; V^T: row bits 1..3), so tile
    // (kt, dt) steps are immediate offsets and a k-chunk step (slot bit 2, inside the XOR) needs a second base register
    int kfo[2], vfo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        kfo[c] = k_off<ROWB>(8 * (lr >> 2) + (lr & 3), c * 4 + g);
        vfo[c] = TILEB + lds_off<ROWB>(lr, 4 * c + g);
    }
    u32x4 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const long long grow = (long long)t.grow0 + wave * 32 + qt * 16 + lr;
#pragma unroll
        for (int c = 0; c < 2; ++c) qf[qt][c] = *reinterpret_cast<const u32x4*>(Q + ((long long)head * R + grow) * 64 + c * 32 + g * 8);
    }
    f32x4 o[4][QT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = 0.f; l_run[qt] = 0.f;          // finite: the accumulators start at -m_run; the first tile always re-bases (below)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (kvlen + ABK - 1) / ABK;
#ifdef LG_ATTN_TIMING
    long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#endif
#ifdef LG_ATTN_WALL    // experiment: workgroup life span on the 100 MHz wall clock + where it ran (no per-phase stamps)
    const long long wall0 = wall_clock64(), cyc0 = clock64();
#endif
    for (int tile = 0; tile < ntiles; ++tile) {
        const int kv0 = tile * ABK;
        const char* bK = smem + (tile & 1) * BUFB; const char* bV = bK + TILEB;
        // The SIMD arbiter issues oldest-wave-first, so of the 4 waves sharing a SIMD the first one ran at nearly solo speed
        // and the last one at a third of it: workgroup lives spread 20 ... 57 us, the second round started ragged and the
        // kernel ended with a 25 us ramp-down at <= 75 % occupancy (tools/attn_wall.py).  Priority by PROGRESS (the wave
        // that is furthest behind issues first) makes the co-resident waves finish together: two sharp rounds, -3.5 %.
        if ((tile & 3) == 0) {
            const int q = (tile * 4) / ntiles;
            if (q == 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2); else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        // my pieces of this tile (and, first time round, my Q fragments).  The BUILTIN, not inline asm: hipcc's waitcnt pass must see this wait —
        // it knows nothing of the asm DMAs, but it does track the Q fragment loads, and with an asm wait it still believed them in flight at
        // their first use in the (peeled) first tile: its own vmcnt(0) there also waited for the NEXT tile's DMAs issued in between (in-order
        // counter), i.e. one exposed DMA round trip per workgroup (ISA check: tools/check_isa.py)
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt / lgkmcnt untouched
        asm volatile("" ::: "memory");
        __syncthreads();
        ATT_TICK(0);
        if (kv0 + ABK > kvlen) {                            // workgroup-uniform: zero the dead key columns of V^T
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + ATHREADS * i, row = c >> 3, slot = c & 7;
                u32x4* p = reinterpret_cast<u32x4*>(const_cast<char*>(bV) + lds_off<ROWB>(row, slot));
                *p = mask_tail<Tag>(*p, kvlen - (kv0 + slot * 8));
            }
            __syncthreads();
        }
        dma_tile((tile + 1) & 1, tile + 1 < ntiles ? kv0 + ABK : kv0);   // not branched (see attn_split_kernel); the last tile re-fetches itself into the idle buffer
        __builtin_amdgcn_sched_barrier(0);
        ATT_TICK(1);

        f32x4 s[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{-m_run[qt], -m_run[qt], -m_run[qt], -m_run[qt]};   // scores arrive as s - m_run (the first MFMA of a chain reads this tuple as its C operand)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(bK + kfo[c] + (32 * (kt >> 1) + 4 * (kt & 1)) * ROWB);   // key row 32 (kt >> 1) + 8 (lr >> 2) + 4 (kt & 1) + (lr & 3)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(s[kt][qt], kf, qf[qt][c]);
            }
        }
        ATT_TICK(2);
        if (kv0 + ABK > kvlen) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r >= kvlen) s[kt][qt][r] = -INFINITY;
        }
        // s holds d = score - m_run.  Row maximum of d; re-base when it exceeds 8 (deferred rescale: p <= 2^8 otherwise) and
        // always on the first tile (m_run = 0 there is arbitrary; the re-base may go DOWN, so alpha is not used for it: l = o = 0)
        float dmax[QT];
        bool grew = tile == 0;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float t0 = vmax3(s[0][qt][0], s[0][qt][1], s[0][qt][2]), t1 = vmax3(s[0][qt][3], s[1][qt][0], s[1][qt][1]);
            const float t2 = vmax3(s[1][qt][2], s[1][qt][3], s[2][qt][0]), t3 = vmax3(s[2][qt][1], s[2][qt][2], s[2][qt][3]);
            const float t4 = vmax3(s[3][qt][0], s[3][qt][1], s[3][qt][2]);
            float mx = vmax2(vmax3(t0, t1, t2), vmax3(t3, t4, s[3][qt][3]));
            dmax[qt] = xor32_max(xor16_max(mx));              // finite: every tile holds >= 1 live key
            grew = grew || (dmax[qt] > 8.f);
        }
        if (__any(grew)) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                // per ROW: a row re-bases only on its own growth (shift 0 -> alpha = 1, s - 0: bit-exact no-ops), so a row's
                // arithmetic does not depend on which rows share its wave — workgroup shapes and batch composition cannot show
                const float shift = tile == 0 ? dmax[qt] : (dmax[qt] > 8.f ? dmax[qt] : 0.f);
                const float alpha = tile == 0 ? 0.f : __builtin_amdgcn_exp2f(-shift);
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
                m_run[qt] += shift;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[kt][qt] -= shift;
            }
        }
        ATT_TICK(3);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float p0 = __builtin_amdgcn_exp2f(s[kt][qt][0]), p1 = __builtin_amdgcn_exp2f(s[kt][qt][1]);
                const float p2 = __builtin_amdgcn_exp2f(s[kt][qt][2]), p3 = __builtin_amdgcn_exp2f(s[kt][qt][3]);
                s[kt][qt][0] = p0; s[kt][qt][1] = p1; s[kt][qt][2] = p2; s[kt][qt][3] = p3;
                rs0 += p0 + p2; rs1 += p1 + p3;
            }
            l_run[qt] += rs0 + rs1;
        }
        ATT_TICK(4);
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            u32x4 pp[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) pp[qt] = pack8<Tag>(s[2 * tp][qt], s[2 * tp + 1][qt]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 vf = *reinterpret_cast<const u32x4*>(bK + vfo[tp] + dt * 16 * ROWB);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(o[dt][qt], vf, pp[qt]);
            }
        }
        ATT_TICK(5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (redundant) tile fetch must have landed before this wave's LDS can be released
#ifdef LG_ATTN_TIMING
    if (a.dbg && lane == 0) {
        long long* d = a.dbg + ((long long)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 6; ++i) d[i] = tacc[i];
        d[6] = ntiles; d[7] = 1;
    }
#endif
#ifdef LG_ATTN_WALL
    if (a.dbg && lane == 0) {
        long long* d = a.dbg + ((long long)blockIdx.x * 4 + wave) * 8;
        d[0] = wall0; d[1] = wall_clock64(); d[2] = clock64() - cyc0; d[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4); d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20); d[6] = ntiles; d[7] = 1;
    }
#endif
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        l = xor32_sum(xor16_sum(l));
        const float inv = 1.f / l;
        const int qrow = t.r0 + wave * 32 + qt * 16 + lr;
        if (qrow < qlen) {
            float* dst = a.ctx + (t.grow0 + wave * 32 + qt * 16 + lr) * 256LL + head * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(dst + dt * 16) = o[dt][qt] * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// SPLIT-f16 attention (attention precision PREC_F16X3, the default): q, k, v arrive as hi + lo f16 planes (lg_proj_body.h,
// 22 operand bits) and the probabilities are split the same way, so both contractions run as three MFMAs per product
//      S^T = Kh Ql^T + Kl Qh^T + Kh Qh^T          O^T = Vh^T Pl^T + Vl^T Ph^T + Vh^T Ph^T
// with fp32 accumulation.  Why: with one f16 plane a base-2 logit of magnitude 30 carries an absolute error of ~1e-2, i.e. 1 % on
// the softmax weights — invisible while attention is diffuse (random-weight fixtures), 2e-2 in the matching scores once the logit
// spread reaches 25 (recipe-D fixtures, DESIGN.md §1: QK^T alone 1.4e-2, P V alone 5e-3; split: 5e-5 / < 5e-4).  There is no cheaper
// operand format with >= 17 bits on gfx950 (no xf32; fp32 MFMA runs at 1/16 of the f16 rate).
// Structure = attn_dma_kernel: LDS-DMA tiles, two buffers, one barrier per tile, wave priority by progress, scores accumulated
// from -m_run.  A buffer holds four 8 KB tiles (K hi, K lo, V^T hi, V^T lo): 64 KB per workgroup, two workgroups per CU.
// QT = 16-row query tiles per wave, NW = waves per workgroup.  Shapes (round-3 A/B at cfg #2, one box each): 8 waves x 16 rows
// (108 VGPRs, four waves per SIMD; the product shape, and 4 x 16 rows = 64-row workgroups for under-filled grids) is 2-3 % faster
// than 4 waves x 32 rows (172 VGPRs, two waves per SIMD).  Lost: a cross-tile software pipeline (S(t+1) MFMAs interleaved with the
// softmax VALU of tile t inside each wave via sched_group_barrier: bit-identical, +7 % time).  SQ counters of the kernel: matrix
// pipe 54 % busy, waves 42 % issue-stalled, 25 % parked; timing ablations: no DMA -9 %, no exponentials -4 % — LAB_NOTES.md.
// Round 6: a PING-PONG form (the two waves a workgroup has on a SIMD half a tile apart — one multiplies while the other runs its softmax —, two barriers per
// tile, K(t + 1) requested behind the first and V(t + 1) behind the second; bit-identical; tools/experiments/attn_pingpong.patch) took MORE cycles per workgroup
// (64.9k vs 63.5k) and +2 ... +6 % time; priorities by phase or none at all: noise.  The kernel is not bound by when its waves do what: inside a forward the chip
// sits at its 1.35 kW power limit and runs this kernel at 1.64 GHz of 2.4 (profiles/r06b_attn_wall_clock.log) — what moves it is energy per tile.
template <int QT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void attn_split_kernel(AttnArgs a) {
    typedef TagF16 Tag;
    typedef f16_t T;
    constexpr int NT = NW * 64, PPW = 8 / NW;      // threads; DMA pieces per wave per (plane, tile)
    constexpr int ABM = 16 * QT * NW, ROWB = 128, TILEB = 64 * ROWB, BUFB = 4 * TILEB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int ntile = gridDim.x >> 2;
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int head = v / ntile;
    const TileLoc t = locate_tile(a.rs, v - head * ntile, ABM);
    const int qlen = a.rs.len[t.seg];
    if (t.r0 >= qlen) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int kvseg = a.cross ? (t.seg ^ 1) : t.seg;
    const int kvlen = a.rs.len[kvseg];
    const long long kvbase = seg_row_base(a.rs, kvseg);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const long long R = a.R, PL = a.plane;
    const T* Q = static_cast<const T*>(a.q);
    const T* Kp = static_cast<const T*>(a.cross ? a.q : a.k);
    const T* Vt = static_cast<const T*>(a.vt);

    if (kvlen == 0) {  // ref :114-115: empty key set -> zeros
        for (int i = tid; i < ABM * 16; i += NT) {
            const int row = i >> 4, c4 = i & 15;
            if (t.r0 + row < qlen) *reinterpret_cast<f32x4*>(a.ctx + (t.grow0 + row) * 256LL + head * 64 + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // DMA source offsets (elements), per lane: piece p = PPW * wave + i covers rows 8p .. 8p + 7 of each of the four tiles
    const int prow = lane >> 3, pslot = lane & 7;
    int koff[PPW], voff[PPW];
    const T* vrow[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = (PPW * wave + i) * 8 + prow;
        koff[i] = row * 64 + ((pslot ^ (((row >> 1) & 1) | (((row >> 3) & 3) << 1))) << 3);          // inverse of k_off<128>
        voff[i] = (pslot ^ ((row >> 1) & 7)) << 3;                                                      // inverse of lds_off<128>; + row * R below
    }
    const T* kseg = Kp + ((long long)head * R + kvbase) * 64;
#pragma unroll
    for (int i = 0; i < PPW; ++i) vrow[i] = Vt + ((long long)head * 64 + (PPW * wave + i) * 8 + prow) * R + kvbase + voff[i];
    auto dma_tile = [&](int buf, int kv0) {
        char* bK = smem + buf * BUFB;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const T* ks = kseg + (long long)kv0 * 64 + koff[i];
            const T* vs = vrow[i] + kv0;
            char* dst = bK + (PPW * wave + i) * 1024;
            lds_dma16(ks, dst); lds_dma16(ks + PL, dst + TILEB);
            lds_dma16(vs, dst + 2 * TILEB); lds_dma16(vs + PL, dst + 3 * TILEB);
        }
    };
    dma_tile(0, 0);

    // LDS fragment offsets (see attn_dma_kernel): per k-chunk c one base for the K tiles and one for the V^T tiles
    int kfo[2], vfo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        kfo[c] = k_off<ROWB>(8 * (lr >> 2) + (lr & 3), c * 4 + g);
        vfo[c] = 2 * TILEB + lds_off<ROWB>(lr, 4 * c + g);
    }
    u32x4 qh[QT][2], ql[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const long long grow = (long long)t.grow0 + wave * (16 * QT) + qt * 16 + lr;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const T* src = Q + ((long long)head * R + grow) * 64 + c * 32 + g * 8;
            qh[qt][c] = *reinterpret_cast<const u32x4*>(src);
            ql[qt][c] = *reinterpret_cast<const u32x4*>(src + PL);
        }
    }
    f32x4 o[4][QT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = 0.f; l_run[qt] = 0.f;          // finite: the accumulators start at -m_run; the first tile always re-bases (below)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int ntiles = (kvlen + ABK - 1) / ABK;
#ifdef LG_ATTN_TIMING
    long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#endif
#ifdef LG_ATTN_WALL    // profiling build: workgroup life span on the 100 MHz wall clock + shader cycles -> the clock the kernel ran at (tools/attn_wall.py)
    const long long wall0 = wall_clock64(), cyc0 = clock64();
#endif
    for (int tile = 0; tile < ntiles; ++tile) {
        const int kv0 = tile * ABK;
        const char* bK = smem + (tile & 1) * BUFB;
        if ((tile & 3) == 0) {   // wave priority by progress, see attn_dma_kernel
            const int q = (tile * 4) / ntiles;
            if (q == 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2); else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        // my pieces of this tile (and, first time round, my Q fragments).  The BUILTIN, not inline asm: hipcc's waitcnt pass must see this wait —
        // it knows nothing of the asm DMAs, but it does track the Q fragment loads, and with an asm wait it still believed them in flight at
        // their first use in the (peeled) first tile: its own vmcnt(0) there also waited for the NEXT tile's DMAs issued in between (in-order
        // counter), i.e. one exposed DMA round trip per workgroup (ISA check: tools/check_isa.py)
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt / lgkmcnt untouched
        asm volatile("" ::: "memory");
        __syncthreads();
        ATT_TICK(0);
        if (kv0 + ABK > kvlen) {                            // workgroup-uniform: zero the dead key columns of both V^T planes
#pragma unroll
            for (int i = 0; i < 1024 / NT; ++i) {         // 512 16-byte chunks per plane
                const int c = (tid + NT * i) & 511, plane = (tid + NT * i) >> 9, row = c >> 3, slot = c & 7;
                u32x4* p = reinterpret_cast<u32x4*>(const_cast<char*>(bK) + (2 + plane) * TILEB + lds_off<ROWB>(row, slot));
                *p = mask_tail<Tag>(*p, kvlen - (kv0 + slot * 8));
            }
            __syncthreads();
        }
        // Fragment reads are software-pipelined one group ahead of the MFMAs (two register sets; sched_barrier pins the order —
        // hipcc otherwise sinks every ds_read next to its use and each group of 12 MFMAs starts with an exposed LDS round trip,
        // 16 of them per tile, with only two waves per SIMD to cover).  Group = (k-chunk, pair of key tiles) for S^T and
        // (P chunk, pair of d tiles) for O^T; the first V^T group is fetched during the last K group, so that its latency
        // hides under the softmax.
        u32x4 fh[2][2], fl[2][2];
        auto load_k = [&](int grp, u32x4 (&h)[2], u32x4 (&l)[2]) {
            const int c = grp >> 1, kp = grp & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kt = 2 * kp + j;
                const char* src = bK + kfo[c] + (32 * (kt >> 1) + 4 * (kt & 1)) * ROWB;   // key row 32 (kt >> 1) + 8 (lr >> 2) + 4 (kt & 1) + (lr & 3)
                h[j] = *reinterpret_cast<const u32x4*>(src);
                l[j] = *reinterpret_cast<const u32x4*>(src + TILEB);
            }
        };
        auto load_v = [&](int grp, u32x4 (&h)[2], u32x4 (&l)[2]) {
            const int tp = grp >> 1, dp = grp & 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* src = bK + vfo[tp] + (2 * dp + j) * 16 * ROWB;
                h[j] = *reinterpret_cast<const u32x4*>(src);
                l[j] = *reinterpret_cast<const u32x4*>(src + TILEB);
            }
        };
        load_k(0, fh[0], fl[0]);
        // NOT branched around: hipcc merges its s_waitcnt bookkeeping conservatively at a join, and every fragment wait of the tile
        // became lgkmcnt(0) — i.e. the prefetched group was waited for as well (measured in the ISA).  The last tile re-fetches
        // itself into the idle buffer instead; the wave drains vmcnt before it leaves the loop (no DMA may land after the exit).
        dma_tile((tile + 1) & 1, tile + 1 < ntiles ? kv0 + ABK : kv0);
        __builtin_amdgcn_sched_barrier(0);
        ATT_TICK(1);

        // ---- S^T - m_run: per k-chunk and pair of key tiles, three products over 2 x QT independent accumulators
        f32x4 s[4][QT];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{-m_run[qt], -m_run[qt], -m_run[qt], -m_run[qt]};
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int c = grp >> 1, kp = grp & 1;
            if (grp < 3) load_k(grp + 1, fh[(grp + 1) & 1], fl[(grp + 1) & 1]);
            else load_v(0, fh[0], fl[0]);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&kh)[2] = fh[grp & 1];
            const u32x4 (&kl)[2] = fl[grp & 1];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(s[2 * kp + j][qt], kh[j], ql[qt][c]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(s[2 * kp + j][qt], kl[j], qh[qt][c]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(s[2 * kp + j][qt], kh[j], qh[qt][c]);
            __builtin_amdgcn_sched_barrier(0);
        }
        ATT_TICK(2);
        if (kv0 + ABK > kvlen) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + 32 * (kt >> 1) + 8 * g + 4 * (kt & 1) + r >= kvlen) s[kt][qt][r] = -INFINITY;
        }
        // ---- online softmax on d = score - m_run (deferred rescale; the first tile always re-bases), as in attn_dma_kernel
        float dmax[QT];
        bool grew = tile == 0;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float t0 = vmax3(s[0][qt][0], s[0][qt][1], s[0][qt][2]), t1 = vmax3(s[0][qt][3], s[1][qt][0], s[1][qt][1]);
            const float t2 = vmax3(s[1][qt][2], s[1][qt][3], s[2][qt][0]), t3 = vmax3(s[2][qt][1], s[2][qt][2], s[2][qt][3]);
            const float t4 = vmax3(s[3][qt][0], s[3][qt][1], s[3][qt][2]);
            float mx = vmax2(vmax3(t0, t1, t2), vmax3(t3, t4, s[3][qt][3]));
            dmax[qt] = xor32_max(xor16_max(mx));              // finite: every tile holds >= 1 live key
            grew = grew || (dmax[qt] > 8.f);
        }
        if (__any(grew)) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                // per ROW: a row re-bases only on its own growth (shift 0 -> alpha = 1, s - 0: bit-exact no-ops), so a row's
                // arithmetic does not depend on which rows share its wave — workgroup shapes and batch composition cannot show
                const float shift = tile == 0 ? dmax[qt] : (dmax[qt] > 8.f ? dmax[qt] : 0.f);
                const float alpha = tile == 0 ? 0.f : __builtin_amdgcn_exp2f(-shift);
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha;
                m_run[qt] += shift;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[kt][qt] -= shift;
            }
        }
        ATT_TICK(3);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const float p0 = __builtin_amdgcn_exp2f(s[kt][qt][0]), p1 = __builtin_amdgcn_exp2f(s[kt][qt][1]);
                const float p2 = __builtin_amdgcn_exp2f(s[kt][qt][2]), p3 = __builtin_amdgcn_exp2f(s[kt][qt][3]);
                s[kt][qt][0] = p0; s[kt][qt][1] = p1; s[kt][qt][2] = p2; s[kt][qt][3] = p3;
                rs0 += p0 + p2; rs1 += p1 + p3;
            }
            l_run[qt] += rs0 + rs1;
        }
        ATT_TICK(4);
        // ---- O^T += V^T P^T with P = Ph + Pl (both f16; the residual p - Ph is exact in fp32): three products per (d tile, query tile)
        u32x4 ph[2][QT], pl[2][QT];
#pragma unroll
        for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) split8_f16<true>(s[2 * tp][qt], s[2 * tp + 1][qt], ph[tp][qt], pl[tp][qt]);
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int tp = grp >> 1, dp = grp & 1;
            if (grp < 3) load_v(grp + 1, fh[(grp + 1) & 1], fl[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&vh)[2] = fh[grp & 1];
            const u32x4 (&vl)[2] = fl[grp & 1];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(o[2 * dp + j][qt], vh[j], pl[tp][qt]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(o[2 * dp + j][qt], vl[j], ph[tp][qt]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) mma_chunk<Tag>(o[2 * dp + j][qt], vh[j], ph[tp][qt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        ATT_TICK(5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (redundant) tile fetch must have landed before this wave's LDS can be released
#ifdef LG_ATTN_TIMING
    if (a.dbg && lane == 0) {
        long long* d = a.dbg + ((long long)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 6; ++i) d[i] = tacc[i];
        d[6] = ntiles; d[7] = 1;
    }
#endif
#ifdef LG_ATTN_WALL
    if (a.dbg && lane == 0) {
        long long* d = a.dbg + ((long long)blockIdx.x * NW + wave) * 8;
        d[0] = wall0; d[1] = wall_clock64(); d[2] = clock64() - cyc0; d[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4); d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20); d[6] = ntiles; d[7] = 1;
    }
#endif
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        l = xor32_sum(xor16_sum(l));
        const float inv = 1.f / l;
        const int qrow = t.r0 + wave * (16 * QT) + qt * 16 + lr;
        if (qrow < qlen) {
            float* dst = a.ctx + (t.grow0 + wave * (16 * QT) + qt * 16 + lr) * 256LL + head * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(dst + dt * 16) = o[dt][qt] * inv;
        }
    }
}

template <int QT, int NW> static hipError_t launch_attn_split(const AttnArgs& a, hipStream_t s) {
    if (a.plane <= 0) return hipErrorInvalidValue;
    constexpr int smem = 2 * 4 * 64 * 128;
    auto kern = attn_split_kernel<QT, NW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    const int tiles = a.rs.B * (a.rs.cap0 + a.rs.cap1) / (16 * QT * NW);
    hipLaunchKernelGGL(kern, dim3(tiles * 4), dim3(NW * 64), smem, s, a);
    return hipGetLastError();
}

template <class Tag> static hipError_t launch_attn_dma(const AttnArgs& a, hipStream_t s) {
    const int tiles = a.rs.B * (a.rs.cap0 + a.rs.cap1) / 128;
    hipLaunchKernelGGL((attn_dma_kernel<Tag>), dim3(tiles * 4), dim3(ATHREADS), 2 * 2 * 64 * 128, s, a);
    return hipGetLastError();
}

template <class Tag, int QT> static hipError_t launch_attn_t(const AttnArgs& a, hipStream_t s) {
    constexpr int ABM = 64 * QT;
    const int tiles = QT <= 2 ? a.rs.B * (a.rs.cap0 + a.rs.cap1) / ABM : a.rs.B * ((a.rs.cap0 + ABM - 1) / ABM + (a.rs.cap1 + ABM - 1) / ABM);
    dim3 grid(tiles * 4);
    constexpr int smem = 2 * 64 * 64 * (int)sizeof(typename Tag::elem);
    hipLaunchKernelGGL((attn_kernel<Tag, QT>), grid, dim3(ATHREADS), smem, s, a);
    return hipGetLastError();
}

// a.rows_per_wave: 32 (two 16-row query tiles per wave, three waves per SIMD) or 64 (four tiles, two waves per SIMD, K/V
// fragments reused twice as often; 16-bit operands only)
hipError_t launch_attention(int attn_prec, const AttnArgs& a, hipStream_t s) {
    const int rpw = a.rows_per_wave;
    switch (attn_prec) {
        case PREC_F32: return launch_attn_t<TagF32, 2>(a, s);
        case PREC_BF16: if (a.dma && rpw == 32) return launch_attn_dma<TagBF16>(a, s); return rpw == 64 ? launch_attn_t<TagBF16, 4>(a, s) : rpw == 16 ? launch_attn_t<TagBF16, 1>(a, s) : launch_attn_t<TagBF16, 2>(a, s);
        case PREC_F16: if (a.dma && rpw == 32) return launch_attn_dma<TagF16>(a, s); return rpw == 64 ? launch_attn_t<TagF16, 4>(a, s) : rpw == 16 ? launch_attn_t<TagF16, 1>(a, s) : launch_attn_t<TagF16, 2>(a, s);
        case PREC_F16X3: return rpw == 16 ? launch_attn_split<1, 4>(a, s) : launch_attn_split<1, 8>(a, s);   // 64-row workgroups for under-filled grids, else 128-row ones (8 waves x 16 rows)
    }
    return hipErrorInvalidValue;
}

}  // namespace lg
