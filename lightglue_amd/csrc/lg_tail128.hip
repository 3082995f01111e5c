// EXPERIMENT BUILDS ONLY (-DLG_EXPERIMENTS; engine option tail_rows = 128).  Measured on MI355X at N=M=1024, B=32: correct
// (bit-for-bit the scores of the 64-row kernel's arithmetic, 0 index mismatches on the reference fixture) but 2 % SLOWER than
// lg_tail.hip: phase A -3 %, LayerNorm -11 %, epilogue -60 % per row, but phase B +33 % (128 accumulator registers leave room for
// a one-chunk-deep W2 ring only, and ~50 VGPRs spill around LayerNorm / the first phase-B step).
// lightglue_amd — fused block tail, 128 keypoint rows per workgroup (split-bf16 operands, f16 attention outputs).
// Same arithmetic, same packed weights and the same transposed-MFMA epilogues as lg_tail.hip (read its header first);
// what changes is the decomposition: every weight fragment a wave pulls from L2 now feeds EIGHT 16-row tiles instead
// of four, and the per-workgroup fixed costs (prologue latency, LayerNorm exchange, the phase barriers, the epilogues'
// load/store latencies) are paid once per 128 rows.  The price is LDS: a 128 x 512 split-bf16 activation tile does not fit,
// so phase A STREAMS [x ; ctx] through a two-slab ring (one 64-wide K stage = 32 KB per slab, one barrier per slab),
// and the accumulators: 128 x 64 per wave = 128 VGPRs, which is why the B-fragment ring is two half-chunks deep and
// activation fragments are read per 16-row tile.
//   LDS map: [0, 128 KB)  phase A: slab ring at [0, 64 KB) (slab = hi tile 16 KB + lo tile 16 KB, [128 rows][128 B] each)
//                         phase B: g double buffer, slot s at s * 64 KB: [plane][2 K stages][128 rows][128 B]
//                         next:    x' tile [plane (64 KB apart)][4 K stages][128 rows][128 B]
//            [128, 136 KB) LayerNorm exchange [128 rows][8 waves] (mean, M2)
#include "lg_proj_body.h"

namespace lg {

constexpr int T8M = 128, T8THREADS = 512;
constexpr int T8_TILE = T8M * 128;              // one plane of one K stage: 16 KB
constexpr int T8_SLAB = 2 * T8_TILE;            // phase A ring slab: hi + lo
constexpr int T8_GSLOT = 4 * T8_TILE;           // phase B slot: 2 planes x 2 stages = 64 KB
constexpr int T8_REGION = 8 * T8_TILE;          // 128 KB
constexpr int T8_LDS = T8_REGION + T8M * 8 * 8;

__device__ __forceinline__ f32x2 gelu8_fast2(f32x2 u) {   // the branch-free GELU of lg_tail.hip
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

// acc (C^T tile) += w x^T: (w_hi x_lo) + (w_lo x_hi) + (w_hi x_hi)
__device__ __forceinline__ void mma3(f32x4& acc, const u32x4* wf, const u32x4* xf) {
    mma_chunk<TagBF16>(acc, wf[0], xf[1]);
    mma_chunk<TagBF16>(acc, wf[1], xf[0]);
    mma_chunk<TagBF16>(acc, wf[0], xf[0]);
}

template <int NEXT>
__global__ __launch_bounds__(T8THREADS) void tail128_kernel(TailArgs a) {
    constexpr int NKC = 16;   // 32-wide k-chunks of the 512-long contractions
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x2* red2 = reinterpret_cast<f32x2*>(smem + T8_REGION);

    const TileLoc t = locate_tile(a.rs, blockIdx.x, T8M);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    // the wave index is made PROVABLY wave-uniform: weight-fragment addresses then split into an SGPR part and one per-lane offset
    // (saves the VGPR pairs hipcc otherwise keeps per address)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    auto stamp = [&](int slot) {
        if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8 + slot] = clock64();
    };
    stamp(0);
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);   // see lg_tail.hip

    const int lane16 = lane * 16;
    auto wfrag = [&](const void* base, int p, long long plane_elems, int nt, int kc) -> u32x4 {
        const char* ptr = static_cast<const char*>(base) + (p ? plane_elems * 2 : 0) + (long long)(nt * NKC + kc) * 1024;   // wave-uniform
        return *reinterpret_cast<const u32x4*>(ptr + lane16);
    };

    // ------------------------------------------------------------------ phase A: h^T = Wcat [x ; ctx]^T, K streamed in 8 slabs
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // slab staging: thread -> row tid >> 2, 16 consecutive floats (two 16-byte operand chunks) at column (tid & 3) * 16
    const int srow = tid >> 2, sq = tid & 3;
    f32x4 st[4];
    auto load_slab = [&](int s) {
        const float* src = (s < 4 ? a.X : a.CTX) + (long long)(t.grow0 + srow) * 256 + (s & 3) * 64 + sq * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) st[j] = *reinterpret_cast<const f32x4*>(src + 4 * j);
    };
    auto store_slab = [&](int s) {
        char* buf = smem + (s & 1) * T8_SLAB;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 hi, lo;
            split8_bf16(st[2 * h], st[2 * h + 1], hi, lo);
            const int off = lds_off<128>(srow, 2 * sq + h);
            *reinterpret_cast<u32x4*>(buf + off) = hi;
            *reinterpret_cast<u32x4*>(buf + T8_TILE + off) = lo;
        }
    };
    // B-fragment unit = (k-chunk kc, n-tile pair np): 2 n-tiles x 2 planes; ring of two units, prefetched one unit ahead
    u32x4 bw[2][2][2];
    auto load_unit = [&](u32x4 (&dst)[2][2], int U) {   // U = kc * 2 + np, clamped by the caller
        const int kc = U >> 1, np = U & 1;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int p = 0; p < 2; ++p) dst[n][p] = wfrag(a.Wcat, p, 512LL * 512, w + 8 * (2 * np + n), kc);
    };
    load_slab(0);
    load_unit(bw[0], 0);
    store_slab(0);
    load_slab(1);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) store_slab(s + 1);                 // loaded during the previous slab; its ring slot was last read two slabs ago
        load_slab(s + 2 < 8 ? s + 2 : 7);                 // clamped, never branched (see lg_tail.hip)
        const char* buf = smem + (s & 1) * T8_SLAB;
#pragma unroll
        for (int u = 0; u < 4; ++u) {                     // units of this slab: k-chunks 2s, 2s+1 x n-tile pairs
            const int U = 4 * s + u, c = u >> 1, np = u & 1;
            load_unit(bw[(u + 1) & 1], U + 1 < 32 ? U + 1 : 31);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                u32x4 af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    af[p] = *reinterpret_cast<const u32x4*>(buf + p * T8_TILE + lds_off<128>(mt * 16 + lr, c * 4 + g));
                mma3(acc[mt][2 * np], bw[u & 1][0], af);
                mma3(acc[mt][2 * np + 1], bw[u & 1][1], af);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                  // slab s + 1 is complete; everyone is done reading slab s
    }
    stamp(1);
    // ------------------------------------------------------------------ bias + LayerNorm(512)
    // acc[mt][nt][r] = h[row mt*16 + lr][hidden (w + 8 nt)*16 + 4g + r]
    {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bcat + (w + 8 * nt) * 16 + 4 * g);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt][nt] += b4;
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
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
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        float mean[8], rstd[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const f32x4* pr = reinterpret_cast<const f32x4*>(red2 + (mt * 16 + lr) * 8);
            const f32x4 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
            const float mu = (((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) * 0.125f;
            float m2 = ((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]));
            float dm = 0.f;
            { float d;
              d = p0[0] - mu; dm += d * d; d = p0[2] - mu; dm += d * d; d = p1[0] - mu; dm += d * d; d = p1[2] - mu; dm += d * d;
              d = p2[0] - mu; dm += d * d; d = p2[2] - mu; dm += d * d; d = p3[0] - mu; dm += d * d; d = p3[2] - mu; dm += d * d; }
            m2 += 64.f * dm;
            mean[mt] = mu; rstd[mt] = __builtin_amdgcn_rsqf(m2 * (1.f / 512.f) + 1e-5f);
            __builtin_amdgcn_sched_barrier(0);   // bound the live range: 8 x 4 hoisted 16-byte reads would spill
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = (w + 8 * nt) * 16 + 4 * g;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + col), bt = *reinterpret_cast<const f32x4*>(a.beta + col);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = (acc[mt][nt][r] - mean[mt]) * rstd[mt] * gm[r] + bt[r];
        }
    }
    stamp(2);
    // ------------------------------------------------------------------ GELU + g -> LDS slots + phase B
    // step j: n-tile j of every wave = hidden [(w + 8j)*16, +16) = K stage (w >> 2) of the step's 128, columns (w & 3)*16 + 4g + r
    auto gelu_store = [&](int j) {
        char* slot = smem + (j & 1) * T8_GSLOT + (w >> 2) * T8_TILE;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const f32x2 v01 = gelu8_fast2(f32x2{acc[mt][j][0], acc[mt][j][1]});
            const f32x2 v23 = gelu8_fast2(f32x2{acc[mt][j][2], acc[mt][j][3]});
            char* dst = slot + lds_off<128>(mt * 16 + lr, (w & 3) * 2 + (g >> 1)) + (g & 1) * 8;
            const float h0 = bf16_round(v01[0]), h1 = bf16_round(v01[1]), h2 = bf16_round(v23[0]), h3 = bf16_round(v23[1]);
            *reinterpret_cast<u32x2*>(dst) = u32x2{pack2_bf16(h0, h1), pack2_bf16(h2, h3)};
            *reinterpret_cast<u32x2*>(dst + 2 * T8_TILE) = u32x2{pack2_bf16(v01[0] - h0, v01[1] - h1), pack2_bf16(v23[0] - h2, v23[1] - h3)};
        }
    };
    f32x4 acc2[8][2];   // out[row mt*16 + lr][column w*32 + nt*16 + 4g + r]
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc2[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    u32x4 b2[2][2][2];  // ring of two k-chunks: 2 out n-tiles x 2 planes
    auto load_b2 = [&](u32x4 (&dst)[2][2], int kc) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int p = 0; p < 2; ++p) dst[n][p] = wfrag(a.W2, p, 256LL * 512, w * 2 + n, kc);
    };
    load_b2(b2[0], 0);
    gelu_store(0);
    __syncthreads();
    stamp(3);
    const int qlen = a.rs.len[t.seg];
    f32x4 xres[8][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j == 3) {   // residual rows: in flight under the last step's MFMAs
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    xres[mt][nt] = *reinterpret_cast<const f32x4*>(a.X + (long long)(t.grow0 + mt * 16 + lr) * 256 + w * 32 + nt * 16 + 4 * g);
        }
        const char* slot = smem + (j & 1) * T8_GSLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kc = 4 * j + i;
            load_b2(b2[(i + 1) & 1], kc + 1 < NKC ? kc + 1 : NKC - 1);
            __builtin_amdgcn_sched_barrier(0);
            const char* tile = slot + (i >> 1) * T8_TILE;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                u32x4 af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    af[p] = *reinterpret_cast<const u32x4*>(tile + p * 2 * T8_TILE + lds_off<128>(mt * 16 + lr, (i & 1) * 4 + g));
                mma3(acc2[mt][0], b2[i & 1][0], af);
                mma3(acc2[mt][1], b2[i & 1][1], af);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j < 3) {
            gelu_store(j + 1);
            __syncthreads();
        }
    }
    stamp(4);
    // ------------------------------------------------------------------ epilogue: + b2, + x, store; next block's activation tile
    if constexpr (NEXT != 0) __syncthreads();   // the x' tile covers both g slots: every wave must be done reading them
    const f32x4 b2v[2] = {*reinterpret_cast<const f32x4*>(a.b2 + w * 32 + 4 * g), *reinterpret_cast<const f32x4*>(a.b2 + w * 32 + 16 + 4 * g)};
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = w * 32 + nt * 16 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int row = mt * 16 + lr;
            const f32x4 xn = xres[mt][nt] + (acc2[mt][nt] + b2v[nt]);
            if (t.r0 + row < qlen) *reinterpret_cast<f32x4*>(a.X + (long long)(t.grow0 + row) * 256 + col) = xn;
            if constexpr (NEXT != 0) {
                char* dst = smem + (col >> 6) * T8_TILE + lds_off<128>(row, (col & 63) >> 3) + (col & 7) * 2;
                *reinterpret_cast<u32x2*>(dst) = u32x2{pack2_f16(xn[0], xn[1]), pack2_f16(xn[2], xn[3])};
            }
        }
    }
    stamp(5);
    if constexpr (NEXT != 0) proj_compute<PREC_QKV_F16W2, f16_t, NEXT == 1 ? 3 : 2, 2, 4 * T8_TILE, 8>(a.next, t, smem, 8);
}

template <int NEXT> static hipError_t launch_tail128_t(const TailArgs& a, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    auto kern = tail128_kernel<NEXT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / T8M), dim3(T8THREADS), T8_LDS, s, a);
    return hipGetLastError();
}

// split-bf16 linear layers + f16 attention operands only (the default precision)
hipError_t launch_tail_rows128(const TailArgs& a, hipStream_t s) {
    if (!a.next.W) return launch_tail128_t<0>(a, s);
    if (a.next.Nout == 768 && a.next.cosb) return launch_tail128_t<1>(a, s);
    if (a.next.Nout == 512 && !a.next.cosb) return launch_tail128_t<2>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace lg
