// lightglue_amd — attention input projections as one kernel per block:
//   SelfBlock  (ref lightglue.py:165-169): qkv = Wqkv x + b, split per head, rotary on q and k
//   CrossBlock (ref :204-209):             qk = to_qk x + b, v = to_v x + b   (same weights for both images)
// Output layouts feed lg_attention.hip directly: q, k [head][row][64], v TRANSPOSED [head][64][row].
//
// Same structure as the fused tail (lg_tail.hip): workgroup = 64 keypoint rows x ALL output columns, 8 waves
// split the columns; the 64 x 256 activation tile is read from HBM exactly once, converted to the operand
// precision and kept in LDS (64 KB as split f16); weight fragments come pre-packed in MFMA order straight
// from L2 (one coalesced 1 KB wave load each).  Wave w owns the n-tiles {w + 8j}: the columns are produced in
// two passes of NTP n-tiles per wave (keeps accumulators + weight ring under the register budget); the outputs
// leave straight from the accumulators (lg_proj_body.h: transposed MFMA form for q/k, plain form for v^T).
#include "lg_proj_body.h"

namespace lg {

// activation tile: HBM -> registers -> operand precision -> LDS (once); no barrier (proj_compute / final_compute synchronise)
template <int PREC>
__device__ __forceinline__ void proj_load_tile(const float* X, const TileLoc& t, char* smA) {
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int EPC = Tag::EPC;
    constexpr int KE = PJL<PREC>::KE, STAGES = PJL<PREC>::STAGES, TILE = PJL<PREC>::TILE, A_PLANE = PJL<PREC>::A_PLANE;
    constexpr int NV = EPC / 4;
    const int tid = threadIdx.x;
    const int srow = tid >> 3, sslot = tid & 7;
    const float* src = X + (long long)(t.grow0 + srow) * 256 + sslot * EPC;
    f32x4 hreg[STAGES][NV];
#pragma unroll
    for (int st = 0; st < STAGES; ++st)
#pragma unroll
        for (int j = 0; j < NV; ++j) hreg[st][j] = *reinterpret_cast<const f32x4*>(src + st * KE + 4 * j);
    const int off = pj_tile_off(srow, sslot);
#pragma unroll
    for (int st = 0; st < STAGES; ++st) {
        char* tile = smA + st * TILE;
        if constexpr (PREC == PREC_F32) {
            *reinterpret_cast<f32x4*>(tile + off) = hreg[st][0];
        } else if constexpr (PJ<PREC>::APART == 2) {
            u32x4 hi, lo;
            split8<Tag>(hreg[st][0], hreg[st][1], hi, lo);
            *reinterpret_cast<u32x4*>(tile + off) = hi;
            *reinterpret_cast<u32x4*>(tile + A_PLANE + off) = lo;
        } else {   // single plane (PREC_QKV_F16W2: one f16 plane)
            *reinterpret_cast<u32x4*>(tile + off) = pack8<Tag>(hreg[st][0], hreg[st][1]);
        }
    }
}

template <int PREC, class TA, int NTP, int NPASS>
__global__ __launch_bounds__(PTHREADS) void proj_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* smA = smem;                                              // [NPART][STAGES][64][128 B]

    const TileLoc t = locate_tile(a.rs, blockIdx.x, PBM);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.dbg && lane == 0) a.dbg[((long long)blockIdx.x * 8 + w) * 8] = clock64();   // profiling tap, slot 0
    RopeRows<4> rr;
    if constexpr (NTP == 3) proj_rope_load<4>(a, t, rr);   // before the x tile: both are cold, in flight together
    proj_load_tile<PREC>(a.X, t, smA);
    proj_compute<PREC, TA, NTP, NPASS>(a, t, smA, 0, NTP == 3 ? &rr : nullptr);
}

// ---- the first SelfBlock projection with the per-keypoint preparation fused in (lg_kernels.h launch_proj_first).  Same arithmetic, in the same
// order, as prep_kernel (lg_pointwise.hip: ref lightglue.py:32-43, :76-81) and proj_kernel, so the outputs are bit-identical to the two-launch form.
template <int PREC, class TA>
__global__ __launch_bounds__(PTHREADS) void proj_first_kernel(ProjArgs a, PrepArgs pa) {
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int EPC = Tag::EPC, NV = EPC / 4;
    constexpr int KE = PJL<PREC>::KE, STAGES = PJL<PREC>::STAGES, TILE = PJL<PREC>::TILE, A_PLANE = PJL<PREC>::A_PLANE, A_BYTES = PJL<PREC>::A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* smA = smem;
    float* ldsC = reinterpret_cast<float*>(smem + A_BYTES);            // [64][32] cos, then [64][32] sin
    float* ldsS = ldsC + PBM * 32;

    const TileLoc t = locate_tile(a.rs, blockIdx.x, PBM);
    const int len = a.rs.len[t.seg];
    if (t.r0 >= len) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const int image = t.seg & 1, n = image ? pa.n1 : pa.n0;
    const int srow = tid >> 3, sslot = tid & 7;
    const int r = t.r0 + srow, rc = r < len ? r : len - 1;             // rows past the segment's count: a finite copy of its last row (never stored)
    const long long in_row = (long long)t.pair * n + rc;
    // ---- descriptor rows (the x tile) requested first: cold, in flight under the table arithmetic
    const float* src = (image ? pa.desc1 : pa.desc0) + in_row * 256 + sslot * EPC;
    f32x4 hreg[STAGES][NV];
#pragma unroll
    for (int st = 0; st < STAGES; ++st)
#pragma unroll
        for (int j = 0; j < NV; ++j) hreg[st][j] = *reinterpret_cast<const f32x4*>(src + st * KE + 4 * j);
    // ---- rotary rows: thread -> keypoint srow, frequencies 4 sslot .. 4 sslot + 3 (prep_kernel's expressions)
    {
        const float* kp = (image ? pa.kpts1 : pa.kpts0) + in_row * 2;
        const float* szp = image ? pa.size1 : pa.size0;
        float sx, sy;
        if (szp) { sx = szp[t.pair * 2]; sy = szp[t.pair * 2 + 1]; }
        else { const float* bb = pa.bbox + t.seg * 4; sx = 1.f + bb[2] - bb[0]; sy = 1.f + bb[3] - bb[1]; }
        const float scale = fmaxf(sx, sy) / 2.f;
        float kn[4];
        kn[0] = (kp[0] - sx / 2.f) / scale;
        kn[1] = (kp[1] - sy / 2.f) / scale;
        if (pa.pos_dim == 4) {
            kn[2] = (image ? pa.scales1 : pa.scales0)[in_row];
            kn[3] = (image ? pa.oris1 : pa.oris0)[in_row];
        }
        f32x4 c4, s4;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float p = 0.f;
            for (int c = 0; c < pa.pos_dim; ++c) p += kn[c] * pa.Wr[(sslot * 4 + f) * pa.pos_dim + c];
            c4[f] = cosf(p); s4[f] = sinf(p);
        }
        *reinterpret_cast<f32x4*>(ldsC + srow * 32 + sslot * 4) = c4;
        *reinterpret_cast<f32x4*>(ldsS + srow * 32 + sslot * 4) = s4;
        if (r < len) {
            const long long grow = t.grow0 + srow;
            *reinterpret_cast<f32x4*>(pa.cosb + grow * 32 + sslot * 4) = c4;
            *reinterpret_cast<f32x4*>(pa.sinb + grow * 32 + sslot * 4) = s4;
            if (sslot == 0) pa.ind[grow] = r;
        }
    }
    // ---- x tile -> fp32 residual stream + operand planes in LDS (proj_load_tile's conversion)
    {
        const int off = pj_tile_off(srow, sslot);
        float* xdst = pa.X + (long long)(t.grow0 + srow) * 256 + sslot * EPC;
#pragma unroll
        for (int st = 0; st < STAGES; ++st) {
            char* tile = smA + st * TILE;
            if (r < len) {
#pragma unroll
                for (int j = 0; j < NV; ++j) *reinterpret_cast<f32x4*>(xdst + st * KE + 4 * j) = hreg[st][j];
            }
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(tile + off) = hreg[st][0];
            } else if constexpr (PJ<PREC>::APART == 2) {
                u32x4 hi, lo;
                split8<Tag>(hreg[st][0], hreg[st][1], hi, lo);
                *reinterpret_cast<u32x4*>(tile + off) = hi;
                *reinterpret_cast<u32x4*>(tile + A_PLANE + off) = lo;
            } else {
                *reinterpret_cast<u32x4*>(tile + off) = pack8<Tag>(hreg[st][0], hreg[st][1]);
            }
        }
    }
    __syncthreads();   // the tile's rotary rows are in LDS
    RopeRows<4> rr;
    {
        const int f0 = ((32 * w) & 63) / 2 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int row = pj_row<4>(mt, lr);
            rr.c[mt] = *reinterpret_cast<const f32x4*>(ldsC + row * 32 + f0);
            rr.s[mt] = *reinterpret_cast<const f32x4*>(ldsS + row * 32 + f0);
        }
    }
    proj_compute<PREC, TA, 3, 2>(a, t, smA, 0, &rr);
}
template <int PREC, class TA> static hipError_t launch_proj_first_t(const ProjArgs& a, const PrepArgs& pa, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    constexpr int smem = PJL<PREC>::A_BYTES + 2 * PBM * 32 * 4;
    auto kern = proj_first_kernel<PREC, TA>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / PBM), dim3(PTHREADS), smem, s, a, pa);
    return hipGetLastError();
}
hipError_t launch_proj_first(int prec, int attn_prec, const ProjArgs& a, const PrepArgs& pa, hipStream_t s) {
    if (!(a.Nout == 768 && a.n_qk_groups == 2 && a.cosb && a.sinb) || pa.input_dim != 256) return hipErrorInvalidValue;
    if (prec == PREC_F32 && attn_prec == PREC_F32) return launch_proj_first_t<PREC_F32, float>(a, pa, s);
    if (prec == PREC_BF16 && attn_prec == PREC_BF16) return launch_proj_first_t<PREC_BF16, bf16_t>(a, pa, s);
    if (prec == PREC_F16 && attn_prec == PREC_F16) return launch_proj_first_t<PREC_F16, f16_t>(a, pa, s);
    if (prec == PREC_F16X3 && attn_prec == PREC_F16X3) return a.plane > 0 ? launch_proj_first_t<PREC_F16X3, f16_t>(a, pa, s) : hipErrorInvalidValue;
    if (prec == PREC_F16X3 && attn_prec == PREC_F16) return launch_proj_first_t<PREC_QKV_F16W2, f16_t>(a, pa, s);
    return hipErrorInvalidValue;
}

// ---- the SelfBlock projection behind a pruning step, with the compaction folded in (lg_kernels.h GatherArgs): proj_first_kernel's shape — the workgroup fetches the
// residual rows and the rotary rows of its 64 NEW keypoint rows from wherever they lived (src map), stores them at their new place in the other buffer set and
// projects them.  The same fp32 values reach the same tile positions as after an in-place compaction, so q / k / v are bit-identical to that path.
template <int PREC, class TA>
__global__ __launch_bounds__(PTHREADS) void proj_gather_kernel(ProjArgs a, GatherArgs ga) {
    typedef typename PJ<PREC>::Tag Tag;
    constexpr int EPC = Tag::EPC, NV = EPC / 4;
    constexpr int KE = PJL<PREC>::KE, STAGES = PJL<PREC>::STAGES, TILE = PJL<PREC>::TILE, A_PLANE = PJL<PREC>::A_PLANE, A_BYTES = PJL<PREC>::A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* smA = smem;
    float* ldsC = reinterpret_cast<float*>(smem + A_BYTES);            // [64][32] cos, then [64][32] sin
    float* ldsS = ldsC + PBM * 32;

    const TileLoc t = locate_tile(a.rs, blockIdx.x, PBM);
    const int len = a.rs.len[t.seg];
    if (t.r0 >= len) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const int srow = tid >> 3, sslot = tid & 7;
    const int r = t.r0 + srow, rc = r < len ? r : len - 1;             // rows past the segment's count: a finite copy of its last row (never stored)
    const long long base = t.grow0 - t.r0;                             // first row of the segment
    const int sr = ga.len_old[t.seg] >= 0 ? ga.src[base + rc] : rc;
    const long long in_row = base + sr;
    const float* src = ga.Xold + in_row * 256 + sslot * EPC;
    f32x4 hreg[STAGES][NV];
#pragma unroll
    for (int st = 0; st < STAGES; ++st)
#pragma unroll
        for (int j = 0; j < NV; ++j) hreg[st][j] = *reinterpret_cast<const f32x4*>(src + st * KE + 4 * j);
    {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(ga.cos_old + in_row * 32 + sslot * 4);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(ga.sin_old + in_row * 32 + sslot * 4);
        *reinterpret_cast<f32x4*>(ldsC + srow * 32 + sslot * 4) = c4;
        *reinterpret_cast<f32x4*>(ldsS + srow * 32 + sslot * 4) = s4;
        if (r < len) {
            const long long grow = t.grow0 + srow;
            *reinterpret_cast<f32x4*>(ga.cos_new + grow * 32 + sslot * 4) = c4;
            *reinterpret_cast<f32x4*>(ga.sin_new + grow * 32 + sslot * 4) = s4;
        }
    }
    {
        const int off = pj_tile_off(srow, sslot);
        float* xdst = ga.Xnew + (long long)(t.grow0 + srow) * 256 + sslot * EPC;
#pragma unroll
        for (int st = 0; st < STAGES; ++st) {
            char* tile = smA + st * TILE;
            if (r < len) {
#pragma unroll
                for (int j = 0; j < NV; ++j) *reinterpret_cast<f32x4*>(xdst + st * KE + 4 * j) = hreg[st][j];
            }
            if constexpr (PREC == PREC_F32) {
                *reinterpret_cast<f32x4*>(tile + off) = hreg[st][0];
            } else if constexpr (PJ<PREC>::APART == 2) {
                u32x4 hi, lo;
                split8<Tag>(hreg[st][0], hreg[st][1], hi, lo);
                *reinterpret_cast<u32x4*>(tile + off) = hi;
                *reinterpret_cast<u32x4*>(tile + A_PLANE + off) = lo;
            } else {
                *reinterpret_cast<u32x4*>(tile + off) = pack8<Tag>(hreg[st][0], hreg[st][1]);
            }
        }
    }
    __syncthreads();   // the tile's rotary rows are in LDS
    RopeRows<4> rr;
    {
        const int f0 = ((32 * w) & 63) / 2 + 4 * g;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int row = pj_row<4>(mt, lr);
            rr.c[mt] = *reinterpret_cast<const f32x4*>(ldsC + row * 32 + f0);
            rr.s[mt] = *reinterpret_cast<const f32x4*>(ldsS + row * 32 + f0);
        }
    }
    proj_compute<PREC, TA, 3, 2>(a, t, smA, 0, &rr);
}
template <int PREC, class TA> static hipError_t launch_proj_gather_t(const ProjArgs& a, const GatherArgs& g, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    constexpr int smem = PJL<PREC>::A_BYTES + 2 * PBM * 32 * 4;
    auto kern = proj_gather_kernel<PREC, TA>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / PBM), dim3(PTHREADS), smem, s, a, g);
    return hipGetLastError();
}
hipError_t launch_proj_gather(int prec, int attn_prec, const ProjArgs& a, const GatherArgs& g, hipStream_t s) {
    if (!(a.Nout == 768 && a.n_qk_groups == 2) || !g.Xold || !g.Xnew || !g.src || !g.len_old) return hipErrorInvalidValue;
    if (prec == PREC_F32 && attn_prec == PREC_F32) return launch_proj_gather_t<PREC_F32, float>(a, g, s);
    if (prec == PREC_BF16 && attn_prec == PREC_BF16) return launch_proj_gather_t<PREC_BF16, bf16_t>(a, g, s);
    if (prec == PREC_F16 && attn_prec == PREC_F16) return launch_proj_gather_t<PREC_F16, f16_t>(a, g, s);
    if (prec == PREC_F16X3 && attn_prec == PREC_F16X3) return a.plane > 0 ? launch_proj_gather_t<PREC_F16X3, f16_t>(a, g, s) : hipErrorInvalidValue;
    if (prec == PREC_F16X3 && attn_prec == PREC_F16) return launch_proj_gather_t<PREC_QKV_F16W2, f16_t>(a, g, s);
    return hipErrorInvalidValue;
}

// final projection of the log assignment as its own launch (adaptive depth: the weights of the layer each pair stopped at)
template <int PREC>
__global__ __launch_bounds__(PTHREADS) void final_proj_kernel(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TileLoc t = locate_tile(a.rs, blockIdx.x, PBM);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    proj_load_tile<PREC>((a.xsel && a.xsel[t.pair]) ? a.X2 : a.X, t, smem);   // gather path: a pair's rows live in the buffer set they were in when it stopped
    final_compute<PREC>(a, t, smem);
}
template <int PREC> static hipError_t launch_final_t(const FinalArgs& a, hipStream_t s) {
    constexpr int smem = PJL<PREC>::A_BYTES;
    auto kern = final_proj_kernel<PREC>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.R / PBM), dim3(PTHREADS), smem, s, a);
    return hipGetLastError();
}
hipError_t launch_final_proj(int prec, const FinalArgs& a, hipStream_t s) {
    switch (prec) {
        case PREC_F32: return launch_final_t<PREC_F32>(a, s);
        case PREC_BF16: return launch_final_t<PREC_BF16>(a, s);
        case PREC_F16: return launch_final_t<PREC_F16>(a, s);
        case PREC_F16X3: return launch_final_t<PREC_F16X3>(a, s);
    }
    return hipErrorInvalidValue;
}

template <int PREC, class TA, int NTP, int NPASS> static hipError_t launch_proj_t(const ProjArgs& a, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    constexpr int smem = PJL<PREC>::A_BYTES;
    auto kern = proj_kernel<PREC, TA, NTP, NPASS>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(R / PBM), dim3(PTHREADS), smem, s, a);
    return hipGetLastError();
}
template <int PREC, class TA> static hipError_t launch_proj_n(const ProjArgs& a, hipStream_t s) {
    if (a.Nout == 768 && a.n_qk_groups == 2 && a.cosb && a.sinb) return launch_proj_t<PREC, TA, 3, 2>(a, s);
    if (a.Nout == 512 && a.n_qk_groups == 1) return launch_proj_t<PREC, TA, 2, 2>(a, s);
    return hipErrorInvalidValue;
}
// (linear precision, attention precision) pairs the engine runs: every single-plane precision with its own element type,
// f16x3 with split q / k / v (default) or with one f16 plane (attention_precision fp16, lg_proj_body.h PREC_QKV_F16W2)
hipError_t launch_proj(int prec, int attn_prec, const ProjArgs& a, hipStream_t s) {
    if (prec == PREC_F32 && attn_prec == PREC_F32) return launch_proj_n<PREC_F32, float>(a, s);
    if (prec == PREC_BF16 && attn_prec == PREC_BF16) return launch_proj_n<PREC_BF16, bf16_t>(a, s);
    if (prec == PREC_F16 && attn_prec == PREC_F16) return launch_proj_n<PREC_F16, f16_t>(a, s);
    if (prec == PREC_F16X3 && attn_prec == PREC_F16X3) return a.plane > 0 ? launch_proj_n<PREC_F16X3, f16_t>(a, s) : hipErrorInvalidValue;
    if (prec == PREC_F16X3 && attn_prec == PREC_F16) return launch_proj_n<PREC_QKV_F16W2, f16_t>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace lg
