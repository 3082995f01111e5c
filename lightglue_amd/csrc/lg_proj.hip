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

// final projection of the log assignment as its own launch (adaptive depth: the weights of the layer each pair stopped at)
template <int PREC>
__global__ __launch_bounds__(PTHREADS) void final_proj_kernel(FinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TileLoc t = locate_tile(a.rs, blockIdx.x, PBM);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    proj_load_tile<PREC>(a.X, t, smem);
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
