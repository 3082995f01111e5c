// lightglue_amd — log-assignment and match filtering (ref lightglue.py:265-277
// sigmoid_log_double_softmax, :302-318 filter_matches, :593-614 output assembly).
// The similarity matrix sim[pair][a][b] (fp32) is produced by launch_sim; everything here is
// HBM-bound streaming over it:
//   pass 1  row log-sum-exp (one wave per row) and column log-sum-exp (64-column strips)
//   pass 2  score(a,b) = ((sim - lse_r[a]) + (sim - lse_c[b])) + (ls0[a] + ls1[b])   [ref :270-274]
//           row max/argmax and column max/argmax with first-index tie-break (torch.max semantics)
//   final   mutual check, exp, threshold, scatter through the index sets into ORIGINAL index
//           space (un-pruning, ref :605-614) and the compact sorted match list (ref :593-602).
// The dustbin row/column (ref :275-276) never influence any output of forward and
// are not materialised.
#include "lg_kernels.h"

namespace lg {

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; s = 0.f; return; }
    s = s * expf(m - mn) + s2 * expf(m2 - mn);
    m = mn;
}

// ---- pass 1a: row LSE.  grid (cap0/4, B), one wave per row
__global__ __launch_bounds__(256) void row_lse_kernel(AssignArgs a) {
    const int pair = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    if (r >= len0) return;
    const float* row = a.sim + ((long long)pair * a.rs.cap0 + r) * a.rs.cap1;
    float m = -INFINITY;
    for (int c = lane * 4; c < len1; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c + i < len1) m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane * 4; c < len1; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c + i < len1) s += expf(v[i] - m);
    }
    s = wave_sum(s);
    if (lane == 0) a.lse_r[(long long)pair * a.rs.cap0 + r] = m + logf(s);
}

// ---- pass 1b: column LSE.  grid (cap1/64, B); thread = (column lane, row group of 4)
__global__ __launch_bounds__(256) void col_lse_kernel(AssignArgs a) {
    const int pair = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    const int c = blockIdx.x * 64 + cl;
    if (blockIdx.x * 64 >= len1) return;
    const float* simp = a.sim + (long long)pair * a.rs.cap0 * a.rs.cap1 + c;
    float m = -INFINITY, s = 0.f;
    for (int r = rg; r < len0; r += 4) {
        const float v = simp[(long long)r * a.rs.cap1];
        if (v > m) { s = s * expf(m - v) + 1.f; m = v; } else s += expf(v - m);
    }
    __shared__ float shm[4][64], shs[4][64];
    shm[rg][cl] = m; shs[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < len1) {
        for (int i = 1; i < 4; ++i) lse_merge(m, s, shm[i][cl], shs[i][cl]);
        a.lse_c[(long long)pair * a.rs.cap1 + c] = m + logf(s);
    }
}

__device__ __forceinline__ float score_of(float sim, float lr, float lc, float cert) {
    return ((sim - lr) + (sim - lc)) + cert;   // ref :271-274 evaluation order
}

// ---- pass 2a: row max / argmax of the score matrix.  one wave per row
__global__ __launch_bounds__(256) void row_argmax_kernel(AssignArgs a) {
    const int pair = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    if (r >= len0) return;
    const float* row = a.sim + ((long long)pair * a.rs.cap0 + r) * a.rs.cap1;
    const float* lsec = a.lse_c + (long long)pair * a.rs.cap1;
    const float* ls1 = a.ls + seg_row_base(a.rs, 2 * pair + 1);
    const float lr = a.lse_r[(long long)pair * a.rs.cap0 + r];
    const float l0 = a.ls[seg_row_base(a.rs, 2 * pair) + r];
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane * 4; c < len1; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
        const f32x4 lc = *reinterpret_cast<const f32x4*>(lsec + c);
        const f32x4 l1 = *reinterpret_cast<const f32x4*>(ls1 + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) if (c + i < len1) {
            const float sc = score_of(v[i], lr, lc[i], l0 + l1[i]);
            if (sc > best || (sc == best && c + i < bi)) { best = sc; bi = c + i; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { a.max0[(long long)pair * a.rs.cap0 + r] = best; a.arg0[(long long)pair * a.rs.cap0 + r] = bi; }
}

// ---- pass 2b: column max / argmax.  grid (cap1/64, B)
__global__ __launch_bounds__(256) void col_argmax_kernel(AssignArgs a) {
    const int pair = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    const int c = blockIdx.x * 64 + cl;
    if (blockIdx.x * 64 >= len1) return;
    const float* simp = a.sim + (long long)pair * a.rs.cap0 * a.rs.cap1 + c;
    const float* lser = a.lse_r + (long long)pair * a.rs.cap0;
    const float* ls0 = a.ls + seg_row_base(a.rs, 2 * pair);
    const float lc = c < len1 ? a.lse_c[(long long)pair * a.rs.cap1 + c] : 0.f;
    const float l1 = c < len1 ? a.ls[seg_row_base(a.rs, 2 * pair + 1) + c] : 0.f;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int r = rg; r < len0; r += 4) {   // ascending r per thread: strict '>' keeps the first index
        const float sc = score_of(simp[(long long)r * a.rs.cap1], lser[r], lc, ls0[r] + l1);
        if (sc > best) { best = sc; bi = r; }
    }
    __shared__ float shb[4][64]; __shared__ int shi[4][64];
    shb[rg][cl] = best; shi[rg][cl] = bi;
    __syncthreads();
    if (rg == 0 && c < len1) {
        for (int i = 1; i < 4; ++i) {
            const float ob = shb[i][cl]; const int oi = shi[i][cl];
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        a.max1[(long long)pair * a.rs.cap1 + c] = best; a.arg1[(long long)pair * a.rs.cap1 + c] = bi;
    }
}

// ---- final: one workgroup per pair
__global__ __launch_bounds__(256) void finalize_kernel(AssignArgs a) {
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    const int base0 = seg_row_base(a.rs, 2 * pair), base1 = seg_row_base(a.rs, 2 * pair + 1);
    const float* max0 = a.max0 + (long long)pair * a.rs.cap0;
    const int* arg0 = a.arg0 + (long long)pair * a.rs.cap0;
    const int* arg1 = a.arg1 + (long long)pair * a.rs.cap1;
    int* m0 = a.m0 + (long long)pair * a.n0; float* s0 = a.s0 + (long long)pair * a.n0;
    int* m1 = a.m1 + (long long)pair * a.n1; float* s1 = a.s1 + (long long)pair * a.n1;
    __shared__ int sh_cnt[4];
    if (len0 == 0 || len1 == 0) { if (tid == 0) a.n_matches[pair] = 0; return; }
    // image 1 side (ref :309, :313, :315, :317)
    for (int b = tid; b < len1; b += 256) {
        const int i = arg1[b];
        const bool mutual1 = arg0[i] == b;
        const float e = expf(max0[i]);            // mutual1 implies mutual0(i), so mscores0[i] = exp(max0[i])
        const bool valid1 = mutual1 && (e > a.filter_threshold);
        const int ob = a.ind[base1 + b];
        m1[ob] = valid1 ? a.ind[base0 + i] : -1;
        s1[ob] = mutual1 ? e : 0.f;
    }
    // image 0 side + compact list, ascending a (ref :308, :312, :314, :316, :595-602)
    int running = 0;
    int* ml = a.matches + (long long)pair * a.max_matches * 2;
    float* msl = a.mscores + (long long)pair * a.max_matches;
    for (int a0 = 0; a0 < len0; a0 += 256) {
        const int r = a0 + tid;
        bool valid0 = false; int oa = 0, ob = -1; float e = 0.f;
        if (r < len0) {
            const int j = arg0[r];
            const bool mutual0 = arg1[j] == r;
            e = mutual0 ? expf(max0[r]) : 0.f;
            valid0 = mutual0 && (e > a.filter_threshold);
            oa = a.ind[base0 + r];
            ob = valid0 ? a.ind[base1 + j] : -1;
            m0[oa] = ob;
            s0[oa] = e;
        }
        const unsigned long long bal = __ballot(valid0);
        const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) sh_cnt[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += sh_cnt[w];
        if (valid0) {
            const int k = running + woff + prefix;
            ml[2 * k] = oa; ml[2 * k + 1] = ob; msl[k] = e;
        }
        running += sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
    }
    if (tid == 0) a.n_matches[pair] = running;
}

hipError_t launch_assign(const AssignArgs& a, hipStream_t s) {
    const int B = a.rs.B;
    hipError_t e;
    if (a.n0 > 0) {
        if ((e = hipMemsetAsync(a.m0, 0xFF, sizeof(int) * (size_t)B * a.n0, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(a.s0, 0, sizeof(float) * (size_t)B * a.n0, s)) != hipSuccess) return e;
    }
    if (a.n1 > 0) {
        if ((e = hipMemsetAsync(a.m1, 0xFF, sizeof(int) * (size_t)B * a.n1, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(a.s1, 0, sizeof(float) * (size_t)B * a.n1, s)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(row_lse_kernel, dim3(a.rs.cap0 / 4, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(col_lse_kernel, dim3(a.rs.cap1 / 64, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(row_argmax_kernel, dim3(a.rs.cap0 / 4, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(col_argmax_kernel, dim3(a.rs.cap1 / 64, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace lg
