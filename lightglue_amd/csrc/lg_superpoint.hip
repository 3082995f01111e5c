// lightglue_amd — SuperPoint descriptor head: the step that PRODUCES the matcher's
// [B, N, 256] descriptor tensors from the dense descriptor map (ref superpoint.py:80-95
// sample_descriptors, :216-228 dense L2 normalisation + per-keypoint sampling + transpose).
// HBM-bound gather work, two kernels:
//   sp_dense_kernel   reads the NCHW map once (coalesced along w), optionally L2-normalises every
//                     location over its 256 channels (ref :218 F.normalize(dim=1)) and writes it
//                     location-major (NHWC) through an LDS tile — the transposition the reference
//                     leaves to grid_sample's strided reads.  Algorithmic bytes: 2 * 4 * C * h * w.
//   sp_sample_kernel  one wave per keypoint: bilinear interpolation of the 4 neighbouring locations
//                     (align_corners=True, zero padding: ref :85-91 + grid_sample), each a contiguous
//                     1 KB row of the NHWC map, then the final L2 normalisation (ref :92-94); writes
//                     [B][N][C] directly (the layout ref :228 transposes to).
//                     Algorithmic bytes per keypoint: 4 * 1 KB read + 1 KB written.
#include "lg_kernels.h"

namespace lg {

constexpr int SPC = 256;   // descriptor_dim of SuperPoint (ref superpoint.py:107)
constexpr int SPT = 32;    // locations per workgroup tile (32 x 257 floats of LDS)

__global__ __launch_bounds__(256) void sp_dense_kernel(SpArgs a) {
    __shared__ float tile[SPT][SPC + 1];
    __shared__ float part[8][SPT];
    const int b = blockIdx.y, loc0 = blockIdx.x * SPT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = a.h * a.w, l = lane & 31, half = lane >> 5, loc = loc0 + l;
    const float* src = a.desc_map + (long long)b * SPC * hw;
    float ss = 0.f;
    for (int c = wave * 2 + half; c < SPC; c += 8) {     // half-wave reads 32 consecutive locations (128 B) of one channel
        const float v = loc < hw ? src[(long long)c * hw + loc] : 0.f;
        tile[l][c] = v;
        ss += v * v;
    }
    part[wave * 2 + half][l] = ss;
    __syncthreads();
    float* dst = a.nhwc + ((long long)b * hw + loc0) * SPC;
    for (int r = wave; r < SPT; r += 4) {                // wave writes one location = 1 KB contiguous
        if (loc0 + r >= hw) break;
        float inv = 1.f;
        if (a.normalize_dense) {                         // F.normalize: x / max(||x||_2, 1e-12)
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) q += part[k][r];
            inv = 1.f / fmaxf(sqrtf(q), 1e-12f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[(long long)r * SPC + lane + 64 * k] = tile[r][lane + 64 * k] * inv;
    }
}

__global__ __launch_bounds__(256) void sp_sample_kernel(SpArgs a) {
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    const int n = a.num ? min(max(a.num[b], 0), a.N) : a.N;
    if (i >= a.N) return;
    float* out = a.out + ((long long)b * a.N + i) * SPC + lane * 4;
    if (i >= n) { *reinterpret_cast<f32x4*>(out) = f32x4{0.f, 0.f, 0.f, 0.f}; return; }   // padding row of a ragged batch
    const float* kp = a.keypoints + ((long long)b * a.N + i) * 2;
    const float s = (float)a.s;
    // ref :83-90: (k - s/2 + 0.5) / (w*s - s/2 - 0.5) in [0,1], *2-1, then align_corners=True un-normalisation
    // ((g + 1) / 2 * (w - 1)), evaluated in the reference's order
    const float gx = (kp[0] - s / 2.f + 0.5f) / ((float)a.w * s - s / 2.f - 0.5f) * 2.f - 1.f;
    const float gy = (kp[1] - s / 2.f + 0.5f) / ((float)a.h * s - s / 2.f - 0.5f) * 2.f - 1.f;
    const float ix = (gx + 1.f) / 2.f * (float)(a.w - 1), iy = (gy + 1.f) / 2.f * (float)(a.h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};   // nw ne sw se
    const float* map = a.nhwc + (long long)b * a.h * a.w * SPC + lane * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + (k & 1), y = y0 + (k >> 1);
        if (x >= 0 && x < a.w && y >= 0 && y < a.h) {    // padding_mode="zeros"
            const f32x4 v = *reinterpret_cast<const f32x4*>(map + ((long long)y * a.w + x) * SPC);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += v[c] * wgt[k];
        }
    }
    const float nrm = sqrtf(wave_sum(acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3]));
    const float inv = 1.f / fmaxf(nrm, 1e-12f);          // ref :92-94
    *reinterpret_cast<f32x4*>(out) = f32x4{acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
}

hipError_t launch_sp_sample(const SpArgs& a, hipStream_t s) {
    const int hw = a.h * a.w;
    hipLaunchKernelGGL(sp_dense_kernel, dim3((hw + SPT - 1) / SPT, a.B), dim3(256), 0, s, a);
    if (a.N > 0) hipLaunchKernelGGL(sp_sample_kernel, dim3((a.N + 3) / 4, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace lg

// ====================================================================================================================
// Keypoint extraction: non-maximum suppression of the dense score map (ref superpoint.py:52-70
// simple_nms: three rounds of (2r+1)^2 max-pooling with equality tests), border removal (:189-194), thresholding and
// row-major compaction (:197-204, the order torch.where produces), optional top-k by score (:207-215, top_k_keypoints
// :73-77) and the (y, x) -> (x, y) float conversion (:218).  All comparisons are exact, so the result is bit-identical
// to the reference's (ties inside top-k, which torch leaves unspecified, break towards the lower row-major index).
// HBM-trivial work (a 1024 x 768 map is 3 MB): written for clarity — one 64 x 16 pixel tile per workgroup with a halo
// of 2r in LDS, separable max / or passes — not tuned.
namespace lg {

constexpr int NTX = 64, NTY = 16, NR = 4;                 // tile, maximum radius
constexpr int NLW = NTX + 4 * NR, NLH = NTY + 4 * NR;     // tile + halo of 2r on every side (80 x 32)

// for every (ly, lx) of the local region [y0, y1) x [x0, x1): body(ly, lx); threads stride over the region
template <class F> __device__ __forceinline__ void for_region(int y0, int y1, int x0, int x1, F body) {
    const int w = x1 - x0, n = (y1 - y0) * w;
    for (int i = threadIdx.x; i < n; i += blockDim.x) body(y0 + i / w, x0 + i % w);
}

// mode 0: mask_out = (S == maxpool(S))                                                     (ref :62)
// mode 1: one suppression round (ref :64-68) mask_in -> mask_out
// mode 2: the same round, but writes nms = mask ? S : 0 (ref :69) instead of the mask
__global__ __launch_bounds__(256) void sp_nms_kernel(SpDetectArgs a, int mode, const unsigned char* mask_in, unsigned char* mask_out) {
    __shared__ float sS[NLH][NLW], sSS[NLH][NLW], sH[NLH][NLW];
    __shared__ unsigned char sM[NLH][NLW], sHM[NLH][NLW], sSupp[NLH][NLW];
    const int b = blockIdx.z, ty0 = blockIdx.y * NTY, tx0 = blockIdx.x * NTX, r = a.radius, O = 2 * NR;   // O: local origin offset
    const long long img = (long long)b * a.H * a.W;
    auto inside = [&](int ly, int lx) { const int y = ty0 + ly - O, x = tx0 + lx - O; return y >= 0 && y < a.H && x >= 0 && x < a.W; };
    // ---- load S (and the mask) for tile +- 2r; outside the image: -inf / 0 (max_pool2d pads with -inf)
    for_region(O - 2 * r, O + NTY + 2 * r, O - 2 * r, O + NTX + 2 * r, [&](int ly, int lx) {
        const bool in = inside(ly, lx);
        const long long idx = img + (long long)(ty0 + ly - O) * a.W + (tx0 + lx - O);
        sS[ly][lx] = in ? a.scores[idx] : -INFINITY;
        sM[ly][lx] = (mode != 0 && in) ? mask_in[idx] : 0;
    });
    __syncthreads();
    if (mode == 0) {
        for_region(O - r, O + NTY + r, O, O + NTX, [&](int ly, int lx) {
            float m = -INFINITY;
            for (int d = -r; d <= r; ++d) m = fmaxf(m, sS[ly][lx + d]);
            sH[ly][lx] = m;
        });
        __syncthreads();
        for_region(O, O + NTY, O, O + NTX, [&](int ly, int lx) {
            if (!inside(ly, lx)) return;
            float m = -INFINITY;
            for (int d = -r; d <= r; ++d) m = fmaxf(m, sH[ly + d][lx]);
            mask_out[img + (long long)(ty0 + ly - O) * a.W + (tx0 + lx - O)] = sS[ly][lx] == m;
        });
        return;
    }
    // ---- supp = maxpool(mask) > 0 on tile +- r (separable or)
    for_region(O - 2 * r, O + NTY + 2 * r, O - r, O + NTX + r, [&](int ly, int lx) {
        unsigned char v = 0;
        for (int d = -r; d <= r; ++d) v |= sM[ly][lx + d];
        sHM[ly][lx] = v;
    });
    __syncthreads();
    for_region(O - r, O + NTY + r, O - r, O + NTX + r, [&](int ly, int lx) {
        unsigned char v = 0;
        for (int d = -r; d <= r; ++d) v |= sHM[ly + d][lx];
        sSupp[ly][lx] = v;
        sSS[ly][lx] = !inside(ly, lx) ? -INFINITY : (v ? 0.f : sS[ly][lx]);   // supp_scores (ref :66)
    });
    __syncthreads();
    // ---- new_max_mask = supp_scores == maxpool(supp_scores) on the tile
    for_region(O - r, O + NTY + r, O, O + NTX, [&](int ly, int lx) {
        float m = -INFINITY;
        for (int d = -r; d <= r; ++d) m = fmaxf(m, sSS[ly][lx + d]);
        sH[ly][lx] = m;
    });
    __syncthreads();
    for_region(O, O + NTY, O, O + NTX, [&](int ly, int lx) {
        if (!inside(ly, lx)) return;
        float m = -INFINITY;
        for (int d = -r; d <= r; ++d) m = fmaxf(m, sH[ly + d][lx]);
        const bool keep = sM[ly][lx] || ((sSS[ly][lx] == m) && !sSupp[ly][lx]);   // ref :67-68
        const long long idx = img + (long long)(ty0 + ly - O) * a.W + (tx0 + lx - O);
        if (mode == 1) mask_out[idx] = keep;
        else a.nms[idx] = keep ? sS[ly][lx] : 0.f;                                  // ref :69
    });
}

// value the reference thresholds: borders are overwritten with -1 (ref :189-194)
__device__ __forceinline__ float sp_value(const SpDetectArgs& a, const float* nms, int y, int x) {
    const bool border = a.border > 0 && (y < a.border || x < a.border || y >= a.H - a.border || x >= a.W - a.border);
    return border ? -1.f : nms[(long long)y * a.W + x];
}

// per image row: number of pixels above the threshold.  grid (H, B)
__global__ __launch_bounds__(256) void sp_row_count_kernel(SpDetectArgs a) {
    const int y = blockIdx.x, b = blockIdx.y;
    const float* nms = a.nms + (long long)b * a.H * a.W;
    int cnt = 0;
    for (int x = threadIdx.x; x < a.W; x += 256) cnt += sp_value(a, nms, y, x) > a.threshold;
    cnt = (int)wave_sum((float)cnt);   // exact: counts <= 2048 per wave
    __shared__ int sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) a.row_counts[b * a.H + y] = sh[0] + sh[1] + sh[2] + sh[3];
}

// row-major compaction (the order of torch.where, ref :197): grid (H, B)
__global__ __launch_bounds__(256) void sp_compact_kernel(SpDetectArgs a) {
    const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* nms = a.nms + (long long)b * a.H * a.W;
    __shared__ int sh[4];
    __shared__ int base_sh;
    int pre = 0;
    for (int yy = tid; yy < y; yy += 256) pre += a.row_counts[b * a.H + yy];
    pre = (int)wave_sum((float)pre);   // exact below 2^24 candidates per image
    if (lane == 0) sh[wave] = pre;
    __syncthreads();
    if (tid == 0) base_sh = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    int running = base_sh;
    for (int x0 = 0; x0 < a.W; x0 += 256) {
        const int x = x0 + tid;
        const float v = x < a.W ? sp_value(a, nms, y, x) : -INFINITY;
        const bool hit = x < a.W && v > a.threshold;
        const unsigned long long bal = __ballot(hit);
        __syncthreads();
        if (lane == 0) sh[wave] = __popcll(bal);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += sh[w];
        off += __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && off < a.max_candidates) {
            a.cand_xy[(long long)b * a.max_candidates + off] = (y << 16) | x;
            a.cand_score[(long long)b * a.max_candidates + off] = v;
        }
        running += sh[0] + sh[1] + sh[2] + sh[3];
    }
    if (y == a.H - 1 && tid == 0) a.cand_total[b] = running;
}

// order-preserving key of a float: larger float <-> larger unsigned
__device__ __forceinline__ unsigned sp_key(float v) { const unsigned u = __float_as_uint(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// per image: keep everything (row-major order) or the top-k by score, sorted descending (ref :73-77).  grid (B), 1024 threads
__global__ __launch_bounds__(1024) void sp_select_kernel(SpDetectArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.totals && tid == 0) a.totals[b] = a.cand_total[b];
    const int total = min(a.cand_total[b], a.max_candidates);
    const int* cxy = a.cand_xy + (long long)b * a.max_candidates;
    const float* csc = a.cand_score + (long long)b * a.max_candidates;
    float* okp = a.keypoints + (long long)b * a.capacity * 2;
    float* osc = a.kp_scores + (long long)b * a.capacity;
    const int K = a.max_keypoints;
    if (K <= 0 || K >= total) {   // ref :74-75: fewer candidates than k -> unchanged, unsorted
        const int n = min(total, a.capacity);
        for (int i = tid; i < n; i += 1024) {
            const int p = cxy[i];
            okp[2 * i] = (float)(p & 0xFFFF); okp[2 * i + 1] = (float)(p >> 16);   // (x, y), ref :218
            osc[i] = csc[i];
        }
        if (tid == 0) a.counts[b] = n;
        return;
    }
    // ---- radix select: key of the K-th largest score, 8 bits per pass from the top
    __shared__ unsigned hist[256];
    __shared__ unsigned prefix_sh, need_sh;
    unsigned prefix = 0, need = (unsigned)K;    // among keys matching `prefix` on the bits fixed so far, we still need `need`
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned fixed_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = tid; i < total; i += 1024) {
            const unsigned k = sp_key(csc[i]);
            if ((k & fixed_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = 0; int d = 255;
            for (; d > 0; --d) { if (acc + hist[d] >= need) break; acc += hist[d]; }
            prefix_sh = prefix | ((unsigned)d << shift); need_sh = need - acc;
        }
        __syncthreads();
        prefix = prefix_sh; need = need_sh;
        __syncthreads();
    }
    // prefix = key of the K-th largest; take every key > prefix and the first `need` keys == prefix (row-major order)
    __shared__ unsigned skey[SP_TOPK_MAX];
    __shared__ int sidx[SP_TOPK_MAX];
    __shared__ int cnt_gt, cnt_eq_base;
    __shared__ int wsum[16];
    if (tid == 0) { cnt_gt = 0; }
    __syncthreads();
    for (int i = tid; i < total; i += 1024) {
        const unsigned k = sp_key(csc[i]);
        if (k > prefix) { const int p = atomicAdd(&cnt_gt, 1); skey[p] = k; sidx[p] = i; }
    }
    __syncthreads();
    if (tid == 0) cnt_eq_base = cnt_gt;
    __syncthreads();
    // equal keys in index order: ordered ballot scan over the candidate list
    int taken = 0;
    for (int i0 = 0; i0 < total && taken < (int)need; i0 += 1024) {
        const int i = i0 + tid;
        const bool eq = i < total && sp_key(csc[i]) == prefix;
        const unsigned long long bal = __ballot(eq);
        __syncthreads();
        if ((tid & 63) == 0) wsum[tid >> 6] = __popcll(bal);
        __syncthreads();
        int off = taken;
        for (int w = 0; w < (tid >> 6); ++w) off += wsum[w];
        off += __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if (eq && off < (int)need) { skey[cnt_eq_base + off] = prefix; sidx[cnt_eq_base + off] = i; }
        int all = 0;
        for (int w = 0; w < 16; ++w) all += wsum[w];
        taken += all;
    }
    __syncthreads();
    // ---- bitonic sort of the K selected entries: score descending, index ascending among equals
    int P = 1; while (P < K) P <<= 1;
    for (int i = K + tid; i < P; i += 1024) { skey[i] = 0u; sidx[i] = 0x7FFFFFFF; }
    __syncthreads();
    auto before = [&](int i, int j) { return skey[i] > skey[j] || (skey[i] == skey[j] && sidx[i] < sidx[j]); };   // i sorts before j
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < P / 2; t += 1024) {
                const int lo = (t / stride) * 2 * stride + (t % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);   // first half of each `size` block in order
                if (before(hi, lo) == up) {
                    const unsigned tk = skey[lo]; skey[lo] = skey[hi]; skey[hi] = tk;
                    const int ti = sidx[lo]; sidx[lo] = sidx[hi]; sidx[hi] = ti;
                }
            }
            __syncthreads();
        }
    const int n = min(K, a.capacity);
    for (int i = tid; i < n; i += 1024) {
        const int c = sidx[i], p = cxy[c];
        okp[2 * i] = (float)(p & 0xFFFF); okp[2 * i + 1] = (float)(p >> 16);
        osc[i] = csc[c];
    }
    if (tid == 0) a.counts[b] = n;
}

hipError_t launch_sp_detect(const SpDetectArgs& a, hipStream_t s) {
    const dim3 grid((a.W + NTX - 1) / NTX, (a.H + NTY - 1) / NTY, a.B);
    hipLaunchKernelGGL(sp_nms_kernel, grid, dim3(256), 0, s, a, 0, (const unsigned char*)nullptr, a.mask_a);
    hipLaunchKernelGGL(sp_nms_kernel, grid, dim3(256), 0, s, a, 1, (const unsigned char*)a.mask_a, a.mask_b);   // ref :63 first round
    hipLaunchKernelGGL(sp_nms_kernel, grid, dim3(256), 0, s, a, 2, (const unsigned char*)a.mask_b, a.mask_a);   // second round + :69
    hipLaunchKernelGGL(sp_row_count_kernel, dim3(a.H, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sp_compact_kernel, dim3(a.H, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sp_select_kernel, dim3(a.B), dim3(1024), 0, s, a);
    return hipGetLastError();
}

}  // namespace lg
