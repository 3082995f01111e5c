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
