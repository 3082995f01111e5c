// lightglue_amd — HBM-bound per-keypoint kernels: keypoint normalisation + learnable Fourier
// encoding (ref lightglue.py:32-43, :76-81), LayerNorm(512)+GELU of the FFN (ref :152-157),
// and the 256->1 heads (token confidence ref :89-94, matchability ref :298-299).
// One wave (64 lanes) per keypoint row; 16-byte loads per lane.
#include "lg_kernels.h"

namespace lg {

// ------------------------------------------------------------------ bbox (only when image_size is absent)
// ref :35-36: size = 1 + max - min over the keypoints of the image
__global__ __launch_bounds__(256) void bbox_kernel(PrepArgs a) {
    const int seg = blockIdx.x, image = seg & 1, pair = seg >> 1;
    const int n = image ? a.n1 : a.n0;
    const float* kp = (image ? a.kpts1 : a.kpts0) + (long long)pair * n * 2;
    const int live = a.rs.len[seg];   // ragged batch: only the pair's own keypoints define its bounding box
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = threadIdx.x; i < live; i += 256) {
        const float x = kp[2 * i], y = kp[2 * i + 1];
        mnx = fminf(mnx, x); mny = fminf(mny, y); mxx = fmaxf(mxx, x); mxy = fmaxf(mxy, y);
    }
    __shared__ float sh[4][4];
    mnx = -wave_max(-mnx); mny = -wave_max(-mny); mxx = wave_max(mxx); mxy = wave_max(mxy);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[w][0] = mnx; sh[w][1] = mny; sh[w][2] = mxx; sh[w][3] = mxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { mnx = fminf(mnx, sh[i][0]); mny = fminf(mny, sh[i][1]); mxx = fmaxf(mxx, sh[i][2]); mxy = fmaxf(mxy, sh[i][3]); }
        float* o = a.bbox + seg * 4;
        o[0] = mnx; o[1] = mny; o[2] = mxx; o[3] = mxy;
    }
}

// ------------------------------------------------------------------ prep: one wave per input keypoint
__global__ __launch_bounds__(256) void prep_kernel(PrepArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seg = blockIdx.y, image = seg & 1, pair = seg >> 1;
    const int n = image ? a.n1 : a.n0;
    const int r = blockIdx.x * 4 + wave;
    if (r >= a.rs.len[seg]) return;   // len <= n: rows past a pair's own count are padding (never read)
    const long long grow = seg_row_base(a.rs, seg) + r;
    const float* kp = (image ? a.kpts1 : a.kpts0) + ((long long)pair * n + r) * 2;
    const float* szp = image ? a.size1 : a.size0;
    float sx, sy;
    if (szp) { sx = szp[pair * 2]; sy = szp[pair * 2 + 1]; }
    else { const float* bb = a.bbox + seg * 4; sx = 1.f + bb[2] - bb[0]; sy = 1.f + bb[3] - bb[1]; }
    const float scale = fmaxf(sx, sy) / 2.f;          // ref :41
    float kn[4];
    kn[0] = (kp[0] - sx / 2.f) / scale;               // ref :40, :42
    kn[1] = (kp[1] - sy / 2.f) / scale;
    if (a.pos_dim == 4) {                             // ref :495-501
        kn[2] = (image ? a.scales1 : a.scales0)[(long long)pair * n + r];
        kn[3] = (image ? a.oris1 : a.oris0)[(long long)pair * n + r];
    }
    if (lane < 32) {                                  // ref :78-79 (32 frequencies, shared by all heads)
        float p = 0.f;
        for (int c = 0; c < a.pos_dim; ++c) p += kn[c] * a.Wr[lane * a.pos_dim + c];
        a.cosb[grow * 32 + lane] = cosf(p);
        a.sinb[grow * 32 + lane] = sinf(p);
    }
    if (lane == 0) a.ind[grow] = r;
    // descriptors -> residual stream (ref :502-503, :521-522 identity case) or input-projection staging
    const float* d = (image ? a.desc1 : a.desc0) + ((long long)pair * n + r) * a.input_dim;
    float* dst = (a.input_dim == 256) ? a.X + grow * 256 : a.Xin + grow * a.input_dim;
    for (int c = lane * 4; c < a.input_dim; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(d + c);
}

hipError_t launch_prep_bbox(const PrepArgs& a, hipStream_t s) {
    if (!a.size0 || !a.size1) hipLaunchKernelGGL(bbox_kernel, dim3(2 * a.rs.B), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_prep(const PrepArgs& a, hipStream_t s) {
    if (a.input_dim % 4) return hipErrorInvalidValue;
    const int nseg = 2 * a.rs.B;
    if (!a.size0 || !a.size1) hipLaunchKernelGGL(bbox_kernel, dim3(nseg), dim3(256), 0, s, a);
    const int nmax = a.n0 > a.n1 ? a.n0 : a.n1;
    if (nmax > 0) hipLaunchKernelGGL(prep_kernel, dim3((nmax + 3) / 4, nseg), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ LayerNorm(512) + exact GELU
__global__ __launch_bounds__(256) void ln_gelu_kernel(LnGeluArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grow = blockIdx.x * 4 + wave;
    if (grow >= a.R) return;
    const TileLoc t = locate_tile(a.rs, grow, 1);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (a.rs.active && !a.rs.active[t.pair]) return;
    const float* h = a.h + (long long)grow * 512;
    f32x4 v0 = *reinterpret_cast<const f32x4*>(h + lane * 4);
    f32x4 v1 = *reinterpret_cast<const f32x4*>(h + 256 + lane * 4);
    float sum = v0[0] + v0[1] + v0[2] + v0[3] + v1[0] + v1[1] + v1[2] + v1[3];
    const float mean = wave_sum(sum) * (1.f / 512.f);
    v0 -= mean; v1 -= mean;
    float sq = v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3] + v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3];
    const float var = wave_sum(sq) * (1.f / 512.f);
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + lane * 4), g1 = *reinterpret_cast<const f32x4*>(a.gamma + 256 + lane * 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.beta + lane * 4), b1 = *reinterpret_cast<const f32x4*>(a.beta + 256 + lane * 4);
    f32x4 y0, y1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = v0[i] * rstd * g0[i] + b0[i], u1 = v1[i] * rstd * g1[i] + b1[i];
        y0[i] = 0.5f * u0 * (1.f + erff(u0 * 0.70710678118654752440f));
        y1[i] = 0.5f * u1 * (1.f + erff(u1 * 0.70710678118654752440f));
    }
    float* g = a.g + (long long)grow * 512;
    *reinterpret_cast<f32x4*>(g + lane * 4) = y0;
    *reinterpret_cast<f32x4*>(g + 256 + lane * 4) = y1;
}
hipError_t launch_ln_gelu(const LnGeluArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(ln_gelu_kernel, dim3((a.R + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ 256 -> 1 heads
__device__ __forceinline__ float act_apply(float z, int act) {
    if (act == 1) return 1.f / (1.f + expf(-z));                    // sigmoid
    if (act == 2) return fminf(z, 0.f) - log1pf(expf(-fabsf(z)));   // logsigmoid
    if (act == 3) return fminf(-z, 0.f) - log1pf(expf(-fabsf(z)));  // logsigmoid(-z): dustbin terms (ref :275-276)
    return z;
}
__global__ __launch_bounds__(256) void rowdot_kernel(RowDotArgs a, int R) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grow = blockIdx.x * 4 + wave;
    if (grow >= R) return;
    const TileLoc t = locate_tile(a.rs, grow, 1);
    if (t.r0 >= a.rs.len[t.seg]) return;
    if (!a.ignore_active && a.rs.active && !a.rs.active[t.pair]) return;
    const int layer = a.layer_of_pair ? a.layer_of_pair[t.pair] : 0;
    const f32x4 x = *reinterpret_cast<const f32x4*>(a.X + (long long)grow * 256 + lane * 4);
    {
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.w0 + (long long)layer * a.w_layer_stride + lane * 4);
        const float z = wave_sum(x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3]) + a.b0[layer];
        if (lane == 0) a.out0[grow] = act_apply(z, a.act0);
    }
    if (a.w1) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.w1 + (long long)layer * a.w_layer_stride + lane * 4);
        const float z = wave_sum(x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3]) + a.b1[layer];
        if (lane == 0) a.out1[grow] = act_apply(z, a.act1);
    }
}
hipError_t launch_rowdot(const RowDotArgs& a, hipStream_t s) {
    const int R = a.rs.B * (a.rs.cap0 + a.rs.cap1);
    hipLaunchKernelGGL(rowdot_kernel, dim3((R + 3) / 4), dim3(256), 0, s, a, R);
    return hipGetLastError();
}

}  // namespace lg
