// lightglue_amd — adaptive depth (early stop, ref lightglue.py:547-550 + :645-656) and adaptive
// width (point pruning, ref :551-566 + :636-643) without leaving the device.
//
// adapt_decide  one workgroup per pair: counts low-confidence tokens, takes the stop decision,
//               and — if the pair continues — turns the keep mask of each image into destination
//               indices with a wavefront ballot + popcount prefix (stable: kept points stay in
//               ascending index order, which is what `torch.where` gives the reference).
// adapt_compact rewrites, IN PLACE, the descriptor rows, the rotary tables and the index set of
//               every pruned segment so that later layers see contiguous rows [0, len).
//               In-place is safe because dst <= src: rows are processed in ascending chunks (1024 rows),
//               each chunk is fully read into registers before a barrier and written after it.
//               Parallelism = segments x 11 column slices (8 x 128 B of the descriptor row,
//               cos, sin, index set).
// Per-pair state (len, active, final_layer) lives in device memory; every later kernel reads it,
// so the whole adaptive forward is one stream of launches with no host synchronisation.
#include "lg_kernels.h"

namespace lg {

__global__ __launch_bounds__(256) void adapt_decide_kernel(AdaptArgs a) {
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int sh_cnt[4];
    __shared__ int sh_stop;
    __shared__ int sh_newlen[2];
    if (!a.active[pair]) {
        // a pair that stopped (or lost all points of an image) at an earlier layer: make sure adapt_compact skips it —
        // len_old of the layer that pruned it must not be replayed at every later layer
        if (tid < 2) a.len_old[2 * pair + tid] = -1;
        return;
    }
    const int len0 = a.len[2 * pair], len1 = a.len[2 * pair + 1];

    if (a.do_stop) {
        // ref :653-656: ratio = 1 - #(conf < thr) / (m + n)  [m, n = ORIGINAL counts], float32
        int cnt = 0;
        for (int image = 0; image < 2; ++image) {
            const int L = image ? len1 : len0;
            const float* c = a.conf + seg_row_base(a.rs, 2 * pair + image);
            for (int r = tid; r < L; r += 256) cnt += (c[r] < a.conf_thr) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) sh_cnt[wave] = cnt;
        __syncthreads();
        if (tid == 0) {
            const int total = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
            const float ratio = 1.0f - (float)total / (float)(a.len_orig[2 * pair] + a.len_orig[2 * pair + 1]);
            const int stop = ratio > a.depth_conf;
            sh_stop = stop;
            if (stop) { a.active[pair] = 0; a.final_layer[pair] = a.layer; }
        }
        __syncthreads();
        if (sh_stop) {
            if (tid < 2) a.len_old[2 * pair + tid] = -1;
            return;
        }
    }
    for (int image = 0; image < 2; ++image) {
        const int seg = 2 * pair + image;
        const int L = image ? len1 : len0;
        if (!a.do_prune || L <= a.pruning_min_kpts) {   // ref :551 / :559
            if (tid == 0) { a.len_old[seg] = -1; sh_newlen[image] = L; }   // "pruning not applied at this layer"
            continue;
        }
        const int base = seg_row_base(a.rs, seg);
        int running = 0;                                 // uniform across the block
        for (int r0 = 0; r0 < L; r0 += 256) {
            const int r = r0 + tid;
            bool keep = false;
            if (r < L) {
                keep = a.mscore[base + r] > a.width_conf;               // ref :640 (width_conf = 1 - width_confidence)
                if (a.do_stop) keep = keep || (a.conf[base + r] <= a.conf_thr);  // ref :641-642
            }
            const unsigned long long bal = __ballot(keep);
            const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
            __syncthreads();                             // sh_cnt reuse
            if (lane == 0) sh_cnt[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; ++w) woff += sh_cnt[w];
            if (r < L) a.dst[base + r] = keep ? (running + woff + prefix) : -1;
            running += sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
        }
        __syncthreads();
        if (tid == 0) { a.len_old[seg] = L; a.len[seg] = running; sh_newlen[image] = running; }
    }
    // every point of one image pruned: the reference leaves its layer loop at the next iteration's
    // empty guard (ref :539-540) and returns the empty result with stop = layer + 2 (ref :568-588)
    if (tid == 0 && (sh_newlen[0] == 0 || sh_newlen[1] == 0)) { a.active[pair] = 0; a.final_layer[pair] = a.layer + 1; }
}

__global__ __launch_bounds__(256) void adapt_compact_kernel(AdaptArgs a) {
    const int seg = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x;
    const int Lold = a.len_old[seg];
    if (Lold < 0) return;                    // pruning not applied to this segment at this layer
    const int Lnew = a.len[seg];
    const int base = seg_row_base(a.rs, seg);
    if (slice == 10) {                       // index set + prune counters (ref :555, :558)
        const int pair = seg >> 1, image = seg & 1;
        int* prune = image ? a.prune1 + (long long)pair * a.n1 : a.prune0 + (long long)pair * a.n0;
        for (int r0 = 0; r0 < Lold; r0 += 256) {
            const int r = r0 + tid;
            int d = -1, v = 0;
            if (r < Lold) { d = a.dst[base + r]; v = a.ind[base + r]; }
            __syncthreads();
            if (d >= 0) { a.ind[base + d] = v; prune[v] += 1; }
        }
        return;
    }
    if (Lnew == Lold) return;                // nothing dropped: rows already in place
    float* buf; int ld, col;
    if (slice < 8) { buf = a.X; ld = 256; col = slice * 32; }
    else { buf = slice == 8 ? a.cosb : a.sinb; ld = 32; col = 0; }
    const int sub = tid & 7, rr = tid >> 3;  // 8 lanes x 16 B = one 128-byte row slice; 32 rows per pass
    // CU = passes per iteration.  The kernel is a chain of (read chunk | barrier | write chunk) steps per segment, i.e. bound by the
    // number of HBM round trips, not by bytes: 8 passes (256 rows) per step ran at 0.9 TB/s in the cfg #3 trace (2048-row segments = 8
    // dependent steps); 32 passes (1024 rows, 128 data VGPRs) make it 2 steps
    constexpr int CU = 32;
    for (int r0 = 0; r0 < Lold; r0 += 32 * CU) {
        f32x4 v[CU]; int d[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int r = r0 + u * 32 + rr;
            d[u] = -1;
            if (r < Lold) {
                d[u] = a.dst[base + r];
                if (d[u] >= 0 && d[u] != r) v[u] = *reinterpret_cast<const f32x4*>(buf + (long long)(base + r) * ld + col + sub * 4);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int r = r0 + u * 32 + rr;
            if (d[u] >= 0 && d[u] != r) *reinterpret_cast<f32x4*>(buf + (long long)(base + d[u]) * ld + col + sub * 4) = v[u];
        }
    }
}

hipError_t launch_adapt(const AdaptArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(adapt_decide_kernel, dim3(a.rs.B), dim3(256), 0, s, a);
    if (a.do_prune) hipLaunchKernelGGL(adapt_compact_kernel, dim3(2 * a.rs.B, 11), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace lg
