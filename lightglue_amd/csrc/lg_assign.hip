// lightglue_amd — log-assignment and match filtering (ref lightglue.py:265-277
// sigmoid_log_double_softmax, :302-318 filter_matches, :593-614 output assembly).
// The similarity matrix sim[pair][a][b] (fp32) is produced by launch_sim; everything here is
// HBM-bound streaming over it:
//   sweep 1 row log-sum-exp and per-row-tile column (max, sum-exp) partials in ONE pass over sim
//   sweep 2 score(a,b) = ((sim - lse_r[a]) + (sim - lse_c[b])) + (ls0[a] + ls1[b])   [ref :270-274] on the fly;
//           row max/argmax and per-row-tile column max/argmax partials, first-index tie-break (torch.max)
//           => 8*N*M algorithmic bytes per pair (two fp32 read sweeps) + 1/8 of that in partials
//   final   mutual check, exp, threshold, scatter through the index sets into ORIGINAL index
//           space (un-pruning, ref :605-614) and the compact sorted match list (ref :593-602).
// The dustbin row/column (ref :275-276) never influence any output of forward and
// are not materialised.
#include "lg_kernels.h"

namespace lg {

// exp via the bare v_exp_f32 (arguments are <= 0 here; a flushed tail term is below fp32 resolution of the sum)
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; s = 0.f; return; }
    s = s * fexp(m - mn) + s2 * fexp(m2 - mn);
    m = mn;
}

// Both sweeps use the same decomposition: workgroup = ART rows x all columns; thread = 4 consecutive columns (one float4
// per row, a wave reads 1 KB contiguous), looping over the tile's rows and, for n > 1024, over column passes.
// The row loop is a REAL loop with a ~200-instruction body: each row's statistic is reduced across the wave right away
// (DPP + permlane, no LDS) and lane 0 folds it into an LDS slot.  The first version kept 32 per-row partials in
// registers and fully unrolled both the row loop and the 32 x 6-step reduction: 58 KB of straight-line code executed
// exactly once per wave, i.e. instruction-fetch-bound (the reduction alone was 54 of the kernel's 114 us).
// Column statistics are per-thread registers across the rows, written as per-row-tile partials [tile][column] and
// merged by the next kernel.
constexpr int ART = 32;

// (m, s) <- log-sum-exp merge with (m2, s2), ONE exponential: e = exp(-|m - m2|) scales the smaller side
__device__ __forceinline__ void lse_merge1(float& m, float& s, float m2, float s2) {
    const float e = (m == m2) ? 1.f : fexp(-fabsf(m - m2));   // m == m2 also covers (-inf, -inf), whose difference is NaN
    const bool keep = m >= m2;
    s = keep ? __builtin_fmaf(s2, e, s) : __builtin_fmaf(s, e, s2);
    m = keep ? m : m2;
}

// ---- sweep 1: row log-sum-exp + column (max, sum-exp) partials.  grid (cap0/ART, B)
__global__ __launch_bounds__(256) void lse_sweep_kernel(AssignArgs a) {
    const int pair = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    const int r0 = tile * ART;
    if (r0 >= len0) return;
    const int rows = min(ART, len0 - r0);
    const int ntiles = a.rs.cap0 / ART;
    const float* simp = a.sim + ((long long)pair * a.rs.cap0 + r0) * a.rs.cap1;
    float* pm = a.cpm + ((long long)pair * ntiles + tile) * a.rs.cap1;
    float* ps = a.cps + ((long long)pair * ntiles + tile) * a.rs.cap1;
    __shared__ float shm[4][ART], shs[4][ART];
    for (int c0 = 0; c0 < len1; c0 += 1024) {             // workgroup-uniform trip count
        const int c = c0 + tid * 4;
        const bool act = c < len1;                         // this thread's 4 columns hold at least one live column
        const float* col = simp + (act ? c : 0);
        float cm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, cs[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 nxt = *reinterpret_cast<const f32x4*>(col);
#pragma unroll 1
        for (int r = 0; r < rows; ++r) {
            f32x4 v = nxt;
            nxt = *reinterpret_cast<const f32x4*>(col + (long long)min(r + 1, rows - 1) * a.rs.cap1);   // next row in flight
#pragma unroll
            for (int i = 0; i < 4; ++i) if (!act || c + i >= len1) v[i] = -INFINITY;
            // row statistic: the WAVE's maximum first (12 DPP / permlane ops), then every lane exponentiates against it and the
            // sums are added across the wave — 4 exponentials per lane and row (the pairwise log-sum-exp merge tree cost 10)
            const float m = wave_allmax(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            float sum = 0.f;
            if (m != -INFINITY) sum = (fexp(v[0] - m) + fexp(v[1] - m)) + (fexp(v[2] - m) + fexp(v[3] - m));   // wave-uniform branch; dead lanes add exp(-inf) = 0
            if (act) {
#pragma unroll
                for (int i = 0; i < 4; ++i) lse_merge1(cm[i], cs[i], v[i], v[i] == -INFINITY ? 0.f : 1.f);
            }
            sum = wave_allsum(sum);
            if (lane == 0) {
                if (c0 == 0) { shm[wave][r] = m; shs[wave][r] = sum; }
                else { float mo = shm[wave][r], so = shs[wave][r]; lse_merge1(mo, so, m, sum); shm[wave][r] = mo; shs[wave][r] = so; }
            }
        }
        if (act) {
            *reinterpret_cast<f32x4*>(pm + c) = f32x4{cm[0], cm[1], cm[2], cm[3]};
            *reinterpret_cast<f32x4*>(ps + c) = f32x4{cs[0], cs[1], cs[2], cs[3]};
        }
    }
    __syncthreads();
    if (tid < rows) {
        float m = shm[0][tid], s = shs[0][tid];
        for (int w = 1; w < 4; ++w) lse_merge1(m, s, shm[w][tid], shs[w][tid]);
        a.lse_r[(long long)pair * a.rs.cap0 + r0 + tid] = m + logf(s);
    }
}

// ---- merge the column partials: lse_c[b] over the live row tiles.  grid (cap1/64, B): a wave = 64 consecutive columns of
// every 4th row tile, so that a thread has <= 8 independent (max, sum) loads in flight per round and the 4 partial results
// meet in LDS.  (One thread per column walking all tiles was latency-bound: 19 us for 8 MB at cfg #2.)
__global__ __launch_bounds__(256) void col_lse_merge_kernel(AssignArgs a) {
    const int pair = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    if (blockIdx.x * 64 >= len1) return;                  // workgroup-uniform
    const int ntiles = a.rs.cap0 / ART, live = (len0 + ART - 1) / ART;
    const int cc = min(c, a.rs.cap1 - 1);                 // (cap1 is a multiple of 128: always in range; keeps the loads unconditional)
    const float* pm = a.cpm + (long long)pair * ntiles * a.rs.cap1 + cc;
    const float* ps = a.cps + (long long)pair * ntiles * a.rs.cap1 + cc;
    float m = -INFINITY, s = 0.f;
#pragma unroll 8
    for (int t = wave; t < live; t += 4) lse_merge1(m, s, pm[(long long)t * a.rs.cap1], ps[(long long)t * a.rs.cap1]);
    __shared__ float shm[4][64], shs[4][64];
    shm[wave][lane] = m; shs[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < len1) {
#pragma unroll
        for (int w = 1; w < 4; ++w) lse_merge1(m, s, shm[w][lane], shs[w][lane]);
        a.lse_c[(long long)pair * a.rs.cap1 + c] = m + logf(s);
    }
}

__device__ __forceinline__ float score_of(float sim, float lr, float lc, float cert) {
    return ((sim - lr) + (sim - lc)) + cert;   // ref :271-274 evaluation order
}

// ---- sweep 2: score matrix on the fly; row max/argmax (final) + column max/argmax partials per row tile
__global__ __launch_bounds__(256) void argmax_sweep_kernel(AssignArgs a) {
    const int pair = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    const int r0 = tile * ART;
    if (r0 >= len0) return;
    const int rows = min(ART, len0 - r0);
    const int ntiles = a.rs.cap0 / ART;
    const float* simp = a.sim + ((long long)pair * a.rs.cap0 + r0) * a.rs.cap1;
    const float* lsec = a.lse_c + (long long)pair * a.rs.cap1;
    const float* ls1 = a.ls + seg_row_base(a.rs, 2 * pair + 1);
    const float* lser = a.lse_r + (long long)pair * a.rs.cap0 + r0;
    const float* ls0 = a.ls + seg_row_base(a.rs, 2 * pair) + r0;
    float* pv = a.cbv + ((long long)pair * ntiles + tile) * a.rs.cap1;
    int* pi = a.cbi + ((long long)pair * ntiles + tile) * a.rs.cap1;
    __shared__ float shb[4][ART]; __shared__ int shi[4][ART];
    for (int c0 = 0; c0 < len1; c0 += 1024) {             // workgroup-uniform trip count
        const int c = c0 + tid * 4;
        const bool act = c < len1;
        const int cc = act ? c : 0;
        const float* col = simp + cc;
        const f32x4 lc = *reinterpret_cast<const f32x4*>(lsec + cc);
        const f32x4 l1 = *reinterpret_cast<const f32x4*>(ls1 + cc);
        float cb[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}; int ci[4] = {0, 0, 0, 0};
        f32x4 nxt = *reinterpret_cast<const f32x4*>(col);
#pragma unroll 1
        for (int r = 0; r < rows; ++r) {
            const f32x4 v = nxt;
            nxt = *reinterpret_cast<const f32x4*>(col + (long long)min(r + 1, rows - 1) * a.rs.cap1);
            const float lr = lser[r], l0 = ls0[r];
            float best = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (act && c + i < len1) {
                    const float sc = score_of(v[i], lr, lc[i], l0 + l1[i]);
                    if (sc > best) { best = sc; bi = c + i; }                 // ascending columns: first index wins
                    if (sc > cb[i]) { cb[i] = sc; ci[i] = r0 + r; }           // ascending rows: first index wins
                }
            }
            // wave maximum by value only (12 ops); its first index is the index held by the LOWEST lane that has it (lanes hold
            // ascending columns): ballot + s_ff1 + v_readlane instead of carrying the index through every reduction step
            const float wbest = wave_allmax(best);
            const unsigned long long owners = __ballot(best == wbest);    // never empty: all lanes hold -inf when the row has no finite score
            const int wbi = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(owners));
            if (lane == 0) {
                if (c0 == 0 || wbest > shb[wave][r]) { shb[wave][r] = wbest; shi[wave][r] = wbi; }   // later passes = higher columns: strict >
            }
        }
        if (act) {
            *reinterpret_cast<f32x4*>(pv + c) = f32x4{cb[0], cb[1], cb[2], cb[3]};
            *reinterpret_cast<int4*>(pi + c) = int4{ci[0], ci[1], ci[2], ci[3]};
        }
    }
    __syncthreads();
    if (tid < rows) {
        float best = shb[0][tid]; int bi = shi[0][tid];
        for (int w = 1; w < 4; ++w) {
            const float ob = shb[w][tid]; const int oi = shi[w][tid];
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        a.max0[(long long)pair * a.rs.cap0 + r0 + tid] = best; a.arg0[(long long)pair * a.rs.cap0 + r0 + tid] = bi;
    }
}

// ---- merge the column argmax partials (first index = lowest row wins).  Same decomposition as col_lse_merge_kernel.
__global__ __launch_bounds__(256) void col_argmax_merge_kernel(AssignArgs a) {
    const int pair = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = blockIdx.x * 64 + lane;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    if (blockIdx.x * 64 >= len1) return;
    const int ntiles = a.rs.cap0 / ART, live = (len0 + ART - 1) / ART;
    const int cc = min(c, a.rs.cap1 - 1);
    const float* pv = a.cbv + (long long)pair * ntiles * a.rs.cap1 + cc;
    const int* pi = a.cbi + (long long)pair * ntiles * a.rs.cap1 + cc;
    float best = -INFINITY; int bi = 0;
#pragma unroll 8
    for (int t = wave; t < live; t += 4) {   // ascending rows within a wave: strict '>'
        const float v = pv[(long long)t * a.rs.cap1];
        const int vi = pi[(long long)t * a.rs.cap1];
        if (v > best) { best = v; bi = vi; }
    }
    __shared__ float shb[4][64]; __shared__ int shi[4][64];
    shb[wave][lane] = best; shi[wave][lane] = bi;
    __syncthreads();
    if (wave == 0 && c < len1) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {        // the waves interleave row tiles: ties go to the lower row
            const float v = shb[w][lane]; const int vi = shi[w][lane];
            if (v > best || (v == best && v != -INFINITY && vi < bi)) { best = v; bi = vi; }
        }
        a.max1[(long long)pair * a.rs.cap1 + c] = best; a.arg1[(long long)pair * a.rs.cap1 + c] = bi;
    }
}

// ---- final: grid (ceil(max(cap0, cap1) / 256), B); workgroup j owns rows [256 j, 256 j + 256) of both images.
// The compact match list is sorted by the image-0 row, so a workgroup needs the number of valid rows below its own: it
// recomputes their validity (3 dependent 4-byte loads per row; at most cap0 / 256 - 1 extra rounds) instead of waiting for
// its predecessors — no inter-workgroup dependency, and one workgroup per pair was 10 us on a 256-CU chip.
struct Row0 { bool valid; int j; float e; };
__device__ __forceinline__ Row0 match_of_row0(const AssignArgs& a, const float* max0, const int* arg0, const int* arg1, int r, int len1) {
    // a row whose scores are all NaN (NaN / Inf in the inputs) keeps the argmax sentinel: treat it as unmatched instead of
    // indexing with it (the reference returns garbage there, but does not fault)
    const int jraw = arg0[r];
    const bool jok = (unsigned)jraw < (unsigned)len1;
    const int j = jok ? jraw : 0;
    const bool mutual0 = jok && arg1[j] == r;
    const float e = mutual0 ? expf(max0[r]) : 0.f;
    return {mutual0 && (e > a.filter_threshold), j, e};
}
__global__ __launch_bounds__(256) void finalize_kernel(AssignArgs a) {
    const int pair = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    const int base0 = seg_row_base(a.rs, 2 * pair), base1 = seg_row_base(a.rs, 2 * pair + 1);
    const float* max0 = a.max0 + (long long)pair * a.rs.cap0;
    const int* arg0 = a.arg0 + (long long)pair * a.rs.cap0;
    const int* arg1 = a.arg1 + (long long)pair * a.rs.cap1;
    int* m0 = a.m0 + (long long)pair * a.n0; float* s0 = a.s0 + (long long)pair * a.n0;
    int* m1 = a.m1 + (long long)pair * a.n1; float* s1 = a.s1 + (long long)pair * a.n1;
    __shared__ int sh_cnt[4], sh_below[4];
    if (len0 == 0 || len1 == 0) { if (blk == 0 && tid == 0) a.n_matches[pair] = 0; return; }
    // image 1 side (ref :309, :313, :315, :317)
    {
        const int b = blk * 256 + tid;
        if (b < len1) {
            const int i = arg1[b];
            const bool mutual1 = arg0[i] == b;
            const float e = expf(max0[i]);            // mutual1 implies mutual0(i), so mscores0[i] = exp(max0[i])
            const bool valid1 = mutual1 && (e > a.filter_threshold);
            const int ob = a.ind[base1 + b];
            m1[ob] = valid1 ? a.ind[base0 + i] : -1;
            s1[ob] = mutual1 ? e : 0.f;
        }
    }
    if (blk * 256 >= len0 && blk != (int)gridDim.x - 1) return;   // (the last workgroup reports the count)
    // image 0 side + compact list, ascending a (ref :308, :312, :314, :316, :595-602)
    int below = 0;                                                // valid rows of the workgroups before this one (wave-uniform partial)
    for (int q = 0; q < blk; ++q) {
        const int r = q * 256 + tid;
        below += __popcll(__ballot(r < len0 && match_of_row0(a, max0, arg0, arg1, min(r, len0 - 1), len1).valid));
    }
    const int r = blk * 256 + tid;
    bool valid0 = false; int oa = 0, ob = -1; float e = 0.f;
    if (r < len0) {
        const Row0 m = match_of_row0(a, max0, arg0, arg1, r, len1);
        valid0 = m.valid; e = m.e;
        oa = a.ind[base0 + r];
        ob = valid0 ? a.ind[base1 + m.j] : -1;
        m0[oa] = ob;
        s0[oa] = e;
    }
    const unsigned long long bal = __ballot(valid0);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) { sh_cnt[wave] = __popcll(bal); sh_below[wave] = below; }
    __syncthreads();
    int woff = sh_below[0] + sh_below[1] + sh_below[2] + sh_below[3];
    for (int w = 0; w < wave; ++w) woff += sh_cnt[w];
    if (valid0) {
        int* ml = a.matches + (long long)pair * a.max_matches * 2;
        float* msl = a.mscores + (long long)pair * a.max_matches;
        const int k = woff + prefix;
        ml[2 * k] = oa; ml[2 * k + 1] = ob; msl[k] = e;
    }
    if (blk == (int)gridDim.x - 1 && tid == 0)
        a.n_matches[pair] = sh_below[0] + sh_below[1] + sh_below[2] + sh_below[3] + sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
}

// ---- optional: materialise the full log-assignment (ref :265-277) in original index space.
// grid (cap0 + 1, B): block r < cap0 writes score row r (+ its dustbin column entry), block cap0 the dustbin row.
__global__ __launch_bounds__(256) void fill_neg_inf_kernel(float* p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = -INFINITY;
}
__global__ __launch_bounds__(256) void log_assignment_kernel(AssignArgs a) {
    const int pair = blockIdx.y, r = blockIdx.x, tid = threadIdx.x;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    const int base0 = seg_row_base(a.rs, 2 * pair), base1 = seg_row_base(a.rs, 2 * pair + 1);
    float* out = a.log_assignment + (long long)pair * (a.n0 + 1) * (a.n1 + 1);
    if (len0 == 0 || len1 == 0) {   // a pair that never ran a layer, or lost every point of an image (ref :539-540: the reference returns before it builds any
        // score): its matchability terms were never written — leave the -inf fill, define the corner
        if (r == a.rs.cap0 && tid == 0) out[(long long)a.n0 * (a.n1 + 1) + a.n1] = 0.f;
        return;
    }
    if (r == a.rs.cap0) {   // dustbin row: logsigmoid(-z1) (ref :276), corner 0 (ref :269 zero-initialised)
        for (int c = tid; c < len1; c += 256) out[(long long)a.n0 * (a.n1 + 1) + a.ind[base1 + c]] = a.lsneg[base1 + c];
        if (tid == 0) out[(long long)a.n0 * (a.n1 + 1) + a.n1] = 0.f;
        return;
    }
    if (r >= len0) return;
    float* orow = out + (long long)a.ind[base0 + r] * (a.n1 + 1);
    const float* simr = a.sim + ((long long)pair * a.rs.cap0 + r) * a.rs.cap1;
    const float lr = a.lse_r[(long long)pair * a.rs.cap0 + r], l0 = a.ls[base0 + r];
    const float* lsec = a.lse_c + (long long)pair * a.rs.cap1;
    for (int c = tid; c < len1; c += 256) orow[a.ind[base1 + c]] = score_of(simr[c], lr, lsec[c], l0 + a.ls[base1 + c]);
    if (tid == 0) orow[a.n1] = a.lsneg[base0 + r];   // dustbin column: logsigmoid(-z0) (ref :275)
}

hipError_t launch_assign(const AssignArgs& a, hipStream_t s) {
    const int B = a.rs.B;
    hipError_t e;
    // outputs are in ORIGINAL index space: rows that are not live any more (pruned, or beyond a ragged count) keep the
    // -1 / 0 fill.  When every row is live (no pruning, no ragged counts) finalize writes all of them and the fill is skipped.
    if (!a.all_rows_live) {
    if (a.n0 > 0) {
        if ((e = hipMemsetAsync(a.m0, 0xFF, sizeof(int) * (size_t)B * a.n0, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(a.s0, 0, sizeof(float) * (size_t)B * a.n0, s)) != hipSuccess) return e;
    }
    if (a.n1 > 0) {
        if ((e = hipMemsetAsync(a.m1, 0xFF, sizeof(int) * (size_t)B * a.n1, s)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(a.s1, 0, sizeof(float) * (size_t)B * a.n1, s)) != hipSuccess) return e;
    }
    }
    hipLaunchKernelGGL(lse_sweep_kernel, dim3(a.rs.cap0 / ART, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(col_lse_merge_kernel, dim3(a.rs.cap1 / 64, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(argmax_sweep_kernel, dim3(a.rs.cap0 / ART, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(col_argmax_merge_kernel, dim3(a.rs.cap1 / 64, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(finalize_kernel, dim3((max(a.rs.cap0, a.rs.cap1) + 255) / 256, B), dim3(256), 0, s, a);
    if (a.log_assignment) {
        const long long total = (long long)B * (a.n0 + 1) * (a.n1 + 1);
        hipLaunchKernelGGL(fill_neg_inf_kernel, dim3(2048), dim3(256), 0, s, a.log_assignment, total);
        hipLaunchKernelGGL(log_assignment_kernel, dim3(a.rs.cap0 + 1, B), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace lg
