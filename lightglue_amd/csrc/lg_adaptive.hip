// lightglue_amd — adaptive depth (early stop, ref lightglue.py:547-550 + :645-656) and adaptive
// width (point pruning, ref :551-566 + :636-643) without leaving the device.
//
// adapt_decide  one workgroup per pair: counts low-confidence tokens, takes the stop decision,
//               and — if the pair continues — turns the keep mask of each image into destination
//               indices with a wavefront ballot + popcount prefix (stable: kept points stay in
//               ascending index order, which is what `torch.where` gives the reference).
// adapt_compact rewrites, IN PLACE, the descriptor rows, the rotary tables and the index set of
//               every pruned segment so that later layers see contiguous rows [0, len).
//               Parallelism = segments x 128-row chunks, every chunk in registers at once (see the kernel).
// Per-pair state (len, active, final_layer) lives in device memory; every later kernel reads it,
// so the whole adaptive forward is one stream of launches with no host synchronisation.
#include "lg_kernels.h"

namespace lg {

constexpr int DNT = 1024, DNW = DNT / 64;   // one workgroup of 16 waves per pair: a 2048-point image is two ballot steps (round 3: 256 threads, eight)
__global__ __launch_bounds__(DNT) void adapt_decide_kernel(AdaptArgs a) {
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int sh_cnt[DNW];
    __shared__ int sh_stop;
    __shared__ int sh_newlen[2];
    if (pair == 0 && tid == 0 && a.compact_ticket) *a.compact_ticket = 0;   // the compaction launched behind this kernel deals its work items from 0
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
            for (int r = tid; r < L; r += DNT) cnt += (c[r] < a.conf_thr) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) sh_cnt[wave] = cnt;
        __syncthreads();
        if (tid == 0) {
            int total = 0;
            for (int w = 0; w < DNW; ++w) total += sh_cnt[w];
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
    if (a.gather && tid == 0) a.xsel[pair] = a.xnext;   // the pair continues: proj_gather_kernel moves its rows to the other buffer set
    for (int image = 0; image < 2; ++image) {
        const int seg = 2 * pair + image;
        const int L = image ? len1 : len0;
        if (!a.do_prune || L <= a.pruning_min_kpts) {   // ref :551 / :559
            if (tid == 0) { a.len_old[seg] = -1; sh_newlen[image] = L; }   // "pruning not applied at this layer"
            continue;
        }
        const int base = seg_row_base(a.rs, seg);
        int* prune = image ? a.prune1 + (long long)pair * a.n1 : a.prune0 + (long long)pair * a.n0;
        int running = 0;                                 // uniform across the block
        for (int r0 = 0; r0 < L; r0 += DNT) {
            const int r = r0 + tid;
            bool keep = false;
            int v = 0;
            if (r < L) {
                keep = a.mscore[base + r] > a.width_conf;               // ref :640 (width_conf = 1 - width_confidence)
                if (a.do_stop) keep = keep || (a.conf[base + r] <= a.conf_thr);  // ref :641-642
                if (a.gather) v = a.ind[base + r];                      // read (and waited for, below) BEFORE the barriers: the in-place index compaction writes rows <= r
            }
            const unsigned long long bal = __ballot(keep);
            const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
            // the in-place index compaction: every read of this chunk's index-set entries must have RETURNED before any wave stores below.  hipcc's __syncthreads
            // does not wait for loads in flight (workgroup scope, one CU, one L1: its memory model needs no vmcnt there — seen in the ISA), so wait explicitly
            if (a.gather) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                             // sh_cnt reuse
            if (lane == 0) sh_cnt[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0, all = 0;
#pragma unroll
            for (int w = 0; w < DNW; ++w) { const int c = sh_cnt[w]; woff += w < wave ? c : 0; all += c; }
            if (r < L) {
                const int d = keep ? (running + woff + prefix) : -1;
                if (!a.gather) a.dst[base + r] = d;
                else if (keep) {   // d <= r and rows < r0 were read by earlier iterations: in place is safe (ref :554-558: ind = ind[keep]; prune[:, ind] += 1)
                    a.src[base + d] = r; a.ind[base + d] = v; prune[v] += 1;
                }
            }
            running += all;
        }
        __syncthreads();
        if (tid == 0) { a.len_old[seg] = L; a.len[seg] = running; sh_newlen[image] = running; }
    }
    // every point of one image pruned: the reference leaves its layer loop at the next iteration's
    // empty guard (ref :539-540) and returns the empty result with stop = layer + 2 (ref :568-588)
    if (tid == 0 && (sh_newlen[0] == 0 || sh_newlen[1] == 0)) { a.active[pair] = 0; a.final_layer[pair] = a.layer + 1; }
}

// Compaction, round 4: parallel over ROW CHUNKS as well (round 3: one workgroup per (segment, 128-byte column slice) walking the rows in dependent
// read | barrier | write steps — latency-bound by construction, 1.3 TB/s).  Work item = (segment, chunk of CROWS rows); a workgroup holds its whole
// chunk — the 1 KB descriptor row, the cos and sin rows (1280 B per keypoint, 80 pieces of 16 bytes, dealt to the 256 threads in row-major order:
// fully coalesced) and the index-set entries — in registers, so every chunk of every segment is read at the same time.
// In place is still safe because dst <= src: a chunk only writes rows of ITSELF or of LOWER chunks of its segment, and it does so after those
// chunks have published "all my rows are in registers" (one relaxed agent-scope flag per chunk, value = this launch's epoch; no payload travels
// through memory between workgroups, so no release / acquire of data is involved — the flag only orders the consumer's stores behind the producer's
// completed loads).  Deadlock freedom (round 5): work items are dealt by an atomic TICKET, not by workgroup id.  Item id = chunk * segments +
// seg, an item waits only for items with LOWER ids, and a lower ticket is always held by a workgroup that is already running (it drew the ticket) —
// so the lowest unfinished item never waits and every wait ends, whatever the grid size, the number of co-resident workgroups (CU masks, partitioned
// modes, a second process on the GPU) or the dispatch order.  The spin is bounded anyway; a wait that expires SKIPS the chunk's stores (rows stay where
// they were instead of overwriting unread ones) and raises a.compact_err, which the forward reports as LG_ERR_DEVICE in io->status.
// Memory ops are raw BUFFER loads / stores: a lane that has nothing to move gets an offset past the end of the buffer, for which the hardware
// returns zeros / drops the store WITHOUT touching memory — 40 independent 16-byte loads per thread with no branch between them (a branch
// around a load makes hipcc wait for it at the join, i.e. one exposed round trip per piece).
#ifndef LG_COMPACT_ROWS
#define LG_COMPACT_ROWS 128
#endif
constexpr int CROWS = LG_COMPACT_ROWS;         // rows per chunk.  64 was measured in round 5 (twice the work items, so that cfg #3' — 128 chunks of 128 rows on 256 CUs —
                                               // fills the chip): SLOWER, 23.8 vs 17.9 us per launch, cfg #3' -1 % (profiles/r05j_*): the launch is a chain of per-item
                                               // latencies (ticket, index loads, row loads, flag wait, stores), not a per-CU bandwidth limit
constexpr int CXU = CROWS / 4, CTU = CROWS / 16;   // descriptor-row steps per wave, rotary-row steps per thread
constexpr unsigned CSKIP = 0xFFFFFFF0u;        // offset no buffer reaches
__device__ __forceinline__ __amdgpu_buffer_rsrc_t compact_rsrc(float* p, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(bytes > 0xFFFFFF00LL ? 0xFFFFFF00LL : bytes), 0x00020000);
}

__global__ __launch_bounds__(256) void adapt_compact_kernel(AdaptArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), nseg = 2 * a.rs.B, nchunk = a.compact_chunks;
    __shared__ int sh_d[CROWS];                // destination row of each row of the chunk, -1 = nothing to move (dropped, or already in place)
    const long long Rrows = (long long)a.rs.B * (a.rs.cap0 + a.rs.cap1);
    const __amdgpu_buffer_rsrc_t rx = compact_rsrc(a.X, Rrows * 1024), rt = compact_rsrc(w < 2 ? a.cosb : a.sinb, Rrows * 128);   // waves 0, 1 move cos rows, waves 2, 3 sin rows
    __shared__ int sh_item;
    int ticket = tid == 0 ? atomicAdd(a.compact_ticket, 1) : 0;
    for (;;) {
        __syncthreads();                         // everybody has read the previous ticket (and the previous item's readers of sh_d are done)
        if (tid == 0) sh_item = ticket;
        __syncthreads();
        const int item = sh_item;
        if (item >= nseg * nchunk) break;
        if (tid == 0) ticket = atomicAdd(a.compact_ticket, 1);   // the next ticket travels while this item is processed (a workgroup's tickets ascend, so the
                                                                 // lowest unfinished item is still always some running workgroup's current or next one)
        const int chunk = item / nseg, seg = item - chunk * nseg;
        const int Lold = a.len_old[seg];
        if (Lold < 0) continue;                  // pruning not applied to this segment at this layer
        const int r0 = chunk * CROWS;
        if (r0 >= Lold) continue;                // no rows: nobody waits for this chunk either (a waiting chunk has rows, so every lower one does)
        const int Lnew = a.len[seg];
        const int base = seg_row_base(a.rs, seg);
        // index set + prune counters (ref :555, :558) — one row per thread of the first half of the workgroup
        int myd = -1, myv = 0;
        if (tid < CROWS && r0 + tid < Lold) { myd = a.dst[base + r0 + tid]; myv = a.ind[base + r0 + tid]; }
        int* prune = (seg & 1) ? a.prune1 + (long long)(seg >> 1) * a.n1 : a.prune0 + (long long)(seg >> 1) * a.n0;
        if (Lnew == Lold) {                      // nothing dropped: rows and index set already in place, only the counters move
            if (myd >= 0) prune[myv] += 1;
            continue;
        }
        if (tid < CROWS) sh_d[tid] = (myd >= 0 && myd != r0 + tid) ? myd : -1;   // (the loop head's barrier: the previous item's readers are done)
        __syncthreads();
        // descriptor rows: wave w, step u -> row 4u + w of the chunk, one 16-byte piece per lane (a whole 1 KB row per wave instruction)
        u32x4 vx[CXU], vt[CTU]; int dx[CXU], dt[CTU];
        const unsigned xrow0 = (unsigned)(base + r0) * 1024u + (unsigned)lane * 16u;
#pragma unroll
        for (int u = 0; u < CXU; ++u) {
            const int row = 4 * u + w;
            dx[u] = sh_d[row];
            vx[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, dx[u] >= 0 ? xrow0 + (unsigned)row * 1024u : CSKIP, 0, 0);
        }
        // rotary tables: a wave pair covers 128 rows x 8 pieces of one table: thread t of the pair, step u -> row 16u + (t >> 3), piece t & 7
        const int tt = tid & 127;
        const unsigned trow0 = (unsigned)(base + r0) * 128u + (unsigned)(tt & 7) * 16u;
#pragma unroll
        for (int u = 0; u < CTU; ++u) {
            const int row = 16 * u + (tt >> 3);
            dt[u] = sh_d[row];
            vt[u] = __builtin_amdgcn_raw_buffer_load_b128(rt, dt[u] >= 0 ? trow0 + (unsigned)row * 128u : CSKIP, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every row of the chunk has ARRIVED in registers (not merely been requested)
        __syncthreads();
        int* flags = a.compact_flags + (long long)seg * nchunk;
        if (tid == 0) __hip_atomic_store(flags + chunk, a.compact_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __shared__ int sh_expired;
        if (tid == 0) sh_expired = 0;
        __syncthreads();
        if (tid < chunk) {                       // lower chunks of this segment: their rows are the only foreign ones this chunk overwrites
            int spins = 0;
            while (__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.compact_epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 20)) { *a.compact_err = 1; sh_expired = 1; break; }
            }
        }
        __syncthreads();
        if (sh_expired) continue;                // never expected (see above): leave this chunk's rows where they are rather than overwrite unread ones
        const unsigned xdst0 = (unsigned)base * 1024u + (unsigned)lane * 16u, tdst0 = (unsigned)base * 128u + (unsigned)(tt & 7) * 16u;
#pragma unroll
        for (int u = 0; u < CXU; ++u) __builtin_amdgcn_raw_buffer_store_b128(vx[u], rx, dx[u] >= 0 ? xdst0 + (unsigned)dx[u] * 1024u : CSKIP, 0, 0);
#pragma unroll
        for (int u = 0; u < CTU; ++u) __builtin_amdgcn_raw_buffer_store_b128(vt[u], rt, dt[u] >= 0 ? tdst0 + (unsigned)dt[u] * 128u : CSKIP, 0, 0);
        if (myd >= 0) { a.ind[base + myd] = myv; prune[myv] += 1; }
    }
}

int compact_chunk_rows() { return CROWS; }

hipError_t launch_adapt(const AdaptArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(adapt_decide_kernel, dim3(a.rs.B), dim3(DNT), 0, s, a);
    if (a.do_prune && !a.gather) {   // gather mode: the next SelfBlock projection moves the rows (proj_gather_kernel)
        const int items = 2 * a.rs.B * a.compact_chunks;   // dealt by ticket: the grid size is a throughput choice only (one workgroup per CU of an MI355X)
        hipLaunchKernelGGL(adapt_compact_kernel, dim3(items < 256 ? items : 256), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace lg
