// lightglue_amd — descriptor similarity matrix of the log assignment (ref lightglue.py:292: sim = einsum("bmd,bnd->bmn", mdesc0, mdesc1)) for the
// split-f16 precision, on operands that ARRIVE as f16 planes: the final projection (final_compute, lg_proj_body.h) stores its result as hi / lo planes
// MD[2][R][256] — the same bytes as the fp32 rows it used to store — so that nothing is converted here.  (The generic sim_kernel of lg_gemm.hip reads fp32
// rows and splits both operand tiles per K stage in every workgroup, 8 x redundantly at N = 1024: 0.086 of the MFMA peak; it stays for the other precisions.)
//
// Shape = the split attention's S phase with d = 256: a workgroup is 8 waves x 16 image-0 rows (their 8 k-chunks x 2 planes live in 64 VGPRs as the MFMA's A
// operand for the whole kernel) and walks up to 512 image-1 rows in 32-row tiles that a wave-cooperative LDS-DMA (lds_dma16, no registers, no conversion) drops
// into two 32 KB half buffers (64 KB: two workgroups per CU), ONE barrier per tile.  Per tile and wave: 48 MFMAs on 32 KB of fragment reads.
// Row dealing: the 32 rows of tile h (0 / 1) of a 64-row super tile T are the rows 64 T + 4 j + 2 h + nb (n-block nb, MFMA column slot j), so that after both
// tiles lane (lr, g) holds FOUR CONSECUTIVE image-1 columns 64 T + 4 lr .. + 3 of its rows 4 g + r: one 16-byte store per row, 256 contiguous bytes per row and
// instruction.  (A free choice: which global row a DMA lane fetches.)  LDS rows are 512 B; the 16-byte slot index is XOR-ed with the row's low 4 bits on the
// DMA's SOURCE side, which makes every ds_read_b128 lane group of the fragment reads conflict-free (rows lr, slot 4 c + g).
// Arithmetic per output element = sim_kernel's: k-chunks ascending, per chunk lo0 x hi1, hi0 x lo1, hi0 x hi1 into one fp32 accumulator with image 0 as the A
// operand — bit-identical results (tests/test_gpu_round6.py).
// Measured (profiles/r06y_sim_planes.log, r06y2_sim_chunk.log; same box): launch time against sim_kernel 74 -> 60 us at cfg #2, 140 -> 104 us at cfg #3', 1.215 -> 0.776 ms
// at cfg #4 (0.142 of the MFMA peak, from 0.091).  512 image-1 rows per workgroup beat 256 (-6 %) and 128 (-20 %); holding a super tile's results in 16 more
// registers so that its stores leave behind the next tile's barrier (instead of right behind their MFMAs, in front of the next vmcnt(0)) measured +4 % at cfg #2: dropped.
#include "lg_kernels.h"

namespace lg {

constexpr int SBM = 128, STHREADS = 512, SHALF = 2 * 32 * 512;   // rows per strip, image-1 rows per workgroup, threads, bytes of one half buffer

__global__ __launch_bounds__(STHREADS, 4) void sim_planes_kernel(SimPlanesArgs a) {
    typedef TagF16 Tag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int SCHUNK = a.chunk;                                  // image-1 rows per workgroup: 512 when the grid fills the chip, less for small batches (launch_sim_planes)
    const int strips = a.rs.cap0 / SBM, chunks = (a.rs.cap1 + SCHUNK - 1) / SCHUNK, per_pair = strips * chunks;
    // XCD-aware order (xcd_remap hands every XCD a contiguous range of virtual ids): a pair's workgroups run back to back on ONE XCD, whose L2 then holds the
    // pair's two descriptor sets for all of them
    const int v = xcd_remap(blockIdx.x, gridDim.x);
    const int pair = v / per_pair, rem = v - pair * per_pair;
    const int a0 = (rem / chunks) * SBM, b0 = (rem % chunks) * SCHUNK;
    const int len0 = a.rs.len[2 * pair], len1 = a.rs.len[2 * pair + 1];
    if (a0 >= len0 || b0 >= len1) return;
    const int nst = (min(len1 - b0, SCHUNK) + 63) >> 6;        // live 64-row super tiles of this chunk (cap1 is a multiple of 128: always inside the segment)
    const int ntile = 2 * nst;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const long long PL = a.plane;
    const f16_t* A = a.md + (seg_row_base(a.rs, 2 * pair) + a0) * 256LL;
    const f16_t* Bm = a.md + (seg_row_base(a.rs, 2 * pair + 1) + b0) * 256LL;

    // DMA pieces: a half buffer is [plane][32 rows][512 B] = 32 pieces of 1 KB (two LDS rows each); wave w moves pieces 4 (w & 3) .. + 3 of plane w >> 2.
    // Lane: LDS row rho = 2 piece + (lane >> 5) = 16 nb + j, LDS slot s' = lane & 31 <- global slot s' ^ (rho & 15) of global row 4 j + nb (+ 64 T + 2 h)
    // Addresses = wave-uniform base (SGPR pair: plane + tile) + a 32-bit per-lane byte offset per piece: no 64-bit VALU arithmetic per request, 4 VGPRs instead of 8.
    uint32_t doff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rho = 2 * (4 * (wave & 3) + i) + (lane >> 5), sl = (lane & 31) ^ (rho & 15);
        doff[i] = (uint32_t)(((4 * (rho & 15) + (rho >> 4)) * 256 + sl * 8) * (int)sizeof(f16_t));
    }
    const f16_t* dbase = Bm + (wave >> 2) * PL;                  // wave-uniform
    auto dma_tile = [&](int tile) {                              // tile = 2 T + h -> half buffer h
        const f16_t* src = dbase + (64 * (tile >> 1) + 2 * (tile & 1)) * 256;
        char* dst = smem + (tile & 1) * SHALF + (wave >> 2) * (32 * 512) + (4 * (wave & 3)) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16_s(src, doff[i], dst + i * 1024);
    };
    dma_tile(0);

    // image-0 fragments (the MFMA's A operand): row a0 + 16 wave + lr, k = 32 c + 8 g .. + 7, both planes
    u32x4 ah[8], al[8];
    {
        const f16_t* src = A + (16 * wave + lr) * 256 + 8 * g;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ah[c] = *reinterpret_cast<const u32x4*>(src + 32 * c);
            al[c] = *reinterpret_cast<const u32x4*>(src + 32 * c + PL);
        }
    }
    // fragment read offsets inside a half buffer: row 16 nb + lr, slot (4 c + g) ^ lr = 4 (c ^ (lr >> 2)) + (g ^ (lr & 3)); c ^ x = (c & 4) | ((c & 3) ^ x)
    int fo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fo[i] = lr * 512 + (((i ^ (lr >> 2)) << 6) | ((g ^ (lr & 3)) << 4));

    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* out = a.sim + ((long long)pair * a.rs.cap0 + a0 + 16 * wave + 4 * g) * a.rs.cap1 + b0 + 4 * lr;
    auto store_tile = [&](int T) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(out + (long long)r * a.rs.cap1 + 64 * T) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    };
    auto compute_half = [&](int h) {                             // 32 image-1 rows: n-blocks nb = 0, 1 -> accumulators 2 h + nb
        const char* base = smem + h * SHALF;
        u32x4 bh[2][2], bl[2][2];                                // [register set][nb]: fragments read one k-chunk ahead of their MFMAs
        auto load_b = [&](int c, u32x4 (&hh)[2], u32x4 (&ll)[2]) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const char* src = base + fo[c & 3] + (c >> 2) * 256 + nb * (16 * 512);
                hh[nb] = *reinterpret_cast<const u32x4*>(src);
                ll[nb] = *reinterpret_cast<const u32x4*>(src + 32 * 512);
            }
        };
        load_b(0, bh[0], bl[0]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < 7) load_b(c + 1, bh[(c + 1) & 1], bl[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) mma_chunk<Tag>(acc[2 * h + nb], al[c], bh[c & 1][nb]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) mma_chunk<Tag>(acc[2 * h + nb], ah[c], bl[c & 1][nb]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) mma_chunk<Tag>(acc[2 * h + nb], ah[c], bh[c & 1][nb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int T = 0; T < nst; ++T) {
        // ---- tile 2 T (half 0).  The BUILTIN wait: hipcc's waitcnt pass must see it (lg_common.h, lds_dma16)
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): my pieces of this tile have landed (first time round: my A fragments too)
        asm volatile("" ::: "memory");
        __syncthreads();                                         // ... and everybody's; every wave is through with half 1
        dma_tile(2 * T + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute_half(0);
        // ---- tile 2 T + 1 (half 1)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        __syncthreads();
        dma_tile(T + 1 < nst ? 2 * T + 2 : 2 * T);               // never branched around; past the end: a harmless re-fetch of a live tile into the idle half
        __builtin_amdgcn_sched_barrier(0);
        compute_half(1);
        store_tile(T);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // no DMA may land in LDS that has been released
    (void)ntile;
}

hipError_t launch_sim_planes(const SimPlanesArgs& a0, hipStream_t s) {
    SimPlanesArgs a = a0;
    if (a.rs.cap0 % SBM || a.rs.cap1 % 64 || a.K != 256 || !a.md || a.plane <= 0) return hipErrorInvalidValue;
    // image-1 rows per workgroup: the longest walk (fewest operand reloads) that still gives the chip two workgroups per CU; single pairs and small batches take shorter
    // walks instead of leaving CUs idle (B = 1, N = 512: 4 workgroups of 512 rows would be 4 CUs busy).  The arithmetic per output element does not depend on it.
    const int strips = a.rs.cap0 / SBM;
    int chunk = 512;
    while (chunk > 64 && (long long)strips * ((a.rs.cap1 + chunk - 1) / chunk) * a.rs.B < 512) chunk >>= 1;
    if (a.chunk <= 0) a.chunk = chunk;
    if (a.chunk % 64) return hipErrorInvalidValue;
    const int chunks = (a.rs.cap1 + a.chunk - 1) / a.chunk;
    hipLaunchKernelGGL(sim_planes_kernel, dim3(strips * chunks * a.rs.B), dim3(STHREADS), 2 * SHALF, s, a);
    return hipGetLastError();
}

}  // namespace lg
