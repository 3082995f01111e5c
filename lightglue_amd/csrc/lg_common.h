// lightglue_amd — common device helpers for gfx950 (CDNA4, wave64).
//
// Vocabulary used by every kernel in this directory:
//   pair      one (image0, image1) matching problem; B pairs per forward.
//   segment   the keypoint set of one image of one pair; seg = 2*pair + image.
//   row       one keypoint.  All per-keypoint tensors are indexed by a GLOBAL row
//             grow = pair*(cap0+cap1) + image*cap0 + r   (r < len[seg] <= cap{image})
//             cap0/cap1 are multiples of 128 so that a 128-row tile never straddles segments.
//   len[seg]  current number of live keypoints of the segment (device int; shrinks when the
//             adaptive-width compaction prunes points).
//   chunk     16 bytes of one operand row along the contraction axis: 8 x bf16/f16 or 4 x f32.
//             One MFMA "step" (mma_chunk) consumes one chunk per lane of A and of B:
//             16-bit: one v_mfma_f32_16x16x32_{bf16,f16};  f32: four v_mfma_f32_16x16x4_f32.
//             Lane l supplies row/col (l & 15) and k-slot group g = l >> 4.  A and B use the SAME
//             (g, element) -> k mapping, so any k permutation applied to both is harmless.
//   C layout  acc[r] of lane l is C[row = 4*(l>>4) + r][col = l & 15]   (16x16 tile).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// This library is written for ONE target.  Hard-wired below and in the kernels: the gfx9 buffer-descriptor word 0x00020000 (weight_rsrc, compact_rsrc),
// the gfx9 s_waitcnt immediate layout (0x0F70 = vmcnt(0) only; lg_attention.hip, lg_tail.hip), v_mfma_f32_16x16x32_{f16,bf16}, v_permlane{16,32}_swap,
// global_load_lds_dwordx4, and dynamic LDS up to 132 KB per workgroup (160 KB per CU).  Any other --offload-arch would miscompile or fail at launch
// (ADVICE r04): refuse it here instead.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "lightglue_amd kernels are gfx950 (MI355X, CDNA4) only: build with --offload-arch=gfx950"
#endif

namespace lg {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Operand precision of a contraction (host enum mirrored in include/lightglue_amd.h)
// (3 was split-bf16, the default of rounds 1-2: same cost as PREC_F16X3, 3x its score error on the GPU — removed in round 3)
enum : int { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2, PREC_F16X3 = 4 };
// PREC_F16X3: x = hi + lo with both planes f16, three MFMAs per product (hi*lo + lo*hi + hi*hi), fp32 accumulate: 22 operand bits;
// the dropped lo*lo term and the lo rounding are 2^-24-class (product error 3.5e-9 of sum |x||w|).  Needs |x| < 65504 (the
// reference's own fp16 mode needs the same); lo planes of small values live in f16 subnormals, which the MFMA honours.
__host__ __device__ constexpr bool prec_is_split(int prec) { return prec == PREC_F16X3; }

// Element tags
struct TagF32 { typedef float elem; static constexpr int EPC = 4; };   // elements per 16-byte chunk
struct TagBF16 { typedef bf16_t elem; static constexpr int EPC = 8; };
struct TagF16 { typedef f16_t elem; static constexpr int EPC = 8; };

// ---- one 16-byte chunk of A and of B per lane -> accumulate into a 16x16 f32 tile
template <class Tag> __device__ __forceinline__ void mma_chunk(f32x4& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma_chunk<TagBF16>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<TagF16>(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma_chunk<TagF32>(f32x4& acc, const u32x4& a, const u32x4& b) {
    const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
}

// ---- fp32 -> operand conversions (round to nearest even, hardware v_cvt)
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pack2_f16(float a, float b) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    f16x2 v = {(f16_t)a, (f16_t)b};
    return __builtin_bit_cast(uint32_t, v);
}
template <class Tag> __device__ __forceinline__ uint32_t pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t pack2<TagBF16>(float a, float b) { return pack2_bf16(a, b); }
template <> __device__ __forceinline__ uint32_t pack2<TagF16>(float a, float b) { return pack2_f16(a, b); }

// 8 floats -> one chunk of a 16-bit operand
template <class Tag> __device__ __forceinline__ u32x4 pack8(const f32x4& lo, const f32x4& hi) {
    u32x4 r;
    r[0] = pack2<Tag>(lo[0], lo[1]); r[1] = pack2<Tag>(lo[2], lo[3]);
    r[2] = pack2<Tag>(hi[0], hi[1]); r[3] = pack2<Tag>(hi[2], hi[3]);
    return r;
}
// split-f16: x = hi + lo, both f16 (lo of small values lives in f16 subnormals, which the MFMA honours).
// Two values at a time in 4 instructions: v_cvt_pk_f16_f32 (hi pair), two v_fma_mix_f32 that read one half of the packed hi register
// as f16 and return p - hi in fp32 (exact: the difference is representable), v_cvt_pk_f16_f32 (lo pair).  hipcc's own lowering of
// the same expression is 8 instructions (v_cvt_f16_f32 x2, v_cvt_f32_f16 x2, v_sub x2, two packs).  The mix instructions read `hi`,
// so they are always at least two instructions behind whatever produced a / b (no transcendental-result hazard can reach them).
__device__ __forceinline__ void split2_f16(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack2_f16(a, b);
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    lo = pack2_f16(la, lb);
}
// 8 values -> one chunk of each plane.  MIX = false leaves the lowering to hipcc: in the fused tail's prologue (x / ctx rows ->
// LDS, 250 live VGPRs) the asm form's operand constraints cost 4-6 spilled registers.
template <bool MIX = true>
__device__ __forceinline__ void split8_f16(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
    if constexpr (MIX) {
        uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
        split2_f16(a[0], a[1], h0, l0); split2_f16(a[2], a[3], h1, l1);
        split2_f16(b[0], b[1], h2, l2); split2_f16(b[2], b[3], h3, l3);
        hi = u32x4{h0, h1, h2, h3}; lo = u32x4{l0, l1, l2, l3};
    } else {
        float h[8], l[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { h[i] = (float)(f16_t)a[i]; l[i] = a[i] - h[i]; h[4 + i] = (float)(f16_t)b[i]; l[4 + i] = b[i] - h[4 + i]; }
        hi[0] = pack2_f16(h[0], h[1]); hi[1] = pack2_f16(h[2], h[3]); hi[2] = pack2_f16(h[4], h[5]); hi[3] = pack2_f16(h[6], h[7]);
        lo[0] = pack2_f16(l[0], l[1]); lo[1] = pack2_f16(l[2], l[3]); lo[2] = pack2_f16(l[4], l[5]); lo[3] = pack2_f16(l[6], l[7]);
    }
}
template <class Tag> __device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo);
template <> __device__ __forceinline__ void split8<TagF16>(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) { split8_f16<false>(a, b, hi, lo); }

// ---- LDS tile addressing.  A tile is [rows][ROWB bytes] with ROWB = 128 or 256; the 16-byte
// slot index within a row is XOR-swizzled with a function of the row so that the 16-lane groups of
// ds_read_b128 / the 32-lane halves of ds_read_b64 hit distinct banks (guide §2 / T2):
//   ROWB = 128: slot ^= (row >> 1) & 7      ROWB = 256: slot ^= row & 15
template <int ROWB> __device__ __forceinline__ int lds_off(int row, int slot16) {
    if constexpr (ROWB == 128) return row * 128 + ((slot16 ^ ((row >> 1) & 7)) << 4);
    else return row * 256 + ((slot16 ^ (row & 15)) << 4);
}

// ---- LDS-DMA: one wave moves 64 x 16 B (1 KB) from global memory straight into LDS; lane l's 16 bytes land at lds + 16 l.
// As inline asm ON PURPOSE: through the builtin (__builtin_amdgcn_global_load_lds) hipcc's s_waitcnt pass books the instruction as
// a FLAT access that may touch LDS, and while one is in flight EVERY later LDS wait becomes lgkmcnt(0) — a fragment prefetched for
// the next group of MFMAs is then waited for together with the current one (seen in the ISA of the attention kernels: 8 exposed LDS
// round trips per 64-key tile).  The asm form is invisible to that pass; completion is tracked by hand (s_waitcnt vmcnt before the
// barrier that publishes the tile), and a compiler-issued vmcnt wait can only over-wait because of it, never under-wait.
// `lds` must be wave-uniform (it travels in M0); one wait state between the M0 write and the DMA instruction.
// INVARIANTS a caller keeps (checked on the compiled ISA by tools/check_isa.py, tests/test_isa_invariants.py):
//   * every barrier that publishes a DMA'd tile is preceded by a vmcnt(0) wait, written as __builtin_amdgcn_s_waitcnt (NOT inline asm: the
//     waitcnt pass must see it, or it keeps waiting for older compiler-visible loads later — and, the counter being in order, for the DMAs
//     issued in between: "over-wait" = an exposed DMA round trip, found in the peeled first tile of the attention kernels in round 4);
//   * no VMEM store sits between a DMA and its wait, and all waits for DMAs are vmcnt(0) (the in-order argument needs nothing else then).
__device__ __forceinline__ void lds_dma16(const void* gptr, void* lds) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(gptr) : "memory", "m0");
}

// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (global_load_lds_dwordx4 v, s[..]): no 64-bit VALU address per request
__device__ __forceinline__ void lds_dma16_s(const void* sbase, uint32_t voff, void* lds) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds);
    const uint64_t b = (uint64_t)(uintptr_t)sbase;
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint64_t sb = ((uint64_t)bhi << 32) | blo;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff), "s"(sb) : "memory", "m0");
}

// ---- weight fragments by buffer load: the fragment-packed weight arrays are read as base descriptor (SGPRs) + one per-lane VGPR offset (16 lane,
// constant for the kernel) + a SCALAR byte offset per fragment.  As plain global loads hipcc built a 64-bit VGPR address per load (v_lshl_add_u64 +
// v_add_co / v_addc pairs and an s_nop in front of each load: 12 VALU + 7 s_nop per k-chunk of the tail's phase A, round-4 ISA), all of it in the gap
// between two MFMA runs; the scalar adds co-issue.  Same cache path, same in-order counter.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ u32x4 weight_frag(__amdgpu_buffer_rsrc_t rs, int lane_off, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, byte_off, 0);
}

// ---- row space
struct RowSpace {
    int B;            // pairs in this forward
    int cap0, cap1;   // per-image row capacity (multiples of 128)
    const int* len;   // [2B] live rows per segment
    const int* active;  // [B] 1 while the pair is still running layers (nullptr = all active)
};
struct TileLoc { int pair, image, seg, r0, grow0; };  // r0 = first local row of the tile
__device__ __forceinline__ TileLoc locate_tile(const RowSpace& rs, int tile, int tile_rows) {
    TileLoc t;
    const int per_pair = rs.cap0 + rs.cap1;
    t.grow0 = tile * tile_rows;
    t.pair = t.grow0 / per_pair;
    const int rem = t.grow0 - t.pair * per_pair;
    t.image = rem >= rs.cap0 ? 1 : 0;
    t.r0 = rem - t.image * rs.cap0;
    t.seg = 2 * t.pair + t.image;
    return t;
}
__device__ __forceinline__ int seg_row_base(const RowSpace& rs, int seg) {
    return (seg >> 1) * (rs.cap0 + rs.cap1) + (seg & 1) * rs.cap0;
}

// Workgroup id runs on XCD id % 8 (round-robin dispatch; tools/tail_wall.py: 1024 of 1024).  For a kernel whose tiles share nothing (the
// fused tail) the identity map hands XCD x every tile = x (mod 8), i.e. one fixed residue of the address bits above the 64 KB tile size,
// for the whole launch — and when the live tiles of a ragged / pruned / partly stopped batch have a period that is a multiple of 8 tiles
// (cfg #5: 40 tiles per pair, live ones at 0..8 and 32..36) some XCDs get 2-3x the work of others.  Rotating the residue by the group
// index gives every XCD every residue in turn and spreads any such pattern; whole inactive pairs still cost every XCD the same (a
// contiguous chunk per XCD, as the attention uses for K / V reuse, does not: cfg #3' -15 %).  Bijective for every grid size.
__device__ __forceinline__ int rr_rotate(int id, int nwg) {
    if (id >= (nwg & ~7)) return id;
    const int q = id >> 3;
    return (q << 3) + ((id + q) & 7);
}

// ---- XCD-aware workgroup order.  The dispatcher places workgroup b on XCD b % 8 (observed, used for
// L2 locality only — never for correctness).  Bijective for any grid size (guide §5 "XCD swizzle must
// be bijective"): XCD x owns the virtual ids [start(x), start(x) + count(x)).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- DPP cross-lane helpers (VALU data path, no LDS traffic unlike ds_bpermute-based __shfl)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Cross-row exchanges without LDS (gfx950 v_permlane16_swap / v_permlane32_swap; __shfl_xor = ds_bpermute is an
// LDS round trip).  With both operands = x: swap16 leaves {x.row0, x.row0, x.row2, x.row2} / {x.row1, x.row1, x.row3,
// x.row3}, swap32 leaves {x.lo, x.lo} / {x.hi, x.hi} — combining the two halves IS the xor-16 / xor-32 reduction step.
// Bare v_max_f32 / v_max3_f32: fmaxf() makes hipcc canonicalise each operand first (v_max x, x, x: one extra VALU op per
// value in IEEE mode, see the guide's T17 / "canonicalising v_max" note); the operands here come straight out of MFMAs
// or other arithmetic and are never signalling NaNs.
__device__ __forceinline__ float vmax2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// (elements are copied to scalars before the bit_cast: __builtin_bit_cast on `r[1]` directly reads element 0 with this clang)
struct SwapPair { float a, b; };
__device__ __forceinline__ SwapPair swap16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned lo = r[0], hi = r[1];
    return {__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)};
}
__device__ __forceinline__ SwapPair swap32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned lo = r[0], hi = r[1];
    return {__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)};
}
__device__ __forceinline__ float xor16_max(float x) { const SwapPair p = swap16(x); return vmax2(p.a, p.b); }
__device__ __forceinline__ float xor32_max(float x) { const SwapPair p = swap32(x); return vmax2(p.a, p.b); }
__device__ __forceinline__ float xor16_sum(float x) { const SwapPair p = swap16(x); return p.a + p.b; }
__device__ __forceinline__ float xor32_sum(float x) { const SwapPair p = swap32(x); return p.a + p.b; }
__device__ __forceinline__ float dpp_xor1(float v) { return dpp_mov<0xB1>(v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float dpp_xor2(float v) { return dpp_mov<0x4E>(v); }   // quad_perm [2,3,0,1]
// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4); every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_mov<0x141>(v);   // row_half_mirror: lane i <- lane 7 - i  (the other quad of the 8-lane half)
    v += dpp_mov<0x140>(v);   // row_mirror:      lane i <- lane 15 - i (the other half of the row)
    return v;
}

// All-reduce of a commutative pair operation over the 64 lanes WITHOUT LDS: four DPP steps inside each row of 16 lanes
// (xor 1, xor 2, half mirror, mirror), then the two permlane swaps across rows.  `__shfl_xor` would be six ds_bpermute
// round trips per value; with 4 waves per SIMD reducing 32 rows each that was 46 % of the log-assignment LSE sweep.
// merge(a0, a1, b0, b1) folds partner (b0, b1) into (a0, a1); both lanes of a pair must end with the same result.
template <class F> __device__ __forceinline__ void wave_allreduce2(float& x0, float& x1, F merge) {
    merge(x0, x1, dpp_xor1(x0), dpp_xor1(x1));
    merge(x0, x1, dpp_xor2(x0), dpp_xor2(x1));
    merge(x0, x1, dpp_mov<0x141>(x0), dpp_mov<0x141>(x1));
    merge(x0, x1, dpp_mov<0x140>(x0), dpp_mov<0x140>(x1));
    { const SwapPair p0 = swap16(x0), p1 = swap16(x1); x0 = p0.a; x1 = p1.a; merge(x0, x1, p0.b, p1.b); }
    { const SwapPair p0 = swap32(x0), p1 = swap32(x1); x0 = p0.a; x1 = p1.a; merge(x0, x1, p0.b, p1.b); }
}

// single-value forms: every lane ends with the maximum / the sum over the 64 lanes (12 VALU ops, no LDS)
__device__ __forceinline__ float wave_allmax(float x) {
    x = vmax2(x, dpp_xor1(x)); x = vmax2(x, dpp_xor2(x)); x = vmax2(x, dpp_mov<0x141>(x)); x = vmax2(x, dpp_mov<0x140>(x));
    return xor32_max(xor16_max(x));
}
__device__ __forceinline__ float wave_allsum(float x) { return xor32_sum(xor16_sum(row16_sum(x))); }

// ---- wave helpers (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace lg
