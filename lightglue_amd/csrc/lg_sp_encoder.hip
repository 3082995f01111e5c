// lightglue_amd — SuperPoint conv stack on the MI355X (the producer of the matcher's inputs):
//   ref superpoint.py:127-141 (layers) and :159-184, :213-214 (forward): shared VGG-style encoder (8 x conv3x3 + ReLU, 3 x
//   maxpool 2x2), keypoint head (conv3x3 + ReLU, conv1x1 -> 65 logits, softmax, drop the dustbin, depth-to-space 8x8) and
//   descriptor head (conv3x3 + ReLU, conv1x1 -> 256 channels; L2 normalisation + sampling live in lg_superpoint.hip).
// Arithmetic is EXACT fp32 (v_mfma_f32_16x16x4_f32): the detector thresholds / NMS-compares the scores, so they must
// agree with an fp32 convolution to round-off; the f32 MFMA rate (157 TFLOP/s) is not the limiter of this stage.
//
// Convolutions are implicit GEMMs on NHWC activations, C[pixel][cout] = sum_{tap, cin} A[pixel + tap][cin] W[tap][cout][cin]:
//   * one MFMA "chunk" = 16 input channels: lane (lr, g) supplies pixel lr / cout lr and the 4 consecutive channels
//     4g..4g+3 -> ONE 16-byte load per fragment for both operands (weights are repacked [tap][cout][cin] once);
//   * wave = 2 image rows x 32 pixels x 64 output channels (4 x 4 tiles, 64 accumulator registers), workgroup = 4 waves =
//     8 rows: the 9 taps re-read the same input lines from L1/L2, nothing is staged through LDS;
//   * zero padding = clamped address + select; bias / ReLU / the 2x2 max-pool run on the accumulators (a lane holds 4
//     horizontally consecutive pixels of one channel and both rows of the pool window).
#include "lg_kernels.h"

namespace lg {

struct ConvArgs {
    const float* in; float* out; const void* w; const float* bias;
    int B, H, W, Cin, Cout, taps;    // taps = 9 (3x3, pad 1) or 1 (1x1)
    int relu, pool, out_nchw;
};

// bias / ReLU / 2x2 max-pool / stores of one wave's 2 rows x 32 pixels x 64 output channels, straight from the accumulators (shared by the conv kernels)
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[4][4], int b, int n0, int ntl, int x0, int y0, int lr, int g) {
    // ---- epilogue: acc[mt][nt][r] = out[pixel (y0 + mt/2, x0 + (mt&1)*16 + 4g + r)][cout n0 + nt*16 + lr]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int co = n0 + nt * 16 + lr;
        if (nt >= ntl || co >= a.Cout) continue;
        const float bv = a.bias[co];
        f32x4 v[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            v[mt] = acc[mt][nt] + bv;
            if (a.relu) { v[mt][0] = fmaxf(v[mt][0], 0.f); v[mt][1] = fmaxf(v[mt][1], 0.f); v[mt][2] = fmaxf(v[mt][2], 0.f); v[mt][3] = fmaxf(v[mt][3], 0.f); }
        }
        if (a.pool) {        // 2x2 max-pool (ref :161-169): rows y0, y0+1; columns (4g, 4g+1), (4g+2, 4g+3)
            const int H2 = a.H >> 1, W2 = a.W >> 1, yo = y0 >> 1;
            if (yo < H2) {
#pragma unroll
                for (int xt = 0; xt < 2; ++xt) {
                    const float p0 = fmaxf(fmaxf(v[xt][0], v[xt][1]), fmaxf(v[2 + xt][0], v[2 + xt][1]));
                    const float p1 = fmaxf(fmaxf(v[xt][2], v[xt][3]), fmaxf(v[2 + xt][2], v[2 + xt][3]));
                    const int xo = (x0 + xt * 16 + 4 * g) >> 1;
                    float* o = a.out + (((long long)b * H2 + yo) * W2 + xo) * a.Cout + co;
                    if (xo < W2) o[0] = p0;
                    if (xo + 1 < W2) o[a.Cout] = p1;
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int y = y0 + (mt >> 1), x = x0 + (mt & 1) * 16 + 4 * g;
                if (y >= a.H) continue;
                if (a.out_nchw) {    // 4 consecutive pixels of one channel: one 16-byte store when the row allows it
                    float* o = a.out + (((long long)b * a.Cout + co) * a.H + y) * a.W + x;
                    if (x + 3 < a.W && (a.W & 3) == 0) *reinterpret_cast<f32x4*>(o) = v[mt];
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (x + r < a.W) o[r] = v[mt][r];
                    }
                } else {
                    float* o = a.out + (((long long)b * a.H + y) * a.W + x) * a.Cout + co;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (x + r < a.W) o[(long long)r * a.Cout] = v[mt][r];
                }
            }
        }
    }
}

// SPLIT (conv_precision "f16x3", opt-in): the same implicit GEMM on split-f16 operands — activations split into hi + lo f16 planes in registers (x = hi + lo, 22 bits),
// weights pre-split by lg_sp_pack_conv_weight_split ([2 planes][tap][cout][cin] f16, the bytes of the fp32 array), three v_mfma_f32_16x16x32_f16 per product with fp32
// accumulation (hi lo + lo hi + hi hi; the dropped lo lo term is 2^-24-class, below the fp32 convolution's own summation-order noise over K = 576 ... 1152).  One MFMA
// chunk = 32 input channels: lane (lr, g) supplies pixel lr / cout lr and the 8 consecutive channels 8g .. 8g + 7 (two 16-byte fp32 loads per activation fragment,
// one 16-byte load per weight plane).  Same tiles, loop order, epilogue and layouts as the fp32 form.  Measured (profiles/r06sp_*): 1.2 x the fp32 form, not the 5 x of
// the MFMA rates — with 16 KB of operand loads per 48 MFMAs and no LDS staging the kernel is bound by its loads and their address / split arithmetic, whatever the matrix
// instruction costs (a one-step-ahead register prefetch at one wave per SIMD: slower; chunk-outer loop order: the same).
template <bool SPLIT>
__global__ __launch_bounds__(256) void sp_conv_kernel(ConvArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const int ngroups = (a.Cout + 63) >> 6;
    const int b = blockIdx.z / ngroups, n0 = (blockIdx.z - b * ngroups) << 6;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8 + wv * 2;
    if (y0 >= a.H) return;
    f32x4 acc[4][4];   // [mt = ry*2 + xt][nt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* inb = a.in + (long long)b * a.H * a.W * a.Cin;
    constexpr int CPL = SPLIT ? 8 : 4;                        // input channels per lane and chunk
    const int nchunk = a.Cin / (4 * CPL);
    const long long wplane = (long long)a.taps * a.Cout * a.Cin;   // SPLIT: f16 elements from the hi to the lo weight plane
    const int ntl = min(4, (a.Cout - n0 + 15) >> 4);          // live n-tiles of this channel group (wave-uniform)
    for (int tap = 0; tap < a.taps; ++tap) {
        const int dy = a.taps == 9 ? tap / 3 - 1 : 0, dx = a.taps == 9 ? tap % 3 - 1 : 0;
        // source pixel of every m-tile for this tap (clamped; `ok` = inside the image)
        long long poff[4]; bool ok[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int yy = y0 + (mt >> 1) + dy, xx = x0 + (mt & 1) * 16 + lr + dx;
            ok[mt] = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const int yc = min(max(yy, 0), a.H - 1), xc = min(max(xx, 0), a.W - 1);
            poff[mt] = ((long long)yc * a.W + xc) * a.Cin + CPL * g;
        }
        long long wrow[4];   // weight row of every n-tile, in elements (clamped into the matrix; dead lanes are zeroed after the load)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wrow[nt] = ((long long)tap * a.Cout + min(n0 + nt * 16 + lr, a.Cout - 1)) * a.Cin + CPL * g;
        if constexpr (SPLIT) {
            const f16_t* wh = static_cast<const f16_t*>(a.w);
            for (int c = 0; c < nchunk; ++c) {
                u32x4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const float* src = inb + poff[mt] + c * 32;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
                    if (!ok[mt]) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
                    split8<TagF16>(v0, v1, ah[mt], al[mt]);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bool live = nt < ntl && n0 + nt * 16 + lr < a.Cout;
                    const u32x4 h = *reinterpret_cast<const u32x4*>(wh + wrow[nt] + c * 32), l = *reinterpret_cast<const u32x4*>(wh + wplane + wrow[nt] + c * 32);
                    bh[nt] = live ? h : u32x4{0u, 0u, 0u, 0u};
                    bl[nt] = live ? l : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        mma_chunk<TagF16>(acc[mt][nt], ah[mt], bl[nt]);
                        mma_chunk<TagF16>(acc[mt][nt], al[mt], bh[nt]);
                        mma_chunk<TagF16>(acc[mt][nt], ah[mt], bh[nt]);
                    }
            }
        } else {
            const float* wf = static_cast<const float*>(a.w);
            for (int c = 0; c < nchunk; ++c) {
                u32x4 af[4], bf[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(inb + poff[mt] + c * 16);
                    af[mt] = ok[mt] ? v : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const bool live = nt < ntl && n0 + nt * 16 + lr < a.Cout;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(wf + wrow[nt] + c * 16);
                    bf[nt] = live ? v : u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) mma_chunk<TagF32>(acc[mt][nt], af[mt], bf[nt]);
            }
        }
    }
    conv_epilogue(a, acc, b, n0, ntl, x0, y0, lr, g);
}

// ---- 3 x 3 convolutions of the split-f16 form, LDS-staged (conv_precision "f16x3": conv1b ... convPa / convDa; cin % 32 == 0, cout % 64 == 0).
// The register-only form above is bound by its operand loads (PMC: matrix pipe 17 % busy at 1.7 waves per SIMD: every tap re-fetches its pixels from L2 and splits them
// again).  Here a workgroup (4 waves = 8 rows x 32 pixels x 64 output channels, as above) stages the (8 + 2) x (32 + 2) pixel halo tile of ONE 32-channel chunk in LDS —
// fetched once, split once into hi / lo f16 planes (43.5 KB), zero padding applied at staging — and the nine taps read their activation fragments from it (slot XOR pixel
// bits 1..2: two-way, the minimum for a 16-byte read); the next chunk's pixels are requested into registers before the taps run.  Weights arrive as in the matcher's tail
// kernel: pre-packed in MFMA-fragment order ([cout group][chunk][tap][n-tile][plane][lane][8 f16], lg_sp_pack_conv_weight_split with k = 3) and read by raw buffer loads
// one (chunk, tap) step ahead — one fully coalesced 1 KB wave load per fragment, no address arithmetic.  Two barriers per chunk (6 912 matrix cycles per wave).
constexpr int C3_HW = 34, C3_HH = 10, C3_NPX = C3_HW * C3_HH, C3_PLANE = C3_NPX * 64, C3_ITEMS = C3_NPX * 4, C3_ROUNDS = (C3_ITEMS + 255) / 256;
__device__ __forceinline__ int c3_off(int q, int slot) { return q * 64 + ((slot ^ ((q >> 1) & 3)) << 4); }

__global__ __launch_bounds__(256, 2) void sp_conv3x3_split_kernel(ConvArgs a) {     // two workgroups per CU (<= 256 VGPRs, 43.5 KB of LDS each): at one wave per SIMD the launch is 20 % slower
    __shared__ __attribute__((aligned(16))) char smA[2 * C3_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, g = lane >> 4;
    const int ngroups = a.Cout >> 6;
    const int b = blockIdx.z / ngroups, grp = blockIdx.z - b * ngroups, n0 = grp << 6;
    const int x0 = blockIdx.x * 32, yt = blockIdx.y * 8, y0 = yt + wv * 2;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* inb = a.in + (long long)b * a.H * a.W * a.Cin;
    const int nchunk = a.Cin >> 5, nsteps = nchunk * 9;
    const __amdgpu_buffer_rsrc_t wrs = weight_rsrc(static_cast<const char*>(a.w) + (long long)grp * nsteps * 8192);
    const int lane16 = lane * 16;
    // staging: item = (halo pixel q, 16-byte slot): 8 channels; thread tid takes items tid, tid + 256, ...  The coordinates are recomputed per use (a few VALU per
    // round against 432 MFMAs per chunk) instead of kept in 18 registers: the kernel has to fit 256 VGPRs for two workgroups per CU.
    auto item_of = [&](int r, int& q, int& slot, bool& inside, int& src_off) {
        const int item = min(tid + 256 * r, C3_ITEMS - 1);
        q = item >> 2; slot = item & 3;
        const int ry = q / C3_HW, rx = q - ry * C3_HW, yy = yt + ry - 1, xx = x0 + rx - 1;
        inside = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        src_off = (min(max(yy, 0), a.H - 1) * a.W + min(max(xx, 0), a.W - 1)) * a.Cin + 8 * slot;      // (element offset inside one image: H W cin < 2^31 in the extractor's envelope)
    };
    f32x4 raw[C3_ROUNDS][2];
    auto stage_load = [&](int c) {
#pragma unroll
        for (int r = 0; r < C3_ROUNDS; ++r) {
            int q, slot, off; bool inside;
            item_of(r, q, slot, inside, off);
            const float* src = inb + off + c * 32;
            raw[r][0] = *reinterpret_cast<const f32x4*>(src); raw[r][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int r = 0; r < C3_ROUNDS; ++r) {
            int q, slot, off; bool inside;
            item_of(r, q, slot, inside, off);
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x4 hi, lo;
            split8<TagF16>(inside ? raw[r][0] : z, inside ? raw[r][1] : z, hi, lo);
            if (tid + 256 * r < C3_ITEMS) {
                const int dst = c3_off(q, slot);
                *reinterpret_cast<u32x4*>(smA + dst) = hi;
                *reinterpret_cast<u32x4*>(smA + C3_PLANE + dst) = lo;
            }
        }
    };
    u32x4 wf[2][4][2];   // [ring][n-tile][plane]
    auto load_w = [&](u32x4 (&dst)[4][2], int step) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) dst[nt][p] = weight_frag(wrs, lane16, step * 8192 + (nt * 2 + p) * 1024);
    };
    stage_load(0);
    load_w(wf[0], 0);
    // chunks in PAIRS (cin = 64 / 128: 2 or 4 chunks): 18 steps per iteration, so that the two-deep weight ring's slot (step & 1) is a compile-time constant everywhere
    for (int c0 = 0; c0 < nchunk; c0 += 2) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int c = c0 + cc;
            stage_store();
            __syncthreads();                                   // the chunk's halo tile is complete
            stage_load(min(c + 1, nchunk - 1));                // next chunk's pixels, in flight under the nine taps (the last chunk re-fetches itself: never branched around)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int sl = (cc * 9 + tap) & 1;             // ring slot of this step's weights
                const int step = c * 9 + tap;
                load_w(wf[sl ^ 1], min(step + 1, nsteps - 1));
                const int dy = tap / 3, dx = tap % 3;         // halo coordinates: + 1 - 1
                u32x4 ah[4], al[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int q = (2 * wv + (mt >> 1) + dy) * C3_HW + (mt & 1) * 16 + lr + dx;
                    const int off = c3_off(q, g);
                    ah[mt] = *reinterpret_cast<const u32x4*>(smA + off);
                    al[mt] = *reinterpret_cast<const u32x4*>(smA + C3_PLANE + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&w)[4][2] = wf[sl];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        mma_chunk<TagF16>(acc[mt][nt], ah[mt], w[nt][1]);
                        mma_chunk<TagF16>(acc[mt][nt], al[mt], w[nt][0]);
                        mma_chunk<TagF16>(acc[mt][nt], ah[mt], w[nt][0]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                   // every wave is past its last read of the tile
        }
    }
    conv_epilogue(a, acc, b, n0, 4, x0, y0, lr, g);
}

// conv1a (ref :127, :159): 1 -> 64 channels, 3x3, pad 1, ReLU, image [B][1][H][W] -> NHWC.  K = 9: plain VALU.
// thread = (pixel, 4 output channels); weights packed [9][64] (tap-major) and read as float4.
__global__ __launch_bounds__(256) void sp_conv1a_kernel(const float* img, const float* w9x64, const float* bias, float* out, int B, int H, int W) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W * 16;
    if (idx >= total) return;
    const int c4 = (int)(idx & 15);
    const long long pix = idx >> 4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    f32x4 s = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        const float p = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[((long long)b * H + yy) * W + xx] : 0.f;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w9x64 + t * 64 + c4 * 4);
        s[0] = __builtin_fmaf(p, wv[0], s[0]); s[1] = __builtin_fmaf(p, wv[1], s[1]); s[2] = __builtin_fmaf(p, wv[2], s[2]); s[3] = __builtin_fmaf(p, wv[3], s[3]);
    }
    s[0] = fmaxf(s[0], 0.f); s[1] = fmaxf(s[1], 0.f); s[2] = fmaxf(s[2], 0.f); s[3] = fmaxf(s[3], 0.f);
    *reinterpret_cast<f32x4*>(out + pix * 64 + c4 * 4) = s;
}

// keypoint scores (ref :176-184): softmax over the 65 logits of a cell, drop the dustbin, depth-to-space: channel c of cell
// (i, j) -> pixel (8i + c / 8, 8j + c % 8).  One wave per cell.
__global__ __launch_bounds__(256) void sp_scores_kernel(const float* logits, float* scores, int B, int h, int w) {
    const int lane = threadIdx.x & 63;
    const long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= (long long)B * h * w) return;
    const float* p = logits + cell * 65;
    const float v = p[lane], d = p[64];
    const float m = fmaxf(wave_max(v), d);
    const float e = expf(v - m), ed = expf(d - m);
    const float sum = wave_sum(e) + ed;
    const int j = (int)(cell % w), i = (int)((cell / w) % h), b = (int)(cell / ((long long)w * h));
    scores[((long long)b * h * 8 + i * 8 + (lane >> 3)) * (w * 8) + j * 8 + (lane & 7)] = e / sum;
}

// repack a conv weight [Cout][Cin][k][k] -> [k*k][Cout][Cin] (conv1a: [9][64])
__global__ __launch_bounds__(256) void sp_pack_weight_kernel(const float* src, float* dst, int Cout, int Cin, int kk) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)Cout * Cin * kk;
    if (i >= total) return;
    const int t = (int)(i % kk), ci = (int)((i / kk) % Cin), co = (int)(i / ((long long)kk * Cin));
    dst[((long long)t * Cout + co) * Cin + ci] = src[i];
}

// SPLIT form, hi = f16(w), lo = f16(w - hi).  1 x 1 layers: [2 planes][Cout][Cin] f16 (sp_conv_kernel<true>).  3 x 3 layers: MFMA-fragment order for
// sp_conv3x3_split_kernel, [cout group of 64][32-channel chunk][tap][n-tile][plane][lane = 16 g + lr][8 f16]: lane (lr, g) of n-tile nt holds cout 64 grp + 16 nt + lr,
// channels 32 c + 8 g .. + 7 — 1 KB per fragment, 8 KB per (chunk, tap) step.
__global__ __launch_bounds__(256) void sp_pack_weight_split_kernel(const float* src, f16_t* dst, int Cout, int Cin, int kk) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = (long long)Cout * Cin * kk;
    if (i >= total) return;
    const int t = (int)(i % kk), ci = (int)((i / kk) % Cin), co = (int)(i / ((long long)kk * Cin));
    const float w = src[i];
    const f16_t hi = (f16_t)w, lo = (f16_t)(w - (float)hi);
    if (kk == 9) {
        const int grp = co >> 6, nt = (co & 63) >> 4, lr = co & 15, c = ci >> 5, g = (ci & 31) >> 3, e = ci & 7, nchunk = Cin >> 5;
        const long long o = ((((long long)(grp * nchunk + c) * 9 + t) * 4 + nt) * 2) * 512 + (g * 16 + lr) * 8 + e;
        dst[o] = hi;
        dst[o + 512] = lo;
    } else {
        const long long o = ((long long)t * Cout + co) * Cin + ci;
        dst[o] = hi;
        dst[total + o] = lo;
    }
}

hipError_t launch_sp_pack_weight_split(const float* src, void* dst, int Cout, int Cin, int k, hipStream_t s) {
    const long long total = (long long)Cout * Cin * k * k;
    hipLaunchKernelGGL(sp_pack_weight_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, static_cast<f16_t*>(dst), Cout, Cin, k * k);
    return hipGetLastError();
}

hipError_t launch_sp_pack_weight(const float* src, float* dst, int Cout, int Cin, int k, hipStream_t s) {
    const long long total = (long long)Cout * Cin * k * k;
    hipLaunchKernelGGL(sp_pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, Cout, Cin, k * k);
    return hipGetLastError();
}

static void conv(bool split, const float* in, float* out, const float* w, const float* bias, int B, int H, int W, int Cin, int Cout, int taps, int relu, int pool,
                 int nchw, hipStream_t s) {
    ConvArgs a{in, out, w, bias, B, H, W, Cin, Cout, taps, relu, pool, nchw};
    const dim3 grid((W + 31) / 32, (H + 7) / 8, B * ((Cout + 63) / 64));
    if (split && taps == 9) hipLaunchKernelGGL(sp_conv3x3_split_kernel, grid, dim3(256), 0, s, a);      // (every 3 x 3 layer of the stack has cin % 32 == 0 and cout % 64 == 0)
    else if (split) hipLaunchKernelGGL(sp_conv_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(sp_conv_kernel<false>, grid, dim3(256), 0, s, a);
}

// params: packed weight / bias pointers in layer order conv1a, conv1b, conv2a, conv2b, conv3a, conv3b, conv4a, conv4b, convPa, convPb,
// convDa, convDb (24 device pointers).  ws: two ping-pong buffers of B*H*W*64 floats each.
// split != 0: every MFMA convolution on split-f16 operands (weights packed by launch_sp_pack_weight_split; conv1a — K = 9, VALU — stays fp32 with its fp32 packing)
hipError_t launch_sp_encode(const float* image, int B, int H, int W, const float* const* P, float* ws, float* scores, float* desc_map, int split_flag, hipStream_t s) {
    const bool split = split_flag != 0;
    const long long half = (long long)B * H * W * 64;
    float* A = ws; float* Bf = ws + half;
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const long long px = (long long)B * H * W * 16;
    hipLaunchKernelGGL(sp_conv1a_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s, image, P[0], P[1], A, B, H, W);
    conv(split, A, Bf, P[2], P[3], B, H, W, 64, 64, 9, 1, 1, 0, s);           // conv1b + pool  -> [H2][W2][64]
    conv(split, Bf, A, P[4], P[5], B, H2, W2, 64, 64, 9, 1, 0, 0, s);         // conv2a
    conv(split, A, Bf, P[6], P[7], B, H2, W2, 64, 64, 9, 1, 1, 0, s);         // conv2b + pool  -> [H4][W4][64]
    conv(split, Bf, A, P[8], P[9], B, H4, W4, 64, 128, 9, 1, 0, 0, s);        // conv3a
    conv(split, A, Bf, P[10], P[11], B, H4, W4, 128, 128, 9, 1, 1, 0, s);     // conv3b + pool  -> [H8][W8][128]
    conv(split, Bf, A, P[12], P[13], B, H8, W8, 128, 128, 9, 1, 0, 0, s);     // conv4a
    conv(split, A, Bf, P[14], P[15], B, H8, W8, 128, 128, 9, 1, 0, 0, s);     // conv4b         -> x in Bf
    float* x = Bf;
    float* t1 = A;                                                     // [H8][W8][256]
    float* t2 = A + (long long)B * H8 * W8 * 256;                      // [H8][W8][65]
    conv(split, x, t1, P[16], P[17], B, H8, W8, 128, 256, 9, 1, 0, 0, s);     // convPa
    conv(split, t1, t2, P[18], P[19], B, H8, W8, 256, 65, 1, 0, 0, 0, s);     // convPb (logits)
    const long long cells = (long long)B * H8 * W8;
    hipLaunchKernelGGL(sp_scores_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, s, t2, scores, B, H8, W8);
    conv(split, x, t1, P[20], P[21], B, H8, W8, 128, 256, 9, 1, 0, 0, s);     // convDa
    conv(split, t1, desc_map, P[22], P[23], B, H8, W8, 256, 256, 1, 0, 0, 1, s);   // convDb -> NCHW raw descriptor map
    return hipGetLastError();
}

}  // namespace lg
