// lightglue_amd — kernel argument structs and host-side launchers (internal C++ interface;
// the public C-ABI is include/lightglue_amd.h).
#pragma once
#include "lg_common.h"

namespace lg {

// ---------------------------------------------------------------- GEMM  (lg_gemm.hip)
// Y[row, n] = sum_k A[row, k] * W[n, k] (+ bias[n]);  A is fp32 in HBM and converted to the operand
// precision while it is staged into LDS; W is pre-packed in the operand precision ([Nout][K], K
// contiguous; hi and lo halves for PREC_F16X3).
enum : int { EPI_STORE = 0, EPI_RESID = 1 };

struct GemmArgs {
    RowSpace rs;
    const float* A;  int lda;     // columns [0, K1) of the contraction
    const float* A2; int lda2;    // columns [K1, K) (the cat[x, msg] of the FFN); nullptr when K1 == K
    int K1, K;
    const void* W;  const void* Wlo;  // packed weights (element type by precision)
    const float* bias;            // [Nout] or nullptr
    int Nout;
    // optional per-pair layer selection (adaptive depth: the final projection uses the weights of
    // the layer at which each pair stopped): W += layer_of_pair[pair] * w_layer_stride (elements)
    const int* layer_of_pair; long long w_layer_stride; long long b_layer_stride;
    // EPI_STORE / EPI_RESID
    float* out; int ldo; float out_scale;
    int R;                        // total rows (= B*(cap0+cap1))
};
hipError_t launch_gemm(int prec, int epi, const GemmArgs& a, hipStream_t s);

// sim[pair][a][b] = sum_k X[row(2*pair, a), k] * X[row(2*pair+1, b), k]   (both operands fp32 rows)
struct SimArgs {
    RowSpace rs;
    const float* X; int ldx; int K;
    float* sim;   // [B][cap0][cap1]
};
hipError_t launch_sim(int prec, const SimArgs& a, hipStream_t s);
// the same product for the split-f16 precision on operands that arrive as f16 planes (lg_sim.hip): md = [2 planes][R][256] f16 (hi, then lo at + plane
// elements), written by the final projection (FinalArgs::planes)
struct SimPlanesArgs {
    RowSpace rs;
    const f16_t* md; long long plane; int K;
    float* sim;   // [B][cap0][cap1]
    int chunk;    // image-1 rows per workgroup (multiple of 64); <= 0: chosen by grid fill
};
hipError_t launch_sim_planes(const SimPlanesArgs& a, hipStream_t s);

// ---------------------------------------------------------------- attention input projections (lg_proj.hip)
// q/k/v (self, rotary on q,k) or qk/v (cross) from the residual stream; weights fragment-packed like TailArgs
// ([Nout][256], columns ordered [group][head][64], hi plane then lo plane).
struct ProjArgs {
    RowSpace rs;
    const float* X;
    const void* W; const float* bias; int Nout;       // 768 (self) or 512 (cross)
    void* q; void* k; void* vt;                      // q,k: [H][R][64]   vt: [H][64][R]; q and k pre-scaled by QK_PRESCALE (lg_proj_body.h)
    long long plane;                                 // split attention (PREC_F16X3): element distance from the hi to the lo plane of q / k / vt
    const float* cosb; const float* sinb;            // rotary tables [R][32] or nullptr
    int n_qk_groups; int R;
    long long* dbg;                                  // profiling tap: [blocks][8 waves][8] shader-clock stamps, or nullptr
    int* range_flag;                                 // [B] or nullptr: LG_FLAG_CHECK_FINITE — set to 1 when a q / k / v value of a live row is not |v| < 65504
};
hipError_t launch_proj(int prec, int attn_prec, const ProjArgs& a, hipStream_t s);
// The SelfBlock projection that follows a pruning step, with the row compaction folded in (round 6): new row r of a segment is the old row src[r]
// (adapt_decide_kernel's inverse map; identity where len_old[seg] < 0), read from the buffers the rows lived in and written — residual row, rotary rows —
// to the OTHER set of buffers (the engine flips the two sets at every pruning layer), so no workgroup ever overwrites a row another one still has to read:
// the in-place compaction kernel, its flags / tickets and its launch are gone from the product path.
struct GatherArgs {
    const float* Xold; const float* cos_old; const float* sin_old;
    float* Xnew; float* cos_new; float* sin_new;
    const int* src;       // [R] segment-local source row of every new row
    const int* len_old;   // [2B] length before this layer's pruning, -1 = pruning not applied to the segment at this layer
};
hipError_t launch_proj_gather(int prec, int attn_prec, const ProjArgs& a, const GatherArgs& g, hipStream_t s);

// ---------------------------------------------------------------- final projection (lg_proj.hip; ref lightglue.py:289-291)
// MD[row, :] = (final_proj(x[row]) ) / 256^0.25 with the weights of the layer each pair stopped at: the same workgroup shape and operand
// scheme as the attention projections (64 rows x all 256 columns, weights fragment-packed like TailArgs), as its own launch (adaptive
// depth: per-pair layer select) or run by the LAST tail kernel on the x tile it has just produced (TailArgs::fin, fixed depth).
struct FinalArgs {
    RowSpace rs;
    const float* X;
    const void* W; const float* bias;               // [layers][256][256] fragment-packed (hi plane then lo plane per layer), [layers][256]
    const int* layer_of_pair; long long w_layer_bytes;   // optional per-pair layer select; nullptr: W / bias point at the layer to use
    float* out; float scale;                         // [R][256] fp32
    int R;
    int planes;                                      // split-f16 precision only: != 0 -> `out` receives f16 hi / lo planes [2][R][256] (same bytes) for launch_sim_planes instead of fp32 rows
    const float* X2; const int* xsel;                // optional: pair p's rows live in X2 when xsel[p] != 0 (the gather path flips buffers per pruning layer; a stopped pair stays where it was)
};
hipError_t launch_final_proj(int prec, const FinalArgs& a, hipStream_t s);

// ---------------------------------------------------------------- fused block tail (lg_tail.hip)
// x <- x + ffn(cat[x, out_proj(ctx)]) with out_proj folded into ffn.0 on the host (see lg_tail.hip).
// Weights are packed in MFMA-fragment order: plane p (hi, lo) at element offset p*rows*512, and within a
// plane element ((nt*NKC + kc)*64 + lane)*EPC + j = W[nt*16 + (lane&15)][kc*4*EPC + (lane>>4)*EPC + j].
struct TailArgs {
    RowSpace rs;
    float* X; const float* CTX;
    const void* Wcat; const float* bcat;     // [512][512] fragment-packed, [512]
    const float* gamma; const float* beta;   // LayerNorm(512)
    const void* W2; const float* b2;         // [256][512] fragment-packed, [256]
    long long* dbg;                          // profiling tap: [blocks][8 waves][8] shader-clock stamps, or nullptr
    int* range_flag;                         // [B] or nullptr: LG_FLAG_CHECK_FINITE — set to 1 when a new x value of a live row is not |x| < 65504 (inf / NaN included)
    int row_tiles;                           // 16-row tiles per workgroup: 4 (default; 0 = 4), 2 or 1 for under-filled grids (16-bit modes)
    // optional 256 -> 1 heads on the NEW x rows, sigmoid applied (token confidence ref :89-94, matchability ref :298-299): up to two
    // weight vectors [256] + bias [1] -> out [R]; nullptr = off.  Replaces a rowdot launch (and its re-read of X) per layer in the
    // adaptive path; the dot products are reduced in a fixed order (lane values, the 4 lane groups, the 8 waves).
    const float* head_w0; const float* head_b0; float* head_out0;
    const float* head_w1; const float* head_b1; float* head_out1;
    // further outputs of the same logits (nullptr = off): logsigmoid(z) and logsigmoid(-z) — the matchability terms of the log assignment
    // (ref :268-276) for the layer a pair ENDS at: written by every CrossBlock tail a pair may end after, so the last write a pair's rows
    // receive is the one of its final layer (replaces the rowdot launch in front of the assignment).  head_out* may be nullptr then.
    float* head_ls0; float* head_lsneg0;
    float* head_ls1; float* head_lsneg1;
    // optional: the NEXT block's q/k/v projection, run on the x tile this kernel has just produced (next.W == nullptr:
    // none).  next.X is unused; supported for 16-bit operand / attention precisions (launch_tail_supports_next).
    ProjArgs next;
    // optional (fin.W != nullptr, instead of `next`): the final projection of the log assignment on the x tile of the LAST block
    FinalArgs fin;
};
bool launch_tail_supports_next(int prec, int attn_prec);
hipError_t launch_tail(int prec, int attn_prec, const TailArgs& a, hipStream_t s);    // 8 waves, one workgroup per CU (lg_tail.hip)
// ---------------------------------------------------------------- attention (lg_attention.hip)
struct AttnArgs {
    RowSpace rs;
    const void* q; const void* k; const void* vt;  // see ProjArgs: q, k arrive pre-scaled, scores are base-2 logits
    long long plane;  // PREC_F16X3: element distance from the hi to the lo plane of q / k / vt
    float* ctx;       // [R][256] fp32, column = head*64 + d
    int R; int cross; // cross: segment s attends to segment s^1 (q and k both read from `q`)
    long long* dbg;   // profiling builds (-DLG_ATTN_TIMING) only: [blocks][4 waves][8] phase clock sums
    int rows_per_wave;   // 16, 32 or 64 query rows per wave (see launch_attention)
    int dma;             // 16-bit, 32 rows per wave: the LDS-DMA kernel (attn_dma_kernel)
};
hipError_t launch_attention(int attn_prec, const AttnArgs& a, hipStream_t s);

// ---------------------------------------------------------------- pointwise (lg_pointwise.hip)
struct PrepArgs {
    RowSpace rs;
    int n0, n1;                    // dense input row counts per image (input tensors are [B][n][..])
    const float* kpts0; const float* kpts1;      // [B][n][2] pixels
    const float* size0; const float* size1;      // [B][2] (w,h) or nullptr -> bbox
    const float* scales0; const float* oris0; const float* scales1; const float* oris1;  // [B][n] or nullptr
    const float* Wr; int pos_dim;  // [32][pos_dim], pos_dim = 2 or 4
    const float* desc0; const float* desc1; int input_dim;  // [B][n][input_dim]
    float* X;                      // [R][256]  (written only when input_dim == 256; else see Xin)
    float* Xin;                    // [R][input_dim] staging for the input projection GEMM
    float* cosb; float* sinb;      // [R][32]
    int* ind;                      // [R] original index of each live row
    float* bbox;                   // [2B][4] scratch (minx,miny,maxx,maxy) when size is absent
};
hipError_t launch_prep(const PrepArgs& a, hipStream_t s);
hipError_t launch_prep_bbox(const PrepArgs& a, hipStream_t s);   // only the bounding boxes (image_size absent); no-op otherwise
// The FIRST SelfBlock projection of a forward with the per-keypoint preparation fused in (input_dim == 256): the workgroup builds the rotary rows of its
// 64 keypoints itself (normalisation + Fourier encoding, written to COS / SIN for the later layers and kept in LDS for its own epilogue), reads the
// descriptor rows straight from the input tensors (and stores them as the fp32 residual stream X) and fills the index set — one launch and one
// 64 MB read + write of the descriptors less than prep_kernel + proj_kernel.  `pa` as for launch_prep (X / cosb / sinb / ind are the outputs).
hipError_t launch_proj_first(int prec, int attn_prec, const ProjArgs& a, const PrepArgs& pa, hipStream_t s);

struct LnGeluArgs { RowSpace rs; const float* h; float* g; const float* gamma; const float* beta; int R; };
hipError_t launch_ln_gelu(const LnGeluArgs& a, hipStream_t s);

// z[row] = x[row,:] . w + b for up to two weight vectors at once (token confidence, matchability)
struct RowDotArgs {
    RowSpace rs; const float* X;
    const float* w0; const float* b0; float* out0; int act0;   // act: 0 raw, 1 sigmoid, 2 logsigmoid(z), 3 logsigmoid(-z)
    const float* w1; const float* b1; float* out1; int act1;   // w1 == nullptr -> skipped
    const int* layer_of_pair; int w_layer_stride;              // optional per-pair layer select
    int ignore_active;
};
hipError_t launch_rowdot(const RowDotArgs& a, hipStream_t s);

// ---------------------------------------------------------------- adaptive (lg_adaptive.hip)
struct AdaptArgs {
    RowSpace rs;            // rs.len / rs.active are read AND written here (non-const alias below)
    int* len; int* active; int* len_old;
    int* final_layer;       // [B]
    int* ind; int* dst;     // [R]
    int* prune0; int* prune1;   // [B][n0], [B][n1] layer counters in ORIGINAL index space
    int n0, n1;
    const int* len_orig;    // [2B] keypoint counts the pair started with (the m + n of ref :550)
    const float* conf; const float* mscore;   // [R] token confidence / matchability (sigmoid)
    float* X; float* cosb; float* sinb;
    int layer; float conf_thr; float depth_conf; float width_conf; int pruning_min_kpts;
    int do_stop, do_prune;
    int compact_chunks;     // chunks per segment = max(cap0, cap1) / compact_chunk_rows()
    int* compact_flags;     // [2B][compact_chunks] "chunk is in registers" flags, value = compact_epoch of the launch that set them
    int compact_epoch;      // > 0, different for every launch (never reset: stale flags of earlier launches compare unequal)
    int* compact_err;       // set to 1 if a bounded flag wait expired (never expected): the chunk's stores are SKIPPED and the forward reports LG_ERR_DEVICE in io->status
    int* compact_ticket;    // work-item counter of adapt_compact_kernel (reset by adapt_decide_kernel of the same launch_adapt)
    // gather mode (round 6; the product path): no compaction launch — adapt_decide_kernel writes the inverse map `src` (new row -> old row), compacts the index
    // set in place, bumps the prune counters (ref :555-558) and records in xsel[pair] which buffer set a continuing pair's rows move to (proj_gather_kernel)
    int gather; int* src; int* xsel; int xnext;
};
hipError_t launch_adapt(const AdaptArgs& a, hipStream_t s);
int compact_chunk_rows();   // rows per compaction work item (lg_adaptive.hip CROWS)

// ---------------------------------------------------------------- assignment (lg_assign.hip)
struct AssignArgs {
    RowSpace rs;
    const float* sim;        // [B][cap0][cap1]
    const float* ls;         // [R] logsigmoid(matchability logit) per row
    float* lse_r; float* lse_c;          // [B][cap0], [B][cap1]
    float* max0; int* arg0;  // [B][cap0] row max / argmax of the score matrix
    float* max1; int* arg1;  // [B][cap1]
    float* cpm; float* cps;  // [B][cap0/32][cap1] column (max, sum-exp) partials per 32-row tile
    float* cbv; int* cbi;    // [B][cap0/32][cap1] column (best score, row) partials per 32-row tile
    const int* ind;          // [R]
    int n0, n1;
    float filter_threshold;
    // outputs in ORIGINAL index space, pre-filled with -1 / 0 by launch_assign
    int* m0; int* m1; float* s0; float* s1;   // [B][n0], [B][n1]
    // compact match list (sorted by index0): [B][min(n0,n1)][2] + count
    int* matches; float* mscores; int* n_matches; int max_matches;
    // optional full log-assignment [B][n0+1][n1+1] (ref :265-277) + logsigmoid(-z) per row for its dustbins
    float* log_assignment; const float* lsneg;
    long long* dbg;   // profiling tap (tail_timing == 4): per sweep workgroup [3] shader-clock stamps
    int all_rows_live; // 1: len == n for every segment (no pruning ever, no ragged counts): the -1 / 0 pre-fill of m/s is skipped
};
hipError_t launch_assign(const AssignArgs& a, hipStream_t s);

// ---------------------------------------------------------------- SuperPoint descriptor head (lg_superpoint.hip)
struct SpArgs {
    const float* desc_map;     // [B][256][h][w] dense descriptor map (NCHW, as the conv stack leaves it)
    float* nhwc;               // [B][h][w][256] workspace: (normalised) location-major copy
    const float* keypoints;    // [B][N][2] pixel (x, y)
    const int* num;            // [B] live keypoints per image or nullptr
    float* out;                // [B][N][256]
    int B, h, w, N, s, normalize_dense;
};
hipError_t launch_sp_sample(const SpArgs& a, hipStream_t s);

// keypoint extraction from the dense score map (ref superpoint.py:52-70 simple_nms, :186-214)
struct SpDetectArgs {
    const float* scores;       // [B][H][W]
    int B, H, W, radius, border; float threshold;
    int max_keypoints;         // top-k (<= 0: keep all, ref :200-208), at most SP_TOPK_MAX
    int capacity;              // rows of the outputs per image
    int max_candidates;        // rows of the candidate buffers per image
    // workspace pieces
    unsigned char* mask_a; unsigned char* mask_b;   // [B][H][W]
    float* nms;                // [B][H][W] scores after non-maximum suppression (0 elsewhere)
    int* row_counts;           // [B][H]
    int* cand_xy; float* cand_score; int* cand_total;   // [B][max_candidates] packed (y << 16 | x), [B][max_candidates], [B]
    // outputs
    float* keypoints; float* kp_scores; int* counts;    // [B][capacity][2] (x, y), [B][capacity], [B]
    int* totals;               // optional [B]: pixels above the threshold before top-k / capacity clipping
};
constexpr int SP_TOPK_MAX = 4096;
// SuperPoint conv stack (lg_sp_encoder.hip): image [B][1][H][W] (any H, W >= 8; floor pooling) -> scores [B][H/8*8][W/8*8], raw descriptor map
// [B][256][H/8][W/8]; P = 24 device pointers (packed weight, bias) x 12 layers; ws = 2 * B*H*W*64 floats
hipError_t launch_sp_encode(const float* image, int B, int H, int W, const float* const* P, float* ws, float* scores, float* desc_map, int split, hipStream_t s);
hipError_t launch_sp_pack_weight_split(const float* src, void* dst, int Cout, int Cin, int k, hipStream_t s);
hipError_t launch_sp_pack_weight(const float* src, float* dst, int Cout, int Cin, int k, hipStream_t s);
hipError_t launch_sp_detect(const SpDetectArgs& a, hipStream_t s);

}  // namespace lg
