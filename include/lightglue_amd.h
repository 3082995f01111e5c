/* lightglue_amd — C ABI of the MI355X-native LightGlue matcher forward path.
 *
 * The reference (cvg/LightGlue) has no FFI / plugin boundary: its matcher is the Python class
 * `lightglue.LightGlue` (lightglue/lightglue.py:321, exported at lightglue/__init__.py:4).  This
 * header is the boundary a binding of that class's hot path would sit on: one opaque engine that
 * replaces `LightGlue.__init__` weight handling (lightglue.py:376-437) and `LightGlue._forward`
 * (lightglue.py:483-629).  Plain pointers and sizes only; every data pointer is a DEVICE pointer
 * owned by the caller unless stated otherwise; kernels are enqueued on the caller's HIP stream.
 * Python binding: lightglue_amd/_cabi.py (ctypes).  See INTEGRATION.md.
 */
#ifndef LIGHTGLUE_AMD_H
#define LIGHTGLUE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Operand precision of the matrix-core contractions (accumulation is always fp32). */
enum {
    LG_PREC_F32 = 0,    /* v_mfma_f32_16x16x4_f32: exact fp32 products (parity anchor)                              */
    LG_PREC_BF16 = 1,   /* v_mfma_f32_16x16x32_bf16, one MFMA per product: fast, does NOT hold the 1e-3 score bar    */
    LG_PREC_F16 = 2,    /* v_mfma_f32_16x16x32_f16, one MFMA per product: fast, does NOT hold the 1e-3 score bar     */
    /* 3 was split-bf16 (rounds 1-2): same cost as LG_PREC_F16X3 at 3x its score error — removed in round 3            */
    LG_PREC_F16X3 = 4   /* DEFAULT.  split f16: x = hi + lo (both f16, 22 bits), 3 MFMAs per product, for every linear
                           layer, the similarity matrix AND — with attn_precision = LG_PREC_F16X3, the default — for q k^T
                           and P V of the attention (q / k / v and the probabilities kept as hi + lo planes).  Holds the
                           bar also when attention logits are sharp (recipe-D fixtures).  Needs |x| < 65504, like the
                           reference's own fp16 mode.  attn_precision = LG_PREC_F16 is the fast opt-in: q / k / v as ONE f16
                           plane (what the reference's GPU path feeds its fp16 SDPA, lightglue.py:119), 2 MFMAs per
                           product in the q/k/v projections — within the bar only while attention is diffuse            */
};

#define LG_OK 0
#define LG_ERR_INVALID 1   /* bad argument (the Python shim raises AssertionError/ValueError)      */
#define LG_ERR_HIP 2       /* a HIP runtime call failed; see lg_last_error()                       */
#define LG_ERR_STATE 3     /* call order violated (e.g. forward before weights were finalised)     */
/* per-pair codes of lg_forward_io.status (device array; the call itself returned LG_OK): */
#define LG_ERR_RANGE 4     /* LG_FLAG_CHECK_FINITE: a value of this pair's residual stream or of its q / k / v left the f16 operand range
                              (|x| >= 65504, inf or NaN: the input domain of LG_PREC_F16X3 / _F16 / _BF16); its scores are not to be trusted */
#define LG_ERR_DEVICE 5    /* an internal device-side wait expired (adaptive compaction): results of this forward are invalid */

/* Envelope of one lg_engine_forward / lg_engine_reserve call (round 6).  Inside it every index of every kernel fits the integer type it is
 * computed in, and the largest shapes are covered by GPU tests (N = M = 8192, B = 1; N = M = 4096, B = 32); a call outside it returns
 * LG_ERR_INVALID with a message instead of writing past 32-bit offsets.  Split larger batches across calls (pairs are independent).
 *   keypoints per image                  n0, n1 <= LG_MAX_KEYPOINTS
 *   rows of one forward                  B * (cap0 + cap1) <= LG_MAX_ROWS          (cap = n rounded up to 128; X is then <= 2 GiB)
 *   similarity-matrix elements           B * cap0 * cap1  <= LG_MAX_SIM_ELEMS
 * Input domain of the score bar (|d score| <= 1e-3 against the fp32 reference, LG_PREC_F16X3): descriptors of norm <= ~30 (unit-norm extractor
 * output times up to 10 x the recipes' norm spread of [0.5, 3]); beyond that the network's own fp32 evaluation differs by more than the bar
 * (DESIGN.md section 1).  Values that leave the f16 operand range are reported per pair as LG_ERR_RANGE when LG_FLAG_CHECK_FINITE is set. */
#define LG_WIRE_WIDTH(n0, n1) (3LL * (n0) + 3LL * (n1) + 2)   /* elements of one lg_forward_io.wire row */
#define LG_MAX_KEYPOINTS 8192
#define LG_MAX_ROWS 2097152            /* 2^21 */
#define LG_MAX_SIM_ELEMS 2147483647LL  /* 2^31 - 1 */

/* Mirrors LightGlue.default_conf (lightglue.py:322-335) for the keys the forward path reads. */
typedef struct lg_config {
    int32_t input_dim;          /* 256 SuperPoint, 128 DISK/ALIKED/SIFT (lightglue.py:351-374)   */
    int32_t descriptor_dim;     /* must be 256                                                    */
    int32_t n_layers;           /* 9                                                              */
    int32_t num_heads;          /* must be 4 (head_dim 64)                                        */
    int32_t add_scale_ori;      /* 0/1: 4-D positional input (lightglue.py:495-501)               */
    double depth_confidence;    /* early stop, <= 0 disables (lightglue.py:528)                   */
    double width_confidence;    /* point pruning, <= 0 disables (lightglue.py:529)                */
    double filter_threshold;    /* lightglue.py:592                                               */
    int32_t pruning_min_kpts;   /* resolved LightGlue.pruning_min_kpts() (lightglue.py:658-662)   */
    int32_t precision;          /* LG_PREC_*                                                      */
    int32_t attn_precision;     /* -1 = `precision`; or LG_PREC_F16 together with LG_PREC_F16X3   */
} lg_config;

typedef struct lg_engine lg_engine;

/* Everything one forward reads and writes.  Inputs are dense [B][n][..] fp32, exactly the tensors of
 * the reference's input dict (lightglue.py:460-468); outputs are the fixed-shape tensors of its output
 * dict (lightglue.py:619-629) in int32 (the Python shim widens to int64) plus a device-built compact
 * match list replacing the per-pair torch.where loop (lightglue.py:593-602). */
#define LG_FLAG_NO_PRUNING 1u /* skip point pruning for this call: the reference's padded/compiled
                                 path does the same (lightglue.py:529 `and not do_compile`)        */
#define LG_FLAG_EXT 2u        /* the caller's struct carries the round-5 extension fields (everything after `log_assignment`); without
                                 the flag the engine never reads them (callers built against the round-4 header keep working)          */
#define LG_FLAG_CHECK_FINITE 4u /* range guard (needs LG_FLAG_EXT and `status`): every fused tail and q/k/v projection also tests the
                                 values it is about to split into f16 planes — one compare per value — and flags the pair           */

typedef struct lg_forward_io {
    int32_t batch, n0, n1;
    uint32_t flags;                        /* LG_FLAG_*                                           */
    const float *kpts0, *kpts1;            /* [B][n][2] pixel (x, y)                              */
    const float *desc0, *desc1;            /* [B][n][input_dim]                                   */
    const float *size0, *size1;            /* [B][2] (w, h) or NULL -> bounding-box normalisation */
    const float *scales0, *oris0, *scales1, *oris1; /* [B][n] iff add_scale_ori, else NULL        */
    int32_t *matches0, *matches1;          /* [B][n0], [B][n1]; -1 = unmatched                    */
    float *scores0, *scores1;              /* [B][n0], [B][n1]                                    */
    int32_t *stop;                         /* [B] number of layers executed per pair              */
    int32_t *prune0, *prune1;              /* [B][n] layer counters; required iff pruning enabled */
    int32_t *matches;                      /* [B][min(n0,n1)][2] sorted by index0                 */
    float *match_scores;                   /* [B][min(n0,n1)]                                     */
    int32_t *n_matches;                    /* [B]                                                 */
    /* Ragged batches (replaces the reference's pad_to_length ones-padding + masks,
     * lightglue.py:46-55, :512-520): optional per-pair keypoint counts, num0[b] <= n0, num1[b] <= n1.
     * Rows >= num of the dense inputs are never read; their outputs are -1 / 0 (prune counters 0).
     * Each pair behaves exactly as a separate B=1 call on its first num rows.  NULL = all rows live. */
    const int32_t *num0, *num1;            /* [B] or NULL                                         */
    /* Optional side output: the full log-assignment matrix of
     * sigmoid_log_double_softmax (lightglue.py:265-277) INCLUDING the dustbin row / column, in ORIGINAL
     * keypoint index space: [B][n0+1][n1+1] fp32.  Rows / columns of keypoints that were pruned (or are
     * padding of a ragged batch) hold -inf; the corner is 0 as in the reference.  NULL = not produced
     * (the match outputs never need the matrix materialised). */
    float *log_assignment;
    /* ---- round-5 extension: read only when flags & LG_FLAG_EXT; every pointer optional (NULL = not produced) ----
     * The reference's output dict holds int64 indices and, without pruning, float `prune0/1` (lightglue.py:616-629); the fields
     * below let the engine write those dtypes itself (one kernel at the end of the forward) instead of the caller widening /
     * filling them with framework kernels.  The int32 outputs above stay required: they are what the engine computes in. */
    int64_t *matches0_i64, *matches1_i64;  /* [B][n0], [B][n1]                                                        */
    int64_t *matches_i64;                  /* [B][min(n0,n1)][2] (rows >= n_matches[b] are not written)               */
    int64_t *stop_i64;                     /* [B]                                                                     */
    int64_t *prune0_i64, *prune1_i64;      /* pruning enabled: the final layer counters (lightglue.py:555-558, :605-614) */
    float *prune0_f32, *prune1_f32;        /* pruning disabled: n_layers for live rows (lightglue.py:616-617), 0 for the
                                              padding rows of a ragged batch                                          */
    /* One packed int32 row per pair, the wire format of the pair-sharded multi-GPU path (lightglue_amd/parallel.py, DESIGN.md
     * section 6) — everything the reference's output dict holds for the pair, so that every rank can rebuild the SAME dict for the whole batch:
     *   [matches0 (n0) | scores0 bit patterns (n0) | matches1 (n1) | scores1 bit patterns (n1) | stop | status | prune0 (n0) | prune1 (n1)]
     * prune0/1 are the int32 counters when this forward prunes, else the bit patterns of the reference's float fill (n_layers, lightglue.py:616-617; 0 on
     * the padding rows of a ragged batch).  Row stride wire_stride >= LG_WIRE_WIDTH(n0, n1) elements.  lg_unpack_wire() is the inverse on the receiving side. */
    int32_t *wire; int64_t wire_stride;
    int32_t *status;                       /* [B] LG_OK / LG_ERR_RANGE / LG_ERR_DEVICE per pair                       */
} lg_forward_io;

/* Last error message of the calling thread (never NULL). */
const char* lg_last_error(void);
/* Library / ABI version string. */
const char* lg_version(void);

/* Create an engine on the current HIP device.  Replaces the module construction of
 * LightGlue.__init__ (lightglue.py:388-413). */
int lg_engine_create(const lg_config* cfg, lg_engine** out);
void lg_engine_destroy(lg_engine* e);

/* Stage one state-dict tensor (HOST fp32 pointer, row-major) under its reference name
 * (e.g. "transformers.3.self_attn.Wqkv.weight"; names/shapes: SURVEY.md §8a, lightglue.py:388-407).
 * Replaces load_state_dict (lightglue.py:421/434). */
int lg_engine_set_weight(lg_engine* e, const char* name, const float* host_data, const int64_t* shape, int32_t ndim);
/* Re-pack all staged tensors into kernel layouts / operand precision and upload them. */
int lg_engine_finalize_weights(lg_engine* e);

/* Pre-size the device workspace (otherwise grown on demand inside forward, which synchronises). */
int lg_engine_reserve(lg_engine* e, int32_t max_batch, int32_t max_n0, int32_t max_n1);

/* Enqueue one forward (lightglue.py:483-629) on `hip_stream` (a hipStream_t, NULL = default stream).
 * Asynchronous: no host synchronisation when the workspace is already large enough. */
int lg_engine_forward(lg_engine* e, const lg_forward_io* io, void* hip_stream);

/* Inverse of lg_forward_io.wire on gathered rows (stateless; device pointers; ONE kernel; replaces the slice / widen / index_select chain and the per-pair
 * torch.where of an all-gather consumer): row r of `wire` goes to output pair order[r] (NULL = r; negative = skip the row: the padding rows of a short shard).
 * Writes the reference's output dtypes (lightglue.py:604-629): int64 indices / stop, fp32 scores, prune0/1 as int64 counters (with_prune) or as the float fill,
 * the sorted match list of lightglue.py:593-602 per pair ([pairs_out][kmax][2] int64 + [pairs_out][kmax] scores, kmax = min(n0, n1)) — and `info`,
 * [3][pairs_out] int32 = stop | number of matches | status per OUTPUT pair: the one small block a caller copies to the host (list lengths, B = 1's int stop, and
 * the status every rank must raise on).  Any output pointer may be NULL. */
typedef struct lg_unpack_io {
    const int32_t *wire; int64_t wire_stride; int32_t rows;   /* gathered rows                                           */
    int32_t n0, n1, with_prune, pairs_out;                    /* pairs_out: leading extent of the outputs / of info's rows */
    const int32_t *order;                                     /* [rows] destination pair, or NULL                        */
    int64_t *matches0, *matches1, *stop;                      /* [pairs_out][n0], [pairs_out][n1], [pairs_out]            */
    float *scores0, *scores1;
    int64_t *prune0_i64, *prune1_i64; float *prune0_f32, *prune1_f32;   /* with_prune: the i64 pair, else the f32 pair    */
    int64_t *matches; float *match_scores;                    /* [pairs_out][kmax][2], [pairs_out][kmax]                 */
    int32_t *info;                                            /* [3][pairs_out]: stop, n_matches, status                 */
} lg_unpack_io;
int lg_unpack_wire(const lg_unpack_io* io, void* hip_stream);

/* Engine options (defaults are the product configuration; the others exist for tests, A/B measurements and profiling):
 *   "fused_tail"   1  out_proj + ffn + LayerNorm + GELU + residual as one kernel (lg_tail.hip); 0 = per-op GEMM kernels
 *   "fused_next"   1  the tail kernel also runs the NEXT block's q/k/v projection on the x tile it has just produced
 *                     (bit-identical to the separate kernel; needs fused_tail and 16-bit operand precisions)
 *   "fused_prep"   1  input_dim == 256: keypoint normalisation, Fourier rotary rows, index set and the descriptor copy run inside the first
 *                     projection launch instead of their own kernel (bit-identical; debug stops use the separate kernel)
 *   "attn_rows"    -  query rows per attention wave: 16 | 32 | 64 (bit-identical outputs; the split attention has 16 | 32).  Unset:
 *                     32, and 16 when a launch has fewer 128-row workgroups than the chip has CUs (single pairs); setting it pins
 *                     the shape
 *   "adapt_gather" 1  adaptive width: the SelfBlock projection behind a pruning step gathers its rows through the decide kernel's index map and writes
 *                     them compacted into a second buffer set (no compaction launch; bit-identical to 0 = the in-place compaction kernel)
 *   "sim_planes"   1  precision f16x3: the final projection stores its rows as f16 hi / lo planes and the similarity matrix is multiplied straight from an LDS-DMA
 *                     ring (lg_sim.hip); 0 = fp32 rows + the generic GEMM that splits them per K stage (bit-identical)
 *   "sim_chunk"    0  image-1 rows per workgroup of that kernel: 0 = by grid fill (512 for batches that fill the chip, down to 64 for single pairs), or a multiple of 64
 *   "attn_dma"     1  single-plane 16-bit attention, 32 rows per wave: K / V^T tiles reach LDS by DMA (two buffers, one barrier per
 *                     tile, 4 waves per SIMD); 0 = the register-staged kernel (bit-identical).  The split attention is always DMA
 *   "tail_row_tiles" 0  16-row tiles per fused-tail workgroup: 4 (64 rows) | 2 | 1, 0 = by grid fill (small grids take the
 *                     smaller shapes; bit-identical outputs; the exact fp32 mode always uses 4)
 *   "profile_only" -1 restrict the HIP-event timing of lg_engine_profile_enable to one kernel class (index of
 *                     lg_profile_class_name), -1 = all classes
 *   "tail_timing"  0  shader-clock taps: 1 tail kernel (+ wall-clock life span and CU of every workgroup, tools/tail_wall.py),
 *                     2 self projection, 3 attention (LG_ATTN_TIMING / LG_ATTN_WALL builds) */
int lg_engine_set_option(lg_engine* e, const char* key, int32_t value);

/* ---- test / profiling taps (not used by the product path) ---- */
/* What the matrix pipe sustains on this box: dense bf16 MFMA spin on every SIMD for ~25 ms -> achieved TFLOP/s and the shader
 * clock it ran at (the nominal 2.5 PFLOP/s assumes 2.4 GHz; explains box-to-box throughput differences of one binary). */
int lg_debug_mfma_sustained(double* tflops, double* mhz, void* hip_stream);
/* Stop the next forwards after pipeline step `step` (-1 = run everything).  Step numbering:
 * 0 = prep (+input projection); 1 + 12*layer + k, k = 0 self-QKV, 1 self-attention, 2 out_proj,
 * 3 ffn.0, 4 LayerNorm+GELU, 5 ffn.3+residual, 6 cross-QK/V, 7 cross-attention, 8 to_out, 9 ffn.0,
 * 10 LayerNorm+GELU, 11 ffn.3+residual. */
int lg_engine_debug_stop_after(lg_engine* e, int32_t step);
/* Copy an internal buffer to host (synchronises the device).  Names: X CTX MSG H1 G Q K VT COS SIN MD
 * SIM LS LSNEG CONF MSCORE LSE_R LSE_C IND LEN FINAL_LAYER.  *nbytes_out receives the buffer's byte size;
 * at most max_bytes are copied. */
int lg_engine_debug_read(lg_engine* e, const char* name, void* host_dst, int64_t max_bytes, int64_t* nbytes_out);
/* Row capacities chosen for the last forward (multiples of 128) -> global row = pair*(cap0+cap1) + image*cap0 + r */
int lg_engine_debug_caps(lg_engine* e, int32_t* cap0, int32_t* cap1);

/* ---- per-kernel-class timing with HIP events on the caller's stream (bench.py roofline leg) ---- */
int32_t lg_profile_num_classes(void);
const char* lg_profile_class_name(int32_t cls);
int lg_engine_profile_enable(lg_engine* e, int32_t on);
/* Accumulated milliseconds and launch-site counts per class since the last read (resets). */
int lg_engine_profile_read(lg_engine* e, double* ms, int64_t* count, int32_t n_classes);

/* ---- SuperPoint descriptor head: the producer of the matcher's descriptor tensors ----
 * Replaces, for descriptor_dim 256 and cell size s (8), the tail of SuperPoint.forward
 * (lightglue/superpoint.py:216-228): optional dense L2 normalisation of the descriptor map (:218),
 * sample_descriptors (:80-95: bilinear grid_sample with align_corners=True and zero padding + L2
 * normalisation) and the transpose to [B][N][256] (:228).  Stateless; all pointers are device pointers.
 *   desc_map   [B][256][h][w] fp32 (NCHW)         keypoints [B][N][2] fp32 pixel (x, y)
 *   num        [B] live keypoints per image or NULL; rows >= num[b] of `out` are zero-filled
 *   workspace  [B][h][w][256] fp32 scratch          out       [B][N][256] fp32 */
int lg_sp_sample_descriptors(const float* desc_map, int32_t batch, int32_t channels, int32_t h, int32_t w,
                             const float* keypoints, const int32_t* num, int32_t n, int32_t cell,
                             int32_t normalize_dense, float* workspace, float* out, void* hip_stream);

/* ---- SuperPoint conv stack (replaces superpoint.py:127-141 layers and :159-184, :213-214 of forward) ----
 * lg_sp_pack_conv_weight: repack one nn.Conv2d weight [cout][cin][k][k] (device fp32) into the layout the kernels read,
 *   [k*k][cout][cin] (conv1a: [9][64]); dst holds cout*cin*k*k floats.
 * lg_sp_encode: image [batch][1][h][w] fp32 (grayscale, any h, w >= 8: the 2x2 max-pools floor like nn.MaxPool2d) ->
 *   scores   [batch][h/8*8][w/8*8]    keypoint probabilities after softmax / dustbin removal / depth-to-space (ref :176-184),
 *                                      i.e. cropped to whole 8 x 8 cells like the reference's; the input of lg_sp_detect
 *   desc_map [batch][256][h/8][w/8]   RAW convDb output (NCHW), the input of lg_sp_sample_descriptors(normalize_dense = 1)
 *   params: 24 device pointers = (packed weight, bias) of conv1a, conv1b, conv2a, conv2b, conv3a, conv3b, conv4a, conv4b,
 *           convPa, convPb, convDa, convDb; workspace: lg_sp_encode_workspace_bytes(batch, h, w) bytes.
 * Exact fp32 arithmetic (f32 MFMA): scores agree with an fp32 convolution to round-off. */
int lg_sp_pack_conv_weight(const float* src, int32_t cout, int32_t cin, int32_t k, float* dst, void* hip_stream);
int64_t lg_sp_encode_workspace_bytes(int32_t batch, int32_t h, int32_t w);
int lg_sp_encode(const float* image, int32_t batch, int32_t h, int32_t w, const float* const* params, void* workspace,
                 int64_t workspace_bytes, float* scores, float* desc_map, void* hip_stream);
/* Opt-in split-f16 form of the same stack (SuperPoint(conv_precision="f16x3")): every MFMA convolution multiplies hi + lo f16 planes of its operands (22 bits) with three
 * v_mfma_f32_16x16x32_f16 per product and fp32 accumulation — the dropped lo x lo term is 2^-24-class, i.e. below the summation-order noise of an fp32 convolution — at a
 * multiple of the f32 MFMA rate.  Values must lie inside the f16 range (|x| < 65504); conv1a (K = 9) stays fp32.
 * lg_sp_pack_conv_weight_split: [cout][cin][k][k] fp32 -> hi / lo f16 planes in the order the kernels read (k = 1: [2 planes][cout][cin]; k = 3: MFMA-fragment order
 *   [cout / 64][cin / 32][tap][4 n-tiles][2 planes][64 lanes][8], cout % 64 == 0); dst holds cout*cin*k*k*4 BYTES (as the fp32 form).  cin % 32 == 0.
 * lg_sp_encode_split: as lg_sp_encode, for images of h * w < 2^25 pixels (32-bit element offsets inside one image; larger ones are refused); params[0..1] (conv1a) packed by lg_sp_pack_conv_weight, the weights of the other eleven layers by lg_sp_pack_conv_weight_split. */
int lg_sp_pack_conv_weight_split(const float* src, int32_t cout, int32_t cin, int32_t k, void* dst, void* hip_stream);
int lg_sp_encode_split(const float* image, int32_t batch, int32_t h, int32_t w, const float* const* params, void* workspace,
                       int64_t workspace_bytes, float* scores, float* desc_map, void* hip_stream);

/* Keypoint extraction from SuperPoint's dense score map: replaces simple_nms (lightglue/superpoint.py:52-70), the
 * border removal, thresholding, per-image split (:186-204), top_k_keypoints (:73-77, :207-215) and the (y, x) -> (x, y)
 * float conversion (:218).  Bit-identical to the reference (all comparisons are exact; ties inside top-k, which
 * torch.topk leaves unspecified, go to the lower row-major index).
 *   scores [B][H][W] fp32 (H, W < 32768); nms_radius <= 4; max_keypoints <= 0 keeps every detection in row-major
 *   order, otherwise (<= 4096) the max_keypoints best sorted by score when more were found (the reference's rule).
 *   keypoints [B][capacity][2] (x, y), kp_scores [B][capacity], counts [B] rows written per image;
 *   totals [B] or NULL: detections above the threshold before top-k / clipping (> max_candidates = overflow).
 *   workspace: lg_sp_detect_workspace_bytes(...) bytes of device scratch. */
int64_t lg_sp_detect_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t max_candidates);
int lg_sp_detect(const float* scores, int32_t batch, int32_t h, int32_t w, int32_t nms_radius, int32_t remove_borders,
                 float detection_threshold, int32_t max_keypoints, int32_t capacity, int32_t max_candidates,
                 void* workspace, int64_t workspace_bytes, float* keypoints, float* kp_scores, int32_t* counts,
                 int32_t* totals, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LIGHTGLUE_AMD_H */
