// lightglue_amd — engine: weight re-packing, workspace, forward orchestration and the C ABI
// (include/lightglue_amd.h).  Replaces LightGlue.__init__ weight handling (ref lightglue.py:376-437)
// and LightGlue._forward (ref :483-629) for the hot path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/lightglue_amd.h"
#include "lg_kernels.h"

using namespace lg;

// kernel classes for lg_engine_profile_read
enum { PC_PREP = 0, PC_GEMM_QKV_SELF, PC_ATTN_SELF, PC_GEMM_OUT, PC_GEMM_FFN1, PC_LN_GELU, PC_GEMM_FFN2, PC_GEMM_QKV_CROSS,
       PC_ATTN_CROSS, PC_ADAPTIVE, PC_ROWDOT, PC_GEMM_FINAL, PC_SIM, PC_ASSIGN, PC_TAIL, LG_PROF_NCLS };
static const char* const kProfNames[LG_PROF_NCLS] = {"prep", "gemm_qkv_self", "attn_self", "gemm_out_proj", "gemm_ffn0", "ln_gelu", "gemm_ffn3_resid",
    "gemm_qkv_cross", "attn_cross", "adaptive", "rowdot", "gemm_final_proj", "sim", "assign", "fused_tail"};

namespace {

thread_local std::string g_err = "";
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail(LG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    } while (0)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- host-side fp32 -> 16-bit conversions (round to nearest even)
inline uint16_t f32_to_bf16(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
inline uint16_t f32_to_f16(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return (uint16_t)sign;               // rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t half_e;
    if (e < -14) { shift = 13 + (-14 - e); half_e = 0; } else { shift = 13; half_e = (uint32_t)(e + 15); }
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (r & 1u))) r++;
    // r includes the implicit bit for normals: (half_e << 10) + (r - 0x400) == ((half_e - 1) << 10) + r
    const uint32_t out = half_e ? (((half_e - 1u) << 10) + r) : r;
    return (uint16_t)(sign | out);
}

inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; ++sh; } u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3ffu) << 13); }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f; std::memcpy(&f, &u, 4); return f;
}

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

// A device weight matrix [rows][K] in operand precision (+ the lo f16 plane for PREC_F16X3)
struct PackedW { void* hi = nullptr; void* lo = nullptr; };

struct DevArena {
    char* base = nullptr; size_t size = 0, used = 0;
    void* take(size_t bytes) { used = (used + 255) & ~size_t(255); void* p = base + used; used += bytes; return p; }
};

}  // namespace

struct lg_engine {
    lg_config cfg{};
    int attn_prec = 0;
    std::map<std::string, HostTensor> staged;
    bool weights_ready = false;
    // ---- device weights (layer-major arrays)
    void* w_arena = nullptr;
    PackedW w_in, w_sout, w_sf1, w_sf2, w_cout, w_cf1, w_cf2;
    float *b_in = nullptr, *b_sqkv = nullptr, *b_sout = nullptr, *b_sf1 = nullptr, *b_sf2 = nullptr, *b_cqkv = nullptr,
          *b_cout = nullptr, *b_cf1 = nullptr, *b_cf2 = nullptr, *b_final = nullptr;
    float *ln_s_g = nullptr, *ln_s_b = nullptr, *ln_c_g = nullptr, *ln_c_b = nullptr;  // [L][512]
    float *w_match = nullptr, *b_match = nullptr, *w_tok = nullptr, *b_tok = nullptr;  // [L][256], [L]
    float* Wr = nullptr;
    // fused block tail (lg_tail.hip): fragment-packed [Wcat | planes] and W2 per layer, folded bias
    char *w_stail_cat = nullptr, *w_stail_2 = nullptr, *w_ctail_cat = nullptr, *w_ctail_2 = nullptr;
    float *b_scat = nullptr, *b_ccat = nullptr;
    size_t tail_cat_layer_bytes = 0, tail_2_layer_bytes = 0;
    char *w_sqkv_p = nullptr, *w_cqkv_p = nullptr, *w_final_p = nullptr;   // fragment-packed projection weights (lg_proj.hip)
    size_t sqkv_layer_bytes = 0, cqkv_layer_bytes = 0, final_layer_bytes = 0;
    bool attn_dma = true;   // option "attn_dma": LDS-DMA attention kernel (16-bit operands, 32 rows per wave)
    int attn_rows = 32;   // query rows per attention wave (32 | 64), option "attn_rows" / env LG_ATTN_ROWS
    int fused_tail = 1, fused_next = 1, fused_prep = 1;   // fused_prep: the per-keypoint preparation inside the first projection launch (input_dim == 256)
    int tail_timing = 0; long long* TAILDBG = nullptr; long long* TAILDBG2 = nullptr;
    int* CFLAGS = nullptr; int compact_epoch = 0; bool cflags_clean = false;   // compaction chunk flags [2B][cap / 128] + 1 error word (lg_adaptive.hip)
    int tail_row_tiles = 0;   // option "tail_row_tiles": 16-row tiles per fused-tail workgroup; 0 = by grid fill (4 | 2 | 1)
    bool attn_auto_rows = true;   // small grids: 16 query rows per attention wave (twice the workgroups); off once "attn_rows" is set
    // ---- workspace
    void* ws = nullptr; size_t ws_bytes = 0;
    int capB = 0, cap0 = 0, cap1 = 0;      // reserved
    int cur_cap0 = 0, cur_cap1 = 0, curB = 0;
    std::map<std::string, std::pair<void*, size_t>> bufs;
    float *X, *CTX, *MSG, *H1, *G, *COS, *SIN, *MD, *SIM, *LS, *LSNEG, *CONF, *MSCORE, *LSE_R, *LSE_C, *MAX0, *MAX1, *BBOX, *XIN, *CPM, *CPS, *CBV;
    int* CBI;
    void *Q, *K, *VT;
    int *IND, *DST, *LEN, *LEN_ORIG, *LEN_OLD, *ACTIVE, *FINAL_LAYER, *ARG0, *ARG1;
    int* RANGEF = nullptr;   // [B] range-guard flags (LG_FLAG_CHECK_FINITE), zeroed by init_state_kernel
    // gather path of the adaptive width (round 6, option "adapt_gather", default on): a second set of residual / rotary buffers — the SelfBlock projection behind
    // a pruning step reads rows from one set and writes the compacted rows to the other (lg_proj.hip proj_gather_kernel) —, and which set each pair's rows are in
    float *X2 = nullptr, *COS2 = nullptr, *SIN2 = nullptr; int* XSEL = nullptr;
    bool adapt_gather = true;
    // split-f16 precision: the final projection stores f16 hi / lo planes and the similarity matrix is sim_planes_kernel (lg_sim.hip); 0 = fp32 rows + the generic sim_kernel (bit-identical)
    bool sim_planes = true;
    int sim_chunk = 0;      // image-1 rows per sim_planes workgroup, 0 = by grid fill (option "sim_chunk": tests / A-B; bit-identical)
    int debug_stop = -1;
    // ---- per-kernel-class HIP-event timing (bench.py roofline leg)
    bool profiling = false, prof_open = false;
    int prof_only = -1;                // kernel class to time alone, -1 = every class
    struct ProfSpan { hipEvent_t a, b; int cls; };
    std::vector<ProfSpan> prof_pool;   // events, reused
    size_t prof_used = 0;
    double prof_ms[LG_PROF_NCLS] = {0};
    long long prof_cnt[LG_PROF_NCLS] = {0};
};

namespace {

size_t elem_size(int prec) { return prec == PREC_F32 ? 4 : 2; }
// bytes per element of the q / k / v^T buffers: the split attention keeps an f16 hi and an f16 lo plane
size_t attn_elem_bytes(int attn_prec) { return attn_prec == PREC_F32 ? 4 : attn_prec == PREC_F16X3 ? 4 : 2; }

// pack a [rows][K] fp32 host matrix into operand precision at device memory (split f16: hi and lo planes)
int upload_packed(int prec, const float* src, size_t n, PackedW& dst, size_t elem_offset) {
    const size_t es = elem_size(prec);
    std::vector<char> hi(n * es), lo;
    if (prec == PREC_F32) std::memcpy(hi.data(), src, n * 4);
    else if (prec == PREC_BF16) { auto* p = reinterpret_cast<uint16_t*>(hi.data()); for (size_t i = 0; i < n; ++i) p[i] = f32_to_bf16(src[i]); }
    else {
        auto* p = reinterpret_cast<uint16_t*>(hi.data());
        for (size_t i = 0; i < n; ++i) p[i] = f32_to_f16(src[i]);
        if (prec == PREC_F16X3) {
            lo.resize(n * es);
            auto* q = reinterpret_cast<uint16_t*>(lo.data());
            for (size_t i = 0; i < n; ++i) q[i] = f32_to_f16(src[i] - f16_to_f32(p[i]));
        }
    }
    HIPCHK(hipMemcpy(static_cast<char*>(dst.hi) + elem_offset * es, hi.data(), n * es, hipMemcpyHostToDevice));
    if (prec_is_split(prec)) HIPCHK(hipMemcpy(static_cast<char*>(dst.lo) + elem_offset * es, lo.data(), n * es, hipMemcpyHostToDevice));
    return LG_OK;
}

// MFMA-fragment order (lg_kernels.h TailArgs): plane-major, then [n-tile][k-chunk][lane][EPC]; split f16: hi = f16(v), lo = f16(v - hi)
int upload_fragment_packed(int prec, const std::vector<double>& W, int rows, int K, char* dst) {
    const size_t es = elem_size(prec);
    const int EPC = prec == PREC_F32 ? 4 : 8, KC = 4 * EPC, NKC = K / KC, NT = rows / 16;
    const bool split = prec_is_split(prec);
    const size_t n = (size_t)rows * K;
    std::vector<char> buf(n * es * (split ? 2 : 1));
    for (int nt = 0; nt < NT; ++nt)
        for (int kc = 0; kc < NKC; ++kc)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < EPC; ++j) {
                    const float v = (float)W[(size_t)(nt * 16 + (lane & 15)) * K + kc * KC + (lane >> 4) * EPC + j];
                    const size_t idx = ((size_t)(nt * NKC + kc) * 64 + lane) * EPC + j;
                    if (prec == PREC_F32) reinterpret_cast<float*>(buf.data())[idx] = v;
                    else if (prec == PREC_BF16) reinterpret_cast<uint16_t*>(buf.data())[idx] = f32_to_bf16(v);
                    else {
                        const uint16_t h = f32_to_f16(v);
                        reinterpret_cast<uint16_t*>(buf.data())[idx] = h;
                        if (split) reinterpret_cast<uint16_t*>(buf.data())[n + idx] = f32_to_f16(v - f16_to_f32(h));
                    }
                }
    HIPCHK(hipMemcpy(dst, buf.data(), buf.size(), hipMemcpyHostToDevice));
    return LG_OK;
}

// Row order of the fragment-packed q/k/v projection weights (lg_proj_body.h pj_tile): n-tile t, MFMA row i -> packed column
// ([group][head][64]).  The n_qk leading groups of 16 tiles each (q, k / qk) are dealt in PAIRS of adjacent tiles whose 32 rows are
// interleaved in blocks of 4 — tile 2p + e, row 4g + r <- channel 32p + 8g + 4e + r — so that a lane of the transposed MFMA form
// ends with 8 consecutive channels (one 16-byte store per plane); v tiles keep the natural order.
std::vector<double> proj_row_permutation(const std::vector<float>& pw, int nout, int n_qk_groups, int K) {
    std::vector<double> out((size_t)nout * K);
    for (int t = 0; t < nout / 16; ++t)
        for (int i = 0; i < 16; ++i) {
            int col = t * 16 + i;
            if (t < 16 * n_qk_groups) {
                const int group = t / 16, tg = t % 16;
                col = group * 256 + 32 * (tg / 2) + 8 * (i >> 2) + 4 * (tg % 2) + (i & 3);
            }
            for (int k = 0; k < K; ++k) out[(size_t)(t * 16 + i) * K + k] = pw[(size_t)col * K + k];
        }
    return out;
}

const HostTensor* find(const lg_engine* e, const std::string& name, std::initializer_list<int64_t> shape, std::string& err) {
    auto it = e->staged.find(name);
    if (it == e->staged.end()) { err = "missing weight '" + name + "'"; return nullptr; }
    if (it->second.shape != std::vector<int64_t>(shape)) { err = "bad shape for '" + name + "'"; return nullptr; }
    return &it->second;
}

__global__ void init_state_kernel(int B, int n0, int n1, int L, const int* num0, const int* num1, int* len, int* len_orig, int* len_old,
                                  int* active, int* final_layer, int* prune0, int* prune1, int* range_flag = nullptr, int* device_err = nullptr, int* xsel = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (range_flag && i < B) range_flag[i] = 0;
    if (xsel && i < B) xsel[i] = 0;
    if (device_err && i == 0) *device_err = 0;
    auto count = [](const int* num, int pair, int n) { int v = num ? num[pair] : n; return v < 0 ? 0 : (v > n ? n : v); };
    if (len && i < B) {
        const int l0 = count(num0, i, n0), l1 = count(num1, i, n1);
        len[2 * i] = l0; len[2 * i + 1] = l1; len_orig[2 * i] = l0; len_orig[2 * i + 1] = l1;
        len_old[2 * i] = -1; len_old[2 * i + 1] = -1;
        // a pair with an empty image never enters the layer loop: stop = 1, empty result (ref :539-540, :568-588)
        const int live = l0 > 0 && l1 > 0;
        active[i] = live; final_layer[i] = live ? L - 1 : 0;
    }
    // prune counters start at 1 for every keypoint (ref :535-536); padding rows of a ragged batch get 0
    if (prune0) for (long long k = i; k < (long long)B * n0; k += (long long)gridDim.x * blockDim.x) prune0[k] = (int)(k % n0) < count(num0, (int)(k / n0), n0);
    if (prune1) for (long long k = i; k < (long long)B * n1; k += (long long)gridDim.x * blockDim.x) prune1[k] = (int)(k % n1) < count(num1, (int)(k / n1), n1);
}
// ---- the last kernel of every forward: `stop` (ref :604 / :575), and — round-5 extension of lg_forward_io — the outputs in the reference's own dtypes
// (int64 indices ref :619-629, float prune0/1 without pruning ref :616-617), the packed wire row of the pair-sharded path, and the per-pair status.
// Replaces the framework kernels the Python shim ran behind the forward (one `.long()` over the int32 block, `torch.full`, the ragged masks) and the
// five slice copies of parallel.py's pack.  grid (ceil(span / 256), B), span = max(n0, n1, 2 * min(n0, n1)).
struct OutArgs {
    int B, n0, n1, L, kmax;
    const int* final_layer; int stop_const;   // final_layer == nullptr: every pair's stop is stop_const (an empty image: 1)
    int* stop;
    const int* m0; const int* m1; const float* s0; const float* s1; const int* matches; const int* n_matches;
    const int* prune0; const int* prune1; const int* num0; const int* num1;
    long long* m0_64; long long* m1_64; long long* matches_64; long long* stop_64; long long* prune0_64; long long* prune1_64;
    float* prune0_f; float* prune1_f;
    int* wire; long long wire_stride; int wire_prune;   // wire_prune: the row's prune block carries the int counters (else the float fill's bits)
    int* status; const int* range_flag; const int* device_err;
};
__global__ __launch_bounds__(256) void write_outputs_kernel(OutArgs a) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int stop = a.final_layer ? a.final_layer[b] + 1 : a.stop_const;
    int* wrow = a.wire ? a.wire + (long long)b * a.wire_stride : nullptr;
    auto count = [](const int* num, int pair, int n) { int v = num ? num[pair] : n; return v < 0 ? 0 : (v > n ? n : v); };
    const int wp = 2 * a.n0 + 2 * a.n1 + 2;   // first element of the wire row's prune block
    if (i == 0) {
        const int status = (a.device_err && *a.device_err) ? LG_ERR_DEVICE : ((a.range_flag && a.range_flag[b]) ? LG_ERR_RANGE : LG_OK);
        a.stop[b] = stop;
        if (a.stop_64) a.stop_64[b] = stop;
        if (wrow) { wrow[wp - 2] = stop; wrow[wp - 1] = status; }
        if (a.status) a.status[b] = status;
    }
    if (i < a.n0) {
        const long long k = (long long)b * a.n0 + i;
        const int m = a.m0[k];
        const float fill = i < count(a.num0, b, a.n0) ? (float)a.L : 0.f;
        if (a.m0_64) a.m0_64[k] = m;
        if (a.prune0_64) a.prune0_64[k] = a.prune0[k];
        if (a.prune0_f) a.prune0_f[k] = fill;
        if (wrow) { wrow[i] = m; wrow[a.n0 + i] = __float_as_int(a.s0[k]); wrow[wp + i] = a.wire_prune ? a.prune0[k] : __float_as_int(fill); }
    }
    if (i < a.n1) {
        const long long k = (long long)b * a.n1 + i;
        const int m = a.m1[k];
        const float fill = i < count(a.num1, b, a.n1) ? (float)a.L : 0.f;
        if (a.m1_64) a.m1_64[k] = m;
        if (a.prune1_64) a.prune1_64[k] = a.prune1[k];
        if (a.prune1_f) a.prune1_f[k] = fill;
        if (wrow) { wrow[2 * a.n0 + i] = m; wrow[2 * a.n0 + a.n1 + i] = __float_as_int(a.s1[k]); wrow[wp + a.n0 + i] = a.wire_prune ? a.prune1[k] : __float_as_int(fill); }
    }
    if (a.matches_64 && i < 2 * a.kmax && (i >> 1) < a.n_matches[b]) {
        const long long k = (long long)b * a.kmax * 2 + i;
        a.matches_64[k] = a.matches[k];
    }
}
// inverse of the wire row on gathered rows (lg_unpack_wire): one workgroup of 1024 threads per gathered row.  Besides the permutation / widening it
// builds the reference's sorted match list (ref :593-602: indices of matches0 > -1 in ascending order, their partners, their scores) by ballot +
// popcount prefix over the row, and the [3][pairs_out] host block (stop | n_matches | status).
struct UnpackArgs {
    const int* wire; long long stride; int n0, n1, with_prune, pairs_out, kmax; const int* order;
    long long* m0; long long* m1; long long* stop; float* s0; float* s1;
    long long* p0_64; long long* p1_64; float* p0_f; float* p1_f;
    long long* matches; float* mscores; int* info;
};
__global__ __launch_bounds__(1024) void unpack_wire_kernel(UnpackArgs a) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n0 = a.n0, n1 = a.n1;
    const int* row = a.wire + (long long)r * a.stride;
    const long long d = a.order ? a.order[r] : r;
    if (d < 0 || d >= a.pairs_out) return;      // a padding row of a short shard
    const int wp = 2 * n0 + 2 * n1 + 2;
    __shared__ int sh_cnt[16];
    int base = 0;                                // matches found in the chunks before this one (the same value in every thread)
    for (int i0 = 0; i0 < n0; i0 += 1024) {
        const int i = i0 + tid;
        int m = -1; float sc = 0.f;
        if (i < n0) {
            m = row[i]; sc = __int_as_float(row[n0 + i]);
            if (a.m0) a.m0[d * n0 + i] = m;
            if (a.s0) a.s0[d * n0 + i] = sc;
            if (a.p0_64) a.p0_64[d * n0 + i] = row[wp + i];
            if (a.p0_f) a.p0_f[d * n0 + i] = __int_as_float(row[wp + i]);
        }
        const bool valid = m > -1;
        const unsigned long long bal = __ballot(valid);
        if (lane == 0) sh_cnt[wv] = __popcll(bal);
        __syncthreads();
        int before = base, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = sh_cnt[w]; if (w < wv) before += c; total += c; }
        if (valid && a.matches) {
            const long long k = d * a.kmax + before + __popcll(bal & ((1ull << lane) - 1ull));
            a.matches[2 * k] = i; a.matches[2 * k + 1] = m;
            if (a.mscores) a.mscores[k] = sc;
        }
        base += total;
        __syncthreads();                         // sh_cnt is rewritten by the next chunk
    }
    for (int i = tid; i < n1; i += 1024) {
        if (a.m1) a.m1[d * n1 + i] = row[2 * n0 + i];
        if (a.s1) a.s1[d * n1 + i] = __int_as_float(row[2 * n0 + n1 + i]);
        if (a.p1_64) a.p1_64[d * n1 + i] = row[wp + n0 + i];
        if (a.p1_f) a.p1_f[d * n1 + i] = __int_as_float(row[wp + n0 + i]);
    }
    if (tid == 0) {
        if (a.stop) a.stop[d] = row[wp - 2];
        if (a.info) { a.info[d] = row[wp - 2]; a.info[a.pairs_out + d] = base; a.info[2 * a.pairs_out + d] = row[wp - 1]; }
    }
}

int ensure_workspace(lg_engine* e, int B, int n0, int n1) {
    const int c0 = round_up(n0 > 0 ? n0 : 1, 128), c1 = round_up(n1 > 0 ? n1 : 1, 128);
    const bool fits = e->ws && B <= e->capB && c0 <= e->cap0 && c1 <= e->cap1;
    if (!fits) {
        const int nB = B > e->capB ? B : e->capB, nc0 = c0 > e->cap0 ? c0 : e->cap0, nc1 = c1 > e->cap1 ? c1 : e->cap1;
        if (e->ws) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(e->ws)); e->ws = nullptr; }
        e->capB = nB; e->cap0 = nc0; e->cap1 = nc1;
        const size_t R = (size_t)nB * (nc0 + nc1), as = attn_elem_bytes(e->attn_prec);
        size_t total = 0;
        auto add = [&](size_t b) { total = ((total + 255) & ~size_t(255)) + b; };
        for (int i = 0; i < 3; ++i) add(R * 256 * 4);           // X CTX MSG
        add(R * 512 * 4); add(R * 512 * 4);                     // H1 G
        add(R * 32 * 4); add(R * 32 * 4);                       // COS SIN
        add(R * 256 * 4);                                       // MD
        add((size_t)nB * nc0 * nc1 * 4);                        // SIM
        for (int i = 0; i < 4; ++i) add(R * 4);                 // LS LSNEG CONF MSCORE
        add((size_t)nB * nc0 * 4); add((size_t)nB * nc1 * 4);   // LSE_R LSE_C
        add((size_t)nB * nc0 * 4); add((size_t)nB * nc1 * 4);   // MAX0 MAX1
        add((size_t)nB * nc0 * 4); add((size_t)nB * nc1 * 4);   // ARG0 ARG1
        for (int i = 0; i < 4; ++i) add((size_t)nB * (nc0 / 32) * nc1 * 4);   // CPM CPS CBV CBI
        add((size_t)nB * 8 * 4);                                // BBOX
        add(R * (size_t)e->cfg.input_dim * 4);                  // XIN
        for (int i = 0; i < 3; ++i) add(R * 256 * as);          // Q K VT
        add(R * 4); add(R * 4);                                 // IND DST
        for (int i = 0; i < 7; ++i) add((size_t)nB * 2 * 4);    // LEN LEN_ORIG LEN_OLD ACTIVE FINAL_LAYER RANGEF XSEL
        add(R * 256 * 4); add(R * 32 * 4); add(R * 32 * 4);     // X2 COS2 SIN2 (gather path)
        add(R / 32 * 8 + 256);                                  // CFLAGS: 2B * max(cap0, cap1) / chunk rows (>= 32) ints, + the error word and the work-item ticket
        add(R * 16); add(R * 16);                               // TAILDBG TAILDBG2 (16 bytes per row: [R / 64 workgroups][8 waves][8 stamps], or [R / 128][16 half-waves ...] of the split attention's taps)
        total += 4096;
        HIPCHK(hipMalloc(&e->ws, total));
        e->ws_bytes = total;
    }
    // carve for the CURRENT (B, c0, c1): buffers are laid out densely for the current shape
    if (fits && e->curB == B && e->cur_cap0 == c0 && e->cur_cap1 == c1 && !e->bufs.empty()) return LG_OK;
    e->curB = B; e->cur_cap0 = c0; e->cur_cap1 = c1;
    DevArena ar{static_cast<char*>(e->ws), e->ws_bytes, 0};
    const size_t R = (size_t)B * (c0 + c1), as = attn_elem_bytes(e->attn_prec);
    e->bufs.clear();
    auto take = [&](const char* name, size_t bytes) { void* p = ar.take(bytes); e->bufs[name] = {p, bytes}; return p; };
    e->X = (float*)take("X", R * 256 * 4); e->CTX = (float*)take("CTX", R * 256 * 4); e->MSG = (float*)take("MSG", R * 256 * 4);
    e->H1 = (float*)take("H1", R * 512 * 4); e->G = (float*)take("G", R * 512 * 4);
    e->COS = (float*)take("COS", R * 32 * 4); e->SIN = (float*)take("SIN", R * 32 * 4);
    e->MD = (float*)take("MD", R * 256 * 4);
    e->SIM = (float*)take("SIM", (size_t)B * c0 * c1 * 4);
    e->LS = (float*)take("LS", R * 4); e->LSNEG = (float*)take("LSNEG", R * 4); e->CONF = (float*)take("CONF", R * 4); e->MSCORE = (float*)take("MSCORE", R * 4);
    e->LSE_R = (float*)take("LSE_R", (size_t)B * c0 * 4); e->LSE_C = (float*)take("LSE_C", (size_t)B * c1 * 4);
    e->MAX0 = (float*)take("MAX0", (size_t)B * c0 * 4); e->MAX1 = (float*)take("MAX1", (size_t)B * c1 * 4);
    e->ARG0 = (int*)take("ARG0", (size_t)B * c0 * 4); e->ARG1 = (int*)take("ARG1", (size_t)B * c1 * 4);
    e->CPM = (float*)take("CPM", (size_t)B * (c0 / 32) * c1 * 4); e->CPS = (float*)take("CPS", (size_t)B * (c0 / 32) * c1 * 4);
    e->CBV = (float*)take("CBV", (size_t)B * (c0 / 32) * c1 * 4); e->CBI = (int*)take("CBI", (size_t)B * (c0 / 32) * c1 * 4);
    e->BBOX = (float*)take("BBOX", (size_t)B * 8 * 4);
    e->XIN = (float*)take("XIN", R * (size_t)e->cfg.input_dim * 4);
    e->Q = take("Q", R * 256 * as); e->K = take("K", R * 256 * as); e->VT = take("VT", R * 256 * as);
    e->IND = (int*)take("IND", R * 4); e->DST = (int*)take("DST", R * 4);
    e->LEN = (int*)take("LEN", (size_t)B * 2 * 4); e->LEN_ORIG = (int*)take("LEN_ORIG", (size_t)B * 2 * 4); e->LEN_OLD = (int*)take("LEN_OLD", (size_t)B * 2 * 4);
    e->ACTIVE = (int*)take("ACTIVE", (size_t)B * 4); e->FINAL_LAYER = (int*)take("FINAL_LAYER", (size_t)B * 4);
    e->RANGEF = (int*)take("RANGEF", (size_t)B * 4);
    e->XSEL = (int*)take("XSEL", (size_t)B * 4);
    e->X2 = (float*)take("X2", R * 256 * 4); e->COS2 = (float*)take("COS2", R * 32 * 4); e->SIN2 = (float*)take("SIN2", R * 32 * 4);
    e->CFLAGS = (int*)take("CFLAGS", (size_t)2 * B * ((c0 > c1 ? c0 : c1) / compact_chunk_rows()) * 4 + 256); e->cflags_clean = false;
    e->TAILDBG = (long long*)take("TAILDBG", R * 16); e->TAILDBG2 = (long long*)take("TAILDBG2", R * 16);
    if (ar.used > e->ws_bytes) return fail(LG_ERR_STATE, "workspace carve overflow");
    return LG_OK;
}

int prof_begin(lg_engine* e, int cls, hipStream_t s) {
    e->prof_open = e->profiling && (e->prof_only < 0 || e->prof_only == cls);   // "profile_only": time one class, leave the rest unbracketed
    if (!e->prof_open) return LG_OK;
    if (e->prof_used == e->prof_pool.size()) {
        lg_engine::ProfSpan sp{};
        HIPCHK(hipEventCreate(&sp.a)); HIPCHK(hipEventCreate(&sp.b));
        e->prof_pool.push_back(sp);
    }
    e->prof_pool[e->prof_used].cls = cls;
    HIPCHK(hipEventRecord(e->prof_pool[e->prof_used].a, s));
    return LG_OK;
}
int prof_end(lg_engine* e, hipStream_t s) {
    if (!e->prof_open) return LG_OK;
    e->prof_open = false;
    HIPCHK(hipEventRecord(e->prof_pool[e->prof_used].b, s));
    e->prof_used++;
    return LG_OK;
}
int prof_collect(lg_engine* e) {
    for (size_t i = 0; i < e->prof_used; ++i) {
        HIPCHK(hipEventSynchronize(e->prof_pool[i].b));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->prof_pool[i].a, e->prof_pool[i].b));
        e->prof_ms[e->prof_pool[i].cls] += ms;
        e->prof_cnt[e->prof_pool[i].cls] += 1;
    }
    e->prof_used = 0;
    return LG_OK;
}

}  // namespace

extern "C" {

int32_t lg_profile_num_classes(void) { return LG_PROF_NCLS; }
const char* lg_profile_class_name(int32_t cls) { return (cls >= 0 && cls < LG_PROF_NCLS) ? kProfNames[cls] : ""; }
int lg_engine_profile_enable(lg_engine* e, int32_t on) {
    if (!e) return fail(LG_ERR_INVALID, "null engine");
    if (!on && e->profiling) { int rc = prof_collect(e); if (rc != LG_OK) return rc; }
    e->profiling = on != 0;
    return LG_OK;
}
int lg_engine_profile_read(lg_engine* e, double* ms, int64_t* count, int32_t n) {
    if (!e || !ms || !count || n < LG_PROF_NCLS) return fail(LG_ERR_INVALID, "bad argument");
    int rc = prof_collect(e);
    if (rc != LG_OK) return rc;
    for (int i = 0; i < LG_PROF_NCLS; ++i) { ms[i] = e->prof_ms[i]; count[i] = e->prof_cnt[i]; e->prof_ms[i] = 0; e->prof_cnt[i] = 0; }
    return LG_OK;
}

const char* lg_last_error(void) { return g_err.c_str(); }
const char* lg_version(void) { return "lightglue_amd 0.4 (gfx950)"; }

int lg_engine_create(const lg_config* cfg, lg_engine** out) {
    if (!cfg || !out) return fail(LG_ERR_INVALID, "null argument");
    if (cfg->descriptor_dim != 256 || cfg->num_heads != 4) return fail(LG_ERR_INVALID, "only descriptor_dim=256, num_heads=4 (head_dim 64) are built");
    if (cfg->n_layers < 1 || cfg->n_layers > 64) return fail(LG_ERR_INVALID, "bad n_layers");
    if (cfg->input_dim <= 0 || cfg->input_dim % 64) return fail(LG_ERR_INVALID, "input_dim must be a positive multiple of 64");
    if (cfg->precision != LG_PREC_F32 && cfg->precision != LG_PREC_BF16 && cfg->precision != LG_PREC_F16 && cfg->precision != LG_PREC_F16X3) return fail(LG_ERR_INVALID, "bad precision");
    auto* e = new lg_engine();
    e->cfg = *cfg;
    int ap = cfg->attn_precision;
    if (ap < 0) ap = cfg->precision;   // f16x3 -> split attention; every other precision -> its own element type
    // pairs with a q/k/v projection kernel (lg_proj.hip): a precision with itself, or f16x3 with one f16 plane (the fast opt-in)
    if (!(ap == cfg->precision || (cfg->precision == PREC_F16X3 && ap == PREC_F16))) { delete e; return fail(LG_ERR_INVALID, "bad attn_precision: must equal `precision`, or LG_PREC_F16 with LG_PREC_F16X3"); }
    e->attn_prec = ap;
    *out = e;
    return LG_OK;
}

void lg_engine_destroy(lg_engine* e) {
    if (!e) return;
    for (auto& sp : e->prof_pool) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    if (e->ws) (void)hipFree(e->ws);
    if (e->w_arena) (void)hipFree(e->w_arena);
    delete e;
}

int lg_engine_set_weight(lg_engine* e, const char* name, const float* host_data, const int64_t* shape, int32_t ndim) {
    if (!e || !name || !host_data || ndim < 0 || ndim > 4) return fail(LG_ERR_INVALID, "bad argument");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(host_data, host_data + n);
    e->staged[name] = std::move(t);
    e->weights_ready = false;
    return LG_OK;
}

int lg_engine_finalize_weights(lg_engine* e) {
    if (!e) return fail(LG_ERR_INVALID, "null engine");
    const int L = e->cfg.n_layers, D = 256, Din = e->cfg.input_dim, prec = e->cfg.precision;
    const int pos_dim = 2 + 2 * (e->cfg.add_scale_ori ? 1 : 0);
    const size_t es = elem_size(prec);
    const bool split = prec_is_split(prec);
    // ---- size the weight arena
    size_t total = 0;
    auto addw = [&](size_t elems) { total = ((total + 255) & ~size_t(255)) + elems * es; if (split) total = ((total + 255) & ~size_t(255)) + elems * es; };
    auto addf = [&](size_t n) { total = ((total + 255) & ~size_t(255)) + n * 4; };
    addw((size_t)D * Din);
    addw((size_t)L * D * D); addw((size_t)L * 512 * 512); addw((size_t)L * D * 512);
    addw((size_t)L * D * D); addw((size_t)L * 512 * 512); addw((size_t)L * D * 512);
    addf(D); addf((size_t)L * 768); addf((size_t)L * D); addf((size_t)L * 512); addf((size_t)L * D);
    addf((size_t)L * 512); addf((size_t)L * D); addf((size_t)L * 512); addf((size_t)L * D); addf((size_t)L * D);
    for (int i = 0; i < 4; ++i) addf((size_t)L * 512);
    addf((size_t)L * D); addf(L); addf((size_t)L * D); addf(L); addf(32 * 4);
    const size_t planes = split ? 2 : 1;
    const size_t cat_layer = (size_t)512 * 512 * es * planes, w2_layer = (size_t)256 * 512 * es * planes;
    for (int i = 0; i < 2; ++i) { total = ((total + 255) & ~size_t(255)) + L * cat_layer; total = ((total + 255) & ~size_t(255)) + L * w2_layer; }
    const size_t sqkv_layer = (size_t)768 * 256 * es * planes, cqkv_layer = (size_t)512 * 256 * es * planes;
    const size_t final_layer = (size_t)256 * 256 * es * planes;
    total = ((total + 255) & ~size_t(255)) + L * sqkv_layer; total = ((total + 255) & ~size_t(255)) + L * cqkv_layer; total = ((total + 255) & ~size_t(255)) + L * final_layer;
    addf((size_t)L * 512); addf((size_t)L * 512);
    total += 4096;
    HIPCHK(hipDeviceSynchronize());   // weights may be in use by forwards still running on any stream
    if (e->w_arena) { HIPCHK(hipFree(e->w_arena)); e->w_arena = nullptr; }
    HIPCHK(hipMalloc(&e->w_arena, total));
    HIPCHK(hipMemset(e->w_arena, 0, total));
    DevArena ar{static_cast<char*>(e->w_arena), total, 0};
    auto takew = [&](PackedW& w, size_t elems) { w.hi = ar.take(elems * es); w.lo = split ? ar.take(elems * es) : nullptr; };
    auto takef = [&](size_t n) { return static_cast<float*>(ar.take(n * 4)); };
    takew(e->w_in, (size_t)D * Din);
    takew(e->w_sout, (size_t)L * D * D); takew(e->w_sf1, (size_t)L * 512 * 512); takew(e->w_sf2, (size_t)L * D * 512);
    takew(e->w_cout, (size_t)L * D * D); takew(e->w_cf1, (size_t)L * 512 * 512); takew(e->w_cf2, (size_t)L * D * 512);
    e->b_in = takef(D); e->b_sqkv = takef((size_t)L * 768); e->b_sout = takef((size_t)L * D); e->b_sf1 = takef((size_t)L * 512); e->b_sf2 = takef((size_t)L * D);
    e->b_cqkv = takef((size_t)L * 512); e->b_cout = takef((size_t)L * D); e->b_cf1 = takef((size_t)L * 512); e->b_cf2 = takef((size_t)L * D); e->b_final = takef((size_t)L * D);
    e->ln_s_g = takef((size_t)L * 512); e->ln_s_b = takef((size_t)L * 512); e->ln_c_g = takef((size_t)L * 512); e->ln_c_b = takef((size_t)L * 512);
    e->w_match = takef((size_t)L * D); e->b_match = takef(L); e->w_tok = takef((size_t)L * D); e->b_tok = takef(L); e->Wr = takef(32 * 4);
    e->tail_cat_layer_bytes = cat_layer; e->tail_2_layer_bytes = w2_layer;
    e->w_stail_cat = static_cast<char*>(ar.take(L * cat_layer)); e->w_stail_2 = static_cast<char*>(ar.take(L * w2_layer));
    e->w_ctail_cat = static_cast<char*>(ar.take(L * cat_layer)); e->w_ctail_2 = static_cast<char*>(ar.take(L * w2_layer));
    e->b_scat = takef((size_t)L * 512); e->b_ccat = takef((size_t)L * 512);
    e->sqkv_layer_bytes = sqkv_layer; e->cqkv_layer_bytes = cqkv_layer;
    e->w_sqkv_p = static_cast<char*>(ar.take(L * sqkv_layer)); e->w_cqkv_p = static_cast<char*>(ar.take(L * cqkv_layer));
    e->final_layer_bytes = final_layer; e->w_final_p = static_cast<char*>(ar.take(L * final_layer));
    if (ar.used > total) return fail(LG_ERR_STATE, "weight arena carve overflow");

    std::string err;
    auto up_f32 = [&](float* dst, const float* src, size_t n) -> int { HIPCHK(hipMemcpy(dst, src, n * 4, hipMemcpyHostToDevice)); return LG_OK; };
#define NEED(var, name, ...) const HostTensor* var = find(e, name, {__VA_ARGS__}, err); if (!var) return fail(LG_ERR_INVALID, err)
#define TRY(x) do { int _rc = (x); if (_rc != LG_OK) return _rc; } while (0)
    {
        NEED(wr, "posenc.Wr.weight", 32, pos_dim);
        TRY(up_f32(e->Wr, wr->data.data(), 32 * (size_t)pos_dim));
    }
    if (Din != D) {
        NEED(w, "input_proj.weight", D, Din); NEED(b, "input_proj.bias", D);
        TRY(upload_packed(prec, w->data.data(), (size_t)D * Din, e->w_in, 0));
        TRY(up_f32(e->b_in, b->data.data(), D));
    }
    for (int i = 0; i < L; ++i) {
        const std::string s = "transformers." + std::to_string(i) + ".self_attn.", c = "transformers." + std::to_string(i) + ".cross_attn.";
        {   // Wqkv: reference channel = head*192 + d*3 + {q,k,v} (ref :166-167) -> packed column = which*256 + head*64 + d
            NEED(w, s + "Wqkv.weight", 768, D); NEED(b, s + "Wqkv.bias", 768);
            std::vector<float> pw((size_t)768 * D), pb(768);
            for (int which = 0; which < 3; ++which) for (int h = 0; h < 4; ++h) for (int d = 0; d < 64; ++d) {
                const int src = h * 192 + d * 3 + which, dst = which * 256 + h * 64 + d;
                std::memcpy(&pw[(size_t)dst * D], &w->data[(size_t)src * D], D * 4);
                pb[dst] = b->data[src];
            }
            TRY(upload_fragment_packed(prec, proj_row_permutation(pw, 768, 2, D), 768, D, e->w_sqkv_p + (size_t)i * sqkv_layer));
            TRY(up_f32(e->b_sqkv + (size_t)i * 768, pb.data(), 768));
        }
        {
            NEED(w, s + "out_proj.weight", D, D); NEED(b, s + "out_proj.bias", D);
            TRY(upload_packed(prec, w->data.data(), (size_t)D * D, e->w_sout, (size_t)i * D * D));
            TRY(up_f32(e->b_sout + (size_t)i * D, b->data.data(), D));
        }
        for (int blk = 0; blk < 2; ++blk) {
            const std::string& p = blk ? c : s;
            NEED(w0, p + "ffn.0.weight", 512, 512); NEED(b0, p + "ffn.0.bias", 512);
            NEED(g, p + "ffn.1.weight", 512); NEED(be, p + "ffn.1.bias", 512);
            NEED(w3, p + "ffn.3.weight", D, 512); NEED(b3, p + "ffn.3.bias", D);
            TRY(upload_packed(prec, w0->data.data(), (size_t)512 * 512, blk ? e->w_cf1 : e->w_sf1, (size_t)i * 512 * 512));
            TRY(up_f32((blk ? e->b_cf1 : e->b_sf1) + (size_t)i * 512, b0->data.data(), 512));
            TRY(up_f32((blk ? e->ln_c_g : e->ln_s_g) + (size_t)i * 512, g->data.data(), 512));
            TRY(up_f32((blk ? e->ln_c_b : e->ln_s_b) + (size_t)i * 512, be->data.data(), 512));
            TRY(upload_packed(prec, w3->data.data(), (size_t)D * 512, blk ? e->w_cf2 : e->w_sf2, (size_t)i * D * 512));
            TRY(up_f32((blk ? e->b_cf2 : e->b_sf2) + (size_t)i * D, b3->data.data(), D));
        }
        for (int blk = 0; blk < 2; ++blk) {   // fused tail: Wcat = [W1x | W1m Wo], bcat = b1 + W1m bo (double precision fold)
            const std::string& p = blk ? c : s;
            const HostTensor* w0 = &e->staged[p + "ffn.0.weight"]; const HostTensor* b0 = &e->staged[p + "ffn.0.bias"];
            const HostTensor* w3 = &e->staged[p + "ffn.3.weight"];
            std::string oname = blk ? c + "to_out" : s + "out_proj";
            NEED(wo, oname + ".weight", D, D); NEED(bo, oname + ".bias", D);
            std::vector<double> cat((size_t)512 * 512), w2d((size_t)256 * 512);
            std::vector<float> bc(512);
            for (int n = 0; n < 512; ++n) {
                const float* w1row = &w0->data[(size_t)n * 512];
                for (int k = 0; k < 256; ++k) cat[(size_t)n * 512 + k] = w1row[k];
                double bacc = b0->data[n];
                for (int j = 0; j < 256; ++j) bacc += (double)w1row[256 + j] * (double)bo->data[j];
                bc[n] = (float)bacc;
                for (int k = 0; k < 256; ++k) {
                    double acc = 0.0;
                    for (int j = 0; j < 256; ++j) acc += (double)w1row[256 + j] * (double)wo->data[(size_t)j * 256 + k];
                    cat[(size_t)n * 512 + 256 + k] = acc;
                }
            }
            for (size_t q = 0; q < w2d.size(); ++q) w2d[q] = w3->data[q];
            TRY(upload_fragment_packed(prec, cat, 512, 512, (blk ? e->w_ctail_cat : e->w_stail_cat) + (size_t)i * cat_layer));
            TRY(upload_fragment_packed(prec, w2d, 256, 512, (blk ? e->w_ctail_2 : e->w_stail_2) + (size_t)i * w2_layer));
            TRY(up_f32((blk ? e->b_ccat : e->b_scat) + (size_t)i * 512, bc.data(), 512));
        }
        {   // cross: [to_qk ; to_v] share one GEMM (both applied to both images, ref :204-205)
            NEED(wq, c + "to_qk.weight", D, D); NEED(bq, c + "to_qk.bias", D);
            NEED(wv, c + "to_v.weight", D, D); NEED(bv, c + "to_v.bias", D);
            NEED(wo, c + "to_out.weight", D, D); NEED(bo, c + "to_out.bias", D);
            std::vector<float> pw((size_t)512 * D), pb(512);
            std::memcpy(pw.data(), wq->data.data(), (size_t)D * D * 4);
            std::memcpy(pw.data() + (size_t)D * D, wv->data.data(), (size_t)D * D * 4);
            std::memcpy(pb.data(), bq->data.data(), D * 4); std::memcpy(pb.data() + D, bv->data.data(), D * 4);
            TRY(upload_fragment_packed(prec, proj_row_permutation(pw, 512, 1, D), 512, D, e->w_cqkv_p + (size_t)i * cqkv_layer));
            TRY(up_f32(e->b_cqkv + (size_t)i * 512, pb.data(), 512));
            TRY(upload_packed(prec, wo->data.data(), (size_t)D * D, e->w_cout, (size_t)i * D * D));
            TRY(up_f32(e->b_cout + (size_t)i * D, bo->data.data(), D));
        }
        {
            const std::string a = "log_assignment." + std::to_string(i) + ".";
            NEED(wf, a + "final_proj.weight", D, D); NEED(bf, a + "final_proj.bias", D);
            NEED(wm, a + "matchability.weight", 1, D); NEED(bm, a + "matchability.bias", 1);
            TRY(upload_fragment_packed(prec, std::vector<double>(wf->data.begin(), wf->data.end()), D, D, e->w_final_p + (size_t)i * final_layer));
            TRY(up_f32(e->b_final + (size_t)i * D, bf->data.data(), D));
            TRY(up_f32(e->w_match + (size_t)i * D, wm->data.data(), D));
            TRY(up_f32(e->b_match + i, bm->data.data(), 1));
        }
        if (i < L - 1) {
            const std::string tkn = "token_confidence." + std::to_string(i) + ".token.0.";
            NEED(wt, tkn + "weight", 1, D); NEED(bt, tkn + "bias", 1);
            TRY(up_f32(e->w_tok + (size_t)i * D, wt->data.data(), D));
            TRY(up_f32(e->b_tok + i, bt->data.data(), 1));
        }
    }
#undef NEED
    e->weights_ready = true;
    return LG_OK;
}

// The envelope of include/lightglue_amd.h (LG_MAX_*): inside it every byte / element offset fits the type it is computed in (the raw-buffer
// descriptors and 32-bit offsets of the compaction, lg_adaptive.hip, reach 4 GB; X is B * (cap0 + cap1) KB) — outside it the hardware would DROP
// stores without an error, so the call is refused instead.
static int check_envelope(long long B, long long n0, long long n1) {
    const long long c0 = (n0 + 127) / 128 * 128, c1 = (n1 + 127) / 128 * 128;
    if (n0 > LG_MAX_KEYPOINTS || n1 > LG_MAX_KEYPOINTS)
        return fail(LG_ERR_INVALID, "more than LG_MAX_KEYPOINTS (8192) keypoints in one image");
    if (B * (c0 + c1) > LG_MAX_ROWS)
        return fail(LG_ERR_INVALID, "batch * (cap0 + cap1) exceeds LG_MAX_ROWS (2^21 keypoint rows per forward): split the batch across calls");
    if (B * c0 * c1 > LG_MAX_SIM_ELEMS)
        return fail(LG_ERR_INVALID, "batch * cap0 * cap1 exceeds LG_MAX_SIM_ELEMS (2^31 - 1 similarity entries per forward): split the batch across calls");
    return LG_OK;
}

int lg_engine_reserve(lg_engine* e, int32_t max_batch, int32_t max_n0, int32_t max_n1) {
    if (!e || max_batch < 1 || max_n0 < 0 || max_n1 < 0) return fail(LG_ERR_INVALID, "bad argument");
    TRY(check_envelope(max_batch, max_n0, max_n1));
    return ensure_workspace(e, max_batch, max_n0, max_n1);
}

int lg_engine_set_option(lg_engine* e, const char* key, int32_t value) {
    if (!e || !key) return fail(LG_ERR_INVALID, "null argument");
    if (std::strcmp(key, "fused_tail") == 0) { e->fused_tail = value != 0; return LG_OK; }
    if (std::strcmp(key, "fused_next") == 0) { e->fused_next = value != 0; return LG_OK; }
    if (std::strcmp(key, "fused_prep") == 0) { e->fused_prep = value != 0; return LG_OK; }
    if (std::strcmp(key, "attn_dma") == 0) { e->attn_dma = value != 0; return LG_OK; }
    if (std::strcmp(key, "adapt_gather") == 0) { e->adapt_gather = value != 0; return LG_OK; }
    if (std::strcmp(key, "sim_planes") == 0) { e->sim_planes = value != 0; return LG_OK; }
    if (std::strcmp(key, "sim_chunk") == 0) { if (value < 0 || value % 64) return fail(LG_ERR_INVALID, "sim_chunk: 0 or a multiple of 64"); e->sim_chunk = value; return LG_OK; }
    if (std::strcmp(key, "attn_rows") == 0) { if (value != 16 && value != 32 && value != 64) return fail(LG_ERR_INVALID, "attn_rows must be 16, 32 or 64"); e->attn_rows = value; e->attn_auto_rows = false; return LG_OK; }
    if (std::strcmp(key, "tail_row_tiles") == 0) { if (value != 0 && value != 1 && value != 2 && value != 4) return fail(LG_ERR_INVALID, "tail_row_tiles must be 0 (automatic), 1, 2 or 4"); e->tail_row_tiles = value; return LG_OK; }
    if (std::strcmp(key, "profile_only") == 0) { e->prof_only = value; return LG_OK; }   // kernel class index, -1 = all classes
    if (std::strcmp(key, "tail_timing") == 0) { e->tail_timing = value; return LG_OK; }   // 1: tail kernel, 2: self projection, 3: self attention (LG_ATTN_TIMING builds), 4: assign sweeps, 5 / 6: layer 0's CrossBlock / SelfBlock tail WITH its fused next projection (stamps of the projection in TAILDBG2)
    return fail(LG_ERR_INVALID, std::string("unknown option '") + key + "'");
}

int lg_engine_debug_stop_after(lg_engine* e, int32_t step) { if (!e) return fail(LG_ERR_INVALID, "null engine"); e->debug_stop = step; return LG_OK; }

int lg_sp_sample_descriptors(const float* desc_map, int32_t batch, int32_t channels, int32_t h, int32_t w, const float* keypoints,
                             const int32_t* num, int32_t n, int32_t cell, int32_t normalize_dense, float* workspace, float* out,
                             void* hip_stream) {
    if (channels != 256) return fail(LG_ERR_INVALID, "descriptor map must have 256 channels");
    if (batch < 1 || h < 1 || w < 1 || n < 0 || cell < 1) return fail(LG_ERR_INVALID, "bad descriptor map / keypoint sizes");
    if (!desc_map || !workspace || (n && (!keypoints || !out))) return fail(LG_ERR_INVALID, "null pointer");
    SpArgs a{desc_map, workspace, keypoints, num, out, batch, h, w, n, cell, normalize_dense ? 1 : 0};
    HIPCHK(launch_sp_sample(a, static_cast<hipStream_t>(hip_stream)));
    return LG_OK;
}

namespace {
// Fused tail: 16-row tiles per workgroup by grid fill.  A 64-row workgroup costs ~114k cycles (matrix-bound), a 32- / 16-row
// one ~69k / ~58k (each streams the full weight set from L2): with R rows on 256 CUs take the shape with the shortest
// critical path — 64 rows as soon as they fill the chip, smaller tiles for single pairs (B = 1, N = 1024: 32 -> 128 workgroups).
int tail_row_tiles_for(int R) {
    int best = 4; long long best_cost = -1;
    const int shapes[3] = {4, 2, 1}; const long long cost[3] = {114, 69, 58};   // measured: 56 / 34 / 28.6 us per launch at B = 1, N = 1024
    for (int i = 0; i < 3; ++i) {
        const long long wgs = R / (16 * shapes[i]), rounds = (wgs + 255) / 256, c = rounds * cost[i];
        if (best_cost < 0 || c < best_cost) { best = shapes[i]; best_cost = c; }
    }
    return best;
}
struct SpLayout { size_t mask_a, mask_b, nms, rows, cxy, csc, ctot, total; };
SpLayout sp_layout(int B, int h, int w, int maxc) {
    SpLayout l{}; size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~size_t(255); return o; };
    const size_t px = (size_t)B * h * w;
    l.mask_a = take(px); l.mask_b = take(px); l.nms = take(px * 4); l.rows = take((size_t)B * h * 4);
    l.cxy = take((size_t)B * maxc * 4); l.csc = take((size_t)B * maxc * 4); l.ctot = take((size_t)B * 4); l.total = off;
    return l;
}
}  // namespace

int lg_sp_pack_conv_weight(const float* src, int32_t cout, int32_t cin, int32_t k, float* dst, void* hip_stream) {
    if (!src || !dst || cout < 1 || cin < 1 || (k != 1 && k != 3)) return fail(LG_ERR_INVALID, "bad conv weight");
    HIPCHK(launch_sp_pack_weight(src, dst, cout, cin, k, static_cast<hipStream_t>(hip_stream)));
    return LG_OK;
}

int64_t lg_sp_encode_workspace_bytes(int32_t batch, int32_t h, int32_t w) {
    if (batch < 1 || h < 8 || w < 8) return 0;
    return (int64_t)2 * batch * h * w * 64 * 4;
}

int lg_sp_encode(const float* image, int32_t batch, int32_t h, int32_t w, const float* const* params, void* workspace,
                 int64_t workspace_bytes, float* scores, float* desc_map, void* hip_stream) {
    if (batch < 1 || h < 8 || w < 8) return fail(LG_ERR_INVALID, "image height / width must be at least 8");
    if (!image || !params || !workspace || !scores || !desc_map) return fail(LG_ERR_INVALID, "null pointer");
    if (workspace_bytes < lg_sp_encode_workspace_bytes(batch, h, w)) return fail(LG_ERR_INVALID, "workspace too small (lg_sp_encode_workspace_bytes)");
    for (int i = 0; i < 24; ++i) if (!params[i]) return fail(LG_ERR_INVALID, "null layer parameter");
    HIPCHK(launch_sp_encode(image, batch, h, w, params, static_cast<float*>(workspace), scores, desc_map, 0, static_cast<hipStream_t>(hip_stream)));
    return LG_OK;
}

int lg_sp_pack_conv_weight_split(const float* src, int32_t cout, int32_t cin, int32_t k, void* dst, void* hip_stream) {
    if (!src || !dst || cout < 1 || cin < 32 || cin % 32 || (k != 1 && k != 3) || (k == 3 && cout % 64)) return fail(LG_ERR_INVALID, "bad conv weight (split form: cin a multiple of 32; 3 x 3 layers: cout a multiple of 64)");
    HIPCHK(launch_sp_pack_weight_split(src, dst, cout, cin, k, static_cast<hipStream_t>(hip_stream)));
    return LG_OK;
}

int lg_sp_encode_split(const float* image, int32_t batch, int32_t h, int32_t w, const float* const* params, void* workspace,
                       int64_t workspace_bytes, float* scores, float* desc_map, void* hip_stream) {
    if (batch < 1 || h < 8 || w < 8) return fail(LG_ERR_INVALID, "image height / width must be at least 8");
    // sp_conv3x3_split_kernel addresses the pixels of ONE image with 32-bit element offsets (h w 64 channels at full resolution)
    if ((int64_t)h * w * 64 >= (int64_t)1 << 31) return fail(LG_ERR_INVALID, "split-f16 conv stack: h * w must stay below 2^25 pixels (use conv_precision = fp32 for larger images)");
    if (!image || !params || !workspace || !scores || !desc_map) return fail(LG_ERR_INVALID, "null pointer");
    if (workspace_bytes < lg_sp_encode_workspace_bytes(batch, h, w)) return fail(LG_ERR_INVALID, "workspace too small (lg_sp_encode_workspace_bytes)");
    for (int i = 0; i < 24; ++i) if (!params[i]) return fail(LG_ERR_INVALID, "null layer parameter");
    HIPCHK(launch_sp_encode(image, batch, h, w, params, static_cast<float*>(workspace), scores, desc_map, 1, static_cast<hipStream_t>(hip_stream)));
    return LG_OK;
}

int64_t lg_sp_detect_workspace_bytes(int32_t batch, int32_t h, int32_t w, int32_t max_candidates) {
    if (batch < 1 || h < 1 || w < 1 || max_candidates < 1) return 0;
    return (int64_t)sp_layout(batch, h, w, max_candidates).total;
}

int lg_sp_detect(const float* scores, int32_t batch, int32_t h, int32_t w, int32_t nms_radius, int32_t remove_borders,
                 float detection_threshold, int32_t max_keypoints, int32_t capacity, int32_t max_candidates, void* workspace,
                 int64_t workspace_bytes, float* keypoints, float* kp_scores, int32_t* counts, int32_t* totals, void* hip_stream) {
    if (batch < 1 || h < 1 || w < 1 || h >= 32768 || w >= 32768) return fail(LG_ERR_INVALID, "bad score map size");
    if (nms_radius < 0 || nms_radius > 4) return fail(LG_ERR_INVALID, "nms_radius must be in [0, 4]");
    if (max_keypoints > SP_TOPK_MAX) return fail(LG_ERR_INVALID, "max_keypoints above 4096");
    if (capacity < 1 || max_candidates < 1 || (max_keypoints > 0 && capacity < max_keypoints)) return fail(LG_ERR_INVALID, "bad capacity / max_candidates");
    if (!scores || !workspace || !keypoints || !kp_scores || !counts) return fail(LG_ERR_INVALID, "null pointer");
    const SpLayout l = sp_layout(batch, h, w, max_candidates);
    if (workspace_bytes < (int64_t)l.total) return fail(LG_ERR_INVALID, "workspace too small (lg_sp_detect_workspace_bytes)");
    char* ws = static_cast<char*>(workspace);
    SpDetectArgs a{};
    a.scores = scores; a.B = batch; a.H = h; a.W = w; a.radius = nms_radius; a.border = remove_borders; a.threshold = detection_threshold;
    a.max_keypoints = max_keypoints; a.capacity = capacity; a.max_candidates = max_candidates;
    a.mask_a = reinterpret_cast<unsigned char*>(ws + l.mask_a); a.mask_b = reinterpret_cast<unsigned char*>(ws + l.mask_b);
    a.nms = reinterpret_cast<float*>(ws + l.nms); a.row_counts = reinterpret_cast<int*>(ws + l.rows);
    a.cand_xy = reinterpret_cast<int*>(ws + l.cxy); a.cand_score = reinterpret_cast<float*>(ws + l.csc); a.cand_total = reinterpret_cast<int*>(ws + l.ctot);
    a.keypoints = keypoints; a.kp_scores = kp_scores; a.counts = counts; a.totals = totals;
    HIPCHK(launch_sp_detect(a, static_cast<hipStream_t>(hip_stream)));
    return LG_OK;
}

namespace {
// matrix-core-dense spin: 2 waves per SIMD, 8 independent accumulators per wave (the pipe never waits), operands with
// pseudo-random bits (data that toggles: an all-zero spin runs ~15 % faster at the same power); block 0 reports its
// shader-clock span
__global__ __launch_bounds__(512) void mfma_spin_kernel(long long* cycles, int iters) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    const unsigned h = (threadIdx.x * 2654435761u + blockIdx.x * 40503u) * 12345u;
    u32x4 xa = {h ^ 0x3f803f80u, (h >> 3) | 0x3c003c00u, (h * 7u) & 0x3fff3fffu, (h * 13u) & 0x3fff3fffu};
    u32x4 xb = {(h * 3u) & 0x3fff3fffu, (h * 5u) & 0x3fff3fffu, (h * 11u) & 0x3fff3fffu, (h * 17u) & 0x3fff3fffu};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(xa), "+v"(xb));
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, xa), __builtin_bit_cast(bf16x8_t, xb), acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    float sacc = 0.f;
    for (int i = 0; i < 8; ++i) sacc += acc[i][0];
    // the LONGEST wave span: the arbiter issues oldest-first, so the older wave of a SIMD can finish in half the kernel's time
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(cycles), (unsigned long long)(t1 - t0));
    if (sacc == 12345.f) cycles[1] = 1;
}
}  // namespace

/* What the matrix pipe SUSTAINS on this box: a dense v_mfma_f32_16x16x32_bf16 spin on every SIMD for ~25 ms (long enough for
 * the power management to settle).  tflops = achieved dense bf16 rate (the nominal 2.5 PFLOP/s assumes 2.4 GHz; under this load
 * the boxes of the pool hold 1.8 - 2.1 GHz), mhz = shader clock during the spin (s_memtime span of one wave / HIP-event time). */
int lg_debug_mfma_sustained(double* tflops, double* mhz, void* hip_stream) {
    if (!tflops || !mhz) return fail(LG_ERR_INVALID, "null pointer");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    long long* d = nullptr;
    HIPCHK(hipMalloc(&d, 16));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const int iters = 400000;
    hipLaunchKernelGGL(mfma_spin_kernel, dim3(256), dim3(512), 0, s, d, iters / 10);   // warm-up / clock ramp
    HIPCHK(hipMemsetAsync(d, 0, 16, s));
    HIPCHK(hipEventRecord(a, s));
    hipLaunchKernelGGL(mfma_spin_kernel, dim3(256), dim3(512), 0, s, d, iters);
    HIPCHK(hipEventRecord(b, s));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f; long long h[2] = {0, 0};
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    HIPCHK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    (void)hipFree(d); (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *mhz = ms > 0.f ? (double)h[0] / (ms * 1e3) : 0.0;
    *tflops = ms > 0.f ? 256.0 * 8.0 * iters * 8.0 * 16384.0 / (ms * 1e9) : 0.0;
    return LG_OK;
}

int lg_engine_debug_caps(lg_engine* e, int32_t* cap0, int32_t* cap1) {
    if (!e || !cap0 || !cap1) return fail(LG_ERR_INVALID, "null argument");
    *cap0 = e->cur_cap0; *cap1 = e->cur_cap1;
    return LG_OK;
}

int lg_engine_debug_read(lg_engine* e, const char* name, void* host_dst, int64_t max_bytes, int64_t* nbytes_out) {
    if (!e || !name) return fail(LG_ERR_INVALID, "null argument");
    auto it = e->bufs.find(name);
    if (it == e->bufs.end()) return fail(LG_ERR_INVALID, std::string("unknown buffer '") + name + "'");
    if (nbytes_out) *nbytes_out = (int64_t)it->second.second;
    HIPCHK(hipDeviceSynchronize());
    if (host_dst && max_bytes > 0) {
        const size_t n = (size_t)max_bytes < it->second.second ? (size_t)max_bytes : it->second.second;
        HIPCHK(hipMemcpy(host_dst, it->second.first, n, hipMemcpyDeviceToHost));
    }
    return LG_OK;
}

int lg_engine_forward(lg_engine* e, const lg_forward_io* io, void* hip_stream) {
    if (!e || !io) return fail(LG_ERR_INVALID, "null argument");
    if (!e->weights_ready) return fail(LG_ERR_STATE, "weights not finalised");
    const int B = io->batch, n0 = io->n0, n1 = io->n1, L = e->cfg.n_layers, D = 256;
    if (B < 1 || n0 < 0 || n1 < 0) return fail(LG_ERR_INVALID, "bad batch / keypoint counts");
    TRY(check_envelope(B, n0, n1));
    if (!io->stop || !io->n_matches) return fail(LG_ERR_INVALID, "null output pointer");
    if ((n0 && (!io->matches0 || !io->scores0)) || (n1 && (!io->matches1 || !io->scores1))) return fail(LG_ERR_INVALID, "null output pointer");
    if (n0 && n1 && (!io->matches || !io->match_scores)) return fail(LG_ERR_INVALID, "null match-list pointer");
    if (n0 && n1 && (!io->kpts0 || !io->kpts1 || !io->desc0 || !io->desc1)) return fail(LG_ERR_INVALID, "null input pointer");
    const bool do_stop = e->cfg.depth_confidence > 0;
    const bool do_prune = e->cfg.width_confidence > 0 && !(io->flags & LG_FLAG_NO_PRUNING);
    if (do_prune && ((n0 && !io->prune0) || (n1 && !io->prune1))) return fail(LG_ERR_INVALID, "prune0/prune1 required when width_confidence > 0");
    if (e->cfg.add_scale_ori && (!io->scales0 || !io->oris0 || !io->scales1 || !io->oris1)) return fail(LG_ERR_INVALID, "scales/oris required");
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    const int max_matches = n0 < n1 ? n0 : n1;
    const bool ext = (io->flags & LG_FLAG_EXT) != 0;            // round-5 extension fields present
    const bool check_finite = ext && (io->flags & LG_FLAG_CHECK_FINITE) != 0;
    if ((io->flags & LG_FLAG_CHECK_FINITE) && (!ext || !io->status)) return fail(LG_ERR_INVALID, "LG_FLAG_CHECK_FINITE needs LG_FLAG_EXT and a status array");
    if (ext && io->wire && io->wire_stride < LG_WIRE_WIDTH(n0, n1)) return fail(LG_ERR_INVALID, "wire_stride < LG_WIRE_WIDTH(n0, n1) = 3 n0 + 3 n1 + 2");
    if (ext && do_prune && ((io->prune0_f32 || io->prune1_f32))) return fail(LG_ERR_INVALID, "prune0_f32 / prune1_f32 are the outputs of a forward WITHOUT pruning");
    if (ext && !do_prune && ((io->prune0_i64 || io->prune1_i64))) return fail(LG_ERR_INVALID, "prune0_i64 / prune1_i64 are the outputs of a forward WITH pruning");
    auto write_outputs = [&](const int* final_layer, int stop_const, const int* range_flag, const int* device_err) -> hipError_t {
        OutArgs o{};
        o.B = B; o.n0 = n0; o.n1 = n1; o.L = L; o.kmax = max_matches; o.final_layer = final_layer; o.stop_const = stop_const; o.stop = io->stop;
        o.m0 = io->matches0; o.m1 = io->matches1; o.s0 = io->scores0; o.s1 = io->scores1; o.matches = io->matches; o.n_matches = io->n_matches;
        o.prune0 = io->prune0; o.prune1 = io->prune1; o.num0 = io->num0; o.num1 = io->num1;
        if (ext) {
            o.m0_64 = (long long*)io->matches0_i64; o.m1_64 = (long long*)io->matches1_i64; o.matches_64 = (long long*)io->matches_i64; o.stop_64 = (long long*)io->stop_i64;
            o.prune0_64 = (long long*)io->prune0_i64; o.prune1_64 = (long long*)io->prune1_i64; o.prune0_f = io->prune0_f32; o.prune1_f = io->prune1_f32;
            o.wire = io->wire; o.wire_stride = io->wire_stride; o.wire_prune = do_prune ? 1 : 0; o.status = io->status; o.range_flag = range_flag; o.device_err = device_err;
        }
        int span = n0 > n1 ? n0 : n1; if (2 * max_matches > span) span = 2 * max_matches; if (span < 1) span = 1;
        hipLaunchKernelGGL(write_outputs_kernel, dim3((span + 255) / 256, B), dim3(256), 0, s, o);
        return hipGetLastError();
    };

    if (n0 == 0 || n1 == 0) {  // ref :539-540, :568-588: well-formed empty result, stop = 1
        if (n0) { HIPCHK(hipMemsetAsync(io->matches0, 0xFF, sizeof(int) * (size_t)B * n0, s)); HIPCHK(hipMemsetAsync(io->scores0, 0, 4 * (size_t)B * n0, s)); }
        if (n1) { HIPCHK(hipMemsetAsync(io->matches1, 0xFF, sizeof(int) * (size_t)B * n1, s)); HIPCHK(hipMemsetAsync(io->scores1, 0, 4 * (size_t)B * n1, s)); }
        HIPCHK(hipMemsetAsync(io->n_matches, 0, sizeof(int) * (size_t)B, s));
        hipLaunchKernelGGL(init_state_kernel, dim3(64), dim3(256), 0, s, B, n0, n1, L, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           (do_prune && n0) ? io->prune0 : nullptr, (do_prune && n1) ? io->prune1 : nullptr);
        HIPCHK(write_outputs(nullptr, 1, nullptr, nullptr));   // stop = 1 for every pair (+ the extension outputs of an empty result)
        return LG_OK;
    }

    TRY(ensure_workspace(e, B, n0, n1));
    const int c0 = e->cur_cap0, c1 = e->cur_cap1, R = B * (c0 + c1);
    RowSpace rs_all{B, c0, c1, e->LEN, nullptr};
    RowSpace rs_act{B, c0, c1, e->LEN, e->ACTIVE};
    const int prec = e->cfg.precision, ap = e->attn_prec;
    const size_t es = elem_size(prec);
    auto woff = [&](const PackedW& w, size_t elems) { PackedW r; r.hi = static_cast<char*>(w.hi) + elems * es; r.lo = w.lo ? static_cast<char*>(w.lo) + elems * es : nullptr; return r; };
    int step = 0;
#define STEP_DONE() do { if (e->debug_stop >= 0 && step >= e->debug_stop) return LG_OK; ++step; } while (0)

    int* const device_err = e->CFLAGS + (size_t)2 * B * ((c0 > c1 ? c0 : c1) / compact_chunk_rows());   // error word + compaction ticket behind the chunk flags (lg_adaptive.hip)
    hipLaunchKernelGGL(init_state_kernel, dim3(64), dim3(256), 0, s, B, n0, n1, L, io->num0, io->num1, e->LEN, e->LEN_ORIG, e->LEN_OLD, e->ACTIVE, e->FINAL_LAYER,
                       do_prune ? io->prune0 : nullptr, do_prune ? io->prune1 : nullptr, e->RANGEF, device_err, e->XSEL);
    int* const range_flag = check_finite ? e->RANGEF : nullptr;
    // prep (+ descriptor copy) as its own launch, or — input_dim == 256, no debug stop — inside the first projection launch (lg_proj.hip proj_first_kernel)
    const bool fuse_prep = e->fused_prep && e->cfg.input_dim == D && e->debug_stop < 0 && e->tail_timing != 2;
    PrepArgs p{};
    p.rs = rs_all; p.n0 = n0; p.n1 = n1; p.kpts0 = io->kpts0; p.kpts1 = io->kpts1; p.size0 = io->size0; p.size1 = io->size1;
    p.scales0 = io->scales0; p.oris0 = io->oris0; p.scales1 = io->scales1; p.oris1 = io->oris1;
    p.Wr = e->Wr; p.pos_dim = 2 + 2 * (e->cfg.add_scale_ori ? 1 : 0);
    p.desc0 = io->desc0; p.desc1 = io->desc1; p.input_dim = e->cfg.input_dim;
    p.X = e->X; p.Xin = e->XIN; p.cosb = e->COS; p.sinb = e->SIN; p.ind = e->IND; p.bbox = e->BBOX;
    TRY(prof_begin(e, PC_PREP, s));
    if (fuse_prep) HIPCHK(launch_prep_bbox(p, s)); else HIPCHK(launch_prep(p, s));
    auto gemm = [&](int epi, const RowSpace& rs, const float* A, int lda, const float* A2, int lda2, int K1, int K,
                    const PackedW& W, const float* bias, int Nout, float* out, int ldo, float scale) -> GemmArgs {
        GemmArgs g{};
        g.rs = rs; g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.K1 = K1; g.K = K; g.W = W.hi; g.Wlo = W.lo; g.bias = bias; g.Nout = Nout;
        g.out = out; g.ldo = ldo; g.out_scale = scale; g.R = R; (void)epi;
        return g;
    };
    if (e->cfg.input_dim != D) {  // ref :521-522
        GemmArgs g = gemm(EPI_STORE, rs_all, e->XIN, e->cfg.input_dim, nullptr, 0, e->cfg.input_dim, e->cfg.input_dim, e->w_in, e->b_in, D, e->X, D, 1.f);
        HIPCHK(launch_gemm(prec, EPI_STORE, g, s));
    }
    TRY(prof_end(e, s));
    STEP_DONE();
    const long long qkv_plane = ap == PREC_F16X3 ? (long long)R * 256 : 0;   // split attention: hi plane, then lo plane, of q / k / v^T

    // current set of residual / rotary buffers: set 0 = X / COS / SIN, set 1 = X2 / COS2 / SIN2; flipped by every gather projection (adaptive width)
    int xcur = 0;
    float* Xs[2] = {e->X, e->X2}; float* Cs[2] = {e->COS, e->COS2}; float* Ss[2] = {e->SIN, e->SIN2};
    bool gather_pending = false;   // the decide step of the previous layer ran in gather mode: the next SelfBlock projection moves the rows
    auto make_proj = [&](int layer, int blk) {   // q/k/v projection of block `blk` of `layer` (lg_proj.hip / fused into lg_tail.hip)
        ProjArgs pj{};
        pj.rs = rs_act; pj.X = Xs[xcur]; pj.R = R; pj.q = e->Q; pj.k = e->K; pj.vt = e->VT; pj.plane = qkv_plane;
        pj.W = blk == 0 ? e->w_sqkv_p + (size_t)layer * e->sqkv_layer_bytes : e->w_cqkv_p + (size_t)layer * e->cqkv_layer_bytes;
        pj.bias = blk == 0 ? e->b_sqkv + (size_t)layer * 768 : e->b_cqkv + (size_t)layer * 512;
        pj.Nout = blk == 0 ? 768 : 512; pj.n_qk_groups = blk == 0 ? 2 : 1;
        pj.cosb = blk == 0 ? Cs[xcur] : nullptr; pj.sinb = blk == 0 ? Ss[xcur] : nullptr;
        pj.dbg = nullptr; pj.range_flag = range_flag;
        return pj;
    };
    // fused_next: a tail kernel also runs the NEXT block's projection on the x tile it has just produced.  Across a
    // layer boundary that is only valid when nothing re-orders rows in between (no early stop / pruning step).
    const bool fuse_next = e->fused_next && e->fused_tail && (e->tail_timing == 0 || e->tail_timing == 5 || e->tail_timing == 6) &&
                           e->debug_stop < 0 && launch_tail_supports_next(prec, ap);
    const bool prune_possible = do_prune && (n0 > e->cfg.pruning_min_kpts || n1 > e->cfg.pruning_min_kpts);
    // gather path: the product configuration only (the per-op / debug-stop paths keep the in-place compaction kernel, which the tests compare it with)
    const bool use_gather = e->adapt_gather && e->fused_tail && e->debug_stop < 0 && e->tail_timing == 0;
    bool proj_done = false, final_done = false;
    auto make_final = [&](const RowSpace& rs, int layer, bool per_pair) {   // final projection (ref :289-291: / d**0.25), lg_proj.hip / fused into the last tail
        FinalArgs f{};
        f.rs = rs; f.X = per_pair ? e->X : Xs[xcur]; f.R = R; f.out = e->MD; f.scale = 0.25f;
        f.planes = (prec == PREC_F16X3 && e->sim_planes) ? 1 : 0;
        if (per_pair) { f.X2 = e->X2; f.xsel = e->XSEL; }    // XSEL stays 0 unless a gather projection ran
        f.W = e->w_final_p + (per_pair ? 0 : (size_t)layer * e->final_layer_bytes); f.bias = e->b_final + (per_pair ? 0 : (size_t)layer * D);
        f.layer_of_pair = per_pair ? e->FINAL_LAYER : nullptr; f.w_layer_bytes = (long long)e->final_layer_bytes;
        return f;
    };
    for (int i = 0; i < L; ++i) {
        for (int blk = 0; blk < 2; ++blk) {  // 0 = SelfBlock (ref :159-172), 1 = CrossBlock (ref :201-230)
            if (!proj_done) {   // otherwise the previous block's tail kernel has already produced q/k/v (fused_next)
                ProjArgs pj = make_proj(i, blk);
                pj.dbg = (e->tail_timing == 2 && blk == 0) ? e->TAILDBG : nullptr;
                TRY(prof_begin(e, blk == 0 ? PC_GEMM_QKV_SELF : PC_GEMM_QKV_CROSS, s));
                if (fuse_prep && i == 0 && blk == 0) {
                    pj.rs = rs_all;                                 // rs_all: what prep covers (pairs with an empty image too)
                    if (launch_proj_first(prec, ap, pj, p, s) != hipSuccess) {
                        //  e.g. the 80 KB of dynamic LDS refused: the separate preparation kernel + the standard projection instead — bit-identical
                        (void)hipGetLastError();
                        HIPCHK(launch_prep(p, s));
                        pj.rs = rs_act;
                        HIPCHK(launch_proj(prec, ap, pj, s));
                    }
                }
                else if (gather_pending && blk == 0) {
                    // the rows of every pair that prunes move to the other buffer set on their way into the projection; pairs that do not prune at this
                    // layer are copied as they are (identity map): one buffer set is current for the whole batch
                    GatherArgs ga{};
                    ga.Xold = Xs[xcur]; ga.cos_old = Cs[xcur]; ga.sin_old = Ss[xcur]; ga.Xnew = Xs[xcur ^ 1]; ga.cos_new = Cs[xcur ^ 1]; ga.sin_new = Ss[xcur ^ 1];
                    ga.src = e->DST; ga.len_old = e->LEN_OLD;
                    HIPCHK(launch_proj_gather(prec, ap, pj, ga, s));
                    xcur ^= 1; gather_pending = false;
                }
                else HIPCHK(launch_proj(prec, ap, pj, s));
                TRY(prof_end(e, s));
            }
            proj_done = false;
            STEP_DONE();
            {
                AttnArgs at{};
                at.rs = rs_act; at.q = e->Q; at.k = e->K; at.vt = e->VT; at.plane = qkv_plane; at.ctx = e->CTX; at.R = R; at.cross = blk;
                at.dbg = (e->tail_timing == 3 && blk == 0) ? e->TAILDBG : nullptr;
                at.rows_per_wave = (e->attn_auto_rows && R / 128 * 4 < 256) ? 16 : e->attn_rows; at.dma = e->attn_dma ? 1 : 0;   // fewer 128-row workgroups than CUs: 64-row ones
                TRY(prof_begin(e, blk == 0 ? PC_ATTN_SELF : PC_ATTN_CROSS, s));
                HIPCHK(launch_attention(ap, at, s));
                TRY(prof_end(e, s));
            }
            STEP_DONE();
            if (e->fused_tail) {   // out_proj + ffn.0 + LayerNorm + GELU + ffn.3 + residual in one kernel (lg_tail.hip)
                TailArgs ta{};
                ta.rs = rs_act; ta.X = Xs[xcur]; ta.CTX = e->CTX; ta.range_flag = range_flag;
                ta.Wcat = (blk ? e->w_ctail_cat : e->w_stail_cat) + (size_t)i * e->tail_cat_layer_bytes;
                ta.bcat = (blk ? e->b_ccat : e->b_scat) + (size_t)i * 512;
                ta.gamma = (blk ? e->ln_c_g : e->ln_s_g) + (size_t)i * 512; ta.beta = (blk ? e->ln_c_b : e->ln_s_b) + (size_t)i * 512;
                ta.W2 = (blk ? e->w_ctail_2 : e->w_stail_2) + (size_t)i * e->tail_2_layer_bytes;
                ta.b2 = (blk ? e->b_cf2 : e->b_sf2) + (size_t)i * D;
                if (blk == 1) {   // 256 -> 1 heads on the rows this CrossBlock tail produces
                    // token confidence of layer i (ref :548) for the stop decision; matchability of layer i (ref :298-299): its sigmoid for
                    // the pruning mask (ref :553) and its log-sigmoids as the assignment's matchability terms (ref :268-276) wherever a pair
                    // may END at this layer — the last one, or any one with early stopping
                    const bool last = i + 1 == L;
                    const bool prune_here = !last && do_prune && prune_possible;
                    const bool want_tok = !last && do_stop, want_ls = last || do_stop;
                    if (want_tok) { ta.head_w0 = e->w_tok + (size_t)i * D; ta.head_b0 = e->b_tok + i; ta.head_out0 = e->CONF; }
                    if (prune_here || want_ls) {
                        float* sig = prune_here ? e->MSCORE : nullptr; float* ls = want_ls ? e->LS : nullptr;
                        float* lsneg = (want_ls && io->log_assignment) ? e->LSNEG : nullptr;
                        if (want_tok) { ta.head_w1 = e->w_match + (size_t)i * D; ta.head_b1 = e->b_match + i; ta.head_out1 = sig; ta.head_ls1 = ls; ta.head_lsneg1 = lsneg; }
                        else { ta.head_w0 = e->w_match + (size_t)i * D; ta.head_b0 = e->b_match + i; ta.head_out0 = sig; ta.head_ls0 = ls; ta.head_lsneg0 = lsneg; }
                    }
                }
                ta.dbg = (e->tail_timing == 1 || (i == 0 && ((e->tail_timing == 5 && blk == 1) || (e->tail_timing == 6 && blk == 0)))) ? e->TAILDBG : nullptr;
                ta.row_tiles = e->tail_row_tiles ? e->tail_row_tiles : tail_row_tiles_for(R);
                // Across a layer boundary the fusion is valid whenever no row can MOVE in between: early stop alone only
                // deactivates a pair (its speculative projection is never read), pruning re-orders rows — but it cannot
                // happen while every segment is at or below the pruning threshold (ref :551 / :559; lengths only shrink).
                if (fuse_next && (blk == 0 || (i + 1 < L && !prune_possible))) {
                    ta.next = blk == 0 ? make_proj(i, 1) : make_proj(i + 1, 0);
                    if (ta.dbg && e->tail_timing >= 5) ta.next.dbg = e->TAILDBG2;
                    proj_done = true;
                } else if (fuse_next && blk == 1 && i + 1 == L && !do_stop) {
                    // fixed depth: every live pair ends here, so the LAST tail also runs the final projection of the log assignment on its x tile
                    ta.fin = make_final(rs_act, L - 1, false);
                    final_done = true;
                }
                TRY(prof_begin(e, PC_TAIL, s));
                HIPCHK(launch_tail(prec, ap, ta, s));
                TRY(prof_end(e, s));
                STEP_DONE(); STEP_DONE(); STEP_DONE(); STEP_DONE();
                continue;
            }
            {
                GemmArgs g = gemm(EPI_STORE, rs_act, e->CTX, D, nullptr, 0, D, D, woff(blk ? e->w_cout : e->w_sout, (size_t)i * D * D),
                                  (blk ? e->b_cout : e->b_sout) + (size_t)i * D, D, e->MSG, D, 1.f);
                TRY(prof_begin(e, PC_GEMM_OUT, s));
                HIPCHK(launch_gemm(prec, EPI_STORE, g, s));
                TRY(prof_end(e, s));
            }
            STEP_DONE();
            {
                GemmArgs g = gemm(EPI_STORE, rs_act, e->X, D, e->MSG, D, D, 512, woff(blk ? e->w_cf1 : e->w_sf1, (size_t)i * 512 * 512),
                                  (blk ? e->b_cf1 : e->b_sf1) + (size_t)i * 512, 512, e->H1, 512, 1.f);
                TRY(prof_begin(e, PC_GEMM_FFN1, s));
                HIPCHK(launch_gemm(prec, EPI_STORE, g, s));
                TRY(prof_end(e, s));
            }
            STEP_DONE();
            {
                LnGeluArgs ln{rs_act, e->H1, e->G, (blk ? e->ln_c_g : e->ln_s_g) + (size_t)i * 512, (blk ? e->ln_c_b : e->ln_s_b) + (size_t)i * 512, R};
                TRY(prof_begin(e, PC_LN_GELU, s));
                HIPCHK(launch_ln_gelu(ln, s));
                TRY(prof_end(e, s));
            }
            STEP_DONE();
            {
                GemmArgs g = gemm(EPI_RESID, rs_act, e->G, 512, nullptr, 0, 512, 512, woff(blk ? e->w_cf2 : e->w_sf2, (size_t)i * D * 512),
                                  (blk ? e->b_cf2 : e->b_sf2) + (size_t)i * D, D, e->X, D, 1.f);
                TRY(prof_begin(e, PC_GEMM_FFN2, s));
                HIPCHK(launch_gemm(prec, EPI_RESID, g, s));
                TRY(prof_end(e, s));
            }
            STEP_DONE();
        }
        if (i == L - 1) break;  // ref :544-545
        const bool prune_now = do_prune && prune_possible;   // below the threshold the pruning branch is never taken (ref :551 / :559)
        if (do_stop || prune_now) {
            if (!e->fused_tail) {   // per-op path: the 256 -> 1 heads as their own pass (the fused CrossBlock tail computes them in its epilogue)
                RowDotArgs rd{};
                rd.rs = rs_act; rd.X = e->X;
                if (do_stop && prune_now) {
                    rd.w0 = e->w_tok + (size_t)i * D; rd.b0 = e->b_tok + i; rd.out0 = e->CONF; rd.act0 = 1;
                    rd.w1 = e->w_match + (size_t)i * D; rd.b1 = e->b_match + i; rd.out1 = e->MSCORE; rd.act1 = 1;
                } else if (do_stop) {
                    rd.w0 = e->w_tok + (size_t)i * D; rd.b0 = e->b_tok + i; rd.out0 = e->CONF; rd.act0 = 1;
                } else {
                    rd.w0 = e->w_match + (size_t)i * D; rd.b0 = e->b_match + i; rd.out0 = e->MSCORE; rd.act0 = 1;
                }
                TRY(prof_begin(e, PC_ROWDOT, s));
                HIPCHK(launch_rowdot(rd, s));
                TRY(prof_end(e, s));
            }
            AdaptArgs ad{};
            ad.rs = rs_act; ad.len = e->LEN; ad.active = e->ACTIVE; ad.len_old = e->LEN_OLD; ad.final_layer = e->FINAL_LAYER;
            ad.ind = e->IND; ad.dst = e->DST; ad.prune0 = io->prune0; ad.prune1 = io->prune1; ad.n0 = n0; ad.n1 = n1; ad.len_orig = e->LEN_ORIG;
            ad.conf = e->CONF; ad.mscore = e->MSCORE; ad.X = e->X; ad.cosb = e->COS; ad.sinb = e->SIN;
            if (use_gather && prune_now) { ad.gather = 1; ad.src = e->DST; ad.xsel = e->XSEL; ad.xnext = xcur ^ 1; gather_pending = true; }
            ad.layer = i;
            // ref :631-634 threshold (float32 buffer), :656 and :640 compare in float32
            ad.conf_thr = (float)std::fmin(std::fmax(0.8 + 0.1 * std::exp(-4.0 * i / L), 0.0), 1.0);
            ad.depth_conf = (float)e->cfg.depth_confidence;
            ad.width_conf = (float)(1.0 - e->cfg.width_confidence);
            ad.pruning_min_kpts = e->cfg.pruning_min_kpts;
            ad.do_stop = do_stop; ad.do_prune = prune_now;
            ad.compact_chunks = (c0 > c1 ? c0 : c1) / compact_chunk_rows(); ad.compact_flags = e->CFLAGS; ad.compact_err = device_err; ad.compact_ticket = device_err + 1;
            if (prune_now && !e->cflags_clean) {   // fresh carve: whatever the arena held there must not look like an epoch
                HIPCHK(hipMemsetAsync(e->CFLAGS, 0, (size_t)2 * B * ad.compact_chunks * 4 + 256, s));
                e->cflags_clean = true;
            }
            ad.compact_epoch = ++e->compact_epoch;
            TRY(prof_begin(e, PC_ADAPTIVE, s));
            HIPCHK(launch_adapt(ad, s));
            TRY(prof_end(e, s));
        }
    }
    // ---- log assignment with the weights of the layer each pair stopped at (ref :591)
    {
        if (!e->fused_tail) {   // per-op path: matchability terms as their own pass (the fused CrossBlock tails emit them from their heads)
            RowDotArgs rd{};
            rd.rs = rs_all; rd.X = e->X; rd.w0 = e->w_match; rd.b0 = e->b_match; rd.out0 = e->LS; rd.act0 = 2;
            rd.layer_of_pair = e->FINAL_LAYER; rd.w_layer_stride = D; rd.ignore_active = 1;
            if (io->log_assignment) { rd.w1 = e->w_match; rd.b1 = e->b_match; rd.out1 = e->LSNEG; rd.act1 = 3; }  // dustbin terms
            TRY(prof_begin(e, PC_ROWDOT, s));
            HIPCHK(launch_rowdot(rd, s));
            TRY(prof_end(e, s));
        }
        if (!final_done) {
            TRY(prof_begin(e, PC_GEMM_FINAL, s));
            HIPCHK(launch_final_proj(prec, make_final(rs_all, 0, true), s));
            TRY(prof_end(e, s));
        }
        TRY(prof_begin(e, PC_SIM, s));
        if (prec == PREC_F16X3 && e->sim_planes) {
            SimPlanesArgs sp{rs_all, reinterpret_cast<const f16_t*>(e->MD), (long long)R * 256, D, e->SIM, e->sim_chunk};
            HIPCHK(launch_sim_planes(sp, s));
        } else {
            SimArgs sm{rs_all, e->MD, D, D, e->SIM};
            HIPCHK(launch_sim(prec, sm, s));
        }
        TRY(prof_end(e, s));
        AssignArgs as{};
        as.rs = rs_all; as.sim = e->SIM; as.ls = e->LS; as.lse_r = e->LSE_R; as.lse_c = e->LSE_C; as.max0 = e->MAX0; as.arg0 = e->ARG0;
        as.max1 = e->MAX1; as.arg1 = e->ARG1; as.cpm = e->CPM; as.cps = e->CPS; as.cbv = e->CBV; as.cbi = e->CBI; as.ind = e->IND; as.n0 = n0; as.n1 = n1; as.filter_threshold = (float)e->cfg.filter_threshold;
        as.m0 = io->matches0; as.m1 = io->matches1; as.s0 = io->scores0; as.s1 = io->scores1;
        as.matches = io->matches; as.mscores = io->match_scores; as.n_matches = io->n_matches; as.max_matches = max_matches;
        as.log_assignment = io->log_assignment; as.lsneg = e->LSNEG;
        as.dbg = e->tail_timing == 4 ? e->TAILDBG : nullptr;
        as.all_rows_live = (!do_prune && !io->num0 && !io->num1) ? 1 : 0;
        TRY(prof_begin(e, PC_ASSIGN, s));
        HIPCHK(launch_assign(as, s));
        TRY(prof_end(e, s));
        HIPCHK(write_outputs(e->FINAL_LAYER, 0, range_flag, device_err));
    }
#undef STEP_DONE
#undef TRY
    return LG_OK;
}

int lg_unpack_wire(const lg_unpack_io* io, void* hip_stream) {
    if (!io || io->rows < 0 || io->n0 < 0 || io->n1 < 0 || io->pairs_out < 0) return fail(LG_ERR_INVALID, "lg_unpack_wire: bad argument");
    if (io->rows == 0 || io->pairs_out == 0) return LG_OK;    // an empty gather (world of one, empty batch) is a no-op
    if (!io->wire || io->wire_stride < LG_WIRE_WIDTH(io->n0, io->n1)) return fail(LG_ERR_INVALID, "lg_unpack_wire: null wire or wire_stride < LG_WIRE_WIDTH(n0, n1)");
    if (io->n0 > LG_MAX_KEYPOINTS || io->n1 > LG_MAX_KEYPOINTS) return fail(LG_ERR_INVALID, "lg_unpack_wire: more than LG_MAX_KEYPOINTS keypoints");
    UnpackArgs a{};
    a.wire = io->wire; a.stride = io->wire_stride; a.n0 = io->n0; a.n1 = io->n1; a.with_prune = io->with_prune; a.pairs_out = io->pairs_out;
    a.kmax = io->n0 < io->n1 ? io->n0 : io->n1; a.order = io->order;
    a.m0 = (long long*)io->matches0; a.m1 = (long long*)io->matches1; a.stop = (long long*)io->stop; a.s0 = io->scores0; a.s1 = io->scores1;
    if (io->with_prune) { a.p0_64 = (long long*)io->prune0_i64; a.p1_64 = (long long*)io->prune1_i64; }
    else { a.p0_f = io->prune0_f32; a.p1_f = io->prune1_f32; }
    a.matches = (long long*)io->matches; a.mscores = io->match_scores; a.info = io->info;
    hipLaunchKernelGGL(unpack_wire_kernel, dim3(io->rows), dim3(1024), 0, static_cast<hipStream_t>(hip_stream), a);
    HIPCHK(hipGetLastError());
    return LG_OK;
}

}  // extern "C"
