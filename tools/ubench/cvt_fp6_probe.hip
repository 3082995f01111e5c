// Probe for a later round: can the fp6 operand of the hi plane be DERIVED in registers from the f16 fragments
// the main product loads anyway?  Needs two facts about v_cvt_scalef32_pk32_fp6_f16 that no document in this image states:
//   (1) the element order: does input element i (dword i/2, half i%2) land in fp6 slot i (bits [6i, 6i+6) of the 192-bit result)?
//   (2) the scale operand: is the result fp6(x / scale), fp6(x * scale), or scaled by the exponent of `scale` only?
// Method: convert 32 known f16 values with scale 1, 2, 0.5 and 8; decode the 192 bits on the host by the e2m3 table (slot i at bits
// [6i, 6i+6) — the layout the scaled MFMA was CONFIRMED to read for the 2xpk16_fp6_f32 conversion, tools/ubench/mfma_mx_probe.hip);
// also convert back on the device with v_cvt_scalef32_pk32_f32_fp6 and feed the result to one scaled MFMA against an identity-like
// operand, so that the MFMA's view of the order is checked too.  Prints everything; a human (or the next session) reads it.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 cvt_fp6_probe.hip -o cvt_fp6_probe
// RESULT on MI355X (profiles/r02f_cvt_fp6_probe.md): (1) pk32_fp6_f16 writes input element i to slot i (natural order) and the
// scaled MFMA reads slot i as k = i of the lane's block; (2) the result is fp6(x / scale), round to nearest even, saturating at 7.5;
// (3) 2xpk16_fp6_f32(a0, a1) INTERLEAVES its two sources: slot 2i = a0[i], slot 2i + 1 = a1[i] (harmless in ffn0_f16_fp6.hip, where
// both operands go through it); (4) with the scale register holding 127 in byte 0 only, the 4 k-blocks summed to 1x the expected
// value instead of 4x — consistent with k-block g of a lane taking scale byte g when op_sel is 0; keep replicating the E8M0 byte
// into all four bytes as ffn0_f16_fp6.hip does (that run of the probe used a lone byte; the source now replicates it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// out[s][0..5] = packed result for scale s; back[s][0..31] = device-side decode; ref6[0..5] = the 2xpk16_fp6_f32 conversion of the same values
__global__ void probe(const float* in, const float* scales, int ns, unsigned* out, float* back, unsigned* ref6, float* mm) {
    if (threadIdx.x >= 64) return;
    v32h h; v16f a0, a1;
    for (int i = 0; i < 32; ++i) { h[i] = (_Float16)in[i]; if (i < 16) a0[i] = in[i]; else a1[i - 16] = in[i]; }
    for (int s = 0; s < ns; ++s) {
        const v6u r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scales[s]);
        const v32f b = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r, 1.0f);
        if (threadIdx.x == 0) { for (int i = 0; i < 6; ++i) out[s * 6 + i] = r[i]; for (int i = 0; i < 32; ++i) back[s * 32 + i] = b[i]; }
    }
    const v6u r2 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, 1.0f);
    if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) ref6[i] = r2[i];
    // MFMA view: A = the pk32-converted row (every lane the same 32 values -> every row of A identical in each k-block),
    // B = one-hot: lane (lr, g) supplies column lr with a single 1.0 at k = 32 g + lr  (lr < 16) -> C[row][lr] = sum_g A[row][32 g + lr]
    const v6u ra = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, 1.0f);
    v16f o0, o1;
    const int lr = threadIdx.x & 15;
    for (int i = 0; i < 16; ++i) { o0[i] = (i == lr) ? 1.0f : 0.0f; o1[i] = 0.0f; }
    const v6u rb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(o0, o1, 1.0f);
    const v8i A = {(int)ra[0], (int)ra[1], (int)ra[2], (int)ra[3], (int)ra[4], (int)ra[5], 0, 0};
    const v8i B = {(int)rb[0], (int)rb[1], (int)rb[2], (int)rb[3], (int)rb[4], (int)rb[5], 0, 0};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);   // E8M0 127 = 2^0 in ALL FOUR bytes (see the header: a lone byte 0 gave 1x instead of 4x)
    if (threadIdx.x < 16) mm[threadIdx.x] = acc[0];   // C[row 0][col lr] (lanes 0..15 hold rows 0..3 of column lr)
}

static float e2m3(unsigned c) {   // sign, 2 exponent bits (bias 1), 3 mantissa bits; subnormal step 0.125
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m * 0.125f : ldexpf(1.0f + m * 0.125f, e - 1);
    return s ? -v : v;
}

int main() {
    float in[32];
    // 32 DISTINCT exactly representable e2m3 magnitudes / signs so that order and scale can be read off unambiguously
    const float mags[16] = {0.125f, 0.25f, 0.375f, 0.5f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f, 2.0f, 2.5f, 3.0f, 3.5f, 4.0f, 5.0f, 6.0f};
    for (int i = 0; i < 32; ++i) in[i] = (i < 16 ? 1.f : -1.f) * mags[i & 15];
    const float scales[4] = {1.0f, 2.0f, 0.5f, 8.0f};
    float *din, *dsc, *dback, *dmm; unsigned *dout, *dref;
    CHK(hipMalloc(&din, sizeof in)); CHK(hipMalloc(&dsc, sizeof scales)); CHK(hipMalloc(&dout, 4 * 6 * 4)); CHK(hipMalloc(&dback, 4 * 32 * 4));
    CHK(hipMalloc(&dref, 6 * 4)); CHK(hipMalloc(&dmm, 16 * 4));
    CHK(hipMemcpy(din, in, sizeof in, hipMemcpyHostToDevice)); CHK(hipMemcpy(dsc, scales, sizeof scales, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dsc, 4, dout, dback, dref, dmm);
    CHK(hipDeviceSynchronize());
    unsigned out[24], ref6[6]; float back[128], mm[16];
    CHK(hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost)); CHK(hipMemcpy(back, dback, sizeof back, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(ref6, dref, sizeof ref6, hipMemcpyDeviceToHost)); CHK(hipMemcpy(mm, dmm, sizeof mm, hipMemcpyDeviceToHost));
    printf("input      :"); for (int i = 0; i < 32; ++i) printf(" %g", in[i]); printf("\n");
    auto slot = [](const unsigned* r, int i) { const int bit = 6 * i; unsigned long long w = r[bit >> 5]; if ((bit >> 5) + 1 < 6) w |= (unsigned long long)r[(bit >> 5) + 1] << 32; return (unsigned)((w >> (bit & 31)) & 63); };
    printf("2xpk16_f32 (scale 1), host decode:"); for (int i = 0; i < 32; ++i) printf(" %g", e2m3(slot(ref6, i))); printf("\n");
    for (int s = 0; s < 4; ++s) {
        printf("pk32_f16 scale %-4g host decode :", scales[s]); for (int i = 0; i < 32; ++i) printf(" %g", e2m3(slot(out + 6 * s, i))); printf("\n");
        printf("pk32_f16 scale %-4g device back :", scales[s]); for (int i = 0; i < 32; ++i) printf(" %g", back[32 * s + i]); printf("\n");
    }
    int same_order = 1; for (int i = 0; i < 6; ++i) same_order &= (out[i] == ref6[i]);
    printf("pk32_fp6_f16(scale 1) bit pattern == 2xpk16_fp6_f32(scale 1) bit pattern: %s\n", same_order ? "YES (same element order as the confirmed conversion)" : "NO");
    printf("MFMA view: C[0][c] = sum_g A[0][32 g + c], with every k-block holding the same 32 values = 4 * in[c]:");
    for (int c = 0; c < 16; ++c) printf(" %g", mm[c]); printf("\n   expected                                                                          :");
    for (int c = 0; c < 16; ++c) printf(" %g", 4 * in[c]); printf("\n");
    return 0;
}
