// standalone check of the permlane-swap reductions against __shfl_xor (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/permlane_check tools/permlane_check.hip && /tmp/permlane_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../lightglue_amd/csrc/lg_common.h"
__global__ void k(const float* in, float* out) {
    const float x = in[threadIdx.x];
    out[threadIdx.x] = lg::xor16_max(x);
    out[64 + threadIdx.x] = fmaxf(x, __shfl_xor(x, 16, 64));
    out[128 + threadIdx.x] = lg::xor32_max(x);
    out[192 + threadIdx.x] = fmaxf(x, __shfl_xor(x, 32, 64));
    out[256 + threadIdx.x] = lg::xor32_sum(lg::xor16_sum(x));
    float l = x; l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
    out[320 + threadIdx.x] = l;
}
int main() {
    float h[64], o[384]; for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64) + 0.25f * i;
    float *d, *e; hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 3; ++t) for (int i = 0; i < 64; ++i) if (o[128 * t + i] != o[128 * t + 64 + i]) { if (bad < 8) printf("mismatch test %d lane %d: %f vs %f\n", t, i, o[128 * t + i], o[128 * t + 64 + i]); ++bad; }
    printf("permlane check: %d mismatches\n", bad);
    return bad != 0;
}
