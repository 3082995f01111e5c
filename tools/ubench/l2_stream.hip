// Microbenchmark (experiment): per-CU L2 -> VGPR streaming rate of 1 KB coalesced fragment loads (the fused tail's weight stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
// every workgroup streams the SAME `bytes`-sized array `passes` times; wave w reads fragments w, w+NW, ... (1 KB each)
template <int DEPTH, int NT>
__global__ __launch_bounds__(NT) void stream(const char* __restrict__ W, int frags, int passes, int rot, unsigned* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = NT / 64;
    unsigned acc = 0;
    const int start = rot ? (blockIdx.x * 37) % frags : 0;   // rot: workgroups start at different places of the array
    for (int p = 0; p < passes; ++p) {
        for (int f0 = w; f0 < frags; f0 += NW * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                int f = f0 + d * NW + start; if (f >= frags) f -= frags; if (f >= frags) f -= frags;
                v[d] = *reinterpret_cast<const u32x4*>(W + (long long)f * 1024 + lane * 16);
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int DEPTH, int NT> void run(const char* dW, size_t bytes, int blocks, int rot, unsigned* sink) {
    const int frags = (int)(bytes / 1024), passes = (int)(((size_t)64 << 20) / bytes);
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((stream<DEPTH, NT>), dim3(blocks), dim3(NT), 0, 0, dW, frags, passes, rot, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL((stream<DEPTH, NT>), dim3(blocks), dim3(NT), 0, 0, dW, frags, passes, rot, sink);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    const double tot = (double)blocks * passes * bytes;
    printf("  array %5zu KB  depth %2d  waves %2d  blocks %4d rot %d: %.3f ms  %.1f TB/s  = %.1f B/clk/CU @2.4GHz (256 CUs)\n", bytes >> 10, DEPTH, NT / 64, blocks, rot, ms, tot / ms / 1e9,
           tot / ms / 1e-3 / 256 / 2.4e9);
}
int main() {
    char* dW; unsigned* sink;
    CHK(hipMalloc(&dW, 64 << 20)); CHK(hipMemset(dW, 1, 64 << 20)); CHK(hipMalloc(&sink, 4));
    for (size_t kb : {16, 1024, 2304, 8192}) {
        for (int rot : {0, 1}) {
            run<8, 512>(dW, kb << 10, 256, rot, sink);
            run<16, 512>(dW, kb << 10, 256, rot, sink);
            run<8, 1024>(dW, kb << 10, 256, rot, sink);
            run<8, 256>(dW, kb << 10, 512, rot, sink);
        }
    }
    return 0;
}
