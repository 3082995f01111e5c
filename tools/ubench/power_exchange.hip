// Exchange rates at the board's power cap (round 6: both big kernels are power-limited, so a launch costs its ENERGY).
// What does a TB/s of LDS fragment reads, of L2 -> VGPR loads, of L2 -> LDS DMA, or a stream of exponentials COST in matrix throughput on this board?
// One 16-wave workgroup per CU: waves 0-7 (two per SIMD) spin on dense f16 16x16x32 MFMAs (own A, shared B: the kernels' operand pattern) — alone they put the board at
// its cap —, waves 8-15 run the OTHER activity at a duty cycle set by s_sleep.  Every wave runs until the same wall-clock deadline and reports its iteration count, so
// both rates are measured over the same interval while a host thread samples the board power sensor (hwmon power1_input) and sclk.
// Reading: at the cap, d(MFMA TFLOP/s) / d(other rate) is the exchange rate; x (board power - idle) / (MFMA rate alone) gives joules per unit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize power_exchange.hip -o power_exchange -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { O_NONE = 0, O_LDS = 1, O_L2 = 2, O_DMA = 3, O_EXP = 4, O_HBM = 5 };
constexpr int REGION = 96 * 1024;          // bytes of global memory a workgroup cycles through for O_L2 / O_DMA: beyond the 32 KB L1, 32 of them inside an XCD's 4 MB L2

__device__ __forceinline__ long long wall() { return (long long)__builtin_amdgcn_s_memrealtime(); }   // 100 MHz
__device__ __forceinline__ u32x4 rnd(unsigned h, unsigned k) {
    u32x4 r;
    for (int i = 0; i < 4; ++i) { h = h * 1664525u + 1013904223u + k; r[i] = (h & 0x83ff83ffu) | 0x20002000u | ((h >> 7) & 0x1c001c00u); }
    return r;
}
#define MMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0)

template <int OTHER>
__global__ __launch_bounds__(1024) void mix(const char* __restrict__ gmem, long long gbytes, float* sink, unsigned long long* counts, long long ticks, int sleep_on) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 1024) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const long long t_end = wall() + ticks;
    unsigned long long n = 0;
    float keep = 0.f;
    if (wave < 8) {
        const unsigned h = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
        u32x4 a[8], b[3];
        for (int i = 0; i < 8; ++i) a[i] = rnd(h, 2 * i + 1);
        for (int i = 0; i < 3; ++i) b[i] = rnd(h, 2 * i + 2);
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        do {
            for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) MMA(acc[i], a[i], b[r]);
            }
            n += 16 * 24;
        } while (wall() < t_end);
        for (int i = 0; i < 8; ++i) keep += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if constexpr (OTHER != O_NONE) {
        const int ow = wave - 8;
        if constexpr (OTHER == O_LDS) {                       // 16 ds_read_b128 per iteration: lane-linear 1 KB rows (conflict-free), 16 KB per wave and iteration
            const char* base = smem + lane * 16;
            u32x4 x = {0, 0, 0, 0};
            do {
#pragma unroll
                for (int k = 0; k < 16; ++k) { const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((k + 2 * ow) & 63) * 1024); x[0] ^= v[0]; x[1] ^= v[1]; x[2] ^= v[2]; x[3] ^= v[3]; }
                n += 16 * 1024;
                if (sleep_on) __builtin_amdgcn_s_sleep(8);
            } while (wall() < t_end);
            keep = (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
        } else if constexpr (OTHER == O_L2 || OTHER == O_HBM) {  // 8 global_load_dwordx4 per iteration: 8 KB per wave and iteration
            const long long span = OTHER == O_L2 ? REGION : gbytes / gridDim.x;
            const char* base = gmem + (long long)blockIdx.x * span;
            long long off = (long long)ow * 8192;
            u32x4 x = {0, 0, 0, 0};
            do {
#pragma unroll
                for (int k = 0; k < 8; ++k) { long long o = off + k * 1024; if (o >= span) o -= span; const u32x4 v = *reinterpret_cast<const u32x4*>(base + o + lane * 16); x[0] ^= v[0]; x[1] ^= v[1]; x[2] ^= v[2]; x[3] ^= v[3]; }
                off += 8 * 8192; if (off >= span) off -= span;
                n += 8 * 1024;
                if (sleep_on) __builtin_amdgcn_s_sleep(8);
            } while (wall() < t_end);
            keep = (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
        } else if constexpr (OTHER == O_DMA) {                // 8 global_load_lds_dwordx4 per iteration into this wave's own 8 KB of LDS
            const char* base = gmem + (long long)blockIdx.x * REGION;
            long long off = (long long)ow * 8192;
            do {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    long long o = off + k * 1024; if (o >= REGION) o -= REGION;
                    const void* g = base + o + lane * 16;
                    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + 65536 + ow * 8192 + k * 1024));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(g) : "memory", "m0");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                off += 8 * 8192; if (off >= REGION) off -= REGION;
                n += 8 * 1024;
                if (sleep_on) __builtin_amdgcn_s_sleep(8);
            } while (wall() < t_end);
        } else if constexpr (OTHER == O_EXP) {                // a softmax-like block: 16 v_exp_f32 + 32 fma per iteration
            float v[16];
            for (int i = 0; i < 16; ++i) v[i] = -0.001f * (lane + i);
            do {
#pragma unroll
                for (int i = 0; i < 16; ++i) { v[i] = __builtin_amdgcn_exp2f(v[i]); v[i] = __builtin_fmaf(v[i], -0.5f, -0.25f); v[i] = __builtin_fmaf(v[i], 0.999f, -0.001f); }
                n += 16;
                if (sleep_on) __builtin_amdgcn_s_sleep(8);
            } while (wall() < t_end);
            for (int i = 0; i < 16; ++i) keep += v[i];
        }
    }
    if (keep == 12345.678f) sink[0] = keep;
    if (lane == 0) counts[blockIdx.x * 16 + wave] = n;
}

struct Sampler {
    std::string power, freq; std::atomic<bool> stop{false}; std::vector<double> w, f; std::thread th;
    static long long rd(const std::string& p) { FILE* fp = fopen(p.c_str(), "r"); if (!fp) return -1; long long v = -1; if (fscanf(fp, "%lld", &v) != 1) v = -1; fclose(fp); return v; }
    void start() { stop = false; w.clear(); f.clear(); th = std::thread([this] { while (!stop) { const long long p = rd(power), q = rd(freq); if (p > 0) w.push_back(p / 1e6); if (q > 0) f.push_back(q / 1e6); std::this_thread::sleep_for(std::chrono::milliseconds(40)); } }); }
    void finish(double& watts, double& mhz) { stop = true; th.join(); auto med = [](std::vector<double> v, size_t skip) { if (v.size() <= skip) return 0.0; v.erase(v.begin(), v.begin() + skip); std::sort(v.begin(), v.end()); return v[v.size() / 2]; }; watts = med(w, w.size() / 3); mhz = med(f, f.size() / 3); }
};

template <int OTHER> void run(const char* name, const char* unit, double unit_scale, Sampler& smp, const char* gmem, long long gbytes, float* sink, unsigned long long* counts, double seconds, int sleep_on) {
    const long long ticks = (long long)(seconds * 1e8);
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(mix<OTHER>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CHK(hipMemset(counts, 0, 256 * 16 * 8));
    smp.start();
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(mix<OTHER>, dim3(256), dim3(1024), 131072, 0, gmem, gbytes, sink, counts, ticks, sleep_on);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double watts, mhz; smp.finish(watts, mhz);
    std::vector<unsigned long long> h(256 * 16); CHK(hipMemcpy(h.data(), counts, h.size() * 8, hipMemcpyDeviceToHost));
    double mf = 0, ot = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) (w < 8 ? mf : ot) += (double)h[b * 16 + w];
    const double t = seconds;   // every wave ran until the same deadline
    printf("%-34s %s  MFMA %7.1f TFLOP/s   other %9.2f %-8s  board %6.1f W  sclk %6.0f MHz  (kernel %.0f ms)\n", name, sleep_on ? "half duty" : "full duty", mf * 16384.0 / t / 1e12, ot * unit_scale / t, unit, watts, mhz, ms);
    fflush(stdout);
}

int main() {
    char bus[64]; CHK(hipDeviceGetPCIBusId(bus, 64, 0));
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    Sampler smp;
    glob_t g; const std::string pat = std::string("/sys/bus/pci/devices/") + bus + "/hwmon/hwmon*";
    if (glob(pat.c_str(), 0, nullptr, &g) == 0 && g.gl_pathc > 0) { smp.power = std::string(g.gl_pathv[0]) + "/power1_input"; smp.freq = std::string(g.gl_pathv[0]) + "/freq1_input"; }
    printf("power sensor: %s (cap %lld W)\n", smp.power.c_str(), smp.power.empty() ? -1 : Sampler::rd(smp.power.substr(0, smp.power.size() - 12) + "power1_cap") / 1000000);
    const long long gbytes = 4LL << 30;
    char* gmem; float* sink; unsigned long long* counts;
    CHK(hipMalloc(&gmem, gbytes)); CHK(hipMemset(gmem, 0x3c, gbytes)); CHK(hipMalloc(&sink, 64)); CHK(hipMalloc(&counts, 256 * 16 * 8));
    { double w, f; smp.start(); std::this_thread::sleep_for(std::chrono::milliseconds(1200)); smp.finish(w, f); printf("idle (no kernel): board %6.1f W  sclk %6.0f MHz\n", w, f); }
    const double S = 1.5;
    for (int rep = 0; rep < 2; ++rep) {
        run<O_NONE>("MFMA alone", "-", 0.0, smp, gmem, gbytes, sink, counts, S, 0);
        for (int sl = 1; sl >= 0; --sl) {
            run<O_LDS>("MFMA + LDS reads (ds_read_b128)", "TB/s", 1e-12, smp, gmem, gbytes, sink, counts, S, sl);
            run<O_L2>("MFMA + L2 -> VGPR loads", "TB/s", 1e-12, smp, gmem, gbytes, sink, counts, S, sl);
            run<O_DMA>("MFMA + L2 -> LDS DMA", "TB/s", 1e-12, smp, gmem, gbytes, sink, counts, S, sl);
            run<O_EXP>("MFMA + exp2 / fma VALU", "Texp/s", 64e-12, smp, gmem, gbytes, sink, counts, S, sl);
            run<O_HBM>("MFMA + HBM stream", "TB/s", 1e-12, smp, gmem, gbytes, sink, counts, S, sl);
        }
    }
    return 0;
}
