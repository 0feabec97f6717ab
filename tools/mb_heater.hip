// Microbenchmark: does a low-priority "keep-warm" kernel on a few CUs keep the chip in its fast clock state while the
// main stream runs long phases of tiny kernels between short MFMA bursts (the training step's pattern)?
//   graph A = 6 "big" launches (MFMA loop on every CU, ~200 us each) + N tiny kernels; graph B = the N tiny kernels.
//   (A - B) / 6 = time of a big launch inside the pattern, with a heater of G workgroups running beside it or not.
// gfx950.  build: hipcc --offload-arch=gfx950 -O3 tools/mb_heater.hip -o tools/_mb_heater
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// iters x 16 MFMAs per wave on 4 independent accumulators; `nap` > 0 inserts s_sleep between bursts (heater intensity)
__global__ __launch_bounds__(256) void k_mfma(int iters, int nap, float* sink) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        if (nap) __builtin_amdgcn_s_sleep(127);
    }
    const float s = c0[0] + c1[1] + c2[2] + c3[3];
    if (s == 12345.678f) sink[threadIdx.x] = s;      // never true: keeps the loop alive
}
__global__ void k_tiny(float* p) { p[threadIdx.x] += 1.0f; }

int main(int argc, char** argv) {
    const int n_tiny = argc > 1 ? atoi(argv[1]) : 3000;
    float* sink; CK(hipMalloc(&sink, 1 << 16)); CK(hipMemset(sink, 0, 1 << 16));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));      // lo = least priority (numerically largest)
    hipStream_t s1, s2;
    CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
    // size the big kernel to ~200 us at full clock: 1024 workgroups (4 per CU) x 4 waves; calibrate iters
    const int BIG_WG = 1024;
    int big_iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_big = [&](int n) {
        hipEventRecord(e0, s1);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_mfma, BIG_WG, 256, 0, s1, big_iters, 0, sink);
        hipEventRecord(e1, s1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / n;
    };
    time_big(300);
    float t = time_big(300);
    big_iters = (int)(big_iters * 200.0f / t);
    printf("big kernel: %d iters, sustained back-to-back %.1f us\n", big_iters, time_big(300));
    hipGraph_t gA, gB; hipGraphExec_t xA, xB;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(k_mfma, BIG_WG, 256, 0, s1, big_iters, 0, sink);
    for (int i = 0; i < n_tiny; ++i) hipLaunchKernelGGL(k_tiny, 1, 64, 0, s1, sink + 4096);
    CK(hipStreamEndCapture(s1, &gA)); CK(hipGraphInstantiate(&xA, gA, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_tiny; ++i) hipLaunchKernelGGL(k_tiny, 1, 64, 0, s1, sink + 4096);
    CK(hipStreamEndCapture(s1, &gB)); CK(hipGraphInstantiate(&xB, gB, nullptr, nullptr, 0));
    auto run = [&](hipGraphExec_t x, int n) {
        hipEventRecord(e0, s1);
        for (int i = 0; i < n; ++i) hipGraphLaunch(x, s1);
        hipEventRecord(e1, s1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / n;
    };
    const int reps = 12;
    struct Cfg { int G, nap; } cfgs[] = {{0, 0}, {8, 0}, {32, 0}, {64, 0}, {256, 0}, {256, 1}, {1024, 1}, {0, 0}};
    for (auto c : cfgs) {
        // heater long enough to cover both measurements (~ 2 * reps * (n_tiny * 2 us + 2 ms)); it ends by itself
        if (c.G) {
            const double want_us = 2.5 * reps * (n_tiny * 2.0 + 2500.0);
            const double per_iter_us = c.nap ? 0.55 : 0.30;     // 16 MFMAs (~0.27 us at 2 GHz) [+ s_sleep 127*64 clk]
            hipLaunchKernelGGL(k_mfma, c.G, 256, 0, s2, (int)(want_us / per_iter_us), c.nap, sink);
        }
        run(xB, 2);
        const float tb = run(xB, reps), ta = run(xA, reps);
        CK(hipStreamSynchronize(s2));
        printf("heater G=%4d%s, 6 big + %d tiny: A %9.1f us  B %9.1f us (%.2f us/tiny) -> %6.1f us per big launch\n", c.G,
               c.nap ? " (napping)" : "          ", n_tiny, ta, tb, tb / n_tiny, (ta - tb) / 6);
        fflush(stdout);
    }
    return 0;
}
