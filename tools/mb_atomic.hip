// Microbenchmark: G workgroups each ADD one float to each of N shared addresses (the BatchNorm statistics
// pattern) with global_atomic_add_f32, against the same workgroups storing a [G][N] slab.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* sums, float* slab, const float* in, int N) {
    float v = in[(blockIdx.x * 256 + threadIdx.x) & 65535];
    for (int c = threadIdx.x; c < N; c += 256) {
        if (MODE == 0) slab[(long)blockIdx.x * N + c] = v;
        else if (MODE == 1) atomicAdd(sums + c, v);                       // every workgroup starts at address 0
        else atomicAdd(sums + (c + blockIdx.x * 64) % N, v);              // staggered start
    }
}
template <int MODE>
float run(float* sums, float* slab, const float* in, int G, int N) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, G, 256, 0, 0, sums, slab, in, N);
    hipEventRecord(a);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k<MODE>, G, 256, 0, 0, sums, slab, in, N);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 100.f;
}
int main() {
    float *sums, *slab, *in;
    CK(hipMalloc(&sums, 1 << 20)); CK(hipMalloc(&slab, 64 << 20)); CK(hipMalloc(&in, 65536 * 4));
    CK(hipMemset(sums, 0, 1 << 20)); CK(hipMemset(in, 0, 65536 * 4));
    for (int G : {32, 64, 128, 256, 1024}) for (int N : {512, 3072}) {
        printf("G %4d workgroups x N %4d addresses: slab store %.2f us   atomics %.2f us   atomics staggered %.2f us\n", G, N,
               run<0>(sums, slab, in, G, N), run<1>(sums, slab, in, G, N), run<2>(sums, slab, in, G, N));
    }
    return 0;
}
