// Microbenchmark: cost of agent-scope fences and same-address atomics in tiny kernels (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int* ctr, const float* in, int groups) {
    float v = in[blockIdx.x * 256 + threadIdx.x];
    if (MODE == 4 || MODE == 5) __hip_atomic_store(out + blockIdx.x * 256 + threadIdx.x, v * 2.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out[blockIdx.x * 256 + threadIdx.x] = v * 2.f;
    if (MODE == 1 || MODE == 3) __threadfence();
    if (MODE == 5) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (MODE >= 2 && threadIdx.x == 0) {
        int old = atomicAdd(ctr + (blockIdx.x % groups), 1);
        if (old == 1 << 30) out[0] = 0.f;
    }
}

template <int MODE>
float run(float* out, int* ctr, const float* in, int blocks, int groups) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, blocks, 256, 0, 0, out, ctr, in, groups);
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<MODE>, blocks, 256, 0, 0, out, ctr, in, groups);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 200.f;
}

int main() {
    float *out, *in; int* ctr;
    CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&in, 4096 * 256 * 4)); CK(hipMalloc(&ctr, 4096 * 4));
    CK(hipMemset(in, 0, 4096 * 256 * 4)); CK(hipMemset(ctr, 0, 4096 * 4));
    for (int blocks : {128, 256, 1024}) for (int groups : {1, 2, 8, 64}) {
        printf("blocks %4d groups %2d: plain %.2f  fence %.2f  atomic %.2f  fence+atomic %.2f  sc1store+atomic %.2f  sc1store+wgfence+atomic %.2f us\n", blocks, groups,
               run<0>(out, ctr, in, blocks, groups), run<1>(out, ctr, in, blocks, groups), run<2>(out, ctr, in, blocks, groups),
               run<3>(out, ctr, in, blocks, groups), run<4>(out, ctr, in, blocks, groups), run<5>(out, ctr, in, blocks, groups));
    }
    return 0;
}
