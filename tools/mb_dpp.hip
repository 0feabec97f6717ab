// Probe: which lane does __builtin_amdgcn_update_dpp(row_shl:1 / row_shr:1) read from?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int v = threadIdx.x;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x101, 0xf, 0xf, false);   // row_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
}
int main() {
    int* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, d);
    int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row_shl:1 ->"); for (int i = 0; i < 34; ++i) printf(" %d", h[i]); printf("\n");
    printf("row_shr:1 ->"); for (int i = 0; i < 34; ++i) printf(" %d", h[64 + i]); printf("\n");
    return 0;
}
