#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
    unsigned x = threadIdx.x * 3u + 1u;
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i : {0, 1, 31, 32, 33, 63}) printf("lane %d: in %u r0 %u r1 %u\n", i, i * 3u + 1u, h[i], h[64 + i]);
    return 0;
}
