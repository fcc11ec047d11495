#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* __restrict__ src, float* __restrict__ out, int n16) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave fills chunk `wave` (1 KB) of LDS; lane l reads global 16-B slot perm(l) ; odd lanes of wave 1 masked
    int slot = wave * 64 + (63 - lane);            // reversed source order
    if (!(wave == 1 && (lane & 1))) {
        __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * slot), (lptr_t)(lds + wave * 1024), 16, 0, 0);
    }
    __syncthreads();
    const float* l = reinterpret_cast<const float*>(lds);
    for (int i = threadIdx.x; i < 4 * 64 * 4; i += blockDim.x) out[i] = l[i];
}
int main() {
    const int N = 4 * 64 * 4;
    float h[N], o[N];
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *d, *e;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(e, 0, sizeof(o));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096 + 1024, 0, d, e, 0);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    // expectation: out slot (wave*64 + lane) holds src slot wave*64 + 63 - lane
    int bad = 0, masked_written = 0;
    for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) {
        float got = o[(w * 64 + l) * 4], exp = (float)((w * 64 + 63 - l) * 4);
        if (w == 1 && (l & 1)) { if (got == exp) masked_written++; continue; }
        if (got != exp) { if (bad < 5) printf("w%d l%d got %g exp %g\n", w, l, got, exp); bad++; }
    }
    printf("bad=%d masked_written=%d (of 32)\n", bad, masked_written);
    return 0;
}
