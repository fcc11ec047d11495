// MFMA issue rate when the accumulators cannot all live in architectural VGPRs (> 256 registers per lane: the
// compiler places them in AGPRs, as in mlp_fused_kernel) against the all-VGPR case of mfma_operands.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NA>
__global__ __launch_bounds__(256, 1) void k(long long* out, int iters, float seed) {
    f32x16 acc[NA];
    for (int a = 0; a < NA; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    h8 xh[3], wh[3];
    for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 8; ++j) { xh[r][j] = (_Float16)(seed + threadIdx.x * 0.001f + j + r); wh[r][j] = (_Float16)(seed * 0.5f + j * 0.01f + r); }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[a % 3], xh[(a / 3) % 3], acc[a], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < NA; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.678f) out[2] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    long long* d; (void)hipMalloc(&d, 64);
    for (int grid : {1, 256}) for (int na : {9, 15, 24}) {
        int iters = 3000;
        (void)hipMemset(d, 0, 64);
        if (na == 9) hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        else if (na == 15) hipLaunchKernelGGL(k<15>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        else hipLaunchKernelGGL(k<24>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("grid %4d, %2d accumulators (%3d registers): %.2f clock64 ticks/MFMA\n", grid, na, na * 16, h[0] / ((double)iters * na));
    }
    return 0;
}
