// MFMA issue rate with DISTINCT A / B operand registers per instruction (the fused MLP's k-block: 3 weight
// fragments x 3 activation fragments x 3 hi/lo terms on 9 accumulators), against the same-operand loop of
// mfma_rate_clock.hip.  One workgroup per CU, 4 waves, full chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(long long* out, int iters, float seed) {
    f32x16 acc[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    h8 xh[3], xl[3], wh[3], wl[3];
    for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 8; ++j) {
            xh[r][j] = (_Float16)(seed + threadIdx.x * 0.001f + j + r); xl[r][j] = (_Float16)(seed * 0.25f + j - r);
            wh[r][j] = (_Float16)(seed * 0.5f + j * 0.01f + r); wl[r][j] = (_Float16)(seed * 0.125f + j * 0.02f - r);
        }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < 3; ++r) acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], xh[0], acc[c][r], 0, 0, 0);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[c], xh[r], acc[c][r], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[c], xl[r], acc[c][r], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[c], xh[r], acc[c][r], 0, 0, 0);
        }
        if (MODE == 2) {     // perturb the operands each iteration like freshly loaded fragments (VALU writes between bursts)
#pragma unroll
            for (int r = 0; r < 3; ++r) { xh[r][0] += (_Float16)1; wh[r][1] += (_Float16)1; xl[r][2] += (_Float16)1; wl[r][3] += (_Float16)1; }
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) for (int e = 0; e < 16; ++e) s += acc[a][b][e];
    if (s == 12345.678f) out[2] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
    long long* d; hipMalloc(&d, 64);
    for (int grid : {1, 256}) for (int mode : {0, 1, 2}) {
        int iters = 2000;
        hipMemset(d, 0, 64);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("grid %4d mode %d (%s): %.2f clock64 ticks/MFMA\n", grid, mode,
               mode == 0 ? "same operands" : mode == 1 ? "3x3 distinct operands" : "distinct + operand writes", h[0] / (iters * 27.0));
    }
    return 0;
}
