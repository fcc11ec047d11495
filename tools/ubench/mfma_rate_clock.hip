#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256, 1) void k(long long* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    h8 x, w;
    for (int j = 0; j < 8; ++j) { x[j] = (_Float16)(seed + threadIdx.x * 0.001f + j); w[j] = (_Float16)(seed * 0.5f + j * 0.01f); }
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc[a], 0, 0, 0);
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 12345.678f) out[2] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
int main() {
    long long* d; hipMalloc(&d, 64);
    for (int grid : {1, 256, 1024}) for (int nacc : {6, 9}) {
        int iters = 2000;
        hipMemset(d, 0, 64);
        if (nacc == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        else hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double n = (double)iters * 3 * nacc;
        printf("grid %4d nacc %d: %.2f clock64 ticks/MFMA, %.2f ns/MFMA (wall_clock64 100MHz), => %.3f GHz tick rate\n", grid, nacc,
               h[0] / n, h[1] * 10.0 / n, (double)h[0] / (h[1] * 10.0));
    }
    return 0;
}
