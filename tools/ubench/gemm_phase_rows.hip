// gemm_phase_rate.hip with 1, 2 or 3 row tiles (32, 64, 96 operand rows) per weight fragment: the weight stream per wave is the
// same 4 KB per 16-deep k-block, the MFMAs it feeds are 6, 12 or 18 -- how far can the rows per workgroup shrink (two
// half-tiles per CU = 48 rows = "1.5") before the CU's address unit (16 cycles per wave load, 4 waves) bounds the phase?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I transhuman_amd/csrc tools/ubench/gemm_phase_rows.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "k_mlp_fused_kernel.h"

template <int RT, int KBT>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ w, long long* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 96 * STR256 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    f32x16 acc[2][RT];
    zero_acc<2, RT>(acc);
    const uint4* wl = w + (long long)wave * KBT * (2 * 2 * 64);
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        gemm_phase<RT, 2, STR256>(lds, lds + 96 * STR256, wl, KBT, lane, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < 2; ++c) for (int r = 0; r < RT; ++r) for (int e = 0; e < 16; ++e) s += acc[c][r][e];
    if (s == 12345.678f) out[2] = 1;
    if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int RT, int KBT>
static void run(const uint4* w, long long* d, int grid) {
    const int reps = 50;
    const int lds = 2 * 96 * STR256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<RT, KBT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipMemset(d, 0, 64);
    hipLaunchKernelGGL((k<RT, KBT>), dim3(grid), dim3(256), lds, 0, w, d, reps);
    long long h[2];
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double mf = reps * KBT * 3.0 * 2 * RT;
    printf("grid %4d rows %2d KB %3d: %.2f cycles per MFMA  (%.0f cycles per k-block: 4 weight loads per wave)\n", grid, 32 * RT,
           KBT, h[0] / mf, h[0] / (double)(reps * KBT));
}

int main() {
    long long* d;
    (void)hipMalloc(&d, 64);
    uint4* w;
    const size_t wbytes = 8ull * 64 * (3 * 2 * 64) * 16;
    (void)hipMalloc(&w, wbytes);
    (void)hipMemset(w, 0x11, wbytes);
    for (int grid : {1, 256}) {
        run<3, 16>(w, d, grid);
        run<2, 16>(w, d, grid);
        run<1, 16>(w, d, grid);
        run<3, 64>(w, d, grid);
        run<2, 64>(w, d, grid);
        run<1, 64>(w, d, grid);
    }
    return 0;
}
