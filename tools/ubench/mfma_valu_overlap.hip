// Do a wave's own VALU instructions issue in the shadow of its MFMAs?  One wave per SIMD (4 per workgroup, one workgroup per CU
// through the LDS size), a loop of v_mfma_f32_32x32x16_f16 on NACC independent accumulators with K independent VALU instructions
// (v_pk_fma_f32 on private registers) behind every MFMA.  Prints shader cycles per MFMA for K = 0 .. 12.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int K, int MODE>
__global__ __launch_bounds__(256, 1) void kern(long long* out, float* sink, int iters) {
    extern __shared__ char lds[];
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x2 v[12];
    for (int i = 0; i < 12; ++i) v[i] = (f32x2){threadIdx.x * 1e-3f + i, 1.0f + i};
    const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, 2e-3f};
    float* lp = reinterpret_cast<float*>(lds) + threadIdx.x * 4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 0) v[k] = __builtin_elementwise_fma(v[k], m, c);                 // VALU
                else if (MODE == 1) v[k][0] += lp[k * 1024];                                 // LDS read + add
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 12; ++i) s += v[i][0] + v[i][1];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K, int MODE>
void run(long long* d_out, float* d_sink) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)kern<K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((kern<K, MODE>), dim3(256), dim3(256), 100 * 1024, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    long long t;
    hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
    printf("mode %d K = %2d: %.2f cycles per MFMA\n", MODE, K, (double)t / (iters * 6.0));
}

int main() {
    long long* d_out; float* d_sink;
    hipMalloc((void**)&d_out, 64); hipMalloc((void**)&d_sink, 64);
    run<0, 0>(d_out, d_sink); run<1, 0>(d_out, d_sink); run<2, 0>(d_out, d_sink); run<4, 0>(d_out, d_sink); run<6, 0>(d_out, d_sink);
    run<7, 0>(d_out, d_sink); run<8, 0>(d_out, d_sink); run<10, 0>(d_out, d_sink); run<12, 0>(d_out, d_sink);
    run<1, 1>(d_out, d_sink); run<2, 1>(d_out, d_sink); run<4, 1>(d_out, d_sink);
    return 0;
}
