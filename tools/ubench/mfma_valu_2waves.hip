// Two waves per SIMD: does wave B's VALU work run under wave A's MFMAs?  512 threads per workgroup (8 waves: two per SIMD), one
// workgroup per CU.  Waves 0-3 run NM MFMAs, waves 4-7 run NV v_pk_fma_f32 (mode 0) / v_fma_f32 (mode 1); either set can be switched off.
// Prints the shader cycles each set took: alone, and together.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu_2waves.hip -o /tmp/mv2 && /tmp/mv2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void kern(long long* out, float* sink, int nm, int nv) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    if (wave < 4) {
        f32x16 acc[6];
        for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        h8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
        for (int it = 0; it < nm; ++it) {
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][7];
    } else {
        f32x2 v[12];
        for (int i = 0; i < 12; ++i) v[i] = (f32x2){threadIdx.x * 1e-3f + i, 1.0f + i};
        const f32x2 m = {0.999f, 1.001f}, c = {1e-3f, 2e-3f};
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                if (MODE == 0) v[k] = __builtin_elementwise_fma(v[k], m, c);
                else v[k][0] = fmaf(v[k][0], m[0], c[0]);
            }
        }
        for (int i = 0; i < 12; ++i) s += v[i][0] + v[i][1];
    }
    const long long t1 = clock64();
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int MODE>
void run(long long* d_out, float* d_sink, int nm, int nv, const char* what) {
    hipFuncSetAttribute((const void*)kern<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((kern<MODE>), dim3(256), dim3(512), 100 * 1024, 0, d_out, d_sink, nm, nv);
    hipDeviceSynchronize();
    long long t[8];
    hipMemcpy(t, d_out, 64, hipMemcpyDeviceToHost);
    printf("%-44s mfma waves %8lld cycles (%.1f per MFMA)   valu waves %8lld cycles (%.2f per instruction)\n", what, t[0],
           nm ? (double)t[0] / (nm * 6.0) : 0.0, t[4], nv ? (double)t[4] / (nv * 12.0) : 0.0);
}

int main() {
    long long* d_out; float* d_sink;
    hipMalloc((void**)&d_out, 64); hipMalloc((void**)&d_sink, 64);
    run<0>(d_out, d_sink, 2000, 0, "MFMA alone");
    run<0>(d_out, d_sink, 0, 8000, "v_pk_fma_f32 alone");
    run<1>(d_out, d_sink, 0, 8000, "v_fma_f32 alone");
    run<0>(d_out, d_sink, 2000, 8000, "MFMA + v_pk_fma_f32 (2 waves per SIMD)");
    run<1>(d_out, d_sink, 2000, 8000, "MFMA + v_fma_f32 (2 waves per SIMD)");
    run<1>(d_out, d_sink, 2000, 4000, "MFMA + half the v_fma_f32");
    return 0;
}
