// Round 6: the premise of a two-waves-per-SIMD fused MLP, measured with hand-placed instruction streams.
//   (1) one wave per SIMD: how many PLAIN single-issue instructions (v_fma_f32, v_cvt_pk_f16_f32, an independent ds_read_b128)
//       hide in the gap behind a v_mfma_f32_32x32x16_f16 -- the round-5 test (mfma_valu_overlap.hip) used v_pk_fma_f32 and a
//       dependent ds_read + add; MI355X_MICROARCH.md says <= 5 plain ones are free and names packed f32 as the anti-lever;
//   (2) VALU / LDS-only streams with one and with two waves per SIMD: does a second wave double the issue rate of the
//       non-matrix phases (fillings, epilogues)?
//   (3) v_mfma_f32_16x16x32_f16 vs 32x32x16: cycles per MFMA of a SIMD with one and two waves issuing.
// Every instruction is an `asm volatile` (program order = issue order).  Prints shader cycles (clock64) per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/issue_rates.hip -o /tmp/ir && /tmp/ir
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA32(ACC) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define MFMA16(ACC) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(m0), "v"(c0))
#define PKFMA(X) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(m2), "v"(c2))
#define CVT(D, X) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(D) : "v"(X), "v"(X))
#define DSR(D, A, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(D) : "v"(A))
#define DSW(A, D, OFF) asm volatile("ds_write_b64 %0, %1 offset:" #OFF : : "v"(A), "v"(D))
#define WAITL() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// TEST:  0 v_fma_f32 x12            1 v_pk_fma_f32 x12         2 ds_read_b128 x8           3 split-epilogue mix
//        4 MFMA32 x6                5 MFMA16 x12
//       10+K  MFMA32 + K v_fma_f32 per gap     30+K  MFMA32 + 1 ds_read_b128 + (K-1) v_fma_f32 per gap
//       50+K  MFMA32 + K v_pk_fma_f32 per gap  70+K  MFMA32 + K v_cvt_pk_f16_f32 per gap
//       90+K  MFMA16 + K v_fma_f32 per gap
template <int NT, int TEST>
__global__ __launch_bounds__(NT, 1) void kern(long long* out, float* sink, int iters) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    f32x16 acc[6];
    f32x4 acq[12];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int i = 0; i < 12; ++i) for (int e = 0; e < 4; ++e) acq[i][e] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float v[12];
    f32x2 p[12];
    u32x4 d[8];
    unsigned cv[12];
    for (int i = 0; i < 12; ++i) { v[i] = threadIdx.x * 1e-3f + i; p[i] = (f32x2){v[i], 1.0f + i}; cv[i] = 0u; }
    for (int i = 0; i < 8; ++i) d[i] = (u32x4){0u, 0u, 0u, 0u};
    const float m0 = 0.999f, c0 = 1e-3f;
    const f32x2 m2 = {0.999f, 1.001f}, c2 = {1e-3f, 2e-3f};
    const unsigned la = (threadIdx.x & 63) * 16 + wave * 4096;       // 1 KiB per wave instruction, conflict-free
    const unsigned lw = (threadIdx.x & 63) * 8 + wave * 4096;
    f32x2 wd = {1.f, 2.f};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (TEST == 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) FMA(v[k]);
        } else if constexpr (TEST == 1) {
#pragma unroll
            for (int k = 0; k < 12; ++k) PKFMA(p[k]);
        } else if constexpr (TEST == 2) {
            DSR(d[0], la, 0); DSR(d[1], la, 1024); DSR(d[2], la, 2048); DSR(d[3], la, 3072);
            DSR(d[4], la, 0); DSR(d[5], la, 1024); DSR(d[6], la, 2048); DSR(d[7], la, 3072);
            WAITL();
        } else if constexpr (TEST == 3) {
            // per pair of values: fma (scale + bias), max (relu), cvt_pk, 2 x fma_mix (lo halves), and, pk_max_u16; a ds_write_b64 per 4
#pragma unroll
            for (int k = 0; k < 12; k += 2) {
                FMA(v[k]); FMA(v[k + 1]);
                asm volatile("v_max_f32 %0, %0, 0" : "+v"(v[k]));
                asm volatile("v_max_f32 %0, %0, 0" : "+v"(v[k + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[k]) : "v"(v[k]), "v"(v[k + 1]));
                asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(cv[k + 1]) : "v"(cv[k]), "v"(v[k]));
                asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(cv[k + 1]) : "v"(cv[k]), "v"(v[k + 1]));
                asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(cv[11 - k]) : "v"(cv[k]));
                if ((k & 2) == 0) DSW(lw, wd, 0);
            }
        } else if constexpr (TEST == 4) {
#pragma unroll
            for (int i = 0; i < 6; ++i) MFMA32(acc[i]);
        } else if constexpr (TEST == 5) {
#pragma unroll
            for (int i = 0; i < 12; ++i) MFMA16(acq[i]);
        } else if constexpr (TEST >= 10 && TEST < 30) {
            constexpr int K = TEST - 10;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                MFMA32(acc[i]);
#pragma unroll
                for (int k = 0; k < K; ++k) FMA(v[(i * K + k) % 12]);
            }
        } else if constexpr (TEST >= 30 && TEST < 50) {
            constexpr int K = TEST - 30;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                MFMA32(acc[i]);
                DSR(d[i], la, 0);
#pragma unroll
                for (int k = 1; k < K; ++k) FMA(v[(i * K + k) % 12]);
            }
            WAITL();
        } else if constexpr (TEST >= 50 && TEST < 70) {
            constexpr int K = TEST - 50;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                MFMA32(acc[i]);
#pragma unroll
                for (int k = 0; k < K; ++k) PKFMA(p[(i * K + k) % 12]);
            }
        } else if constexpr (TEST >= 70 && TEST < 90) {
            constexpr int K = TEST - 70;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                MFMA32(acc[i]);
#pragma unroll
                for (int k = 0; k < K; ++k) CVT(cv[(i * K + k) % 12], v[k]);
            }
        } else {
            constexpr int K = TEST - 90;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                MFMA16(acq[i]);
#pragma unroll
                for (int k = 0; k < K; ++k) FMA(v[(i * K + k) % 12]);
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 12; ++i) s += v[i] + p[i][0] + p[i][1] + acq[i][0] + (float)cv[i];
    for (int i = 0; i < 8; ++i) s += (float)d[i][0];
    if (s == 123.456f) sink[0] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int NT, int TEST>
double run(long long* d_out, float* d_sink, int per_iter) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)kern<NT, TEST>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((kern<NT, TEST>), dim3(256), dim3(NT), 100 * 1024, 0, d_out, d_sink, iters);
    hipDeviceSynchronize();
    long long t[8];
    hipMemcpy(t, d_out, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < NT / 64; ++w) mx = t[w] > mx ? t[w] : mx;
    return (double)mx / ((double)iters * per_iter);
}

#define BOTH(TEST, PER, WHAT)                                                                                      \
    do {                                                                                                            \
        const double a1 = run<256, TEST>(d_out, d_sink, PER), a2 = run<512, TEST>(d_out, d_sink, PER);              \
        printf("%-58s 1 wave/SIMD %7.2f   2 waves/SIMD %7.2f per wave = %7.2f per SIMD\n", WHAT, a1, a2, a2 / 2.0); \
    } while (0)

int main() {
    long long* d_out; float* d_sink;
    hipMalloc((void**)&d_out, 64); hipMalloc((void**)&d_sink, 64);
    printf("cycles per instruction of a wave (VALU / LDS streams) or per MFMA (matrix streams); full chip, 256 workgroups\n");
    BOTH(0, 12, "v_fma_f32 (12 independent chains)");
    BOTH(1, 12, "v_pk_fma_f32");
    BOTH(2, 8, "ds_read_b128 (8 in flight, 1 KiB each)");
    BOTH(3, 45, "split epilogue mix (45 instr: fma max cvt mix pkmax dsw)");
    BOTH(4, 6, "v_mfma_f32_32x32x16_f16 alone");
    BOTH(5, 12, "v_mfma_f32_16x16x32_f16 alone");
    BOTH(11, 6, "MFMA32 + 1 v_fma_f32 per gap");
    BOTH(12, 6, "MFMA32 + 2 v_fma_f32");
    BOTH(13, 6, "MFMA32 + 3 v_fma_f32");
    BOTH(14, 6, "MFMA32 + 4 v_fma_f32");
    BOTH(15, 6, "MFMA32 + 5 v_fma_f32");
    BOTH(16, 6, "MFMA32 + 6 v_fma_f32");
    BOTH(18, 6, "MFMA32 + 8 v_fma_f32");
    BOTH(22, 6, "MFMA32 + 12 v_fma_f32");
    BOTH(31, 6, "MFMA32 + 1 ds_read_b128");
    BOTH(33, 6, "MFMA32 + 1 ds_read_b128 + 2 v_fma_f32");
    BOTH(35, 6, "MFMA32 + 1 ds_read_b128 + 4 v_fma_f32");
    BOTH(51, 6, "MFMA32 + 1 v_pk_fma_f32");
    BOTH(52, 6, "MFMA32 + 2 v_pk_fma_f32");
    BOTH(54, 6, "MFMA32 + 4 v_pk_fma_f32");
    BOTH(72, 6, "MFMA32 + 2 v_cvt_pk_f16_f32");
    BOTH(74, 6, "MFMA32 + 4 v_cvt_pk_f16_f32");
    BOTH(91, 12, "MFMA16 + 1 v_fma_f32 per gap");
    BOTH(92, 12, "MFMA16 + 2 v_fma_f32");
    BOTH(93, 12, "MFMA16 + 3 v_fma_f32");
    return 0;
}
