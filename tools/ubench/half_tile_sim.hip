// Would TWO co-resident half-tile workgroups per CU beat the one 96-row tile per CU of mlp_fused_kernel?  
// A skeleton with the fused kernel's resource shape, not its arithmetic: a workgroup = 4 waves (<= 256 registers each) owns 48
// operand rows (16 samples x 3 views) as fp16 hi / lo planes in LDS and runs NPH dense layers 256 -> 256 on
// v_mfma_f32_16x16x32_f16 (three products per term, weights streamed from an L2-resident image: 8 KB per wave and 32-deep
// k-block), each followed by the relu / hi-lo split epilogue back into the operand planes and a filler of VALU work standing
// for the cross-view attention; two LDS-DMA fillings of 48 KB per tile from a buffer far larger than the caches stand for the
// pixel-feature rows.  Per half-tile and wave: NPH * 8 * 36 MFMAs of 16 cycles (NPH = 7: 32.3 k cycles; the real kernel issues
// 2070 MFMAs of 32 cycles per 32-sample tile = 33.1 k per half).  Run with 79 KB of LDS (two workgroups per CU) and with 120 KB
// (one per CU): the ratio is what the second workgroup hides; the absolute time per PAIR of half-tiles compares with the
// real kernel's 32-sample tile (about 64 us at the clock the chip sustains under this load).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/half_tile_sim.hip -o tools/ubench/_bin/half_tile_sim
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int ROWS = 48, K = 256, STR = 2 * K + 16, PLANE = ROWS * STR;      // 25 344 B per plane
constexpr int KB = K / 32;                                                    // 8 k-blocks of 32
constexpr int WPB = 8 * 64;                                                   // uint4 per wave and k-block (4 col tiles x hi/lo)

#ifndef NPH
#define NPH 7
#endif
#ifndef FILL
#define FILL 160        // dependent-chain VALU filler per phase and lane (x 4 chains)
#endif

#ifndef RING
#define RING 3          // weight blocks in registers: RING - 1 requested ahead of the one the MFMAs read
#endif
__device__ __forceinline__ void gemm(const char* __restrict__ hi, const char* __restrict__ lo, const uint4* __restrict__ wl,
                                     int lane, f4 (&acc)[4][3], int wpb) {
    uint4 wr[RING][8];
#pragma unroll
    for (int b = 0; b < RING - 1; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[b][i] = wl[b * wpb + i * 64 + lane];
    const int boff = (lane & 15) * STR + (lane >> 4) * 16;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + RING - 1 < KB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wr[(kb + RING - 1) % RING][i] = wl[(kb + RING - 1) * wpb + i * 64 + lane];
        }
        h8 bh[3], bl[3];
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            bh[rt] = *reinterpret_cast<const h8*>(hi + rt * 16 * STR + boff + kb * 64);
            bl[rt] = *reinterpret_cast<const h8*>(lo + rt * 16 * STR + boff + kb * 64);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const h8 ah = *reinterpret_cast<const h8*>(&wr[kb % RING][2 * ct]), al = *reinterpret_cast<const h8*>(&wr[kb % RING][2 * ct + 1]);
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[rt], acc[ct][rt], 0, 0, 0);
                acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[rt], acc[ct][rt], 0, 0, 0);
                acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[rt], acc[ct][rt], 0, 0, 0);
            }
        }
    }
}

__device__ __forceinline__ void epilogue(char* __restrict__ hi, char* __restrict__ lo, int wave, int lane, f4 (&acc)[4][3],
                                         const float* __restrict__ bias) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int ch = wave * 64 + ct * 16 + 4 * (lane >> 4);
        const f4 b = *reinterpret_cast<const f4*>(bias + ch);
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            h4 vh, vl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaxf(fmaf(acc[ct][rt][i], 0.25f, b[i]), 0.f);
                vh[i] = (_Float16)v;
                vl[i] = (_Float16)(v - (float)vh[i]);
                acc[ct][rt][i] = 0.f;
            }
            const int row = 16 * rt + (lane & 15);
            *reinterpret_cast<h4*>(hi + row * STR + 2 * ch) = vh;
            *reinterpret_cast<h4*>(lo + row * STR + 2 * ch) = vl;
        }
    }
}

template <int NBLK>
__global__ __launch_bounds__(256, NBLK) void sim(const uint4* __restrict__ w, const char* __restrict__ rows,
                                                 const float* __restrict__ bias, float* __restrict__ sink, int stage_on, int flags) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* hi = lds;
    char* lo = lds + PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f4 acc[4][3];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[c][r] = f4{0.f, 0.f, 0.f, 0.f};
    float fl[4] = {1.f + lane, 2.f, 3.f, 4.f};
    const char* myrows = rows + (long long)blockIdx.x * (2 * 48 * 1024);
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph == 0 || ph == NPH / 2) {            // a filling of the operand planes (48 KB) by LDS-DMA
            if (stage_on) {
                const char* g = myrows + (ph ? 48 * 1024 : 0) + lane * 16;
                for (int i = wave; i < 48; i += 4)
                    __builtin_amdgcn_global_load_lds((gptr_t)(g + i * 1024), (lptr_t)(lds + i * 1024), 16, 0, 0);
            } else {
                for (int i = tid; i < 48 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0x3c3c3c3cu, 0x3c3c3c3cu, 0x3c3c3c3cu, 0x3c3c3c3cu);
            }
            __syncthreads();
        }
        if (flags & 1) gemm(hi, lo, (flags & 8) ? w : w + ((long long)ph * 4 + wave) * KB * WPB, lane, acc, (flags & 8) ? 0 : WPB);
        __syncthreads();                            // every wave has read the operand
        if (flags & 2) epilogue(hi, lo, wave, lane, acc, bias + ph * 256);
#pragma unroll 4
        for (int i = 0; i < ((flags & 4) ? FILL : 0); ++i) {           // cross-view attention / means / heads stand-in: four chains of VALU
            fl[0] = fmaf(fl[0], 1.0001f, 0.5f);
            fl[1] = fmaf(fl[1], 0.9999f, 0.25f);
            fl[2] = fmaf(fl[2], 1.0002f, 0.125f);
            fl[3] = fmaf(fl[3], 0.9998f, 0.75f);
        }
        __syncthreads();
    }
    float s = fl[0] + fl[1] + fl[2] + fl[3] + (float)*reinterpret_cast<_Float16*>(hi + lane * 2);
    if (s == 12345.678f) sink[0] = s;
}

template <int NBLK>
static float run(const uint4* w, const char* rows, const float* bias, float* sink, int grid, int lds, int stage_on, int flags = 7) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sim<NBLK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 12; ++it) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(sim<NBLK>, dim3(grid), dim3(256), lds, 0, w, rows, bias, sink, stage_on, flags);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    return best;
}

int main() {
    const int grid = 256 * 2 * 16;                  // 8192 half-tiles = 4096 tiles of 32 samples
    uint4* w;
    const size_t wbytes = (size_t)NPH * 4 * KB * WPB * 16;
    (void)hipMalloc(&w, wbytes);
    (void)hipMemset(w, 0x11, wbytes);
    char* rows;
    const size_t rbytes = (size_t)grid * 2 * 48 * 1024;
    (void)hipMalloc(&rows, rbytes);
    (void)hipMemset(rows, 0x3c, rbytes);
    float *bias, *sink;
    (void)hipMalloc(&bias, NPH * 256 * 4);
    (void)hipMemset(bias, 0, NPH * 256 * 4);
    (void)hipMalloc(&sink, 64);
    hipError_t err = hipGetLastError();
    printf("weights %.2f MB, rows %.2f GB, %d phases, %d MFMA(16x16x32) per wave and half-tile, alloc: %s\n", wbytes / 1e6, rbytes / 1e9,
           NPH, NPH * KB * 36, hipGetErrorString(err));
    const double pairs_per_cu = grid / 2.0 / 256.0;
    for (int i = 0; i < 3; ++i) (void)run<2>(w, rows, bias, sink, grid, 79 * 1024, 0);        // clocks up
    for (int stage_on : {1, 0, 1}) {
        const float t2 = run<2>(w, rows, bias, sink, grid, 79 * 1024, stage_on);
        const float t1 = run<1>(w, rows, bias, sink, grid, 120 * 1024, stage_on);
        const float t1r = run<2>(w, rows, bias, sink, grid, 120 * 1024, stage_on);     // the 256-register build, one per CU
        printf("staging %s: two per CU %.3f ms (%.1f us per pair of half-tiles per CU) | one per CU %.3f ms (512-register build), %.3f ms (256-register build) -> x%.2f\n",
               stage_on ? "HBM rows by LDS-DMA" : "none (LDS fill)   ", t2, 1e3 * t2 / pairs_per_cu, t1, t1r, t1r / t2);
    }
    for (int flags : {0, 1, 9, 3, 7, 15}) {
        const float a = run<2>(w, rows, bias, sink, grid, 79 * 1024, 1, flags), b = run<2>(w, rows, bias, sink, grid, 79 * 1024, 0, flags);
        printf("two per CU, parts %2d (1 gemm, 2 epilogue, 4 filler, 8 every weight load reads the same 8 KB): staged %.3f ms, LDS fill %.3f ms\n", flags, a, b);
    }
    return 0;
}
