// K6 (fused form): the whole per-point multi-view MLP in ONE kernel.
//
// Network._multiview_agg / cross_attention / _alpha_forward / _RGB_forward
// (cross_transformer.py:128-149, :291-353) for a tile of 32 samples x V views.
//
// Arithmetic: every dense layer runs on v_mfma_f32_32x32x16_f16 with BOTH
// operands split into fp16 hi + lo halves (x = hi + lo exactly to ~2^-22) and
// three products accumulated in fp32:  W_hi*x_hi + W_hi*x_lo + W_lo*x_hi.
// The dropped W_lo*x_lo term is 2^-22 relative, i.e. fp32-class accuracy
// (measured: raw logits within 7e-6 of the fp32 oracle) at 16/3 = 5.3x the
// fp32-MFMA rate.  Weights are pre-scaled by a power of two per layer so their
// lo halves stay in fp16's normal range; the scale is undone in fp32.
//
// Data flow per workgroup (256 threads = 4 waves, one per SIMD, 1 workgroup/CU):
//   activations that feed a GEMM live in LDS as fp16 hi/lo planes
//   [row = view*32 + sample][K] (row stride = 2K+16 B: conflict-free
//   ds_read_b128 fragments); each wave owns a 64-column slice of every layer
//   and keeps its outputs in MFMA accumulators (the layers are evaluated
//   transposed, out^T = W * in^T, so a lane holds 4 consecutive output channels
//   of ONE sample row -> 8-byte LDS stores, and the V views of a sample sit in
//   the same lane/register of V accumulator tiles: the 3x3 cross-view softmax
//   and the view means are pure register arithmetic).  Weight fragments are
//   streamed from the L2-resident packed image straight into VGPRs
//   (double-buffered), no LDS round trip.
// HBM traffic per sample: 3 KB (h) + 2 x 4.6 KB (f, read again for the RGB
// branch) + 124 B; every intermediate of the reference's ~40 kernels/chunk
// (~70 KB/sample of HBM round trips) stays on chip.
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "th_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define FM_PTS 32
#define STR256 528   // bytes per LDS row, K = 256 halves (+16)
#define STR192 400   // K = 192
#define STR288 592   // K = 288
#define STR128 272   // K = 128
#define KSTR 132     // floats per row of the fp32 key buffers

// ---- packed fused-layer image ---------------------------------------------------
// [wave 4][kb][ct][plane hi/lo][lane 64][8 halves]; lane = (kh<<5)|i holds
// W[col0(wave,ct)+i][16*kb + 8*kh + j] * 2^scale_log2 (zero padded).
// (struct FusedLayer / FusedParams: th_internal.h)

__device__ __forceinline__ void split_h(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// stage ROWS x KC fp32 values (global, row r -> sample (r&31), view (r>>5)) into hi/lo planes
template <int V, int KC, int STR>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, int ld, int coff, int pbase, int npts,
                                           char* __restrict__ hi, char* __restrict__ lo, int tid) {
    constexpr int C4 = KC / 4;
    constexpr int TOTAL = 32 * V * C4;
    constexpr int ITERS = (TOTAL + 255) / 256;
    f32x4v v[ITERS];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        int idx = tid + 256 * i;
        int row = idx / C4, c4 = idx % C4;
        int p = row & 31, vw = row >> 5;
        v[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        if (idx < TOTAL && p < npts)
            v[i] = __builtin_nontemporal_load(
                reinterpret_cast<const f32x4v*>(src + ((long long)(pbase + p) * V + vw) * ld + coff + 4 * c4));
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        int idx = tid + 256 * i;
        if (idx < TOTAL) {
            int row = idx / C4, c4 = idx % C4;
            h4 a, b;
            _Float16 x, y;
            split_h(v[i].x, x, y); a[0] = x; b[0] = y;
            split_h(v[i].y, x, y); a[1] = x; b[1] = y;
            split_h(v[i].z, x, y); a[2] = x; b[2] = y;
            split_h(v[i].w, x, y); a[3] = x; b[3] = y;
            *reinterpret_cast<h4*>(hi + row * STR + 8 * c4) = a;
            *reinterpret_cast<h4*>(lo + row * STR + 8 * c4) = b;
        }
    }
}

// one k-block (16 deep): acc[c][r] += W(c) * X(r)^T as three fp16 products, ordered term-major so
// consecutive MFMAs hit different accumulators (no back-to-back dependent issue)
template <int RT, int CT, int STR>
__device__ __forceinline__ void gemm_kblock(const char* __restrict__ ahi, const char* __restrict__ alo, int aoff, int kb,
                                            const uint4 (&w)[CT][2], f32x16 (&acc)[CT][RT]) {
    h8 xh[RT], xl[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        xh[r] = *reinterpret_cast<const h8*>(ahi + r * 32 * STR + aoff + kb * 32);
        xl[r] = *reinterpret_cast<const h8*>(alo + r * 32 * STR + aoff + kb * 32);
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&w[c][1]), xh[r], acc[c][r], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&w[c][0]), xl[r], acc[c][r], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&w[c][0]), xh[r], acc[c][r], 0, 0, 0);
}

template <int CT>
__device__ __forceinline__ void load_wfrag(const uint4* __restrict__ wl, int kb, uint4 (&w)[CT][2]) {
    const uint4* p = wl + (long long)kb * (CT * 2 * 64);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        w[c][0] = p[(c * 2 + 0) * 64];
        w[c][1] = p[(c * 2 + 1) * 64];
    }
}

// acc[ct][rt] += W_tile(ct) * A_rows(rt)^T over KB (even) k-blocks of 16.  The weight fragments
// ping-pong between two register sets (unrolled by two, no copies), so the loads of block k+1 are
// in flight during the whole MFMA burst of block k and are waited for only at their first use.
template <int RT, int CT, int STR>
__device__ __forceinline__ void gemm_phase(const char* __restrict__ ahi, const char* __restrict__ alo,
                                           const uint4* __restrict__ wp, int KB, int lane, f32x16 (&acc)[CT][RT]) {
    const uint4* wl = wp + lane;     // this wave's stream: per kb: CT x {hi, lo} x 64 lanes x 16 B
    const int aoff = (lane & 31) * STR + (lane >> 5) * 16;
    uint4 w0[CT][2], w1[CT][2];
    load_wfrag<CT>(wl, 0, w0);
    // never fully unroll (a constant KB would hoist every weight load of the phase -> VGPR spills)
#pragma unroll 1
    for (int kb = 0; kb < KB; kb += 2) {
        load_wfrag<CT>(wl, kb + 1, w1);
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ABOVE this block's MFMA burst
        gemm_kblock<RT, CT, STR>(ahi, alo, aoff, kb, w0, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 2 < KB) load_wfrag<CT>(wl, kb + 2, w0);
        __builtin_amdgcn_sched_barrier(0);
        gemm_kblock<RT, CT, STR>(ahi, alo, aoff, kb + 1, w1, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int CT, int RT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[CT][RT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][r][e] = 0.f;
}

// channel of accumulator register e (within a 32-wide column tile) for this lane
__device__ __forceinline__ int acc_chan(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// y = acc*inv_scale + bias (per output channel), optional relu, in place
template <int RT>
__device__ __forceinline__ void finish_tile(f32x16 (&acc)[RT], const float* __restrict__ bias, int col0, float inv_scale,
                                            bool relu, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 b = *reinterpret_cast<const float4*>(bias + col0 + 8 * g + 4 * (lane >> 5));
        float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y = acc[r][4 * g + q] * inv_scale + bb[q];
                acc[r][4 * g + q] = relu ? fmaxf(y, 0.f) : y;
            }
    }
}

// write one 32x32 output tile (this lane: row = rt*32 + (lane&31), 4x4 channels) as hi/lo halves
template <int STR>
__device__ __forceinline__ void store_tile_h(const f32x16& t, int row, int col0, char* __restrict__ hi,
                                             char* __restrict__ lo, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int c = col0 + 8 * g + 4 * (lane >> 5);
        h4 a, b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            _Float16 x, y;
            split_h(t[4 * g + q], x, y);
            a[q] = x; b[q] = y;
        }
        *reinterpret_cast<h4*>(hi + row * STR + 2 * c) = a;
        *reinterpret_cast<h4*>(lo + row * STR + 2 * c) = b;
    }
    // keep the scheduler from hoisting the next tile's accumulator reads/conversions above these
    // stores: VALU temporaries must be arch VGPRs (<= 256) while the big tensors sit in AGPRs
    __builtin_amdgcn_sched_barrier(0);
}
// same tile as fp32 (key buffers for the cross-view dots)
__device__ __forceinline__ void store_tile_f(const f32x16& t, int row, int col0, float* __restrict__ dst, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int c = col0 + 8 * g + 4 * (lane >> 5);
        *reinterpret_cast<float4*>(dst + row * KSTR + c) =
            make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
    }
}

// barrier + optional cycle stamp (developer aid: TH_FUSED_DBG=1 prints per-phase cycles of one tile)
#define FM_SYNC()                                                                   \
    do {                                                                            \
        __syncthreads();                                                            \
        if (P.dbg != nullptr && tid == 0 && blockIdx.x == gridDim.x / 2) P.dbg[dbg_i++] = clock64(); \
    } while (0)

#define ABUF_BYTES (2 * 96 * STR288)
#define MBUF_BYTES (2 * 32 * STR256)
#define MISC_FLOATS (9 * 32 + 32 * 4 + 32 * 28 + 4 * 32 * 4 + 8)
#define FUSED_LDS_BYTES (ABUF_BYTES + MBUF_BYTES + MISC_FLOATS * 4)

template <int V>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* abuf = lds;
    char* mbuf = lds + ABUF_BYTES;
    float* misc = reinterpret_cast<float*>(lds + ABUF_BYTES + MBUF_BYTES);
    float* probs = misc;                  // [V*V][32]
    float* sig = misc + 9 * 32;           // [32] (+ padding)
    float* vds = sig + 32 * 4;            // [32][28]
    float* part = vds + 32 * 28;          // [4 waves][32][4]
    int* flag = reinterpret_cast<int*>(part + 4 * 32 * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pbase = blockIdx.x * FM_PTS;
    const int npts = min(FM_PTS, P.P - pbase);
    constexpr int ROWS = 32 * V;
    const int myrow = lane & 31;
    int dbg_i = 0;
    if (P.dbg != nullptr && tid == 0 && blockIdx.x == gridDim.x / 2) P.dbg[dbg_i++] = clock64();

    // ================= token branch: s = relu(fc_0 h); ks|vs = kv1(s) =================
    stage_rows<V, 256, STR256>(P.h, 256, 0, pbase, npts, abuf, abuf + ROWS * STR256, tid);
    for (int i = tid; i < 32 * 28; i += 256) {
        int p = i / 28, c = i % 28;
        vds[i] = (p < npts && c < 27) ? P.vd[(long long)(pbase + p) * 27 + c] : 0.f;
    }
    FM_SYNC();
    f32x16 acc2[2][V];
    zero_acc<2, V>(acc2);
    gemm_phase<V, 2, STR256>(abuf, abuf + ROWS * STR256, P.fc_0.w + (long long)wave * P.fc_0.KB * (2 * 2 * 64), P.fc_0.KB,
                             lane, acc2);
    FM_SYNC();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        finish_tile<V>(acc2[c], P.fc_0.bias, wave * 64 + c * 32, P.fc_0.inv_scale, true, lane);
#pragma unroll
        for (int r = 0; r < V; ++r)
            store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, abuf + ROWS * STR256, lane);
    }
    FM_SYNC();
    // kv layers: column tile 0 = key tile `wave` (cols wave*32..), tiles 1,2 = value cols 128 + wave*64 ..
    f32x16 ks[1][V], vs[2][V];
    {
        f32x16 acc3[3][V];
        zero_acc<3, V>(acc3);
        gemm_phase<V, 3, STR256>(abuf, abuf + ROWS * STR256, P.kv1.w + (long long)wave * P.kv1.KB * (3 * 2 * 64), P.kv1.KB,
                                 lane, acc3);
        finish_tile<V>(acc3[0], P.kv1.bias, wave * 32, P.kv1.inv_scale, false, lane);
        finish_tile<V>(acc3[1], P.kv1.bias, 128 + wave * 64, P.kv1.inv_scale, false, lane);
        finish_tile<V>(acc3[2], P.kv1.bias, 128 + wave * 64 + 32, P.kv1.inv_scale, false, lane);
#pragma unroll
        for (int r = 0; r < V; ++r) { ks[0][r] = acc3[0][r]; vs[0][r] = acc3[1][r]; vs[1][r] = acc3[2][r]; }
    }
    FM_SYNC();

    // ================= pixel branch: p = relu(alpha_res_0 f); kp|vp = kv0(p) =================
    zero_acc<2, V>(acc2);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        stage_rows<V, 192, STR192>(P.f, 384, half * 192, pbase, npts, abuf, abuf + ROWS * STR192, tid);
        FM_SYNC();
        gemm_phase<V, 2, STR192>(abuf, abuf + ROWS * STR192,
                                 P.ar0.w + ((long long)wave * P.ar0.KB + half * 12) * (2 * 2 * 64), 12, lane, acc2);
        FM_SYNC();
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        finish_tile<V>(acc2[c], P.ar0.bias, wave * 64 + c * 32, P.ar0.inv_scale, true, lane);
#pragma unroll
        for (int r = 0; r < V; ++r)
            store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, abuf + ROWS * STR256, lane);
    }
    FM_SYNC();
    f32x16 kp[1][V], vp[2][V];
    {
        f32x16 acc3[3][V];
        zero_acc<3, V>(acc3);
        gemm_phase<V, 3, STR256>(abuf, abuf + ROWS * STR256, P.kv0.w + (long long)wave * P.kv0.KB * (3 * 2 * 64), P.kv0.KB,
                                 lane, acc3);
        finish_tile<V>(acc3[0], P.kv0.bias, wave * 32, P.kv0.inv_scale, false, lane);
        finish_tile<V>(acc3[1], P.kv0.bias, 128 + wave * 64, P.kv0.inv_scale, false, lane);
        finish_tile<V>(acc3[2], P.kv0.bias, 128 + wave * 64 + 32, P.kv0.inv_scale, false, lane);
#pragma unroll
        for (int r = 0; r < V; ++r) { kp[0][r] = acc3[0][r]; vp[0][r] = acc3[1][r]; vp[1][r] = acc3[2][r]; }
    }
    FM_SYNC();

    // ================= cross-view attention (cross_transformer.py:128-149) =================
    {
        float* kpb = reinterpret_cast<float*>(abuf);                 // [ROWS][KSTR]
        float* ksb = kpb + ROWS * KSTR;
#pragma unroll
        for (int r = 0; r < V; ++r) {
            store_tile_f(kp[0][r], r * 32 + myrow, wave * 32, kpb, lane);
            store_tile_f(ks[0][r], r * 32 + myrow, wave * 32, ksb, lane);
        }
        FM_SYNC();
        // A[j][i] = kp_j . ks_i / sqrt(128)
        for (int t = tid; t < 32 * V * V; t += 256) {
            int p = t & 31, ji = t >> 5, j = ji / V, i = ji % V;
            const float4* a = reinterpret_cast<const float4*>(kpb + (j * 32 + p) * KSTR);
            const float4* b = reinterpret_cast<const float4*>(ksb + (i * 32 + p) * KSTR);
            float s = 0.f;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                float4 x = a[c], y = b[c];
                s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
            }
            probs[ji * 32 + p] = s / 11.313708498984761f;
        }
        FM_SYNC();
        for (int t = tid; t < 32 * V; t += 256) {                    // softmax over j for each (sample, i)
            int p = t & 31, i = t >> 5;
            float m = -3.0e38f;
#pragma unroll
            for (int j = 0; j < V; ++j) m = fmaxf(m, probs[(j * V + i) * 32 + p]);
            float e[V], se = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) { e[j] = expf(probs[(j * V + i) * 32 + p] - m); se = se + e[j]; }
#pragma unroll
            for (int j = 0; j < V; ++j) probs[(j * V + i) * 32 + p] = e[j] / se;
        }
        FM_SYNC();
        float A[V][V];
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) A[j][i] = probs[(j * V + i) * 32 + myrow];
        // n_i = vs_i + sum_j vp_j A[j][i]  -> hi/lo planes (K = 256) for fc_1
        // (the key buffers were last read before the two barriers above -> ABUF is reusable; each
        // output tile is stored as soon as it is formed to keep accumulator->VGPR copies short-lived)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                f32x16 n;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float t = vp[c][0][e] * A[0][i];
#pragma unroll
                    for (int j = 1; j < V; ++j) t = t + vp[c][j][e] * A[j][i];
                    n[e] = vs[c][i][e] + t;
                }
                store_tile_h<STR256>(n, i * 32 + myrow, wave * 64 + c * 32, abuf, abuf + ROWS * STR256, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        FM_SYNC();
    }

    // ================= fc_1, fc_2 =================
    zero_acc<2, V>(acc2);
    gemm_phase<V, 2, STR256>(abuf, abuf + ROWS * STR256, P.fc_1.w + (long long)wave * P.fc_1.KB * (2 * 2 * 64), P.fc_1.KB,
                             lane, acc2);
    FM_SYNC();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        finish_tile<V>(acc2[c], P.fc_1.bias, wave * 64 + c * 32, P.fc_1.inv_scale, true, lane);
#pragma unroll
        for (int r = 0; r < V; ++r)
            store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, abuf + ROWS * STR256, lane);
    }
    FM_SYNC();
    zero_acc<2, V>(acc2);
    gemm_phase<V, 2, STR256>(abuf, abuf + ROWS * STR256, P.fc_2.w + (long long)wave * P.fc_2.KB * (2 * 2 * 64), P.fc_2.KB,
                             lane, acc2);
    FM_SYNC();
    // inter = relu(.) stays in registers (acc2) and goes to ABUF for feature_fc; its view mean -> MBUF
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        finish_tile<V>(acc2[c], P.fc_2.bias, wave * 64 + c * 32, P.fc_2.inv_scale, true, lane);
        f32x16 m;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float a = acc2[c][0][e];
#pragma unroll
            for (int r = 1; r < V; ++r) a = a + acc2[c][r][e];
            m[e] = a / (float)V;
        }
        store_tile_h<STR256>(m, myrow, wave * 64 + c * 32, mbuf, mbuf + 32 * STR256, lane);
#pragma unroll
        for (int r = 0; r < V; ++r)
            store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, abuf + ROWS * STR256, lane);
    }
    FM_SYNC();

    // ================= sigma head: relu(fc_3 m) . alpha_w + b =================
    {
        f32x16 a1[2][1];
        zero_acc<2, 1>(a1);
        gemm_phase<1, 2, STR256>(mbuf, mbuf + 32 * STR256, P.fc_3.w + (long long)wave * P.fc_3.KB * (2 * 2 * 64), P.fc_3.KB,
                                 lane, a1);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            finish_tile<1>(a1[c], P.fc_3.bias, wave * 64 + c * 32, P.fc_3.inv_scale, true, lane);
#pragma unroll
            for (int e = 0; e < 16; ++e) s = fmaf(a1[c][0][e], P.alpha_w[wave * 64 + c * 32 + acc_chan(e, lane)], s);
        }
        s += __shfl_xor(s, 32);
        if (lane < 32) part[(wave * 32 + lane) * 4] = s;
        FM_SYNC();
        if (tid < 32) {
            float t = part[tid * 4] + part[(32 + tid) * 4] + part[(64 + tid) * 4] + part[(96 + tid) * 4] + P.alpha_b[0];
            sig[tid] = t;
        }
        if (tid == 0) *flag = 0;
        FM_SYNC();
        if (tid < npts && (P.rgb_all || sig[tid] > 0.f)) *flag = 1;
        FM_SYNC();
    }
    const bool need_rgb = *flag != 0;
    float rgb_out[3] = {0.f, 0.f, 0.f};
    if (need_rgb) {
        // ================= RGB branch (cross_transformer.py:330-353) =================
        // feat = feature_fc(inter) + rgb_res_0(f)   (one accumulator: both layers share a scale)
        f32x16 r1[1][V];
        zero_acc<2, V>(acc2);
        zero_acc<1, V>(r1);
        gemm_phase<V, 2, STR256>(abuf, abuf + ROWS * STR256, P.feat.w + (long long)wave * P.feat.KB * (2 * 2 * 64),
                                 P.feat.KB, lane, acc2);
        FM_SYNC();
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            stage_rows<V, 192, STR192>(P.f, 384, half * 192, pbase, npts, abuf, abuf + ROWS * STR192, tid);
            FM_SYNC();
            gemm_phase<V, 2, STR192>(abuf, abuf + ROWS * STR192,
                                     P.rr0.w + ((long long)wave * P.rr0.KB + half * 12) * (2 * 2 * 64), 12, lane, acc2);
            gemm_phase<V, 1, STR192>(abuf, abuf + ROWS * STR192,
                                     P.rr1.w + ((long long)wave * P.rr1.KB + half * 12) * (1 * 2 * 64), 12, lane, r1);
            FM_SYNC();
        }
        // feat (+ both biases) | viewdir -> [ROWS][288]
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            finish_tile<V>(acc2[c], P.feat.bias, wave * 64 + c * 32, P.feat.inv_scale, false, lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 b = *reinterpret_cast<const float4*>(P.rr0.bias + wave * 64 + c * 32 + 8 * g + 4 * (lane >> 5));
#pragma unroll
                for (int r = 0; r < V; ++r) {
                    acc2[c][r][4 * g] += b.x; acc2[c][r][4 * g + 1] += b.y;
                    acc2[c][r][4 * g + 2] += b.z; acc2[c][r][4 * g + 3] += b.w;
                }
            }
#pragma unroll
            for (int r = 0; r < V; ++r)
                store_tile_h<STR288>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, abuf + ROWS * STR288, lane);
        }
        for (int i = tid; i < ROWS * 32; i += 256) {
            int row = i >> 5, c = i & 31;
            float x = (c < 27) ? vds[(row & 31) * 28 + c] : 0.f;
            _Float16 a, b;
            split_h(x, a, b);
            *reinterpret_cast<_Float16*>(abuf + row * STR288 + 2 * (256 + c)) = a;
            *reinterpret_cast<_Float16*>(abuf + ROWS * STR288 + row * STR288 + 2 * (256 + c)) = b;
        }
        FM_SYNC();
        f32x16 vf[1][V];
        zero_acc<1, V>(vf);
        gemm_phase<V, 1, STR288>(abuf, abuf + ROWS * STR288, P.vfc.w + (long long)wave * P.vfc.KB * (1 * 2 * 64), P.vfc.KB,
                                 lane, vf);
        finish_tile<V>(vf[0], P.vfc.bias, wave * 32, P.vfc.inv_scale, true, lane);
        finish_tile<V>(r1[0], P.rr1.bias, wave * 32, P.rr1.inv_scale, false, lane);
        {
            f32x16 m;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float a = vf[0][0][e] + r1[0][0][e];
#pragma unroll
                for (int r = 1; r < V; ++r) a = a + (vf[0][r][e] + r1[0][r][e]);
                m[e] = a / (float)V;
            }
            store_tile_h<STR128>(m, myrow, wave * 32, mbuf, mbuf + 32 * STR128, lane);
        }
        FM_SYNC();
        f32x16 a4[1][1];
        zero_acc<1, 1>(a4);
        gemm_phase<1, 1, STR128>(mbuf, mbuf + 32 * STR128, P.fc_4.w + (long long)wave * P.fc_4.KB * (1 * 2 * 64), P.fc_4.KB,
                                 lane, a4);
        finish_tile<1>(a4[0], P.fc_4.bias, wave * 32, P.fc_4.inv_scale, true, lane);
        float s3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int ch = wave * 32 + acc_chan(e, lane);
#pragma unroll
            for (int o = 0; o < 3; ++o) s3[o] = fmaf(a4[0][0][e], P.rgb_w[o * 128 + ch], s3[o]);
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) s3[o] += __shfl_xor(s3[o], 32);
        if (lane < 32) {
            part[(wave * 32 + lane) * 4 + 0] = s3[0];
            part[(wave * 32 + lane) * 4 + 1] = s3[1];
            part[(wave * 32 + lane) * 4 + 2] = s3[2];
        }
        FM_SYNC();
        if (tid < 32) {
#pragma unroll
            for (int o = 0; o < 3; ++o)
                rgb_out[o] = part[tid * 4 + o] + part[(32 + tid) * 4 + o] + part[(64 + tid) * 4 + o] +
                             part[(96 + tid) * 4 + o] + P.rgb_b[o];
        }
    }
    if (tid < npts)
        *reinterpret_cast<float4*>(P.raw_c + (long long)(pbase + tid) * 4) =
            make_float4(rgb_out[0], rgb_out[1], rgb_out[2], sig[tid]);
}

// ---- host: packing -------------------------------------------------------------------
// cols[] gives, for every (wave, ct), the first output column of that 32-wide tile.
__global__ void pack_fused_kernel(const float* __restrict__ W, int N, int K, int KB, int CT, const int* __restrict__ cols,
                                  float scale, uint4* __restrict__ out) {
    long long total = 4LL * KB * CT * 2 * 64;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        int lane = (int)(o & 63);
        long long q = o >> 6;
        int plane = (int)(q & 1); q >>= 1;
        int ct = (int)(q % CT); q /= CT;
        int kb = (int)(q % KB);
        int wave = (int)(q / KB);
        int col = cols[wave * CT + ct] + (lane & 31);
        int k0 = 16 * kb + 8 * (lane >> 5);
        h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int k = k0 + j;
            float x = (col < N && k < K) ? W[(long long)col * K + k] * scale : 0.f;
            _Float16 hi, lo;
            split_h(x, hi, lo);
            v[j] = plane == 0 ? hi : lo;
        }
        out[o] = *reinterpret_cast<uint4*>(&v);
    }
}

__global__ void absmax_kernel(const float* __restrict__ w, long long n, unsigned int* __restrict__ out) {
    float m = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

size_t th_fused_pack_bytes() {
    // every layer, both planes, K padded to 16 (+ biases / column tables / scratch in a 64 KiB tail)
    size_t halves = 0;
    const int dims[][2] = {{256, 256}, {384, 256}, {256, 384}, {384, 256}, {256, 256}, {256, 256}, {256, 256},
                           {256, 256}, {256, 384}, {128, 288}, {128, 384}, {128, 128}};
    for (auto& d : dims) halves += (size_t)d[0] * d[1] * 2;
    return th_align(halves * 2) + 16 * 256 + 64 * 1024;
}

// power-of-two scale putting max|W| into [2^12, 2^13): hi fits fp16, lo stays a normal number
static int layer_scale_log2(const float* w, long long n, unsigned int* amax_dev, int* out, hipStream_t s) {
    TH_HIP(hipMemsetAsync(amax_dev, 0, 4, s));
    hipLaunchKernelGGL(absmax_kernel, dim3(64), dim3(256), 0, s, w, n, amax_dev);
    unsigned int bits = 0;
    TH_HIP(hipMemcpyAsync(&bits, amax_dev, 4, hipMemcpyDeviceToHost, s));
    TH_HIP(hipStreamSynchronize(s));
    float amax;
    memcpy(&amax, &bits, 4);
    if (!(amax > 0.f) || !(amax < 3.0e38f)) amax = 1.f;
    int e;
    frexpf(amax, &e);            // amax = m * 2^e, m in [0.5,1)
    *out = 13 - e;
    return 0;
}

struct PackCursor {
    char* w;        // packed halves
    float* bias;    // fp32 biases
    int* cols;      // device column tables (16 ints per layer)
};

static int pack_layer(const float* w, const float* b, int N, int K, int CT, const int* cols_host, int sl2,
                      PackCursor& cur, FusedLayer* out, hipStream_t s) {
    const int KB = (K + 15) / 16;
    if (b) TH_HIP(hipMemcpyAsync(cur.bias, b, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    else TH_HIP(hipMemsetAsync(cur.bias, 0, (size_t)N * 4, s));
    TH_HIP(hipMemcpyAsync(cur.cols, cols_host, 4 * CT * sizeof(int), hipMemcpyHostToDevice, s));
    uint4* dst = (uint4*)cur.w;
    long long total = 4LL * KB * CT * 2 * 64;
    hipLaunchKernelGGL(pack_fused_kernel, dim3(256), dim3(256), 0, s, w, N, K, KB, CT, cur.cols, ldexpf(1.f, sl2), dst);
    TH_HIP(hipStreamSynchronize(s));   // cols_host is a caller stack array
    out->w = dst;
    out->bias = cur.bias;
    out->inv_scale = ldexpf(1.f, -sl2);
    out->CT = CT;
    out->KB = KB;
    cur.w += th_align((size_t)total * 16);
    cur.bias += (N + 3) & ~3;
    cur.cols += 16;
    return 0;
}

// Builds the fused image from the fp32 layers.  `store` = th_fused_pack_bytes() of device memory.
int th_fused_pack(const th_mlp_weights* w, void* store, FusedParams* out, hipStream_t s) {
    char* tail = (char*)store + th_fused_pack_bytes() - 64 * 1024;
    PackCursor cur{(char*)store, (float*)tail, (int*)(tail + 40 * 1024)};
    unsigned int* amax = (unsigned int*)(tail + 60 * 1024);
    int c256[8], c128[4], ckv[12];
    for (int wv = 0; wv < 4; ++wv) {
        c256[wv * 2] = wv * 64; c256[wv * 2 + 1] = wv * 64 + 32;
        c128[wv] = wv * 32;
        // stacked [key(128); value(256)]: tile 0 = key cols wave*32, tiles 1,2 = value cols 128 + wave*64 (+32)
        ckv[wv * 3] = wv * 32; ckv[wv * 3 + 1] = 128 + wv * 64; ckv[wv * 3 + 2] = 128 + wv * 64 + 32;
    }
    int sl2, sl2b;
#define PACK_SIMPLE(LAYER, L, N_, K_, CT_, COLS)                                       \
    TH_TRY(layer_scale_log2((L).w, (long long)(N_) * (K_), amax, &sl2, s));            \
    TH_TRY(pack_layer((L).w, (L).b, N_, K_, CT_, COLS, sl2, cur, &out->LAYER, s))
    PACK_SIMPLE(fc_0, w->fc_0, 256, 255, 2, c256);
    PACK_SIMPLE(ar0, w->alpha_res_0, 256, 384, 2, c256);
    PACK_SIMPLE(fc_1, w->fc_1, 256, 256, 2, c256);
    PACK_SIMPLE(fc_2, w->fc_2, 256, 256, 2, c256);
    PACK_SIMPLE(fc_3, w->fc_3, 256, 256, 2, c256);
    PACK_SIMPLE(vfc, w->view_fc, 128, 283, 1, c128);
    PACK_SIMPLE(rr1, w->rgb_res_1, 128, 384, 1, c128);
    PACK_SIMPLE(fc_4, w->fc_4, 128, 128, 1, c128);
#undef PACK_SIMPLE
    // feature_fc and rgb_res_0 accumulate into ONE register tile -> they must share a scale
    TH_TRY(layer_scale_log2(w->feature_fc.w, 256LL * 256, amax, &sl2, s));
    TH_TRY(layer_scale_log2(w->rgb_res_0.w, 256LL * 384, amax, &sl2b, s));
    if (sl2b < sl2) sl2 = sl2b;
    TH_TRY(pack_layer(w->feature_fc.w, w->feature_fc.b, 256, 256, 2, c256, sl2, cur, &out->feat, s));
    TH_TRY(pack_layer(w->rgb_res_0.w, w->rgb_res_0.b, 256, 384, 2, c256, sl2, cur, &out->rr0, s));
    // stacked key/value layers need a contiguous [384,256] weight + [384] bias
    float* tw = nullptr;
    TH_HIP(hipMalloc((void**)&tw, (size_t)(384 * 256 + 384) * 4));
    float* tb = tw + 384 * 256;
    for (int which = 0; which < 2; ++which) {
        const th_linear& k = which == 0 ? w->key1 : w->key0;
        const th_linear& v = which == 0 ? w->val1 : w->val0;
        TH_HIP(hipMemcpyAsync(tw, k.w, 128 * 256 * 4, hipMemcpyDeviceToDevice, s));
        TH_HIP(hipMemcpyAsync(tw + 128 * 256, v.w, 256 * 256 * 4, hipMemcpyDeviceToDevice, s));
        if (k.b) TH_HIP(hipMemcpyAsync(tb, k.b, 128 * 4, hipMemcpyDeviceToDevice, s));
        else TH_HIP(hipMemsetAsync(tb, 0, 128 * 4, s));
        if (v.b) TH_HIP(hipMemcpyAsync(tb + 128, v.b, 256 * 4, hipMemcpyDeviceToDevice, s));
        else TH_HIP(hipMemsetAsync(tb + 128, 0, 256 * 4, s));
        TH_TRY(layer_scale_log2(tw, 384LL * 256, amax, &sl2, s));
        TH_TRY(pack_layer(tw, tb, 384, 256, 3, ckv, sl2, cur, which == 0 ? &out->kv1 : &out->kv0, s));
    }
    TH_HIP(hipStreamSynchronize(s));
    TH_HIP(hipFree(tw));
    TH_REQUIRE(cur.w <= tail, "fused pack overflow");
    TH_REQUIRE((char*)cur.bias <= tail + 40 * 1024 && (char*)cur.cols <= tail + 60 * 1024, "fused pack tail overflow");
    return 0;
}

int th_mlp_fused_forward(const FusedParams& base, const ThMlpPacked& heads, int V, int P, const float* h, const float* f,
                         const float* vd, int rgb_all, float* raw_c, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE(V >= 1 && V <= 3, "fused MLP supports 1..3 reference views");
    FusedParams p = base;
    p.alpha_w = heads.alpha_w; p.alpha_b = heads.alpha_b; p.rgb_w = heads.rgb_w; p.rgb_b = heads.rgb_b;
    p.h = h; p.f = f; p.vd = vd; p.raw_c = raw_c; p.P = P; p.rgb_all = rgb_all;
    static bool attr = false;
    if (!attr) {
        TH_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
        TH_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
        TH_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_BYTES));
        attr = true;
    }
    dim3 grid(th_cdiv(P, FM_PTS));
    // developer experiment (timing only, results are wrong): alias every layer's weights onto fc_1's image
    // so the weight working set is 256 KB -> shows how much of a phase is L2-capacity/latency
    static int alias_w = getenv("TH_FUSED_ALIAS_W") ? 1 : 0;
    if (alias_w) {
        FusedLayer* ls[] = {&p.fc_0, &p.kv1, &p.ar0, &p.kv0, &p.fc_2, &p.fc_3, &p.feat, &p.rr0, &p.vfc, &p.rr1, &p.fc_4};
        for (auto* l : ls) l->w = p.fc_1.w;
    }
    // developer aid: TH_FUSED_DBG=1 -> cycle stamps of the middle tile after every barrier (first big launch only)
    static int dbg_state = getenv("TH_FUSED_DBG") ? 1 : 0;
    static long long* dbg_dev = nullptr;
    p.dbg = nullptr;
    const bool dbg_now = dbg_state == 1 && P >= 4096;
    if (dbg_now) {
        if (!dbg_dev) TH_HIP(hipMalloc((void**)&dbg_dev, 64 * sizeof(long long)));
        TH_HIP(hipMemsetAsync(dbg_dev, 0, 64 * sizeof(long long), s));
        p.dbg = dbg_dev;
    }
    switch (V) {
        case 1: hipLaunchKernelGGL(mlp_fused_kernel<1>, grid, dim3(256), FUSED_LDS_BYTES, s, p); break;
        case 2: hipLaunchKernelGGL(mlp_fused_kernel<2>, grid, dim3(256), FUSED_LDS_BYTES, s, p); break;
        default: hipLaunchKernelGGL(mlp_fused_kernel<3>, grid, dim3(256), FUSED_LDS_BYTES, s, p); break;
    }
    TH_LAUNCH_CHECK();
    if (dbg_now) {
        long long st[64];
        TH_HIP(hipStreamSynchronize(s));
        TH_HIP(hipMemcpy(st, dbg_dev, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "[TH_FUSED_DBG] tile %d of %d, cycles between barriers:", grid.x / 2, grid.x);
        for (int i = 1; i < 64 && st[i] != 0; ++i) fprintf(stderr, " %lld", st[i] - st[i - 1]);
        fprintf(stderr, "\n");
        dbg_state = 2;
    }
    return 0;
}
