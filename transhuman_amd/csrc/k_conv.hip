// K12: the convolutions of the ResNet stem of SpatialEncoder (encoder.py:114-126; torchvision resnet18: conv1 7x7/2,
// layer1: four 3x3 64->64, layer2: 3x3/2 64->128, three 3x3 128->128, 1x1/2 64->128 downsample; all bias-free).
//
// Implicit GEMM on v_mfma_f32_32x32x16_f16 with both operands split into fp16 hi + lo (three products, fp32
// accumulate: the fp32-class scheme of the fused MLP): out^T = W * patches^T, one MFMA row tile =
// 32 consecutive output pixels of one image row, K runs over (tap, 16 input channels).
//   * the input tile (+ halo) of a workgroup is staged ONCE into LDS as fp16 hi / lo planes [pixel][CIN] (row stride
//     2 CIN + 16 B: conflict-free ds_read_b128 fragments), converted on the way in from the NCHW fp32 tensor (loads
//     coalesced along x, four channels per thread -> 8-byte LDS stores); every tap reads it at a shifted pixel offset;
//   * weights are pre-packed (th_conv_pack) into the exact per-lane A fragments [tap][k-block][col tile][hi|lo][lane],
//     pre-scaled by a power of two so that their lo halves stay normal fp16 numbers, and stream from L2 into registers;
//   * a lane's accumulators are 16 output channels of ONE pixel, lanes 0..31 are 32 consecutive x: every store
//     instruction writes two 128-byte runs of the NCHW output.
// Tile shapes: COUT = 64: 4 rows x 32 pixels per workgroup, waves = 2 row pairs x 2 column tiles (each: 2 row tiles x 1
// column tile); COUT = 128: 2 rows x 32 pixels, the 4 waves split the output channels (each: 2 row tiles x 1 column tile);
// two workgroups fit a CU (LDS 59 / 74 KB), so one stages while the other multiplies.
#include <string.h>

#include "th_internal.h"

typedef float cv_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 cv_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void cv_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
// range guard of the split input (th_range_read, slot TH_RANGE_CONV): running maximum of |x| as fp32 bit patterns
// (inf / NaN order above every finite value); merged into the launch-wide table with an atomic only when it grows
__device__ __forceinline__ void cv_range_acc(unsigned& rm, float x) { rm = max(rm, __float_as_uint(x) & 0x7fffffffu); }
__device__ __forceinline__ void cv_range_commit(unsigned* __restrict__ table, unsigned rm) {
    if (table != nullptr && rm > table[TH_RANGE_CONV]) atomicMax(table + TH_RANGE_CONV, rm);
}

// ---- weight packing ---------------------------------------------------------------------------------------------
__global__ void conv_absmax_kernel(const float* __restrict__ w, long long n, unsigned int* __restrict__ out) {
    float m = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// W [COUT][CIN][KS][KS] -> uint4 [tap][kb][ct][plane][lane]: lane l holds channel ct*32 + (l & 31),
// input channels kb*16 + 8 (l >> 5) .. +7 of tap (kh, kw); K (= CIN) zero-padded to a multiple of 16
__global__ void conv_pack_kernel(const float* __restrict__ W, int COUT, int CIN, int KS, int KB, int CT, float scale,
                                 uint4* __restrict__ out) {
    const long long total = (long long)KS * KS * KB * CT * 2 * 64;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(o & 63);
        long long q = o >> 6;
        const int plane = (int)(q & 1); q >>= 1;
        const int ct = (int)(q % CT); q /= CT;
        const int kb = (int)(q % KB);
        const int tap = (int)(q / KB);
        const int co = ct * 32 + (lane & 31);
        const int ci0 = kb * 16 + 8 * (lane >> 5);
        cv_h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = ci0 + j;
            const float x = (co < COUT && ci < CIN) ? W[((long long)co * CIN + ci) * KS * KS + tap] * scale : 0.f;
            _Float16 hi, lo;
            cv_split(x, hi, lo);
            v[j] = plane == 0 ? hi : lo;
        }
        out[o] = *reinterpret_cast<uint4*>(&v);
    }
}

size_t th_conv_pack_size(int COUT, int CIN, int KS) {
    if (CIN == 3 && KS == 7) { CIN = 3 * 49; KS = 1; }
    const int KB = (CIN + 15) / 16, CT = (COUT + 31) / 32;
    return th_align((size_t)KS * KS * KB * CT * 2 * 64 * sizeof(uint4)) + 256;
}

int th_conv_pack_launch(const float* w, int COUT, int CIN, int KS, void* out, size_t out_bytes, float* inv_scale_host,
                        hipStream_t s) {
    TH_REQUIRE(out_bytes >= th_conv_pack_size(COUT, CIN, KS), "pack buffer too small");
    if (CIN == 3 && KS == 7) { CIN = 3 * 49; KS = 1; }      // conv1: im2col form, K = (c, kh, kw) flattened
    const int KB = (CIN + 15) / 16, CT = (COUT + 31) / 32;
    unsigned int* amax_dev = (unsigned int*)((char*)out + th_conv_pack_size(COUT, CIN, KS) - 256);
    TH_HIP(hipMemsetAsync(amax_dev, 0, 4, s));
    const long long n = (long long)COUT * CIN * KS * KS;
    hipLaunchKernelGGL(conv_absmax_kernel, dim3(64), dim3(256), 0, s, w, n, amax_dev);
    unsigned int bits = 0;
    TH_HIP(hipMemcpyAsync(&bits, amax_dev, 4, hipMemcpyDeviceToHost, s));
    TH_HIP(hipStreamSynchronize(s));
    float amax;
    memcpy(&amax, &bits, 4);
    if (!(amax > 0.f) || !(amax < 3.0e38f)) amax = 1.f;
    int e;
    frexpf(amax, &e);                       // amax = m 2^e, m in [0.5, 1)
    const int sl2 = 13 - e;                 // max |W| 2^sl2 in [2^12, 2^13): hi fits fp16, lo stays a normal number
    hipLaunchKernelGGL(conv_pack_kernel, dim3(256), dim3(256), 0, s, w, COUT, CIN, KS, KB, CT, ldexpf(1.f, sl2),
                       (uint4*)out);
    TH_LAUNCH_CHECK();
    *inv_scale_host = ldexpf(1.f, -sl2);
    return 0;
}

// ---- the convolution ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_chan(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

template <int CIN, int COUT, int S, int KS, int TR, int WC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                        float inv_scale, float* __restrict__ y, int H, int W, int Ho,
                                                        int Wo, unsigned* __restrict__ range,
                                                        float2* __restrict__ stats, int NP) {
    constexpr int PAD = KS / 2;
    constexpr int IR = (TR - 1) * S + KS, IC = 31 * S + KS;
    constexpr int STRB = 2 * CIN + 16;
    constexpr int PLANE = IR * IC * STRB;
    constexpr int KB = CIN / 16, CT = COUT / 32;
    constexpr int WR = 4 / WC;                             // the 4 waves: WR along the rows x WC along the column tiles
    constexpr int RTW = TR / WR;                           // row tiles per wave
    constexpr int CTW = CT / WC;                           // column tiles per wave (waves that share a column tile load
                                                           // the same weight fragments: L1 traffic, keep WC large)
    static_assert(CIN % 16 == 0 && COUT % 32 == 0, "channel counts");
    static_assert(WC * WR == 4 && TR % WR == 0 && CT % WC == 0, "wave split");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* xhi = lds;
    char* xlo = lds + PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.z, oy0 = blockIdx.y * TR, ox0 = blockIdx.x * 32;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;

    // ---- stage the input tile: fp32 NCHW -> fp16 hi / lo planes [pixel][CIN] ----
    // batches of SB items per thread: all 4 SB loads of a batch are in flight before the first conversion (one
    // workgroup or two per CU: nothing else hides the global-load latency)
    {
        constexpr int ITEMS = IR * IC * (CIN / 4);
        constexpr int SB = 6;
        const float* xn = x + (long long)n * CIN * H * W;
        const long long HW = (long long)H * W;
        unsigned rmax = 0u;
        for (int it0 = tid; it0 < ITEMS; it0 += 256 * SB) {
            float v[SB][4];
            int off[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const int it = it0 + 256 * b;
                const int px = it % IC, t2 = it / IC, r = t2 % IR, c4 = t2 / IR;
                const int gy = iy0 + r, gx = ix0 + px;
                off[b] = it < ITEMS ? (r * IC + px) * STRB + 8 * c4 : -1;
                const bool in = it < ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const float* p = xn + ((long long)(4 * c4) * H + gy) * W + gx;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[b][e] = in ? p[e * HW] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                cv_h4 a, bl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 hi, lo;
                    cv_split(v[b][e], hi, lo);
                    cv_range_acc(rmax, v[b][e]);
                    a[e] = hi; bl[e] = lo;
                }
                if (off[b] >= 0) {
                    *reinterpret_cast<cv_h4*>(xhi + off[b]) = a;
                    *reinterpret_cast<cv_h4*>(xlo + off[b]) = bl;
                }
            }
        }
        cv_range_commit(range, rmax);
    }
    __syncthreads();

    const int rbase = (wave / WC) * RTW;
    const int cbase = (wave % WC) * CTW;
    cv_f32x16 acc[CTW][RTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][r][e] = 0.f;

    // K loop over (tap, k-block).  A step is only 6 MFMAs (192 cycles): the weight fragments (L2, ~1-2 k cycles away)
    // run DW - 1 steps ahead in a register ring, the activation fragments (LDS) one step ahead.
    constexpr int STEPS = KS * KS * KB;
    constexpr int DW = 6;
    static_assert((DW & 1) == 0, "ring depth must be even (activation ping-pong parity)");
    const uint4* wl = wp + (long long)cbase * 2 * 64 + lane;
    const int lane_off = (lane & 31) * S * STRB + (lane >> 5) * 16;
    uint4 wq[DW][CTW][2];
    cv_h8 xh[2][RTW], xl[2][RTW];
    auto load_w = [&](int st, int slot) {
        const uint4* p = wl + (long long)st * (CT * 2 * 64);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            wq[slot][c][0] = p[(c * 2 + 0) * 64];
            wq[slot][c][1] = p[(c * 2 + 1) * 64];
        }
    };
    auto load_x = [&](int st, int buf) {
        const int tap = st / KB, kb = st - tap * KB;
        const int kh = tap / KS, kw = tap - kh * KS;
#pragma unroll
        for (int r = 0; r < RTW; ++r) {
            const int off = (((rbase + r) * S + kh) * IC + kw) * STRB + lane_off + kb * 32;
            xh[buf][r] = *reinterpret_cast<const cv_h8*>(xhi + off);
            xl[buf][r] = *reinterpret_cast<const cv_h8*>(xlo + off);
        }
    };
#pragma unroll
    for (int j = 0; j < DW - 1; ++j)
        if (j < STEPS) load_w(j, j);
    load_x(0, 0);
#pragma unroll 1
    for (int st0 = 0; st0 < STEPS; st0 += DW) {
#pragma unroll
        for (int j = 0; j < DW; ++j) {
            const int st = st0 + j;
            if (st < STEPS) {
                if (st + DW - 1 < STEPS) load_w(st + DW - 1, (j + DW - 1) % DW);
                if (st + 1 < STEPS) load_x(st + 1, (j + 1) & 1);
                const int cur = j & 1;
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int r = 0; r < RTW; ++r)
                        acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const cv_h8*>(&wq[j][c][1]),
                                                                           xh[cur][r], acc[c][r], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int r = 0; r < RTW; ++r)
                        acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const cv_h8*>(&wq[j][c][0]),
                                                                           xl[cur][r], acc[c][r], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < CTW; ++c)
#pragma unroll
                    for (int r = 0; r < RTW; ++r)
                        acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const cv_h8*>(&wq[j][c][0]),
                                                                           xh[cur][r], acc[c][r], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: NCHW fp32 ----
    const int ox = ox0 + (lane & 31);
    if (ox < Wo) {
        float* yn = y + (long long)n * COUT * Ho * Wo;
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int r = 0; r < RTW; ++r) {
                const int oy = oy0 + rbase + r;
                if (oy < Ho) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int co = (cbase + c) * 32 + cv_chan(e, lane);
                        yn[((long long)co * Ho + oy) * Wo + ox] = acc[c][r][e] * inv_scale;
                    }
                }
            }
    }
    // ---- BatchNorm statistics of the tile (th_conv2d_stats): per output channel the sum and the sum of squares of the
    // values just stored, over this wave's pixels, as one float2 partial per (channel, wave-tile) -- K11's apply pass adds
    // the partials of a channel in float64 in a fixed order.  Saves the statistics pass its launch and its read of y.
    if (stats != nullptr) {
        const int pidx = ((n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * WR + wave / WC;
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float sm = 0.f, sq = 0.f;
#pragma unroll
                for (int r = 0; r < RTW; ++r) {
                    const float v = (ox < Wo && oy0 + rbase + r < Ho) ? acc[c][r][e] * inv_scale : 0.f;
                    sm += v;
                    sq = fmaf(v, v, sq);
                }
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
                if ((lane & 31) == 0) stats[(long long)((cbase + c) * 32 + cv_chan(e, lane)) * NP + pidx] = make_float2(sm, sq);
            }
    }
}

template <int CIN, int COUT, int S, int KS, int TR, int WC>
static int conv_launch_t(const float* x, const uint4* wp, float inv_scale, float* y, int N, int H, int W, int Ho, int Wo,
                         unsigned* range, float2* stats, int* np_out,
                         hipStream_t s) {
    constexpr int IR = (TR - 1) * S + KS, IC = 31 * S + KS, STRB = 2 * CIN + 16;
    constexpr size_t lds = (size_t)2 * IR * IC * STRB;
    static_assert(lds <= 160 * 1024, "input tile does not fit in LDS");
    auto kern = conv_mfma_kernel<CIN, COUT, S, KS, TR, WC>;
    static unsigned long long attr_done = 0ull;
    if (th_lds_attr_needed(&attr_done))
        TH_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(th_cdiv(Wo, 32), th_cdiv(Ho, TR), N);
    const int NP = (int)(grid.x * grid.y * grid.z) * (4 / WC);
    if (np_out) *np_out = NP;
    if (x == nullptr) return 0;                            // (size query)
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, wp, inv_scale, y, H, W, Ho, Wo, range, stats, NP);
    TH_LAUNCH_CHECK();
    return 0;
}

// ---- conv1: 7x7 / 2, 3 -> 64 (K = 147: too few input channels for the per-tap form) ----------------------------------
// The workgroup (2 output rows x 32 pixels) stages its raw fp32 input patch (9 x 69 x 3) in LDS, expands it there into
// the im2col operand [pixel][k = (c*7 + kh)*7 + kw, padded to 160] as fp16 hi / lo planes, and runs ONE 10-k-block GEMM
// (the weight [64][3][7][7] is already [64][147] row-major: packed by conv_pack_kernel as a 1-tap layer with CIN = 147).
// Wave w owns output row (w & 1) and column tile (w >> 1) of the tile; three workgroups fit a CU.
#define C1_TR 2
#define C1_K 147
#define C1_KB 10
#define C1_STRB (2 * 16 * C1_KB + 16)
#define C1_IR ((C1_TR - 1) * 2 + 7)
#define C1_IC (31 * 2 + 7)

__global__ __launch_bounds__(256) void conv1_mfma_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                         float inv_scale, float* __restrict__ y, int H, int W, int Ho,
                                                         int Wo, unsigned* __restrict__ range,
                                                         float2* __restrict__ stats, int NP) {
    constexpr int PLANE = C1_TR * 32 * C1_STRB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* xhi = lds;
    char* xlo = lds + PLANE;
    float* raw = reinterpret_cast<float*>(lds + 2 * PLANE);             // [3][C1_IR][C1_IC]
    int* koff = reinterpret_cast<int*>(raw + 3 * C1_IR * C1_IC);          // [160] raw offset of tap k (-1: padding)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.z, oy0 = blockIdx.y * C1_TR, ox0 = blockIdx.x * 32;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    const float* xn = x + (long long)n * 3 * H * W;
    {
        constexpr int RAW = 3 * C1_IR * C1_IC, RB = (RAW + 255) / 256;
        float rv[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {                                    // all loads in flight, then the LDS stores
            const int it = tid + 256 * b;
            const int px = it % C1_IC, t2 = it / C1_IC, r = t2 % C1_IR, c = t2 / C1_IR;
            const int gy = iy0 + r, gx = ix0 + px;
            rv[b] = (it < RAW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? xn[((long long)c * H + gy) * W + gx] : 0.f;
        }
        unsigned rmax = 0u;
#pragma unroll
        for (int b = 0; b < RB; ++b)
            if (tid + 256 * b < RAW) { raw[tid + 256 * b] = rv[b]; cv_range_acc(rmax, rv[b]); }
        cv_range_commit(range, rmax);
    }
    if (tid < 16 * C1_KB) {
        const int c = tid / 49, rem = tid - c * 49, kh = rem / 7, kw = rem - kh * 7;
        koff[tid] = tid < C1_K ? (c * C1_IR + kh) * C1_IC + kw : -1;
    }
    __syncthreads();
    for (int it = tid; it < C1_TR * 32 * (4 * C1_KB); it += 256) {       // (pixel, group of 4 consecutive k)
        const int g = it % (4 * C1_KB), p = it / (4 * C1_KB);
        const int base = ((p >> 5) * 2) * C1_IC + (p & 31) * 2;
        cv_h4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = koff[4 * g + e];
            const float v = o >= 0 ? raw[o + base] : 0.f;
            _Float16 hi, lo;
            cv_split(v, hi, lo);
            a[e] = hi; b[e] = lo;
        }
        *reinterpret_cast<cv_h4*>(xhi + p * C1_STRB + 8 * g) = a;
        *reinterpret_cast<cv_h4*>(xlo + p * C1_STRB + 8 * g) = b;
    }
    __syncthreads();
    cv_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int rt = wave & 1, ct = wave >> 1;
    const uint4* wl = wp + lane;
    const int xoff = (rt * 32 + (lane & 31)) * C1_STRB + (lane >> 5) * 16;
    uint4 wh[C1_KB], wlo[C1_KB];
#pragma unroll
    for (int kb = 0; kb < C1_KB; ++kb) {
        wh[kb] = wl[((kb * 2 + ct) * 2 + 0) * 64];
        wlo[kb] = wl[((kb * 2 + ct) * 2 + 1) * 64];
    }
#pragma unroll
    for (int kb = 0; kb < C1_KB; ++kb) {
        const cv_h8 xh = *reinterpret_cast<const cv_h8*>(xhi + xoff + kb * 32);
        const cv_h8 xl = *reinterpret_cast<const cv_h8*>(xlo + xoff + kb * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const cv_h8*>(&wlo[kb]), xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const cv_h8*>(&wh[kb]), xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const cv_h8*>(&wh[kb]), xh, acc, 0, 0, 0);
    }
    const int ox = ox0 + (lane & 31), oy = oy0 + rt;
    if (ox < Wo && oy < Ho) {
        float* yn = y + (long long)n * 64 * Ho * Wo;
#pragma unroll
        for (int e = 0; e < 16; ++e) yn[((long long)(ct * 32 + cv_chan(e, lane)) * Ho + oy) * Wo + ox] = acc[e] * inv_scale;
    }
    if (stats != nullptr) {                                              // (see conv_mfma_kernel)
        const int pidx = ((n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * C1_TR + rt;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = (ox < Wo && oy < Ho) ? acc[e] * inv_scale : 0.f;
            float sm = v, sq = v * v;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { sm += __shfl_xor(sm, o); sq += __shfl_xor(sq, o); }
            if ((lane & 31) == 0) stats[(long long)(ct * 32 + cv_chan(e, lane)) * NP + pidx] = make_float2(sm, sq);
        }
    }
}

static int conv1_launch(const float* x, const uint4* wp, float inv_scale, float* y, int N, int H, int W, int Ho, int Wo,
                        unsigned* range, float2* stats, int* np_out,
                        hipStream_t s) {
    constexpr size_t lds = (size_t)2 * C1_TR * 32 * C1_STRB + 3 * C1_IR * C1_IC * 4 + 16 * C1_KB * 4;
    static unsigned long long attr_done = 0ull;
    if (th_lds_attr_needed(&attr_done))
        TH_HIP(hipFuncSetAttribute((const void*)conv1_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(th_cdiv(Wo, 32), th_cdiv(Ho, C1_TR), N);
    const int NP = (int)(grid.x * grid.y * grid.z) * C1_TR;
    if (np_out) *np_out = NP;
    if (x == nullptr) return 0;
    hipLaunchKernelGGL(conv1_mfma_kernel, grid, dim3(256), lds, s, x, wp, inv_scale, y, H, W, Ho, Wo, range, stats, NP);
    TH_LAUNCH_CHECK();
    return 0;
}

// ---- max pooling 3x3 / 2, padding 1 (resnet.maxpool), NCHW ---------------------------------------------------------------
// One thread = 4 consecutive outputs of a row: their 3 x 9 input window is read as two float4 and one scalar per input row
// (the one-output-per-thread form issued 9 stride-2 scalar loads per output: 143 us for the stem's 3 x 64 planes of 256^2,
// a kernel that moves 63 MB).  Same comparisons in the same order per output (max is exact: order-free anyway).
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ x, int H, int W, int Ho, int Wo,
                                                           long long total4, float* __restrict__ y) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int Wq = (Wo + 3) >> 2;
    const int q = (int)(i % Wq);
    const long long t = i / Wq;
    const int oy = (int)(t % Ho);
    const long long plane = t / Ho;
    const float* p = x + plane * H * W;
    const int ox0 = 4 * q, gx0 = 2 * ox0;                       // window columns gx0 - 1 .. gx0 + 7
    const bool fast = (W & 3) == 0 && gx0 + 8 <= W;
    float m[4] = {-3.4028235e38f, -3.4028235e38f, -3.4028235e38f, -3.4028235e38f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int gy = 2 * oy - 1 + dy;
        if (gy < 0 || gy >= H) continue;
        const float* r = p + (long long)gy * W;
        float v[9];
        if (fast) {
            const float4 a = *reinterpret_cast<const float4*>(r + gx0), b = *reinterpret_cast<const float4*>(r + gx0 + 4);
            v[0] = gx0 > 0 ? r[gx0 - 1] : -3.4028235e38f;
            v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w; v[5] = b.x; v[6] = b.y; v[7] = b.z; v[8] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int gx = gx0 - 1 + k;
                v[k] = (gx >= 0 && gx < W) ? r[gx] : -3.4028235e38f;
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) m[o] = fmaxf(m[o], fmaxf(fmaxf(v[2 * o], v[2 * o + 1]), v[2 * o + 2]));
    }
    float* dst = y + (plane * Ho + oy) * (long long)Wo + ox0;
    if ((Wo & 3) == 0) *reinterpret_cast<float4*>(dst) = make_float4(m[0], m[1], m[2], m[3]);
    else
        for (int o = 0; o < 4 && ox0 + o < Wo; ++o) dst[o] = m[o];
}

int th_maxpool3x3s2_launch(const float* x, int planes, int H, int W, float* y, hipStream_t s) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total4 = (long long)planes * Ho * ((Wo + 3) / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)th_cdiv(total4, 256)), dim3(256), 0, s, x, H, W, Ho, Wo, total4, y);
    TH_LAUNCH_CHECK();
    return 0;
}

// the shapes of the ResNet18 stem (bias-free, padding KS/2)
int th_conv2d_launch(const float* x, int N, int CIN, int H, int W, const void* packed, float inv_scale, int COUT, int KS,
                     int stride, float* y, hipStream_t s, unsigned int* range, float2* stats, int* np_out) {
    const int PAD = KS / 2;
    const int Ho = (H + 2 * PAD - KS) / stride + 1, Wo = (W + 2 * PAD - KS) / stride + 1;
    const uint4* wp = (const uint4*)packed;
    if (CIN == 3 && COUT == 64 && KS == 7 && stride == 2) return conv1_launch(x, wp, inv_scale, y, N, H, W, Ho, Wo, range, stats, np_out, s);
    if (CIN == 64 && COUT == 64 && KS == 3 && stride == 1)
#define CV_TR64 4
        return conv_launch_t<64, 64, 1, 3, CV_TR64, 2>(x, wp, inv_scale, y, N, H, W, Ho, Wo, range, stats, np_out, s);
    if (CIN == 64 && COUT == 128 && KS == 3 && stride == 2)
        return conv_launch_t<64, 128, 2, 3, 2, 4>(x, wp, inv_scale, y, N, H, W, Ho, Wo, range, stats, np_out, s);
    if (CIN == 128 && COUT == 128 && KS == 3 && stride == 1)
        return conv_launch_t<128, 128, 1, 3, 2, 4>(x, wp, inv_scale, y, N, H, W, Ho, Wo, range, stats, np_out, s);
    if (CIN == 64 && COUT == 128 && KS == 1 && stride == 2)
        return conv_launch_t<64, 128, 2, 1, 2, 4>(x, wp, inv_scale, y, N, H, W, Ho, Wo, range, stats, np_out, s);
    TH_REQUIRE(false, "th_conv2d: shape not built (ResNet18 stem shapes only)");
    return 1;
}

bool th_conv2d_built(int CIN, int COUT, int KS, int stride) {
    return (CIN == 3 && COUT == 64 && KS == 7 && stride == 2) || (CIN == 64 && COUT == 64 && KS == 3 && stride == 1) || (CIN == 64 && COUT == 128 && KS == 3 && stride == 2) ||
           (CIN == 128 && COUT == 128 && KS == 3 && stride == 1) || (CIN == 64 && COUT == 128 && KS == 1 && stride == 2);
}
