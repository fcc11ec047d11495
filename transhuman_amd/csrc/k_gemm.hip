// Dense layer on the fp32 matrix pipe: C[M,N] = act(A[M,K] W^T + b) (+C).
//
// Replaces every torch Conv1d(k=1)/Linear GEMM of the hot path
// (cross_transformer.py:315-351 and vision_transformer.py:271-279,250-253).
// Numerics: v_mfma_f32_16x16x4_f32 is bit-for-bit an fp32 fmaf chain, i.e. the
// same precision class as the reference's cuBLAS/cuDNN fp32 GEMMs (no TF32
// exists on gfx950).  Roofline: fp32 MFMA, 157.3 TFLOP/s.
//
// Tiling (wave64, 4 waves / workgroup):
//   workgroup tile 64 rows x (4 waves * NT * 16) columns, K in chunks of 128;
//   each wave owns 4 row tiles x NT column tiles of 16x16 (f32x4 accumulators);
//   A chunk  [64][128] staged in LDS with a +4 float row pad (b128 fragment
//   reads, <=2-way bank conflicts);
//   B (weights) pre-packed by th_pack_linear into the exact per-lane fragment
//   image, so one coalesced 1 KiB global_load_dwordx4 per wave fetches a whole
//   16(k) x 16(n) block straight into the MFMA operand registers (weights are
//   <= 1.2 MB per layer and stay L2-resident; no LDS round trip for them).
#include <stdlib.h>

#include "th_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GEMM_BM 64
#define GEMM_KC 128
#define GEMM_LDS_STRIDE (GEMM_KC + 4)

__global__ void pack_linear_kernel(const float* __restrict__ W, const float* __restrict__ b, int N, int K, int NB,
                                   int KB, float* __restrict__ wp, float* __restrict__ bp) {
    long long total = (long long)NB * KB * 256;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total;
         o += (long long)gridDim.x * blockDim.x) {
        int e = (int)(o & 3);
        int lane = (int)((o >> 2) & 63);
        long long blk = o >> 8;
        int kb = (int)(blk % KB);
        int nb = (int)(blk / KB);
        int n = nb * 16 + (lane & 15);
        int k = kb * 16 + 4 * (lane >> 4) + e;
        wp[o] = (n < N && k < K) ? W[(long long)n * K + k] : 0.0f;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NB * 16; i += gridDim.x * blockDim.x)
        bp[i] = (b != nullptr && i < N) ? b[i] : 0.0f;
}

int th_pack_linear(const th_linear& lin, void* storage, ThPacked* out, hipStream_t s) {
    TH_REQUIRE(lin.w != nullptr && lin.out_f > 0 && lin.in_f > 0, "bad layer");
    out->N = lin.out_f;
    out->K = lin.in_f;
    out->NB = (lin.out_f + 15) / 16;
    out->KB = (lin.in_f + 15) / 16;
    out->w = (float*)storage;
    out->b = (float*)((char*)storage + th_align((size_t)out->NB * out->KB * 256 * sizeof(float)));
    long long total = (long long)out->NB * out->KB * 256;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_linear_kernel, dim3(blocks), dim3(256), 0, s, lin.w, lin.b, lin.out_f, lin.in_f, out->NB,
                       out->KB, out->w, out->b);
    TH_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ float th_gelu_erf(float x) {
    // nn.GELU (exact): x * 0.5 * (1 + erf(x / sqrt(2)))
    return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}

// LN = true (MT = 1, K <= 256, K % 4 == 0): the rows of A are layer-normalised on their way into LDS,
// a = (a - mean_row) * rstd_row * ln_w[k] + ln_b[k] with the two-pass statistics of layernorm_kernel (k_vit.hip) --
// the pre-LN of a transformer block costs no launch of its own (every column workgroup of a row block recomputes the
// 16 row statistics: 3072 values, a fraction of a microsecond).
template <int NT, int MT, bool LN = false>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float* __restrict__ A, int lda, int M, int Kreal,
                                                            const float* __restrict__ Wp,
                                                            const float* __restrict__ bias, int N, int NB, int KB,
                                                            float* __restrict__ C, int ldc, int flags,
                                                            const float* __restrict__ ln_w = nullptr,
                                                            const float* __restrict__ ln_b = nullptr, float ln_eps = 0.f) {
    extern __shared__ __attribute__((aligned(16))) float As[];   // [64][GEMM_LDS_STRIDE]
    __shared__ float ln_mean[16], ln_rstd[16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    constexpr int BM = 16 * MT;                          // rows per workgroup (MT = 4: 64, MT = 1: 16)
    const int m0 = blockIdx.x * BM;
    const int nb0 = (blockIdx.y * 4 + wave) * NT;       // first 16-col tile of this wave
    const bool wave_active = nb0 < NB;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = (KB * 16 + GEMM_KC - 1) / GEMM_KC;
    // weight fragments: ring of GEMM_PF + 1 register sets, fragment kb + GEMM_PF is requested while block kb
    // multiplies (the ring index is static: a chunk has 8 = 2 * (GEMM_PF + 1) k-blocks and is fully unrolled)
    constexpr int GEMM_PF = 3;
    f32x4 bq[GEMM_PF + 1][NT];
    auto load_b = [&](int kb, f32x4 (&dst)[NT]) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int nb = nb0 + j;
            dst[j] = (nb < NB) ? *reinterpret_cast<const f32x4*>(Wp + ((long long)nb * KB + kb) * 256 + lane * 4)
                               : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    if (wave_active) {
#pragma unroll
        for (int i = 0; i < GEMM_PF; ++i)
            if (i < KB) load_b(i, bq[i]);
    }
    // A chunk [BM][128]: fetched into registers one chunk ahead (the loads of chunk ch+1 fly while chunk ch
    // multiplies), written to LDS at the top of its own iteration
    float4 areg[2 * MT];
    auto load_a = [&](int ch) {
        const int k0 = ch * GEMM_KC;
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            int idx = tid + 256 * i;
            int row = idx >> 5;
            int c4 = idx & 31;
            int gm = m0 + row;
            int gk = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                const float* src = A + (long long)gm * lda + gk;
                if (gk + 3 < Kreal) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (gk < Kreal) v.x = src[0];
                    if (gk + 1 < Kreal) v.y = src[1];
                    if (gk + 2 < Kreal) v.z = src[2];
                }
            }
            areg[i] = v;
        }
    };
    if (LN) {
        // 16 threads per row: K / 16 (<= 16) values each, 16-lane shuffle reductions; two passes like layernorm_kernel
        const int row = tid >> 4, sub = tid & 15, gm = m0 + row;
        float v[16];
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * (sub + 16 * q);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M && k < Kreal) x = *reinterpret_cast<const float4*>(A + (long long)gm * lda + k);
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
            sum += (x.x + x.y) + (x.z + x.w);
        }
        sum = th_row16_sum(sum);
        const float mean = sum / (float)Kreal;
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * (sub + 16 * q);
            if (k < Kreal) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float d = v[4 * q + e] - mean; ss += d * d; }
            }
        }
        ss = th_row16_sum(ss);
        if (sub == 0) { ln_mean[row] = mean; ln_rstd[row] = 1.0f / __fsqrt_rn(ss / (float)Kreal + ln_eps); }
        __syncthreads();
    }
    load_a(0);
    for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
        for (int i = 0; i < 2 * MT; ++i) {
            int idx = tid + 256 * i;
            if (LN) {
                const int row = idx >> 5, gk = ch * GEMM_KC + 4 * (idx & 31);
                if (gk < Kreal) {
                    const float mu = ln_mean[row], rs = ln_rstd[row];
                    const float4 w4 = *reinterpret_cast<const float4*>(ln_w + gk), b4 = *reinterpret_cast<const float4*>(ln_b + gk);
                    areg[i].x = (areg[i].x - mu) * rs * w4.x + b4.x;
                    areg[i].y = (areg[i].y - mu) * rs * w4.y + b4.y;
                    areg[i].z = (areg[i].z - mu) * rs * w4.z + b4.z;
                    areg[i].w = (areg[i].w - mu) * rs * w4.w + b4.w;
                }
            }
            *reinterpret_cast<float4*>(&As[(idx >> 5) * GEMM_LDS_STRIDE + 4 * (idx & 31)]) = areg[i];
        }
        __syncthreads();
        if (ch + 1 < nchunks) load_a(ch + 1);
        if (wave_active) {
            const int kb_lo = ch * (GEMM_KC / 16);
#pragma unroll
            for (int kl8 = 0; kl8 < GEMM_KC / 16; ++kl8) {
                const int kb = kb_lo + kl8;
                if (kb < KB) {
                    if (kb + GEMM_PF < KB) load_b(kb + GEMM_PF, bq[(kl8 + GEMM_PF) % (GEMM_PF + 1)]);
                    f32x4 a[MT];
                    const int kl = kl8 * 16 + 4 * (lane >> 4);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        a[i] = *reinterpret_cast<const f32x4*>(&As[(i * 16 + (lane & 15)) * GEMM_LDS_STRIDE + kl]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], bq[kl8 % (GEMM_PF + 1)][j][e],
                                                                                 acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    if (!wave_active) return;
    const int act = flags & 15;
    const bool accum = (flags & TH_GEMM_ACCUM) != 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int col = (nb0 + j) * 16 + (lane & 15);
        if (col >= N) continue;
        float bv = bias[col];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = m0 + i * 16 + 4 * (lane >> 4) + r;
                if (row < M) {
                    float v = acc[i][j][r] + bv;
                    if (act == TH_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (act == TH_ACT_GELU) v = th_gelu_erf(v);
                    float* dst = C + (long long)row * ldc + col;
                    if (accum) v = *dst + v;
                    *dst = v;
                }
            }
        }
    }
}

static int pick_nt(int NB) {
    // columns per workgroup = 4 waves * NT tiles; minimise padded tiles, prefer big NT
    int best = 1, best_waste = 1 << 30;
    for (int nt = 4; nt >= 1; --nt) {
        int per = 4 * nt;
        int blocks = (NB + per - 1) / per;
        int waste = blocks * per - NB;
        if (waste < best_waste) { best_waste = waste; best = nt; }
    }
    return best;
}

static int gemm_launch(const float* A, int lda, int M, const ThPacked& W, int flags, float* C, int ldc,
                       const float* ln_w, const float* ln_b, float ln_eps, hipStream_t s);

int th_gemm(const float* A, int lda, int M, const ThPacked& W, int flags, float* C, int ldc, hipStream_t s) {
    return gemm_launch(A, lda, M, W, flags, C, ldc, nullptr, nullptr, 0.f, s);
}

bool th_gemm_ln_ok(int M, const ThPacked& W) { return M > 0 && M <= 8192 && W.K <= 256 && (W.K & 3) == 0; }

// C = act(LayerNorm(A; ln_w, ln_b, eps) W^T + b): rows normalised inside the GEMM (th_gemm_ln_ok shapes only)
int th_gemm_ln(const float* A, int lda, int M, const ThPacked& W, const float* ln_w, const float* ln_b, float eps,
               int flags, float* C, int ldc, hipStream_t s) {
    TH_REQUIRE(th_gemm_ln_ok(M, W) && ln_w && ln_b, "th_gemm_ln: unsupported shape");
    return gemm_launch(A, lda, M, W, flags, C, ldc, ln_w, ln_b, eps, s);
}

static int gemm_launch(const float* A, int lda, int M, const ThPacked& W, int flags, float* C, int ldc,
                       const float* ln_w, const float* ln_b, float ln_eps, hipStream_t s) {
    if (M <= 0) return 0;
    TH_REQUIRE((lda & 3) == 0 && (((uintptr_t)A) & 15) == 0, "A must be 16-byte aligned with lda % 4 == 0");
    TH_REQUIRE(W.w != nullptr, "weights not packed");
    // few rows (the ViT: V*N_c = 900..4500 tokens): 16-row workgroups of 64 columns so the launch still covers
    // the chip several times over (these launches are latency-bound: co-resident workgroups hide the L2 round
    // trips of each other's weight fragments)
    const bool small = M <= 8192;
    static const int small_nt = getenv("TH_GEMM_SMALL_NT") ? atoi(getenv("TH_GEMM_SMALL_NT")) : 1;
    int nt = small ? (small_nt >= 1 && small_nt <= 4 ? small_nt : pick_nt(W.NB)) : pick_nt(W.NB);
    const int bm = small ? 16 : GEMM_BM;
    dim3 grid(th_cdiv(M, bm), th_cdiv(W.NB, 4 * nt));
    size_t lds = (size_t)bm * GEMM_LDS_STRIDE * sizeof(float);
#define LAUNCH(NT_, MT_)                                                                                            \
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<NT_, MT_>), grid, dim3(256), lds, s, A, lda, M, W.K, W.w, W.b, W.N, W.NB, \
                       W.KB, C, ldc, flags)
#define LAUNCH_LN(NT_)                                                                                              \
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<NT_, 1, true>), grid, dim3(256), lds, s, A, lda, M, W.K, W.w, W.b, W.N, W.NB, \
                       W.KB, C, ldc, flags, ln_w, ln_b, ln_eps)
#define LAUNCH_NT(NT_)                 \
    do {                               \
        if (ln_w) LAUNCH_LN(NT_);      \
        else if (small) LAUNCH(NT_, 1); \
        else LAUNCH(NT_, 4);           \
    } while (0)
    switch (nt) {
        case 4: LAUNCH_NT(4); break;
        case 3: LAUNCH_NT(3); break;
        case 2: LAUNCH_NT(2); break;
        default: LAUNCH_NT(1); break;
    }
#undef LAUNCH_NT
#undef LAUNCH_LN
#undef LAUNCH
    TH_LAUNCH_CHECK();
    return 0;
}


// ---- fp16-split form for the small-M layers of TransHE ---------------------------------------------------------------
// The ViT's dense layers are 1500..4500 rows x 192..768: on the fp32 matrix pipe a 16-row workgroup spends 48..192
// MFMAs of 32 cycles per column tile and the launch is as long as that chain (8..15 us per layer, 890 us per forward).
// Here both operands are split into fp16 hi + lo (x = hi + lo to 2^-22) and three v_mfma_f32_16x16x32_f16 products
// accumulate in fp32 (hi*lo + lo*hi + hi*hi: the fused MLP's fp32-class scheme): 18..72 MFMAs of 16 cycles.
//   workgroup = 16 rows x 64 columns (4 waves x one 16-column tile): ceil(M/16) x ceil(N/64) workgroups;
//   the WHOLE K range of the 16 rows is converted once into LDS planes [16][K] hi | lo (row stride 2K + 16 B: an odd
//   number of 16-byte slots, conflict-free ds_read_b128 A fragments); LN = true normalises the rows on the way in;
//   weight fragments (pre-split, pre-scaled, th_pack_linear_h3) stream from L2 through a 6-deep register ring.
typedef _Float16 g_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 g_h4 __attribute__((ext_vector_type(4)));

__global__ void h3_absmax_kernel(const float* __restrict__ w, long long n, unsigned int* __restrict__ out) {
    float m = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__global__ void h3_pack_kernel(const float* __restrict__ W, int N, int K, int NB, int KB32,
                               const unsigned int* __restrict__ amax_bits, uint4* __restrict__ out, float* __restrict__ inv_out) {
    float amax = __uint_as_float(*amax_bits);
    if (!(amax > 0.f) || !(amax < 3.0e38f)) amax = 1.f;
    int e;
    frexpf(amax, &e);                          // amax = m * 2^e, m in [0.5, 1)
    const float scale = ldexpf(1.f, 13 - e);
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_out[0] = ldexpf(1.f, e - 13);
    const long long total = (long long)NB * KB32 * 2 * 64;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(o & 63);
        long long q = o >> 6;
        const int plane = (int)(q & 1); q >>= 1;
        const int kb = (int)(q % KB32);
        const int nb = (int)(q / KB32);
        const int n = nb * 16 + (lane & 15);
        const int k0 = kb * 32 + 8 * (lane >> 4);
        g_h8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const float x = (n < N && k < K) ? W[(long long)n * K + k] * scale : 0.f;
            const _Float16 hi = (_Float16)x;
            v[j] = plane == 0 ? hi : (_Float16)(x - (float)hi);
        }
        out[o] = *reinterpret_cast<uint4*>(&v);
    }
}

int th_pack_linear_h3(const th_linear& lin, void* storage_h3, ThPacked* out, hipStream_t s) {
    TH_REQUIRE(lin.w != nullptr && out->w != nullptr && out->N == lin.out_f && out->K == lin.in_f, "pack the fp32 image first");
    out->KB32 = (lin.in_f + 31) / 32;
    uint4* w16 = (uint4*)storage_h3;
    const size_t img = th_align((size_t)out->NB * out->KB32 * 2 * 64 * 16);
    float* sc = (float*)((char*)storage_h3 + img);
    unsigned int* amax = (unsigned int*)(sc + 4);
    TH_HIP(hipMemsetAsync(amax, 0, 4, s));
    hipLaunchKernelGGL(h3_absmax_kernel, dim3(64), dim3(256), 0, s, lin.w, (long long)lin.out_f * lin.in_f, amax);
    hipLaunchKernelGGL(h3_pack_kernel, dim3(256), dim3(256), 0, s, lin.w, lin.out_f, lin.in_f, out->NB, out->KB32, amax, w16, sc);
    TH_LAUNCH_CHECK();
    out->w16 = w16;
    out->scale16 = sc;
    return 0;
}

#define H3_RING 6
// RT row tiles of 16 per workgroup.  A workgroup streams its 64 columns of the weight image (4 K bytes per column) once per
// 16 RT rows: with RT = 1 the V N_c = 1500 rows of the ViT re-read every layer's weights 94 times (41-55 MB of L2 traffic
// per GEMM: that, not latency, bounded the 9-10 us launches); RT = 2 / 4 halves / quarters it and gives a wave 2 / 4
// independent accumulators per product term.
template <bool LN, int RT>
__global__ __launch_bounds__(256) void gemm_h3_kernel(const float* __restrict__ A, int lda, int M, int Kreal,
                                                      const uint4* __restrict__ W16, const float* __restrict__ inv_scale,
                                                      const float* __restrict__ bias, int N, int NB, int KB32,
                                                      float* __restrict__ C, int ldc, int flags,
                                                      const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                      float ln_eps, unsigned int* __restrict__ range, ThQkvSplit qs) {
    extern __shared__ __attribute__((aligned(16))) char h3_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 16 * RT;
    const int nb = blockIdx.y * 4 + wave;
    const bool wave_active = nb < NB;
    const int KP = KB32 * 32;
    const int stride = 2 * KP + 16;                        // bytes per LDS row of a plane
    char* a_hi = h3_lds;
    char* a_lo = h3_lds + 16 * RT * stride;

    // weight fragments of the first ring stages: requested before the operand is staged
    uint4 bq[H3_RING][2];
    const uint4* wl = W16 + ((long long)nb * KB32) * 128 + lane;
    if (wave_active) {
#pragma unroll
        for (int j = 0; j < H3_RING; ++j)
            if (j < KB32) { bq[j][0] = wl[j * 128]; bq[j][1] = wl[j * 128 + 64]; }
    }
    // epilogue operands (scale, bias, the residual of TH_GEMM_ACCUM) are requested here as well, not behind the MFMAs
    const int act = flags & 15;
    const bool accum = (flags & TH_GEMM_ACCUM) != 0;
    const int col = nb * 16 + (lane & 15);
    const bool col_ok = wave_active && col < N;
    const float inv = inv_scale[0], bv = col_ok ? bias[col] : 0.f;
    float cres[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * t + 4 * (lane >> 4) + r;
            cres[t][r] = (accum && col_ok && row < M) ? C[(long long)row * ldc + col] : 0.f;
        }
    unsigned rmax = 0u;
    if (LN) {
        // 16 threads per row hold the row in registers (K <= 256): two-pass statistics like layernorm_kernel (k_vit.hip)
        // with DPP row reductions (every lane gets mean / rstd: no LDS, no barrier), normalised and split in place --
        // the row is read ONCE (it used to be re-read by the staging loop behind a barrier)
        const int sub = tid & 15;
        float4 w4[4], b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * (sub + 16 * q);
            w4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            b4[q] = w4[q];
            if (k < Kreal) {
                w4[q] = *reinterpret_cast<const float4*>(ln_w + k);
                b4[q] = *reinterpret_cast<const float4*>(ln_b + k);
            }
        }
        float v[RT][16];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int gm = m0 + 16 * t + (tid >> 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * (sub + 16 * q);
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gm < M && k < Kreal) x = *reinterpret_cast<const float4*>(A + (long long)gm * lda + k);
                v[t][4 * q] = x.x; v[t][4 * q + 1] = x.y; v[t][4 * q + 2] = x.z; v[t][4 * q + 3] = x.w;
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int row = 16 * t + (tid >> 4), gm = m0 + row;
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) sum += (v[t][4 * q] + v[t][4 * q + 1]) + (v[t][4 * q + 2] + v[t][4 * q + 3]);
            sum = th_row16_sum(sum);
            const float mean = sum / (float)Kreal;
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * (sub + 16 * q);
                if (k < Kreal) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float d = v[t][4 * q + e] - mean; ss += d * d; }
                }
            }
            ss = th_row16_sum(ss);
            const float rs = 1.0f / __fsqrt_rn(ss / (float)Kreal + ln_eps);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * (sub + 16 * q);
                if (k < KP) {
                    const float wv[4] = {w4[q].x, w4[q].y, w4[q].z, w4[q].w}, bb[4] = {b4[q].x, b4[q].y, b4[q].z, b4[q].w};
                    g_h4 hv, lv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = 0.f;
                        if (gm < M && k < Kreal) x = (v[t][4 * q + e] - mean) * rs * wv[e] + bb[e];
                        const _Float16 hi = (_Float16)x;
                        hv[e] = hi;
                        lv[e] = (_Float16)(x - (float)hi);
                        rmax = max(rmax, (unsigned)(__builtin_bit_cast(unsigned short, hi) & 0x7fffu));
                    }
                    *reinterpret_cast<g_h4*>(a_hi + row * stride + 2 * k) = hv;
                    *reinterpret_cast<g_h4*>(a_lo + row * stride + 2 * k) = lv;
                }
            }
        }
    } else {
        // stage + split the 16 RT x KP operand
        const int c4n = KP / 4;
        for (int idx = tid; idx < 16 * RT * c4n; idx += 256) {
            const int row = idx / c4n, c4 = idx - row * c4n, gm = m0 + row, gk = 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M && gk < Kreal) v = *reinterpret_cast<const float4*>(A + (long long)gm * lda + gk);     // (K % 4 == 0 is required)
            const float x4[4] = {v.x, v.y, v.z, v.w};
            g_h4 hv, lv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 hi = (_Float16)x4[e];
                hv[e] = hi;
                lv[e] = (_Float16)(x4[e] - (float)hi);
                rmax = max(rmax, (unsigned)(__builtin_bit_cast(unsigned short, hi) & 0x7fffu));
            }
            *reinterpret_cast<g_h4*>(a_hi + row * stride + 8 * c4) = hv;
            *reinterpret_cast<g_h4*>(a_lo + row * stride + 8 * c4) = lv;
        }
    }
    if (range != nullptr && rmax > range[TH_RANGE_VIT]) atomicMax(range + TH_RANGE_VIT, rmax);
    __syncthreads();
    if (!wave_active) return;

    // one accumulator per row tile and product term (hi*lo, lo*hi, hi*hi): independent MFMA chains
    f32x4 acc[RT][3];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int e = 0; e < 3; ++e) acc[t][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int aoff = (lane & 15) * stride + 16 * (lane >> 4);
    for (int kb0 = 0; kb0 < KB32; kb0 += H3_RING) {
#pragma unroll
        for (int j = 0; j < H3_RING; ++j) {
            const int kb = kb0 + j;
            if (kb < KB32) {
                const g_h8 bh = *reinterpret_cast<const g_h8*>(&bq[j][0]);
                const g_h8 bl = *reinterpret_cast<const g_h8*>(&bq[j][1]);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const g_h8 ah = *reinterpret_cast<const g_h8*>(a_hi + 16 * t * stride + aoff + kb * 64);
                    const g_h8 al = *reinterpret_cast<const g_h8*>(a_lo + 16 * t * stride + aoff + kb * 64);
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t][2], 0, 0, 0);
                }
                if (kb + H3_RING < KB32) { bq[j][0] = wl[(kb + H3_RING) * 128]; bq[j][1] = wl[(kb + H3_RING) * 128 + 64]; }
            }
        }
    }
    if (col >= N) return;
    // qkv epilogue: this wave's 16 columns are queries (fp32 as usual), keys or values (split planes) -- wave-uniform
    const int part = (qs.Kp != nullptr) ? col / qs.dim : 0;
    const int hd = (qs.Kp != nullptr) ? (col - part * qs.dim) : 0;       // head * 64 + d
    const long long plane = (long long)qs.Npad * 64;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const f32x4 a = (acc[t][0] + acc[t][1]) + acc[t][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * t + 4 * (lane >> 4) + r;
            if (row < M) {
                float v = a[r] * inv + bv;
                if (act == TH_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (act == TH_ACT_GELU) v = th_gelu_erf(v);
                if (accum) v = cres[t][r] + v;
                if (part == 0) C[(long long)row * ldc + col] = v;
                else {
                    const int view = row / qs.N, key = row - view * qs.N;
                    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
                    _Float16* base = (part == 1 ? qs.Kp : qs.Vp) + ((long long)view * qs.heads + (hd >> 6)) * 2 * plane;
                    long long o;
                    if (part == 1) o = (long long)key * 64 + (hd & 63);
                    else {      // V^T row d, keys in the fragment order of attn2_kernel (see kv_split_kernel)
                        const int ko = key & 31;
                        o = (long long)(hd & 63) * qs.Npad + (key & ~31) + 8 * ((ko & 15) >> 2) + 4 * (ko >> 4) + (ko & 3);
                    }
                    base[o] = hi;
                    base[plane + o] = lo;
                }
            }
        }
    }
}

bool th_gemm_h3_ok(int M, const ThPacked& W, bool ln) {
    return W.w16 != nullptr && M > 0 && M <= 8192 && (W.K & 3) == 0 && W.K <= 768 && (!ln || W.K <= 256);
}

int th_gemm_h3(const float* A, int lda, int M, const ThPacked& W, const float* ln_w, const float* ln_b, float eps, int flags,
               float* C, int ldc, unsigned int* range, hipStream_t s, const ThQkvSplit* qkv) {
    const bool ln = ln_w != nullptr;
    TH_REQUIRE(th_gemm_h3_ok(M, W, ln), "th_gemm_h3: unsupported shape / layer not packed for the fp16-split path");
    ThQkvSplit qs{};
    if (qkv != nullptr) {
        qs = *qkv;
        TH_REQUIRE(W.N == 3 * qs.dim && (qs.dim & 63) == 0 && qs.N > 0 && M % qs.N == 0 && (flags & TH_GEMM_ACCUM) == 0,
                   "th_gemm_h3: the qkv epilogue needs N = 3 dim, dim = heads * 64, M = views * tokens");
    }
    TH_REQUIRE((lda & 3) == 0 && (((uintptr_t)A) & 15) == 0, "A must be 16-byte aligned with lda % 4 == 0");
    const int KP = W.KB32 * 32;
    static unsigned long long attr = 0ull;
    if (th_lds_attr_needed(&attr)) {
        const int mx = 2 * 16 * 4 * (2 * 256 + 16), mx2 = 2 * 16 * 2 * (2 * 768 + 16);
        TH_HIP(hipFuncSetAttribute((const void*)gemm_h3_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx2));
        TH_HIP(hipFuncSetAttribute((const void*)gemm_h3_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, mx2));
        TH_HIP(hipFuncSetAttribute((const void*)gemm_h3_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, mx2));
        TH_HIP(hipFuncSetAttribute((const void*)gemm_h3_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, mx2));
        TH_HIP(hipFuncSetAttribute((const void*)gemm_h3_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
        TH_HIP(hipFuncSetAttribute((const void*)gemm_h3_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, mx));
    }
    // rows per workgroup: 16 (RT = 1).  Larger tiles re-read the weights less often (RT = 2 / 4: half / a quarter of the
    // 41-55 MB of L2 traffic per GEMM of the ViT) but were measured SLOWER at V N_c = 900 .. 4500 rows (0.79 / 0.86 /
    // 0.91 ms per forward at RT = 1 / 2 / 4, N_c = 500): these launches are bound by the length of the per-workgroup
    // dependency chain, not by bandwidth.  TH_GEMM_H3_RT = 2 | 4 selects the larger tiles (A/B switch).
    static const int rt_env = getenv("TH_GEMM_H3_RT") ? atoi(getenv("TH_GEMM_H3_RT")) : 0;
    const int gy = th_cdiv(W.NB, 4);
    int rt = 1;
    if (rt_env == 2 || (rt_env == 4 && KP <= 256)) rt = rt_env;
    const size_t lds = (size_t)2 * 16 * rt * (2 * KP + 16);
    dim3 grid(th_cdiv(M, 16 * rt), gy);
#define H3_LAUNCH(LN_, RT_)                                                                                              \
    hipLaunchKernelGGL((gemm_h3_kernel<LN_, RT_>), grid, dim3(256), lds, s, A, lda, M, W.K, W.w16, W.scale16, W.b, W.N, W.NB, \
                       W.KB32, C, ldc, flags, ln_w, ln_b, eps, range, qs)
    if (ln) { if (rt == 4) H3_LAUNCH(true, 4); else if (rt == 2) H3_LAUNCH(true, 2); else H3_LAUNCH(true, 1); }
    else    { if (rt == 4) H3_LAUNCH(false, 4); else if (rt == 2) H3_LAUNCH(false, 2); else H3_LAUNCH(false, 1); }
#undef H3_LAUNCH
    TH_LAUNCH_CHECK();
    return 0;
}
