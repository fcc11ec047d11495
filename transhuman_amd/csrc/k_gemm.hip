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
        for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
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
        for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o);
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
