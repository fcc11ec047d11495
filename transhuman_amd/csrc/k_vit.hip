// K3: TransHE -- ViT-tiny over the N_c canonical-body tokens.
//
// VisionTransformer.forward (lib/networks/vision_transformer.py:371-383):
// x += PE; depth x [ pre-LN (eps 1e-6) -> fused qkv Linear -> 3-head
// softmax(q k^T * 0.125) v -> proj -> residual ; pre-LN -> FC 192->768 ->
// GELU(erf) -> FC 768->192 -> residual ] ; final LN   (:257-307).
// Dense layers: th_gemm (fp32 MFMA).  Attention: flash-style, one workgroup =
// 64 queries (16 per wave) of one (view, head); K/V tiles of 64 keys staged in
// LDS; q k^T and p v on v_mfma_f32_16x16x4_f32; online softmax in registers with
// 16-lane shuffle reductions.  The [V,3,N,N] probability tensor the reference
// materialises (81 MB at N_c = 1500) never exists.
// Per-frame cost 12/23/110 GFLOP at N_c = 300/500/1500 (fp32 MFMA bound).
#include "th_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                           float* __restrict__ o) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

// wave per row, dim <= 256*... (dim=192 -> 3 values per lane)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int rows, int dim,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, float* __restrict__ o) {
    const int lane = threadIdx.x & 63;
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + (long long)r * dim;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = lane + 64 * i;
        v[i] = (c < dim) ? xr[c] : 0.f;
        s += v[i];
    }
    for (int q = 32; q > 0; q >>= 1) s += __shfl_xor(s, q);
    float mean = s / (float)dim;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float d = (lane + 64 * i < dim) ? v[i] - mean : 0.f;
        ss += d * d;
    }
    for (int q = 32; q > 0; q >>= 1) ss += __shfl_xor(ss, q);
    float rstd = 1.0f / __fsqrt_rn(ss / (float)dim + eps);
    float* orow = o + (long long)r * dim;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = lane + 64 * i;
        if (c < dim) orow[c] = (v[i] - mean) * rstd * w[c] + b[c];
    }
}

// ---- attention -------------------------------------------------------------------
// qkv rows: [T, 3*dim] with col = which*dim + head*64 + d  (:271).  head_dim = 64.
#define AT_Q 64
#define AT_K 64
#define AT_STR 68      // LDS row stride (floats) for K, V and P tiles

__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, int N, int dim, float scale,
                                                   float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Ks[AT_K * AT_STR];
    __shared__ __attribute__((aligned(16))) float Vs[AT_K * AT_STR];
    __shared__ __attribute__((aligned(16))) float Psh[4][16 * AT_STR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y, view = blockIdx.z;
    const int q0 = blockIdx.x * AT_Q + wave * 16;
    const int ld = 3 * dim;
    const float* base = qkv + (long long)view * N * ld;
    const int qcol = head * 64, kcol = dim + head * 64, vcol = 2 * dim + head * 64;

    // Q fragments: A[i = lane&15][k = 16*kb + 4*(lane>>4) + e]
    f32x4 qf[4];
    {
        int qi = q0 + (lane & 15);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (qi < N) qf[kb] = *reinterpret_cast<const f32x4*>(base + (long long)qi * ld + qcol + kb * 16 + 4 * (lane >> 4));
            else qf[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 oacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) oacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrun[r] = -3.0e38f; lrun[r] = 0.f; }

    // K / V tiles (64 keys x 64 floats each = 1024 float4 per tile): fetched into registers one tile ahead so
    // the L2 round trip of tile t+1 overlaps the two MFMA passes and the softmax of tile t
    float4 kreg[4], vreg[4];
    auto fetch_kv = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = tid + 256 * i;
            int row = idx >> 4, c4 = idx & 15;
            int key = k0 + row;
            kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            vreg[i] = kreg[i];
            if (key < N) {
                kreg[i] = *reinterpret_cast<const float4*>(base + (long long)key * ld + kcol + 4 * c4);
                vreg[i] = *reinterpret_cast<const float4*>(base + (long long)key * ld + vcol + 4 * c4);
            }
        }
    };
    fetch_kv(0);
    for (int k0 = 0; k0 < N; k0 += AT_K) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = tid + 256 * i;
            int row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4*>(&Ks[row * AT_STR + 4 * c4]) = kreg[i];
            *reinterpret_cast<float4*>(&Vs[row * AT_STR + 4 * c4]) = vreg[i];
        }
        __syncthreads();
        if (k0 + AT_K < N) fetch_kv(k0 + AT_K);
        // S = Q K^T : B[k=d][j=key]  lane (j = lane&15, kq = lane>>4) reads K[key][16kb+4kq .. +3]
        f32x4 sacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            sacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f32x4 bf = *reinterpret_cast<const f32x4*>(&Ks[(nt * 16 + (lane & 15)) * AT_STR + kb * 16 + 4 * (lane >> 4)]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb][e], bf[e], sacc[nt], 0, 0, 0);
            }
        }
        // C layout: row = 4*(lane>>4)+r, col = nt*16 + (lane&15)
        float mnew[4], corr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -3.0e38f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int key = k0 + nt * 16 + (lane & 15);
                float sv = (key < N) ? sacc[nt][r] * scale : -3.0e38f;
                sacc[nt][r] = sv;
                m = fmaxf(m, sv);
            }
            for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
            mnew[r] = fmaxf(mrun[r], m);
            corr[r] = expf(mrun[r] - mnew[r]);
            mrun[r] = mnew[r];
        }
        float* Pw = Psh[wave];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float ls = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int key = k0 + nt * 16 + (lane & 15);
                float pv = (key < N) ? expf(sacc[nt][r] - mnew[r]) : 0.f;
                ls += pv;
                Pw[(4 * (lane >> 4) + r) * AT_STR + nt * 16 + (lane & 15)] = pv;
            }
            for (int o = 1; o < 16; o <<= 1) ls += __shfl_xor(ls, o);
            lrun[r] = lrun[r] * corr[r] + ls;
#pragma unroll
            for (int j = 0; j < 4; ++j) oacc[j][r] *= corr[r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): P tile of this wave is written
        __builtin_amdgcn_wave_barrier();
        // O += P V : A[i=q][k=key] from Pw, B[k=key][j=d] = V[key][d]
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 af = *reinterpret_cast<const f32x4*>(&Pw[(lane & 15) * AT_STR + kb * 16 + 4 * (lane >> 4)]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bv = Vs[(kb * 16 + 4 * (lane >> 4) + e) * AT_STR + j * 16 + (lane & 15)];
                    oacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bv, oacc[j], 0, 0, 0);
                }
            }
        }
    }
    // out[t][head*64 + d] = O / l      (x = (attn @ v).transpose(1,2).reshape(B,N,C), :278)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int qi = q0 + 4 * (lane >> 4) + r;
        if (qi < N) {
            float inv = 1.0f / lrun[r];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                out[((long long)view * N + qi) * dim + head * 64 + j * 16 + (lane & 15)] = oacc[j][r] * inv;
        }
    }
}

size_t th_vit_ws(int V, int N, int dim, int heads) {
    size_t T = (size_t)V * N;
    return 2 * th_align(T * dim * 4) + th_align(T * 4 * dim * 4);
}

int th_vit_launch(const ThVitPacked& W, const float* x, const float* pe, int V, int N, float* out, void* ws,
                  size_t ws_bytes, hipStream_t s) {
    TH_REQUIRE(W.ready, "ViT weights not set (th_set_vit_weights)");
    const int dim = W.dim, heads = W.heads;
    TH_REQUIRE(dim == heads * 64, "attention kernel is built for head_dim 64 (ViT-tiny: 192 = 3 x 64)");
    TH_REQUIRE(dim <= 512, "dim too large for the layernorm kernel");
    TH_REQUIRE(ws_bytes >= th_vit_ws(V, N, dim, heads), "workspace too small");
    ThArena ar(ws, ws_bytes);
    const int T = V * N;
    float* X = ar.take<float>((size_t)T * dim);
    float* Y = ar.take<float>((size_t)T * dim);
    float* Q = ar.take<float>((size_t)T * 4 * dim);     // qkv (3*dim) or mlp hidden (4*dim)
    TH_REQUIRE(Q != nullptr, "workspace carve failed");
    long long n = (long long)T * dim;
    hipLaunchKernelGGL(add_kernel, dim3(th_cdiv(n, 256)), dim3(256), 0, s, x, pe, n, X);
    const float scale = 0.125f;   // head_dim ** -0.5
    for (int b = 0; b < W.depth; ++b) {
        const ThVitBlockPacked& B = W.blocks[b];
        hipLaunchKernelGGL(layernorm_kernel, dim3(th_cdiv(T, 4)), dim3(256), 0, s, X, T, dim, B.ln1_w, B.ln1_b, 1e-6f, Y);
        TH_TRY(th_gemm(Y, dim, T, B.qkv, TH_ACT_NONE, Q, 3 * dim, s));
        hipLaunchKernelGGL(attn_kernel, dim3(th_cdiv(N, AT_Q), heads, V), dim3(256), 0, s, Q, N, dim, scale, Y);
        TH_TRY(th_gemm(Y, dim, T, B.proj, TH_ACT_NONE | TH_GEMM_ACCUM, X, dim, s));
        hipLaunchKernelGGL(layernorm_kernel, dim3(th_cdiv(T, 4)), dim3(256), 0, s, X, T, dim, B.ln2_w, B.ln2_b, 1e-6f, Y);
        TH_TRY(th_gemm(Y, dim, T, B.fc1, TH_ACT_GELU, Q, 4 * dim, s));
        TH_TRY(th_gemm(Q, 4 * dim, T, B.fc2, TH_ACT_NONE | TH_GEMM_ACCUM, X, dim, s));
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3(th_cdiv(T, 4)), dim3(256), 0, s, X, T, dim, W.norm_w, W.norm_b, 1e-6f, out);
    TH_LAUNCH_CHECK();
    return 0;
}
