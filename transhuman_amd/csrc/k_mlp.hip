// K6 (layer-by-layer form): the per-point multi-view colour/density MLP.
//
// Network._multiview_agg / cross_attention / _alpha_forward / _RGB_forward
// (cross_transformer.py:128-149, :313-353) for P compacted samples and V views.
// Every dense layer is one th_gemm launch on the fp32 MFMA pipe (k_gemm.hip);
// the kernels in this file are the non-GEMM glue: the 3x3 cross-view softmax
// attention, view means, view-direction concat, the 1- and 3-wide heads and the
// scatter back to the dense raw[R*S,4] buffer.  Activations are row-major
// [sample][view][channel] so the V rows of a sample are adjacent.
// (A fused single-kernel form that keeps activations in LDS is the follow-up;
// this form is the parity reference for it on the GPU.)
#include "th_internal.h"

// ---- cross-view attention (cross_transformer.py:128-149) -----------------------
// kvp/kvs rows: [key(128) | value(256)] of the pixel / token branch.
// A[j][i] = kp_j . ks_i / sqrt(128); softmax over j (pixel views, dim=1 :144);
// n_i = vs_i + sum_j vp_j A[j][i].
template <int V>
__global__ __launch_bounds__(256) void xattn_kernel(const float* __restrict__ kvp, const float* __restrict__ kvs,
                                                    int P, float* __restrict__ n) {
    const int lane = threadIdx.x & 63;
    int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const float* bp = kvp + (long long)p * V * 384;
    const float* bs = kvs + (long long)p * V * 384;
    float kp[V][2], ks[V][2];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        kp[v][0] = bp[v * 384 + lane]; kp[v][1] = bp[v * 384 + 64 + lane];
        ks[v][0] = bs[v * 384 + lane]; ks[v][1] = bs[v * 384 + 64 + lane];
    }
    float A[V][V];
#pragma unroll
    for (int j = 0; j < V; ++j)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float t = kp[j][0] * ks[i][0] + kp[j][1] * ks[i][1];
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            A[j][i] = t / 11.313708498984761f;          // math.sqrt(128), :142
        }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        float m = A[0][i];
#pragma unroll
        for (int j = 1; j < V; ++j) m = fmaxf(m, A[j][i]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < V; ++j) { A[j][i] = expf(A[j][i] - m); s = s + A[j][i]; }
#pragma unroll
        for (int j = 0; j < V; ++j) A[j][i] = A[j][i] / s;
    }
    float4 vp[V];
#pragma unroll
    for (int j = 0; j < V; ++j) vp[j] = *reinterpret_cast<const float4*>(bp + j * 384 + 128 + 4 * lane);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        float4 vs = *reinterpret_cast<const float4*>(bs + i * 384 + 128 + 4 * lane);
        float4 t;
        t.x = vp[0].x * A[0][i]; t.y = vp[0].y * A[0][i]; t.z = vp[0].z * A[0][i]; t.w = vp[0].w * A[0][i];
#pragma unroll
        for (int j = 1; j < V; ++j) {
            t.x = t.x + vp[j].x * A[j][i]; t.y = t.y + vp[j].y * A[j][i];
            t.z = t.z + vp[j].z * A[j][i]; t.w = t.w + vp[j].w * A[j][i];
        }
        float4 r = make_float4(vs.x + t.x, vs.y + t.y, vs.z + t.z, vs.w + t.w);
        *reinterpret_cast<float4*>(n + ((long long)p * V + i) * 256 + 4 * lane) = r;
    }
}

// combine_interleaved(..., "average") (cross_transformer.py:61-72): mean over views
__global__ void view_mean_kernel(const float* __restrict__ x, int P, int V, int C, float* __restrict__ out) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)P * C) return;
    int p = (int)(i / C), c = (int)(i % C);
    const float* b = x + (long long)p * V * C + c;
    float a = b[0];
    for (int v = 1; v < V; ++v) a = a + b[(long long)v * C];
    out[i] = a / (float)V;
}

// feat rows are [rows,288]: cols 256..282 <- viewdir[p] (27), 283..287 <- 0  (:338-340)
__global__ void viewdir_fill_kernel(const float* __restrict__ vd, int P, int V, float* __restrict__ feat) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)P * V * 32) return;
    long long row = i >> 5;
    int c = (int)(i & 31);
    int p = (int)(row / V);
    feat[row * 288 + 256 + c] = (c < 27) ? vd[(long long)p * 27 + c] : 0.f;
}

// alpha_fc (256 -> 1) / rgb_fc (128 -> 3): wave per sample
template <int OUT>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, int P, int K, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    float acc[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) acc[o] = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = x[(long long)p * K + k];
#pragma unroll
        for (int o = 0; o < OUT; ++o) acc[o] = fmaf(xv, w[o * K + k], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        float t = acc[o];
        for (int s = 32; s > 0; s >>= 1) t += __shfl_xor(t, s);
        if (lane == 0) out[(long long)p * ldo + o] = t + b[o];
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int width, const int32_t* __restrict__ sel,
                                   int div, int P, float* __restrict__ out) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)P * width) return;
    int p = (int)(i / width), c = (int)(i % width);
    long long q = sel ? sel[p] : p;
    out[i] = src[(q / div) * width + c];
}
int th_gather_rows_launch(const float* src, int width, const int32_t* sel, int div, int P, float* out,
                          hipStream_t s) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(th_cdiv((long long)P * width, 256)), dim3(256), 0, s, src, width, sel,
                       div, P, out);
    TH_LAUNCH_CHECK();
    return 0;
}

// raw[sel[p]] = raw_c[p]; progressive rule (:296-305): rgb stays 0 where sigma <= 0
__global__ void scatter_raw_kernel(const float4* __restrict__ raw_c, const int32_t* __restrict__ sel, int P,
                                   int rgb_all, float4* __restrict__ raw) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float4 r = raw_c[p];
    if (!rgb_all && !(r.w > 0.f)) { r.x = 0.f; r.y = 0.f; r.z = 0.f; }
    raw[sel ? sel[p] : p] = r;
}
int th_scatter_raw_launch(const float* raw_c, const int32_t* sel, int P, int rgb_all, float* raw, hipStream_t s) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(scatter_raw_kernel, dim3(th_cdiv(P, 256)), dim3(256), 0, s, (const float4*)raw_c, sel, P,
                       rgb_all, (float4*)raw);
    TH_LAUNCH_CHECK();
    return 0;
}

size_t th_mlp_ws(int V, int P) {
    size_t rows = (size_t)V * P;
    return 2 * th_align(rows * 256 * 4) + 2 * th_align(rows * 384 * 4) + 2 * th_align((size_t)P * 256 * 4);
}

int th_mlp_forward(const ThMlpPacked& W, int V, int P, const float* h, const float* f, int f_ld, const float* vd,
                   float* raw_c, void* ws, size_t ws_bytes, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE(W.ready, "MLP weights not set (th_set_mlp_weights)");
    TH_REQUIRE(f_ld == 384 || (f_ld == 272 && W.compact_ready),
               "pixel-feature rows must be 384 wide, or 272 wide with upsample_color weights uploaded");
    const bool cf = f_ld == 272;
    const ThPacked& L_ar0 = cf ? W.alpha_res_0c : W.alpha_res_0;
    const ThPacked& L_rr0 = cf ? W.rgb_res_0c : W.rgb_res_0;
    const ThPacked& L_rr1 = cf ? W.rgb_res_1c : W.rgb_res_1;
    TH_REQUIRE(V >= 1 && V <= 4, "supported reference-view counts: 1..4");
    TH_REQUIRE(ws_bytes >= th_mlp_ws(V, P), "workspace too small");
    ThArena ar(ws, ws_bytes);
    const int rows = V * P;
    float* B1 = ar.take<float>((size_t)rows * 256);
    float* B2 = ar.take<float>((size_t)rows * 256);
    float* B3 = ar.take<float>((size_t)rows * 384);
    float* B4 = ar.take<float>((size_t)rows * 384);
    float* M1 = ar.take<float>((size_t)P * 256);
    float* M2 = ar.take<float>((size_t)P * 256);
    TH_REQUIRE(M2 != nullptr, "workspace carve failed");

    // _multiview_agg :313-322
    TH_TRY(th_gemm(h, 256, rows, W.fc_0, TH_ACT_RELU, B1, 256, s));              // s
    TH_TRY(th_gemm(f, f_ld, rows, L_ar0, TH_ACT_RELU, B2, 256, s));       // p
    TH_TRY(th_gemm(B1, 256, rows, W.kv1, TH_ACT_NONE, B3, 384, s));              // ks|vs
    TH_TRY(th_gemm(B2, 256, rows, W.kv0, TH_ACT_NONE, B4, 384, s));              // kp|vp
    dim3 g4(th_cdiv(P, 4));
    switch (V) {
        case 1: hipLaunchKernelGGL(xattn_kernel<1>, g4, dim3(256), 0, s, B4, B3, P, B1); break;
        case 2: hipLaunchKernelGGL(xattn_kernel<2>, g4, dim3(256), 0, s, B4, B3, P, B1); break;
        case 3: hipLaunchKernelGGL(xattn_kernel<3>, g4, dim3(256), 0, s, B4, B3, P, B1); break;
        default: hipLaunchKernelGGL(xattn_kernel<4>, g4, dim3(256), 0, s, B4, B3, P, B1); break;
    }
    TH_TRY(th_gemm(B1, 256, rows, W.fc_1, TH_ACT_RELU, B2, 256, s));
    TH_TRY(th_gemm(B2, 256, rows, W.fc_2, TH_ACT_RELU, B1, 256, s));             // inter
    // _alpha_forward :324-328
    hipLaunchKernelGGL(view_mean_kernel, dim3(th_cdiv((long long)P * 256, 256)), dim3(256), 0, s, B1, P, V, 256, M1);
    TH_TRY(th_gemm(M1, 256, P, W.fc_3, TH_ACT_RELU, M2, 256, s));
    hipLaunchKernelGGL(head_kernel<1>, g4, dim3(256), 0, s, M2, P, 256, W.alpha_w, W.alpha_b, raw_c + 3, 4);
    // _RGB_forward :330-353 (evaluated for every compacted sample; the
    // sigma>0 gate is applied when scattering)
    TH_TRY(th_gemm(B1, 256, rows, W.feature_fc, TH_ACT_NONE, B4, 288, s));
    TH_TRY(th_gemm(f, f_ld, rows, L_rr0, TH_ACT_NONE | TH_GEMM_ACCUM, B4, 288, s));
    hipLaunchKernelGGL(viewdir_fill_kernel, dim3(th_cdiv((long long)rows * 32, 256)), dim3(256), 0, s, vd, P, V, B4);
    TH_TRY(th_gemm(B4, 288, rows, W.view_fc, TH_ACT_RELU, B3, 128, s));
    TH_TRY(th_gemm(f, f_ld, rows, L_rr1, TH_ACT_NONE | TH_GEMM_ACCUM, B3, 128, s));
    hipLaunchKernelGGL(view_mean_kernel, dim3(th_cdiv((long long)P * 128, 256)), dim3(256), 0, s, B3, P, V, 128, M1);
    TH_TRY(th_gemm(M1, 128, P, W.fc_4, TH_ACT_RELU, M2, 128, s));
    hipLaunchKernelGGL(head_kernel<3>, g4, dim3(256), 0, s, M2, P, 128, W.rgb_w, W.rgb_b, raw_c, 4);
    TH_LAUNCH_CHECK();
    return 0;
}
