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
// branch; 2 x 3.3 KB with compact rows) + 124 B; every intermediate of the
// reference's ~40 kernels/chunk (~70 KB/sample of HBM round trips) stays on chip.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "k_mlp_fused_kernel.h"
#include "k_mlp_fused8_kernel.h"

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

// 16x16x32 form (8 waves): lane = 16 g + l holds column cols[wave][ct] + l, the 8 k of 16-byte slot `sigma` of the operand row:
// natural order sigma = 4 t + g; PERM (K = 256 operands, see k_mlp_fused8_kernel.h) sigma = 2 t + (g >> 1) + 16 (g & 1)
struct Cols24 { int c[24]; };
__global__ void pack_fused16_kernel(const float* __restrict__ W, int N, int K, int T, int CT, Cols24 cols, float scale, int perm,
                                    uint4* __restrict__ out) {
    long long total = 8LL * T * CT * 2 * 64;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        int lane = (int)(o & 63);
        long long q = o >> 6;
        int plane = (int)(q & 1); q >>= 1;
        int ct = (int)(q % CT); q /= CT;
        int t = (int)(q % T);
        int wave = (int)(q / T);
        int col = cols.c[wave * CT + ct] + (lane & 15);
        int g = lane >> 4;
        int sigma = perm ? 2 * t + (g >> 1) + 16 * (g & 1) : 4 * t + g;
        int k0 = 8 * sigma;
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
    const int dims[][2] = {{256, 64},  {256, 384}, {256, 272}, {384, 256}, {384, 256}, {256, 256}, {256, 256},
                           {128, 256}, {128, 32},  {256, 384}, {256, 272}, {128, 128}};
    // fc_0pe, ar0, ar0c, kv1, kv0, fc_2, fc_3, vfA, vfD, rst, rstc, fc_4
    for (auto& d : dims) halves += (size_t)d[0] * d[1] * 2 + 4096;
    // the 16x16x32 images of the 8-wave kernel: fc_0pe, kv1, kv0, fc_2, fc_3, vfA, vfD, fc_4 (K padded to 32)
    const int dims16[][2] = {{256, 64}, {384, 256}, {384, 256}, {256, 256}, {256, 256}, {128, 256}, {128, 32}, {128, 128}};
    for (auto& d : dims16) halves += (size_t)d[0] * d[1] * 2 + 4096;
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

// the same layer (same source, same power-of-two scale, same bias storage) as 16x16x32 fragments for 8 waves
static int pack_layer16(const float* w, int N, int K, int CT, const int* cols_host, const FusedLayer& base, int sl2, PackCursor& cur,
                        FusedLayer* out, hipStream_t s) {
    const int T = (K + 31) / 32;
    Cols24 cols{};
    for (int i = 0; i < 8 * CT; ++i) cols.c[i] = cols_host[i];
    uint4* dst = (uint4*)cur.w;
    const long long total = 8LL * T * CT * 2 * 64;
    hipLaunchKernelGGL(pack_fused16_kernel, dim3(256), dim3(256), 0, s, w, N, K, T, CT, cols, ldexpf(1.f, sl2), K == 256 ? 1 : 0, dst);
    TH_LAUNCH_CHECK();
    *out = base;
    out->w = dst;
    out->CT = CT;
    out->KB = T;
    cur.w += th_align((size_t)total * 16);
    return 0;
}

// ---- algebraic fold of value_embed into fc_1 ------------------------------------------------------------
// cross_attention returns n_i = (V1 s_i + b1) + sum_j A[j][i] (V0 p_j + b0)  (cross_transformer.py:145-147)
// and the very next op is fc_1 (linear, :319).  softmax over j sums to 1, hence
//   fc_1(n_i) = (F V1) s_i + sum_j A[j][i] (F V0) p_j + F (b1 + b0) + b_F .
// The fused kernel therefore evaluates the two products F*V1, F*V0 in place of the value projections and has
// no separate fc_1 GEMM: one 256x256 layer per row (8.5 % of the MACs) and one barrier-delimited phase less.
// The products are formed once at weight-upload time with fp64 accumulation.
__global__ void fold_matmul_kernel(const float* __restrict__ F, const float* __restrict__ Vw, float* __restrict__ out) {
    int o = blockIdx.x, k = threadIdx.x;
    double acc = 0.0;
    for (int m = 0; m < 256; ++m) acc += (double)F[o * 256 + m] * (double)Vw[m * 256 + k];
    out[o * 256 + k] = (float)acc;
}
__global__ void fold_bias_kernel(const float* __restrict__ F, const float* __restrict__ bF, const float* __restrict__ b1,
                                 const float* __restrict__ b0, float* __restrict__ out) {
    int o = threadIdx.x;
    double acc = bF ? (double)bF[o] : 0.0;
    for (int m = 0; m < 256; ++m) {
        double bm = (b1 ? (double)b1[m] : 0.0) + (b0 ? (double)b0[m] : 0.0);
        acc += (double)F[o * 256 + m] * bm;
    }
    out[o] = (float)acc;
}

// ---- algebraic fold of feature_fc / rgb_res_0 into view_fc ----------------------------------------------
// _RGB_forward (cross_transformer.py:330-343): features = feature_fc(inter) + rgb_res_0(f); net =
// relu(view_fc(cat(features, viewdir))).  There is no nonlinearity between the two linear maps, so with
// view_fc.W = [Wa (128 x 256) | Wd (128 x 27)]:
//   view_fc(.) = (Wa F) inter + (Wa R0) f + Wd viewdir + [b_vf + Wa (b_F + b_R0)]
// 128-wide products replace the two 256-wide layers (206 848 -> 106 496 MACs per row of the RGB branch, one
// barrier-delimited phase and one [96][256] hi/lo epilogue less).  The products are formed once at weight
// upload with fp64 accumulation; rgb_res_1(f) (added after the relu, :346) rides along as the second
// column-tile family of the same pass over f.
// out[m][n] = sum_k A[m*lda + k] * B[k*ldb + n]
__global__ void fold_mm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, int K, int N,
                               float* __restrict__ out, int ldo) {
    const int m = blockIdx.x;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += (double)A[(long long)m * lda + k] * (double)B[(long long)k * ldb + n];
        out[(long long)m * ldo + n] = (float)acc;
    }
}
// out[m] = b0[m] + sum_k A[m*lda + k] * (b1[k] + b2[k])   (null biases count as zero)
__global__ void fold_mv_kernel(const float* __restrict__ A, int lda, int K, const float* __restrict__ b0,
                               const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = b0 ? (double)b0[m] : 0.0;
    for (int k = 0; k < K; ++k)
        acc += (double)A[(long long)m * lda + k] * ((b1 ? (double)b1[k] : 0.0) + (b2 ? (double)b2[k] : 0.0));
    out[m] = (float)acc;
}
// dst[r][:] = src[r][:] * scale (a power of two: exact)
__global__ void scale_rows_kernel(const float* __restrict__ src, int lds_, int rows, int cols, float scale,
                                  float* __restrict__ dst, int ldd) {
    long long n = (long long)rows * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int r = (int)(i / cols), c = (int)(i % cols);
        dst[(long long)r * ldd + c] = src[(long long)r * lds_ + c] * scale;
    }
}

// Builds the fused image from the fp32 layers.  `store` = th_fused_pack_bytes() of device memory.
int th_fused_pack(const th_mlp_weights* w, const th_linear* folded, void* store, FusedParams* out, hipStream_t s) {
    char* tail = (char*)store + th_fused_pack_bytes() - 64 * 1024;
    PackCursor cur{(char*)store, (float*)tail, (int*)(tail + 40 * 1024)};
    unsigned int* amax = (unsigned int*)(tail + 60 * 1024);
    int c256[8], c128[4], ckv[12], cst[8];
    int c256_16[16], c128_16[8], ckv16[24];       // 8 waves, 16-wide column tiles (k_mlp_fused8_kernel.h)
    for (int wv = 0; wv < 8; ++wv) {
        c256_16[wv * 2] = wv * 32; c256_16[wv * 2 + 1] = wv * 32 + 16;
        c128_16[wv] = wv * 16;
        ckv16[wv * 3] = wv * 16; ckv16[wv * 3 + 1] = 128 + wv * 32; ckv16[wv * 3 + 2] = 128 + wv * 32 + 16;
    }
    for (int wv = 0; wv < 4; ++wv) {
        c256[wv * 2] = wv * 64; c256[wv * 2 + 1] = wv * 64 + 32;
        c128[wv] = wv * 32;
        // stacked [key(128); value(256)]: tile 0 = key cols wave*32, tiles 1,2 = value cols 128 + wave*64 (+32)
        ckv[wv * 3] = wv * 32; ckv[wv * 3 + 1] = 128 + wv * 64; ckv[wv * 3 + 2] = 128 + wv * 64 + 32;
        // stacked [Wa R0 (128); rgb_res_1 (128)]: tile 0 = view_fc cols wave*32, tile 1 = rgb_res_1 cols wave*32
        cst[wv * 2] = wv * 32; cst[wv * 2 + 1] = 128 + wv * 32;
    }
    int sl2, sl2b;
#define PACK_SIMPLE(LAYER, L, N_, K_, CT_, COLS)                                       \
    TH_TRY(layer_scale_log2((L).w, (long long)(N_) * (K_), amax, &sl2, s));            \
    TH_TRY(pack_layer((L).w, (L).b, N_, K_, CT_, COLS, sl2, cur, &out->LAYER, s))
    // scratch for the folds (fp32): wpe [256,63] | XF [128,256] | XR [128,384] | XRc [128,260] | st [256,384] | bias [256]
    float* scratch = nullptr;
    const size_t n_scratch = 256 * 63 + 128 * 256 + 128 * 384 + 128 * 260 + 256 * 384 + 256 + (384 * 256 + 384);
    TH_HIP(hipMalloc((void**)&scratch, n_scratch * 4));
    struct Free { float* p; ~Free() { (void)hipFree(p); } } free_scratch{scratch};
    float* wpe = scratch;
    float* XF = wpe + 256 * 63;
    float* XR = XF + 128 * 256;
    float* XRc = XR + 128 * 384;
    float* st = XRc + 128 * 260;
    float* stb = st + 256 * 384;
    float* tw = stb + 256;
    {   // fc_0: only the 63 positional-encoding columns [192, 255) stay in the kernel (token columns -> T', th_api.hip)
        TH_HIP(hipMemcpy2DAsync(wpe, 63 * 4, w->fc_0.w + 192, 255 * 4, 63 * 4, 256, hipMemcpyDeviceToDevice, s));
        TH_TRY(layer_scale_log2(wpe, 256LL * 63, amax, &sl2, s));
        TH_TRY(pack_layer(wpe, w->fc_0.b, 256, 63, 2, c256, sl2, cur, &out->fc_0pe, s));
        TH_TRY(pack_layer16(wpe, 256, 63, 2, c256_16, out->fc_0pe, sl2, cur, &out->w16.fc_0pe, s));
    }
    PACK_SIMPLE(ar0, w->alpha_res_0, 256, 384, 2, c256);
    PACK_SIMPLE(fc_2, w->fc_2, 256, 256, 2, c256);
    TH_TRY(pack_layer16(w->fc_2.w, 256, 256, 2, c256_16, out->fc_2, sl2, cur, &out->w16.fc_2, s));
    PACK_SIMPLE(fc_3, w->fc_3, 256, 256, 2, c256);
    TH_TRY(pack_layer16(w->fc_3.w, 256, 256, 2, c256_16, out->fc_3, sl2, cur, &out->w16.fc_3, s));
    PACK_SIMPLE(fc_4, w->fc_4, 128, 128, 1, c128);
    TH_TRY(pack_layer16(w->fc_4.w, 128, 128, 1, c128_16, out->fc_4, sl2, cur, &out->w16.fc_4, s));
#undef PACK_SIMPLE
    out->compact_ready = false;
    if (folded) {
        TH_TRY(layer_scale_log2(folded[0].w, 256LL * 260, amax, &sl2, s));
        TH_TRY(pack_layer(folded[0].w, folded[0].b, 256, 260, 2, c256, sl2, cur, &out->ar0c, s));
    }
    // ---- RGB branch: view_fc folded over feature_fc / rgb_res_0 (see above) ----
    {
        const float* Wvf = w->view_fc.w;                  // [128, 283] = [Wa | Wd]
        hipLaunchKernelGGL(fold_mm_kernel, dim3(128), dim3(256), 0, s, Wvf, 283, w->feature_fc.w, 256, 256, 256, XF, 256);
        hipLaunchKernelGGL(fold_mm_kernel, dim3(128), dim3(256), 0, s, Wvf, 283, w->rgb_res_0.w, 384, 256, 384, XR, 384);
        if (folded)
            hipLaunchKernelGGL(fold_mm_kernel, dim3(128), dim3(256), 0, s, Wvf, 283, folded[1].w, 260, 256, 260, XRc, 260);
        TH_LAUNCH_CHECK();
        // the three K ranges of the folded view_fc accumulate into ONE register tile -> one scale
        TH_TRY(layer_scale_log2(XF, 128LL * 256, amax, &sl2, s));
        TH_TRY(layer_scale_log2(XR, 128LL * 384, amax, &sl2b, s));
        if (sl2b < sl2) sl2 = sl2b;
        if (folded) {
            TH_TRY(layer_scale_log2(XRc, 128LL * 260, amax, &sl2b, s));
            if (sl2b < sl2) sl2 = sl2b;
        }
        {   // Wd: strided view of view_fc -> contiguous [128, 27] in `st`
            TH_HIP(hipMemcpy2DAsync(st, 27 * 4, Wvf + 256, 283 * 4, 27 * 4, 128, hipMemcpyDeviceToDevice, s));
            TH_TRY(layer_scale_log2(st, 128LL * 27, amax, &sl2b, s));
            if (sl2b < sl2) sl2 = sl2b;
            TH_TRY(pack_layer(st, nullptr, 128, 27, 1, c128, sl2, cur, &out->vfD, s));
            TH_TRY(pack_layer16(st, 128, 27, 1, c128_16, out->vfD, sl2, cur, &out->w16.vfD, s));
        }
        const int s_vf = sl2;
        TH_TRY(pack_layer(XF, nullptr, 128, 256, 1, c128, s_vf, cur, &out->vfA, s));
        TH_TRY(pack_layer16(XF, 128, 256, 1, c128_16, out->vfA, s_vf, cur, &out->w16.vfA, s));
        for (int variant = 0; variant < (folded ? 2 : 1); ++variant) {
            const int K = variant ? 260 : 384;
            const float* xr = variant ? XRc : XR;
            const th_linear& r0 = variant ? folded[1] : w->rgb_res_0;
            const th_linear& r1 = variant ? folded[2] : w->rgb_res_1;
            int s_r1;
            TH_TRY(layer_scale_log2(r1.w, 128LL * K, amax, &s_r1, s));
            hipLaunchKernelGGL(scale_rows_kernel, dim3(128), dim3(256), 0, s, xr, K, 128, K, ldexpf(1.f, s_vf), st, K);
            hipLaunchKernelGGL(scale_rows_kernel, dim3(128), dim3(256), 0, s, r1.w, K, 128, K, ldexpf(1.f, s_r1),
                               st + 128 * K, K);
            // bias rows: [b_vf + Wa (b_F + b_R0) | b_R1]
            hipLaunchKernelGGL(fold_mv_kernel, dim3(1), dim3(128), 0, s, Wvf, 283, 256, w->view_fc.b, w->feature_fc.b,
                               r0.b, stb);
            if (r1.b) TH_HIP(hipMemcpyAsync(stb + 128, r1.b, 128 * 4, hipMemcpyDeviceToDevice, s));
            else TH_HIP(hipMemsetAsync(stb + 128, 0, 128 * 4, s));
            TH_LAUNCH_CHECK();
            FusedLayer* dst = variant ? &out->rstc : &out->rst;
            TH_TRY(pack_layer(st, stb, 256, K, 2, cst, 0, cur, dst, s));
            dst->inv_scale = ldexpf(1.f, -s_vf);
            dst->inv_scale2 = ldexpf(1.f, -s_r1);
        }
        out->vfA.inv_scale = out->vfD.inv_scale = ldexpf(1.f, -s_vf);
        out->compact_ready = folded != nullptr;
    }
    // stacked key/value layers need a contiguous [384,256] weight + [384] bias
    float* tb = tw + 384 * 256;
    for (int which = 0; which < 2; ++which) {
        const th_linear& k = which == 0 ? w->key1 : w->key0;
        const th_linear& v = which == 0 ? w->val1 : w->val0;
        TH_HIP(hipMemcpyAsync(tw, k.w, 128 * 256 * 4, hipMemcpyDeviceToDevice, s));
        // value rows <- fc_1.W * value_embed.W (fold, see above); their bias moves into fc_1's folded bias
        hipLaunchKernelGGL(fold_matmul_kernel, dim3(256), dim3(256), 0, s, w->fc_1.w, v.w, tw + 128 * 256);
        if (k.b) TH_HIP(hipMemcpyAsync(tb, k.b, 128 * 4, hipMemcpyDeviceToDevice, s));
        else TH_HIP(hipMemsetAsync(tb, 0, 128 * 4, s));
        TH_HIP(hipMemsetAsync(tb + 128, 0, 256 * 4, s));
        TH_TRY(layer_scale_log2(tw, 384LL * 256, amax, &sl2, s));
        TH_TRY(pack_layer(tw, tb, 384, 256, 3, ckv, sl2, cur, which == 0 ? &out->kv1 : &out->kv0, s));
        TH_TRY(pack_layer16(tw, 384, 256, 3, ckv16, which == 0 ? out->kv1 : out->kv0, sl2, cur, which == 0 ? &out->w16.kv1 : &out->w16.kv0, s));
    }
    // folded bias of fc_1 (fc_1 itself lives inside the value projections)
    hipLaunchKernelGGL(fold_bias_kernel, dim3(1), dim3(256), 0, s, w->fc_1.w, w->fc_1.b, w->val1.b, w->val0.b, cur.bias);
    out->fc_1 = FusedLayer{};
    out->fc_1.bias = cur.bias;
    cur.bias += 256;
    TH_HIP(hipStreamSynchronize(s));
    TH_REQUIRE(cur.w <= tail, "fused pack overflow");
    TH_REQUIRE((char*)cur.bias <= tail + 40 * 1024 && (char*)cur.cols <= tail + 60 * 1024, "fused pack tail overflow");
    return 0;
}

// ---- per-frame token table as split rows (TH_ROWS_NBR path) ---------------------------------------------------
__global__ __launch_bounds__(256) void tok_split_kernel(float* __restrict__ t, int rows, float* __restrict__ sc,
                                                        unsigned int* __restrict__ range) {
    const unsigned bits = reinterpret_cast<const unsigned*>(sc)[1];
    const int E = (int)((bits >> 23) & 255u);
    int sl2 = 0;                                           // zero / non-finite table: scale 1
    if (E == 255) { if (range != nullptr && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(range + TH_RANGE_VIT, 0x7c00u); }
    else if (bits != 0u) sl2 = min(139 - E, 100);          // max |T'| = m 2^e, m in [0.5, 1): e = E - 126, k = 13 - e
    const float scale = __builtin_bit_cast(float, (unsigned)(sl2 + 127) << 23);
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[0] = __builtin_bit_cast(float, (unsigned)(127 - sl2) << 23);
    const int lane = threadIdx.x & 63;
    // one wave per row, in place: every lane holds its float4 before the first half is written over the row
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        float* row = t + (long long)r * 256;
        const float4 x = reinterpret_cast<const float4*>(row)[lane];
        const float v[4] = {x.x * scale, x.y * scale, x.z * scale, x.w * scale};
        h4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 hi, lo;
            split_h(v[e], hi, lo);
            hv[e] = hi;
            lv[e] = lo;
        }
        __builtin_amdgcn_s_waitcnt(0);                    // (the loads of ALL lanes have returned: same instruction)
        _Float16* rh = reinterpret_cast<_Float16*>(row);
        *reinterpret_cast<h4*>(rh + 4 * lane) = hv;
        *reinterpret_cast<h4*>(rh + 256 + 4 * lane) = lv;
    }
}

int th_tok_split(float* tprime, int rows, float* sc, unsigned int* range, hipStream_t s) {
    if (rows <= 0) return 0;
    unsigned int* amax = reinterpret_cast<unsigned int*>(sc) + 1;
    TH_HIP(hipMemsetAsync(amax, 0, 4, s));
    hipLaunchKernelGGL(absmax_kernel, dim3(64), dim3(256), 0, s, tprime, (long long)rows * 256, amax);
    hipLaunchKernelGGL(tok_split_kernel, dim3(th_cdiv(rows, 4) < 1024 ? th_cdiv(rows, 4) : 1024), dim3(256), 0, s, tprime, rows, sc,
                       range);
    TH_LAUNCH_CHECK();
    return 0;
}

// map_split: TH_MAP_SPLIT ([V][H*W][256] latents, then [V][H*W][4] colours); fold: [2][V][H*W][256] (fold0, then fold12)
int th_map_fold_launch(const FusedParams& base, const float* map_split, int V, int H, int W, const int32_t* box, float* fold,
                       unsigned int* range, hipStream_t s, const unsigned* demand) {
    TH_REQUIRE(base.compact_ready, "the map fold needs the colour-folded layers (th_mlp_weights.upsample_color)");
    TH_REQUIRE(V >= 1 && H >= 1 && W >= 1 && map_split && fold, "bad argument");
    MapFoldParams p;
    p.ar0 = base.ar0c; p.rst = base.rstc;
    p.lat = map_split; p.rgb = map_split + (size_t)V * H * W * 256;
    p.box = box; p.V = V; p.H = H; p.W = W;
    p.out0 = fold; p.out12 = fold + (size_t)V * H * W * 256;
    p.range = range;
    p.list = nullptr; p.count = nullptr;
    static unsigned long long attr_done = 0ull;
    if (th_lds_attr_needed(&attr_done)) {
        TH_HIP(hipFuncSetAttribute((const void*)map_fold_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MF_TEX * STR272));
        TH_HIP(hipFuncSetAttribute((const void*)map_fold_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MF_TEX * STR272));
    }
    if (demand != nullptr) {
        // demand-driven map (k_demand.hip): the listed texels only.  The list's length lives on the device: two workgroups per CU
        // walk it in strides of the grid.
        const size_t NW = (size_t)V * H * W / 32;
        p.count = demand + 2 * NW;
        p.list = reinterpret_cast<const int32_t*>(demand + 2 * NW + 16);
        hipLaunchKernelGGL(map_fold_kernel<true>, dim3(512), dim3(256), 2 * MF_TEX * STR272, s, p);
        TH_LAUNCH_CHECK();
        return 0;
    }
    const int tpr = (W + MF_TEX - 1) / MF_TEX;
    hipLaunchKernelGGL(map_fold_kernel<false>, dim3((unsigned)(V * H * tpr)), dim3(256), 2 * MF_TEX * STR272, s, p);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_mlp_fused_forward(const FusedParams& base, const ThMlpPacked& heads, int V, int P, const float* stok,
                         const void* pe, const void* f, int f_ld, const float* vd, const int32_t* vd_sel, int vd_div, int rgb_all, float* raw_c,
                         unsigned int* range, hipStream_t s, const void* tsplit, const float* t_inv, int t_nc, const float* tex_map,
                         size_t tex_stride) {
    if (P <= 0) return 0;
    TH_REQUIRE(V >= 1 && V <= 3, "fused MLP supports 1..3 reference views");
    TH_REQUIRE(f_ld == 384 || (f_ld == 272 && base.compact_ready),
               "pixel-feature rows must be 384 wide, or 272 wide with upsample_color weights uploaded");
    const bool cf = f_ld == 272;
    FusedParams p = base;
    if (cf) { p.ar0 = base.ar0c; p.rst = base.rstc; }
    p.alpha_w = heads.alpha_w; p.alpha_b = heads.alpha_b; p.rgb_w = heads.rgb_w; p.rgb_b = heads.rgb_b;
    TH_REQUIRE(tsplit == nullptr || (t_inv != nullptr && t_nc >= 7 && t_nc <= 4096), "split token table: scale word and 7 <= N_c <= 4096");
    p.tsplit = (const _Float16*)tsplit; p.t_inv = t_inv; p.t_nc = t_nc;
    const bool tex = tex_map != nullptr;
    TH_REQUIRE(!tex || (cf && f != nullptr), "texel hand-over: compact (272-wide) operand planes only");
    p.tex_hdr = nullptr; p.tex_rec = nullptr; p.tex_map = nullptr; p.tex_map2 = nullptr;
    if (tex) {          // `f` is K5t's block: tile headers, then the per-row records (th_pixtex_launch)
        p.tex_hdr = (const unsigned*)f;
        p.tex_rec = p.tex_hdr + (size_t)th_cdiv(P, FM_PTS) * 512;
        p.tex_map = tex_map;                                 // fold0, fold12 behind it (th_map_fold_launch)
        p.tex_map2 = tex_map + tex_stride;
        f = nullptr;
    }
    p.stok = stok; p.pe = (const _Float16*)pe; p.f = (const _Float16*)f; p.vd = vd; p.vd_sel = vd_sel; p.vd_div = vd_div > 0 ? vd_div : 1; p.raw_c = raw_c; p.P = P; p.rgb_all = rgb_all; p.range = range;
    static unsigned long long attr_done = 0ull;
    if (th_lds_attr_needed(&attr_done)) {
#define FM_ATTR(V_, F_)                                                                                       \
    TH_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<V_, F_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               FUSED_LDS_BYTES))
        FM_ATTR(1, 0); FM_ATTR(2, 0); FM_ATTR(3, 0); FM_ATTR(1, 1); FM_ATTR(2, 1); FM_ATTR(3, 1);
#undef FM_ATTR
#define FM_ATTR_T(V_)                                                                                              \
    TH_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<V_, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               FUSED_LDS_BYTES))
        FM_ATTR_T(1); FM_ATTR_T(2); FM_ATTR_T(3);
#undef FM_ATTR_T
    }
    dim3 grid(tex ? 8 * th_cdiv(th_cdiv(P, FM_PTS), 8) : th_cdiv(P, FM_PTS));     // (TEX: XCD-contiguous tile order)
    // the 8-wave kernel (two waves per SIMD) serves the hand-overs the frame-level entry points run: texel lists + neighbour records
    const bool eight = base.waves == 8 && tex && tsplit != nullptr;
    static unsigned long long attr8_done = 0ull;
    if (eight && th_lds_attr_needed(&attr8_done)) {
        TH_HIP(hipFuncSetAttribute((const void*)mlp_fused8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_BYTES));
        TH_HIP(hipFuncSetAttribute((const void*)mlp_fused8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_BYTES));
        TH_HIP(hipFuncSetAttribute((const void*)mlp_fused8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_BYTES));
    }
    // developer aid: TH_FUSED_DBG=1 -> average cycles between barriers over every 16th tile (first big launch only)
    static int dbg_state = getenv("TH_FUSED_DBG") ? 1 : 0;
    static long long* dbg_dev = nullptr;
    p.dbg = base.cycles_buf;                     // (th_fused_cycles: the caller's counters, every launch)
    const bool dbg_now = dbg_state == 1 && P >= 4096 && base.cycles_buf == nullptr;
    hipEvent_t dbg_e0 = nullptr, dbg_e1 = nullptr;
    if (dbg_now) {
        if (!dbg_dev) TH_HIP(hipMalloc((void**)&dbg_dev, 64 * sizeof(long long)));
        TH_HIP(hipMemsetAsync(dbg_dev, 0, 64 * sizeof(long long), s));
        p.dbg = dbg_dev;
        TH_HIP(hipEventCreate(&dbg_e0));
        TH_HIP(hipEventCreate(&dbg_e1));
        TH_HIP(hipStreamSynchronize(s));                   // (the launch is timed alone on its stream)
        TH_HIP(hipEventRecord(dbg_e0, s));
    }
#define FM_LAUNCH(V_, F_) hipLaunchKernelGGL((mlp_fused_kernel<V_, F_>), grid, dim3(256), FUSED_LDS_BYTES, s, p)
#define FM_LAUNCH_T(V_) hipLaunchKernelGGL((mlp_fused_kernel<V_, 1, true>), grid, dim3(256), FUSED_LDS_BYTES, s, p)
#define FM_LAUNCH_8(V_) hipLaunchKernelGGL((mlp_fused8_kernel<V_>), grid, dim3(F8_THREADS), F8_LDS_BYTES, s, p)
    if (eight) {
        if (V == 1) FM_LAUNCH_8(1); else if (V == 2) FM_LAUNCH_8(2); else FM_LAUNCH_8(3);
    } else if (tex) {
        if (V == 1) FM_LAUNCH_T(1); else if (V == 2) FM_LAUNCH_T(2); else FM_LAUNCH_T(3);
    } else
    switch (V * 2 + (cf ? 1 : 0)) {
        case 2: FM_LAUNCH(1, 0); break;
        case 3: FM_LAUNCH(1, 1); break;
        case 4: FM_LAUNCH(2, 0); break;
        case 5: FM_LAUNCH(2, 1); break;
        case 6: FM_LAUNCH(3, 0); break;
        default: FM_LAUNCH(3, 1); break;
    }
#undef FM_LAUNCH
    // developer aid: TH_FUSED_CHECK=1 -> the first big 8-wave launch is repeated by the 4-wave kernel into a scratch buffer and the two
    // raw outputs are compared on the host (per channel: max difference, worst sample)
    static int chk_state = getenv("TH_FUSED_CHECK") ? 1 : 0;
    if (eight && chk_state == 1 && P >= 4096) {
        chk_state = 2;
        float* raw4 = nullptr;
        TH_HIP(hipMalloc((void**)&raw4, (size_t)P * 16));
        FusedParams q = p;
        q.raw_c = raw4; q.dbg = nullptr; q.range = nullptr;
        if (V == 1) hipLaunchKernelGGL((mlp_fused_kernel<1, 1, true>), grid, dim3(256), FUSED_LDS_BYTES, s, q);
        else if (V == 2) hipLaunchKernelGGL((mlp_fused_kernel<2, 1, true>), grid, dim3(256), FUSED_LDS_BYTES, s, q);
        else hipLaunchKernelGGL((mlp_fused_kernel<3, 1, true>), grid, dim3(256), FUSED_LDS_BYTES, s, q);
        TH_HIP(hipStreamSynchronize(s));
        float* h8v = (float*)malloc((size_t)P * 16);
        float* h4v = (float*)malloc((size_t)P * 16);
        TH_HIP(hipMemcpy(h8v, raw_c, (size_t)P * 16, hipMemcpyDeviceToHost));
        TH_HIP(hipMemcpy(h4v, raw4, (size_t)P * 16, hipMemcpyDeviceToHost));
        double mx[4] = {0, 0, 0, 0};
        long long wi[4] = {0, 0, 0, 0}, nbad[4] = {0, 0, 0, 0};
        for (long long i = 0; i < P; ++i)
            for (int ch = 0; ch < 4; ++ch) {
                const double d = fabs((double)h8v[i * 4 + ch] - (double)h4v[i * 4 + ch]);
                if (!(d <= mx[ch])) { mx[ch] = d; wi[ch] = i; }
                if (!(d <= 1e-4)) ++nbad[ch];
            }
        for (int ch = 0; ch < 4; ++ch)
            fprintf(stderr, "[TH_FUSED_CHECK] raw ch %d: max |8w - 4w| = %.3e at sample %lld (tile %lld, row %lld): %.6f vs %.6f; %lld of %d above 1e-4\n", ch,
                    mx[ch], wi[ch], wi[ch] / 32, wi[ch] % 32, h8v[wi[ch] * 4 + ch], h4v[wi[ch] * 4 + ch], nbad[ch], P);
        // the first tile with a bad green value, row by row
        for (long long i = 0; i < P; ++i)
            if (fabs(h8v[i * 4 + 1] - h4v[i * 4 + 1]) > 1e-4) {
                const long long t0 = i / 32 * 32;
                for (long long r = t0; r < t0 + 32 && r < P; ++r)
                    fprintf(stderr, "   row %2lld  8w %9.5f %9.5f %9.5f %9.5f   4w %9.5f %9.5f %9.5f %9.5f\n", r - t0, h8v[r * 4], h8v[r * 4 + 1], h8v[r * 4 + 2],
                            h8v[r * 4 + 3], h4v[r * 4], h4v[r * 4 + 1], h4v[r * 4 + 2], h4v[r * 4 + 3]);
                break;
            }
        free(h8v); free(h4v);
        TH_HIP(hipFree(raw4));
    }
#undef FM_LAUNCH_T
#undef FM_LAUNCH_8
    TH_LAUNCH_CHECK();
    if (dbg_now) {
        long long st[64];
        TH_HIP(hipEventRecord(dbg_e1, s));
        TH_HIP(hipStreamSynchronize(s));
        float dbg_ms = 0.f;
        TH_HIP(hipEventElapsedTime(&dbg_ms, dbg_e0, dbg_e1));
        fprintf(stderr, "[TH_FUSED_DBG] launch %.3f ms (events), %d tiles\n", dbg_ms, grid.x);
        TH_HIP(hipMemcpy(st, dbg_dev, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "[TH_FUSED_DBG] %lld tiles sampled of %d, average cycles between barriers:", st[0], grid.x);
        long long tot = 0;
        for (int i = 1; i < 62 && st[i] != 0; ++i) {
            fprintf(stderr, " %lld", st[i] / (st[0] > 0 ? st[0] : 1));
            tot += st[i] / (st[0] > 0 ? st[0] : 1);
        }
        fprintf(stderr, "  | total %lld", tot);
        if (st[63] > 0)      // whole sampled tiles: shader cycles per 100 MHz tick = the shader clock the chip ran at INSIDE this launch
            fprintf(stderr, "  | %.1f us per tile at %.3f GHz inside the launch", (double)st[63] / (double)(st[0] > 0 ? st[0] : 1) * 0.01,
                    (double)st[62] / ((double)st[63] * 10.0));
        fprintf(stderr, "\n");
        dbg_state = 2;
    }
    return 0;
}
