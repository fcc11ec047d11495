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
#include <stdlib.h>

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
// softmax(q k^T * 0.125) v per (view, head), flash-style: 64-key tiles, online softmax in registers, the
// [V,3,N,N] probability tensor never exists.  Both products run on v_mfma_f32_16x16x32_f16 with the operands
// split into fp16 hi + lo (x = hi + lo to 2^-22; three products hi*hi + hi*lo + lo*hi, fp32 accumulate -- the
// same fp32-class scheme as the fused MLP): 48 MFMAs of 16 cycles per tile and wave instead of the
// 128 fp32 MFMAs of 32 cycles the 16x16x4 form needs.
//   S = Q K^T : A = Q [16 q][k = d]   (registers, split once per workgroup)
//               B = K^T: lane (key = nt*16 + lane&15, d = 32 s + 8 (lane>>4) ..+7) -> 16 bytes of a K row in LDS
//   O = P V   : A = P [16 q][k = key] (softmax output, C layout -> LDS -> A layout, split)
//               B = V  : lane (d = dt*16 + lane&15, key = 32 s + 8 (lane>>4) ..+7) -> V is staged TRANSPOSED
// One workgroup = 4 waves = 64 queries; K / V tiles are fetched into registers one tile ahead.
#define AT_Q 64
#define AT_K 64
#define AT_HS 144      // LDS row stride in bytes of a 64-half row (+16: an odd number of 16-byte slots)

typedef _Float16 at_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 at_h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void at_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, int N, int dim, float scale,
                                                   float* __restrict__ out) {
    // planes: [hi | lo][64 rows][AT_HS bytes]
    __shared__ __attribute__((aligned(16))) char Ks[2 * AT_K * AT_HS];      // K  [key][d]
    __shared__ __attribute__((aligned(16))) char Vt[2 * 64 * AT_HS];        // V^T [d][key]
    __shared__ __attribute__((aligned(16))) char Ps[4][2 * 16 * AT_HS];     // P  [q][key] per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y, view = blockIdx.z;
    const int q0 = blockIdx.x * AT_Q + wave * 16;
    const int ld = 3 * dim;
    const float* base = qkv + (long long)view * N * ld;
    const int qcol = head * 64, kcol = dim + head * 64, vcol = 2 * dim + head * 64;

    // Q fragments (A operand): row q0 + (lane&15), d = 32 s + 8 (lane>>4) .. +7
    at_h8 qh[2], ql[2];
    {
        const int qi = q0 + (lane & 15);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = 0.f;
            if (qi < N) {
                const float* src = base + (long long)qi * ld + qcol + 32 * s2 + 8 * (lane >> 4);
                float4 a = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
                v8[0] = a.x; v8[1] = a.y; v8[2] = a.z; v8[3] = a.w; v8[4] = b4.x; v8[5] = b4.y; v8[6] = b4.z; v8[7] = b4.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 x, y;
                at_split(v8[e], x, y);
                qh[s2][e] = x; ql[s2][e] = y;
            }
        }
    }
    f32x4 oacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) oacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrun[r] = -3.0e38f; lrun[r] = 0.f; }

    // K / V tiles (64 keys x 64 floats each = 1024 float4 per tile): fetched into registers one tile ahead
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
    char* Pw = Ps[wave];
    for (int k0 = 0; k0 < N; k0 += AT_K) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int idx = tid + 256 * i;
            int row = idx >> 4, c4 = idx & 15;                  // key row, 4 consecutive d
            const float kv[4] = {kreg[i].x, kreg[i].y, kreg[i].z, kreg[i].w};
            const float vv[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
            at_h4 kh, kl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 x, y;
                at_split(kv[e], x, y);
                kh[e] = x; kl[e] = y;
                at_split(vv[e], x, y);
                *reinterpret_cast<_Float16*>(Vt + (4 * c4 + e) * AT_HS + 2 * row) = x;                     // V^T[d][key]
                *reinterpret_cast<_Float16*>(Vt + 64 * AT_HS + (4 * c4 + e) * AT_HS + 2 * row) = y;
            }
            *reinterpret_cast<at_h4*>(Ks + row * AT_HS + 8 * c4) = kh;
            *reinterpret_cast<at_h4*>(Ks + AT_K * AT_HS + row * AT_HS + 8 * c4) = kl;
        }
        __syncthreads();
        if (k0 + AT_K < N) fetch_kv(k0 + AT_K);
        // ---- S = Q K^T (C layout: row q = 4*(lane>>4)+r, col key = nt*16 + (lane&15)) ----
        f32x4 sacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            sacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int off = (nt * 16 + (lane & 15)) * AT_HS + 64 * s2 + 16 * (lane >> 4);
                const at_h8 bh = *reinterpret_cast<const at_h8*>(Ks + off);
                const at_h8 bl = *reinterpret_cast<const at_h8*>(Ks + AT_K * AT_HS + off);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql[s2], bh, sacc[nt], 0, 0, 0);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[s2], bl, sacc[nt], 0, 0, 0);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[s2], bh, sacc[nt], 0, 0, 0);
            }
        }
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
            m = th_row16_max(m);
            mnew[r] = fmaxf(mrun[r], m);
            corr[r] = expf(mrun[r] - mnew[r]);
            mrun[r] = mnew[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float ls = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int key = k0 + nt * 16 + (lane & 15);
                float pv = (key < N) ? expf(sacc[nt][r] - mnew[r]) : 0.f;
                ls += pv;
                _Float16 x, y;
                at_split(pv, x, y);
                const int po = (4 * (lane >> 4) + r) * AT_HS + 2 * (nt * 16 + (lane & 15));
                *reinterpret_cast<_Float16*>(Pw + po) = x;
                *reinterpret_cast<_Float16*>(Pw + 16 * AT_HS + po) = y;
            }
            ls = th_row16_sum(ls);
            lrun[r] = lrun[r] * corr[r] + ls;
#pragma unroll
            for (int j = 0; j < 4; ++j) oacc[j][r] *= corr[r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): P tile of this wave is written
        __builtin_amdgcn_wave_barrier();
        // ---- O += P V ----
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int pof = (lane & 15) * AT_HS + 64 * s2 + 16 * (lane >> 4);
            const at_h8 ph = *reinterpret_cast<const at_h8*>(Pw + pof);
            const at_h8 pl = *reinterpret_cast<const at_h8*>(Pw + 16 * AT_HS + pof);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int vof = (j * 16 + (lane & 15)) * AT_HS + 64 * s2 + 16 * (lane >> 4);
                const at_h8 vh = *reinterpret_cast<const at_h8*>(Vt + vof);
                const at_h8 vl = *reinterpret_cast<const at_h8*>(Vt + 64 * AT_HS + vof);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl, vh, oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vl, oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vh, oacc[j], 0, 0, 0);
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


// ---- attention, second form: operands split ONCE per layer, no LDS staging ------------------------------------
// attn_kernel converts every K / V tile fp32 -> fp16 hi/lo inside every workgroup (the V tile transposed with
// 2-byte LDS stores) between two block barriers: 8 q-blocks repeat the conversion of the same (view, head) and
// the 8-tile loop is a 31 us dependent chain.  Here kv_split_kernel writes, once per layer,
//   Kp [V][heads][2 planes][Npad][64]   K rows as fp16 hi | lo          (rows >= N zero)
//   Vp [V][heads][2 planes][64][Npad]   V^T rows (one d; keys in the fragment order of attn2_kernel) as fp16 hi | lo
// -- exactly the 16-byte B-operand fragments of the two products -- and attn2_kernel (ONE wave = 16 queries per
// workgroup, 288 workgroups at N = 500) loads them straight from L2 into registers: no block barrier, no conversion
// in the loop, V fragments of a tile requested before its S product, K fragments of the next tile before its P V.
// (Splitting the keys of a query block over 4 waves with a final merge, and feeding the LDS tiling from the pre-split
// planes, were both measured: no gain / slower -- this form is bound by its L2 traffic, the LDS form by its barriers.)
__global__ __launch_bounds__(256) void kv_split_kernel(const float* __restrict__ qkv, int N, int Npad, int dim,
                                                       _Float16* __restrict__ Kp, _Float16* __restrict__ Vp) {
    __shared__ float vt[64][65];
    const int tid = threadIdx.x, head = blockIdx.y, view = blockIdx.z, heads = gridDim.y;
    const int k0 = blockIdx.x * 64;
    const int ld = 3 * dim;
    const float* base = qkv + (long long)view * N * ld;
    const long long plane = (long long)Npad * 64;
    _Float16* kp = Kp + ((long long)view * heads + head) * 2 * plane;
    _Float16* vp = Vp + ((long long)view * heads + head) * 2 * plane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i, row = idx >> 4, c4 = idx & 15, key = k0 + row;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (key < N) {
            kv = *reinterpret_cast<const float4*>(base + (long long)key * ld + dim + head * 64 + 4 * c4);
            vv = *reinterpret_cast<const float4*>(base + (long long)key * ld + 2 * dim + head * 64 + 4 * c4);
        }
        const float kk[4] = {kv.x, kv.y, kv.z, kv.w};
        at_h4 kh, kl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 x, y;
            at_split(kk[e], x, y);
            kh[e] = x; kl[e] = y;
        }
        *reinterpret_cast<at_h4*>(kp + (long long)key * 64 + 4 * c4) = kh;
        *reinterpret_cast<at_h4*>(kp + plane + (long long)key * 64 + 4 * c4) = kl;
        vt[row][4 * c4 + 0] = vv.x; vt[row][4 * c4 + 1] = vv.y; vt[row][4 * c4 + 2] = vv.z; vt[row][4 * c4 + 3] = vv.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i, d = idx >> 4, c4 = idx & 15;     // d row, 4 consecutive keys
        at_h4 vh, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            _Float16 x, y;
            at_split(vt[4 * c4 + e][d], x, y);
            vh[e] = x; vl[e] = y;
        }
        // keys 4 c4 .. 4 c4 + 3 of the 64-key block; within a 32-key block key offset 16 n + 4 g + j sits at position
        // 8 g + 4 n + j (the B-operand order of attn2_kernel's P^T fragments)
        const int ko = (4 * c4) & 31, pos = ((4 * c4) & 32) + 8 * ((ko & 15) >> 2) + 4 * (ko >> 4);
        *reinterpret_cast<at_h4*>(vp + (long long)d * Npad + k0 + pos) = vh;
        *reinterpret_cast<at_h4*>(vp + plane + (long long)d * Npad + k0 + pos) = vl;
    }
}

// The products are formed TRANSPOSED: S^T = K Q^T and O^T = V^T P^T.  An MFMA result tile holds, per lane, 4 consecutive
// rows of ONE column; with queries as columns a lane owns one query (lane & 15) and its 16 scores of a 64-key tile sit
// in its own registers: the row maximum / sum are in-register reductions plus two cross-row steps, the probabilities are
// already laid out as the B operand of the second product (k index = 8 (lane >> 4) + j  <->  key 16 (j >> 2) + 4 (lane >> 4)
// + (j & 3) of a 32-key block: kv_split_kernel stores V^T in that key order), and the softmax statistics of a query live
// in the lane that holds its output column -- no LDS, no barrier, no cross-lane traffic for the rescaling.  (The
// row-major form wrote P through LDS with 32 two-byte stores per tile and reduced 4 rows x 4 steps across lanes:
// 21.4 us per layer at N = 500; this form 16.6 us.  Splitting the keys of a query block over 4 waves with an LDS merge
// of the partial (m, l, O^T) triples was measured again on this form: 17.7 us -- the launch is not bound by the length of
// the per-wave chain.)
__global__ __launch_bounds__(64) void attn2_kernel(const float* __restrict__ qkv, const _Float16* __restrict__ Kp,
                                                   const _Float16* __restrict__ Vp, int N, int Npad, int dim,
                                                   float scale, float* __restrict__ out) {
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int kbeg = 0, kend = N;
    const int head = blockIdx.y, view = blockIdx.z, heads = gridDim.y;
    const int q0 = blockIdx.x * 16;
    const int ld = 3 * dim;
    const float* base = qkv + (long long)view * N * ld;
    const long long plane = (long long)Npad * 64;
    const _Float16* kp = Kp + ((long long)view * heads + head) * 2 * plane;
    const _Float16* vp = Vp + ((long long)view * heads + head) * 2 * plane;

    // Q^T fragments (B operand: column = query c, k = d = 32 s2 + 8 g + j)
    at_h8 qh[2], ql[2];
    {
        const int qi = q0 + c;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = 0.f;
            if (qi < N) {
                const float* src = base + (long long)qi * ld + head * 64 + 32 * s2 + 8 * g;
                float4 a = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
                v8[0] = a.x; v8[1] = a.y; v8[2] = a.z; v8[3] = a.w; v8[4] = b4.x; v8[5] = b4.y; v8[6] = b4.z; v8[7] = b4.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 x, y;
                at_split(v8[e], x, y);
                qh[s2][e] = x; ql[s2][e] = y;
            }
        }
    }
    f32x4 oacc[4];              // O^T tiles: d = 16 jd + 4 g + r, column = query c
#pragma unroll
    for (int j = 0; j < 4; ++j) oacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -3.0e38f, lrun = 0.f;

    // fragments of one 64-key tile: K [nt][s2] (A operand: row = key k0 + 16 nt + c, k = d = 32 s2 + 8 g ..+7),
    //                               V^T [jd][s2] (A operand: row = d = 16 jd + c, k = position 8 g ..+7 of 32-key block s2)
    at_h8 kh[4][2], kl[4][2], vh[4][2], vl[4][2];
    auto load_k = [&](int k0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const long long o = (long long)(k0 + nt * 16 + c) * 64 + 32 * s2 + 8 * g;
                kh[nt][s2] = *reinterpret_cast<const at_h8*>(kp + o);
                kl[nt][s2] = *reinterpret_cast<const at_h8*>(kp + plane + o);
            }
    };
    auto load_v = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const long long o = (long long)(j * 16 + c) * Npad + k0 + 32 * s2 + 8 * g;
                vh[j][s2] = *reinterpret_cast<const at_h8*>(vp + o);
                vl[j][s2] = *reinterpret_cast<const at_h8*>(vp + plane + o);
            }
    };
    if (kbeg < kend) load_k(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += AT_K) {
        load_v(k0);
        f32x4 sacc[4];          // S^T tiles: key = k0 + 16 nt + 4 g + r, column = query c
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            sacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[nt][s2], ql[s2], sacc[nt], 0, 0, 0);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[nt][s2], qh[s2], sacc[nt], 0, 0, 0);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[nt][s2], qh[s2], sacc[nt], 0, 0, 0);
            }
        }
        if (k0 + AT_K < kend) load_k(k0 + AT_K);
        float m = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + nt * 16 + 4 * g + r;
                const float sv = (key < N) ? sacc[nt][r] * scale : -3.0e38f;
                sacc[nt][r] = sv;
                m = fmaxf(m, sv);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mnew = fmaxf(mrun, m);
        const float corr = expf(mrun - mnew);
        mrun = mnew;
        float ls = 0.f;
        at_h8 ph[2], pl[2];     // P^T fragments (B operand) of the two 32-key blocks
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + nt * 16 + 4 * g + r;
                const float pv = (key < N) ? expf(sacc[nt][r] - mnew) : 0.f;
                ls += pv;
                _Float16 x, y;
                at_split(pv, x, y);
                ph[nt >> 1][4 * (nt & 1) + r] = x;
                pl[nt >> 1][4 * (nt & 1) + r] = y;
            }
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        lrun = lrun * corr + ls;
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[j] *= corr;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[j][s2], pl[s2], oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[j][s2], ph[s2], oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[j][s2], ph[s2], oacc[j], 0, 0, 0);
            }
    }
    // out[t][head*64 + d] = O / l      (x = (attn @ v).transpose(1,2).reshape(B,N,C), :278): 4 consecutive d per lane
    const int qi = q0 + c;
    if (qi < N) {
        const float inv = 1.0f / lrun;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(out + ((long long)view * N + qi) * dim + head * 64 + j * 16 + 4 * g) =
                make_float4(oacc[j][0] * inv, oacc[j][1] * inv, oacc[j][2] * inv, oacc[j][3] * inv);
    }
}

// ---- attention, third form (round 6): attn2's arithmetic, K / V^T tiles shared by four waves through LDS ---------------------------
// attn2_kernel (one wave = 16 queries per workgroup) streams the whole K / V^T planes of its (view, head) from L2 per wave: at
// N = 1500 that is 650 MB per layer and the launch takes 172 us (3.8 TB/s: the L2's rate for this pattern); the LDS-staged
// attn_kernel converts and transposes every tile inside every workgroup (134 us).  Here one workgroup = 4 waves = 64 queries; a
// 64-key tile of the PRE-SPLIT planes (the qkv GEMM's epilogue writes them in fragment order) is copied into LDS by LDS-DMA, one
// plane per wave (K hi, K lo, V^T hi, V^T lo: 8 KB = 8 wave instructions each), double-buffered, ONE barrier per tile; every wave
// reads its fragments with ds_read_b128 and runs attn2's transposed products / online softmax in registers.  A quarter of the L2
// traffic, no conversion, no 2-byte LDS stores.  LDS rows are 128 B (8 slots of 16 B); slot s of row r sits at s ^ (r & 7): the
// 16 lanes a ds_read_b128 services per cycle (16 rows, two slot columns) then hit 16 different positions of the bank row
// (LDS-DMA writes lane-linear, so the swizzle is applied to the SOURCE address a lane fetches).
#define AT3_PLANE (64 * 128)           // bytes of one tile plane
__global__ __launch_bounds__(256) void attn3_kernel(const float* __restrict__ qkv, const _Float16* __restrict__ Kp,
                                                    const _Float16* __restrict__ Vp, int N, int Npad, int dim,
                                                    float scale, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char tiles[2][4 * AT3_PLANE];      // [buffer][K hi | K lo | V^T hi | V^T lo]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y, view = blockIdx.z, heads = gridDim.y;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const int ld = 3 * dim;
    const float* base = qkv + (long long)view * N * ld;
    const long long plane = (long long)Npad * 64;
    const _Float16* kp = Kp + ((long long)view * heads + head) * 2 * plane;
    const _Float16* vp = Vp + ((long long)view * heads + head) * 2 * plane;

    // this wave's share of a tile copy: plane `wave` (0: K hi, 1: K lo, 2: V^T hi, 3: V^T lo), 8 x 1 KiB
    typedef __attribute__((address_space(1))) const void* at_gptr;
    typedef __attribute__((address_space(3))) void* at_lptr;
    auto stage = [&](int k0, int buf) {
        char* dst = tiles[buf] + wave * AT3_PLANE;
        const _Float16* src = (wave < 2 ? kp : vp) + (long long)(wave & 1) * plane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int L = i * 64 + lane, r = L >> 3, sl = (L & 7) ^ (r & 7);          // LDS slot L holds logical slot sl of row r
            const _Float16* gsrc = wave < 2 ? src + (long long)(k0 + r) * 64 + 8 * sl               // K row = key k0 + r
                                            : src + (long long)r * Npad + k0 + 8 * sl;              // V^T row = d
            __builtin_amdgcn_global_load_lds((at_gptr)gsrc, (at_lptr)(dst + i * 1024), 16, 0, 0);
        }
    };

    // Q^T fragments (B operand: column = query c, k = d = 32 s2 + 8 g + j)
    at_h8 qh[2], ql[2];
    {
        const int qi = q0 + c;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = 0.f;
            if (qi < N) {
                const float* src = base + (long long)qi * ld + head * 64 + 32 * s2 + 8 * g;
                float4 a = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
                v8[0] = a.x; v8[1] = a.y; v8[2] = a.z; v8[3] = a.w; v8[4] = b4.x; v8[5] = b4.y; v8[6] = b4.z; v8[7] = b4.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 x, y;
                at_split(v8[e], x, y);
                qh[s2][e] = x; ql[s2][e] = y;
            }
        }
    }
    f32x4 oacc[4];              // O^T tiles: d = 16 jd + 4 g + r, column = query c
#pragma unroll
    for (int j = 0; j < 4; ++j) oacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -3.0e38f, lrun = 0.f;
    const int foff = c * 128;                         // this lane's row inside a 16-row block of a tile plane
    const int sw = c & 7;

    stage(0, 0);
    int buf = 0;
    for (int k0 = 0; k0 < N; k0 += AT_K, buf ^= 1) {
        __builtin_amdgcn_s_waitcnt(0x0070);           // vmcnt(0): this wave's plane of the tile has landed
        __syncthreads();                              // ... and everybody's; everybody is done with the other buffer
        if (k0 + AT_K < N) stage(k0 + AT_K, buf ^ 1);
        const char* T = tiles[buf];
        f32x4 sacc[4];          // S^T tiles: key = k0 + 16 nt + 4 g + r, column = query c
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            sacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int o = nt * 16 * 128 + foff + (((4 * s2 + g) ^ sw) << 4);
                const at_h8 kh = *reinterpret_cast<const at_h8*>(T + o);
                const at_h8 kl = *reinterpret_cast<const at_h8*>(T + AT3_PLANE + o);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[s2], sacc[nt], 0, 0, 0);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[s2], sacc[nt], 0, 0, 0);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[s2], sacc[nt], 0, 0, 0);
            }
        }
        float m = -3.0e38f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + nt * 16 + 4 * g + r;
                const float sv = (key < N) ? sacc[nt][r] * scale : -3.0e38f;
                sacc[nt][r] = sv;
                m = fmaxf(m, sv);
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const float mnew = fmaxf(mrun, m);
        const float corr = expf(mrun - mnew);
        mrun = mnew;
        float ls = 0.f;
        at_h8 ph[2], pl[2];     // P^T fragments (B operand) of the two 32-key blocks (attn2_kernel's key order)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + nt * 16 + 4 * g + r;
                const float pv = (key < N) ? expf(sacc[nt][r] - mnew) : 0.f;
                ls += pv;
                _Float16 x, y;
                at_split(pv, x, y);
                ph[nt >> 1][4 * (nt & 1) + r] = x;
                pl[nt >> 1][4 * (nt & 1) + r] = y;
            }
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        lrun = lrun * corr + ls;
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[j] *= corr;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = j * 16 * 128 + foff + (((4 * s2 + g) ^ sw) << 4);
                const at_h8 vh = *reinterpret_cast<const at_h8*>(T + 2 * AT3_PLANE + o);
                const at_h8 vl = *reinterpret_cast<const at_h8*>(T + 3 * AT3_PLANE + o);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[s2], oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[s2], oacc[j], 0, 0, 0);
                oacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[s2], oacc[j], 0, 0, 0);
            }
    }
    const int qi = q0 + c;
    if (qi < N) {
        const float inv = 1.0f / lrun;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(out + ((long long)view * N + qi) * dim + head * 64 + j * 16 + 4 * g) =
                make_float4(oacc[j][0] * inv, oacc[j][1] * inv, oacc[j][2] * inv, oacc[j][3] * inv);
    }
}

static int vit_npad(int N) { return (N + 63) / 64 * 64; }

// zero fill of the K / V^T planes as a KERNEL, not hipMemsetAsync: this forward is also replayed as a hipGraph (hip.vit_forward(
// graph=True)), and with the clear as a memset node the replayed graph produced non-finite tokens in a third of the runs of
// tests/test_gpu_parity.py::test_render_sequence_equals_per_frame_render on ROCm 7.2 (0 of 15 with
// DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, 0 of 40 with this kernel; the stem's graph -- kernel nodes only -- never failed; a stand-alone
// memset -> kernel graph does not fail either: tools/ubench/graph_memset_node.py).  profiles/r05_l_vit_graph_memset_node.txt
__global__ __launch_bounds__(256) void zero16_kernel(uint4* __restrict__ p, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256)
        p[i] = make_uint4(0u, 0u, 0u, 0u);
}

size_t th_vit_ws(int V, int N, int dim, int heads) {
    size_t T = (size_t)V * N;
    // X, Y, qkv / hidden, and the split K / V^T planes of one layer (2 x [V][heads][2][Npad][64] halves)
    return 2 * th_align(T * dim * 4) + th_align(T * 4 * dim * 4) +
           2 * th_align((size_t)V * heads * 2 * vit_npad(N) * 64 * sizeof(_Float16));
}

int th_vit_launch(const ThVitPacked& W, const float* x, const float* pe, int V, int N, float* out, void* ws,
                  size_t ws_bytes, hipStream_t s, unsigned int* range, bool allow_h3) {
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
    const int Npad = vit_npad(N);
    _Float16* Kp = ar.take<_Float16>((size_t)V * heads * 2 * Npad * 64);
    _Float16* Vp = ar.take<_Float16>((size_t)V * heads * 2 * Npad * 64);
    TH_REQUIRE(Vp != nullptr, "workspace too small");
    // few tokens: the register-fed form (more, shorter workgroups); many: the LDS-staged form (64 queries share every K / V
    // tile: a quarter of the L2 traffic, which is what bounds the register-fed form from N ~ 1000 on).
    // Measured ViT forward, N_c = 300 / 500 / 1500: 0.80 / 0.99 / 2.15 ms register-fed, 0.86 / 1.07 / 2.06 ms LDS-staged.
    static const char* attn_env = getenv("TH_ATTN_FORM");               // "lds" | "reg": A/B switch
    // (transposed register-fed form, late round 2: 0.77 / 0.95 / 1.19 / 2.06 ms at N_c = 500 / 800 / 1000 / 1500 against 0.93 / 1.16 /
    // 1.27 / 1.76 ms LDS-staged)
    // (round 6: from N ~ 700 on the third form -- attn2's arithmetic on K / V^T tiles shared through LDS, attn3_kernel; "lds": the
    // round-2 form that converts inside the workgroup, "reg" / "tile": force the register-fed / the tiled form)
    const bool attn_lds = attn_env ? attn_env[0] == 'l' : false;
    const bool attn_tile = attn_env ? attn_env[0] == 't' : N > 700;
    TH_REQUIRE(Q != nullptr, "workspace carve failed");
    long long n = (long long)T * dim;
    hipLaunchKernelGGL(add_kernel, dim3(th_cdiv(n, 256)), dim3(256), 0, s, x, pe, n, X);
    const float scale = 0.125f;   // head_dim ** -0.5
    // the two pre-LayerNorms of a block run inside the GEMM that consumes them (TH_VIT_SEPARATE_LN=1: own launches)
    static const bool separate_ln = getenv("TH_VIT_SEPARATE_LN") != nullptr;
    const bool fuse_ln = !separate_ln && th_gemm_ln_ok(T, W.blocks[0].qkv) && th_gemm_ln_ok(T, W.blocks[0].fc1);
    // dense layers on the fp16-split MFMA path (th_gemm_h3; TH_VIT_GEMM_F32=1: the fp32 MFMA GEMMs)
    static const bool f32_gemm = getenv("TH_VIT_GEMM_F32") != nullptr;
    const bool h3 = allow_h3 && !f32_gemm && !separate_ln && th_gemm_h3_ok(T, W.blocks[0].qkv, true) && th_gemm_h3_ok(T, W.blocks[0].fc1, true) &&
                    th_gemm_h3_ok(T, W.blocks[0].proj, false) && th_gemm_h3_ok(T, W.blocks[0].fc2, false);
    // the K / V^T operand planes of the register-fed attention are written by the qkv GEMM's epilogue (no kv_split launch);
    // the padding keys (N .. Npad) of the planes must read as zero: cleared once per forward
    static const bool no_fuse_split = getenv("TH_VIT_KV_SPLIT") != nullptr;       // A/B switch: the separate launch
    const bool fuse_split = h3 && !attn_lds && !no_fuse_split;
    const ThQkvSplit qs{Kp, Vp, N, Npad, heads, dim};
    if (fuse_split && Npad != N)
    {
        const long long n16 = (long long)((((char*)Vp - (char*)Kp) + (size_t)V * heads * 2 * Npad * 64 * sizeof(_Float16)) / 16);   // Kp .. end of Vp
        hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)(th_cdiv(n16, 256) < 1024 ? th_cdiv(n16, 256) : 1024)), dim3(256), 0, s,
                           reinterpret_cast<uint4*>(Kp), n16);
    }
    for (int b = 0; b < W.depth; ++b) {
        const ThVitBlockPacked& B = W.blocks[b];
        if (h3) {
            TH_TRY(th_gemm_h3(X, dim, T, B.qkv, B.ln1_w, B.ln1_b, 1e-6f, TH_ACT_NONE, Q, 3 * dim, range, s, fuse_split ? &qs : nullptr));
        } else if (fuse_ln) {
            TH_TRY(th_gemm_ln(X, dim, T, B.qkv, B.ln1_w, B.ln1_b, 1e-6f, TH_ACT_NONE, Q, 3 * dim, s));
        } else {
            hipLaunchKernelGGL(layernorm_kernel, dim3(th_cdiv(T, 4)), dim3(256), 0, s, X, T, dim, B.ln1_w, B.ln1_b, 1e-6f, Y);
            TH_TRY(th_gemm(Y, dim, T, B.qkv, TH_ACT_NONE, Q, 3 * dim, s));
        }
        if (attn_lds) {
            hipLaunchKernelGGL(attn_kernel, dim3(th_cdiv(N, AT_Q), heads, V), dim3(256), 0, s, Q, N, dim, scale, Y);
        } else {
            if (!(h3 && fuse_split))
                hipLaunchKernelGGL(kv_split_kernel, dim3(Npad / 64, heads, V), dim3(256), 0, s, Q, N, Npad, dim, Kp, Vp);
            if (attn_tile) hipLaunchKernelGGL(attn3_kernel, dim3(th_cdiv(N, 64), heads, V), dim3(256), 0, s, Q, Kp, Vp, N, Npad, dim, scale, Y);
            else hipLaunchKernelGGL(attn2_kernel, dim3(th_cdiv(N, 16), heads, V), dim3(64), 0, s, Q, Kp, Vp, N, Npad, dim, scale, Y);
        }
        if (h3) {
            TH_TRY(th_gemm_h3(Y, dim, T, B.proj, nullptr, nullptr, 0.f, TH_ACT_NONE | TH_GEMM_ACCUM, X, dim, range, s));
            TH_TRY(th_gemm_h3(X, dim, T, B.fc1, B.ln2_w, B.ln2_b, 1e-6f, TH_ACT_GELU, Q, 4 * dim, range, s));
            TH_TRY(th_gemm_h3(Q, 4 * dim, T, B.fc2, nullptr, nullptr, 0.f, TH_ACT_NONE | TH_GEMM_ACCUM, X, dim, range, s));
            continue;
        }
        TH_TRY(th_gemm(Y, dim, T, B.proj, TH_ACT_NONE | TH_GEMM_ACCUM, X, dim, s));
        if (fuse_ln) {
            TH_TRY(th_gemm_ln(X, dim, T, B.fc1, B.ln2_w, B.ln2_b, 1e-6f, TH_ACT_GELU, Q, 4 * dim, s));
        } else {
            hipLaunchKernelGGL(layernorm_kernel, dim3(th_cdiv(T, 4)), dim3(256), 0, s, X, T, dim, B.ln2_w, B.ln2_b, 1e-6f, Y);
            TH_TRY(th_gemm(Y, dim, T, B.fc1, TH_ACT_GELU, Q, 4 * dim, s));
        }
        TH_TRY(th_gemm(Q, 4 * dim, T, B.fc2, TH_ACT_NONE | TH_GEMM_ACCUM, X, dim, s));
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3(th_cdiv(T, 4)), dim3(256), 0, s, X, T, dim, W.norm_w, W.norm_b, 1e-6f, out);
    TH_LAUNCH_CHECK();
    return 0;
}
