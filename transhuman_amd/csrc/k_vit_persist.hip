// K3, one launch (opt-in: th_set_vit_mode(ctx, 2) / TH_VIT_PERSIST=1): TransHE (VisionTransformer.forward,
// lib/networks/vision_transformer.py:371-383, blocks :257-307) as a PERSISTENT kernel.
//
// The per-layer form (k_vit.hip: 63 launches, 16-row x 64-column GEMM workgroups, one wave per 16 queries) is shaped for
// latency on an idle device: every launch spreads ~1000 short workgroups over all 256 CUs.  Under the frame pipeline the
// device is not idle -- a fused-MLP workgroup owns a whole CU (150 KB of LDS, 4 x 512 registers) for 55 us, a TransHE
// workgroup can only start on a CU such a tile has just released and keeps the next tile off it while it waits for its
// loads: the 0.09 ms of TransHE arithmetic cost the frame 0.3 - 0.6 ms (profiles/r03_e_stream_experiments.txt).  What
// counts there is CU-TIME.  Here V * ceil(N_c / 32) workgroups (48 at V = 3, N_c = 500) each own 32 tokens of one view
// for the whole forward:
//   * everything but attention is row-local (LayerNorm, qkv / proj / fc1 / fc2, GELU, residuals): the residual rows and the
//     block's biases live in LDS, the rows' operand planes too ([32][K] fp16 hi | lo, the layout of gemm_h3_kernel); the four
//     waves walk the column tiles of a layer with the weight fragments streaming through a 12-deep register ring ACROSS tiles.
//     The stream is fully unrolled (compile-time tile and step counts) with a scheduling barrier per step: with run-time
//     bounds around the ring's loads the wait-count pass emitted vmcnt(0) at every step (one miss latency per 6 MFMAs),
//     without the barriers the scheduler hoisted the stream's loads and spilled 100+ registers; the ring is addressed as
//     scalar base + one lane offset.  For K = 192 the 24 A fragments of the rows are read from LDS once per layer and kept in
//     registers (every column tile uses the same ones), for K = 768 the next step's fragments are requested before a step's
//     MFMAs.  Epilogues write straight into the next stage's planes (attention output -> proj operand, GELU(fc1) -> fc2
//     operand) or into the residual rows;
//   * attention needs the keys / values of the whole view: the qkv epilogue publishes this workgroup's K / V^T in
//     attn2_kernel's operand-plane layout, ONE device-wide barrier per block (release / acquire at agent scope through the
//     compiler's memory model: the eight XCDs' L2s are not coherent with each other), then one wave per head runs
//     attn2_kernel's register-fed flash loop for BOTH 16-query tiles of the rows against every K / V^T fragment (tiles
//     double-buffered); the planes are double-buffered so a block needs no second barrier.
// Arithmetic: gemm_h3_kernel's three-term fp16-split products in the same k order, the same LayerNorm, attn2's softmax
// recurrence -- with v_exp_f32 (x log2 e) instead of expf and a 1.5e-7 polynomial erf in GELU (the softmax of 32 queries runs
// on three waves here: with the library functions the stage was VALU-bound), and the final LayerNorm reduced by 16 lanes per
// row: within 2e-6 of the per-layer path (tests/test_gpu_round3.py), deterministic.  2 launches per forward (one memset of the
// key padding + barrier word, this kernel) instead of 63.
// Measured (N_c = 500, V = 3): 1.00 - 1.03 ms stand-alone on 48 CUs against 0.70 ms for the per-layer path on the whole chip;
// cycles of one block on workgroup 0 (-DVP_DBG): ln1 7.5 k, qkv 30 k, barrier 17-20 k, attention 80 k (40 k without its K / V
// loads), proj 7 k, ln2 2.6 k, fc1 32 k, fc2 17 k = 196 k.  In the frame pipeline it is neutral (21.0 - 21.4 ms either way):
// 48 CUs x 1.2 ms is about the CU-time the 63 launches take.  Hence opt-in; what is left is the attention stage's loads and
// the barrier.
#include <stdlib.h>

#include "th_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 vp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 vp_h4 __attribute__((ext_vector_type(4)));

#define VP_ROWS 32
#define VP_DIM 192
#define VP_HID 768
#define VP_SA (2 * VP_DIM + 16)      // bytes per LDS row of a K = 192 plane
#define VP_SH (2 * VP_HID + 16)      // ... of the K = 768 plane (fc2's operand)
#define VP_MAX_DEPTH 12

struct VpLayer {
    const uint4* w16;
    const float* inv;
    const float* bias;
    int N, NB;
};
struct VpBlock {
    VpLayer qkv, proj, fc1, fc2;
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
};
struct VpParams {
    VpBlock blk[VP_MAX_DEPTH];
    int depth;
    const float* x;
    const float* pe;
    float* X;              // [V N][192] residual stream
    float* Qb;             // [V N][576] queries (fp32, columns 0..191 used)
    float* out;
    _Float16* Kp[2];
    _Float16* Vp[2];
    int V, N, Npad, heads;
    const float *norm_w, *norm_b;
    unsigned* bar;
    unsigned* range;
    float scale;
};

__device__ __forceinline__ void vp_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
// exp on the transcendental unit (v_exp_f32 of x log2 e: ~1 ulp; the library expf is ~40 VALU instructions, and the softmax of
// a workgroup's 32 queries x N_c keys x 3 heads runs on three waves here -- with expf the attention stage was VALU-bound, 88 of
// a block's 218 k cycles)
__device__ __forceinline__ float vp_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
// nn.GELU (exact form): x/2 (1 + erf(x / sqrt 2)); erf by Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7) instead of the library
// erff (~60 instructions per value, 8 values per lane and column tile in fc1's epilogue)
__device__ __forceinline__ float vp_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = 1.0f - poly * vp_exp(-z * z);                 // erf(|x| / sqrt 2)
    return x * 0.5f * (1.0f + copysignf(e, x));
}

// device-wide barrier: every workgroup of the launch is resident (48 .. 141 workgroups of one per CU); counter only grows
__device__ __forceinline__ void vp_grid_sync(unsigned* bar, unsigned target) {
    __threadfence();                    // release (agent scope): this thread's K / V stores are written back beyond the XCD's L2
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    __threadfence();                    // acquire side: later loads of every thread see the other workgroups' planes
}

// rows [nrows <= 32] x 192 of src (row stride 192) -> LayerNorm (eps 1e-6, two-pass statistics like layernorm_kernel) ->
// either the K = 192 operand planes (OUT == nullptr) or fp32 rows of OUT
__device__ __forceinline__ void vp_stage_ln(const float* __restrict__ src, int nrows, const float* __restrict__ ln_w,
                                            const float* __restrict__ ln_b, char* a_hi, char* a_lo, float* __restrict__ OUT,
                                            unsigned& rmax) {
    const int tid = threadIdx.x, sub = tid & 15;
    float4 w4[3], b4[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int k = 4 * (sub + 16 * q);
        w4[q] = *reinterpret_cast<const float4*>(ln_w + k);
        b4[q] = *reinterpret_cast<const float4*>(ln_b + k);
    }
    float v[2][12];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 16 * t + (tid >> 4);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = 4 * (sub + 16 * q);
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nrows) x = *reinterpret_cast<const float4*>(src + (long long)row * VP_DIM + k);
            v[t][4 * q] = x.x; v[t][4 * q + 1] = x.y; v[t][4 * q + 2] = x.z; v[t][4 * q + 3] = x.w;
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = 16 * t + (tid >> 4);
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) sum += (v[t][4 * q] + v[t][4 * q + 1]) + (v[t][4 * q + 2] + v[t][4 * q + 3]);
        sum += 0.f;                                            // (the per-layer kernel adds a fourth, all-zero float4)
        sum = th_row16_sum(sum);
        const float mean = sum / (float)VP_DIM;
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[t][4 * q + e] - mean; ss += d * d; }
        ss = th_row16_sum(ss);
        const float rs = 1.0f / __fsqrt_rn(ss / (float)VP_DIM + 1e-6f);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int k = 4 * (sub + 16 * q);
            const float wv[4] = {w4[q].x, w4[q].y, w4[q].z, w4[q].w}, bb[4] = {b4[q].x, b4[q].y, b4[q].z, b4[q].w};
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = row < nrows ? (v[t][4 * q + e] - mean) * rs * wv[e] + bb[e] : 0.f;
            if (OUT != nullptr) {
                if (row < nrows) *reinterpret_cast<float4*>(OUT + (long long)row * VP_DIM + k) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
                vp_h4 hv, lv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 hi, lo;
                    vp_split(y[e], hi, lo);
                    hv[e] = hi; lv[e] = lo;
                    rmax = max(rmax, (unsigned)(__builtin_bit_cast(unsigned short, hi) & 0x7fffu));
                }
                *reinterpret_cast<vp_h4*>(a_hi + row * VP_SA + 2 * k) = hv;
                *reinterpret_cast<vp_h4*>(a_lo + row * VP_SA + 2 * k) = lv;
            }
        }
    }
}

// C[32 rows][N] = A planes (LDS) x W^T on v_mfma_f32_16x16x32_f16, three product terms with their own accumulators
// (hi*lo, lo*hi, hi*hi: gemm_h3_kernel's scheme and order).  Wave w takes the 16-column tiles w, w + 4, ...; the weight
// fragments of ALL its tiles form one stream through a VP_RING-deep register ring.
//   MODE 0: qkv   -- queries -> Qb (fp32), keys / values -> the operand planes of attn2_kernel (ThQkvSplit's layout)
//   MODE 1: C += (proj, fc2: residual rows X in global memory, own rows)
//   MODE 2: GELU -> the K = 768 planes (fc1 -> fc2 operand)
template <int KB32, int MODE, int NTILE, int VP_RING>
__device__ __forceinline__ void vp_gemm(const char* a_hi, const char* a_lo, const int stride, const VpLayer& L, const VpParams& P,
                                        const int view, const int q0, const int nrows, const int buf, char* h_hi, char* h_lo,
                                        float* xs, const float* bsl, unsigned& rmax) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4, c = lane & 15;
    constexpr int S = NTILE * KB32;       // compile-time: the stream below is straight-line code
    const long long m0 = (long long)view * P.N + q0;
    const float inv = L.inv[0];
    uint4 ring[VP_RING][2];
    // scalar base (wave-uniform) + one 32-bit lane offset: the saddr form of global_load, no per-step address registers
    const unsigned loff = (unsigned)lane * 16u;
    auto wptr = [&](int step) {
        const int ti = step / KB32, kb = step - ti * KB32;
        const uint4* sb = L.w16 + ((long long)(wave + 4 * ti) * KB32 + kb) * 128;
        return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sb) + loff);
    };
#pragma unroll
    for (int j = 0; j < VP_RING; ++j)
        if (j < S) { const uint4* p = wptr(j); ring[j][0] = p[0]; ring[j][1] = p[64]; }
    f32x4 acc[2][3];
    float bv = 0.f, cres[2][4];
    const int aoff = c * stride + 16 * g;
    // A fragments: K = 192 -- the same 24 fragments serve every column tile of the layer: read from LDS ONCE, kept in registers
    // (no ds_read in the stream); K = 768 -- the fragments of step s + 1 are requested before the MFMAs of step s
    constexpr bool AREG = KB32 <= 6;
    vp_h8 afr[AREG ? 2 : 1][AREG ? KB32 : 1][2];
    vp_h8 acur[2][2], anxt[2][2];
    if (AREG) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kb = 0; kb < KB32; ++kb) {
                afr[AREG ? t : 0][AREG ? kb : 0][0] = *reinterpret_cast<const vp_h8*>(a_hi + 16 * t * stride + aoff + kb * 64);
                afr[AREG ? t : 0][AREG ? kb : 0][1] = *reinterpret_cast<const vp_h8*>(a_lo + 16 * t * stride + aoff + kb * 64);
            }
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            acur[t][0] = *reinterpret_cast<const vp_h8*>(a_hi + 16 * t * stride + aoff);
            acur[t][1] = *reinterpret_cast<const vp_h8*>(a_lo + 16 * t * stride + aoff);
        }
    }
#pragma unroll
    for (int base = 0; base < S; base += VP_RING) {
#pragma unroll
        for (int j = 0; j < VP_RING; ++j) {
            const int step = base + j;
            if (step < S) {
                const int ti = step / KB32, kb = step % KB32;
                const int nb = wave + 4 * ti, col = nb * 16 + c;
                if (kb == 0) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int e = 0; e < 3; ++e) acc[t][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    // epilogue operands of this tile: out of LDS (bias of the layer, residual rows) -- a global load here would
                    // make the epilogue's wait drain the weight ring
                    bv = bsl[col];
                    if (MODE == 1) {
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) cres[t][r] = xs[(16 * t + 4 * g + r) * VP_DIM + col];
                    }
                }
                const vp_h8 bh = *reinterpret_cast<const vp_h8*>(&ring[j][0]);
                const vp_h8 bl = *reinterpret_cast<const vp_h8*>(&ring[j][1]);
                if (!AREG) {
                    const int kn = (kb + 1) % KB32;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        anxt[t][0] = *reinterpret_cast<const vp_h8*>(a_hi + 16 * t * stride + aoff + kn * 64);
                        anxt[t][1] = *reinterpret_cast<const vp_h8*>(a_lo + 16 * t * stride + aoff + kn * 64);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const vp_h8 ah = AREG ? afr[AREG ? t : 0][AREG ? kb : 0][0] : acur[t][0];
                    const vp_h8 al = AREG ? afr[AREG ? t : 0][AREG ? kb : 0][1] : acur[t][1];
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t][2], 0, 0, 0);
                }
                if (!AREG) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) { acur[t][0] = anxt[t][0]; acur[t][1] = anxt[t][1]; }
                }
                if (step + VP_RING < S) { const uint4* p = wptr(step + VP_RING); ring[j][0] = p[0]; ring[j][1] = p[64]; }
                __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from hoisting the whole stream's loads: spills)
                if (kb == KB32 - 1 && col < L.N) {
                    // (queries, keys or values: a 16-column tile never straddles two of them, dim = 12 tiles)
                    const int part = MODE == 0 ? col / VP_DIM : 0;
                    const int hd = MODE == 0 ? col - part * VP_DIM : 0;              // head * 64 + d
                    const long long plane = (long long)P.Npad * 64;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4 a = (acc[t][0] + acc[t][1]) + acc[t][2];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int lr = 16 * t + 4 * g + r;
                            float val = a[r] * inv + bv;
                            if (MODE == 2) {
                                val = vp_gelu(val);
                                _Float16 hi, lo;
                                vp_split(val, hi, lo);
                                *reinterpret_cast<_Float16*>(h_hi + lr * VP_SH + 2 * col) = hi;
                                *reinterpret_cast<_Float16*>(h_lo + lr * VP_SH + 2 * col) = lo;
                                if (lr < nrows) rmax = max(rmax, (unsigned)(__builtin_bit_cast(unsigned short, hi) & 0x7fffu));
                            } else if (lr < nrows) {
                                if (MODE == 1) xs[lr * VP_DIM + col] = cres[t][r] + val;
                                else if (part == 0) P.Qb[(m0 + lr) * (3 * VP_DIM) + col] = val;
                                else {
                                    const int key = q0 + lr;
                                    _Float16 hi, lo;
                                    vp_split(val, hi, lo);
                                    _Float16* pb = (part == 1 ? P.Kp[buf] : P.Vp[buf]) + ((long long)view * P.heads + (hd >> 6)) * 2 * plane;
                                    long long o;
                                    if (part == 1) o = (long long)key * 64 + (hd & 63);
                                    else {      // V^T row d, keys in the fragment order of attn2_kernel (see kv_split_kernel)
                                        const int ko = key & 31;
                                        o = (long long)(hd & 63) * P.Npad + (key & ~31) + 8 * ((ko & 15) >> 2) + 4 * (ko >> 4) + (ko & 3);
                                    }
                                    pb[o] = hi;
                                    pb[plane + o] = lo;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// attn2_kernel's loop for one head and BOTH 16-query tiles of the workgroup's rows, run by one wave: every K / V^T fragment
// is loaded once and used for the two tiles (the per-query arithmetic and its order are attn2_kernel's); the normalised
// output goes into the K = 192 operand planes (columns 64 head ..): proj's A operand
__device__ __forceinline__ void vp_attn_head(const VpParams& P, const int view, const int q0, const int nrows, const int head,
                                             const int buf, char* a_hi, char* a_lo, unsigned& rmax) {
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int N = P.N, Npad = P.Npad;
    const long long plane = (long long)Npad * 64;
    const _Float16* kp = P.Kp[buf] + ((long long)view * P.heads + head) * 2 * plane;
    const _Float16* vp = P.Vp[buf] + ((long long)view * P.heads + head) * 2 * plane;
    vp_h8 qh[2][2], ql[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int lq = 16 * qt + c;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = 0.f;
            if (lq < nrows) {
                const float* src = P.Qb + ((long long)view * N + q0 + lq) * (3 * VP_DIM) + head * 64 + 32 * s2 + 8 * g;
                float4 a = *reinterpret_cast<const float4*>(src), b4 = *reinterpret_cast<const float4*>(src + 4);
                v8[0] = a.x; v8[1] = a.y; v8[2] = a.z; v8[3] = a.w; v8[4] = b4.x; v8[5] = b4.y; v8[6] = b4.z; v8[7] = b4.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 x, y;
                vp_split(v8[e], x, y);
                qh[qt][s2][e] = x; ql[qt][s2][e] = y;
            }
        }
    }
    f32x4 oacc[2][4];
    float mrun[2] = {-3.0e38f, -3.0e38f}, lrun[2] = {0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[qt][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K and V^T fragments of a 64-key tile, double-buffered: tile t + 1 is requested before tile t is multiplied
    vp_h8 kh[2][4][2], kl[2][4][2], vh[2][4][2], vl[2][4][2];
    auto load_kv = [&](int k0, int b) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const long long o = (long long)(k0 + nt * 16 + c) * 64 + 32 * s2 + 8 * g;
                kh[b][nt][s2] = *reinterpret_cast<const vp_h8*>(kp + o);
                kl[b][nt][s2] = *reinterpret_cast<const vp_h8*>(kp + plane + o);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const long long o = (long long)(j * 16 + c) * Npad + k0 + 32 * s2 + 8 * g;
                vh[b][j][s2] = *reinterpret_cast<const vp_h8*>(vp + o);
                vl[b][j][s2] = *reinterpret_cast<const vp_h8*>(vp + plane + o);
            }
    };
    auto tile = [&](int k0, int b) {
        f32x4 sacc[2][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                sacc[qt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    sacc[qt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[b][nt][s2], ql[qt][s2], sacc[qt][nt], 0, 0, 0);
                    sacc[qt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl[b][nt][s2], qh[qt][s2], sacc[qt][nt], 0, 0, 0);
                    sacc[qt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[b][nt][s2], qh[qt][s2], sacc[qt][nt], 0, 0, 0);
                }
            }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float m = -3.0e38f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + nt * 16 + 4 * g + r;
                    const float sv = (key < N) ? sacc[qt][nt][r] * P.scale : -3.0e38f;
                    sacc[qt][nt][r] = sv;
                    m = fmaxf(m, sv);
                }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            const float mnew = fmaxf(mrun[qt], m);
            const float corr = vp_exp(mrun[qt] - mnew);
            mrun[qt] = mnew;
            float ls = 0.f;
            vp_h8 ph[2], pl[2];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + nt * 16 + 4 * g + r;
                    const float pv = (key < N) ? vp_exp(sacc[qt][nt][r] - mnew) : 0.f;
                    ls += pv;
                    _Float16 x, y;
                    vp_split(pv, x, y);
                    ph[nt >> 1][4 * (nt & 1) + r] = x;
                    pl[nt >> 1][4 * (nt & 1) + r] = y;
                }
            ls += __shfl_xor(ls, 16);
            ls += __shfl_xor(ls, 32);
            lrun[qt] = lrun[qt] * corr + ls;
#pragma unroll
            for (int j = 0; j < 4; ++j) oacc[qt][j] *= corr;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    oacc[qt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[b][j][s2], pl[s2], oacc[qt][j], 0, 0, 0);
                    oacc[qt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl[b][j][s2], ph[s2], oacc[qt][j], 0, 0, 0);
                    oacc[qt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh[b][j][s2], ph[s2], oacc[qt][j], 0, 0, 0);
                }
        }
    };
    // (the planes are padded to Npad keys with zeros: a clamped request past the last tile is harmless)
    // tiles in pairs, no branch in the loop (a tile past the end has every key masked: it changes nothing)
    load_kv(0, 0);
#ifdef VP_EXP_NOKV        // timing experiment (wrong results): every tile multiplies the first tile's fragments
    load_kv(0, 1);
    for (int k0 = 0; k0 < N; k0 += 128) { tile(k0, 0); tile(k0 + 64, 1); }
#else
    for (int k0 = 0; k0 < N; k0 += 128) {
        load_kv(min(k0 + 64, Npad - 64), 1);
        tile(k0, 0);
        load_kv(min(k0 + 128, Npad - 64), 0);
        tile(k0 + 64, 1);
    }
#endif
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int lq = 16 * qt + c;
        const float inv = 1.0f / lrun[qt];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vp_h4 hv, lv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float val = lq < nrows ? oacc[qt][j][r] * inv : 0.f;
                _Float16 hi, lo;
                vp_split(val, hi, lo);
                hv[r] = hi; lv[r] = lo;
                rmax = max(rmax, (unsigned)(__builtin_bit_cast(unsigned short, hi) & 0x7fffu));
            }
            *reinterpret_cast<vp_h4*>(a_hi + lq * VP_SA + 2 * (head * 64 + j * 16 + 4 * g)) = hv;
            *reinterpret_cast<vp_h4*>(a_lo + lq * VP_SA + 2 * (head * 64 + j * 16 + 4 * g)) = lv;
        }
    }
}

__global__ __launch_bounds__(256) void vit_persist_kernel(const VpParams P) {
    extern __shared__ __attribute__((aligned(16))) char vp_lds[];
    char* a_hi = vp_lds;
    char* a_lo = vp_lds + VP_ROWS * VP_SA;
    char* h_hi = vp_lds + 2 * VP_ROWS * VP_SA;
    char* h_lo = h_hi + VP_ROWS * VP_SH;
    float* xs = reinterpret_cast<float*>(h_lo + VP_ROWS * VP_SH);      // [32][192] residual stream of the workgroup's rows
    float* bs = xs + VP_ROWS * VP_DIM;                                  // biases of the block's four layers: 576 | 192 | 768 | 192
    const int tid = threadIdx.x, wave = tid >> 6;
    const int G = (P.N + VP_ROWS - 1) / VP_ROWS;
    if ((int)blockIdx.x >= P.V * G) {
        // Weight prefetchers: gridDim.x - V G extra workgroups (8: with round-robin placement one per XCD, speed only) that
        // never arrive at the barrier.  A block's four weight images (1.77 MB) reach a compute workgroup from HBM / MALL at
        // the latency of a 24-deep register ring -- a third of the rate it gets out of its XCD's L2 -- so each prefetcher
        // touches every 128-byte line of block b + 1 while block b computes (paced by the barrier word).
        const unsigned ncomp = (unsigned)(P.V * G);
        unsigned sink = 0u;
        for (int b = 0; b < P.depth; ++b) {
            if (b > 0 && tid == 0)
                while (__hip_atomic_load(P.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(b - 1) * ncomp + 1u) __builtin_amdgcn_s_sleep(8);
            __syncthreads();
            const VpBlock& B = P.blk[b];
            const VpLayer* Ls[4] = {&B.qkv, &B.proj, &B.fc1, &B.fc2};
            const int kb32[4] = {VP_DIM / 32, VP_DIM / 32, VP_DIM / 32, VP_HID / 32};
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const unsigned* w = reinterpret_cast<const unsigned*>(Ls[l]->w16);
                const long long lines = (long long)Ls[l]->NB * kb32[l] * 2 * 64 * 16 / 128;
                // 16 independent requests in flight per thread (a dependent chain of touches would run at one miss latency each)
                for (long long i0 = tid; i0 < lines; i0 += 256 * 16) {
                    unsigned t16[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const long long i = i0 + 256LL * u;
                        t16[u] = w[(i < lines ? i : (long long)tid) * 32];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) sink += t16[u];
                }
            }
        }
        if (sink == 0x9e3779b9u) P.bar[1] = sink;          // (keeps the loads)
        return;
    }
    const int view = blockIdx.x / G, q0 = (blockIdx.x % G) * VP_ROWS;
    const int nrows = min(VP_ROWS, P.N - q0);
    const long long m0 = (long long)view * P.N + q0;
    const float* Xr = xs;
    unsigned rmax = 0u;
    // x += PE (:373-383), own rows (rows beyond nrows: zeros)
    for (int i = tid; i < VP_ROWS * (VP_DIM / 4); i += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < nrows * (VP_DIM / 4)) {
            const float4 a = reinterpret_cast<const float4*>(P.x + m0 * VP_DIM)[i], b = reinterpret_cast<const float4*>(P.pe + m0 * VP_DIM)[i];
            v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        reinterpret_cast<float4*>(xs)[i] = v;
    }
    __syncthreads();
    const unsigned nwg = (unsigned)(P.V * G);
#ifdef VP_DBG
    long long stamp[10];
#define VP_STAMP(i) if (b == 5) stamp[i] = clock64()
#else
#define VP_STAMP(i)
#endif
    for (int b = 0; b < P.depth; ++b) {
        const VpBlock& B = P.blk[b];
        const int buf = b & 1;
        VP_STAMP(0);
        // biases of the block's four layers -> LDS (read by the epilogues; the previous block's readers are behind a barrier)
        for (int i = tid; i < 3 * VP_DIM + VP_DIM + VP_HID + VP_DIM; i += 256) {
            const float* src = i < 3 * VP_DIM ? B.qkv.bias + i : (i < 4 * VP_DIM ? B.proj.bias + (i - 3 * VP_DIM) :
                               (i < 4 * VP_DIM + VP_HID ? B.fc1.bias + (i - 4 * VP_DIM) : B.fc2.bias + (i - 4 * VP_DIM - VP_HID)));
            bs[i] = *src;
        }
        vp_stage_ln(Xr, nrows, B.ln1_w, B.ln1_b, a_hi, a_lo, nullptr, rmax);
        __syncthreads();
        VP_STAMP(1);
        vp_gemm<VP_DIM / 32, 0, 9, 12>(a_hi, a_lo, VP_SA, B.qkv, P, view, q0, nrows, buf, h_hi, h_lo, xs, bs, rmax);
        VP_STAMP(2);
        vp_grid_sync(P.bar, (unsigned)(b + 1) * nwg);
        VP_STAMP(3);
        for (int h = wave; h < P.heads; h += 4) vp_attn_head(P, view, q0, nrows, h, buf, a_hi, a_lo, rmax);
        __syncthreads();
        VP_STAMP(4);
        vp_gemm<VP_DIM / 32, 1, 3, 12>(a_hi, a_lo, VP_SA, B.proj, P, view, q0, nrows, buf, h_hi, h_lo, xs, bs + 3 * VP_DIM, rmax);
        __syncthreads();
        VP_STAMP(5);
        vp_stage_ln(Xr, nrows, B.ln2_w, B.ln2_b, a_hi, a_lo, nullptr, rmax);
        __syncthreads();
        VP_STAMP(6);
        vp_gemm<VP_DIM / 32, 2, 12, 12>(a_hi, a_lo, VP_SA, B.fc1, P, view, q0, nrows, buf, h_hi, h_lo, xs, bs + 4 * VP_DIM, rmax);
        __syncthreads();
        VP_STAMP(7);
        vp_gemm<VP_HID / 32, 1, 3, 12>(h_hi, h_lo, VP_SH, B.fc2, P, view, q0, nrows, buf, h_hi, h_lo, xs, bs + 4 * VP_DIM + VP_HID, rmax);
        __syncthreads();
        VP_STAMP(8);
    }
    vp_stage_ln(Xr, nrows, P.norm_w, P.norm_b, a_hi, a_lo, P.out + m0 * VP_DIM, rmax);
    if (P.range != nullptr && rmax > P.range[TH_RANGE_VIT]) atomicMax(P.range + TH_RANGE_VIT, rmax);
#ifdef VP_DBG
    if (blockIdx.x == 0 && tid == 0)
        printf("[VP_DBG] block 5 cycles: ln1 %lld qkv %lld barrier %lld attn %lld proj %lld ln2 %lld fc1 %lld fc2 %lld | total %lld\n",
               stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4],
               stamp[6] - stamp[5], stamp[7] - stamp[6], stamp[8] - stamp[7], stamp[8] - stamp[0]);
#endif
}

bool th_vit_persist_ok(const ThVitPacked& W, int V, int N) {
    if (!W.ready || W.dim != VP_DIM || W.heads != 3 || W.depth > VP_MAX_DEPTH || N < 1 || N > 1100) return false;
    if ((long long)V * ((N + VP_ROWS - 1) / VP_ROWS) > 200) return false;          // every workgroup must be resident (one per CU)
    for (int b = 0; b < W.depth; ++b) {
        const ThVitBlockPacked& B = W.blocks[b];
        if (!B.qkv.w16 || !B.proj.w16 || !B.fc1.w16 || !B.fc2.w16) return false;
        if (B.qkv.K != VP_DIM || B.qkv.N != 3 * VP_DIM || B.proj.K != VP_DIM || B.proj.N != VP_DIM || B.fc1.K != VP_DIM ||
            B.fc1.N != VP_HID || B.fc2.K != VP_HID || B.fc2.N != VP_DIM)
            return false;
        if (B.qkv.NB != 36 || B.proj.NB != 12 || B.fc1.NB != 48 || B.fc2.NB != 12 || B.qkv.KB32 != 6 || B.fc2.KB32 != 24) return false;
    }
    return true;
}

size_t th_vit_persist_extra_ws(int V, int N, int heads) {
    const int Npad = (N + 63) / 64 * 64;
    return 2 * th_align((size_t)V * heads * 2 * Npad * 64 * sizeof(_Float16)) + th_align(256);
}

// X, Qb: [V N][192] / [V N][768] fp32 rows of the caller's workspace; planes: 4 x th_align([V][heads][2][Npad][64] halves)
// followed by a 256-byte block for the barrier word, all contiguous (cleared here with one memset)
int th_vit_persist_launch(const ThVitPacked& W, const float* x, const float* pe, int V, int N, float* out, float* X, float* Qb,
                          char* planes, hipStream_t s, unsigned int* range) {
    const int Npad = (N + 63) / 64 * 64;
    const size_t pl = th_align((size_t)V * W.heads * 2 * Npad * 64 * sizeof(_Float16));
    VpParams P{};
    for (int b = 0; b < W.depth; ++b) {
        const ThVitBlockPacked& B = W.blocks[b];
        auto L = [](const ThPacked& p) { return VpLayer{p.w16, p.scale16, p.b, p.N, p.NB}; };
        P.blk[b] = VpBlock{L(B.qkv), L(B.proj), L(B.fc1), L(B.fc2), B.ln1_w, B.ln1_b, B.ln2_w, B.ln2_b};
    }
    P.depth = W.depth;
    P.x = x; P.pe = pe; P.X = X; P.Qb = Qb; P.out = out;
    P.Kp[0] = (_Float16*)planes; P.Vp[0] = (_Float16*)(planes + pl);
    P.Kp[1] = (_Float16*)(planes + 2 * pl); P.Vp[1] = (_Float16*)(planes + 3 * pl);
    P.bar = (unsigned*)(planes + 4 * pl);
    P.V = V; P.N = N; P.Npad = Npad; P.heads = W.heads;
    P.norm_w = W.norm_w; P.norm_b = W.norm_b;
    P.range = range;
    P.scale = 0.125f;           // head_dim ** -0.5
    TH_HIP(hipMemsetAsync(planes, 0, 4 * pl + 256, s));      // key padding N .. Npad of the planes reads as zero; barrier word
    const int lds = 2 * VP_ROWS * VP_SA + 2 * VP_ROWS * VP_SH + VP_ROWS * VP_DIM * 4 + (5 * VP_DIM + VP_HID) * 4;
    static bool attr = false;
    if (!attr) {
        TH_HIP(hipFuncSetAttribute((const void*)vit_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    const int G = (N + VP_ROWS - 1) / VP_ROWS;
    static const int npf = getenv("TH_VIT_PREFETCHERS") ? atoi(getenv("TH_VIT_PREFETCHERS")) : 0;
    hipLaunchKernelGGL(vit_persist_kernel, dim3(V * G + (npf > 0 ? npf : 0)), dim3(256), lds, s, P);
    TH_LAUNCH_CHECK();
    return 0;
}
