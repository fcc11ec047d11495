// K4: DPaRF point encoding.
//
// Network.get_human_representation + get_dist_weight
// (cross_transformer.py:151-205): 7 nearest token centres of every valid sample
// (squared L2 ascending, ties -> lower index; the reference calls pytorch3d
// knn_points :170), d = sqrt, w = softmax(-d / 0.5) (:153-154), offsets rotated
// by the cluster-mean blend rotation (:183-188), 10-octave sin-cos PE with a
// single-rounding fma argument (vision_transformer.py:131-132), and for every
// view sum_k w_k * [token_v[k] (192) | PE_k (63)] (:197-200).
//
// Layout: one workgroup = 256 threads = 128 samples.
//  Phase 1 (two lanes per sample): each lane keeps a register top-7 over the even / odd token centres
//   (staged in LDS, N_c*12 B <= 72 KB); the pair's lists are merged by (distance, index) -- same result as
//   one ordered scan; softmax weights and the 7 rotated offsets go to LDS.
//  Phase 2 (wave per sample, 32 samples per wave): lanes 0..47 fetch a token row as one float4 each
//   (768 B coalesced, L2-resident table), lanes 0..62 evaluate one PE channel each (7 sines, dp_sin).
// Output rows [sample][view][256] (255 + zero pad) feed fc_0.
// FOLDED = true (fused MLP path): `tokens` is the per-frame table T' = tokens fc_0[:, :192]^T, 256 wide; since
// fc_0 is linear, fc_0(h)'s token part is the same neighbour blend applied to T' rows.  Then out = that blend,
// fp32 [sample][view][256] (64 lanes x float4: every lane busy), and pe_out = the blended 63-wide positional
// encoding, ONE split-f16 row (64 hi + 64 lo halves) per sample -- it is the same for every view.
// Bound: L2 gather of 7*V*768 B per sample; HBM write 3 KB per sample.
#include "th_internal.h"

#define DP_K 7
#define DP_SAMPLES 128
#define DP_THREADS 256

struct DpNbr {
    float w[DP_K];
    int idx[DP_K];
    float def[DP_K][3];
};

// insert (cd, ci) into the ascending list; on equal distance the smaller index stays first
__device__ __forceinline__ void dp_insert(float (&bd)[DP_K], int (&bi)[DP_K], float cd, int ci) {
#pragma unroll
    for (int k = 0; k < DP_K; ++k) {
        bool sw = (cd < bd[k]) || (cd == bd[k] && ci < bi[k]);
        float td = sw ? bd[k] : cd;
        int ti = sw ? bi[k] : ci;
        bd[k] = sw ? cd : bd[k];
        bi[k] = sw ? ci : bi[k];
        cd = td; ci = ti;
    }
}

__device__ __forceinline__ void dp_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
typedef _Float16 dp_h4 __attribute__((ext_vector_type(4)));

// sin(a) for |a| < 2^12 (the PE arguments are fma(x, pi*2^k, phase), k <= 9, |x| of the order of the body size).
// Three-constant Cody-Waite reduction by 2*pi (every fma below is exact or rounds once at the 2^-22 level), then
// the hardware sine of the reduced angle.  Absolute error measured against the fp64 sine of the same fp32
// argument: < 1e-6 (tests/test_gpu_parity.py::test_dparf_vs_golden holds the channel tolerance 1e-4); about a
// fifth of the instructions of the full-range library sinf, which made the PE channels the largest VALU block
// of this kernel.
__device__ __forceinline__ float dp_sin(float a) {
    const float k = rintf(a * 0.15915494309189535f);
    float r = fmaf(-k, 6.28125f, a);                       // 2*pi = 6.28125 + 1.93500518798828125e-3 + 3.0199159819e-7
    r = fmaf(-k, 1.93500518798828125e-3f, r);
    r = fmaf(-k, 3.0199159819e-7f, r);
    return __builtin_amdgcn_sinf(r * 0.15915494309189535f);   // v_sin_f32 takes revolutions
}

template <bool FOLDED>
__global__ __launch_bounds__(DP_THREADS) void dparf_kernel(const float* __restrict__ pts_smpl, ThPointSrc ps,
                                                           const float* __restrict__ Rh, const float* __restrict__ Th,
                                                           const int32_t* __restrict__ sel, int P,
                                                           const float* __restrict__ centres,
                                                           const float* __restrict__ rot,
                                                           const float* __restrict__ tokens, int V, int nc,
                                                           float alpha, float* __restrict__ out,
                                                           float* __restrict__ pe_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cen = lds;                                   // [nc*3]
    DpNbr* nb = reinterpret_cast<DpNbr*>(lds + ((nc * 3 + 3) & ~3));
    for (int i = threadIdx.x; i < nc * 3; i += DP_THREADS) cen[i] = centres[i];
    __syncthreads();

    // ---- phase 1: lanes (2s, 2s+1) share sample s ----
    const int ls = threadIdx.x >> 1, half = threadIdx.x & 1;
    const int p = blockIdx.x * DP_SAMPLES + ls;
    if (p < P) {                                        // (both lanes of a pair take the same branch)
        float x, y, z;
        long long q = sel ? sel[p] : p;
        if (pts_smpl) {
            x = pts_smpl[3 * q]; y = pts_smpl[3 * q + 1]; z = pts_smpl[3 * q + 2];
        } else {
            float wx, wy, wz;
            th_get_point(ps, q, wx, wy, wz);
            // world2smpl, if_clight_renderer.py:289-295: (p - Th) @ Rh
            float ax = wx - Th[0], ay = wy - Th[1], az = wz - Th[2];
            x = fmaf(az, Rh[6], fmaf(ay, Rh[3], ax * Rh[0]));
            y = fmaf(az, Rh[7], fmaf(ay, Rh[4], ax * Rh[1]));
            z = fmaf(az, Rh[8], fmaf(ay, Rh[5], ax * Rh[2]));
        }
        float bd[DP_K];
        int bi[DP_K];
#pragma unroll
        for (int k = 0; k < DP_K; ++k) { bd[k] = 3.0e38f; bi[k] = 0x7fffffff; }
        for (int c = half; c < nc; c += 2) {
            float dx = x - cen[3 * c], dy = y - cen[3 * c + 1], dz = z - cen[3 * c + 2];
            float d2 = dx * dx + dy * dy;
            d2 = d2 + dz * dz;
            if (d2 < bd[DP_K - 1]) dp_insert(bd, bi, d2, c);   // ascending c: a tie never displaces an earlier index
        }
        // merge the partner's list (7 candidates) -> global top-7 ordered by (d2, index)
        float od[DP_K];
        int oi[DP_K];
#pragma unroll
        for (int k = 0; k < DP_K; ++k) { od[k] = __shfl_xor(bd[k], 1); oi[k] = __shfl_xor(bi[k], 1); }
#pragma unroll
        for (int k = 0; k < DP_K; ++k)
            if (od[k] < bd[DP_K - 1] || (od[k] == bd[DP_K - 1] && oi[k] < bi[DP_K - 1])) dp_insert(bd, bi, od[k], oi[k]);
        if (half == 0) {
            // softmax(-d/alpha) over the K neighbours
            float xs[DP_K], mx = -3.0e38f;
#pragma unroll
            for (int k = 0; k < DP_K; ++k) {
                xs[k] = (-__fsqrt_rn(bd[k])) / alpha;              // cross_transformer.py:153-154
                mx = fmaxf(mx, xs[k]);
            }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < DP_K; ++k) { xs[k] = expf(xs[k] - mx); se = se + xs[k]; }
            DpNbr& o = nb[ls];
#pragma unroll
            for (int k = 0; k < DP_K; ++k) {
                int c = bi[k];
                o.w[k] = xs[k] / se;
                o.idx[k] = c;
                float rx = x - cen[3 * c], ry = y - cen[3 * c + 1], rz = z - cen[3 * c + 2];
                const float* Rm = rot + 9 * c;
                o.def[k][0] = fmaf(rz, Rm[6], fmaf(ry, Rm[3], rx * Rm[0]));
                o.def[k][1] = fmaf(rz, Rm[7], fmaf(ry, Rm[4], rx * Rm[1]));
                o.def[k][2] = fmaf(rz, Rm[8], fmaf(ry, Rm[5], rx * Rm[2]));
            }
        }
    }
    __syncthreads();

    // ---- phase 2: wave per sample ----
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar)
    const float PI_F = 3.14159274101257324219f;          // fp32(pi)
    const float HALF_PI_F = 1.57079637050628662109f;     // fp32(pi/2)
    // PE channel `lane` (0..62): 0..2 raw xyz; then per octave f: sin xyz, cos xyz
    int axis = 0, oct = 0;
    float phase = 0.f;
    if (lane < 3) axis = lane;
    else if (lane < 63) {
        int qq = lane - 3;
        oct = qq / 6;
        int r = qq % 6;
        axis = r % 3;
        phase = (r >= 3) ? HALF_PI_F : 0.f;
    }
    const float freq = PI_F * (float)(1 << oct);
    for (int lp = wave; lp < DP_SAMPLES; lp += DP_THREADS / 64) {
        int gp = blockIdx.x * DP_SAMPLES + lp;
        if (gp >= P) break;
        const DpNbr& n = nb[lp];
        float w[DP_K];
        int id[DP_K];
#pragma unroll
        for (int k = 0; k < DP_K; ++k) {
            // wave-uniform (one sample per wave): keep them in scalar registers so the row addresses are scalar
            // arithmetic and the weights are scalar operands of the FMAs
            w[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, n.w[k])));
            id[k] = __builtin_amdgcn_readfirstlane(n.idx[k]);
        }
        float pe = 0.f;
        if (lane < 63) {
#pragma unroll
            for (int k = 0; k < DP_K; ++k) {
                float xv = n.def[k][axis];
                float val = (lane < 3) ? xv : dp_sin(fmaf(xv, freq, phase));
                pe = (k == 0) ? w[k] * val : fmaf(w[k], val, pe);
            }
        }
        if (FOLDED) {
            for (int v = 0; v < V; ++v) {
                const float* tv = tokens + (long long)v * nc * 256;
                float4 r[DP_K];
#pragma unroll
                for (int k = 0; k < DP_K; ++k) r[k] = *reinterpret_cast<const float4*>(tv + (long long)id[k] * 256 + 4 * lane);
                float4 acc = make_float4(w[0] * r[0].x, w[0] * r[0].y, w[0] * r[0].z, w[0] * r[0].w);
#pragma unroll
                for (int k = 1; k < DP_K; ++k) {
                    acc.x = fmaf(w[k], r[k].x, acc.x); acc.y = fmaf(w[k], r[k].y, acc.y);
                    acc.z = fmaf(w[k], r[k].z, acc.z); acc.w = fmaf(w[k], r[k].w, acc.w);
                }
                *reinterpret_cast<float4*>(out + ((long long)gp * V + v) * 256 + 4 * lane) = acc;
            }
            _Float16* ph = reinterpret_cast<_Float16*>(pe_out) + (long long)gp * 128;
            _Float16 x, y;
            dp_split((lane < 63) ? pe : 0.f, x, y);
            ph[lane] = x;
            ph[64 + lane] = y;
            continue;
        }
        for (int v = 0; v < V; ++v) {
            const float* tv = tokens + (long long)v * nc * 192;
            float* o = out + ((long long)gp * V + v) * 256;
            if (lane < 48) {
                float4 r[DP_K];
#pragma unroll
                for (int k = 0; k < DP_K; ++k) r[k] = *reinterpret_cast<const float4*>(tv + (long long)id[k] * 192 + 4 * lane);
                float4 acc = make_float4(w[0] * r[0].x, w[0] * r[0].y, w[0] * r[0].z, w[0] * r[0].w);
#pragma unroll
                for (int k = 1; k < DP_K; ++k) {
                    acc.x = fmaf(w[k], r[k].x, acc.x); acc.y = fmaf(w[k], r[k].y, acc.y);
                    acc.z = fmaf(w[k], r[k].z, acc.z); acc.w = fmaf(w[k], r[k].w, acc.w);
                }
                *reinterpret_cast<float4*>(o + 4 * lane) = acc;
            }
            o[192 + lane] = (lane < 63) ? pe : 0.f;
        }
    }
}

int th_dparf_launch(const float* pts_smpl, const ThPointSrc* ps, const float* Rh, const float* Th,
                    const int32_t* sel, int P, const float* centres, const float* rot, const float* tokens, int V,
                    int nc, float alpha, float* out, float* pe_out, int fmt, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE(nc >= DP_K, "need at least 7 token centres");
    ThPointSrc src;
    if (ps) src = *ps; else { src = ThPointSrc{}; }
    size_t lds = ((size_t)((nc * 3 + 3) & ~3)) * sizeof(float) + DP_SAMPLES * sizeof(DpNbr);
    TH_REQUIRE(lds <= 160 * 1024, "too many token centres for LDS staging");
    static bool attr_set = false;
    if (!attr_set) {
        TH_HIP(hipFuncSetAttribute((const void*)dparf_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TH_HIP(hipFuncSetAttribute((const void*)dparf_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    TH_REQUIRE(fmt == TH_ROWS_F32 || (fmt == TH_ROWS_FOLDED && pe_out != nullptr), "K4 writes fp32 rows or the folded form");
    if (fmt == TH_ROWS_FOLDED)
        hipLaunchKernelGGL(dparf_kernel<true>, dim3(th_cdiv(P, DP_SAMPLES)), dim3(DP_THREADS), lds, s, pts_smpl, src, Rh,
                           Th, sel, P, centres, rot, tokens, V, nc, alpha, out, pe_out);
    else
        hipLaunchKernelGGL(dparf_kernel<false>, dim3(th_cdiv(P, DP_SAMPLES)), dim3(DP_THREADS), lds, s, pts_smpl, src, Rh,
                           Th, sel, P, centres, rot, tokens, V, nc, alpha, out, pe_out);
    TH_LAUNCH_CHECK();
    return 0;
}
