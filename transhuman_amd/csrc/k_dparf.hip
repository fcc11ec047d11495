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
// Layout: one workgroup = 128 samples.  Phase 1 (thread per sample): brute-force
// top-7 over the N_c centres staged in LDS (N_c*12 B <= 72 KB), weights and the
// 7 rotated offsets go to LDS.  Phase 2 (wave per sample): lanes span channels,
// token rows are gathered as coalesced 768 B reads from the L2-resident table,
// lanes 0..62 evaluate one PE channel each (7 accurate sinf per lane).
// Output rows [sample][view][256] (255 + zero pad) feed the fc_0 GEMM.
// Bound: L2 gather of 7*V*768 B per sample; HBM write 3 KB per sample.
#include "th_internal.h"

#define DP_K 7
#define DP_BLOCK 128
#define DP_FREQ 10

struct DpNbr {
    float w[DP_K];
    int idx[DP_K];
    float def[DP_K][3];
};

__global__ __launch_bounds__(DP_BLOCK) void dparf_kernel(const float* __restrict__ pts_smpl, ThPointSrc ps,
                                                         const float* __restrict__ Rh, const float* __restrict__ Th,
                                                         const int32_t* __restrict__ sel, int P,
                                                         const float* __restrict__ centres,
                                                         const float* __restrict__ rot,
                                                         const float* __restrict__ tokens, int V, int nc,
                                                         float alpha, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cen = lds;                                   // [nc*3]
    DpNbr* nb = reinterpret_cast<DpNbr*>(lds + ((nc * 3 + 3) & ~3));
    for (int i = threadIdx.x; i < nc * 3; i += DP_BLOCK) cen[i] = centres[i];
    __syncthreads();

    const int p = blockIdx.x * DP_BLOCK + threadIdx.x;
    if (p < P) {
        float x, y, z;
        if (pts_smpl) {
            long long q = sel ? sel[p] : p;
            x = pts_smpl[3 * q]; y = pts_smpl[3 * q + 1]; z = pts_smpl[3 * q + 2];
        } else {
            long long q = sel ? sel[p] : p;
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
        for (int k = 0; k < DP_K; ++k) { bd[k] = 3.0e38f; bi[k] = 0; }
        for (int c = 0; c < nc; ++c) {
            float dx = x - cen[3 * c], dy = y - cen[3 * c + 1], dz = z - cen[3 * c + 2];
            float d2 = dx * dx + dy * dy;
            d2 = d2 + dz * dz;
            if (d2 < bd[DP_K - 1]) {
                // stable insertion: goes after every element <= d2
                float cd = d2;
                int ci = c;
#pragma unroll
                for (int k = 0; k < DP_K; ++k) {
                    bool sw = cd < bd[k];
                    float td = sw ? bd[k] : cd;
                    int ti = sw ? bi[k] : ci;
                    bd[k] = sw ? cd : bd[k];
                    bi[k] = sw ? ci : bi[k];
                    cd = td; ci = ti;
                }
            }
        }
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
        DpNbr& o = nb[threadIdx.x];
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
    __syncthreads();

    // ---- phase 2: wave per sample ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float PI_F = 3.14159274101257324219f;          // fp32(pi)
    const float HALF_PI_F = 1.57079637050628662109f;     // fp32(pi/2)
    for (int lp = wave; lp < DP_BLOCK; lp += DP_BLOCK / 64) {
        int gp = blockIdx.x * DP_BLOCK + lp;
        if (gp >= P) break;
        const DpNbr& n = nb[lp];
        float w[DP_K];
        int id[DP_K];
#pragma unroll
        for (int k = 0; k < DP_K; ++k) { w[k] = n.w[k]; id[k] = n.idx[k]; }
        // PE channel `lane` (0..62): 0..2 raw xyz; then per octave f: sin xyz, cos xyz
        float pe = 0.f;
        if (lane < 63) {
            int axis, oct = 0;
            float phase = 0.f;
            if (lane < 3) axis = lane;
            else {
                int q = lane - 3;
                oct = q / 6;
                int r = q % 6;
                axis = r % 3;
                phase = (r >= 3) ? HALF_PI_F : 0.f;
            }
            float freq = PI_F * (float)(1 << oct);
#pragma unroll
            for (int k = 0; k < DP_K; ++k) {
                float xv = n.def[k][axis];
                float val = (lane < 3) ? xv : sinf(fmaf(xv, freq, phase));
                float t = w[k] * val;
                pe = (k == 0) ? t : pe + t;
            }
        }
        for (int v = 0; v < V; ++v) {
            const float* tv = tokens + (long long)v * nc * 192;
            float* o = out + ((long long)gp * V + v) * 256;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                int c = lane + 64 * j;
                float acc = w[0] * tv[(long long)id[0] * 192 + c];
#pragma unroll
                for (int k = 1; k < DP_K; ++k) acc = acc + w[k] * tv[(long long)id[k] * 192 + c];
                o[c] = acc;
            }
            o[192 + lane] = (lane < 63) ? pe : 0.f;
        }
    }
}

int th_dparf_launch(const float* pts_smpl, const ThPointSrc* ps, const float* Rh, const float* Th,
                    const int32_t* sel, int P, const float* centres, const float* rot, const float* tokens, int V,
                    int nc, float alpha, float* out, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE(nc >= DP_K, "need at least 7 token centres");
    ThPointSrc src;
    if (ps) src = *ps; else { src = ThPointSrc{}; }
    size_t lds = ((size_t)((nc * 3 + 3) & ~3)) * sizeof(float) + DP_BLOCK * sizeof(DpNbr);
    TH_REQUIRE(lds <= 160 * 1024, "too many token centres for LDS staging");
    static bool attr_set = false;
    if (!attr_set) {
        TH_HIP(hipFuncSetAttribute((const void*)dparf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(dparf_kernel, dim3(th_cdiv(P, DP_BLOCK)), dim3(DP_BLOCK), lds, s, pts_smpl, src, Rh, Th, sel,
                       P, centres, rot, tokens, V, nc, alpha, out);
    TH_LAUNCH_CHECK();
    return 0;
}
