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
#include <stdlib.h>

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

// ---- candidate grid over the token centres (exact pruning of the 7-NN scan) ---------------------------------
// Cell size g; for a cell with centre q and half-diagonal h let d7(q) be the distance from q to its 7th nearest
// token centre.  For any point p of the cell d7(p) <= d7(q) + |p - q| <= d7(q) + h, hence every one of p's 7
// nearest centres t satisfies |t - q| <= |t - p| + |p - q| <= d7(q) + 2h.  The cell's list holds all centres
// within that radius (plus a rounding margin), in ascending index order: scanning it gives the same 7
// neighbours, distances and tie order as scanning all N_c centres, with ~70 instead of N_c candidates.  Points
// outside the grid fall back to the full scan.  Built once per frame (thousands of tiny blocks, ~20 us).
// Round 6: the cell size follows the token density -- g = DPG_CELL cbrt(500 / N_c) -- so a list stays at ~55 candidates whatever N_c
// (with the fixed 0.075 m cell the lists of the reference's kmeans_dict_1500 held ~3 x as many: K4 3.1 ms against 0.57 at
// N_c = 500).  Lists live in slots of DPG_STRIDE entries; a cell whose list does not fit is marked (count -1) and its points
// take the full scan -- exact either way.
#define DPG_MAXCELLS 32768
#define DPG_STRIDE 192
#define DPG_CELL 0.075f        // cell size asked for at N_c = 500; 0.1: lists of ~70 candidates, 0.075: ~55, K4 0.73 -> 0.71 ms; (dpgrid_setup_kernel grows it until the grid fits DPG_MAXCELLS)
struct DpGrid {
    float gmin[3];
    float g, inv_g;
    int dim[3];
    int ncell;          // 0: grid disabled (too many cells)
};

__global__ __launch_bounds__(256) void dpgrid_setup_kernel(const float* __restrict__ centres, int nc, float g, float margin,
                                                           DpGrid* __restrict__ gi) {
    __shared__ float red[6][256];
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = threadIdx.x; i < nc; i += 256)
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], centres[3 * i + a]); mx[a] = fmaxf(mx[a], centres[3 * i + a]); }
    for (int a = 0; a < 3; ++a) { red[a][threadIdx.x] = mn[a]; red[3 + a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int a = 0; a < 3; ++a) {
                red[a][threadIdx.x] = fminf(red[a][threadIdx.x], red[a][threadIdx.x + s]);
                red[3 + a][threadIdx.x] = fmaxf(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float ext[3];
        for (int a = 0; a < 3; ++a) ext[a] = (red[3 + a][0] - red[a][0]) + 2.f * margin;
        g = fminf(fmaxf(g * cbrtf(500.f / (float)nc), 0.04f), 0.12f);
        // grow the cell until the grid fits
        for (int it = 0; it < 32; ++it) {
            long long n = 1;
            for (int a = 0; a < 3; ++a) n *= (long long)ceilf(ext[a] / g);
            if (n <= DPG_MAXCELLS) break;
            g *= 1.25f;
        }
        long long n = 1;
        for (int a = 0; a < 3; ++a) {
            gi->gmin[a] = red[a][0] - margin;
            gi->dim[a] = max(1, (int)ceilf(ext[a] / g));
            n *= gi->dim[a];
        }
        gi->g = g;
        gi->inv_g = 1.0f / g;
        gi->ncell = (n <= DPG_MAXCELLS && nc >= DP_K) ? (int)n : 0;
    }
}

// one WAVE per cell (four cells per workgroup): radius = d7(cell centre) + 2h, list of the centres inside it (ascending
// index).  No barrier anywhere: the seven selection rounds are wave reductions over (value, index) pairs, the ordered
// compaction is a ballot + prefix count per 64 consecutive indices.  (The first form -- one 256-thread workgroup per cell,
// seven block-wide reductions and a serial scan -- took 90 us per frame in front of K4; same lists, entry for entry.)
__global__ __launch_bounds__(256) void dpgrid_fill_kernel(const float* __restrict__ centres, int nc,
                                                          const DpGrid* __restrict__ gi, int* __restrict__ cell_count,
                                                          int* __restrict__ cand) {
    extern __shared__ float dsq_all[];                   // [4][nc] squared distances to the cell centres
    const DpGrid g = *gi;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cell = blockIdx.x * 4 + wave;
    if (cell >= g.ncell) return;
    float* dsq = dsq_all + (size_t)wave * nc;
    const int cx = cell % g.dim[0], cy = (cell / g.dim[0]) % g.dim[1], cz = cell / (g.dim[0] * g.dim[1]);
    const float qx = g.gmin[0] + (cx + 0.5f) * g.g, qy = g.gmin[1] + (cy + 0.5f) * g.g, qz = g.gmin[2] + (cz + 0.5f) * g.g;
    for (int i = lane; i < nc; i += 64) {
        float dx = qx - centres[3 * i], dy = qy - centres[3 * i + 1], dz = qz - centres[3 * i + 2];
        dsq[i] = dx * dx + dy * dy + dz * dz;
    }
    // A cell whose centre is further than `skip` from EVERY token centre holds no sample of the hull (samples lie within 0.1 m of a
    // posed vertex, a vertex within its cluster's radius of the cluster mean): no list -- a point that falls into it anyway takes the
    // full scan (count -1), exact like every other.  Most cells of the body's bounding box are such air: at N_c = 1500 (32 k cells)
    // the seven selection rounds over 1500 distances per cell made this kernel 0.30 ms per frame.
    {
        float mn = 3e38f;
        for (int i = lane; i < nc; i += 64) mn = fminf(mn, dsq[i]);
        for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
        const float skip = 0.30f + 0.8660254f * g.g;
        if (mn > skip * skip) {
            if (lane == 0) cell_count[cell] = -1;
            return;
        }
    }
    // 7th smallest squared distance: seven rounds of (min over values after the previous pick, by (value, index))
    float pv = -1.f;
    int pi = -1;
    for (int round = 0; round < DP_K; ++round) {
        float bv = 3e38f;
        int bi = 0x7fffffff;
        for (int i = lane; i < nc; i += 64) {
            const float v = dsq[i];                      // (a lane reads back what it wrote: no cross-lane hazard)
            const bool after = (v > pv) || (v == pv && i > pi);
            if (after && (v < bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        pv = bv; pi = bi;
    }
    const float h = 0.8660254f * g.g;                    // half diagonal of the cell
    const float rad = sqrtf(pv) * 1.0001f + 2.f * h * 1.0001f + 1e-5f;
    const float r2 = rad * rad;
    int* dst = cand + (long long)cell * DPG_STRIDE;
    int o = 0;
    for (int base = 0; base < nc; base += 64) {
        const int i = base + lane;
        const bool in = i < nc && dsq[i] <= r2;
        const unsigned long long m = __ballot(in);
        const int pos = o + __popcll(m & ((1ull << lane) - 1ull));
        if (in && pos < DPG_STRIDE) dst[pos] = i;
        o += __popcll(m);
    }
    if (lane == 0) cell_count[cell] = o <= DPG_STRIDE ? o : -1;      // (-1: the list does not fit its slot -> full scan)
}

size_t th_dparf_grid_ws(int nc) {
    return th_align(sizeof(DpGrid)) + th_align((size_t)DPG_MAXCELLS * 4) + th_align((size_t)DPG_MAXCELLS * DPG_STRIDE * 4);
}

int th_dparf_grid_build(const float* centres, int nc, void* ws, size_t ws_bytes, hipStream_t s) {
    TH_REQUIRE(ws_bytes >= th_dparf_grid_ws(nc), "grid workspace too small");
    TH_REQUIRE(th_dparf_grid_ok(nc), "too many (or fewer than 7) token centres for the grid builder");
    ThArena ar(ws, ws_bytes);
    DpGrid* gi = ar.take<DpGrid>(1);
    int* cnt = ar.take<int>(DPG_MAXCELLS);
    int* cand = ar.take<int>((size_t)DPG_MAXCELLS * DPG_STRIDE);
    TH_REQUIRE(cand != nullptr, "grid workspace carve failed");
    hipLaunchKernelGGL(dpgrid_setup_kernel, dim3(1), dim3(256), 0, s, centres, nc, DPG_CELL, 0.25f, gi);
    hipLaunchKernelGGL(dpgrid_fill_kernel, dim3(DPG_MAXCELLS / 4), dim3(256), (size_t)nc * 16, s, centres, nc, gi, cnt, cand);
    TH_LAUNCH_CHECK();
    return 0;
}

template <bool FOLDED, int VT = 0>
__global__ __launch_bounds__(DP_THREADS) void dparf_kernel(const float* __restrict__ pts_smpl, ThPointSrc ps,
                                                           const float* __restrict__ Rh, const float* __restrict__ Th,
                                                           const int32_t* __restrict__ sel, int P,
                                                           const float* __restrict__ centres,
                                                           const float* __restrict__ rot,
                                                           const float* __restrict__ tokens, int V, int nc,
                                                           float alpha, float* __restrict__ out,
                                                           float* __restrict__ pe_out, const DpGrid* __restrict__ gi,
                                                           const int* __restrict__ cell_count,
                                                           const int* __restrict__ cand) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* cen = lds;                                   // [nc*3]
    DpNbr* nb = reinterpret_cast<DpNbr*>(lds + ((nc * 3 + 3) & ~3));
    // TH_ROWS_NBR: per 32-sample tile of the fused kernel (4 per workgroup) a bitmap over the centre indices and the slot
    // of every bitmap word's first set bit
    unsigned* ubm = reinterpret_cast<unsigned*>(nb + DP_SAMPLES);       // [4][128]
    unsigned* ubw = ubm + 4 * 128;                                       // [4][128]
    if constexpr (FOLDED && VT < 0) {
        for (int i = threadIdx.x; i < 4 * 128; i += DP_THREADS) ubm[i] = 0u;
    }
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
        // candidate list of this point's grid cell (exact superset of its 7 nearest centres), else all centres
        const int* list = nullptr;
        int nlist = nc;
        if (gi != nullptr && gi->ncell > 0) {
            const int cx = (int)floorf((x - gi->gmin[0]) * gi->inv_g), cy = (int)floorf((y - gi->gmin[1]) * gi->inv_g),
                      cz = (int)floorf((z - gi->gmin[2]) * gi->inv_g);
            if (cx >= 0 && cx < gi->dim[0] && cy >= 0 && cy < gi->dim[1] && cz >= 0 && cz < gi->dim[2]) {
                const int cell = (cz * gi->dim[1] + cy) * gi->dim[0] + cx;
                const int cnt = cell_count[cell];
                if (cnt >= 0) {
                    list = cand + (long long)cell * DPG_STRIDE;
                    nlist = cnt;
                }
            }
        }
        for (int j = half; j < nlist; j += 2) {
            const int c = list ? list[j] : j;
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
                if constexpr (FOLDED && VT < 0) atomicOr(&ubm[(ls >> 5) * 128 + (c >> 5)], 1u << (c & 31));
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
    if constexpr (FOLDED && VT < 0) {
        // Tile header (512 B behind the records): hdr[0] = U, the size of the union of the tile's neighbour sets, then
        // the centre of every slot as unsigned short, ascending.  Wave t serves tile t of this workgroup.
        const long long tile = (long long)blockIdx.x * 4 + wave;
        if (tile * 32 < P) {
            unsigned* hdr = reinterpret_cast<unsigned*>(out) + (long long)((P + 31) / 32 * 32) * 16 + tile * 128;
            unsigned short* scl = reinterpret_cast<unsigned short*>(hdr) + 2;
            unsigned w0 = ubm[wave * 128 + 2 * lane], w1 = ubm[wave * 128 + 2 * lane + 1];
            const int c0 = __popc(w0), c1 = __popc(w1);
            int incl = c0 + c1;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            int base = incl - c0 - c1;
            ubw[wave * 128 + 2 * lane] = (unsigned)base;
            ubw[wave * 128 + 2 * lane + 1] = (unsigned)(base + c0);
            if (lane == 63) hdr[0] = (unsigned)incl;
            while (w0) { const int bit = __ffs(w0) - 1; scl[base++] = (unsigned short)(64 * lane + bit); w0 &= w0 - 1; }
            while (w1) { const int bit = __ffs(w1) - 1; scl[base++] = (unsigned short)(64 * lane + 32 + bit); w1 &= w1 - 1; }
        }
        __syncthreads();
    }
    const float PI_F = 3.14159274101257324219f;          // fp32(pi)
    const float HALF_PI_F = 1.57079637050628662109f;     // fp32(pi/2)
    if constexpr (FOLDED && VT < 0) {
        // TH_ROWS_NBR writes no table rows, so nothing here is 64 lanes wide: FOUR samples per wave step, 16 lanes each.
        // Lane q of a sample owns the PE channels 4 q .. 4 q + 3 (channel 63 is the row's zero pad) -- the same 441 sines per
        // sample, every term and the order of the seven-term sums as in the wave-per-sample loop below (bit-identical rows),
        // but the per-sample overhead of that loop (weight broadcasts, record lanes, two 2-byte stores per lane) is paid
        // once per four samples and the PE halves leave as 8-byte stores.
        const int sub = lane >> 4, l16 = lane & 15;
        unsigned aoff[4];
        float fr[4], phs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = 4 * l16 + q;
            int a = 0, oc = 0;
            float ph0 = 0.f;
            if (ch < 3) a = ch;
            else if (ch < 63) {
                const int qq = ch - 3, r = qq % 6;
                oc = qq / 6;
                a = r % 3;
                ph0 = (r >= 3) ? HALF_PI_F : 0.f;
            }
            aoff[q] = (unsigned)a;
            fr[q] = PI_F * (float)(1 << oc);
            phs[q] = ph0;
        }
        for (int it = 0; it < DP_SAMPLES / 16; ++it) {
            const int lp = it * 16 + wave * 4 + sub;
            const int gp = blockIdx.x * DP_SAMPLES + lp;
            if (gp >= P) continue;
            const DpNbr& n = nb[lp];
            float pe[4];
#pragma unroll
            for (int k = 0; k < DP_K; ++k) {
                const float wk = n.w[k];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xv = n.def[k][aoff[q]];
                    float val = dp_sin(fmaf(xv, fr[q], phs[q]));
                    if (q < 3) val = (4 * l16 + q < 3) ? xv : val;          // channels 0..2: the raw offsets
                    pe[q] = (k == 0) ? wk * val : fmaf(wk, val, pe[q]);
                }
            }
            if (l16 == 15) pe[3] = 0.f;                                     // channel 63: pad
            unsigned rec = 0u;                                              // the neighbour record (16 dwords)
            if (l16 < 7) {
                const int c = n.idx[l16], t = lp >> 5;
                rec = ubw[t * 128 + (c >> 5)] + __popc(ubm[t * 128 + (c >> 5)] & ((1u << (c & 31)) - 1u));   // slot of centre c
            } else if (l16 >= 8 && l16 < 15) rec = __builtin_bit_cast(unsigned, n.w[l16 - 8]);
            reinterpret_cast<unsigned*>(out)[(long long)gp * 16 + l16] = rec;
            dp_h4 hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                _Float16 x, y;
                dp_split(pe[q], x, y);
                hi[q] = x;
                lo[q] = y;
            }
            _Float16* ph = reinterpret_cast<_Float16*>(pe_out) + (long long)gp * 128;
            *reinterpret_cast<dp_h4*>(ph + 4 * l16) = hi;
            *reinterpret_cast<dp_h4*>(ph + 64 + 4 * l16) = lo;
        }
        return;
    }
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
        if constexpr (FOLDED && VT < 0) {
            // TH_ROWS_NBR: the neighbour record (16 dwords) + the positional-encoding row; no table access
            if (lane < 16) {
                unsigned rec = 0u;
                if (lane < 7) {
                    const int c = n.idx[lane], t = lp >> 5;
                    rec = ubw[t * 128 + (c >> 5)] + __popc(ubm[t * 128 + (c >> 5)] & ((1u << (c & 31)) - 1u));   // slot of centre c
                } else if (lane >= 8 && lane < 15) rec = __builtin_bit_cast(unsigned, n.w[lane - 8]);
                reinterpret_cast<unsigned*>(out)[(long long)gp * 16 + lane] = rec;
            }
            _Float16* ph = reinterpret_cast<_Float16*>(pe_out) + (long long)gp * 128;
            _Float16 x, y;
            dp_split((lane < 63) ? pe : 0.f, x, y);
            ph[lane] = x;
            ph[64 + lane] = y;
            continue;
        }
        if constexpr (FOLDED && VT > 0) {
            // all VT * 7 row loads of the sample in flight before the first blend (one exposed round trip per sample
            // instead of one per view)
            float4 r[VT > 0 ? VT : 1][DP_K];
#pragma unroll
            for (int v = 0; v < VT; ++v)
#pragma unroll
                for (int k = 0; k < DP_K; ++k)
                    r[v][k] = *reinterpret_cast<const float4*>(tokens + ((long long)v * nc + id[k]) * 256 + 4 * lane);
#pragma unroll
            for (int v = 0; v < VT; ++v) {
                float4 acc = make_float4(w[0] * r[v][0].x, w[0] * r[v][0].y, w[0] * r[v][0].z, w[0] * r[v][0].w);
#pragma unroll
                for (int k = 1; k < DP_K; ++k) {
                    acc.x = fmaf(w[k], r[v][k].x, acc.x); acc.y = fmaf(w[k], r[v][k].y, acc.y);
                    acc.z = fmaf(w[k], r[v][k].z, acc.z); acc.w = fmaf(w[k], r[v][k].w, acc.w);
                }
                *reinterpret_cast<float4*>(out + ((long long)gp * VT + v) * 256 + 4 * lane) = acc;
            }
            _Float16* ph = reinterpret_cast<_Float16*>(pe_out) + (long long)gp * 128;
            _Float16 x, y;
            dp_split((lane < 63) ? pe : 0.f, x, y);
            ph[lane] = x;
            ph[64 + lane] = y;
            continue;
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
                    int nc, float alpha, float* out, float* pe_out, int fmt, const void* grid_ws, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE(nc >= DP_K, "need at least 7 token centres");
    ThPointSrc src;
    if (ps) src = *ps; else { src = ThPointSrc{}; }
    size_t lds = ((size_t)((nc * 3 + 3) & ~3)) * sizeof(float) + DP_SAMPLES * sizeof(DpNbr) + 2 * 4 * 128 * sizeof(unsigned);
    TH_REQUIRE(lds <= 160 * 1024, "too many token centres for LDS staging");
    static unsigned long long attr_set = 0ull;
    if (th_lds_attr_needed(&attr_set)) {
        TH_HIP(hipFuncSetAttribute((const void*)dparf_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TH_HIP(hipFuncSetAttribute((const void*)dparf_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    TH_REQUIRE(fmt == TH_ROWS_F32 || ((fmt == TH_ROWS_FOLDED || fmt == TH_ROWS_NBR) && pe_out != nullptr),
               "K4 writes fp32 rows, the folded form or neighbour records");
    // optional candidate grid (th_dparf_grid_build into grid_ws): same carve as the builder
    const DpGrid* gi = nullptr;
    const int *cnt = nullptr, *cand = nullptr;
    if (grid_ws != nullptr) {
        ThArena ar(const_cast<void*>(grid_ws), th_dparf_grid_ws(nc));
        gi = ar.take<DpGrid>(1);
        cnt = ar.take<int>(DPG_MAXCELLS);
        cand = ar.take<int>((size_t)DPG_MAXCELLS * DPG_STRIDE);
    }
    static const bool per_view = getenv("TH_DPARF_PER_VIEW") != nullptr;       // A/B switch
    if (fmt == TH_ROWS_NBR) {
        static unsigned long long attrn = 0ull;
        if (th_lds_attr_needed(&attrn))
            TH_HIP(hipFuncSetAttribute((const void*)dparf_kernel<true, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((dparf_kernel<true, -1>), dim3(th_cdiv(P, DP_SAMPLES)), dim3(DP_THREADS), lds, s, pts_smpl, src, Rh,
                           Th, sel, P, centres, rot, tokens, V, nc, alpha, out, pe_out, gi, cnt, cand);
    } else if (fmt == TH_ROWS_FOLDED && V == 3 && !per_view) {
        static unsigned long long attr3 = 0ull;
        if (th_lds_attr_needed(&attr3))
            TH_HIP(hipFuncSetAttribute((const void*)dparf_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((dparf_kernel<true, 3>), dim3(th_cdiv(P, DP_SAMPLES)), dim3(DP_THREADS), lds, s, pts_smpl, src, Rh,
                           Th, sel, P, centres, rot, tokens, V, nc, alpha, out, pe_out, gi, cnt, cand);
    } else if (fmt == TH_ROWS_FOLDED)
        hipLaunchKernelGGL(dparf_kernel<true>, dim3(th_cdiv(P, DP_SAMPLES)), dim3(DP_THREADS), lds, s, pts_smpl, src, Rh,
                           Th, sel, P, centres, rot, tokens, V, nc, alpha, out, pe_out, gi, cnt, cand);
    else
        hipLaunchKernelGGL(dparf_kernel<false>, dim3(th_cdiv(P, DP_SAMPLES)), dim3(DP_THREADS), lds, s, pts_smpl, src, Rh,
                           Th, sel, P, centres, rot, tokens, V, nc, alpha, out, pe_out, gi, cnt, cand);
    TH_LAUNCH_CHECK();
    return 0;
}
