// K5t: texel hand-over of the pixel-aligned features (TH_ROWS_TEX).
//
// get_pixel_aligned_feature (if_clight_renderer.py:210-269) bilinearly samples the pixel map at the projection of every
// sample in every reference view.  K5 (k_pixfeat.hip) forms those rows and hands them to the fused MLP through HBM: 1088
// bytes written per (sample, view) and read back twice (sigma and RGB branch), 20 GB per 512 x 512 x 64 frame, plus a
// 2.4 ms kernel whose only product they are.  The 32 samples of a tile of the fused kernel are image neighbours in every
// view (depth-major sample list inside 16-ray groups), so their 3 x 128 corner texels are only ~79 DISTINCT 1 KiB texel
// rows of the map.  This kernel therefore hands over, per tile, the LIST of distinct texels (all views in one pool of rows)
// and, per (sample, view), the four row numbers + the four bilinear weights; the fused kernel copies the listed rows into LDS
// and blends them itself (K5's weights and term order).  Since late round 4 the rows it copies are not the latents but
// the texels of the two FOLDED maps (map_fold_kernel, k_mlp_fused_kernel.h: alpha_res_0 / rgb_res_0 / rgb_res_1 applied to the
// map once per frame -- bilinear sampling commutes with linear layers), so this kernel needs no map at all: cameras only.
//
// Output (per launch of P samples, T = ceil(P / 32) tiles):
//   hdr  [T][4 passes][128 words]   word 0 = U | npass << 16 (U = texel rows of this pass, npass = 1, 2 or 4),
//                                   words 8 .. 8 + U - 1 = global texel index (view * H * W + y * W + x) of row 0 .. U - 1
//   rec  [T][V][32 samples][8 words] {w00, w01, w10, w11} {byte offsets of the nw, ne, sw, se texel rows in the fused kernel's
//                                   row buffer: row number x 1040}
// A pass holds at most TX_CAP rows (what the fused kernel's operand buffer takes).  5.8 % of the headline frame's tiles need
// more: their samples are split into halves (2 passes) or quarters (4 passes: 8 samples x 4 corners x 3 views = 96 rows
// always fit), each with its own row list; sample s belongs to pass s / (32 / npass).
// One wave per tile; lane l and lane l + 32 both carry sample l (the upper half idles through the ballots).
#include "th_internal.h"

#define TX_CAP 103
#define TX_HDR_WORDS 512
#define TX_MAXV 3
#define TX_ROW_STRIDE 1040     // bytes per texel row in the fused kernel's row buffer (fill_tex: TSTR)

// ---- distinct texels of one view over the samples of a mask ------------------------------------------------------------
// Lane l (and its mirror l + 32) carries sample l: the clamped texel coordinates of its four corners, packed as
// px = x0 | x1 << 16, py = y0 | y1 << 16 (corner order nw ne sw se = (x0,y0) (x1,y0) (x0,y1) (x1,y1)).  The texels are numbered
// from `base` on; slot[k] = number of corner k's texel; the global texel index (goff + y W + x) of number u goes to hw[8 + u].
// Returns the number of distinct texels of the view.
//
// Fast form (94-96 % of the (tile, view) pairs of the headline frame): the corners' bounding box holds at most 64 texels -> one
// bit per texel of the box, the lanes' bits OR-reduced over the wave, numbers = population counts below the bit (texels in
// raster order of the box), and lane b writes the index of texel b.  ~110 instructions instead of ~30 per distinct texel.
template <int CTRL>
__device__ __forceinline__ unsigned tx_dpp(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
typedef unsigned short tx_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned tx_pkmin(unsigned a, unsigned b) {
    tx_us2 r = __builtin_elementwise_min(__builtin_bit_cast(tx_us2, a), __builtin_bit_cast(tx_us2, b));
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned tx_pkmax(unsigned a, unsigned b) {
    tx_us2 r = __builtin_elementwise_max(__builtin_bit_cast(tx_us2, a), __builtin_bit_cast(tx_us2, b));
    return __builtin_bit_cast(unsigned, r);
}
// reductions over lanes 0..31 (rows 0 and 1 of the wave; the mirrored upper half holds the same values): every lane of a
// 16-lane row ends with the row's result (quad_perm x 2, row_half_mirror, row_mirror), the two rows meet through readlane
#define TX_ROW_REDUCE(v, OP)                \
    v = OP(v, tx_dpp<0xB1>(v));             \
    v = OP(v, tx_dpp<0x4E>(v));             \
    v = OP(v, tx_dpp<0x141>(v));            \
    v = OP(v, tx_dpp<0x140>(v));
__device__ __forceinline__ unsigned tx_or(unsigned a, unsigned b) { return a | b; }

__device__ __forceinline__ int tx_dedup(unsigned px, unsigned py, unsigned M, bool inM, int base, int goff, int W, int lane,
                                        int (&slot)[4], unsigned* __restrict__ hw) {
    const int x0 = (int)(px & 0xffffu), x1 = (int)(px >> 16), y0 = (int)(py & 0xffffu), y1 = (int)(py >> 16);
    // bounding box of the mask's corners: packed 16-bit minima (x0 | y0 << 16) and maxima (x1 | y1 << 16)
    unsigned mn = inM ? ((unsigned)x0 | ((unsigned)y0 << 16)) : 0xffffffffu;
    unsigned mx = inM ? ((unsigned)x1 | ((unsigned)y1 << 16)) : 0u;
    TX_ROW_REDUCE(mn, tx_pkmin)
    TX_ROW_REDUCE(mx, tx_pkmax)
    const unsigned mn2 = tx_pkmin((unsigned)__builtin_amdgcn_readlane((int)mn, 0), (unsigned)__builtin_amdgcn_readlane((int)mn, 16));
    const unsigned mx2 = tx_pkmax((unsigned)__builtin_amdgcn_readlane((int)mx, 0), (unsigned)__builtin_amdgcn_readlane((int)mx, 16));
    const int xmin = (int)(mn2 & 0xffffu), ymin = (int)(mn2 >> 16);
    const int bw = (int)(mx2 & 0xffffu) - xmin + 1, bh = (int)(mx2 >> 16) - ymin + 1;
    if (bw * bh <= 64) {
        const int bx0 = x0 - xmin, bx1 = x1 - xmin, r0 = (y0 - ymin) * bw, r1 = (y1 - ymin) * bw;
        const int bit[4] = {r0 + bx0, r0 + bx1, r1 + bx0, r1 + bx1};
        unsigned lo = 0u, hi = 0u;
        if (inM) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (bit[k] < 32) lo |= 1u << bit[k];
                else hi |= 1u << (bit[k] - 32);
            }
        }
        TX_ROW_REDUCE(lo, tx_or)
        TX_ROW_REDUCE(hi, tx_or)
        const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)lo, 0) | (unsigned)__builtin_amdgcn_readlane((int)lo, 16);
        const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)hi, 0) | (unsigned)__builtin_amdgcn_readlane((int)hi, 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = bit[k];
            const unsigned ml = b < 32 ? ((1u << b) - 1u) : 0xffffffffu;
            const unsigned mh = b < 32 ? 0u : ((1u << (b - 32)) - 1u);
            slot[k] = base + __builtin_popcount(mlo & ml) + __builtin_popcount(mhi & mh);
        }
        // lane b: texel b of the box, if any corner touches it
        const bool on = lane < 32 ? ((mlo >> lane) & 1u) != 0u : ((mhi >> (lane - 32)) & 1u) != 0u;
        const int rank = (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        // lane / bw by a float product (lane + 0.5 is never within 0.5 / 64 of a multiple of bw: the floor is exact)
        const int q = (int)(((float)lane + 0.5f) * (1.0f / (float)bw));
        const int id = (ymin + q) * W + xmin + (lane - q * bw);
        if (on) hw[8 + base + rank] = (unsigned)(goff + id);
        return __builtin_popcount(mlo) + __builtin_popcount(mhi);
    }
    // general form: texels numbered in order of first appearance, one ballot round per distinct texel
    const int c[4] = {y0 * W + x0, y0 * W + x1, y1 * W + x0, y1 * W + x1};
    unsigned un[4] = {M, M, M, M};
    int U = 0;
    while ((un[0] | un[1] | un[2] | un[3]) != 0u) {
        const int k = un[0] ? 0 : un[1] ? 1 : un[2] ? 2 : 3;
        const int src = __builtin_ctz(un[k]);
        const int id = k == 0 ? __builtin_amdgcn_readlane(c[0], src)
                     : k == 1 ? __builtin_amdgcn_readlane(c[1], src)
                     : k == 2 ? __builtin_amdgcn_readlane(c[2], src)
                              : __builtin_amdgcn_readlane(c[3], src);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool eq = inM && c[j] == id;
            const unsigned m = (unsigned)__builtin_amdgcn_ballot_w64(eq) & un[j];
            if (eq) slot[j] = base + U;
            un[j] &= ~m;
        }
        if (lane == 0) hw[8 + base + U] = (unsigned)(goff + id);
        ++U;
    }
    return U;
}

template <int V>
__global__ __launch_bounds__(256) void pixtex_kernel(int H, int W, ThPointSrc ps,
                                                     const int32_t* __restrict__ sel, int P, const float* __restrict__ cams,
                                                     const float* __restrict__ scale, unsigned* __restrict__ hdr,
                                                     unsigned* __restrict__ rec, int cap) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pbase = tile * 32;
    if (pbase >= P) return;
    const int npts = min(32, P - pbase);
    const int smp = lane & 31;
    const int HW = H * W;
    // ---- projection, corners and weights of this lane's sample in every view (K5's phase 1)
    unsigned px[V], py[V];
    {
        const int p = pbase + min(smp, npts - 1);          // ragged last tile: the last sample again (rows never stored)
        const long long s = sel ? sel[p] : p;
        float x, y, z;
        th_get_point(ps, s, x, y, z);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float uu, vv;
            th_project(cams + 21 * v, x, y, z, uu, vv);
            const Bilin b = th_bilinear_setup(uu, vv, scale[0], scale[1], H, W);
            px[v] = (unsigned)b.x0 | ((unsigned)b.x1 << 16);
            py[v] = (unsigned)b.y0 | ((unsigned)b.y1 << 16);
            if (lane < 32)
                *reinterpret_cast<uint4*>(rec + ((long long)(tile * V + v) * 32 + smp) * 8) =
                    make_uint4(__builtin_bit_cast(unsigned, b.w00), __builtin_bit_cast(unsigned, b.w01),
                               __builtin_bit_cast(unsigned, b.w10), __builtin_bit_cast(unsigned, b.w11));
        }
    }
    // ---- row lists: one pass if the tile's distinct texels fit, else halves, else quarters
    unsigned* hb = hdr + (long long)tile * TX_HDR_WORDS;
    for (int np = 1; np <= 4; np *= 2) {
        bool ok = true;
        const int per = 32 / np;
        for (int p = 0; p < np && ok; ++p) {
            const unsigned M = per == 32 ? 0xffffffffu : (((1u << per) - 1u) << (p * per));
            const bool inM = ((M >> smp) & 1u) != 0u;
            int rows[V][4];
            int U = 0;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                int slot[4] = {0, 0, 0, 0};
                U += tx_dedup(px[v], py[v], M, inM, U, v * HW, W, lane, slot, hb + p * 128);
#pragma unroll
                for (int j = 0; j < 4; ++j) rows[v][j] = slot[j] * TX_ROW_STRIDE;
                if (U > cap && np < 4) break;
            }
            if (U > cap && np < 4) { ok = false; break; }
            if (lane == 0) hb[p * 128] = (unsigned)U | ((unsigned)np << 16);
            if (inM && lane < 32) {
#pragma unroll
                for (int v = 0; v < V; ++v)
                    *reinterpret_cast<uint4*>(rec + ((long long)(tile * V + v) * 32 + smp) * 8 + 4) =
                        make_uint4((unsigned)rows[v][0], (unsigned)rows[v][1], (unsigned)rows[v][2], (unsigned)rows[v][3]);
            }
        }
        if (ok) break;
    }
}

size_t th_pixtex_bytes(int V, long long P) {
    const long long T = (P + 31) / 32;
    return (size_t)T * TX_HDR_WORDS * 4 + (size_t)T * V * 32 * 8 * 4;
}

// hdr = out, rec = hdr + T * TX_HDR_WORDS (words)
int th_pixtex_launch(int V, int H, int W, const ThPointSrc* ps, const int32_t* sel, int P, const float* cams,
                     const float* scale, void* out, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE(V >= 1 && V <= TX_MAXV && (long long)V * H * W < (1LL << 22), "texel hand-over: 1..3 views, V*H*W < 2^22 texels (4 GiB of map)");
    const int T = th_cdiv(P, 32);
    unsigned* hdr = reinterpret_cast<unsigned*>(out);
    unsigned* rec = hdr + (size_t)T * TX_HDR_WORDS;
    const dim3 grid(th_cdiv(T, 4)), block(256);
    // developer / test switch: a smaller row budget per pass sends more tiles down the 2- and 4-pass forms (same results)
    // (read per launch: a test flips it between two renders of one process)
    const char* e = getenv("TH_TEX_CAP");
    const int cv = e ? atoi(e) : TX_CAP;
    const int cap = cv >= 8 && cv <= TX_CAP ? cv : TX_CAP;
    switch (V) {
        case 1: hipLaunchKernelGGL(pixtex_kernel<1>, grid, block, 0, s, H, W, *ps, sel, P, cams, scale, hdr, rec, cap); break;
        case 2: hipLaunchKernelGGL(pixtex_kernel<2>, grid, block, 0, s, H, W, *ps, sel, P, cams, scale, hdr, rec, cap); break;
        default: hipLaunchKernelGGL(pixtex_kernel<3>, grid, block, 0, s, H, W, *ps, sel, P, cams, scale, hdr, rec, cap); break;
    }
    TH_LAUNCH_CHECK();
    return 0;
}
