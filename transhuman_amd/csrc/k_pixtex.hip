// K5t: texel hand-over of the pixel-aligned features (TH_ROWS_TEX).
//
// get_pixel_aligned_feature (if_clight_renderer.py:210-269) bilinearly samples the pixel map at the projection of every
// sample in every reference view.  K5 (k_pixfeat.hip) forms those rows and hands them to the fused MLP through HBM: 1088
// bytes written per (sample, view) and read back twice (sigma and RGB branch), 20 GB per 512 x 512 x 64 frame, plus a
// 2.4 ms kernel whose only product they are.  The 32 samples of a tile of the fused kernel are image neighbours in every
// view (depth-major sample list inside 16-ray groups), so their 3 x 128 corner texels are only ~79 DISTINCT 1 KiB texel
// rows of the map.  This kernel therefore hands over, per tile, the LIST of distinct texels (all views in one pool of rows)
// and, per (sample, view), the four row numbers + the four bilinear weights + the blended colour; the fused kernel copies the
// rows from the (L2-resident) map into LDS by LDS-DMA and blends them itself with K5's arithmetic, in K5's term order: the
// operand planes it multiplies are bit-identical to K5's rows.
//
// Output (per launch of P samples, T = ceil(P / 32) tiles):
//   hdr  [T][4 passes][128 words]   word 0 = U | npass << 16 (U = texel rows of this pass, npass = 1, 2 or 4),
//                                   words 8 .. 8 + U - 1 = global texel index (view * H * W + y * W + x) of row 0 .. U - 1
//   rec  [T][V][32 samples][8 words] {w00, w01, w10, w11} {byte offsets of the nw, ne, sw, se texel rows in the fused kernel's
//                                   row buffer: row number x 1040}
//   col  [T][V][32 samples][4 words] {r, g, b, 0}: the blended colour texels (channels 256..258 of the row), fp32
// A pass holds at most TX_CAP rows (what the fused kernel's operand buffer takes).  5.8 % of the headline frame's tiles need
// more: their samples are split into halves (2 passes) or quarters (4 passes: 8 samples x 4 corners x 3 views = 96 rows
// always fit), each with its own row list; sample s belongs to pass s / (32 / npass).
// One wave per tile; lane l and lane l + 32 both carry sample l (the upper half idles through the ballots).
#include "th_internal.h"

#define TX_CAP 103
#define TX_HDR_WORDS 512
#define TX_MAXV 3
#define TX_ROW_STRIDE 1040     // bytes per texel row in the fused kernel's row buffer (fill_tex: TSTR)

// distinct texels of corner registers c[0..3] over the lanes of `M` (a mask over the 32 samples): slot[k] = number of the
// texel of corner k in order of first appearance, starting at `base`; the texel of row `base + u` is left in lane
// (8 + base + u) of the header image (ir0: words 0..63, ir1: words 64..127).  Returns the number of distinct texels.
__device__ __forceinline__ int tx_dedup(const int (&c)[4], unsigned M, bool inM, int base, int goff, int lane, int (&slot)[4],
                                        unsigned& ir0, unsigned& ir1) {
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
        const int w = 8 + base + U;
        if (lane == w) ir0 = (unsigned)(goff + id);
        if (lane + 64 == w) ir1 = (unsigned)(goff + id);
        ++U;
    }
    return U;
}

template <int V>
__global__ __launch_bounds__(256) void pixtex_kernel(const float* __restrict__ map, int H, int W, ThPointSrc ps,
                                                     const int32_t* __restrict__ sel, int P, const float* __restrict__ cams,
                                                     const float* __restrict__ scale, unsigned* __restrict__ hdr,
                                                     unsigned* __restrict__ rec, unsigned* __restrict__ col, int cap) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pbase = tile * 32;
    if (pbase >= P) return;
    const int npts = min(32, P - pbase);
    const int smp = lane & 31;
    const int HW = H * W;
    // ---- projection, corners, weights and the colour tail of this lane's sample in every view (K5's phase 1 + tail)
    int cid[V][4];
    {
        const int p = pbase + min(smp, npts - 1);          // ragged last tile: the last sample again (rows never stored)
        const long long s = sel ? sel[p] : p;
        float x, y, z;
        th_get_point(ps, s, x, y, z);
        const float4* rgbp = reinterpret_cast<const float4*>(map + (long long)V * HW * 256);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float uu, vv;
            th_project(cams + 21 * v, x, y, z, uu, vv);
            const Bilin b = th_bilinear_setup(uu, vv, scale[0], scale[1], H, W);
            cid[v][0] = b.i00; cid[v][1] = b.i01; cid[v][2] = b.i10; cid[v][3] = b.i11;
            const float4 a = rgbp[(long long)v * HW + b.i00], bb = rgbp[(long long)v * HW + b.i01],
                         cc = rgbp[(long long)v * HW + b.i10], d = rgbp[(long long)v * HW + b.i11];
            // (pixgather_s256_kernel's colour tail: the same term order)
            const float r = fmaf(d.x, b.w11, fmaf(cc.x, b.w10, fmaf(bb.x, b.w01, a.x * b.w00)));
            const float g = fmaf(d.y, b.w11, fmaf(cc.y, b.w10, fmaf(bb.y, b.w01, a.y * b.w00)));
            const float bl = fmaf(d.z, b.w11, fmaf(cc.z, b.w10, fmaf(bb.z, b.w01, a.z * b.w00)));
            if (lane < 32) {
                unsigned* o = rec + ((long long)(tile * V + v) * 32 + smp) * 8;
                *reinterpret_cast<uint4*>(o) = make_uint4(__builtin_bit_cast(unsigned, b.w00), __builtin_bit_cast(unsigned, b.w01),
                                                          __builtin_bit_cast(unsigned, b.w10), __builtin_bit_cast(unsigned, b.w11));
                *reinterpret_cast<uint4*>(col + ((long long)(tile * V + v) * 32 + smp) * 4) =
                    make_uint4(__builtin_bit_cast(unsigned, r), __builtin_bit_cast(unsigned, g), __builtin_bit_cast(unsigned, bl), 0u);
            }
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
            unsigned ir0 = 0u, ir1 = 0u;
            int rows[V][4];
            int U = 0;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                int slot[4] = {0, 0, 0, 0};
                U += tx_dedup(cid[v], M, inM, U, v * HW, lane, slot, ir0, ir1);
#pragma unroll
                for (int j = 0; j < 4; ++j) rows[v][j] = slot[j] * TX_ROW_STRIDE;
                if (U > cap && np < 4) break;
            }
            if (U > cap && np < 4) { ok = false; break; }
            if (lane == 0) ir0 = (unsigned)U | ((unsigned)np << 16);
            hb[p * 128 + lane] = ir0;
            hb[p * 128 + 64 + lane] = ir1;
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
    return (size_t)T * TX_HDR_WORDS * 4 + (size_t)T * V * 32 * (8 + 4) * 4;
}

// hdr = out, rec = hdr + T * TX_HDR_WORDS (words), col = rec + T * V * 256; map: TH_MAP_SPLIT ([V][H*W][256] latents, then [V][H*W][4] colours)
int th_pixtex_launch(const float* map, int V, int H, int W, const ThPointSrc* ps, const int32_t* sel, int P, const float* cams,
                     const float* scale, void* out, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE(V >= 1 && V <= TX_MAXV && (long long)V * H * W < (1LL << 31), "texel hand-over: 1..3 views, V*H*W < 2^31");
    const int T = th_cdiv(P, 32);
    unsigned* hdr = reinterpret_cast<unsigned*>(out);
    unsigned* rec = hdr + (size_t)T * TX_HDR_WORDS;
    unsigned* col = rec + (size_t)T * V * 32 * 8;
    const dim3 grid(th_cdiv(T, 4)), block(256);
    // developer / test switch: a smaller row budget per pass sends more tiles down the 2- and 4-pass forms (same results)
    // (read per launch: a test flips it between two renders of one process)
    const char* e = getenv("TH_TEX_CAP");
    const int cv = e ? atoi(e) : TX_CAP;
    const int cap = cv >= 8 && cv <= TX_CAP ? cv : TX_CAP;
    switch (V) {
        case 1: hipLaunchKernelGGL(pixtex_kernel<1>, grid, block, 0, s, map, H, W, *ps, sel, P, cams, scale, hdr, rec, col, cap); break;
        case 2: hipLaunchKernelGGL(pixtex_kernel<2>, grid, block, 0, s, map, H, W, *ps, sel, P, cams, scale, hdr, rec, col, cap); break;
        default: hipLaunchKernelGGL(pixtex_kernel<3>, grid, block, 0, s, map, H, W, *ps, sel, P, cams, scale, hdr, rec, col, cap); break;
    }
    TH_LAUNCH_CHECK();
    return 0;
}
