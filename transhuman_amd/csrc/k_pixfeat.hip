// K5: pixel-aligned feature gather.
//
// get_pixel_aligned_feature (if_clight_renderer.py:210-269): project a world
// point into the V reference cameras and bilinearly sample the 384-channel
// pixel_feat_map (grid_sample, align_corners=True, border).  The reference
// samples an NCHW map for *all* chunk points (384 strided cache lines per
// corner) and masks afterwards (cross_transformer.py:235); here the map is
// channels-last (th_nchw_to_nhwc, once per frame) so each corner is one
// contiguous 1.5 KB read, and only hull-valid samples are gathered.
// One wave per 16 consecutive samples of one view; split rows: a half-wave per row, two float4 per lane and corner
// (pixgather_kernel); fp32 rows: a wave per row, one float4 per lane and corner (pixgather_f32_kernel).
// The map has C channels per pixel (384 full / 260 compact, see k_encoder.hip); output rows are ldo floats
// wide (>= C; the tail is zero-filled so the row can feed a K-padded GEMM directly: compact rows are 272).
// Bound: L2/HBM gather, 4 * 4C B per (sample, view) in, 4*ldo B out.
#include <stdlib.h>

#include "th_internal.h"

#define PG_G 16
#define PG_B 4      // rows per batch: 4 PG_B corner loads in flight per wave
typedef _Float16 pg_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pg_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// SPLIT: rows are written as ldo fp16 hi halves followed by ldo fp16 lo halves (TH_ROWS_SPLIT) instead of ldo floats
// range guard of the split rows (see store_tile_h in k_mlp_fused_kernel.h): running maximum of the |hi| halves as
// 15-bit integers (inf / NaN order above every finite value), one v_and + v_pk_max_u16 per pair of values
typedef unsigned short pg_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pg_range_acc(unsigned& rm, unsigned hi2) {
    unsigned a = hi2 & 0x7fff7fffu;
    pg_us2 m = __builtin_elementwise_max(*reinterpret_cast<pg_us2*>(&rm), *reinterpret_cast<pg_us2*>(&a));
    rm = *reinterpret_cast<unsigned*>(&m);
}
__device__ __forceinline__ void pg_range_commit(unsigned* __restrict__ table, unsigned rm) {
    const unsigned m = max(rm & 0xffffu, rm >> 16);
    if (table != nullptr && m > table[TH_RANGE_F]) atomicMax(table + TH_RANGE_F, m);
}

// SPLIT rows (TH_ROWS_SPLIT): groups of 8 channels, each 32 bytes = [8 hi halves | 8 lo halves].  A 16-byte piece is
// then one plane's 8 consecutive halves -- what the fused kernel's LDS-DMA staging moves and its MFMA fragments read --
// and a lane that owns 8 channels writes its 32 bytes with two full-width store instructions per ROW PAIR.  (Two 512-byte
// plane stores per row were what bounded this kernel: 2.67 ms per frame, 2.03 ms without the second store, see DESIGN.md.)
template <bool SPLIT>
__device__ __forceinline__ void pg_store8(float* __restrict__ orow, int g, float4 a, float4 bq, unsigned& rm) {
    if (!SPLIT) {
        reinterpret_cast<float4*>(orow)[2 * g] = a;
        reinterpret_cast<float4*>(orow)[2 * g + 1] = bq;
    } else {
        const float v[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        typedef _Float16 pg_h8 __attribute__((ext_vector_type(8)));
        pg_h8 hv, lv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 x, y;
            pg_split(v[e], x, y);
            hv[e] = x;
            lv[e] = y;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pg_range_acc(rm, reinterpret_cast<const unsigned*>(&hv)[e]);
        _Float16* og = reinterpret_cast<_Float16*>(orow) + 16 * g;
        *reinterpret_cast<pg_h8*>(og) = hv;
        *reinterpret_cast<pg_h8*>(og + 8) = lv;
    }
}
// Contiguous form of pg_store8 for the latents of a split row: lane gl of a half-wave holds channels 4 gl .. 4 gl + 3 (a) and
// 128 + 4 gl .. + 3 (b).  Neighbouring lanes (2k, 2k+1) trade halves with one DPP swap of four dwords, after which lane 2k holds
// the 8 hi halves and lane 2k+1 the 8 lo halves of channel group k (a) and of group 16 + k (b): piece gl of the row's first 512
// bytes and piece gl of its second -- two store instructions that each write 512 contiguous bytes per half-wave.
// Every lane of the wave must call this (DPP reads the neighbour); `live` masks the stores.
#define PG_CONTIG 1
__device__ __forceinline__ unsigned pg_swap1(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void pg_store_pair(float* __restrict__ orow, int gl, float4 a, float4 bq, unsigned& rm, bool live) {
    typedef _Float16 pg_h2 __attribute__((ext_vector_type(2)));
    const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {bq.x, bq.y, bq.z, bq.w};
    unsigned ha[2], la[2], hb[2], lb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        _Float16 x0, y0, x1, y1;
        pg_h2 t;
        pg_split(va[2 * e], x0, y0); pg_split(va[2 * e + 1], x1, y1);
        t[0] = x0; t[1] = x1; ha[e] = __builtin_bit_cast(unsigned, t);
        t[0] = y0; t[1] = y1; la[e] = __builtin_bit_cast(unsigned, t);
        pg_split(vb[2 * e], x0, y0); pg_split(vb[2 * e + 1], x1, y1);
        t[0] = x0; t[1] = x1; hb[e] = __builtin_bit_cast(unsigned, t);
        t[0] = y0; t[1] = y1; lb[e] = __builtin_bit_cast(unsigned, t);
        pg_range_acc(rm, ha[e]);
        pg_range_acc(rm, hb[e]);
    }
    const bool odd = gl & 1;
    // even lanes send their lo halves and receive the neighbour's hi halves; odd lanes the other way round
    unsigned ra[2], rb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        ra[e] = pg_swap1(odd ? ha[e] : la[e]);
        rb[e] = pg_swap1(odd ? hb[e] : lb[e]);
    }
    uint4 s0, s1;
    s0.x = odd ? ra[0] : ha[0]; s0.y = odd ? ra[1] : ha[1]; s0.z = odd ? la[0] : ra[0]; s0.w = odd ? la[1] : ra[1];
    s1.x = odd ? rb[0] : hb[0]; s1.y = odd ? rb[1] : hb[1]; s1.z = odd ? lb[0] : rb[0]; s1.w = odd ? lb[1] : rb[1];
    if (live) {
        uint4* o = reinterpret_cast<uint4*>(orow);
        o[gl] = s0;
        o[32 + gl] = s1;
    }
}

// bilinear blend of four corner texels (grid_sample's term order; fused multiply-adds: the reference kernel is
// compiled with FMA contraction too, and the parity bar for this stage is 2e-5)
__device__ __forceinline__ float4 pg_blend(float4 a, float4 bb, float4 cc, float4 d, float w00, float w01, float w10,
                                           float w11) {
    float4 r;
    r.x = fmaf(d.x, w11, fmaf(cc.x, w10, fmaf(bb.x, w01, a.x * w00)));
    r.y = fmaf(d.y, w11, fmaf(cc.y, w10, fmaf(bb.y, w01, a.y * w00)));
    r.z = fmaf(d.z, w11, fmaf(cc.z, w10, fmaf(bb.z, w01, a.z * w00)));
    r.w = fmaf(d.w, w11, fmaf(cc.w, w10, fmaf(bb.w, w01, a.w * w00)));
    return r;
}

template <bool SPLIT>
__device__ __forceinline__ void pg_store4(float* __restrict__ orow, int ldo, int c4, float4 r, unsigned& rm) {
    if (!SPLIT) reinterpret_cast<float4*>(orow)[c4] = r;
    else {
        _Float16* oh = reinterpret_cast<_Float16*>(orow);
        pg_h4 hv, lv;
        _Float16 x, y;
        pg_split(r.x, x, y); hv[0] = x; lv[0] = y;
        pg_split(r.y, x, y); hv[1] = x; lv[1] = y;
        pg_split(r.z, x, y); hv[2] = x; lv[2] = y;
        pg_split(r.w, x, y); hv[3] = x; lv[3] = y;
        pg_range_acc(rm, reinterpret_cast<const unsigned*>(&hv)[0]);
        pg_range_acc(rm, reinterpret_cast<const unsigned*>(&hv)[1]);
        *reinterpret_cast<pg_h4*>(oh + 4 * c4) = hv;
        *reinterpret_cast<pg_h4*>(oh + ldo + 4 * c4) = lv;
    }
}

// fp32 rows (TH_ROWS_F32: the per-layer MLP path, paint_neural_human): one row per wave pass, lanes = float4 columns,
// any row stride that is a multiple of 4
template <bool SPLIT>
__global__ __launch_bounds__(256) void pixgather_f32_kernel(const float* __restrict__ map, int V, int C, int H, int W,
                                                        const float* __restrict__ pts_world, ThPointSrc ps,
                                                        const int32_t* __restrict__ sel, int P,
                                                        const float* __restrict__ cams,
                                                        const float* __restrict__ scale, float* __restrict__ out,
                                                        int ldo, unsigned* __restrict__ range) {
    const int lane = threadIdx.x & 63;
    unsigned rm = 0u;
    // XCD-aware remap (speed only): workgroup b runs on XCD b % 8, each XCD has its own L2: give every XCD a
    // CONTIGUOUS range of groups: logical block = (b % 8) * ceil(nb / 8) + b / 8 (bijective incl. ragged tails
    // via the bounds check below).
    const long long nb8 = ((long long)gridDim.x + 7) / 8;
    const long long lb = (long long)(blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
    const long long grp = lb * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (sample group, view): scalar
    const long long ngrp = (long long)((P + PG_G - 1) / PG_G) * V;
    if (grp >= ngrp) return;
    const int v = (int)(grp % V);
    const int p0 = (int)(grp / V) * PG_G;
    const int nrow = min(PG_G, P - p0);
    const float* m = map + (long long)v * H * W * C;
    // split layout (C == 256): the r, g, b, 0 texels live in a [V,H,W,4] plane behind the latents; they are output
    // column 64 (float4) of a row exactly like channels 256..259 of the interleaved 260-channel map
    const float4* rgbp = C == 256 ? reinterpret_cast<const float4*>(map + (long long)V * H * W * 256) + (long long)v * H * W : nullptr;
    const int C4 = C / 4, L4 = ldo / 4;

    // ---- phase 1 ----
    Bilin b;
    {
        const int p = p0 + min(lane, nrow - 1);
        long long s = sel ? sel[p] : p;
        float x, y, z;
        if (pts_world) { x = pts_world[3 * s]; y = pts_world[3 * s + 1]; z = pts_world[3 * s + 2]; }
        else th_get_point(ps, s, x, y, z);
        float uu, vv;
        th_project(cams + 21 * v, x, y, z, uu, vv);
        b = th_bilinear_setup(uu, vv, scale[0], scale[1], H, W);
    }
    // ---- phase 2 ----
    for (int r0 = 0; r0 < nrow; r0 += PG_B) {
        float4 q[PG_B][4];
        float w[PG_B][4];
#pragma unroll
        for (int j = 0; j < PG_B; ++j) {
            const int i = min(r0 + j, nrow - 1);                           // ragged tail: duplicate loads, no store
            const int i00 = __builtin_amdgcn_readlane(b.i00, i), i01 = __builtin_amdgcn_readlane(b.i01, i);
            const int i10 = __builtin_amdgcn_readlane(b.i10, i), i11 = __builtin_amdgcn_readlane(b.i11, i);
            w[j][0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b.w00), i));
            w[j][1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b.w01), i));
            w[j][2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b.w10), i));
            w[j][3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b.w11), i));
            if (lane < C4) {
                q[j][0] = reinterpret_cast<const float4*>(m + (long long)i00 * C)[lane];
                q[j][1] = reinterpret_cast<const float4*>(m + (long long)i01 * C)[lane];
                q[j][2] = reinterpret_cast<const float4*>(m + (long long)i10 * C)[lane];
                q[j][3] = reinterpret_cast<const float4*>(m + (long long)i11 * C)[lane];
            }
        }
        // columns >= 256 of the 4 rows: lane = 4*t + j handles float4 column 64 + t of row r0 + j
        if (C4 <= 65 && L4 <= 68 && L4 > 64) {
            const int j = lane & (PG_B - 1), t = lane / PG_B, i = min(r0 + j, nrow - 1);
            const int c4 = 64 + t;
            // this lane's row parameters live in lane i (phase 1): fetch them across lanes
            const int i00 = __shfl(b.i00, i), i01 = __shfl(b.i01, i), i10 = __shfl(b.i10, i), i11 = __shfl(b.i11, i);
            const float w00 = __shfl(b.w00, i), w01 = __shfl(b.w01, i), w10 = __shfl(b.w10, i), w11 = __shfl(b.w11, i);
            if (lane < 4 * PG_B && c4 < L4 && r0 + j < nrow) {
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rgbp != nullptr && c4 == 64)
                    r = pg_blend(rgbp[i00], rgbp[i01], rgbp[i10], rgbp[i11], w00, w01, w10, w11);
                else if (c4 < C4)
                    r = pg_blend(reinterpret_cast<const float4*>(m + (long long)i00 * C)[c4],
                                 reinterpret_cast<const float4*>(m + (long long)i01 * C)[c4],
                                 reinterpret_cast<const float4*>(m + (long long)i10 * C)[c4],
                                 reinterpret_cast<const float4*>(m + (long long)i11 * C)[c4], w00, w01, w10, w11);
                pg_store4<SPLIT>(out + ((long long)(p0 + r0 + j) * V + v) * ldo, ldo, c4, r, rm);
            }
        }
#pragma unroll
        for (int j = 0; j < PG_B; ++j) {
            if (r0 + j >= nrow) break;
            float* orow = out + ((long long)(p0 + r0 + j) * V + v) * ldo;
            if (lane < C4)
                pg_store4<SPLIT>(orow, ldo, lane, pg_blend(q[j][0], q[j][1], q[j][2], q[j][3], w[j][0], w[j][1], w[j][2], w[j][3]), rm);
            if (!(C4 <= 65 && L4 <= 68) && L4 > 64) {
                // wide maps (full 384-channel map): remaining columns row by row
                const int i = r0 + j;
                const int i00 = __builtin_amdgcn_readlane(b.i00, i), i01 = __builtin_amdgcn_readlane(b.i01, i);
                const int i10 = __builtin_amdgcn_readlane(b.i10, i), i11 = __builtin_amdgcn_readlane(b.i11, i);
                for (int c4 = 64 + lane; c4 < L4; c4 += 64) {
                    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c4 < C4)
                        r = pg_blend(reinterpret_cast<const float4*>(m + (long long)i00 * C)[c4],
                                     reinterpret_cast<const float4*>(m + (long long)i01 * C)[c4],
                                     reinterpret_cast<const float4*>(m + (long long)i10 * C)[c4],
                                     reinterpret_cast<const float4*>(m + (long long)i11 * C)[c4], w[j][0], w[j][1], w[j][2],
                                     w[j][3]);
                    pg_store4<SPLIT>(orow, ldo, c4, r, rm);
                }
            }
        }
    }
    if (SPLIT) pg_range_commit(range, rm);
}


// One wave = PG_G consecutive samples of ONE view (their projections are neighbours in that view's map).
//  phase 1: lane i < PG_G projects sample i and derives its four corner indices + weights -- the per-row setup
//           (two divisions, floor/clamp, ~140 instructions) runs once per PG_G rows instead of once per row in
//           every lane (PMC: the row-per-wave form spent 78 % of its time in VALU issue, 325 instructions/row);
//  phase 2: rows in batches of 4: the corner indices / weights come back as wave-uniform scalars
//           (v_readlane), so the corner addresses are scalar arithmetic and the 16 corner loads of a batch
//           (4 KiB each row) are all issued before the first blend.  Lanes span channels (float4 per lane:
//           64 lanes = the 256 latent channels); columns >= 256 (compact map: r g b 0 + zero tail of the row)
//           are produced for the 4 rows of a batch at once by lanes 0..15.
template <bool SPLIT>
__global__ __launch_bounds__(256) void pixgather_kernel(const float* __restrict__ map, int V, int C, int H, int W,
                                                        const float* __restrict__ pts_world, ThPointSrc ps,
                                                        const int32_t* __restrict__ sel, int P,
                                                        const float* __restrict__ cams,
                                                        const float* __restrict__ scale, float* __restrict__ out,
                                                        int ldo, unsigned* __restrict__ range) {
    const int lane = threadIdx.x & 63;
    unsigned rm = 0u;
    // XCD-aware remap (speed only): workgroup b runs on XCD b % 8, each XCD has its own L2: give every XCD a
    // CONTIGUOUS range of groups: logical block = (b % 8) * ceil(nb / 8) + b / 8 (bijective incl. ragged tails
    // via the bounds check below).
    const long long nb8 = ((long long)gridDim.x + 7) / 8;
    const long long lb = (long long)(blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
    const long long grp = lb * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (sample group, view): scalar
    const long long ngrp = (long long)((P + PG_G - 1) / PG_G) * V;
    if (grp >= ngrp) return;
    const int v = (int)(grp % V);
    const int p0 = (int)(grp / V) * PG_G;
    const int nrow = min(PG_G, P - p0);
    const float* m = map + (long long)v * H * W * C;
    // split layout (C == 256): the r, g, b, 0 texels live in a [V,H,W,4] plane behind the latents; they are output
    // column 64 (float4) of a row exactly like channels 256..259 of the interleaved 260-channel map
    const float4* rgbp = C == 256 ? reinterpret_cast<const float4*>(map + (long long)V * H * W * 256) + (long long)v * H * W : nullptr;
    const int C4 = C / 4, L4 = ldo / 4;

    // ---- phase 1 ----
    Bilin b;
    {
        const int p = p0 + min(lane, nrow - 1);
        long long s = sel ? sel[p] : p;
        float x, y, z;
        if (pts_world) { x = pts_world[3 * s]; y = pts_world[3 * s + 1]; z = pts_world[3 * s + 2]; }
        else th_get_point(ps, s, x, y, z);
        float uu, vv;
        th_project(cams + 21 * v, x, y, z, uu, vv);
        b = th_bilinear_setup(uu, vv, scale[0], scale[1], H, W);
    }
    // ---- phase 2 ----
    // Rows in batches of PG_B / 2 row pairs.  Lanes 0..31 work on the first row of a pair, lanes 32..63 on the second;
    // gl = lane & 31 owns two float4 per corner (channels 4 gl .. and 128 + 4 gl ..: every load instruction reads 512
    // contiguous bytes per half-wave; pg_store_pair regroups them into the [8 hi | 8 lo] groups of a split row): the corner
    // loads of a batch are all requested before the first blend, the corner indices / weights of the two rows come back as
    // scalars (v_readlane) and are selected per half-wave.
    const int half = lane >> 5, gl = lane & 31;
    const int G = ldo / 8, TG = G - 32;                 // 8-channel groups per output row; groups beyond the 256 latents
    for (int r0 = 0; r0 < nrow; r0 += PG_B) {
        float4 q[PG_B / 2][4][2];
        float w[PG_B / 2][4];
#pragma unroll
        for (int jj = 0; jj < PG_B / 2; ++jj) {
            const int iA = min(r0 + 2 * jj, nrow - 1), iB = min(r0 + 2 * jj + 1, nrow - 1);   // ragged tail: duplicate loads, no store
            const int a00 = __builtin_amdgcn_readlane(b.i00, iA), a01 = __builtin_amdgcn_readlane(b.i01, iA);
            const int a10 = __builtin_amdgcn_readlane(b.i10, iA), a11 = __builtin_amdgcn_readlane(b.i11, iA);
            const int b00 = __builtin_amdgcn_readlane(b.i00, iB), b01 = __builtin_amdgcn_readlane(b.i01, iB);
            const int b10 = __builtin_amdgcn_readlane(b.i10, iB), b11 = __builtin_amdgcn_readlane(b.i11, iB);
            const float* wsrc[4] = {&b.w00, &b.w01, &b.w10, &b.w11};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float wa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, *wsrc[c]), iA));
                const float wb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, *wsrc[c]), iB));
                w[jj][c] = half ? wb : wa;
            }
            const int ic[4] = {half ? b00 : a00, half ? b01 : a01, half ? b10 : a10, half ? b11 : a11};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#if PG_CONTIG
                // each half-wave reads 512 contiguous bytes per instruction (channels 4 gl.. and 128 + 4 gl..): 8 L1
                // lines per wave load instead of 16 half-used ones
                const float4* src = reinterpret_cast<const float4*>(m + (long long)ic[c] * C) + gl;
                q[jj][c][0] = src[0];
                q[jj][c][1] = src[32];
#else
                const float4* src = reinterpret_cast<const float4*>(m + (long long)ic[c] * C) + 2 * gl;
                q[jj][c][0] = src[0];
                q[jj][c][1] = src[1];
#endif
            }
        }
        // groups beyond the latents (compact rows: r g b 0 | zeros; full rows: the 128 colour-lift channels) for the rows of
        // the batch at once: lane = row * TG + group
        if (TG > 0) {
            const int j = min(lane / TG, PG_B - 1), t = lane % TG, i = min(r0 + j, nrow - 1);
            // this lane's row parameters live in lane i (phase 1): fetch them across lanes (every lane takes part: a
            // ds_bpermute reads nothing useful from a lane that is masked off)
            const int i00 = __shfl(b.i00, i), i01 = __shfl(b.i01, i), i10 = __shfl(b.i10, i), i11 = __shfl(b.i11, i);
            const float w00 = __shfl(b.w00, i), w01 = __shfl(b.w01, i), w10 = __shfl(b.w10, i), w11 = __shfl(b.w11, i);
            if (lane < PG_B * TG) {
            float4 r2[2];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const int c4 = 2 * (32 + t) + hq;
                float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c4 < C4)
                    r = pg_blend(reinterpret_cast<const float4*>(m + (long long)i00 * C)[c4],
                                 reinterpret_cast<const float4*>(m + (long long)i01 * C)[c4],
                                 reinterpret_cast<const float4*>(m + (long long)i10 * C)[c4],
                                 reinterpret_cast<const float4*>(m + (long long)i11 * C)[c4], w00, w01, w10, w11);
                else if (rgbp != nullptr && c4 == 64)
                    r = pg_blend(rgbp[i00], rgbp[i01], rgbp[i10], rgbp[i11], w00, w01, w10, w11);
                r2[hq] = r;
            }
            if (r0 + j < nrow) pg_store8<SPLIT>(out + ((long long)(p0 + r0 + j) * V + v) * ldo, 32 + t, r2[0], r2[1], rm);
            }
        }
#pragma unroll
        for (int jj = 0; jj < PG_B / 2; ++jj) {
            const int i = r0 + 2 * jj + half;
            const float4 ra = pg_blend(q[jj][0][0], q[jj][1][0], q[jj][2][0], q[jj][3][0], w[jj][0], w[jj][1], w[jj][2], w[jj][3]);
            const float4 rb = pg_blend(q[jj][0][1], q[jj][1][1], q[jj][2][1], q[jj][3][1], w[jj][0], w[jj][1], w[jj][2], w[jj][3]);
#if PG_CONTIG
            pg_store_pair(out + ((long long)(p0 + min(i, nrow - 1)) * V + v) * ldo, gl, ra, rb, rm, i < nrow);
#else
            if (i < nrow) pg_store8<SPLIT>(out + ((long long)(p0 + i) * V + v) * ldo, gl, ra, rb, rm);
#endif
        }
    }
    if (SPLIT) pg_range_commit(range, rm);
}


// ---- the frame-level form: split map (C = 256 latents + the [V,H,W,4] colour plane) -> 272-wide split rows -------------------
// Same rows, bit for bit, as pixgather_kernel<true> on that input; written after measuring what the launch costs when every
// corner load hits L1 (tools/k5_locality.py: 1.72 ms of the frame's 2.5 ms -- instruction issue, not memory): 365 VALU
// instructions and 20 + 5 vector-memory instructions per batch of 4 rows there, a third of them bookkeeping --
//  * the two rows of a pair need DIFFERENT corner indices / weights in the two half-waves: v_readlane x 2 + v_mov +
//    v_cndmask per value and 64-bit vector address arithmetic per load.  Here phase 1 runs on every lane (lane l sets up row
//    l >> 2: same instruction count as on 16 lanes), a half-wave fetches its row's eight values with ds_bpermute_b32 (the LDS
//    crossbar, no LDS memory) and the corner loads take a scalar base (the view's map) + a 32-bit lane offset;
//  * the colour texels (channels 256..258) cost 4 load instructions + a blend + a store per BATCH for 16 bytes per row: with
//    lane = (row, corner) the whole wave's 64 colour texels are ONE load instruction, the blend runs inside the quad
//    (quad_perm broadcasts, same term order) and the 64-byte row tails (r g b 0 | zeros, hi | lo) are one store instruction;
//  * blend on packed fp32 FMAs, hi / lo split by v_fma_mix (split_pair of k_mlp_fused_kernel.h: bit-identical to pg_split).
typedef float pg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pg_bperm(int byte_addr, int x) { return __builtin_amdgcn_ds_bpermute(byte_addr, x); }
__device__ __forceinline__ float pg_bpermf(int byte_addr, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, x)));
}
template <int CTRL>
__device__ __forceinline__ float pg_quad(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// two values -> packed hi pair + packed lo pair (lo = fp16(x - fp32(hi)); x - hi is exact in fp32: same bits as pg_split)
__device__ __forceinline__ void pg_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    typedef _Float16 pg_h2 __attribute__((ext_vector_type(2)));
    pg_h2 h;
    h[0] = (_Float16)x0;
    h[1] = (_Float16)x1;
    hi = __builtin_bit_cast(unsigned, h);
    asm("" : "+v"(hi));
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    lo = l;
}
__device__ __forceinline__ float4 pg_blend2(float4 a, float4 bb, float4 cc, float4 d, float w00, float w01, float w10, float w11) {
    const pg_f2 W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
    pg_f2 lo = (pg_f2){a.x, a.y} * W00, hi = (pg_f2){a.z, a.w} * W00;
    lo = __builtin_elementwise_fma((pg_f2){bb.x, bb.y}, W01, lo);
    hi = __builtin_elementwise_fma((pg_f2){bb.z, bb.w}, W01, hi);
    lo = __builtin_elementwise_fma((pg_f2){cc.x, cc.y}, W10, lo);
    hi = __builtin_elementwise_fma((pg_f2){cc.z, cc.w}, W10, hi);
    lo = __builtin_elementwise_fma((pg_f2){d.x, d.y}, W11, lo);
    hi = __builtin_elementwise_fma((pg_f2){d.z, d.w}, W11, hi);
    return make_float4(lo[0], lo[1], hi[0], hi[1]);
}

__global__ __launch_bounds__(256) void pixgather_s256_kernel(const float* __restrict__ map, int V, int H, int W,
                                                             const float* __restrict__ pts_world, ThPointSrc ps,
                                                             const int32_t* __restrict__ sel, int P,
                                                             const float* __restrict__ cams, const float* __restrict__ scale,
                                                             float* __restrict__ out, unsigned* __restrict__ range) {
    constexpr int LDO = 272;                        // floats per output row (1088 bytes: 34 groups of [8 hi | 8 lo] halves)
    const int lane = threadIdx.x & 63;
    unsigned rm = 0u;
    const long long nb8 = ((long long)gridDim.x + 7) / 8;                   // XCD-aware remap: see pixgather_kernel
    const long long lb = (long long)(blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
    // the four waves of a workgroup take four CONSECUTIVE 16-sample groups of ONE view (adjacent depths of the same 16 rays in
    // the depth-major list: their footprints in that view overlap), not one group in V views
    const int v = (int)(lb % V);
    const long long g16 = (lb / V) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (g16 >= (P + PG_G - 1) / PG_G) return;
    const int p0 = (int)g16 * PG_G;
    const int nrow = min(PG_G, P - p0);
    const char* mb = reinterpret_cast<const char*>(map + (long long)v * H * W * 256);             // wave-uniform bases
    const float4* rgbp = reinterpret_cast<const float4*>(map + (long long)V * H * W * 256) + (long long)v * H * W;
    char* ob = reinterpret_cast<char*>(out + ((long long)p0 * V + v) * LDO);

    // ---- phase 1: lane l sets up row l >> 2 (the four lanes of a quad hold the same row: the colour tail below uses them as
    //      the row's four corners; the row pairs fetch their values from lane 4 * row)
    Bilin b;
    {
        const int p = p0 + min(lane >> 2, nrow - 1);
        long long s = sel ? sel[p] : p;
        float x, y, z;
        if (pts_world) { x = pts_world[3 * s]; y = pts_world[3 * s + 1]; z = pts_world[3 * s + 2]; }
        else th_get_point(ps, s, x, y, z);
        float uu, vv;
        th_project(cams + 21 * v, x, y, z, uu, vv);
        b = th_bilinear_setup(uu, vv, scale[0], scale[1], H, W);
    }
    // ---- colour tail of the 16 rows: lane = (row, corner)
    float4 tq;
    float tw[4] = {b.w00, b.w01, b.w10, b.w11};
    {
        const int c = lane & 3;
        const int ti = c == 0 ? b.i00 : c == 1 ? b.i01 : c == 2 ? b.i10 : b.i11;
        tq = rgbp[ti];
    }
    const int half = lane >> 5, gl = lane & 31;
    const unsigned o00 = (unsigned)b.i00 << 10, o01 = (unsigned)b.i01 << 10, o10 = (unsigned)b.i10 << 10, o11 = (unsigned)b.i11 << 10;
    for (int r0 = 0; r0 < nrow; r0 += PG_B) {
        float4 q[PG_B / 2][4][2];
        float w[PG_B / 2][4];
#pragma unroll
        for (int jj = 0; jj < PG_B / 2; ++jj) {
            const int i = min(r0 + 2 * jj + half, nrow - 1);              // ragged tail: duplicate loads, no store
            const int src = i << 4;                                        // byte address of lane 4 i for ds_bpermute
            const unsigned oc[4] = {(unsigned)pg_bperm(src, (int)o00), (unsigned)pg_bperm(src, (int)o01),
                                    (unsigned)pg_bperm(src, (int)o10), (unsigned)pg_bperm(src, (int)o11)};
            w[jj][0] = pg_bpermf(src, b.w00); w[jj][1] = pg_bpermf(src, b.w01);
            w[jj][2] = pg_bpermf(src, b.w10); w[jj][3] = pg_bpermf(src, b.w11);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // each half-wave reads 512 contiguous bytes per instruction (channels 4 gl .. and 128 + 4 gl ..)
                const float4* sp = reinterpret_cast<const float4*>(mb + (size_t)(oc[c] + (unsigned)gl * 16u));
                q[jj][c][0] = sp[0];
                q[jj][c][1] = sp[32];
            }
        }
#pragma unroll
        for (int jj = 0; jj < PG_B / 2; ++jj) {
            const int i = r0 + 2 * jj + half;
            const float4 ra = pg_blend2(q[jj][0][0], q[jj][1][0], q[jj][2][0], q[jj][3][0], w[jj][0], w[jj][1], w[jj][2], w[jj][3]);
            const float4 rb = pg_blend2(q[jj][0][1], q[jj][1][1], q[jj][2][1], q[jj][3][1], w[jj][0], w[jj][1], w[jj][2], w[jj][3]);
            unsigned ha[2], la[2], hb[2], lb2[2];
            pg_split2(ra.x, ra.y, ha[0], la[0]); pg_split2(ra.z, ra.w, ha[1], la[1]);
            pg_split2(rb.x, rb.y, hb[0], lb2[0]); pg_split2(rb.z, rb.w, hb[1], lb2[1]);
            pg_range_acc(rm, ha[0]); pg_range_acc(rm, hb[0]); pg_range_acc(rm, ha[1]); pg_range_acc(rm, hb[1]);
            // lanes (2k, 2k+1) trade halves (see pg_store_pair): lane 2k ends with the 8 hi halves, lane 2k+1 with the 8 lo halves
            const bool odd = gl & 1;
            unsigned xa[2], xb[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                xa[e] = pg_swap1(odd ? ha[e] : la[e]);
                xb[e] = pg_swap1(odd ? hb[e] : lb2[e]);
            }
            uint4 s0, s1;
            s0.x = odd ? xa[0] : ha[0]; s0.y = odd ? xa[1] : ha[1]; s0.z = odd ? la[0] : xa[0]; s0.w = odd ? la[1] : xa[1];
            s1.x = odd ? xb[0] : hb[0]; s1.y = odd ? xb[1] : hb[1]; s1.z = odd ? lb2[0] : xb[0]; s1.w = odd ? lb2[1] : xb[1];
            if (i < nrow) {
                uint4* o = reinterpret_cast<uint4*>(ob + (size_t)((unsigned)i * (unsigned)(V * LDO * 4) + (unsigned)gl * 16u));
                o[0] = s0;
                o[32] = s1;
            }
        }
    }
    // ---- colour tail: blend inside the quad (corner values by quad_perm broadcast; pg_blend's term order), then lane c of a quad
    //      writes 16-byte piece c of the row's last 64 bytes: hi halves of (r g b 0 0 0 0 0), their lo halves, zeros, zeros
    {
        float r3[4];
        const float tv[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = pg_quad<0x00>(tv[e]), bb = pg_quad<0x55>(tv[e]), cc = pg_quad<0xAA>(tv[e]), d = pg_quad<0xFF>(tv[e]);
            r3[e] = fmaf(d, tw[3], fmaf(cc, tw[2], fmaf(bb, tw[1], a * tw[0])));
        }
        unsigned h01, l01, h23, l23;
        pg_split2(r3[0], r3[1], h01, l01);
        pg_split2(r3[2], r3[3], h23, l23);
        pg_range_acc(rm, h01);
        pg_range_acc(rm, h23);
        const int c = lane & 3, row = lane >> 2;
        uint4 t = make_uint4(0u, 0u, 0u, 0u);
        if (c == 0) { t.x = h01; t.y = h23; }
        if (c == 1) { t.x = l01; t.y = l23; }
        if (row < nrow)
            *reinterpret_cast<uint4*>(ob + (size_t)((unsigned)row * (unsigned)(V * LDO * 4) + 1024u + (unsigned)c * 16u)) = t;
    }
    pg_range_commit(range, rm);
}

int th_pixgather_launch(const float* map, int V, int C, int H, int W, const float* pts_world, const ThPointSrc* ps,
                        const int32_t* sel, int P, const float* cams, const float* scale, float* out, int ldo,
                        int fmt, hipStream_t s, unsigned int* range) {
    if (P <= 0) return 0;
    TH_REQUIRE((C & 3) == 0 && (ldo & 3) == 0 && ldo >= C, "channel count / row stride must be multiples of 4, ldo >= C");
    TH_REQUIRE(fmt != TH_ROWS_SPLIT || ((ldo & 7) == 0 && C >= 256 && ldo <= 256 + 8 * (64 / PG_B)),
               "split rows: >= 256 map channels, row stride a multiple of 8 with at most 16 groups beyond the latents");
    ThPointSrc src = ps ? *ps : ThPointSrc{};
    const long long groups = (long long)th_cdiv(P, PG_G) * V;
    const int nblk = 8 * th_cdiv(th_cdiv(groups, 4), 8);     // multiple of 8 so the XCD remap is onto
    const char* k5g = getenv("TH_K5_GENERIC");                       // developer / test switch: the generic kernel on the same input
    const bool generic = k5g != nullptr && atoi(k5g) != 0;
    const int nblk4 = 8 * th_cdiv((long long)th_cdiv(th_cdiv(P, PG_G), 4) * V, 8);      // workgroups of 4 groups x 1 view
    if (fmt == TH_ROWS_SPLIT && C == 256 && ldo == 272 && (long long)H * W <= (1 << 21) && !generic)
        hipLaunchKernelGGL(pixgather_s256_kernel, dim3(nblk4), dim3(256), 0, s, map, V, H, W, pts_world, src, sel, P, cams, scale,
                           out, range);
    else if (fmt == TH_ROWS_SPLIT)
        hipLaunchKernelGGL(pixgather_kernel<true>, dim3(nblk), dim3(256), 0, s, map, V, C, H, W, pts_world, src, sel, P,
                           cams, scale, out, ldo, range);
    else
        hipLaunchKernelGGL(pixgather_f32_kernel<false>, dim3(nblk), dim3(256), 0, s, map, V, C, H, W, pts_world, src, sel, P,
                           cams, scale, out, ldo, nullptr);
    TH_LAUNCH_CHECK();
    return 0;
}

// Network.forward drop-in path: the caller already sampled pixel_feat as
// [V, C, Pall] (channel-major, cross_transformer.py:235 masks it).  Gather the
// selected points and transpose to rows [P][V][C] through LDS tiles.
template <bool SPLIT>
__global__ __launch_bounds__(256) void gather_chan_major_kernel(const float* __restrict__ pf, int V, int C,
                                                                long long Pall, const int32_t* __restrict__ sel,
                                                                int P, float* __restrict__ out,
                                                                unsigned* __restrict__ range) {
    __shared__ float tile[32][33];
    unsigned rm = 0u;
    int v = blockIdx.z;
    int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    int p = p0 + tx;
    long long q = (p < P) ? (sel ? (long long)sel[p] : (long long)p) : -1;
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r;
        tile[r][tx] = (q >= 0 && c < C) ? pf[((long long)v * C + c) * Pall + q] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int pp = p0 + r, c = c0 + tx;
        if (pp < P && c < C) {
            if (!SPLIT) out[((long long)pp * V + v) * C + c] = tile[tx][r];
            else {
                _Float16* oh = reinterpret_cast<_Float16*>(out + ((long long)pp * V + v) * C);
                _Float16 x, y;
                pg_split(tile[tx][r], x, y);
                oh[(c >> 3) * 16 + (c & 7)] = x;            // [8 hi | 8 lo] groups (see pg_store8)
                oh[(c >> 3) * 16 + 8 + (c & 7)] = y;
                pg_range_acc(rm, (unsigned)__builtin_bit_cast(unsigned short, x));
            }
        }
    }
    if (SPLIT) pg_range_commit(range, rm);
}
int th_gather_chan_major_launch(const float* pf, int V, int C, long long Pall, const int32_t* sel, int P, float* out,
                                int fmt, hipStream_t s, unsigned int* range) {
    if (P <= 0) return 0;
    if (fmt == TH_ROWS_SPLIT)
        hipLaunchKernelGGL(gather_chan_major_kernel<true>, dim3(th_cdiv(P, 32), th_cdiv(C, 32), V), dim3(256), 0, s, pf, V,
                           C, Pall, sel, P, out, range);
    else
        hipLaunchKernelGGL(gather_chan_major_kernel<false>, dim3(th_cdiv(P, 32), th_cdiv(C, 32), V), dim3(256), 0, s, pf, V,
                           C, Pall, sel, P, out, nullptr);
    TH_LAUNCH_CHECK();
    return 0;
}
