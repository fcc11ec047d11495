// K5: pixel-aligned feature gather.
//
// get_pixel_aligned_feature (if_clight_renderer.py:210-269): project a world
// point into the V reference cameras and bilinearly sample the 384-channel
// pixel_feat_map (grid_sample, align_corners=True, border).  The reference
// samples an NCHW map for *all* chunk points (384 strided cache lines per
// corner) and masks afterwards (cross_transformer.py:235); here the map is
// channels-last (th_nchw_to_nhwc, once per frame) so each corner is one
// contiguous 1.5 KB read, and only hull-valid samples are gathered.
// One wave per (sample, view) row; lanes span channels (float4 per lane).
// The map has C channels per pixel (384 full / 260 compact, see k_encoder.hip); output rows are ldo floats
// wide (>= C; the tail is zero-filled so the row can feed a K-padded GEMM directly: compact rows are 272).
// Bound: L2/HBM gather, 4 * 4C B per (sample, view) in, 4*ldo B out.
#include "th_internal.h"

typedef _Float16 pg_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pg_split(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// SPLIT: rows are written as ldo fp16 hi halves followed by ldo fp16 lo halves (TH_ROWS_SPLIT) instead of ldo floats
template <bool SPLIT>
__global__ __launch_bounds__(256) void pixgather_kernel(const float* __restrict__ map, int V, int C, int H, int W,
                                                        const float* __restrict__ pts_world, ThPointSrc ps,
                                                        const int32_t* __restrict__ sel, int P,
                                                        const float* __restrict__ cams,
                                                        const float* __restrict__ scale, float* __restrict__ out,
                                                        int ldo) {
    const int lane = threadIdx.x & 63;
    // XCD-aware remap (speed only): workgroup b runs on XCD b % 8, each XCD has its own L2.  Neighbouring
    // samples of a ray share bilinear corners, so give every XCD a CONTIGUOUS range of rows: logical block
    // = (b % 8) * ceil(nb / 8) + b / 8 (bijective incl. ragged tails via the bounds check below).
    const long long nb8 = ((long long)gridDim.x + 7) / 8;
    const long long lb = (long long)(blockIdx.x & 7) * nb8 + (blockIdx.x >> 3);
    long long row = lb * 4 + (threadIdx.x >> 6);                          // (sample, view)
    if (row >= (long long)P * V) return;
    int p = (int)(row / V), v = (int)(row % V);
    long long q = sel ? sel[p] : p;
    float x, y, z;
    if (pts_world) { x = pts_world[3 * q]; y = pts_world[3 * q + 1]; z = pts_world[3 * q + 2]; }
    else th_get_point(ps, q, x, y, z);
    float uu, vv;
    th_project(cams + 21 * v, x, y, z, uu, vv);
    Bilin b = th_bilinear_setup(uu, vv, scale[0], scale[1], H, W);
    const float* m = map + (long long)v * H * W * C;
    const float4* p00 = reinterpret_cast<const float4*>(m + (long long)b.i00 * C);
    const float4* p01 = reinterpret_cast<const float4*>(m + (long long)b.i01 * C);
    const float4* p10 = reinterpret_cast<const float4*>(m + (long long)b.i10 * C);
    const float4* p11 = reinterpret_cast<const float4*>(m + (long long)b.i11 * C);
    float4* o = reinterpret_cast<float4*>(out + row * ldo);
    _Float16* oh = reinterpret_cast<_Float16*>(out + row * ldo);
    // 16 B per lane: C = 384 -> 96 float4 per corner row = 1.5 wave-loads (the second one half masked)
    for (int c4 = lane; c4 < ldo / 4; c4 += 64) {
        if (c4 >= C / 4) {
            if (!SPLIT) o[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                pg_h4 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                *reinterpret_cast<pg_h4*>(oh + 4 * c4) = z;
                *reinterpret_cast<pg_h4*>(oh + ldo + 4 * c4) = z;
            }
            continue;
        }
        float4 a = p00[c4], bb = p01[c4], cc = p10[c4], d = p11[c4];
        float4 r;
        r.x = a.x * b.w00; r.x = r.x + bb.x * b.w01; r.x = r.x + cc.x * b.w10; r.x = r.x + d.x * b.w11;
        r.y = a.y * b.w00; r.y = r.y + bb.y * b.w01; r.y = r.y + cc.y * b.w10; r.y = r.y + d.y * b.w11;
        r.z = a.z * b.w00; r.z = r.z + bb.z * b.w01; r.z = r.z + cc.z * b.w10; r.z = r.z + d.z * b.w11;
        r.w = a.w * b.w00; r.w = r.w + bb.w * b.w01; r.w = r.w + cc.w * b.w10; r.w = r.w + d.w * b.w11;
        if (!SPLIT) o[c4] = r;
        else {
            pg_h4 hv, lv;
            _Float16 x, y;
            pg_split(r.x, x, y); hv[0] = x; lv[0] = y;
            pg_split(r.y, x, y); hv[1] = x; lv[1] = y;
            pg_split(r.z, x, y); hv[2] = x; lv[2] = y;
            pg_split(r.w, x, y); hv[3] = x; lv[3] = y;
            *reinterpret_cast<pg_h4*>(oh + 4 * c4) = hv;
            *reinterpret_cast<pg_h4*>(oh + ldo + 4 * c4) = lv;
        }
    }
}

int th_pixgather_launch(const float* map, int V, int C, int H, int W, const float* pts_world, const ThPointSrc* ps,
                        const int32_t* sel, int P, const float* cams, const float* scale, float* out, int ldo,
                        int fmt, hipStream_t s) {
    if (P <= 0) return 0;
    TH_REQUIRE((C & 3) == 0 && (ldo & 3) == 0 && ldo >= C, "channel count / row stride must be multiples of 4, ldo >= C");
    ThPointSrc src = ps ? *ps : ThPointSrc{};
    long long rows = (long long)P * V;
    const int nblk = 8 * th_cdiv(th_cdiv(rows, 4), 8);      // multiple of 8 so the XCD remap is onto
    if (fmt == TH_ROWS_SPLIT)
        hipLaunchKernelGGL(pixgather_kernel<true>, dim3(nblk), dim3(256), 0, s, map, V, C, H, W, pts_world, src, sel, P,
                           cams, scale, out, ldo);
    else
        hipLaunchKernelGGL(pixgather_kernel<false>, dim3(nblk), dim3(256), 0, s, map, V, C, H, W, pts_world, src, sel, P,
                           cams, scale, out, ldo);
    TH_LAUNCH_CHECK();
    return 0;
}

// Network.forward drop-in path: the caller already sampled pixel_feat as
// [V, C, Pall] (channel-major, cross_transformer.py:235 masks it).  Gather the
// selected points and transpose to rows [P][V][C] through LDS tiles.
template <bool SPLIT>
__global__ __launch_bounds__(256) void gather_chan_major_kernel(const float* __restrict__ pf, int V, int C,
                                                                long long Pall, const int32_t* __restrict__ sel,
                                                                int P, float* __restrict__ out) {
    __shared__ float tile[32][33];
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
                oh[c] = x;
                oh[C + c] = y;
            }
        }
    }
}
int th_gather_chan_major_launch(const float* pf, int V, int C, long long Pall, const int32_t* sel, int P, float* out,
                                int fmt, hipStream_t s) {
    if (P <= 0) return 0;
    if (fmt == TH_ROWS_SPLIT)
        hipLaunchKernelGGL(gather_chan_major_kernel<true>, dim3(th_cdiv(P, 32), th_cdiv(C, 32), V), dim3(256), 0, s, pf, V,
                           C, Pall, sel, P, out);
    else
        hipLaunchKernelGGL(gather_chan_major_kernel<false>, dim3(th_cdiv(P, 32), th_cdiv(C, 32), V), dim3(256), 0, s, pf, V,
                           C, Pall, sel, P, out);
    TH_LAUNCH_CHECK();
    return 0;
}
