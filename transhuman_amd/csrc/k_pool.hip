// K2: paint SMPL vertices with image features, then pool them to tokens.
//
// paint_neural_human (if_clight_renderer.py:95-184): project the 6890 posed
// vertices into each reference camera (x_cam = R x + T, uvw = K x_cam, :123-126),
// bilinear grid_sample(align_corners=True, padding_mode="border") of the NCHW
// holder feature map (:186-208), zero the vertices the vizmap marks invisible
// (:176-182).  voxelization / can_body_grouping (:356-371, :415-427): per-cluster
// arithmetic mean over the CSR member list (the reference loops over N_c
// clusters x V views in Python, ~10^4 tiny launches at N_c = 1500).
// Per-frame, ~20 MB of traffic: latency-, not bandwidth-bound.
#include "th_internal.h"

// grid (n_verts, V); threads over channels
__global__ void paint_kernel(const float* __restrict__ map, int C, int H, int W, const float* __restrict__ verts,
                             int nv, const float* __restrict__ cams, const float* __restrict__ scale,
                             const uint8_t* __restrict__ viz, float* __restrict__ out) {
    int vert = blockIdx.x, view = blockIdx.y;
    float* o = out + ((long long)view * nv + vert) * C;
    if (viz && !viz[(long long)view * nv + vert]) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) o[c] = 0.0f;
        return;
    }
    float u, v;
    th_project(cams + 21 * view, verts[3 * vert], verts[3 * vert + 1], verts[3 * vert + 2], u, v);
    Bilin b = th_bilinear_setup(u, v, scale[0], scale[1], H, W);
    long long hw = (long long)H * W;
    const float* m = map + (long long)view * C * hw;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float* p = m + c * hw;
        float r = p[b.i00] * b.w00;
        r = r + p[b.i01] * b.w01;
        r = r + p[b.i10] * b.w10;
        r = r + p[b.i11] * b.w11;
        o[c] = r;
    }
}

int th_paint_launch(const float* map, int V, int C, int H, int W, const float* verts, int nv, const float* cams,
                    const float* scale, const uint8_t* viz, float* painted, hipStream_t s) {
    int th = C <= 64 ? 64 : (C <= 128 ? 128 : 192);
    hipLaunchKernelGGL(paint_kernel, dim3(nv, V), dim3(th), 0, s, map, C, H, W, verts, nv, cams, scale, viz, painted);
    TH_LAUNCH_CHECK();
    return 0;
}

// grid (n_clusters, batch); mean over the member rows in stored order
__global__ void segmean_kernel(const float* __restrict__ src, long long batch_stride, int width,
                               const int32_t* __restrict__ off, const int32_t* __restrict__ mem, int nc,
                               float* __restrict__ out) {
    int c = blockIdx.x, b = blockIdx.y;
    int s0 = off[c], s1 = off[c + 1];
    const float* base = src + b * batch_stride;
    float n = (float)(s1 - s0);
    for (int k = threadIdx.x; k < width; k += blockDim.x) {
        float acc = 0.f;
        for (int j = s0; j < s1; ++j) acc = acc + base[(long long)mem[j] * width + k];
        out[((long long)b * nc + c) * width + k] = acc / n;
    }
}
int th_segmean_launch(const float* src, int batch, long long batch_stride, int width, const int32_t* off,
                      const int32_t* mem, int nc, float* out, hipStream_t s) {
    int th = width <= 64 ? 64 : (width <= 128 ? 128 : 256);
    hipLaunchKernelGGL(segmean_kernel, dim3(nc, batch), dim3(th), 0, s, src, batch_stride, width, off, mem, nc, out);
    TH_LAUNCH_CHECK();
    return 0;
}

// blend [nv,4,4] f64 -> cluster mean in f64 -> rot[c][i*3+j] = (float)mean[i][j]
// (if_clight_renderer.py:544 then cross_transformer.py:185)
__global__ void segmean_rot_kernel(const double* __restrict__ blend, const int32_t* __restrict__ off,
                                   const int32_t* __restrict__ mem, int nc, float* __restrict__ rot) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    int s0 = off[c], s1 = off[c + 1];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = s0; j < s1; ++j) {
        const double* m = blend + (long long)mem[j] * 16;
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) acc[i * 3 + k] += m[i * 4 + k];
    }
    double n = (double)(s1 - s0);
    for (int q = 0; q < 9; ++q) rot[(long long)c * 9 + q] = (float)(acc[q] / n);
}
int th_segmean_rot_launch(const double* blend, const int32_t* off, const int32_t* mem, int nc, float* rot,
                          hipStream_t s) {
    hipLaunchKernelGGL(segmean_rot_kernel, dim3(th_cdiv(nc, 64)), dim3(64), 0, s, blend, off, mem, nc, rot);
    TH_LAUNCH_CHECK();
    return 0;
}

// [V,C,HW] -> [V,HW,C] through a padded 32x32 LDS tile; one 1.2 GB read + write
// per frame at C=384 (HBM-bound).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, long long HW,
                                                           float* __restrict__ dst) {
    __shared__ float tile[32][33];
    int v = blockIdx.z;
    long long p0 = (long long)blockIdx.x * 32;
    int c0 = blockIdx.y * 32;
    const float* s = src + (long long)v * C * HW;
    float* d = dst + (long long)v * C * HW;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r;
        long long p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? s[(long long)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        long long p = p0 + r;
        int c = c0 + tx;
        if (c < C && p < HW) d[p * C + c] = tile[tx][r];
    }
}
int th_nchw_to_nhwc_launch(const float* src, int V, int C, int H, int W, float* dst, hipStream_t s) {
    long long HW = (long long)H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(th_cdiv(HW, 32), th_cdiv(C, 32), V), dim3(256), 0, s, src, C, HW,
                       dst);
    TH_LAUNCH_CHECK();
    return 0;
}
