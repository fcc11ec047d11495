// Demand-driven pixel map (round 5): which texels of the map does THIS frame's per-sample stage read?
//
// The reference writes pixel_feat_map over the whole image (encoder.py:133-146) and reads it at the valid samples of the frame
// (get_pixel_aligned_feature, if_clight_renderer.py:210-269, behind the hull mask :440-444) and at the projected input vertices
// (paint_neural_human :168-172).  Round 3 cropped the map to the box / row spans those reads CAN touch (th_map_box, from the
// vertices alone).  Once the frame's hull stage has run, the texels they DO touch are known exactly: the four bilinear corners
// of every valid sample in every view.  For one GPU that is ~70 % of the spans; for a rank of an N-rank job (its 8 x 8 pixel
// tiles of the frame) about a sixth of them -- and the map write and the map fold are the part of a rank's per-frame front that
// does not shrink with N.
//
// th_render_predemand (th_api.hip) runs these kernels behind a th_render_prepass, in its workspace:
//   demand buffer (uint32 words, NW = V * H * W / 32):
//     [0, NW)            need_fold   one bit per texel: a corner of a valid sample  -> map_fold_kernel<LIST> evaluates it
//     [NW, 2 NW)         need_map    need_fold | corners of the painted vertices     -> upsample_concat writes it
//     [2 NW]             n_list      number of set bits of need_fold
//     [2 NW + 16, ...)   list        their texel indices (view * H * W + y * W + x), ascending
//     behind the list    chunk sums  scratch of the compaction
// No host synchronisation: the valid-sample count is read on the device (the workspace's `info` words).
#include "th_internal.h"

__device__ __forceinline__ void dm_set(unsigned* __restrict__ bits, unsigned t) {
    unsigned* w = bits + (t >> 5);
    const unsigned b = 1u << (t & 31u);
    if ((*reinterpret_cast<volatile unsigned*>(w) & b) == 0u) atomicOr(w, b);     // (a stale "0" only costs a redundant atomic)
}
// the same for a wave whose consecutive lanes carry consecutive valid samples (neighbouring depths of a ray, neighbouring rays):
// a lane whose texel is its predecessor's has nothing to do -- most of a wave drops out before touching memory
__device__ __forceinline__ void dm_set_wave(unsigned* __restrict__ bits, unsigned t, bool on) {
    const unsigned prev = (unsigned)__builtin_amdgcn_update_dpp((int)~0u, (int)t, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    if (on && t != prev) dm_set(bits, t);
}

template <int V>
__global__ __launch_bounds__(256) void demand_samples_kernel(ThPointSrc ps, const int32_t* __restrict__ sel,
                                                             const int32_t* __restrict__ info, const float* __restrict__ cams,
                                                             const float* __restrict__ scale, int H, int W,
                                                             unsigned* __restrict__ need_fold) {
    const int n = info[2];                        // valid samples of the prepass (device-side count)
    const int HW = H * W;
    const float sx = scale[0], sy = scale[1];
    // (whole waves stay in the loop together: the DPP shift below reads the neighbour lane)
    for (int p0 = blockIdx.x * 256; p0 < n; p0 += gridDim.x * 256) {
        const int p = p0 + threadIdx.x;
        const bool on = p < n;
        float x, y, z;
        th_get_point(ps, sel[on ? p : n - 1], x, y, z);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float uu, vv;
            th_project(cams + 21 * v, x, y, z, uu, vv);
            const Bilin b = th_bilinear_setup(uu, vv, sx, sy, H, W);
            const unsigned r0 = (unsigned)(v * HW + b.y0 * W), r1 = (unsigned)(v * HW + b.y1 * W);
            dm_set_wave(need_fold, r0 + (unsigned)b.x0, on);
            dm_set_wave(need_fold, r0 + (unsigned)b.x1, on);
            dm_set_wave(need_fold, r1 + (unsigned)b.x0, on);
            dm_set_wave(need_fold, r1 + (unsigned)b.x1, on);
        }
    }
}

__global__ __launch_bounds__(256) void demand_verts_kernel(const float* __restrict__ verts, int nv, const float* __restrict__ cams,
                                                           int V, const float* __restrict__ scale, int H, int W,
                                                           unsigned* __restrict__ need_paint) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nv * V) return;
    const int v = i / nv, k = i - v * nv;
    float uu, vv;
    th_project(cams + 21 * v, verts[3 * k], verts[3 * k + 1], verts[3 * k + 2], uu, vv);
    const Bilin b = th_bilinear_setup(uu, vv, scale[0], scale[1], H, W);
    const unsigned r0 = (unsigned)(v * H * W + b.y0 * W), r1 = (unsigned)(v * H * W + b.y1 * W);
    dm_set(need_paint, r0 + (unsigned)b.x0);
    dm_set(need_paint, r0 + (unsigned)b.x1);
    dm_set(need_paint, r1 + (unsigned)b.x0);
    dm_set(need_paint, r1 + (unsigned)b.x1);
}

// need_map |= need_fold; the set bits of need_fold -> list (ascending) + count.  Two launches of NW / 1024 workgroups: the number of
// set bits per 1024-word chunk (a scratch tail of the buffer, th_demand_bytes), then every chunk emits its indices behind the
// chunks before it (one workgroup scanning all words took 107 us on a rank's dependent chain).
__global__ __launch_bounds__(1024) void demand_count_kernel(unsigned* __restrict__ dm, int NW, unsigned* __restrict__ csum) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x * 1024 + tid;
    const unsigned bits = w < NW ? dm[w] : 0u;
    if (w < NW && bits != 0u) dm[NW + w] |= bits;
    int c = __builtin_popcount(bits);
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int k = 0; k < 16; ++k) t += wsum[k];
        csum[blockIdx.x] = (unsigned)t;
    }
}
__global__ __launch_bounds__(1024) void demand_emit_kernel(unsigned* __restrict__ dm, int NW, const unsigned* __restrict__ csum) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int t = 0;
        for (int k = 0; k < (int)blockIdx.x; ++k) t += (int)csum[k];
        base_s = t;
        if (blockIdx.x == gridDim.x - 1) dm[2 * (size_t)NW] = (unsigned)(t + (int)csum[blockIdx.x]);
    }
    int* list = reinterpret_cast<int*>(dm + 2 * (size_t)NW + 16);
    const int w = blockIdx.x * 1024 + tid;
    const unsigned bits = w < NW ? dm[w] : 0u;
    const int c = __builtin_popcount(bits);
    int s = c;
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(s, d);
        if (lane >= d) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int off = base_s;
    for (int k = 0; k < wave; ++k) off += wsum[k];
    int o = off + s - c;
    unsigned b = bits;
    while (b) {
        const int k = __builtin_ctz(b);
        list[o++] = w * 32 + k;
        b &= b - 1u;
    }
}

size_t th_demand_bytes(int V, int H, int W) {
    const size_t T = (size_t)V * H * W;
    return (2 * (T / 32) + 16 + T + (T / 32 + 1023) / 1024 + 16) * 4;      // bitmaps, count, list, chunk sums
}

int th_demand_launch(const ThPointSrc& ps, const int32_t* sel, const int32_t* info, const float* cams, const float* scale, int V,
                     int H, int W, const float* verts_paint, int n_paint, void* demand, hipStream_t s) {
    TH_REQUIRE(V >= 1 && V <= 3 && (W % 64) == 0 && ((long long)V * H * W) % 32 == 0 && (long long)V * H * W < (1LL << 31),
               "demand-driven map: 1..3 views, image width a multiple of 64");
    const int NW = (int)((long long)V * H * W / 32);
    unsigned* dm = reinterpret_cast<unsigned*>(demand);
    TH_HIP(hipMemsetAsync(dm, 0, (2 * (size_t)NW + 16) * 4, s));
    const dim3 grid(1024), block(256);
    switch (V) {
        case 1: hipLaunchKernelGGL(demand_samples_kernel<1>, grid, block, 0, s, ps, sel, info, cams, scale, H, W, dm); break;
        case 2: hipLaunchKernelGGL(demand_samples_kernel<2>, grid, block, 0, s, ps, sel, info, cams, scale, H, W, dm); break;
        default: hipLaunchKernelGGL(demand_samples_kernel<3>, grid, block, 0, s, ps, sel, info, cams, scale, H, W, dm); break;
    }
    if (verts_paint != nullptr && n_paint > 0)
        hipLaunchKernelGGL(demand_verts_kernel, dim3(th_cdiv(n_paint * V, 256)), block, 0, s, verts_paint, n_paint, cams, V, scale, H,
                           W, dm + NW);
    unsigned* csum = dm + 2 * (size_t)NW + 16 + (size_t)V * H * W;
    const int chunks = th_cdiv(NW, 1024);
    hipLaunchKernelGGL(demand_count_kernel, dim3(chunks), dim3(1024), 0, s, dm, NW, csum);
    hipLaunchKernelGGL(demand_emit_kernel, dim3(chunks), dim3(1024), 0, s, dm, NW, csum);
    TH_LAUNCH_CHECK();
    return 0;
}
