// K8: tail of SpatialEncoder.forward written channels-last, + painting from the channels-last map.
//
// encoder.py:133-146: the three ResNet latents (64ch @H/2, 64ch @H/4, 128ch @H/8) are bilinearly
// upsampled (align_corners=True) to HxW, concatenated with upsample_color(img) (1x1 conv 3->128) into the
// 384-channel pixel_feat_map, and reduction_layer (1x1 conv 384->192) makes holder_feat_map.  The
// reference materialises both maps NCHW (1.2 GB + 0.6 GB per frame) through six torch kernels; the
// per-sample gather (K5) wants channels-last.  Here ONE kernel writes the channels-last map directly
// (1.2 GB written once, nothing re-read), and holder_feat_map is never built: it is only ever sampled at
// the 3 x 6890 projected SMPL vertices (if_clight_renderer.py:168-172), and a 1x1 conv commutes with
// bilinear sampling, so the 384-ch map is sampled at the vertices and the 384->192 layer is applied to
// those 20 670 rows (th_gemm) before the visibility mask and the cluster mean.
// Compact form (wc == nullptr): the colour lift is not applied here at all -- the map carries the raw
// r,g,b after the 256 latent channels ([V,H,W,260]) and the lift is folded into the three MLP layers and
// the reduction layer that consume those channels (see th_set_mlp_weights): 0.82 GB written instead of
// 1.2 GB and a third fewer bytes per gathered sample.
// Bound: HBM write stream; the latents (75 MB) stay L2/MALL resident.
#include "th_internal.h"

struct UpsSrc {
    const float* p;    // [V, C, h, w]
    int C, h, w;
    int c0;            // first output channel
};

// torch upsample_bilinear2d(align_corners=True): src = dst * (in-1)/(out-1)
__device__ __forceinline__ void ups_coord(int dst, int in, int out, int& i0, int& i1, float& l0, float& l1) {
    float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float src = scale * (float)dst;
    i0 = (int)src;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

// grid (W/64, H, V*NG): z%NG selects a 64-channel group: 0 lat0, 1 lat1, 2..3 lat2, then either 4..5 lifted
// colour (NG = 6, CO = 384) or 4 = raw r,g,b,0 (NG = 5, CO = 260).
// Phase 1: lane = x (coalesced source reads along x), waves stride over the group's 64 channels -> LDS tile.
// Phase 2: 64 channels of a pixel are 256 contiguous bytes of the NHWC map.
__global__ __launch_bounds__(256) void upsample_concat_nhwc_kernel(UpsSrc s0, UpsSrc s1, UpsSrc s2,
                                                                   const float* __restrict__ img,
                                                                   const float* __restrict__ wc,
                                                                   const float* __restrict__ bc, int H, int W,
                                                                   int NG, int CO, float* __restrict__ out,
                                                                   float* __restrict__ rgb4,
                                                                   const int32_t* __restrict__ box,
                                                                   const unsigned* __restrict__ need) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y;
    const int v = blockIdx.z / NG, grp = blockIdx.z % NG;
    // demand-driven map (k_demand.hip; W % 64 == 0): one bit per texel -- this workgroup's 64 texels are two words
    unsigned long long want = ~0ull;
    if (need != nullptr) {
        const unsigned* nw = need + ((((long long)v * H + y) * W + (long long)blockIdx.x * 64) >> 5);
        want = (unsigned long long)nw[0] | ((unsigned long long)nw[1] << 32);
        if (want == 0ull) return;
    }
    if (box != nullptr) {               // cropped map (map_box_kernel): 64-pixel spans outside the view's box are not written
        const int by0 = box[4 * v + 1], by1 = box[4 * v + 3];
        if (y < by0 || y > by1) return;
        // (the row's own span [x0, x1] behind the boxes: a tighter outline of the body than the box)
        const int32_t* sp = box + 4 * (gridDim.z / NG) + ((long long)v * H + y) * 2;
        if ((int)blockIdx.x * 64 + 63 < sp[0] || (int)blockIdx.x * 64 > sp[1]) return;
    }
    int cout0;
    if (grp == 4 && NG == 5) {          // compact map: channels 256..259 = r, g, b, 0 (one float4 per pixel)
        if (wave == 0 && x < W && ((want >> lane) & 1ull)) {
            long long hw = (long long)H * W;
            const float* ip = img + (long long)v * 3 * hw + (long long)y * W + x;
            // (split layout: the colour plane [V,H,W,4] behind the 256-channel latent plane)
            float* dst = rgb4 ? rgb4 + (((long long)v * H + y) * W + x) * 4 : out + (((long long)v * H + y) * W + x) * CO + 256;
            *reinterpret_cast<float4*>(dst) = make_float4(ip[0], ip[hw], ip[2 * hw], 0.f);
        }
        return;
    }
    if (grp < 4) {
        UpsSrc s = grp == 0 ? s0 : (grp == 1 ? s1 : s2);
        int cin0 = grp == 3 ? 64 : 0;
        cout0 = s.c0 + cin0;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        ups_coord(y, s.h, H, y0, y1, ly0, ly1);
        ups_coord(x < W ? x : W - 1, s.w, W, x0, x1, lx0, lx1);
        const float* base = s.p + ((long long)v * s.C + cin0) * s.h * s.w;
        for (int c = wave; c < 64; c += 4) {
            const float* pl = base + (long long)c * s.h * s.w;
            float v00 = pl[y0 * s.w + x0], v01 = pl[y0 * s.w + x1], v10 = pl[y1 * s.w + x0], v11 = pl[y1 * s.w + x1];
            tile[lane][c] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        }
    } else {
        int cin0 = (grp - 4) * 64;
        cout0 = 256 + cin0;
        int xx = x < W ? x : W - 1;
        long long hw = (long long)H * W;
        const float* ip = img + (long long)v * 3 * hw + (long long)y * W + xx;
        float r = ip[0], g = ip[hw], b = ip[2 * hw];
        for (int c = wave; c < 64; c += 4) {
            const float* w = wc + (cin0 + c) * 3;
            tile[lane][c] = fmaf(b, w[2], fmaf(g, w[1], r * w[0])) + bc[cin0 + c];
        }
    }
    __syncthreads();
    float* orow = out + (((long long)v * H + y) * W + (long long)blockIdx.x * 64) * CO + cout0;
    for (int px = wave; px < 64; px += 4)
        if (blockIdx.x * 64 + px < W && ((want >> px) & 1ull)) orow[(long long)px * CO + lane] = tile[px][lane];
}

int th_upsample_concat_launch(const float* img, const float* lat0, const float* lat1, const float* lat2,
                              const int* dims /* h0,w0,h1,w1,h2,w2 */, int V, int H, int W, const float* wc,
                              const float* bc, float* out, hipStream_t s, int split, const int32_t* box, const unsigned* need) {
    UpsSrc s0{lat0, 64, dims[0], dims[1], 0};
    UpsSrc s1{lat1, 64, dims[2], dims[3], 64};
    UpsSrc s2{lat2, 128, dims[4], dims[5], 128};
    const int NG = wc ? 6 : 5, CO = wc ? 384 : (split ? 256 : 260);
    float* rgb4 = (!wc && split) ? out + (long long)V * H * W * 256 : nullptr;
    hipLaunchKernelGGL(upsample_concat_nhwc_kernel, dim3(th_cdiv(W, 64), H, V * NG), dim3(256), 0, s, s0, s1, s2, img, wc,
                       bc, H, W, NG, CO, out, rgb4, box, need);
    TH_LAUNCH_CHECK();
    return 0;
}

// The part of a view's map the per-sample stage can touch.  Every point it gathers at lies within `reach` of a vertex
// (hull-valid samples: within hull_thresh of a target vertex, if_clight_renderer.py:440-444; painting: the input
// vertices themselves), i.e. inside the axis-aligned cube of half-width `reach` around that vertex.  u = px / pz and
// v = py / pz are linear-fractional in the point, so over a cube that lies in front of the camera (pz > 0 at its eight
// corners) their extrema sit at corners: the box of the projected corners of all cubes, in map texel coordinates (the
// expression of th_bilinear_setup), widened by two texels for the second bilinear corner and fp32 rounding, contains
// every texel those gathers read.  Border clamping is monotone, so clamping the box to the image keeps that true.  A cube
// that reaches behind a camera makes that view's box the whole image.
// Three small launches (one workgroup per view did all of it in 190 us -- on the side stream's dependent chain, which is what a rank
// of an 8-rank job waits for): init (boxes empty, spans empty, flags 0), NB blocks per view over the vertices (block-local spans
// in LDS, merged with integer atomics), finalise (a view with a cube at or behind its camera plane, or no vertices: whole image).
// Buffer: [V][4] boxes, [V][H][2] spans, [V] flags.
#define MAPBOX_NB 8
__global__ void map_box_init_kernel(int V, int H, int W, int32_t* __restrict__ box) {
    const int v = blockIdx.x;
    int32_t* sp = box + 4 * V + (long long)v * H * 2;
    for (int i = threadIdx.x; i < H; i += blockDim.x) { sp[2 * i] = W; sp[2 * i + 1] = -1; }
    if (threadIdx.x == 0) {
        box[4 * v] = 0x7fffffff; box[4 * v + 1] = 0x7fffffff; box[4 * v + 2] = -1; box[4 * v + 3] = -1;
        box[4 * V + (long long)V * H * 2 + v] = 0;
    }
}
__global__ __launch_bounds__(1024) void map_box_kernel(const float* __restrict__ va, int na, const float* __restrict__ vb,
                                                       int nb, const float* __restrict__ cams,
                                                       const float* __restrict__ scale, int H, int W, float reach,
                                                       int32_t* __restrict__ box) {
    const int v = blockIdx.x, V = gridDim.x, tid = threadIdx.x;
    const float* cam = cams + 21 * v;
    bool bad = false;
    // per image row the span [x0, x1] of the projected cubes that touch it (LDS, H <= 4096): the union of the vertices' own
    // boxes row by row -- every gather within reach of a vertex reads inside that vertex's box, hence inside its rows' spans
    extern __shared__ int span_l[];                 // [H][2]
    __shared__ int bx[4];
    for (int i = tid; i < H; i += 1024) { span_l[2 * i] = W; span_l[2 * i + 1] = -1; }
    if (tid == 0) { bx[0] = 0x7fffffff; bx[1] = 0x7fffffff; bx[2] = -1; bx[3] = -1; }
    __syncthreads();
    for (int i = blockIdx.y * 1024 + tid; i < na + nb; i += 1024 * gridDim.y) {
        const float* p = i < na ? va + 3 * i : vb + 3 * (i - na);
        const float x = p[0], y = p[1], z = p[2];
        float vx0 = 3.0e38f, vx1 = -3.0e38f, vy0 = 3.0e38f, vy1 = -3.0e38f;
        bool vbad = false;
        for (int k = 0; k < 8; ++k) {
            const float qx = x + ((k & 1) ? reach : -reach), qy = y + ((k & 2) ? reach : -reach),
                        qz = z + ((k & 4) ? reach : -reach);
            const float cx = fmaf(cam[2], qz, fmaf(cam[1], qy, cam[0] * qx)) + cam[9];
            const float cy = fmaf(cam[5], qz, fmaf(cam[4], qy, cam[3] * qx)) + cam[10];
            const float cz = fmaf(cam[8], qz, fmaf(cam[7], qy, cam[6] * qx)) + cam[11];
            const float* K = cam + 12;
            const float pz = fmaf(K[8], cz, fmaf(K[7], cy, K[6] * cx));
            float u, w;
            th_project(cam, qx, qy, qz, u, w);
            const float ix = ((u * scale[0] - 1.0f + 1.0f) / 2.0f) * (float)(W - 1);
            const float iy = ((w * scale[1] - 1.0f + 1.0f) / 2.0f) * (float)(H - 1);
            if (!(pz > 1e-6f) || !(fabsf(ix) < 1.0e9f) || !(fabsf(iy) < 1.0e9f)) vbad = true;
            vx0 = fminf(vx0, ix); vx1 = fmaxf(vx1, ix);
            vy0 = fminf(vy0, iy); vy1 = fmaxf(vy1, iy);
        }
        bad = bad || vbad;
        if (!vbad) {            // this vertex's box (widened by two texels for the second bilinear corner and fp32 rounding, clamped)
            const int ax0 = (int)fminf(fmaxf(floorf(vx0) - 2.0f, 0.0f), (float)(W - 1));
            const int ax1 = (int)fminf(fmaxf(floorf(vx1) + 3.0f, 0.0f), (float)(W - 1));
            const int ay0 = (int)fminf(fmaxf(floorf(vy0) - 2.0f, 0.0f), (float)(H - 1));
            const int ay1 = (int)fminf(fmaxf(floorf(vy1) + 3.0f, 0.0f), (float)(H - 1));
            atomicMin(&bx[0], ax0); atomicMin(&bx[1], ay0); atomicMax(&bx[2], ax1); atomicMax(&bx[3], ay1);
            for (int yy = ay0; yy <= ay1; ++yy) {
                atomicMin(&span_l[2 * yy], ax0);
                atomicMax(&span_l[2 * yy + 1], ax1);
            }
        }
    }
    if (bad) atomicOr(&box[4 * V + (long long)V * H * 2 + v], 1);
    __syncthreads();
    int32_t* sp = box + 4 * V + (long long)v * H * 2;
    for (int i = tid; i < H; i += 1024)
        if (span_l[2 * i + 1] >= span_l[2 * i]) {
            atomicMin(&sp[2 * i], span_l[2 * i]);
            atomicMax(&sp[2 * i + 1], span_l[2 * i + 1]);
        }
    if (tid == 0 && bx[2] >= 0) {
        atomicMin(&box[4 * v], bx[0]); atomicMin(&box[4 * v + 1], bx[1]);
        atomicMax(&box[4 * v + 2], bx[2]); atomicMax(&box[4 * v + 3], bx[3]);
    }
}
__global__ void map_box_final_kernel(int V, int H, int W, int nverts, int32_t* __restrict__ box) {
    const int v = blockIdx.x;
    const bool full = box[4 * V + (long long)V * H * 2 + v] != 0 || nverts <= 0 || box[4 * v + 2] < 0;
    if (!full) return;
    int32_t* sp = box + 4 * V + (long long)v * H * 2;
    for (int i = threadIdx.x; i < H; i += blockDim.x) { sp[2 * i] = 0; sp[2 * i + 1] = W - 1; }
    if (threadIdx.x == 0) { box[4 * v] = 0; box[4 * v + 1] = 0; box[4 * v + 2] = W - 1; box[4 * v + 3] = H - 1; }
}
int th_map_box_launch(const float* va, int na, const float* vb, int nb, const float* cams, int V, const float* scale, int H,
                      int W, float reach, int32_t* box, hipStream_t s) {
    TH_REQUIRE(H <= 4096, "th_map_box: at most 4096 image rows (row spans in LDS)");
    hipLaunchKernelGGL(map_box_init_kernel, dim3(V), dim3(256), 0, s, V, H, W, box);
    hipLaunchKernelGGL(map_box_kernel, dim3(V, MAPBOX_NB), dim3(1024), (size_t)H * 2 * sizeof(int), s, va, na, vb, nb, cams, scale, H,
                       W, reach, box);
    hipLaunchKernelGGL(map_box_final_kernel, dim3(V), dim3(256), 0, s, V, H, W, na + nb, box);
    TH_LAUNCH_CHECK();
    return 0;
}

// Colour-lift fold: a layer L reading the full 384-channel feature [latent(256) | Wc rgb + bc] equals the layer
//   W' = [W[:, :256] | W[:, 256:] Wc | 0] (in_f 260),  b' = b + W[:, 256:] bc   reading [latent | r g b | 0].
__global__ void fold_color_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ wc,
                                  const float* __restrict__ bc, int N, float* __restrict__ Wo, float* __restrict__ bo) {
    int o = blockIdx.x;
    for (int k = threadIdx.x; k < 256; k += blockDim.x) Wo[(long long)o * 260 + k] = W[(long long)o * 384 + k];
    if (threadIdx.x < 4) {
        double acc = 0.0;
        if (threadIdx.x < 3)
            for (int c = 0; c < 128; ++c) acc += (double)W[(long long)o * 384 + 256 + c] * (double)wc[c * 3 + threadIdx.x];
        else {
            acc = b ? (double)b[o] : 0.0;
            for (int c = 0; c < 128; ++c) acc += (double)W[(long long)o * 384 + 256 + c] * (bc ? (double)bc[c] : 0.0);
        }
        if (threadIdx.x < 3) Wo[(long long)o * 260 + 256 + threadIdx.x] = (float)acc;
        else { Wo[(long long)o * 260 + 259] = 0.f; bo[o] = (float)acc; }
    }
}
int th_fold_color_launch(const float* W, const float* b, const float* wc, const float* bc, int N, float* Wo, float* bo,
                         hipStream_t s) {
    hipLaunchKernelGGL(fold_color_kernel, dim3(N), dim3(256), 0, s, W, b, wc, bc, N, Wo, bo);
    TH_LAUNCH_CHECK();
    return 0;
}

// tokens[v][c][:] = mean over cluster members m of ( visible(v,m) ? rows[(m*V + v)][:] : 0 )
// (rows come from the [P,V,C] gather layout; invisible vertices count as zeros, if_clight_renderer.py:181-182,:364)
__global__ void segmean_masked_kernel(const float* __restrict__ rows, int V, int width, const uint8_t* __restrict__ viz,
                                      int nv, const int32_t* __restrict__ off, const int32_t* __restrict__ mem, int nc,
                                      float* __restrict__ out) {
    int c = blockIdx.x, v = blockIdx.y;
    int s0 = off[c], s1 = off[c + 1];
    float n = (float)(s1 - s0);
    for (int k = threadIdx.x; k < width; k += blockDim.x) {
        float acc = 0.f;
        for (int j = s0; j < s1; ++j) {
            int m = mem[j];
            float val = (!viz || viz[(long long)v * nv + m]) ? rows[((long long)m * V + v) * width + k] : 0.f;
            acc = acc + val;
        }
        out[((long long)v * nc + c) * width + k] = acc / n;
    }
}

int th_segmean_masked_launch(const float* rows, int V, int width, const uint8_t* viz, int nv, const int32_t* off,
                             const int32_t* mem, int nc, float* out, hipStream_t s) {
    int th = width <= 64 ? 64 : (width <= 128 ? 128 : 256);
    hipLaunchKernelGGL(segmean_masked_kernel, dim3(nc, V), dim3(th), 0, s, rows, V, width, viz, nv, off, mem, nc, out);
    TH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// K11: train-mode BatchNorm2d (+ residual) (+ ReLU) on NCHW tensors -- the elementwise tail of every ResNet stage of
// SpatialEncoder (encoder.py:114-126 runs torchvision's BasicBlocks with the network in train(), run.py:29: batch
// statistics, running statistics updated).  torch issues BN, the num_batches_tracked increment, the ReLU and the
// residual add as 3-4 launches per site; here: one statistics pass + one apply pass.
//   y = (x - mean_c) * rsqrt(var_c + eps) * gamma_c + beta_c  [+ res]  [relu]       (var biased, like F.batch_norm)
//   running_mean = (1-m) running_mean + m mean ; running_var = (1-m) running_var + m var * n/(n-1)
// Statistics: fixed-order partial sums in float64 (per (n, c) plane slice) -> deterministic, and at least as accurate
// as the fp32 reductions of the stock kernels.
// ---------------------------------------------------------------------------
#define BN_SPLIT_ELEMS 16384       // elements of one (n, c) plane per statistics workgroup

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int C, int HW, int splits,
                                                       double* __restrict__ part /*[C][N*splits][2]*/, int NS) {
    const int c = blockIdx.x, ns = blockIdx.y, n = ns / splits, sp = ns - n * splits;
    const int per = (HW + splits - 1) / splits;
    const int lo = sp * per, hi = min(HW, lo + per);
    const float* p = x + ((long long)n * C + c) * HW;
    double s = 0.0, q = 0.0;
    if ((HW & 3) == 0 && (lo & 3) == 0) {
        const int hi4 = lo + ((hi - lo) & ~3);
        for (int i = lo + 4 * threadIdx.x; i < hi4; i += 4 * 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
        for (int i = hi4 + threadIdx.x; i < hi; i += 256) { s += p[i]; q += (double)p[i] * p[i]; }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) { s += p[i]; q += (double)p[i] * p[i]; }
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    __shared__ double sh[8];
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = s; sh[2 * (threadIdx.x >> 6) + 1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long long)c * NS + ns) * 2 + 0] = (sh[0] + sh[2]) + (sh[4] + sh[6]);
        part[((long long)c * NS + ns) * 2 + 1] = (sh[1] + sh[3]) + (sh[5] + sh[7]);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, int C,
                                                       int HW, int splits, const double* __restrict__ part, int NS,
                                                       long long count, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float momentum,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       int relu, float* __restrict__ y,
                                                       const float2* __restrict__ fpart, int NP) {
    const int c = blockIdx.x, ns = blockIdx.y, n = ns / splits, sp = ns - n * splits;
    __shared__ float ss[2];
    __shared__ double red[8];
    if (fpart != nullptr) {
        // statistics left by the producing convolution (th_conv2d_stats): NP float2 partials per channel, added here in
        // float64 in a fixed order (thread t takes t, t + 256, ...; then lanes, then waves)
        double s = 0.0, q = 0.0;
        for (int k = threadIdx.x; k < NP; k += 256) {
            const float2 v = fpart[(long long)c * NP + k];
            s += (double)v.x;
            q += (double)v.y;
        }
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
        if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = s; red[2 * (threadIdx.x >> 6) + 1] = q; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && part == nullptr && fpart == nullptr) {
        // eval mode (nn.BatchNorm2d.eval(): the reference's Trainer.val, trainer.py:131): the running statistics ARE the
        // statistics, nothing is updated
        const float g = gamma ? gamma[c] : 1.f;
        ss[0] = (float)(1.0 / sqrt((double)run_var[c] + (double)eps)) * g;
        ss[1] = run_mean[c];
    } else if (threadIdx.x == 0) {
        double s = 0.0, q = 0.0;
        if (fpart != nullptr) {
            s = (red[0] + red[2]) + (red[4] + red[6]);
            q = (red[1] + red[3]) + (red[5] + red[7]);
        } else {
            for (int k = 0; k < NS; ++k) { s += part[((long long)c * NS + k) * 2]; q += part[((long long)c * NS + k) * 2 + 1]; }
        }
        const double mean = s / (double)count;
        double var = q / (double)count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        ss[0] = invstd * g;
        ss[1] = (float)mean;
        if (ns == 0 && run_mean != nullptr) {
            const double unb = count > 1 ? var * ((double)count / (double)(count - 1)) : var;
            run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * mean);
            run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unb);
        }
    }
    __syncthreads();
    const float sc = ss[0], mean = ss[1], b = beta ? beta[c] : 0.f;
    const int per = (HW + splits - 1) / splits;
    const int lo = sp * per, hi = min(HW, lo + per);
    const long long base = ((long long)n * C + c) * HW;
    const float* p = x + base;
    const float* r = res ? res + base : nullptr;
    float* o = y + base;
    if ((HW & 3) == 0 && (lo & 3) == 0) {
        const int hi4 = lo + ((hi - lo) & ~3);
        for (int i = lo + 4 * threadIdx.x; i < hi4; i += 4 * 256) {
            float4 v = *reinterpret_cast<const float4*>(p + i);
            v.x = (v.x - mean) * sc + b; v.y = (v.y - mean) * sc + b; v.z = (v.z - mean) * sc + b; v.w = (v.w - mean) * sc + b;
            if (r) {
                const float4 a = *reinterpret_cast<const float4*>(r + i);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(o + i) = v;
        }
        for (int i = hi4 + threadIdx.x; i < hi; i += 256) {
            float v = (p[i] - mean) * sc + b;
            if (r) v += r[i];
            o[i] = relu ? fmaxf(v, 0.f) : v;
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            float v = (p[i] - mean) * sc + b;
            if (r) v += r[i];
            o[i] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

size_t th_bn_ws(int N, int C, int HW) {
    const int splits = (HW + BN_SPLIT_ELEMS - 1) / BN_SPLIT_ELEMS;
    return th_align((size_t)C * N * splits * 2 * sizeof(double));
}

int th_bn_act_launch(const float* x, const float* res, int N, int C, int HW, const float* gamma, const float* beta,
                     float eps, float momentum, float* run_mean, float* run_var, int relu, float* y, void* ws,
                     size_t ws_bytes, hipStream_t s, int eval, const void* conv_stats, int conv_np) {
    TH_REQUIRE(N > 0 && C > 0 && HW > 0, "empty tensor");
    TH_REQUIRE(eval || conv_stats || ws_bytes >= th_bn_ws(N, C, HW), "workspace too small");
    TH_REQUIRE(!conv_stats || (!eval && conv_np > 0), "convolution statistics: train mode, at least one partial");
    TH_REQUIRE(!eval || (run_mean && run_var), "eval-mode BatchNorm needs the running statistics");
    TH_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "tensors must be 16-byte aligned");
    const int splits = (HW + BN_SPLIT_ELEMS - 1) / BN_SPLIT_ELEMS;
    const int NS = N * splits;
    TH_REQUIRE(NS <= 65535, "too many plane slices");
    double* part = (eval || conv_stats) ? nullptr : (double*)ws;
    if (part) hipLaunchKernelGGL(bn_stats_kernel, dim3(C, NS), dim3(256), 0, s, x, C, HW, splits, part, NS);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(C, NS), dim3(256), 0, s, x, res, C, HW, splits, part, NS,
                       (long long)N * HW, gamma, beta, eps, momentum, run_mean, run_var, relu, y,
                       (const float2*)conv_stats, conv_np);
    TH_LAUNCH_CHECK();
    return 0;
}
