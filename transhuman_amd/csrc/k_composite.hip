// K7: alpha-composited ray integration + view-direction embedding.
//
// raw2outputs (lib/networks/renderer/nerf_net_utils.py:14-59): delta_i =
// (z_{i+1}-z_i)*|d| with delta_last = 1e10*|d|, c = sigmoid(raw_rgb),
// alpha = 1 - exp(-relu(sigma)*delta), T_i = prod_{j<i}(1 - alpha_j + 1e-10),
// w = alpha*T, rgb = sum w c, depth = sum w z, acc = sum w.
// One wavefront per ray: with the reference's 64 samples/ray a ray is exactly
// one wave64; the exclusive transmittance product is a shuffle scan and the
// three sums are shuffle reductions (no LDS).  S != 64 runs in 64-sample
// passes carrying T across passes.
// Bound: HBM streaming, 16 B/sample in (raw) + 20 B/ray out.
#include "th_internal.h"

__global__ __launch_bounds__(256) void composite_kernel(const float4* __restrict__ raw, const float* __restrict__ zin,
                                                        ThPointSrc ps, int white, float* __restrict__ rgb,
                                                        float* __restrict__ acc, float* __restrict__ depth,
                                                        float* __restrict__ wout, const uint8_t* __restrict__ mask,
                                                        const int32_t* __restrict__ ray_hit) {
    const int lane = threadIdx.x & 63;
    int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= ps.R) return;
    const int S = ps.S;
    // A ray that misses the hull has no live sample (the hull stage raises ray_hit with every mask bit it sets): every weight is
    // 0 and the outputs are exact zeros (no white background either, see below) -- written without the 64 exponentials and the
    // scan.  4/5 of the headline frame's rays: 265 -> 90 us.
    if (ray_hit != nullptr && mask != nullptr && ray_hit[ray] == 0) {
        if (wout)
            for (int s = lane; s < S; s += 64) wout[(long long)ray * S + s] = 0.f;
        if (lane == 0) {
            rgb[3 * ray] = 0.f; rgb[3 * ray + 1] = 0.f; rgb[3 * ray + 2] = 0.f;
            acc[ray] = 0.f;
            depth[ray] = 0.f;
        }
        return;
    }
    float dx = ps.ray_d[3 * ray], dy = ps.ray_d[3 * ray + 1], dz = ps.ray_d[3 * ray + 2];
    float nd = dx * dx + dy * dy;
    nd = __fsqrt_rn(nd + dz * dz);                       // torch.norm(rays_d)
    float carryT = 1.0f;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    // (render_fast composites the rays that hit the hull only, if_clight_renderer.py:467-476: a ray that misses stays zero)
    const bool noisy = ps.noise != nullptr && (ray_hit == nullptr || ray_hit[ray] != 0);
    for (int base = 0; base < S; base += 64) {
        int s = base + lane;
        bool ok = s < S;
        float z = 0.f, zn = 0.f;
        if (ok) {
            z = zin ? zin[(long long)ray * S + s] : th_sample_z(ps, ray, s);
            if (s + 1 < S) zn = zin ? zin[(long long)ray * S + s + 1] : th_sample_z(ps, ray, s + 1);
        }
        float delta = (s + 1 < S) ? (zn - z) : 1e10f;
        delta = delta * nd;
        // mask given: samples outside the hull carry raw = 0 (cross_transformer.py:231) without being stored or read
        const bool live = ok && (mask == nullptr || mask[(long long)ray * S + s] != 0);
        float4 r = live ? raw[(long long)ray * S + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        // density noise (nerf_net_utils.py:39-44): on every sample of a composited ray, the zeros outside the hull included
        if (noisy && ok) r.w = r.w + ps.noise[(long long)ray * S + s];
        float alpha = ok ? 1.0f - expf(-fmaxf(r.w, 0.0f) * delta) : 0.0f;
        float t = (1.0f - alpha) + 1e-10f;                // factor contributed to later samples
        if (!ok) t = 1.0f;
        // inclusive product scan over the wave, then shift to exclusive
        float inc = t;
        for (int o = 1; o < 64; o <<= 1) {
            float u = __shfl_up(inc, o);
            if (lane >= o) inc = inc * u;
        }
        float excl = __shfl_up(inc, 1);
        if (lane == 0) excl = 1.0f;
        float T = carryT * excl;
        float w = alpha * T;
        carryT = carryT * __shfl(inc, 63);
        if (ok && wout) wout[(long long)ray * S + s] = w;
        float cr = 1.0f / (1.0f + expf(-r.x)), cg = 1.0f / (1.0f + expf(-r.y)), cb = 1.0f / (1.0f + expf(-r.z));
        sr += w * cr; sg += w * cg; sb += w * cb; sd += w * z; sa += w;
    }
    for (int o = 32; o > 0; o >>= 1) {
        sr += __shfl_xor(sr, o); sg += __shfl_xor(sg, o); sb += __shfl_xor(sb, o);
        sd += __shfl_xor(sd, o); sa += __shfl_xor(sa, o);
    }
    if (lane == 0) {
        // white background: render_fast composites only the rays that hit the hull and scatters them into zeros
        // (if_clight_renderer.py:467-476), so a ray that misses stays black even with white_bkgd; Renderer.render
        // (:486-498) composites every ray (ray_hit == nullptr or all ones)
        if (white && (ray_hit == nullptr || ray_hit[ray] != 0)) { float bg = 1.0f - sa; sr += bg; sg += bg; sb += bg; }
        rgb[3 * ray] = sr; rgb[3 * ray + 1] = sg; rgb[3 * ray + 2] = sb;
        acc[ray] = sa;
        depth[ray] = sd;
    }
}

int th_composite_launch(const float* raw, const float* z, const ThPointSrc& ps, int white, float* rgb, float* acc,
                        float* depth, float* wout, const uint8_t* mask, hipStream_t s, const int32_t* ray_hit) {
    if (ps.R <= 0) return 0;
    hipLaunchKernelGGL(composite_kernel, dim3(th_cdiv(ps.R, 4)), dim3(256), 0, s, (const float4*)raw, z, ps, white, rgb,
                       acc, depth, wout, mask, ray_hit);
    TH_LAUNCH_CHECK();
    return 0;
}

// view embedding: v = d/|d| ; [v, sin(2^k v), cos(2^k v)] (embedder.py:9-35, view_res=4)
// `hit` (optional): rays whose flag is zero are skipped -- their rows are never read (a view direction is looked up per
// SHADED sample, and every shaded sample belongs to a ray that touches the hull): 4/5 of the frame's 24 sin / cos per ray
__global__ void view_embed_kernel(const float* __restrict__ d, int R, int res, float* __restrict__ out,
                                  const int32_t* __restrict__ hit) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    if (hit != nullptr && hit[r] == 0) return;
    float x = d[3 * r], y = d[3 * r + 1], z = d[3 * r + 2];
    float n = x * x + y * y;
    n = __fsqrt_rn(n + z * z);
    float v[3] = {x / n, y / n, z / n};
    float* o = out + (long long)r * (3 + 6 * res);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    for (int k = 0; k < res; ++k) {
        float f = (float)(1 << k);
        for (int a = 0; a < 3; ++a) {
            o[3 + 6 * k + a] = sinf(v[a] * f);
            o[3 + 6 * k + 3 + a] = cosf(v[a] * f);
        }
    }
}
int th_view_embed_launch(const float* d, int R, int res, float* out, hipStream_t s, const int32_t* hit) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(view_embed_kernel, dim3(th_cdiv(R, 256)), dim3(256), 0, s, d, R, res, out, hit);
    TH_LAUNCH_CHECK();
    return 0;
}
