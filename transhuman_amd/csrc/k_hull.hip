// K1: sample placement + SMPL-hull membership + compaction of valid samples.
//
// Replaces get_sampling_points (if_clight_renderer.py:271-287), the brute-force
// pytorch3d knn_points(K=1) over all R*S samples vs 6890 posed vertices (:440),
// sqrt / "< 0.1" / per-ray any (:441-444) and the boolean-mask compactions
// (:459-465, cross_transformer.py:230-236).  Same test in if_mesh_renderer.py:53-56.
//
// The predicate evaluated per (sample, candidate vertex) is exactly the
// reference's: d2 = dx*dx + dy*dy + dz*dz in fp32 (x,y,z order, no FMA),
// sqrtf correctly rounded, compared with fp32 0.1.  "min over all vertices <
// 0.1" == "exists a vertex with dist < 0.1", so a uniform grid with cell size
// >= 0.1*1.05 only prunes vertices that cannot pass; the outcome is bit-identical
// to brute force while doing ~27 cells x a few vertices instead of 6890.
// Bound: HBM/L2 streaming of 32 B/ray in + 1 B/sample out (bytes, not FLOPs).
#include <stdlib.h>

#include "th_internal.h"

struct GridInfo {
    float gmin[3];
    float inv_h;
    int dim[3];
    int ncell;
};

#define GRID_MAX_DIM 64
#define GRID_MAX_CELLS (GRID_MAX_DIM * GRID_MAX_DIM * GRID_MAX_DIM)

__device__ __forceinline__ int cell_coord(float x, float gmin, float inv_h) {
    return (int)floorf((x - gmin) * inv_h);
}

// one block: AABB of the vertices -> GridInfo, and zero the counters
__global__ __launch_bounds__(1024) void grid_setup_kernel(const float* __restrict__ verts, int nv, float h0,
                                                          GridInfo* __restrict__ gi, int* __restrict__ counts) {
    __shared__ float smin[3][16], smax[3][16];
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = threadIdx.x; i < nv; i += blockDim.x)
        for (int a = 0; a < 3; ++a) {
            float v = verts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
        if ((threadIdx.x & 63) == 0) {
            smin[a][threadIdx.x >> 6] = mn[a];
            smax[a][threadIdx.x >> 6] = mx[a];
        }
    }
    __syncthreads();
    __shared__ GridInfo g;
    if (threadIdx.x == 0) {
        float ext = 0.f;
        int nw = blockDim.x >> 6;
        for (int a = 0; a < 3; ++a) {
            float lo = smin[a][0], hi = smax[a][0];
            for (int w = 1; w < nw; ++w) { lo = fminf(lo, smin[a][w]); hi = fmaxf(hi, smax[a][w]); }
            g.gmin[a] = lo;
            smax[a][0] = hi;
            ext = fmaxf(ext, hi - lo);
        }
        float h = fmaxf(h0, ext / (float)(GRID_MAX_DIM - 2));
        g.inv_h = 1.0f / h;
        int nc = 1;
        for (int a = 0; a < 3; ++a) {
            int d = cell_coord(smax[a][0], g.gmin[a], g.inv_h) + 1;
            d = d < 1 ? 1 : (d > GRID_MAX_DIM ? GRID_MAX_DIM : d);
            g.dim[a] = d;
            nc *= d;
        }
        g.ncell = nc;
        *gi = g;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g.ncell + 1; i += blockDim.x) counts[i] = 0;
}

__device__ __forceinline__ int vert_cell(const GridInfo& g, float x, float y, float z) {
    int cx = cell_coord(x, g.gmin[0], g.inv_h), cy = cell_coord(y, g.gmin[1], g.inv_h),
        cz = cell_coord(z, g.gmin[2], g.inv_h);
    cx = min(max(cx, 0), g.dim[0] - 1);
    cy = min(max(cy, 0), g.dim[1] - 1);
    cz = min(max(cz, 0), g.dim[2] - 1);
    return (cz * g.dim[1] + cy) * g.dim[0] + cx;
}

__global__ void grid_count_kernel(const float* __restrict__ verts, int nv, const GridInfo* __restrict__ gi,
                                  int* __restrict__ counts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    GridInfo g = *gi;
    atomicAdd(&counts[vert_cell(g, verts[3 * i], verts[3 * i + 1], verts[3 * i + 2])], 1);
}

// single block exclusive scan counts[0..ncell) -> starts[0..ncell]; cursor := starts
__global__ __launch_bounds__(1024) void grid_scan_kernel(const GridInfo* __restrict__ gi, int* __restrict__ counts,
                                                         int* __restrict__ starts, int* __restrict__ cursor) {
    __shared__ int wsum[16];
    __shared__ int carry;
    int n = gi->ncell;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < n) ? counts[i] : 0;
        int x = v;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(x, o);
            if ((threadIdx.x & 63) >= o) x += t;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
        int excl = carry + woff + x - v;
        if (i < n) { starts[i] = excl; cursor[i] = excl; }
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) starts[n] = carry;
}

__global__ void grid_fill_kernel(const float* __restrict__ verts, int nv, const GridInfo* __restrict__ gi,
                                 int* __restrict__ cursor, float* __restrict__ sorted) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    GridInfo g = *gi;
    float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
    int pos = atomicAdd(&cursor[vert_cell(g, x, y, z)], 1);
    sorted[3 * pos] = x; sorted[3 * pos + 1] = y; sorted[3 * pos + 2] = z;
}

// bounding box of the vertices of every cell (empty cell: inverted box): lets the mask kernel skip a whole cell
// whose content is provably farther than the threshold -- at 6890 vertices a 0.1 m cell that touches the surface
// holds ~40 of them, and a sample just outside the hull used to test all ~400 vertices of its 27 cells
__global__ void grid_bbox_kernel(const GridInfo* __restrict__ gi, const int* __restrict__ starts,
                                 const float* __restrict__ sorted, float* __restrict__ bbox) {
    // 8 lanes per cell (a surface cell holds ~40 vertices: one thread per cell was a 40-deep dependent load chain)
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const bool live = c < gi->ncell;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (live)
        for (int v = starts[c] + sub, e = starts[c + 1]; v < e; v += 8)
            for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], sorted[3 * v + a]); mx[a] = fmaxf(mx[a], sorted[3 * v + a]); }
    for (int o = 1; o < 8; o <<= 1)
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
    if (live && sub == 0)
        for (int a = 0; a < 3; ++a) { bbox[6 * c + a] = mn[a]; bbox[6 * c + 3 + a] = mx[a]; }
}

// Per cell of the grid EXTENDED by one layer (a sample one cell outside still reaches the boundary cells): the 27-bit
// word of its non-empty neighbours, bit j = the cell at offset (j + 13) % 27 of the 3 x 3 x 3 block (bit 0 = the cell
// itself) is inside the grid and holds a vertex.  The mask kernel screens only those; a sample in empty space reads 0.
#define GRID_EXT_CELLS ((GRID_MAX_DIM + 2) * (GRID_MAX_DIM + 2) * (GRID_MAX_DIM + 2))
__device__ __forceinline__ void grid_nbr_phase(const GridInfo& g, const int* __restrict__ starts, unsigned* __restrict__ nbr,
                                               int tid, int nthreads) {
    const int ex = g.dim[0] + 2, ey = g.dim[1] + 2, ez = g.dim[2] + 2;
    for (int e = tid; e < ex * ey * ez; e += nthreads) {
        const int cx = e % ex - 1, cy = (e / ex) % ey - 1, cz = e / (ex * ey) - 1;
        unsigned m = 0u;
        for (int j = 0; j < 27; ++j) {
            const int o = (j + 13) % 27;
            const int X = cx + o % 3 - 1, Y = cy + (o / 3) % 3 - 1, Z = cz + o / 9 - 1;
            if (X >= 0 && X < g.dim[0] && Y >= 0 && Y < g.dim[1] && Z >= 0 && Z < g.dim[2]) {
                const int c = (Z * g.dim[1] + Y) * g.dim[0] + X;
                if (starts[c + 1] > starts[c]) m |= 1u << j;
            }
        }
        nbr[e] = m;
    }
}
__global__ void grid_nbr_kernel(const GridInfo* __restrict__ gi, const int* __restrict__ starts, unsigned* __restrict__ nbr) {
    grid_nbr_phase(*gi, starts, nbr, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The whole grid build as ONE single-workgroup launch (6890 vertices: 7 per thread): AABB -> GridInfo, zeroed counters,
// per-cell counts, exclusive scan, bucket fill, per-cell bounding boxes -- the five launches above cost ~30 us alone
// but ~80 us EACH when they have to squeeze in between the MLP tiles of a concurrent frame (frame pipeline, sharded
// frames).  Also clears the per-ray hit flags and the 16-int info block of the caller (two memset launches less).
// Same arithmetic per phase as the separate kernels; the order of the vertices inside a cell is as arbitrary as before
// (atomics) and does not matter to the predicate.
__global__ __launch_bounds__(1024) void grid_build_kernel(const float* __restrict__ verts, int nv, float h0,
                                                          GridInfo* __restrict__ gi, int* __restrict__ counts,
                                                          int* __restrict__ starts, int* __restrict__ cursor,
                                                          float* __restrict__ sorted, float* __restrict__ bbox,
                                                          int32_t* __restrict__ ray_hit, int R,
                                                          int32_t* __restrict__ info_zero, unsigned* __restrict__ nbr) {
    __shared__ float smin[3][16], smax[3][16];
    __shared__ GridInfo g;
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (info_zero != nullptr && tid < 16) info_zero[tid] = 0;
    if (ray_hit != nullptr)
        for (int i = tid; i < R; i += 1024) ray_hit[i] = 0;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < nv; i += 1024)
        for (int a = 0; a < 3; ++a) {
            float v = verts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
        if ((tid & 63) == 0) {
            smin[a][tid >> 6] = mn[a];
            smax[a][tid >> 6] = mx[a];
        }
    }
    __syncthreads();
    if (tid == 0) {
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) {
            float lo = smin[a][0], hi = smax[a][0];
            for (int w = 1; w < 16; ++w) { lo = fminf(lo, smin[a][w]); hi = fmaxf(hi, smax[a][w]); }
            g.gmin[a] = lo;
            smax[a][0] = hi;
            ext = fmaxf(ext, hi - lo);
        }
        float h = fmaxf(h0, ext / (float)(GRID_MAX_DIM - 2));
        g.inv_h = 1.0f / h;
        int nc = 1;
        for (int a = 0; a < 3; ++a) {
            int d = cell_coord(smax[a][0], g.gmin[a], g.inv_h) + 1;
            d = d < 1 ? 1 : (d > GRID_MAX_DIM ? GRID_MAX_DIM : d);
            g.dim[a] = d;
            nc *= d;
        }
        g.ncell = nc;
        *gi = g;
        carry = 0;
    }
    __syncthreads();
    const int n = g.ncell;
    for (int i = tid; i < n + 1; i += 1024) counts[i] = 0;
    __syncthreads();
    for (int i = tid; i < nv; i += 1024) atomicAdd(&counts[vert_cell(g, verts[3 * i], verts[3 * i + 1], verts[3 * i + 2])], 1);
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {            // exclusive scan, 1024 cells per round
        int i = base + tid;
        int v = (i < n) ? counts[i] : 0;
        int x = v;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(x, o);
            if ((tid & 63) >= o) x += t;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (tid >> 6); ++w) woff += wsum[w];
        int excl = carry + woff + x - v;
        if (i < n) { starts[i] = excl; cursor[i] = excl; }
        __syncthreads();
        if (tid == 1023) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) starts[n] = carry;
    __syncthreads();
    for (int i = tid; i < nv; i += 1024) {
        float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
        int pos = atomicAdd(&cursor[vert_cell(g, x, y, z)], 1);
        sorted[3 * pos] = x; sorted[3 * pos + 1] = y; sorted[3 * pos + 2] = z;
    }
    __syncthreads();
    for (int c = tid >> 3; c < n; c += 128) {                // 8 lanes per cell
        const int sub = tid & 7;
        float bmn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, bmx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int v = starts[c] + sub, e = starts[c + 1]; v < e; v += 8)
            for (int a = 0; a < 3; ++a) { bmn[a] = fminf(bmn[a], sorted[3 * v + a]); bmx[a] = fmaxf(bmx[a], sorted[3 * v + a]); }
        for (int o = 1; o < 8; o <<= 1)
            for (int a = 0; a < 3; ++a) { bmn[a] = fminf(bmn[a], __shfl_xor(bmn[a], o)); bmx[a] = fmaxf(bmx[a], __shfl_xor(bmx[a], o)); }
        if (sub == 0)
            for (int a = 0; a < 3; ++a) { bbox[6 * c + a] = bmn[a]; bbox[6 * c + 3 + a] = bmx[a]; }
    }
    grid_nbr_phase(g, starts, nbr, tid, 1024);               // (starts[] is complete since the barrier after the scan)
}


// one thread per sample; a wave covers 64 consecutive samples (= one ray at S=64).  The first form of the test (kept as the
// A/B partner, TH_HULL_SEQ=1): every lane walks its own candidate vertices -- a wave takes as long as its slowest lane, and
// on a ray through the body that is the one or two samples just OUTSIDE the hull that must test every vertex of every
// cell whose box is in reach (100-300 vertices) while the samples inside leave after a few.
__global__ __launch_bounds__(256) void hull_mask_seq_kernel(ThPointSrc ps, long long P, const GridInfo* __restrict__ gi,
                                                        const int* __restrict__ starts,
                                                        const float* __restrict__ sorted,
                                                        const float* __restrict__ bbox, float thresh,
                                                        uint8_t* __restrict__ mask, int32_t* __restrict__ ray_hit) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= P) return;
    GridInfo g = *gi;
    float px, py, pz;
    th_get_point(ps, i, px, py, pz);
    int cx = cell_coord(px, g.gmin[0], g.inv_h), cy = cell_coord(py, g.gmin[1], g.inv_h),
        cz = cell_coord(pz, g.gmin[2], g.inv_h);
    bool hit = false;
    if (cx >= -1 && cx <= g.dim[0] && cy >= -1 && cy <= g.dim[1] && cz >= -1 && cz <= g.dim[2]) {
        int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
        int y0 = max(cy - 1, 0), y1 = min(cy + 1, g.dim[1] - 1);
        int z0 = max(cz - 1, 0), z1 = min(cz + 1, g.dim[2] - 1);
        // a cell is skipped when the sample is farther than the threshold (+ a rounding margin: never a false skip)
        // from the bounding box of its vertices; the per-vertex predicate is the reference's, sqrt(d2) < thresh
        // (:440-442), evaluated exactly: sqrt is monotone, so d2 below / above a band of +-4e-6 relative around
        // thresh^2 decides without it (fp32 sqrt is correctly rounded: 6e-8) and only the band takes the square root.
        const float rej = thresh * 1.001f + 1e-6f, rej2 = rej * rej;
        const float t2 = thresh * thresh, t2lo = t2 * (1.0f - 4e-6f), t2hi = t2 * (1.0f + 4e-6f);
        auto cell = [&](int c) {
            const float* bb = bbox + 6 * c;
            float ex = fmaxf(fmaxf(bb[0] - px, px - bb[3]), 0.f), ey = fmaxf(fmaxf(bb[1] - py, py - bb[4]), 0.f),
                  ez = fmaxf(fmaxf(bb[2] - pz, pz - bb[5]), 0.f);
            if (!(ex * ex + ey * ey + ez * ez <= rej2)) return;     // (also skips empty cells: inverted box)
            for (int v = starts[c], e = starts[c + 1]; v < e; ++v) {
                float dx = px - sorted[3 * v], dy = py - sorted[3 * v + 1], dz = pz - sorted[3 * v + 2];
                float d2 = dx * dx + dy * dy;
                d2 = d2 + dz * dz;
                if (d2 < t2hi && (d2 < t2lo || __fsqrt_rn(d2) < thresh)) { hit = true; break; }
            }
        };
        // the sample's own cell first: a sample inside the hull (2 of 3 samples that get this far) usually finds its
        // vertex there and leaves after ~40 tests instead of walking up to 13 neighbour cells first
        const bool own = cx >= 0 && cx < g.dim[0] && cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2];
        const int c0 = own ? (cz * g.dim[1] + cy) * g.dim[0] + cx : -1;
        if (own) cell(c0);
        for (int zz = z0; zz <= z1 && !hit; ++zz)
            for (int yy = y0; yy <= y1 && !hit; ++yy) {
                int rowbase = (zz * g.dim[1] + yy) * g.dim[0];
                for (int xx = x0; xx <= x1 && !hit; ++xx)
                    if (rowbase + xx != c0) cell(rowbase + xx);
            }
    }
    mask[i] = hit ? 1 : 0;
    if (hit && ray_hit) ray_hit[(int)(i / ps.S)] = 1;
}

// The default form: the same predicate on the same (sample, vertex) pairs, "exists" evaluated in another order.
// (1) per lane: the 27 cells of the neighbourhood are screened against their bounding boxes -> a 27-bit word (bit 0 = the
// sample's own cell); the first HULL_PROBE vertices of the own cell are tested by the lane itself (two of three samples
// inside the hull leave here).  (2) per wave: the samples still open are taken one at a time by the WHOLE wave -- lane l
// < 27 looks up the vertex range of screened cell l, then the 64 lanes test 64 vertices of a cell per step and the first
// ballot with a set bit ends the sample.  The outcome is bit-identical (the predicate and its operands are unchanged, an
// "or" over the same set); the near-miss samples cost ~5 wave steps instead of ~200 lane iterations.
#define HULL_PROBE 8
__global__ __launch_bounds__(256) void hull_mask_kernel(ThPointSrc ps, long long P, const GridInfo* __restrict__ gi,
                                                        const int* __restrict__ starts,
                                                        const float* __restrict__ sorted,
                                                        const float* __restrict__ bbox, float thresh,
                                                        uint8_t* __restrict__ mask, int32_t* __restrict__ ray_hit,
                                                        const unsigned* __restrict__ nbr) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < P;                           // (no early return: every lane serves the wave stage)
    const GridInfo g = *gi;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (valid) th_get_point(ps, i, px, py, pz);
    const int cx = cell_coord(px, g.gmin[0], g.inv_h), cy = cell_coord(py, g.gmin[1], g.inv_h),
              cz = cell_coord(pz, g.gmin[2], g.inv_h);
    // thresholds as in the sequential form: box rejection with a rounding margin (never a false skip), sqrt only inside
    // a +-4e-6 band around thresh^2
    const float rej = thresh * 1.001f + 1e-6f, rej2 = rej * rej;
    const float t2 = thresh * thresh, t2lo = t2 * (1.0f - 4e-6f), t2hi = t2 * (1.0f + 4e-6f);
    auto within = [&](float x, float y, float z, int v) -> bool {
        float dx = x - sorted[3 * v], dy = y - sorted[3 * v + 1], dz = z - sorted[3 * v + 2];
        float d2 = dx * dx + dy * dy;
        d2 = d2 + dz * dz;
        return d2 < t2hi && (d2 < t2lo || __fsqrt_rn(d2) < thresh);
    };
    unsigned pmask = 0u;                                // bit j: cell at offset (j + 13) % 27 of the 3 x 3 x 3 block is in reach
    bool hit = false;
    if (valid && cx >= -1 && cx <= g.dim[0] && cy >= -1 && cy <= g.dim[1] && cz >= -1 && cz <= g.dim[2]) {
        const unsigned occ = nbr[((cz + 1) * (g.dim[1] + 2) + (cy + 1)) * (g.dim[0] + 2) + (cx + 1)];
        // (cell indices relative to the own cell's; byte offsets as unsigned 32-bit lane offsets from a scalar base)
        const int cb = (cz * g.dim[1] + cy) * g.dim[0] + cx, sy = g.dim[0], sz = g.dim[0] * g.dim[1];
        if (occ != 0u)
#pragma unroll
        for (int j = 0; j < 27; ++j) {
            const int o = (j + 13) % 27;
            if ((occ >> j) & 1u) {                      // (in the grid and not empty)
                const unsigned c = (unsigned)(cb + (o / 9 - 1) * sz + ((o / 3) % 3 - 1) * sy + (o % 3 - 1));
                const float* bb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(bbox) + c * 24u);
                float ex = fmaxf(fmaxf(bb[0] - px, px - bb[3]), 0.f), ey = fmaxf(fmaxf(bb[1] - py, py - bb[4]), 0.f),
                      ez = fmaxf(fmaxf(bb[2] - pz, pz - bb[5]), 0.f);
                if (ex * ex + ey * ey + ez * ez <= rej2) pmask |= 1u << j;
            }
        }
        if (pmask & 1u) {
            const int c0 = (cz * g.dim[1] + cy) * g.dim[0] + cx;
            const int s0 = starts[c0], e0 = min(starts[c0 + 1], s0 + HULL_PROBE);
            for (int v = s0; v < e0; ++v)
                if (within(px, py, pz, v)) { hit = true; break; }
        }
    }
    // ---- wave stage ----
    const int lo = (lane + 13) % 27;                    // (lanes >= 27 hold no cell)
    const int ldx = lo % 3 - 1, ldy = (lo / 3) % 3 - 1, ldz = lo / 9 - 1;
    unsigned long long todo = __ballot(pmask != 0u && !hit);
    while (todo) {
        const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, px), b));
        const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, py), b));
        const float sz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pz), b));
        const int scx = __builtin_amdgcn_readlane(cx, b), scy = __builtin_amdgcn_readlane(cy, b),
                  scz = __builtin_amdgcn_readlane(cz, b);
        const unsigned pm = (unsigned)__builtin_amdgcn_readlane((int)pmask, b);
        int vs = 0, ve = 0;
        if (lane < 27 && ((pm >> lane) & 1u)) {
            const int c = ((scz + ldz) * g.dim[1] + (scy + ldy)) * g.dim[0] + (scx + ldx);
            vs = starts[c];
            ve = starts[c + 1];
            if (lane == 0) vs = min(ve, vs + HULL_PROBE);          // the lane's own probe covered these
        }
        unsigned cells = (unsigned)__ballot(ve > vs);
        bool h = false;
        while (cells && !h) {
            const int l = __builtin_amdgcn_readfirstlane(__ffs((int)cells) - 1);
            cells &= cells - 1;
            const int s0 = __builtin_amdgcn_readlane(vs, l), e0 = __builtin_amdgcn_readlane(ve, l);
            for (int v0 = s0; v0 < e0 && !h; v0 += 64) {
                const int v = v0 + lane;
                const bool t = v < e0 && within(sx, sy, sz, v);
                h = __ballot(t) != 0ull;
            }
        }
        if (h && lane == b) hit = true;
    }
    if (valid) {
        mask[i] = hit ? 1 : 0;
        if (hit && ray_hit) ray_hit[(int)(i / ps.S)] = 1;
    }
}

size_t th_hull_ws(int n_verts) {
    return th_align(sizeof(GridInfo)) + 3 * th_align((GRID_MAX_CELLS + 1) * sizeof(int)) +
           th_align((size_t)n_verts * 3 * sizeof(float)) + th_align((size_t)GRID_MAX_CELLS * 6 * sizeof(float)) +
           th_align((size_t)GRID_EXT_CELLS * sizeof(unsigned));
}

int th_hull_mask_launch(const ThPointSrc& ps, long long P, const float* verts, int nv, float thresh, uint8_t* mask,
                        int32_t* ray_hit, void* ws, size_t ws_bytes, hipStream_t s, int32_t* info_zero) {
    TH_REQUIRE(ws_bytes >= th_hull_ws(nv), "workspace too small");
    ThArena ar(ws, ws_bytes);
    GridInfo* gi = ar.take<GridInfo>(1);
    int* counts = ar.take<int>(GRID_MAX_CELLS + 1);
    int* starts = ar.take<int>(GRID_MAX_CELLS + 1);
    int* cursor = ar.take<int>(GRID_MAX_CELLS + 1);
    float* sorted = ar.take<float>((size_t)nv * 3);
    float* bbox = ar.take<float>((size_t)GRID_MAX_CELLS * 6);
    unsigned* nbr = ar.take<unsigned>(GRID_EXT_CELLS);
    TH_REQUIRE(sorted != nullptr && bbox != nullptr && nbr != nullptr, "workspace carve failed");
    // cell size: thresh plus 5 % so fp rounding of the cell index can never
    // separate a vertex within `thresh` from the 3x3x3 neighbourhood
    float h0 = thresh * 1.05f;
    static const bool split_build = getenv("TH_HULL_SPLIT_BUILD") != nullptr;     // A/B switch: the five-launch build
    if (split_build) {
        hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(1024), 0, s, verts, nv, h0, gi, counts);
        hipLaunchKernelGGL(grid_count_kernel, dim3(th_cdiv(nv, 256)), dim3(256), 0, s, verts, nv, gi, counts);
        hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, s, gi, counts, starts, cursor);
        hipLaunchKernelGGL(grid_fill_kernel, dim3(th_cdiv(nv, 256)), dim3(256), 0, s, verts, nv, gi, cursor, sorted);
        hipLaunchKernelGGL(grid_bbox_kernel, dim3(th_cdiv(GRID_MAX_CELLS * 8, 256)), dim3(256), 0, s, gi, starts, sorted, bbox);
        hipLaunchKernelGGL(grid_nbr_kernel, dim3(64), dim3(256), 0, s, gi, starts, nbr);
        if (ray_hit) TH_HIP(hipMemsetAsync(ray_hit, 0, sizeof(int32_t) * (size_t)ps.R, s));
        if (info_zero) TH_HIP(hipMemsetAsync(info_zero, 0, 16 * sizeof(int32_t), s));
    } else {
        hipLaunchKernelGGL(grid_build_kernel, dim3(1), dim3(1024), 0, s, verts, nv, h0, gi, counts, starts, cursor, sorted, bbox,
                           ray_hit, ray_hit ? ps.R : 0, info_zero, nbr);
    }
    const bool seq = getenv("TH_HULL_SEQ") != nullptr;          // A/B switch, read per launch: one lane per sample throughout
    if (seq)
        hipLaunchKernelGGL(hull_mask_seq_kernel, dim3(th_cdiv(P, 256)), dim3(256), 0, s, ps, P, gi, starts, sorted, bbox, thresh,
                           mask, ray_hit);
    else
        hipLaunchKernelGGL(hull_mask_kernel, dim3(th_cdiv(P, 256)), dim3(256), 0, s, ps, P, gi, starts, sorted, bbox, thresh,
                           mask, ray_hit, nbr);
    TH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// small-frame rule, if_clight_renderer.py:551: when the number of rays that
// touch the hull is <= 2400 the reference takes the un-chunked branch and calls
// the network WITHOUT pts_mask, i.e. every sample of every hit ray is shaded
// (MLP_forward_ori).  Reproduced on device: no host round trip.
// ---------------------------------------------------------------------------
// info[0] (zeroed by the caller) += number of hit rays; integer atomics: the sum is order-independent
__global__ __launch_bounds__(256) void count_hits_kernel(const int32_t* __restrict__ ray_hit, int R,
                                                         int32_t* __restrict__ info) {
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R; i += gridDim.x * blockDim.x) c += ray_hit[i] != 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c != 0) atomicAdd(&info[0], c);
}
// if #hit rays <= thr: mask := ray_hit for every sample of the ray; info[1] = 1 records the mode
__global__ __launch_bounds__(256) void small_frame_apply_kernel(uint8_t* __restrict__ mask,
                                                                const int32_t* __restrict__ ray_hit, long long P, int S,
                                                                int thr, int32_t* __restrict__ info) {
    if (info[0] > thr) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) info[1] = 1;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x)
        mask[i] = ray_hit[(int)(i / S)] ? 1 : 0;
}
int th_small_frame_rule(uint8_t* mask, const int32_t* ray_hit, int R, int S, int thr, int32_t* dev_info,
                        hipStream_t s) {
    // dev_info[0..1] must be zero on entry (shade_points clears the block)
    hipLaunchKernelGGL(count_hits_kernel, dim3(R >= 65536 ? 256 : th_cdiv(R, 256)), dim3(256), 0, s, ray_hit, R, dev_info);
    long long P = (long long)R * S;
    const int nb = (int)(P >= (1LL << 20) ? 1024 : th_cdiv(P, 256));
    hipLaunchKernelGGL(small_frame_apply_kernel, dim3(nb), dim3(256), 0, s, mask, ray_hit, P, S, thr, dev_info);
    TH_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// stream compaction of the mask -> index list (deterministic)
// ---------------------------------------------------------------------------
// Order of the list (round 4): groups of CMP_G = 16 consecutive rays, DEPTH-MAJOR inside a group -- position e of the
// permuted index space maps to ray g * 16 + (e mod 16), sample (e / 16) mod S, g = e / (16 S) -- so consecutive entries
// are the same depth of neighbouring rays (a caller that hands rays over in 8 x 8 pixel tiles, like bench.py and
// dist.shard_ray_indices, gets 8 x 2 pixel blocks) instead of consecutive depths of one ray.  A batch of 32 entries is then
// a compact blob in space: its bilinear footprints in a reference view share texel rows (tools/k5_replay.py: 20 % of
// the corner reads are distinct rows against 51 % ray-major; K5 loads each distinct row once), and its 7-NN sets share
// token centres (K6's tile unions).  Every consumer of the list is order-agnostic (K4 / K5 / K6 work per sample, the raw
// values are scattered back to their dense positions, compositing reads them through the mask).  S = 1 (point lists,
// th_compact_mask) degenerates to the ascending order.
#define CMP_ITEMS 4096   // per block (256 threads x 16)
#define CMP_G 16

// Optional small-frame rule (if_clight_renderer.py:551) evaluated on the fly: when rule.ray_hit != nullptr and the
// number of hit rays info[0] is <= thr, the effective mask of sample i is "ray i / S was hit" (every sample of a hit
// ray is shaded); cmp_write_kernel then also stores that mask and records the mode in info[1].  Same result as the
// separate small_frame_apply_kernel pass in front of a plain compaction, one launch and one 16 MB sweep less.
struct CmpRule {
    const int32_t* ray_hit;
    int32_t* info;
    int S, thr;
    int R;          // rays (points when S == 1)
};
__device__ __forceinline__ bool cmp_rule_on(const CmpRule& r) { return r.ray_hit != nullptr && r.info[0] <= r.thr; }
// a thread's 16 consecutive positions: ray group g, depth s -> dense indices (g * 16 + k) * S + s, k = 0 .. 15
__device__ __forceinline__ void cmp_thread_span(const CmpRule& r, long long e0, long long& ray0, int& s) {
    const long long gs = (long long)CMP_G * r.S;
    const long long g = e0 / gs;
    s = (int)((e0 - g * gs) >> 4);
    ray0 = g * CMP_G;
}

__global__ __launch_bounds__(256) void cmp_count_kernel(const uint8_t* __restrict__ mask, long long P,
                                                        int* __restrict__ bcount, CmpRule rule) {
    __shared__ int ws[4];
    long long e0 = (long long)blockIdx.x * CMP_ITEMS + threadIdx.x * 16;
    long long ray0;
    int sd;
    cmp_thread_span(rule, e0, ray0, sd);
    int c = 0;
    if (cmp_rule_on(rule)) {
        for (int k = 0; k < 16; ++k)
            if (ray0 + k < rule.R) c += rule.ray_hit[ray0 + k] != 0;
    } else if (rule.S == 1 && ray0 + 16 <= rule.R) {
        uint4 v = *reinterpret_cast<const uint4*>(mask + ray0);
        c = __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) +
            __popc(v.w & 0x01010101u);
    } else {
        for (int k = 0; k < 16; ++k) if (ray0 + k < rule.R) c += mask[(ray0 + k) * rule.S + sd] != 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bcount[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void cmp_scan_kernel(int* __restrict__ bcount, int nb, int* __restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < nb) ? bcount[i] : 0;
        int x = v;
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(x, o);
            if ((threadIdx.x & 63) >= o) x += t;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
        int excl = carry + woff + x - v;
        if (i < nb) bcount[i] = excl;
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void cmp_write_kernel(uint8_t* __restrict__ mask, long long P,
                                                        const int* __restrict__ boff, int32_t* __restrict__ idx,
                                                        CmpRule rule) {
    __shared__ int ws[4];
    long long e0 = (long long)blockIdx.x * CMP_ITEMS + threadIdx.x * 16;
    long long ray0;
    int sd;
    cmp_thread_span(rule, e0, ray0, sd);
    uint8_t m[16];
    int c = 0;
    if (cmp_rule_on(rule)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) rule.info[1] = 1;
        for (int k = 0; k < 16; ++k) {
            const bool in = ray0 + k < rule.R;
            m[k] = (in && rule.ray_hit[ray0 + k] != 0) ? 1 : 0;
            if (in) mask[(ray0 + k) * rule.S + sd] = m[k];
            c += m[k] != 0;
        }
    } else
    for (int k = 0; k < 16; ++k) {
        m[k] = (ray0 + k < rule.R) ? mask[(ray0 + k) * rule.S + sd] : 0;
        c += m[k] != 0;
    }
    int x = c;
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(x, o);
        if ((threadIdx.x & 63) >= o) x += t;
    }
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (threadIdx.x >> 6); ++w) woff += ws[w];
    int pos = boff[blockIdx.x] + woff + x - c;
    for (int k = 0; k < 16; ++k)
        if (m[k]) idx[pos++] = (int32_t)((ray0 + k) * rule.S + sd);
}

// blocks over the permuted index space: ceil(R / 16) ray groups x 16 S positions
static long long cmp_space(long long R, int S) { return (R + CMP_G - 1) / CMP_G * CMP_G * S; }
// (P + 16 S covers the padded last ray group of any (R, S) factorisation of P with S <= 1024)
size_t th_compact_ws(long long P) { return th_align((size_t)(th_cdiv(P + 16 * 1024, CMP_ITEMS) + 2) * sizeof(int)); }

static int compact_launch(uint8_t* mask, long long P, int32_t* idx_out, int32_t* dev_count, void* ws, size_t ws_bytes,
                          const CmpRule& rule, hipStream_t s) {
    TH_REQUIRE(P < (1LL << 31), "too many points for int32 indices");
    TH_REQUIRE(rule.S >= 1 && rule.S <= 1024 && (long long)rule.R * rule.S == P, "compaction: P must be R x S, S <= 1024");
    int nb = th_cdiv(cmp_space(rule.R, rule.S), CMP_ITEMS);
    TH_REQUIRE(ws_bytes >= th_compact_ws(P), "workspace too small");
    int* bcount = (int*)ws;
    hipLaunchKernelGGL(cmp_count_kernel, dim3(nb), dim3(256), 0, s, mask, P, bcount, rule);
    hipLaunchKernelGGL(cmp_scan_kernel, dim3(1), dim3(1024), 0, s, bcount, nb, dev_count);
    hipLaunchKernelGGL(cmp_write_kernel, dim3(nb), dim3(256), 0, s, mask, P, bcount, idx_out, rule);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_compact_mask(const uint8_t* mask, long long P, int32_t* idx_out, int32_t* dev_count, void* ws,
                    size_t ws_bytes, hipStream_t s) {
    return compact_launch(const_cast<uint8_t*>(mask), P, idx_out, dev_count, ws, ws_bytes, CmpRule{nullptr, nullptr, 1, 0, (int)P}, s);
}

// hit-ray count -> info[0]; then the compaction with the small-frame rule applied on the fly (info[1] = mode)
int th_compact_mask_rule(uint8_t* mask, long long P, const int32_t* ray_hit, int R, int S, int thr, int32_t* dev_info,
                         int32_t* idx_out, int32_t* dev_count, void* ws, size_t ws_bytes, hipStream_t s) {
    // dev_info[0..1] must be zero on entry
    hipLaunchKernelGGL(count_hits_kernel, dim3(R >= 65536 ? 256 : th_cdiv(R, 256)), dim3(256), 0, s, ray_hit, R, dev_info);
    return compact_launch(mask, P, idx_out, dev_count, ws, ws_bytes, CmpRule{ray_hit, dev_info, S, thr, R}, s);
}
