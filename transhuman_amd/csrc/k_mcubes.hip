// K13: marching cubes over the sigma cube, on device.
//
// if_mesh_renderer.py:99-109: `cube = np.pad(alpha, 10)`, `vertices, triangles = mcubes.marching_cubes(cube,
// cfg.mesh_th)`, vertices -> world coordinates.  PyMCubes is third-party and absent (PARITY UNPINNED against mcubes
// itself); what is restated here is the published algorithm it implements (Lorensen & Cline 1987 with the
// Bourke / Bloyd case table, csrc/mc_tables.h) in PyMCubes' conventions:
//   * cell (i, j, k) has corners v0 (i,j,k) v1 (i+1,j,k) v2 (i+1,j+1,k) v3 (i,j+1,k) v4..v7 the same at k+1;
//     edges e0 v0v1, e1 v1v2, e2 v2v3, e3 v3v0, e4..e7 the same on the k+1 face, e8..e11 the verticals v0v4 .. v3v7;
//   * corner m sets bit m of the case index when value <= iso;
//   * a cut edge carries ONE vertex shared by the (up to four) cells around it, at the linearly interpolated
//     position p + (iso - f(p)) / (f(q) - f(p)) along the edge from its lower grid point p to q (float64, like
//     mc_isovalue_interpolation; the midpoint if f(p) == f(q));
//   * triangles: the table row of the cell's case, three edge numbers at a time.
// Output order (ours, deterministic; PyMCubes' is a property of its traversal, a mesh is a set): vertices by owning
// grid point in row-major (x, y, z) order, at a point the +x edge, then +y, then +z; triangles by cell in the same
// order, inside a cell in table order.
//
// Three passes over the [X][Y][Z] fp32 cube (21 M points at 256^3 + padding: 84 MB read per pass, HBM-bound):
//   count   per grid point: which of its three outgoing edges are cut (3 flag bits) and, for the cell whose lowest
//           corner it is, the number of triangles;
//   scan    exclusive prefix sums of both counts (block sums -> one block -> add: deterministic);
//   emit    vertices (float64 [nv,3], scaled to world) and triangles (int32 [nt,3], global vertex indices looked up
//           through the packed (prefix << 3 | flags) word of the edge's owning point).
// An x-range [x0, x1) restricts the emit pass to the points / cells of a slab (multi-GPU: grid slabs per rank,
// prefix sums are global so the slabs' outputs are disjoint contiguous ranges of the same arrays).
#include "mc_tables.h"
#include "th_internal.h"

__constant__ signed char c_mc_tri[256][16];
__constant__ unsigned char c_mc_ntri[256];

struct McDims { int X, Y, Z; };

__device__ __forceinline__ long long mc_lin(const McDims& d, int x, int y, int z) { return ((long long)x * d.Y + y) * d.Z + z; }

// pass 1: flags (bit a: the edge from this point along +axis a is cut) and triangle count of the cell at this point
__global__ __launch_bounds__(256) void mc_count_kernel(const float* __restrict__ cube, McDims d, float iso,
                                                       int* __restrict__ vcount, int* __restrict__ tcount) {
    const long long n = (long long)d.X * d.Y * d.Z;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int z = (int)(i % d.Z), y = (int)((i / d.Z) % d.Y), x = (int)(i / ((long long)d.Z * d.Y));
    const bool in0 = cube[i] <= iso;
    const bool hx = x + 1 < d.X, hy = y + 1 < d.Y, hz = z + 1 < d.Z;
    int flags = 0;
    if (hx && (cube[i + (long long)d.Y * d.Z] <= iso) != in0) flags |= 1;
    if (hy && (cube[i + d.Z] <= iso) != in0) flags |= 2;
    if (hz && (cube[i + 1] <= iso) != in0) flags |= 4;
    vcount[i] = flags;                       // (bits; the scan adds their population count)
    int nt = 0;
    if (hx && hy && hz) {
        const long long sx = (long long)d.Y * d.Z, sy = d.Z;
        int c = in0 ? 1 : 0;
        c |= (cube[i + sx] <= iso) ? 2 : 0;
        c |= (cube[i + sx + sy] <= iso) ? 4 : 0;
        c |= (cube[i + sy] <= iso) ? 8 : 0;
        c |= (cube[i + 1] <= iso) ? 16 : 0;
        c |= (cube[i + sx + 1] <= iso) ? 32 : 0;
        c |= (cube[i + sx + sy + 1] <= iso) ? 64 : 0;
        c |= (cube[i + sy + 1] <= iso) ? 128 : 0;
        nt = c_mc_ntri[c];
    }
    tcount[i] = nt;
}

// pass 2: exclusive scans.  1024 elements per block; `pop`: the element value is a 3-bit flag word whose population
// count is scanned and the result is packed as (prefix << 3) | flags.
#define MC_SCAN_B 1024
template <bool POP>
__global__ __launch_bounds__(256) void mc_scan_block_kernel(int* __restrict__ a, long long n, long long* __restrict__ sums) {
    __shared__ int ws[4];
    const long long base = (long long)blockIdx.x * MC_SCAN_B + 4 * threadIdx.x;
    int v[4], raw[4];
    int t = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        raw[k] = base + k < n ? a[base + k] : 0;
        v[k] = POP ? __popc(raw[k]) : raw[k];
        t += v[k];
    }
    // inclusive scan of the per-thread totals: wave shuffles, then the four wave sums
    int inc = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) ws[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += ws[w];
    int run = woff + inc - t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) a[base + k] = POP ? ((run << 3) | raw[k]) : run;
        run += v[k];
    }
    if (threadIdx.x == 255) sums[blockIdx.x] = (long long)run;
}
// one block: exclusive scan of the block sums in place; total -> sums[nb]
__global__ __launch_bounds__(1024) void mc_scan_sums_kernel(long long* __restrict__ sums, int nb) {
    __shared__ long long part[1024];
    const int per = (nb + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(nb, lo + per);
    long long t = 0;
    for (int i = lo; i < hi; ++i) t += sums[i];
    part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 1024; ++i) { const long long v = part[i]; part[i] = run; run += v; }
        sums[nb] = run;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const long long v = sums[i]; sums[i] = run; run += v; }
}
template <bool POP>
__global__ __launch_bounds__(256) void mc_scan_add_kernel(int* __restrict__ a, long long n, const long long* __restrict__ sums) {
    const long long base = (long long)blockIdx.x * MC_SCAN_B + 4 * threadIdx.x;
    const int off = (int)sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) a[base + k] += POP ? (off << 3) : off;
}

// pass 3a: vertices of the points of the slab
__global__ __launch_bounds__(256) void mc_emit_verts_kernel(const float* __restrict__ cube, McDims d, float iso,
                                                            const int* __restrict__ vpack, int x0, int x1, double sx_,
                                                            double sy_, double sz_, double ox, double oy, double oz,
                                                            double* __restrict__ verts) {
    const long long plane = (long long)d.Y * d.Z;
    const long long i = (long long)x0 * plane + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)x1 * plane) return;
    const int pk = vpack[i];
    const int flags = pk & 7;
    if (!flags) return;
    const int z = (int)(i % d.Z), y = (int)((i / d.Z) % d.Y), x = (int)(i / plane);
    long long vi = pk >> 3;
    const double f0 = (double)cube[i];
    const long long step[3] = {plane, (long long)d.Z, 1};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!(flags & (1 << a))) continue;
        const double f1 = (double)cube[i + step[a]];
        const double t = f1 == f0 ? 0.5 : ((double)iso - f0) / (f1 - f0);
        double p[3] = {(double)x, (double)y, (double)z};
        p[a] = p[a] + t;
        verts[3 * vi] = p[0] * sx_ + ox;
        verts[3 * vi + 1] = p[1] * sy_ + oy;
        verts[3 * vi + 2] = p[2] * sz_ + oz;
        ++vi;
    }
}

// pass 3b: triangles of the cells of the slab.  Edge e of cell (x,y,z) is owned by point (x,y,z) + MC_EP[e] along axis MC_EA[e].
__constant__ signed char c_mc_ep[12][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 0}, {0, 0, 1}, {1, 0, 1},
                                           {0, 1, 1}, {0, 0, 1}, {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}};
__constant__ signed char c_mc_ea[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};

__global__ __launch_bounds__(256) void mc_emit_tris_kernel(const float* __restrict__ cube, McDims d, float iso,
                                                           const int* __restrict__ vpack, const int* __restrict__ tbase,
                                                           int x0, int x1, int* __restrict__ tris) {
    const long long plane = (long long)d.Y * d.Z;
    const long long i = (long long)x0 * plane + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)x1 * plane) return;
    const int z = (int)(i % d.Z), y = (int)((i / d.Z) % d.Y), x = (int)(i / plane);
    if (x + 1 >= d.X || y + 1 >= d.Y || z + 1 >= d.Z) return;
    const long long sx = plane, sy = d.Z;
    int c = (cube[i] <= iso) ? 1 : 0;
    c |= (cube[i + sx] <= iso) ? 2 : 0;
    c |= (cube[i + sx + sy] <= iso) ? 4 : 0;
    c |= (cube[i + sy] <= iso) ? 8 : 0;
    c |= (cube[i + 1] <= iso) ? 16 : 0;
    c |= (cube[i + sx + 1] <= iso) ? 32 : 0;
    c |= (cube[i + sx + sy + 1] <= iso) ? 64 : 0;
    c |= (cube[i + sy + 1] <= iso) ? 128 : 0;
    const int nt = c_mc_ntri[c];
    if (nt == 0) return;
    long long t = tbase[i];
    for (int k = 0; k < 3 * nt; ++k) {
        const int e = c_mc_tri[c][k];
        const int a = c_mc_ea[e];
        const int pk = vpack[mc_lin(d, x + c_mc_ep[e][0], y + c_mc_ep[e][1], z + c_mc_ep[e][2])];
        const int fl = pk & 7;
        // rank of axis a among the owner's cut edges
        const int r = __popc(fl & ((1 << a) - 1));
        tris[3 * t + k] = (pk >> 3) + r;
    }
}

static int mc_tables_upload() {
    static bool done = false;
    if (done) return 0;
    TH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_mc_tri), MC_TRI_TABLE, sizeof(MC_TRI_TABLE)));
    TH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_mc_ntri), MC_NUM_TRIS, sizeof(MC_NUM_TRIS)));
    done = true;
    return 0;
}

size_t th_mc_ws(int X, int Y, int Z) {
    const size_t n = (size_t)X * Y * Z;
    const size_t nb = (n + MC_SCAN_B - 1) / MC_SCAN_B;
    return 2 * th_align(n * sizeof(int)) + 2 * th_align((nb + 2) * sizeof(long long));
}

struct McWs { int *vpack, *tbase; long long *vsums, *tsums; size_t nb; };
static McWs mc_carve(void* ws, int X, int Y, int Z) {
    const size_t n = (size_t)X * Y * Z;
    McWs w;
    w.nb = (n + MC_SCAN_B - 1) / MC_SCAN_B;
    char* p = (char*)ws;
    w.vpack = (int*)p; p += th_align(n * sizeof(int));
    w.tbase = (int*)p; p += th_align(n * sizeof(int));
    w.vsums = (long long*)p; p += th_align((w.nb + 2) * sizeof(long long));
    w.tsums = (long long*)p;
    return w;
}

// passes 1 + 2; totals (vertices, triangles) -> counts_dev[0..1]
int th_mc_count_launch(const float* cube, int X, int Y, int Z, float iso, void* ws, size_t ws_bytes, long long* counts_dev,
                       hipStream_t s) {
    TH_REQUIRE(ws_bytes >= th_mc_ws(X, Y, Z), "workspace too small");
    TH_REQUIRE(X >= 2 && Y >= 2 && Z >= 2, "marching cubes needs at least 2 grid points per axis");
    TH_REQUIRE((long long)X * Y * Z <= 80000000LL, "grid too large for the packed 28-bit vertex prefix (80 M points)");
    TH_TRY(mc_tables_upload());
    McWs w = mc_carve(ws, X, Y, Z);
    const long long n = (long long)X * Y * Z;
    McDims d{X, Y, Z};
    hipLaunchKernelGGL(mc_count_kernel, dim3(th_cdiv(n, 256)), dim3(256), 0, s, cube, d, iso, w.vpack, w.tbase);
    hipLaunchKernelGGL(mc_scan_block_kernel<true>, dim3((unsigned)w.nb), dim3(256), 0, s, w.vpack, n, w.vsums);
    hipLaunchKernelGGL(mc_scan_block_kernel<false>, dim3((unsigned)w.nb), dim3(256), 0, s, w.tbase, n, w.tsums);
    hipLaunchKernelGGL(mc_scan_sums_kernel, dim3(1), dim3(1024), 0, s, w.vsums, (int)w.nb);
    hipLaunchKernelGGL(mc_scan_sums_kernel, dim3(1), dim3(1024), 0, s, w.tsums, (int)w.nb);
    hipLaunchKernelGGL(mc_scan_add_kernel<true>, dim3((unsigned)w.nb), dim3(256), 0, s, w.vpack, n, w.vsums);
    hipLaunchKernelGGL(mc_scan_add_kernel<false>, dim3((unsigned)w.nb), dim3(256), 0, s, w.tbase, n, w.tsums);
    TH_LAUNCH_CHECK();
    TH_HIP(hipMemcpyAsync(counts_dev, w.vsums + w.nb, sizeof(long long), hipMemcpyDeviceToDevice, s));
    TH_HIP(hipMemcpyAsync(counts_dev + 1, w.tsums + w.nb, sizeof(long long), hipMemcpyDeviceToDevice, s));
    return 0;
}

int th_mc_emit_launch(const float* cube, int X, int Y, int Z, float iso, const void* ws, int x0, int x1, const double* scale,
                      const double* origin, double* verts, int* tris, hipStream_t s) {
    McWs w = mc_carve(const_cast<void*>(ws), X, Y, Z);
    McDims d{X, Y, Z};
    x0 = x0 < 0 ? 0 : x0;
    x1 = x1 > X ? X : x1;
    if (x1 <= x0) return 0;
    const long long m = (long long)(x1 - x0) * Y * Z;
    hipLaunchKernelGGL(mc_emit_verts_kernel, dim3(th_cdiv(m, 256)), dim3(256), 0, s, cube, d, iso, w.vpack, x0, x1, scale[0],
                       scale[1], scale[2], origin[0], origin[1], origin[2], verts);
    hipLaunchKernelGGL(mc_emit_tris_kernel, dim3(th_cdiv(m, 256)), dim3(256), 0, s, cube, d, iso, w.vpack, w.tbase, x0, x1, tris);
    TH_LAUNCH_CHECK();
    return 0;
}

// (vertex, triangle) prefix at the first point of plane x (x == X: the totals): the output range of a slab
__global__ void mc_prefix_at_kernel(const int* __restrict__ vpack, const int* __restrict__ tbase, long long i, long long n,
                                    const long long* __restrict__ vtot, const long long* __restrict__ ttot,
                                    long long* __restrict__ out) {
    out[0] = i < n ? (long long)(vpack[i] >> 3) : *vtot;
    out[1] = i < n ? (long long)tbase[i] : *ttot;
}
int th_mc_prefix_launch(const void* ws, int X, int Y, int Z, int x, long long* out_dev, hipStream_t s) {
    McWs w = mc_carve(const_cast<void*>(ws), X, Y, Z);
    const long long n = (long long)X * Y * Z;
    hipLaunchKernelGGL(mc_prefix_at_kernel, dim3(1), dim3(1), 0, s, w.vpack, w.tbase, (long long)x * Y * Z, n, w.vsums + w.nb,
                       w.tsums + w.nb, out_dev);
    TH_LAUNCH_CHECK();
    return 0;
}
