// NOT PART OF THE BUILD.  Round-6 experiment, kept as the record of what was measured (LOG.md):
//   * built in one sitting on top of mlp_fused8_kernel's helpers, 255 VGPRs, no scratch, 77.7 KB of LDS: two workgroups per CU;
//   * token branch, pixel rows and the token keys bit-identical with the 8-wave kernel (phase dumps: FH_DUMP_AT / FH_DUMPF_AT /
//     FH_DUMPSUM_AT, macros that lived in k_mlp_fused8_kernel.h while this was debugged); the PIXEL KEYS (kv0) come out wrong on
//     0.6 % of the rows -- only when two workgroups share a CU (TH_FH_LDS=83000, one per CU: bit-identical), the operand rows
//     intact before and after the product, the ISA around the barriers in order: not found when it was stopped;
//   * 101 k cycles per HALF tile with two workgroups on the CU at 1.95 GHz = 51.9 us per 32 samples against the 8-wave kernel's
//     96.8 k at 2.04 GHz = 47.5 us; frame 15.3 ms against 13.8 ms on the same box.  Why: (a) the two co-resident workgroups ran
//     their GEMM phases TOGETHER (kv1 15.1 k cycles for half the rows = the 8-wave kernel's time for all of them): equal phase
//     lengths keep two tiles that start together in step, nothing anti-aligns them; (b) the fillings straight from L2 took
//     13.7 k + 10.3 k cycles per half tile (8-wave kernel, staged through LDS: 9.1 k + 7.5 k per whole tile): a scalar
//     look-up chain per texel row and two batches in flight do not cover two L2 round trips.
// To try again: anti-align by construction (the second workgroup of a CU starts with the pixel branch AND half a GEMM later), and
// give K5t a 48-row pass budget (its `cap` argument) so that a half tile stages its own list like the 8-wave kernel does.
// To build it: copy next to k_mlp_fused8_kernel.h, include it from k_mlp_fused_host.hip and launch mlp_fusedh_kernel<V> on
// 8 * ceil(2 * tiles / 8) workgroups of 256 threads with FH_LDS_BYTES of dynamic LDS.
//
// K6 "h" (round 6, experimental, TH_FUSED_HALF=1): the fused per-point MLP on HALF tiles -- 16 samples x V views per workgroup of
// 256 threads, 78 KB of LDS and <= 256 registers per wave, so that TWO workgroups share a CU and a SIMD holds two waves that are
// in DIFFERENT phases of two tiles (one in a GEMM segment while the other fills, blends or splits): what mlp_fused8_kernel's two
// waves per SIMD cannot do, because a barrier keeps them in the same phase.
//
// Same arithmetic, weight images (the 16-form of k_mlp_fused_host.hip: wave w of 4 runs the slices of "virtual" waves 2 w and
// 2 w + 1 of the 8-wave kernel, one after the other) and hand-overs as mlp_fused8_kernel:
//   * K4's records are per sample; the tile header (slot -> centre) is the 32-sample tile's, the T' rows are staged 16 slots x V
//     views per pass (the K = 32 blend GEMM runs half empty);
//   * K5t's records are per sample and view; the texel rows are NOT staged: a row of the operand is blended straight out of its
//     four texel rows in L2 (the row numbers are looked up in the tile's list through the record's row offsets) -- a half tile
//     has no room for the 32-sample tile's list, and needs no pass logic this way.
// Row r of a plane = view (r >> 4), sample (r & 15) of the half.
#pragma once
#include "k_mlp_fused8_kernel.h"

#define FH_THREADS 256
#define FH_ABUF_BYTES (2 * 48 * STR256)                    // operand planes hi | lo (>= 48 rows x 1040 B: T' rows, fp32 RGB rows)
#define FH_MBUF_BYTES (48 * KSTR * 4)                      // fp32 keys | means | small planes
#define FH_PSTR 12
#define FH_PART_FLOATS (3 * 4 * 16)                        // [3 outputs][4 waves][16 samples]
#define FH_MISC_FLOATS (FH_PSTR * 16 + FH_PART_FLOATS + 16 + 8)   // probs | part | sig | flag + cycle stamps
#define FH_LDS_BYTES (FH_ABUF_BYTES + FH_MBUF_BYTES + FH_MISC_FLOATS * 4)
#define FH_MBUF_FC4_OFF 8192
#define FH_MBUF_W_OFF 8192
static_assert(2 * FH_LDS_BYTES <= 163840, "two half-tile workgroups must fit the CU's LDS");
static_assert(48 * 1040 <= FH_ABUF_BYTES, "fp32 rows must fit the operand buffer");

// acc[C0 + c][r] (+)= W(c) X(r)^T for one k-step: three fp16 products, term-major
template <int CT, int RT, int NC, int C0, bool FIRST>
__device__ __forceinline__ void fh_mfma(const uint4 (&w)[CT][2], const h8 (&xh)[RT], const h8 (&xl)[RT], f8_f4 (&acc)[NC][RT]) {
    const f8_f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[C0 + c][r] = F8_MFMA(*reinterpret_cast<const h8*>(&w[c][1]), xh[r], FIRST ? zero : acc[C0 + c][r]);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[C0 + c][r] = F8_MFMA(*reinterpret_cast<const h8*>(&w[c][0]), xl[r], acc[C0 + c][r]);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[C0 + c][r] = F8_MFMA(*reinterpret_cast<const h8*>(&w[c][0]), xh[r], acc[C0 + c][r]);
}

// acc[C0 ..][.] = W A^T over T k-steps, all row tiles at once (RT = V), weights D steps deep in a register ring, activations
// double-buffered.  (Latency that this leaves exposed is the OTHER workgroup's issue time: two tiles share the SIMD.)
template <int CT, int RT, int NC, int C0, int T, int ROWSTEP, bool PERM, int D>
__device__ __forceinline__ void fh_gemm(const char* __restrict__ ahi, const char* __restrict__ alo, int str,
                                        const uint4* __restrict__ wp, int lane, f8_f4 (&acc)[NC][RT]) {
    const uint4* wl = wp + lane;
    const int aoff = f8_aoff<PERM>(lane, str);
    uint4 w[D][CT][2];
    h8 xh[2][RT], xl[2][RT];
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < T) f8_load_w<CT>(wl, d, w[d]);
    f8_load_x<RT, ROWSTEP, PERM>(ahi, alo, aoff, 0, xh[0], xl[0]);
    FM_SB();
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (t + D - 1 < T) f8_load_w<CT>(wl, t + D - 1, w[(t + D - 1) % D]);
        if (t + 1 < T) f8_load_x<RT, ROWSTEP, PERM>(ahi, alo, aoff, t + 1, xh[(t + 1) & 1], xl[(t + 1) & 1]);
        if (t == 0) fh_mfma<CT, RT, NC, C0, true>(w[0], xh[0], xl[0], acc);
        else fh_mfma<CT, RT, NC, C0, false>(w[t % D], xh[t & 1], xl[t & 1], acc);
        FM_SB();
    }
}

template <int V>
__global__ __launch_bounds__(FH_THREADS, 2) void mlp_fusedh_kernel(FusedParams P_arg) {
#define PK P_arg
    struct F8Dbg { long long* dbg; };
    const F8Dbg P{PK.dbg};
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* abuf = lds;
    char* mbuf = lds + FH_ABUF_BYTES;
    float* misc = reinterpret_cast<float*>(lds + FH_ABUF_BYTES + FH_MBUF_BYTES);
    float* probs = misc;                       // [16][FH_PSTR]
    float* part = misc + FH_PSTR * 16;         // [3][4 waves][16]
    float* sig = part + FH_PART_FLOATS;        // [16]
    int* flag = reinterpret_cast<int*>(sig + 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    // XCD-contiguous order of the HALF tiles: the two halves of a tile are neighbours on one XCD (they share the tile's lists)
    const int ht = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int tile = ht >> 1, half = ht & 1;
    const int pbase = tile * FM_PTS + 16 * half;
    if (pbase >= PK.P) return;
    const int npts = min(16, PK.P - pbase);
    const int fm_dbg_tile = ht;
    constexpr int ROWS = 16 * V, RT = V;
    constexpr int RS256 = 16 * STR256;
    int dbg_i = 1;
    long long dbg_t = 0;
    long long* dbg_keep = reinterpret_cast<long long*>(flag + 2);
    constexpr float inv_v = 1.0f / (float)V;
    unsigned rmax = 0u;
    unsigned seen_s = 0u, seen_p = 0u, seen_n = 0u, seen_i = 0u, seen_4 = 0u;
    char* a256_lo = abuf + ROWS * STR256;
    const int wv = __builtin_amdgcn_readfirstlane(wave);

    // ---- texel hand-over: the four pass headers of the 32-sample tile (which of them hold this half's lists depends on the
    // tile's pass count, known with the first) and this half's 16 V row records
    struct TexPre { unsigned h[4][2]; fm_u4 rq; };
    auto tex_fetch = [&]() __attribute__((always_inline)) {
        int tl = tile;
        asm volatile("" : "+s"(tl));
        TexPre t;
        const unsigned* hb = PK.tex_hdr + (long long)tl * 512;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            t.h[p][0] = hb[p * 128 + lane];
            t.h[p][1] = hb[p * 128 + 64 + lane];
        }
        t.rq = (fm_u4){0u, 0u, 0u, 0u};
        if (tid < 32 * V) {
            const int row = tid >> 1, v = row >> 4, s = row & 15;
            t.rq = *reinterpret_cast<const fm_u4*>(PK.tex_rec + ((long long)(tl * V + v) * 32 + 16 * half + s) * 8 + (tid & 1) * 4);
        }
        return t;
    };
    // p = blend of 4 texel rows per operand row, straight from the folded map (L2): wave w blends rows w, w + 4, ...: row
    // w + 4 k is view k >> 2, sample w + 4 (k & 3); batches of four rows (one view), two batches in flight
    auto fill_tex = [&](const TexPre& pre, auto rgb, auto&& under) __attribute__((always_inline)) {
        constexpr bool RGB = decltype(rgb)::value;
        char* recl = reinterpret_cast<char*>(misc);
        static_assert(16 * V * 32 <= (FH_PSTR * 16 + FH_PART_FLOATS) * 4, "row records must fit probs + part");
        const int np = __builtin_amdgcn_readfirstlane((int)(pre.h[0][0] >> 16));
        // list of the samples' quarter q (0 .. 3 of the 32-sample tile): pass q >> (2 - log2 np)
        const int sh = np == 1 ? 2 : np == 2 ? 1 : 0;
        const int pa = (2 * half) >> sh, pb = (2 * half + 1) >> sh;
        unsigned ha0 = pre.h[0][0], ha1 = pre.h[0][1], hb0 = pre.h[0][0], hb1 = pre.h[0][1];
#pragma unroll
        for (int p = 1; p < 4; ++p) {
            if (pa == p) { ha0 = pre.h[p][0]; ha1 = pre.h[p][1]; }
            if (pb == p) { hb0 = pre.h[p][0]; hb1 = pre.h[p][1]; }
        }
        f32x2 bias_lo = {0.f, 0.f}, bias_hi = {0.f, 0.f};
        if constexpr (!RGB) {
            const float4 b4 = *reinterpret_cast<const float4*>(PK.ar0.bias + 4 * lane);
            bias_lo = (f32x2){b4.x, b4.y};
            bias_hi = (f32x2){b4.z, b4.w};
        }
        if (tid < 32 * V) *reinterpret_cast<fm_u4*>(recl + tid * 16) = pre.rq;
        FM_SYNCL();                                          // the records are in place (and every wave is done with ABUF)
        const char* mbase = reinterpret_cast<const char*>(RGB ? PK.tex_map2 : PK.tex_map);
        const unsigned loff = (unsigned)lane * 16u;
        struct RowIn { float4 a, b, c, d; };
        auto issue = [&](int k, RowIn& r) __attribute__((always_inline)) {
            const fm_u4 o = *reinterpret_cast<const fm_u4*>(recl + (wv + 4 * k) * 32 + 16);
            const bool second = ((k & 3) >> 1) != 0;         // sample w + 4 (k & 3) >= 8: the half's second quarter
            const unsigned h0 = second ? hb0 : ha0, h1 = second ? hb1 : ha1;
            auto row = [&](unsigned off) __attribute__((always_inline)) {
                const int wi = __builtin_amdgcn_readfirstlane((int)(8u + off / 1040u));
                const unsigned id = wi < 64 ? (unsigned)__builtin_amdgcn_readlane((int)h0, wi)
                                            : (unsigned)__builtin_amdgcn_readlane((int)h1, wi - 64);
                return *reinterpret_cast<const float4*>(mbase + TX_ADDR(id));
            };
            r.a = row(o[0]);
            r.b = row(o[1]);
            r.c = row(o[2]);
            r.d = row(o[3]);
        };
        auto blend = [&](int k, const RowIn& r) __attribute__((always_inline)) {
            const fm_u4 q0 = *reinterpret_cast<const fm_u4*>(recl + (wv + 4 * k) * 32);
            // (the elements are copied out first: __builtin_bit_cast(float, q0[i]) on the vector's element lvalue compiled into
            // element 0 four times)
            const unsigned u0 = q0[0], u1 = q0[1], u2 = q0[2], u3 = q0[3];
            const float w00 = __builtin_bit_cast(float, u0), w01 = __builtin_bit_cast(float, u1),
                        w10 = __builtin_bit_cast(float, u2), w11 = __builtin_bit_cast(float, u3);
            // (pg_blend2 of k_pixfeat.hip: a w00, then fused multiply-adds in the order ne, sw, se)
            const f32x2 W00 = {w00, w00}, W01 = {w01, w01}, W10 = {w10, w10}, W11 = {w11, w11};
            f32x2 lo = (f32x2){r.a.x, r.a.y} * W00, hi = (f32x2){r.a.z, r.a.w} * W00;
            lo = __builtin_elementwise_fma((f32x2){r.b.x, r.b.y}, W01, lo);
            hi = __builtin_elementwise_fma((f32x2){r.b.z, r.b.w}, W01, hi);
            lo = __builtin_elementwise_fma((f32x2){r.c.x, r.c.y}, W10, lo);
            hi = __builtin_elementwise_fma((f32x2){r.c.z, r.c.w}, W10, hi);
            lo = __builtin_elementwise_fma((f32x2){r.d.x, r.d.y}, W11, lo);
            hi = __builtin_elementwise_fma((f32x2){r.d.z, r.d.w}, W11, hi);
            const int row = wv + 4 * k;
            if constexpr (RGB) {
                *reinterpret_cast<float4*>(abuf + row * 1040 + lane * 16) = make_float4(lo[0], lo[1], hi[0], hi[1]);
            } else {
                lo = __builtin_elementwise_max(lo + bias_lo, (f32x2){0.f, 0.f});
                hi = __builtin_elementwise_max(hi + bias_hi, (f32x2){0.f, 0.f});
                unsigned n0, n1, n2, n3;
                split_pair(lo[0], lo[1], n0, n2);
                split_pair(hi[0], hi[1], n1, n3);
                range_acc<true>(rmax, n0);
                range_acc<true>(rmax, n1);
                *reinterpret_cast<uint2*>(abuf + row * STR256 + lane * 8) = make_uint2(n0, n1);
                *reinterpret_cast<uint2*>(a256_lo + row * STR256 + lane * 8) = make_uint2(n2, n3);
            }
        };
        RowIn in[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) issue(q, in[0][q]);
        under();
#pragma unroll
        for (int b = 0; b < V; ++b) {
            if (b + 1 < V) {
#pragma unroll
                for (int q = 0; q < 4; ++q) issue(4 * (b + 1) + q, in[(b + 1) & 1][q]);
            }
            FM_SB();
#pragma unroll
            for (int q = 0; q < 4; ++q) blend(4 * b + q, in[b & 1][q]);
            FM_SB();
        }
        if constexpr (!RGB) range_commit(PK.range, TH_RANGE_P, seen_p, rmax);
    };

    if (PK.dbg != nullptr && tid == 0 && (ht & 15) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(PK.dbg), 1ull);
        dbg_t = clock64();
        dbg_keep[0] = dbg_t;
        dbg_keep[1] = wall_clock64();
    }
    fm_u4 rv4 = {0u, 0u, 0u, 0u};
    unsigned rv1 = 0u;
    if (PK.range != nullptr) {
        const unsigned* rtab = PK.range;
        rv4 = (fm_u4){rtab[1], rtab[2], rtab[3], rtab[4]};
        rv1 = rtab[5];
    }
    TexPre tex_pre, tex_pre2;

    // ================= token branch: s = relu(fc_0 h); ks|vs = kv1(s) =================
    int vsel[2] = {0, 0};
    float vdv[2] = {0.f, 0.f};
    f8_f4 acc2[4][RT];                                        // [virtual wave x column tile][view]
    {
        constexpr int STOK_STR = 1040;
        static_assert(16 * V * STOK_STR <= FH_ABUF_BYTES, "T' rows must fit the operand buffer");
        char* pe_hi = mbuf;
        char* pe_lo = mbuf + 16 * STR64;
        // pe: 16 rows x (64 hi | 64 lo halves): one 16-byte piece per thread
        const int ppl = tid >> 7, pt = tid & 127, prow = pt >> 3, pc = pt & 7;
        const int psrc = min(prow, npts - 1);
        const uint4 pe_v = *reinterpret_cast<const uint4*>(PK.pe + (long long)(pbase + psrc) * 128 + 64 * ppl + 8 * pc);
        char* wsp_hi = mbuf + FH_MBUF_W_OFF;                               // W [sample][slot] halves, K = 32 (16 used) per pass
        char* wsp_lo = wsp_hi + 16 * STRVD;
        const unsigned* hdr = reinterpret_cast<const unsigned*>(PK.stok) + (long long)((PK.P + 31) / 32 * 32) * 16 + (long long)tile * 128;
        const unsigned h0 = hdr[lane], h1 = hdr[64 + lane];
        const int ns = tid / 7, nk = tid - 7 * ns;
        int slot = -1;
        float nw = 0.f;
        if (tid < 112) {
            const unsigned* rec = reinterpret_cast<const unsigned*>(PK.stok) + (long long)(pbase + min(ns, npts - 1)) * 16;
            slot = (int)rec[nk];
            nw = __builtin_bit_cast(float, rec[8 + nk]);
        }
        const unsigned zq = 0u;
        const float inv_t = PK.t_inv[0];
        uint4 wq[2][2][2][2];                                              // [virtual wave][k-step][column tile][plane]
        float4 b0[4];
#pragma unroll
        for (int vw = 0; vw < 2; ++vw) {
            const uint4* wl = F8_WSLICE(PK.w16.fc_0pe, 2 * wave + vw, 2) + lane;
            f8_load_w<2>(wl, 0, wq[vw][0]);
            f8_load_w<2>(wl, 1, wq[vw][1]);
            b0[2 * vw] = f8_bias(PK.fc_0pe.bias, (2 * wave + vw) * 32, lane);
            b0[2 * vw + 1] = f8_bias(PK.fc_0pe.bias, (2 * wave + vw) * 32 + 16, lane);
        }
        if (tid < 2 * 16 * STRVD / 16) reinterpret_cast<uint4*>(wsp_hi)[tid] = make_uint4(zq, zq, zq, zq);
        *reinterpret_cast<uint4*>((ppl ? pe_lo : pe_hi) + prow * STR64 + 16 * pc) = pe_v;
        const int U = __builtin_amdgcn_readfirstlane((int)h0);
        FM_SB();
        auto slot_centre = [&](int u) {
            const int d = (2 + u) >> 1;
            const unsigned src = d < 64 ? (unsigned)__builtin_amdgcn_readlane((int)h0, d) : (unsigned)__builtin_amdgcn_readlane((int)h1, d - 64);
            return (int)((src >> (16 * ((2 + u) & 1))) & 0xffffu);
        };
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc2[c][r] = (f8_f4){0.f, 0.f, 0.f, 0.f};
        f8_f4 a1[4][1];
#pragma unroll
        for (int c = 0; c < 4; ++c) a1[c][0] = (f8_f4){0.f, 0.f, 0.f, 0.f};
        const int aoffw = f8_aoff<false>(lane, STRVD), aoffp = f8_aoff<false>(lane, STR64);
        for (int u0 = 0; u0 < U; u0 += 16) {
            const int nU = min(16, U - u0);
            if (u0 > 0) {
                FM_SYNCL();
                if (tid < 2 * 16 * STRVD / 16) reinterpret_cast<uint4*>(wsp_hi)[tid] = make_uint4(zq, zq, zq, zq);
            }
            for (int u = wv; u < nU; u += 4) {
                const int cu = slot_centre(u0 + u);
                const char* g = reinterpret_cast<const char*>(PK.tsplit) + (long long)cu * 1024 + lane * 16;
#pragma unroll
                for (int vv = 0; vv < V; ++vv)
                    __builtin_amdgcn_global_load_lds((fm_gptr)(g + (long long)vv * PK.t_nc * 1024),
                                                     (fm_lptr)(abuf + (vv * 16 + u) * STOK_STR), 16, 0, 0);
            }
            FM_SYNCL();                                          // W is cleared (and the pe rows are in place)
            if (slot >= u0 && slot < u0 + 16) {
                _Float16 hi, lo;
                split_h(nw, hi, lo);
                *reinterpret_cast<_Float16*>(wsp_hi + ns * STRVD + 2 * (slot - u0)) = hi;
                *reinterpret_cast<_Float16*>(wsp_lo + ns * STRVD + 2 * (slot - u0)) = lo;
            }
            if (u0 == 0) {                                       // W_pe pe under the row loads
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    h8 xh[1], xl[1];
                    f8_load_x<1, 16 * STR64, false>(pe_hi, pe_lo, aoffp, t, xh, xl);
                    if (t == 0) {
                        fh_mfma<2, 1, 4, 0, true>(wq[0][0], xh, xl, a1);
                        fh_mfma<2, 1, 4, 2, true>(wq[1][0], xh, xl, a1);
                    } else {
                        fh_mfma<2, 1, 4, 0, false>(wq[0][1], xh, xl, a1);
                        fh_mfma<2, 1, 4, 2, false>(wq[1][1], xh, xl, a1);
                    }
                }
            }
            FM_SYNC();                                           // rows (LDS-DMA) and W are in place
            {
                h8 xh[1], xl[1];
                f8_load_x<1, 16 * STRVD, false>(wsp_hi, wsp_lo, aoffw, 0, xh, xl);
                // A operand = T'^T through transposing reads (see mlp_fused8_kernel); k-groups 2, 3 (slots 16 .. 31) read the last
                // row again against zero weights
                const int tq = l15 >> 2, tc = 8 * (l15 & 3);
                const int ro0 = min(8 * g4 + tq, nU - 1) * STOK_STR + tc, ro1 = min(8 * g4 + 4 + tq, nU - 1) * STOK_STR + tc;
                typedef __attribute__((address_space(3))) f8_s4* f8_lp;
#pragma unroll
                for (int r = 0; r < V; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const char* rb = abuf + r * 16 * STOK_STR + 2 * ((2 * wave + (c >> 1)) * 32 + (c & 1) * 16);
                        const f8_s4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro0));
                        const f8_s4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro1));
                        const f8_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro0 + 512));
                        const f8_s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro1 + 512));
                        const f8_s8 ahs = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const f8_s8 als = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const h8 ah = __builtin_bit_cast(h8, ahs), al = __builtin_bit_cast(h8, als);
                        acc2[c][r] = F8_MFMA(al, xh[0], acc2[c][r]);
                        acc2[c][r] = F8_MFMA(ah, xl[0], acc2[c][r]);
                        acc2[c][r] = F8_MFMA(ah, xh[0], acc2[c][r]);
                    }
            }
        }
        FM_SYNCL();                                   // every wave is done reading the T' rows: ABUF may take s
        {
            asm volatile("" : "+v"(rv4), "+v"(rv1));
            seen_s = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[0]);
            seen_p = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[1]);
            seen_n = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[2]);
            seen_i = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[3]);
            seen_4 = (unsigned)__builtin_amdgcn_readfirstlane((int)rv1);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f8_f4 pe2 = f8_finish(a1[c][0], b0[c], PK.fc_0pe.inv_scale, false);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f32x2 it2 = {inv_t, inv_t};
                f32x2 u01 = __builtin_elementwise_fma((f32x2){acc2[c][r][0], acc2[c][r][1]}, it2, (f32x2){pe2[0], pe2[1]});
                f32x2 u23 = __builtin_elementwise_fma((f32x2){acc2[c][r][2], acc2[c][r][3]}, it2, (f32x2){pe2[2], pe2[3]});
                u01 = __builtin_elementwise_max(u01, (f32x2){0.f, 0.f});
                u23 = __builtin_elementwise_max(u23, (f32x2){0.f, 0.f});
                const f8_f4 u = {u01[0], u01[1], u23[0], u23[1]};
                f8_store_h<STR256>(u, r * 16 + l15, (2 * wave + (c >> 1)) * 32 + (c & 1) * 16, abuf, a256_lo, lane, rmax);
            }
        }
        range_commit(PK.range, TH_RANGE_S, seen_s, rmax);
    }
    tex_pre = tex_fetch();                                  // (the round trip runs under kv1)
    FM_SYNCL();
    FH_DUMP_AT(1, abuf, a256_lo, STR256, 16)
    // kv layers, per virtual wave: column tile 0 = key cols 16 vwave .., tiles 1, 2 = value cols 128 + 32 vwave ..
    f8_f4 vs[4][RT];
    float* ksb = reinterpret_cast<float*>(mbuf);                    // [ROWS][KSTR] fp32 keys of the token branch
#pragma unroll
    for (int vw = 0; vw < 2; ++vw) {
        const int vwave = 2 * wave + vw;
        f8_f4 acc3[3][RT];
        fh_gemm<3, RT, 3, 0, 8, RS256, true, 2>(abuf, a256_lo, STR256, F8_WSLICE(PK.w16.kv1, vwave, 3), lane, acc3);
        FM_SB();
        const float4 bk = f8_bias(PK.kv1.bias, vwave * 16, lane), bv0 = f8_bias(PK.kv1.bias, 128 + vwave * 32, lane),
                     bv1 = f8_bias(PK.kv1.bias, 128 + vwave * 32 + 16, lane);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const f8_f4 k = f8_finish(acc3[0][r], bk, PK.kv1.inv_scale, false);
            *reinterpret_cast<float4*>(ksb + (r * 16 + l15) * KSTR + vwave * 16 + 4 * g4) = make_float4(k[0], k[1], k[2], k[3]);
            vs[2 * vw][r] = f8_finish(acc3[1][r], bv0, PK.kv1.inv_scale, false);
            vs[2 * vw + 1][r] = f8_finish(acc3[2][r], bv1, PK.kv1.inv_scale, false);
        }
    }

    // ================= pixel branch: p = relu(alpha_res_0 f) = blend of fold0 rows; kp|vp = kv0(p) =================
    FM_SB();
    fill_tex(tex_pre, std::false_type{}, [] {});            // (its first barrier: every wave is done reading s)
    FM_SB();
    FM_SYNCL();
    FH_DUMP_AT(2, abuf, a256_lo, STR256, 16)
    f8_f4 vp[4][RT];
    {
        f8_f4 kk[2][RT];
#pragma unroll
        for (int vw = 0; vw < 2; ++vw) {
            const int vwave = 2 * wave + vw;
            f8_f4 acc3[3][RT];
            fh_gemm<3, RT, 3, 0, 8, RS256, true, 2>(abuf, a256_lo, STR256, F8_WSLICE(PK.w16.kv0, vwave, 3), lane, acc3);
            FM_SB();
            const float4 bk = f8_bias(PK.kv0.bias, vwave * 16, lane), bv0 = f8_bias(PK.kv0.bias, 128 + vwave * 32, lane),
                         bv1 = f8_bias(PK.kv0.bias, 128 + vwave * 32 + 16, lane);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                kk[vw][r] = f8_finish(acc3[0][r], bk, PK.kv0.inv_scale, false);
                vp[2 * vw][r] = f8_finish(acc3[1][r], bv0, PK.kv0.inv_scale, false);
                vp[2 * vw + 1][r] = f8_finish(acc3[2][r], bv1, PK.kv0.inv_scale, false);
            }
        }
        FM_SYNCL();                                                  // every wave is done reading p from ABUF
        FH_DUMP_AT(8, abuf, a256_lo, STR256, 16)
        FH_DUMPSUM_AT(9, abuf, a256_lo, STR256, 16)
        float* kpb = reinterpret_cast<float*>(abuf);                // [ROWS][KSTR]
#pragma unroll
        for (int vw = 0; vw < 2; ++vw)
#pragma unroll
            for (int r = 0; r < RT; ++r)
                *reinterpret_cast<float4*>(kpb + (r * 16 + l15) * KSTR + (2 * wave + vw) * 16 + 4 * g4) =
                    make_float4(kk[vw][r][0], kk[vw][r][1], kk[vw][r][2], kk[vw][r][3]);
    }
    FM_SYNCL();

    FH_DUMPF_AT(5, reinterpret_cast<const float*>(abuf), KSTR, 16, 0, 41, 77, 127)
    FH_DUMPF_AT(6, ksb, KSTR, 16, 0, 41, 77, 127)
    // ================= cross-view attention (cross_transformer.py:128-149) =================
    {
        const float* kpb = reinterpret_cast<const float*>(abuf);
        float4 bn[4];
        {
            const int p = tid >> 4, c16 = tid & 15;
            f32x2 accp[V * V];
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) accp[ji] = (f32x2){0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 kx[V], sx[V];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    kx[v] = *reinterpret_cast<const float4*>(kpb + (v * 16 + p) * KSTR + 4 * (c16 + 16 * q));
                    sx[v] = *reinterpret_cast<const float4*>(ksb + (v * 16 + p) * KSTR + 4 * (c16 + 16 * q));
                }
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        accp[j * V + i] = __builtin_elementwise_fma((f32x2){kx[j].x, kx[j].y}, (f32x2){sx[i].x, sx[i].y}, accp[j * V + i]);
                        accp[j * V + i] = __builtin_elementwise_fma((f32x2){kx[j].z, kx[j].w}, (f32x2){sx[i].z, sx[i].w}, accp[j * V + i]);
                    }
            }
            float acc[V * V];
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) {
                float s = accp[ji][0] + accp[ji][1];
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, true));
                acc[ji] = s;
            }
            float a[V], m = -3.0e38f;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float x = acc[j * V];
#pragma unroll
                for (int i = 1; i < V; ++i) x = c16 == i ? acc[j * V + i] : x;
                a[j] = x / 11.313708498984761f;
                m = fmaxf(m, a[j]);
            }
            float e[V], se = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) { e[j] = expf(a[j] - m); se = se + e[j]; }
            if (c16 < V) {
#pragma unroll
                for (int j = 0; j < V; ++j) probs[p * FH_PSTR + j * V + c16] = e[j] / se;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) bn[c] = f8_bias(PK.fc_1.bias, (2 * wave + (c >> 1)) * 32 + (c & 1) * 16, lane);
        FM_SYNCL();
        FH_DUMPF_AT(7, probs, FH_PSTR, -1, 0, 3, 4, 8)
        {
            const float* pr = probs + l15 * FH_PSTR;
            const float4 q0 = *reinterpret_cast<const float4*>(pr), q1 = *reinterpret_cast<const float4*>(pr + 4);
            const float q2 = pr[8];
            const float A[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 b01 = {bn[c].x, bn[c].y}, b23 = {bn[c].z, bn[c].w};
                f32x2 t0[V], t1[V];
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    t0[i] = (f32x2){vs[c][i][0], vs[c][i][1]} + b01;
                    t1[i] = (f32x2){vs[c][i][2], vs[c][i][3]} + b23;
                }
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const f32x2 v0 = {vp[c][j][0], vp[c][j][1]}, v1 = {vp[c][j][2], vp[c][j][3]};
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const f32x2 a2 = {A[j * V + i], A[j * V + i]};
                        t0[i] = __builtin_elementwise_fma(v0, a2, t0[i]);
                        t1[i] = __builtin_elementwise_fma(v1, a2, t1[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    t0[i] = __builtin_elementwise_max(t0[i], (f32x2){0.f, 0.f});
                    t1[i] = __builtin_elementwise_max(t1[i], (f32x2){0.f, 0.f});
                    const f8_f4 n = {t0[i][0], t0[i][1], t1[i][0], t1[i][1]};
                    f8_store_h<STR256>(n, i * 16 + l15, (2 * wave + (c >> 1)) * 32 + (c & 1) * 16, abuf, a256_lo, lane, rmax);
                }
            }
        }
        range_commit(PK.range, TH_RANGE_N, seen_n, rmax);
    }
    FH_DUMP_AT(3, abuf, a256_lo, STR256, 16)

    // ================= fc_2 (fc_1 is folded into the value projections) =================
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + FH_THREADS * q, row = i >> 5, c = i & 31;
        int x = pbase + row;
        if (PK.vd_sel != nullptr && PK.rgb_all != 2 && c < 27 && row < npts) x = PK.vd_sel[pbase + row];
        vsel[q] = x;
    }
    FM_SYNCL();
    fh_gemm<2, RT, 4, 0, 8, RS256, true, 3>(abuf, a256_lo, STR256, F8_WSLICE(PK.w16.fc_2, 2 * wave, 2), lane, acc2);
    FM_SB();
    fh_gemm<2, RT, 4, 2, 8, RS256, true, 3>(abuf, a256_lo, STR256, F8_WSLICE(PK.w16.fc_2, 2 * wave + 1, 2), lane, acc2);
    FM_SB();
    float4 bi[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bi[c] = f8_bias(PK.fc_2.bias, (2 * wave + (c >> 1)) * 32 + (c & 1) * 16, lane);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + FH_THREADS * q, row = i >> 5, c = i & 31;
        float x = 0.f;
        if (PK.rgb_all != 2 && c < 27 && row < npts) {
            const long long vr = PK.vd_sel ? (long long)(vsel[q] / PK.vd_div) : (long long)(pbase + row);
            x = PK.vd[vr * 27 + c];
        }
        vdv[q] = x;
    }
    FM_SYNCL();
    // inter = relu(.) -> ABUF (operand of the folded view_fc); its view mean -> MBUF (operand of fc_3)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col0 = (2 * wave + (c >> 1)) * 32 + (c & 1) * 16;
#pragma unroll
        for (int r = 0; r < RT; ++r) acc2[c][r] = f8_finish(acc2[c][r], bi[c], PK.fc_2.inv_scale, true);
        f8_f4 m = acc2[c][0];
#pragma unroll
        for (int r = 1; r < V; ++r) m = m + acc2[c][r];
        m = m * (f8_f4){inv_v, inv_v, inv_v, inv_v};
        f8_store_h<STR256>(m, l15, col0, mbuf, mbuf + 16 * STR256, lane, rmax);
#pragma unroll
        for (int r = 0; r < RT; ++r) f8_store_h<STR256>(acc2[c][r], r * 16 + l15, col0, abuf, a256_lo, lane, rmax);
    }
    range_commit(PK.range, TH_RANGE_INTER, seen_i, rmax);
    FM_SYNCL();
    FH_DUMP_AT(4, abuf, a256_lo, STR256, 16)

    // ================= sigma head: relu(fc_3 m) . alpha_w + b   ||   folded view_fc on inter =================
    char* vd_hi = mbuf;
    char* vd_lo = vd_hi + 16 * STRVD;
    tex_pre2 = tex_fetch();                              // (for the RGB branch's filling: the round trip runs under fc_3)
    f8_f4 va[2][RT];                                     // [virtual wave]: 16 of the 128 view_fc outputs, all views
    {
        f8_f4 a3[4][1];                                  // [virtual wave x column tile]: fc_3 on the mean rows
        if (PK.rgb_all != 2) {
            constexpr int T = 8;
            const int aoff = f8_aoff<true>(lane, STR256);
            const char* mhi = mbuf;
            const char* mlo = mbuf + 16 * STR256;
            const uint4* w3l[2] = {F8_WSLICE(PK.w16.fc_3, 2 * wave, 2) + lane, F8_WSLICE(PK.w16.fc_3, 2 * wave + 1, 2) + lane};
            const uint4* wal[2] = {F8_WSLICE(PK.w16.vfA, 2 * wave, 1) + lane, F8_WSLICE(PK.w16.vfA, 2 * wave + 1, 1) + lane};
            uint4 r3[2][2][2][2], ra[2][2][1][2];        // [buffer][virtual wave]...
            h8 mh[2][1], ml[2][1], xh[2][V], xl[2][V];
#pragma unroll
            for (int vw = 0; vw < 2; ++vw) {
                f8_load_w<2>(w3l[vw], 0, r3[0][vw]);
                f8_load_w<1>(wal[vw], 0, ra[0][vw]);
            }
            f8_load_x<1, RS256, true>(mhi, mlo, aoff, 0, mh[0], ml[0]);
            f8_load_x<V, RS256, true>(abuf, a256_lo, aoff, 0, xh[0], xl[0]);
            FM_SB();
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int cb = t & 1, nb = cb ^ 1;
                if (t + 1 < T) {
#pragma unroll
                    for (int vw = 0; vw < 2; ++vw) {
                        f8_load_w<2>(w3l[vw], t + 1, r3[nb][vw]);
                        f8_load_w<1>(wal[vw], t + 1, ra[nb][vw]);
                    }
                    f8_load_x<1, RS256, true>(mhi, mlo, aoff, t + 1, mh[nb], ml[nb]);
                    f8_load_x<V, RS256, true>(abuf, a256_lo, aoff, t + 1, xh[nb], xl[nb]);
                }
                if (t == 0) {
                    fh_mfma<2, 1, 4, 0, true>(r3[cb][0], mh[cb], ml[cb], a3);
                    fh_mfma<1, RT, 2, 0, true>(ra[cb][0], xh[cb], xl[cb], va);
                    fh_mfma<2, 1, 4, 2, true>(r3[cb][1], mh[cb], ml[cb], a3);
                    fh_mfma<1, RT, 2, 1, true>(ra[cb][1], xh[cb], xl[cb], va);
                } else {
                    fh_mfma<2, 1, 4, 0, false>(r3[cb][0], mh[cb], ml[cb], a3);
                    fh_mfma<1, RT, 2, 0, false>(ra[cb][0], xh[cb], xl[cb], va);
                    fh_mfma<2, 1, 4, 2, false>(r3[cb][1], mh[cb], ml[cb], a3);
                    fh_mfma<1, RT, 2, 1, false>(ra[cb][1], xh[cb], xl[cb], va);
                }
                FM_SB();
            }
        } else {
#pragma unroll
            for (int vw = 0; vw < 2; ++vw)
#pragma unroll
                for (int r = 0; r < RT; ++r) va[vw][r] = (f8_f4){0.f, 0.f, 0.f, 0.f};      // (never read on this path)
            fh_gemm<2, 1, 4, 0, 8, RS256, true, 3>(mbuf, mbuf + 16 * STR256, STR256, F8_WSLICE(PK.w16.fc_3, 2 * wave, 2), lane, a3);
            fh_gemm<2, 1, 4, 2, 8, RS256, true, 3>(mbuf, mbuf + 16 * STR256, STR256, F8_WSLICE(PK.w16.fc_3, 2 * wave + 1, 2), lane, a3);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col0 = (2 * wave + (c >> 1)) * 32 + (c & 1) * 16;
            const float4 aw = f8_bias(PK.alpha_w, col0, lane), b3 = f8_bias(PK.fc_3.bias, col0, lane);
            const f8_f4 y = f8_finish(a3[c][0], b3, PK.fc_3.inv_scale, true);
            s = fmaf(y[0], aw.x, s);
            s = fmaf(y[1], aw.y, s);
            s = fmaf(y[2], aw.z, s);
            s = fmaf(y[3], aw.w, s);
        }
        s += f8_xor(s, lane, 16);
        s += f8_xor(s, lane, 32);
        if (g4 == 0) part[wave * 16 + l15] = s;
        if (tid == 0) *flag = 0;
        FM_SYNCL();                                  // every wave is done reading the means (MBUF) and inter (ABUF)
        if (tid < 16) {
            float sg = PK.alpha_b[0];
#pragma unroll
            for (int w = 0; w < 4; ++w) sg += part[w * 16 + tid];
            sig[tid] = sg;
            if (tid < npts && PK.rgb_all != 2 && (PK.rgb_all == 1 || sg > 0.f)) *flag = 1;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + FH_THREADS * q, row = i >> 5, c = i & 31;
            _Float16 a, b;
            split_h(vdv[q], a, b);
            *reinterpret_cast<_Float16*>(vd_hi + row * STRVD + 2 * c) = a;
            *reinterpret_cast<_Float16*>(vd_lo + row * STRVD + 2 * c) = b;
        }
        FM_SYNCL();
    }
    const bool need_rgb = *flag != 0;
    float rgb_out[3] = {0.f, 0.f, 0.f};
    if (need_rgb) {
        // ================= RGB branch (cross_transformer.py:330-353) =================
        uint4 wvd[2][1][2];
#pragma unroll
        for (int vw = 0; vw < 2; ++vw) f8_load_w<1>(F8_WSLICE(PK.w16.vfD, 2 * wave + vw, 1) + lane, 0, wvd[vw]);
        FM_SB();
        fill_tex(tex_pre2, std::true_type{}, [&]() __attribute__((always_inline)) {
            h8 xh[1], xl[1];
            f8_load_x<1, 16 * STRVD, false>(vd_hi, vd_lo, f8_aoff<false>(lane, STRVD), 0, xh, xl);
#pragma unroll
            for (int vw = 0; vw < 2; ++vw)
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    va[vw][r] = F8_MFMA(*reinterpret_cast<const h8*>(&wvd[vw][0][1]), xh[0], va[vw][r]);
                    va[vw][r] = F8_MFMA(*reinterpret_cast<const h8*>(&wvd[vw][0][0]), xl[0], va[vw][r]);
                    va[vw][r] = F8_MFMA(*reinterpret_cast<const h8*>(&wvd[vw][0][0]), xh[0], va[vw][r]);
                }
        });
        FM_SYNCL();
        uint4 w4[2][4][1][2];
#pragma unroll
        for (int vw = 0; vw < 2; ++vw) {
            const uint4* wl4 = F8_WSLICE(PK.w16.fc_4, 2 * wave + vw, 1) + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) f8_load_w<1>(wl4, t, w4[vw][t]);
        }
        char* f4_hi = mbuf + FH_MBUF_FC4_OFF;
        char* f4_lo = f4_hi + 16 * STR128;
#pragma unroll
        for (int vw = 0; vw < 2; ++vw) {
            const int vwave = 2 * wave + vw;
            const float4 bt = f8_bias(PK.rst.bias, vwave * 16, lane), br = f8_bias(PK.rst.bias, 128 + vwave * 16, lane);
            const char* mb = abuf + l15 * 1040 + 4 * (vwave * 16 + 4 * g4);
            f8_f4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float4 m1 = *reinterpret_cast<const float4*>(mb + r * 16 * 1040);
                const float4 m2 = *reinterpret_cast<const float4*>(mb + r * 16 * 1040 + 512);
                const f8_f4 t = f8_finish(va[vw][r], bt, PK.rst.inv_scale, false);
                const f32x2 z2 = {0.f, 0.f};
                const f32x2 u01 = __builtin_elementwise_max((f32x2){t[0], t[1]} + (f32x2){m1.x, m1.y}, z2) + ((f32x2){m2.x, m2.y} + (f32x2){br.x, br.y});
                const f32x2 u23 = __builtin_elementwise_max((f32x2){t[2], t[3]} + (f32x2){m1.z, m1.w}, z2) + ((f32x2){m2.z, m2.w} + (f32x2){br.z, br.w});
                const f8_f4 u = {u01[0], u01[1], u23[0], u23[1]};
                m = r == 0 ? u : m + u;
            }
            m = m * (f8_f4){inv_v, inv_v, inv_v, inv_v};
            f8_store_h<STR128, false>(m, l15, vwave * 16, f4_hi, f4_lo, lane, rmax);      // (signed: relu(.) + rgb_res_1)
        }
        range_commit(PK.range, TH_RANGE_F4, seen_4, rmax);
        FM_SYNCL();
        float s3[3] = {0.f, 0.f, 0.f};
        {
            const int aoff4 = f8_aoff<false>(lane, STR128);
            f8_f4 a4[2][1][1];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                h8 xh4[1], xl4[1];
                f8_load_x<1, 16 * STR128, false>(f4_hi, f4_lo, aoff4, t, xh4, xl4);
#pragma unroll
                for (int vw = 0; vw < 2; ++vw) {
                    if (t == 0) fh_mfma<1, 1, 1, 0, true>(w4[vw][t], xh4, xl4, a4[vw]);
                    else fh_mfma<1, 1, 1, 0, false>(w4[vw][t], xh4, xl4, a4[vw]);
                }
            }
#pragma unroll
            for (int vw = 0; vw < 2; ++vw) {
                const int vwave = 2 * wave + vw;
                const float4 b4 = f8_bias(PK.fc_4.bias, vwave * 16, lane);
                const f8_f4 y = f8_finish(a4[vw][0][0], b4, PK.fc_4.inv_scale, true);
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float4 rw = f8_bias(PK.rgb_w + o * 128, vwave * 16, lane);
                    float s = s3[o];
                    s = fmaf(y[0], rw.x, s);
                    s = fmaf(y[1], rw.y, s);
                    s = fmaf(y[2], rw.z, s);
                    s = fmaf(y[3], rw.w, s);
                    s3[o] = s;
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float s = s3[o];
            s += f8_xor(s, lane, 16);
            s += f8_xor(s, lane, 32);
            if (g4 == 0) part[(o * 4 + wave) * 16 + l15] = s;
        }
        FM_SYNCL();
        if (tid < 16) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float s = PK.rgb_b[o];
#pragma unroll
                for (int w = 0; w < 4; ++w) s += part[(o * 4 + w) * 16 + tid];
                rgb_out[o] = s;
            }
        }
    }
    if (tid < npts)
        *reinterpret_cast<float4*>(PK.raw_c + (long long)(pbase + tid) * 4) = make_float4(rgb_out[0], rgb_out[1], rgb_out[2], sig[tid]);
    if (FM_DBG_SAMPLED) {
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + 62), (unsigned long long)(clock64() - dbg_keep[0]));
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + 63), (unsigned long long)(wall_clock64() - dbg_keep[1]));
    }
#undef PK
}
