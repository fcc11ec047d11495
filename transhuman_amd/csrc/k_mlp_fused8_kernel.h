// K6 v2 (round 6): the fused per-point MLP with TWO waves per SIMD.
//
// Same tile (32 samples x V views), same LDS map, same hand-overs (K4's neighbour records, K5t's texel lists, the folded
// maps) and same arithmetic (fp16 hi/lo split, three MFMA products per fp32 MAC, fp32 accumulate) as mlp_fused_kernel
// (k_mlp_fused_kernel.h) -- what changes is who does the work: one workgroup = 512 threads = 8 waves, two per SIMD, every wave
// owning HALF the output columns of a layer (32 of 256, 16 of the 128 keys).  Why (profiles/r06_a_issue_rates.txt, measured):
//   * a wave alone on its SIMD issues one VALU / LDS instruction every 7.4 - 9.6 cycles; two waves issue 1.5 - 2 x that
//     together.  59 of the 4-wave kernel's 117 k cycles per tile are fillings, epilogues and the attention, i.e. plain
//     instruction issue at the lone-wave rate (9 850 non-matrix instructions per wave and tile);
//   * half the columns = half the accumulators: 232 + 256 registers become <= 256, which is what lets two waves share a SIMD.
// The matrix instruction is v_mfma_f32_16x16x32_f16 (16 output channels x 16 samples x 32 deep): 24 key/value column tiles
// = 3 per wave, 16 column tiles of the 256-wide layers = 2 per wave, 6 row tiles (view x half of the samples) -- every
// layer splits evenly over 8 waves, no K-splits, no exchanges.  It issues every 16.6 cycles with two waves feeding a SIMD
// (32x32x16: 32.0 -- 3.6 % more pipe time per FLOP; 17.4 from one wave).
// Fragment layout (lane = 16 g + l): A (weights) lane holds column col0 + l, 8 consecutive k of k-group g; B (activations)
// row l of the row tile, the same 8 k; D lane holds sample l, channels col0 + 4 g .. 4 g + 3.
// K = 256 operands are read from the LDS planes (row stride 528 B = 33 slots of 16 B) in a PERMUTED k order: step t, group g
// reads slot 2 t + (g >> 1) + 16 (g & 1) -- the 16 lanes a ds_read_b128 services per LDS cycle ({0-3, 12-15, 20-27}, ...)
// then touch 16 different slots of the 256-byte bank row (natural order: rows 4-11 of group 1 collide with rows 0-3, 12-15
// of group 0).  The weight image is packed in the same order (k_mlp_fused_host.hip pack_fused16_kernel).
// Reference: Network._multiview_agg / cross_attention / _alpha_forward / _RGB_forward, cross_transformer.py:128-149, :291-353.
#pragma once
#include "k_mlp_fused_kernel.h"

typedef float f8_f4 __attribute__((ext_vector_type(4)));
typedef short f8_s4 __attribute__((ext_vector_type(4)));
typedef short f8_s8 __attribute__((ext_vector_type(8)));

#define F8_THREADS 512
#define F8_PART_FLOATS (3 * 8 * 32)                        // [3 outputs][8 waves][32 samples] cross-wave partial sums
#define F8_PSTR 12                                         // floats per sample of the softmax table ([j][i], 9 used)
#define F8_MISC_FLOATS (F8_PSTR * 32 + F8_PART_FLOATS + 32 + 8)  // probs | part | sig | flag
#define F8_LDS_BYTES (ABUF_BYTES + MBUF_BYTES + F8_MISC_FLOATS * 4)
static_assert(F8_LDS_BYTES <= 163840, "the 8-wave tile must fit the CU's LDS");

#define F8_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16((A), (B), (C), 0, 0, 0)

// this wave's weight stream: per k-step t: CT x {hi, lo} x 64 lanes x 16 B
template <int CT>
__device__ __forceinline__ void f8_load_w(const uint4* __restrict__ wl, int t, uint4 (&w)[CT][2]) {
    const uint4* p = wl + (long long)t * (CT * 2 * 64);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        w[c][0] = p[(c * 2 + 0) * 64];
        w[c][1] = p[(c * 2 + 1) * 64];
    }
}
#define F8_WSLICE(L, wave, ct) ((L).w + (long long)(wave) * (L).KB * ((ct) * 2 * 64))   // (KB = number of 32-deep k-steps of a 16-form image)

// NR row tiles (16 rows each, ROWSTEP bytes apart) of one k-step
template <int NR, int ROWSTEP, bool PERM>
__device__ __forceinline__ void f8_load_x(const char* __restrict__ ahi, const char* __restrict__ alo, int aoff, int t,
                                          h8 (&xh)[NR], h8 (&xl)[NR]) {
    const int ko = (PERM ? 32 : 64) * t;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        xh[r] = *reinterpret_cast<const h8*>(ahi + r * ROWSTEP + aoff + ko);
        xl[r] = *reinterpret_cast<const h8*>(alo + r * ROWSTEP + aoff + ko);
    }
}
// per-lane byte offset of a fragment inside a plane (row l of the tile, k-group g)
template <bool PERM>
__device__ __forceinline__ int f8_aoff(int lane, int str) {
    const int l = lane & 15, g = lane >> 4;
    return PERM ? l * str + 16 * (g >> 1) + 256 * (g & 1) : l * str + 16 * g;
}

// acc[c][R0 + r] (+)= W(c) X(r)^T for one k-step: three fp16 products, term-major (an accumulator recurs every CT * NR MFMAs)
template <int CT, int RT, int NR, int R0, bool FIRST>
__device__ __forceinline__ void f8_mfma_half(const uint4 (&w)[CT][2], const h8 (&xh)[NR], const h8 (&xl)[NR], f8_f4 (&acc)[CT][RT]) {
    const f8_f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < NR; ++r)
            acc[c][R0 + r] = F8_MFMA(*reinterpret_cast<const h8*>(&w[c][1]), xh[r], FIRST ? zero : acc[c][R0 + r]);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[c][R0 + r] = F8_MFMA(*reinterpret_cast<const h8*>(&w[c][0]), xl[r], acc[c][R0 + r]);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[c][R0 + r] = F8_MFMA(*reinterpret_cast<const h8*>(&w[c][0]), xh[r], acc[c][R0 + r]);
}

// one MFMA, one memory instruction ... : the loads of a half-step issue in the shadow of its MFMAs
template <int NMEM, int NMF>
__device__ __forceinline__ void f8_interleave() {
    constexpr int NP = NMEM < NMF ? NMEM : NMF;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);
    }
    if constexpr (NMF > NP) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NP, 0);
    if constexpr (NMEM > NP) __builtin_amdgcn_sched_group_barrier(0x120, NMEM - NP, 0);
    FM_SB();
}

// acc = W A^T over T k-steps (T even, or 1).  Row tiles in two halves A = [0, RT/2), B = [RT/2, RT): while the MFMAs of half A
// run, half B's fragments (and the next step's weights) are in flight, and half A's registers take the NEXT step's rows as
// soon as its MFMAs have issued -- one set of activation registers, weights double-buffered (a k-step is 54 MFMAs of a
// 3 x 6 phase: ~ 1.8 k cycles with two waves on the pipe, several L2 round trips).
// PRE: block 0 of the weights was requested by the caller (in front of the barrier that publishes the operand).
template <int RT, int CT, int ROWSTEP, bool PERM, bool PRE = false>
__device__ __forceinline__ void f8_gemm(const char* __restrict__ ahi, const char* __restrict__ alo, int str,
                                        const uint4* __restrict__ wp, int T, int lane, f8_f4 (&acc)[CT][RT], uint4 (&w)[2][CT][2]) {
    static_assert(RT % 2 == 0, "row tiles come in halves");
    constexpr int NR = RT / 2;
    const uint4* wl = wp + lane;
    const int aoff = f8_aoff<PERM>(lane, str);
    h8 xah[NR], xal[NR], xbh[NR], xbl[NR];
    if (!PRE) f8_load_w<CT>(wl, 0, w[0]);
    f8_load_x<NR, ROWSTEP, PERM>(ahi, alo, aoff, 0, xah, xal);
    FM_SB();
    // first step peeled: its MFMAs start the accumulators from the inline constant 0
    auto step = [&](auto first, auto par, int t) __attribute__((always_inline)) {
        constexpr bool F = decltype(first)::value;
        constexpr int PB = decltype(par)::value;
        const int tn = t + 1 < T ? t + 1 : T - 1;             // (clamped: the last step re-requests what nobody consumes)
        f8_load_x<NR, ROWSTEP, PERM>(ahi + NR * ROWSTEP, alo + NR * ROWSTEP, aoff, t, xbh, xbl);
        f8_load_w<CT>(wl, tn, w[PB ^ 1]);
        f8_mfma_half<CT, RT, NR, 0, F>(w[PB], xah, xal, acc);
        f8_interleave<2 * NR + 2 * CT, 3 * CT * NR>();
        f8_load_x<NR, ROWSTEP, PERM>(ahi, alo, aoff, tn, xah, xal);
        f8_mfma_half<CT, RT, NR, NR, F>(w[PB], xbh, xbl, acc);
        f8_interleave<2 * NR, 3 * CT * NR>();
    };
    step(std::true_type{}, std::integral_constant<int, 0>{}, 0);
    if (T > 1) {
        step(std::false_type{}, std::integral_constant<int, 1>{}, 1);
#pragma unroll 1
        for (int t = 2; t < T; t += 2) {
            step(std::false_type{}, std::integral_constant<int, 0>{}, t);
            step(std::false_type{}, std::integral_constant<int, 1>{}, t + 1);
        }
    }
}

// The same product with the row tiles in THREE groups of RT / 3 (V = 3: view by view): two fragment buffers of RT / 3 row tiles
// ping-pong through the 3 T sub-steps -- 32 fragment registers instead of 48.  With halves the kv0 phase (72 accumulator, 48
// parked value, 48 weight and 48 fragment registers) spilled 20 registers per wave: 2.4 GB of scratch writes per frame in the
// round-6 PMC pass (profiles/r06_y_pmc_mlp.txt, WRITE_SIZE).  T even.
template <int RT, int CT, int ROWSTEP, bool PERM, bool PRE = false>
__device__ __forceinline__ void f8_gemm3(const char* __restrict__ ahi, const char* __restrict__ alo, int str,
                                         const uint4* __restrict__ wp, int T, int lane, f8_f4 (&acc)[CT][RT], uint4 (&w)[2][CT][2]) {
    static_assert(RT % 3 == 0, "row tiles come in thirds");
    constexpr int NR = RT / 3;
    const uint4* wl = wp + lane;
    const int aoff = f8_aoff<PERM>(lane, str);
    h8 xh[2][NR], xl[2][NR];
    if (!PRE) f8_load_w<CT>(wl, 0, w[0]);
    f8_load_x<NR, ROWSTEP, PERM>(ahi, alo, aoff, 0, xh[0], xl[0]);
    FM_SB();
    // sub-step S of a pair of k-steps (t, t + 1), t even: group S % 3 of step t + S / 3 out of buffer S & 1
    auto sub = [&](auto first, auto sidx, int t) __attribute__((always_inline)) {
        constexpr int S = decltype(sidx)::value, G = S % 3, PB = S / 3, CUR = S & 1;
        constexpr bool F = decltype(first)::value && S < 3;
        constexpr int GN = (G + 1) % 3;
        const int tt = t + PB;
        const int tx = G == 2 ? (tt + 1 < T ? tt + 1 : T - 1) : tt;          // (clamped: the last sub-step re-requests what nobody consumes)
        f8_load_x<NR, ROWSTEP, PERM>(ahi + GN * NR * ROWSTEP, alo + GN * NR * ROWSTEP, aoff, tx, xh[CUR ^ 1], xl[CUR ^ 1]);
        if constexpr (G == 0) f8_load_w<CT>(wl, tt + 1 < T ? tt + 1 : T - 1, w[PB ^ 1]);
        f8_mfma_half<CT, RT, NR, G * NR, F>(w[PB], xh[CUR], xl[CUR], acc);
        f8_interleave<2 * NR + (G == 0 ? 2 * CT : 0), 3 * CT * NR>();
    };
    auto pair = [&](auto first, int t) __attribute__((always_inline)) {
        sub(first, std::integral_constant<int, 0>{}, t); sub(first, std::integral_constant<int, 1>{}, t);
        sub(first, std::integral_constant<int, 2>{}, t); sub(first, std::integral_constant<int, 3>{}, t);
        sub(first, std::integral_constant<int, 4>{}, t); sub(first, std::integral_constant<int, 5>{}, t);
    };
    pair(std::true_type{}, 0);
#pragma unroll 1
    for (int t = 2; t < T; t += 2) pair(std::false_type{}, t);
}
// (row tiles in thirds where they divide by three, else in halves)
template <int RT, int CT, int ROWSTEP, bool PERM, bool PRE = false>
__device__ __forceinline__ void f8_gemm_rt(const char* __restrict__ ahi, const char* __restrict__ alo, int str,
                                           const uint4* __restrict__ wp, int T, int lane, f8_f4 (&acc)[CT][RT], uint4 (&w)[2][CT][2]) {
    if constexpr (RT % 3 == 0) f8_gemm3<RT, CT, ROWSTEP, PERM, PRE>(ahi, alo, str, wp, T, lane, acc, w);
    else f8_gemm<RT, CT, ROWSTEP, PERM, PRE>(ahi, alo, str, wp, T, lane, acc, w);
}

// 16 x 16 output tile of this lane (sample l, channels col0 + 4 g ..): hi / lo halves into the operand planes
template <int STR, bool NONNEG = true>
__device__ __forceinline__ void f8_store_h(const f8_f4& t, int row, int col0, char* __restrict__ hi, char* __restrict__ lo, int lane,
                                           unsigned& rm) {
    const int c = col0 + 4 * (lane >> 4);
    uint2 a, b;
    split_pair(t[0], t[1], a.x, b.x);
    split_pair(t[2], t[3], a.y, b.y);
    range_acc<NONNEG>(rm, a.x);
    range_acc<NONNEG>(rm, a.y);
    *reinterpret_cast<uint2*>(hi + row * STR + 2 * c) = a;
    *reinterpret_cast<uint2*>(lo + row * STR + 2 * c) = b;
}
// y = acc * inv_scale + bias (inv_scale a power of two: the fused form rounds like mul + add), optional relu
__device__ __forceinline__ f8_f4 f8_finish(const f8_f4& a, const float4& b, float inv_scale, bool relu) {
    const f32x2 sc = {inv_scale, inv_scale};
    f32x2 y01 = __builtin_elementwise_fma((f32x2){a[0], a[1]}, sc, (f32x2){b.x, b.y});
    f32x2 y23 = __builtin_elementwise_fma((f32x2){a[2], a[3]}, sc, (f32x2){b.z, b.w});
    if (relu) {
        y01 = __builtin_elementwise_max(y01, (f32x2){0.f, 0.f});
        y23 = __builtin_elementwise_max(y23, (f32x2){0.f, 0.f});
    }
    return (f8_f4){y01[0], y01[1], y23[0], y23[1]};
}
// value of lane (lane ^ mask): ds_bpermute addressed from the kernel's own (per-tile) lane number (__shfl_xor derives the lane from
// mbcnt: loop-invariant, kept live across the tile loop)
__device__ __forceinline__ float f8_xor(float v, int lane, int mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ mask) << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float4 f8_bias(const float* __restrict__ bias, int col0, int lane) {
    return *reinterpret_cast<const float4*>(bias + col0 + 4 * (lane >> 4));
}

// one half-step of the fc_3 || view_fc loop: a3[c][H] (2 column tiles on ONE mean row tile) and va[0][R0 ..] (one column tile on NV row
// tiles), term by term with the two products interleaved -- an accumulator recurs every 2 + NV MFMAs (fc_3's two accumulators
// alone would issue dependent MFMAs two apart)
template <int RT, int NV, int H, int R0, bool FIRST>
__device__ __forceinline__ void f8_mfma_dual(const uint4 (&r3)[2][2], const uint4 (&ra)[1][2], const h8 (&mh)[1], const h8 (&ml)[1],
                                             const h8 (&xh)[NV], const h8 (&xl)[NV], f8_f4 (&a3)[2][2], f8_f4 (&va)[1][RT]) {
    const f8_f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int wp = t == 0 ? 1 : 0;                     // weight plane: lo, hi, hi
        const bool lo = t == 1;                            // activation plane: hi, lo, hi
        const bool first = FIRST && t == 0;
        a3[0][H] = F8_MFMA(*reinterpret_cast<const h8*>(&r3[0][wp]), lo ? ml[0] : mh[0], first ? zero : a3[0][H]);
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            va[0][R0 + r] = F8_MFMA(*reinterpret_cast<const h8*>(&ra[0][wp]), lo ? xl[r] : xh[r], first ? zero : va[0][R0 + r]);
            if (r == 0) a3[1][H] = F8_MFMA(*reinterpret_cast<const h8*>(&r3[1][wp]), lo ? ml[0] : mh[0], first ? zero : a3[1][H]);
        }
    }
}

template <int V>
__global__ __launch_bounds__(F8_THREADS, 2) void mlp_fused8_kernel(FusedParams P_arg) {
#define PK P_arg
    struct F8Dbg { long long* dbg; };
    const F8Dbg P{PK.dbg};                                // (the cycle-accounting macros of k_mlp_fused_kernel.h read P.dbg)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* abuf = lds;
    char* mbuf = lds + ABUF_BYTES;
    float* misc = reinterpret_cast<float*>(lds + ABUF_BYTES + MBUF_BYTES);
    float* probs = misc;                       // [32][F8_PSTR]
    float* part = misc + F8_PSTR * 32;         // [3][8 waves][32]  (probs + part take the tile's 3 KB of row records while a
    float* sig = part + F8_PART_FLOATS;        // [32]               filling runs: both are dead then, sig is not)
    int* flag = reinterpret_cast<int*>(sig + 32);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    // One workgroup per tile, XCD-contiguous order: the tiles of an XCD (workgroups b, b + 8, ... share an L2) are CONSECUTIVE tiles of
    // the sample list -- their texel rows overlap, so a row missing in L2 is fetched once per XCD instead of once per tile.
    // (A persistent form -- 256 workgroups claiming 2 .. 255 tiles each from per-XCD counters -- was built and measured: the same
    // launch time when the kernel has the device to itself, and a LONGER frame in the pipeline, because a resident workgroup holds
    // all of a CU's LDS and registers and the other streams' kernels live on the CUs that change hands between tiles:
    // profiles/r06_o_tiles_per_workgroup.txt.)
    const int tile = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int pbase = tile * FM_PTS;
    if (pbase >= PK.P) return;
    const int npts = min(FM_PTS, PK.P - pbase);
    const int fm_dbg_tile = tile;                       // (cycle accounting samples every 16th TILE)
    const int zoff = 0;
    constexpr int ROWS = 32 * V, RT = 2 * V;
    constexpr int RS256 = 16 * STR256;           // bytes between row tiles of a K = 256 plane
    int dbg_i = 1;
    long long dbg_t = 0;
    long long* dbg_keep = reinterpret_cast<long long*>(flag + 2);      // (tile-start stamps of the cycle accounting: LDS, not two register pairs)
    constexpr float inv_v = 1.0f / (float)V;
    unsigned rmax = 0u;
    unsigned seen_s = 0u, seen_p = 0u, seen_n = 0u, seen_i = 0u, seen_4 = 0u;
    char* a256_lo = abuf + ROWS * STR256;

    // ---- texel hand-over (see mlp_fused_kernel: fill_tex); 8 waves: texel row i of the list is copied by wave i & 7, operand
    // row r is blended by wave r & 7 (12 rows per wave at V = 3)
    struct TexPre { unsigned h0, h1; fm_u4 rq; };
    auto tex_fetch = [&]() __attribute__((always_inline)) {
        int tl = tile;
        asm volatile("" : "+s"(tl));
        TexPre t;
        const unsigned* hb = PK.tex_hdr + (long long)tl * 512;
        t.h0 = hb[lane];
        t.h1 = hb[64 + lane];
        t.rq = (fm_u4){0u, 0u, 0u, 0u};
        if (tid < 64 * V) t.rq = *reinterpret_cast<const fm_u4*>(PK.tex_rec + (long long)tl * V * 32 * 8 + tid * 4);
        return t;
    };
    auto fill_tex = [&](const TexPre& pre, auto rgb, auto&& under) __attribute__((always_inline)) {
        constexpr bool RGB = decltype(rgb)::value;
        constexpr int TSTR = 1040, TMAX = 103, NK = (TMAX + 7) / 8, NR = 4 * V;
        static_assert(TMAX * TSTR <= ABUF_BYTES, "a pass of texel rows must fit the operand buffer");
        int wv = __builtin_amdgcn_readfirstlane(wave), tl = tile;
        asm volatile("" : "+s"(wv), "+s"(tl));
        const unsigned* hb = PK.tex_hdr + (long long)tl * 512;
        unsigned h0 = pre.h0, h1 = pre.h1;
        const int npass = __builtin_amdgcn_readfirstlane((int)(h0 >> 16));
        char* recl = reinterpret_cast<char*>(misc);
        static_assert(32 * V * 32 <= (F8_PSTR * 32 + F8_PART_FLOATS) * 4, "row records must fit probs + part");
        unsigned fv[NR][4] = {};
        f32x2 bias_lo = {0.f, 0.f}, bias_hi = {0.f, 0.f};
        if constexpr (!RGB) {
            const float4 b4 = *reinterpret_cast<const float4*>(PK.ar0.bias + 4 * lane);
            bias_lo = (f32x2){b4.x, b4.y};
            bias_hi = (f32x2){b4.z, b4.w};
        }
        for (int p = 0; p < npass; ++p) {
            if (p > 0) {
                FM_SYNCL();
                h0 = hb[p * 128 + lane];
                h1 = hb[p * 128 + 64 + lane];
            }
            const int U = __builtin_amdgcn_readfirstlane((int)(h0 & 0xffffu));
            {
                // rows wv, wv + 8, ...: the first 7 of a wave always (list entries 0 .. 55: header word 8 + i is in h0), 3 more for
                // lists longer than 56, the last 3 for lists longer than 80 (entries 56 ..: word i - 56 of h1)
                constexpr int NA = 7, NB = 10;
                const char* mbase = reinterpret_cast<const char*>(RGB ? PK.tex_map2 : PK.tex_map);
                const unsigned loff = (unsigned)lane * 16u;
                const int last = U - 1;
                fm_u4 ta[NA], tb[NB - NA], tc[NK - NB];
#pragma unroll
                for (int k = 0; k < NA; ++k) {
                    const int i = min(wv + 8 * k, last);
                    const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)h0, 8 + i);
                    ta[k] = *reinterpret_cast<const fm_u4*>(mbase + TX_ADDR(id));
                }
                const bool more = U > 8 * NA, most = U > 8 * NB;
                if (more) {
#pragma unroll
                    for (int k = NA; k < NB; ++k) {
                        const int i = min(wv + 8 * k, last);
                        const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)h1, i - 56);
                        tb[k - NA] = *reinterpret_cast<const fm_u4*>(mbase + TX_ADDR(id));
                    }
                }
                if (most) {
#pragma unroll
                    for (int k = NB; k < NK; ++k) {
                        const int i = min(wv + 8 * k, last);
                        const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)h1, i - 56);
                        tc[k - NB] = *reinterpret_cast<const fm_u4*>(mbase + TX_ADDR(id));
                    }
                }
                if (p == 0) {
                    under();
                    if (tid < 64 * V) *reinterpret_cast<fm_u4*>(recl + tid * 16) = pre.rq;
                }
                int wv2 = wv;
                asm volatile("" : "+s"(wv2));
#pragma unroll
                for (int k = 0; k < NA; ++k) *reinterpret_cast<fm_u4*>(abuf + min(wv2 + 8 * k, last) * TSTR + lane * 16) = ta[k];
                if (more) {
#pragma unroll
                    for (int k = NA; k < NB; ++k) *reinterpret_cast<fm_u4*>(abuf + min(wv2 + 8 * k, last) * TSTR + lane * 16) = tb[k - NA];
                }
                if (most) {
#pragma unroll
                    for (int k = NB; k < NK; ++k) *reinterpret_cast<fm_u4*>(abuf + min(wv2 + 8 * k, last) * TSTR + lane * 16) = tc[k - NB];
                }
            }
            FM_SYNCL();                                      // the texel rows (and the records) are in place
            // operand row wv + 8 k is sample wv + 8 (k & 3): its pass is (k & 3) >> (sh - 1)
            const int sh2 = npass == 1 ? 2 : npass == 2 ? 1 : 0;
            const int cofs = lane * 16;
            struct RowIn { float4 a, b, c, d; fm_u4 q0; };
            auto issue = [&](int k, const fm_u4& o, RowIn& r) __attribute__((always_inline)) {
                r.q0 = *reinterpret_cast<const fm_u4*>(recl + (wv + 8 * k) * 32);
                r.a = *reinterpret_cast<const float4*>(abuf + (o[0] + cofs));
                r.b = *reinterpret_cast<const float4*>(abuf + (o[1] + cofs));
                r.c = *reinterpret_cast<const float4*>(abuf + (o[2] + cofs));
                r.d = *reinterpret_cast<const float4*>(abuf + (o[3] + cofs));
            };
            auto offs = [&](int k) __attribute__((always_inline)) {
                return *reinterpret_cast<const fm_u4*>(recl + (wv + 8 * k) * 32 + 16);
            };
            auto blend = [&](int k, const RowIn& r, auto sel) __attribute__((always_inline)) {
                const unsigned u0 = r.q0[0], u1 = r.q0[1], u2 = r.q0[2], u3 = r.q0[3];
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
                unsigned n0, n1, n2, n3;
                if constexpr (RGB) {
                    const float l0 = lo[0], l1 = lo[1], h0f = hi[0], h1f = hi[1];
                    n0 = __builtin_bit_cast(unsigned, l0); n1 = __builtin_bit_cast(unsigned, l1);
                    n2 = __builtin_bit_cast(unsigned, h0f); n3 = __builtin_bit_cast(unsigned, h1f);
                } else {
                    lo = __builtin_elementwise_max(lo + bias_lo, (f32x2){0.f, 0.f});
                    hi = __builtin_elementwise_max(hi + bias_hi, (f32x2){0.f, 0.f});
                    split_pair(lo[0], lo[1], n0, n2);
                    split_pair(hi[0], hi[1], n1, n3);
                }
                if constexpr (decltype(sel)::value) {
                    const bool mine = ((k & 3) >> sh2) == p;
                    fv[k][0] = mine ? n0 : fv[k][0]; fv[k][1] = mine ? n1 : fv[k][1];
                    fv[k][2] = mine ? n2 : fv[k][2]; fv[k][3] = mine ? n3 : fv[k][3];
                } else {
                    fv[k][0] = n0; fv[k][1] = n1; fv[k][2] = n2; fv[k][3] = n3;
                }
            };
            auto rows_loop = [&](auto sel) __attribute__((always_inline)) {
                RowIn in[2];
                fm_u4 of[2];
                of[0] = offs(0);
                of[1] = offs(1);
                issue(0, of[0], in[0]);
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    if (k + 1 < NR) issue(k + 1, of[(k + 1) & 1], in[(k + 1) & 1]);
                    if (k + 2 < NR) of[k & 1] = offs(k + 2);
                    FM_SB();
                    blend(k, in[k & 1], sel);
                    FM_SB();
                }
            };
            if (npass == 1) rows_loop(std::false_type{});
            else rows_loop(std::true_type{});
        }
        FM_SYNCL();                                          // every wave is done reading texel rows: ABUF takes the result
        if constexpr (RGB) {
#pragma unroll
            for (int k = 0; k < NR; ++k)
                *reinterpret_cast<fm_u4*>(abuf + (wv + 8 * k) * TSTR + lane * 16) = (fm_u4){fv[k][0], fv[k][1], fv[k][2], fv[k][3]};
        } else {
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                range_acc<true>(rmax, fv[k][0]);
                range_acc<true>(rmax, fv[k][1]);
                *reinterpret_cast<uint2*>(abuf + (wv + 8 * k) * STR256 + lane * 8) = make_uint2(fv[k][0], fv[k][1]);
                *reinterpret_cast<uint2*>(a256_lo + (wv + 8 * k) * STR256 + lane * 8) = make_uint2(fv[k][2], fv[k][3]);
            }
            range_commit(PK.range, TH_RANGE_P, seen_p, rmax);
        }
    };

    if (PK.dbg != nullptr && tid == 0 && (tile & 15) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(PK.dbg), 1ull);
        dbg_t = clock64();
        dbg_keep[0] = dbg_t;
        dbg_keep[1] = wall_clock64();
    }
    // range guard: launch-wide maxima as they stand when this tile starts (scalar loads through the constant address space: the
    // table is only a hint here -- a stale smaller value costs an atomic, never a result)
    // All five requested at once (one 16-byte + one 4-byte vector load: coherent with the other workgroups' atomics -- through the
    // scalar cache a workgroup saw the table as it stood when its CU first read it, i.e. zeros after a reset, and every lane of
    // every tile raised atomics: 2.5 x the tile's cycles in render_fast) and moved to scalar registers BEHIND the first barrier.
    // As `readfirstlane(rtab[i])` they had compiled into five load / s_waitcnt vmcnt(0) pairs in a row: five serial L2 round trips
    // in front of every other request of the tile.
    fm_u4 rv4 = {0u, 0u, 0u, 0u};
    unsigned rv1 = 0u;
    static_assert(TH_RANGE_S == 1 && TH_RANGE_P == 2 && TH_RANGE_N == 3 && TH_RANGE_INTER == 4 && TH_RANGE_F4 == 5, "range slots 1..5");
    if (PK.range != nullptr) {
        const unsigned* rtab = PK.range + zoff;
        rv4 = (fm_u4){rtab[1], rtab[2], rtab[3], rtab[4]};
        rv1 = rtab[5];
    }
    TexPre tex_pre = tex_fetch(), tex_pre2;

    // ================= token branch: s = relu(fc_0 h); ks|vs = kv1(s) =================
    // (mlp_fused_kernel, TH_ROWS_NBR form: the 7-neighbour blend of T' rows on the matrix pipe + W_pe pe)
    // (the view-direction rows of the RGB branch sit behind an index: the index is requested in front of fc_2, the rows behind it --
    // both HBM round trips run under the GEMMs; requested at the top of the tile the four registers were spilled across kv0)
    int vsel[2] = {0, 0};
    float vdv[2] = {0.f, 0.f};
    f8_f4 acc2[2][RT];
    {
        constexpr int STOK_STR = 1040;
        static_assert(32 * V * STOK_STR <= ABUF_BYTES, "T' rows must fit the operand buffer");
        char* pe_hi = mbuf;
        char* pe_lo = mbuf + 32 * STR64;
        // pe: 32 rows x (64 hi | 64 lo halves): one 16-byte piece per thread
        const int pt = tid & 255, prow = pt >> 3, pc = pt & 7, ppl = tid >> 8;
        const int psrc = min(prow, npts - 1);
        const uint4 pe_v = *reinterpret_cast<const uint4*>(PK.pe + (long long)(pbase + psrc) * 128 + 64 * ppl + 8 * pc);
        char* wsp_hi = mbuf + 16384;                                     // W [sample][slot] halves, K = 32 per pass
        char* wsp_lo = wsp_hi + 32 * STRVD;
        const unsigned* hdr = reinterpret_cast<const unsigned*>(PK.stok) + (long long)((PK.P + 31) / 32 * 32) * 16 + (long long)tile * 128;
        const unsigned h0 = hdr[lane], h1 = hdr[64 + lane];
        const int ns = tid / 7, nk = tid - 7 * ns;
        int slot = -1;
        float nw = 0.f;
        if (tid < 224) {
            const unsigned* rec = reinterpret_cast<const unsigned*>(PK.stok) + (long long)(pbase + min(ns, npts - 1)) * 16;
            slot = (int)rec[nk];
            nw = __builtin_bit_cast(float, rec[8 + nk]);
        }
        const unsigned zq = 0u;
        const float inv_t = PK.t_inv[zoff];
        uint4 wq[2][2][2];
        const uint4* wl = F8_WSLICE(PK.w16.fc_0pe, wave, 2) + lane;
        f8_load_w<2>(wl, 0, wq[0]);
        f8_load_w<2>(wl, 1, wq[1]);
        const float4 b0[2] = {f8_bias(PK.fc_0pe.bias, wave * 32, lane), f8_bias(PK.fc_0pe.bias, wave * 32 + 16, lane)};
        if (tid < 2 * 32 * STRVD / 16) reinterpret_cast<uint4*>(wsp_hi)[tid] = make_uint4(zq, zq, zq, zq);
        *reinterpret_cast<uint4*>((ppl ? pe_lo : pe_hi) + prow * STR64 + 16 * pc) = pe_v;
        const int U = __builtin_amdgcn_readfirstlane((int)h0);
        FM_SB();
        auto slot_centre = [&](int u) {
            const int d = (2 + u) >> 1;
            const unsigned src = d < 64 ? (unsigned)__builtin_amdgcn_readlane((int)h0, d) : (unsigned)__builtin_amdgcn_readlane((int)h1, d - 64);
            return (int)((src >> (16 * ((2 + u) & 1))) & 0xffffu);
        };
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc2[c][r] = (f8_f4){0.f, 0.f, 0.f, 0.f};
        f8_f4 a1[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int aoffw = f8_aoff<false>(lane, STRVD), aoffp = f8_aoff<false>(lane, STR64);
        for (int u0 = 0; u0 < U; u0 += 32) {
            const int nU = min(32, U - u0);
            if (u0 > 0) {
                FM_SYNCL();
                if (tid < 2 * 32 * STRVD / 16) reinterpret_cast<uint4*>(wsp_hi)[tid] = make_uint4(zq, zq, zq, zq);
            }
            for (int u = wv; u < nU; u += 8) {
                const int cu = slot_centre(u0 + u);
                const char* g = reinterpret_cast<const char*>(PK.tsplit) + (long long)cu * 1024 + lane * 16;
#pragma unroll
                for (int vw = 0; vw < V; ++vw)
                    __builtin_amdgcn_global_load_lds((fm_gptr)(g + (long long)vw * PK.t_nc * 1024),
                                                     (fm_lptr)(abuf + (vw * 32 + u) * STOK_STR), 16, 0, 0);
            }
            FM_SYNCL();                                          // W is cleared (and the pe rows are in place)
            if (slot >= u0 && slot < u0 + 32) {
                _Float16 hi, lo;
                split_h(nw, hi, lo);
                *reinterpret_cast<_Float16*>(wsp_hi + ns * STRVD + 2 * (slot - u0)) = hi;
                *reinterpret_cast<_Float16*>(wsp_lo + ns * STRVD + 2 * (slot - u0)) = lo;
            }
            if (u0 == 0) {                                       // W_pe pe under the row loads
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    h8 xh[2], xl[2];
                    f8_load_x<2, 16 * STR64, false>(pe_hi, pe_lo, aoffp, t, xh, xl);
                    if (t == 0) f8_mfma_half<2, 2, 2, 0, true>(wq[0], xh, xl, a1);
                    else f8_mfma_half<2, 2, 2, 0, false>(wq[1], xh, xl, a1);
                }
            }
            FM_SYNC();                                           // rows (LDS-DMA) and W are in place
            {
                h8 xh[2], xl[2];
                f8_load_x<2, 16 * STRVD, false>(wsp_hi, wsp_lo, aoffw, 0, xh, xl);
                // The A operand is T'^T: lane (channel l15, slot group g4) wants 8 SLOTS of one channel out of rows that hold 256
                // channels of one slot.  ds_read_b64_tr_b16 does the transpose: the 16 lanes of a group each hand in the address
                // of 4 consecutive halves of a [4 slots][16 channels] block (lane i: slot i / 4, channels 4 (i % 4) ..) and lane i
                // receives channel i of the 4 slots (tools/ubench/tr16_semantics.hip) -- two reads per plane and fragment
                // instead of eight 2-byte reads and their packing (slots past the list's end: the last row again, weight 0).
                const int tq = l15 >> 2, tc = 8 * (l15 & 3);
                const int ro0 = min(8 * g4 + tq, nU - 1) * STOK_STR + tc, ro1 = min(8 * g4 + 4 + tq, nU - 1) * STOK_STR + tc;
                typedef __attribute__((address_space(3))) f8_s4* f8_lp;
#pragma unroll
                for (int r = 0; r < V; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const char* rb = abuf + r * 32 * STOK_STR + 2 * (wave * 32 + c * 16);
                        const f8_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro0));
                        const f8_s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro1));
                        const f8_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro0 + 512));
                        const f8_s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((f8_lp)(rb + ro1 + 512));
                        const f8_s8 ahs = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const f8_s8 als = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const h8 ah = __builtin_bit_cast(h8, ahs), al = __builtin_bit_cast(h8, als);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            acc2[c][2 * r + h] = F8_MFMA(al, xh[h], acc2[c][2 * r + h]);
                            acc2[c][2 * r + h] = F8_MFMA(ah, xl[h], acc2[c][2 * r + h]);
                            acc2[c][2 * r + h] = F8_MFMA(ah, xh[h], acc2[c][2 * r + h]);
                        }
                    }
            }
        }
        FM_SYNCL();                                   // every wave is done reading the T' rows: ABUF may take s
        {                                             // (the range table's words have long arrived)
            asm volatile("" : "+v"(rv4), "+v"(rv1));
            seen_s = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[0]);
            seen_p = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[1]);
            seen_n = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[2]);
            seen_i = (unsigned)__builtin_amdgcn_readfirstlane((int)rv4[3]);
            seen_4 = (unsigned)__builtin_amdgcn_readfirstlane((int)rv1);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f8_f4 pe2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) pe2[h] = f8_finish(a1[c][h], b0[c], PK.fc_0pe.inv_scale, false);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f32x2 it2 = {inv_t, inv_t};
                f32x2 u01 = __builtin_elementwise_fma((f32x2){acc2[c][r][0], acc2[c][r][1]}, it2, (f32x2){pe2[r & 1][0], pe2[r & 1][1]});
                f32x2 u23 = __builtin_elementwise_fma((f32x2){acc2[c][r][2], acc2[c][r][3]}, it2, (f32x2){pe2[r & 1][2], pe2[r & 1][3]});
                u01 = __builtin_elementwise_max(u01, (f32x2){0.f, 0.f});
                u23 = __builtin_elementwise_max(u23, (f32x2){0.f, 0.f});
                const f8_f4 u = {u01[0], u01[1], u23[0], u23[1]};
                f8_store_h<STR256>(u, r * 16 + l15, wave * 32 + c * 16, abuf, a256_lo, lane, rmax);
            }
        }
        range_commit(PK.range, TH_RANGE_S, seen_s, rmax);
    }
    uint4 wk3[2][3][2];
    FM_SB();
    f8_load_w<3>(F8_WSLICE(PK.w16.kv1, wave, 3) + lane, 0, wk3[0]);
    FM_SYNCL();
    // kv layers: column tile 0 = key cols 16 wave .., tiles 1, 2 = value cols 128 + 32 wave ..
    f8_f4 vs[2][RT];
    float* ksb = reinterpret_cast<float*>(mbuf);                    // [ROWS][KSTR] fp32 keys of the token branch
    {
        f8_f4 acc3[3][RT];
        f8_gemm_rt<RT, 3, RS256, true, true>(abuf, a256_lo, STR256, F8_WSLICE(PK.w16.kv1, wave, 3), 8, lane, acc3, wk3);
        const float4 bk = f8_bias(PK.kv1.bias, wave * 16, lane), bv0 = f8_bias(PK.kv1.bias, 128 + wave * 32, lane),
                     bv1 = f8_bias(PK.kv1.bias, 128 + wave * 32 + 16, lane);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const f8_f4 k = f8_finish(acc3[0][r], bk, PK.kv1.inv_scale, false);
            *reinterpret_cast<float4*>(ksb + (r * 16 + l15) * KSTR + wave * 16 + 4 * g4) = make_float4(k[0], k[1], k[2], k[3]);
            vs[0][r] = f8_finish(acc3[1][r], bv0, PK.kv1.inv_scale, false);
            vs[1][r] = f8_finish(acc3[2][r], bv1, PK.kv1.inv_scale, false);
        }
    }
    FM_SYNCL();

    // ================= pixel branch: p = relu(alpha_res_0 f) = blend of fold0 rows; kp|vp = kv0(p) =================
    FM_SB();
    fill_tex(tex_pre, std::false_type{}, [] {});
    FM_SB();
    f8_load_w<3>(F8_WSLICE(PK.w16.kv0, wave, 3) + lane, 0, wk3[0]);
    FM_SYNCL();
    f8_f4 vp[2][RT];
    {
        f8_f4 acc3[3][RT];
        f8_gemm_rt<RT, 3, RS256, true, true>(abuf, a256_lo, STR256, F8_WSLICE(PK.w16.kv0, wave, 3), 8, lane, acc3, wk3);
        const float4 bk = f8_bias(PK.kv0.bias, wave * 16, lane), bv0 = f8_bias(PK.kv0.bias, 128 + wave * 32, lane),
                     bv1 = f8_bias(PK.kv0.bias, 128 + wave * 32 + 16, lane);
        FM_SYNCL();                                                  // every wave is done reading p from ABUF
        float* kpb = reinterpret_cast<float*>(abuf);                // [ROWS][KSTR]
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const f8_f4 k = f8_finish(acc3[0][r], bk, PK.kv0.inv_scale, false);
            *reinterpret_cast<float4*>(kpb + (r * 16 + l15) * KSTR + wave * 16 + 4 * g4) = make_float4(k[0], k[1], k[2], k[3]);
            vp[0][r] = f8_finish(acc3[1][r], bv0, PK.kv0.inv_scale, false);
            vp[1][r] = f8_finish(acc3[2][r], bv1, PK.kv0.inv_scale, false);
        }
    }
    FM_SYNCL();

    // ================= cross-view attention (cross_transformer.py:128-149) =================
    {
        const float* kpb = reinterpret_cast<const float*>(abuf);
        float4 bn[2];
        // A[j][i] = kp_j . ks_i / sqrt(128).  Thread (p = tid >> 4, c16 = tid & 15) owns float4 columns c16, c16 + 16 of sample p
        // (packed FMAs over the component pairs); the 16 partials of a sample are summed with four DPP steps.  The softmax over j of
        // column i is formed by lane c16 = i ONLY (selects, no indexed register array): three exponentials per wave instruction
        // stream instead of nine -- with every lane forming all of them this phase took longer on 8 waves than on 4.
        {
            const int p = tid >> 4, c16 = tid & 15;
            f32x2 acc2[V * V];
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) acc2[ji] = (f32x2){0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 kx[V], sx[V];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    kx[v] = *reinterpret_cast<const float4*>(kpb + (v * 32 + p) * KSTR + 4 * (c16 + 16 * q));
                    sx[v] = *reinterpret_cast<const float4*>(ksb + (v * 32 + p) * KSTR + 4 * (c16 + 16 * q));
                }
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        acc2[j * V + i] = __builtin_elementwise_fma((f32x2){kx[j].x, kx[j].y}, (f32x2){sx[i].x, sx[i].y}, acc2[j * V + i]);
                        acc2[j * V + i] = __builtin_elementwise_fma((f32x2){kx[j].z, kx[j].w}, (f32x2){sx[i].z, sx[i].w}, acc2[j * V + i]);
                    }
            }
            float acc[V * V];
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) {
                float s = acc2[ji][0] + acc2[ji][1];
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));  // row_half_mirror
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xF, 0xF, true));  // row_mirror
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
                for (int j = 0; j < V; ++j) probs[p * F8_PSTR + j * V + c16] = e[j] / se;
            }
        }
        bn[0] = f8_bias(PK.fc_1.bias, wave * 32, lane);              // (the round trip runs under the barrier)
        bn[1] = f8_bias(PK.fc_1.bias, wave * 32 + 16, lane);
        FM_SYNCL();
        // fc_1 pre-activation of view i = vs_i + sum_j vp_j A[j][i] + folded bias; relu; -> operand of fc_2 (packed FMAs: two
        // channels per instruction; the probabilities of a sample are 9 consecutive floats: two 16-byte reads and one dword)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* pr = probs + (h * 16 + l15) * F8_PSTR;
            const float4 q0 = *reinterpret_cast<const float4*>(pr), q1 = *reinterpret_cast<const float4*>(pr + 4);
            const float q2 = pr[8];
            const float A[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x2 b01 = {bn[c].x, bn[c].y}, b23 = {bn[c].z, bn[c].w};
                f32x2 t0[V], t1[V];
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    t0[i] = (f32x2){vs[c][2 * i + h][0], vs[c][2 * i + h][1]} + b01;
                    t1[i] = (f32x2){vs[c][2 * i + h][2], vs[c][2 * i + h][3]} + b23;
                }
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const f32x2 v0 = {vp[c][2 * j + h][0], vp[c][2 * j + h][1]}, v1 = {vp[c][2 * j + h][2], vp[c][2 * j + h][3]};
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
                    f8_store_h<STR256>(n, (2 * i + h) * 16 + l15, wave * 32 + c * 16, abuf, a256_lo, lane, rmax);
                }
            }
        }
        range_commit(PK.range, TH_RANGE_N, seen_n, rmax);
    }

    // ================= fc_2 (fc_1 is folded into the value projections) =================
    uint4 wk2[2][2][2];
    FM_SB();
    f8_load_w<2>(F8_WSLICE(PK.w16.fc_2, wave, 2) + lane, 0, wk2[0]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + F8_THREADS * q, row = i >> 5, c = i & 31;
        int x = pbase + row;
        if (PK.vd_sel != nullptr && PK.rgb_all != 2 && c < 27 && row < npts) x = PK.vd_sel[pbase + row];
        vsel[q] = x;
    }
    FM_SYNCL();
    f8_gemm_rt<RT, 2, RS256, true, true>(abuf, a256_lo, STR256, F8_WSLICE(PK.w16.fc_2, wave, 2), 8, lane, acc2, wk2);
    const float4 bi[2] = {f8_bias(PK.fc_2.bias, wave * 32, lane), f8_bias(PK.fc_2.bias, wave * 32 + 16, lane)};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + F8_THREADS * q, row = i >> 5, c = i & 31;
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
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < RT; ++r) acc2[c][r] = f8_finish(acc2[c][r], bi[c], PK.fc_2.inv_scale, true);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f8_f4 m = acc2[c][h];
#pragma unroll
            for (int r = 1; r < V; ++r) m = m + acc2[c][2 * r + h];
            m = m * (f8_f4){inv_v, inv_v, inv_v, inv_v};
            f8_store_h<STR256>(m, h * 16 + l15, wave * 32 + c * 16, mbuf, mbuf + 32 * STR256, lane, rmax);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) f8_store_h<STR256>(acc2[c][r], r * 16 + l15, wave * 32 + c * 16, abuf, a256_lo, lane, rmax);
    }
    range_commit(PK.range, TH_RANGE_INTER, seen_i, rmax);
    FM_SYNCL();

    // ================= sigma head: relu(fc_3 m) . alpha_w + b   ||   folded view_fc on inter =================
    char* vd_hi = mbuf + MBUF_VD_OFF;
    char* vd_lo = vd_hi + 32 * STRVD;
    tex_pre2 = tex_fetch();                              // (for the RGB branch's filling: the round trip runs under fc_3)
    f8_f4 va[1][RT];                                     // this wave's 16 of the 128 view_fc outputs, all rows
    {
        f8_f4 a3[2][2];
        const float4 aw[2] = {f8_bias(PK.alpha_w, wave * 32, lane), f8_bias(PK.alpha_w, wave * 32 + 16, lane)};
        const float4 b3[2] = {f8_bias(PK.fc_3.bias, wave * 32, lane), f8_bias(PK.fc_3.bias, wave * 32 + 16, lane)};
        if (PK.rgb_all != 2) {
            // fc_3 on the 32 mean rows (MBUF) and the folded view_fc on the 32 V rows of inter (ABUF) in ONE loop (see
            // gemm_dual_fc3_vfa): halves A = {mean rows 0-15, inter row tiles 0 .. V-1}, B = {mean rows 16-31, the other V}
            constexpr int T = 8;
            const uint4* w3l = F8_WSLICE(PK.w16.fc_3, wave, 2) + lane;
            const uint4* wal = F8_WSLICE(PK.w16.vfA, wave, 1) + lane;
            const int aoff = f8_aoff<true>(lane, STR256);
            const char* mhi = mbuf;
            const char* mlo = mbuf + 32 * STR256;
            uint4 r3[2][2][2], ra[2][1][2];
            h8 mah[1], mal[1], xah[V], xal[V], mbh[1], mbl[1], xbh[V], xbl[V];
            f8_load_w<2>(w3l, 0, r3[0]);
            f8_load_w<1>(wal, 0, ra[0]);
            f8_load_x<1, RS256, true>(mhi, mlo, aoff, 0, mah, mal);
            f8_load_x<V, RS256, true>(abuf, a256_lo, aoff, 0, xah, xal);
            FM_SB();
            auto step = [&](auto first, auto par, int t) __attribute__((always_inline)) {
                constexpr bool F = decltype(first)::value;
                constexpr int PB = decltype(par)::value;
                const int tn = t + 1 < T ? t + 1 : T - 1;
                f8_load_x<1, RS256, true>(mhi + RS256, mlo + RS256, aoff, t, mbh, mbl);
                f8_load_x<V, RS256, true>(abuf + V * RS256, a256_lo + V * RS256, aoff, t, xbh, xbl);
                f8_load_w<2>(w3l, tn, r3[PB ^ 1]);
                f8_load_w<1>(wal, tn, ra[PB ^ 1]);
                // (the two products interleaved term by term: 2 + V accumulators between two uses of one)
                f8_mfma_dual<RT, V, 0, 0, F>(r3[PB], ra[PB], mah, mal, xah, xal, a3, va);
                f8_interleave<2 + 2 * V + 6, 3 * (2 + V)>();
                f8_load_x<1, RS256, true>(mhi, mlo, aoff, tn, mah, mal);
                f8_load_x<V, RS256, true>(abuf, a256_lo, aoff, tn, xah, xal);
                f8_mfma_dual<RT, V, 1, V, F>(r3[PB], ra[PB], mbh, mbl, xbh, xbl, a3, va);
                f8_interleave<2 + 2 * V, 3 * (2 + V)>();
            };
            step(std::true_type{}, std::integral_constant<int, 0>{}, 0);
            step(std::false_type{}, std::integral_constant<int, 1>{}, 1);
#pragma unroll 1
            for (int t = 2; t < T; t += 2) {
                step(std::false_type{}, std::integral_constant<int, 0>{}, t);
                step(std::false_type{}, std::integral_constant<int, 1>{}, t + 1);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RT; ++r) va[0][r] = (f8_f4){0.f, 0.f, 0.f, 0.f};      // (never read on this path)
            uint4 w3[2][2][2];
            f8_gemm<2, 2, RS256, true, false>(mbuf, mbuf + 32 * STR256, STR256, F8_WSLICE(PK.w16.fc_3, wave, 2), 8, lane, a3, w3);
        }
        float s2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f8_f4 y = f8_finish(a3[c][h], b3[c], PK.fc_3.inv_scale, true);
                s = fmaf(y[0], aw[c].x, s);
                s = fmaf(y[1], aw[c].y, s);
                s = fmaf(y[2], aw[c].z, s);
                s = fmaf(y[3], aw[c].w, s);
            }
            s += f8_xor(s, lane, 16);
            s += f8_xor(s, lane, 32);
            s2[h] = s;
        }
        // (every lane group holds both sums: group 0 stores the first half's samples, group 1 the second half's -- two predicated
        // stores; written as a select between s2[0] and s2[1] the array went to scratch memory and was indexed there)
        if (g4 == 0) part[wave * 32 + l15] = s2[0];
        if (g4 == 1) part[wave * 32 + 16 + l15] = s2[1];
        if (tid == 0) *flag = 0;
        FM_SYNCL();                                  // every wave is done reading the means (MBUF) and inter (ABUF)
        if (tid < 32) {
            float sg = PK.alpha_b[zoff];
#pragma unroll
            for (int w = 0; w < 8; ++w) sg += part[w * 32 + tid];
            sig[tid] = sg;
            if (tid < npts && PK.rgb_all != 2 && (PK.rgb_all == 1 || sg > 0.f)) *flag = 1;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + F8_THREADS * q, row = i >> 5, c = i & 31;
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
        //   t = relu((Wa F) inter + Wd viewdir + blend(fold12[:, :128]) + b') ; u = t + blend(fold12[:, 128:]) + b_R1 ; mean over views ;
        //   fc_4 ; rgb_fc
        uint4 wvd[1][2];
        f8_load_w<1>(F8_WSLICE(PK.w16.vfD, wave, 1) + lane, 0, wvd);
        FM_SB();
        fill_tex(tex_pre2, std::true_type{}, [&]() __attribute__((always_inline)) {
            h8 xh[2], xl[2];
            f8_load_x<2, 16 * STRVD, false>(vd_hi, vd_lo, f8_aoff<false>(lane, STRVD), 0, xh, xl);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                va[0][r] = F8_MFMA(*reinterpret_cast<const h8*>(&wvd[0][1]), xh[r & 1], va[0][r]);
                va[0][r] = F8_MFMA(*reinterpret_cast<const h8*>(&wvd[0][0]), xl[r & 1], va[0][r]);
                va[0][r] = F8_MFMA(*reinterpret_cast<const h8*>(&wvd[0][0]), xh[r & 1], va[0][r]);
            }
        });
        FM_SYNCL();
        // fc_4 weights (4 k-steps) and the rgb_fc rows of this wave's channels: requested before the epilogue
        uint4 w4[4][1][2];
        {
            const uint4* wl4 = F8_WSLICE(PK.w16.fc_4, wave, 1) + lane;
#pragma unroll
            for (int t = 0; t < 4; ++t) f8_load_w<1>(wl4, t, w4[t]);
        }
        float4 rw[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) rw[o] = f8_bias(PK.rgb_w + o * 128, wave * 16, lane);
        const float4 b4 = f8_bias(PK.fc_4.bias, wave * 16, lane);
        const float4 bt = f8_bias(PK.rst.bias, wave * 16, lane), br = f8_bias(PK.rst.bias, 128 + wave * 16, lane);
        char* f4_hi = mbuf + MBUF_FC4_OFF;
        char* f4_lo = f4_hi + 32 * STR128;
        {
            const char* mb = abuf + l15 * 1040 + 4 * (wave * 16 + 4 * g4);
            f8_f4 u[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float4 m1 = *reinterpret_cast<const float4*>(mb + r * 16 * 1040);
                const float4 m2 = *reinterpret_cast<const float4*>(mb + r * 16 * 1040 + 512);
                const f8_f4 t = f8_finish(va[0][r], bt, PK.rst.inv_scale, false);
                const f32x2 z2 = {0.f, 0.f};
                const f32x2 u01 = __builtin_elementwise_max((f32x2){t[0], t[1]} + (f32x2){m1.x, m1.y}, z2) + ((f32x2){m2.x, m2.y} + (f32x2){br.x, br.y});
                const f32x2 u23 = __builtin_elementwise_max((f32x2){t[2], t[3]} + (f32x2){m1.z, m1.w}, z2) + ((f32x2){m2.z, m2.w} + (f32x2){br.z, br.w});
                u[r] = (f8_f4){u01[0], u01[1], u23[0], u23[1]};
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f8_f4 m = u[h];
#pragma unroll
                for (int r = 1; r < V; ++r) m = m + u[2 * r + h];
                m = m * (f8_f4){inv_v, inv_v, inv_v, inv_v};
                f8_store_h<STR128, false>(m, h * 16 + l15, wave * 16, f4_hi, f4_lo, lane, rmax);      // (signed: relu(.) + rgb_res_1)
            }
            range_commit(PK.range, TH_RANGE_F4, seen_4, rmax);
        }
        FM_SYNCL();
        f8_f4 a4[1][2];
        {
            const int aoff4 = f8_aoff<false>(lane, STR128);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                h8 xh4[2], xl4[2];
                f8_load_x<2, 16 * STR128, false>(f4_hi, f4_lo, aoff4, t, xh4, xl4);
                if (t == 0) f8_mfma_half<1, 2, 2, 0, true>(w4[t], xh4, xl4, a4);
                else f8_mfma_half<1, 2, 2, 0, false>(w4[t], xh4, xl4, a4);
            }
        }
        float s3[3][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f8_f4 y = f8_finish(a4[0][h], b4, PK.fc_4.inv_scale, true);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float s = 0.f;
                s = fmaf(y[0], rw[o].x, s);
                s = fmaf(y[1], rw[o].y, s);
                s = fmaf(y[2], rw[o].z, s);
                s = fmaf(y[3], rw[o].w, s);
                s += f8_xor(s, lane, 16);
                s += f8_xor(s, lane, 32);
                s3[o][h] = s;
            }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            if (g4 == 0) part[(o * 8 + wave) * 32 + l15] = s3[o][0];
            if (g4 == 1) part[(o * 8 + wave) * 32 + 16 + l15] = s3[o][1];
        }
        FM_SYNCL();
        if (tid < 32) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                float s = PK.rgb_b[zoff + o];
#pragma unroll
                for (int w = 0; w < 8; ++w) s += part[(o * 8 + w) * 32 + tid];
                rgb_out[o] = s;
            }
        }
    }
    if (tid < npts)
        *reinterpret_cast<float4*>(PK.raw_c + (long long)(pbase + tid) * 4) = make_float4(rgb_out[0], rgb_out[1], rgb_out[2], sig[tid]);
    if (FM_DBG_SAMPLED) {        // dbg[62] / dbg[63]: shader cycles and 100 MHz ticks of the sampled tiles, first stamp to here: the clock INSIDE the launch
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + 62), (unsigned long long)(clock64() - dbg_keep[0]));
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + 63), (unsigned long long)(wall_clock64() - dbg_keep[1]));
    }
#undef PK
}
