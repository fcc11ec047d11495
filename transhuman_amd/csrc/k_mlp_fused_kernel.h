// Device code of the fused per-point MLP (included by k_mlp_fused_host.hip only).
//
// One workgroup = 256 threads = 4 waves (one per SIMD, up to 512 registers each) shades a tile of
// 32 samples x V views.  See k_mlp_fused_host.hip for the arithmetic (fp16 hi/lo split MFMA) and the
// data-flow overview.  LDS map (bytes, V = 3):
//   ABUF  107 520  activation operand of the running GEMM: fp16 hi + lo planes [row = view*32+sample][K],
//                  row stride 2K+16 B (an odd number of 16-B slots: conflict-free ds_read_b128); aliased by
//                  the fp32 key buffer kp
//   MBUF   50 688  ks (fp32 keys of the token branch, parked here so they do not occupy accumulators
//                  during the pixel branch) -> later the view-mean operand of fc_3 -> later viewdir + fc_4 operand
//   MISC    3 744  softmax probabilities, sigma, cross-wave partial sums
// The pixel features f arrive from the producer kernel (k_pixfeat) already split into fp16 hi|lo halves per
// row, so that operand is staged by global_load_lds_dwordx4 (LDS-DMA): no staging registers, no conversion
// pass, no ds_write.  The token-branch input arrives pre-multiplied (see the token branch below).
#pragma once
#include <hip/hip_fp16.h>

#include <utility>

#include "th_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned fm_u4 __attribute__((ext_vector_type(4)));

#define FM_PTS 32
#define STR272 560   // bytes per LDS row = 2K + 16, K = 272 halves (compact pixel-feature rows)
#define STR256 528   // K = 256
#define STR192 400   // K = 192
#define STR128 272   // K = 128
#define STR64 144    // K = 64 (positional-encoding rows of fc_0)
#define STRVD 80     // K = 32 (view-direction block of view_fc)
#define KSTR 132     // floats per row of the fp32 key buffers

#define ABUF_BYTES (2 * 96 * STR272)
#define MBUF_BYTES (96 * KSTR * 4)
#define MISC_FLOATS (9 * 32 + 128 + 4 * 32 * 4 + 8)
#define FUSED_LDS_BYTES (ABUF_BYTES + MBUF_BYTES + MISC_FLOATS * 4)
#define MBUF_VD_OFF 0
#define MBUF_FC4_OFF 8192

// barrier + optional cycle accounting (developer aid: TH_FUSED_DBG=1 prints the average cycles between
// consecutive barriers over every 16th tile of one launch)
#define FM_SYNC_(BARRIER)                                                                                 \
    do {                                                                                                 \
        BARRIER;                                                                                         \
        if (FM_DBG_SAMPLED) {                                                                            \
            long long now_ = clock64();                                                                  \
            atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + dbg_i++), (unsigned long long)(now_ - dbg_t)); \
            dbg_t = now_;                                                                                \
        }                                                                                                \
    } while (0)
// which workgroups are sampled: a variable `fm_dbg_tile` of the enclosing kernel (its tile number)
#define FM_DBG_SAMPLED (P.dbg != nullptr && tid == 0 && (fm_dbg_tile & 15) == 0)
// full barrier (waits for every outstanding memory operation: needed behind LDS-DMA staging, whose LDS writes are
// tracked by vmcnt)
#define FM_SYNC() FM_SYNC_(__syncthreads())
// LDS-only barrier: orders LDS traffic (s_waitcnt lgkmcnt(0) + s_barrier) and leaves global loads in flight -- bias /
// head-row / weight-fragment requests issued in front of it keep travelling while the waves meet
#define FM_LDS_BARRIER()                                                      \
    do {                                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");       \
        __builtin_amdgcn_s_barrier();                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");       \
    } while (0)
#define FM_SYNCL() FM_SYNC_(FM_LDS_BARRIER())

// cycle stamp without a barrier (same accounting as FM_SYNC: wave 0 of every 16th tile)
#define FM_STAMP()                                                                                       \
    do {                                                                                                 \
        if (FM_DBG_SAMPLED) {                                                                            \
            long long now_ = clock64();                                                                  \
            atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + dbg_i++), (unsigned long long)(now_ - dbg_t)); \
            dbg_t = now_;                                                                                \
        }                                                                                                \
    } while (0)

__device__ __forceinline__ void split_h(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two values -> packed hi pair + packed lo pair: one v_cvt_pk_f16_f32 and two mixed-precision FMAs
// (lo = fp16(x - hi), the f16 -> f32 widening of hi, the subtraction and the narrowing in ONE instruction each;
// x - hi is exact in fp32, so this is bit-identical to split_h at 1.5 instead of 3.25 VALU instructions per value)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    h2v h;
    h[0] = (_Float16)x0;
    h[1] = (_Float16)x1;
    hi = *reinterpret_cast<unsigned*>(&h);
    asm("" : "+v"(hi));          // one packed register from here on (keeps the compiler from re-deriving the halves)
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
    lo = l;
}

// ---- global -> LDS staging by LDS-DMA ------------------------------------------------------------------
// Source rows (one per (sample, view)) hold KROW / 8 groups of [8 hi halves | 8 lo halves].  Columns
// [coff, coff + KC) of both planes are copied into the LDS images hi/lo (row = view*32 + sample, stride STR
// bytes).  global_load_lds_dwordx4 writes LDS at a wave-uniform base + lane*16, reading a per-lane global
// address: every wave fills 1 KiB slices of the plane image and each lane derives the (row, column) its
// 16-byte slot belongs to; slots in the 16-byte row pad fetch column 0 (never read back).  Rows past the
// end of a ragged last tile re-read the last valid sample (their results are never stored).
typedef __attribute__((address_space(1))) const void* fm_gptr;
typedef __attribute__((address_space(3))) void* fm_lptr;

template <int V, int KROW, int KC, int STR>
__device__ __forceinline__ void stage_glds(const _Float16* __restrict__ src, int coff, int pbase, int npts,
                                           char* __restrict__ hi, char* __restrict__ lo, int wave, int lane) {
    // 16-byte slots: SL per LDS row (the last one is the pad), DS of them carry data; a plane image is NS slots,
    // filled 64 slots (1 KiB) per load.  A lane's slot index advances by 256 per round of the four waves: its
    // (row, slot) pair is stepped incrementally (one division per call instead of one per load -- the address
    // arithmetic, not the memory system, used to bound this loop: 45 VALU instructions per load under a vector
    // loop counter, 8-9 k cycles per filling).
    constexpr int SL = STR / 16, DS = (2 * KC) / 16, NS = 32 * V * SL;
    constexpr int NCH = (NS + 63) / 64;
    constexpr int DR = 256 / SL, DSL = 256 % SL;
    static_assert(STR % 16 == 0 && (2 * KC) % 16 == 0 && DS < SL, "row stride must hold the data slots plus a pad slot");
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int q0 = wv * 64 + lane;
    int row = q0 / SL, slot = q0 - row * SL;
    // (source rows are groups of 8 channels, 32 bytes = [8 hi halves | 8 lo halves]: k_pixfeat.hip pg_store8)
    static_assert(KC % 8 == 0 && KROW % 8 == 0, "rows are made of 8-channel groups");
    const char* gbase = reinterpret_cast<const char*>(src) + 4 * coff;
#pragma unroll 1
    for (int c = wv; c < NCH; c += 4) {
        if (c * 64 + lane < NS) {
            int p = row & 31;
            const int vw = row >> 5;
            p = p < npts ? p : npts - 1;
            const int col = slot < DS ? slot * 32 : 0;
            const char* g = gbase + (long long)((pbase + p) * V + vw) * (4 * KROW) + col;
#define FM_STAGE_AUX 0      // cache-policy bits of the staging loads (1 = sc0, 2 = nt, 16 = sc1); nt / nt + sc1: +5 % per tile
                            // (135.5 k vs 130 k cycles): the RGB branch's second filling lives on what the first left in L2
            __builtin_amdgcn_global_load_lds((fm_gptr)g, (fm_lptr)(hi + c * 1024), 16, 0, FM_STAGE_AUX);
            __builtin_amdgcn_global_load_lds((fm_gptr)(g + 16), (fm_lptr)(lo + c * 1024), 16, 0, FM_STAGE_AUX);
        }
        row += DR;
        slot += DSL;
        if (slot >= SL) { slot -= SL; row += 1; }
    }
}

// ---- GEMM phase ---------------------------------------------------------------------------------------
template <int RT, int STR, int ROWSTEP>
__device__ __forceinline__ void load_xfrag(const char* __restrict__ ahi, const char* __restrict__ alo, int aoff, int kb,
                                           h8 (&xh)[RT], h8 (&xl)[RT]) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        xh[r] = *reinterpret_cast<const h8*>(ahi + r * ROWSTEP + aoff + kb * 32);
        xl[r] = *reinterpret_cast<const h8*>(alo + r * ROWSTEP + aoff + kb * 32);
    }
}

// one k-block (16 deep): acc[c][r] += W(c) * X(r)^T as three fp16 products (lo*hi, hi*lo, hi*hi),
// term-major so consecutive MFMAs hit different accumulators
// FIRST (bit c = column tile c): those accumulators start from the inline constant 0 (no zero-initialisation pass:
// 16 v_accvgpr_write per tile)
template <int RT, int CT, int FIRST = 0>
__device__ __forceinline__ void mfma_kblock(const uint4 (&w)[CT][2], const h8 (&xh)[RT], const h8 (&xl)[RT],
                                            f32x16 (&acc)[CT][RT]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&w[c][1]), xh[r],
                                                               ((FIRST >> c) & 1) ? zero : acc[c][r], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&w[c][0]), xl[r], acc[c][r], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
            acc[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&w[c][0]), xh[r], acc[c][r], 0, 0, 0);
}

template <int CT>
__device__ __forceinline__ void load_wfrag(const uint4* __restrict__ wl, int kb, uint4 (&w)[CT][2]) {
    const uint4* p = wl + (long long)kb * (CT * 2 * 64);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        w[c][0] = p[(c * 2 + 0) * 64];
        w[c][1] = p[(c * 2 + 1) * 64];
    }
}

#define FM_SB() __builtin_amdgcn_sched_barrier(0)

// Layout of the pixel-feature rows f and how their K range maps onto ABUF fillings:
//   FM = 0  full rows, K = 384 (pixel_feat_map as the reference builds it): two fillings 192 + 192, 12 + 12 k-blocks
//   FM = 1  compact rows, K = 272 = 256 latent | r g b | 0 (colour lift folded into the weights): ONE filling,
//           17 k-blocks (the second "filling" is empty)
// LDS row strides are 2K + 16 B, i.e. an odd number of 16-byte slots: the 16 lanes of a ds_read_b128 lane
// group (16 different rows, same column) then hit 16 different slots of the 256-byte bank row.
template <int FM> struct FLay;
template <> struct FLay<0> { static constexpr int LD = 384, KA = 192, KB2 = 192, SA = STR192, SB = STR192, NA = 12, NB = 12; };
template <> struct FLay<1> { static constexpr int LD = 272, KA = 272, KB2 = 0, SA = STR272, SB = STR272, NA = 17, NB = 0; };

// acc[ct][rt] += W_tile(ct) * A_rows(rt)^T over KB k-blocks of 16.
//  * weight fragments stream from the (L2-resident) packed image through a ring of D register sets: block
//    k+D-1 is requested before the MFMA burst of block k (sched_barrier pins the loads there) -- D-1 bursts
//    of latency tolerance, no register copies (loop unrolled by D, D even);
//  * activation fragments (LDS, ds_read_b128) ping-pong between two register sets: block k+1 is read
//    while block k multiplies, so no burst starts with an exposed LDS round trip.
#define FM_RING_D 4       // weight ring depth of the 3-column-tile (key/value) phases: register-tight
#define FM_RING_D2 4      // ... of the 1- and 2-column-tile phases
template <int CT, int RT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[CT][RT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][r][e] = 0.f;
}

struct FmNoStamp { __device__ __forceinline__ void operator()() const {} };

// Weight ring of a GEMM phase.  A wave's dwordx4 load costs the CU's address unit ~16 cycles whatever it hits, so the
// D-1 blocks a steady-state ring keeps in flight are NOT requested up front (3 x 2*CT loads x 4 waves = 1.1 k cycles
// of address-unit time in front of the first MFMA at CT = 3): only block 0 is, and the first ring step requests blocks
// 1 .. D-1 in the shadow of block 0's MFMAs (RAMP).  Block 0 does not depend on the LDS operand: a caller may request
// it BEFORE the barrier in front of the phase (ring_prefetch0 + PRE) -- the barriers around the GEMM phases order LDS
// only (FM_SYNCL), so the L2 round trip runs under the barrier.
template <int CT, int D>
__device__ __forceinline__ void ring_prefetch0(const uint4* __restrict__ wp, int lane, uint4 (&w)[D][CT][2]) {
    load_wfrag<CT>(wp + lane, 0, w[0]);
}
// (phases shorter than the ring, KB < D: everything up front)
template <int CT, int D>
__device__ __forceinline__ void ring_prefetch(const uint4* __restrict__ wp, int KB, int lane, uint4 (&w)[D][CT][2]) {
    const uint4* wl = wp + lane;
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
        if (j < KB) load_wfrag<CT>(wl, j, w[j]);
}

// one ring step: request block kb+j+D-1 (weights; RAMP: blocks 1 .. D-1) and kb+j+1 (activations), multiply block kb+j
template <int RT, int CT, int STR, int ROWSTEP, int D, bool XPP, int FIRST, int J, bool RAMP>
__device__ __forceinline__ void ring_step(const char* __restrict__ ahi, const char* __restrict__ alo, const uint4* __restrict__ wl,
                                          int aoff, int kb, int KB, uint4 (&w)[D][CT][2], h8 (&xh)[2][RT], h8 (&xl)[2][RT],
                                          f32x16 (&acc)[CT][RT]) {
    constexpr int j = J;
    // branch-free (indices clamped: the last blocks re-request a fragment nobody consumes) so that
    // the prefetch and the MFMA burst form ONE scheduling region, then ask for one memory
    // instruction after each of the first MFMAs: the 2*CT global loads + 2*RT ds_reads issue in the
    // shadow of running MFMAs instead of in front of the burst.
    const int kx = (kb + j + 1 < KB) ? kb + j + 1 : KB - 1;
    if (RAMP) {
#pragma unroll
        for (int q = 1; q < D; ++q) load_wfrag<CT>(wl, q < KB ? q : KB - 1, w[q]);
    } else {
        const int kw = (kb + j + D - 1 < KB) ? kb + j + D - 1 : KB - 1;
        load_wfrag<CT>(wl, kw, w[(j + D - 1) % D]);
    }
    if (XPP) {
        load_xfrag<RT, STR, ROWSTEP>(ahi, alo, aoff, kx, xh[(j + 1) & 1], xl[(j + 1) & 1]);
        mfma_kblock<RT, CT, FIRST>(w[j], xh[j & 1], xl[j & 1], acc);
    } else {   // register-tight phases: one activation set, read right before its burst
        load_xfrag<RT, STR, ROWSTEP>(ahi, alo, aoff, kb + j, xh[0], xl[0]);
        mfma_kblock<RT, CT, FIRST>(w[j], xh[0], xl[0], acc);
    }
    constexpr int NMEM = 2 * CT * (RAMP ? D - 1 : 1) + 2 * RT, NMF = 3 * CT * RT;
    if constexpr (NMEM <= NMF) {
#pragma unroll
        for (int q = 0; q < NMEM; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);     // one VMEM read or DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - NMEM, 0);
    } else {        // more memory instructions than MFMAs (ramp-up of a small tile): spread evenly
        constexpr int BASE = NMEM / NMF, EXTRA = NMEM % NMF;
#pragma unroll
        for (int q = 0; q < EXTRA; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x120, BASE + 1, 0);
        }
#pragma unroll
        for (int q = EXTRA; q < NMF; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x120, BASE, 0);
        }
    }
    FM_SB();
}

template <int RT, int CT, int STR, int ROWSTEP, int D, bool XPP, int FIRST, bool RAMP, int... Js>
__device__ __forceinline__ void ring_round(std::integer_sequence<int, Js...>, const char* __restrict__ ahi,
                                           const char* __restrict__ alo, const uint4* __restrict__ wl, int aoff, int kb, int KB,
                                           uint4 (&w)[D][CT][2], h8 (&xh)[2][RT], h8 (&xl)[2][RT], f32x16 (&acc)[CT][RT]) {
    (ring_step<RT, CT, STR, ROWSTEP, D, XPP, (Js == 0 ? FIRST : 0), Js, (RAMP && Js == 0)>(ahi, alo, wl, aoff, kb, KB, w, xh, xl, acc), ...);
}

//  * ZMASK (bit c = column tile c): those tiles are acc = W * A^T, the others accumulate.  The first MFMA of a ZMASK
//    accumulator takes the inline constant 0 as C (no accumulator clearing pass).
//  * RAMP (KB >= D): the first ring round is peeled out of the loop: it carries the ZMASK MFMAs and the ring ramp-up.
//    Without it (KB < D) the phase is the up-front prefetch plus the tail.
//  * PRE: the caller has already requested block 0 (ring_prefetch0) / the first D-1 blocks (ring_prefetch, !RAMP).
template <int RT, int CT, int STR, int ROWSTEP, int D, bool XPP, int ZMASK, bool PRE, bool RAMP = true, class ST = FmNoStamp>
__device__ __forceinline__ void gemm_phase_core(const char* __restrict__ ahi, const char* __restrict__ alo,
                                                const uint4* __restrict__ wp, int KB, int lane, f32x16 (&acc)[CT][RT],
                                                uint4 (&w)[D][CT][2], ST stamp = ST()) {
    static_assert((D & 1) == 0, "ring depth must be even (activation ping-pong parity)");
    static_assert(RAMP || ZMASK == 0, "zero-start tiles need the peeled first round");
    const uint4* wl = wp + lane;     // this wave's stream: per kb: CT x {hi, lo} x 64 lanes x 16 B
    const int aoff = (lane & 31) * STR + (lane >> 5) * 16;
    h8 xh[2][RT], xl[2][RT];
    if (!PRE) {
        if (RAMP) ring_prefetch0<CT, D>(wp, lane, w);
        else ring_prefetch<CT, D>(wp, KB, lane, w);
    }
    load_xfrag<RT, STR, ROWSTEP>(ahi, alo, aoff, 0, xh[0], xl[0]);
    FM_SB();
    stamp();
    int kb = 0;
    if (RAMP) {
        ring_round<RT, CT, STR, ROWSTEP, D, XPP, ZMASK, true>(std::make_integer_sequence<int, D>{}, ahi, alo, wl, aoff, 0, KB, w, xh, xl, acc);
        kb = D;
    }
#pragma unroll 1
    for (; kb + D <= KB; kb += D)
        ring_round<RT, CT, STR, ROWSTEP, D, XPP, 0, false>(std::make_integer_sequence<int, D>{}, ahi, alo, wl, aoff, kb, KB, w, xh, xl, acc);
    stamp();
    // tail (KB % D blocks): their weight fragments were requested by the clamped loads above
#pragma unroll
    for (int j = 0; j < D - 1; ++j)
        if (kb + j < KB) {
            if (XPP) {
                if (kb + j + 1 < KB) load_xfrag<RT, STR, ROWSTEP>(ahi, alo, aoff, kb + j + 1, xh[(j + 1) & 1], xl[(j + 1) & 1]);
                FM_SB();
                mfma_kblock<RT, CT>(w[j], xh[j & 1], xl[j & 1], acc);
            } else {
                load_xfrag<RT, STR, ROWSTEP>(ahi, alo, aoff, kb + j, xh[0], xl[0]);
                mfma_kblock<RT, CT>(w[j], xh[0], xl[0], acc);
            }
            FM_SB();
        }
}

template <int RT, int CT, int STR, int ROWSTEP, int D, bool XPP, bool ZERO, bool RAMP = true>
__device__ __forceinline__ void gemm_phase_impl(const char* __restrict__ ahi, const char* __restrict__ alo,
                                                const uint4* __restrict__ wp, int KB, int lane, f32x16 (&acc)[CT][RT]) {
    uint4 w[D][CT][2];
    gemm_phase_core<RT, CT, STR, ROWSTEP, D, XPP, (ZERO ? (1 << CT) - 1 : 0), false, RAMP>(ahi, alo, wp, KB, lane, acc, w);
}

// acc += W * A^T
template <int RT, int CT, int STR, int ROWSTEP = 32 * STR, int D = (CT >= 3 ? FM_RING_D : FM_RING_D2), bool XPP = true>
__device__ __forceinline__ void gemm_phase(const char* __restrict__ ahi, const char* __restrict__ alo,
                                           const uint4* __restrict__ wp, int KB, int lane, f32x16 (&acc)[CT][RT]) {
    gemm_phase_impl<RT, CT, STR, ROWSTEP, D, XPP, false>(ahi, alo, wp, KB, lane, acc);
}
// acc = W * A^T
template <int RT, int CT, int STR, int ROWSTEP = 32 * STR, int D = (CT >= 3 ? FM_RING_D : FM_RING_D2), bool XPP = true>
__device__ __forceinline__ void gemm_phase_z(const char* __restrict__ ahi, const char* __restrict__ alo,
                                             const uint4* __restrict__ wp, int KB, int lane, f32x16 (&acc)[CT][RT]) {
    gemm_phase_impl<RT, CT, STR, ROWSTEP, D, XPP, true>(ahi, alo, wp, KB, lane, acc);
}

// Two independent products in ONE software-pipelined loop over 16 k-blocks (K = 256):
//   a3 (2 column tiles x 1 row tile)  += W3 * M^T      fc_3 on the 32 view-mean rows (MBUF)
//   va (1 column tile  x V row tiles) += WA * X^T      folded view_fc (Wa F) on the 32*V rows of inter (ABUF)
// fc_3 alone re-uses every weight fragment on ONE row tile: its 256 KB of weights per tile ask for 85 B/clk/CU
// of L2 bandwidth and the phase took 11 k cycles for 3 k cycles of MFMA issue.  Interleaved with the (MFMA-heavy,
// 3 row tiles per fragment) view_fc product the pair streams 6 KB of weights per 15 MFMAs per wave (50 B/clk/CU).
template <int V>
__device__ __forceinline__ void gemm_dual_fc3_vfa(const char* __restrict__ mhi, const char* __restrict__ mlo,
                                                  const char* __restrict__ xhi, const char* __restrict__ xlo,
                                                  const uint4* __restrict__ w3, const uint4* __restrict__ wa, int lane,
                                                  f32x16 (&a3)[2][1], f32x16 (&va)[1][V]) {
    constexpr int KB = 16, D = 4;
    const uint4* w3l = w3 + lane;
    const uint4* wal = wa + lane;
    const int aoff = (lane & 31) * STR256 + (lane >> 5) * 16;
    uint4 r3[D][2][2], ra[D][1][2];
    h8 mh[2][1], ml[2][1], xh[2][V], xl[2][V];
    // ring ramp-up as in gemm_phase_core: block 0 up front, blocks 1 .. D-1 under the MFMAs of block 0
    load_wfrag<2>(w3l, 0, r3[0]);
    load_wfrag<1>(wal, 0, ra[0]);
    load_xfrag<1, STR256, 0>(mhi, mlo, aoff, 0, mh[0], ml[0]);
    load_xfrag<V, STR256, 32 * STR256>(xhi, xlo, aoff, 0, xh[0], xl[0]);
    FM_SB();
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // (the first MFMA of every accumulator takes the inline constant 0 as C: no clearing pass;
    // fully unrolled: as a rolled loop the five loop-carried accumulator tiles were copied between AGPR ranges
    // every iteration -- 413 v_accvgpr_mov in the body -- and the phase ran at half its MFMA rate)
#pragma unroll
    for (int kb = 0; kb < KB; kb += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int kw = (kb + j + D - 1 < KB) ? kb + j + D - 1 : KB - 1;
            const int kx = (kb + j + 1 < KB) ? kb + j + 1 : KB - 1;
            const bool ramp = kb == 0 && j == 0;
            if (ramp) {
#pragma unroll
                for (int q = 1; q < D; ++q) {
                    load_wfrag<2>(w3l, q, r3[q]);
                    load_wfrag<1>(wal, q, ra[q]);
                }
            } else {
                load_wfrag<2>(w3l, kw, r3[(j + D - 1) % D]);
                load_wfrag<1>(wal, kw, ra[(j + D - 1) % D]);
            }
            load_xfrag<1, STR256, 0>(mhi, mlo, aoff, kx, mh[(j + 1) & 1], ml[(j + 1) & 1]);
            load_xfrag<V, STR256, 32 * STR256>(xhi, xlo, aoff, kx, xh[(j + 1) & 1], xl[(j + 1) & 1]);
            // term-major over the 2 + V accumulators (an accumulator recurs every 2 + V MFMAs: the 2-accumulator fc_3
            // block alone would issue dependent MFMAs back to back)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int wp = t == 0 ? 1 : 0;                    // weight plane: lo, hi, hi
                const bool xlo = t == 1;                           // activation plane: hi, lo, hi
                const bool first = kb == 0 && j == 0 && t == 0;
                a3[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&r3[j][0][wp]),
                                                                  xlo ? ml[j & 1][0] : mh[j & 1][0], first ? zero : a3[0][0], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < V; ++r) {
                    va[0][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&ra[j][0][wp]),
                                                                      xlo ? xl[j & 1][r] : xh[j & 1][r], first ? zero : va[0][r], 0, 0, 0);
                    if (r == 0)
                        a3[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const h8*>(&r3[j][1][wp]),
                                                                          xlo ? ml[j & 1][0] : mh[j & 1][0], first ? zero : a3[1][0], 0, 0, 0);
                }
            }
            constexpr int NMF = 6 + 3 * V;
            if (ramp) {     // 6 * (D - 1) + 2 + 2 V memory instructions, spread evenly behind the MFMAs
                constexpr int NMEM = 6 * (D - 1) + 2 + 2 * V;
                constexpr int BASE = NMEM / NMF, EXTRA = NMEM % NMF;
#pragma unroll
                for (int q = 0; q < EXTRA; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x120, BASE + 1, 0);
                }
#pragma unroll
                for (int q = EXTRA; q < NMF; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x120, BASE, 0);
                }
            } else {
                constexpr int NMEM = 6 + 2 + 2 * V;
                constexpr int NPAIR = NMEM < NMF ? NMEM : NMF;
#pragma unroll
                for (int q = 0; q < NPAIR; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);     // one VMEM read or DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMF - NPAIR, 0);
            }
            FM_SB();
        }
    }
}

// channel of accumulator register e (within a 32-wide column tile) for this lane
__device__ __forceinline__ int acc_chan(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// y = acc*inv_scale + bias (per output channel), optional relu, in place
template <int RT>
__device__ __forceinline__ void finish_tile(f32x16 (&acc)[RT], const float* __restrict__ bias, int col0, float inv_scale,
                                            bool relu, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 b = *reinterpret_cast<const float4*>(bias + col0 + 8 * g + 4 * (lane >> 5));
        float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float y = acc[r][4 * g + q] * inv_scale + bb[q];
                acc[r][4 * g + q] = relu ? fmaxf(y, 0.f) : y;
            }
    }
}

// The bias values a lane needs for one 32-wide column tile: 4 groups of 4 consecutive channels.  They are
// requested as one batch BEFORE the barrier / arithmetic in front of the epilogue (the per-group loads inside
// finish_tile used to be waited for one by one: 8-12 exposed L2 round trips per epilogue).
struct BiasT { float4 g[4]; };
__device__ __forceinline__ BiasT load_bias(const float* __restrict__ bias, int col0, int lane) {
    BiasT b;
#pragma unroll
    for (int g = 0; g < 4; ++g) b.g[g] = *reinterpret_cast<const float4*>(bias + col0 + 8 * g + 4 * (lane >> 5));
    return b;
}
// y = acc*inv_scale + bias with a preloaded bias (inv_scale is a power of two: the fused form rounds like mul + add)
template <int RT>
__device__ __forceinline__ void finish_tile_b(f32x16 (&acc)[RT], const BiasT& b, float inv_scale, bool relu) {
    const f32x2 sc = {inv_scale, inv_scale}, zero = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x2 b01 = {b.g[g].x, b.g[g].y}, b23 = {b.g[g].z, b.g[g].w};
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            f32x2 y01 = {acc[r][4 * g], acc[r][4 * g + 1]}, y23 = {acc[r][4 * g + 2], acc[r][4 * g + 3]};
            y01 = __builtin_elementwise_fma(y01, sc, b01);
            y23 = __builtin_elementwise_fma(y23, sc, b23);
            if (relu) { y01 = __builtin_elementwise_max(y01, zero); y23 = __builtin_elementwise_max(y23, zero); }
            acc[r][4 * g] = y01[0]; acc[r][4 * g + 1] = y01[1];
            acc[r][4 * g + 2] = y23[0]; acc[r][4 * g + 3] = y23[1];
        }
    }
}

// write one 32x32 output tile (this lane: row, 4 groups of 4 consecutive channels) as hi/lo halves
// Range guard of the fp16 hi/lo split (x = hi + lo needs |x| < 65504; below 2^-14 the halves are subnormal and the
// absolute resolution stops at 2^-25).  Every value that is split passes through store_tile_h: the |hi| halves
// (as 15-bit integers: inf and NaN order above every finite value) are folded into a per-lane running maximum
// with one v_and + one v_pk_max_u16 per PAIR of values, and after each activation the lane maximum is merged into
// the launch-wide table P.range[slot] -- with an atomic only when it exceeds the value read at kernel start, i.e.
// almost never after the first tiles.  The host reads the table per frame (th_range_read) and re-renders on the
// fp32 MFMA path when a slot reached 6e4 (overflow / NaN) or stayed below 2^-6 (resolution).
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
template <bool NONNEG = false>
__device__ __forceinline__ void range_acc(unsigned& rm, unsigned hi2) {
    unsigned a = NONNEG ? hi2 : (hi2 & 0x7fff7fffu);      // relu outputs carry no sign bit: no masking needed
    us2v m = __builtin_elementwise_max(*reinterpret_cast<us2v*>(&rm), *reinterpret_cast<us2v*>(&a));
    rm = *reinterpret_cast<unsigned*>(&m);
}
__device__ __forceinline__ void range_commit(unsigned* __restrict__ table, int slot, unsigned seen, unsigned& rm) {
    const unsigned m = max(rm & 0xffffu, rm >> 16);
    if (table != nullptr && m > seen) atomicMax(table + slot, m);
    rm = 0u;
}

template <int STR, bool NONNEG = true>
__device__ __forceinline__ void store_tile_h(const f32x16& t, int row, int col0, char* __restrict__ hi,
                                             char* __restrict__ lo, int lane, unsigned& rm) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int c = col0 + 8 * g + 4 * (lane >> 5);
        uint2 a, b;
        split_pair(t[4 * g], t[4 * g + 1], a.x, b.x);
        split_pair(t[4 * g + 2], t[4 * g + 3], a.y, b.y);
        range_acc<NONNEG>(rm, a.x);
        range_acc<NONNEG>(rm, a.y);
        *reinterpret_cast<uint2*>(hi + row * STR + 2 * c) = a;
        *reinterpret_cast<uint2*>(lo + row * STR + 2 * c) = b;
    }
}
// same tile as fp32 (key buffers for the cross-view dots)
__device__ __forceinline__ void store_tile_f(const f32x16& t, int row, int col0, float* __restrict__ dst, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int c = col0 + 8 * g + 4 * (lane >> 5);
        *reinterpret_cast<float4*>(dst + row * KSTR + c) =
            make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
    }
    FM_SB();
}

// this wave's slice of a packed layer (+ k-block offset)
__device__ __forceinline__ const uint4* wslice(const FusedLayer& L, int wave, int ct, int kb0) {
    return L.w + ((long long)wave * L.KB + kb0) * (ct * 2 * 64);
}

// FM_NUM_VGPR (build experiment): cap the architectural VGPRs so that the wave's unified allocation (VGPRs + AGPRs)
// stays below 512 and a low-register kernel of another stream can become co-resident on the same SIMDs
#define FM_VGPR_ATTR
template <int V, int FM, bool TEX = false>
__global__ __launch_bounds__(256, 1) FM_VGPR_ATTR void mlp_fused_kernel(FusedParams P) {
    using FL = FLay<FM>;
    static_assert(!TEX || FM == 1, "the texel hand-over produces compact (272-wide) operand planes");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* abuf = lds;
    char* mbuf = lds + ABUF_BYTES;
    float* misc = reinterpret_cast<float*>(lds + ABUF_BYTES + MBUF_BYTES);
    float* probs = misc;                  // [V*V][32]
    float* part = misc + 9 * 32;          // [4 waves][32][4]   (TEX: probs + part take the tile's 3 KB of row records while
    float* sig = part + 4 * 32 * 4;       // [32] (+ padding)    a filling runs: both are dead then, sig is not)
    int* flag = reinterpret_cast<int*>(sig + 128);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // TEX: the tiles of an XCD (workgroups b, b + 8, b + 16 ... share an L2) are CONSECUTIVE tiles of the sample list -- their
    // texel rows overlap (adjacent depths of the same rays, neighbouring ray groups), so a row missing in L2 is fetched
    // once per XCD instead of once per tile.  (The launcher rounds the grid up to a multiple of 8.)
    const int tile = TEX ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int pbase = tile * FM_PTS;
    if (TEX && pbase >= P.P) return;
    const int npts = min(FM_PTS, P.P - pbase);
    constexpr int ROWS = 32 * V;
    const int myrow = lane & 31;
    int dbg_i = 1;                      // dbg[0] counts sampled tiles, dbg[i] accumulates the cycles of interval i
    long long dbg_t = 0, dbg_w = 0, dbg_t0 = 0;
    const int fm_dbg_tile = (int)blockIdx.x;
    if (P.dbg != nullptr && tid == 0 && (blockIdx.x & 15) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg), 1ull);
        dbg_t = clock64();
        dbg_t0 = dbg_t;
        dbg_w = wall_clock64();
    }

    // view means multiply by 1/V (one rounding away from torch.mean's division; 10 VALU instructions less per value)
    constexpr float inv_v = 1.0f / (float)V;
    // range guard: launch-wide maxima as they stood when this tile started (uniform -> scalar loads)
    unsigned rmax = 0u;
    const unsigned seen_s = P.range ? P.range[TH_RANGE_S] : 0u, seen_p = P.range ? P.range[TH_RANGE_P] : 0u,
                   seen_n = P.range ? P.range[TH_RANGE_N] : 0u, seen_i = P.range ? P.range[TH_RANGE_INTER] : 0u,
                   seen_4 = P.range ? P.range[TH_RANGE_F4] : 0u;

    char* a256_lo = abuf + ROWS * STR256;
    char* fa_lo = abuf + ROWS * FL::SA;      // lo planes of the two f fillings
    char* fb_lo = abuf + ROWS * FL::SB;

    // ---- TEX (TH_ROWS_TEX): the pixel-feature operand is formed HERE from the texels of the map --------------------------
    // K5t (k_pixtex.hip) hands over, per tile, the list of DISTINCT corner texels of its 32 x V (sample, view) rows (79 on
    // average for the headline frame, at most 103 per pass) and per row four row numbers + four bilinear weights + the
    // blended colour.  Every listed texel is one 1 KiB row of the map (256 fp32 latents, L2 / MALL resident):
    //  1. a wave copies its share of the texel rows into ABUF (rows of 1040 bytes) -- plain 16-byte loads, all of a wave's
    //     requests in flight, then ds_write_b128 (LDS-DMA lands ~12 B/clk/CU whatever the source: 11 k cycles for 80 KB);
    //  2. a wave blends one operand row per step: lane l owns channels 4 l .. 4 l + 3, the four corner rows are read with one
    //     conflict-free ds_read_b128 each (64 lanes = the whole 1 KiB row), row numbers and weights are wave-uniform (scalar
    //     loads of the record), K5's packed-FMA term order -- the values are K5's, bit for bit -- and the split hi / lo halves
    //     stay in registers (4 dwords per row, 8 V rows per wave) across the barrier that retires the texel rows;
    //  3. the halves are written as the operand planes over the same buffer (512 contiguous bytes per instruction).
    // Nothing of f travels through HBM; the second use (RGB branch) repeats the filling from rows that are L2-warm by then.
    // `under` runs between the row requests and their arrival.
    // (tex_fetch: the tile header and this thread's piece of the row records, requested a phase ahead of the filling that
    // uses them -- an HBM round trip otherwise sits in front of the first texel-row request)
    struct TexPre { unsigned h0, h1; fm_u4 rq; };
    auto tex_fetch = [&]() __attribute__((always_inline)) {
        int tl = tile;
        asm volatile("" : "+s"(tl));        // (laundered per call: addresses derived from it are recomputed, not kept live)
        TexPre t;
        const unsigned* hb = P.tex_hdr + (long long)tl * 512;
        t.h0 = hb[lane];
        t.h1 = hb[64 + lane];
        t.rq = (fm_u4){0u, 0u, 0u, 0u};
        if (tid < 64 * V) t.rq = *reinterpret_cast<const fm_u4*>(P.tex_rec + (long long)tl * V * 32 * 8 + tid * 4);
        return t;
    };
#define TX_ADDR(id) (size_t)(((id) << 10) + loff)
    // `rgb` (a std::bool_constant): false = pixel branch: rows of fold0 (alpha_res_0 of the texels), p = relu(blend + bias) written
    // as the hi / lo operand planes of kv0; true = RGB branch: rows of fold12 ([Wa rgb_res_0 | rgb_res_1] of the texels), the
    // blended fp32 rows written to ABUF ([row][256], 1040-byte rows) for the accumulator-layout reads of the epilogue.
    auto fill_tex = [&](const TexPre& pre, auto rgb, auto&& under) __attribute__((always_inline)) {
        constexpr bool RGB = decltype(rgb)::value;
        constexpr int TSTR = 1040, TMAX = 103, NK = (TMAX + 3) / 4, NR = 8 * V;
        static_assert(TMAX * TSTR <= ABUF_BYTES, "a pass of texel rows must fit the operand buffer");
        int wv = __builtin_amdgcn_readfirstlane(wave), tl = tile;
        asm volatile("" : "+s"(wv), "+s"(tl));
        const unsigned* hb = P.tex_hdr + (long long)tl * 512;
        unsigned h0 = pre.h0, h1 = pre.h1;
        const int npass = __builtin_amdgcn_readfirstlane((int)(h0 >> 16));
        // the tile's row records ({w00, w01, w10, w11} {byte offsets of the four corner rows} per operand row, 32 V rows) go to
        // LDS (MISC: free here)
        char* recl = reinterpret_cast<char*>(misc);
        static_assert(32 * V * 32 <= (9 * 32 + 4 * 32 * 4) * 4, "row records must fit probs + part");
        // operand row wv + 4 k, channels 4 lane .. 4 lane + 3: packed hi / lo halves of relu(blend + bias) (pixel branch: fv[k][0..1]
        // hi, [2..3] lo) or the four blended fp32 values (RGB branch)
        unsigned fv[NR][4] = {};
        f32x2 bias_lo = {0.f, 0.f}, bias_hi = {0.f, 0.f};
        if constexpr (!RGB) {
            const float4 b4 = *reinterpret_cast<const float4*>(P.ar0.bias + 4 * lane);
            bias_lo = (f32x2){b4.x, b4.y};
            bias_hi = (f32x2){b4.z, b4.w};
        }
        for (int p = 0; p < npass; ++p) {
            if (p > 0) {
                FM_SYNCL();                                  // the previous pass's rows have been read
                h0 = hb[p * 128 + lane];
                h1 = hb[p * 128 + 64 + lane];
            }
            const int U = __builtin_amdgcn_readfirstlane((int)(h0 & 0xffffu));
            {
                // rows wv, wv + 4, ... of the list in three groups: the first 14 of a wave always (indices past the end clamped:
                // the last row again, same bytes to the same place), 6 more for lists longer than 56, the last 6 for lists longer
                // than 80 -- the header word of a request (8 + row) is then always in h0 for the first group and always in h1
                // for the others (no branch between the loads of a group; the average list of 79 rows has no clamped request).
                // Addresses are a scalar base + a 32-bit lane offset (map < 4 GiB: launcher).
                constexpr int NA = 14, NB = 20;
                const char* mbase = reinterpret_cast<const char*>(RGB ? P.tex_map2 : P.tex_map);
                const unsigned loff = (unsigned)lane * 16u;
                const int last = U - 1;
                fm_u4 ta[NA], tb[NB - NA], tc[NK - NB];
#pragma unroll
                for (int k = 0; k < NA; ++k) {
                    const int i = min(wv + 4 * k, last);
                    const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)h0, 8 + i);
                    ta[k] = *reinterpret_cast<const fm_u4*>(mbase + TX_ADDR(id));
                }
                const bool more = U > 4 * NA, most = U > 4 * NB;
                if (more) {
#pragma unroll
                    for (int k = NA; k < NB; ++k) {
                        const int i = min(wv + 4 * k, last);
                        const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)h1, i - 56);
                        tb[k - NA] = *reinterpret_cast<const fm_u4*>(mbase + TX_ADDR(id));
                    }
                }
                if (most) {
#pragma unroll
                    for (int k = NB; k < NK; ++k) {
                        const int i = min(wv + 4 * k, last);
                        const unsigned id = (unsigned)__builtin_amdgcn_readlane((int)h1, i - 56);
                        tc[k - NB] = *reinterpret_cast<const fm_u4*>(mbase + TX_ADDR(id));
                    }
                }
                if (p == 0) {
                    under();
                    if (tid < 64 * V) *reinterpret_cast<fm_u4*>(recl + tid * 16) = pre.rq;
                }
                int wv2 = wv;                               // (recomputed row numbers: no scalar lives across the loads)
                asm volatile("" : "+s"(wv2));
#pragma unroll
                for (int k = 0; k < NA; ++k) *reinterpret_cast<fm_u4*>(abuf + min(wv2 + 4 * k, last) * TSTR + lane * 16) = ta[k];
                if (more) {
#pragma unroll
                    for (int k = NA; k < NB; ++k) *reinterpret_cast<fm_u4*>(abuf + min(wv2 + 4 * k, last) * TSTR + lane * 16) = tb[k - NA];
                }
                if (most) {
#pragma unroll
                    for (int k = NB; k < NK; ++k) *reinterpret_cast<fm_u4*>(abuf + min(wv2 + 4 * k, last) * TSTR + lane * 16) = tc[k - NB];
                }
            }
            FM_SYNCL();                                      // the texel rows (and the records) are in place
            // One operand row per step and wave, software-pipelined by hand: the reads of row k + 1, then the corner offsets of
            // row k + 2 (its weights: the same address in every lane, a broadcast; its four corner rows) are issued before row k
            // is blended -- written row by row the compiler waits for every row's record, then for its corners: two exposed LDS
            // round trips per row, 8.7 k cycles.  With one wave per SIMD every instruction costs the wave an issue slot
            // (4 cycles): the loop is bound by its instruction count, so the producer stores byte offsets, not row numbers.
            // Operand row wv + 4 k is sample wv + 4 (k & 7): its pass is (k & 7) >> (sh - 2), known per k.
            const int sh2 = npass == 1 ? 3 : npass == 2 ? 2 : 1;
            const int cofs = lane * 16;
            struct RowIn { float4 a, b, c, d; fm_u4 q0; };
            auto issue = [&](int k, const fm_u4& o, RowIn& r) __attribute__((always_inline)) {
                r.q0 = *reinterpret_cast<const fm_u4*>(recl + (wv + 4 * k) * 32);     // the four weights: one broadcast read
                r.a = *reinterpret_cast<const float4*>(abuf + (o[0] + cofs));
                r.b = *reinterpret_cast<const float4*>(abuf + (o[1] + cofs));
                r.c = *reinterpret_cast<const float4*>(abuf + (o[2] + cofs));
                r.d = *reinterpret_cast<const float4*>(abuf + (o[3] + cofs));
            };
            auto offs = [&](int k) __attribute__((always_inline)) {                   // (a broadcast read as well)
                return *reinterpret_cast<const fm_u4*>(recl + (wv + 4 * k) * 32 + 16);
            };
            auto blend = [&](int k, const RowIn& r, auto sel) __attribute__((always_inline)) {
                // (plain copies first: __builtin_bit_cast applied to a vector COMPONENT reads component 0)
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
                    const float l0 = lo[0], l1 = lo[1], h0f = hi[0], h1f = hi[1];      // (copies: bit_cast of a vector component reads [0])
                    n0 = __builtin_bit_cast(unsigned, l0); n1 = __builtin_bit_cast(unsigned, l1);
                    n2 = __builtin_bit_cast(unsigned, h0f); n3 = __builtin_bit_cast(unsigned, h1f);
                } else {                                     // p = relu(alpha_res_0 f): + bias, relu, hi / lo split
                    lo = __builtin_elementwise_max(lo + bias_lo, (f32x2){0.f, 0.f});
                    hi = __builtin_elementwise_max(hi + bias_hi, (f32x2){0.f, 0.f});
                    split_pair(lo[0], lo[1], n0, n2);
                    split_pair(hi[0], hi[1], n1, n3);
                }
                if constexpr (decltype(sel)::value) {        // multi-pass tile: a row is kept in its own pass only
                    const bool mine = ((k & 7) >> sh2) == p;
                    fv[k][0] = mine ? n0 : fv[k][0]; fv[k][1] = mine ? n1 : fv[k][1];
                    fv[k][2] = mine ? n2 : fv[k][2]; fv[k][3] = mine ? n3 : fv[k][3];
                } else {
                    fv[k][0] = n0; fv[k][1] = n1; fv[k][2] = n2; fv[k][3] = n3;
                }
            };
            auto rows_loop = [&](auto sel) __attribute__((always_inline)) {
                RowIn in[2];                                 // the reads of row k + 1 in flight under the blend of row k
                fm_u4 of[2];                                 // (a second row of look-ahead changes nothing: issue-bound)
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
                *reinterpret_cast<fm_u4*>(abuf + (wv + 4 * k) * TSTR + lane * 16) = (fm_u4){fv[k][0], fv[k][1], fv[k][2], fv[k][3]};
        } else {
            const unsigned seen_p = P.range ? P.range[TH_RANGE_P] : 0u;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                range_acc<true>(rmax, fv[k][0]);
                range_acc<true>(rmax, fv[k][1]);
                *reinterpret_cast<uint2*>(abuf + (wv + 4 * k) * STR256 + lane * 8) = make_uint2(fv[k][0], fv[k][1]);
                *reinterpret_cast<uint2*>(a256_lo + (wv + 4 * k) * STR256 + lane * 8) = make_uint2(fv[k][2], fv[k][3]);
            }
            range_commit(P.range, TH_RANGE_P, seen_p, rmax);
        }
    };

    TexPre tex_pre{}, tex_pre2{};
    if constexpr (TEX) tex_pre = tex_fetch();

    // ================= token branch: s = relu(fc_0 h); ks|vs = kv1(s) =================
    // fc_0 is linear and h = sum_k w_k [token_v[k] | PE_k]: the token part of fc_0(h) is sum_k w_k (W_tok token_v[k]),
    // i.e. the SAME 7-neighbour blend applied to the per-frame table T' = tokens W_tok^T (one small GEMM per frame,
    // th_api.hip).  K4 therefore hands over `stok` = that blend (fp32, [P][V][256]) and the blended 63-wide
    // positional encoding `pe` (one split-f16 row per SAMPLE: it is the same for every view); what is left of
    // fc_0 here is W_pe pe: 4 k-blocks on 32 rows instead of 16 k-blocks on 32*V rows.
    // The 96 KB of stok of a tile are one contiguous block of global memory ([sample][view][256] fp32): every
    // (sample, view) row is copied by ONE LDS-DMA load (1 KiB, fully coalesced) into ABUF rows of STOK_STR bytes
    // (an odd number of 16-byte slots) and read back in the accumulator layout with conflict-free ds_read_b128.
    // (Fetching that layout straight from global memory took 96 loads per lane of 32 B segments from 32 different
    // rows each: the texture path, not the memory latency, bounded the phase at 9 k cycles.)
    // The view-direction rows of the tile (used by the RGB branch, 27 of 32 columns) sit behind an index (the valid-sample
    // list): the indices are requested here, in FRONT of the token rows (vmcnt returns in order: behind them they would
    // arrive last), and the rows themselves in front of the pixel-feature staging -- both HBM round trips then run under
    // waits that exist anyway (the dependent pair used to sit in front of the first LDS-DMA load: 2 k cycles per tile).
    int vsel[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + 256 * q, row = i >> 5, c = i & 31;
        int x = pbase + row;
        if (P.vd_sel != nullptr && P.rgb_all != 2 && c < 27 && row < npts) x = P.vd_sel[pbase + row];
        vsel[q] = x;
    }
    float vdv[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_vd = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q, row = i >> 5, c = i & 31;
            float x = 0.f;
            if (P.rgb_all != 2 && c < 27 && row < npts) {
                const long long vr = P.vd_sel ? (long long)(vsel[q] / P.vd_div) : (long long)(pbase + row);
                x = P.vd[vr * 27 + c];
            }
            vdv[q] = x;
        }
    };
    f32x16 acc2[2][V];
    {
        constexpr int STOK_STR = 1040;
        static_assert(32 * V * STOK_STR <= ABUF_BYTES, "stok rows must fit the operand buffer");
        char* pe_hi = mbuf;
        char* pe_lo = mbuf + 32 * STR64;
        // pe: 32 rows x (64 hi | 64 lo halves) = 8 KiB: one 16-byte piece per thread and plane
        const int prow = tid >> 3, pc = tid & 7;
        const int psrc = min(prow, npts - 1);
        const uint4 pe_h = *reinterpret_cast<const uint4*>(P.pe + (long long)(pbase + psrc) * 128 + 8 * pc);
        const uint4 pe_l = *reinterpret_cast<const uint4*>(P.pe + (long long)(pbase + psrc) * 128 + 64 + 8 * pc);
        const int aoff = (lane & 31) * STR64 + (lane >> 5) * 16;
        if (P.tsplit != nullptr) {
            // ---- TH_ROWS_NBR: the blend of T' rows is formed HERE, on the matrix pipe -------------------------------
            // K4 hands over, per sample, the 7 nearest token centres and their softmax weights (64 bytes instead of
            // 3 KB of blended rows through HBM).  The 32 samples of a tile share most of their neighbours: the union U
            // of their centres (typically 15-25) defines "slots"; stok^T[ch][sample] = sum_slot T'^T[ch][slot] W[slot][sample]
            // is one more GEMM of the tile with K = U: its "weights" are the U x V rows of the per-frame table T' (split
            // fp16 hi | lo, 1 KiB per row, L2-resident: 60 KB per tile at the L2 rate instead of 96 KB at the per-CU HBM
            // rate), its activations the sparse weight matrix W (7 non-zeros per sample).  Passes of 32 slots.
            // K4 also forms the union per tile (it works on the same 32-sample groups): a 512-byte tile header -- U and the
            // centre of every slot -- and per sample the SLOTS of its 7 neighbours.  Every wave reads the header into
            // registers (two dwords per lane) and addresses its share of the row loads through v_readlane: no LDS
            // bookkeeping and no barrier in front of the LDS-DMA loads.
            char* wsp_hi = mbuf + 16384;                                     // W [sample][slot] halves, K = 32 per pass
            char* wsp_lo = wsp_hi + 32 * STRVD;
            const unsigned* hdr = reinterpret_cast<const unsigned*>(P.stok) + (long long)((P.P + 31) / 32 * 32) * 16 +
                                  (long long)tile * 128;
            const unsigned h0 = hdr[lane], h1 = hdr[64 + lane];
            const int ns = tid / 7, nk = tid - 7 * ns;
            int slot = -1;
            float nw = 0.f;
            if (tid < 224) {
                const unsigned* rec = reinterpret_cast<const unsigned*>(P.stok) + (long long)(pbase + min(ns, npts - 1)) * 16;
                slot = (int)rec[nk];
                nw = __builtin_bit_cast(float, rec[8 + nk]);
            }
            // (the L2-resident fc_0 weights / bias are requested now: they travel during the header's HBM round trip)
            const float inv_t = P.t_inv[0];
            uint4 wq[4][2][2];
            const uint4* wl = wslice(P.fc_0pe, wave, 2, 0) + lane;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) load_wfrag<2>(wl, kb, wq[kb]);
            const BiasT b0[2] = {load_bias(P.fc_0pe.bias, wave * 64, lane), load_bias(P.fc_0pe.bias, wave * 64 + 32, lane)};
            for (int i = tid; i < 2 * 32 * STRVD / 16; i += 256) reinterpret_cast<uint4*>(wsp_hi)[i] = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(pe_hi + prow * STR64 + 16 * pc) = pe_h;
            *reinterpret_cast<uint4*>(pe_lo + prow * STR64 + 16 * pc) = pe_l;
            const int U = __builtin_amdgcn_readfirstlane((int)h0);          // (hdr[0])
            // the view-direction rows (their indices arrived in front of the header): one more HBM round trip that runs
            // beside the T' rows instead of in front of the pixel-feature staging
            load_vd();
            FM_SB();
            auto slot_centre = [&](int u) {                                  // wave-uniform u
                const int d = (2 + u) >> 1;
                const unsigned src = d < 64 ? (unsigned)__builtin_amdgcn_readlane((int)h0, d) : (unsigned)__builtin_amdgcn_readlane((int)h1, d - 64);
                return (int)((src >> (16 * ((2 + u) & 1))) & 0xffffu);
            };
            zero_acc<2, V>(acc2);
            f32x16 a1[2][1];
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const int aoffw = (lane & 31) * STRVD + (lane >> 5) * 16;
            for (int u0 = 0; u0 < U; u0 += 32) {
                const int nU = min(32, U - u0), KBu = (nU + 15) >> 4;
                if (u0 > 0) {
                    FM_SYNCL();                                      // the previous pass's operands have been read
                    for (int i = tid; i < 2 * 32 * STRVD / 16; i += 256) reinterpret_cast<uint4*>(wsp_hi)[i] = make_uint4(0u, 0u, 0u, 0u);
                }
                // T' rows of the pass (only the nU real ones: the k-padding of the last block re-reads row nU - 1 below)
                // (slot-major: the centre of a slot is looked up once and serves its V rows; no division, scalar addresses --
                // as a flat loop over (view, slot) the address arithmetic bounded this loop, like stage_glds once did)
                for (int u = wv; u < nU; u += 4) {
                    const int cu = slot_centre(u0 + u);
                    const char* g = reinterpret_cast<const char*>(P.tsplit) + (long long)cu * 1024 + lane * 16;
#pragma unroll
                    for (int vw = 0; vw < V; ++vw)
                        __builtin_amdgcn_global_load_lds((fm_gptr)(g + (long long)vw * P.t_nc * 1024),
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
                    zero_acc<2, 1>(a1);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        h8 xh[1], xl[1];
                        load_xfrag<1, STR64, 0>(pe_hi, pe_lo, aoff, kb, xh, xl);
                        mfma_kblock<1, 2>(wq[kb], xh, xl, a1);
                    }
                }
                FM_SYNC();                                           // rows (LDS-DMA) and W are in place
                for (int kb = 0; kb < KBu; ++kb) {
                    h8 xh[1], xl[1];
                    load_xfrag<1, STRVD, 0>(wsp_hi, wsp_lo, aoffw, kb, xh, xl);
                    int roff[8];                                     // slot rows of this lane's 8 k values (padding clamped)
#pragma unroll
                    for (int j = 0; j < 8; ++j) roff[j] = min(kb * 16 + 8 * (lane >> 5) + j, nU - 1) * STOK_STR;
#pragma unroll
                    for (int r = 0; r < V; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const char* rb = abuf + r * 32 * STOK_STR + 2 * (wave * 64 + c * 32 + (lane & 31));
                            h8 ah, al;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                ah[j] = *reinterpret_cast<const _Float16*>(rb + roff[j]);
                                al[j] = *reinterpret_cast<const _Float16*>(rb + roff[j] + 512);
                            }
                            acc2[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[0], acc2[c][r], 0, 0, 0);
                            acc2[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[0], acc2[c][r], 0, 0, 0);
                            acc2[c][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[0], acc2[c][r], 0, 0, 0);
                        }
                }
            }
            FM_SYNCL();                                   // every wave is done reading the T' rows: ABUF may take s
            const f32x2 it2 = {inv_t, inv_t};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                finish_tile_b<1>(a1[c], b0[c], P.fc_0pe.inv_scale, false);
#pragma unroll
                for (int r = 0; r < V; ++r) {
                    f32x2 u[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        u[q] = __builtin_elementwise_fma((f32x2){acc2[c][r][2 * q], acc2[c][r][2 * q + 1]}, it2,
                                                         (f32x2){a1[c][0][2 * q], a1[c][0][2 * q + 1]});
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        acc2[c][r][2 * q] = fmaxf(u[q][0], 0.f);
                        acc2[c][r][2 * q + 1] = fmaxf(u[q][1], 0.f);
                    }
                    store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, a256_lo, lane, rmax);
                }
            }
        } else {
        FM_SB();
        {   // the HBM filling first: the (L2-resident) fc_0 weights and bias queue behind it, not in front of it
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const char* sg = reinterpret_cast<const char*>(P.stok) + lane * 16;
#pragma unroll 4
            for (int i = wv; i < 32 * V; i += 4) {                  // LDS row i = view * 32 + sample (wave-uniform)
                const int sp = min(i & 31, npts - 1), vw = i >> 5;
                const char* g = sg + ((long long)(pbase + sp) * V + vw) * 1024;
                __builtin_amdgcn_global_load_lds((fm_gptr)g, (fm_lptr)(abuf + i * STOK_STR), 16, 0, 0);
            }
        }
        FM_SB();
        uint4 wq[4][2][2];
        const uint4* wl = wslice(P.fc_0pe, wave, 2, 0) + lane;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) load_wfrag<2>(wl, kb, wq[kb]);
        const BiasT b0[2] = {load_bias(P.fc_0pe.bias, wave * 64, lane), load_bias(P.fc_0pe.bias, wave * 64 + 32, lane)};
        FM_SB();
        *reinterpret_cast<uint4*>(pe_hi + prow * STR64 + 16 * pc) = pe_h;
        *reinterpret_cast<uint4*>(pe_lo + prow * STR64 + 16 * pc) = pe_l;
        FM_SYNC();
        f32x16 a1[2][1];
        zero_acc<2, 1>(a1);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            h8 xh[1], xl[1];
            load_xfrag<1, STR64, 0>(pe_hi, pe_lo, aoff, kb, xh, xl);
            mfma_kblock<1, 2>(wq[kb], xh, xl, a1);
        }
        float4 st[2][V][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < V; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    st[c][r][g] = *reinterpret_cast<const float4*>(abuf + (r * 32 + myrow) * STOK_STR +
                                                                   4 * (wave * 64 + c * 32 + 8 * g + 4 * (lane >> 5)));
        FM_SYNCL();                                   // every wave has its stok values in registers: ABUF may take s
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            finish_tile_b<1>(a1[c], b0[c], P.fc_0pe.inv_scale, false);
#pragma unroll
            for (int r = 0; r < V; ++r) {
                // packed adds first, the relus after them: independent consecutive instructions
                f32x2 u[8];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u[2 * g] = (f32x2){a1[c][0][4 * g + 0], a1[c][0][4 * g + 1]} + (f32x2){st[c][r][g].x, st[c][r][g].y};
                    u[2 * g + 1] = (f32x2){a1[c][0][4 * g + 2], a1[c][0][4 * g + 3]} + (f32x2){st[c][r][g].z, st[c][r][g].w};
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc2[c][r][2 * q] = fmaxf(u[q][0], 0.f);
                    acc2[c][r][2 * q + 1] = fmaxf(u[q][1], 0.f);
                }
                store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, a256_lo, lane, rmax);
            }
        }
        }
        range_commit(P.range, TH_RANGE_S, seen_s, rmax);
    }
    // The first weight block of a phase does not depend on the LDS operand: it is requested in FRONT of the barrier
    // that publishes the operand (the L2 round trip runs under the barrier instead of behind it).
    uint4 wk3[FM_RING_D][3][2];
    FM_SB();
    ring_prefetch0<3, FM_RING_D>(wslice(P.kv1, wave, 3, 0), lane, wk3);
    FM_SYNCL();
    // kv layers: column tile 0 = key tile `wave` (cols wave*32..), tiles 1,2 = value cols 128 + wave*64 ..
    f32x16 vs[2][V];
    float* ksb = reinterpret_cast<float*>(mbuf);                    // [ROWS][KSTR] fp32 keys of the token branch
    {
        f32x16 acc3[3][V];
        gemm_phase_core<V, 3, STR256, 32 * STR256, FM_RING_D, true, 7, true>(abuf, a256_lo, wslice(P.kv1, wave, 3, 0), P.kv1.KB, lane,
                                                                            acc3, wk3);
        {
            const BiasT bk = load_bias(P.kv1.bias, wave * 32, lane), bv0 = load_bias(P.kv1.bias, 128 + wave * 64, lane),
                        bv1 = load_bias(P.kv1.bias, 128 + wave * 64 + 32, lane);
            FM_SB();
            finish_tile_b<V>(acc3[0], bk, P.kv1.inv_scale, false);
            finish_tile_b<V>(acc3[1], bv0, P.kv1.inv_scale, false);
            finish_tile_b<V>(acc3[2], bv1, P.kv1.inv_scale, false);
        }
#pragma unroll
        for (int r = 0; r < V; ++r) {
            store_tile_f(acc3[0][r], r * 32 + myrow, wave * 32, ksb, lane);
            vs[0][r] = acc3[1][r];
            vs[1][r] = acc3[2][r];
        }
    }
    FM_SYNCL();

    // ================= pixel branch: p = relu(alpha_res_0 f); kp|vp = kv0(p) =================
    if (P.tsplit == nullptr) load_vd();      // (TH_ROWS_NBR: requested in the token branch, behind the tile header)
    FM_SB();
    uint4 wk2[FM_RING_D2][2][2];
    if constexpr (TEX) {
        // alpha_res_0 was applied to the texels of the map once per frame (map_fold_kernel below): p is the blend of fold0 rows
        fill_tex(tex_pre, std::false_type{}, [] {});
    } else {
    stage_glds<V, FL::LD, FL::KA, FL::SA>(P.f, 0, pbase, npts, abuf, fa_lo, wave, lane);
    ring_prefetch0<2, FM_RING_D2>(wslice(P.ar0, wave, 2, 0), lane, wk2);
    FM_SYNC();
    gemm_phase_core<V, 2, FL::SA, 32 * FL::SA, FM_RING_D2, true, 3, true>(abuf, fa_lo, wslice(P.ar0, wave, 2, 0), FL::NA, lane, acc2, wk2);
    const BiasT bp[2] = {load_bias(P.ar0.bias, wave * 64, lane), load_bias(P.ar0.bias, wave * 64 + 32, lane)};
    FM_SYNCL();
    if constexpr (FL::NB > 0) {
        stage_glds<V, FL::LD, FL::KB2, FL::SB>(P.f, FL::KA, pbase, npts, abuf, fb_lo, wave, lane);
        FM_SYNC();
        gemm_phase<V, 2, FL::SB>(abuf, fb_lo, wslice(P.ar0, wave, 2, FL::NA), FL::NB, lane, acc2);
        FM_SYNCL();
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        finish_tile_b<V>(acc2[c], bp[c], P.ar0.inv_scale, true);
#pragma unroll
        for (int r = 0; r < V; ++r)
            store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, a256_lo, lane, rmax);
    }
    range_commit(P.range, TH_RANGE_P, seen_p, rmax);
    }
    FM_SB();
    ring_prefetch0<3, FM_RING_D>(wslice(P.kv0, wave, 3, 0), lane, wk3);
    FM_SYNCL();
    f32x16 vp[2][V];
    {
        f32x16 acc3[3][V];
        gemm_phase_core<V, 3, STR256, 32 * STR256, FM_RING_D, true, 7, true>(abuf, a256_lo, wslice(P.kv0, wave, 3, 0), P.kv0.KB, lane,
                                                                            acc3, wk3);
        {
            const BiasT bk = load_bias(P.kv0.bias, wave * 32, lane), bv0 = load_bias(P.kv0.bias, 128 + wave * 64, lane),
                        bv1 = load_bias(P.kv0.bias, 128 + wave * 64 + 32, lane);
            FM_SB();
            finish_tile_b<V>(acc3[0], bk, P.kv0.inv_scale, false);
            finish_tile_b<V>(acc3[1], bv0, P.kv0.inv_scale, false);
            finish_tile_b<V>(acc3[2], bv1, P.kv0.inv_scale, false);
        }
        FM_SYNCL();                                                  // every wave is done reading p from ABUF
        float* kpb = reinterpret_cast<float*>(abuf);                // [ROWS][KSTR]
#pragma unroll
        for (int r = 0; r < V; ++r) {
            store_tile_f(acc3[0][r], r * 32 + myrow, wave * 32, kpb, lane);
            vp[0][r] = acc3[1][r];
            vp[1][r] = acc3[2][r];
        }
    }
    FM_SYNCL();

    // ================= cross-view attention (cross_transformer.py:128-149) =================
    {
        const float* kpb = reinterpret_cast<const float*>(abuf);
        const BiasT bn[2] = {load_bias(P.fc_1.bias, wave * 64, lane), load_bias(P.fc_1.bias, wave * 64 + 32, lane)};
        // A[j][i] = kp_j . ks_i / sqrt(128).  Thread (p = tid >> 3, c8 = tid & 7) owns float4 columns c8, c8 + 8,
        // c8 + 16, c8 + 24 of sample p (8 lanes read 128 contiguous bytes of a key row): it loads the V pixel-branch
        // and the V token-branch keys once, forms all V*V partial products, and the 8 partials of a sample are
        // summed with three xor-shuffles.  All 256 threads carry the same load (the one-thread-per-dot form took
        // two rounds for 288 dots: 6.8 k cycles against 2 k).
        {
            const int p = tid >> 3, c8 = tid & 7;
            float acc[V * V];
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) acc[ji] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 kx[V], sx[V];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    kx[v] = *reinterpret_cast<const float4*>(kpb + (v * 32 + p) * KSTR + 4 * (c8 + 8 * q));
                    sx[v] = *reinterpret_cast<const float4*>(ksb + (v * 32 + p) * KSTR + 4 * (c8 + 8 * q));
                }
                // component outermost: the V*V accumulation chains advance together (independent consecutive FMAs)
#pragma unroll
                for (int comp = 0; comp < 4; ++comp)
#pragma unroll
                    for (int j = 0; j < V; ++j)
#pragma unroll
                        for (int i = 0; i < V; ++i) {
                            const float a = comp == 0 ? kx[j].x : comp == 1 ? kx[j].y : comp == 2 ? kx[j].z : kx[j].w;
                            const float b = comp == 0 ? sx[i].x : comp == 1 ? sx[i].y : comp == 2 ? sx[i].z : sx[i].w;
                            acc[j * V + i] = fmaf(a, b, acc[j * V + i]);
                        }
            }
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) {
                float s = acc[ji];
                // sum over the 8 lanes of a sample with DPP moves (VALU, no LDS round trip like ds_bpermute): quad
                // neighbours, quad halves, then the mirrored lane of the other quad of the 8-lane group
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));  // row_half_mirror
                acc[ji] = s / 11.313708498984761f;          // every lane of the 8-lane group holds the full sum
            }
            // softmax over j for each (sample, i), computed by every lane of the group (9 exponentials; as its own
            // phase on 32*V threads it cost a barrier and 1.9 k cycles), written by one
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float m = -3.0e38f;
#pragma unroll
                for (int j = 0; j < V; ++j) m = fmaxf(m, acc[j * V + i]);
                float e[V], se = 0.f;
#pragma unroll
                for (int j = 0; j < V; ++j) { e[j] = expf(acc[j * V + i] - m); se = se + e[j]; }
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (c8 == 0) probs[(j * V + i) * 32 + p] = e[j] / se;
            }
        }
        FM_SYNCL();
        float A[V][V];
#pragma unroll
        for (int j = 0; j < V; ++j)
#pragma unroll
            for (int i = 0; i < V; ++i) A[j][i] = probs[(j * V + i) * 32 + myrow];
        // vs / vp hold (F V1) s_i and (F V0) p_j  (value_embed folded into fc_1, see k_mlp_fused_host.hip):
        //   fc_1 pre-activation of view i = vs_i + sum_j vp_j A[j][i] + folded bias ; relu ; -> operand of fc_2.
        // The key buffer in ABUF was last read before the barrier above, so each tile is stored as
        // soon as it is formed.
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // channel pairs outermost: every vp value leaves its accumulator register once and serves the V outputs
            // (the accumulators live in AGPRs: each use from VALU costs a v_accvgpr_read)
            f32x16 n[V];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = bn[c].g[g];
                const f32x2 bb[2] = {{b4.x, b4.y}, {b4.z, b4.w}};
                // packed fp32 FMAs (two channels per instruction); the 2 x V accumulation chains of a group advance
                // together, j outermost, so that consecutive instructions are independent (written chain by chain the
                // compiler emitted one serial dependency chain through a single temporary: 5 k cycles for this block)
                f32x2 v2[2][V], t[2][V];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 4 * g + 2 * h;
#pragma unroll
                    for (int j = 0; j < V; ++j) v2[h][j] = (f32x2){vp[c][j][e], vp[c][j][e + 1]};
#pragma unroll
                    for (int i = 0; i < V; ++i) t[h][i] = (f32x2){vs[c][i][e], vs[c][i][e + 1]} + bb[h];
                }
#pragma unroll
                for (int j = 0; j < V; ++j)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < V; ++i) {
                            const f32x2 a2 = {A[j][i], A[j][i]};
                            t[h][i] = __builtin_elementwise_fma(v2[h][j], a2, t[h][i]);
                        }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        n[i][4 * g + 2 * h] = fmaxf(t[h][i][0], 0.f);
                        n[i][4 * g + 2 * h + 1] = fmaxf(t[h][i][1], 0.f);
                    }
            }
#pragma unroll
            for (int i = 0; i < V; ++i)
                store_tile_h<STR256>(n[i], i * 32 + myrow, wave * 64 + c * 32, abuf, a256_lo, lane, rmax);
        }
        range_commit(P.range, TH_RANGE_N, seen_n, rmax);
    }

    // ================= fc_2 (fc_1 is folded into the value projections) =================
    FM_SB();
    ring_prefetch0<2, FM_RING_D2>(wslice(P.fc_2, wave, 2, 0), lane, wk2);
    FM_SYNCL();
    gemm_phase_core<V, 2, STR256, 32 * STR256, FM_RING_D2, true, 3, true>(abuf, a256_lo, wslice(P.fc_2, wave, 2, 0), P.fc_2.KB, lane, acc2,
                                                                         wk2);
    const BiasT bi[2] = {load_bias(P.fc_2.bias, wave * 64, lane), load_bias(P.fc_2.bias, wave * 64 + 32, lane)};
    FM_SYNCL();
    // inter = relu(.) -> ABUF (operand of feature_fc); its view mean -> MBUF (operand of fc_3)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        finish_tile_b<V>(acc2[c], bi[c], P.fc_2.inv_scale, true);
        // view mean: packed adds, the 8 chains advance together (same order of operations as the scalar form)
        f32x16 m;
        {
            f32x2 m2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) m2[q] = (f32x2){acc2[c][0][2 * q], acc2[c][0][2 * q + 1]};
#pragma unroll
            for (int r = 1; r < V; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) m2[q] = m2[q] + (f32x2){acc2[c][r][2 * q], acc2[c][r][2 * q + 1]};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                m2[q] = m2[q] * (f32x2){inv_v, inv_v};
                m[2 * q] = m2[q][0];
                m[2 * q + 1] = m2[q][1];
            }
        }
        store_tile_h<STR256>(m, myrow, wave * 64 + c * 32, mbuf, mbuf + 32 * STR256, lane, rmax);
#pragma unroll
        for (int r = 0; r < V; ++r)
            store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, a256_lo, lane, rmax);
    }
    range_commit(P.range, TH_RANGE_INTER, seen_i, rmax);
    FM_SYNCL();

    // ================= sigma head: relu(fc_3 m) . alpha_w + b =================
    // (the view-direction values vdv were requested at the top of the tile; they are parked in MBUF once every wave is
    // done reading the fc_3 operand)
    char* vd_hi = mbuf + MBUF_VD_OFF;
    char* vd_lo = vd_hi + 32 * STRVD;
    // acc2[0] collects the three K ranges of the folded view_fc (this wave's 32 of its 128 outputs), acc2[1] is
    // rgb_res_1: the pass over f multiplies the stacked [Wa R0 ; R1] image as two column tiles.
    if constexpr (TEX) {
        if (P.rgb_all != 2) tex_pre2 = tex_fetch();      // (for the RGB branch's filling: the round trip runs under fc_3)
    }
    f32x16 (&vf)[1][V] = *reinterpret_cast<f32x16 (*)[1][V]>(&acc2[0]);
    {
        f32x16 a1[2][1];
        // head rows / biases of this lane's channels, requested ahead of the GEMM
        float4 aw[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                aw[c][g] = *reinterpret_cast<const float4*>(P.alpha_w + wave * 64 + c * 32 + 8 * g + 4 * (lane >> 5));
        const BiasT b3[2] = {load_bias(P.fc_3.bias, wave * 64, lane), load_bias(P.fc_3.bias, wave * 64 + 32, lane)};
        // the view_fc product on inter does not depend on sigma: it shares the loop with fc_3 (see gemm_dual_fc3_vfa)
        // unless no sample can need colour (sigma-only consumers)
        if (P.rgb_all != 2)
            gemm_dual_fc3_vfa<V>(mbuf, mbuf + 32 * STR256, abuf, a256_lo, wslice(P.fc_3, wave, 2, 0), wslice(P.vfA, wave, 1, 0),
                                 lane, a1, vf);
        else
            gemm_phase_z<1, 2, STR256, 32 * STR256, 6>(mbuf, mbuf + 32 * STR256, wslice(P.fc_3, wave, 2, 0), P.fc_3.KB, lane, a1);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            finish_tile_b<1>(a1[c], b3[c], P.fc_3.inv_scale, true);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                s = fmaf(a1[c][0][4 * g + 0], aw[c][g].x, s);
                s = fmaf(a1[c][0][4 * g + 1], aw[c][g].y, s);
                s = fmaf(a1[c][0][4 * g + 2], aw[c][g].z, s);
                s = fmaf(a1[c][0][4 * g + 3], aw[c][g].w, s);
            }
        }
        s += __shfl_xor(s, 32);
        if (lane < 32) part[(wave * 32 + lane) * 4] = s;
        if (tid == 0) *flag = 0;
        FM_SYNCL();                                  // every wave is done reading the means (MBUF) and inter (ABUF)
        // rgb_all: 0 progressive (sigma > 0 only, :296-305), 1 every sample (MLP_forward_ori), 2 none (sigma grid)
        if (tid < 32) {
            const float sg = part[tid * 4] + part[(32 + tid) * 4] + part[(64 + tid) * 4] + part[(96 + tid) * 4] + P.alpha_b[0];
            sig[tid] = sg;
            if (tid < npts && P.rgb_all != 2 && (P.rgb_all == 1 || sg > 0.f)) *flag = 1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int i = tid + 256 * q, row = i >> 5, c = i & 31;
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
        // view_fc is folded over feature_fc / rgb_res_0 (k_mlp_fused_host.hip):
        //   t = relu((Wa F) inter + Wd viewdir + (Wa R0) f + b') ; u = t + rgb_res_1(f) ; mean over views ; fc_4 ; rgb_fc
        // ABUF has been free since the barrier behind the fc_3 / view_fc loop: f is requested first, the two
        // view-direction k-blocks (operand in MBUF) multiply while it arrives.
        // (vmcnt returns in order: the view-direction weights are requested BEFORE the staging loads, or their GEMM
        // would wait for the whole filling)
        uint4 wvd[FM_RING_D2][1][2];
        ring_prefetch<1, FM_RING_D2>(wslice(P.vfD, wave, 1, 0), 2, lane, wvd);
        FM_SB();
        if constexpr (TEX) {
            // (Wa rgb_res_0) f and rgb_res_1 f are blends of fold12 rows ([128 | 128] channels per texel): fp32 rows in ABUF
            fill_tex(tex_pre2, std::true_type{}, [&]() __attribute__((always_inline)) {
                gemm_phase_core<V, 1, STRVD, 0, FM_RING_D2, true, 0, true, false>(vd_hi, vd_lo, wslice(P.vfD, wave, 1, 0), 2, lane, vf, wvd);   // KB < D
            });
            FM_SYNCL();
        } else {
        stage_glds<V, FL::LD, FL::KA, FL::SA>(P.f, 0, pbase, npts, abuf, fa_lo, wave, lane);
        ring_prefetch0<2, FM_RING_D2>(wslice(P.rst, wave, 2, 0), lane, wk2);
        FM_SB();
        gemm_phase_core<V, 1, STRVD, 0, FM_RING_D2, true, 0, true, false>(vd_hi, vd_lo, wslice(P.vfD, wave, 1, 0), 2, lane, vf, wvd);   // KB < D
#pragma unroll
        for (int r = 0; r < V; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[1][r][e] = 0.f;
        FM_SYNC();
        }
        if constexpr (!TEX) {
        gemm_phase_core<V, 2, FL::SA, 32 * FL::SA, FM_RING_D2, true, 0, true>(abuf, fa_lo, wslice(P.rst, wave, 2, 0), FL::NA, lane, acc2, wk2);
        if constexpr (FL::NB > 0) {
            FM_SYNCL();
            stage_glds<V, FL::LD, FL::KB2, FL::SB>(P.f, FL::KA, pbase, npts, abuf, fb_lo, wave, lane);
            FM_SYNC();
            gemm_phase<V, 2, FL::SB>(abuf, fb_lo, wslice(P.rst, wave, 2, FL::NA), FL::NB, lane, acc2);
        }
        }
        // fc_4 weights (this wave's 8 k-blocks) and the rgb_fc rows of its channels: requested before the epilogue
        uint4 w4[8][1][2];
        {
            const uint4* wl4 = wslice(P.fc_4, wave, 1, 0) + lane;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) load_wfrag<1>(wl4, kb, w4[kb]);
        }
        float4 rw[3][4];
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                rw[o][g] = *reinterpret_cast<const float4*>(P.rgb_w + o * 128 + wave * 32 + 8 * g + 4 * (lane >> 5));
        const BiasT b4 = load_bias(P.fc_4.bias, wave * 32, lane);
        {
            const BiasT bt = load_bias(P.rst.bias, wave * 32, lane), br = load_bias(P.rst.bias, 128 + wave * 32, lane);
            FM_SB();
            if constexpr (TEX) {
                // t = relu(vf 2^-s + b' + blend(fold12[:, :128])), tile 1 = blend(fold12[:, 128:]) + b_R1: the blended rows are read
                // back in the accumulator layout (row = view * 32 + sample, this wave's 32 channels, 4 per group)
                finish_tile_b<V>(acc2[0], bt, P.rst.inv_scale, false);
                const char* mb = abuf + myrow * 1040 + 4 * (wave * 32 + 4 * (lane >> 5));
#pragma unroll
                for (int r = 0; r < V; ++r)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 m1 = *reinterpret_cast<const float4*>(mb + r * 32 * 1040 + 32 * g);
                        const float4 m2 = *reinterpret_cast<const float4*>(mb + r * 32 * 1040 + 512 + 32 * g);
                        const float4 b2 = br.g[g];
                        acc2[0][r][4 * g] = fmaxf(acc2[0][r][4 * g] + m1.x, 0.f);
                        acc2[0][r][4 * g + 1] = fmaxf(acc2[0][r][4 * g + 1] + m1.y, 0.f);
                        acc2[0][r][4 * g + 2] = fmaxf(acc2[0][r][4 * g + 2] + m1.z, 0.f);
                        acc2[0][r][4 * g + 3] = fmaxf(acc2[0][r][4 * g + 3] + m1.w, 0.f);
                        acc2[1][r][4 * g] = m2.x + b2.x;
                        acc2[1][r][4 * g + 1] = m2.y + b2.y;
                        acc2[1][r][4 * g + 2] = m2.z + b2.z;
                        acc2[1][r][4 * g + 3] = m2.w + b2.w;
                    }
            } else {
            finish_tile_b<V>(acc2[0], bt, P.rst.inv_scale, true);
            finish_tile_b<V>(acc2[1], br, P.rst.inv_scale2, false);
            }
        }
        char* f4_hi = mbuf + MBUF_FC4_OFF;
        char* f4_lo = f4_hi + 32 * STR128;
        {
            f32x16 m;
            {
                f32x2 u2[V][8];
#pragma unroll
                for (int r = 0; r < V; ++r)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        u2[r][q] = (f32x2){acc2[0][r][2 * q], acc2[0][r][2 * q + 1]} + (f32x2){acc2[1][r][2 * q], acc2[1][r][2 * q + 1]};
#pragma unroll
                for (int r = 1; r < V; ++r)
#pragma unroll
                    for (int q = 0; q < 8; ++q) u2[0][q] = u2[0][q] + u2[r][q];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    u2[0][q] = u2[0][q] * (f32x2){inv_v, inv_v};
                    m[2 * q] = u2[0][q][0];
                    m[2 * q + 1] = u2[0][q][1];
                }
            }
            store_tile_h<STR128, false>(m, myrow, wave * 32, f4_hi, f4_lo, lane, rmax);      // (signed: relu(.) + rgb_res_1)
            range_commit(P.range, TH_RANGE_F4, seen_4, rmax);
        }
        FM_SYNCL();
        f32x16 a4[1][1];
        zero_acc<1, 1>(a4);
        {
            const int aoff4 = (lane & 31) * STR128 + (lane >> 5) * 16;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                h8 xh4[1], xl4[1];
                load_xfrag<1, STR128, 0>(f4_hi, f4_lo, aoff4, kb, xh4, xl4);
                mfma_kblock<1, 1>(w4[kb], xh4, xl4, a4);
            }
        }
        finish_tile_b<1>(a4[0], b4, P.fc_4.inv_scale, true);
        float s3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                s3[o] = fmaf(a4[0][0][4 * g + 0], rw[o][g].x, s3[o]);
                s3[o] = fmaf(a4[0][0][4 * g + 1], rw[o][g].y, s3[o]);
                s3[o] = fmaf(a4[0][0][4 * g + 2], rw[o][g].z, s3[o]);
                s3[o] = fmaf(a4[0][0][4 * g + 3], rw[o][g].w, s3[o]);
            }
#pragma unroll
        for (int o = 0; o < 3; ++o) s3[o] += __shfl_xor(s3[o], 32);
        if (lane < 32) {
            part[(wave * 32 + lane) * 4 + 0] = s3[0];
            part[(wave * 32 + lane) * 4 + 1] = s3[1];
            part[(wave * 32 + lane) * 4 + 2] = s3[2];
        }
        FM_SYNCL();
        if (tid < 32) {
#pragma unroll
            for (int o = 0; o < 3; ++o)
                rgb_out[o] = part[tid * 4 + o] + part[(32 + tid) * 4 + o] + part[(64 + tid) * 4 + o] +
                             part[(96 + tid) * 4 + o] + P.rgb_b[o];
        }
    }
    if (tid < npts)
        *reinterpret_cast<float4*>(P.raw_c + (long long)(pbase + tid) * 4) =
            make_float4(rgb_out[0], rgb_out[1], rgb_out[2], sig[tid]);
    if (FM_DBG_SAMPLED) {        // dbg[62] / dbg[63]: shader cycles and 100 MHz ticks of the sampled tiles: the clock INSIDE the launch
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + 62), (unsigned long long)(clock64() - dbg_t0));
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg + 63), (unsigned long long)(wall_clock64() - dbg_w));
    }
}

// ---- the f-consuming layers applied to the MAP (TH_ROWS_TEX, late round 4) ------------------------------------------------------
// alpha_res_0, rgb_res_0 and rgb_res_1 (cross_transformer.py:316, :334, :346) are linear maps applied DIRECTLY to the bilinear
// samples of the pixel map (grid_sample, if_clight_renderer.py:255-265), and bilinear sampling is linear in the map's texels:
//   L(sum_c w_c texel_c) = sum_c w_c L(texel_c)          (the weights of a sample sum to 1: the bias is added after the blend)
// so the three layers are evaluated ONCE PER TEXEL of the (cropped) map -- ~0.26 M texel rows per frame instead of 6.3 M
// (sample, view) rows -- and the fused kernel blends texel rows of the two transformed maps where it used to blend the latents
// and multiply: its alpha_res_0 and stacked [Wa R0 ; R1] GEMMs (26 k of 141 k cycles per tile) are gone.  Same packed weight
// images, same fp16 hi/lo x 3 MFMA arithmetic, same power-of-two scales as in the fused kernel; outputs are the plain fp32
// values (accumulator x 2^-scale, no bias):
//   fold0 [V][H*W][256]  = alpha_res_0' [lat | rgb]                (K = 260 colour-folded form)
//   fold12[V][H*W][256]  = [ (Wa rgb_res_0')  (128) | rgb_res_1' (128) ] [lat | rgb]
// One workgroup = 96 consecutive texels of one image row inside the view's box (3 row tiles of 32: the fused kernel's operand
// shape, so gemm_phase_core and the weight slices are used as they are).
struct MapFoldParams {
    FusedLayer ar0, rst;           // the colour-folded (K = 272) images
    const float* lat;              // [V][H*W][256]
    const float* rgb;              // [V][H*W][4]
    const int32_t* box;            // device [V][4] x0 y0 x1 y1 inclusive, or nullptr: the whole map
    int V, H, W;
    float* out0;
    float* out12;
    unsigned int* range;           // TH_RANGE_F takes max |hi half| of the texels that are split
    const int32_t* list;           // LIST form (demand-driven map, k_demand.hip): texel indices, any order ...
    const unsigned* count;         // ... and their number (device)
};

// MF_RT row tiles of 32 texels per workgroup.  Measured on the headline frame's boxes (342 k texels, tools/fold_time.py): 3 (96
// texels, one workgroup per CU) 0.55 ms; 2 (64 texels, two workgroups per CU, meant to overlap one's staging with the other's
// GEMMs) 0.65 ms -- the launch is bound by the weight stream, 557 KB per workgroup out of L2 whatever its row count (by
// parts, round 4: stores 0.20 ms, the two GEMMs 0.28 ms, staging + launch 0.17 ms, additive), so more rows per workgroup win.
#define MF_STORE(c) (c)
#define MF_KB 17
#define MF_RT 3
#define MF_TEX (32 * MF_RT)
// LIST: the workgroup's 96 texels are entries blockIdx.x * 96 ... of a list of texel indices (the texels this frame's samples
// read, k_demand.hip) instead of a run of an image row: every tile but the last is full, wherever the texels lie.
template <bool LIST>
__global__ __launch_bounds__(256, 1) void map_fold_kernel(MapFoldParams P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* hi_pl = lds;
    char* lo_pl = lds + MF_TEX * STR272;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int xs = 0, x1 = 0;
    long long trow = 0;
    if constexpr (LIST) {
        const int n = (int)*P.count;
        xs = (int)blockIdx.x * MF_TEX;           // (position in the list; "x" below is a list position)
        x1 = n - 1;
        if (xs > x1) return;
    } else {
    const int tpr = (P.W + MF_TEX - 1) / MF_TEX;
    const int v = blockIdx.x / (P.H * tpr), rem = blockIdx.x - v * (P.H * tpr);
    const int y = rem / tpr, xt = rem - y * tpr;
    int x0 = 0;
    x1 = P.W - 1;
    if (P.box != nullptr) {             // th_map_box: [V][4] boxes, then [V][H][2] row spans (empty rows: x1 < x0)
        if (y < P.box[4 * v + 1] || y > P.box[4 * v + 3]) return;
        const int32_t* sp = P.box + 4 * P.V + ((long long)v * P.H + y) * 2;
        x0 = sp[0];
        x1 = sp[1];
    }
    xs = x0 + xt * MF_TEX;
    if (xs > x1) return;
    trow = ((long long)v * P.H + y) * P.W;      // texel index of (v, y, 0)
    }
    // texel index of position x of this tile's run
    auto texel = [&](int x) __attribute__((always_inline)) -> long long {
        if constexpr (LIST) return (long long)P.list[x];
        else return trow + x;
    };
    // (LIST: a workgroup walks the list in strides of the grid -- the list's length is only known on the device)
    for (;;) {
    unsigned rmax = 0u;
    const unsigned seen_f = P.range ? P.range[TH_RANGE_F] : 0u;
    // ---- the MF_TEX texels as fp16 hi / lo planes [texel][272] (texels past the box's edge: the edge texel again, never stored)
    {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll 4
        for (int k = 0; k < MF_TEX / 4; ++k) {
            const int r = wv + 4 * k;
            const long long t = texel(min(xs + r, x1));
            const float4 q = *reinterpret_cast<const float4*>(P.lat + t * 256 + 4 * lane);
            uint2 h, l;
            split_pair(q.x, q.y, h.x, l.x);
            split_pair(q.z, q.w, h.y, l.y);
            range_acc<false>(rmax, h.x);
            range_acc<false>(rmax, h.y);
            *reinterpret_cast<uint2*>(hi_pl + r * STR272 + lane * 8) = h;
            *reinterpret_cast<uint2*>(lo_pl + r * STR272 + lane * 8) = l;
        }
        if (tid < MF_TEX) {             // channels 256..258 = r g b, 259..271 = 0
            const long long t = texel(min(xs + tid, x1));
            const float4 c = *reinterpret_cast<const float4*>(P.rgb + t * 4);
            uint4 th = make_uint4(0u, 0u, 0u, 0u), tl = make_uint4(0u, 0u, 0u, 0u);
            split_pair(c.x, c.y, th.x, tl.x);
            split_pair(c.z, 0.f, th.y, tl.y);
            range_acc<false>(rmax, th.x);
            range_acc<false>(rmax, th.y);
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(hi_pl + tid * STR272 + 512) = th;
            *reinterpret_cast<uint4*>(hi_pl + tid * STR272 + 528) = z4;
            *reinterpret_cast<uint4*>(lo_pl + tid * STR272 + 512) = tl;
            *reinterpret_cast<uint4*>(lo_pl + tid * STR272 + 528) = z4;
        }
        range_commit(P.range, TH_RANGE_F, seen_f, rmax);
    }
    __syncthreads();
    const int myrow = lane & 31;
    uint4 wk2[FM_RING_D2][2][2];
    f32x16 acc[2][MF_RT];
    // ---- alpha_res_0'
    gemm_phase_core<MF_RT, 2, STR272, 32 * STR272, FM_RING_D2, true, 3, false>(hi_pl, lo_pl, wslice(P.ar0, wave, 2, 0), MF_KB, lane, acc, wk2);
    {
        const float sc = P.ar0.inv_scale;
#pragma unroll
        for (int r = 0; r < MF_RT; ++r) {
            const int x = xs + r * 32 + myrow;
            if (MF_STORE(x <= x1)) {
                float* o = P.out0 + texel(x) * 256 + wave * 64 + 4 * (lane >> 5);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(o + c * 32 + 8 * g) =
                            make_float4(acc[c][r][4 * g] * sc, acc[c][r][4 * g + 1] * sc, acc[c][r][4 * g + 2] * sc, acc[c][r][4 * g + 3] * sc);
            }
        }
    }
    // ---- stacked [Wa rgb_res_0' ; rgb_res_1']: column tile 0 = this wave's 32 of the 128 view_fc outputs, tile 1 = its 32 of rgb_res_1
    gemm_phase_core<MF_RT, 2, STR272, 32 * STR272, FM_RING_D2, true, 3, false>(hi_pl, lo_pl, wslice(P.rst, wave, 2, 0), MF_KB, lane, acc, wk2);
    {
        const float s0 = P.rst.inv_scale, s1 = P.rst.inv_scale2;
#pragma unroll
        for (int r = 0; r < MF_RT; ++r) {
            const int x = xs + r * 32 + myrow;
            if (MF_STORE(x <= x1)) {
                float* o = P.out12 + texel(x) * 256 + wave * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<float4*>(o + 8 * g) =
                        make_float4(acc[0][r][4 * g] * s0, acc[0][r][4 * g + 1] * s0, acc[0][r][4 * g + 2] * s0, acc[0][r][4 * g + 3] * s0);
                    *reinterpret_cast<float4*>(o + 128 + 8 * g) =
                        make_float4(acc[1][r][4 * g] * s1, acc[1][r][4 * g + 1] * s1, acc[1][r][4 * g + 2] * s1, acc[1][r][4 * g + 3] * s1);
                }
            }
        }
    }
    if (!LIST) break;
    xs += (int)gridDim.x * MF_TEX;
    if (xs > x1) break;
    __syncthreads();            // (the operand planes are read by every wave until its last MFMA)
    }
}
