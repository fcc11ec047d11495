// Second-generation fused per-point MLP (included by k_mlp_fused_host.hip behind k_mlp_fused_kernel.h, whose GEMM /
// epilogue building blocks it uses).  Same arithmetic, same packed weights, same inputs and outputs as
// mlp_fused_kernel; what changes is WHEN the pixel-feature rows f move and how often:
//
//   * f is read ONCE.  The two f-consuming products of the RGB branch ([Wa R0' ; rgb_res_1'] f, 256 outputs) are formed
//     in the pixel branch, while f sits in LDS for alpha_res_0, and wait in 96 accumulator registers for the RGB tail
//     (the first-generation kernel staged the same 104 KB a second time: 4.9 k cycles per tile and 0.85 GB of HBM reads
//     per 512 Ki-sample launch).  Register room comes from splitting the stacked key|value layer of the pixel branch
//     into a key pass (one column tile) and a value pass (two) and from NOT parking keys in LDS:
//   * the keys never leave registers.  Each wave owns 32 of the 128 key channels of all rows: the cross-view dots
//     A[j][i] = kp_j . ks_i are partial sums over a lane's 16 channels, a lane-pair add and a 4-wave sum through a 4.6 KB
//     LDS table (the fp32 key buffers -- 2 x 50 KB of LDS stores and loads, and the whole of MBUF during the pixel branch
//     -- are gone), every lane then forms its sample's softmax itself.
//   * MBUF, now free during the token branch's key/value GEMM, receives the first 128 columns of f by LDS-DMA UNDER that
//     GEMM (requests issued from inline asm between ring steps: fm_dma16; vmcnt returns in order, so a request has
//     D - 1 ring steps to come back before it can hold up a weight fragment).  alpha_res_0 then starts on those columns
//     with pre-loaded weights while the remaining columns stream into ABUF: the staging burst that nothing could hide
//     (7.7 k cycles per tile) runs under ~30 k cycles of MFMA work.
//
// LDS map (bytes, V = 3):
//   ABUF  101 376  operand planes [row][K <= 256] hi | lo (T' rows of the token blend first: 99 840)
//   MBUF   52 224  pe / W of the token blend -> f[:, 0:128] planes (stride STR128) -> view means -> viewdir + fc_4 operand
//   MISC    7 200  cross-view dot partials [9][32][4 waves], sigma, cross-wave partial sums, flag
#pragma once
#include "k_mlp_fused_kernel.h"

#define F2_K1 128
#define F2_ABUF_BYTES (2 * 96 * STR256)
#define F2_MBUF_BYTES (2 * 96 * STR128)
#define F2_DOTP_FLOATS (9 * 32 * 4)
#define F2_MISC_FLOATS (F2_DOTP_FLOATS + 128 + 4 * 32 * 4 + 8)
#define FUSED2_LDS_BYTES (F2_ABUF_BYTES + F2_MBUF_BYTES + F2_MISC_FLOATS * 4)

#ifndef FM2_RING_D1
#define FM2_RING_D1 6      // weight ring of the token-branch key/value GEMM: its depth is the latency budget of the f prefetch
#endif

// columns of f behind the prefetched 128: compact rows 144 (9 k-blocks), full rows 256 (16)
template <int FM> struct FLay2;
template <> struct FLay2<0> { static constexpr int LD = 384, K2 = 256, S2 = STR256, NB2 = 16; };
template <> struct FLay2<1> { static constexpr int LD = 272, K2 = 144, S2 = 2 * 144 + 16, NB2 = 9; };

// Incremental form of stage_glds: one call = this wave's next 1 KiB slice of the hi plane and of the lo plane (two
// LDS-DMA requests issued as raw ISA).  (row, slot) of the lane's 16-byte slot are stepped, not divided.
template <int V, int KROW, int KC, int STR>
struct StageGen {
    static constexpr int SL = STR / 16, DS = (2 * KC) / 16, NS = 32 * V * SL, NCH = (NS + 63) / 64;
    static constexpr int DR = 256 / SL, DSL = 256 % SL;
    static constexpr int ITERS = (NCH + 3) / 4;            // calls that can still request something (per wave, upper bound)
    static_assert(STR % 16 == 0 && (2 * KC) % 16 == 0 && DS < SL && KC % 8 == 0 && KROW % 8 == 0, "bad plane geometry");
    const char* gbase;
    char *hi, *lo;
    int pbase, npts, lane, row, slot, c;
    __device__ __forceinline__ void init(const _Float16* src, int coff, int pbase_, int npts_, char* hi_, char* lo_, int wave,
                                         int lane_) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        gbase = reinterpret_cast<const char*>(src) + 4 * coff;
        hi = hi_; lo = lo_; pbase = pbase_; npts = npts_; lane = lane_;
        const int q0 = wv * 64 + lane_;
        row = q0 / SL;
        slot = q0 - row * SL;
        c = wv;
    }
    __device__ __forceinline__ void issue() {
        if (c < NCH) {                                      // (wave-uniform)
            if (c * 64 + lane < NS) {
                int p = row & 31;
                const int vw = row >> 5;
                p = p < npts ? p : npts - 1;
                const int col = slot < DS ? slot * 32 : 0;
                const char* g = gbase + (long long)((pbase + p) * V + vw) * (4 * KROW) + col;
                fm_dma16(g, hi + c * 1024);
                fm_dma16(g + 16, lo + c * 1024);
            }
            row += DR;
            slot += DSL;
            if (slot >= SL) { slot -= SL; row += 1; }
            c += 4;
        }
    }
};

// hooks of the two GEMMs that carry staging requests
template <class G> struct HookEvery2 {            // one slice pair behind every second ring step
    G& g;
    template <int J> __device__ __forceinline__ void step(int) {
#ifndef FM2_EXP_NO_PREFETCH       // timing experiment only (wrong results): no requests under the key/value GEMM
        if ((J & 1) == 0) g.issue();
#endif
    }
};
template <class G, int PER, int STEPS> struct HookFront {   // PER slice pairs behind each of the first STEPS steps
    G& g;
    template <int J> __device__ __forceinline__ void step(int) {
        if (J < STEPS) {
#pragma unroll
            for (int q = 0; q < PER; ++q) g.issue();
        }
    }
};

// acc (+)= W * A^T over KB k-blocks whose weight fragments are already in registers (requested before a staging burst:
// nothing of this phase may queue BEHIND the burst, vmcnt returns in order); activations ping-pong as in the ring form
template <int RT, int CT, int STR, int ROWSTEP, int KB, int ZMASK, class HK, int... Ks>
__device__ __forceinline__ void gemm_preloaded_steps(std::integer_sequence<int, Ks...>, const char* __restrict__ ahi,
                                                     const char* __restrict__ alo, const uint4 (&w)[KB][CT][2], int aoff,
                                                     h8 (&xh)[2][RT], h8 (&xl)[2][RT], f32x16 (&acc)[CT][RT], HK& hook) {
    (([&]() __attribute__((always_inline)) {
         constexpr int k = Ks;
         if (k + 1 < KB) load_xfrag<RT, STR, ROWSTEP>(ahi, alo, aoff, k + 1, xh[(k + 1) & 1], xl[(k + 1) & 1]);
         mfma_kblock<RT, CT, (k == 0 ? ZMASK : 0)>(w[k], xh[k & 1], xl[k & 1], acc);
         constexpr int NMEM = (k + 1 < KB) ? 2 * RT : 0, NMF = 3 * CT * RT;
#pragma unroll
         for (int q = 0; q < NMEM; ++q) {
             __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
             __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
         }
         __builtin_amdgcn_sched_group_barrier(0x008, NMF - NMEM, 0);
         FM_SB();
         hook.template step<k>(0);
     }()),
     ...);
}
template <int RT, int CT, int STR, int ROWSTEP, int KB, int ZMASK, class HK>
__device__ __forceinline__ void gemm_preloaded(const char* __restrict__ ahi, const char* __restrict__ alo,
                                               const uint4 (&w)[KB][CT][2], int lane, f32x16 (&acc)[CT][RT], HK& hook) {
    const int aoff = (lane & 31) * STR + (lane >> 5) * 16;
    h8 xh[2][RT], xl[2][RT];
    load_xfrag<RT, STR, ROWSTEP>(ahi, alo, aoff, 0, xh[0], xl[0]);
    FM_SB();
    gemm_preloaded_steps<RT, CT, STR, ROWSTEP, KB, ZMASK, HK>(std::make_integer_sequence<int, KB>{}, ahi, alo, w, aoff, xh, xl, acc, hook);
}

// LDS-only barrier as raw ISA (the staging requests of fm_dma16 stay in flight)
#define FM2_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int V, int FM>
__global__ __launch_bounds__(256, 1) void mlp_fused2_kernel(FusedParams P) {
    using FL = FLay2<FM>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* abuf = lds;
    char* mbuf = lds + F2_ABUF_BYTES;
    float* misc = reinterpret_cast<float*>(lds + F2_ABUF_BYTES + F2_MBUF_BYTES);
    float* dotp = misc;                          // [V*V][32][4 waves]
    float* sig = misc + F2_DOTP_FLOATS;          // [32] (+ padding)
    float* part = sig + 128;                     // [4 waves][32][4]
    int* flag = reinterpret_cast<int*>(part + 4 * 32 * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pbase = blockIdx.x * FM_PTS;
    const int npts = min(FM_PTS, P.P - pbase);
    constexpr int ROWS = 32 * V;
    const int myrow = lane & 31;
    int dbg_i = 1;
    long long dbg_t = 0;
    if (P.dbg != nullptr && tid == 0 && (blockIdx.x & 15) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long*>(P.dbg), 1ull);
        dbg_t = clock64();
    }
    constexpr float inv_v = 1.0f / (float)V;
    unsigned rmax = 0u;
    const unsigned seen_s = P.range ? P.range[TH_RANGE_S] : 0u, seen_p = P.range ? P.range[TH_RANGE_P] : 0u,
                   seen_n = P.range ? P.range[TH_RANGE_N] : 0u, seen_i = P.range ? P.range[TH_RANGE_INTER] : 0u,
                   seen_4 = P.range ? P.range[TH_RANGE_F4] : 0u;
    char* a256_lo = abuf + ROWS * STR256;
    char* f1_hi = mbuf;                           // f[:, 0:128] planes
    char* f1_lo = mbuf + ROWS * STR128;
    char* f2_lo = abuf + ROWS * FL::S2;           // f[:, 128:] planes: hi at abuf

    // ================= token branch: s = relu(fc_0 h) (mlp_fused_kernel's, unchanged) =================
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
    {
        f32x16 acc2[2][V];
        constexpr int STOK_STR = 1040;
        static_assert(32 * V * STOK_STR <= F2_ABUF_BYTES, "stok rows must fit the operand buffer");
        char* pe_hi = mbuf;
        char* pe_lo = mbuf + 32 * STR64;
        const int prow = tid >> 3, pc = tid & 7;
        const int psrc = min(prow, npts - 1);
        const uint4 pe_h = *reinterpret_cast<const uint4*>(P.pe + (long long)(pbase + psrc) * 128 + 8 * pc);
        const uint4 pe_l = *reinterpret_cast<const uint4*>(P.pe + (long long)(pbase + psrc) * 128 + 64 + 8 * pc);
        const int aoff = (lane & 31) * STR64 + (lane >> 5) * 16;
        if (P.tsplit != nullptr) {
            // ---- TH_ROWS_NBR: the blend of T' rows on the matrix pipe (see mlp_fused_kernel) ----
            char* wsp_hi = mbuf + 16384;
            char* wsp_lo = wsp_hi + 32 * STRVD;
            const unsigned* hdr = reinterpret_cast<const unsigned*>(P.stok) + (long long)((P.P + 31) / 32 * 32) * 16 +
                                  (long long)blockIdx.x * 128;
            const unsigned h0 = hdr[lane], h1 = hdr[64 + lane];
            const int ns = tid / 7, nk = tid - 7 * ns;
            int slot = -1;
            float nw = 0.f;
            if (tid < 224) {
                const unsigned* rec = reinterpret_cast<const unsigned*>(P.stok) + (long long)(pbase + min(ns, npts - 1)) * 16;
                slot = (int)rec[nk];
                nw = __builtin_bit_cast(float, rec[8 + nk]);
            }
            const float inv_t = P.t_inv[0];
            uint4 wq[4][2][2];
            const uint4* wl = wslice(P.fc_0pe, wave, 2, 0) + lane;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) load_wfrag<2>(wl, kb, wq[kb]);
            const BiasT b0[2] = {load_bias(P.fc_0pe.bias, wave * 64, lane), load_bias(P.fc_0pe.bias, wave * 64 + 32, lane)};
            for (int i = tid; i < 2 * 32 * STRVD / 16; i += 256) reinterpret_cast<uint4*>(wsp_hi)[i] = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(pe_hi + prow * STR64 + 16 * pc) = pe_h;
            *reinterpret_cast<uint4*>(pe_lo + prow * STR64 + 16 * pc) = pe_l;
            const int U = __builtin_amdgcn_readfirstlane((int)h0);
            load_vd();
            FM_SB();
            auto slot_centre = [&](int u) {
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
                    FM_SYNCL();
                    for (int i = tid; i < 2 * 32 * STRVD / 16; i += 256) reinterpret_cast<uint4*>(wsp_hi)[i] = make_uint4(0u, 0u, 0u, 0u);
                }
                for (int u = wv; u < nU; u += 4) {
                    const int cu = slot_centre(u0 + u);
                    const char* g = reinterpret_cast<const char*>(P.tsplit) + (long long)cu * 1024 + lane * 16;
#pragma unroll
                    for (int vw = 0; vw < V; ++vw)
                        __builtin_amdgcn_global_load_lds((fm_gptr)(g + (long long)vw * P.t_nc * 1024),
                                                         (fm_lptr)(abuf + (vw * 32 + u) * STOK_STR), 16, 0, 0);
                }
                FM_SYNCL();
                if (slot >= u0 && slot < u0 + 32) {
                    _Float16 hi, lo;
                    split_h(nw, hi, lo);
                    *reinterpret_cast<_Float16*>(wsp_hi + ns * STRVD + 2 * (slot - u0)) = hi;
                    *reinterpret_cast<_Float16*>(wsp_lo + ns * STRVD + 2 * (slot - u0)) = lo;
                }
                if (u0 == 0) {
                    zero_acc<2, 1>(a1);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        h8 xh[1], xl[1];
                        load_xfrag<1, STR64, 0>(pe_hi, pe_lo, aoff, kb, xh, xl);
                        mfma_kblock<1, 2>(wq[kb], xh, xl, a1);
                    }
                }
                FM_SYNC();
                for (int kb = 0; kb < KBu; ++kb) {
                    h8 xh[1], xl[1];
                    load_xfrag<1, STRVD, 0>(wsp_hi, wsp_lo, aoffw, kb, xh, xl);
                    int roff[8];
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
            FM_SYNCL();
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
            // ---- TH_ROWS_FOLDED: blended rows from K4 ----
            FM_SB();
            {
                const int wv = __builtin_amdgcn_readfirstlane(wave);
                const char* sg = reinterpret_cast<const char*>(P.stok) + lane * 16;
#pragma unroll 4
                for (int i = wv; i < 32 * V; i += 4) {
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
            load_vd();
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
            FM_SYNCL();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                finish_tile_b<1>(a1[c], b0[c], P.fc_0pe.inv_scale, false);
#pragma unroll
                for (int r = 0; r < V; ++r) {
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

    // ================= ks | vs = kv1(s), with f[:, 0:128] arriving in MBUF under the GEMM =================
    f32x16 vs[2][V];           // (F V1) s: raw accumulators until the combine needs them (finished in place below)
    f32x16 ks[V];              // this wave's 32 key channels of the token branch, finished, for all rows
    uint4 wp1[F2_K1 / 16][2][2];   // alpha_res_0 fragments of the prefetched columns (requested before the second staging burst)
    {
        uint4 wk1[FM2_RING_D1][3][2];
        FM_SB();
        ring_prefetch0<3, FM2_RING_D1>(wslice(P.kv1, wave, 3, 0), lane, wk1);
        FM_SYNCL();                        // s is published; pe / W of the token blend (MBUF) are dead
        StageGen<V, FL::LD, F2_K1, STR128> g1;
        g1.init(P.f, 0, pbase, npts, f1_hi, f1_lo, wave, lane);
        HookEvery2<decltype(g1)> hk1{g1};
        static_assert(decltype(g1)::ITERS <= 8, "the prefetch must fit the 16 ring steps of the key/value GEMM");
        f32x16 acc3[3][V];
        gemm_phase_core_h<V, 3, STR256, 32 * STR256, FM2_RING_D1, true, 7, true, true, FmNoStamp, decltype(hk1)>(
            abuf, a256_lo, wslice(P.kv1, wave, 3, 0), P.kv1.KB, lane, acc3, wk1, FmNoStamp(), hk1);
#ifdef FM_STAMPS
        FM_STAMP();
#endif
        // this wave's prefetch requests have landed (issued >= 2 ring steps ago) -- and nothing tracked is in flight when
        // the fragments of the first alpha_res_0 part are requested
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            const uint4* wl = wslice(P.ar0, wave, 2, 0) + lane;
#pragma unroll
            for (int kb = 0; kb < F2_K1 / 16; ++kb) load_wfrag<2>(wl, kb, wp1[kb]);
        }
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
            ks[r] = acc3[0][r];
            vs[0][r] = acc3[1][r];
            vs[1][r] = acc3[2][r];
        }
    }
    // every wave is done reading s (ABUF may take f[:, 128:]) and has seen its share of f[:, 0:128] land
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the pre-loaded fragments: requested a whole epilogue ago)
    FM_SYNC_(FM2_LDS_BARRIER());

    // ================= pixel branch: p = relu(alpha_res_0 f); rr = [Wa R0 ; rgb_res_1] f =================
    f32x16 rr[2][V];           // tile 0 continues as the folded view_fc sum in the RGB tail, tile 1 = rgb_res_1 f
    f32x16 acc2[2][V];
    {
        StageGen<V, FL::LD, FL::K2, FL::S2> g2;
        g2.init(P.f, F2_K1, pbase, npts, abuf, f2_lo, wave, lane);
        constexpr int PER = (decltype(g2)::ITERS + 3) / 4;
        HookFront<decltype(g2), PER, 4> hk2{g2};
        g2.issue();                        // (the first slices before the first MFMA: the burst is what bounds this part)
        gemm_preloaded<V, 2, STR128, 32 * STR128, F2_K1 / 16, 3>(f1_hi, f1_lo, wp1, lane, acc2, hk2);
#ifdef FM_STAMPS
        FM_STAMP();
#endif
    }
    uint4 wk2[FM_RING_D2][2][2];
    FM_SYNC_(fm_dma_wait_barrier());       // f[:, 128:] is in place for every wave
    gemm_phase_core<V, 2, FL::S2, 32 * FL::S2, FM_RING_D2, true, 0, false>(abuf, f2_lo, wslice(P.ar0, wave, 2, F2_K1 / 16), FL::NB2, lane,
                                                                          acc2, wk2);
#ifdef FM_STAMPS
    FM_STAMP();
#endif
    if (P.rgb_all != 2) {
        gemm_phase_core<V, 2, FL::S2, 32 * FL::S2, FM_RING_D2, true, 3, false>(abuf, f2_lo, wslice(P.rst, wave, 2, F2_K1 / 16), FL::NB2,
                                                                              lane, rr, wk2);
        gemm_phase_core<V, 2, STR128, 32 * STR128, FM_RING_D2, true, 0, false>(f1_hi, f1_lo, wslice(P.rst, wave, 2, 0), F2_K1 / 16, lane,
                                                                              rr, wk2);
    }
#ifdef FM_STAMPS
    FM_STAMP();
#endif
    {
        const BiasT bp[2] = {load_bias(P.ar0.bias, wave * 64, lane), load_bias(P.ar0.bias, wave * 64 + 32, lane)};
        FM_SYNCL();                        // every wave is done reading f
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            finish_tile_b<V>(acc2[c], bp[c], P.ar0.inv_scale, true);
#pragma unroll
            for (int r = 0; r < V; ++r)
                store_tile_h<STR256>(acc2[c][r], r * 32 + myrow, wave * 64 + c * 32, abuf, a256_lo, lane, rmax);
        }
        range_commit(P.range, TH_RANGE_P, seen_p, rmax);
    }

    // ================= kp = key0(p) -> cross-view dot partials; vp = (F V0) p =================
    f32x16 vp[2][V];
    {
        uint4 wkk[FM_RING_D2][1][2];
        FM_SB();
        ring_prefetch0<1, FM_RING_D2>(wslice(P.k0, wave, 1, 0), lane, wkk);
        FM_SYNCL();                        // p is published
        f32x16 kq[1][V];
        gemm_phase_core<V, 1, STR256, 32 * STR256, FM_RING_D2, true, 1, true>(abuf, a256_lo, wslice(P.k0, wave, 1, 0), P.k0.KB, lane, kq,
                                                                             wkk);
        ring_prefetch0<2, FM_RING_D2>(wslice(P.v0, wave, 2, 0), lane, wk2);
        {
            const BiasT bk = load_bias(P.kv0.bias, wave * 32, lane);
            FM_SB();
            finish_tile_b<V>(kq[0], bk, P.kv0.inv_scale, false);
        }
        // partial dots over this lane's 16 of the wave's 32 key channels: A[j][i] = kp_j . ks_i (j: pixel-branch view)
        float pd[V * V];
#pragma unroll
        for (int ji = 0; ji < V * V; ++ji) pd[ji] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int j = 0; j < V; ++j)
#pragma unroll
                for (int i = 0; i < V; ++i) pd[j * V + i] = fmaf(kq[0][j][e], ks[i][e], pd[j * V + i]);
#pragma unroll
        for (int ji = 0; ji < V * V; ++ji) pd[ji] += __shfl_xor(pd[ji], 32);
        if (lane < 32) {
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) dotp[(ji * 32 + lane) * 4 + wave] = pd[ji];
        }
        FM_SB();
        gemm_phase_core<V, 2, STR256, 32 * STR256, FM_RING_D2, true, 3, true>(abuf, a256_lo, wslice(P.v0, wave, 2, 0), P.v0.KB, lane, vp,
                                                                             wk2);
        {
            const BiasT bv0 = load_bias(P.kv0.bias, 128 + wave * 64, lane), bv1 = load_bias(P.kv0.bias, 128 + wave * 64 + 32, lane);
            FM_SB();
            finish_tile_b<V>(vp[0], bv0, P.kv0.inv_scale, false);
            finish_tile_b<V>(vp[1], bv1, P.kv0.inv_scale, false);
        }
    }

    // ================= cross-view attention (cross_transformer.py:128-149) =================
    {
        const BiasT bn[2] = {load_bias(P.fc_1.bias, wave * 64, lane), load_bias(P.fc_1.bias, wave * 64 + 32, lane)};
        FM_SYNCL();                        // dot partials published; every wave is done reading p
        float A[V][V];
        {
            float d[V * V];
#pragma unroll
            for (int ji = 0; ji < V * V; ++ji) {
                const float4 q = *reinterpret_cast<const float4*>(dotp + (ji * 32 + myrow) * 4);
                d[ji] = ((q.x + q.y) + (q.z + q.w)) / 11.313708498984761f;
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float m = -3.0e38f;
#pragma unroll
                for (int j = 0; j < V; ++j) m = fmaxf(m, d[j * V + i]);
                float e[V], se = 0.f;
#pragma unroll
                for (int j = 0; j < V; ++j) { e[j] = expf(d[j * V + i] - m); se = se + e[j]; }
#pragma unroll
                for (int j = 0; j < V; ++j) A[j][i] = e[j] / se;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x16 n[V];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = bn[c].g[g];
                const f32x2 bb[2] = {{b4.x, b4.y}, {b4.z, b4.w}};
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
#ifdef FM_STAMPS
    FM_STAMP();
#endif
    {
        const BiasT bi[2] = {load_bias(P.fc_2.bias, wave * 64, lane), load_bias(P.fc_2.bias, wave * 64 + 32, lane)};
        FM_SYNCL();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            finish_tile_b<V>(acc2[c], bi[c], P.fc_2.inv_scale, true);
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
    }
    FM_SYNCL();

    // ================= sigma head: relu(fc_3 m) . alpha_w + b; folded view_fc on inter accumulates onto rr[0] =================
    char* vd_hi = mbuf + MBUF_VD_OFF;
    char* vd_lo = vd_hi + 32 * STRVD;
    f32x16 (&vf)[1][V] = *reinterpret_cast<f32x16 (*)[1][V]>(&rr[0]);
    {
        f32x16 a1[2][1];
        float4 aw[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                aw[c][g] = *reinterpret_cast<const float4*>(P.alpha_w + wave * 64 + c * 32 + 8 * g + 4 * (lane >> 5));
        const BiasT b3[2] = {load_bias(P.fc_3.bias, wave * 64, lane), load_bias(P.fc_3.bias, wave * 64 + 32, lane)};
#ifdef FM_STAMPS
        FM_STAMP();
#endif
        if (P.rgb_all != 2)
            gemm_dual_fc3_vfa<V, true>(mbuf, mbuf + 32 * STR256, abuf, a256_lo, wslice(P.fc_3, wave, 2, 0), wslice(P.vfA, wave, 1, 0),
                                       lane, a1, vf);
        else
            gemm_phase_z<1, 2, STR256, 32 * STR256, 6>(mbuf, mbuf + 32 * STR256, wslice(P.fc_3, wave, 2, 0), P.fc_3.KB, lane, a1);
#ifdef FM_STAMPS
        FM_STAMP();
#endif
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
        FM_SYNCL();
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
        // ================= RGB tail (cross_transformer.py:330-353): no pass over f any more =================
        //   t = relu((Wa F) inter + Wd viewdir + (Wa R0) f + b') ; u = t + rgb_res_1(f) ; mean over views ; fc_4 ; rgb_fc
        uint4 wvd[FM_RING_D2][1][2];
        ring_prefetch<1, FM_RING_D2>(wslice(P.vfD, wave, 1, 0), 2, lane, wvd);
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
        const BiasT bt = load_bias(P.rst.bias, wave * 32, lane), br = load_bias(P.rst.bias, 128 + wave * 32, lane);
        FM_SB();
        gemm_phase_core<V, 1, STRVD, 0, FM_RING_D2, true, 0, true, false>(vd_hi, vd_lo, wslice(P.vfD, wave, 1, 0), 2, lane, vf, wvd);   // KB < D
        finish_tile_b<V>(rr[0], bt, P.rst.inv_scale, true);
        finish_tile_b<V>(rr[1], br, P.rst.inv_scale2, false);
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
                        u2[r][q] = (f32x2){rr[0][r][2 * q], rr[0][r][2 * q + 1]} + (f32x2){rr[1][r][2 * q], rr[1][r][2 * q + 1]};
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
            store_tile_h<STR128, false>(m, myrow, wave * 32, f4_hi, f4_lo, lane, rmax);
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
}
