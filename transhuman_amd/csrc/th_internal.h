// Internal declarations shared by the HIP translation units of
// libtranshuman_hip.so (gfx950 only).  Public ABI: include/transhuman_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <time.h>
#include <string>

#include "transhuman_hip.h"

#define TH_WAVE 64

// ---- error plumbing ---------------------------------------------------------
void th_set_error(const std::string& msg);
#define TH_FAIL(msg)                                                                \
    do {                                                                            \
        th_set_error(std::string(__func__) + ": " + (msg));                         \
        return -1;                                                                  \
    } while (0)
#define TH_REQUIRE(cond, msg)                                                       \
    do {                                                                            \
        if (!(cond)) TH_FAIL(msg);                                                  \
    } while (0)
#define TH_HIP(call)                                                                \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) {                                                    \
            th_set_error(std::string(__func__) + ": " #call " -> " + hipGetErrorString(e__)); \
            return -2;                                                              \
        }                                                                           \
    } while (0)
#define TH_LAUNCH_CHECK() TH_HIP(hipGetLastError())
#define TH_TRY(call)                                                                \
    do {                                                                            \
        int r__ = (call);                                                           \
        if (r__ != 0) return r__;                                                   \
    } while (0)

static inline size_t th_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int th_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bump allocator over the caller-supplied workspace.
struct ThArena {
    char* base;
    size_t cap, off;
    ThArena(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = th_align(count * sizeof(T));
        if (off + bytes > cap) return nullptr;
        T* r = (T*)(base + off);
        off += bytes;
        return r;
    }
};

// ---- packed dense layer (MFMA 16x16x4 f32 B-operand image) --------------------
// w: [NB][KB][64 lanes][4] with lane = (kq<<4)|j, value = W[nb*16+j][kb*16+4*kq+e]
// (zero padded); b: [NB*16] zero padded.
struct ThPacked {
    float* w = nullptr;
    float* b = nullptr;
    int N = 0, K = 0, NB = 0, KB = 0;
    // optional second image for th_gemm_h3 (fp16 hi/lo split x3 on v_mfma_f32_16x16x32_f16, the fused MLP's fp32-class
    // arithmetic): [NB][KB32][hi | lo][64 lanes] 16-byte B-operand fragments of the weights scaled by a power of two
    // (max |w| -> [2^12, 2^13): lo halves stay normal numbers); scale16[0] = its inverse (device scalar)
    const uint4* w16 = nullptr;
    const float* scale16 = nullptr;
    int KB32 = 0;
    static size_t bytes(int out_f, int in_f) {
        size_t nb = (out_f + 15) / 16, kb = (in_f + 15) / 16;
        return th_align(nb * kb * 256 * sizeof(float)) + th_align(nb * 16 * sizeof(float));
    }
    static size_t bytes_h3(int out_f, int in_f) {
        size_t nb = (out_f + 15) / 16, kb32 = (in_f + 31) / 32;
        return th_align(nb * kb32 * 2 * 64 * 16) + th_align(64);
    }
};

enum { TH_ACT_NONE = 0, TH_ACT_RELU = 1, TH_ACT_GELU = 2, TH_GEMM_ACCUM = 16 };

int th_pack_linear(const th_linear& lin, void* storage, ThPacked* out, hipStream_t s);
// adds the fp16-split image (storage_h3: ThPacked::bytes_h3 bytes) to an already packed layer
int th_pack_linear_h3(const th_linear& lin, void* storage_h3, ThPacked* out, hipStream_t s);
// C = act([LayerNorm](A) W^T + b) (+C) on the fp16-split MFMA path (layers packed with th_pack_linear_h3, M <= 8192,
// K <= 768); range: the guard's slot TH_RANGE_VIT takes max |a| of the split operand
bool th_gemm_h3_ok(int M, const ThPacked& W, bool ln);
// qkv epilogue of the ViT (optional): output columns >= dim (the keys and values of row = view * N + key) are not stored
// as fp32 but as the fp16 hi | lo operand planes of attn2_kernel (k_vit.hip: Kp [V][heads][2][Npad][64],
// Vp [V][heads][2][64][Npad] in the fragment key order) -- the per-layer kv_split launch disappears
struct ThQkvSplit {
    _Float16* Kp;
    _Float16* Vp;
    int N, Npad, heads, dim;
};
int th_gemm_h3(const float* A, int lda, int M, const ThPacked& W, const float* ln_w, const float* ln_b, float eps, int flags,
               float* C, int ldc, unsigned int* range, hipStream_t s, const ThQkvSplit* qkv = nullptr);
// C[M,N] = act(A[M,K] W^T + b) (+ C if TH_GEMM_ACCUM)
int th_gemm(const float* A, int lda, int M, const ThPacked& W, int flags, float* C, int ldc, hipStream_t s);
// C = act(LayerNorm(A rows; ln_w, ln_b, eps) W^T + b) (+ C): the normalisation happens inside the GEMM
bool th_gemm_ln_ok(int M, const ThPacked& W);
int th_gemm_ln(const float* A, int lda, int M, const ThPacked& W, const float* ln_w, const float* ln_b, float eps,
               int flags, float* C, int ldc, hipStream_t s);

// ---- point source ---------------------------------------------------------------
struct ThPointSrc {
    const float* pts;
    const float* ray_o;
    const float* ray_d;
    const float* near;
    const float* far;
    const float* tv;
    const float* omt;
    int R, S;
    const float* zv;           // th_points.z_vals (explicit depths [R,S]) or nullptr
    const float* noise;        // th_points.sigma_noise [R,S] or nullptr
};
static inline ThPointSrc th_src(const th_points* p) {
    ThPointSrc s;
    s.pts = p->pts; s.ray_o = p->ray_o; s.ray_d = p->ray_d; s.near = p->near; s.far = p->far;
    s.tv = p->t_vals; s.omt = p->one_minus_t; s.R = p->R; s.S = p->S;
    s.zv = p->z_vals; s.noise = p->sigma_noise;
    return s;
}

#ifdef __HIPCC__
// z = near*(1-t) + far*t ; p = o + d*z  -- separate roundings like the torch
// elementwise ops of if_clight_renderer.py:274,285 (contraction is disabled
// for the whole library with -ffp-contract=off).
// Reductions over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15) with data-parallel-primitive moves: VALU speed,
// no LDS round trip like the ds_bpermute behind __shfl_xor.  Every lane ends up with the full result; the pairing of
// the first two steps is xor 1, xor 2 and the mirrored lanes of the last two steps hold the same partial results as
// lane ^ 4 / lane ^ 8 would, so sums are bit-identical to the xor butterfly.
template <int CTRL>
__device__ __forceinline__ float th_dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float th_row16_sum(float s) {
    s += th_dpp_f<0xB1>(s);      // quad_perm [1,0,3,2]
    s += th_dpp_f<0x4E>(s);      // quad_perm [2,3,0,1]
    s += th_dpp_f<0x141>(s);     // row_half_mirror
    s += th_dpp_f<0x140>(s);     // row_mirror
    return s;
}
__device__ __forceinline__ float th_row16_max(float m) {
    m = fmaxf(m, th_dpp_f<0xB1>(m));
    m = fmaxf(m, th_dpp_f<0x4E>(m));
    m = fmaxf(m, th_dpp_f<0x141>(m));
    m = fmaxf(m, th_dpp_f<0x140>(m));
    return m;
}

__device__ __forceinline__ float th_sample_z(const ThPointSrc& ps, int ray, int s) {
    if (ps.zv != nullptr) return ps.zv[(long long)ray * ps.S + s];      // (wave-uniform: one scalar test)
    return ps.near[ray] * ps.omt[s] + ps.far[ray] * ps.tv[s];
}
__device__ __forceinline__ void th_get_point(const ThPointSrc& ps, long long i, float& x, float& y, float& z) {
    if (ps.pts) {
        x = ps.pts[3 * i]; y = ps.pts[3 * i + 1]; z = ps.pts[3 * i + 2];
    } else {
        int ray = (int)(i / ps.S), s = (int)(i % ps.S);
        float t = th_sample_z(ps, ray, s);
        x = ps.ray_o[3 * ray] + ps.ray_d[3 * ray] * t;
        y = ps.ray_o[3 * ray + 1] + ps.ray_d[3 * ray + 1] * t;
        z = ps.ray_o[3 * ray + 2] + ps.ray_d[3 * ray + 2] * t;
    }
}

// ---- projection + bilinear set-up shared by K2 (paint) and K5 (pixel gather) ----
struct Bilin {
    int i00, i01, i10, i11;     // linear y*W+x of nw, ne, sw, se (clamped)
    float w00, w01, w10, w11;   // weights (0 where the corner is out of range)
    int x0, y0, x1, y1;         // the clamped texel coordinates behind the indices
};

// uv -> corner indices/weights exactly as torch's grid_sampler_2d does for
// bilinear / align_corners=True / border padding:
//   g = uv*scale - 1 ; ix = ((g+1)/2)*(W-1) ; clamp to [0,W-1] ; floor ; weights.
__device__ __forceinline__ Bilin th_bilinear_setup(float u, float v, float sx, float sy, int H, int W) {
    float gx = u * sx - 1.0f, gy = v * sy - 1.0f;
    float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));
    iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
    float x0 = floorf(ix), y0 = floorf(iy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    Bilin b;
    b.w00 = (x1 - ix) * (y1 - iy);
    b.w01 = (ix - x0) * (y1 - iy);
    b.w10 = (x1 - ix) * (iy - y0);
    b.w11 = (ix - x0) * (iy - y0);
    int xi0 = (int)x0, yi0 = (int)y0, xi1 = xi0 + 1, yi1 = yi0 + 1;
    bool bx1 = xi1 <= W - 1, by1 = yi1 <= H - 1;
    if (!bx1) { b.w01 = 0.f; b.w11 = 0.f; xi1 = W - 1; }
    if (!by1) { b.w10 = 0.f; b.w11 = 0.f; yi1 = H - 1; }
    b.i00 = yi0 * W + xi0; b.i01 = yi0 * W + xi1; b.i10 = yi1 * W + xi0; b.i11 = yi1 * W + xi1;
    b.x0 = xi0; b.y0 = yi0; b.x1 = xi1; b.y1 = yi1;
    return b;
}

// cams: per view 21 floats  R[9] T[3] K[9]
__device__ __forceinline__ void th_project(const float* __restrict__ cam, float x, float y, float z, float& u,
                                           float& v) {
    float cx = fmaf(cam[2], z, fmaf(cam[1], y, cam[0] * x)) + cam[9];
    float cy = fmaf(cam[5], z, fmaf(cam[4], y, cam[3] * x)) + cam[10];
    float cz = fmaf(cam[8], z, fmaf(cam[7], y, cam[6] * x)) + cam[11];
    const float* K = cam + 12;
    float px = fmaf(K[2], cz, fmaf(K[1], cy, K[0] * cx));
    float py = fmaf(K[5], cz, fmaf(K[4], cy, K[3] * cx));
    float pz = fmaf(K[8], cz, fmaf(K[7], cy, K[6] * cx));
    u = px / pz;
    v = py / pz;
}

#endif

// ---- context ----------------------------------------------------------------------
struct ThMlpPacked {
    ThPacked fc_0, alpha_res_0, kv0, kv1, fc_1, fc_2, fc_3, feature_fc, rgb_res_0, view_fc, rgb_res_1, fc_4;
    ThPacked fc_0tok;   // fc_0[:, :192] without bias: per-frame table T' = tokens W_tok^T of the fused path
    // colour-folded forms of the three layers that read the pixel feature (in_f 260, rows 272 floats wide)
    ThPacked alpha_res_0c, rgb_res_0c, rgb_res_1c;
    bool compact_ready = false;
    // tiny heads kept as plain rows
    float *alpha_w = nullptr, *alpha_b = nullptr;   // [256], [1]
    float *rgb_w = nullptr, *rgb_b = nullptr;       // [3*128], [3]
    bool ready = false;
};
struct ThVitBlockPacked {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    ThPacked qkv, proj, fc1, fc2;
};
struct ThVitPacked {
    int depth = 0, dim = 0, heads = 0;
    ThVitBlockPacked* blocks = nullptr;   // host array
    float *norm_w = nullptr, *norm_b = nullptr;
    bool ready = false;
};

// fused MLP image (k_mlp_fused.hip): per layer [wave 4][kb][ct][plane hi/lo][lane 64][8 halves]
struct FusedLayer {
    const uint4* w;      // packed fp16 halves
    const float* bias;   // [N]
    float inv_scale;     // 2^-scale_log2
    int CT, KB;
    float inv_scale2;    // stacked layers: scale of the second column-tile family (rows N/2 ..)
};
// the dense layers of the 8-wave kernel (k_mlp_fused8_kernel.h) as v_mfma_f32_16x16x32_f16 fragments: [wave 8][k-step][column tile]
// [hi | lo][64 lanes][8 halves]; KB = number of 32-deep k-steps; bias / scales are those of the 32x32x16 images
struct Fused16 {
    FusedLayer fc_0pe, kv1, kv0, fc_2, fc_3, fc_4, vfA, vfD;
};
struct FusedParams {
    Fused16 w16;
    int waves;                     // 4: mlp_fused_kernel, 8: mlp_fused8_kernel where the hand-overs allow it (th_set_mlp_mode 1 | 2)
    FusedLayer fc_0pe;             // fc_0[:, 192:255] (the positional-encoding columns) + the fc_0 bias
    FusedLayer kv1, ar0, kv0, fc_1, fc_2, fc_3, fc_4;
    // RGB branch after the view_fc fold (k_mlp_fused_host.hip): vfA = view_fc[:, :256] feature_fc (K 256, on inter),
    // vfD = view_fc[:, 256:283] (K 32, on the view-direction rows), rst = [view_fc[:, :256] rgb_res_0 ; rgb_res_1]
    // stacked along out_f (K 384, on f): column tile 0 accumulates onto vfA / vfD, tile 1 is rgb_res_1
    FusedLayer vfA, vfD, rst;
    FusedLayer ar0c, rstc;         // colour-folded (K = 272) forms; the launcher copies them over ar0 / rst
    bool compact_ready;
    const float *alpha_w, *alpha_b, *rgb_w, *rgb_b;
    // token branch, written by K4 in TH_ROWS_FOLDED form: the neighbour blend of T' = tokens W_tok^T (fp32) and
    // the blended positional encoding (split-f16: 64 hi halves then 64 lo halves per sample)
    const float* stok;      // [P][V][256]; with tsplit != nullptr (TH_ROWS_NBR): [P][16] neighbour records instead
    const _Float16* tsplit; // nullptr, or the per-frame table T' as split rows [V][t_nc][256 hi | 256 lo] (th_tok_split)
    const float* t_inv;     // device word: 1 / (power-of-two scale of tsplit)
    int t_nc;
    const _Float16* pe;     // [P][2][64]
    // split-f16 rows written by K5 (TH_ROWS_SPLIT): K / 8 groups of [8 hi | 8 lo] halves per (sample, view)
    const _Float16* f;  // [P][V][2][384] (full) or [P][V][2][272] (compact: 256 latent | r g b | 0...)
    // TH_ROWS_TEX (k_pixtex.hip): instead of f, per tile the list of distinct corner texels and per (sample, view) four row
    // numbers + bilinear weights + the blended colour; the kernel blends the rows of tex_map (TH_MAP_SPLIT latents) itself
    const unsigned* tex_hdr;   // [tiles][4][128]
    const unsigned* tex_rec;   // [tiles][V][32][8]  {w00 w01 w10 w11} {byte offsets of the four corner rows}
    const float* tex_map;      // fold0  [V][H*W][256]: alpha_res_0' of the map's texels (map_fold_kernel)
    const float* tex_map2;     // fold12 [V][H*W][256]: [Wa rgb_res_0' (128) | rgb_res_1' (128)] of the texels
    const float* vd;    // view-direction rows [.][27]: row of compacted sample p = vd_sel ? vd_sel[p] / vd_div : p
    const int32_t* vd_sel;
    int vd_div;
    float* raw_c;       // [P][4]
    int P;
    int rgb_all;        // 0: RGB where sigma > 0, 1: everywhere, 2: nowhere (sigma-only consumers)
    unsigned int* range;   // [TH_RANGE_SLOTS] launch-wide max |hi half| (fp16 bits) per split activation, or nullptr
    long long* dbg;     // optional cycle stamps (TH_FUSED_DBG, th_fused_cycles)
    long long* cycles_buf;  // th_fused_cycles: the caller's counters (nullptr: off)
};
// range-guard table slots (th_range_read): the tensors that pass through the fp16 hi/lo split
enum { TH_RANGE_F = 0, TH_RANGE_S = 1, TH_RANGE_P = 2, TH_RANGE_N = 3, TH_RANGE_INTER = 4, TH_RANGE_F4 = 5, TH_RANGE_CONV = 6,
       TH_RANGE_VIT = 7 };
size_t th_fused_pack_bytes();
// folded: nullptr or the three colour-folded fp32 layers {alpha_res_0, rgb_res_0, rgb_res_1} (in_f 260)
int th_fused_pack(const th_mlp_weights* w, const th_linear* folded, void* store, FusedParams* out, hipStream_t s);
// stok / pe: TH_ROWS_FOLDED output of K4; f: TH_ROWS_SPLIT rows (4 * K bytes per (sample, view))
// vd rows are addressed through vd_sel / vd_div (the per-RAY embedding table is read in place: sample index / S),
// or directly by the compacted sample index when vd_sel == nullptr
// tsplit / t_inv / t_nc: nullptr / nullptr / 0, or the split token table of th_tok_split -- `stok` then holds K4's
// TH_ROWS_NBR neighbour records
int th_mlp_fused_forward(const FusedParams& base, const ThMlpPacked& heads, int V, int P, const float* stok,
                         const void* pe, const void* f, int f_ld, const float* vd, const int32_t* vd_sel, int vd_div, int rgb_all,
                         float* raw_c, unsigned int* range, hipStream_t s, const void* tsplit = nullptr,
                         const float* t_inv = nullptr, int t_nc = 0, const float* tex_map = nullptr, size_t tex_stride = 0);
// The > 64 KB dynamic-LDS attribute of a kernel belongs to the DEVICE's copy of it: set once per device, not once per process (a
// process that drives two GPUs otherwise launches with the default limit on the second).  `done` = a static mask per call site.
static inline bool th_lds_attr_needed(unsigned long long* done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    if ((*done >> dev) & 1ull) return false;
    *done |= 1ull << dev;
    return true;
}
// alpha_res_0 / rgb_res_0 / rgb_res_1 applied to the texels of the (cropped) split map: fold [2][V][H*W][256] (k_mlp_fused_kernel.h)
int th_map_fold_launch(const FusedParams& base, const float* map_split, int V, int H, int W, const int32_t* box, float* fold,
                       unsigned int* range, hipStream_t s, const unsigned* demand = nullptr /* k_demand.hip buffer: listed texels only */);
// K5t (k_pixtex.hip): texel lists + records for P samples into `out` (th_pixtex_bytes); with tex_map != nullptr
// th_mlp_fused_forward reads `f` as that block instead of rows
size_t th_pixtex_bytes(int V, long long P);
int th_pixtex_launch(int V, int H, int W, const ThPointSrc* ps, const int32_t* sel, int P, const float* cams,
                     const float* scale, void* out, hipStream_t s);
// In place: the per-frame table T' [rows][256] fp32 -> [rows][256 hi | 256 lo] fp16 halves of T' * 2^k, k chosen on the
// device so that max |T'| lands in [2^12, 2^13) (lo halves stay normal numbers); sc[0] = 2^-k, sc[1] scratch (the
// maximum's bits).  A non-finite table raises the guard's TH_RANGE_VIT slot (its producer is the ViT).
int th_tok_split(float* tprime, int rows, float* sc, unsigned int* range, hipStream_t s);

struct th_ctx {
    void* fused_store = nullptr;
    FusedParams fused{};
    bool fused_ready = false;
    int mlp_mode = 1;                 // 1: fused fp16x3-split MFMA kernel, 0: layer-by-layer fp32 MFMA
    int vit_mode = 1;                 // 1: TransHE dense layers on the fp16-split MFMA path (th_gemm_h3), 0: fp32 MFMA
    int tok_gather = 1;               // 1: TH_ROWS_NBR hand-over (token blend inside the fused kernel), 0: TH_ROWS_FOLDED
    int tex_rows = 1;                 // 1: TH_ROWS_TEX hand-over of the pixel features (texel lists, blend inside the fused kernel)
    int device = 0;
    void* mlp_store = nullptr;
    void* vit_store = nullptr;
    ThMlpPacked mlp;
    ThVitPacked vit;
    int32_t* host_pinned = nullptr;   // small pinned read-back buffer ([0..15] immediate reads, [16..31] prepass slots)
    // th_render_prepass tokens: the hull / compaction stage of a coming th_render_rays already ran into `ws`.
    // Several may be pending (a frame pipeline keeps the ray-only stage of the next frames in flight, each in its
    // own workspace); th_render_rays picks the one queued for its workspace.  Counts land in host_pinned[16 + 4*slot].
    struct Prepass {
        hipEvent_t ev = nullptr;
        const void* ws = nullptr;
        const void* rays = nullptr;
        int R = 0, S = 0;
        bool valid = false;
        // th_render_pregather: the pixel rows (K5) and neighbour records (K4) of the first `npre` valid SAMPLES are already
        // in region A of the shading pool `pre_pool`, written for this map / these token centres; ev2 marks their completion
        hipEvent_t ev2 = nullptr;
        int npre = 0;
        const void* pre_map = nullptr;
        const void* pre_centres = nullptr;
        const void* pre_pool = nullptr;
        const void* grid_centres = nullptr;         // th_render_pregrid: K4's candidate grid of these centres is in the workspace
        const void* pre_tokens = nullptr;           // th_render_pregather_early: T' of these tokens already sits in the workspace
        const void* demand = nullptr;               // th_render_predemand: this demand buffer describes the prepass's sample list
        const void* map_done = nullptr;      // the cropped map this prepass's frame has already completed (written once)
    };
    static constexpr int kPrepassSlots = 4;
    Prepass prepass[kPrepassSlots];
    int prepass_rr = 0;
    int n_cu = 256;
    // second stream of the pre-gather stage: K4 (neighbour records: VALU / LDS work, no row gather since TH_ROWS_NBR)
    // runs beside K5 (pixel-feature gather: texture-path bound) instead of behind it
    const void* map_completed = nullptr;    // a demand-driven map that a later call had to write in full (shade_points) ...
    const void* map_completed_demand = nullptr;   // ... and the demand buffer it was made from (the pair identifies the frame)
    hipStream_t aux = nullptr, aux2 = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr, aux2_join = nullptr;
    // th_render_pregather_early: the point of the last th_render_rays' stream where its per-sample stage (fused MLP +
    // scatter) was complete -- recorded before the compositing -- and the stream / shading pool it belongs to
    hipEvent_t after_shade = nullptr;
    hipStream_t after_shade_stream = nullptr;
    const void* after_shade_pool = nullptr;
    bool after_shade_valid = false, pregather_early = false;
    void* prof = nullptr;             // ThProf (th_api.hip)
    // range guard (th_range_*): device table the kernels merge their maxima into, pinned snapshots + events
    unsigned int* range_dev = nullptr;
    unsigned int* range_host = nullptr;            // [kRangeSnaps][TH_RANGE_SLOTS]
    static constexpr int kRangeSnaps = 8;
    hipEvent_t range_ev[kRangeSnaps] = {};
    int range_gen[kRangeSnaps] = {-1, -1, -1, -1, -1, -1, -1, -1};   // generation each pinned buffer holds
    int range_gen_next = 0, range_last = -1;
    // host time spent inside BLOCKING waits of the entry points (hipEventSynchronize / hipStreamSynchronize on counts
    // and range snapshots): bench.py subtracts it from its host-side loop time to get the pure queueing cost
    double host_wait_ms = 0.0;
};
// RAII stopwatch around a blocking wait (th_host_wait_read drains the sum)
struct ThWaitClock {
    th_ctx* c;
    timespec t0;
    explicit ThWaitClock(th_ctx* c_) : c(c_) { clock_gettime(CLOCK_MONOTONIC, &t0); }
    ~ThWaitClock() {
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        c->host_wait_ms += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    }
};

// ---- launchers (one group per .hip file) -------------------------------------------
// k_hull.hip
size_t th_hull_ws(int n_verts);
// (info_zero: optional 16-int block cleared by the grid build -- saves the caller a memset launch)
int th_hull_mask_launch(const ThPointSrc& ps, long long P, const float* verts, int nv, float thresh,
                        uint8_t* mask, int32_t* ray_hit, void* ws, size_t ws_bytes, hipStream_t s,
                        int32_t* info_zero = nullptr);
// count of hit rays -> dev_info[0], then compaction with the reference's R' <= thr rule applied on the fly
// (dev_info[1] = 1 and mask rewritten when it fires); dev_info[0..1] zero on entry
int th_compact_mask_rule(uint8_t* mask, long long P, const int32_t* ray_hit, int R, int S, int thr, int32_t* dev_info,
                         int32_t* idx_out, int32_t* dev_count, void* ws, size_t ws_bytes, hipStream_t s);
// compaction helpers (k_hull.hip)
size_t th_compact_ws(long long P);
// counts[0]=n_valid written to dev_count; idx_out ascending
int th_compact_mask(const uint8_t* mask, long long P, int32_t* idx_out, int32_t* dev_count, void* ws,
                    size_t ws_bytes, hipStream_t s);
// small-frame rule: if #hit rays <= thr, mask := ray_hit (all samples of hit rays); dev_info[0]=hit rays, [1]=mode
int th_small_frame_rule(uint8_t* mask, const int32_t* ray_hit, int R, int S, int thr, int32_t* dev_info,
                        hipStream_t s);

// k_pool.hip
int th_paint_launch(const float* map, int V, int C, int H, int W, const float* verts, int nv, const float* cams,
                    const float* scale, const uint8_t* viz, float* painted, hipStream_t s);
int th_segmean_launch(const float* src, int batch, long long batch_stride, int width, const int32_t* off,
                      const int32_t* mem, int nc, float* out, hipStream_t s);
int th_segmean_rot_launch(const double* blend, const int32_t* off, const int32_t* mem, int nc, float* rot,
                          hipStream_t s);
int th_nchw_to_nhwc_launch(const float* src, int V, int C, int H, int W, float* dst, hipStream_t s);

// Row format of the per-(sample, view) operand rows h / f handed from the producer kernels to the MLP stage:
//   TH_ROWS_F32    K fp32 values (layer-by-layer fp32 MFMA path, public th_dparf_encode / th_pixel_gather)
//   TH_ROWS_SPLIT  K / 8 groups of [8 fp16 "hi" halves | 8 fp16 "lo" halves], x = hi + lo to 2^-22 (fused kernel: the
//                  operand is copied to LDS by LDS-DMA, 16-byte pieces = one plane's 8 consecutive halves, with no
//                  conversion pass); same 4K bytes per row
//   TH_ROWS_FOLDED (K4 only) the token table handed to K4 is T' = tokens fc_0[:, :192]^T (256 wide): `out` gets the
//                  fp32 neighbour blend of T' rows [P][V][256], `pe_out` the blended 63-wide positional encoding
//                  as one split-f16 row of 64 + 64 halves per SAMPLE (fused kernel, see its token branch)
//   TH_ROWS_NBR    (K4 only) no rows at all: `out` gets one 64-byte neighbour record per sample -- int idx[7], 0,
//                  float w[7], 0 (the 7 nearest token centres in (distance, index) order and their softmax weights) --
//                  and `pe_out` the split-f16 positional encoding like TH_ROWS_FOLDED.  The fused kernel then forms the
//                  blend of T' rows itself, on the matrix pipe, from the L2-resident table (k_mlp_fused_kernel.h)
enum { TH_ROWS_F32 = 0, TH_ROWS_SPLIT = 1, TH_ROWS_FOLDED = 2, TH_ROWS_NBR = 3 };
// k_dparf.hip
int th_dparf_launch(const float* pts_smpl, const ThPointSrc* ps, const float* Rh, const float* Th,
                    const int32_t* sel, int P, const float* centres, const float* rot, const float* tokens,
                    int V, int nc, float alpha, float* out, float* pe_out, int fmt, const void* grid_ws,
                    hipStream_t s);
// exact candidate grid over the token centres for the 7-NN scan of K4 (per frame); grid_ws = nullptr: full scan
// (the grid builder keeps four cells' squared distances in the default 64 KiB of dynamic LDS: nc <= 4096)
static inline bool th_dparf_grid_ok(int nc) { return nc >= 7 && (size_t)nc * 16 <= 64 * 1024; }
size_t th_dparf_grid_ws(int nc);
int th_dparf_grid_build(const float* centres, int nc, void* ws, size_t ws_bytes, hipStream_t s);
// k_pixfeat.hip
int th_pixgather_launch(const float* map, int V, int C, int H, int W, const float* pts_world,
                        const ThPointSrc* ps, const int32_t* sel, int P, const float* cams, const float* scale,
                        float* out, int ldo, int fmt, hipStream_t s, unsigned int* range = nullptr);
int th_gather_chan_major_launch(const float* pf /*[V,C,Pall]*/, int V, int C, long long Pall, const int32_t* sel,
                                int P, float* out /*[P,V,C]*/, int fmt, hipStream_t s, unsigned int* range = nullptr);
// k_mlp.hip
size_t th_mlp_ws(int V, int P);
// h [P*V,256], f [P*V,f_ld] (f_ld 384: full rows, 272: compact rows + colour-folded layers), vd rows [P,27]
// (gathered), -> raw_c [P,4]
int th_mlp_forward(const ThMlpPacked& W, int V, int P, const float* h, const float* f, int f_ld, const float* vd,
                   float* raw_c, void* ws, size_t ws_bytes, hipStream_t s);
int th_gather_rows_launch(const float* src, int width, const int32_t* sel, int div, int P, float* out, hipStream_t s);
// raw[sel[p]] = raw_c[p] (rgb zeroed where sigma<=0 unless rgb_all)
int th_scatter_raw_launch(const float* raw_c, const int32_t* sel, int P, int rgb_all, float* raw, hipStream_t s);
// k_composite.hip
// mask (optional, uint8 per sample): raw is only read where mask != 0, elsewhere it counts as zero
int th_composite_launch(const float* raw, const float* z, const ThPointSrc& ps, int white, float* rgb, float* acc,
                        float* depth, float* wout, const uint8_t* mask, hipStream_t s, const int32_t* ray_hit = nullptr);
int th_view_embed_launch(const float* d, int R, int res, float* out, hipStream_t s, const int32_t* hit = nullptr);
// k_vit.hip
size_t th_vit_ws(int V, int N, int dim, int heads);
int th_vit_launch(const ThVitPacked& W, const float* x, const float* pe, int V, int N, float* out, void* ws,
                  size_t ws_bytes, hipStream_t s, unsigned int* range = nullptr, bool allow_h3 = true);

// k_smpl.hip
size_t th_smpl_ws(int nv);
int th_smpl_launch(const th_smpl_model& m, const float* pose_aa, const float* R, const double* beta, double* verts,
                   double* joints, double* T, void* ws, size_t ws_bytes, hipStream_t s);
// k_rays.hip
int th_gen_rays_launch(const float* K, const float* R, const float* T, const float* bounds, int H, int W, float* ray_o,
                       float* ray_d, float* near_out, float* far_out, uint8_t* mask, hipStream_t s);
// k_mcubes.hip
size_t th_mc_ws(int X, int Y, int Z);
int th_mc_count_launch(const float* cube, int X, int Y, int Z, float iso, void* ws, size_t ws_bytes, long long* counts_dev,
                       hipStream_t s);
int th_mc_emit_launch(const float* cube, int X, int Y, int Z, float iso, const void* ws, int x0, int x1, const double* scale,
                      const double* origin, double* verts, int* tris, hipStream_t s);
int th_mc_prefix_launch(const void* ws, int X, int Y, int Z, int x, long long* out_dev, hipStream_t s);
int th_bound_mask_launch(const int32_t* corners_xy /* host [8][2] */, int H, int W, uint8_t* mask, hipStream_t s);
// k_encoder.hip
int th_upsample_concat_launch(const float* img, const float* lat0, const float* lat1, const float* lat2,
                              const int* dims, int V, int H, int W, const float* wc, const float* bc, float* out,
                              hipStream_t s, int split, const int32_t* box /* device [V][4] or null: th_map_box_launch */,
                              const unsigned* need = nullptr /* demand-driven map: one bit per texel (k_demand.hip), W % 64 == 0 */);
// k_demand.hip: the texels the valid samples `sel[0 .. info[2])` (+ the painted vertices) read -> demand buffer
size_t th_demand_bytes(int V, int H, int W);
int th_demand_launch(const ThPointSrc& ps, const int32_t* sel, const int32_t* info, const float* cams, const float* scale, int V,
                     int H, int W, const float* verts_paint, int n_paint, void* demand, hipStream_t s);
int th_map_box_launch(const float* va, int na, const float* vb, int nb, const float* cams, int V, const float* scale, int H,
                      int W, float reach, int32_t* box, hipStream_t s);
// W' [N,260] = [W[:, :256] | W[:, 256:384] Wc | 0], b' = b + W[:, 256:384] bc  (fp64 accumulation)
// K12 (k_conv.hip): convolutions of the ResNet stem on the fp16-split MFMA path
size_t th_conv_pack_size(int COUT, int CIN, int KS);
int th_conv_pack_launch(const float* w, int COUT, int CIN, int KS, void* out, size_t out_bytes, float* inv_scale_host,
                        hipStream_t s);
bool th_conv2d_built(int CIN, int COUT, int KS, int stride);
int th_conv2d_launch(const float* x, int N, int CIN, int H, int W, const void* packed, float inv_scale, int COUT, int KS,
                     int stride, float* y, hipStream_t s, unsigned int* range = nullptr, float2* stats = nullptr,
                     int* np_out = nullptr);
int th_maxpool3x3s2_launch(const float* x, int planes, int H, int W, float* y, hipStream_t s);
size_t th_bn_ws(int N, int C, int HW);
int th_bn_act_launch(const float* x, const float* res, int N, int C, int HW, const float* gamma, const float* beta,
                     float eps, float momentum, float* run_mean, float* run_var, int relu, float* y, void* ws,
                     size_t ws_bytes, hipStream_t s, int eval = 0, const void* conv_stats = nullptr, int conv_np = 0);
int th_fold_color_launch(const float* W /*[N,384]*/, const float* b, const float* wc /*[128,3]*/, const float* bc,
                         int N, float* Wo /*[N,260]*/, float* bo /*[N]*/, hipStream_t s);
int th_segmean_masked_launch(const float* rows, int V, int width, const uint8_t* viz, int nv, const int32_t* off,
                             const int32_t* mem, int nc, float* out, hipStream_t s);
