/*
 * transhuman_hip.h -- C ABI of libtranshuman_hip.so (gfx950 / MI355X only).
 *
 * The reference (pansanity666/TransHuman) is 100 % Python; its "native" hot
 * path is a sequence of torch / pytorch3d kernels launched from
 *   lib/networks/renderer/if_clight_renderer.py  (Renderer.render_fast/_render/batchify_rays)
 *   lib/networks/cross_transformer.py            (Network.forward and helpers)
 *   lib/networks/vision_transformer.py           (VisionTransformer.forward)
 *   lib/networks/renderer/nerf_net_utils.py      (raw2outputs)
 *   lib/networks/renderer/if_mesh_renderer.py    (Renderer.render)
 * Each entry point below names the reference lines it replaces.  The Python
 * classes in transhuman_amd/networks/ keep the reference's operator API and
 * bind these symbols with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - all floating point data is fp32 unless stated (blend matrices: f64);
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *  - functions return 0 on success, <0 on error; th_last_error() gives the text;
 *  - no function allocates device memory: scratch comes from the caller via
 *    (workspace, workspace_bytes), sized with the matching *_workspace_bytes();
 *  - a th_ctx is bound to one device and must be used from one thread at a time.
 */
#ifndef TRANSHUMAN_HIP_H
#define TRANSHUMAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TH_ABI_VERSION 12

typedef struct th_ctx th_ctx;
typedef void* th_stream;

/* ---- library / context ------------------------------------------------ */
int         th_abi_version(void);
const char* th_last_error(void);
int         th_ctx_create(int device, th_ctx** out);
void        th_ctx_destroy(th_ctx* ctx);
/* sizeof() of a struct of this header as the LIBRARY was compiled with it: "th_points", "th_frame", "th_map_source",
 * "th_linear", "th_mlp_weights", "th_vit_block", "th_smpl_model"; 0 for an unknown name.  A binding written in another language
 * (INTEGRATION.md's ctypes stub) checks its own struct definitions against it -- a struct that is short by one field
 * makes the library read garbage pointers (ABI 10 grew th_points by two). */
size_t      th_sizeof(const char* type_name);

/* ---- stage timing (HIP events on the launch stream) --------------------- */
/* When enabled, the frame-level entry points bracket their stages with
 * hipEventRecord on `stream`; th_profile_read drains the accumulated
 * per-phase milliseconds / launch counts (arrays of TH_PROF_PHASES). */
enum { TH_PROF_HULL = 0, TH_PROF_DPARF = 1, TH_PROF_GATHER = 2, TH_PROF_MLP = 3, TH_PROF_COMPOSITE = 4,
       TH_PROF_VIT = 5, TH_PROF_FOLD = 6 /* th_map_fold: the f-consuming layers applied to the map's texels */, TH_PROF_PHASES = 8 };
int th_profile_enable(th_ctx* ctx, int on);
int th_profile_read(th_ctx* ctx, double* ms_out, int64_t* count_out);
/* Host milliseconds the calling thread spent inside the BLOCKING waits of the entry points (sample counts of
 * th_render_rays / th_render_pregather / th_eval_sigma_grid, th_range_read) since the last call; reads and clears.
 * bench.py subtracts it from the wall time of its queueing loop: what is left is the host's own cost per frame
 * (Python + ctypes + launch calls), the figure that decides whether a short multi-GPU frame is host-bound. */
int th_host_wait_read(th_ctx* ctx, double* ms_out);
/* Shader-clock probe: one wave, ~15 us, queued on `stream`; out_dev[0] = shader-clock ticks, out_dev[1] = the same
 * interval in 10 ns units of the constant 100 MHz counter (GHz = ticks / (10 * units)).  bench.py queues one behind
 * the dominant kernel of every timed step and reports the median: the frequency the chip sustains under the load
 * the roofline is priced at. */
int th_clock_probe(th_ctx* ctx, int64_t* out_dev /* [3] */, th_stream stream);
/* Cycle accounting of the fused MLP kernel (K6) from INSIDE its launches: `counters_dev` = 64 zeroed int64 words of device
 * memory (NULL switches it off).  While set, thread 0 of every 16th tile adds: [0] sampled tiles, [1..61] shader cycles
 * between consecutive workgroup barriers (the tile's phases, in order), [62] shader cycles and [63] ticks of the constant
 * 100 MHz counter over the whole tile -- [62] / (10 [63]) is the shader clock in GHz the chip ran at UNDER this kernel (the
 * probe above runs on whatever CU has room: beside the kernel, not inside it).  Costs ~0.5 % of the launch; bench.py uses it
 * for a short post-run measurement. */
int th_fused_cycles(th_ctx* ctx, int64_t* counters_dev);

/* ---- weights ----------------------------------------------------------- */
/* One dense layer: weight row-major [out_f][in_f] (a Conv1d(k=1)/Linear
 * weight with the trailing 1 squeezed), bias [out_f] or NULL. */
typedef struct {
    const float* w;
    const float* b;
    int out_f;
    int in_f;
} th_linear;

/* Per-point MLP of Network (cross_transformer.py:96-126). */
typedef struct {
    th_linear fc_0, alpha_res_0;
    th_linear key0, val0;          /* spatial_key_value_0 (pixel branch)  */
    th_linear key1, val1;          /* spatial_key_value_1 (token branch)  */
    th_linear fc_1, fc_2, fc_3, alpha_fc;
    th_linear feature_fc, rgb_res_0, view_fc, rgb_res_1, fc_4, rgb_fc;
    /* optional (w == NULL: absent): encoder.upsample_color, the 1x1 conv 3->128 that produces the last 128
     * channels of pixel_feat_map (encoder.py:95,:139).  When given, the three layers that read the 384-channel
     * pixel feature (alpha_res_0, rgb_res_0, rgb_res_1) are ALSO kept in a colour-folded form
     *   W' = [W[:, :256] | W[:, 256:] Wc | 0]  (in_f 260),  b' = b + W[:, 256:] bc
     * which is what the per-sample stage uses with a compact map (th_frame.map_channels = 260): bilinear
     * sampling is linear with weights summing to 1, so sampling r,g,b and lifting afterwards equals
     * sampling the lifted map. */
    th_linear upsample_color;
} th_mlp_weights;

/* One transformer block (vision_transformer.py:285-307). */
typedef struct {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    th_linear qkv, proj, fc1, fc2;
} th_vit_block;

/* Copies + re-packs the weights into the MFMA operand layout owned by ctx.
 * `dparf_alpha`/`n_freq`/`knn` = cfg.KNN_DIST_ALPHA / KNN_FREQ / KNN. */
int th_set_mlp_weights(th_ctx* ctx, const th_mlp_weights* w, th_stream stream);
/* K6 implementation switch: 1 (default) = one fused kernel per 32-sample tile, dense layers on
 * v_mfma_f32_32x32x16_f16 with fp16 hi/lo operand splitting (3 products, fp32 accumulate: fp32-class
 * accuracy); 0 = one fp32-MFMA GEMM launch per layer (exact fp32 products; also the path for V = 4). */
int th_set_mlp_mode(th_ctx* ctx, int mode);
/* Form of the fused kernel (mode 1) on the frame-level path (texel lists + neighbour records): 8 (default) = 512-thread
 * workgroups, two waves per SIMD, every wave owning half the output columns of a layer, dense layers on
 * v_mfma_f32_16x16x32_f16 (k_mlp_fused8_kernel.h, round 6); 4 = the 256-thread form of rounds 2-5 (one wave per SIMD,
 * v_mfma_f32_32x32x16_f16), which also serves every other hand-over (row operands, V = 1..3).  Same tile, same
 * arithmetic (three fp16 products per fp32 MAC); results differ by fp32 summation order only.  New contexts read
 * TH_FUSED_WAVES. */
int th_set_fused_waves(th_ctx* ctx, int waves);
/* K4 -> K6 hand-over of the token branch on the fused path (Network.get_human_representation, cross_transformer.py:151-205,
 * feeding fc_0, :291-295): 1 (default) = K4 writes, per sample, its 7 nearest token centres (as slots of the per-tile
 * union) and their softmax weights, and the fused kernel blends the rows of the per-frame table tokens fc_0[:, :192]^T on
 * the matrix pipe (0.34 KB per sample through HBM); 0 = K4 blends the rows in fp32 and hands them over (3.3 KB per
 * sample; a ray shard then equals the whole frame bit for bit instead of to fp32 rounding).  Env TH_TOK_GATHER=0 makes
 * 0 the default of new contexts. */
int th_set_tok_gather(th_ctx* ctx, int on);
/* K5 -> K6 hand-over of the pixel-aligned features on the fused path (get_pixel_aligned_feature,
 * if_clight_renderer.py:210-269, feeding alpha_res_0 / rgb_res_0 / rgb_res_1, cross_transformer.py:300-346), for frames
 * whose map is TH_MAP_SPLIT: 1 (default) = per 32-sample tile the list of DISTINCT corner texels of its V x 32 rows
 * (~79 of 384 on the headline frame) and per row four row numbers + bilinear weights + the blended colour (160 B per
 * sample); the fused kernel copies those texels -- of the FOLDED maps, th_map_fold / th_frame.map_fold: the layers that read
 * the features are applied to the map's texels once per frame -- into LDS and blends them itself with K5's weights; nothing of
 * the rows travels through HBM and two of the kernel's GEMMs are gone; 0 (or a frame without map_fold) = K5 writes the rows
 * (3.3 KB per sample, read back twice).  Env TH_ROWS_TEX=0 makes 0 the default of new contexts.  Ask th_shade_pool_bytes again
 * after a change. */
int th_set_tex_rows(th_ctx* ctx, int on);

/* Range guard of the fp16 hi/lo split arithmetic (fused MLP kernel, its producer K5, the ResNet-stem convolutions).
 * The reference computes this path in fp32 (cross_transformer.py:291-353 has no autocast); the fused kernel is
 * fp32-class only while every split value satisfies |x| < 65504 (fp16 range) and the tensor it belongs to is not
 * uniformly tiny (below 2^-14 the halves are subnormal: absolute resolution 2^-25).  The kernels keep launch-wide
 * maxima of |x| per split tensor in a device table; th_range_snapshot queues a copy of the table (and clears it)
 * behind the work issued so far on `stream` and returns a snapshot id (>= 0, monotonic; 8 pinned buffers rotate),
 * th_range_read waits for that copy and returns the TH_RANGE_SLOTS words (return value 2: the id is older than the 8
 * most recent snapshots and its buffer has been reused -- the caller must treat the frame as unchecked and render it
 * again): slots 0..5 = fp16 bit pattern of max |hi half| of
 * {f rows (K5), s, p, n, inter, fc_4 operand}; slot 6 = fp32 bit pattern of max |input| of the stem convolutions;
 * inf / NaN show up as values >= 0x7C00 (fp16 slots) / 0x7F800000 (fp32 slot).  th_render_rays,
 * th_eval_sigma_grid and th_network_forward take a snapshot at their end (th_range_last_slot).  A caller that finds
 * a slot >= TH_RANGE_FP16_LIMIT, or a non-zero slot below TH_RANGE_FP16_FLOOR, re-renders with th_set_mlp_mode(ctx, 0)
 * (per-layer fp32 MFMA): transhuman_amd/hip.py does exactly that. */
#define TH_RANGE_SLOTS 8
#define TH_RANGE_FP16_LIMIT 0x7B53u     /* 6.0e4 */
#define TH_RANGE_FP16_FLOOR 0x2400u     /* 2^-6  */
/* slot 7: fp16 bit pattern of max |a| of the operands of TransHE's dense layers (th_gemm_h3).  Slots 6 and 7 are
 * written by the stream that computes a frame's constants and are sticky (not cleared by a snapshot; slot 7 is cleared by
 * th_set_vit_weights, slot 6 by th_set_mlp_weights).  th_set_vit_mode(ctx, 0) moves TransHE's dense layers back to the fp32 MFMA GEMMs;
 * 1 (default): fp16-split arithmetic.  (ABI 5 had a mode 2, the whole forward as one persistent launch: correct, but slower
 * than the 63 launches and resident-workgroup-count dependent -- removed in ABI 6) */
int th_set_vit_mode(th_ctx* ctx, int mode);
int th_range_snapshot(th_ctx* ctx, th_stream stream);
int th_range_read(th_ctx* ctx, int slot, uint32_t* out /* [TH_RANGE_SLOTS] */);
int th_range_last_slot(th_ctx* ctx);
/* Samples shaded per pass of the per-sample stage (process-wide; default 524288, the reference's
 * batchify_rays chunk is 32768, if_clight_renderer.py:575).  Results do not depend on it; workspace
 * sizes do, so call it before the *_workspace_bytes() queries. */
int th_set_chunk_samples(int n);
int th_set_vit_weights(th_ctx* ctx, int depth, int dim, int heads, const th_vit_block* blocks,
                       const float* norm_w, const float* norm_b, th_stream stream);

/* ---- generic dense layer on the fp32 MFMA pipe (building block) -------- */
/* C[M, out_f] = act(A[M, in_f] * W^T + b);  act: 0 none, 1 relu, 2 gelu(erf).
 * lda/ldc in floats.  Packs W on the fly into `workspace`. */
size_t th_linear_workspace_bytes(int out_f, int in_f);
int th_linear_forward(th_ctx* ctx, const float* A, int lda, int M, const th_linear* lin, int act,
                      float* C, int ldc, void* workspace, size_t workspace_bytes, th_stream stream);

/* ---- K1: sample placement + SMPL-hull mask ------------------------------ */
/* if_clight_renderer.py:271-287 (get_sampling_points) + :440-444
 * (knn_points K=1, sqrt, < 0.1, per-ray any) and if_mesh_renderer.py:53-56.
 * Points come either from rays (pts==NULL: p = o + d*(near*(1-t)+far*t),
 * t_vals/one_minus_t are the S linspace values) or explicitly (pts[P,3]).
 * mask_out: uint8[P] (P = R*S or P); ray_hit_out: int32[R] or NULL. */
typedef struct {
    const float* pts;          /* explicit points [P,3] or NULL            */
    const float* ray_o;        /* [R,3]                                    */
    const float* ray_d;        /* [R,3]                                    */
    const float* near;         /* [R]                                      */
    const float* far;          /* [R]                                      */
    const float* t_vals;       /* [S] torch.linspace(0,1,S)                */
    const float* one_minus_t;  /* [S] 1 - t_vals                           */
    int R, S;                  /* rays, samples per ray (S=1 with pts)     */
    /* ABI 10 -- the reference's two sampling randomisations (NULL = off, what run.py:22,68,123 renders with):          */
    const float* z_vals;       /* [R,S] explicit sample depths: the stratified jitter of get_sampling_points            */
                               /* (if_clight_renderer.py:276-283, cfg.perturb > 0 in train() mode) replaces            */
                               /* near*(1-t)+far*t in every stage (hull test, neighbour records, texel lists, deltas)  */
    const float* sigma_noise;  /* [R,S] added to sigma in front of the relu of raw2alpha (nerf_net_utils.py:39-44,      */
                               /* cfg.raw_noise_std > 0: randn * std) on the rays that are composited (hit rays)       */
} th_points;

size_t th_hull_workspace_bytes(int n_verts);
int th_hull_mask(th_ctx* ctx, const th_points* p, const float* verts_world, int n_verts, float thresh,
                 uint8_t* mask_out, int32_t* ray_hit_out, void* workspace, size_t workspace_bytes,
                 th_stream stream);

/* ---- K2: paint SMPL vertices + cluster mean pooling --------------------- */
/* paint_neural_human :95-184 (project, bilinear grid_sample(align_corners,
 * border) of the NCHW holder map, zero invisible vertices) followed by
 * voxelization/can_body_grouping :356-371,:415-427 (CSR segmented mean).
 * cams: per view R[9], T[3], K[9] row-major = 21 floats.  scale_xy: the two
 * floats of sample_from_feature_map :193.  tokens_out [V,N_c,C]. */
int th_paint_group(th_ctx* ctx, const float* holder_map_nchw, int V, int C, int H, int W,
                   const float* verts_world, int n_verts, const float* cams, const float* scale_xy,
                   const uint8_t* vizmap, const int32_t* csr_offsets, const int32_t* csr_members,
                   int n_clusters, float* painted_out /* [V,n_verts,C] or NULL */, float* tokens_out,
                   th_stream stream);
/* voxelization of per-vertex rows: fp32 [n_verts,width] -> [N_c,width] */
int th_segment_mean_f32(th_ctx* ctx, const float* src, int width, const int32_t* csr_offsets,
                        const int32_t* csr_members, int n_clusters, float* out, th_stream stream);
/* blend matrices: f64 [n_verts,16] -> mean in f64 -> top-left 3x3 as fp32
 * [N_c,9] (cross_transformer.py:185) */
int th_segment_mean_rot_f64(th_ctx* ctx, const double* blend, const int32_t* csr_offsets,
                            const int32_t* csr_members, int n_clusters, float* rot_out,
                            th_stream stream);

/* channel counts of the channels-last pixel map */
#define TH_MAP_FULL    384   /* pixel_feat_map as the reference builds it: 256 latent + 128 lifted colour  */
#define TH_MAP_COMPACT 260   /* 256 latent | r g b | 0 : the colour lift is folded into the consumers        */
#define TH_MAP_SPLIT   256   /* the compact map as two planes: [V,H,W,256] latents (1 KiB rows: one aligned wave load per
                                corner texel) immediately followed by [V,H,W,4] (r, g, b, 0); same consumers / folds as 260 */

/* ---- K8: encoder tail written channels-last + painting from that map ------ */
/* encoder.py:133-146: bilinear-upsample (align_corners=True) the three ResNet latents
 * lat0 [V,64,h0,w0], lat1 [V,64,h1,w1], lat2 [V,128,h2,w2] to HxW, append upsample_color(img)
 * (1x1 conv 3->128, color_w [128,3], color_b [128]) and write pixel_feat_map DIRECTLY channels-last
 * [V,H,W,384] (the layout K5 gathers from).  dims_host = {h0,w0,h1,w1,h2,w2} (host ints).
 * color_w == NULL: write the COMPACT map [V,H,W,260] = 256 upsampled latent channels | r g b | 0 instead
 * (a third less HBM written per frame and read per sample; consumers use colour-folded weights). */
int th_upsample_concat_nhwc(th_ctx* ctx, const float* img, const float* lat0, const float* lat1,
                            const float* lat2, const int32_t* dims_host, int V, int H, int W,
                            const float* color_w, const float* color_b, float* out_nhwc, th_stream stream);
/* the compact map in the TH_MAP_SPLIT layout: out = [V,H,W,256] latents followed by [V,H,W,4] (r, g, b, 0) */
int th_upsample_concat_split(th_ctx* ctx, const float* img, const float* lat0, const float* lat1, const float* lat2,
                             const int32_t* dims_host, int V, int H, int W, float* out, th_stream stream);
/* Cropped map.  The reference builds pixel_feat_map over the whole image (encoder.py:133-146) and then samples it only
 * at points within hull_thresh of a target vertex (if_clight_renderer.py:440-444 -> :210-269) and at the projected input
 * vertices (:168-172).  th_map_box computes, per view, the texel box [x0,y0,x1,y1] (inclusive, box_out: device int32
 * [V][4]) that contains every texel such a gather can read: the projections of the eight corners of the axis-aligned cube
 * of half-width `reach` around every vertex of verts_a [na,3] and verts_b [nb,3] (world space; u, v are linear-fractional,
 * so their extrema over a cube in front of the camera sit at corners), in the texel coordinates of grid_sample
 * (align_corners=True, scale_xy as in th_pixel_gather), widened by two texels and clamped to the image (border padding
 * clamps monotonically); the whole image for a view with a cube corner at or behind its camera plane.
 * Behind the V boxes box_out also receives, per view and image row, the SPAN [x0, x1] (inclusive; x1 < x0: empty row) of
 * the vertices' own projected boxes that touch the row -- the body's outline inside the box, about half of it: box_out holds
 * V * 4 + V * H * 2 + V int32 (H <= 4096; the last V words are scratch of the kernels).  th_upsample_concat_split_box writes only the 64-texel runs of `out` that meet their
 * row's span (box == NULL: everything), th_map_fold only the texels inside it; the rest of `out` stays as it was.  No host
 * synchronisation: boxes and spans live on the device. */
int th_map_box(th_ctx* ctx, const float* verts_a, int na, const float* verts_b, int nb, const float* cams, int V,
               const float* scale_xy, int H, int W, float reach, int32_t* box_out, th_stream stream);
int th_upsample_concat_split_box(th_ctx* ctx, const float* img, const float* lat0, const float* lat1, const float* lat2,
                                 const int32_t* dims_host, int V, int H, int W, float* out, const int32_t* box,
                                 th_stream stream);
/* Demand-driven map (round 5).  Behind a th_render_prepass the texels a frame reads are known exactly: the four bilinear
 * corners, in every view, of the prepass's valid samples (get_pixel_aligned_feature, if_clight_renderer.py:210-269 behind the
 * hull mask :440-444) and of the painted input vertices (:168-172).  th_render_predemand (below, with the frame-level entry
 * points) marks them in a demand buffer of th_map_demand_bytes(V, H, W) bytes -- on the device, no host synchronisation --;
 * th_upsample_concat_split_demand writes only those texels of `out` (inside `box`, which may be NULL) and th_map_fold_demand
 * evaluates the folded layers only at the samples' texels (a compacted list: full tiles wherever the texels lie).  A rank of an
 * N-rank job reads about 1 / N of what the whole frame reads, so the map write and the fold shrink with the ray shard.  W must
 * be a multiple of 64.  Frames built this way carry the buffer in th_map_source.demand: a frame-level call whose sample list is
 * not the one the buffer was made from writes the rest of the map first (like the other premises of a cropped map). */
size_t th_map_demand_bytes(int V, int H, int W);
int th_upsample_concat_split_demand(th_ctx* ctx, const float* img, const float* lat0, const float* lat1, const float* lat2,
                                    const int32_t* dims_host, int V, int H, int W, float* out, const int32_t* box,
                                    const void* demand, th_stream stream);
/* paint_neural_human + can_body_grouping without materialising holder_feat_map: reduction_layer
 * (1x1 conv C->out_f, encoder.py:85,146) commutes with the bilinear sampling at :168-172, so the C-channel
 * channels-last map is sampled at the projected vertices and the layer is applied to those V*n_verts
 * rows; then the vizmap zeroing (:181-182) and the cluster mean (:356-371).  tokens_out [V,N_c,out_f].
 * C = 384 (full map) or 260 (compact map: `color_lift` = upsample_color is then required and is folded into
 * the 384-input reduction layer on the fly); color_lift may be NULL for C = 384. */
size_t th_paint_group_nhwc_workspace_bytes(int V, int n_verts, int C, int out_f);
int th_paint_group_nhwc(th_ctx* ctx, const float* map_nhwc, int V, int H, int W, int C,
                        const float* verts_world, int n_verts, const float* cams, const float* scale_xy,
                        const uint8_t* vizmap, const th_linear* reduction, const th_linear* color_lift,
                        const int32_t* csr_offsets,
                        const int32_t* csr_members, int n_clusters, float* tokens_out, void* workspace,
                        size_t workspace_bytes, th_stream stream);

/* K11 -- train-mode BatchNorm2d (+ residual) (+ ReLU), NCHW fp32: the elementwise tail of every ResNet stage of
 * SpatialEncoder.forward (encoder.py:114-126; torchvision BasicBlock: bn -> relu, bn -> += identity -> relu) with
 * the network in train() as run.py:29 leaves it: batch statistics (biased variance), running statistics updated with
 * `momentum` (unbiased variance), exactly F.batch_norm(training=True).  res may be NULL; running_* may be NULL (no
 * update); gamma/beta may be NULL (1 / 0).  y may alias x.  workspace: th_bn_workspace_bytes(N, C, H*W). */
/* K12 -- the bias-free convolutions of the ResNet18 stem (encoder.py:114-126: 7x7/2 3->64, 3x3 64->64, 3x3/2 64->128,
 * 3x3 128->128, 1x1/2 64->128; padding KS/2), NCHW fp32 in and out, as implicit GEMMs on v_mfma_f32_32x32x16_f16 with fp16 hi/lo split
 * operands (fp32-class accuracy).  th_conv_pack turns a [COUT,CIN,KS,KS] weight into the per-lane fragment image
 * (th_conv_pack_bytes bytes, device) and returns the power-of-two output scale to pass to th_conv2d; re-pack when the
 * weight changes.  th_conv2d_supported tells whether a shape is built. */
size_t th_conv_pack_bytes(int cout, int cin, int ks);
int th_conv_pack(th_ctx* ctx, const float* w, int cout, int cin, int ks, void* packed, size_t packed_bytes,
                 float* inv_scale_out, th_stream stream);
int th_conv2d_supported(int cin, int cout, int ks, int stride);
int th_conv2d(th_ctx* ctx, const float* x, int N, int cin, int H, int W, const void* packed, float inv_scale, int cout,
              int ks, int stride, float* y, th_stream stream);

/* nn.MaxPool2d(3, 2, 1) of the stem on [planes, H, W] fp32 -> [planes, (H-1)/2+1, (W-1)/2+1] */
int th_maxpool3x3s2(th_ctx* ctx, const float* x, int planes, int H, int W, float* y, th_stream stream);

size_t th_bn_workspace_bytes(int N, int C, int HW);
int th_bn_act(th_ctx* ctx, const float* x, const float* residual, int N, int C, int HW, const float* gamma,
              const float* beta, float eps, float momentum, float* running_mean, float* running_var, int relu,
              float* y, void* workspace, size_t workspace_bytes, th_stream stream);
/* The same sites with the module in eval() mode (the reference's Trainer.val, lib/train/trainers/trainer.py:131-150, runs
 * network.eval()): y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta [+ residual] [relu]; ONE launch, no
 * statistics pass, nothing updated. */
int th_bn_act_eval(th_ctx* ctx, const float* x, const float* residual, int N, int C, int HW, const float* gamma,
                   const float* beta, float eps, const float* running_mean, const float* running_var, int relu,
                   float* y, th_stream stream);
/* ABI 12 -- a convolution followed by a train-mode BatchNorm (every conv -> bn site of encoder.py:114-126) in TWO launches
 * instead of three: th_conv2d_stats is th_conv2d whose epilogue also leaves, per output channel, the sum and the sum of
 * squares of the values it stored as th_conv2d_stats_partials(...) float2 partials (`stats`: [cout][partials] float2, device,
 * 8-byte aligned); th_bn_act_stats is th_bn_act reading those instead of running its own statistics pass over y (the
 * partials of a channel are added in float64 in a fixed order: deterministic).  Same results as th_conv2d + th_bn_act up
 * to the rounding of the statistics (fp32 partial sums over <= 128 pixels). */
int th_conv2d_stats_partials(int N, int cin, int H, int W, int cout, int ks, int stride);
int th_conv2d_stats(th_ctx* ctx, const float* x, int N, int cin, int H, int W, const void* packed, float inv_scale,
                    int cout, int ks, int stride, float* y, void* stats, size_t stats_bytes, th_stream stream);
int th_bn_act_stats(th_ctx* ctx, const float* x, const float* residual, int N, int C, int HW, const void* stats,
                    int n_partials, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int relu, float* y, th_stream stream);

/* ---- K3: TransHE (ViT-tiny) ---------------------------------------------- */
/* VisionTransformer.forward, vision_transformer.py:371-383.  x [V,N,dim]
 * tokens, pe [V,N,dim] sin-cos table (host-built, see vision_transformer.py),
 * out [V,N,dim]. */
size_t th_vit_workspace_bytes(int V, int N, int dim, int heads);
int th_vit_forward(th_ctx* ctx, const float* x, const float* pe, int V, int N, float* out,
                   void* workspace, size_t workspace_bytes, th_stream stream);

/* ---- K4: DPaRF encoding --------------------------------------------------- */
/* Network.get_human_representation, cross_transformer.py:158-205.
 * pts_smpl [P,3]; centres [N_c,3]; rot [N_c,9]; tokens [V,N_c,192];
 * sel: int32[P] indices into pts or NULL (identity).
 * out: [P, V, 256] row-major (255 features + one zero pad column). */
int th_dparf_encode(th_ctx* ctx, const float* pts_smpl, const int32_t* sel, int P, const float* centres,
                    const float* rot, const float* tokens, int V, int n_clusters, float* out,
                    th_stream stream);

/* ---- K5: pixel-aligned feature gather -------------------------------------- */
/* get_pixel_aligned_feature :210-269 on a channels-last map
 * pixel_map_nhwc [V,H,W,C]; pts_world [P,3]; out [P,V,ldo] (ldo >= C floats per row, a multiple of 4;
 * columns C..ldo-1 are written as zeros). */
int th_nchw_to_nhwc(th_ctx* ctx, const float* src, int V, int C, int H, int W, float* dst, th_stream stream);
int th_pixel_gather(th_ctx* ctx, const float* pixel_map_nhwc, int V, int C, int H, int W,
                    const float* pts_world, const int32_t* sel, int P, const float* cams,
                    const float* scale_xy, float* out, int ldo, th_stream stream);
/* The form of the same gather that the frame-level entry points run (ABI 6): the split compact map (th_upsample_concat_split:
 * [V,H,W,256] latents followed by a [V,H,W,4] r g b 0 plane) in, TH_ROWS_SPLIT rows out -- per (sample, view) 272 / 8 groups
 * of [8 fp16 hi halves | 8 fp16 lo halves], x = hi + lo, the layout the fused MLP kernel stages by LDS-DMA; out_rows holds
 * P * V * ldo * 4 bytes.  (Exposed so that the split-row kernel can be tested on its own against th_pixel_gather.) */
int th_pixel_gather_split(th_ctx* ctx, const float* map_split, int V, int H, int W, const float* pts_world,
                          const int32_t* sel, int P, const float* cams, const float* scale_xy, void* out_rows,
                          int ldo, th_stream stream);

/* The layers that read the pixel-aligned features -- alpha_res_0, rgb_res_0 (under the folded view_fc) and rgb_res_1,
 * cross_transformer.py:316, :334, :346 -- applied to the TEXELS of a TH_MAP_SPLIT map instead of to every (sample, view) row:
 * they are linear and act directly on grid_sample's bilinear blends (if_clight_renderer.py:255-265), so they commute with the
 * sampling.  fold [2][V,H,W,256] fp32: plane 0 = alpha_res_0' texel, plane 1 = [Wa rgb_res_0' (128) | rgb_res_1' (128)] texel
 * (no biases; the colour lift is inside, th_mlp_weights.upsample_color).  box: th_map_box's output (boxes + row spans) -> only
 * texels inside each row's span are computed, or NULL: the whole map.  Once per frame, after th_set_mlp_weights; th_frame.map_fold. */
int th_map_fold(th_ctx* ctx, const float* map_split, int V, int H, int W, const int32_t* box, float* fold, th_stream stream);
int th_map_fold_demand(th_ctx* ctx, const float* map_split, int V, int H, int W, const void* demand, float* fold, th_stream stream);

/* K5t, the producer of the texel hand-over (th_set_tex_rows; k_pixtex.hip), on its own -- exposed for tests.  Samples are taken
 * in tiles of 32 consecutive entries (of `sel`, or of the points when sel is NULL); out (th_pixel_texlist_bytes(V, P) bytes,
 * T = ceil(P / 32) tiles) receives
 *   lists   [T][4 passes][128] uint32: word 0 = U | npass << 16 (U <= 103 texel rows of this pass; npass = 1, 2 or 4: sample s of
 *           the tile belongs to pass s / (32 / npass)), words 8 .. 8 + U - 1 = view * H * W + y * W + x of the pass's distinct
 *           corner texels;
 *   records [T][V][32][8]: {w00, w01, w10, w11} (float bits, grid_sample's bilinear weights, if_clight_renderer.py:210-269) and
 *           {o00, o01, o10, o11} = 1040 * (number of the corner's texel in its pass's list).
 * (map_split is not read: the lists depend on the cameras only.) */
size_t th_pixel_texlist_bytes(int V, int P);
int th_pixel_texlist(th_ctx* ctx, const float* map_split, int V, int H, int W, const float* pts_world, const int32_t* sel,
                     int P, const float* cams, const float* scale_xy, void* out, size_t out_bytes, th_stream stream);

/* ---- K6: per-point multi-view MLP ------------------------------------------ */
/* Network.forward, cross_transformer.py:207-353, on already-gathered inputs.
 * pixel_feat [V,384,P] (the reference's channel-major layout); viewdir [P,27];
 * pts_smpl [P,3]; mask uint8[P] or NULL.  raw_out [P,4] (rgb logits, sigma).
 * With a mask: progressive RGB (sigma>0 only) and zero rows where masked out;
 * without: RGB everywhere (MLP_forward_ori :280-289). */
size_t th_network_workspace_bytes(int V, int P);
int th_network_forward(th_ctx* ctx, const float* pixel_feat, const float* viewdir, const float* pts_smpl,
                       const uint8_t* mask, int P, const float* centres, const float* rot,
                       const float* tokens, int V, int n_clusters, float* raw_out, void* workspace,
                       size_t workspace_bytes, th_stream stream);

/* ---- K7: alpha compositing --------------------------------------------------- */
/* raw2outputs, nerf_net_utils.py:14-59.  raw [R,S,4]; z [R,S] or NULL (then
 * recomputed from near/far/t_vals); ray_d [R,3].  Outputs rgb [R,3], acc [R],
 * depth [R]; weights_out [R,S] or NULL. */
int th_composite(th_ctx* ctx, const float* raw, const float* z, const th_points* rays, int white_bkgd,
                 float* rgb, float* acc, float* depth, float* weights_out, th_stream stream);

/* ---- K9: ray generation for a target camera --------------------------- */
/* lib/utils/if_nerf/if_nerf_data_utils.py:11-30 (get_rays) + :65-97 (get_near_far) as the test split of
 * sample_ray_h36m uses them (:271-283): one ray per pixel of an H x W camera (K [3,3], R [3,3], T [3] float32,
 * HOST pointers -- 84 bytes of parameters), intersected with the box `bounds_host` = {min xyz, max xyz} of the
 * posed body (padded by 0.01, float64 arithmetic like the reference).  Dense outputs over the H*W pixels in
 * row-major order: ray_o/ray_d [H*W,3] (ray_d carries the reference's |d| < 1e-5 -> 1e-5 clamp), near/far [H*W]
 * (0 where the ray misses), mask_at_box uint8[H*W] (1: exactly two faces hit).  Compact with the mask to get the
 * reference's ray list. */
int th_gen_rays(th_ctx* ctx, const float* K_host, const float* R_host, const float* T_host,
                const float* bounds_host, int H, int W, float* ray_o, float* ray_d, float* near_out,
                float* far_out, uint8_t* mask_at_box, th_stream stream);

/* lib/utils/if_nerf/if_nerf_data_utils.py:49-62 (get_bound_2d_mask): the projected body box as an H x W uint8 mask
 * (used by sample_ray_h36m :211-213 to restrict the foreground mask).  corners_xy_host = the eight box corners
 * (get_bound_corners order, :33-46) projected with base_utils.project and rounded with np.round(..).astype(int)
 * on the HOST (8 points; transhuman_amd/hip.py::bound_2d_mask keeps the reference's numpy expressions); the kernel
 * rasterises the six faces with cv2.fillPoly's scan conversion (even-odd spans + 8-connected edges), vertex lists
 * as in :55-60.  OpenCV is third-party and absent: parity unpinned against cv2 itself, bit-exact against the
 * restatement in oracle/th_oracle.py. */
int th_bound_mask(th_ctx* ctx, const int32_t* corners_xy_host /* [8][2] */, int H, int W, uint8_t* mask,
                  th_stream stream);

/* ---- K13: marching cubes over the sigma cube ------------------------------ */
/* lib/networks/renderer/if_mesh_renderer.py:99-109: `mcubes.marching_cubes(np.pad(cube, 10), cfg.mesh_th)` and the
 * index -> world transform `vertices * voxel_size + (can_bounds[0] - 10 voxel_size)`.  PyMCubes is third-party and
 * absent (parity unpinned against mcubes itself): the published table-driven algorithm is restated in PyMCubes'
 * conventions (corner / edge numbering of the Bourke table, corner "inside" <=> value <= iso, one shared vertex per
 * cut edge at the float64 linear interpolation, table order of triangles) -- see csrc/k_mcubes.hip.
 * cube: [X][Y][Z] fp32 on the device (already padded by the caller).  Usage:
 *   th_marching_cubes_count  passes 1-2 (classification + prefix sums into `workspace`), waits for the stream and
 *                            returns {n_vertices, n_triangles} in counts_host;
 *   th_marching_cubes_emit   writes vertices (float64 [n_vertices,3] = index * scale + origin) and triangles
 *                            (int32 [n_triangles,3], indices into the vertex array) for the grid slab
 *                            x0 <= x < x1 of points / cells (0, X for everything); slabs of one count pass write
 *                            disjoint contiguous ranges of the same arrays (multi-GPU: one slab per rank);
 *   th_marching_cubes_range  {vertex, triangle} prefix at the first point of plane x (x = X: the totals): the
 *                            range a slab [x0, x1) fills is [range(x0), range(x1)).
 * Vertex order: owning grid point row-major, +x / +y / +z edge; triangle order: cell row-major, table order. */
size_t th_marching_cubes_workspace_bytes(int X, int Y, int Z);
int th_marching_cubes_count(th_ctx* ctx, const float* cube, int X, int Y, int Z, float iso, void* workspace,
                            size_t workspace_bytes, int64_t* counts_host /* [2] */, th_stream stream);
int th_marching_cubes_range(th_ctx* ctx, const void* workspace, int X, int Y, int Z, int x,
                            int64_t* prefix_host /* [2] */, th_stream stream);
int th_marching_cubes_emit(th_ctx* ctx, const float* cube, int X, int Y, int Z, float iso, const void* workspace,
                           int x0, int x1, const double* scale_host /* [3] */, const double* origin_host /* [3] */,
                           double* verts, int32_t* tris, th_stream stream);

/* ---- K10: SMPL linear blend skinning ------------------------------------ */
/* SMPL._call, lib/utils/SMPL.py:114-186, float64 like the reference.  Model arrays (DEVICE pointers, the fields
 * the reference reads from the SMPL pickle, :83-89): v_template [nv,3], shapedirs [nv,3,10], posedirs [nv,3,207],
 * J_regressor [24,nv] dense, weights [nv,24], parent int32[24] (parent[0] = -1).
 * Pose: either pose_aa = 72 float32 axis-angle values (cv2.Rodrigues form, :135-139) or rot = [24,3,3] float32
 * rotation matrices (:131-132); exactly one of the two is non-NULL.  beta: 10 float64 (device).
 * Outputs (device, float64): verts [nv,3] (SMPL-space posed vertices, :186), joints [24,3] (:163),
 * T [nv,4,4] (= the path's blend_mtx, :176). */
typedef struct {
    const double* v_template;
    const double* shapedirs;
    const double* posedirs;
    const double* J_regressor;
    const double* weights;
    const int32_t* parent;
    int n_verts;
} th_smpl_model;
size_t th_smpl_workspace_bytes(int n_verts);
int th_smpl_lbs(th_ctx* ctx, const th_smpl_model* model, const float* pose_aa, const float* rot, const double* beta,
                double* verts, double* joints, double* T, void* workspace, size_t workspace_bytes,
                th_stream stream);

/* view-direction embedding, if_clight_renderer.py:525-526 + embedder.py:9-35:
 * ray_d [R,3] -> [R, 3 + 6*view_res] */
int th_view_embed(th_ctx* ctx, const float* ray_d, int R, int view_res, float* out, th_stream stream);

/* ---- frame-level entry points -------------------------------------------------- */
/* What a cropped TH_MAP_SPLIT map was made from (host struct, device pointers; everything must stay alive as long as the
 * frame is used).  The frame-level entry points write the rest of the map themselves -- on their stream, in front of the
 * gather -- when a call leaves the crop's premise: the un-masked branch (R' <= small_frame_rays shades EVERY sample of
 * the hit rays), hull_thresh < 0 (no hull test) or hull_thresh > reach. */
typedef struct {
    const int32_t* box;            /* device [V][4] + [V][H][2], th_map_box    */
    float          reach;          /* the reach the box was computed with      */
    const float*   img;            /* arguments of th_upsample_concat_split    */
    const float*   lat0;
    const float*   lat1;
    const float*   lat2;
    int32_t        dims[6];
    const void*    demand;         /* NULL, or the th_render_predemand buffer the map (and its fold) was written for: only the
                                      texels the valid samples of THAT th_render_prepass (and the painted vertices) read exist */
} th_map_source;

/* Per-frame constants produced by th_paint_group / th_vit_forward / the
 * segment means, consumed by the per-sample stage. */
typedef struct {
    const float* verts_world;      /* [n_verts,3] tar_smpl_vertice            */
    int          n_verts;
    const float* Rh;               /* [9]                                     */
    const float* Th;               /* [3]                                     */
    const float* cams;             /* [V,21]                                  */
    const float* scale_xy;         /* [2]                                     */
    const float* pixel_map_nhwc;   /* [V,H,W,map_channels]                    */
    int          V, H, W;
    int          map_channels;     /* TH_MAP_FULL (384), TH_MAP_COMPACT (260) or TH_MAP_SPLIT (256 + 4; the last two need
                                      upsample_color weights) */
    const float* tokens;           /* [V,N_c,192] ViT output                  */
    const float* centres;          /* [N_c,3]                                 */
    const float* rot;              /* [N_c,9]                                 */
    int          n_clusters;
    float        hull_thresh;      /* 0.1                                     */
    int          small_frame_rays; /* 2400: R' <= this -> un-masked branch    */
    const th_map_source* map_source; /* NULL: pixel_map_nhwc is complete; else it is cropped to map_source->box */
    const float* map_fold;         /* NULL, or th_map_fold's output for pixel_map_nhwc (TH_MAP_SPLIT, same crop): [2][V,H,W,256];
                                      with it (and th_set_tex_rows 1) the per-sample stage takes the texel hand-over */
} th_frame;

/* Renderer.render_fast :429-484 incl. _render/batchify_rays/raw2outputs for a
 * range of rays (the unit that is sharded across GPUs).  Outputs are dense
 * over the R rays (zeros for rays that miss the hull).
 * stats_host (optional, int64[4]): hit rays, valid samples, range-guard snapshot slot (th_range_read), mode.
 *
 * Two caller-supplied buffers (ABI 6; one 25 GB worst-case workspace per frame in flight before):
 *  - `workspace` (th_render_workspace_bytes: ~25 B per SAMPLE of the ray list -- hull mask, sample list, dense raw, counts,
 *    view embeddings, the per-frame token table): one per frame in flight, it is what th_render_prepass writes;
 *  - `shade_pool` (th_shade_pool_bytes: ~0.5 KB per VALID sample on the fused path with the texel hand-over -- K5t's texel
 *    lists and row records (get_pixel_aligned_feature :210-269), the neighbour records and positional encodings of K4;
 *    ~3.6 KB with th_set_tex_rows(ctx, 0), when K5 writes the pixel-feature rows and K6 reads them back): ONE per context, shared by every workspace (the per-sample stage of a context runs
 *    on one stream at a time).  It is sized from the frame's valid-sample count, which th_render_prepass_wait puts on
 *    the host before the stage is queued; without a prepass size it for n_valid = R * S (then only the chunk buffers of
 *    th_set_chunk_samples samples are needed: the count bounds the chunk, not the pool). */
size_t th_render_workspace_bytes(const th_frame* f, int R, int S);
/* `f` needs V, H, W and map_channels; the result depends on the context's MLP / token-gather / texel hand-over modes (ask
 * again after th_set_mlp_mode / th_set_tok_gather / th_set_tex_rows).  with_pregather = 1: large enough for th_render_pregather + th_render_rays. */
size_t th_shade_pool_bytes(th_ctx* ctx, const th_frame* f, long long n_valid, int with_pregather);
int th_render_rays(th_ctx* ctx, const th_frame* f, const th_points* rays, float* rgb, float* acc,
                   float* depth, int white_bkgd, void* workspace, size_t workspace_bytes,
                   void* shade_pool, size_t shade_pool_bytes, int64_t* stats_host, th_stream stream);

/* Optional: run the ray-only front of th_render_rays (sample placement + hull mask :440-444, the R' <= 2400
 * rule :551, compaction, view-direction embedding) ahead of time.  It needs only the rays and, of `f`, the
 * fields verts_world / n_verts / V / hull_thresh / small_frame_rays -- so it can be queued BEFORE the per-frame
 * constants (encoder, tokens) are produced; its sample count then reaches the host while that work runs and the
 * following th_render_rays (same ctx, same workspace, same ray arrays) neither repeats the stage nor stalls the
 * queue on the read-back.  `stream` may differ from the stream of th_render_rays (which waits on an event): on a
 * side stream the hull pass fills the chip while the latency-bound encoder / TransHE launches run.  The caller
 * orders the prepass after the previous use of the workspace.  Up to 4 prepasses may be pending at once, each in
 * its own workspace (a frame pipeline keeps the ray-only stage of the next frames in flight); th_render_rays
 * consumes the one queued for ITS workspace.  Results are identical with or without it. */
int th_render_prepass(th_ctx* ctx, const th_frame* f, const th_points* rays, void* workspace,
                      size_t workspace_bytes, th_stream stream);
/* Optional second ahead-of-time stage, behind a th_render_prepass of the same workspace / ray arrays: the pixel-aligned
 * feature rows (get_pixel_aligned_feature, :210-269) and the 7-neighbour records of the human representation
 * (cross_transformer.py:151-205) of the first chunks of valid samples.  Neither needs the TransHE tokens on the
 * th_set_tok_gather(ctx, 1) path: `f` must be complete EXCEPT f->tokens, so a caller can run TransHE on another stream
 * beside this (texture-path-bound) stage instead of in front of it; the following th_render_rays (same workspace, same
 * map / token centres, complete frame) then only runs the fused MLP for those chunks.  Waits for the prepass's sample
 * count on the host.  A no-op (return 0) without a matching prepass or off the fused neighbour-record path; results are
 * identical with or without it. */
int th_render_pregather(th_ctx* ctx, const th_frame* f, const th_points* rays, void* workspace,
                        size_t workspace_bytes, void* shade_pool, size_t shade_pool_bytes, th_stream stream);
/* ABI 8.  Optional, behind a th_render_prepass of the same workspace / ray arrays and before its th_render_pregather: builds
 * the candidate grid of the 7-neighbour search (cross_transformer.py:151-160; exact pruning) for f->centres
 * into the workspace on `stream` -- a frame pipeline calls it on the stream that produced the centres (th_paint_group),
 * long before the pre-gather stage, which then starts with the neighbour search itself.  The caller orders `stream` before
 * the pre-gather stage (as for the prepass).  A no-op without a matching prepass; results are identical. */
int th_render_pregrid(th_ctx* ctx, const th_frame* f, const th_points* rays, void* workspace, size_t workspace_bytes,
                      th_stream stream);
/* ABI 9.  Optional, behind a th_render_prepass of the same workspace / ray arrays: marks, on `stream` (which is first ordered
 * behind the prepass), the map texels the prepass's valid samples read in the V views of f (f->cams, f->scale_xy, f->V, f->H,
 * f->W; the other fields as for th_render_prepass) and those the n_paint vertices of verts_paint read (NULL / 0: none -- e.g. a
 * rank that receives this frame's tokens) into `demand` (th_map_demand_bytes).  See th_upsample_concat_split_demand.  A frame
 * whose th_map_source.demand is this buffer is complete for exactly this prepass; returns 1 (and leaves `demand` untouched) without
 * a matching prepass. */
int th_render_predemand(th_ctx* ctx, const th_frame* f, const th_points* rays, void* workspace, size_t workspace_bytes,
                        const float* verts_paint, int n_paint, void* demand, size_t demand_bytes, th_stream stream);
/* ABI 8.  th_render_pregather for a frame pipeline that queues frame i+1's pre-gather stage right behind frame i's
 * th_render_rays on the same stream and shading pool: the neighbour-record producer (on the context's second stream) is
 * ordered behind the PER-SAMPLE stage of that th_render_rays -- the last user of the pool regions it writes -- instead of
 * behind everything queued on `stream` since, so frame i's compositing and whatever the caller queued after it (image
 * assembly, an all-gather) run beside the producer instead of in front of it.  The caller promises that everything the
 * stage reads -- the prepass of this workspace, `f` except f->tokens, the ray arrays -- was complete on `stream` BEFORE
 * that th_render_rays was queued (e.g. the stream waited on the event of the front that produced them); a non-NULL
 * f->tokens is taken as complete on `stream` now, and the per-frame token table is then queued here as well.  Without a
 * th_render_rays on the same stream and pool since the last pre-gather it behaves exactly as th_render_pregather;
 * `stream` has waited for the whole stage when the call returns, as there.  Results are identical. */
int th_render_pregather_early(th_ctx* ctx, const th_frame* f, const th_points* rays, void* workspace,
                              size_t workspace_bytes, void* shade_pool, size_t shade_pool_bytes, th_stream stream);
/* Waits (host) for the counts of the prepass pending in `workspace`: counts_host[0] = hit rays, [1] = 1 when the
 * R' <= small_frame_rays rule fired (un-masked branch), [2] = valid samples -- what th_shade_pool_bytes wants.
 * Returns 1 (and leaves counts_host alone) when no prepass is pending for that workspace. */
int th_render_prepass_wait(th_ctx* ctx, const void* workspace, int64_t* counts_host /* [3] */);
/* Drops a pending prepass (a caller that abandons the frame it was queued for must not let a later
 * th_render_rays that happens to reuse the same buffers pick it up). */
int th_render_prepass_cancel(th_ctx* ctx);                          /* all pending prepasses */
int th_render_prepass_drop(th_ctx* ctx, const void* workspace);     /* the one queued for this workspace */

/* if_mesh_renderer.Renderer.render :46-100 up to `cube`: sigma_raw per grid
 * point (0 outside the hull).  pts [P,3] world space. */
size_t th_sigma_grid_workspace_bytes(const th_frame* f, int P);
int th_eval_sigma_grid(th_ctx* ctx, const th_frame* f, const float* pts, int P, float* sigma_out,
                       void* workspace, size_t workspace_bytes, void* shade_pool, size_t shade_pool_bytes,
                       int64_t* stats_host, th_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* TRANSHUMAN_HIP_H */
