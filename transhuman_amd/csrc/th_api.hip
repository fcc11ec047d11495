// C-ABI of libtranshuman_hip.so: context, weight packing, per-kernel entry
// points and the frame-level orchestration (th_render_rays /
// th_eval_sigma_grid / th_network_forward).  See include/transhuman_hip.h.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "th_internal.h"

static thread_local std::string g_err;
void th_set_error(const std::string& msg) { g_err = msg; }

// ---------------------------------------------------------------------------
// stage profiling: HIP events recorded on the stream the kernels are launched
// on (bench.py reads them back for the live roofline numbers)
// ---------------------------------------------------------------------------
struct ThProf {
    bool on = false;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    struct Span { int phase; hipEvent_t a, b; };
    std::vector<Span> spans;
    hipEvent_t get() {
        if (used == pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
};
struct ProfScope {
    ThProf* p; int phase; hipStream_t s; hipEvent_t a = nullptr;
    ProfScope(ThProf* p_, int phase_, hipStream_t s_) : p(p_), phase(phase_), s(s_) {
        if (p && p->on) {
            // (a stream under capture -- the callers that replay the stem / TransHE as hipGraphs -- is not timed: an event node
            // recorded in a graph has no timestamp of its own)
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return;
            a = p->get();
            if (a) (void)hipEventRecord(a, s);
        }
    }
    void close() {
        if (a) { hipEvent_t b = p->get(); if (b) { (void)hipEventRecord(b, s); p->spans.push_back({phase, a, b}); } }
        a = nullptr;
    }
    ~ProfScope() { close(); }
};
static ThProf* prof_of(th_ctx* c);
// one wave: shader-clock ticks (s_memtime) against the constant 100 MHz counter (s_memrealtime) over a short dependent
// VALU chain -> out[0] = ticks, out[1] = 10 ns units.  The clock domain is chip-wide, so a probe queued right behind a
// kernel reads the frequency the DVFS governor holds under that kernel's load.
__global__ void clock_probe_kernel(long long* __restrict__ out, int iters) {
    float x = 1.0f + threadIdx.x * 1e-3f;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) x = fmaf(x, 0.999f, 1e-3f);
    const long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (x == 12345.678f) out[2] = 1;
}


extern "C" {

int th_abi_version(void) { return TH_ABI_VERSION; }
const char* th_last_error(void) { return g_err.c_str(); }
size_t th_sizeof(const char* type_name) {
    if (type_name == nullptr) return 0;
#define TH_SZ(T) if (strcmp(type_name, #T) == 0) return sizeof(T)
    TH_SZ(th_points); TH_SZ(th_frame); TH_SZ(th_map_source); TH_SZ(th_linear); TH_SZ(th_mlp_weights); TH_SZ(th_vit_block); TH_SZ(th_smpl_model);
#undef TH_SZ
    return 0;
}

int th_ctx_create(int device, th_ctx** out) {
    TH_REQUIRE(out != nullptr, "null out");
    int count = 0;
    TH_HIP(hipGetDeviceCount(&count));
    TH_REQUIRE(device >= 0 && device < count, "no such device");
    TH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    TH_HIP(hipGetDeviceProperties(&prop, device));
    TH_REQUIRE(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0,
               std::string("libtranshuman_hip is built for gfx950 only, device is ") + prop.gcnArchName);
    th_ctx* c = new th_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    if (const char* e = getenv("TH_TOK_GATHER")) c->tok_gather = e[0] == '0' ? 0 : 1;
    if (const char* e = getenv("TH_ROWS_TEX")) c->tex_rows = e[0] == '0' ? 0 : 1;
    c->fused.waves = 8;                                      // (th_set_fused_waves)
    if (const char* e = getenv("TH_FUSED_WAVES")) c->fused.waves = e[0] == '4' ? 4 : 8;
    TH_HIP(hipHostMalloc((void**)&c->host_pinned, 64 * sizeof(int32_t), hipHostMallocDefault));
    TH_HIP(hipMalloc((void**)&c->range_dev, TH_RANGE_SLOTS * sizeof(unsigned int)));
    TH_HIP(hipMemset(c->range_dev, 0, TH_RANGE_SLOTS * sizeof(unsigned int)));
    TH_HIP(hipHostMalloc((void**)&c->range_host, th_ctx::kRangeSnaps * TH_RANGE_SLOTS * sizeof(unsigned int),
                         hipHostMallocDefault));
    memset(c->range_host, 0, th_ctx::kRangeSnaps * TH_RANGE_SLOTS * sizeof(unsigned int));
    *out = c;
    return 0;
}

void th_ctx_destroy(th_ctx* c) {
    if (!c) return;
    if (c->mlp_store) (void)hipFree(c->mlp_store);
    if (c->vit_store) (void)hipFree(c->vit_store);
    if (c->fused_store) (void)hipFree(c->fused_store);
    if (c->host_pinned) (void)hipHostFree(c->host_pinned);
    if (c->range_dev) (void)hipFree(c->range_dev);
    if (c->range_host) (void)hipHostFree(c->range_host);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->aux2) (void)hipStreamDestroy(c->aux2);
    if (c->aux2_join) (void)hipEventDestroy(c->aux2_join);
    if (c->aux_fork) (void)hipEventDestroy(c->aux_fork);
    if (c->aux_join) (void)hipEventDestroy(c->aux_join);
    if (c->after_shade) (void)hipEventDestroy(c->after_shade);
    for (auto& e : c->range_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& t : c->prepass)
        if (t.ev) (void)hipEventDestroy(t.ev);
    delete[] c->vit.blocks;
    if (c->prof) {
        ThProf* p = (ThProf*)c->prof;
        for (auto e : p->pool) (void)hipEventDestroy(e);
        delete p;
    }
    delete c;
}

int th_profile_enable(th_ctx* c, int on) {
    TH_REQUIRE(c, "null ctx");
    ThProf* p = prof_of(c);
    p->on = on != 0;
    p->used = 0;
    p->spans.clear();
    return 0;
}

int th_clock_probe(th_ctx* c, int64_t* out_dev, th_stream stream) {
    TH_REQUIRE(c && out_dev, "null argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)out_dev, 4096);
    TH_LAUNCH_CHECK();
    return 0;
}

int th_fused_cycles(th_ctx* c, int64_t* counters_dev) {
    TH_REQUIRE(c, "null ctx");
    c->fused.cycles_buf = (long long*)counters_dev;
    return 0;
}

int th_host_wait_read(th_ctx* c, double* ms_out) {
    TH_REQUIRE(c && ms_out, "null argument");
    *ms_out = c->host_wait_ms;
    c->host_wait_ms = 0.0;
    return 0;
}

int th_profile_read(th_ctx* c, double* ms_out, int64_t* count_out) {
    TH_REQUIRE(c && ms_out && count_out, "null argument");
    ThProf* p = prof_of(c);
    for (int i = 0; i < TH_PROF_PHASES; ++i) { ms_out[i] = 0.0; count_out[i] = 0; }
    for (auto& sp : p->spans) {
        TH_HIP(hipEventSynchronize(sp.b));
        float ms = 0.f;
        TH_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
        if (sp.phase >= 0 && sp.phase < TH_PROF_PHASES) { ms_out[sp.phase] += ms; count_out[sp.phase] += 1; }
    }
    p->used = 0;
    p->spans.clear();
    return 0;
}

// ---------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------
static bool lin_ok(const th_linear& l, int out_f, int in_f) {
    return l.w != nullptr && l.out_f == out_f && l.in_f == in_f;
}

int th_set_mlp_weights(th_ctx* c, const th_mlp_weights* w, th_stream stream) {
    TH_REQUIRE(c && w, "null argument");
    hipStream_t s = (hipStream_t)stream;
    // shapes of cross_transformer.py:96-126
    TH_REQUIRE(lin_ok(w->fc_0, 256, 255) && lin_ok(w->alpha_res_0, 256, 384) && lin_ok(w->key0, 128, 256) &&
                   lin_ok(w->val0, 256, 256) && lin_ok(w->key1, 128, 256) && lin_ok(w->val1, 256, 256) &&
                   lin_ok(w->fc_1, 256, 256) && lin_ok(w->fc_2, 256, 256) && lin_ok(w->fc_3, 256, 256) &&
                   lin_ok(w->alpha_fc, 1, 256) && lin_ok(w->feature_fc, 256, 256) &&
                   lin_ok(w->rgb_res_0, 256, 384) && lin_ok(w->view_fc, 128, 283) &&
                   lin_ok(w->rgb_res_1, 128, 384) && lin_ok(w->fc_4, 128, 128) && lin_ok(w->rgb_fc, 3, 128),
               "unexpected layer shape (expects the reference Network of cross_transformer.py:96-126)");
    struct Item { const th_linear* l; ThPacked* dst; int out_f, in_f; };
    ThMlpPacked& M = c->mlp;
    size_t total = 0;
    Item items[] = {{&w->fc_0, &M.fc_0, 256, 255},          {&w->alpha_res_0, &M.alpha_res_0, 256, 384},
                    {nullptr, &M.kv0, 384, 256},            {nullptr, &M.kv1, 384, 256},
                    {&w->fc_1, &M.fc_1, 256, 256},          {&w->fc_2, &M.fc_2, 256, 256},
                    {&w->fc_3, &M.fc_3, 256, 256},          {&w->feature_fc, &M.feature_fc, 256, 256},
                    {&w->rgb_res_0, &M.rgb_res_0, 256, 384}, {&w->view_fc, &M.view_fc, 128, 283},
                    {&w->rgb_res_1, &M.rgb_res_1, 128, 384}, {&w->fc_4, &M.fc_4, 128, 128}};
    for (auto& it : items) total += ThPacked::bytes(it.out_f, it.in_f);
    // colour-folded forms (optional): packed layers + fp32 fold scratch [N,260] + [N]
    const bool have_lift = w->upsample_color.w != nullptr;
    if (have_lift) TH_REQUIRE(lin_ok(w->upsample_color, 128, 3), "upsample_color must be a 3 -> 128 layer (encoder.py:95)");
    const size_t cpk_off = total;
    if (have_lift) total += 2 * ThPacked::bytes(256, 260) + ThPacked::bytes(128, 260);
    const size_t cfold_off = total;
    if (have_lift) total += 3 * th_align((size_t)(256 * 260 + 256) * sizeof(float));
    // token columns of fc_0 (no bias): packed layer + contiguous [256,192] copy
    const size_t tokpk_off = total;
    total += ThPacked::bytes(256, 192);
    const size_t tokw_off = total;
    total += th_align((size_t)256 * 192 * sizeof(float));
    size_t heads_off = total;
    total += th_align((256 + 1 + 3 * 128 + 3) * sizeof(float));
    size_t tmp_off = total;
    total += 2 * th_align((size_t)(384 * 256 + 384) * sizeof(float));   // concat scratch for kv0 / kv1
    if (c->mlp_store) { TH_HIP(hipFree(c->mlp_store)); c->mlp_store = nullptr; }
    TH_HIP(hipMalloc(&c->mlp_store, total));
    char* base = (char*)c->mlp_store;
    size_t off = 0;
    int kv_i = 0;
    for (auto& it : items) {
        th_linear src;
        if (it.l) src = *it.l;
        else {
            // [key_embed ; value_embed] stacked along out_f -> one 256 -> 384 layer
            const th_linear& k = kv_i == 0 ? w->key0 : w->key1;
            const th_linear& v = kv_i == 0 ? w->val0 : w->val1;
            float* tw = (float*)(base + tmp_off + kv_i * th_align((size_t)(384 * 256 + 384) * sizeof(float)));
            float* tb = tw + 384 * 256;
            TH_HIP(hipMemcpyAsync(tw, k.w, 128 * 256 * 4, hipMemcpyDeviceToDevice, s));
            TH_HIP(hipMemcpyAsync(tw + 128 * 256, v.w, 256 * 256 * 4, hipMemcpyDeviceToDevice, s));
            if (k.b) TH_HIP(hipMemcpyAsync(tb, k.b, 128 * 4, hipMemcpyDeviceToDevice, s));
            else TH_HIP(hipMemsetAsync(tb, 0, 128 * 4, s));
            if (v.b) TH_HIP(hipMemcpyAsync(tb + 128, v.b, 256 * 4, hipMemcpyDeviceToDevice, s));
            else TH_HIP(hipMemsetAsync(tb + 128, 0, 256 * 4, s));
            src.w = tw; src.b = tb; src.out_f = 384; src.in_f = 256;
            ++kv_i;
        }
        TH_TRY(th_pack_linear(src, base + off, it.dst, s));
        off += ThPacked::bytes(it.out_f, it.in_f);
    }
    float* hd = (float*)(base + heads_off);
    M.alpha_w = hd; M.alpha_b = hd + 256; M.rgb_w = hd + 257; M.rgb_b = hd + 257 + 384;
    TH_HIP(hipMemcpyAsync(M.alpha_w, w->alpha_fc.w, 256 * 4, hipMemcpyDeviceToDevice, s));
    TH_HIP(hipMemcpyAsync(M.rgb_w, w->rgb_fc.w, 384 * 4, hipMemcpyDeviceToDevice, s));
    if (w->alpha_fc.b) TH_HIP(hipMemcpyAsync(M.alpha_b, w->alpha_fc.b, 4, hipMemcpyDeviceToDevice, s));
    else TH_HIP(hipMemsetAsync(M.alpha_b, 0, 4, s));
    if (w->rgb_fc.b) TH_HIP(hipMemcpyAsync(M.rgb_b, w->rgb_fc.b, 12, hipMemcpyDeviceToDevice, s));
    else TH_HIP(hipMemsetAsync(M.rgb_b, 0, 12, s));
    {
        float* tw = (float*)(base + tokw_off);
        TH_HIP(hipMemcpy2DAsync(tw, 192 * 4, w->fc_0.w, 255 * 4, 192 * 4, 256, hipMemcpyDeviceToDevice, s));
        th_linear tok{tw, nullptr, 256, 192};
        TH_TRY(th_pack_linear(tok, base + tokpk_off, &M.fc_0tok, s));
    }
    th_linear folded[3] = {};
    M.compact_ready = false;
    if (have_lift) {
        const th_linear* src3[3] = {&w->alpha_res_0, &w->rgb_res_0, &w->rgb_res_1};
        ThPacked* dst3[3] = {&M.alpha_res_0c, &M.rgb_res_0c, &M.rgb_res_1c};
        size_t po = cpk_off;
        for (int i = 0; i < 3; ++i) {
            float* fw = (float*)(base + cfold_off + i * th_align((size_t)(256 * 260 + 256) * sizeof(float)));
            float* fb = fw + 256 * 260;
            const int N = src3[i]->out_f;
            TH_TRY(th_fold_color_launch(src3[i]->w, src3[i]->b, w->upsample_color.w, w->upsample_color.b, N, fw, fb, s));
            folded[i].w = fw; folded[i].b = fb; folded[i].out_f = N; folded[i].in_f = 260;
            TH_TRY(th_pack_linear(folded[i], base + po, dst3[i], s));
            po += ThPacked::bytes(N, 260);
        }
        M.compact_ready = true;
    }
    // new weights: the sticky maximum of the stem convolutions' input starts over (the encoder belongs to the same
    // parameter set; hip.py re-enables the HIP convolutions at the same moment)
    TH_HIP(hipMemsetAsync(c->range_dev + TH_RANGE_CONV, 0, sizeof(unsigned int), s));
    TH_HIP(hipStreamSynchronize(s));
    M.ready = true;
    // fused-kernel image (fp16 hi/lo split, per-wave fragment order)
    c->fused_ready = false;
    if (c->fused_store) { TH_HIP(hipFree(c->fused_store)); c->fused_store = nullptr; }
    TH_HIP(hipMalloc(&c->fused_store, th_fused_pack_bytes()));
    TH_TRY(th_fused_pack(w, have_lift ? folded : nullptr, c->fused_store, &c->fused, s));
    c->fused_ready = true;
    return 0;
}

// Snapshots rotate through kRangeSnaps pinned buffers.  The id handed out is a GENERATION (monotonic; buffer = id %
// kRangeSnaps): a caller that holds an id across more than kRangeSnaps - 1 later snapshots (a frame pipeline whose
// consumer issues other guarded calls between two frames) is told so (th_range_read returns 2) instead of silently
// reading another call's maxima.
int th_range_snapshot(th_ctx* c, th_stream stream) {
    TH_REQUIRE(c, "null ctx");
    hipStream_t s = (hipStream_t)stream;
    const int gen = c->range_gen_next;
    const int slot = gen % th_ctx::kRangeSnaps;
    c->range_gen_next = gen == 0x3fffffff ? 0 : gen + 1;
    if (!c->range_ev[slot]) TH_HIP(hipEventCreateWithFlags(&c->range_ev[slot], hipEventDisableTiming));
    TH_HIP(hipMemcpyAsync(c->range_host + slot * TH_RANGE_SLOTS, c->range_dev, TH_RANGE_SLOTS * sizeof(unsigned int),
                          hipMemcpyDeviceToHost, s));
    // per-frame slots only: the stem-convolution and TransHE slots are written from the stream that computes the NEXT
    // frame's constants (clearing them here could erase that frame's maxima) -- they are sticky until new weights arrive
    TH_HIP(hipMemsetAsync(c->range_dev, 0, TH_RANGE_CONV * sizeof(unsigned int), s));
    TH_HIP(hipEventRecord(c->range_ev[slot], s));
    c->range_gen[slot] = gen;
    c->range_last = gen;
    return gen;
}

int th_range_read(th_ctx* c, int id, uint32_t* out) {
    TH_REQUIRE(c && out, "null argument");
    TH_REQUIRE(id >= 0, "no such range snapshot");
    const int slot = id % th_ctx::kRangeSnaps;
    TH_REQUIRE(c->range_ev[slot], "no such range snapshot");
    if (c->range_gen[slot] != id) {
        th_set_error("th_range_read: snapshot " + std::to_string(id) + " has been overwritten by later snapshots");
        return 2;
    }
    {
        ThWaitClock wc(c);
        TH_HIP(hipEventSynchronize(c->range_ev[slot]));
    }
    memcpy(out, c->range_host + slot * TH_RANGE_SLOTS, TH_RANGE_SLOTS * sizeof(unsigned int));
    return 0;
}

int th_range_last_slot(th_ctx* c) { return c ? c->range_last : -1; }

int th_set_vit_mode(th_ctx* c, int mode) {
    TH_REQUIRE(c && (mode == 0 || mode == 1), "mode must be 0 (fp32 MFMA GEMMs) or 1 (fp16-split MFMA GEMMs)");
    c->vit_mode = mode;
    return 0;
}

int th_set_tok_gather(th_ctx* c, int on) {
    TH_REQUIRE(c && (on == 0 || on == 1), "th_set_tok_gather: 0 (blended rows from K4) or 1 (neighbour records, blend in the fused kernel)");
    c->tok_gather = on;
    return 0;
}

int th_set_tex_rows(th_ctx* c, int on) {
    TH_REQUIRE(c && (on == 0 || on == 1), "th_set_tex_rows: 0 (pixel-feature rows from K5) or 1 (texel lists, blend in the fused kernel)");
    c->tex_rows = on;
    return 0;
}

int th_set_mlp_mode(th_ctx* c, int mode) {
    TH_REQUIRE(c && (mode == 0 || mode == 1), "mode must be 0 (layer-by-layer fp32 MFMA) or 1 (fused fp16-split MFMA)");
    c->mlp_mode = mode;
    return 0;
}

int th_set_fused_waves(th_ctx* c, int waves) {
    TH_REQUIRE(c && (waves == 4 || waves == 8), "th_set_fused_waves: 4 (one wave per SIMD) or 8 (two per SIMD)");
    c->fused.waves = waves;
    return 0;
}

int th_set_vit_weights(th_ctx* c, int depth, int dim, int heads, const th_vit_block* blocks, const float* norm_w,
                       const float* norm_b, th_stream stream) {
    TH_REQUIRE(c && blocks && norm_w && norm_b, "null argument");
    TH_REQUIRE(depth > 0 && dim == heads * 64, "ViT must have head_dim 64");
    hipStream_t s = (hipStream_t)stream;
    size_t per = ThPacked::bytes(3 * dim, dim) + ThPacked::bytes(dim, dim) + ThPacked::bytes(4 * dim, dim) +
                 ThPacked::bytes(dim, 4 * dim) + th_align(4 * dim * sizeof(float)) +
                 ThPacked::bytes_h3(3 * dim, dim) + ThPacked::bytes_h3(dim, dim) + ThPacked::bytes_h3(4 * dim, dim) +
                 ThPacked::bytes_h3(dim, 4 * dim);
    size_t total = per * depth + th_align(2 * dim * sizeof(float));
    if (c->vit_store) { TH_HIP(hipFree(c->vit_store)); c->vit_store = nullptr; }
    TH_HIP(hipMalloc(&c->vit_store, total));
    delete[] c->vit.blocks;
    c->vit.blocks = new ThVitBlockPacked[depth];
    c->vit.depth = depth; c->vit.dim = dim; c->vit.heads = heads;
    char* base = (char*)c->vit_store;
    size_t off = 0;
    for (int b = 0; b < depth; ++b) {
        const th_vit_block& B = blocks[b];
        ThVitBlockPacked& P = c->vit.blocks[b];
        TH_REQUIRE(lin_ok(B.qkv, 3 * dim, dim) && lin_ok(B.proj, dim, dim) && lin_ok(B.fc1, 4 * dim, dim) &&
                       lin_ok(B.fc2, dim, 4 * dim) && B.ln1_w && B.ln1_b && B.ln2_w && B.ln2_b,
                   "unexpected ViT block shape");
        TH_TRY(th_pack_linear(B.qkv, base + off, &P.qkv, s)); off += ThPacked::bytes(3 * dim, dim);
        TH_TRY(th_pack_linear(B.proj, base + off, &P.proj, s)); off += ThPacked::bytes(dim, dim);
        TH_TRY(th_pack_linear(B.fc1, base + off, &P.fc1, s)); off += ThPacked::bytes(4 * dim, dim);
        TH_TRY(th_pack_linear(B.fc2, base + off, &P.fc2, s)); off += ThPacked::bytes(dim, 4 * dim);
        // fp16-split images of the same four layers (th_gemm_h3: the ViT's small-M GEMMs on the fp16 matrix pipe)
        TH_TRY(th_pack_linear_h3(B.qkv, base + off, &P.qkv, s)); off += ThPacked::bytes_h3(3 * dim, dim);
        TH_TRY(th_pack_linear_h3(B.proj, base + off, &P.proj, s)); off += ThPacked::bytes_h3(dim, dim);
        TH_TRY(th_pack_linear_h3(B.fc1, base + off, &P.fc1, s)); off += ThPacked::bytes_h3(4 * dim, dim);
        TH_TRY(th_pack_linear_h3(B.fc2, base + off, &P.fc2, s)); off += ThPacked::bytes_h3(dim, 4 * dim);
        float* ln = (float*)(base + off); off += th_align(4 * dim * sizeof(float));
        P.ln1_w = ln; P.ln1_b = ln + dim; P.ln2_w = ln + 2 * dim; P.ln2_b = ln + 3 * dim;
        TH_HIP(hipMemcpyAsync(P.ln1_w, B.ln1_w, dim * 4, hipMemcpyDeviceToDevice, s));
        TH_HIP(hipMemcpyAsync(P.ln1_b, B.ln1_b, dim * 4, hipMemcpyDeviceToDevice, s));
        TH_HIP(hipMemcpyAsync(P.ln2_w, B.ln2_w, dim * 4, hipMemcpyDeviceToDevice, s));
        TH_HIP(hipMemcpyAsync(P.ln2_b, B.ln2_b, dim * 4, hipMemcpyDeviceToDevice, s));
    }
    float* nf = (float*)(base + off);
    c->vit.norm_w = nf; c->vit.norm_b = nf + dim;
    TH_HIP(hipMemcpyAsync(c->vit.norm_w, norm_w, dim * 4, hipMemcpyDeviceToDevice, s));
    TH_HIP(hipMemcpyAsync(c->vit.norm_b, norm_b, dim * 4, hipMemcpyDeviceToDevice, s));
    TH_HIP(hipMemsetAsync(c->range_dev + TH_RANGE_VIT, 0, sizeof(unsigned int), s));   // new weights: sticky maximum cleared
    TH_HIP(hipStreamSynchronize(s));
    c->vit.ready = true;
    return 0;
}

// ---------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------
size_t th_linear_workspace_bytes(int out_f, int in_f) { return ThPacked::bytes(out_f, in_f); }

int th_linear_forward(th_ctx* c, const float* A, int lda, int M, const th_linear* lin, int act, float* C, int ldc,
                      void* ws, size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && A && lin && C && ws, "null argument");
    TH_REQUIRE(ws_bytes >= ThPacked::bytes(lin->out_f, lin->in_f), "workspace too small");
    TH_REQUIRE(act >= 0 && act <= 2, "act must be 0 (none), 1 (relu) or 2 (gelu)");
    ThPacked P;
    TH_TRY(th_pack_linear(*lin, ws, &P, (hipStream_t)stream));
    return th_gemm(A, lda, M, P, act, C, ldc, (hipStream_t)stream);
}

size_t th_hull_workspace_bytes(int n_verts) { return th_hull_ws(n_verts); }

int th_hull_mask(th_ctx* c, const th_points* p, const float* verts, int nv, float thresh, uint8_t* mask,
                 int32_t* ray_hit, void* ws, size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && p && verts && mask && ws, "null argument");
    TH_REQUIRE(p->pts || (p->ray_o && p->ray_d && p->near && p->far && p->t_vals && p->one_minus_t),
               "need explicit pts or a complete ray description");
    ThPointSrc ps = th_src(p);
    long long P = (long long)p->R * p->S;
    return th_hull_mask_launch(ps, P, verts, nv, thresh, mask, ray_hit, ws, ws_bytes, (hipStream_t)stream);
}

int th_paint_group(th_ctx* c, const float* map, int V, int C, int H, int W, const float* verts, int nv,
                   const float* cams, const float* scale, const uint8_t* viz, const int32_t* off, const int32_t* mem,
                   int nc, float* painted, float* tokens, th_stream stream) {
    TH_REQUIRE(c && map && verts && cams && scale && off && mem && tokens, "null argument");
    TH_REQUIRE(painted != nullptr, "painted_out scratch [V,n_verts,C] is required");
    hipStream_t s = (hipStream_t)stream;
    TH_TRY(th_paint_launch(map, V, C, H, W, verts, nv, cams, scale, viz, painted, s));
    return th_segmean_launch(painted, V, (long long)nv * C, C, off, mem, nc, tokens, s);
}

int th_segment_mean_f32(th_ctx* c, const float* src, int width, const int32_t* off, const int32_t* mem, int nc,
                        float* out, th_stream stream) {
    TH_REQUIRE(c && src && off && mem && out, "null argument");
    return th_segmean_launch(src, 1, 0, width, off, mem, nc, out, (hipStream_t)stream);
}

int th_segment_mean_rot_f64(th_ctx* c, const double* blend, const int32_t* off, const int32_t* mem, int nc,
                            float* rot, th_stream stream) {
    TH_REQUIRE(c && blend && off && mem && rot, "null argument");
    return th_segmean_rot_launch(blend, off, mem, nc, rot, (hipStream_t)stream);
}

int th_upsample_concat_split(th_ctx* c, const float* img, const float* lat0, const float* lat1, const float* lat2,
                             const int32_t* dims_host, int V, int H, int W, float* out, th_stream stream) {
    TH_REQUIRE(c && img && lat0 && lat1 && lat2 && dims_host && out, "null argument");
    return th_upsample_concat_launch(img, lat0, lat1, lat2, dims_host, V, H, W, nullptr, nullptr, out, (hipStream_t)stream, 1,
                                     nullptr);
}

int th_upsample_concat_split_box(th_ctx* c, const float* img, const float* lat0, const float* lat1, const float* lat2,
                                 const int32_t* dims_host, int V, int H, int W, float* out, const int32_t* box,
                                 th_stream stream) {
    TH_REQUIRE(c && img && lat0 && lat1 && lat2 && dims_host && out, "null argument");
    return th_upsample_concat_launch(img, lat0, lat1, lat2, dims_host, V, H, W, nullptr, nullptr, out, (hipStream_t)stream, 1,
                                     box);
}

size_t th_map_demand_bytes(int V, int H, int W) { return th_demand_bytes(V, H, W); }

int th_upsample_concat_split_demand(th_ctx* c, const float* img, const float* lat0, const float* lat1, const float* lat2,
                                    const int32_t* dims_host, int V, int H, int W, float* out, const int32_t* box,
                                    const void* demand, th_stream stream) {
    TH_REQUIRE(c && img && lat0 && lat1 && lat2 && dims_host && out && demand, "null argument");
    TH_REQUIRE((W % 64) == 0, "demand-driven map: image width must be a multiple of 64");
    const unsigned* need_map = reinterpret_cast<const unsigned*>(demand) + (size_t)V * H * W / 32;
    return th_upsample_concat_launch(img, lat0, lat1, lat2, dims_host, V, H, W, nullptr, nullptr, out, (hipStream_t)stream, 1,
                                     box, need_map);
}

int th_map_box(th_ctx* c, const float* verts_a, int na, const float* verts_b, int nb, const float* cams, int V,
               const float* scale_xy, int H, int W, float reach, int32_t* box_out, th_stream stream) {
    TH_REQUIRE(c && cams && scale_xy && box_out && (verts_a || na == 0) && (verts_b || nb == 0), "null argument");
    TH_REQUIRE(V > 0 && H > 0 && W > 0 && na >= 0 && nb >= 0 && reach >= 0.f, "bad sizes");
    return th_map_box_launch(verts_a, na, verts_b, nb, cams, V, scale_xy, H, W, reach, box_out, (hipStream_t)stream);
}

int th_upsample_concat_nhwc(th_ctx* c, const float* img, const float* lat0, const float* lat1, const float* lat2,
                            const int32_t* dims_host, int V, int H, int W, const float* color_w, const float* color_b,
                            float* out_nhwc, th_stream stream) {
    TH_REQUIRE(c && img && lat0 && lat1 && lat2 && dims_host && out_nhwc, "null argument");
    TH_REQUIRE(color_w == nullptr || color_b != nullptr, "color_b is required with color_w");
    return th_upsample_concat_launch(img, lat0, lat1, lat2, dims_host, V, H, W, color_w, color_b, out_nhwc,
                                     (hipStream_t)stream, 0, nullptr);
}

size_t th_paint_group_nhwc_workspace_bytes(int V, int n_verts, int C, int out_f) {
    if (C == TH_MAP_SPLIT) C = TH_MAP_COMPACT;       // gathered rows are 260 wide either way
    size_t rows = (size_t)V * n_verts;
    return th_align(rows * C * 4) + th_align(rows * out_f * 4) + ThPacked::bytes(out_f, C) +
           th_align((size_t)(out_f * 260 + out_f) * 4);
}

int th_paint_group_nhwc(th_ctx* c, const float* map_nhwc, int V, int H, int W, int C, const float* verts, int nv,
                        const float* cams, const float* scale, const uint8_t* viz, const th_linear* reduction,
                        const th_linear* color_lift, const int32_t* off, const int32_t* mem, int nc, float* tokens,
                        void* ws, size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && map_nhwc && verts && cams && scale && reduction && off && mem && tokens && ws, "null argument");
    const bool compact = C == TH_MAP_COMPACT || C == TH_MAP_SPLIT;
    TH_REQUIRE(reduction->in_f == (compact ? TH_MAP_FULL : C), "reduction layer must take the (full) map's channel count");
    TH_REQUIRE(!compact || (color_lift && color_lift->w && lin_ok(*color_lift, 128, 3)),
               "a compact map needs the upsample_color layer (3 -> 128) to fold into the reduction layer");
    const int out_f = reduction->out_f;
    TH_REQUIRE(ws_bytes >= th_paint_group_nhwc_workspace_bytes(V, nv, C, out_f), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ThArena ar(ws, ws_bytes);
    const size_t rows = (size_t)V * nv;
    const int CR = C == TH_MAP_SPLIT ? TH_MAP_COMPACT : C;     // width of a gathered row
    float* g = ar.take<float>(rows * CR);             // [nv][V][CR]
    float* r = ar.take<float>(rows * out_f);          // [nv][V][out_f]
    void* pk = ar.take<char>(ThPacked::bytes(out_f, CR));
    float* fold = ar.take<float>((size_t)out_f * 260 + out_f);
    TH_REQUIRE(fold != nullptr, "workspace carve failed");
    // sample the channels-last map at the projected vertices, then the 1x1 conv on those rows only
    TH_TRY(th_pixgather_launch(map_nhwc, V, C, H, W, verts, nullptr, nullptr, nv, cams, scale, g, CR, TH_ROWS_F32, s));
    ThPacked P;
    th_linear red = *reduction;
    if (compact) {
        TH_TRY(th_fold_color_launch(reduction->w, reduction->b, color_lift->w, color_lift->b, out_f, fold,
                                    fold + (size_t)out_f * 260, s));
        red.w = fold; red.b = fold + (size_t)out_f * 260; red.in_f = 260;
    }
    TH_TRY(th_pack_linear(red, pk, &P, s));
    TH_TRY(th_gemm(g, CR, (int)rows, P, TH_ACT_NONE, r, out_f, s));
    return th_segmean_masked_launch(r, V, out_f, viz, nv, off, mem, nc, tokens, s);
}

int th_bn_act_eval(th_ctx* c, const float* x, const float* residual, int N, int C, int HW, const float* gamma, const float* beta,
                   float eps, const float* running_mean, const float* running_var, int relu, float* y, th_stream stream) {
    TH_REQUIRE(c && x && y && running_mean && running_var, "null argument");
    return th_bn_act_launch(x, residual, N, C, HW, gamma, beta, eps, 0.f, const_cast<float*>(running_mean),
                            const_cast<float*>(running_var), relu, y, nullptr, 0, (hipStream_t)stream, 1);
}

size_t th_vit_workspace_bytes(int V, int N, int dim, int heads) { return th_vit_ws(V, N, dim, heads); }

size_t th_conv_pack_bytes(int cout, int cin, int ks) { return th_conv_pack_size(cout, cin, ks); }

int th_conv_pack(th_ctx* c, const float* w, int cout, int cin, int ks, void* packed, size_t packed_bytes,
                 float* inv_scale_out, th_stream stream) {
    TH_REQUIRE(c && w && packed && inv_scale_out, "null argument");
    return th_conv_pack_launch(w, cout, cin, ks, packed, packed_bytes, inv_scale_out, (hipStream_t)stream);
}

int th_conv2d_supported(int cin, int cout, int ks, int stride) { return th_conv2d_built(cin, cout, ks, stride) ? 1 : 0; }

int th_conv2d(th_ctx* c, const float* x, int N, int cin, int H, int W, const void* packed, float inv_scale, int cout, int ks,
              int stride, float* y, th_stream stream) {
    TH_REQUIRE(c && x && packed && y, "null argument");
    TH_REQUIRE(N > 0 && H > 0 && W > 0, "empty tensor");
    return th_conv2d_launch(x, N, cin, H, W, packed, inv_scale, cout, ks, stride, y, (hipStream_t)stream, c->range_dev);
}

int th_conv2d_stats_partials(int N, int cin, int H, int W, int cout, int ks, int stride) {
    int np = 0;
    if (!th_conv2d_built(cin, cout, ks, stride) || N <= 0 || H <= 0 || W <= 0) return 0;
    if (th_conv2d_launch(nullptr, N, cin, H, W, nullptr, 0.f, cout, ks, stride, nullptr, nullptr, nullptr, nullptr, &np)) return 0;
    return np;
}

int th_conv2d_stats(th_ctx* c, const float* x, int N, int cin, int H, int W, const void* packed, float inv_scale, int cout,
                    int ks, int stride, float* y, void* stats, size_t stats_bytes, th_stream stream) {
    TH_REQUIRE(c && x && packed && y && stats, "null argument");
    TH_REQUIRE(N > 0 && H > 0 && W > 0, "empty tensor");
    const int np = th_conv2d_stats_partials(N, cin, H, W, cout, ks, stride);
    TH_REQUIRE(np > 0, "th_conv2d_stats: shape not built");
    TH_REQUIRE(stats_bytes >= (size_t)cout * np * sizeof(float2), "statistics buffer too small");
    TH_REQUIRE(((uintptr_t)stats & 7) == 0, "statistics buffer must be 8-byte aligned");
    return th_conv2d_launch(x, N, cin, H, W, packed, inv_scale, cout, ks, stride, y, (hipStream_t)stream, c->range_dev,
                            (float2*)stats, nullptr);
}

int th_bn_act_stats(th_ctx* c, const float* x, const float* residual, int N, int C, int HW, const void* stats, int n_partials,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, int relu, float* y, th_stream stream) {
    TH_REQUIRE(c && x && y && stats && n_partials > 0, "null argument");
    TH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean / running_var go together");
    return th_bn_act_launch(x, residual, N, C, HW, gamma, beta, eps, momentum, running_mean, running_var, relu, y, nullptr, 0,
                            (hipStream_t)stream, 0, stats, n_partials);
}

int th_maxpool3x3s2(th_ctx* c, const float* x, int planes, int H, int W, float* y, th_stream stream) {
    TH_REQUIRE(c && x && y && planes > 0 && H > 0 && W > 0, "bad argument");
    return th_maxpool3x3s2_launch(x, planes, H, W, y, (hipStream_t)stream);
}

size_t th_bn_workspace_bytes(int N, int C, int HW) { return th_bn_ws(N, C, HW); }

int th_bn_act(th_ctx* c, const float* x, const float* residual, int N, int C, int HW, const float* gamma, const float* beta,
              float eps, float momentum, float* running_mean, float* running_var, int relu, float* y, void* ws,
              size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && x && y && ws, "null argument");
    TH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running_mean / running_var go together");
    return th_bn_act_launch(x, residual, N, C, HW, gamma, beta, eps, momentum, running_mean, running_var, relu, y, ws,
                            ws_bytes, (hipStream_t)stream);
}

int th_vit_forward(th_ctx* c, const float* x, const float* pe, int V, int N, float* out, void* ws, size_t ws_bytes,
                   th_stream stream) {
    TH_REQUIRE(c && x && pe && out && ws, "null argument");
    ProfScope sc(prof_of(c), TH_PROF_VIT, (hipStream_t)stream);
    return th_vit_launch(c->vit, x, pe, V, N, out, ws, ws_bytes, (hipStream_t)stream, c->vit_mode >= 1 ? c->range_dev : nullptr,
                         c->vit_mode >= 1);
}

int th_dparf_encode(th_ctx* c, const float* pts, const int32_t* sel, int P, const float* centres, const float* rot,
                    const float* tokens, int V, int nc, float* out, th_stream stream) {
    TH_REQUIRE(c && pts && centres && rot && tokens && out, "null argument");
    return th_dparf_launch(pts, nullptr, nullptr, nullptr, sel, P, centres, rot, tokens, V, nc, 0.5f, out, nullptr,
                           TH_ROWS_F32, nullptr, (hipStream_t)stream);
}

int th_nchw_to_nhwc(th_ctx* c, const float* src, int V, int C, int H, int W, float* dst, th_stream stream) {
    TH_REQUIRE(c && src && dst, "null argument");
    return th_nchw_to_nhwc_launch(src, V, C, H, W, dst, (hipStream_t)stream);
}

int th_pixel_gather(th_ctx* c, const float* map, int V, int C, int H, int W, const float* pts, const int32_t* sel,
                    int P, const float* cams, const float* scale, float* out, int ldo, th_stream stream) {
    TH_REQUIRE(c && map && pts && cams && scale && out, "null argument");
    return th_pixgather_launch(map, V, C, H, W, pts, nullptr, sel, P, cams, scale, out, ldo, TH_ROWS_F32,
                               (hipStream_t)stream);
}

int th_pixel_gather_split(th_ctx* c, const float* map_split, int V, int H, int W, const float* pts, const int32_t* sel, int P,
                          const float* cams, const float* scale, void* out_rows, int ldo, th_stream stream) {
    TH_REQUIRE(c && map_split && pts && cams && scale && out_rows, "null argument");
    TH_REQUIRE(ldo == 272, "split rows of the compact map are 272 wide (256 latents | r g b 0 | zeros)");
    return th_pixgather_launch(map_split, V, TH_MAP_SPLIT, H, W, pts, nullptr, sel, P, cams, scale, (float*)out_rows, ldo,
                               TH_ROWS_SPLIT, (hipStream_t)stream, nullptr);
}

int th_map_fold(th_ctx* c, const float* map_split, int V, int H, int W, const int32_t* box, float* fold, th_stream stream) {
    TH_REQUIRE(c && map_split && fold, "null argument");
    TH_REQUIRE(c->fused_ready && c->fused.compact_ready, "th_map_fold needs the MLP weights incl. upsample_color (th_set_mlp_weights)");
    ProfScope sc(prof_of(c), TH_PROF_FOLD, (hipStream_t)stream);
    return th_map_fold_launch(c->fused, map_split, V, H, W, box, fold, c->range_dev, (hipStream_t)stream);
}

int th_map_fold_demand(th_ctx* c, const float* map_split, int V, int H, int W, const void* demand, float* fold, th_stream stream) {
    TH_REQUIRE(c && map_split && fold && demand, "null argument");
    TH_REQUIRE(c->fused_ready && c->fused.compact_ready, "th_map_fold needs the MLP weights incl. upsample_color (th_set_mlp_weights)");
    ProfScope sc(prof_of(c), TH_PROF_FOLD, (hipStream_t)stream);
    return th_map_fold_launch(c->fused, map_split, V, H, W, nullptr, fold, c->range_dev, (hipStream_t)stream,
                              reinterpret_cast<const unsigned*>(demand));
}

size_t th_pixel_texlist_bytes(int V, int P) { return th_pixtex_bytes(V, P); }

int th_pixel_texlist(th_ctx* c, const float* map_split, int V, int H, int W, const float* pts, const int32_t* sel, int P,
                     const float* cams, const float* scale, void* out, size_t out_bytes, th_stream stream) {
    TH_REQUIRE(c && pts && cams && scale && out, "null argument");
    (void)map_split;               // (the lists depend on the cameras only; the argument is kept for the map's identity)
    TH_REQUIRE(out_bytes >= th_pixtex_bytes(V, P), "output too small (th_pixel_texlist_bytes)");
    ThPointSrc ps{};
    ps.pts = pts;
    return th_pixtex_launch(V, H, W, &ps, sel, P, cams, scale, out, (hipStream_t)stream);
}

int th_composite(th_ctx* c, const float* raw, const float* z, const th_points* rays, int white, float* rgb,
                 float* acc, float* depth, float* wout, th_stream stream) {
    TH_REQUIRE(c && raw && rays && rgb && acc && depth, "null argument");
    TH_REQUIRE(rays->ray_d != nullptr, "ray_d required");
    TH_REQUIRE(z || (rays->near && rays->far && rays->t_vals && rays->one_minus_t), "need z or near/far/t_vals");
    return th_composite_launch(raw, z, th_src(rays), white, rgb, acc, depth, wout, nullptr, (hipStream_t)stream);
}

int th_gen_rays(th_ctx* c, const float* K_host, const float* R_host, const float* T_host, const float* bounds_host, int H,
                int W, float* ray_o, float* ray_d, float* near_out, float* far_out, uint8_t* mask_at_box,
                th_stream stream) {
    TH_REQUIRE(c && K_host && R_host && T_host && bounds_host && ray_o && ray_d && near_out && far_out && mask_at_box,
               "null argument");
    return th_gen_rays_launch(K_host, R_host, T_host, bounds_host, H, W, ray_o, ray_d, near_out, far_out, mask_at_box,
                              (hipStream_t)stream);
}

int th_bound_mask(th_ctx* c, const int32_t* corners_xy_host, int H, int W, uint8_t* mask, th_stream stream) {
    TH_REQUIRE(c && corners_xy_host && mask && H > 0 && W > 0, "bad argument");
    return th_bound_mask_launch(corners_xy_host, H, W, mask, (hipStream_t)stream);
}

size_t th_marching_cubes_workspace_bytes(int X, int Y, int Z) { return th_mc_ws(X, Y, Z) + 256; }

int th_marching_cubes_count(th_ctx* c, const float* cube, int X, int Y, int Z, float iso, void* ws, size_t ws_bytes,
                            int64_t* counts_host, th_stream stream) {
    TH_REQUIRE(c && cube && ws && counts_host, "null argument");
    TH_REQUIRE(ws_bytes >= th_marching_cubes_workspace_bytes(X, Y, Z), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    long long* cnt = (long long*)((char*)ws + th_mc_ws(X, Y, Z));
    TH_TRY(th_mc_count_launch(cube, X, Y, Z, iso, ws, th_mc_ws(X, Y, Z), cnt, s));
    long long h[2];
    TH_HIP(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, s));
    TH_HIP(hipStreamSynchronize(s));
    counts_host[0] = h[0]; counts_host[1] = h[1];
    return 0;
}

int th_marching_cubes_range(th_ctx* c, const void* ws, int X, int Y, int Z, int x, int64_t* prefix_host, th_stream stream) {
    TH_REQUIRE(c && ws && prefix_host && x >= 0 && x <= X, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    long long* cnt = (long long*)((char*)ws + th_mc_ws(X, Y, Z)) + 4;
    TH_TRY(th_mc_prefix_launch(ws, X, Y, Z, x, cnt, s));
    long long h[2];
    TH_HIP(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, s));
    TH_HIP(hipStreamSynchronize(s));
    prefix_host[0] = h[0]; prefix_host[1] = h[1];
    return 0;
}

int th_marching_cubes_emit(th_ctx* c, const float* cube, int X, int Y, int Z, float iso, const void* ws, int x0, int x1,
                           const double* scale_host, const double* origin_host, double* verts, int32_t* tris,
                           th_stream stream) {
    TH_REQUIRE(c && cube && ws && scale_host && origin_host && verts && tris, "null argument");
    return th_mc_emit_launch(cube, X, Y, Z, iso, ws, x0, x1, scale_host, origin_host, verts, tris, (hipStream_t)stream);
}

size_t th_smpl_workspace_bytes(int n_verts) { return th_smpl_ws(n_verts); }

int th_smpl_lbs(th_ctx* c, const th_smpl_model* m, const float* pose_aa, const float* rot, const double* beta,
                double* verts, double* joints, double* T, void* ws, size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && m && beta && verts && joints && T && ws, "null argument");
    TH_REQUIRE(m->v_template && m->shapedirs && m->posedirs && m->J_regressor && m->weights && m->parent && m->n_verts > 0,
               "incomplete th_smpl_model");
    TH_REQUIRE((pose_aa != nullptr) != (rot != nullptr), "give exactly one of pose_aa (72 axis-angle) / rot ([24,3,3])");
    return th_smpl_launch(*m, pose_aa, rot, beta, verts, joints, T, ws, ws_bytes, (hipStream_t)stream);
}

int th_view_embed(th_ctx* c, const float* d, int R, int res, float* out, th_stream stream) {
    TH_REQUIRE(c && d && out, "null argument");
    return th_view_embed_launch(d, R, res, out, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// chunked per-sample stage shared by the three frame-level entry points
// ---------------------------------------------------------------------------
// Samples per pass of the per-sample stage.  The reference chunks at 32768 (batchify_rays :575) to fit a
// 2021 GPU; results are chunk-size invariant (the network is strictly per-sample).  With 288 GB of HBM a
// pass of 512 Ki samples (3.3 GB of h/f rows; 12 GB of workspace incl. the per-layer path's scratch) makes every
// launch several full waves of workgroups: the DPaRF kernel (6 workgroups/CU resident) loses a third of a 256 Ki
// launch to its partial last wave.  Measured frame 58 -> 33.1 -> 32.1 ms for 32 Ki / 256 Ki / 512 Ki chunks (1 Mi: no
// further gain).  th_set_chunk_samples() / TH_CHUNK_SAMPLES override.
static int th_chunk_init() {
    const char* e = getenv("TH_CHUNK_SAMPLES");
    long v = e ? atol(e) : 0;
    if (v >= 1024 && v <= (1L << 24)) return (int)v;
    return 524288;
}
static int TH_CHUNK = th_chunk_init();

int th_set_chunk_samples(int n) {
    TH_REQUIRE(n >= 1024 && n <= (1 << 24), "chunk must be in [1024, 2^24] samples");
    TH_CHUNK = n;
    return 0;
}

struct ChunkBufs {
    float *h, *f, *vdc, *raw_c;
    float* pe;          // [CH][2][64] halves: blended positional encoding of the folded K4 form
    void* mlp_ws;
    size_t mlp_ws_bytes;
};

// K6 dispatch: fused fp16x3-split kernel (default, V <= 3) or the layer-by-layer fp32 MFMA form
// which K6 form runs for V views, and therefore which row format the producers must emit into cb.h / cb.f
static bool mlp_is_fused(const th_ctx* c, int V) { return c->mlp_mode == 1 && c->fused_ready && V <= 3; }
static int mlp_row_format(const th_ctx* c, int V) { return mlp_is_fused(c, V) ? TH_ROWS_SPLIT : TH_ROWS_F32; }      // K5
static bool tok_gather(const th_ctx* c, int V);
static int dparf_row_format(const th_ctx* c, int V) {                                                                  // K4
    return mlp_is_fused(c, V) ? (tok_gather(c, V) ? TH_ROWS_NBR : TH_ROWS_FOLDED) : TH_ROWS_F32;
}

// bytes of K4's TH_ROWS_NBR output for m samples: records rounded up to whole tiles + one 512-byte header per tile
static size_t nbr_bytes(size_t m) { return (m + 32) * 16 * 4 + (m / 32 + 2) * 128 * 4; }
// TH_ROWS_TEX: on the fused path a frame with a split map (TH_MAP_SPLIT) hands the fused kernel texel lists instead of
// pixel-feature rows (k_pixtex.hip; th_set_tex_rows(ctx, 0) / TH_ROWS_TEX=0: K5's rows through HBM as before)
// (the fused kernel addresses a texel row as a 32-bit byte offset into the map: V * H * W texels of 1 KiB must stay below 4 GiB --
// three views of up to 1182 x 1182; larger maps keep K5's rows)
static bool tex_rows(const th_ctx* c, const th_frame* f) {
    return mlp_is_fused(c, f->V) && c->tex_rows == 1 && f->map_channels == TH_MAP_SPLIT && f->map_fold != nullptr &&
           (long long)f->V * f->H * f->W < (1LL << 22);
}

// ---- the shading pool -------------------------------------------------------------------------------------------
// Everything the per-sample stage needs PER VALID SAMPLE lives in a second caller-supplied buffer, the shading pool,
// sized from the frame's valid-sample count (on the host before the stage is queued: th_render_prepass_wait) and shared
// by all of a context's ray workspaces -- the stage runs on one stream at a time.  Round 3 carved the worst case
// (5 pre-gather sets + the per-layer path's scratch, 25 GB) into EVERY workspace: 105 GiB for the headline frame.
//   region A (fused neighbour-record path behind a prepass): the pixel-feature rows, neighbour records + tile headers and
//            positional encodings of the first pre_n = min(n, TH_PRE_SAMPLES) valid samples as ONE contiguous block each,
//            written by one K5 and one K4 launch (th_render_pregather) and read by ONE launch of the fused kernel;
//   region B the chunk buffers (TH_CHUNK samples, row format of the active path) for whatever region A does not cover.
static long th_pre_init() {
    const char* e = getenv("TH_PRE_SAMPLES");
    long v = e ? atol(e) : 0;
    return (v >= 1024 && v <= (1L << 26)) ? v : 5L * 524288;
}
static const long TH_PRE_SAMPLES = th_pre_init();
struct PoolPlan {
    long long pre_n = 0;        // samples of region A
    int ch = 0;                 // samples per pass of region B (0: no region B)
    size_t a_f = 0, a_h = 0, a_pe = 0, b_h = 0, b_f = 0, b_vdc = 0, b_pe = 0, b_mlp = 0, raw_c = 0, total = 0;
    size_t b_mlp_bytes = 0;
};
static PoolPlan pool_plan(const th_ctx* c, int V, int f_ld, long long n, bool with_pre, bool tex) {
    PoolPlan p;
    const bool fused = mlp_is_fused(c, V);
    const bool can_pre = fused && tok_gather(c, V);
    if (n < 0) n = 0;
    p.pre_n = (with_pre && can_pre) ? (n < TH_PRE_SAMPLES ? n : TH_PRE_SAMPLES) : 0;
    const long long rest = n - p.pre_n;
    p.ch = (int)(rest < TH_CHUNK ? rest : TH_CHUNK);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += th_align(bytes); return o; };
    if (p.pre_n > 0) {
        p.a_f = take(tex ? th_pixtex_bytes(V, p.pre_n) : (size_t)p.pre_n * V * f_ld * 4);
        p.a_h = take(nbr_bytes((size_t)p.pre_n));
        p.a_pe = take((size_t)p.pre_n * 64 * 4);
    }
    if (p.ch > 0) {
        const size_t rows = (size_t)V * p.ch;
        // h: fp32 rows / folded rows [rows][256], or the neighbour records of the chunk
        p.b_h = take(can_pre ? nbr_bytes((size_t)p.ch) : rows * 256 * 4);
        p.b_f = take(tex ? th_pixtex_bytes(V, p.ch) : rows * (fused ? f_ld : 384) * 4);
        p.b_vdc = take((size_t)p.ch * 27 * 4);
        p.b_pe = take((size_t)p.ch * 64 * 4);
        p.b_mlp_bytes = fused ? 0 : th_mlp_ws(V, p.ch);
        p.b_mlp = take(p.b_mlp_bytes);
    }
    const long long rc = p.pre_n > p.ch ? p.pre_n : p.ch;
    p.raw_c = take((size_t)(rc > 0 ? rc : 1) * 4 * 4);
    p.total = off;
    return p;
}

// Fused path: per-frame table T' = tokens fc_0[:, :192]^T ([V*N_c, 256] fp32, one small GEMM) that K4 blends instead
// of the raw tokens (fc_0 is linear; see the token branch of the fused kernel).  Returns the table K4 must read.
// TH_ROWS_NBR (default on the fused path, TH_TOK_GATHER=0 switches back to TH_ROWS_FOLDED): K4 hands the fused kernel the
// 7 neighbours + weights of every sample and the kernel blends the rows of T' itself, from the split form of the table
// (th_tok_split, in place; the scale word sits behind the table: TPRIME_FLOATS).
#define TH_MAX_CLUSTERS 4096     // T' scratch is sized for this many tokens per view
#define TPRIME_FLOATS(V) ((size_t)(V) * TH_MAX_CLUSTERS * 256 + 64)
static bool tok_gather(const th_ctx* c, int V) { return mlp_is_fused(c, V) && c->tok_gather == 1; }
static float* tprime_scale(float* tprime, int V) { return tprime + (size_t)V * TH_MAX_CLUSTERS * 256; }
static int token_table(th_ctx* c, const float* tokens, int V, int nc, float* tprime, const float** table, hipStream_t s) {
    *table = tokens;
    if (!mlp_is_fused(c, V)) return 0;
    TH_TRY(th_gemm(tokens, 192, V * nc, c->mlp.fc_0tok, TH_ACT_NONE, tprime, 256, s));
    if (tok_gather(c, V)) TH_TRY(th_tok_split(tprime, V * nc, tprime_scale(tprime, V), c->range_dev, s));
    *table = tprime;
    return 0;
}

// f_ld: floats per pixel-feature row of cb.f (384 full / 272 compact).  View directions: `vd_table` rows are
// addressed as vd_sel[p] / vd_div (vd_sel == nullptr: row p); the fused kernel reads the table in place, the
// per-layer form wants them gathered into cb.vdc first.
static int mlp_dispatch(th_ctx* c, int V, int m, const ChunkBufs& cb, int f_ld, const float* vd_table,
                        const int32_t* vd_sel, int vd_div, int rgb_all, hipStream_t s, float* tprime = nullptr, int nc = 0,
                        const float* tex_map = nullptr, size_t tex_stride = 0) {
    TH_REQUIRE(c->mlp.ready, "MLP weights not set (th_set_mlp_weights)");
    if (mlp_is_fused(c, V)) {
        const bool nbr = tok_gather(c, V);
        TH_REQUIRE(!nbr || tprime != nullptr, "the neighbour-record path needs the split token table");
        return th_mlp_fused_forward(c->fused, c->mlp, V, m, cb.h, cb.pe, cb.f, f_ld, vd_table, vd_sel, vd_div, rgb_all,
                                    cb.raw_c, c->range_dev, s, nbr ? tprime : nullptr, nbr ? tprime_scale(tprime, V) : nullptr, nc,
                                    tex_map, tex_stride);
    }
    const float* vd = vd_table;
    if (vd_sel != nullptr || vd_table != cb.vdc) {
        TH_TRY(th_gather_rows_launch(vd_table, 27, vd_sel, vd_div, m, cb.vdc, s));
        vd = cb.vdc;
    }
    return th_mlp_forward(c->mlp, V, m, cb.h, cb.f, f_ld, vd, cb.raw_c, cb.mlp_ws, cb.mlp_ws_bytes, s);
}

static size_t chunk_bytes(int V, int CH) {
    size_t rows = (size_t)V * CH;
    return th_align(rows * 256 * 4) + th_align(rows * 384 * 4) + th_align((size_t)CH * 27 * 4) +
           th_align((size_t)CH * 4 * 4) + th_align((size_t)CH * 64 * 4) + th_mlp_ws(V, CH);
}
static int chunk_carve(ThArena& ar, int V, int CH, ChunkBufs* b) {
    size_t rows = (size_t)V * CH;
    b->h = ar.take<float>(rows * 256);
    b->f = ar.take<float>(rows * 384);
    b->vdc = ar.take<float>((size_t)CH * 27);
    b->raw_c = ar.take<float>((size_t)CH * 4);
    b->pe = ar.take<float>((size_t)CH * 64);
    b->mlp_ws_bytes = th_mlp_ws(V, CH);
    b->mlp_ws = ar.take<char>(b->mlp_ws_bytes);
    TH_REQUIRE(b->mlp_ws != nullptr, "workspace too small");
    return 0;
}

size_t th_network_workspace_bytes(int V, int P) {
    int CH = P < TH_CHUNK ? (P > 0 ? P : 1) : TH_CHUNK;
    return chunk_bytes(V, CH) + th_align((size_t)P * 4) + th_compact_ws(P) + th_align(64) +
           th_align(TPRIME_FLOATS(V) * 4);
}

int th_network_forward(th_ctx* c, const float* pixel_feat, const float* viewdir, const float* pts_smpl,
                       const uint8_t* mask, int P, const float* centres, const float* rot, const float* tokens,
                       int V, int nc, float* raw_out, void* ws, size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && pixel_feat && viewdir && pts_smpl && centres && rot && tokens && raw_out && ws, "null argument");
    hipStream_t s = (hipStream_t)stream;
    if (P <= 0) return 0;
    TH_REQUIRE(ws_bytes >= th_network_workspace_bytes(V, P), "workspace too small");
    ThArena ar(ws, ws_bytes);
    int CH = P < TH_CHUNK ? P : TH_CHUNK;
    ChunkBufs cb;
    TH_TRY(chunk_carve(ar, V, CH, &cb));
    TH_REQUIRE(nc <= TH_MAX_CLUSTERS, "too many token clusters");
    float* tprime = ar.take<float>(TPRIME_FLOATS(V));
    TH_REQUIRE(tprime != nullptr, "workspace too small");
    const float* table = nullptr;
    TH_TRY(token_table(c, tokens, V, nc, tprime, &table, s));
    int32_t* idx = nullptr;
    int n = P;
    if (mask) {
        idx = ar.take<int32_t>((size_t)P);
        void* cws = ar.take<char>(th_compact_ws(P));
        int32_t* dcount = ar.take<int32_t>(16);
        TH_REQUIRE(dcount != nullptr, "workspace too small");
        TH_TRY(th_compact_mask(mask, P, idx, dcount, cws, th_compact_ws(P), s));
        TH_HIP(hipMemsetAsync(raw_out, 0, (size_t)P * 16, s));     // raw_temp zeros, :231
        TH_HIP(hipMemcpyAsync(c->host_pinned, dcount, 4, hipMemcpyDeviceToHost, s));
        TH_HIP(hipStreamSynchronize(s));                           // the reference syncs here too (:232)
        n = c->host_pinned[0];
    }
    for (int o = 0; o < n; o += CH) {
        int m = (n - o) < CH ? (n - o) : CH;
        const int32_t* sel = idx ? idx + o : nullptr;
        const float* pts = idx ? pts_smpl : pts_smpl + 3LL * o;
        const int fmt = mlp_row_format(c, V);
        TH_TRY(th_dparf_launch(pts, nullptr, nullptr, nullptr, sel, m, centres, rot, table, V, nc, 0.5f, cb.h, cb.pe,
                               dparf_row_format(c, V), nullptr, s));
        if (idx) TH_TRY(th_gather_chan_major_launch(pixel_feat, V, 384, P, sel, m, cb.f, fmt, s, c->range_dev));
        else TH_TRY(th_gather_chan_major_launch(pixel_feat + o, V, 384, P, nullptr, m, cb.f, fmt, s, c->range_dev));
        if (idx) TH_TRY(mlp_dispatch(c, V, m, cb, 384, viewdir, sel, 1, 0, s, tprime, nc));
        else TH_TRY(mlp_dispatch(c, V, m, cb, 384, viewdir + 27LL * o, nullptr, 1, 1, s, tprime, nc));
        if (idx) TH_TRY(th_scatter_raw_launch(cb.raw_c, sel, m, 0, raw_out, s));
        else TH_TRY(th_scatter_raw_launch(cb.raw_c, nullptr, m, 1, raw_out + 4LL * o, s));
    }
    return th_range_snapshot(c, stream) < 0 ? -1 : 0;
}

// ---------------------------------------------------------------------------
// frame-level
// ---------------------------------------------------------------------------
static int frame_ok(const th_frame* f) {
    TH_REQUIRE(f->verts_world && f->Rh && f->Th && f->cams && f->scale_xy && f->pixel_map_nhwc && f->tokens &&
                   f->centres && f->rot,
               "incomplete th_frame");
    TH_REQUIRE(f->V >= 1 && f->V <= 4, "supported reference-view counts: 1..4");
    TH_REQUIRE(f->map_channels == TH_MAP_FULL || f->map_channels == TH_MAP_COMPACT || f->map_channels == TH_MAP_SPLIT,
               "th_frame.map_channels must be 384 (full pixel_feat_map), 260 (compact map) or 256 (compact map, split planes)");
    return 0;
}

// ray-stage workspace (per frame in flight): hull mask, hit flags, grid, compaction scratch, sample list, counts, view
// embeddings, dense raw, the per-frame token table and K4's candidate grid.  Nothing here scales with the VALID samples.
static size_t shade_ws_bytes(const th_frame* f, long long P, int R) {
    return th_align((size_t)P) + th_align((size_t)R * 4) + th_hull_ws(f->n_verts) + th_compact_ws(P) +
           th_align((size_t)P * 4) + th_align(64) + th_align((size_t)R * 27 * 4) + th_align((size_t)P * 16) +
           th_align(TPRIME_FLOATS(f->V) * 4) + th_dparf_grid_ws(f->n_clusters > 0 ? f->n_clusters : 1);
}
// (one carve for every entry point that works in a ray workspace)
struct ShadeWs {
    uint8_t* mask; int32_t* ray_hit; void* hws; size_t hws_b; void* cws; size_t cws_b; int32_t* idx; int32_t* info;
    float* vd_all; float* raw; float* tprime; void* gws; size_t gws_b;
};
static ShadeWs carve_shade_ws(const th_frame* f, long long P, int R, ThArena& ar) {
    ShadeWs w{};
    w.mask = ar.take<uint8_t>((size_t)P);
    w.ray_hit = ar.take<int32_t>((size_t)R);
    w.hws_b = th_hull_ws(f->n_verts);
    w.hws = ar.take<char>(w.hws_b);
    w.cws_b = th_compact_ws(P);
    w.cws = ar.take<char>(w.cws_b);
    w.idx = ar.take<int32_t>((size_t)P);
    w.info = ar.take<int32_t>(16);
    w.vd_all = ar.take<float>((size_t)R * 27);
    w.raw = ar.take<float>((size_t)P * 4);
    w.tprime = ar.take<float>(TPRIME_FLOATS(f->V));
    w.gws_b = th_dparf_grid_ws(f->n_clusters > 0 ? f->n_clusters : 1);
    w.gws = ar.take<char>(w.gws_b);
    return w;
}
static int frame_f_ld(const th_frame* f) {
    return (f->map_channels == TH_MAP_COMPACT || f->map_channels == TH_MAP_SPLIT) ? 272 : 384;
}

// hull mask -> (small-frame rule) -> compaction -> chunked DPaRF + gather + MLP -> dense raw[P,4]
// stage A (hull mask, small-frame rule, compaction, view embedding, cleared raw) may run ahead of time
// (th_render_prepass): it needs only the rays, the posed vertices and the two thresholds.  `prepass` = 1: run
// stage A only and leave the counts on their way to host_pinned[16..]; 2: stage A already ran into this workspace;
// 3: stage A ran, queue the pre-gather stage (region A of the pool) and return.
static int shade_points(th_ctx* c, const th_frame* f, const ThPointSrc& ps, long long P, bool ray_mode, ThArena& ar,
                        void* pool, size_t pool_bytes, float** raw_out, const uint8_t** mask_out, int64_t* stats_host,
                        hipStream_t s, int prepass = 0, int slot = 0, const int32_t** ray_hit_out = nullptr) {
    const int R = ps.R, S = ps.S, V = f->V;
    const bool compact = f->map_channels == TH_MAP_COMPACT || f->map_channels == TH_MAP_SPLIT;
    const int f_ld = frame_f_ld(f);
    const int fmt = mlp_row_format(c, V);
    const bool tex = tex_rows(c, f);
    const float* tex_map = tex ? f->map_fold : nullptr;                  // fold0, fold12 one map size behind it
    const size_t tex_stride = (size_t)f->V * f->H * f->W * 256;
    TH_REQUIRE(prepass == 1 || !compact || c->mlp.compact_ready,
               "compact pixel map needs th_mlp_weights.upsample_color (colour-folded layers) to be uploaded");
    const ShadeWs w = carve_shade_ws(f, P, R, ar);
    uint8_t* mask = w.mask;
    int32_t* ray_hit = w.ray_hit;
    const size_t hws_b = w.hws_b, cws_b = w.cws_b, gws_b = w.gws_b;
    void *hws = w.hws, *cws = w.cws, *gws = w.gws;
    int32_t *idx = w.idx, *info = w.info;
    float *vd_all = w.vd_all, *raw = w.raw, *tprime = w.tprime;
    TH_REQUIRE(raw != nullptr && tprime != nullptr && gws != nullptr, "workspace too small");

    ThProf* pf = prof_of(c);
    int32_t* hp = c->host_pinned;
    // Any per-sample stage that is handed a pool (th_render_rays, th_eval_sigma_grid) becomes the pool's last user: the
    // "pool is free behind the previous th_render_rays" event th_render_pregather_early relies on is void from here on and
    // is re-armed only by a th_render_rays that ran to its end (an error return leaves it void: the early path then falls
    // back to ordering K4 behind everything queued on the stream).
    if ((prepass == 0 || prepass == 2) && pool != nullptr) c->after_shade_valid = false;
    if (prepass == 2 || prepass == 3) {
        hp = c->host_pinned + 16 + 4 * slot;
        TH_HIP(hipStreamWaitEvent(s, c->prepass[slot].ev, 0));      // the prepass may have run on another stream
        ThWaitClock wc(c);
        TH_HIP(hipEventSynchronize(c->prepass[slot].ev));
    } else {
    ProfScope sc_hull(pf, TH_PROF_HULL, s);          // (RAII: an early error return closes the span)
    const bool no_hull = f->hull_thresh < 0.f;   // Renderer.render (:486-498): every sample shaded, RGB everywhere
    if (no_hull) {
        TH_HIP(hipMemsetAsync(info, 0, 16 * 4, s));
        TH_HIP(hipMemsetAsync(mask, 1, (size_t)P, s));
        if (ray_mode) TH_HIP(hipMemsetAsync(ray_hit, 1, (size_t)R * 4, s));   // any non-zero marks a hit
    } else {
        // (the grid build clears `info` and the per-ray hit flags: no memset launches)
        TH_TRY(th_hull_mask_launch(ps, P, f->verts_world, f->n_verts, f->hull_thresh, mask,
                                   ray_mode ? ray_hit : nullptr, hws, hws_b, s, info));
    }
    if (ray_mode) {
        // hit-ray count, the R' <= 2400 rule (:551) and the compaction in one chain (no_hull: threshold R makes the
        // rule fire -> un-masked mode, rgb for all samples)
        TH_TRY(th_view_embed_launch(ps.ray_d, R, 4, vd_all, s, ray_hit));      // (behind the hull test: hit rays only)
        TH_TRY(th_compact_mask_rule(mask, P, ray_hit, R, S, no_hull ? R : f->small_frame_rays, info, idx, info + 2, cws,
                                    cws_b, s));
    } else {
        TH_TRY(th_compact_mask(mask, P, idx, info + 2, cws, cws_b, s));
    }
    // (raw is NOT cleared: its consumers read it through the mask -- 268 MB of memset + dense re-read saved per frame)
    sc_hull.close();
    if (prepass == 1) {
        TH_HIP(hipMemcpyAsync(c->host_pinned + 16 + 4 * slot, info, 4 * 4, hipMemcpyDeviceToHost, s));
        if (!c->prepass[slot].ev) TH_HIP(hipEventCreateWithFlags(&c->prepass[slot].ev, hipEventDisableTiming));
        TH_HIP(hipEventRecord(c->prepass[slot].ev, s));
        return 0;
    }
    TH_HIP(hipMemcpyAsync(c->host_pinned, info, 4 * 4, hipMemcpyDeviceToHost, s));
    ThWaitClock wc(c);
    TH_HIP(hipStreamSynchronize(s));
    }
    const int hit_rays = hp[0], unmasked = hp[1], n = hp[2];
    if (stats_host) { stats_host[0] = hit_rays; stats_host[1] = n; stats_host[2] = -1; stats_host[3] = unmasked; }
    th_ctx::Prepass* tk = (prepass == 2 || prepass == 3) ? &c->prepass[slot] : nullptr;
    // A cropped map (th_frame.map_source) holds the texels within reach of the hull only.  The un-masked branch (:551,
    // R' <= small_frame_rays: every sample of the hit rays is shaded), a frame without a hull test and a hull test wider
    // than the crop's reach gather outside it: the rest of the map is written first (same values inside the box) -- once
    // per frame (the pre-gather stage and the shading of one prepass share the completed map).
    // A demand-driven map (th_map_source.demand, th_render_predemand) additionally holds only the texels of ONE prepass's sample
    // list: any other caller of the frame (other rays, a sigma grid, a render without the prepass) gets the rest first too.
    const bool demand_miss = f->map_source != nullptr && f->map_source->demand != nullptr &&
                             !(tk && tk->demand == f->map_source->demand);
    if (f->map_source != nullptr && n > 0 && !(tk && tk->map_done == f->pixel_map_nhwc) &&
        !(f->map_source->demand != nullptr && c->map_completed != nullptr && c->map_completed == f->pixel_map_nhwc &&
          c->map_completed_demand == (const void*)f->map_source->demand) &&       // (demand maps only, keyed on the PAIR: a freed
                                                                                   // block handed to a later cropped map must not match)
        (f->map_source->demand != nullptr ? demand_miss      // (a demand made from THIS sample list covers every branch)
                                          : (unmasked || f->hull_thresh < 0.f || f->hull_thresh > f->map_source->reach))) {
        const th_map_source* ms = f->map_source;
        TH_REQUIRE(f->map_channels == TH_MAP_SPLIT && ms->img && ms->lat0 && ms->lat1 && ms->lat2, "map_source: split map only");
        TH_TRY(th_upsample_concat_launch(ms->img, ms->lat0, ms->lat1, ms->lat2, ms->dims, V, f->H, f->W, nullptr, nullptr,
                                         const_cast<float*>(f->pixel_map_nhwc), s, 1, nullptr));
        // (the folded maps of the texel hand-over follow the map: alpha_res_0 / rgb_res_0 / rgb_res_1 of EVERY texel now)
        if (tex) TH_TRY(th_map_fold_launch(c->fused, f->pixel_map_nhwc, V, f->H, f->W, nullptr, const_cast<float*>(f->map_fold),
                                           c->range_dev, s));
        if (tk) tk->map_done = f->pixel_map_nhwc;
        if (f->map_source->demand != nullptr) {                                          // (until the next th_render_predemand / prepass)
            c->map_completed = f->pixel_map_nhwc;
            c->map_completed_demand = (const void*)f->map_source->demand;
        }
    }
    const bool can_pre = ray_mode && tok_gather(c, V) && fmt == TH_ROWS_SPLIT;
    char* pb = (char*)pool;
    if (prepass == 3) {
        // pre-gather stage: K5 + K4 of the first pre_n valid samples, ONE launch each (needs the map, the cameras, the
        // token centres -- not the tokens)
        th_ctx::Prepass& t = c->prepass[slot];
        t.npre = 0;
        if (!can_pre || n <= 0) return 0;
        const PoolPlan pl = pool_plan(c, V, f_ld, n, true, tex);
        TH_REQUIRE(pool != nullptr && pool_bytes >= pl.total, "shading pool too small (th_shade_pool_bytes)");
        const int m = (int)pl.pre_n;
        float* a_f = (float*)(pb + pl.a_f);
        float* a_h = (float*)(pb + pl.a_h);
        float* a_pe = (float*)(pb + pl.a_pe);
        const bool grid = getenv("TH_DPARF_NOGRID") == nullptr && th_dparf_grid_ok(f->n_clusters);
        // K4 on the context's second stream, K5 on `s`: the two producers share nothing but the sample list -- K5 sits on
        // the texture path (TA busy 80-90 %, VALU 43 %), K4 since TH_ROWS_NBR is a 7-NN scan out of LDS (no row gather) --
        // so their waves co-reside on the CUs instead of running back to back (TH_K4_SIDE=0: one stream, K4 then K5).
        static const bool k4_side = !(getenv("TH_K4_SIDE") && getenv("TH_K4_SIDE")[0] == '0');
        hipStream_t s4 = s, s5 = s;
        if (k4_side) {
            if (!c->aux) {
                TH_HIP(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
                TH_HIP(hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
                TH_HIP(hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
                TH_HIP(hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking));
                TH_HIP(hipEventCreateWithFlags(&c->aux2_join, hipEventDisableTiming));
            }
            // th_render_pregather_early: K4 is ordered behind the per-sample stage of the previous th_render_rays on this
            // stream and pool (the last user of the pool's record regions) instead of behind everything queued on `s` since
            // -- that frame's compositing and whatever the caller queued after it run beside K4, not in front of it
            const bool early = c->pregather_early && c->after_shade_valid && c->after_shade_stream == s &&
                               c->after_shade_pool == pool;
            c->after_shade_valid = false;
            // K5t (the texel hand-over's producer: rays, cameras, sample list -- no map) gets a stream of its own under the same
            // rule: in the early form it starts with K4 behind the previous frame's per-sample stage, beside that frame's
            // compositing and the consumer's image assembly, instead of behind them on `s` (the window between two launches of
            // the fused kernel of a rank of 8: 360 -> 240 us)
            if (early) {
                TH_HIP(hipStreamWaitEvent(c->aux, c->after_shade, 0));
                if (tex) TH_HIP(hipStreamWaitEvent(c->aux2, c->after_shade, 0));
            } else {
                TH_HIP(hipEventRecord(c->aux_fork, s));      // sample list, candidate grid and every earlier user of the pool
                TH_HIP(hipStreamWaitEvent(c->aux, c->aux_fork, 0));
                if (tex) TH_HIP(hipStreamWaitEvent(c->aux2, c->aux_fork, 0));
            }
            s4 = c->aux;
            if (tex) s5 = c->aux2;
        }
        // From here on work may be in flight on the second stream: whatever happens, `s` waits for it before this call
        // returns (a caller that frees or reuses the pool after an error must not race with K4).
        int rc = 0;
        {
            // (the candidate grid of the 7-NN scan is K4's alone: built on K4's stream, not in front of the pixel gather)
            if (grid && t.grid_centres != f->centres) rc = th_dparf_grid_build(f->centres, f->n_clusters, gws, gws_b, s4);
            t.grid_centres = nullptr;                 // (th_render_pregrid's grid is consumed: chunks beyond the stage rebuild)
            if (rc == 0) {
                ProfScope ps1(pf, TH_PROF_DPARF, s4);
                rc = th_dparf_launch(nullptr, &ps, f->Rh, f->Th, idx, m, f->centres, f->rot, nullptr, V, f->n_clusters, 0.5f, a_h,
                                     a_pe, TH_ROWS_NBR, grid ? gws : nullptr, s4);
            }
            if (rc == 0) {
                ProfScope ps2(pf, TH_PROF_GATHER, s5);
                rc = tex ? th_pixtex_launch(V, f->H, f->W, &ps, idx, m, f->cams, f->scale_xy, a_f, s5)
                         : th_pixgather_launch(f->pixel_map_nhwc, V, f->map_channels, f->H, f->W, nullptr, &ps, idx, m, f->cams,
                                               f->scale_xy, a_f, f_ld, fmt, s, c->range_dev);
            }
        }
        // th_render_pregather_early with the tokens already in the frame: the token table T' (one small GEMM + its split)
        // is queued here, beside K4, instead of between K4's end and the fused MLP's start
        t.pre_tokens = nullptr;
        if (rc == 0 && c->pregather_early && f->tokens != nullptr) {
            const float* table = nullptr;
            rc = token_table(c, f->tokens, V, f->n_clusters, tprime, &table, s);
            if (rc == 0) t.pre_tokens = f->tokens;
        }
        if (k4_side) {
            const hipError_t e1 = hipEventRecord(c->aux_join, c->aux);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, c->aux_join, 0) : e1;
            if (e2 != hipSuccess) (void)hipStreamSynchronize(c->aux);      // last resort: nothing left in flight
            if (s5 != s) {
                const hipError_t e3 = hipEventRecord(c->aux2_join, c->aux2);
                const hipError_t e4 = e3 == hipSuccess ? hipStreamWaitEvent(s, c->aux2_join, 0) : e3;
                if (e4 != hipSuccess) (void)hipStreamSynchronize(c->aux2);
            }
        }
        if (rc != 0) return rc;
        if (!t.ev2) TH_HIP(hipEventCreateWithFlags(&t.ev2, hipEventDisableTiming));
        TH_HIP(hipEventRecord(t.ev2, s));
        t.npre = m;
        t.pre_map = f->pixel_map_nhwc;
        t.pre_centres = f->centres;
        t.pre_pool = pool;
        return 0;
    }
    int npre = 0;                        // valid samples whose rows / records sit in region A of the pool
    if (prepass == 2) {
        th_ctx::Prepass& t = c->prepass[slot];
        if (t.npre > 0 && can_pre && t.pre_map == f->pixel_map_nhwc && t.pre_centres == f->centres && t.pre_pool == pool) {
            npre = t.npre;
            TH_HIP(hipStreamWaitEvent(s, t.ev2, 0));
        }
        t.npre = 0;
    }
    const PoolPlan pl = pool_plan(c, V, f_ld, n, npre > 0, tex);
    TH_REQUIRE(n <= 0 || (pool != nullptr && pool_bytes >= pl.total), "shading pool too small (th_shade_pool_bytes)");
    TH_REQUIRE(npre == 0 || npre == pl.pre_n, "pre-gather stage and shading disagree on the pool layout");
    ChunkBufs cb{};
    const int CH = pl.ch;
    cb.raw_c = (float*)(pb + pl.raw_c);
    if (CH > 0) {
        cb.h = (float*)(pb + pl.b_h); cb.f = (float*)(pb + pl.b_f); cb.vdc = (float*)(pb + pl.b_vdc);
        cb.pe = (float*)(pb + pl.b_pe); cb.mlp_ws = pb + pl.b_mlp; cb.mlp_ws_bytes = pl.b_mlp_bytes;
    }
    if (!ray_mode && CH > 0) TH_HIP(hipMemsetAsync(cb.vdc, 0, (size_t)CH * 27 * 4, s));   // zero view dirs, if_mesh_renderer.py:62
    TH_REQUIRE(f->n_clusters <= TH_MAX_CLUSTERS, "too many token clusters");
    const float* table = nullptr;
    if (n > 0) {
        if (npre > 0 && c->prepass[slot].pre_tokens == f->tokens && mlp_is_fused(c, V)) table = tprime;    // (queued by the pre-gather stage)
        else TH_TRY(token_table(c, f->tokens, V, f->n_clusters, tprime, &table, s));
    }
    if (prepass == 2) c->prepass[slot].pre_tokens = nullptr;
    // exact candidate grid for the 7-NN scan of K4 (TH_DPARF_NOGRID=1: full scan, same result)
    // (not needed when every sample's records were pre-gathered: K4 does not run again)
    const bool use_grid = n > npre && getenv("TH_DPARF_NOGRID") == nullptr && th_dparf_grid_ok(f->n_clusters);
    if (use_grid) TH_TRY(th_dparf_grid_build(f->centres, f->n_clusters, gws, gws_b, s));
    if (npre > 0) {          // rows and records of these samples were written by th_render_pregather: ONE fused launch
        ChunkBufs pc = cb;
        pc.f = (float*)(pb + pl.a_f); pc.h = (float*)(pb + pl.a_h); pc.pe = (float*)(pb + pl.a_pe);
        {
            ProfScope ps3(pf, TH_PROF_MLP, s);
            TH_TRY(mlp_dispatch(c, V, npre, pc, f_ld, vd_all, idx, S, unmasked, s, tprime, f->n_clusters, tex_map, tex_stride));
        }
        ProfScope ps4(pf, TH_PROF_COMPOSITE, s);
        TH_TRY(th_scatter_raw_launch(cb.raw_c, idx, npre, unmasked, raw, s));
    }
    for (int o = npre; o < n; o += CH) {
        int m = (n - o) < CH ? (n - o) : CH;
        const int32_t* sel = idx + o;
        // (K5 first: K4's small output -- records, tile headers, positional encodings, 0.3 KB per sample -- is then the
        // last thing written before the fused kernel reads it at the start of every tile: MALL instead of HBM)
        {
            ProfScope ps2(pf, TH_PROF_GATHER, s);
            if (tex) TH_TRY(th_pixtex_launch(V, f->H, f->W, &ps, sel, m, f->cams, f->scale_xy, cb.f, s));
            else TH_TRY(th_pixgather_launch(f->pixel_map_nhwc, V, f->map_channels, f->H, f->W, nullptr, &ps, sel, m, f->cams,
                                            f->scale_xy, cb.f, f_ld, fmt, s, c->range_dev));
        }
        {
            ProfScope ps1(pf, TH_PROF_DPARF, s);
            TH_TRY(th_dparf_launch(nullptr, &ps, f->Rh, f->Th, sel, m, f->centres, f->rot, table, V,
                                   f->n_clusters, 0.5f, cb.h, cb.pe, dparf_row_format(c, V), use_grid ? gws : nullptr, s));
        }
        {
            ProfScope ps3(pf, TH_PROF_MLP, s);
            // ray mode: the [R,27] embedding table is indexed sample -> ray (sel / S); mesh mode: zero rows (cb.vdc)
            if (ray_mode) TH_TRY(mlp_dispatch(c, V, m, cb, f_ld, vd_all, sel, S, unmasked, s, tprime, f->n_clusters, tex_map, tex_stride));
            // (the sigma grid never looks at colour: skip the RGB branch, which the reference evaluates and drops,
            // if_mesh_renderer.py:84-99)
            else TH_TRY(mlp_dispatch(c, V, m, cb, f_ld, cb.vdc, nullptr, 1, 2, s, tprime, f->n_clusters, tex_map, tex_stride));
        }
        ProfScope ps4(pf, TH_PROF_COMPOSITE, s);
        TH_TRY(th_scatter_raw_launch(cb.raw_c, sel, m, unmasked, raw, s));
    }
    *raw_out = raw;
    *mask_out = mask;
    if (ray_hit_out) *ray_hit_out = ray_mode ? ray_hit : nullptr;
    return 0;
}

size_t th_render_workspace_bytes(const th_frame* f, int R, int S) {
    return shade_ws_bytes(f, (long long)R * S, R);
}

size_t th_shade_pool_bytes(th_ctx* c, const th_frame* f, long long n_valid, int with_pregather) {
    if (!c || !f || f->V < 1) return 0;
    const int f_ld = frame_f_ld(f);
    const bool tex = tex_rows(c, f);
    size_t a = pool_plan(c, f->V, f_ld, n_valid, false, tex).total;
    if (with_pregather) {
        const size_t b = pool_plan(c, f->V, f_ld, n_valid, true, tex).total;
        if (b > a) a = b;
    }
    return a;
}

int th_render_prepass_wait(th_ctx* c, const void* ws, int64_t* counts_host) {
    TH_REQUIRE(c && ws && counts_host, "null argument");
    for (int k = 0; k < th_ctx::kPrepassSlots; ++k) {
        th_ctx::Prepass& t = c->prepass[k];
        if (!t.valid || t.ws != ws || !t.ev) continue;
        {
            ThWaitClock wc(c);
            TH_HIP(hipEventSynchronize(t.ev));
        }
        const int32_t* hp = c->host_pinned + 16 + 4 * k;
        counts_host[0] = hp[0]; counts_host[1] = hp[1]; counts_host[2] = hp[2];
        return 0;
    }
    return 1;                                   // no prepass pending for this workspace
}

int th_render_rays(th_ctx* c, const th_frame* f, const th_points* rays, float* rgb, float* acc, float* depth,
                   int white_bkgd, void* ws, size_t ws_bytes, void* pool, size_t pool_bytes, int64_t* stats_host,
                   th_stream stream) {
    TH_REQUIRE(c && f && rays && rgb && acc && depth && ws, "null argument");
    TH_TRY(frame_ok(f));
    TH_REQUIRE(rays->pts == nullptr && rays->ray_o && rays->ray_d && rays->near && rays->far && rays->t_vals &&
                   rays->one_minus_t,
               "th_render_rays needs a complete ray description");
    hipStream_t s = (hipStream_t)stream;
    const int R = rays->R, S = rays->S;
    if (R <= 0) return 0;
    long long P = (long long)R * S;
    TH_REQUIRE(P < (1LL << 31), "R*S must fit in int32");
    TH_REQUIRE(ws_bytes >= shade_ws_bytes(f, P, R), "workspace too small");
    ThArena ar(ws, ws_bytes);
    ThPointSrc ps = th_src(rays);
    float* raw = nullptr;
    const uint8_t* mask = nullptr;
    // a matching th_render_prepass (same workspace, same ray arrays) already ran the hull / compaction stage
    int slot = -1;
    for (int k = 0; k < th_ctx::kPrepassSlots; ++k) {
        th_ctx::Prepass& t = c->prepass[k];
        if (!t.valid || t.ws != ws) continue;
        t.valid = false;                              // consumed, or stale for this workspace
        if (t.rays == (const void*)rays->ray_o && t.R == R && t.S == S) slot = k;
    }
    const int32_t* ray_hit = nullptr;
    TH_TRY(shade_points(c, f, ps, P, true, ar, pool, pool_bytes, &raw, &mask, stats_host, s, slot >= 0 ? 2 : 0,
                        slot >= 0 ? slot : 0, &ray_hit));
    // (for th_render_pregather_early of the next frame: the pool's record regions are free from here on)
    if (!c->after_shade) TH_HIP(hipEventCreateWithFlags(&c->after_shade, hipEventDisableTiming));
    TH_HIP(hipEventRecord(c->after_shade, s));
    c->after_shade_stream = s;
    c->after_shade_pool = pool;
    c->after_shade_valid = true;
    ProfScope sc(prof_of(c), TH_PROF_COMPOSITE, s);
    TH_TRY(th_composite_launch(raw, nullptr, ps, white_bkgd, rgb, acc, depth, nullptr, mask, s, ray_hit));
    const int snap = th_range_snapshot(c, stream);
    TH_REQUIRE(snap >= 0, "range snapshot failed");
    if (stats_host) stats_host[2] = snap;
    return 0;
}

int th_render_prepass(th_ctx* c, const th_frame* f, const th_points* rays, void* ws, size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && f && rays && ws, "null argument");
    TH_REQUIRE(f->verts_world && f->n_verts > 0 && f->V >= 1 && f->V <= 4, "prepass needs verts_world, n_verts and V");
    TH_REQUIRE(rays->pts == nullptr && rays->ray_o && rays->ray_d && rays->near && rays->far && rays->t_vals &&
                   rays->one_minus_t,
               "th_render_prepass needs a complete ray description");
    hipStream_t s = (hipStream_t)stream;
    const int R = rays->R, S = rays->S;
    // slot: the one already tied to this workspace, else a free one, else the oldest (round robin)
    int slot = -1;
    for (int k = 0; k < th_ctx::kPrepassSlots && slot < 0; ++k)
        if (c->prepass[k].ws == ws) slot = k;
    for (int k = 0; k < th_ctx::kPrepassSlots && slot < 0; ++k)
        if (!c->prepass[k].valid) slot = k;
    if (slot < 0) slot = c->prepass_rr = (c->prepass_rr + 1) % th_ctx::kPrepassSlots;
    th_ctx::Prepass& t = c->prepass[slot];
    t.valid = false;
    t.npre = 0;                                   // (a pre-gather of an abandoned frame must not outlive its prepass)
    t.map_done = nullptr;
    t.pre_tokens = nullptr;
    t.grid_centres = nullptr;
    t.demand = nullptr;
    if (R <= 0) return 0;
    long long P = (long long)R * S;
    TH_REQUIRE(P < (1LL << 31), "R*S must fit in int32");
    TH_REQUIRE(ws_bytes >= shade_ws_bytes(f, P, R), "workspace too small");
    ThArena ar(ws, ws_bytes);
    ThPointSrc ps = th_src(rays);
    float* raw = nullptr;
    const uint8_t* mask = nullptr;
    TH_TRY(shade_points(c, f, ps, P, true, ar, nullptr, 0, &raw, &mask, nullptr, s, 1, slot));
    t.ws = ws; t.rays = rays->ray_o; t.R = R; t.S = S;
    t.valid = true;
    return 0;
}

int th_render_pregather(th_ctx* c, const th_frame* f, const th_points* rays, void* ws, size_t ws_bytes, void* pool,
                        size_t pool_bytes, th_stream stream) {
    TH_REQUIRE(c && f && rays && ws, "null argument");
    TH_REQUIRE(f->verts_world && f->Rh && f->Th && f->cams && f->scale_xy && f->pixel_map_nhwc && f->centres && f->rot &&
                   f->n_clusters >= 7 && f->V >= 1 && f->V <= 4,
               "th_render_pregather needs every th_frame field except the tokens");
    TH_REQUIRE(f->map_channels == TH_MAP_FULL || f->map_channels == TH_MAP_COMPACT || f->map_channels == TH_MAP_SPLIT,
               "th_frame.map_channels must be 384, 260 or 256");
    int slot = -1;
    for (int k = 0; k < th_ctx::kPrepassSlots; ++k) {
        th_ctx::Prepass& t = c->prepass[k];
        if (t.valid && t.ws == ws && t.rays == (const void*)rays->ray_o && t.R == rays->R && t.S == rays->S) slot = k;
    }
    if (slot < 0) return 0;                      // no matching th_render_prepass: nothing to do (th_render_rays runs everything)
    const int R = rays->R, S = rays->S;
    long long P = (long long)R * S;
    TH_REQUIRE(ws_bytes >= shade_ws_bytes(f, P, R), "workspace too small");
    ThArena ar(ws, ws_bytes);
    ThPointSrc ps = th_src(rays);
    float* raw = nullptr;
    const uint8_t* mask = nullptr;
    return shade_points(c, f, ps, P, true, ar, pool, pool_bytes, &raw, &mask, nullptr, (hipStream_t)stream, 3, slot);
}

int th_render_pregrid(th_ctx* c, const th_frame* f, const th_points* rays, void* ws, size_t ws_bytes, th_stream stream) {
    TH_REQUIRE(c && f && rays && ws && f->centres, "null argument");
    if (getenv("TH_DPARF_NOGRID") != nullptr || !th_dparf_grid_ok(f->n_clusters)) return 0;
    int slot = -1;
    for (int k = 0; k < th_ctx::kPrepassSlots; ++k) {
        th_ctx::Prepass& t = c->prepass[k];
        if (t.valid && t.ws == ws && t.rays == (const void*)rays->ray_o && t.R == rays->R && t.S == rays->S) slot = k;
    }
    if (slot < 0) return 0;                      // no matching th_render_prepass: the pre-gather stage builds the grid itself
    const long long P = (long long)rays->R * rays->S;
    TH_REQUIRE(ws_bytes >= shade_ws_bytes(f, P, rays->R), "workspace too small");
    ThArena ar(ws, ws_bytes);
    const ShadeWs w = carve_shade_ws(f, P, rays->R, ar);
    TH_REQUIRE(w.gws != nullptr, "workspace too small");
    c->prepass[slot].grid_centres = nullptr;
    TH_TRY(th_dparf_grid_build(f->centres, f->n_clusters, w.gws, w.gws_b, (hipStream_t)stream));
    c->prepass[slot].grid_centres = f->centres;
    return 0;
}

int th_render_predemand(th_ctx* c, const th_frame* f, const th_points* rays, void* ws, size_t ws_bytes, const float* verts_paint,
                        int n_paint, void* demand, size_t demand_bytes, th_stream stream) {
    TH_REQUIRE(c && f && rays && ws && demand && f->cams && f->scale_xy, "null argument");
    TH_REQUIRE(f->V >= 1 && f->V <= 3 && f->H >= 1 && f->W >= 64, "th_render_predemand: 1..3 views and their map size");
    TH_REQUIRE(demand_bytes >= th_demand_bytes(f->V, f->H, f->W), "demand buffer too small (th_map_demand_bytes)");
    int slot = -1;
    for (int k = 0; k < th_ctx::kPrepassSlots; ++k) {
        th_ctx::Prepass& t = c->prepass[k];
        if (t.valid && t.ws == ws && t.rays == (const void*)rays->ray_o && t.R == rays->R && t.S == rays->S) slot = k;
    }
    if (slot < 0) return 1;
    const long long P = (long long)rays->R * rays->S;
    TH_REQUIRE(ws_bytes >= shade_ws_bytes(f, P, rays->R), "workspace too small");
    ThArena ar(ws, ws_bytes);
    const ShadeWs w = carve_shade_ws(f, P, rays->R, ar);
    TH_REQUIRE(w.gws != nullptr, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    th_ctx::Prepass& t = c->prepass[slot];
    t.demand = nullptr;
    c->map_completed = nullptr;
    c->map_completed_demand = nullptr;
    TH_HIP(hipStreamWaitEvent(s, t.ev, 0));                      // the prepass may have run on another stream
    TH_TRY(th_demand_launch(th_src(rays), w.idx, w.info, f->cams, f->scale_xy, f->V, f->H, f->W, verts_paint, n_paint, demand, s));
    t.demand = demand;
    return 0;
}

int th_render_pregather_early(th_ctx* c, const th_frame* f, const th_points* rays, void* ws, size_t ws_bytes, void* pool,
                              size_t pool_bytes, th_stream stream) {
    TH_REQUIRE(c != nullptr, "null argument");
    c->pregather_early = true;
    const int rc = th_render_pregather(c, f, rays, ws, ws_bytes, pool, pool_bytes, stream);
    c->pregather_early = false;
    return rc;
}

int th_render_prepass_cancel(th_ctx* c) {
    TH_REQUIRE(c, "null ctx");
    for (auto& t : c->prepass) t.valid = false;
    return 0;
}

int th_render_prepass_drop(th_ctx* c, const void* ws) {
    TH_REQUIRE(c, "null ctx");
    for (auto& t : c->prepass)
        if (t.ws == ws) t.valid = false;
    return 0;
}

__global__ void extract_sigma_kernel(const float4* __restrict__ raw, const uint8_t* __restrict__ mask, long long P,
                                     float* __restrict__ out) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < P) out[i] = mask[i] ? raw[i].w : 0.f;       // zero outside the hull (raw is only written where shaded)
}

size_t th_sigma_grid_workspace_bytes(const th_frame* f, int P) { return shade_ws_bytes(f, P, P); }

int th_eval_sigma_grid(th_ctx* c, const th_frame* f, const float* pts, int P, float* sigma_out, void* ws,
                       size_t ws_bytes, void* pool, size_t pool_bytes, int64_t* stats_host, th_stream stream) {
    TH_REQUIRE(c && f && pts && sigma_out && ws, "null argument");
    TH_TRY(frame_ok(f));
    hipStream_t s = (hipStream_t)stream;
    if (P <= 0) return 0;
    TH_REQUIRE(ws_bytes >= shade_ws_bytes(f, P, P), "workspace too small");
    ThArena ar(ws, ws_bytes);
    ThPointSrc ps{};
    ps.pts = pts; ps.R = P; ps.S = 1;
    float* raw = nullptr;
    const uint8_t* mask = nullptr;
    TH_TRY(shade_points(c, f, ps, P, false, ar, pool, pool_bytes, &raw, &mask, stats_host, s));
    hipLaunchKernelGGL(extract_sigma_kernel, dim3(th_cdiv(P, 256)), dim3(256), 0, s, (const float4*)raw, mask, (long long)P,
                       sigma_out);
    TH_LAUNCH_CHECK();
    const int snap = th_range_snapshot(c, stream);
    TH_REQUIRE(snap >= 0, "range snapshot failed");
    if (stats_host) stats_host[2] = snap;
    return 0;
}

}  // extern "C"

static ThProf* prof_of(th_ctx* c) {
    if (!c->prof) c->prof = new ThProf();
    return (ThProf*)c->prof;
}
