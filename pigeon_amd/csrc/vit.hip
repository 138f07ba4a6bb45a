// vit.hip -- the pg_vit handle (packed bf16 weights in HBM), the forward-pass orchestration of the ViT-L/14-336
// encoder, and the exported C ABI (include/pigeon_hip.h) for everything except head/refine.
//
// Forward pass for a chunk of n images (M = 577 n rows), all launches asynchronous on the caller's stream:
//   im2col            pixels (n,3,336,336) -> P (576 n, 640) bf16
//   gemm<PATCH>       P x Wpatch^T (+ position)            -> X rows 1..576 of every image (fp32 residual stream)
//   pre_layernorm     class token + position[0], LN in place on X
//   24 x { LN1 -> Xn bf16 ; gemm<QKV> -> QKV bf16 ; attention -> O bf16 (aliases Xn) ; gemm<RESID>(O, Wo) -> X
//          LN2 -> Xn bf16 ; gemm<GELU> -> H bf16 (aliases QKV) ; gemm<RESID>(H, W2) -> X }
//   token_mean        X -> emb (n,1024) fp32
// HBM layout per chunk: X 4 KB/row fp32, Xn/O 2 KB/row bf16, QKV/H/P 8 KB/row bf16  => 14 KB per token row,
// 8.1 MB per image; weights 0.61 GB (16-bit) stay resident.  The residual stream and all LayerNorm / softmax
// statistics are fp32; only MFMA operands are 16-bit: fp16 by default, bf16 with cfg.mma_dtype / PIGEON_MMA_DTYPE=bf16.
// Round 4: everything between im2col and token_mean is replayed from a hipGraph per (workspace, n) key (vit_forward_body_graphed),
// and pg_vit_forward_precise runs the same network in near-fp32 arithmetic for the exact mode (split-fp16 GEMM operands on the same
// MFMA kernels, fp32 attention / LayerNorm / QuickGELU: precise.hip; needs cfg.precise for the 3x weight copy).
#include "common.h"
#include "pigeon_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void pg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int pg_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pg_set_error("launch of %s failed: %s", what, hipGetErrorString(e)); return PG_EHIP; }
    return PG_OK;
}
int pg_default_gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PIGEON_GEMM_VARIANT");
        // 56: persistent ping-pong kernel with 384 x 256 tiles for the QKV / fc1 GEMMs (gemm_pp6.hip), 256 x 256 (variant 36:
        // 8x4 super-tile raster, gemm_pp.hip) for everything else
        v = e ? atoi(e) : 56;
        if (v <= 0) v = 56;
    }
    return v;
}

#ifndef PG_DEFAULT_VIT_STREAMS
#define PG_DEFAULT_VIT_STREAMS 1
#endif
#ifndef PG_DEFAULT_GEMM_STAGGER
#define PG_DEFAULT_GEMM_STAGGER 0.0f
#endif
int pg_gemm_block_cap() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PIGEON_GEMM_BLOCKS"); v = e ? atoi(e) : 0; if (v < 0) v = 0; }
    return v;
}
int pg_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
// Tail split of pg_gemm_launch (gemm_tail.hip).  ROWS: most rows handed to the small-tile kernel instead of giving them a (mostly
// idle) last round of the persistent kernels; 768 = two 384-row panels.  The split is used only where it measured positive on the
// 512-image step (profiles/r02/gemm_tail.txt): K >= 2048 (fc2: a 120 us round saved for a 43 us tail launch, +0.4 % end to end)
// or N >= 4096 (fc1, +0.1 %); for out-projection and QKV the tail launch costs what the half-idle round did (-0.2 % with all four).
#define PG_DEFAULT_GEMM_TAIL_ROWS 768
#define PG_DEFAULT_GEMM_TAIL_MIN_K 2048
#define PG_DEFAULT_GEMM_TAIL_MIN_N 4096
static int g_tail_rows = -1, g_tail_min_k = -1, g_tail_min_n = -1;
// every pg_tune_* call bumps the epoch: the knobs are baked into captured launches, so a hipGraph of an older epoch is re-captured
static unsigned long long g_tune_epoch = 1;
unsigned long long pg_tune_epoch() { return g_tune_epoch; }
static int g_raster_gn = -2;
int pg_gemm_raster_gn() {
    if (g_raster_gn == -2) { const char* e = getenv("PIGEON_GEMM_RASTER_GN"); g_raster_gn = e ? atoi(e) : 0; if (g_raster_gn < -1) g_raster_gn = 0; }
    return g_raster_gn;
}
extern "C" int pg_tune_gemm_raster(int gn) {
    if (gn < -1 || gn > 64) { pg_set_error("tune_gemm_raster: gn must be -1 (all N tiles), 0 (default) or 1..64"); return PG_EINVAL; }
    g_raster_gn = gn; ++g_tune_epoch;
    return PG_OK;
}
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    return v < 0 ? 0 : v;
}
int pg_gemm_tail_rows() { if (g_tail_rows < 0) g_tail_rows = env_int("PIGEON_GEMM_TAIL_ROWS", PG_DEFAULT_GEMM_TAIL_ROWS); return g_tail_rows; }
int pg_gemm_tail_min_k() { if (g_tail_min_k < 0) g_tail_min_k = env_int("PIGEON_GEMM_TAIL_MIN_K", PG_DEFAULT_GEMM_TAIL_MIN_K); return g_tail_min_k; }
int pg_gemm_tail_min_n() { if (g_tail_min_n < 0) g_tail_min_n = env_int("PIGEON_GEMM_TAIL_MIN_N", PG_DEFAULT_GEMM_TAIL_MIN_N); return g_tail_min_n; }
extern "C" int pg_tune_gemm_tail_rows(int rows) {
    if (rows < 0 || rows > (1 << 20)) { pg_set_error("tune_gemm_tail_rows: rows must be in [0, 2^20]"); return PG_EINVAL; }
    g_tail_rows = rows; ++g_tune_epoch;
    return PG_OK;
}
extern "C" int pg_tune_gemm_tail_shape(int min_k, int min_n) {
    if (min_k < 0 || min_n < 0) { pg_set_error("tune_gemm_tail_shape: negative threshold"); return PG_EINVAL; }
    g_tail_min_k = min_k; g_tail_min_n = min_n; ++g_tune_epoch;
    return PG_OK;
}
static int g_gemm_mid = -1;
bool pg_gemm_mid_on() {
    if (g_gemm_mid < 0) { const char* e = getenv("PIGEON_GEMM_MID"); g_gemm_mid = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); }
    return g_gemm_mid != 0;
}
bool pg_gemm_route_pp256() { return pg_gemm_mid_on() && g_gemm_mid != 2; }   // 2 (A/B arm): gemm_mid.hip is the only alternative
extern "C" int pg_tune_gemm_mid(int on) {
    g_gemm_mid = on == 2 ? 2 : (on ? 1 : 0); ++g_tune_epoch;
    return PG_OK;
}
static float g_stagger = -1.f;
float pg_gemm_stagger_fraction() {
    if (g_stagger < 0.f) {
        const char* e = getenv("PIGEON_GEMM_STAGGER");
        g_stagger = e ? (float)atof(e) : PG_DEFAULT_GEMM_STAGGER;
        if (!(g_stagger >= 0.f && g_stagger <= 4.f)) g_stagger = 0.f;
    }
    return g_stagger;
}
extern "C" int pg_tune_gemm_stagger(float fraction) {
    if (!(fraction >= 0.f && fraction <= 4.f)) { pg_set_error("tune_gemm_stagger: fraction must be in [0, 4]"); return PG_EINVAL; }
    g_stagger = fraction; ++g_tune_epoch;
    return PG_OK;
}

extern "C" const char* pg_last_error(void) { return g_err; }
extern "C" int pg_abi_version(void) { return PG_ABI_VERSION; }
extern "C" int pg_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { pg_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return PG_EHIP; }
    return n;
}

// ------------------------------------------------------------------------------------------------ handle
struct LayerW {
    uint16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;     // bf16 [N][K]
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
    float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    // LayerNorm folded into the following GEMM (cfg ln_fold): wqkv / w1 then hold gamma-scaled weights, bqkv / b1 hold
    // beta.W^T + b, and sqkv / s1 the row sums of the ROUNDED folded weights
    float *sqkv = nullptr, *s1 = nullptr;
    // exact mode (cfg.precise): split-fp16 triples [N][3K] = [Wh | Wh | (W - Wh) * 2^8] of the UNFOLDED weights (precise.hip) and the
    // raw fp32 biases of the two GEMMs whose fast-path bias carries the LayerNorm fold
    uint16_t *wqkv3 = nullptr, *wo3 = nullptr, *w13 = nullptr, *w23 = nullptr;
    float *bqkv_raw = nullptr, *b1_raw = nullptr;
};

struct pg_vit {
    pg_vit_cfg cfg;
    int device = 0;
    bool finalized = false;
    bool ln_fold = true;                                   // LayerNorm folded into the GEMMs (env PIGEON_LN_FOLD=0: separate kernels)
    std::map<std::string, std::vector<float>> host;      // staged fp32 parameters until finalize
    std::vector<void*> allocs;
    uint16_t* wpatch = nullptr;                           // [1024][640] bf16 (K zero padded)
    uint16_t* wpatch3 = nullptr;                          // exact mode: [1024][3 * 640] split-fp16 triple
    float *cls = nullptr, *pos = nullptr, *preg = nullptr, *preb = nullptr;
    std::vector<LayerW> layers;
    // two half-batches on two HIP streams (env PIGEON_VIT_STREAMS=2): the tail of a persistent GEMM of one half -- a partial last
    // round with a handful of CUs busy -- is filled by the other half's next kernel instead of idling the chip
    int streams = 1;                                       // 1..4 parts; part 0 runs on the caller's stream
    hipStream_t sx[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // fp16 saturation scan (debug): device counter of 16-bit activations sitting on +-65504
    bool sat_check = false;
    unsigned long long* sat_counter = nullptr;
    // always-on fp16 range alarm (rowstat_finalize_kernel): rows of the residual stream whose sum of squares reaches 65504^2
    unsigned long long* range_alarm = nullptr;
    float range_alarm_sumsq = 0.f;
    // hipGraph of the encoder body (round 4): the ~250 launches between im2col and the token mean touch only the workspace and the
    // weights, so one captured graph per (workspace, n_images) replays them with one host call.  Built at the SECOND forward of a
    // key (the first runs eagerly: it also sets the kernels' LDS attributes, which is not a stream operation), on an internal
    // stream (torch's current stream is usually the legacy default stream, which cannot be captured), launched on the caller's.
    // Off while profiling events / the saturation scan / the multi-stream mode are on, and with env PIGEON_VIT_GRAPH=0.
    struct GraphEntry { const void* ws; int n; int seen; hipGraph_t graph; hipGraphExec_t exec; unsigned long long last_use, epoch; hipEvent_t done; };
    std::vector<GraphEntry> graphs;
    bool use_graph = true;
    hipStream_t capture_stream = nullptr;
    unsigned long long graph_clock = 0;
    long long graph_replays = 0, graph_captures = 0;
    // profiling
    bool prof = false;
    unsigned prof_mask = 0xFFFFFFFFu;                      // classes bracketed while prof is on (bit c = class c)
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> evs;
    int64_t prof_launches[PG_PROF_CLASSES] = {0};
    double prof_ms[PG_PROF_CLASSES] = {0};
};

#ifndef PG_DEFAULT_GEMM_STAGGER
#define PG_DEFAULT_GEMM_STAGGER 0.0f
#endif
static const float kQScale = 0.125f * 1.4426950408889634f;    // head_dim^-0.5 * log2(e)

static std::string canon(const char* name) {
    std::string s(name);
    const std::string pre = "vision_model.";
    if (s.compare(0, pre.size(), pre) == 0) s = s.substr(pre.size());
    return s;
}

extern "C" int pg_vit_create(pg_vit** out, int device, const pg_vit_cfg* cfg) {
    if (!out || !cfg) { pg_set_error("vit_create: null argument"); return PG_EINVAL; }
    if (cfg->image_size != 336 || cfg->patch != 14 || cfg->hidden != 1024 || cfg->heads != 16 || cfg->mlp != 4096 ||
        cfg->layers < 1) {
        pg_set_error("vit_create: only the ViT-L/14-336 geometry is supported (336/14/1024/16/4096, layers>=1)");
        return PG_EINVAL;
    }
    int n = 0;
    PG_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { pg_set_error("vit_create: device %d of %d", device, n); return PG_EINVAL; }
    pg_vit* h = new pg_vit();
    h->cfg = *cfg;
    if (h->cfg.ln_eps <= 0) h->cfg.ln_eps = 1e-5f;
    if (h->cfg.max_chunk <= 0) h->cfg.max_chunk = 512;
    if (h->cfg.mma_dtype == 0) {
        const char* e = getenv("PIGEON_MMA_DTYPE");
        h->cfg.mma_dtype = (e && (!strcmp(e, "bf16") || !strcmp(e, "BF16"))) ? PG_DTYPE_BF16 : PG_DTYPE_F16;
    }
    if (h->cfg.mma_dtype != PG_DTYPE_F16 && h->cfg.mma_dtype != PG_DTYPE_BF16) {
        delete h;
        pg_set_error("vit_create: mma_dtype must be 0 (default), PG_DTYPE_F16 or PG_DTYPE_BF16");
        return PG_EINVAL;
    }
    h->device = device;
    // LayerNorm folded into the next GEMM (gamma into the weights, per-row (rstd, mean*rstd) applied in the epilogue): removes the
    // two 0.31 ms LayerNorm launches per layer (1.8 GB of HBM traffic each) for heavier epilogues.  First measured SLOWER
    // (231.5 vs 229.6 ms per step); with the packed 16-bit conversions, the early residual fetch and the C = 0 first k-step
    // it is 1.3 % faster in the same run (2354 vs 2324 img/s) at the same embedding error (2.7e-4), so it is the default.
    // PIGEON_LN_FOLD=0 selects the separate-LayerNorm chain (kept as the A/B arm and for the non-persistent GEMM variants).
    { const char* e = getenv("PIGEON_LN_FOLD"); h->ln_fold = !(e && e[0] == '0'); }
    if (pg_default_gemm_variant() < 30) h->ln_fold = false;   // the folded epilogues exist only in the persistent GEMM
    h->layers.resize(cfg->layers);
    if (h->ln_fold && h->cfg.mma_dtype == PG_DTYPE_F16) {   // the always-on range alarm of the fp16 residual copies
        PG_HIP(hipSetDevice(device));
        PG_HIP(hipMalloc((void**)&h->range_alarm, sizeof(unsigned long long)));
        h->allocs.push_back(h->range_alarm);
        PG_HIP(hipMemset(h->range_alarm, 0, sizeof(unsigned long long)));
        h->range_alarm_sumsq = 65504.0f * 65504.0f;
    }
    { const char* e = getenv("PIGEON_VIT_GRAPH"); h->use_graph = !(e && e[0] == '0'); }
    { const char* e = getenv("PIGEON_VIT_STREAMS"); h->streams = e ? atoi(e) : PG_DEFAULT_VIT_STREAMS; }
    if (h->streams < 1 || h->streams > 4) h->streams = 1;
    if (h->streams > 1) {
        PG_HIP(hipSetDevice(device));
        PG_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int i = 0; i + 1 < h->streams; ++i) {
            PG_HIP(hipStreamCreateWithFlags(&h->sx[i], hipStreamNonBlocking));
            PG_HIP(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
    }
    *out = h;
    return PG_OK;
}

extern "C" int pg_vit_load_weight(pg_vit* h, const char* name, const void* data, int dtype, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape) { pg_set_error("vit_load_weight: null argument"); return PG_EINVAL; }
    if (dtype != PG_DTYPE_F32) { pg_set_error("vit_load_weight: only fp32 host data is accepted"); return PG_EINVAL; }
    if (h->finalized) { pg_set_error("vit_load_weight: handle already finalized"); return PG_ESTATE; }
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    std::string key = canon(name);
    const float* f = (const float*)data;
    h->host[key] = std::vector<float>(f, f + n);
    return PG_OK;
}

static uint16_t host_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static uint16_t host_f16(float f) {                       // round to nearest even, saturating
    if (f > 65504.f) f = 65504.f;
    if (f < -65504.f) f = -65504.f;
    _Float16 h = (_Float16)f;
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}
static uint16_t host_cvt(int dtype, float f) { return dtype == PG_DTYPE_F16 ? host_f16(f) : host_bf16(f); }
static float host_uncvt(int dtype, uint16_t b) {
    if (dtype == PG_DTYPE_F16) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
    uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f;
}
// W' = cvt(gamma o W) [N][K]; colsum[n] = sum_k float(W'[n][k]) (of the ROUNDED values, so that the identity
// LN(x).W^T = rstd (x.W'^T - mean colsum) + beta.W^T holds exactly for what the MFMA multiplies); cbias = beta.W^T + b.
static void fold_ln(int dt, const float* W, const float* b, const float* gamma, const float* beta, size_t N, size_t K,
                    std::vector<uint16_t>& w16, std::vector<float>& colsum, std::vector<float>& cbias) {
    w16.resize(N * K); colsum.resize(N); cbias.resize(N);
    for (size_t n = 0; n < N; ++n) {
        double s = 0.0, c = 0.0;
        for (size_t k = 0; k < K; ++k) {
            const uint16_t q = host_cvt(dt, W[n * K + k] * gamma[k]);
            w16[n * K + k] = q;
            s += (double)host_uncvt(dt, q);
            c += (double)beta[k] * (double)W[n * K + k];
        }
        colsum[n] = (float)s;
        cbias[n] = (float)(c + (double)b[n]);
    }
}

// exact mode (precise.hip): row n of W [N][K] -> [Wh | Wh | fp16((W - Wh) * 2^8)], K zero padded to Kpad
static void pack_x3(const float* W, size_t N, size_t K, size_t Kpad, std::vector<uint16_t>& w3) {
    w3.assign(N * 3 * Kpad, 0);
    for (size_t n = 0; n < N; ++n) {
        uint16_t* r = w3.data() + n * 3 * Kpad;
        for (size_t k = 0; k < K; ++k) {
            const float w = W[n * K + k];
            const uint16_t hb = host_f16(w);
            const float hf = host_uncvt(PG_DTYPE_F16, hb);
            r[k] = hb; r[Kpad + k] = hb;
            r[2 * Kpad + k] = host_f16((w - hf) * 256.0f);
        }
    }
}

static int need(pg_vit* h, const std::string& k, size_t n, const std::vector<float>** out) {
    auto it = h->host.find(k);
    if (it == h->host.end()) { pg_set_error("vit_finalize: missing parameter %s", k.c_str()); return PG_ESTATE; }
    if (it->second.size() != n) {
        pg_set_error("vit_finalize: parameter %s has %zu elements, expected %zu", k.c_str(), it->second.size(), n);
        return PG_EINVAL;
    }
    *out = &it->second;
    return PG_OK;
}

static int upload_f32(pg_vit* h, const float* src, size_t n, float** dst) {
    PG_HIP(hipMalloc((void**)dst, n * sizeof(float)));
    h->allocs.push_back(*dst);
    PG_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return PG_OK;
}
static int upload_bf16(pg_vit* h, const std::vector<uint16_t>& src, uint16_t** dst) {
    PG_HIP(hipMalloc((void**)dst, src.size() * 2));
    h->allocs.push_back(*dst);
    PG_HIP(hipMemcpy(*dst, src.data(), src.size() * 2, hipMemcpyHostToDevice));
    return PG_OK;
}

#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

extern "C" int pg_vit_finalize(pg_vit* h) {
    if (!h) { pg_set_error("vit_finalize: null handle"); return PG_EINVAL; }
    if (h->finalized) return PG_OK;
    PG_HIP(hipSetDevice(h->device));
    const size_t D = VIT_HIDDEN, F = VIT_MLP;
    const int dt = h->cfg.mma_dtype;
    const std::vector<float>* p;
    // patch embedding [1024,3,14,14] -> bf16 [1024][640], zero padded along K
    RC(need(h, "embeddings.patch_embedding.weight", D * VIT_PATCH_K, &p));
    {
        std::vector<uint16_t> w(D * VIT_PATCH_KPAD, 0);
        for (size_t n = 0; n < D; ++n)
            for (size_t k = 0; k < VIT_PATCH_K; ++k) w[n * VIT_PATCH_KPAD + k] = host_cvt(dt, (*p)[n * VIT_PATCH_K + k]);
        RC(upload_bf16(h, w, &h->wpatch));
        if (h->cfg.precise) { std::vector<uint16_t> w3; pack_x3(p->data(), D, VIT_PATCH_K, VIT_PATCH_KPAD, w3); RC(upload_bf16(h, w3, &h->wpatch3)); }
    }
    RC(need(h, "embeddings.class_embedding", D, &p));                 RC(upload_f32(h, p->data(), D, &h->cls));
    RC(need(h, "embeddings.position_embedding.weight", VIT_TOKENS * D, &p)); RC(upload_f32(h, p->data(), VIT_TOKENS * D, &h->pos));
    RC(need(h, "pre_layrnorm.weight", D, &p));                        RC(upload_f32(h, p->data(), D, &h->preg));
    RC(need(h, "pre_layrnorm.bias", D, &p));                          RC(upload_f32(h, p->data(), D, &h->preb));
    for (int l = 0; l < h->cfg.layers; ++l) {
        LayerW& L = h->layers[l];
        const std::string pre = "encoder.layers." + std::to_string(l) + ".";
        const std::vector<float>*wq, *wk, *wv, *bq, *bk, *bv;
        RC(need(h, pre + "self_attn.q_proj.weight", D * D, &wq)); RC(need(h, pre + "self_attn.k_proj.weight", D * D, &wk));
        RC(need(h, pre + "self_attn.v_proj.weight", D * D, &wv)); RC(need(h, pre + "self_attn.q_proj.bias", D, &bq));
        RC(need(h, pre + "self_attn.k_proj.bias", D, &bk));       RC(need(h, pre + "self_attn.v_proj.bias", D, &bv));
        const std::vector<float>*g1, *be1, *g2, *be2;
        RC(need(h, pre + "layer_norm1.weight", D, &g1)); RC(need(h, pre + "layer_norm1.bias", D, &be1));
        RC(need(h, pre + "layer_norm2.weight", D, &g2)); RC(need(h, pre + "layer_norm2.bias", D, &be2));
        {
            std::vector<float> wf(3 * D * D), b(3 * D);
            for (size_t i = 0; i < D * D; ++i) { wf[i] = (*wq)[i]; wf[D * D + i] = (*wk)[i]; wf[2 * D * D + i] = (*wv)[i]; }
            for (size_t i = 0; i < D; ++i) { b[i] = (*bq)[i]; b[D + i] = (*bk)[i]; b[2 * D + i] = (*bv)[i]; }
            if (h->cfg.precise) {
                std::vector<uint16_t> w3; pack_x3(wf.data(), 3 * D, D, D, w3);
                RC(upload_bf16(h, w3, &L.wqkv3));
                RC(upload_f32(h, b.data(), 3 * D, &L.bqkv_raw));
            }
            if (h->ln_fold) {
                std::vector<uint16_t> w; std::vector<float> cs, cb;
                fold_ln(dt, wf.data(), b.data(), g1->data(), be1->data(), 3 * D, D, w, cs, cb);
                RC(upload_bf16(h, w, &L.wqkv));
                RC(upload_f32(h, cb.data(), 3 * D, &L.bqkv));
                RC(upload_f32(h, cs.data(), 3 * D, &L.sqkv));
            } else {
                std::vector<uint16_t> w(3 * D * D);
                for (size_t i = 0; i < 3 * D * D; ++i) w[i] = host_cvt(dt, wf[i]);
                RC(upload_bf16(h, w, &L.wqkv));
                RC(upload_f32(h, b.data(), 3 * D, &L.bqkv));
            }
        }
        auto up_w = [&](const std::string& k, size_t n, uint16_t** dst) -> int {
            const std::vector<float>* q;
            RC(need(h, k, n, &q));
            std::vector<uint16_t> w(n);
            for (size_t i = 0; i < n; ++i) w[i] = host_cvt(dt, (*q)[i]);
            return upload_bf16(h, w, dst);
        };
        auto up_f = [&](const std::string& k, size_t n, float** dst) -> int {
            const std::vector<float>* q;
            RC(need(h, k, n, &q));
            return upload_f32(h, q->data(), n, dst);
        };
        RC(up_w(pre + "self_attn.out_proj.weight", D * D, &L.wo)); RC(up_f(pre + "self_attn.out_proj.bias", D, &L.bo));
        if (h->ln_fold) {
            const std::vector<float>*w1f, *b1f;
            RC(need(h, pre + "mlp.fc1.weight", F * D, &w1f)); RC(need(h, pre + "mlp.fc1.bias", F, &b1f));
            std::vector<uint16_t> w; std::vector<float> cs, cb;
            fold_ln(dt, w1f->data(), b1f->data(), g2->data(), be2->data(), F, D, w, cs, cb);
            RC(upload_bf16(h, w, &L.w1));
            RC(upload_f32(h, cb.data(), F, &L.b1));
            RC(upload_f32(h, cs.data(), F, &L.s1));
        } else {
            RC(up_w(pre + "mlp.fc1.weight", F * D, &L.w1));        RC(up_f(pre + "mlp.fc1.bias", F, &L.b1));
        }
        RC(up_w(pre + "mlp.fc2.weight", D * F, &L.w2));            RC(up_f(pre + "mlp.fc2.bias", D, &L.b2));
        if (h->cfg.precise) {
            auto up_w3 = [&](const std::string& k, size_t N, size_t K, uint16_t** dst) -> int {
                const std::vector<float>* q;
                RC(need(h, k, N * K, &q));
                std::vector<uint16_t> w3; pack_x3(q->data(), N, K, K, w3);
                return upload_bf16(h, w3, dst);
            };
            RC(up_w3(pre + "self_attn.out_proj.weight", D, D, &L.wo3));
            RC(up_w3(pre + "mlp.fc1.weight", F, D, &L.w13));
            RC(up_w3(pre + "mlp.fc2.weight", D, F, &L.w23));
            RC(up_f(pre + "mlp.fc1.bias", F, &L.b1_raw));
        }
        RC(up_f(pre + "layer_norm1.weight", D, &L.ln1g));          RC(up_f(pre + "layer_norm1.bias", D, &L.ln1b));
        RC(up_f(pre + "layer_norm2.weight", D, &L.ln2g));          RC(up_f(pre + "layer_norm2.bias", D, &L.ln2b));
    }
    h->host.clear();
    h->finalized = true;
    return PG_OK;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// X fp32 | Xn 16-bit | big 16-bit | (LN fold:) Xn2 16-bit | statpart [M][16][2] fp32 | rowstat A, B [M][2] fp32
static size_t ws_bytes_for(int chunk, bool ln_fold) {
    const size_t M = (size_t)chunk * VIT_TOKENS;
    size_t b = align_up(M * VIT_HIDDEN * 4, 256) + align_up(M * VIT_HIDDEN * 2, 256) + align_up(M * VIT_MLP * 2, 256) + 256;
    if (ln_fold) b += align_up(M * VIT_HIDDEN * 2, 256) + align_up(M * (VIT_HIDDEN / 64) * 8, 256) + 2 * align_up(M * 8, 256);
    return b;
}

// two-stream mode: the batch is cut in two halves, each with a workspace of its own
static int split_parts(const pg_vit* h, int n_images) { return (h->streams > 1 && n_images >= 32 * h->streams) ? h->streams : 1; }
extern "C" int pg_vit_workspace_bytes(const pg_vit* h, int n_images, size_t* bytes) {
    if (!h || !bytes || n_images < 0) { pg_set_error("vit_workspace_bytes: bad argument"); return PG_EINVAL; }
    const int parts = split_parts(h, n_images);
    if (parts > 1) {
        const int per = (n_images + parts - 1) / parts;
        const int chunk = per < h->cfg.max_chunk ? per : h->cfg.max_chunk;
        *bytes = (size_t)parts * align_up(ws_bytes_for(chunk, h->ln_fold), 256);
        return PG_OK;
    }
    const int chunk = n_images < h->cfg.max_chunk ? n_images : h->cfg.max_chunk;
    *bytes = ws_bytes_for(chunk > 0 ? chunk : 1, h->ln_fold);
    return PG_OK;
}

struct ProfScope {
    pg_vit* h; hipStream_t s; int cls; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(pg_vit* h_, hipStream_t s_, int c) : h(h_), s(s_), cls(c) {
        if (h->prof && ((h->prof_mask >> c) & 1u)) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~ProfScope() {
        if (a) { (void)hipEventRecord(b, s); h->evs.push_back({a, b, cls}); }
    }
};

#define SAT(buf, rows, cols, ld) do { if (h->sat_check) RC(pg_count_sat16_launch((buf), (rows), (cols), (ld), dt, h->sat_counter, s)); } while (0)

// everything between the im2col and the token mean: reads / writes the workspace and the handle's weights only
static int vit_forward_body(pg_vit* h, int n, char* ws, hipStream_t s) {
    const int64_t M = (int64_t)n * VIT_TOKENS;
    float* X = (float*)ws;
    uint16_t* Xn = (uint16_t*)(ws + align_up((size_t)M * VIT_HIDDEN * 4, 256));
    uint16_t* big = (uint16_t*)((char*)Xn + align_up((size_t)M * VIT_HIDDEN * 2, 256));
    const float eps = h->cfg.ln_eps;
    const int dt = h->cfg.mma_dtype;
    SAT(big, (int64_t)n * VIT_PATCHES, VIT_PATCH_KPAD, VIT_PATCH_KPAD);
    { ProfScope p(h, s, 4);
      RC(pg_gemm_launch(dt, big, VIT_PATCH_KPAD, h->wpatch, VIT_PATCH_KPAD, nullptr, X, VIT_HIDDEN, n * VIT_PATCHES, VIT_HIDDEN, VIT_PATCH_KPAD,
                        EPI_PATCH, 1.f, 0, h->pos, 0, s)); }
    if (!h->ln_fold) { ProfScope p(h, s, 6); RC(pg_preln_launch(X, h->cls, h->pos, h->preg, h->preb, M, eps, s)); }
    if (h->ln_fold) {
        // LayerNorm folded into the GEMM that consumes it: the residual GEMMs (out_proj, fc2) emit, next to the fp32
        // residual row, its 16-bit copy and per-slice partial (sum, sum of squares); a per-row finalize turns those into
        // (rstd, mean*rstd); the QKV / fc1 GEMMs multiply the RAW 16-bit row with gamma-scaled weights and apply
        // rstd * acc - mean*rstd * colsum + (beta.W^T + b) in their epilogue.  No stand-alone LayerNorm launch in the
        // layer loop (two 6 KB/row streaming passes per layer gone); tools/precision_sim-style emulation and the golden
        // tests show the same embedding error as the unfused order (the rounding point moves from LN(x) to x).
        uint16_t* Xn2 = (uint16_t*)((char*)big + align_up((size_t)M * VIT_MLP * 2, 256));
        float* statpart = (float*)((char*)Xn2 + align_up((size_t)M * VIT_HIDDEN * 2, 256));
        float* rsA = (float*)((char*)statpart + align_up((size_t)M * (VIT_HIDDEN / 64) * 8, 256));
        float* rsB = (float*)((char*)rsA + align_up((size_t)M * 8, 256));
        const int slots = VIT_HIDDEN / 64;
        // class token + position + pre_layrnorm, and in the same pass the 16-bit copy + row statistics layer 0's folded LN1 needs
        { ProfScope p(h, s, 6); RC(pg_preln_launch(X, h->cls, h->pos, h->preg, h->preb, M, eps, s, Xn, dt, rsA)); }
        SAT(Xn, M, VIT_HIDDEN, VIT_HIDDEN);
        for (int l = 0; l < h->cfg.layers; ++l) {
            const LayerW& L = h->layers[l];
            const bool last = l + 1 == h->cfg.layers;
            PgGemmExtra ex;
            { ProfScope p(h, s, 0);
              ex = PgGemmExtra(); ex.colsum = L.sqkv; ex.rowstat = rsA;
              RC(pg_gemm_launch(dt, Xn, VIT_HIDDEN, L.wqkv, VIT_HIDDEN, L.bqkv, big, 3 * VIT_HIDDEN, (int)M, 3 * VIT_HIDDEN, VIT_HIDDEN,
                                EPI_QKV_LN, kQScale, VIT_HIDDEN, nullptr, 0, s, &ex)); }
            SAT(big, M, 3 * VIT_HIDDEN, 3 * VIT_HIDDEN);
            { ProfScope p(h, s, 5); RC(pg_attention_launch(dt, big, Xn, n, s)); }
            SAT(Xn, M, VIT_HIDDEN, VIT_HIDDEN);
            { ProfScope p(h, s, 1);
              ex = PgGemmExtra(); ex.x16 = Xn2; ex.ldx = VIT_HIDDEN; ex.statpart = statpart;
              RC(pg_gemm_launch(dt, Xn, VIT_HIDDEN, L.wo, VIT_HIDDEN, L.bo, X, VIT_HIDDEN, (int)M, VIT_HIDDEN, VIT_HIDDEN, EPI_RESID_STAT,
                                1.f, 0, nullptr, 0, s, &ex)); }
            SAT(Xn2, M, VIT_HIDDEN, VIT_HIDDEN);
            { ProfScope p(h, s, 6); RC(pg_rowstat_finalize_launch(statpart, slots, rsB, M, eps, s, h->range_alarm, h->range_alarm_sumsq)); }
            { ProfScope p(h, s, 2);
              ex = PgGemmExtra(); ex.colsum = L.s1; ex.rowstat = rsB;
              RC(pg_gemm_launch(dt, Xn2, VIT_HIDDEN, L.w1, VIT_HIDDEN, L.b1, big, VIT_MLP, (int)M, VIT_MLP, VIT_HIDDEN, EPI_GELU_LN, 1.f, 0,
                                nullptr, 0, s, &ex)); }
            SAT(big, M, VIT_MLP, VIT_MLP);
            { ProfScope p(h, s, 3);
              if (last) {
                  RC(pg_gemm_launch(dt, big, VIT_MLP, L.w2, VIT_MLP, L.b2, X, VIT_HIDDEN, (int)M, VIT_HIDDEN, VIT_MLP, EPI_RESID, 1.f, 0, nullptr, 0, s));
              } else {
                  ex = PgGemmExtra(); ex.x16 = Xn; ex.ldx = VIT_HIDDEN; ex.statpart = statpart;
                  RC(pg_gemm_launch(dt, big, VIT_MLP, L.w2, VIT_MLP, L.b2, X, VIT_HIDDEN, (int)M, VIT_HIDDEN, VIT_MLP, EPI_RESID_STAT, 1.f, 0,
                                    nullptr, 0, s, &ex));
              } }
            if (!last) SAT(Xn, M, VIT_HIDDEN, VIT_HIDDEN);
            if (!last) { ProfScope p(h, s, 6); RC(pg_rowstat_finalize_launch(statpart, slots, rsA, M, eps, s, h->range_alarm, h->range_alarm_sumsq)); }
        }
    } else
    for (int l = 0; l < h->cfg.layers; ++l) {
        const LayerW& L = h->layers[l];
        { ProfScope p(h, s, 6); RC(pg_layernorm_launch(X, L.ln1g, L.ln1b, Xn, dt, M, eps, s)); }
        SAT(Xn, M, VIT_HIDDEN, VIT_HIDDEN);
        { ProfScope p(h, s, 0);
          RC(pg_gemm_launch(dt, Xn, VIT_HIDDEN, L.wqkv, VIT_HIDDEN, L.bqkv, big, 3 * VIT_HIDDEN, (int)M, 3 * VIT_HIDDEN, VIT_HIDDEN, EPI_QKV,
                            kQScale, VIT_HIDDEN, nullptr, 0, s)); }
        SAT(big, M, 3 * VIT_HIDDEN, 3 * VIT_HIDDEN);
        { ProfScope p(h, s, 5); RC(pg_attention_launch(dt, big, Xn, n, s)); }
        SAT(Xn, M, VIT_HIDDEN, VIT_HIDDEN);
        { ProfScope p(h, s, 1);
          RC(pg_gemm_launch(dt, Xn, VIT_HIDDEN, L.wo, VIT_HIDDEN, L.bo, X, VIT_HIDDEN, (int)M, VIT_HIDDEN, VIT_HIDDEN, EPI_RESID, 1.f, 0, nullptr, 0, s)); }
        { ProfScope p(h, s, 6); RC(pg_layernorm_launch(X, L.ln2g, L.ln2b, Xn, dt, M, eps, s)); }
        SAT(Xn, M, VIT_HIDDEN, VIT_HIDDEN);
        { ProfScope p(h, s, 2);
          RC(pg_gemm_launch(dt, Xn, VIT_HIDDEN, L.w1, VIT_HIDDEN, L.b1, big, VIT_MLP, (int)M, VIT_MLP, VIT_HIDDEN, EPI_GELU, 1.f, 0, nullptr, 0, s)); }
        SAT(big, M, VIT_MLP, VIT_MLP);
        { ProfScope p(h, s, 3);
          RC(pg_gemm_launch(dt, big, VIT_MLP, L.w2, VIT_MLP, L.b2, X, VIT_HIDDEN, (int)M, VIT_HIDDEN, VIT_MLP, EPI_RESID, 1.f, 0, nullptr, 0, s)); }
    }
    return PG_OK;
}

// An exec may still be running on the caller's stream when its entry is evicted or found stale: wait for the event recorded behind
// its last launch before destroying it.
static void graph_entry_free(pg_vit::GraphEntry& e) {
    if (e.done) { (void)hipEventSynchronize(e.done); (void)hipEventDestroy(e.done); }
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (e.graph) (void)hipGraphDestroy(e.graph);
    e.exec = nullptr; e.graph = nullptr; e.done = nullptr;
}
// Run the body through its captured graph if there is one for (ws, n); capture it at the second sight of the key; else eagerly.
static int vit_forward_body_graphed(pg_vit* h, int n, char* ws, hipStream_t s) {
    const bool ok = h->use_graph && !h->prof && !h->sat_check && h->streams == 1;
    if (!ok) return vit_forward_body(h, n, ws, s);
    pg_vit::GraphEntry* ent = nullptr;
    for (auto& e : h->graphs) if (e.ws == ws && e.n == n) { ent = &e; break; }
    if (!ent) {
        if (h->graphs.size() >= 8) {                          // evict the least recently used key
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
            graph_entry_free(h->graphs[lru]);
            h->graphs.erase(h->graphs.begin() + lru);
        }
        h->graphs.push_back({ws, n, 0, nullptr, nullptr, 0, pg_tune_epoch(), nullptr});
        ent = &h->graphs.back();
    }
    ent->last_use = ++h->graph_clock;
    if (ent->exec && ent->epoch != pg_tune_epoch()) {         // a pg_tune_* call since the capture: the baked-in knobs are stale
        graph_entry_free(*ent);
        ent->seen = 1;                                         // kernel attributes are set already: capture right away
    }
    if (ent->exec) {
        PG_HIP(hipGraphLaunch(ent->exec, s));
        if (ent->done) (void)hipEventRecord(ent->done, s);
        ++h->graph_replays;
        return PG_OK;
    }
    if (ent->seen++ == 0) return vit_forward_body(h, n, ws, s);      // first sight: eager (sets kernel attributes, warms caches)
    if (!h->capture_stream) PG_HIP(hipStreamCreateWithFlags(&h->capture_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { (void)hipGetLastError(); h->use_graph = false; return vit_forward_body(h, n, ws, s); }
    const int rc = vit_forward_body(h, n, ws, h->capture_stream);
    e = hipStreamEndCapture(h->capture_stream, &graph);
    if (rc != PG_OK || e != hipSuccess || !graph) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        h->use_graph = false;                                  // capture is not available here: stay eager from now on
        return vit_forward_body(h, n, ws, s);
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess || !exec) { (void)hipGetLastError(); (void)hipGraphDestroy(graph); h->use_graph = false; return vit_forward_body(h, n, ws, s); }
    ent->graph = graph; ent->exec = exec; ent->epoch = pg_tune_epoch();
    if (!ent->done) (void)hipEventCreateWithFlags(&ent->done, hipEventDisableTiming);
    ++h->graph_captures;
    PG_HIP(hipGraphLaunch(exec, s));
    if (ent->done) (void)hipEventRecord(ent->done, s);
    ++h->graph_replays;
    return PG_OK;
}

static int vit_forward_chunk(pg_vit* h, const void* pixels, int pix_dtype, int n, float* emb_out, float* hidden_out,
                             char* ws, hipStream_t s) {
    const int64_t M = (int64_t)n * VIT_TOKENS;
    float* X = (float*)ws;
    uint16_t* Xn = (uint16_t*)(ws + align_up((size_t)M * VIT_HIDDEN * 4, 256));
    uint16_t* big = (uint16_t*)((char*)Xn + align_up((size_t)M * VIT_HIDDEN * 2, 256));
    { ProfScope p(h, s, 7); RC(pg_im2col_launch(pixels, pix_dtype, big, h->cfg.mma_dtype, n, s)); }
    RC(vit_forward_body_graphed(h, n, ws, s));
    { ProfScope p(h, s, 8); RC(pg_token_mean_launch(X, emb_out, n, s)); }
    if (hidden_out) PG_HIP(hipMemcpyAsync(hidden_out, X, (size_t)M * VIT_HIDDEN * 4, hipMemcpyDeviceToDevice, s));
    return PG_OK;
}

extern "C" int pg_vit_forward_hidden(pg_vit* h, const void* pixels, int pix_dtype, int n_images, float* emb_out,
                                     float* hidden_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h) { pg_set_error("vit_forward: null handle"); return PG_EINVAL; }
    if (!h->finalized) { pg_set_error("vit_forward: handle not finalized"); return PG_ESTATE; }
    if (n_images < 0) { pg_set_error("vit_forward: n_images = %d", n_images); return PG_EINVAL; }
    if (n_images == 0) return PG_OK;                       // an empty batch is a no-op: its (empty) buffers may be NULL
    if (!pixels || !emb_out || !workspace) { pg_set_error("vit_forward: null argument"); return PG_EINVAL; }
    if (pix_dtype != PG_DTYPE_F32 && pix_dtype != PG_DTYPE_BF16 && pix_dtype != PG_DTYPE_F16) { pg_set_error("vit_forward: bad pixel dtype"); return PG_EINVAL; }
    size_t needb = 0;
    pg_vit_workspace_bytes(h, n_images, &needb);
    if (workspace_bytes < needb) { pg_set_error("vit_forward: workspace %zu < required %zu bytes", workspace_bytes, needb); return PG_ENOMEM; }
    if (((uintptr_t)workspace & 255) != 0) { pg_set_error("vit_forward: workspace must be 256-byte aligned"); return PG_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = pix_dtype == PG_DTYPE_F32 ? 4 : 2;
    const size_t img_elems = (size_t)3 * VIT_IMG * VIT_IMG;
    auto run_range = [&](int first, int last, char* ws, hipStream_t st) -> int {
        for (int s0 = first; s0 < last; s0 += h->cfg.max_chunk) {
            const int n = (last - s0) < h->cfg.max_chunk ? (last - s0) : h->cfg.max_chunk;
            RC(vit_forward_chunk(h, (const char*)pixels + (size_t)s0 * img_elems * esz, pix_dtype, n,
                                 emb_out + (size_t)s0 * VIT_HIDDEN,
                                 hidden_out ? hidden_out + (size_t)s0 * VIT_TOKENS * VIT_HIDDEN : nullptr, ws, st));
        }
        return PG_OK;
    };
    const int parts = split_parts(h, n_images);
    if (parts > 1) {
        const int per = (n_images + parts - 1) / parts;
        const size_t wsp = needb / parts;
        PG_HIP(hipEventRecord(h->ev_fork, s));               // the side streams start where the caller's stream stands
        for (int i = 0; i + 1 < parts; ++i) PG_HIP(hipStreamWaitEvent(h->sx[i], h->ev_fork, 0));
        int rc = PG_OK;
        for (int i = 0; i < parts && rc == PG_OK; ++i) {
            const int first = i * per, last = (i + 1) * per < n_images ? (i + 1) * per : n_images;
            if (first >= last) continue;
            rc = run_range(first, last, (char*)workspace + (size_t)i * wsp, i == 0 ? s : h->sx[i - 1]);
        }
        // ... and the caller's stream continues when every part is done.  Also after a failed launch in the middle: work already
        // queued on the side streams still writes the caller's buffers, so the caller's stream must not run ahead of it (the
        // first error code is what is returned; the join's own errors only matter if there was none).
        for (int i = 0; i + 1 < parts; ++i) {
            hipError_t e = hipEventRecord(h->ev_join[i], h->sx[i]);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, h->ev_join[i], 0);
            if (e != hipSuccess && rc == PG_OK) { pg_set_error("vit_forward: joining side stream %d failed: %s", i, hipGetErrorString(e)); rc = PG_EHIP; }
        }
        return rc;
    }
    return run_range(0, n_images, (char*)workspace, s);
}

extern "C" int pg_vit_forward(pg_vit* h, const void* pixels, int pix_dtype, int n_images, float* emb_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
    return pg_vit_forward_hidden(h, pixels, pix_dtype, n_images, emb_out, nullptr, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ exact mode
// pg_vit_forward_precise: the same encoder in near-fp32 arithmetic (precise.hip: split-fp16 GEMM operands on the persistent MFMA
// kernels, fp32 LayerNorm / attention / QuickGELU / residual), for the panoramas whose top-1 margin the 16-bit path cannot decide.
// Workspace per token row: X fp32 4 KB | T3 triple of a 1024-wide row 6 KB | F fp32 QKV (12 KB) / fc1 (16 KB) | G3 triple of the
// 4096-wide activation 24 KB (the im2col triple and the fp32 attention output live there too) | (round 5) PP, the K-split partial
// products of one GEMM, 48 KB (fc1: 3 x 4096 fp32) = 98 KB; equal chunks of <= 128 images.
//
// Round 5: a GEMM of the layer loop whose tiles would not fill the chip is cut along K' into S parts (the products hi.Wh | lo.Wh |
// hi.Wl, or halves of them) that run as ONE persistent launch over S x tilesM x tilesN tiles (PgGemmExtra::parts) into fp32 partial
// buffers, which sum_parts_kernel then adds in a fixed order.  The exact tier runs on the handful of panoramas a step finds
// uncertain -- 4 to 16 images, 10 to 40 row panels: a K' = 12288 launch then puts 40 tiles of 192 K tiles each on 256 CUs (326 us per
// layer for fc2 alone: latency, not work); 240 tiles of 32 K tiles fill the chip instead.  S is chosen per shape by a small cost
// model (precise_parts).  (First tried as S concurrent launches on side streams: the launches did not overlap -- 4 images 15.9
// against 14.3 ms, 52 images 112 against 76 ms, gpurun_out/r05/exact_small_batches_ksplit.txt -- hence the in-kernel form.)
#define PG_PRECISE_CHUNK 128   // images per internal pass (98 KB of workspace per token row: 7.2 GB at 128); a longer batch is cut into EQUAL chunks
// pg_tune_exact_products (round 6): how many of the three partial products hi.Wh | lo.Wh | hi.Wl the exact mode's weight GEMMs run.
// 3 = all (the exact tier).  2 = the first two: the activations keep both halves, the weights only their fp16 value -- K' = 2K on the
// SAME triple operands (the third K-third of A' / W' is simply not visited).  What is left is the weights' rounding, identical for
// every token of every image: a systematic embedding error that a calibration can measure (pigeon_amd/certainty.py), at 2/3 of the
// exact tier's GEMM work.  The attention (activation x activation) keeps its three products.
static int g_exact_products = 3;
extern "C" int pg_tune_exact_products(int n) {
    if (n != 2 && n != 3) { pg_set_error("pg_tune_exact_products: 2 or 3"); return PG_EINVAL; }
    g_exact_products = n;
    return PG_OK;
}
static size_t precise_ws_bytes_for(int chunk) {
    const size_t M = (size_t)chunk * VIT_TOKENS;
    return align_up(M * VIT_HIDDEN * 4, 256) + align_up(M * 3 * VIT_HIDDEN * 2, 256) + align_up(M * VIT_MLP * 4, 256) +
           align_up(M * 3 * VIT_MLP * 2, 256) + align_up(M * 3 * VIT_MLP * 4, 256) + 256;
}
extern "C" int pg_vit_precise_workspace_bytes(const pg_vit* h, int n_images, size_t* bytes) {
    if (!h || !bytes || n_images < 0) { pg_set_error("vit_precise_workspace_bytes: bad argument"); return PG_EINVAL; }
    const int chunk = n_images < PG_PRECISE_CHUNK ? n_images : PG_PRECISE_CHUNK;
    *bytes = precise_ws_bytes_for(chunk > 0 ? chunk : 1);
    return PG_OK;
}

// How many K-parts for one GEMM of the exact mode: the S (out of `cand`) with the smallest modelled time on `ncu` CUs --
//   rounds(S) x (K tiles per part x 1.7 us + epilogue) + the sum pass (S + 1 or S + 2 streams of M x N floats at ~4 TB/s).
// A pure function of the shape: the same batch size always takes the same path (results differ between S only in fp32 summation
// order, ~1e-7 relative, two orders below the exact tier's own floor).  Measured on MI355X (gpurun_out/r05/exact8_kernel_stats.csv,
// 8 images): out-projection 107 -> 54 us, fc2 351 -> 158 us with S = 3 / 6; QKV and fc1 (216 / 288 tiles of 48 K tiles) gain nothing
// from a split at that size and lose at larger ones, which the model reproduces.
static int precise_parts(int M, int N, int Ktot, bool resid, const int* cand, int ncand, double* best_us_out = nullptr) {
    const double ncu = (double)pg_num_cus();
    const double tiles = (double)((M + 255) / 256) * (N / 256);
    int best = 1; double best_us = 1e30;
    for (int i = 0; i < ncand; ++i) {
        const int S = cand[i];
        if (Ktot % S || (Ktot / S) % 128) continue;
        const double rounds = ceil(S * tiles / ncu);
        const double epi = (S == 1 && resid) ? 25.0 : 12.0;
        double us = rounds * ((Ktot / S / 64) * 1.7 + epi);
        if (S > 1) us += 4.0 + (double)(S + (resid ? 2 : 1)) * M * N * 4.0 / 4.0e6;
        if (us < best_us) { best_us = us; best = S; }
    }
    if (best_us_out) *best_us_out = best_us;
    return best;
}

// One GEMM of the exact mode.  S = 1: the plain launch (EPI_F32 into dst / EPI_RESID onto dst).  S > 1: cut along K' into S parts
// that run as ONE persistent launch (PgGemmExtra::parts: S x tilesM x tilesN tiles on the 256 CUs): part p multiplies columns
// [p Kp, (p + 1) Kp) of the triple operands into the fp32 partial buffer p (the bias rides in part 0); then
// dst = (resid ? dst : 0) + sum of the parts, in part order (sum_parts_kernel).
// Which of the three forms a shape takes (a pure function of the shape): 0 = gemm_mid.hip, S >= 1 = the persistent kernel in S parts.
#ifndef PG_EXACT_MID_US_KT
#define PG_EXACT_MID_US_KT 0.6
#endif
static double precise_mid_us_kt() {     // microseconds per K tile of a gemm_mid round in the exact tier's routing (env PIGEON_EXACT_MID_US: A/B)
    static double v = -1.0;
    if (v < 0.0) { const char* e = getenv("PIGEON_EXACT_MID_US"); v = e ? atof(e) : PG_EXACT_MID_US_KT; if (!(v > 0.0)) v = PG_EXACT_MID_US_KT; }
    return v;
}
static int precise_route(int M, int N, int Ktot, bool resid, const int* cand, int ncand) {
    double parts_us = 0.0;
    const int S = precise_parts(M, N, Ktot, resid, cand, ncand, &parts_us);
    // Round 6: a handful of images (a settled-at-once exact pass: serving, certain_forward) -- the 128 x 128 one-tile-per-block kernel
    // (gemm_mid.hip) keeps the whole K' in one chain like S = 1 and still fills the chip; same cost model as pg_gemm_launch
    // (0.6 us per K tile of a round + epilogue).  Bit-identical to the S = 1 persistent launch.
    if (pg_gemm_mid_on() && N % 128 == 0) {
        const double rounds_m = ceil((double)((M + 127) / 128) * (N / 128) / (double)pg_num_cus());
        const double mid_us = rounds_m * ((Ktot / 64) * precise_mid_us_kt() + (resid ? 6.0 : 5.0));
        if (mid_us < parts_us) return 0;
    }
    return S;
}
static int precise_gemm(pg_vit* h, const uint16_t* A3, int64_t lda, const uint16_t* W3, int64_t ldw, const float* bias, float* parts,
                        float* dst, bool resid, int M, int N, int Ktot, const int* cand, int ncand, hipStream_t s) {
    (void)h;
    const int S = precise_route(M, N, Ktot, resid, cand, ncand);
    if (S == 0)
        return pg_gemm_launch(PG_DTYPE_F16, A3, lda, W3, ldw, bias, dst, N, M, N, Ktot, resid ? EPI_RESID : EPI_F32, 1.f, 0, nullptr, 71, s);
    if (S == 1)
        return pg_gemm_launch(PG_DTYPE_F16, A3, lda, W3, ldw, bias, dst, N, M, N, Ktot, resid ? EPI_RESID : EPI_F32, 1.f, 0, nullptr, 36, s);
    const int Kp = Ktot / S;
    const int64_t part_elems = (int64_t)M * N;
    PgGemmExtra ex;
    ex.parts = S; ex.a_part = Kp; ex.w_part = Kp; ex.c_part = part_elems;
    RC(pg_gemm_launch(PG_DTYPE_F16, A3, lda, W3, ldw, bias, parts, N, M, N, Kp, EPI_F32, 1.f, 0, nullptr, 36, s, &ex));
    return pg_sum_parts_launch(parts, S, part_elems, dst, part_elems, resid ? 1 : 0, s);
}

// pg_tune_exact_fusion (round 6): the exact pass's two activation splits ride in their producers -- the attention
// writes the triple the out-projection reads (attention_x3_kernel<true>), fc1's epilogue applies QuickGELU and writes the triple fc2
// reads (EPI_GELU_X3, where fc1 runs as one persistent launch: S = 1) -- instead of an fp32 buffer plus a split_x3 launch each.  The
// arithmetic per element is the same expression in both forms (x3.h): results are bit-identical (tests/test_gpu_precise.py).  0 = the
// unfused form (the A/B arm and the checker).
static int g_exact_fusion = -1;
extern "C" int pg_tune_exact_fusion(int on) {
    g_exact_fusion = on ? 1 : 0;
    return PG_OK;
}
static bool exact_fusion_on() {
    if (g_exact_fusion < 0) { const char* e = getenv("PIGEON_EXACT_FUSION"); g_exact_fusion = (e && e[0] == '0') ? 0 : 1; }
    return g_exact_fusion != 0;
}

static int vit_precise_chunk(pg_vit* h, const void* pixels, int pix_dtype, int n, float* emb_out, float* hidden_out, char* ws,
                             hipStream_t s) {
    const int64_t M = (int64_t)n * VIT_TOKENS;
    const int D = VIT_HIDDEN, F = VIT_MLP;
    float* X = (float*)ws;
    uint16_t* T3 = (uint16_t*)(ws + align_up((size_t)M * D * 4, 256));
    float* Fb = (float*)((char*)T3 + align_up((size_t)M * 3 * D * 2, 256));
    uint16_t* G3 = (uint16_t*)((char*)Fb + align_up((size_t)M * F * 4, 256));
    float* O = (float*)G3;                                    // fp32 attention output, dead before G3 is written
    float* PP = (float*)((char*)G3 + align_up((size_t)M * 3 * F * 2, 256));     // K-split partial products of one GEMM
    const float eps = h->cfg.ln_eps;
    const int dt = PG_DTYPE_F16, V = 36;                      // the 256 x 256 persistent kernel takes every epilogue used here
    static const int S3[2] = {1, 3}, S6[4] = {1, 2, 3, 6};    // K-part counts precise_parts may choose from (K' = 3072 / 12288)
    static const int S2[2] = {1, 2}, S4[3] = {1, 2, 4};       // ... with two products (K' = 2048 / 8192)
    const int np = g_exact_products;                          // partial products per weight GEMM (3; 2: pg_tune_exact_products)
    const int* c1 = np == 3 ? S3 : S2; const int* c2 = np == 3 ? S6 : S4; const int n2 = np == 3 ? 4 : 3;
    const bool fuse = exact_fusion_on();
    const bool fuse_fc1 = fuse && precise_route((int)M, F, np * D, false, c1, 2) == 1;    // one persistent launch: the epilogue can finish the job
    RC(pg_x3_im2col_launch(pixels, pix_dtype, G3, n, s));
    RC(pg_gemm_launch(dt, G3, 3 * VIT_PATCH_KPAD, h->wpatch3, 3 * VIT_PATCH_KPAD, nullptr, X, D, n * VIT_PATCHES, D, np * VIT_PATCH_KPAD,
                      EPI_PATCH, 1.f, 0, h->pos, V, s));
    RC(pg_preln_launch(X, h->cls, h->pos, h->preg, h->preb, M, eps, s));
    for (int l = 0; l < h->cfg.layers; ++l) {
        const LayerW& L = h->layers[l];
        RC(pg_x3_ln_launch(X, L.ln1g, L.ln1b, T3, M, eps, s));
        RC(precise_gemm(h, T3, 3 * D, L.wqkv3, 3 * D, L.bqkv_raw, PP, Fb, false, (int)M, 3 * D, np * D, c1, 2, s));
        if (fuse && pg_attention_x3out_available()) {
            RC(pg_attention_x3out_launch(Fb, T3, n, s));         // (T3's LayerNorm triple is dead: the QKV GEMM has consumed it)
        } else {
            RC(pg_attention_f32_launch(Fb, O, n, s));
            RC(pg_x3_split_launch(O, T3, M, D, 0, s));
        }
        RC(precise_gemm(h, T3, 3 * D, L.wo3, 3 * D, L.bo, PP, X, true, (int)M, D, np * D, c1, 2, s));
        RC(pg_x3_ln_launch(X, L.ln2g, L.ln2b, T3, M, eps, s));
        if (fuse_fc1) {
            RC(pg_gemm_launch(dt, T3, 3 * D, L.w13, 3 * D, L.b1_raw, G3, 3 * F, (int)M, F, np * D, EPI_GELU_X3, 1.f, 0, nullptr, V, s));
        } else {
            RC(precise_gemm(h, T3, 3 * D, L.w13, 3 * D, L.b1_raw, PP, Fb, false, (int)M, F, np * D, c1, 2, s));
            RC(pg_x3_split_launch(Fb, G3, M, F, 1, s));
        }
        RC(precise_gemm(h, G3, 3 * F, L.w23, 3 * F, L.b2, PP, X, true, (int)M, D, np * F, c2, n2, s));
    }
    RC(pg_token_mean_launch(X, emb_out, n, s));
    if (hidden_out) PG_HIP(hipMemcpyAsync(hidden_out, X, (size_t)M * D * 4, hipMemcpyDeviceToDevice, s));
    return PG_OK;
}

extern "C" int pg_vit_forward_precise(pg_vit* h, const void* pixels, int pix_dtype, int n_images, float* emb_out, float* hidden_out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!h) { pg_set_error("vit_forward_precise: null handle"); return PG_EINVAL; }
    if (!h->finalized) { pg_set_error("vit_forward_precise: handle not finalized"); return PG_ESTATE; }
    if (!h->cfg.precise) { pg_set_error("vit_forward_precise: the handle was created without cfg.precise (no split-weight copy)"); return PG_ESTATE; }
    if (n_images < 0) { pg_set_error("vit_forward_precise: n_images = %d", n_images); return PG_EINVAL; }
    if (n_images == 0) return PG_OK;
    if (!pixels || !emb_out || !workspace) { pg_set_error("vit_forward_precise: null argument"); return PG_EINVAL; }
    if (pix_dtype != PG_DTYPE_F32 && pix_dtype != PG_DTYPE_BF16 && pix_dtype != PG_DTYPE_F16) { pg_set_error("vit_forward_precise: bad pixel dtype"); return PG_EINVAL; }
    size_t needb = 0;
    pg_vit_precise_workspace_bytes(h, n_images, &needb);
    if (workspace_bytes < needb) { pg_set_error("vit_forward_precise: workspace %zu < required %zu bytes", workspace_bytes, needb); return PG_ENOMEM; }
    if (((uintptr_t)workspace & 255) != 0) { pg_set_error("vit_forward_precise: workspace must be 256-byte aligned"); return PG_EINVAL; }
    const size_t esz = pix_dtype == PG_DTYPE_F32 ? 4 : 2;
    const size_t img_elems = (size_t)3 * VIT_IMG * VIT_IMG;
    // equal chunks (130 images: 65 + 65, not 128 + 2 -- a two-image pass costs as much as a sixteen-image one)
    const int nchunks = (n_images + PG_PRECISE_CHUNK - 1) / PG_PRECISE_CHUNK;
    const int per = (n_images + nchunks - 1) / nchunks;
    for (int s0 = 0; s0 < n_images; s0 += per) {
        const int n = (n_images - s0) < per ? (n_images - s0) : per;
        RC(vit_precise_chunk(h, (const char*)pixels + (size_t)s0 * img_elems * esz, pix_dtype, n, emb_out + (size_t)s0 * VIT_HIDDEN,
                             hidden_out ? hidden_out + (size_t)s0 * VIT_TOKENS * VIT_HIDDEN : nullptr, (char*)workspace, (hipStream_t)stream));
    }
    return PG_OK;
}

extern "C" int pg_vit_destroy(pg_vit* h) {
    if (!h) return PG_OK;
    for (void* p : h->allocs) (void)hipFree(p);
    for (int i = 0; i < 3; ++i) { if (h->sx[i]) (void)hipStreamDestroy(h->sx[i]); if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (auto& e : h->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& g : h->graphs) graph_entry_free(g);
    if (h->capture_stream) (void)hipStreamDestroy(h->capture_stream);
    delete h;
    return PG_OK;
}

extern "C" int pg_vit_range_alarm_read(pg_vit* h, int64_t* rows, int reset) {
    if (!h || !rows) { pg_set_error("range_alarm_read: null argument"); return PG_EINVAL; }
    *rows = 0;
    if (!h->range_alarm) return PG_OK;                       // bf16 operands, or the separate-LayerNorm chain: no alarm
    int cur = 0;
    PG_HIP(hipGetDevice(&cur));
    PG_HIP(hipSetDevice(h->device));                         // the counter lives on the handle's device, whatever is current
    hipError_t e = hipDeviceSynchronize();
    unsigned long long v = 0;
    if (e == hipSuccess) e = hipMemcpy(&v, h->range_alarm, sizeof(v), hipMemcpyDeviceToHost);
    if (e == hipSuccess && reset) e = hipMemset(h->range_alarm, 0, sizeof(v));
    (void)hipSetDevice(cur);                                 // ... and the caller's current device is put back
    if (e != hipSuccess) { pg_set_error("range_alarm_read: %s", hipGetErrorString(e)); return PG_EHIP; }
    *rows = (int64_t)v;
    return PG_OK;
}

extern "C" int pg_vit_saturation_check(pg_vit* h, int on) {
    if (!h) { pg_set_error("saturation_check: null handle"); return PG_EINVAL; }
    if (on && !h->sat_counter) {
        PG_HIP(hipSetDevice(h->device));
        PG_HIP(hipMalloc((void**)&h->sat_counter, sizeof(unsigned long long)));
        h->allocs.push_back(h->sat_counter);
        PG_HIP(hipMemset(h->sat_counter, 0, sizeof(unsigned long long)));
    }
    h->sat_check = on != 0;
    return PG_OK;
}
extern "C" int pg_vit_saturation_read(pg_vit* h, int64_t* count, int reset) {
    if (!h || !count) { pg_set_error("saturation_read: null argument"); return PG_EINVAL; }
    *count = 0;
    if (!h->sat_counter) return PG_OK;
    int cur = 0;
    PG_HIP(hipGetDevice(&cur));
    PG_HIP(hipSetDevice(h->device));
    hipError_t e = hipDeviceSynchronize();
    unsigned long long v = 0;
    if (e == hipSuccess) e = hipMemcpy(&v, h->sat_counter, sizeof(v), hipMemcpyDeviceToHost);
    if (e == hipSuccess && reset) e = hipMemset(h->sat_counter, 0, sizeof(v));
    (void)hipSetDevice(cur);
    if (e != hipSuccess) { pg_set_error("saturation_read: %s", hipGetErrorString(e)); return PG_EHIP; }
    *count = (int64_t)v;
    return PG_OK;
}

extern "C" int pg_vit_graph(pg_vit* h, int on, int64_t* replays, int64_t* captures) {
    if (!h) { pg_set_error("vit_graph: null handle"); return PG_EINVAL; }
    if (on == 0 || on == 1) h->use_graph = on != 0;          // any other value: query only
    if (replays) *replays = h->graph_replays;
    if (captures) *captures = h->graph_captures;
    return PG_OK;
}

extern "C" int pg_vit_profile_enable(pg_vit* h, int on) {
    if (!h) { pg_set_error("profile_enable: null handle"); return PG_EINVAL; }
    h->prof = on != 0;
    h->prof_mask = (on == 1 || on == 0) ? 0xFFFFFFFFu : ((unsigned)on >> 1);     // on >= 2: bit (c + 1) selects class c
    return PG_OK;
}
static int prof_drain(pg_vit* h) {
    for (auto& e : h->evs) {
        PG_HIP(hipEventSynchronize(e.b));
        float ms = 0.f;
        PG_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        h->prof_ms[e.cls] += ms;
        h->prof_launches[e.cls] += 1;
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    h->evs.clear();
    return PG_OK;
}
extern "C" int pg_vit_profile_read(pg_vit* h, int64_t* launches, double* ms) {
    if (!h || !launches || !ms) { pg_set_error("profile_read: null argument"); return PG_EINVAL; }
    RC(prof_drain(h));
    for (int i = 0; i < PG_PROF_CLASSES; ++i) { launches[i] = h->prof_launches[i]; ms[i] = h->prof_ms[i]; }
    return PG_OK;
}
extern "C" int pg_vit_profile_reset(pg_vit* h) {
    if (!h) { pg_set_error("profile_reset: null handle"); return PG_EINVAL; }
    RC(prof_drain(h));
    for (int i = 0; i < PG_PROF_CLASSES; ++i) { h->prof_launches[i] = 0; h->prof_ms[i] = 0; }
    return PG_OK;
}

// ------------------------------------------------------------------------------------------------ op-level ABI
extern "C" int pg_vit_mma_dtype(const pg_vit* h) { return h ? h->cfg.mma_dtype : PG_EINVAL; }

extern "C" int pg_op_gemm16(int dtype, const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldc,
                            int M, int N, int K, int epi, float qscale, int qcols, const float* aux, int variant,
                            void* stream) {
    if (!A || !W || !out) { pg_set_error("op_gemm16: null argument"); return PG_EINVAL; }
    return pg_gemm_launch(dtype, A, lda, W, K, bias, out, ldc, M, N, K, epi, qscale, qcols, aux, variant, (hipStream_t)stream);
}
extern "C" int pg_op_gemm16_ld(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out,
                               int64_t ldc, int M, int N, int K, int epi, float qscale, int qcols, const float* aux,
                               int variant, void* stream) {
    if (!A || !W || !out) { pg_set_error("op_gemm16_ld: null argument"); return PG_EINVAL; }
    return pg_gemm_launch(dtype, A, lda, W, ldw, bias, out, ldc, M, N, K, epi, qscale, qcols, aux, variant, (hipStream_t)stream);
}
extern "C" int pg_op_rowstat_cast(const float* x, void* x16, int dtype, float* rowstat, int64_t rows, float eps, void* stream) {
    if (!x || !x16 || !rowstat) { pg_set_error("op_rowstat_cast: null argument"); return PG_EINVAL; }
    return pg_rowstat_cast_launch(x, x16, dtype, rowstat, rows, eps, (hipStream_t)stream);
}
extern "C" int pg_op_rowstat_finalize(const float* statpart, int slots, float* rowstat, int64_t rows, float eps, void* stream) {
    if (!statpart || !rowstat || slots <= 0) { pg_set_error("op_rowstat_finalize: bad argument"); return PG_EINVAL; }
    return pg_rowstat_finalize_launch(statpart, slots, rowstat, rows, eps, (hipStream_t)stream);
}
extern "C" int pg_op_gemm16_resid_stat(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* X,
                                       int64_t ldc, void* x16, int64_t ldx, float* statpart, int M, int N, int K, int variant,
                                       void* stream) {
    if (!A || !W || !X || !x16 || !statpart) { pg_set_error("op_gemm16_resid_stat: null argument"); return PG_EINVAL; }
    PgGemmExtra ex; ex.x16 = x16; ex.ldx = ldx; ex.statpart = statpart;
    return pg_gemm_launch(dtype, A, lda, W, ldw, bias, X, ldc, M, N, K, EPI_RESID_STAT, 1.f, 0, nullptr, variant ? variant : 36,
                          (hipStream_t)stream, &ex);
}
extern "C" int pg_op_gemm16_ln(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                               const float* colsum, const float* rowstat, void* out, int64_t ldc, int M, int N, int K, int epi,
                               float qscale, int qcols, int variant, void* stream) {
    if (!A || !W || !out || !colsum || !rowstat) { pg_set_error("op_gemm16_ln: null argument"); return PG_EINVAL; }
    if (epi != EPI_QKV_LN && epi != EPI_GELU_LN) { pg_set_error("op_gemm16_ln: epi must be 6 or 7"); return PG_EINVAL; }
    PgGemmExtra ex; ex.colsum = colsum; ex.rowstat = rowstat;
    return pg_gemm_launch(dtype, A, lda, W, ldw, bias, out, ldc, M, N, K, epi, qscale, qcols, nullptr, variant ? variant : 36,
                          (hipStream_t)stream, &ex);
}
extern "C" int pg_op_layernorm(const float* x, const float* gamma, const float* beta, void* y, int out_dtype,
                               int64_t rows, float eps, void* stream) {
    if (!x || !gamma || !beta || !y) { pg_set_error("op_layernorm: null argument"); return PG_EINVAL; }
    return pg_layernorm_launch(x, gamma, beta, y, out_dtype, rows, eps, (hipStream_t)stream);
}
extern "C" int pg_op_attention(int dtype, const void* qkv, void* out, int n_images, void* stream) {
    if (!qkv || !out) { pg_set_error("op_attention: null argument"); return PG_EINVAL; }
    return pg_attention_launch(dtype, qkv, out, n_images, (hipStream_t)stream);
}
extern "C" int pg_op_im2col(const void* pixels, int pix_dtype, void* out, int out_dtype, int n_images, void* stream) {
    if (!pixels || !out) { pg_set_error("op_im2col: null argument"); return PG_EINVAL; }
    return pg_im2col_launch(pixels, pix_dtype, out, out_dtype, n_images, (hipStream_t)stream);
}
extern "C" int pg_op_token_mean(const float* x, float* out, int n_images, void* stream) {
    if (!x || !out) { pg_set_error("op_token_mean: null argument"); return PG_EINVAL; }
    return pg_token_mean_launch(x, out, n_images, (hipStream_t)stream);
}
extern "C" int pg_op_cast_f32(const float* x, void* y, int out_dtype, int64_t n, void* stream) {
    if (!x || !y) { pg_set_error("op_cast_f32: null argument"); return PG_EINVAL; }
    return pg_cast_f32_launch(x, y, out_dtype, n, (hipStream_t)stream);
}

extern "C" int pg_op_gemm16_parts(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* parts,
                                  int M, int N, int Kp, int S, void* stream) {
    if (!A || !W || !parts) { pg_set_error("op_gemm16_parts: null argument"); return PG_EINVAL; }
    if (S < 1 || S > 8) { pg_set_error("op_gemm16_parts: 1 <= S <= 8"); return PG_EINVAL; }
    PgGemmExtra ex;
    ex.parts = S; ex.a_part = Kp; ex.w_part = Kp; ex.c_part = (int64_t)M * N;
    return pg_gemm_launch(dtype, A, lda, W, ldw, bias, parts, N, M, N, Kp, EPI_F32, 1.f, 0, nullptr, 36, (hipStream_t)stream, &ex);
}

// exact-mode building blocks (precise.hip), exported for the parity tests
extern "C" int pg_op_x3_split(const float* x, void* y3, int64_t rows, int cols, int gelu, void* stream) {
    if (!x || !y3) { pg_set_error("op_x3_split: null argument"); return PG_EINVAL; }
    return pg_x3_split_launch(x, y3, rows, cols, gelu, (hipStream_t)stream);
}
extern "C" int pg_op_x3_layernorm(const float* x, const float* gamma, const float* beta, void* y3, int64_t rows, float eps, void* stream) {
    if (!x || !gamma || !beta || !y3) { pg_set_error("op_x3_layernorm: null argument"); return PG_EINVAL; }
    return pg_x3_ln_launch(x, gamma, beta, y3, rows, eps, (hipStream_t)stream);
}
extern "C" int pg_op_attention_f32(const float* qkv, float* out, int n_images, void* stream) {
    if (!qkv || !out) { pg_set_error("op_attention_f32: null argument"); return PG_EINVAL; }
    return pg_attention_f32_launch(qkv, out, n_images, (hipStream_t)stream);
}
