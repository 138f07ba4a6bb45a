/* pigeon_hip.h -- C ABI of libpigeon_hip.so: the MI355X (gfx950) kernels behind PIGEON's inference hot path.
 *
 * The reference (LukasHaas/PIGEON) is 100% Python and has NO native boundary of its own:
 * every GPU FLOP is issued by stock PyTorch ops reached through HuggingFace transformers.  The drop-in
 * boundary is therefore the Python class surface (CLIPEmbedding / SuperGuessr / ProtoRefiner, mirrored in
 * pigeon_amd/), and THIS header is the FFI those mirrors bind with ctypes -- exactly what a maintainer of
 * the reference would bind to replace the library kernels it reaches today.  Each entry point cites the
 * reference call site(s) whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C types only; every pointer documented as DEVICE (HBM, on the handle's device) or HOST;
 *   - every function returns 0 on success or a negative PG_E* code; pg_last_error() gives a message;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is asynchronous on it, the
 *     library never synchronises the device inside a forward call;
 *   - the caller owns all inputs, outputs and workspaces (torch tensors in the Python host); the library
 *     owns only the packed weight copies held inside a pg_vit handle;
 *   - handles are not thread-safe: one per (device, stream).
 */
#ifndef PIGEON_HIP_H
#define PIGEON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK            0
#define PG_EINVAL       -1   /* bad argument / unsupported shape */
#define PG_ENOMEM       -2   /* device allocation failed / workspace too small */
#define PG_EHIP         -3   /* a HIP runtime call failed (see pg_last_error) */
#define PG_ESTATE       -4   /* handle not finalized / weight missing */

#define PG_DTYPE_F32     0
#define PG_DTYPE_BF16    1
#define PG_DTYPE_F16     2
#define PG_DTYPE_F64     3

#define PG_ABI_VERSION   5   /* 2: pg_vit_cfg.precise, pg_vit_forward_precise, pg_head_margin (round 4); 3: pg_head_certainty, pg_refine_forward_ex,
                              * pg_refine_certainty, pg_tune_gemm_raster (round 5); 4: the deferred exact tier -- pg_requeue_append,
                              * pg_rows_to_slots, pg_requeue_take, pg_scatter_rows, pg_head_wstats (round 6); 5: pg_embedding_debias (round 6) */

const char* pg_last_error(void);
int pg_abi_version(void);
/* Number of HIP devices visible; <0 on error.  Lets the host fail loudly when there is no GPU. */
int pg_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * ViT-L/14-336 image encoder.
 * Replaces: transformers CLIPVisionModel.forward as called at reference models/clip_embedder.py:63 and
 * models/super_guessr.py:395, plus the token mean at models/clip_embedder.py:64-65 /
 * models/super_guessr.py:397-398 (embedding = mean over ALL 577 tokens of last_hidden_state, taken
 * before post_layernorm).
 * ------------------------------------------------------------------------------------------------ */
typedef struct pg_vit pg_vit;

typedef struct pg_vit_cfg {
    int32_t layers;       /* 24 for ViT-L; any >=1 accepted (tests use 2) */
    int32_t image_size;   /* must be 336 */
    int32_t patch;        /* must be 14  */
    int32_t hidden;       /* must be 1024 */
    int32_t heads;        /* must be 16  */
    int32_t mlp;          /* must be 4096 */
    float   ln_eps;       /* 1e-5 */
    int32_t max_chunk;    /* images processed per internal pass (0 = default 512) */
    int32_t mma_dtype;    /* 16-bit MFMA operand format for weights and activations: PG_DTYPE_F16, PG_DTYPE_BF16, or
                             0 = default (fp16, unless env PIGEON_MMA_DTYPE=bf16).  Same MFMA rate on gfx950; fp16
                             keeps embeddings within 1e-3 of the fp32 reference, bf16 measures 2e-3. */
    int32_t precise;      /* != 0: pg_vit_finalize also packs the split-fp16 weight copy pg_vit_forward_precise needs (3x the
                             16-bit weight memory: 1.8 GB for ViT-L); 0: the exact mode is not available on this handle */
} pg_vit_cfg;

int pg_vit_create(pg_vit** out, int device, const pg_vit_cfg* cfg);
/* Copy one parameter into the handle.  `name` is the state-dict key of transformers' CLIPVisionModel, either
 * layout: 4.23.1 "vision_model.embeddings.patch_embedding.weight" or 5.x "embeddings.patch_embedding.weight"
 * (reference loads by name: models/utils.py:24-45, models/super_guessr.py:222-238).  `data` is a HOST pointer
 * to contiguous fp32 (PG_DTYPE_F32); the library converts GEMM weights to bf16 and keeps its own device copy.
 * Unknown names (post_layernorm.*, position_ids) are accepted and ignored (dead on this path). */
int pg_vit_load_weight(pg_vit* h, const char* name, const void* data, int dtype,
                       const int64_t* shape, int ndim);
/* Checks that every required parameter was loaded and uploads fused/packed forms. */
int pg_vit_finalize(pg_vit* h);
/* Bytes of DEVICE workspace pg_vit_forward needs for n_images (caller allocates, e.g. a torch uint8 tensor). */
int pg_vit_workspace_bytes(const pg_vit* h, int n_images, size_t* bytes);
/* pixels: DEVICE (n_images,3,336,336) contiguous NCHW, fp32 (PG_DTYPE_F32), fp16 (what pg_prep_forward writes) or bf16.
 * emb_out: DEVICE (n_images,1024) fp32 -- mean over the 577 tokens of last_hidden_state.
 * n_images = 0 is a no-op (PG_OK; the buffers of an empty batch may be NULL), n_images < 0 is PG_EINVAL -- the same holds for
 * pg_prep_forward, pg_head_forward (B) and pg_refine_forward (B). */
int pg_vit_forward(pg_vit* h, const void* pixels, int pix_dtype, int n_images, float* emb_out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Same, additionally copying the final residual stream (n_images,577,1024) fp32 to `hidden_out` (DEVICE,
 * may be NULL).  Used by parity tests to compare last_hidden_state itself. */
int pg_vit_forward_hidden(pg_vit* h, const void* pixels, int pix_dtype, int n_images, float* emb_out,
                          float* hidden_out, void* workspace, size_t workspace_bytes, void* stream);
/* EXACT MODE (round 4).  The same encoder in near-fp32 arithmetic, for the few inputs whose downstream decision the 16-bit
 * operands cannot settle -- the reference's geocell `torch.argmax` (models/super_guessr.py:454) is fp32 end to end, and a panorama
 * whose top-1 / top-2 logit margin lies inside the fast path's error band (pg_head_margin below) may flip.  Every GEMM operand is
 * split into two fp16 halves (x = hi + lo, W = Wh + Wl) and the products hi.Wh + lo.Wh + hi.Wl are accumulated in fp32 by the same
 * persistent MFMA kernels over a 3x longer K; LayerNorm, attention (fp32 MFMA), QuickGELU (expf, IEEE division) and the residual
 * stream are fp32.  Measured against the fp32 reference: embeddings to ~1e-6 relative (fast path: 2.7e-4), at ~5x the time per
 * image.  Needs cfg.precise at creation.  Workspace from pg_vit_precise_workspace_bytes (98 KB per token row, equal chunks of at most 128 images).
 * hidden_out (DEVICE (n_images,577,1024) fp32, may be NULL) receives last_hidden_state. */
int pg_vit_precise_workspace_bytes(const pg_vit* h, int n_images, size_t* bytes);
int pg_vit_forward_precise(pg_vit* h, const void* pixels, int pix_dtype, int n_images, float* emb_out, float* hidden_out,
                           void* workspace, size_t workspace_bytes, void* stream);
/* hipGraph of the encoder (round 4).  pg_vit_forward replays the ~250 launches between im2col and the token mean -- they touch only
 * the workspace and the handle's weights -- from a graph captured at the second forward with the same (workspace, n_images); default
 * on (env PIGEON_VIT_GRAPH=0: off).  Results are bit-identical either way (same kernels, same order).  Forwards run un-graphed while
 * pg_vit_profile_enable / pg_vit_saturation_check are on.  on = 1 / 0 switches it, any other value only queries; replays / captures
 * (may be NULL) return the counts since creation. */
int pg_vit_graph(pg_vit* h, int on, int64_t* replays, int64_t* captures);
int pg_vit_destroy(pg_vit* h);
/* The operand format the handle resolved to (PG_DTYPE_F16 or PG_DTYPE_BF16). */
int pg_vit_mma_dtype(const pg_vit* h);

/* Per-kernel-class timing (HIP events on `stream`), for bench.py's roofline object.
 * pg_vit_profile_enable(h,1) makes subsequent forwards bracket every launch with events; a value >= 2 is a class mask
 * shifted left by one (bit c+1 = class c, e.g. 1 << 3 = the fc1 GEMMs only): bench.py --profile dominant keeps just the
 * dominant class's events inside its timed region; 0 switches the events off;
 * pg_vit_profile_read synchronises those events and returns, per class, launches and total milliseconds.
 * Classes: 0 gemm_qkv 1 gemm_out 2 gemm_fc1 3 gemm_fc2 4 gemm_patch 5 attention 6 layernorm 7 im2col
 *          8 token_mean  (PG_PROF_CLASSES entries). */
#define PG_PROF_CLASSES 9
int pg_vit_profile_enable(pg_vit* h, int on);
int pg_vit_profile_read(pg_vit* h, int64_t* launches /*[PG_PROF_CLASSES]*/, double* ms /*[PG_PROF_CLASSES]*/);
int pg_vit_profile_reset(pg_vit* h);

/* fp16 saturation check (debug aid, off by default; costs one scan per 16-bit activation buffer per layer).  The
 * encoder's fp32 -> fp16 conversions clamp at +-65504 instead of overflowing to inf; with the check enabled the forward
 * pass counts, buffer by buffer, the 16-bit activations sitting exactly on that limit (bf16 operands: the +-inf).
 * pg_vit_saturation_read returns the total since the last reset (synchronises the device); 0 = nothing was clamped. */
/* Tuning knob (process-wide, default from the build / env PIGEON_GEMM_STAGGER): the persistent GEMMs start XCD x of the 8
 * XCDs x/8 * fraction of a tile period late, so that the epilogue (HBM) phase of one XCD overlaps the mainloop (MFMA)
 * phases of the others; 0 = all blocks start together.  Changes timing only, never results. */
int pg_tune_gemm_stagger(float fraction);
/* Tuning knob (also env PIGEON_GEMM_TAIL_ROWS; default 768): when a GEMM's tiles do not fill the last round of the persistent
 * kernels and at most `rows` rows lie beyond the last whole round, those rows go to a small-tile kernel (csrc/gemm_tail.hip)
 * that spreads them over all CUs; 0 = never.  Both kernels produce the same bits for a row: timing only, never results. */
int pg_tune_gemm_tail_rows(int rows);
/* ... and only for GEMMs with K >= min_k or N >= min_n (env PIGEON_GEMM_TAIL_MIN_K / PIGEON_GEMM_TAIL_MIN_N; defaults 2048 / 4096:
 * of the model's four GEMMs the split pays for fc2 and fc1 only).  (0, 0) = every shape.  Timing only, never results. */
int pg_tune_gemm_tail_shape(int min_k, int min_n);
/* Raster of the 384 x 256 persistent GEMM (QKV, fc1, fc2; also env PIGEON_GEMM_RASTER_GN): N tiles per group an XCD's round walks
 * before it moves to the next row panels.  0 = default (4: 8 x 4 super-tiles, 2 MB of weights resident per XCD), -1 = all N tiles
 * (each activation panel crosses the fabric once, the weight panels are re-streamed per XCD), 1..64 = explicit (honoured for a GEMM only when it divides that
 * GEMM's count of 256-column tiles, or is at least that count; otherwise the default stays).  Timing only, never results. */
int pg_tune_gemm_raster(int gn);
/* Small and middle batches (round 6; also env PIGEON_GEMM_MID=0 / 1 / 2): a GEMM launch of up to ~64 images (40 000 token rows) is
 * routed between the variant's own persistent kernel (384 x 256 tiles where they exist), the 256 x 256 persistent kernel and a 128 x 128
 * one-tile-per-block kernel (csrc/gemm_mid.hip) by a cost model of how their row panels fill rounds of the CUs (csrc/gemm_bf16.hip
 * gemm_model_us).  1 = on (default), 0 = a variant always means its own kernel, 2 = gemm_mid.hip is the only alternative (A/B arm).
 * All GEMM kernels produce the same bits for a row: timing only, never results. */
int pg_tune_gemm_mid(int on);
/* Which kernel pg_op_gemm16* / the encoder would launch for this shape under the current knobs: *kind = 0 the 384 x 256 persistent
 * kernel, 1 the 256 x 256 one, 2 csrc/gemm_mid.hip, -1 none of them (a non-persistent variant or shape).  variant 0 = the default.
 * Host arithmetic only: no launch, no device work (tests/test_host_cpu.py checks the picks against profiles/r06/gemm_three_sweep.txt). */
int pg_gemm_route(int variant, int epi, int M, int N, int K, int* kind);
/* Exact mode's attention (also env PIGEON_EXACT_ATTN=f32): 0 = split-fp16 operands on v_mfma_f32_32x32x16_f16 (default, round 5),
 * 1 = plain fp32 on v_mfma_f32_32x32x2_f32 (round 4's kernel; the A/B arm).  Both are fp32-grade (6e-7 / 8e-7 against fp64). */
int pg_tune_exact_attention(int use_f32_mfma);
/* Exact mode's weight GEMMs (round 6, experimental): 3 = all three partial products hi.Wh + lo.Wh + hi.Wl (default: the exact tier), 2 = the
 * first two (both halves of the activations, the weights at their fp16 value: K' = 2K on the same operands) -- what remains is the
 * weights' rounding, a systematic embedding error the same for every image up to its dependence on the activations; 2/3 of the GEMM
 * work.  Process-wide; anything else is PG_EINVAL. */
int pg_tune_exact_products(int n);
/* Exact mode's activation splits (round 6; also env PIGEON_EXACT_FUSION=0): 1 = fused into their producers (default: the attention writes
 * the split-fp16 triple the out-projection reads, fc1's epilogue applies QuickGELU and writes the triple fc2 reads -- csrc/x3.h holds
 * the one definition of both forms' arithmetic), 0 = an fp32 buffer plus a split kernel each (the checker).  Bit-identical results. */
int pg_tune_exact_fusion(int on);
int pg_vit_saturation_check(pg_vit* h, int on);
int pg_vit_saturation_read(pg_vit* h, int64_t* count, int reset);
/* Always-on range alarm of the fp16 operand path (no scan, no cost worth naming): the kernel that turns the residual GEMMs' row
 * statistics into (rstd, mean rstd) also counts the rows whose sum of squares reaches 65504^2 -- a NECESSARY condition for an
 * element of that row's 16-bit copy (the next GEMM's operand) to have been clamped at the fp16 limit.  0 = no residual row ever
 * came near the limit since the last reset; > 0 = run pg_vit_saturation_check for the exact element count, or switch the
 * handle to bf16 operands.  Always 0 with bf16 operands or the separate-LayerNorm chain.  Synchronises the device. */
int pg_vit_range_alarm_read(pg_vit* h, int64_t* rows, int reset);

/* ------------------------------------------------------------------------------------------------
 * SuperGuessr geocell head.
 * Replaces: models/super_guessr.py:437 (panel mean), :447 (cell_layer Linear), :448 (softmax), :454 (argmax),
 * :455 (index_select of lla_geocells), :459 (topk).  fp32 throughout; centroids float64.
 *   emb      DEVICE (B,P,1024) fp32, P = 4 (panorama) or 1
 *   W, bias  DEVICE (C,1024) / (C) fp32 -- cell_layer.weight / .bias
 *   centroids DEVICE (C,2) float64 [lng,lat] -- lla_geocells
 *   logits   DEVICE (B,C) fp32 out (required; also scratch)
 *   topk_val DEVICE (B,k) fp32 out: softmax probabilities, descending; ties -> lower index first
 *   topk_idx DEVICE (B,k) int64 out
 *   argmax   DEVICE (B) int64 out (== topk_idx[:,0])
 *   pred_llh DEVICE (B,2) float64 out = centroids[argmax]
 * Limits: 1 <= k <= C, P >= 1; anything else is PG_EINVAL.  Up to C = 38 400 a row's probabilities live in LDS while its top k are
 * picked (the reference's geocell sets have 2 000 - 11 000 cells); larger heads take the same kernel over a stream-ordered device
 * scratch of B x C floats (hipMallocAsync / hipFreeAsync on `stream`), same results.
 * ------------------------------------------------------------------------------------------------ */
int pg_head_forward(const float* emb, int B, int P, const float* W, const float* bias,
                    const double* centroids, int C, int k, float* logits,
                    float* topk_val, int64_t* topk_idx, int64_t* argmax, double* pred_llh, void* stream);

/* Certainty of the top-1 (round 4; reference models/super_guessr.py:454 `torch.argmax`): per row of `logits` (B,C) as written by
 * pg_head_forward,
 *   margin DEVICE (B) fp32 out = logit(top-1) - logit(top-2)   (value desc, index asc -- the order of the selection above)
 *   sens   DEVICE (B) fp32 out = |mean_p emb|_2 * |W[top1] - W[top2]|_2 / sqrt(1024): the change of the margin per unit RELATIVE
 *          embedding error in a random direction; the host calls a row certain when margin > kappa * eps * sens, eps = its bound
 *          on the encoder's relative embedding error (pigeon_amd/super_guessr.py)
 *   top2   DEVICE (B) int64 out, may be NULL: the runner-up cell.   C == 1: margin = +inf, sens = 0, top2 = top1. */
int pg_head_margin(const float* logits, int B, int C, const float* emb, int P, const float* W, float* margin, float* sens,
                   int64_t* top2, void* stream);

/* Certainty of the top-1 against EVERY cell (round 5; supersedes pg_head_margin's top-1 / top-2 pair).  The embedding this path
 * feeds the head differs from the reference's fp32 one by  |e| (beta + r):  beta a calibrated systematic part (relative to |e|,
 * may be NULL), r of relative RMS norm eps in an unknown direction.  Per row,
 *   tol  DEVICE (B) fp32 out = min over cells c != top1 of (logit(top1) - logit(c) - |e| g.beta) / (|e| |g| / 32),  g = W[top1] - W[c]:
 *        the largest eps, in standard deviations of the margin change, that the reference's argmax (models/super_guessr.py:454) survives;
 *        the cells are the kx listed in topk_idx (as pg_head_forward wrote them, descending) and -- bounded with the list's last logit,
 *        |g| <= |W[top1]| + wstats[0] and g.beta <= W[top1].beta + wstats[1] -- every cell not listed.  The host calls a row certain
 *        when tol > kappa * eps.
 *   code DEVICE (B) int32 out: the list position j >= 1 that sets tol, -1 = the cells beyond the list, -2 = bad index in the list
 *   margin, sens DEVICE (B) fp32 out, may be NULL: pg_head_margin's pair (top-1 against top-2), for reports
 *   wstats DEVICE (2) fp32: [0] the largest row norm of W, [1] max over cells of |W[c].beta| (0 without beta).
 *   Limits: 1 <= kx <= C; kx == C: nothing beyond the list. */
int pg_head_certainty(const float* logits, int B, int C, const float* emb, int P, const float* W, const int64_t* topk_idx, int kx,
                      const float* beta, const float* wstats, float* tol, int32_t* code, float* margin, float* sens, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ProtoRefiner prototype-distance refinement over a CSR prototype bank.
 * Replaces: models/proto_refiner.py:154-222 (the per-sample / per-candidate Python loop), :233-255
 * (within-cluster refinement: FARTHEST member), :332-344 (torch.cdist L2), :346-357 (temperature softmax
 * without max shift) and preprocessing/geo_utils.py:40-55 (haversine veto: float64, the refined fp32 point promoted after deg2rad / cos as torch does).
 * The bank arrays are BORROWED device pointers (caller keeps the tensors alive):
 *   proto_emb (P,1024) f32; cell_off (C+1) i64; proto_lnglat (P,2) f32; proto_count (P) i32;
 *   member_off (P+1) i64; member_idx (Nm) i64; train_emb (Ntr,1024) f32; train_lnglat (Ntr,2) f32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct pg_bank {
    const float*   proto_emb;
    const int64_t* cell_off;
    const float*   proto_lnglat;
    const int32_t* proto_count;
    const int64_t* member_off;
    const int64_t* member_idx;
    const float*   train_emb;
    const float*   train_lnglat;
    int64_t num_cells;
    int64_t num_protos;
    int64_t num_train;
} pg_bank;

/*   q          DEVICE (B,P,1024) fp32 query embeddings (P panels are averaged first: proto_refiner.py:139-140)
 *   init_llh   DEVICE (B,2) float64 initial [lng,lat] predictions
 *   cand       DEVICE (B,k) int64 candidate geocells;  cand_prob DEVICE (B,k) fp32 (NULL -> [1,0,0,...], :143-145)
 *   topk <= k, topk <= 64
 *   scratch    DEVICE >= B*topk*4 floats; on return (score, lng, lat, bank rows streamed) per (query, candidate)
 *   out_llh    DEVICE (B,2) fp32; out_cell DEVICE (B) int64; out_choice DEVICE (B) int32 (index of the chosen
 *              candidate, the reference's guess_index, proto_refiner.py:220)                              */
int pg_refine_forward(const pg_bank* bank, const float* q, int B, int P, const double* init_llh,
                      const int64_t* cand, const float* cand_prob, int k, int topk,
                      float temperature, double max_refine_km, float* scratch,
                      float* out_llh, int64_t* out_cell, int32_t* out_choice, void* stream);

/* pg_refine_forward with n_eval >= topk candidates evaluated per query (topk <= n_eval <= k): the selection is pg_refine_forward's,
 * over the first topk only; the candidates beyond take no part in it and exist for pg_refine_certainty (could one of them enter the
 * set and win?).  scratch12: DEVICE >= B*n_eval*12 floats; per (query, candidate) on return
 *   [0] score = -distance to the nearest prototype (-100000: empty cell)  [1] lng  [2] lat  [3] bank rows streamed
 *   [4] distance of the runner-up prototype (+inf: none)   [5] / [6] bank rows of the nearest / runner-up prototype
 *   [7] / [8] distance of the farthest / second farthest member of the chosen cluster (-1: none)
 *   [9] / [10] their training-bank rows   [11] member count of the chosen prototype      ([5] [6] [9] [10] [11]: int32 bit patterns, -1 = none)
 * out_refined DEVICE (B) int32, may be NULL: the candidate picked BEFORE the haversine veto (out_choice: after it). */
int pg_refine_forward_ex(const pg_bank* bank, const float* q, int B, int P, const double* init_llh,
                         const int64_t* cand, const float* cand_prob, int k, int topk, int n_eval,
                         float temperature, double max_refine_km, float* scratch12,
                         float* out_llh, int64_t* out_cell, int32_t* out_choice, int32_t* out_refined, void* stream);

/* Certainty of the refined cell and point (round 5; reference models/proto_refiner.py:176-222).  Same error model and units as
 * pg_head_certainty; the decisions of a row are: the winning candidate r against every other candidate of the set
 * (s_j = log p_j - d_j / T; gradient W[c_r] - W[c_j] - (u_r - u_j) / T with u_j the unit vector from candidate j's nearest prototype to
 * the query), every evaluated candidate outside the set (it must get into the set AND -- unless it pushes r out -- win), the cells
 * beyond the evaluated ones (getting in counts; only when n_eval > topk, i.e. when the caller supplied candidates past the set), and
 * for the refined and the finally chosen candidate the nearest-prototype and farthest-member picks.  The haversine veto compares
 * two discrete points and has no margin.   W (C,1024) = the head's weights (the candidates' log-probabilities move with the
 * embedding through them), wstats as above, refined / choice as pg_refine_forward_ex wrote them.
 *   tol DEVICE (B) fp32 out;  code DEVICE (B) int32 out: 1000 + j / 2000 + j / 2999 / 3000 + w / 4000 + w (see csrc/certainty.hip; round 6: the nearest
 *   prototype / farthest member are checked against EVERY other prototype of the cell / member of the cluster, whose rows are streamed again), -9 = the winning product underflows in fp32, or
 *   an empty cell wins a set that is not all empty (uncertain), -8 = refined / choice outside [0, topk) (tol 0), 0 = nothing can change
 *   the row.   Limits: topk <= 64, n_eval <= min(k, 96). */
int pg_refine_certainty(const pg_bank* bank, const float* q, int B, int P, const int64_t* cand, const float* cand_prob, int k,
                        int topk, int n_eval, const float* scratch12, const float* W, int C, const float* beta,
                        const float* wstats, float temperature, const int32_t* refined, const int32_t* choice,
                        float* tol, int32_t* code, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The deferred exact tier (round 6; csrc/requeue.hip).  The reference's discrete outputs are fp32 end to end
 * (models/super_guessr.py:447-459, models/proto_refiner.py:176-222); a sample that pg_head_certainty / pg_refine_certainty cannot call
 * certain is re-encoded by pg_vit_forward_precise.  These entry points keep that decision ON THE DEVICE: no `nonzero`, no host
 * synchronisation per step.  The caller owns a circular queue of `cap` slots -- the pixels of the queued rows (cap x row_bytes) and, per
 * slot, the row of its result ring the sample belongs to -- and a result ring that the exact tier's outputs are scattered into before
 * the host collects anything (the reference concatenates at the end of its loop: training/train_eval_loop.py:98-112).
 *
 * pg_requeue_append: per row r < B
 *     certain[r] = head_tol[r] > thr && (refine_tol == NULL || refine_tol[r] > thr) && !force_all     (a NaN tolerance is not certain)
 *     cause[r]   = 0 certain | 1 the head's top-1 | else refine_code[r] (pg_refine_certainty's decision code; 2 when that is 0 / NULL)
 *   and, for the rows that are not certain and while the queue has room (appended - flushed < cap), in row order:
 *     slot = appended % cap;  slot_dst[slot] = dst_base + r;  row_slot[r] = slot;  ++appended
 *   row_slot[r] = -1 for certain rows, -2 for rows that did not fit (counted in counters[1]; the host sizes the queue so that this
 *   cannot happen and checks the counter).  counters DEVICE int64[2] = { rows ever appended, rows that did not fit }; `flushed` = rows
 *   the host has already taken out (it knows: it enqueued those flushes).  cap = 0: flags and causes only (counters, slot_dst,
 *   row_slot may be NULL) -- how the exact tier's own results are judged.   certain DEVICE uint8[B], cause DEVICE int32[B], may be NULL.
 * pg_rows_to_slots:  dst + row_slot[r] * row_bytes <- src + r * row_bytes for every r < B with row_slot[r] >= 0 (the queued pixels).
 * pg_requeue_take:   dst_out[i] = i < n_valid ? slot_dst[(head + i) % cap] : -1,  i < n_pad   (ring rows of the slots being flushed;
 *                    the padding rows exist so that every rank of a data-parallel job runs the exact tier on the same batch size).
 * pg_scatter_rows:   dst + d * row_bytes <- src + i * row_bytes, d = dst_row[i] >= 0, i < n.  remap_wb > 0: dst_row addresses the
 *                    GATHERED ring (slabs of remap_wb rows, this rank's rows at [remap_off, remap_off + remap_b) of a slab) while dst is
 *                    a local-only array with slabs of remap_b rows: d = (dst_row / remap_wb) * remap_b + dst_row % remap_wb - remap_off,
 *                    rows of other ranks are skipped.  Rows with d >= dst_rows are skipped.
 * pg_head_wstats:    out2 DEVICE float[2] = { max_c |W[c]|_2, max_c |W[c] . beta| } (beta DEVICE (1024), may be NULL -> 0): the bounds
 *                    pg_head_certainty / pg_refine_certainty take for the cells they do not visit one by one.
 * All asynchronous on `stream`. */
int pg_requeue_append(const float* head_tol, const float* refine_tol, const int32_t* refine_code, int B, float thr, int force_all,
                      int64_t dst_base, int64_t flushed, int64_t cap, int64_t* counters, int64_t* slot_dst, int32_t* row_slot,
                      uint8_t* certain, int32_t* cause, void* stream);
int pg_rows_to_slots(const void* src, int64_t row_bytes, const int32_t* row_slot, int B, void* dst, void* stream);
int pg_requeue_take(const int64_t* slot_dst, int64_t cap, int64_t head, int n_valid, int n_pad, int64_t* dst_out, void* stream);
int pg_scatter_rows(const void* src, int64_t row_bytes, const int64_t* dst_row, int n, void* dst, int64_t dst_rows,
                    int64_t remap_wb, int64_t remap_b, int64_t remap_off, void* stream);
int pg_head_wstats(const float* W, int C, const float* beta, float* out2, void* stream);

/* The systematic part of the 16-bit encoder's error taken out of its embeddings, in place:  emb[r] <- emb[r] - |emb[r]|_2 * bias
 * for every row r < n of the DEVICE (n, dim) fp32 matrix (dim must be 1024); bias DEVICE (1024) fp32 = the mean over a few calibration
 * images of (fast - exact) / |exact| (pigeon_amd/certainty.py: measured once per set of weights by sending the same images through
 * pg_vit_forward and pg_vit_forward_precise).  The reference's encoder is fp32 (models/clip_embedder.py:63-65,
 * models/super_guessr.py:395-398) and has no such term; what is left of this path's error after the subtraction is the part that
 * differs from image to image, which pg_head_certainty / pg_refine_certainty account for.  The norm is summed in a fixed order (a row's
 * bits do not depend on the batch).  Asynchronous on `stream`. */
int pg_embedding_debias(float* emb, int64_t n, int dim, const float* bias, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CLIP image preprocessing (the step in front of the encoder): uint8 RGB -> pixel_values.
 * Replaces `CLIPProcessor(images=..., return_tensors='pt')` (reference models/clip_embedder.py:52,
 * dataset_creation/finetune/embed_dataset.py:20, preprocessing/dataset_preprocessing.py:193,
 * dataset_creation/benchmark/benchmark_dataset.py:99): resize shorter edge to 336 with Pillow's fixed-point BICUBIC,
 * centre crop 336x336, float32 /255.0, (x-mean)/std, HWC->CHW -- bit-exact with Pillow + numpy float32.
 * A handle is bound to one input geometry (in_h, in_w); coefficient tables live on the device.
 * ------------------------------------------------------------------------------------------------ */
typedef struct pg_prep pg_prep;
int pg_prep_create(pg_prep** out, int device, int in_h, int in_w);
int pg_prep_destroy(pg_prep* h);
/* out6 = { resized_h, resized_w, crop_top, crop_left, first_source_row, source_rows_used } */
int pg_prep_geometry(const pg_prep* h, int32_t* out6);
int pg_prep_workspace_bytes(const pg_prep* h, int n_images, size_t* bytes);
/* images_u8: DEVICE (n_images, in_h, in_w, 3) uint8 RGB; out: DEVICE (n_images,3,336,336) fp32 (PG_DTYPE_F32) or fp16
 * (PG_DTYPE_F16, the fp32 value rounded to nearest even -- exactly what the encoder's im2col would do to the fp32 value).
 * Asynchronous on `stream`. */
int pg_prep_forward(pg_prep* h, const void* images_u8, int n_images, void* out, int out_dtype, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The one collective of the path: all-gather over RCCL (xGMI inside a node).
 * Replaces: `accelerator.gather(index)` / `accelerator.gather(output)` at preprocessing/embed.py:36-37 (rank-major
 * concatenation on every rank).  One process per GPU; rank 0 creates the PG_COMM_ID_BYTES-byte id and the host program
 * ships it to the other ranks over its own bootstrap channel (a TCP store, MPI, a file ...); all ranks then call
 * pg_comm_init_rank concurrently with the HIP device they own set current.  RCCL is bound at run time
 * (dlopen "librccl.so.1"; env PIGEON_RCCL_LIB overrides), so a process that already carries an RCCL keeps a single copy.
 *   send  DEVICE bytes_per_rank bytes; recv DEVICE nranks * bytes_per_rank bytes (rank r's block at r * bytes_per_rank);
 *   asynchronous on `stream`, no host synchronisation. */
#define PG_COMM_ID_BYTES 128
int pg_comm_unique_id(void* id_out /*[PG_COMM_ID_BYTES]*/);
int pg_comm_init_rank(void** comm, int nranks, const void* unique_id /*[PG_COMM_ID_BYTES]*/, int rank);
int pg_comm_count(void* comm, int* nranks);          /* ranks as RCCL sees them */
int pg_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);
/* `count` buffers gathered in one RCCL group (one fused launch); buffer i: send[i] bytes_per_rank[i] bytes ->
 * recv[i] nranks * bytes_per_rank[i] bytes, rank-major.  The data-parallel step gathers embeddings, candidate cells /
 * probabilities, initial predictions and sample indices this way. */
int pg_allgather_many(void* comm, int count, const void* const* send, void* const* recv, const size_t* bytes_per_rank,
                      void* stream);
int pg_comm_destroy(void* comm);
int pg_comm_rccl_version(void);                       /* NCCL_VERSION_CODE of the bound library, < 0 on failure */

/* ------------------------------------------------------------------------------------------------
 * Around the hot path.
 * ------------------------------------------------------------------------------------------------ */
/* Prototype construction (reference models/proto_refiner.py:359-384): proto_emb[p] = fp32 mean, in member order, of
 * train_emb[member_idx[member_off[p] .. member_off[p+1])], each member first averaged over its `panels` (1 or 4) panel
 * embeddings; an empty prototype gets zeros.  train_emb: DEVICE (num_train, panels, 1024) fp32; proto_emb: DEVICE
 * (num_protos,1024) fp32.  Bit-identical to torch's `embeddings.mean(dim=1).mean(dim=0)` on the CPU. */
int pg_proto_build(const float* train_emb, int panels, int64_t num_train, const int64_t* member_off,
                   const int64_t* member_idx, int64_t num_protos, float* proto_emb, void* stream);
/* Great-circle distance matrix (reference preprocessing/geo_utils.py:58-74): x DEVICE (N,2) [lng,lat] degrees, fp32
 * (PG_DTYPE_F32: deg2rad / cos(lat) in fp32 then promoted, torch's type promotion) or fp64; y DEVICE (M,2) fp64 (note:
 * row-major (M,2), i.e. lla_geocells itself, where the reference passes its transpose); out DEVICE (N,M) fp64 km. */
int pg_haversine_matrix(const void* x, int x_dtype, const double* y, int N, int M, double* out, void* stream);
/* Row-paired great-circle distance (reference preprocessing/geo_utils.py:40-55 `haversine(x, y)`): x DEVICE (N,2) fp64,
 * y DEVICE (N,2) fp32 or fp64 (fp32: deg2rad / cos(lat) of y in fp32, then promoted -- the dtypes of the refiner's veto
 * call, models/proto_refiner.py:198-202); out DEVICE (N,) fp64 km. */
int pg_haversine_pairs(const double* x, const void* y, int y_dtype, int64_t N, double* out, void* stream);
/* Label smoothing (reference preprocessing/utils.py:7-19): out = exp(-(d - rowmin(d)) / constant), NaN/inf -> 0.
 * distances, out: DEVICE (N,M) fp64 (may alias). */
int pg_smooth_labels(const double* distances, int N, int M, double constant, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Building-block ops (exported so the parity tests can check every kernel in isolation through the ABI).
 * ------------------------------------------------------------------------------------------------ */
/* `dtype` below is the 16-bit operand/activation format: PG_DTYPE_F16 or PG_DTYPE_BF16.
 * C = A (M,K 16-bit, row stride lda) x W^T (N,K 16-bit) with fp32 accumulation and a fused epilogue:
 *   epi 0: out 16-bit (M,ldc) = acc + bias; columns < qcols additionally scaled by qscale     (QKV)
 *   epi 1: out 16-bit = quick_gelu(acc + bias) = x*sigmoid(1.702x)                            (fc1)
 *   epi 2: out fp32 (M,ldc) += acc + bias  (in-place residual add)                            (out_proj, fc2)
 *   epi 3: patch embed: row = img*576+p -> out fp32 row (img*577+1+p) = acc + aux[(1+p)*N + col] (aux = pos emb)
 *   epi 4: out fp32 = acc + bias (bias may be NULL)                                           (tests)
 * N must be a multiple of 256, K a multiple of 64.  variant selects the tile configuration (0 = default). */
int pg_op_gemm16(int dtype, const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldc,
                 int M, int N, int K, int epi, float qscale, int qcols, const float* aux,
                 int variant, void* stream);
/* Same, with an explicit row stride ldw (elements, >= K, multiple of 8) for W: padded strides keep the rows of an
 * operand panel off the same L2 / memory channels (power-of-two strides camp on a few channels). */
int pg_op_gemm16_ld(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out,
                    int64_t ldc, int M, int N, int K, int epi, float qscale, int qcols, const float* aux,
                    int variant, void* stream);
/* The "LayerNorm folded into the GEMM" building blocks the encoder uses (persistent kernel only: N % 256 == 0,
 * K % 128 == 0; see csrc/vit.hip):
 *   pg_op_rowstat_cast      x fp32 (rows,1024) -> x16 (rows,1024) 16-bit copy, rowstat (rows,2) = (rstd, mean*rstd)
 *   pg_op_gemm16_resid_stat X (M,ldc) fp32 += A.W^T + bias; x16 (M,ldx) = 16-bit copy of the new rows; statpart
 *                           (N/64, M, 2) fp32 = per-64-column (sum, sum of squares) of the new rows; ldx == ldc
 *   pg_op_rowstat_finalize  statpart -> rowstat (rstd, mean*rstd) over `slots` slices of a 1024-wide row
 *   pg_op_gemm16_ln         out 16-bit = epi(rowstat[m].rstd * acc - rowstat[m].mean_rstd * colsum[n] + bias[n]),
 *                           epi 6 = QKV form (columns < qcols scaled by qscale), 7 = QuickGELU form.
 * Memory contract of the fp32 residual epilogues (epi 2 of pg_op_gemm16, pg_op_gemm16_resid_stat): the last, partial row tile of the
 * persistent kernels READS (never writes) up to 383 rows past row M of X / out -- the row term of those loads rides in the SGPR
 * offset, which the buffer bounds check does not cover.  X must be followed by at least 384 * ldc * 4 readable bytes (inside
 * pg_vit_forward it is: the next buffer of the workspace). */
int pg_op_rowstat_cast(const float* x, void* x16, int dtype, float* rowstat, int64_t rows, float eps, void* stream);
int pg_op_gemm16_resid_stat(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* X,
                            int64_t ldc, void* x16, int64_t ldx, float* statpart, int M, int N, int K, int variant, void* stream);
int pg_op_rowstat_finalize(const float* statpart, int slots, float* rowstat, int64_t rows, float eps, void* stream);
int pg_op_gemm16_ln(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* colsum,
                    const float* rowstat, void* out, int64_t ldc, int M, int N, int K, int epi, float qscale, int qcols,
                    int variant, void* stream);
/* y = LayerNorm(x) over the last dim (1024), eps, gamma/beta fp32.  x fp32 (rows,1024).
 * out_dtype PG_DTYPE_F16/BF16 -> y 16-bit (rows,1024); PG_DTYPE_F32 -> fp32 (may alias x). */
int pg_op_layernorm(const float* x, const float* gamma, const float* beta, void* y, int out_dtype,
                    int64_t rows, float eps, void* stream);
/* Multi-head attention over the fused QKV buffer (n_images*577, 3072) 16-bit -> out (n_images*577,1024) 16-bit.
 * Q must already carry the factor log2(e)/sqrt(64) (the GEMM epilogue applies it); softmax in fp32. */
int pg_op_attention(int dtype, const void* qkv, void* out, int n_images, void* stream);
/* fp32/fp16/bf16 NCHW pixels -> 16-bit patch matrix (n_images*576, 640), k = c*196+ky*14+kx, cols 588..639 zero. */
int pg_op_im2col(const void* pixels, int pix_dtype, void* out, int out_dtype, int n_images, void* stream);
/* mean over the 577 tokens: x fp32 (n_images,577,1024) -> (n_images,1024). */
int pg_op_token_mean(const float* x, float* out, int n_images, void* stream);
/* fp32 -> fp16/bf16 (round to nearest even; fp16 saturates), n elements. */
int pg_op_cast_f32(const float* x, void* y, int out_dtype, int64_t n, void* stream);
/* Exact-mode building blocks (csrc/precise.hip).  A "triple" of an fp32 row of `cols` values is the fp16 row
 * [hi | lo | hi * 2^-8] of 3 * cols values (hi = fp16(x), lo = fp16(x - hi)); multiplied by weight rows [Wh | Wh | (W - Wh) * 2^8]
 * through pg_op_gemm16 (epi 2 / 3 / 4, K = 3 * cols) it yields the fp32-grade product.
 *   pg_op_x3_split      x fp32 (rows,cols) -> triple (rows, 3 cols); gelu != 0: through QuickGELU first (expf, IEEE division)
 *   pg_op_x3_layernorm  LayerNorm(x fp32 (rows,1024)) -> triple (rows, 3072)
 *   pg_op_attention_f32 fused QKV fp32 (n_images*577, 3072), Q NOT pre-scaled -> softmax(q k^T / 8) v, fp32 (n_images*577, 1024) */
/* S independent products in ONE persistent launch (the exact mode's K-split GEMMs, csrc/vit.hip precise_gemm): part p computes
 * parts[p] (M,N) fp32 = A[:, p*Kp:(p+1)*Kp] x W[:, p*Kp:(p+1)*Kp]^T (+ bias for p = 0); A (M, >= S*Kp) and W (N, >= S*Kp) 16-bit with
 * leading dimensions lda / ldw.  N % 256 == 0, Kp % 128 == 0, 1 <= S <= 8.  Each part is bit-identical to pg_op_gemm16_ld on its slice. */
int pg_op_gemm16_parts(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* parts,
                       int M, int N, int Kp, int S, void* stream);
int pg_op_x3_split(const float* x, void* y3, int64_t rows, int cols, int gelu, void* stream);
int pg_op_x3_layernorm(const float* x, const float* gamma, const float* beta, void* y3, int64_t rows, float eps, void* stream);
int pg_op_attention_f32(const float* qkv, float* out, int n_images, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PIGEON_HIP_H */
