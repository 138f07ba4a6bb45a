// pigeon_internal.h -- declarations shared between the translation units of libpigeon_hip.so (not exported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pigeon_hip.h"

// GEMM epilogues.  5..7 are the "LayerNorm folded into the GEMM" forms (gemm_pp.hip only, see vit.hip):
//   EPI_RESID_STAT  fp32 residual add as EPI_RESID, plus the 16-bit copy of the new residual row (the next GEMM's A
//                   operand) and per-(row, 64-column slice) partial sums / sums of squares for the row statistics;
//   EPI_QKV_LN / EPI_GELU_LN  out = epi(rstd[m] * acc - (mean*rstd)[m] * colsum[n] + bias[n]) where the weight already
//                   carries gamma and bias carries beta.W^T + b.
// 8 (round 6, gemm_pp.hip's 256 x 256 kernel only; the exact mode's fc1):
//   EPI_GELU_X3     out = the split-fp16 triple [M][3N] (ldc = 3N fp16 elements) of QuickGELU(acc + bias) with the accurate expf / IEEE
//                   division: what EPI_F32 followed by split_x3_kernel<true> (precise.hip) writes, bit for bit, without the fp32 round trip.
enum { EPI_QKV = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3, EPI_F32 = 4, EPI_RESID_STAT = 5, EPI_QKV_LN = 6, EPI_GELU_LN = 7,
       EPI_GELU_X3 = 8 };

struct PgGemmExtra {
    const float* colsum = nullptr;     // [N]   EPI_*_LN: row sums of the (rounded) gamma-folded weight
    const float* rowstat = nullptr;    // [M][2] EPI_*_LN: (rstd, mean*rstd) per A row
    void* x16 = nullptr;               // [M][ldx] EPI_RESID_STAT: 16-bit copy of the updated residual rows
    int64_t ldx = 0;
    float* statpart = nullptr;         // [N/64][stat_rows][2] EPI_RESID_STAT: partial (sum, sum of squares) per 64-column slice
    int64_t stat_rows = 0;             // rows per slice of statpart; 0 = M (pg_gemm_launch fills it in before it splits a problem)
    // EPI_F32 on the 256 x 256 persistent kernel only (exact mode, round 5): the problem is `parts` independent products that share
    // M, N, K and the leading dimensions -- part p reads A + p * a_part and W + p * w_part (elements) and writes out + p * c_part
    // (floats); the bias goes into part 0 only.  All parts' tiles form ONE persistent launch (parts x tilesM x tilesN tiles).
    int parts = 1;
    int64_t a_part = 0, w_part = 0, c_part = 0;
};

void pg_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int pg_check_launch(const char* what);          // hipGetLastError -> PG_EHIP + message
int pg_default_gemm_variant();                   // env PIGEON_GEMM_VARIANT or the built-in default
int pg_gemm_block_cap();                         // env PIGEON_GEMM_BLOCKS: cap on the persistent GEMMs' grid (0 = one block per CU)
int pg_num_cus();                               // compute units of the current device (256 on MI355X)
int pg_gemm_tail_min_n();                        // env PIGEON_GEMM_TAIL_MIN_N: ... or smallest N
int pg_gemm_tail_min_k();                        // env PIGEON_GEMM_TAIL_MIN_K: smallest K for which the tail split is used
int pg_gemm_tail_rows();                         // env PIGEON_GEMM_TAIL_ROWS / pg_tune_gemm_tail_rows: most rows handed to gemm_tail.hip (0 = never)
int pg_gemm_raster_gn();                       // pg_tune_gemm_raster / env PIGEON_GEMM_RASTER_GN: N tiles per raster group of the 384 x 256 kernel (0 = default 4, -1 = all)
bool pg_gemm_route_pp256();                     // the small-batch routing may also swap the 384 x 256 kernel for the 256 x 256 one (PIGEON_GEMM_MID=2: no)
bool pg_gemm_mid_on();                          // pg_tune_gemm_mid / env PIGEON_GEMM_MID: small batches through gemm_mid.hip when the cost model says so
unsigned long long pg_tune_epoch();             // bumped by every pg_tune_* call: captured hipGraphs of an older epoch are stale
float pg_gemm_stagger_fraction();                // env PIGEON_GEMM_STAGGER / pg_tune_gemm_stagger: XCD start spread, fraction of a tile period

#define PG_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            pg_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PG_EHIP;                                                                   \
        }                                                                                     \
    } while (0)

// gemm_bf16.hip
int pg_gemm_launch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldc,
                   int M, int N, int K, int epi, float qscale, int qcols, const float* aux, int variant,
                   hipStream_t s, const PgGemmExtra* extra = nullptr);
// rowops.hip
int pg_layernorm_launch(const float* x, const float* gamma, const float* beta, void* y, int out_dtype,
                        int64_t rows, float eps, hipStream_t s);
int pg_preln_launch(float* x, const float* cls, const float* pos0, const float* gamma, const float* beta,
                    int64_t rows, float eps, hipStream_t s, void* x16 = nullptr, int x16_dtype = 0, float* rowstat = nullptr);
int pg_im2col_launch(const void* pixels, int pix_dtype, void* out, int out_dtype, int n_images, hipStream_t s);
int pg_token_mean_launch(const float* x, float* out, int n_images, hipStream_t s);
int pg_cast_f32_launch(const float* x, void* y, int out_dtype, int64_t n, hipStream_t s);
int pg_rowstat_cast_launch(const float* x, void* x16, int out_dtype, float* rowstat, int64_t rows, float eps, hipStream_t s);
int pg_rowstat_finalize_launch(const float* statpart, int slots, float* rowstat, int64_t rows, float eps, hipStream_t s,
                               unsigned long long* alarm = nullptr, float alarm_sumsq = 0.f);
int pg_count_sat16_launch(const void* buf, int64_t rows, int cols, int64_t ld, int dtype, unsigned long long* counter, hipStream_t s);
// attention.hip
int pg_attention_launch(int dtype, const void* qkv, void* out, int n_images, hipStream_t s);
// precise.hip (exact mode)
int pg_x3_ln_launch(const float* x, const float* gamma, const float* beta, void* y3, int64_t rows, float eps, hipStream_t s);
int pg_x3_split_launch(const float* x, void* y3, int64_t rows, int C, int gelu, hipStream_t s);
int pg_x3_im2col_launch(const void* pixels, int pix_dtype, void* out3, int n_images, hipStream_t s);
int pg_attention_f32_launch(const float* qkv, float* out, int n_images, hipStream_t s);
bool pg_attention_x3out_available();
int pg_attention_x3out_launch(const float* qkv, void* out3, int n_images, hipStream_t s);
int pg_sum_parts_launch(const float* parts, int S, int64_t part_elems, float* dst, int64_t n, int resid, hipStream_t s);
