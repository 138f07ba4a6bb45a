// pigeon_internal.h -- declarations shared between the translation units of libpigeon_hip.so (not exported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pigeon_hip.h"

enum { EPI_QKV = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_PATCH = 3, EPI_F32 = 4 };

void pg_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int pg_check_launch(const char* what);          // hipGetLastError -> PG_EHIP + message
int pg_default_gemm_variant();                   // env PIGEON_GEMM_VARIANT or the built-in default

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
                   hipStream_t s);
// rowops.hip
int pg_layernorm_launch(const float* x, const float* gamma, const float* beta, void* y, int out_dtype,
                        int64_t rows, float eps, hipStream_t s);
int pg_preln_launch(float* x, const float* cls, const float* pos0, const float* gamma, const float* beta,
                    int64_t rows, float eps, hipStream_t s);
int pg_im2col_launch(const void* pixels, int pix_dtype, void* out, int out_dtype, int n_images, hipStream_t s);
int pg_token_mean_launch(const float* x, float* out, int n_images, hipStream_t s);
int pg_cast_f32_launch(const float* x, void* y, int out_dtype, int64_t n, hipStream_t s);
// attention.hip
int pg_attention_launch(int dtype, const void* qkv, void* out, int n_images, hipStream_t s);
