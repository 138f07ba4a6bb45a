// rowops.hip -- the HBM-bound row kernels of the ViT: LayerNorm (fp32 stats, bf16 or fp32 out), the
// CLS/position/pre-LayerNorm fix-up, the patch im2col, the token mean and a dtype cast.
//
// Replaces: nn.LayerNorm (modeling_clip.py CLIPEncoderLayer.layer_norm1/2, CLIPVisionTransformer.pre_layrnorm),
// the class-token concat + position add of CLIPVisionEmbeddings.forward (:202-218), the unfold half of the
// patch Conv2d, and torch.mean(last_hidden_state, dim=1) (reference models/clip_embedder.py:65,
// models/super_guessr.py:398).
//
// All of these are pure streaming kernels: one wave owns one 1024-float row (16 floats per lane as four
// 16-byte loads), statistics are reduced with cross-lane shuffles, no LDS.
#include "common.h"
#include "pigeon_internal.h"

// ---- LayerNorm over 1024 columns; one wave per row, 4 rows per 256-thread block ----------------------
template <int OUT>   // 0 fp32, else T id (1 bf16, 2 fp16)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ y,
                                                        int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * VIT_HIDDEN;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) * (1.0f / VIT_HIDDEN);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float var = wave_sum(q) * (1.0f / VIT_HIDDEN);      // biased variance, as torch
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 g4 = *(const f32x4*)(gamma + c);
        const f32x4 b4 = *(const f32x4*)(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
        if (OUT != 0) {
            u32x2 pk;
            if (OUT == 1) { pk[0] = pack16x2<T_BF16>(o[0], o[1]); pk[1] = pack16x2<T_BF16>(o[2], o[3]); }
            else { pk[0] = pack16x2<T_F16>(o[0], o[1]); pk[1] = pack16x2<T_F16>(o[2], o[3]); }
            *(u32x2*)((uint16_t*)y + row * VIT_HIDDEN + c) = pk;
        } else {
            *(f32x4*)((float*)y + row * VIT_HIDDEN + c) = o;
        }
    }
}

int pg_layernorm_launch(const float* x, const float* gamma, const float* beta, void* y, int out_dtype,
                        int64_t rows, float eps, hipStream_t s) {
    if (rows <= 0) return PG_OK;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (out_dtype == PG_DTYPE_BF16) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, gamma, beta, y, rows, eps);
    else if (out_dtype == PG_DTYPE_F16) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, gamma, beta, y, rows, eps);
    else if (out_dtype == PG_DTYPE_F32) hipLaunchKernelGGL(layernorm_kernel<0>, grid, block, 0, s, x, gamma, beta, y, rows, eps);
    else { pg_set_error("layernorm: bad output dtype %d", out_dtype); return PG_EINVAL; }
    return pg_check_launch("layernorm");
}

// ---- class token + position + pre_layrnorm, in place on the fp32 residual stream ----------------------
// Row r of x is token t = r % 577 of image r / 577.  Patch rows already hold conv + position (written by the
// patch GEMM epilogue); row t == 0 is synthesised here as class_embedding + position_embedding[0]
// (modeling_clip.py:212-217), then every row gets pre_layrnorm.
// STAT (LayerNorm-folded chain): the same pass also emits what layer 0's LN1 fold needs from the NEW row -- its 16-bit copy
// (the QKV GEMM's A operand) and (rstd, mean*rstd), two-pass like rowstat_cast_kernel (bit-identical to running that kernel
// afterwards, minus one 4 KB/row read of the residual stream).
template <typename T, bool STAT>
__global__ __launch_bounds__(256) void preln_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                    const float* __restrict__ pos0, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int64_t rows, float eps,
                                                    uint16_t* __restrict__ x16, float* __restrict__ rowstat) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool is_cls = (row % VIT_TOKENS) == 0;
    float* xr = x + row * VIT_HIDDEN;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + lane * 4;
        if (is_cls) v[i] = *(const f32x4*)(cls + c) + *(const f32x4*)(pos0 + c);
        else v[i] = *(const f32x4*)(xr + c);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) * (1.0f / VIT_HIDDEN);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / VIT_HIDDEN) + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 g4 = *(const f32x4*)(gamma + c);
        const f32x4 b4 = *(const f32x4*)(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
        *(f32x4*)(xr + c) = o;
        v[i] = o;
    }
    if (STAT) {
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s2 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean2 = wave_sum(s2) * (1.0f / VIT_HIDDEN);
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean2; q2 += d * d; }
        const float rstd2 = 1.0f / sqrtf(wave_sum(q2) * (1.0f / VIT_HIDDEN) + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 pk;
            pk[0] = pack16x2<T>(v[i][0], v[i][1]); pk[1] = pack16x2<T>(v[i][2], v[i][3]);
            *(u32x2*)(x16 + row * VIT_HIDDEN + i * 256 + lane * 4) = pk;
        }
        if (lane == 0) { rowstat[2 * row] = rstd2; rowstat[2 * row + 1] = mean2 * rstd2; }
    }
}

int pg_preln_launch(float* x, const float* cls, const float* pos0, const float* gamma, const float* beta,
                    int64_t rows, float eps, hipStream_t s, void* x16, int x16_dtype, float* rowstat) {
    if (rows <= 0) return PG_OK;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (!x16) hipLaunchKernelGGL((preln_kernel<T_F16, false>), grid, block, 0, s, x, cls, pos0, gamma, beta, rows, eps, nullptr, nullptr);
    else if (!rowstat) { pg_set_error("pre_layernorm: x16 without rowstat"); return PG_EINVAL; }
    else if (x16_dtype == PG_DTYPE_F16) hipLaunchKernelGGL((preln_kernel<T_F16, true>), grid, block, 0, s, x, cls, pos0, gamma, beta, rows, eps, (uint16_t*)x16, rowstat);
    else if (x16_dtype == PG_DTYPE_BF16) hipLaunchKernelGGL((preln_kernel<T_BF16, true>), grid, block, 0, s, x, cls, pos0, gamma, beta, rows, eps, (uint16_t*)x16, rowstat);
    else { pg_set_error("pre_layernorm: bad 16-bit dtype %d", x16_dtype); return PG_EINVAL; }
    return pg_check_launch("pre_layernorm");
}

// ---- row statistics for the LayerNorm-folded GEMMs (vit.hip, gemm_pp.hip EPI_*_LN) -----------------------------------
// rowstat_cast: x fp32 (rows,1024) -> 16-bit copy + (rstd, mean*rstd) per row, two-pass variance as layernorm_kernel.
// Used once per forward (the input of layer 0's LN1); later rows get their statistics from the residual GEMM epilogues.
template <typename T>
__global__ __launch_bounds__(256) void rowstat_cast_kernel(const float* __restrict__ x, uint16_t* __restrict__ x16,
                                                           float* __restrict__ rowstat, int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * VIT_HIDDEN;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) * (1.0f / VIT_HIDDEN);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / VIT_HIDDEN) + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32x2 pk;
        pk[0] = pack16x2<T>(v[i][0], v[i][1]); pk[1] = pack16x2<T>(v[i][2], v[i][3]);
        *(u32x2*)(x16 + row * VIT_HIDDEN + i * 256 + lane * 4) = pk;
    }
    if (lane == 0) { rowstat[2 * row] = rstd; rowstat[2 * row + 1] = mean * rstd; }
}

int pg_rowstat_cast_launch(const float* x, void* x16, int out_dtype, float* rowstat, int64_t rows, float eps, hipStream_t s) {
    if (rows <= 0) return PG_OK;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (out_dtype == PG_DTYPE_F16) hipLaunchKernelGGL(rowstat_cast_kernel<T_F16>, grid, block, 0, s, x, (uint16_t*)x16, rowstat, rows, eps);
    else if (out_dtype == PG_DTYPE_BF16) hipLaunchKernelGGL(rowstat_cast_kernel<T_BF16>, grid, block, 0, s, x, (uint16_t*)x16, rowstat, rows, eps);
    else { pg_set_error("rowstat_cast: bad dtype %d", out_dtype); return PG_EINVAL; }
    return pg_check_launch("rowstat_cast");
}

// rowstat_finalize: (sum, sum of squares) partials per 64-column slice, written slot-major [slot][row][2] by the
// EPI_RESID_STAT epilogue, summed in slice order -> (rstd, mean*rstd).  One thread per row (coalesced 8-byte reads).
// Range alarm (always on with fp16 operands): a row whose sum of squares reaches 65504^2 MAY hold an element that the 16-bit copy
// of the residual stream clamped (|x| >= 65504 implies it; the converse does not hold) -- one compare per row here, an atomic only
// when it fires.  The exact, per-element count is the debug scan pg_vit_saturation_check.
__global__ __launch_bounds__(256) void rowstat_finalize_kernel(const float* __restrict__ part, int slots, float* __restrict__ rowstat,
                                                               int64_t rows, float eps, unsigned long long* __restrict__ alarm,
                                                               float alarm_sumsq) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < slots; ++i) {
        const f32x2 v = *(const f32x2*)(part + ((int64_t)i * rows + row) * 2);
        s1 += v[0]; s2 += v[1];
    }
    const float mean = s1 * (1.0f / VIT_HIDDEN);
    const float var = fmaxf(s2 * (1.0f / VIT_HIDDEN) - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    rowstat[2 * row] = rstd;
    rowstat[2 * row + 1] = mean * rstd;
    if (alarm && !(s2 < alarm_sumsq)) atomicAdd(alarm, 1ull);         // (NaN rows count too)
}

int pg_rowstat_finalize_launch(const float* statpart, int slots, float* rowstat, int64_t rows, float eps, hipStream_t s,
                               unsigned long long* alarm, float alarm_sumsq) {
    if (rows <= 0) return PG_OK;
    hipLaunchKernelGGL(rowstat_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, statpart, slots, rowstat, rows, eps,
                       alarm, alarm_sumsq);
    return pg_check_launch("rowstat_finalize");
}

// ---- im2col of the 14x14 stride-14 patch stream -------------------------------------------------------
// One block per (image, patch-row py).  The 42 source rows (3 channels x 14 ky) of that patch row are each
// 336 contiguous pixels: reads are fully coalesced; every pixel lands at
//   out[(img*576 + py*24 + px)*640 + c*196 + ky*14 + kx]   (k order == Conv2d weight [1024,3,14,14] flattened)
// Columns 588..639 are zero so the GEMM can run K = 640 = 10 x 64.
template <typename PIX, typename T, bool PIX_F16 = false>
__global__ __launch_bounds__(256) void im2col_kernel(const PIX* __restrict__ pix, uint16_t* __restrict__ out) {
    const int img = blockIdx.x / 24, py = blockIdx.x % 24;
    const PIX* src = pix + (int64_t)img * 3 * VIT_IMG * VIT_IMG;
    uint16_t* dst = out + ((int64_t)img * VIT_PATCHES + py * 24) * VIT_PATCH_KPAD;
    for (int idx = threadIdx.x; idx < 42 * VIT_IMG; idx += 256) {
        const int line = idx / VIT_IMG, xcol = idx - line * VIT_IMG;   // line = c*14 + ky
        const int c = line / 14, ky = line - c * 14;
        const int px = xcol / 14, kx = xcol - px * 14;
        float v;
        if (sizeof(PIX) == 4) v = (float)src[((int64_t)c * VIT_IMG + py * 14 + ky) * VIT_IMG + xcol];
        else if (PIX_F16) v = f16_bits_to_f32((uint16_t)src[((int64_t)c * VIT_IMG + py * 14 + ky) * VIT_IMG + xcol]);
        else v = bf16_bits_to_f32((uint16_t)src[((int64_t)c * VIT_IMG + py * 14 + ky) * VIT_IMG + xcol]);
        dst[px * VIT_PATCH_KPAD + c * 196 + ky * 14 + kx] = T::bits(v);
    }
    for (int idx = threadIdx.x; idx < 24 * (VIT_PATCH_KPAD - VIT_PATCH_K); idx += 256) {
        const int px = idx / (VIT_PATCH_KPAD - VIT_PATCH_K), k = idx % (VIT_PATCH_KPAD - VIT_PATCH_K);
        dst[px * VIT_PATCH_KPAD + VIT_PATCH_K + k] = 0;
    }
}

int pg_im2col_launch(const void* pixels, int pix_dtype, void* out, int out_dtype, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    dim3 grid(n_images * 24), block(256);
    if (out_dtype != PG_DTYPE_F16 && out_dtype != PG_DTYPE_BF16) { pg_set_error("im2col: bad output dtype %d", out_dtype); return PG_EINVAL; }
    const bool h = out_dtype == PG_DTYPE_F16;
    if (pix_dtype == PG_DTYPE_F32) {
        if (h) hipLaunchKernelGGL((im2col_kernel<float, T_F16>), grid, block, 0, s, (const float*)pixels, (uint16_t*)out);
        else hipLaunchKernelGGL((im2col_kernel<float, T_BF16>), grid, block, 0, s, (const float*)pixels, (uint16_t*)out);
    } else if (pix_dtype == PG_DTYPE_BF16) {
        if (h) hipLaunchKernelGGL((im2col_kernel<uint16_t, T_F16>), grid, block, 0, s, (const uint16_t*)pixels, (uint16_t*)out);
        else hipLaunchKernelGGL((im2col_kernel<uint16_t, T_BF16>), grid, block, 0, s, (const uint16_t*)pixels, (uint16_t*)out);
    } else if (pix_dtype == PG_DTYPE_F16) {                  // what pg_prep_forward writes (preprocess.hip)
        if (h) hipLaunchKernelGGL((im2col_kernel<uint16_t, T_F16, true>), grid, block, 0, s, (const uint16_t*)pixels, (uint16_t*)out);
        else hipLaunchKernelGGL((im2col_kernel<uint16_t, T_BF16, true>), grid, block, 0, s, (const uint16_t*)pixels, (uint16_t*)out);
    } else { pg_set_error("im2col: unsupported pixel dtype %d", pix_dtype); return PG_EINVAL; }
    return pg_check_launch("im2col");
}

// ---- token mean: (n,577,1024) fp32 -> (n,1024) ---------------------------------------------------------
// Block = (image, 256-column slab); a thread owns one column and walks the 577 rows (coalesced across threads).
// Four partial sums, rows dealt round-robin: s_(t mod 4) += x[t] -- that order is part of the result.  Round 6: 16 rows are LOADED
// before they are added (16 loads in flight instead of 4; the kernel is latency-bound at small batches: one block per 256 columns of an
// image).  The additions and their order are unchanged -- bit-identical, pinned by test_token_mean_summation_order_is_pinned; the
// encoder's latency at one panorama did not move (3.86 ms), the launch is 1 % of it.
__global__ __launch_bounds__(256) void token_mean_kernel(const float* __restrict__ x, float* __restrict__ out) {
    const int img = blockIdx.x >> 2, col = (blockIdx.x & 3) * 256 + threadIdx.x;
    const float* p = x + (int64_t)img * VIT_TOKENS * VIT_HIDDEN + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = 0;
    for (; t + 16 <= VIT_TOKENS; t += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = p[(int64_t)(t + i) * VIT_HIDDEN];
#pragma unroll
        for (int i = 0; i < 16; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    }
    for (; t + 4 <= VIT_TOKENS; t += 4) {
        s0 += p[(int64_t)(t + 0) * VIT_HIDDEN];
        s1 += p[(int64_t)(t + 1) * VIT_HIDDEN];
        s2 += p[(int64_t)(t + 2) * VIT_HIDDEN];
        s3 += p[(int64_t)(t + 3) * VIT_HIDDEN];
    }
    for (; t < VIT_TOKENS; ++t) s0 += p[(int64_t)t * VIT_HIDDEN];
    out[(int64_t)img * VIT_HIDDEN + col] = ((s0 + s1) + (s2 + s3)) / (float)VIT_TOKENS;
}

int pg_token_mean_launch(const float* x, float* out, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    hipLaunchKernelGGL(token_mean_kernel, dim3(n_images * 4), dim3(256), 0, s, x, out);
    return pg_check_launch("token_mean");
}

// ---- fp32 -> bf16 cast ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void f32_to_16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) {
        const f32x4 v = *(const f32x4*)(x + i);
        u32x2 pk;
        pk[0] = pack16x2<T>(v[0], v[1]);
        pk[1] = pack16x2<T>(v[2], v[3]);
        *(u32x2*)(y + i) = pk;
    }
    if (i < n) for (int64_t j = i; j < n; ++j) y[j] = T::bits(x[j]);   // ragged tail (at most one thread)
}

int pg_cast_f32_launch(const float* x, void* y, int out_dtype, int64_t n, hipStream_t s) {
    if (n <= 0) return PG_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (out_dtype == PG_DTYPE_F16) hipLaunchKernelGGL(f32_to_16_kernel<T_F16>, dim3((unsigned)blocks), dim3(256), 0, s, x, (uint16_t*)y, n);
    else if (out_dtype == PG_DTYPE_BF16) hipLaunchKernelGGL(f32_to_16_kernel<T_BF16>, dim3((unsigned)blocks), dim3(256), 0, s, x, (uint16_t*)y, n);
    else { pg_set_error("cast: bad output dtype %d", out_dtype); return PG_EINVAL; }
    return pg_check_launch("cast_f32");
}

// ---- fp16 saturation scan (debug) ----------------------------------------------------------------------
// Every fp32 -> fp16 conversion of the encoder saturates at +-65504 (common.h) instead of producing inf.  A clamped
// conversion leaves exactly +-65504 (0x7bff / 0xfbff) behind; an un-clamped value rounds there only from the 16-wide
// interval [65488, 65520).  With pg_vit_saturation_check(h, 1) the forward pass scans every 16-bit activation buffer right
// after the kernel that wrote it and adds the number of such elements to a counter (pg_vit_saturation_read): 0 means no
// activation ever touched the fp16 range limit.  bf16 has the fp32 exponent range: the scan then counts +-inf.
__global__ __launch_bounds__(256) void count_sat16_kernel(const uint16_t* __restrict__ buf, int64_t rows, int cols, int64_t ld,
                                                          int is_f16, unsigned long long* __restrict__ counter) {
    const int64_t vec_per_row = cols / 8;
    const int64_t total = rows * vec_per_row;
    unsigned long long n = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vec_per_row, c = (i - r * vec_per_row) * 8;
        const u32x4 v = *(const u32x4*)(buf + r * ld + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t lo = v[e] & 0x7fffu, hi = (v[e] >> 16) & 0x7fffu;
            if (is_f16) { n += (lo == 0x7bffu) + (hi == 0x7bffu); }
            else { n += (lo == 0x7f80u) + (hi == 0x7f80u); }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(counter, n);
}

int pg_count_sat16_launch(const void* buf, int64_t rows, int cols, int64_t ld, int dtype, unsigned long long* counter, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return PG_OK;
    if (cols % 8 || ld % 8) { pg_set_error("count_sat16: cols / ld must be multiples of 8"); return PG_EINVAL; }
    hipLaunchKernelGGL(count_sat16_kernel, dim3(2048), dim3(256), 0, s, (const uint16_t*)buf, rows, cols, ld,
                       dtype == PG_DTYPE_F16 ? 1 : 0, counter);
    return pg_check_launch("count_sat16");
}
