// gemm_epi.h -- argument block and fused epilogue element functions shared by the GEMM kernels
// (gemm_bf16.hip: one-tile-per-block kernels; gemm_pp.hip: persistent ping-pong kernel).
#pragma once
#include "common.h"
#include "pigeon_internal.h"

#define BK 64
#define ROWB 128   // bytes per LDS row (BK bf16)

struct GemmArgs {
    const uint16_t* A; int64_t lda;
    const uint16_t* W; int64_t ldw;   // [N][K], row stride ldw elements
    const float* bias;            // [N] or null
    void* out; int64_t ldc;
    int M, N, K;
    float qscale; int qcols;
    const float* aux;             // epi 3: position embedding [577][N]
    int tilesM, tilesN, ntiles;
    int part_tiles;               // tiles of ONE part (tilesM x tilesN); ntiles = ex.parts x part_tiles (EPI_F32, gemm_pp.hip)
    int gn;                       // N tiles per raster group (see tile_coords)
    int stagger;                  // gemm_pp (tools build): shader cycles of one output tile, per-CU start stagger (0 = off)
    int xcd_stagger_ticks;        // persistent kernels: XCD x starts x * ticks / 8 wall-clock ticks (100 MHz) late (0 = off)
    PgGemmExtra ex;               // LayerNorm-fold epilogues (EPI_RESID_STAT / EPI_QKV_LN / EPI_GELU_LN)
};

// gemm_pp.hip: persistent ping-pong kernel (variants 30..39); tilesM/tilesN/ntiles are filled in by the callee
int pg_gemm_pp_launch(int dtype, GemmArgs g, int epi, int variant, hipStream_t s);
// gemm_pp6.hip: the same kernel with a 384 x 256 block tile, 16-bit-output epilogues only (variant 56, experimental)
bool pg_gemm_pp6_supported(int epi, int N, int K);
int pg_gemm_pp6_launch(int dtype, GemmArgs g, int epi, hipStream_t s);

// XCD-level start stagger of the persistent GEMMs.  Every tile of a launch takes the same time, so blocks that start
// together reach their epilogues together: all 256 CUs then hit HBM at once (the fp32 residual read-modify-write of an
// out-proj / fc2 tile is 640 KB per CU, 164 MB per round) while the matrix pipes idle, and during the mainloops HBM idles.
// Delaying XCD x by x/8 of a tile period keeps the 32 CUs of an XCD in lock step -- they share operand panels through
// their L2 at the same K position, which the per-CU stagger tried in round 1 destroyed -- but lets one XCD's epilogue run
// under the other XCDs' mainloops.  The launch ends with the partial last round anyway (its few tiles go to XCD 0, which is
// not delayed), so a spread below one tile period adds no tail.  Wall clock (100 MHz s_memrealtime), immune to DVFS.
__device__ __forceinline__ void xcd_stagger_wait(int ticks) {
    if (ticks <= 0) return;
    const int xcd = blockIdx.x & 7;
    if (xcd == 0) return;
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)ticks * xcd / 8;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(16);
}

// gemm_w4.hip (tools build only, variant 64, experimental): persistent kernel with one wave per SIMD (4 waves, 128 x 128 wave
// tiles, accumulators in AGPRs).  Level with the ping-pong kernels per K tile, slower end to end.
#ifdef PIGEON_ABLATIONS
bool pg_gemm_w4_supported(int epi, int N, int K);
int pg_gemm_w4_launch(int dtype, GemmArgs g, int epi, hipStream_t s);
#endif

// gemm_tail.hip: rows [m_begin, M) of a problem in 32 x 64 one-wave tiles, bit-identical to the persistent kernels (variant 70
// runs a whole problem through it; pg_gemm_launch uses it for the rows that do not fill the persistent kernels' last round)
bool pg_gemm_tail_supported(int epi, int N, int K);
int pg_gemm_tail_launch(int dtype, GemmArgs g, int epi, int m_begin, hipStream_t s);
// gemm_mid.hip (round 6): a whole problem in 128 x 128 one-tile-per-block tiles through a 3-stage LDS ring, bit-identical to the
// persistent kernels (variant 71 forces it; pg_gemm_launch picks it for batches too small to fill the persistent kernels' first round)
bool pg_gemm_mid_supported(int epi, int N, int K);
int pg_gemm_mid_launch(int dtype, GemmArgs g, int epi, hipStream_t s, int m_begin = 0);

// Tools build only: wall-clock stamps (100 MHz) from inside the persistent kernels, blocks 0 and 100, every wave, first 16 tiles:
// buf[((blk * 16 + tile) * 8 + wave) * 12 + slot].  Armed by pg_dbg_timestamps(buf) (gemm_bf16.hip), read by tools/epi_timeline.py.
#ifdef PIGEON_ABLATIONS
#define PG_TS(g, iter, wave, slot)                                                                                             \
    do {                                                                                                                       \
        if ((g).stagger == -7 && (blockIdx.x == 0 || blockIdx.x == 100) && (threadIdx.x & 63) == 0 && (iter) < 16)             \
            ((unsigned long long*)(g).aux)[(((blockIdx.x ? 1 : 0) * 16 + (iter)) * 8 + (wave)) * 12 + (slot)] =                \
                __builtin_amdgcn_s_memrealtime();                                                                              \
    } while (0)
#else
#define PG_TS(g, iter, wave, slot) do {} while (0)
#endif

__device__ __forceinline__ void glds16(const void* gptr, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
}

// QuickGELU x * sigmoid(1.702 x) (modeling_clip.py QuickGELUActivation) as mul, v_exp_f32 (2^x), add, v_rcp_f32, mul:
// the IEEE division of the obvious form expands to ~10 VALU instructions and made the fc1 epilogue cost ~6 us per tile.
__device__ __forceinline__ float quick_gelu(float v) {
    const float e = __builtin_amdgcn_exp2f(-2.4554669595930157f * v);      // exp(-1.702 v); 1.702 * log2(e)
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// Packed-fp32 forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE fp32 operations per instruction, bit-identical to the
// scalar ones).  They run at the plain VALU rate when no MFMA stream is active on the CU (tools/pipe_rate.hip: 5 cycles alone, 37
// next to MFMAs -- they share the matrix pipe), which is exactly the situation of a persistent kernel's epilogue: the 16-bit
// epilogues are VALU-bound there (gemm_pp6.hip, tools/epi_timeline.py), so halving their fma / mul / add count is time saved.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 quick_gelu2(f32x2 v) {
    const f32x2 t = v * -2.4554669595930157f;
    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = 1.0f + e;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return v * r;
}
// y = acc * rstd + (-(mean rstd) * colsum + c) on 4 columns, as two packed pairs (same two roundings per element as the fmaf form)
__device__ __forceinline__ f32x4 ln_fold4(f32x4 acc, float rstd, float mrs, const f32x4& colsum, const f32x4& c) {
    const f32x2 r2 = {rstd, rstd}, m2 = {-mrs, -mrs};
    const f32x2 a0 = pk_fma(f32x2{acc[0], acc[1]}, r2, pk_fma(m2, f32x2{colsum[0], colsum[1]}, f32x2{c[0], c[1]}));
    const f32x2 a1 = pk_fma(f32x2{acc[2], acc[3]}, r2, pk_fma(m2, f32x2{colsum[2], colsum[3]}, f32x2{c[2], c[3]}));
    return f32x4{a0[0], a0[1], a1[0], a1[1]};
}
__device__ __forceinline__ f32x4 quick_gelu4(f32x4 v) {
    const f32x2 a0 = quick_gelu2(f32x2{v[0], v[1]}), a1 = quick_gelu2(f32x2{v[2], v[3]});
    return f32x4{a0[0], a0[1], a1[0], a1[1]};
}

// ==== ONE definition of the fused epilogue ARITHMETIC for the slab-transposing kernels (gemm_pp / gemm_pp6 / gemm_tail, and
// gemm_w4 in the tools build).  Round 2 kept a hand copy of these expressions in every file and held them together with
// bit-compare tests only; a row's value must not depend on which kernel (persistent tile, tail tile) computed it, so the
// expressions -- including the association order of the row statistics -- live here and nowhere else.  A lane holds 8 outputs of
// one row as two f32x4 (`lo`, `hi`); which columns those are (8 consecutive, or 4k.. and 32+4k.. for EPI_RESID_STAT) is the
// caller's geometry, the arithmetic does not depend on it.
template <int EPI> constexpr bool epi_is_ln() { return EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool epi_is_qkv() { return EPI == EPI_QKV || EPI == EPI_QKV_LN; }
template <int EPI> constexpr bool epi_is_out16() { return EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }

// 16-bit-output epilogues (EPI_QKV, EPI_GELU and their LayerNorm-fold forms): 8 accumulators -> 8 packed 16-bit outputs.
//   plain: y = acc + b;   LN fold: y = rstd * acc - (mean rstd) * colsum + c   (c = beta.W^T + b, s = colsum)
//   QKV:   the Q strip (q_strip: the tile's columns are below qcols -- qcols is a multiple of 8, so a lane's 8 columns are all in
//          or all out) is scaled by qsc; K / V strips skip the multiply;   GELU: QuickGELU.
// (rstd, mrs) must come through registers of their own (callers move each half of the loaded pair through an asm v_mov:
// hipcc, ROCm 7.2, SLP-packs fmas whose multipliers are the two halves of one dwordx2 and drops the op_sel of the high half).
template <typename T, int EPI>
__device__ __forceinline__ u32x4 epi16_finish(f32x4 lo, f32x4 hi, const f32x4& b_lo, const f32x4& b_hi, const f32x4& s_lo,
                                              const f32x4& s_hi, float rstd, float mrs, bool q_strip, float qsc) {
    if constexpr (epi_is_ln<EPI>()) {
        lo = ln_fold4(lo, rstd, mrs, s_lo, b_lo);
        hi = ln_fold4(hi, rstd, mrs, s_hi, b_hi);
    } else {
        lo += b_lo; hi += b_hi;
    }
    if constexpr (epi_is_qkv<EPI>()) {
        if (q_strip) { lo *= qsc; hi *= qsc; }
    } else {
        lo = quick_gelu4(lo); hi = quick_gelu4(hi);
    }
    u32x4 pk;
    pk[0] = pack16x2<T>(lo[0], lo[1]); pk[1] = pack16x2<T>(lo[2], lo[3]);
    pk[2] = pack16x2<T>(hi[0], hi[1]); pk[3] = pack16x2<T>(hi[2], hi[3]);
    return pk;
}

// fp32 residual epilogues (EPI_RESID, EPI_RESID_STAT): the new residual values of 4 columns
__device__ __forceinline__ f32x4 epi_resid4(f32x4 x, const f32x4& acc, const f32x4& b) {
    x += acc + b;
    return x;
}
// EPI_RESID_STAT: 16-bit copy of 4 new residual values
template <typename T>
__device__ __forceinline__ u32x2 epi_copy16x4(const f32x4& x) {
    u32x2 h;
    h[0] = pack16x2<T>(x[0], x[1]); h[1] = pack16x2<T>(x[2], x[3]);
    return h;
}
// EPI_RESID_STAT: this lane's share (its 8 new values x, y) of the row's partial (sum, sum of squares) over 64 columns -- the
// association order is part of the result; the 8 lanes of a row are then combined by row8_sum below.
__device__ __forceinline__ void epi_stat8(const f32x4& x, const f32x4& y, float& s1, float& s2) {
    s1 = ((x[0] + x[1]) + (x[2] + x[3])) + ((y[0] + y[1]) + (y[2] + y[3]));
    s2 = ((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) + ((y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]));
}
// sum over the 8 lanes (lane & 7 = 0..7) that hold one row, fixed association: pairs, quads, then the two quads
template <int CTRL>
__device__ __forceinline__ float epi_dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row8_sum(float v) {
    v += epi_dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    v += epi_dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    v += epi_dpp_mov<0x141>(v);      // row_half_mirror: lane i <-> 7 - i inside each group of 8
    return v;
}

// Apply the epilogue to 4 consecutive columns [col, col+4) of one output row (fp32-out epilogues).
template <int EPI>
__device__ __forceinline__ void epi_store_f32x4(const GemmArgs& g, int row, int col, f32x4 v, const f32x4& b4) {
    if (EPI == EPI_RESID) {
        float* p = (float*)g.out + (int64_t)row * g.ldc + col;
        f32x4 x = *(const f32x4*)p;
        x += v + b4;
        *(f32x4*)p = x;
    } else if (EPI == EPI_PATCH) {
        const int img = row / VIT_PATCHES, p = row - img * VIT_PATCHES;
        float* o = (float*)g.out + ((int64_t)img * VIT_TOKENS + 1 + p) * g.ldc + col;
        const f32x4 pos = *(const f32x4*)(g.aux + (int64_t)(1 + p) * g.N + col);
        *(f32x4*)o = v + pos;
    } else {  // EPI_F32
        *(f32x4*)((float*)g.out + (int64_t)row * g.ldc + col) = v + b4;
    }
}

// 16-bit-out epilogues on 8 consecutive columns.
template <typename T, int EPI>
__device__ __forceinline__ void epi_store_bf16x8(const GemmArgs& g, int row, int col, f32x4 lo, f32x4 hi,
                                                 const f32x4& b_lo, const f32x4& b_hi) {
    lo += b_lo; hi += b_hi;
    if (EPI == EPI_QKV) {
        if (col < g.qcols) { lo *= g.qscale; hi *= g.qscale; }   // qcols is a multiple of 8
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = quick_gelu(lo[e]); hi[e] = quick_gelu(hi[e]); }
    }
    u32x4 pk;
    pk[0] = pack16x2<T>(lo[0], lo[1]); pk[1] = pack16x2<T>(lo[2], lo[3]);
    pk[2] = pack16x2<T>(hi[0], hi[1]); pk[3] = pack16x2<T>(hi[2], hi[3]);
    *(u32x4*)((uint16_t*)g.out + (int64_t)row * g.ldc + col) = pk;
}

template <typename T, int EPI>
__device__ __forceinline__ void epi_store_scalar(const GemmArgs& g, int row, int col, float v) {
    if (EPI == EPI_QKV) {
        if (g.bias) v += g.bias[col];
        if (col < g.qcols) v *= g.qscale;
        ((uint16_t*)g.out)[(int64_t)row * g.ldc + col] = T::bits(v);
    } else if (EPI == EPI_GELU) {
        v = quick_gelu(v + g.bias[col]);
        ((uint16_t*)g.out)[(int64_t)row * g.ldc + col] = T::bits(v);
    } else if (EPI == EPI_RESID) {
        float* p = (float*)g.out + (int64_t)row * g.ldc + col;
        *p = *p + (v + g.bias[col]);
    } else if (EPI == EPI_PATCH) {
        const int img = row / VIT_PATCHES, p = row - img * VIT_PATCHES;
        float* o = (float*)g.out + ((int64_t)img * VIT_TOKENS + 1 + p) * g.ldc + col;
        *o = v + g.aux[(int64_t)(1 + p) * g.N + col];
    } else {
        if (g.bias) v += g.bias[col];
        ((float*)g.out)[(int64_t)row * g.ldc + col] = v;
    }
}

