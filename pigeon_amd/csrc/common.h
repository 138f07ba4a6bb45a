// common.h -- shared device helpers for the gfx950 (CDNA4, wave64) kernels of libpigeon_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment: 8 bf16 = 4 VGPRs
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;   // same, fp16
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

#define PG_WAVE 64

// geometry of the ViT-L/14-336 vision tower (reference config.py:6-7)
#define VIT_TOKENS 577
#define VIT_PATCHES 576
#define VIT_HIDDEN 1024
#define VIT_HEADS 16
#define VIT_HEAD_DIM 64
#define VIT_MLP 4096
#define VIT_PATCH_K 588
#define VIT_PATCH_KPAD 640
#define VIT_IMG 336

// fp32 -> bf16 bits, round to nearest even (matches torch's .to(bfloat16)); NaN stays NaN
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}

__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) {
    f = fminf(fmaxf(f, -65504.f), 65504.f);              // saturate instead of overflowing to inf
    return __builtin_bit_cast(uint16_t, (_Float16)f);     // v_cvt_f16_f32, round to nearest even
}
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }

// The two 16-bit operand formats the MFMA path can run in (same MFMA rate on gfx950).  fp16 is the default:
// its 11-bit significand keeps the embedding within 1e-3 of the fp32 reference; bf16 (8 bits) measures 2e-3,
// dominated by WEIGHT rounding, which is coherent across tokens and survives the token mean.
struct T_BF16 {
    typedef bf16x8 v8;
    static constexpr int id = 1;
    static __device__ __forceinline__ uint16_t bits(float f) { return f32_to_bf16_bits(f); }
    static __device__ __forceinline__ float val(uint16_t b) { return bf16_bits_to_f32(b); }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // 16 x 16 x 32: half as many accumulator registers moved per flop as 32 x 32 x 16 -- 2.06-2.09 against 1.72-1.74 PFLOP/s at
    // the 1400 W package cap (tools/mfma_issue.hip, profiles/r02/mfma_issue.txt)
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    // the same as inline asm with the accumulator TIED ("+v"): written with the builtin, hipcc picks the early-clobber form of
    // the 4-pass MFMA for most of a phase's instructions (destination != source accumulator), which doubles the live
    // accumulator registers of a phase and spills (first 16x16x32 build of gemm_pp6.hip: 300-430 bytes of scratch per lane).
    // Inside asm the compiler pads no hazards: callers keep MFMA phases free of anything that reads their results.
    static __device__ __forceinline__ void mfma16_acc(f32x4& c, const v8& a, const v8& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma16_init(f32x4& c, const v8& a, const v8& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
    }
    // c + a.lo*b.lo + a.hi*b.hi on packed 16-bit pairs (v_dot2c_f32_bf16)
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
    }
    // two finite floats -> one dword, round to nearest even (v_cvt_pk_bf16_f32); no NaN fix-up: for softmax weights
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    }
};
struct T_F16 {
    typedef f16x8 v8;
    static constexpr int id = 2;
    static __device__ __forceinline__ uint16_t bits(float f) { return f32_to_f16_bits(f); }
    static __device__ __forceinline__ float val(uint16_t b) { return f16_bits_to_f32(b); }
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mfma16_acc(f32x4& c, const v8& a, const v8& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma16_init(f32x4& c, const v8& a, const v8& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
    }
    // two floats in [0, 65504] -> one dword, round to nearest even (v_cvt_pk_f16_f32); no saturation: for softmax weights
    static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
    }
};
// Two fp32 -> one dword of 16-bit values, same results as T::bits() on each half (round to nearest even; fp16 saturates at
// +-65504 instead of overflowing to inf), in 3 / 1 instructions per pair: v_med3_f32 x2 + v_cvt_pk_f16_f32, or
// v_cvt_pk_bf16_f32.  The scalar T::bits() forms cost 7 / 10 VALU operations per pair, which was ~1500 cycles of every
// 256x256 GEMM tile's epilogue (128 accumulator registers per wave).
template <typename T>
__device__ __forceinline__ uint32_t pack16x2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack16x2<T_F16>(float lo, float hi) {
    const f32x2 v = {__builtin_amdgcn_fmed3f(lo, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(hi, -65504.f, 65504.f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
template <>
__device__ __forceinline__ uint32_t pack16x2<T_BF16>(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 3-input max: nested __builtin_fmaxf folds to one v_max3_f32 (HIP's fmaxf() wrapper first canonicalises every input with
// a v_max_f32 x, x, x, which tripled the instruction count of the softmax row maximum).  Deliberately NOT inline asm: the
// compiler pads the MFMA-result -> VALU-read hazard (s_nop) only for instructions it can see; an asm v_max3_f32 reading a
// fresh accumulator gave wrong maxima as soon as the schedule placed it right behind the MFMA.
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap of a 1-D block id (cdna guide T1): hardware places block b on XCD b%8; give each
// XCD a contiguous chunk of logical ids so that neighbouring tiles (which share operand panels) share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int xcd = bid % NX, idx = bid / NX;
    int q = nwg / NX, r = nwg % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
