// x3.h -- the split-fp16 triple of the exact mode (precise.hip), shared with the GEMM epilogue that writes it (gemm_pp.hip
// EPI_GELU_X3): x = hi + lo (two fp16 halves, 22 significant bits), third piece hi * 2^-8 (the partner of the weights' Wl * 2^8).
#pragma once
#include "common.h"

#define X3_SHIFT_DOWN 0.00390625f          // 2^-8 on the activation side ...
#define X3_SHIFT_UP 256.0f                 // ... 2^8 on the weight side (vit.hip packs the weights with it)

struct X3 { uint16_t hi, lo, hs; };
__device__ __forceinline__ X3 x3_split(float v) {
    X3 r;
    r.hi = f32_to_f16_bits(v);                                   // RNE, saturating at +-65504
    const float hf = f16_bits_to_f32(r.hi);
    r.lo = __builtin_bit_cast(uint16_t, (_Float16)(v - hf));     // exact difference, |lo| <= ulp16(v) / 2
    r.hs = __builtin_bit_cast(uint16_t, (_Float16)(hf * X3_SHIFT_DOWN));
    return r;
}
// 4 values -> the three 8-byte pieces (hi, lo, hi * 2^-8) of 4 consecutive columns
__device__ __forceinline__ void x3_pack4(const f32x4& v, u32x2& hi, u32x2& lo, u32x2& hs) {
    const X3 a = x3_split(v[0]), b = x3_split(v[1]), d = x3_split(v[2]), e = x3_split(v[3]);
    hi = u32x2{(uint32_t)a.hi | ((uint32_t)b.hi << 16), (uint32_t)d.hi | ((uint32_t)e.hi << 16)};
    lo = u32x2{(uint32_t)a.lo | ((uint32_t)b.lo << 16), (uint32_t)d.lo | ((uint32_t)e.lo << 16)};
    hs = u32x2{(uint32_t)a.hs | ((uint32_t)b.hs << 16), (uint32_t)d.hs | ((uint32_t)e.hs << 16)};
}
// 4 consecutive columns c..c+3 of logical width C -> the triple row [hi | lo | hi * 2^-8] (3C fp16)
__device__ __forceinline__ void x3_store4(uint16_t* row, int C, int c, const f32x4& v) {
    u32x2 hi, lo, hs;
    x3_pack4(v, hi, lo, hs);
    *(u32x2*)(row + c) = hi;
    *(u32x2*)(row + C + c) = lo;
    *(u32x2*)(row + 2 * C + c) = hs;
}
// QuickGELU x * sigmoid(1.702 x) (modeling_clip.py QuickGELUActivation) with the accurate expf and an IEEE division (the fast
// path's v_exp / v_rcp forms are good to ~1e-6 relative, not to the last ulp).  ONE definition: split_x3_kernel<true> and the fused
// fc1 epilogue must give the same bits.
__device__ __forceinline__ float x3_quick_gelu(float v) { return v / (1.0f + expf(-1.702f * v)); }
