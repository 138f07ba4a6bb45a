// attention_common.h -- geometry constants and the two device helpers shared by the product attention kernel (attention.hip, v8)
// and the superseded generations kept for A/B work in the tools build (tools/csrc/attention_old.hip).
#pragma once
#include "common.h"
#include "pigeon_internal.h"

#define ATT_KT 64
#define ATT_QB 128                       // query rows per block
#define ATT_NQB 5                        // ceil(577 / 128)
#define ATT_NT 10                        // ceil(577 / 64)
#define K_ROWB 128
#define VT_STRIDE 136                    // bytes per VT row: 64 keys * 2 B + 8 B pad (conflict-free b64 reads); v1 / v4 only
#define K_TILE_BYTES (ATT_KT * K_ROWB)   // 8192
#define VT_TILE_BYTES (64 * VT_STRIDE)   // 8704
#define QKV_LD 3072
// m is only raised when a tile's maximum exceeds the reference by more than 2^8 (P <= 256 is exact range for both 16-bit formats)
#define ATT_LAZY_THR 8.0f

typedef __attribute__((address_space(3))) void att_lds_void;

// Transposing LDS read as inline asm.  Written with the ds_read_tr builtin, hipcc puts an `s_waitcnt vmcnt(0)` in front of the
// first read: it assumes the read may alias the direct-to-LDS DMA of the NEXT tile issued at the top of the loop (other stage,
// never the same bytes), which parks the wave until that DMA has landed.
template <int OFF>
__device__ __forceinline__ u32x2 att_tr_read(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// a packed pair of ones in the operand format
template <typename T> struct AttOnes;
template <> struct AttOnes<T_F16> { static constexpr uint32_t v = 0x3C003C00u; };
template <> struct AttOnes<T_BF16> { static constexpr uint32_t v = 0x3F803F80u; };
