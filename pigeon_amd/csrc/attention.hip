// attention.hip -- multi-head self-attention of the ViT (16 heads x 64, 577 tokens, no mask), flash style.
//
// Replaces: transformers CLIPAttention.forward + eager_attention_forward (modeling_clip.py:259-335): per
// (image, head) softmax(Q K^T / 8) V with the softmax in fp32 -- SURVEY.md section 2c row K5.
//
// Input is the fused QKV activation (n_img*577, 3072) fp16/bf16 exactly as the QKV GEMM writes it (token-major;
// a head's Q/K/V rows are 128-byte contiguous segments), with Q pre-multiplied by log2(e)/8 so the kernel
// can use v_exp_f32 (2^x) directly.  Output (n_img*577, 1024) in the same 16-bit type, column = head*64 + d.
//
// This file holds four generations (selected by PIGEON_ATTN_VARIANT, see pg_attention_launch at the bottom):
//   v1 attention_kernel   register-staged K/V, the structure described below;
//   v4 attention4_kernel  v1 with the softmax instruction diet and 3 waves per SIMD (+ ablation switches);
//   v5 attention5_kernel  round 2's product: K and V by direct-to-LDS DMA, V row-major + ds_read_b64_tr_b16, single-key tail;
//   v8 attention8_kernel  DEFAULT (round 3): the same data movement with both GEMMs on v_mfma_f32_16x16x32 (section "v8" below).
// The product library instantiates v8 only; the older generations compile into the tools build (-DPIGEON_ABLATIONS).
//
// Structure common to all (gfx950, wave64):
//   * block = 4 waves = 128 query rows of one (image, head); 5 blocks cover the 577 queries.  The 5 blocks of
//     a pair are mapped to the SAME XCD (block b runs on XCD b%8) so K/V are fetched into one L2 once.
//   * K/V are walked in 64-key tiles, register-staged (loads for tile t+1 are issued before tile t is
//     multiplied, written to LDS after) into a double-buffered LDS image, one barrier per tile.
//   * S^T = K Q^T: mfma_32x32x16(A = K rows, B = Q rows) leaves each LANE owning one query and 16 keys per
//     32-key block, so the row max / row sum are in-lane reductions plus one lane^32 exchange.
//   * P feeds the PV MFMA straight from those registers as the B operand (O^T = V^T P^T).  The key order a
//     lane holds (keys 4h+{0..3}, 8+4h+{0..3} per 16-wide k-step) is simply used as the contraction order
//     on BOTH operands: V is stored transposed in LDS (VT[d][key], built with packed ds_write_b32 from the
//     register-staged rows) and the A operand gathers the same keys with two ds_read_b64.
//   * O^T accumulators keep lane == query, so the online-softmax rescale is a per-lane scalar multiply.
//   * key padding: 577 = 9*64 + 1; the last tile masks keys >= 577 to -1e30 before the max.
#include "common.h"
#include "pigeon_internal.h"

#include <cstdlib>
#include <type_traits>

#define ATT_KT 64
#define ATT_QB 128                       // query rows per block
#define ATT_NQB 5                        // ceil(577 / 128)
#define ATT_NT 10                        // ceil(577 / 64)
#define K_ROWB 128
#define VT_STRIDE 136                    // bytes per VT row: 64 keys * 2 B + 8 B pad (conflict-free b64 reads)
#define K_TILE_BYTES (ATT_KT * K_ROWB)   // 8192
#define VT_TILE_BYTES (64 * VT_STRIDE)   // 8704
#define QKV_LD 3072

struct StageRegs { u32x4 k[2]; u32x4 v[2]; };

__device__ __forceinline__ void att_load_tile(StageRegs& st, const uint16_t* __restrict__ qkv, int64_t base,
                                              int head, int t, int tid) {
    const int key0 = t * ATT_KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cid = tid + i * 256, row = cid >> 3, c = cid & 7;
        int key = key0 + row; key = key < VIT_TOKENS ? key : VIT_TOKENS - 1;
        st.k[i] = *(const u32x4*)(qkv + (base + key) * QKV_LD + 1024 + head * 64 + c * 8);
    }
    const int j = tid & 31, c = tid >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int key = key0 + 2 * j + i; key = key < VIT_TOKENS ? key : VIT_TOKENS - 1;
        st.v[i] = *(const u32x4*)(qkv + (base + key) * QKV_LD + 2048 + head * 64 + c * 8);
    }
}

__device__ __forceinline__ void att_store_tile(const StageRegs& st, char* ks, char* vt, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cid = tid + i * 256, row = cid >> 3, c = cid & 7;
        *(u32x4*)(ks + row * K_ROWB + ((c ^ ((row >> 1) & 7)) << 4)) = st.k[i];
    }
    const int j = tid & 31, c = tid >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t w0 = st.v[0][e >> 1], w1 = st.v[1][e >> 1];
        const uint32_t lo = (e & 1) ? (w0 >> 16) : (w0 & 0xffffu);
        const uint32_t hi = (e & 1) ? (w1 >> 16) : (w1 & 0xffffu);
        *(uint32_t*)(vt + (c * 8 + e) * VT_STRIDE + j * 4) = lo | (hi << 16);   // keys 2j (low), 2j+1 (high)
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 2 * VT_TILE_BYTES];
    char* ks0 = smem;
    char* vt0 = smem + 2 * K_TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, g = lane >> 5;

    // XCD-aware decode: the ATT_NQB query blocks of one (image, head) pair share an XCD
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qb = slot % ATT_NQB;
    const int pair = (slot / ATT_NQB) * 8 + xcd;
    const int img = pair >> 4, head = pair & 15;
    const int64_t base = (int64_t)img * VIT_TOKENS;

    const int q_first = qb * ATT_QB + wave * 32;            // wave-uniform
    const bool wave_active = q_first < VIT_TOKENS;
    const int qrow = q_first + lq;
    const int qr = qrow < VIT_TOKENS ? qrow : VIT_TOKENS - 1;

    typename T::v8 qf[4];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi)
        qf[ksi] = *(const typename T::v8*)(qkv + (base + qr) * QKV_LD + head * 64 + ksi * 16 + g * 8);

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -1e30f, l = 0.f;

    int kxoff[4];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi) kxoff[ksi] = ((ksi * 2 + g) ^ ((lq >> 1) & 7)) << 4;

    StageRegs st;
    att_load_tile(st, qkv, base, head, 0, tid);
    att_store_tile(st, ks0, vt0, tid);
    __syncthreads();

    for (int t = 0; t < ATT_NT; ++t) {
        const int cur = t & 1;
        if (t + 1 < ATT_NT) att_load_tile(st, qkv, base, head, t + 1, tid);   // in flight during the math
        const char* ks = ks0 + cur * K_TILE_BYTES;
        const char* vt = vt0 + cur * VT_TILE_BYTES;

        if (wave_active) {
            // ---- S^T = K Q^T : lane owns query lq, keys (r&3)+8*(r>>2)+4g of each 32-key block ----
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
                for (int ksi = 0; ksi < 4; ++ksi) {
                    const typename T::v8 kf = *(const typename T::v8*)(ks + (kb * 32 + lq) * K_ROWB + kxoff[ksi]);
                    s[kb] = T::mfma(kf, qf[ksi], s[kb]);
                }
            }
            if (t == ATT_NT - 1) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * ATT_KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (key >= VIT_TOKENS) s[kb][r] = -1e30f;
                    }
            }
            // ---- online softmax (base 2; Q carries log2(e)/8) ----
            float tmax = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
                    s[kb][r] = p;
                    psum += p;
                }
            l = l * alpha + psum;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

            // ---- O^T += V^T P^T : B operand = this lane's own P registers, 8 per 16-wide k-step ----
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 pw;
#pragma unroll
                    for (int w = 0; w < 4; ++w) pw[w] = pack16x2<T>(s[kb][8 * s2 + 2 * w], s[kb][8 * s2 + 2 * w + 1]);
                    const typename T::v8 pf = __builtin_bit_cast(typename T::v8, pw);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const char* vrow = vt + (db * 32 + lq) * VT_STRIDE + (kb * 32 + 16 * s2 + 4 * g) * 2;
                        const u32x2 lo = *(const u32x2*)(vrow);        // keys +0..3
                        const u32x2 hi = *(const u32x2*)(vrow + 16);   // keys +8..11
                        u32x4 vw; vw[0] = lo[0]; vw[1] = lo[1]; vw[2] = hi[0]; vw[3] = hi[1];
                        const typename T::v8 vf = __builtin_bit_cast(typename T::v8, vw);
                        o[db] = T::mfma(vf, pf, o[db]);
                    }
                }
            }
        }

        if (t + 1 < ATT_NT) att_store_tile(st, ks0 + (cur ^ 1) * K_TILE_BYTES, vt0 + (cur ^ 1) * VT_TILE_BYTES, tid);
        __syncthreads();
    }

    if (wave_active) {
        const float ltot = l + __shfl_xor(l, 32, 64);
        const float inv = 1.0f / ltot;
        if (qrow < VIT_TOKENS) {
            uint16_t* orow = out + (base + qrow) * VIT_HIDDEN + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    u32x2 pk;
                    pk[0] = pack16x2<T>(o[db][4 * q4] * inv, o[db][4 * q4 + 1] * inv);
                    pk[1] = pack16x2<T>(o[db][4 * q4 + 2] * inv, o[db][4 * q4 + 3] * inv);
                    *(u32x2*)(orow + db * 32 + 8 * q4 + 4 * g) = pk;
                }
        }
    }
}

// ================================================================================================================
// Building blocks of the current kernels (v4, v5): QK^T for one 32-query block, the softmax with its VALU diet, PV.
// Two earlier structures were measured and removed: 64 queries per wave with K/V fragments shared by two query blocks
// (half the LDS reads, but 224 VGPRs -> 2 waves per SIMD: same time as v4), and the same with the two query blocks
// staggered by sched_group_barrier so that one block's MFMAs sit beside the other's softmax (spills, -25 %).
// ================================================================================================================
template <typename T, bool LAST, bool COND_RESCALE = false>
__device__ __forceinline__ void att3_softmax(f32x16 (&s)[2], f32x16 (&o)[2], float& m, float& l, typename T::v8 (&pf)[2][2],
                                             int t, int g) {
    if (LAST) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * ATT_KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (key >= VIT_TOKENS) s[kb][r] = -1e30f;
            }
    }
    float tmax = max3f(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = max3f(tmax, s[0][r], s[0][r + 1]);
    tmax = max3f(tmax, s[0][15], s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, s[1][r], s[1][r + 1]);
    tmax = max3f(tmax, s[1][15], s[1][15]);
    const float m_new = max3f(tmax, __shfl_xor(tmax, 32, 64), m);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    // wave-uniform: did any row maximum of this wave move?  (alpha == 1.0 exactly for the rows that did not)
    const bool moved = !COND_RESCALE || __builtin_amdgcn_ballot_w64(m_new > m) != 0;
    m = m_new;
    const f32x2 m2 = {m_new, m_new};
    f32x2 ps2 = {0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4 pw;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x2 sv = {s[kb][8 * s2 + 2 * w], s[kb][8 * s2 + 2 * w + 1]};
                const f32x2 d = sv - m2;
                const f32x2 pv = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                ps2 += pv;
                pw[w] = T::pack2(pv[0], pv[1]);
            }
            pf[kb][s2] = __builtin_bit_cast(typename T::v8, pw);
        }
    l = l * alpha + (ps2[0] + ps2[1]);
    if (moved) {                                             // multiplying by exactly 1.0 otherwise: skipping is bit-identical
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
}

// S^T = K Q^T for one 32-query block.  All eight K fragments are fetched first (32 VGPRs) and the two 32-key accumulator
// chains are INTERLEAVED: as the compiler scheduled the naive loop it issued {ds_read, s_waitcnt, mfma} eight times with
// four dependent MFMAs in a row per chain -- every MFMA paid an LDS latency plus the 64-cycle dependent-accumulator
// latency instead of the 32-cycle issue rate.
template <typename T>
__device__ __forceinline__ void att3_qk(f32x16 (&s)[2], const typename T::v8 (&qf)[4], const char* ks, const int (&kxoff)[4], int lq) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typename T::v8 kf[4][2];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            kf[ksi][kb] = *(const typename T::v8*)(ks + (kb * 32 + lq) * K_ROWB + kxoff[ksi]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            s[kb] = T::mfma(kf[ksi][kb], qf[ksi], ksi == 0 ? zero16 : s[kb]);
}

template <typename T>
__device__ __forceinline__ void att3_pv(f32x16 (&o)[2], const typename T::v8 (&pf)[2][2], const char* vt, int lq, int g) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const char* vrow = vt + (db * 32 + lq) * VT_STRIDE + (kb * 32 + 16 * s2 + 4 * g) * 2;
                const u32x2 lo = *(const u32x2*)(vrow);        // keys +0..3
                const u32x2 hi = *(const u32x2*)(vrow + 16);   // keys +8..11
                u32x4 vw; vw[0] = lo[0]; vw[1] = lo[1]; vw[2] = hi[0]; vw[3] = hi[1];
                o[db] = T::mfma(__builtin_bit_cast(typename T::v8, vw), pf[kb][s2], o[db]);
            }
}

// ================================================================================================================
// v4: v1's geometry (32 queries per wave, 128-query blocks, 5 blocks per (image, head)) with the softmax instruction diet
// of v2/v3 (att3_softmax): the small per-wave state (O 32 + S 32 + Q 16 registers) is what lets 3-4 waves share a SIMD, and
// with that many independent waves the hardware overlaps one wave's MFMAs with another's softmax by itself.
// ================================================================================================================
// K tile straight into LDS (buffer_load_dwordx4 ... lds, as the GEMM stages its operands): no VGPR round trip, no
// ds_write, the bank swizzle applied to the source address.  A wave issues 2 of the tile's 8 DMAs (8 keys x 128 B each).
typedef __attribute__((address_space(3))) void att_lds_void;
__device__ __forceinline__ void att_dma_k(__amdgpu_buffer_rsrc_t rk, char* ks, int wave, int lane, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int grp = wave + 4 * i;                      // 8-key group inside the 64-key tile
        const int row = grp * 8 + (lane >> 3);
        int key = t * ATT_KT + row; key = key < VIT_TOKENS ? key : VIT_TOKENS - 1;
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (att_lds_void*)(ks + grp * 8 * K_ROWB), 16, key * (QKV_LD * 2) + c * 16, 0, 0, 0);
    }
}
__device__ __forceinline__ void att_load_v(StageRegs& st, const uint16_t* __restrict__ qkv, int64_t base, int head, int t, int tid) {
    const int j = tid & 31, c = tid >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int key = t * ATT_KT + 2 * j + i; key = key < VIT_TOKENS ? key : VIT_TOKENS - 1;
        st.v[i] = *(const u32x4*)(qkv + (base + key) * QKV_LD + 2048 + head * 64 + c * 8);
    }
}
__device__ __forceinline__ void att_store_v(const StageRegs& st, char* vt, int tid) {
    const int j = tid & 31, c = tid >> 5;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t w0 = st.v[0][e >> 1], w1 = st.v[1][e >> 1];
        const uint32_t lo = (e & 1) ? (w0 >> 16) : (w0 & 0xffffu);
        const uint32_t hi = (e & 1) ? (w1 >> 16) : (w1 & 0xffffu);
        *(uint32_t*)(vt + (c * 8 + e) * VT_STRIDE + j * 4) = lo | (hi << 16);
    }
}

template <typename T, int WAVES_PER_SIMD, int ABL = 0, bool KDMA = false>   // ABL (timing only, wrong results): 1 no K/V staging in the loop, 2 no softmax
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void attention4_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 2 * VT_TILE_BYTES];
    char* ks0 = smem;
    char* vt0 = smem + 2 * K_TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, g = lane >> 5;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qb = slot % ATT_NQB;
    const int pair = (slot / ATT_NQB) * 8 + xcd;
    const int img = pair >> 4, head = pair & 15;
    const int64_t base = (int64_t)img * VIT_TOKENS;
    const int q_first = qb * ATT_QB + wave * 32;            // wave-uniform
    const bool wave_active = q_first < VIT_TOKENS;
    const int qrow = q_first + lq;
    const int qr = qrow < VIT_TOKENS ? qrow : VIT_TOKENS - 1;

    typename T::v8 qf[4];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi)
        qf[ksi] = *(const typename T::v8*)(qkv + (base + qr) * QKV_LD + head * 64 + ksi * 16 + g * 8);
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -1e30f, l = 0.f;
    int kxoff[4];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi) kxoff[ksi] = ((ksi * 2 + g) ^ ((lq >> 1) & 7)) << 4;

    StageRegs st;
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(qkv + base * QKV_LD + 1024 + head * 64), (short)0, (int)(VIT_TOKENS * QKV_LD * 2), 0x00020000);
    if (KDMA) {
        att_dma_k(rk, ks0, wave, lane, 0);
        att_load_v(st, qkv, base, head, 0, tid);
        att_store_v(st, vt0, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        att_load_tile(st, qkv, base, head, 0, tid);
        att_store_tile(st, ks0, vt0, tid);
    }
    __syncthreads();
    for (int t = 0; t < ATT_NT; ++t) {
        const int cur = t & 1;
        if (t + 1 < ATT_NT && !(ABL & 1)) {
            if (KDMA) {
                att_dma_k(rk, ks0 + (cur ^ 1) * K_TILE_BYTES, wave, lane, t + 1);     // lands under this tile's math
                att_load_v(st, qkv, base, head, t + 1, tid);
            } else att_load_tile(st, qkv, base, head, t + 1, tid);
        }
        const char* ks = ks0 + cur * K_TILE_BYTES;
        const char* vt = vt0 + cur * VT_TILE_BYTES;
        if (wave_active) {
            f32x16 sA[2];
            typename T::v8 pfA[2][2];
            att3_qk<T>(sA, qf, ks, kxoff, lq);
            if (ABL & 2) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        u32x4 pw;
#pragma unroll
                        for (int w = 0; w < 4; ++w) pw[w] = __builtin_bit_cast(uint32_t, sA[kb][8 * s2 + 2 * w]);
                        pfA[kb][s2] = __builtin_bit_cast(typename T::v8, pw);
                    }
            } else if (t < ATT_NT - 1) att3_softmax<T, false>(sA, o, m, l, pfA, t, g);
            else att3_softmax<T, true>(sA, o, m, l, pfA, t, g);
            att3_pv<T>(o, pfA, vt, lq, g);
        }
        if (t + 1 < ATT_NT && !(ABL & 1)) {
            if (KDMA) {
                att_store_v(st, vt0 + (cur ^ 1) * VT_TILE_BYTES, tid);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // this wave's K DMAs have landed
            } else att_store_tile(st, ks0 + (cur ^ 1) * K_TILE_BYTES, vt0 + (cur ^ 1) * VT_TILE_BYTES, tid);
        }
        if (!(ABL & 4)) __syncthreads();
    }
    if (wave_active) {
        const float ltot = l + __shfl_xor(l, 32, 64);
        const float inv = 1.0f / ltot;
        if (qrow < VIT_TOKENS) {
            uint16_t* orow = out + (base + qrow) * VIT_HIDDEN + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    u32x2 pk;
                    pk[0] = pack16x2<T>(o[db][4 * q4] * inv, o[db][4 * q4 + 1] * inv);
                    pk[1] = pack16x2<T>(o[db][4 * q4 + 2] * inv, o[db][4 * q4 + 3] * inv);
                    *(u32x2*)(orow + db * 32 + 8 * q4 + 4 * g) = pk;
                }
        }
    }
}

// ================================================================================================================
// v5 (variant 11): both K and V tiles arrive by direct-to-LDS DMA, V stays ROW-major in LDS and the PV operand is read with
// the transposing LDS load.  Staging V through registers (2 global loads, 16 VALU unpacks, 8 ds_write_b32 per thread and
// tile, to build V^T) was the largest removable part of v4 (ablation: 0.37 of 1.35 ms).
//
// ds_read_b64_tr_b16 (measured with tools/tr_probe.hip): inside a 16-lane group, input lane j = 4k + r supplies 4 contiguous
// 16-bit values In[j][0..3]; output lane i receives In[4k + i/4][i%4] for k = 0..3.  Pointing lane j at
// V[key0 + j/4][d0 + 4 (j%4) ..+3] therefore hands lane i the four keys key0..key0+3 of column d0 + i -- the k-contiguous
// A fragment of O^T += V^T P^T -- from a row-major image.  The PV contraction order of a 16-key step is keys
// {4g..4g+3, 8+4g..8+4g+3} (what the lane's P registers hold), i.e. two such reads per MFMA, as many as v4 issued.
// Bank conflicts: one ds_read_b64 pass covers 32 lanes = 4 key rows x 64 B; rows are 128 B apart, so rows r and r+2 would
// share banks; 64-byte halves of a row are swapped when bit 1 of the key index is set (applied on the DMA source).
// ================================================================================================================
typedef __attribute__((ext_vector_type(4))) short att_s16x4;

// v5 staging: the lane's byte offset inside a 64-key tile is loop-invariant (dvo[i], i = the wave's two 8-key groups), the
// tile advance is an SGPR offset, and keys past token 576 need no clamp: the descriptor ends at the image's last row, the
// DMA writes zeros there (their scores are masked to -1e30 before the softmax anyway).
__device__ __forceinline__ void att5_dma(__amdgpu_buffer_rsrc_t r, char* dst, int wave, const int (&dvo)[2], int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (att_lds_void*)(dst + (wave + 4 * i) * 8 * K_ROWB), 16, dvo[i],
                                                 t * (ATT_KT * QKV_LD * 2), 0, 0);
}

// All 16 transposing reads of a tile are two per-lane base addresses plus compile-time offsets: the 64-byte swizzle bit of
// a key row ((key >> 1) & 1) depends only on the lane (bit 3 of its index in the 16-lane group), so it just selects which
// of the two 32-column blocks (db) sits in which 64-byte half.
// The 16 transposing V reads of a tile are issued as inline asm, right after the QK^T MFMAs and before the softmax, and
// waited for (att5_wait_v) just before the PV MFMAs.  Written with the ds_read_tr builtin, hipcc puts an `s_waitcnt
// vmcnt(0)` in front of the first read: it assumes the read may alias the direct-to-LDS DMA of the NEXT tile issued at the
// top of the loop (other stage, never the same bytes), which parks the wave until that DMA has landed.
template <int OFF>
__device__ __forceinline__ u32x2 att_tr_read(uint32_t addr) {
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <typename T>
__device__ __forceinline__ void att5_load_v(u32x4 (&vf)[2][2][2], const char* vs, const int (&vbase)[2]) {
    const uint32_t vs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)vs;
    const uint32_t a0 = vs_lds + vbase[0], a1 = vs_lds + vbase[1];
    __builtin_amdgcn_sched_barrier(0);                       // after the last QK^T MFMA: its lgkmcnt waits must not see these
#define ATT_TR2(kb, s2)                                                                                       \
    {                                                                                                         \
        const u32x2 l0 = att_tr_read<((kb) * 32 + 16 * (s2)) * K_ROWB>(a0);                                   \
        const u32x2 h0 = att_tr_read<((kb) * 32 + 16 * (s2) + 8) * K_ROWB>(a0);                               \
        const u32x2 l1 = att_tr_read<((kb) * 32 + 16 * (s2)) * K_ROWB>(a1);                                   \
        const u32x2 h1 = att_tr_read<((kb) * 32 + 16 * (s2) + 8) * K_ROWB>(a1);                               \
        vf[kb][s2][0] = u32x4{l0[0], l0[1], h0[0], h0[1]};                                                    \
        vf[kb][s2][1] = u32x4{l1[0], l1[1], h1[0], h1[1]};                                                    \
    }
    ATT_TR2(0, 0) ATT_TR2(0, 1) ATT_TR2(1, 0) ATT_TR2(1, 1)
#undef ATT_TR2
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void att5_wait_v(u32x4 (&vf)[2][2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vf[0][0][0]), "+v"(vf[0][0][1]), "+v"(vf[0][1][0]), "+v"(vf[0][1][1]),
                   "+v"(vf[1][0][0]), "+v"(vf[1][0][1]), "+v"(vf[1][1][0]), "+v"(vf[1][1][1])
                 :: "memory");
}
// O^T += V^T P^T with the fragments already in registers; consecutive MFMAs alternate the two accumulators (db)
template <typename T>
__device__ __forceinline__ void att5_pv(f32x16 (&o)[2], const typename T::v8 (&pf)[2][2], const u32x4 (&vf)[2][2][2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db)
                o[db] = T::mfma(__builtin_bit_cast(typename T::v8, vf[kb][s2][db]), pf[kb][s2], o[db]);
}

// ---- v6 softmax ("lazy" running maximum).  tools/pipe_rate.hip shows what the v5 softmax costs on gfx950:
//   * v_pk_{add,mul}_f32 do NOT co-issue with MFMA (37 cycles per instruction while another wave streams MFMAs, 5 alone),
//     so the 41 packed ops per tile serialised the softmax of one wave with the MFMA phase of its SIMD neighbours;
//   * v_exp_f32 is 8.75 cycles, v_cvt_pk_f16_f32 8, plain fp32 VALU 4.8 and these DO overlap MFMA.
// So: no packed-fp32 ops, and fewer VALU ops altogether:
//   * the QK^T accumulator starts from -m (a 16-register block kept equal to minus the lane's reference maximum), so the
//     MFMA delivers s - m and the 32 subtractions disappear;
//   * m is only raised when a tile's maximum exceeds it by more than ATT_LAZY_THR (2^8: P <= 256 is exact range for both
//     16-bit formats, l and O are fp32) -- after the first tiles that is rare, so the 32-multiply O rescale, the l rescale
//     and the alpha exp are skipped (wave-uniform branch);
//   * the row sum is accumulated with v_dot2c (P pair . (1,1) + l): 16 ops instead of 31 adds, and it sums the ROUNDED P,
//     the same values the PV MFMA uses as numerator.
//     (v_dot2c does not co-issue with MFMA either; 32 plain v_add_f32 instead measured 2.5 % SLOWER on the full chip --
//     the GPU runs this kernel at its 1400 W power cap, where instruction count matters more than pipe overlap; raising
//     the MFMA phases with s_setprio changed nothing.)
// exp2(s - m_ref) / sum is invariant under the choice of m_ref, so the result only differs from v5 in rounding.
#define ATT_LAZY_THR 8.0f
template <typename T> struct AttOnes;
template <> struct AttOnes<T_F16> { static constexpr uint32_t v = 0x3C003C00u; };
template <> struct AttOnes<T_BF16> { static constexpr uint32_t v = 0x3F803F80u; };

template <typename T>
__device__ __forceinline__ void att6_qk(f32x16 (&s)[2], const typename T::v8 (&qf)[4], const f32x16& negm, const char* ks,
                                        const int (&kxoff)[4], int lq) {
    typename T::v8 kf[4][2];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            kf[ksi][kb] = *(const typename T::v8*)(ks + (kb * 32 + lq) * K_ROWB + kxoff[ksi]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            s[kb] = T::mfma(kf[ksi][kb], qf[ksi], ksi == 0 ? negm : s[kb]);
}

template <typename T>
__device__ __forceinline__ void att6_softmax(f32x16 (&s)[2], f32x16 (&o)[2], f32x16& negm, float& l, typename T::v8 (&pf)[2][2],
                                             bool first) {
    float tmax = max3f(s[0][0], s[0][1], s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = max3f(tmax, s[0][r], s[0][r + 1]);
    tmax = max3f(tmax, s[0][15], s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, s[1][r], s[1][r + 1]);
    tmax = __builtin_fmaxf(tmax, s[1][15]);
    tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, 32, 64));  // the partner lane holds the other 32 keys of this query: the
                                                             // two lanes MUST agree on the reference (their P meet in one MFMA)
    // tmax is relative to the reference maximum already (s = q.k - m)
    if (first || __builtin_amdgcn_ballot_w64(tmax > ATT_LAZY_THR) != 0) {
        const float delta = (first || tmax > ATT_LAZY_THR) ? tmax : 0.f;
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);       // O = l = 0 before the first tile
        const float nm = negm[0] - delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = nm;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        l *= alpha;
    }
    float l0 = l, l1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4 pw;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                pw[w] = T::pack2(__builtin_amdgcn_exp2f(s[kb][8 * s2 + 2 * w]), __builtin_amdgcn_exp2f(s[kb][8 * s2 + 2 * w + 1]));
                if (w & 1) l1 = T::dot2(pw[w], AttOnes<T>::v, l1);
                else l0 = T::dot2(pw[w], AttOnes<T>::v, l0);
            }
            pf[kb][s2] = __builtin_bit_cast(typename T::v8, pw);
        }
    l = l0 + l1;
}

// KTAIL false: ten 64-key tiles, the last one masked (A/B arm).  ABL (timing-only ablations, results are garbage):
// 1 no K/V DMA inside the tile loop, 2 no per-tile vmcnt wait / barrier.
template <typename T, int WAVES_PER_SIMD, bool KTAIL = true, bool LAZY = true, int ABL = 0>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void attention5_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char smem[4 * K_TILE_BYTES];              // K0 K1 V0 V1, 8 KB each
    char* ks0 = smem;
    char* vs0 = smem + 2 * K_TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, g = lane >> 5;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qb = slot % ATT_NQB;
    const int pair = (slot / ATT_NQB) * 8 + xcd;
    const int img = pair >> 4, head = pair & 15;
    const int64_t base = (int64_t)img * VIT_TOKENS;
    const int q_first = qb * ATT_QB + wave * 32;            // wave-uniform
    const bool wave_active = q_first < VIT_TOKENS;
    const int qrow = q_first + lq;
    const int qr = qrow < VIT_TOKENS ? qrow : VIT_TOKENS - 1;

    typename T::v8 qf[4];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi)
        qf[ksi] = *(const typename T::v8*)(qkv + (base + qr) * QKV_LD + head * 64 + ksi * 16 + g * 8);
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m = -1e30f, l = 0.f;
    f32x16 negm;                                            // LAZY: minus the reference maximum, the QK^T accumulator seed
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    int kxoff[4];
#pragma unroll
    for (int ksi = 0; ksi < 4; ++ksi) kxoff[ksi] = ((ksi * 2 + g) ^ ((lq >> 1) & 7)) << 4;

    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(qkv + base * QKV_LD + 1024 + head * 64), (short)0, (int)(VIT_TOKENS * QKV_LD * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(qkv + base * QKV_LD + 2048 + head * 64), (short)0, (int)(VIT_TOKENS * QKV_LD * 2), 0x00020000);
    int dvo_k[2], dvo_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 4 * i) * 8 + (lane >> 3);
        dvo_k[i] = row * (QKV_LD * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        dvo_v[i] = row * (QKV_LD * 2) + (((lane & 7) ^ (((row >> 1) & 1) << 2)) << 4);
    }
    int vbase[2];
    {
        const int j = lane & 15, dh = (lane >> 4) & 1, sw = (j >> 3) & 1;
        const int lane_off = (4 * g + (j >> 2)) * K_ROWB + (dh * 16 + 4 * (j & 3)) * 2;
        vbase[0] = lane_off + (sw ? 64 : 0);
        vbase[1] = lane_off + (sw ? 0 : 64);
    }
    // 577 = 9 x 64 + 1: nine full key tiles go through the MFMA loop, the last key (token 576) is a VALU tail (a tenth
    // tile would spend a whole tile's MFMA, softmax and DMA work on one valid key: 9 % of the kernel).
    static_assert(VIT_TOKENS == 9 * ATT_KT + 1, "key tail assumes 577 tokens");
    static_assert(KTAIL || !LAZY, "the lazy softmax has no masked-tile path");
    constexpr int NFULL = KTAIL ? 9 : ATT_NT;
    att5_dma(rk, ks0, wave, dvo_k, 0);
    att5_dma(rv, vs0, wave, dvo_v, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < NFULL; ++t) {
        const int cur = t & 1;
        if (ABL != 1 && t + 1 < NFULL) {                      // both tiles of step t+1 land under this tile's math
            att5_dma(rk, ks0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_k, t + 1);
            att5_dma(rv, vs0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_v, t + 1);
        }
        const char* ks = ks0 + cur * K_TILE_BYTES;
        const char* vs = vs0 + cur * K_TILE_BYTES;
        if (wave_active) {
            f32x16 sA[2];
            typename T::v8 pfA[2][2];
            u32x4 vf[2][2][2];
            if (LAZY) {
                att6_qk<T>(sA, qf, negm, ks, kxoff, lq);
                att5_load_v<T>(vf, vs, vbase);                // 16 transposing reads in flight under the softmax
                att6_softmax<T>(sA, o, negm, l, pfA, t == 0);
                att5_wait_v(vf);
            } else {
                att3_qk<T>(sA, qf, ks, kxoff, lq);
                att5_load_v<T>(vf, vs, vbase);
                if (KTAIL || t < ATT_NT - 1) att3_softmax<T, false>(sA, o, m, l, pfA, t, g);
                else att3_softmax<T, true>(sA, o, m, l, pfA, t, g);
                att5_wait_v(vf);
            }
            att5_pv<T>(o, pfA, vf);
        }
        if (ABL != 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMAs of step t+1 have landed
            __syncthreads();
        }
    }
    if (KTAIL && wave_active) {
        // ---- key 576: s = q . k (this lane holds 32 of the 64 dims, its lane^32 partner the rest), one online-softmax step,
        // O += p * v over the lane's 32 columns.  p stays fp32 here (the MFMA path rounds P to 16 bits first).
        const uint16_t* krow = qkv + (base + (VIT_TOKENS - 1)) * QKV_LD + 1024 + head * 64;
        const uint16_t* vrow = krow + 1024;
        float sp = 0.f;
#pragma unroll
        for (int ksi = 0; ksi < 4; ++ksi) {
            const u32x4 kq = *(const u32x4*)(krow + ksi * 16 + g * 8);
            const u32x4 qq = __builtin_bit_cast(u32x4, qf[ksi]);
#pragma unroll
            for (int w = 0; w < 4; ++w) sp = T::dot2(qq[w], kq[w], sp);
        }
        const float sc = sp + __shfl_xor(sp, 32, 64);
        if (LAZY) m = -negm[0];
        const float m_new = fmaxf(m, sc);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        const float pk = __builtin_amdgcn_exp2f(sc - m_new);
        m = m_new;
        l = l * alpha + (g == 0 ? pk : 0.f);                 // the two lanes of a row are summed below: count the key once
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const u32x2 vv = *(const u32x2*)(vrow + db * 32 + 8 * q4 + 4 * g);   // columns (r&3) + 8 (r>>2) + 4g, r = 4 q4 ..
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t hb = (uint16_t)(vv[e >> 1] >> (16 * (e & 1)));
                    o[db][4 * q4 + e] = fmaf(pk, T::val(hb), o[db][4 * q4 + e] * alpha);
                }
            }
    }
    if (wave_active) {
        const float ltot = l + __shfl_xor(l, 32, 64);
        const float inv = 1.0f / ltot;
        if (qrow < VIT_TOKENS) {
            uint16_t* orow = out + (base + qrow) * VIT_HIDDEN + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    u32x2 pk;
                    pk[0] = pack16x2<T>(o[db][4 * q4] * inv, o[db][4 * q4 + 1] * inv);
                    pk[1] = pack16x2<T>(o[db][4 * q4 + 2] * inv, o[db][4 * q4 + 3] * inv);
                    *(u32x2*)(orow + db * 32 + 8 * q4 + 4 * g) = pk;
                }
        }
    }
}

// ================================================================================================================
// v8 (variant 21 = product default; 19 = 64-queries-per-wave arm): both GEMMs of the attention on v_mfma_f32_16x16x32.
//
// Why: at the 1400 W package cap the 16x16x32 form moves half as many accumulator registers per flop and sustains 2.06-2.09
// PFLOP/s where 32x32x16 sustains 1.72-1.74 (tools/mfma_issue.hip, profiles/r02/mfma_issue.txt) -- the persistent GEMMs gained
// 4.5 % from the same change in round 2.  The attention kernel spends ~0.41 ms of its 1.07 ms worth of energy in MFMAs.
//
// Geometry (r16 = lane & 15, g4 = lane >> 4):
//   S^T = K Q^T   A = K block (16 keys x 32 d): lane holds K[key 16 kb + r16][d 32 ks + 8 g4 ..+7]     (ds_read_b128, GEMM swizzle)
//                 B = Q block (32 d x 16 queries): lane holds Q[query 16 qb + r16][d 32 ks + 8 g4 ..+7] (registers, loaded once)
//                 D = S^T block: lane holds keys 16 kb + 4 g4 + e (e = 0..3) of query 16 qb + r16
//   O^T += V^T P^T  B = P block (32 keys x 16 queries): the lane's own D registers of key blocks 2j and 2j+1, packed: k index
//                 8 g4 + e  <->  key 32 j + 4 g4 + e (e < 4), 32 j + 16 + 4 g4 + (e - 4) (e >= 4) -- any order will do as long
//                 as A uses the same;  A = V^T block (16 d x 32 keys): two ds_read_b64_tr_b16 per fragment (keys 32 j + 4 g4 ..
//                 and 32 j + 16 + 4 g4 .., column 16 db + r16);  D = O^T block: lane holds d 16 db + 4 g4 + e of query r16.
//   A lane therefore holds NQB queries (one per 16-query block), 16 keys of each per tile; the 64 keys of a query are spread
//   over the 4 lanes r16 + 16 g4.  The row sum comes out of the matrix pipe: a fifth 16-row "d" block whose V^T rows are (1, 0, ..., 0)
//   -- two more MFMAs per tile and query block instead of 16 v_dot2c per tile, which share the matrix pipe and cost more than the
//   MFMAs do (-2 %, profiles/r03/attention_v7_v8_ab.txt); with LSUM = false it is a per-lane partial (combined once at the end); the reference maximum (first
//   tile / careful path) needs two lane exchanges (xor 16, xor 32).
// LDS: K tile as before (row = key, 16-byte chunk c at c ^ ((key >> 1) & 7)).  V tile row-major with the 32-byte column block
//   db stored at db ^ ((key >> 1) & 3): one transposing read pass (lanes 0-31) then covers 8 key rows x 32 B = all 64 banks once.
// NQB x WAVES = 2 x 4 (32 queries per wave, 3 waves per SIMD, as v5) or 4 x 2 (64 queries per wave, 2 waves per SIMD: each K / V
//   fragment read from LDS feeds twice the MFMAs).  Block = 128 queries either way, 5 blocks per (image, head) on one XCD.
// ================================================================================================================
// LSUM: true = the row sums come out of the matrix pipe (an extra "ones" row block in the PV product, attention8_kernel) and this
// function only exponentiates and packs; false = v_dot2c on the packed P (16 per tile and wave; v_dot2c shares the matrix pipe).
template <typename T, int NQB, bool LSUM = false>
__device__ __forceinline__ void att8_exp_pack(const f32x4 (&S)[4][NQB], float (&l)[NQB], typename T::v8 (&pf)[NQB][2]) {
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        float l0 = l[qb], l1 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 pw;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const f32x4& sv = S[2 * j + h][qb];
                    pw[2 * h + w] = T::pack2(__builtin_amdgcn_exp2f(sv[2 * w]), __builtin_amdgcn_exp2f(sv[2 * w + 1]));
                    if constexpr (!LSUM) {
                        if (w) l1 = T::dot2(pw[2 * h + w], AttOnes<T>::v, l1);
                        else l0 = T::dot2(pw[2 * h + w], AttOnes<T>::v, l0);
                    }
                }
            pf[qb][j] = __builtin_bit_cast(typename T::v8, pw);
        }
        if constexpr (!LSUM) l[qb] = l0 + l1;
    }
}

// careful softmax step: per-query maximum of the tile (relative to the reference, the scores arrive as s - m), lazy rescale
template <typename T, int NQB, bool LSUM = false>
__device__ __forceinline__ void att8_softmax(f32x4 (&S)[4][NQB], f32x4 (&O)[4][NQB], f32x4 (&negm)[NQB], float (&l)[NQB],
                                             typename T::v8 (&pf)[NQB][2], bool first, f32x4 (&L)[NQB]) {
    float tmax[NQB];
    bool over = false;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        float t = max3f(S[0][qb][0], S[0][qb][1], S[0][qb][2]);
        t = max3f(t, S[0][qb][3], S[1][qb][0]);
        t = max3f(t, S[1][qb][1], S[1][qb][2]);
        t = max3f(t, S[1][qb][3], S[2][qb][0]);
        t = max3f(t, S[2][qb][1], S[2][qb][2]);
        t = max3f(t, S[2][qb][3], S[3][qb][0]);
        t = max3f(t, S[3][qb][1], S[3][qb][2]);
        t = __builtin_fmaxf(t, S[3][qb][3]);
        t = __builtin_fmaxf(t, __shfl_xor(t, 16, 64));       // the 4 lanes of a query MUST agree on the reference: their P
        t = __builtin_fmaxf(t, __shfl_xor(t, 32, 64));       // registers meet in one MFMA operand
        tmax[qb] = t;
        over = over || t > ATT_LAZY_THR;
    }
    if (first || __builtin_amdgcn_ballot_w64(over) != 0) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const float delta = (first || tmax[qb] > ATT_LAZY_THR) ? tmax[qb] : 0.f;
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);   // O = l = 0 before the first tile
            const float nm = negm[qb][0] - delta;
            negm[qb] = f32x4{nm, nm, nm, nm};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) S[kb][qb] -= delta;
#pragma unroll
            for (int db = 0; db < 4; ++db) O[db][qb] *= alpha;
            if constexpr (LSUM) L[qb] *= alpha;
            else l[qb] *= alpha;
        }
    }
    att8_exp_pack<T, NQB, LSUM>(S, l, pf);
}

// Everything that is not valid HOST code lives in __device__ functions, not in the kernel body or its lambdas: those are also
// instantiated for the host, and one address-space cast / "v" asm constraint / buffer-resource parameter there makes hipcc drop the
// host side of the kernel template WITHOUT a diagnostic (the library then fails to load with an undefined kernel symbol).
__device__ __forceinline__ void att8_dma(__amdgpu_buffer_rsrc_t r, char* dst, int wave, const int* dvo, int ndma, int waves, int t) {
#pragma unroll
    for (int i = 0; i < ndma; ++i)                           // ndma / waves are compile-time constants at every (inlined) call site
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (att_lds_void*)(dst + (wave + waves * i) * 8 * K_ROWB), 16, dvo[i],
                                                 t * (ATT_KT * QKV_LD * 2), 0, 0);
}
// the 16 transposing V reads of a tile, right behind the last QK^T MFMA (inline asm: see att5_load_v)
__device__ __forceinline__ void att8_load_v(u32x4 (&vf)[4][2], const char* vs, const int (&vb)[4]) {
    const uint32_t vs_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)vs;
    __builtin_amdgcn_sched_barrier(0);
#define ATT8_TR(db, j)                                                                                        \
    {                                                                                                         \
        const u32x2 lo = att_tr_read<((j) * 32) * K_ROWB>(vs_lds + vb[db]);                                   \
        const u32x2 hi = att_tr_read<((j) * 32 + 16) * K_ROWB>(vs_lds + vb[db]);                              \
        vf[db][j] = u32x4{lo[0], lo[1], hi[0], hi[1]};                                                        \
    }
    ATT8_TR(0, 0) ATT8_TR(1, 0) ATT8_TR(2, 0) ATT8_TR(3, 0) ATT8_TR(0, 1) ATT8_TR(1, 1) ATT8_TR(2, 1) ATT8_TR(3, 1)
#undef ATT8_TR
    __builtin_amdgcn_sched_barrier(0);
}
// (a __device__ function, not inline in the kernel's lambda: a lambda body is also instantiated for the host, where the "v"
// constraint does not exist -- hipcc then silently drops the HOST side of the whole kernel template and the library fails to load)
__device__ __forceinline__ void att8_wait_v(u32x4 (&vf)[4][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vf[0][0]), "+v"(vf[1][0]), "+v"(vf[2][0]), "+v"(vf[3][0]), "+v"(vf[0][1]), "+v"(vf[1][1]),
                   "+v"(vf[2][1]), "+v"(vf[3][1])
                 :: "memory");
}

template <typename T, int NQB, int WAVES, int WPS, bool LSUM = false>
__global__ __launch_bounds__(256, WPS) void attention8_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    static_assert(NQB * WAVES * 16 == ATT_QB, "a block covers 128 queries");
    static_assert(VIT_TOKENS == 9 * ATT_KT + 1, "key tail assumes 577 tokens");
    __shared__ __attribute__((aligned(16))) char smem[4 * K_TILE_BYTES];              // K0 K1 V0 V1, 8 KB each
    char* ks0 = smem;
    char* vs0 = smem + 2 * K_TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qblk = slot % ATT_NQB;
    const int pair = (slot / ATT_NQB) * 8 + xcd;
    const int img = pair >> 4, head = pair & 15;
    const int64_t base = (int64_t)img * VIT_TOKENS;
    const int q_first = qblk * ATT_QB + wave * (NQB * 16);   // wave-uniform
    const bool wave_active = q_first < VIT_TOKENS;

    typename T::v8 qf[NQB][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        int qr = q_first + 16 * qb + r16;
        qr = qr < VIT_TOKENS ? qr : VIT_TOKENS - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qb][ks] = *(const typename T::v8*)(qkv + (base + qr) * QKV_LD + head * 64 + ks * 32 + g4 * 8);
    }
    f32x4 O[4][NQB], negm[NQB];
    f32x4 L[NQB];                                            // LSUM: row sums as a fifth "d" block whose V^T rows are (1, 0, 0, ...): l = L[qb][0] in lanes g4 == 0
    float l[NQB];
    // A fragment of that block: row m = 0 is all ones (lanes r16 == 0 hold eight 1.0), rows 1..15 zero
    const u32x4 ones_rows = r16 == 0 ? u32x4{AttOnes<T>::v, AttOnes<T>::v, AttOnes<T>::v, AttOnes<T>::v} : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        l[qb] = 0.f;
        L[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < 4; ++db) O[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // K fragment addressing: row r16 of a 16-key block, 16-byte chunk 4 ks + g4, swizzled with the row (as the GEMM operands)
    int kx[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kx[ks] = r16 * K_ROWB + (((4 * ks + g4) ^ ((r16 >> 1) & 7)) << 4);
    // V transposing reads: input lane jj = r16 of group g4 points at V[key 4 g4 + (jj >> 2) (+ 32 j + 16 h)][16 db + 4 (jj & 3) ..+3];
    // the 32-byte block db of that row sits at db ^ ((key >> 1) & 3), and (key >> 1) & 3 = (2 g4 + (jj >> 3)) & 3 is a lane constant
    int vb[4];
    {
        const int f = (2 * g4 + (r16 >> 3)) & 3;
#pragma unroll
        for (int db = 0; db < 4; ++db) vb[db] = (4 * g4 + (r16 >> 2)) * K_ROWB + ((db ^ f) << 5) + ((r16 & 3) << 3);
    }
    __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(qkv + base * QKV_LD + 1024 + head * 64), (short)0, (int)(VIT_TOKENS * QKV_LD * 2), 0x00020000);
    __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(qkv + base * QKV_LD + 2048 + head * 64), (short)0, (int)(VIT_TOKENS * QKV_LD * 2), 0x00020000);
    constexpr int NDMA = 8 / WAVES;                          // 8-key groups of a 64-key tile this wave stages, per operand
    int dvo_k[NDMA], dvo_v[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int row = (wave + WAVES * i) * 8 + (lane >> 3);
        dvo_k[i] = row * (QKV_LD * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        dvo_v[i] = row * (QKV_LD * 2) + (((lane & 7) ^ (((row >> 1) & 3) << 1)) << 4);
    }
    constexpr int NFULL = 9;

    att8_dma(rk, ks0, wave, dvo_k, NDMA, WAVES, 0);
    att8_dma(rv, vs0, wave, dvo_v, NDMA, WAVES, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < NFULL; ++t) {
        const int cur = t & 1;
        if (t + 1 < NFULL) {                                  // both tiles of step t+1 land under this tile's math
            att8_dma(rk, ks0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_k, NDMA, WAVES, t + 1);
            att8_dma(rv, vs0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_v, NDMA, WAVES, t + 1);
        }
        const char* ks = ks0 + cur * K_TILE_BYTES;
        const char* vs = vs0 + cur * K_TILE_BYTES;
        if (wave_active) {
            f32x4 S[4][NQB];
            typename T::v8 pf[NQB][2];
            {
                typename T::v8 kf[4][2];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int ks_ = 0; ks_ < 2; ++ks_) kf[kb][ks_] = *(const typename T::v8*)(ks + kb * 16 * K_ROWB + kx[ks_]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks_ = 0; ks_ < 2; ++ks_)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb)
                            S[kb][qb] = T::mfma16(kf[kb][ks_], qf[qb][ks_], ks_ == 0 ? negm[qb] : S[kb][qb]);
            }
            // the 16 transposing V reads, right behind the last QK^T MFMA (inline asm: see att5_load_v)
            u32x4 vf[4][2];
            att8_load_v(vf, vs, vb);
            att8_softmax<T, NQB, LSUM>(S, O, negm, l, pf, t == 0, L);
            att8_wait_v(vf);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb)
                        O[db][qb] = T::mfma16(__builtin_bit_cast(typename T::v8, vf[db][j]), pf[qb][j], O[db][qb]);
            if constexpr (LSUM) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb)
                        L[qb] = T::mfma16(__builtin_bit_cast(typename T::v8, ones_rows), pf[qb][j], L[qb]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMAs of step t+1 have landed
        __syncthreads();
    }
    if (!wave_active) return;

    // ---- key 576 (VALU): s = q . k over the lane's 16 of the 64 dims, summed over the query's 4 lanes; one online-softmax step;
    // O += p * v on the lane's 16 columns (d = 16 db + 4 g4 + e).  p stays fp32 here.
    const uint16_t* krow = qkv + (base + (VIT_TOKENS - 1)) * QKV_LD + 1024 + head * 64;
    const uint16_t* vrow = krow + 1024;
    u32x4 kq[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kq[ks] = *(const u32x4*)(krow + ks * 32 + g4 * 8);
    u32x2 vv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) vv[db] = *(const u32x2*)(vrow + db * 16 + g4 * 4);
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        float sp = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4 qq = __builtin_bit_cast(u32x4, qf[qb][ks]);
#pragma unroll
            for (int w = 0; w < 4; ++w) sp = T::dot2(qq[w], kq[ks][w], sp);
        }
        sp += __shfl_xor(sp, 16, 64);
        const float sc = sp + __shfl_xor(sp, 32, 64);
        if constexpr (LSUM) l[qb] = L[qb][0];                // lanes g4 == 0: the row sum; the others: 0
        const float m = -negm[qb][0];
        const float m_new = fmaxf(m, sc);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        const float pk = __builtin_amdgcn_exp2f(sc - m_new);
        l[qb] = l[qb] * alpha + (g4 == 0 ? pk : 0.f);        // the 4 lanes of a query are summed below: count the key once
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint16_t hb = (uint16_t)(vv[db][e >> 1] >> (16 * (e & 1)));
                O[db][qb][e] = fmaf(pk, T::val(hb), O[db][qb][e] * alpha);
            }
        float lt = l[qb] + __shfl_xor(l[qb], 16, 64);
        lt += __shfl_xor(lt, 32, 64);
        const float inv = 1.0f / lt;
        const int qrow = q_first + 16 * qb + r16;
        if (qrow < VIT_TOKENS) {
            uint16_t* orow = out + (base + qrow) * VIT_HIDDEN + head * 64;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                u32x2 pk2;
                pk2[0] = pack16x2<T>(O[db][qb][0] * inv, O[db][qb][1] * inv);
                pk2[1] = pack16x2<T>(O[db][qb][2] * inv, O[db][qb][3] * inv);
                *(u32x2*)(orow + db * 16 + g4 * 4) = pk2;
            }
        }
    }
}

static int attention_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PIGEON_ATTN_VARIANT");
        v = e ? atoi(e) : 21;
        if (v < 1 || v > 21) v = 21;
    }
    return v;
}

// Variants (env PIGEON_ATTN_VARIANT): 21 (default, the only one in the product library) v8 = v6's structure on v_mfma_f32_16x16x32
// (attention8_kernel, 32 queries per wave, row sums out of the matrix pipe); 20 = the same with v_dot2c row sums, 19 = 64 queries
// per wave (A/B arms); 11 v6 (round 2's product) = K and V by DMA,
// transposing LDS reads, single-key tail, lazy softmax on 32x32x16 MFMAs;
// 13 the same with the v5 softmax (running maximum updated every tile, packed fp32 ops); 12 = 13 with a masked tenth key
// tile instead of the tail (A/B arms); 14 / 15 timing-only ablations of 11 (no DMA in the loop / no per-tile barrier); 4 / 10 v4 register-staged / K by DMA;
// 5 v4 forced to 4 waves per SIMD (spills); 6..9 timing-only ablations of v4; 1 the first kernel.
template <typename KF, typename KB>
static int att_launch2(int dtype, KF kf, KB kb, dim3 grid, const void* qkv, void* out, hipStream_t s) {
    if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL(kf, grid, dim3(256), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    else hipLaunchKernelGGL(kb, grid, dim3(256), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    return pg_check_launch("attention");
}

template <typename KF, typename KB>
static int att_launch3(int dtype, KF kf, KB kb, dim3 grid, int threads, const void* qkv, void* out, hipStream_t s) {
    if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL(kf, grid, dim3(threads), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    else hipLaunchKernelGGL(kb, grid, dim3(threads), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    return pg_check_launch("attention");
}

int pg_attention_launch(int dtype, const void* qkv, void* out, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    if (dtype != PG_DTYPE_F16 && dtype != PG_DTYPE_BF16) { pg_set_error("attention: dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16"); return PG_EINVAL; }
    const int pairs = n_images * VIT_HEADS;                  // always a multiple of 8
    const dim3 grid(pairs * ATT_NQB);
    const int variant = attention_variant();
#ifdef PIGEON_ABLATIONS                                   // tools build only: older generations, A/B arms, timing-only ablations (6..9, 14, 15: WRONG RESULTS)
    switch (variant) {
        case 1: return att_launch2(dtype, attention_kernel<T_F16>, attention_kernel<T_BF16>, grid, qkv, out, s);
        case 4: return att_launch2(dtype, attention4_kernel<T_F16, 3>, attention4_kernel<T_BF16, 3>, grid, qkv, out, s);
        case 5: return att_launch2(dtype, attention4_kernel<T_F16, 4>, attention4_kernel<T_BF16, 4>, grid, qkv, out, s);
        case 6: return att_launch2(dtype, attention4_kernel<T_F16, 3, 1>, attention4_kernel<T_BF16, 3, 1>, grid, qkv, out, s);
        case 7: return att_launch2(dtype, attention4_kernel<T_F16, 3, 2>, attention4_kernel<T_BF16, 3, 2>, grid, qkv, out, s);
        case 8: return att_launch2(dtype, attention4_kernel<T_F16, 3, 5>, attention4_kernel<T_BF16, 3, 5>, grid, qkv, out, s);
        case 9: return att_launch2(dtype, attention4_kernel<T_F16, 3, 7>, attention4_kernel<T_BF16, 3, 7>, grid, qkv, out, s);
        case 10: return att_launch2(dtype, attention4_kernel<T_F16, 3, 0, true>, attention4_kernel<T_BF16, 3, 0, true>, grid, qkv, out, s);
        case 12: return att_launch2(dtype, attention5_kernel<T_F16, 3, false, false>, attention5_kernel<T_BF16, 3, false, false>, grid, qkv, out, s);
        case 14: return att_launch2(dtype, attention5_kernel<T_F16, 3, true, true, 1>, attention5_kernel<T_BF16, 3, true, true, 1>, grid, qkv, out, s);
        case 15: return att_launch2(dtype, attention5_kernel<T_F16, 3, true, true, 2>, attention5_kernel<T_BF16, 3, true, true, 2>, grid, qkv, out, s);
        case 13: return att_launch2(dtype, attention5_kernel<T_F16, 3, true, false>, attention5_kernel<T_BF16, 3, true, false>, grid, qkv, out, s);
        case 19: return att_launch3(dtype, attention8_kernel<T_F16, 4, 2, 2>, attention8_kernel<T_BF16, 4, 2, 2>, grid, 128, qkv, out, s);
        case 20: return att_launch3(dtype, attention8_kernel<T_F16, 2, 4, 3, false>, attention8_kernel<T_BF16, 2, 4, 3, false>, grid, 256, qkv, out, s);
        case 11: return att_launch2(dtype, attention5_kernel<T_F16, 3>, attention5_kernel<T_BF16, 3>, grid, qkv, out, s);
        default: break;
    }
#endif
    if (variant != 21) {
        pg_set_error("attention: PIGEON_ATTN_VARIANT=%d is not part of this build (product: 21; others need -DPIGEON_ABLATIONS)", variant);
        return PG_EINVAL;
    }
    return att_launch3(dtype, attention8_kernel<T_F16, 2, 4, 3, true>, attention8_kernel<T_BF16, 2, 4, 3, true>, grid, 256, qkv, out, s);
}
