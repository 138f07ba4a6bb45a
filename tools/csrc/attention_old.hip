// attention_old.hip -- TOOLS BUILD ONLY (python -m pigeon_amd.build --dev): the attention kernel generations the product kernel
// (pigeon_amd/csrc/attention.hip, v8) superseded, kept as A/B arms and for the timing-only ablations quoted in DESIGN.md
// (PIGEON_ATTN_VARIANT 1, 4..15).  Variants 6..9, 14, 15 compute WRONG results by construction.
//   v1 attention_kernel   register-staged K/V;
//   v4 attention4_kernel  v1 with the softmax instruction diet and 3 waves per SIMD (+ ablation switches);
//   v5 attention5_kernel  round 2's product: K and V by direct-to-LDS DMA, V row-major + ds_read_b64_tr_b16, single-key tail, lazy softmax (v6).
#include "attention_common.h"

#include <cstdlib>
#include <type_traits>
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

template <typename KF, typename KB>
static int att_old_launch2(int dtype, KF kf, KB kb, dim3 grid, const void* qkv, void* out, hipStream_t s) {
    if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL(kf, grid, dim3(256), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    else hipLaunchKernelGGL(kb, grid, dim3(256), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    return pg_check_launch("attention (old generation)");
}

// Variants: 11 v6 (round 2's product) = K and V by DMA, transposing LDS reads, single-key tail, lazy softmax on 32x32x16 MFMAs;
// 13 the same with the v5 softmax (running maximum updated every tile, packed fp32 ops); 12 = 13 with a masked tenth key tile
// instead of the tail; 14 / 15 timing-only ablations of 11 (no DMA in the loop / no per-tile barrier); 4 / 10 v4 register-staged /
// K by DMA; 5 v4 forced to 4 waves per SIMD (spills); 6..9 timing-only ablations of v4; 1 the first kernel.
// Returns 1 if `variant` is not one of this file's.
int pg_attention_old_launch(int variant, int dtype, const void* qkv, void* out, dim3 grid, hipStream_t s, int* rc) {
#define ATT_OLD(K1, K2) *rc = att_old_launch2(dtype, K1, K2, grid, qkv, out, s); return 0;
    switch (variant) {
        case 1: ATT_OLD(attention_kernel<T_F16>, attention_kernel<T_BF16>)
        case 4: ATT_OLD((attention4_kernel<T_F16, 3>), (attention4_kernel<T_BF16, 3>))
        case 5: ATT_OLD((attention4_kernel<T_F16, 4>), (attention4_kernel<T_BF16, 4>))
        case 6: ATT_OLD((attention4_kernel<T_F16, 3, 1>), (attention4_kernel<T_BF16, 3, 1>))
        case 7: ATT_OLD((attention4_kernel<T_F16, 3, 2>), (attention4_kernel<T_BF16, 3, 2>))
        case 8: ATT_OLD((attention4_kernel<T_F16, 3, 5>), (attention4_kernel<T_BF16, 3, 5>))
        case 9: ATT_OLD((attention4_kernel<T_F16, 3, 7>), (attention4_kernel<T_BF16, 3, 7>))
        case 10: ATT_OLD((attention4_kernel<T_F16, 3, 0, true>), (attention4_kernel<T_BF16, 3, 0, true>))
        case 11: ATT_OLD((attention5_kernel<T_F16, 3>), (attention5_kernel<T_BF16, 3>))
        case 12: ATT_OLD((attention5_kernel<T_F16, 3, false, false>), (attention5_kernel<T_BF16, 3, false, false>))
        case 13: ATT_OLD((attention5_kernel<T_F16, 3, true, false>), (attention5_kernel<T_BF16, 3, true, false>))
        case 14: ATT_OLD((attention5_kernel<T_F16, 3, true, true, 1>), (attention5_kernel<T_BF16, 3, true, true, 1>))
        case 15: ATT_OLD((attention5_kernel<T_F16, 3, true, true, 2>), (attention5_kernel<T_BF16, 3, true, true, 2>))
        default: return 1;
    }
#undef ATT_OLD
}
