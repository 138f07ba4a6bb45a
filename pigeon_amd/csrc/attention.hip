// attention.hip -- multi-head self-attention of the ViT (16 heads x 64, 577 tokens, no mask), flash style.
//
// Replaces: transformers CLIPAttention.forward + eager_attention_forward (modeling_clip.py:259-335): per
// (image, head) softmax(Q K^T / 8) V with the softmax in fp32 -- SURVEY.md section 2c row K5.
//
// Input is the fused QKV activation (n_img*577, 3072) fp16/bf16 exactly as the QKV GEMM writes it (token-major;
// a head's Q/K/V rows are 128-byte contiguous segments), with Q pre-multiplied by log2(e)/8 so the kernel
// can use v_exp_f32 (2^x) directly.  Output (n_img*577, 1024) in the same 16-bit type, column = head*64 + d.
//
// Structure (gfx950, wave64):
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
// v2: 64 queries per wave.  The v1 kernel above is LDS-bound: per 64-key tile every wave re-reads the K fragments
// (8 ds_read_b128) and V^T fragments (16 ds_read_b64) for only 32 queries, and a 128-query block re-stages the whole
// K/V tile -- LDS read + write cycles per CU exceed the VALU/MFMA cycles of the same work (rocprof: SQ_LDS busy > 100 %
// of the softmax-bound time).  Here a wave owns TWO 32-query blocks: each K / V^T fragment read feeds two MFMAs, the
// block covers 256 queries per staged tile (3 blocks per (image, head): 256 + 256 + 65), so LDS traffic per query
// halves, and the two independent softmax chains of a wave give the scheduler MFMA work of one query block to put
// under the VALU work of the other.  A wave whose second query block lies entirely past token 576 skips it
// (wave-uniform branch), so the 577th query costs half a wave, as in v1.  The rescale of O is skipped when no row
// maximum of the wave moved (multiplying by exactly 1.0) -- bit-identical, saves 32 VALU per tile most of the time.
// ================================================================================================================
#define ATT2_QB 256
#define ATT2_NQB 3                       // ceil(577 / 256)

template <typename T>
__global__ __launch_bounds__(256, 2) void attention2_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 2 * VT_TILE_BYTES];
    char* ks0 = smem;
    char* vt0 = smem + 2 * K_TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, g = lane >> 5;

    // XCD-aware decode: the ATT2_NQB query blocks of one (image, head) pair share an XCD
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qb = slot % ATT2_NQB;
    const int pair = (slot / ATT2_NQB) * 8 + xcd;
    const int img = pair >> 4, head = pair & 15;
    const int64_t base = (int64_t)img * VIT_TOKENS;

    const int q_first = qb * ATT2_QB + wave * 64;           // wave-uniform
    const bool act0 = q_first < VIT_TOKENS;                  // first 32-query block has at least one valid query
    const bool act1 = q_first + 32 < VIT_TOKENS;             // second one too

    typename T::v8 qf[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qrow = q_first + b * 32 + lq;
        const int qr = qrow < VIT_TOKENS ? qrow : VIT_TOKENS - 1;
#pragma unroll
        for (int ksi = 0; ksi < 4; ++ksi)
            qf[b][ksi] = *(const typename T::v8*)(qkv + (base + qr) * QKV_LD + head * 64 + ksi * 16 + g * 8);
    }

    f32x16 o[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][db][r] = 0.f;
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};

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

        if (act0) {
            // ---- S^T = K Q^T for both query blocks: one K fragment read, two MFMAs (first one with C = 0) ----
            f32x16 s[2][2];
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ksi = 0; ksi < 4; ++ksi) {
                    const typename T::v8 kf = *(const typename T::v8*)(ks + (kb * 32 + lq) * K_ROWB + kxoff[ksi]);
                    s[0][kb] = T::mfma(kf, qf[0][ksi], ksi == 0 ? zero16 : s[0][kb]);
                    if (act1) s[1][kb] = T::mfma(kf, qf[1][ksi], ksi == 0 ? zero16 : s[1][kb]);
                }
            // ---- online softmax (base 2), per query block.  Instruction diet (the loop is VALU-bound: 32 scores per lane
            // per query block): row max as 16 v_max3_f32, s - m and the row sum as packed fp32 adds (2 per instruction),
            // 32 v_exp_f32, and P goes to 16 bits with the packed converts in the PV section below. ----
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b == 1 && !act1) break;
                if (t == ATT_NT - 1) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = t * ATT_KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                            if (key >= VIT_TOKENS) s[b][kb][r] = -1e30f;
                        }
                }
                float tmax = max3f(s[b][0][0], s[b][0][1], s[b][0][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) tmax = max3f(tmax, s[b][0][r], s[b][0][r + 1]);
                tmax = max3f(tmax, s[b][0][15], s[b][1][0]);
#pragma unroll
                for (int r = 1; r < 15; r += 2) tmax = max3f(tmax, s[b][1][r], s[b][1][r + 1]);
                tmax = max3f(tmax, s[b][1][15], s[b][1][15]);
                const float m_new = max3f(tmax, __shfl_xor(tmax, 32, 64), m[b]);   // the other half of the row's keys
                const bool moved = __builtin_amdgcn_ballot_w64(m_new > m[b]) != 0;   // wave-uniform
                const float alpha = __builtin_amdgcn_exp2f(m[b] - m_new);            // == 1.0f when this row's max stayed
                m[b] = m_new;
                const f32x2 m2 = {m_new, m_new};
                f32x2 ps2 = {0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 sv = {s[b][kb][r], s[b][kb][r + 1]};
                        const f32x2 d = sv - m2;
                        const f32x2 pv = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
                        s[b][kb][r] = pv[0]; s[b][kb][r + 1] = pv[1];
                        ps2 += pv;
                    }
                l[b] = l[b] * alpha + (ps2[0] + ps2[1]);
                if (moved) {
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[b][db][r] *= alpha;
                }
            }
            // ---- O^T += V^T P^T: one V^T fragment read, two MFMAs ----
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    typename T::v8 pf[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        u32x4 pw;
#pragma unroll
                        for (int w = 0; w < 4; ++w) pw[w] = T::pack2(s[b][kb][8 * s2 + 2 * w], s[b][kb][8 * s2 + 2 * w + 1]);
                        pf[b] = __builtin_bit_cast(typename T::v8, pw);
                    }
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const char* vrow = vt + (db * 32 + lq) * VT_STRIDE + (kb * 32 + 16 * s2 + 4 * g) * 2;
                        const u32x2 lo = *(const u32x2*)(vrow);        // keys +0..3
                        const u32x2 hi = *(const u32x2*)(vrow + 16);   // keys +8..11
                        u32x4 vw; vw[0] = lo[0]; vw[1] = lo[1]; vw[2] = hi[0]; vw[3] = hi[1];
                        const typename T::v8 vf = __builtin_bit_cast(typename T::v8, vw);
                        o[0][db] = T::mfma(vf, pf[0], o[0][db]);
                        if (act1) o[1][db] = T::mfma(vf, pf[1], o[1][db]);
                    }
                }
            }
        }

        if (t + 1 < ATT_NT) att_store_tile(st, ks0 + (cur ^ 1) * K_TILE_BYTES, vt0 + (cur ^ 1) * VT_TILE_BYTES, tid);
        __syncthreads();
    }

#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b == 0 ? !act0 : !act1) break;
        const float ltot = l[b] + __shfl_xor(l[b], 32, 64);
        const float inv = 1.0f / ltot;
        const int qrow = q_first + b * 32 + lq;
        if (qrow < VIT_TOKENS) {
            uint16_t* orow = out + (base + qrow) * VIT_HIDDEN + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    u32x2 pk;
                    pk[0] = pack16x2<T>(o[b][db][4 * q4] * inv, o[b][db][4 * q4 + 1] * inv);
                    pk[1] = pack16x2<T>(o[b][db][4 * q4 + 2] * inv, o[b][db][4 * q4 + 3] * inv);
                    *(u32x2*)(orow + db * 32 + 8 * q4 + 4 * g) = pk;
                }
        }
    }
}

// ================================================================================================================
// v3: v2's 64-query waves with the two query blocks of a wave STAGGERED so that, inside one wave's instruction stream,
// the matrix pipe always has work queued beside the softmax VALU work (rocprof on v2: MFMA busy 22 %, VALU active 40 %,
// waves parked 44 % -- the two co-resident waves of a SIMD run in lock step, so nothing overlapped):
//     QK(A) | QK(B) + softmax(A) | PV(A) + softmax(B) | PV(B)
// QK(B)'s MFMAs do not depend on softmax(A) and PV(A)'s do not depend on softmax(B); sched_group_barrier pins the
// interleave (1 MFMA : its share of VALU / transcendental / LDS-read instructions) the compiler would otherwise undo by
// clustering.  K / V^T fragments are re-read per query block (the LDS was never the limit).  The O rescale is
// unconditional (16 packed multiplies) to keep the tile body one basic block.
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

template <typename T>
__device__ __forceinline__ void att3_qk(f32x16 (&s)[2], const typename T::v8 (&qf)[4], const char* ks, const int (&kxoff)[4], int lq) {
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int ksi = 0; ksi < 4; ++ksi) {
            const typename T::v8 kf = *(const typename T::v8*)(ks + (kb * 32 + lq) * K_ROWB + kxoff[ksi]);
            s[kb] = T::mfma(kf, qf[ksi], ksi == 0 ? zero16 : s[kb]);
        }
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

// One 64-key tile for a wave.  NQ = 2: both query blocks (staggered); NQ = 1: only the first (the wave that holds token 576).
template <typename T, int NQ, bool LAST>
__device__ __forceinline__ void att3_tile(const char* ks, const char* vt, const typename T::v8 (&qf)[2][4], f32x16 (&o)[2][2],
                                          float (&m)[2], float (&l)[2], const int (&kxoff)[4], int lq, int g, int t) {
    f32x16 sA[2], sB[2];
    typename T::v8 pfA[2][2], pfB[2][2];
    att3_qk<T>(sA, qf[0], ks, kxoff, lq);                                      // QK(A)
    if (NQ == 2) att3_qk<T>(sB, qf[1], ks, kxoff, lq);                         // QK(B)      beside
    att3_softmax<T, LAST>(sA, o[0], m[0], l[0], pfA, t, g);                    // softmax(A)
    att3_pv<T>(o[0], pfA, vt, lq, g);                                          // PV(A)      beside
    if (NQ == 2) {
        att3_softmax<T, LAST>(sB, o[1], m[1], l[1], pfB, t, g);                // softmax(B)
        att3_pv<T>(o[1], pfB, vt, lq, g);                                      // PV(B)
    }
    if (NQ == 2) {
        // stage 1: QK(A) alone: 8 x {LDS read, MFMA}
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        // stage 2: QK(B) beside softmax(A): 8 x {LDS read, MFMA, 12 VALU, 4 transcendental}
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        // stage 3: PV(A) beside softmax(B): 8 x {2 LDS reads, MFMA, 12 VALU, 4 transcendental}
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        // stage 4: PV(B) alone
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attention3_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char smem[2 * K_TILE_BYTES + 2 * VT_TILE_BYTES];
    char* ks0 = smem;
    char* vt0 = smem + 2 * K_TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, g = lane >> 5;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qb = slot % ATT2_NQB;
    const int pair = (slot / ATT2_NQB) * 8 + xcd;
    const int img = pair >> 4, head = pair & 15;
    const int64_t base = (int64_t)img * VIT_TOKENS;

    const int q_first = qb * ATT2_QB + wave * 64;           // wave-uniform
    const bool act0 = q_first < VIT_TOKENS;
    const bool act1 = q_first + 32 < VIT_TOKENS;

    typename T::v8 qf[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int qrow = q_first + b * 32 + lq;
        const int qr = qrow < VIT_TOKENS ? qrow : VIT_TOKENS - 1;
#pragma unroll
        for (int ksi = 0; ksi < 4; ++ksi)
            qf[b][ksi] = *(const typename T::v8*)(qkv + (base + qr) * QKV_LD + head * 64 + ksi * 16 + g * 8);
    }
    f32x16 o[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][db][r] = 0.f;
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};
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
        if (act1) {
            if (t < ATT_NT - 1) att3_tile<T, 2, false>(ks, vt, qf, o, m, l, kxoff, lq, g, t);
            else att3_tile<T, 2, true>(ks, vt, qf, o, m, l, kxoff, lq, g, t);
        } else if (act0) {
            if (t < ATT_NT - 1) att3_tile<T, 1, false>(ks, vt, qf, o, m, l, kxoff, lq, g, t);
            else att3_tile<T, 1, true>(ks, vt, qf, o, m, l, kxoff, lq, g, t);
        }
        if (t + 1 < ATT_NT) att_store_tile(st, ks0 + (cur ^ 1) * K_TILE_BYTES, vt0 + (cur ^ 1) * VT_TILE_BYTES, tid);
        __syncthreads();
    }

#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b == 0 ? !act0 : !act1) break;
        const float ltot = l[b] + __shfl_xor(l[b], 32, 64);
        const float inv = 1.0f / ltot;
        const int qrow = q_first + b * 32 + lq;
        if (qrow < VIT_TOKENS) {
            uint16_t* orow = out + (base + qrow) * VIT_HIDDEN + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    u32x2 pk;
                    pk[0] = pack16x2<T>(o[b][db][4 * q4] * inv, o[b][db][4 * q4 + 1] * inv);
                    pk[1] = pack16x2<T>(o[b][db][4 * q4 + 2] * inv, o[b][db][4 * q4 + 3] * inv);
                    *(u32x2*)(orow + db * 32 + 8 * q4 + 4 * g) = pk;
                }
        }
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

__device__ __forceinline__ void att_dma_v(__amdgpu_buffer_rsrc_t rv, char* vs, int wave, int lane, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int grp = wave + 4 * i;
        const int row = grp * 8 + (lane >> 3);
        int key = t * ATT_KT + row; key = key < VIT_TOKENS ? key : VIT_TOKENS - 1;
        const int c = (lane & 7) ^ (((row >> 1) & 1) << 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (att_lds_void*)(vs + grp * 8 * K_ROWB), 16, key * (QKV_LD * 2) + c * 16, 0, 0, 0);
    }
}

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
template <typename T>
__device__ __forceinline__ void att5_pv(f32x16 (&o)[2], const typename T::v8 (&pf)[2][2], const char* vs, const int (&vbase)[2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const char* p = vs + vbase[db] + (kb * 32 + 16 * s2) * K_ROWB;
                const att_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(p));
                const att_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(p + 8 * K_ROWB));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                u32x4 vw; vw[0] = l2[0]; vw[1] = l2[1]; vw[2] = h2[0]; vw[3] = h2[1];
                o[db] = T::mfma(__builtin_bit_cast(typename T::v8, vw), pf[kb][s2], o[db]);
            }
}

template <typename T, int WAVES_PER_SIMD>
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
    att5_dma(rk, ks0, wave, dvo_k, 0);
    att5_dma(rv, vs0, wave, dvo_v, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ATT_NT; ++t) {
        const int cur = t & 1;
        if (t + 1 < ATT_NT) {                                 // both tiles of step t+1 land under this tile's math
            att5_dma(rk, ks0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_k, t + 1);
            att5_dma(rv, vs0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_v, t + 1);
        }
        const char* ks = ks0 + cur * K_TILE_BYTES;
        const char* vs = vs0 + cur * K_TILE_BYTES;
        if (wave_active) {
            f32x16 sA[2];
            typename T::v8 pfA[2][2];
            att3_qk<T>(sA, qf, ks, kxoff, lq);
            if (t < ATT_NT - 1) att3_softmax<T, false>(sA, o, m, l, pfA, t, g);   // (conditional O rescale: no gain measured)
            else att3_softmax<T, true>(sA, o, m, l, pfA, t, g);
            att5_pv<T>(o, pfA, vs, vbase);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMAs of step t+1 have landed
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

static int attention_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PIGEON_ATTN_VARIANT");
        v = e ? atoi(e) : 11;
        if (v < 1 || v > 12) v = 11;
    }
    return v;
}

int pg_attention_launch(int dtype, const void* qkv, void* out, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    const int pairs = n_images * VIT_HEADS;                  // always a multiple of 8
    const int var = attention_variant();
    const bool v2 = var == 2;
    const dim3 grid(pairs * ((var == 2 || var == 3) ? ATT2_NQB : ATT_NQB)), block(256);
    if ((var == 11 || var == 12) && (dtype == PG_DTYPE_F16 || dtype == PG_DTYPE_BF16)) {   // K and V by DMA, transposing LDS reads
        if (var == 11) {
            if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL((attention5_kernel<T_F16, 3>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
            else hipLaunchKernelGGL((attention5_kernel<T_BF16, 3>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        } else {
            if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL((attention5_kernel<T_F16, 4>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
            else hipLaunchKernelGGL((attention5_kernel<T_BF16, 4>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        }
        return pg_check_launch("attention");
    }
    if (var == 10 && (dtype == PG_DTYPE_F16 || dtype == PG_DTYPE_BF16)) {     // variant 4 with the K tile by direct-to-LDS DMA
        if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL((attention4_kernel<T_F16, 3, 0, true>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        else hipLaunchKernelGGL((attention4_kernel<T_BF16, 3, 0, true>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        return pg_check_launch("attention");
    }
    if (var >= 6 && var <= 9 && dtype == PG_DTYPE_F16) {         // ablations of variant 4 (timing only)
        if (var == 6) hipLaunchKernelGGL((attention4_kernel<T_F16, 3, 1>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        if (var == 7) hipLaunchKernelGGL((attention4_kernel<T_F16, 3, 2>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        if (var == 8) hipLaunchKernelGGL((attention4_kernel<T_F16, 3, 5>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        if (var == 9) hipLaunchKernelGGL((attention4_kernel<T_F16, 3, 7>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        return pg_check_launch("attention");
    }
    if ((var == 4 || var == 5) && (dtype == PG_DTYPE_F16 || dtype == PG_DTYPE_BF16)) {
        if (var == 4) {
            if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL((attention4_kernel<T_F16, 3>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
            else hipLaunchKernelGGL((attention4_kernel<T_BF16, 3>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        } else {
            if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL((attention4_kernel<T_F16, 4>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
            else hipLaunchKernelGGL((attention4_kernel<T_BF16, 4>), grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        }
        return pg_check_launch("attention");
    }
    if (var == 3 && (dtype == PG_DTYPE_F16 || dtype == PG_DTYPE_BF16)) {
        if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL(attention3_kernel<T_F16>, grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        else hipLaunchKernelGGL(attention3_kernel<T_BF16>, grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        return pg_check_launch("attention");
    }
    if (dtype == PG_DTYPE_F16) {
        if (v2) hipLaunchKernelGGL(attention2_kernel<T_F16>, grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        else hipLaunchKernelGGL(attention_kernel<T_F16>, grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    } else if (dtype == PG_DTYPE_BF16) {
        if (v2) hipLaunchKernelGGL(attention2_kernel<T_BF16>, grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        else hipLaunchKernelGGL(attention_kernel<T_BF16>, grid, block, 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    } else { pg_set_error("attention: dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16"); return PG_EINVAL; }
    return pg_check_launch("attention");
}
