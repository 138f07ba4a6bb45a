// attention.hip -- multi-head self-attention of the ViT (16 heads x 64, 577 tokens, no mask), flash style: the PRODUCT kernel
// (attention8_kernel, "v8") and its launcher.  The generations it superseded (v1, v4, v5 / v6): attention_old.hip on the branch archive/kernel-generations-r04
// and compile into the tools build only (python -m pigeon_amd.build --dev, PIGEON_ATTN_VARIANT 1, 4..15).
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
//   * K and V are walked in 64-key tiles, double-buffered in LDS, both by direct-to-LDS DMA (buffer_load ... lds) with
//     loop-invariant lane offsets and an SGPR tile offset; the descriptor's bounds check zero-fills keys past token 576.
//   * S^T = K Q^T leaves each lane owning queries, so row max / row sum are in-lane reductions plus lane exchanges; P feeds the PV
//     MFMA straight from those registers as the B operand (O^T = V^T P^T), V^T fragments come from transposing LDS reads.
//   * lazy online softmax in base 2: the QK^T accumulator is seeded with -m (minus the lane's reference maximum), m is raised only
//     when a tile's maximum exceeds it by more than 2^8 (wave-uniform branch).
//   * 577 = 9*64 + 1: nine full tiles through the MFMAs, key 576 as a single-key VALU step at the end.
#include "attention_common.h"

#include <cstdlib>
#include <type_traits>

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

// QB = NQB * WAVES * 16 queries per block:
//   128 (4 waves x 32)  5 blocks per (image, head) cover 640 query slots (63 idle in the last block), 5 K/V streams per head;
//   192 (6 waves x 32)  round 4: 577 = 3 x 192 + 1 -- 3 blocks cover queries 0..575 with no idle slot and 3 K/V streams per head;
//                       token 576's query is left to attention_last_query_kernel below.  Same 32 queries per wave, same registers
//                       (3 waves per SIMD: two 6-wave blocks per CU where three 4-wave blocks ran); waves 0..3 issue the tile DMAs.
template <typename T, int NQB, int WAVES, int WPS, bool LSUM = false>
__global__ __launch_bounds__(WAVES * 64, WPS) void attention8_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    constexpr int QB = NQB * WAVES * 16;
    static_assert(QB == 128 || QB == 192, "a block covers 128 or 192 queries");
    constexpr int NQBLK = QB == 192 ? 3 : ATT_NQB;
    constexpr int DW = WAVES < 4 ? WAVES : 4;               // waves that stage the K / V tiles
    static_assert(VIT_TOKENS == 9 * ATT_KT + 1, "key tail assumes 577 tokens");
    __shared__ __attribute__((aligned(16))) char smem[4 * K_TILE_BYTES];              // K0 K1 V0 V1, 8 KB each
    char* ks0 = smem;
    char* vs0 = smem + 2 * K_TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int qblk = slot % NQBLK;
    const int pair = (slot / NQBLK) * 8 + xcd;
    const int img = pair >> 4, head = pair & 15;
    const int64_t base = (int64_t)img * VIT_TOKENS;
    const int q_first = qblk * QB + wave * (NQB * 16);       // wave-uniform
    const bool wave_active = q_first < VIT_TOKENS;
    constexpr int Q_END = QB == 192 ? VIT_TOKENS - 1 : VIT_TOKENS;   // 192-query blocks stop at query 575

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
    constexpr int NDMA = 8 / DW;                             // 8-key groups of a 64-key tile a staging wave moves, per operand
    const bool stager = wave < DW;
    int dvo_k[NDMA], dvo_v[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int row = ((wave & (DW - 1)) + DW * i) * 8 + (lane >> 3);
        dvo_k[i] = row * (QKV_LD * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        dvo_v[i] = row * (QKV_LD * 2) + (((lane & 7) ^ (((row >> 1) & 3) << 1)) << 4);
    }
    constexpr int NFULL = 9;

    if (stager) {
        att8_dma(rk, ks0, wave, dvo_k, NDMA, DW, 0);
        att8_dma(rv, vs0, wave, dvo_v, NDMA, DW, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < NFULL; ++t) {
        const int cur = t & 1;
        if (t + 1 < NFULL && stager) {                        // both tiles of step t+1 land under this tile's math
            att8_dma(rk, ks0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_k, NDMA, DW, t + 1);
            att8_dma(rv, vs0 + (cur ^ 1) * K_TILE_BYTES, wave, dvo_v, NDMA, DW, t + 1);
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
        if (qrow < Q_END) {
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


#ifdef PIGEON_ABLATIONS
// ---- token 576's query of every (image, head), for the 192-query blocking (tools build: the A/B arms 22 / 23 below): one wave per pair, VALU.  s_k = q . k over the 577 keys
// (8 lanes share a key: 16-byte chunks of its row, shuffle-reduced), softmax in base 2 over LDS-resident scores (fp32 throughout:
// P is not rounded to 16 bits here), O = sum_k p_k v_k with the same 8-keys-per-pass walk.  ~150 KB of K/V per wave from L2:
// the main kernel has just streamed the same rows.
template <typename T>
__global__ __launch_bounds__(64) void attention_last_query_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out) {
    __shared__ float sc[VIT_TOKENS + 7];
    const int pair = blockIdx.x, img = pair >> 4, head = pair & 15, lane = threadIdx.x;
    const int sub = lane >> 3, ch = lane & 7;
    const int64_t base = (int64_t)img * VIT_TOKENS;
    const uint16_t* qrow = qkv + (base + VIT_TOKENS - 1) * QKV_LD + head * 64 + ch * 8;
    const u32x4 qq = *(const u32x4*)qrow;
    float q[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) { q[2 * w] = T::val((uint16_t)(qq[w] & 0xffffu)); q[2 * w + 1] = T::val((uint16_t)(qq[w] >> 16)); }
    const uint16_t* kbase = qkv + base * QKV_LD + 1024 + head * 64 + ch * 8;
    // 32 keys per pass (4 independent 16-byte loads per lane in flight), 8 lanes per key
    for (int k0 = 0; k0 < VIT_TOKENS; k0 += 32) {
        u32x4 kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + 8 * u + sub;
            kk[u] = key < VIT_TOKENS ? *(const u32x4*)(kbase + (int64_t)key * QKV_LD) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + 8 * u + sub;
            float p = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                p = fmaf(q[2 * w], T::val((uint16_t)(kk[u][w] & 0xffffu)), p);
                p = fmaf(q[2 * w + 1], T::val((uint16_t)(kk[u][w] >> 16)), p);
            }
            p += __shfl_xor(p, 1, 64); p += __shfl_xor(p, 2, 64); p += __shfl_xor(p, 4, 64);
            if (ch == 0 && key < VIT_TOKENS) sc[key] = p;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int k = lane; k < VIT_TOKENS; k += 64) m = fmaxf(m, sc[k]);
    m = wave_max(m);
    float l = 0.f;
    for (int k = lane; k < VIT_TOKENS; k += 64) { const float e = __builtin_amdgcn_exp2f(sc[k] - m); sc[k] = e; l += e; }
    l = wave_sum(l);
    __syncthreads();
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    const uint16_t* vbase = kbase + 1024;
    for (int k0 = 0; k0 < VIT_TOKENS; k0 += 32) {
        u32x4 vv[4];
        float pp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + 8 * u + sub;
            const bool ok = key < VIT_TOKENS;
            vv[u] = ok ? *(const u32x4*)(vbase + (int64_t)key * QKV_LD) : u32x4{0u, 0u, 0u, 0u};
            pp[u] = ok ? sc[key] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                o[2 * w] = fmaf(pp[u], T::val((uint16_t)(vv[u][w] & 0xffffu)), o[2 * w]);
                o[2 * w + 1] = fmaf(pp[u], T::val((uint16_t)(vv[u][w] >> 16)), o[2 * w + 1]);
            }
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = o[e];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        o[e] = v * inv;
    }
    if (sub == 0) {
        u32x4 pk;
#pragma unroll
        for (int w = 0; w < 4; ++w) pk[w] = pack16x2<T>(o[2 * w], o[2 * w + 1]);
        *(u32x4*)(out + (base + VIT_TOKENS - 1) * VIT_HIDDEN + head * 64 + ch * 8) = pk;
    }
}

#endif

#ifndef PG_DEFAULT_ATTN_VARIANT
#define PG_DEFAULT_ATTN_VARIANT 21
#endif
static int attention_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PIGEON_ATTN_VARIANT");
        v = e ? atoi(e) : PG_DEFAULT_ATTN_VARIANT;
        if (v < 1 || v > 23) v = PG_DEFAULT_ATTN_VARIANT;
    }
    return v;
}

template <typename KF, typename KB>
static int att_launch3(int dtype, KF kf, KB kb, dim3 grid, int threads, const void* qkv, void* out, hipStream_t s) {
    if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL(kf, grid, dim3(threads), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    else hipLaunchKernelGGL(kb, grid, dim3(threads), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
    return pg_check_launch("attention");
}

#if defined(PIGEON_ABLATIONS) && defined(PIGEON_OLD_GENERATIONS)
// attention_old.hip: the generations v1 / v4 / v5 / v6, archived on the branch archive/kernel-generations-r04 (round 5); a tools build
// that finds the file under tools/csrc/ again compiles and dispatches to it
int pg_attention_old_launch(int variant, int dtype, const void* qkv, void* out, dim3 grid, hipStream_t s, int* rc);
#endif

// Variants (env PIGEON_ATTN_VARIANT): 21 (default, the only one in the product library) = attention8_kernel, 32 queries per wave, row
// sums out of the matrix pipe.  Tools build only: 20 = the same with v_dot2c row sums, 19 = 64 queries per wave (A/B arms of this
// kernel); 1, 4..15 = the older generations (attention_old.hip, archived: branch archive/kernel-generations-r04).
int pg_attention_launch(int dtype, const void* qkv, void* out, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    if (dtype != PG_DTYPE_F16 && dtype != PG_DTYPE_BF16) { pg_set_error("attention: dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16"); return PG_EINVAL; }
    const int pairs = n_images * VIT_HEADS;                  // always a multiple of 8
    const dim3 grid(pairs * ATT_NQB);
    const int variant = attention_variant();
#ifdef PIGEON_ABLATIONS
    // Round 4 A/B arms (tools build; profiles/r04/attention_192_query_blocks_ab.txt): 192-query blocks (577 = 3 x 192 + 1) + the last
    // query as a micro-kernel.  22 = 6 waves x 32 queries: 1.46 vs 1.04 ms -- at 156 VGPRs (3 waves per SIMD) a CU places ONE 6-wave
    // block (2,2,1,1 waves per SIMD; a second one would need 4 on SIMD 0), where three 4-wave blocks ran.  23 = 4 waves x 48 queries
    // (212 VGPRs, 2 waves per SIMD): main kernel 0.97 ms (-6 %), but the last query alone streams every K / V row once more --
    // 1.2 GB through L2, 0.24 ms as a kernel of its own: 1.21 ms in total.  It would have to ride in a block that already stages K / V.
    if (variant == 22 || variant == 23) {                    // 192-query blocks + the last query's micro-kernel
        const int rc = variant == 22
            ? att_launch3(dtype, attention8_kernel<T_F16, 2, 6, 3, true>, attention8_kernel<T_BF16, 2, 6, 3, true>, dim3(pairs * 3), 384, qkv, out, s)
            : att_launch3(dtype, attention8_kernel<T_F16, 3, 4, 2, true>, attention8_kernel<T_BF16, 3, 4, 2, true>, dim3(pairs * 3), 256, qkv, out, s);
        if (rc != PG_OK) return rc;
        if (dtype == PG_DTYPE_F16) hipLaunchKernelGGL(attention_last_query_kernel<T_F16>, dim3(pairs), dim3(64), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        else hipLaunchKernelGGL(attention_last_query_kernel<T_BF16>, dim3(pairs), dim3(64), 0, s, (const uint16_t*)qkv, (uint16_t*)out);
        return pg_check_launch("attention (last query)");
    }
    if (variant == 19) return att_launch3(dtype, attention8_kernel<T_F16, 4, 2, 2>, attention8_kernel<T_BF16, 4, 2, 2>, grid, 128, qkv, out, s);
    if (variant == 20) return att_launch3(dtype, attention8_kernel<T_F16, 2, 4, 3, false>, attention8_kernel<T_BF16, 2, 4, 3, false>, grid, 256, qkv, out, s);
#ifdef PIGEON_OLD_GENERATIONS
    if (variant != 21) {
        int rc = PG_OK;
        if (pg_attention_old_launch(variant, dtype, qkv, out, grid, s, &rc) == 0) return rc;
    }
#endif
#endif
    if (variant != 21) {
        pg_set_error("attention: PIGEON_ATTN_VARIANT=%d is not part of this build (product: 21; others need -DPIGEON_ABLATIONS)", variant);
        return PG_EINVAL;
    }
    return att_launch3(dtype, attention8_kernel<T_F16, 2, 4, 3, true>, attention8_kernel<T_BF16, 2, 4, 3, true>, grid, 256, qkv, out, s);
}
