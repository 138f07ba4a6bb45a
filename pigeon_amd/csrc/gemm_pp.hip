// gemm_pp.hip -- persistent "ping-pong" GEMM: C[M,N] = A[M,K] * W[N,K]^T, fp16/bf16 operands, fp32 MFMA accumulation,
// the same fused epilogues as gemm_bf16.hip (which stays as the one-tile-per-block reference implementation).
//
// Replaces the nn.Linear GEMMs the reference reaches through transformers (modeling_clip.py CLIPAttention.q/k/v/
// out_proj, CLIPMLP.fc1/fc2, CLIPVisionEmbeddings.patch_embedding) -- SURVEY.md section 2c rows K1,K4,K6,K7,K8.
//
// What is different from gemm_bf16.hip (all three address losses measured in profiles/r01):
//   1. PERSISTENT blocks: one 512-thread block per CU walks tiles L0, L0+G, L0+2G, ...  The first K tile of the NEXT
//      output tile is DMA'd into LDS stage 0 before the epilogue of the current tile starts, the epilogue's global
//      stores are never waited for inside the epilogue, and no block launch / prologue latency sits between tiles.
//      (One-tile-per-block: every CU finishes its mainloop at the same moment, all 256 epilogues hit HBM together and
//      nothing computes meanwhile: ~8 us of a ~34 us tile.)
//   2. PING-PONG phases: the 8 waves form two groups (waves 0-3 / 4-7 = the two waves of every SIMD).  A k-step is
//      {LOAD phase: 6 ds_read_b128 fragments + this k-step's share of the next K tile's DMA; barrier; MFMA phase:
//      8 x v_mfma_f32_32x32x16 at raised priority; barrier}.  Group 1 runs one barrier behind group 0, so on every SIMD
//      one wave is in its MFMA phase while the other is in its LOAD phase: the matrix pipe always has exactly one
//      wave feeding it and LDS/DMA issue stalls never sit in front of an MFMA (cdna guide T3/T4/T5).
//   3. DMA through buffer descriptors (buffer_load_dwordx4 ... lds): one 32-bit VGPR offset per DMA that never
//      changes, the K advance is an SGPR offset, the M tail is handled by the descriptor's bounds check (rows past
//      M read as zero) -- no 64-bit per-lane address arithmetic in the loop.
//
// LDS map (160 KB): stage 0 [0,64K) | stage 1 [64K,128K) | spare [128K,160K).  A stage holds 256 A rows then 256 W rows
// of one K tile (BK = 64 -> 128-byte rows), lane-linear as the DMA requires, 16-byte chunk c of row r stored at
// chunk c ^ ((r>>1)&7) (swizzle applied on the source address) so ds_read_b128 fragments are conflict-free.
// K/64 is even for every GEMM of the model (16, 64, 10), so the last K tile of an output tile always sits in stage 1:
// the epilogue's transpose slabs (8 waves x 8.5 KB) live in [64K, 132K) while stage 0 already receives the next tile.
//
// Synchronisation (B_n = n-th s_barrier of an output tile; S = 4*K/64 k-steps):
//   group 0: LOAD(s) in (B_2s, B_2s+1), MFMA(s) in (B_2s+1, B_2s+2);  group 1: one interval later.
//   RAW  K tile t+1 is first read by group 0 after B_8t+8; every wave waits vmcnt(0) for its own DMAs of that tile at
//        the end of LOAD(4t+3): group 0 before B_8t+7, group 1 before B_8t+8.
//   WAR  stage of K tile t-1 is last read in group 1's LOAD(4t-1), which ends with lgkmcnt(0) before B_8t; the first
//        DMA into it is group 0's LOAD(4t), after B_8t.
//   Both groups execute 2S+2 barriers per tile (group 1: one extra after B_0, group 0: one extra at the end).
#include "gemm_epi.h"
#include "x3.h"
#include <type_traits>

namespace {

constexpr int PP_BM = 256, PP_BN = 256;
constexpr int PP_STAGE = (PP_BM + PP_BN) * ROWB;          // 64 KB
constexpr int PP_W_OFF = PP_BM * ROWB;                     // W rows start 32 KB into a stage
constexpr int PP_SLAB_OFF = PP_STAGE;                      // epilogue slabs overlay stage 1 (+ spare)
constexpr int PP_SLAB_ROWF = 64 + 4;                       // padded slab row, floats
constexpr int PP_SLAB_BYTES = 32 * PP_SLAB_ROWF * 4;       // 8704 B per wave
constexpr int PP_LDS = 160 * 1024;
#ifndef PP_DMA_AUX
#define PP_DMA_AUX 0                                     // cache policy of the operand DMAs (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
#ifndef PP_STORE_AUX
#define PP_STORE_AUX 0                                   // cache policy of the epilogue stores (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_uniform, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_uniform, 16, voff, soff, 0, PP_DMA_AUX);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

struct TileCtx {
    __amdgpu_buffer_rsrc_t ra, rw;
    int m0, n0;
    int part;                                                // PARTS (EPI_F32 with PgGemmExtra::parts > 1): which product this tile belongs to
};

// Tile order.  Logical ids are laid out band by band (a band = 32/gn M-panels x all N-panels); inside a band the ids
// walk gn-wide super-tiles: N fastest inside the super-tile, then M, then the next super-tile.  With gn = tilesN this
// is plain N-fastest order.  A persistent round gives every XCD 32 consecutive ids, i.e. one (32/gn) x gn super-tile:
// the unique operand bytes an XCD's L2 must fetch per K step are (32/gn + gn) panels (12 for 8 x 4, against 14.7 /
// 18 for N-fastest rows of 12 / 16 tiles).
template <int ABL, bool PARTS = false>
__device__ __forceinline__ TileCtx make_tile(const GemmArgs& g, int L) {
    TileCtx c;
    c.part = 0;
    const uint16_t* Ab = g.A;
    const uint16_t* Wb = g.W;
    if constexpr (PARTS) {                                   // tiles of part 0 first, then part 1, ...: same raster inside each part
        c.part = L / g.part_tiles;
        L -= c.part * g.part_tiles;
        Ab += (int64_t)c.part * g.ex.a_part;
        Wb += (int64_t)c.part * g.ex.w_part;
    }
    const int gmax = g.gn >= 32 ? 1 : 32 / g.gn;
    const int band_sz = gmax * g.tilesN;
    const int band = L / band_sz, rem = L - band * band_sz;
    const int gm = min(gmax, g.tilesM - band * gmax);
    const int sup = rem / (gm * g.gn), rem2 = rem - sup * gm * g.gn;
    const int im = rem2 / g.gn, in = rem2 - im * g.gn;
    const int tm = band * gmax + im, tn = sup * g.gn + in;
    c.m0 = tm * PP_BM; c.n0 = tn * PP_BN;
    const int rows = min(PP_BM, g.M - c.m0);
    if constexpr ((ABL & 2) != 0) {                          // ablation: every tile streams operand panel 0 (all L2 hits)
        c.ra = make_rsrc(g.A, (uint32_t)PP_BM * (uint32_t)g.lda * 2u);
        c.rw = make_rsrc(g.W, (uint32_t)PP_BN * (uint32_t)g.ldw * 2u);
    } else {
        c.ra = make_rsrc(Ab + (int64_t)c.m0 * g.lda, (uint32_t)rows * (uint32_t)g.lda * 2u);
        c.rw = make_rsrc(Wb + (int64_t)c.n0 * g.ldw, (uint32_t)PP_BN * (uint32_t)g.ldw * 2u);
    }
    return c;
}

// DMA d of a K tile: d 0..3 = this wave's four 8-row groups of A, d 4..7 = of W.
// SKIP (timing-only ablations 44 / 45): bit mask of DMAs left out -- what would 12.5 % / 25 % fewer operand bytes per tile buy?
template <int FROM, int CNT, int SKIP = 0>
__device__ __forceinline__ void issue_dma(const TileCtx& c, char* stage, int wave, const int (&voffA)[4], const int (&voffW)[4],
                                          int soff) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        if (d < FROM || d >= FROM + CNT) continue;
        if ((SKIP >> d) & 1) continue;
        if (d < 4) dma16(c.ra, stage + (wave + 8 * d) * 8 * ROWB, voffA[d], soff);
        else dma16(c.rw, stage + PP_W_OFF + (wave + 8 * (d - 4)) * 8 * ROWB, voffW[d - 4], soff);
    }
}

// MFMA shape: v_mfma_f32_16x16x32 (round 2; 2.06-2.09 PF/s against 1.72-1.74 for 32x32x16 at the power cap, tools/mfma_issue.hip).
// A k-step of 32 is computed in TWO halves of the wave's 128 rows, so the four {LOAD; barrier; MFMA; barrier} phases per 64-wide
// K tile of the 32x32x16 version stay as they were (256 matrix-pipe cycles each) at the price of 8 more fragment registers:
// phase (s, half) reads the four 16-row A blocks of rows [64 half, 64 half + 64) at k-step s (and, in the first half, the four
// 16-column W blocks, kept for the second half) and runs 16 MFMAs into accumulator blocks [4 half, 4 half + 4) x [0, 4).
typedef f32x4 AccPP[8][4];
template <typename T> struct Frag { typename T::v8 a[4], b[4]; };

template <typename T, bool WITH_B>
__device__ __forceinline__ void load_frag(Frag<T>& f, const char* sa, const char* sb, int xo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = *(const typename T::v8*)(sa + i * 16 * ROWB + xo);
    if constexpr (WITH_B) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *(const typename T::v8*)(sb + j * 16 * ROWB + xo);
    }
}

// swapped operands (weights as "A"): D[n][m] -> a lane owns output row m = lane & 15 of a 16 x 16 block and the 4 consecutive
// columns 4 (lane >> 4) ..  The MFMAs are inline asm with the accumulator tied (common.h mfma16_acc).
// ZERO: first k-step of an output tile -- the accumulator operand is the inline constant 0, so the tile loop does not
// spend 128 v_mov per wave (~512 cycles, 2.5 % of a K = 1024 tile) clearing registers between the epilogue and the mainloop.
template <typename T, bool ZERO, int HALF>
__device__ __forceinline__ void mma16(AccPP& acc, const Frag<T>& f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ZERO) T::mfma16_init(acc[HALF * 4 + i][j], f.b[j], f.a[i]);
            else T::mfma16_acc(acc[HALF * 4 + i][j], f.b[j], f.a[i]);
        }
}

// PG_PP_PHASES = 2 (round 4 experiment, tools/gemm_ab.py): a phase covers a whole k-step of 32 over all 128 rows of the wave -- 8 A
// + 4 W fragments (48 registers instead of 32), 32 MFMAs -- so a K tile has 2 phases and 4 CU-wide barriers per wave group instead
// of 4 and 8.  Same MFMA chain per accumulator over k: bit-identical.
#ifndef PG_PP_PHASES
#define PG_PP_PHASES 4
#endif
template <typename T> struct Frag2 { typename T::v8 a[8], b[4]; };
template <typename T>
__device__ __forceinline__ void load_frag2(Frag2<T>& f, const char* sa, const char* sb, int xo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f.a[i] = *(const typename T::v8*)(sa + i * 16 * ROWB + xo);
#pragma unroll
    for (int j = 0; j < 4; ++j) f.b[j] = *(const typename T::v8*)(sb + j * 16 * ROWB + xo);
}
template <typename T, bool ZERO>
__device__ __forceinline__ void mma32(AccPP& acc, const Frag2<T>& f) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ZERO) T::mfma16_init(acc[i][j], f.b[j], f.a[i]);
            else T::mfma16_acc(acc[i][j], f.b[j], f.a[i]);
        }
}

__device__ __forceinline__ void wait_lgkm0() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_vm0() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void raw_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// EPI_RESID: addressing of the fp32 residual rows this lane adds into (store side: 16 lanes per row, 4 rows per
// instruction, 8 instructions per 32-row slab).  The rows of slabs 0 and 1 are fetched during the SECOND-TO-LAST K tile of
// the mainloop (their 64 registers are idle there), slabs 2 and 3 as soon as the epilogue has parked the accumulators of
// slabs 0 / 1 in LDS: the epilogue used to pay one full memory latency per slab (~14 us per tile, rocprof ablations),
// now roughly one per tile.
struct XCtx {
    __amdgpu_buffer_rsrc_t ro;
    int voff, rstep, sstep;
};
// WIDE: the 8-columns-per-lane geometry of EPI_RESID_STAT (8 lanes per row, 8 rows per instruction, two 16-byte pieces)
template <bool WIDE = false>
__device__ __forceinline__ XCtx make_xctx(const GemmArgs& g, int row0, int col0, int lane) {
    XCtx x;
    int rv = g.M - row0; rv = rv < 0 ? 0 : (rv > 128 ? 128 : rv);
    rv = __builtin_amdgcn_readfirstlane(rv);   // descriptor stays in SGPRs (hipcc clamps with v_med3_i32, see gemm_pp6.hip rowstat_rsrc6)
    const uint32_t nbytes = rv > 0 ? (uint32_t)(((int64_t)(rv - 1) * g.ldc + 64) * 4) : 0u;
    x.ro = make_rsrc((const char*)g.out + ((int64_t)row0 * g.ldc + col0) * 4, nbytes);
    if (WIDE) {                                              // EPI_RESID_STAT: split-halves geometry, see pp_epilogue
        x.voff = ((lane >> 3) * (int)g.ldc + (lane & 7) * 4) * 4;
        x.rstep = 8 * (int)g.ldc * 4;
    } else {
        x.voff = ((lane >> 4) * (int)g.ldc + (lane & 15) * 4) * 4;
        x.rstep = 4 * (int)g.ldc * 4;
    }
    x.sstep = 32 * (int)g.ldc * 4;
    return x;
}
// slab 0 of the wide geometry: dst[it * 2 + h]
__device__ __forceinline__ void fetch_xrows_wide(u32x4 (&dst)[8], const XCtx& x) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int h = 0; h < 2; ++h)      // (readfirstlane: the context is built under a wave-uniform `if`, hipcc keeps rstep in a VGPR and
            dst[it * 2 + h] =            //  wraps every load in a waterfall loop otherwise)
                __builtin_amdgcn_raw_buffer_load_b128(x.ro, x.voff + 128 * h, __builtin_amdgcn_readfirstlane(it * x.rstep), 0);
}
__device__ __forceinline__ void fetch_xrows(u32x4 (&dst)[8], const XCtx& x, int slab) {
#pragma unroll
    for (int it = 0; it < 8; ++it)
        dst[it] = __builtin_amdgcn_raw_buffer_load_b128(x.ro, x.voff, __builtin_amdgcn_readfirstlane(slab * x.sstep + it * x.rstep), 0);
}

// One K tile in ping-pong form.  D0..D2 = number of this wave's 8 DMAs issued in LOAD phases 0..2 (rest in phase 3).
// XF: this call may also issue the early residual fetch (phases 2 and 3, AFTER the tile's DMAs, so that the counted
// vmcnt(16) at the end of phase 3 still means "my DMAs of the next K tile have landed").
// XF 2: EPI_RESID_STAT fetches slab 0 only (eight loads in phase 3; its slab-ahead double buffer covers the rest).
template <typename T, int D0, int D1, int D2, int ABL, int XF = 0, bool ZERO = false>
__device__ __forceinline__ void ktile_pp(AccPP& acc, const char* cur, char* nxt, int a_base, int b_base,
                                         const int (&xoff)[2], const TileCtx& c, int wave, const int (&voffA)[4],
                                         const int (&voffW)[4], int soff_next, bool has_next, bool xf, const XCtx& xc,
                                         u32x4 (&xq)[4][8]) {
    constexpr int D3 = 8 - D0 - D1 - D2;
    static_assert(D3 >= 0, "DMA schedule");
#if PG_PP_PHASES == 2
    {
        Frag2<T> f2;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            load_frag2<T>(f2, cur + a_base, cur + b_base, xoff[ss]);
            if (has_next) {
                if (ss == 0) issue_dma<0, D0 + D1, 0>(c, nxt, wave, voffA, voffW, soff_next);
                if (ss == 1) issue_dma<D0 + D1, D2 + D3, 0>(c, nxt, wave, voffA, voffW, soff_next);
            }
            if constexpr (XF == 1) {
                if (xf && ss == 1) { fetch_xrows(xq[0], xc, 0); fetch_xrows(xq[1], xc, 1); }
            }
            if constexpr (XF == 2) {
                if (xf && ss == 1) fetch_xrows_wide(xq[0], xc);
            }
            if (ss == 1 && has_next) {
                if (XF == 1 && xf) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (XF == 2 && xf) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wait_lgkm0();
            raw_barrier();
            __builtin_amdgcn_s_setprio(1);
            if (ss == 0) mma32<T, ZERO>(acc, f2);
            else mma32<T, false>(acc, f2);
            __builtin_amdgcn_s_setprio(0);
            raw_barrier();
        }
        return;
    }
#endif
#if PG_PP_PHASES == 8
    {   // eight phases per K tile: (k-step s, quarter q of the wave's 128 rows), 2 A fragments per phase, the 4 W fragments of a k-step
        // read in its first quarter; 8 MFMAs per phase, 16 barriers per K tile and wave group
        typename T::v8 fa[2], fb[4];
#pragma unroll
        for (int ph = 0; ph < 8; ++ph) {
            const int ss = ph >> 2, q = ph & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const typename T::v8*)(cur + a_base + (q * 2 + i) * 16 * ROWB + xoff[ss]);
            if (q == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *(const typename T::v8*)(cur + b_base + j * 16 * ROWB + xoff[ss]);
            }
            if (has_next) {
                if (ph == 0) issue_dma<0, 2, 0>(c, nxt, wave, voffA, voffW, soff_next);
                if (ph == 1) issue_dma<2, 2, 0>(c, nxt, wave, voffA, voffW, soff_next);
                if (ph == 2) issue_dma<4, 2, 0>(c, nxt, wave, voffA, voffW, soff_next);
                if (ph == 3) issue_dma<6, 2, 0>(c, nxt, wave, voffA, voffW, soff_next);
            }
            if constexpr (XF == 1) {
                if (xf && ph == 5) fetch_xrows(xq[0], xc, 0);
                if (xf && ph == 7) fetch_xrows(xq[1], xc, 1);
            }
            if constexpr (XF == 2) {
                if (xf && ph == 7) fetch_xrows_wide(xq[0], xc);
            }
            if (ph == 7 && has_next) {
                if (XF == 1 && xf) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (XF == 2 && xf) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            wait_lgkm0();
            raw_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ZERO && ss == 0) T::mfma16_init(acc[q * 2 + i][j], fb[j], fa[i]);
                    else T::mfma16_acc(acc[q * 2 + i][j], fb[j], fa[i]);
                }
            __builtin_amdgcn_s_setprio(0);
            raw_barrier();
        }
        return;
    }
#endif
    Frag<T> f;
    if constexpr ((ABL & 4) != 0) {                          // ablation: fragments read once per K tile (wrong results)
        load_frag<T, true>(f, cur + a_base, cur + b_base, xoff[0]);
        asm volatile("" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]));
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        // phase kk = (k-step s = kk >> 1, half = kk & 1)
        if constexpr ((ABL & 4) == 0) {
            if ((kk & 1) == 0) load_frag<T, true>(f, cur + a_base, cur + b_base, xoff[kk >> 1]);
            else load_frag<T, false>(f, cur + a_base + 64 * ROWB, cur + b_base, xoff[kk >> 1]);
        }
        if (has_next && (ABL & 1) == 0) {                    // ABL&1: ablation, no DMA inside the K loop (wrong results)
            constexpr int SKIP = (ABL & 8) ? 0x80 : ((ABL & 16) ? 0x88 : 0);
            if (kk == 0) issue_dma<0, D0, SKIP>(c, nxt, wave, voffA, voffW, soff_next);
            if (kk == 1) issue_dma<D0, D1, SKIP>(c, nxt, wave, voffA, voffW, soff_next);
            if (kk == 2) issue_dma<D0 + D1, D2, SKIP>(c, nxt, wave, voffA, voffW, soff_next);
            if (kk == 3) issue_dma<D0 + D1 + D2, D3, SKIP>(c, nxt, wave, voffA, voffW, soff_next);
        }
        if constexpr (XF == 1) {
            if (xf && kk == 2) fetch_xrows(xq[0], xc, 0);
            if (xf && kk == 3) fetch_xrows(xq[1], xc, 1);
        }
        if constexpr (XF == 2) {
            if (xf && kk == 3) fetch_xrows_wide(xq[0], xc);
        }
        if (kk == 3 && has_next) {
            if (XF == 1 && xf) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (XF == 2 && xf) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        wait_lgkm0();
        raw_barrier();
        __builtin_amdgcn_s_setprio(1);
        if (kk == 0) mma16<T, ZERO, 0>(acc, f);
        if (kk == 1) mma16<T, ZERO, 1>(acc, f);
        if (kk == 2) mma16<T, false, 0>(acc, f);
        if (kk == 3) mma16<T, false, 1>(acc, f);
        __builtin_amdgcn_s_setprio(0);
        raw_barrier();
    }
}

// Free-running form of the same K tile (MODE 0): one barrier per K tile (taken by the caller), fragments double
// buffered in registers, the DMAs of the next tile spread 2 per k-step between the MFMA groups (= gemm_bf16 variant 8).
template <typename T, bool ZERO = false>
__device__ __forceinline__ void ktile_free(AccPP& acc, const char* cur, char* nxt, int a_base, int b_base,
                                           const int (&xoff)[2], const TileCtx& c, int wave, const int (&voffA)[4],
                                           const int (&voffW)[4], int soff_next, bool has_next) {
    // (tools build only, variant 30; since the 16x16x32 conversion the fragments are read per phase, not double buffered)
    Frag<T> f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if ((kk & 1) == 0) load_frag<T, true>(f, cur + a_base, cur + b_base, xoff[kk >> 1]);
        else load_frag<T, false>(f, cur + a_base + 64 * ROWB, cur + b_base, xoff[kk >> 1]);
        if (has_next) {
            if (kk == 0) issue_dma<0, 2>(c, nxt, wave, voffA, voffW, soff_next);
            if (kk == 1) issue_dma<2, 2>(c, nxt, wave, voffA, voffW, soff_next);
            if (kk == 2) issue_dma<4, 2>(c, nxt, wave, voffA, voffW, soff_next);
            if (kk == 3) issue_dma<6, 2>(c, nxt, wave, voffA, voffW, soff_next);
        }
        wait_lgkm0();
        __builtin_amdgcn_s_setprio(1);
        if (kk == 0) mma16<T, ZERO, 0>(acc, f);
        if (kk == 1) mma16<T, ZERO, 1>(acc, f);
        if (kk == 2) mma16<T, false, 0>(acc, f);
        if (kk == 3) mma16<T, false, 1>(acc, f);
        __builtin_amdgcn_s_setprio(0);
    }
}

// ---- epilogue: per wave, four 32-row x 64-column fp32 slabs transposed through LDS -------------------------------
// A lane owns row m = lane&31 of the wave's 32x32 MFMA blocks; the slab turns that into row-major 16-byte pieces so a
// wave-wide store covers whole 128-byte (16-bit out) / 256-byte (fp32 out) row segments.  The slab is private to the
// wave, so only wave-level ordering is needed between its ds_writes and ds_reads (no block barrier).
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Bias of the lane's row-major columns, fetched at the START of an output tile (before the K loop) and forced to
// retire there: a global_load that is still "pending" in the compiler's scoreboard when the epilogue runs would make
// it insert vmcnt(0) in front of every use -- which also drains the epilogue's own stores and the next tile's DMA.
template <int EPI> constexpr bool epi_out16() { return EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool epi_ln() { return EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
// 8 columns per lane on the store side (16-bit outputs, and the fp32 residual epilogue that also emits the 16-bit copy)
template <int EPI> constexpr bool epi_wide() { return epi_out16<EPI>() || EPI == EPI_RESID_STAT; }

// Per-tile epilogue operands fetched at the START of the tile (or, for the next tile, before the current epilogue's
// stores): bias (lo/hi), for the LN epilogues colsum (slo/shi) and (rstd, mean*rstd) of the 16 rows this lane will store.
template <int EPI> struct EpiBias { f32x4 lo, hi, slo, shi; };

template <int EPI>
__device__ __forceinline__ void load_bias(EpiBias<EPI>& b, const GemmArgs& g, int col) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    b.lo = z; b.hi = z; b.slo = z; b.shi = z;
    if (g.bias) {
        b.lo = *(const f32x4*)(g.bias + col);
        if (epi_wide<EPI>()) b.hi = *(const f32x4*)(g.bias + col + (EPI == EPI_RESID_STAT ? 32 : 4));
    }
    if constexpr (epi_ln<EPI>()) {
        b.slo = *(const f32x4*)(g.ex.colsum + col);
        b.shi = *(const f32x4*)(g.ex.colsum + col + 4);
    }
}
template <int EPI>
__device__ __forceinline__ void zero_bias(EpiBias<EPI>& b) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    b.lo = z; b.hi = z;
}
// (rstd, mean*rstd) of the 16 rows this lane stores: issued right after the tile's first wait, lands under the K loop
template <int EPI>
__device__ __forceinline__ void load_rowstat(u32x2 (&rs)[4][4], const GemmArgs& g, int row0, int rr) {
    int rvs = g.M - row0; rvs = rvs < 0 ? 0 : (rvs > 128 ? 128 : rvs);
    rvs = __builtin_amdgcn_readfirstlane(rvs);   // descriptor stays in SGPRs (hipcc clamps with v_med3_i32, see gemm_pp6.hip rowstat_rsrc6)
    __amdgpu_buffer_rsrc_t rr_s = make_rsrc((const char*)g.ex.rowstat + (int64_t)row0 * 8, (uint32_t)rvs * 8u);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int it = 0; it < 4; ++it)
            rs[i][it] = __builtin_amdgcn_raw_buffer_load_b64(rr_s, (rr + i * 32 + it * 8) * 8, 0, 0);
}
template <int EPI>
__device__ __forceinline__ void pin_bias(EpiBias<EPI>& b) {
    // a use: the compiler's wait for the loads lands here
    if constexpr (epi_ln<EPI>()) asm volatile("" : "+v"(b.lo), "+v"(b.hi), "+v"(b.slo), "+v"(b.shi));
    else if constexpr (epi_wide<EPI>()) asm volatile("" : "+v"(b.lo), "+v"(b.hi));
    else asm volatile("" : "+v"(b.lo));
}

// Sum over the 8 consecutive lanes that hold one output row on the 8-columns-per-lane store side, result in every lane;
// fixed association order, so the row statistics depend on nothing but the data.
// (row8_sum and the epilogue element functions: gemm_epi.h)

template <typename T, int EPI, int XEARLY, typename PREFETCH>
__device__ __forceinline__ void pp_epilogue(AccPP& acc, const GemmArgs& g, char* smem, int wave, int lane,
                                            int row0, int col0, const EpiBias<EPI>& bias, const u32x2 (&rs)[4][4],
                                            PREFETCH&& prefetch_next, const XCtx& xc, u32x4 (&xq)[4][8], int dbg_iter = 0) {
    constexpr bool OUT16 = epi_out16<EPI>();
    constexpr bool LN = epi_ln<EPI>();
    constexpr bool STAT = (EPI == EPI_RESID_STAT);
    constexpr bool RESID = (EPI == EPI_RESID || EPI == EPI_RESID_STAT);
    constexpr bool WIDE = epi_wide<EPI>();
    constexpr bool X3 = (EPI == EPI_GELU_X3);                // fp16 triple [M][3N]: 4 columns per lane like EPI_F32, three 8-byte stores
    constexpr int ROWPF = PP_SLAB_ROWF;
    constexpr int ESZ = (OUT16 || X3) ? 2 : 4;
    constexpr int CPL = WIDE ? 8 : 4;                        // columns per lane on the row-major side
    constexpr int LPR = 64 / CPL, RPI = 64 / LPR, ITS = 32 / RPI;
    const int l15 = lane & 15, lq = lane >> 4;               // MFMA side: row inside a 16-row block, column quad
    float* slab = (float*)(smem + PP_SLAB_OFF + wave * PP_SLAB_BYTES);
    // EPI_RESID_STAT, "split halves": a lane holds columns 4k..4k+3 and 32+4k..32+4k+3 (k = lane & 7) of its row instead of the
    // 8 consecutive ones the 16-bit epilogues need.  Every fp32 load / store instruction then covers 8 rows x 128 CONTIGUOUS
    // bytes; with 8 consecutive columns per lane the two 16-byte halves of a lane were 32 bytes apart, each instruction touched
    // 64 half-used 32-byte pieces, and the CU's memory pipeline -- not HBM -- set the epilogue time (tools/epi_timeline.py: the
    // epilogue took 13 us per tile even with 16 of 256 CUs running).  The 16-bit copy becomes two 8-byte stores per lane.
    constexpr int HOFF = STAT ? 32 : 4;                      // column distance between the lane's two f32x4
    const int rr = lane / LPR, cc = STAT ? (lane & 7) * 4 : (lane % LPR) * CPL;
    const int col = col0 + cc;
    const float qsc = (EPI == EPI_QKV || EPI == EPI_QKV_LN) && col < g.qcols ? g.qscale : 1.f;

    if constexpr (EPI == EPI_PATCH) {
        prefetch_next();
        // rows are re-mapped (image, patch) -> token row and the position embedding is added: generic guarded path
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (ib * 16 + l15) * ROWPF + j * 16 + 4 * lq) = acc[2 * i + ib][j];
            wave_lds_fence();
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int r = it * RPI + rr;
                const int row = row0 + i * 32 + r;
                const f32x4 lo = *(const f32x4*)(slab + r * ROWPF + cc);
                if (row < g.M) epi_store_f32x4<EPI>(g, row, col, lo, bias.lo);
            }
            wave_lds_fence();
        }
        return;
    } else {
        // Output (and, for EPI_RESID*, residual input) through a buffer descriptor based at the wave's (row0, col0):
        // rows past M fail the bounds check (stores dropped, loads return 0), so there is no exec-mask branching and
        // no per-store 64-bit address arithmetic: voffset is one VGPR, the slab/iteration row offset is an SGPR.
        int rv = g.M - row0; rv = rv < 0 ? 0 : (rv > 128 ? 128 : rv);
        rv = __builtin_amdgcn_readfirstlane(rv);   // descriptor stays in SGPRs (hipcc clamps with v_med3_i32, see gemm_pp6.hip rowstat_rsrc6)
        // (EPI_GELU_X3: a row's three pieces sit N fp16 elements apart -- the range reaches 2N past the lane's first piece)
        const uint32_t nbytes = rv > 0 ? (uint32_t)(((int64_t)(rv - 1) * g.ldc + 64 + (X3 ? 2 * g.N : 0)) * ESZ) : 0u;
        __amdgpu_buffer_rsrc_t ro = make_rsrc((const char*)g.out + ((int64_t)row0 * g.ldc + col0) * ESZ, nbytes);
        const int voff = (rr * (int)g.ldc + cc) * ESZ;
        int rstep = RPI * (int)g.ldc * ESZ;                  // bytes between two store iterations
        int sstep = 32 * (int)g.ldc * ESZ;                   // bytes between two slabs
        // opaque to the optimiser: otherwise all 32 row offsets are hoisted out of the persistent tile loop and pinned
        // in SGPRs across the K loop (the fp32 epilogues then spill)
        asm volatile("" : "+s"(rstep), "+s"(sstep));

        // EPI_RESID_STAT: 16-bit copy of the new rows (one 16-byte store per lane; ldx == ldc, checked on the host, so its
        // byte offsets are half the fp32 ones) and per-row partial statistics (sum, sum of squares over this wave's 64
        // columns), parked in the 4 padding floats of the slab rows and written slot-major -- statpart[slot][row][2] --
        // with one coalesced 256-byte store per slab.
        __amdgpu_buffer_rsrc_t rx16 = ro;
        float* stat_base = nullptr;
        if constexpr (STAT) {
            const uint32_t nb16 = rv > 0 ? (uint32_t)(((int64_t)(rv - 1) * g.ldc + 64) * 2) : 0u;
            rx16 = make_rsrc((const char*)g.ex.x16 + ((int64_t)row0 * g.ldc + col0) * 2, nb16);
            stat_base = g.ex.statpart + ((int64_t)(col0 / 64) * g.ex.stat_rows + row0) * 2;
        }
        // LN epilogues issue the next tile's prefetch after slab 0 (their row statistics were loaded at the tile top and
        // must be consumed without waiting for anything younger); everything else issues it first.
        if constexpr (!LN) prefetch_next();
        // EPI_RESID*: the fp32 residual rows are fetched one slab ahead of their use (two register sets)
        constexpr int XPI = WIDE ? 2 : 1;                    // 16-byte pieces per lane per iteration
        u32x4 xr[2][RESID ? ITS : 1][RESID ? XPI : 1];
        auto fetch_x = [&](int i, int set) {
            if constexpr (RESID) {
#pragma unroll
                for (int it = 0; it < ITS; ++it)
#pragma unroll
                    for (int h = 0; h < XPI; ++h)
                        xr[set][it][h] = __builtin_amdgcn_raw_buffer_load_b128(ro, voff + HOFF * 4 * h, i * sstep + it * rstep, 0);
            }
        };
        if constexpr (XEARLY == 0) fetch_x(0, 0);
        if constexpr (XEARLY == 2) {                         // slab 0 came in during the last K-tile pair
#pragma unroll
            for (int it = 0; it < ITS; ++it)
#pragma unroll
                for (int h = 0; h < XPI; ++h) xr[0][it][h] = xq[0][it * 2 + h];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (XEARLY != 1) { if (i + 1 < 4) fetch_x(i + 1, (i + 1) & 1); }
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (ib * 16 + l15) * ROWPF + j * 16 + 4 * lq) = acc[2 * i + ib][j];
            // slabs 0 / 1 came in during the mainloop; with their accumulators parked, fetch the rows of slab i + 2
            if constexpr (XEARLY == 1) { if (i + 2 < 4) fetch_xrows(xq[i + 2], xc, i + 2); }
            wave_lds_fence();
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int r = it * RPI + rr;
                f32x4 lo = *(const f32x4*)(slab + r * ROWPF + cc);
                f32x4 hi = lo;
                if constexpr (WIDE) hi = *(const f32x4*)(slab + r * ROWPF + cc + HOFF);
                // The row offset of every store goes into the VGPR offset, NOT the SGPR soffset: with a register soffset
                // hipcc pads no wait states after a >64-bit buffer store and lets the next VALU overwrite the data
                // registers while the store is still reading them (measured on gfx950: dword 3 of ~3 % of such stores
                // corrupted).
                const int ooff = voff + (i * sstep + it * rstep);
                if constexpr (OUT16) {
                    // (rstd, mean * rstd) through an asm move into registers of their own: with the two halves of the loaded
                    // dwordx2 used directly, hipcc (ROCm 7.2) SLP-packs the fmas and broadcasts the LOW half for both (op_sel of
                    // the high half dropped): every row used rstd in place of mean*rstd (caught by test_ln_fold_building_blocks).
                    float rstd = 0.f, mrs = 0.f;
                    if constexpr (LN) {
                        asm("v_mov_b32 %0, %1" : "=v"(rstd) : "v"(rs[i][it][0]));
                        asm("v_mov_b32 %0, %1" : "=v"(mrs) : "v"(rs[i][it][1]));
                    }
                    const u32x4 pk = epi16_finish<T, EPI>(lo, hi, bias.lo, bias.hi, bias.slo, bias.shi, rstd, mrs, col0 < g.qcols, qsc);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, ro, ooff, 0, PP_STORE_AUX);
                } else if constexpr (RESID) {
                    const f32x4 x = epi_resid4(__builtin_bit_cast(f32x4, XEARLY == 1 ? xq[i][it] : xr[i & 1][it][0]), lo, bias.lo);
#ifdef PIGEON_ABLATIONS
                    // timing-only ablation (env PIGEON_EPI_ABL=1, WRONG RESULTS): the fp32 rows written as 8 instead of 16 bytes per
                    // lane -- the write traffic a residual stream stored as an fp16 value + fp16 correction pair would have
                    // (8 instead of 10 bytes per element in all)
                    const bool half_store = STAT && g.stagger == -11;
                    if (half_store) {
                        u32x2 hx2; hx2[0] = __builtin_bit_cast(u32x4, x)[0]; hx2[1] = __builtin_bit_cast(u32x4, x)[1];
                        __builtin_amdgcn_raw_buffer_store_b64(hx2, ro, ooff, 0, 0);
                    } else
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), ro, ooff, 0, PP_STORE_AUX);
                    if constexpr (STAT) {
                        const f32x4 y = epi_resid4(__builtin_bit_cast(f32x4, xr[i & 1][it][1]), hi, bias.hi);
#ifdef PIGEON_ABLATIONS
                        if (half_store) {
                            u32x2 hy2; hy2[0] = __builtin_bit_cast(u32x4, y)[0]; hy2[1] = __builtin_bit_cast(u32x4, y)[1];
                            __builtin_amdgcn_raw_buffer_store_b64(hy2, ro, ooff + HOFF * 4, 0, 0);
                        } else
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro, ooff + HOFF * 4, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(epi_copy16x4<T>(x), rx16, ooff >> 1, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(epi_copy16x4<T>(y), rx16, (ooff >> 1) + HOFF * 2, 0, 0);
                        float s1, s2;
                        epi_stat8(x, y, s1, s2);
                        s1 = row8_sum(s1);
                        s2 = row8_sum(s2);
                        if ((lane & 7) == 0) { slab[r * ROWPF + 64] = s1; slab[r * ROWPF + 65] = s2; }
                    }
                } else if constexpr (X3) {                   // EPI_GELU_X3: split_x3_kernel<true>'s arithmetic on (acc + bias)
                    f32x4 v = lo + bias.lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = x3_quick_gelu(v[e]);
                    u32x2 ph, pl, ps;
                    x3_pack4(v, ph, pl, ps);
                    const int nb = g.N * 2;                  // bytes between the pieces of a row
                    __builtin_amdgcn_raw_buffer_store_b64(ph, ro, ooff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(pl, ro, ooff + nb, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(ps, ro, ooff + 2 * nb, 0, 0);
                } else {                                     // EPI_F32
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo + bias.lo), ro, ooff, 0, 0);
                }
            }
            wave_lds_fence();                                // slab reads retired before the next slab overwrites it
            if constexpr (STAT) {
                if (lane < 32 && i * 32 + lane < rv)
                    *(u32x2*)(stat_base + (i * 32 + lane) * 2) = *(const u32x2*)(slab + lane * ROWPF + 64);
                wave_lds_fence();
            }
            PG_TS(g, dbg_iter, wave, 3 + i);
            if constexpr (LN) { if (i == 0) prefetch_next(); }
        }
    }
}

// MODE 0: free-running (one barrier per K tile).  MODE 1: ping-pong.  MODE 2: ping-pong phases without the stagger
// (both groups in lock step; A/B arm only).
template <typename T, int EPI, int MODE, int D0, int D1, int D2, int ABL>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool follower = (MODE == 1) && (wm == 1);

    // per-lane DMA offsets inside a tile (bytes), constant for the whole kernel
    int voffA[4], voffW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave + 8 * i) * 8 + (lane >> 3);      // row inside the 256-row operand panel
        const int c = (lane & 7) ^ ((r >> 1) & 7);           // logical 16-byte chunk this lane must fetch
        voffA[i] = r * (int)g.lda * 2 + c * 16;
        voffW[i] = r * (int)g.ldw * 2 + c * 16;
    }
    // fragment row lane & 15 of a 16-row block, k chunk (16 bytes) 4 s + (lane >> 4) of k-step s, XOR-swizzled with the row
    const int l15 = lane & 15, lq = lane >> 4;
    const int sw = (lane >> 1) & 7;
    int xoff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xoff[ks] = ((ks * 4 + lq) ^ sw) << 4;
    const int a_base = (wm * 128 + l15) * ROWB;
    const int b_base = PP_W_OFF + (wn * 64 + l15) * ROWB;

    constexpr int ECPL = epi_wide<EPI>() ? 8 : 4;
    // the lane's first column inside the wave's 64 on the store side (EPI_RESID_STAT: split halves, second f32x4 32 columns on)
    const int ecc = EPI == EPI_RESID_STAT ? (lane & 7) * 4 : (lane % (64 / ECPL)) * ECPL;
    const int nt = g.K / BK;                                 // even (checked on the host)
    const int nblk = gridDim.x;
    // EPI_F32 only (the exact mode's GEMMs): several independent products in one launch, see PgGemmExtra::parts.  The other
    // epilogues -- every kernel of the fast path -- compile exactly as before.
    constexpr bool PARTS = (EPI == EPI_F32) && ABL == 0;
    int L = xcd_remap(blockIdx.x, nblk);
    if (L >= g.ntiles) return;
    xcd_stagger_wait(g.xcd_stagger_ticks);
    if (g.stagger > 0) {
        // De-synchronise the persistent blocks: every tile takes the same time, so blocks that start together also
        // reach their epilogues together and HBM sees bursts (all CUs storing / fetching residual rows) separated by
        // idle mainloop stretches.  Slot s of an XCD starts s/32 of a tile period late.
        const int slot = (blockIdx.x >> 3) & 31;
        const long long until = (long long)__builtin_readcyclecounter() + (long long)g.stagger * slot / 32;
        while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(8);
    }
    TileCtx c = make_tile<ABL, PARTS>(g, L);
    issue_dma<0, 8>(c, smem, wave, voffA, voffW, 0);         // K tile 0 of the first output tile -> stage 0
    EpiBias<EPI> bias;
    const int err = lane / (64 / ECPL);                      // the lane's first row inside a 32-row slab on the store side
    load_bias<EPI>(bias, g, c.n0 + wn * 64 + ecc);
    if constexpr (PARTS) { if (c.part > 0) zero_bias<EPI>(bias); }       // the bias rides in part 0 only
    pin_bias<EPI>(bias);
    // Number of global stores an epilogue issues AFTER its last load / DMA.  vmcnt retires in order, so at the top of
    // the next tile `vmcnt(NST)` means "the prefetched K tile 0 (and the next bias) has landed" while the epilogue's
    // stores are still draining to HBM underneath the first K tile.  (The patch epilogue skips stores of out-of-range
    // waves, so its count is not static: full drain.)
    // (EPI_RESID_STAT issues more; 16 is a safe lower bound.  The LN epilogues prefetch after slab 0: 12 stores follow.)
    // (EPI_GELU_X3 issues 96; vmcnt counts to 63)
    constexpr int NST = (EPI == EPI_PATCH) ? 0 : (EPI == EPI_F32 ? 32 : (EPI == EPI_GELU_X3 ? 63 : (epi_ln<EPI>() ? 12 : 16)));
    bool first = true;
    int dbg_iter = 0;                                        // tile counter of the tools build's time stamps (dead code otherwise)

    while (true) {
        AccPP acc;                                           // not cleared: the first k-step of the tile runs with C = 0

        if (first || NST == 0) {
            wait_vm0();
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_lgkm0();
        raw_barrier();                                       // B_0: K tile 0 visible, previous epilogue's slabs released
        PG_TS(g, dbg_iter, wave, 0);
        u32x2 rs[4][4];
        if constexpr (epi_ln<EPI>()) load_rowstat<EPI>(rs, g, c.m0 + wm * 128, err);
        constexpr int XEARLY = (MODE == 0 || ABL != 0) ? 0 : (EPI == EPI_RESID ? 1 : (EPI == EPI_RESID_STAT ? 2 : 0));
        XCtx xc;
        u32x4 xq[4][8];
        if constexpr (MODE == 0) {
            for (int t = 0; t < nt; t += 2) {
                if (t > 0) { wait_vm0(); wait_lgkm0(); raw_barrier(); }
                if (t == 0) ktile_free<T, true>(acc, smem, smem + PP_STAGE, a_base, b_base, xoff, c, wave, voffA, voffW, ROWB, true);
                else ktile_free<T>(acc, smem, smem + PP_STAGE, a_base, b_base, xoff, c, wave, voffA, voffW, (t + 1) * ROWB, true);
                wait_vm0(); wait_lgkm0(); raw_barrier();
                ktile_free<T>(acc, smem + PP_STAGE, smem, a_base, b_base, xoff, c, wave, voffA, voffW, (t + 2) * ROWB, t + 2 < nt);
            }
            wait_lgkm0(); raw_barrier();
        } else {
            if (follower) raw_barrier();
            int t0 = 0;
            if (nt > 2) {                                     // first K-tile pair peeled: its first k-step runs with C = 0
                ktile_pp<T, D0, D1, D2, ABL, 0, true>(acc, smem, smem + PP_STAGE, a_base, b_base, xoff, c, wave, voffA, voffW,
                                                          ROWB, true, false, xc, xq);
                ktile_pp<T, D0, D1, D2, ABL, 0>(acc, smem + PP_STAGE, smem, a_base, b_base, xoff, c, wave, voffA, voffW,
                                                    2 * ROWB, true, false, xc, xq);
                t0 = 2;
            } else {                                          // K = 128: one pair, which may carry the early residual fetch
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
            }
            for (int t = t0; t < nt; t += 2) {
                const bool xf = XEARLY != 0 && (t + 2 == nt);
                if (XEARLY != 0 && xf) xc = make_xctx<XEARLY == 2>(g, c.m0 + wm * 128, c.n0 + wn * 64, lane);
                ktile_pp<T, D0, D1, D2, ABL, XEARLY>(acc, smem, smem + PP_STAGE, a_base, b_base, xoff, c, wave, voffA, voffW,
                                                     (t + 1) * ROWB, true, xf, xc, xq);
                ktile_pp<T, D0, D1, D2, ABL, 0>(acc, smem + PP_STAGE, smem, a_base, b_base, xoff, c, wave, voffA, voffW,
                                                    (t + 2) * ROWB, t + 2 < nt, false, xc, xq);
            }
            PG_TS(g, dbg_iter, wave, 1);
            if (MODE == 1 && !follower) raw_barrier();       // re-align: every wave has left the mainloop
        }

        // the MFMAs are inline asm, so hipcc pads no "matrix-pipe write -> VALU / LDS read" hazard for the accumulators; a
        // follower wave comes here straight from its last MFMA phase (one barrier, which normally covers the 4-pass latency)
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        PG_TS(g, dbg_iter, wave, 2);
        const int row0 = c.m0 + wm * 128, col0 = c.n0 + wn * 64;
        const int part_now = c.part;                         // (prefetch_next below replaces c by the NEXT tile's context)
        L += nblk;
        const bool more = L < g.ntiles;
        EpiBias<EPI> bias_next = bias;
        // UNCONDITIONAL on purpose (the last tile of a block re-fetches its own first K tile into the idle stage 0, 64 KB of
        // wasted DMA per block): a VMEM block under `if (more)` between the early residual loads and their use makes hipcc
        // count its in-order vmcnt waits along the path WITHOUT the block, i.e. on the common path it waits for ten extra
        // (younger) operations -- a full memory latency at the start of every epilogue.
        auto prefetch_next = [&]() {
            c = make_tile<ABL, PARTS>(g, more ? L : L - nblk);
            issue_dma<0, 8>(c, smem, wave, voffA, voffW, 0);      // next output tile's K tile 0 -> stage 0 (free since K tile nt-2)
            load_bias<EPI>(bias_next, g, c.n0 + wn * 64 + ecc);   // older than the epilogue's last stores: see NST
            if constexpr (PARTS) { if (c.part > 0) zero_bias<EPI>(bias_next); }
        };
        if constexpr (PARTS) {
            GemmArgs ge = g;                                 // what the epilogue sees: this tile's own output buffer
            ge.out = (float*)g.out + (int64_t)part_now * g.ex.c_part;
            pp_epilogue<T, EPI, XEARLY>(acc, ge, smem, wave, lane, row0, col0, bias, rs, prefetch_next, xc, xq, dbg_iter);
        } else {
            pp_epilogue<T, EPI, XEARLY>(acc, g, smem, wave, lane, row0, col0, bias, rs, prefetch_next, xc, xq, dbg_iter);
        }
        PG_TS(g, dbg_iter, wave, 9);
        ++dbg_iter;
        if (!more) break;
        pin_bias<EPI>(bias_next);
        bias = bias_next;
        first = false;
    }
}

template <typename T, int EPI, int MODE, int D0, int D1, int D2, int ABL>
int launch_pp(const GemmArgs& g, int nblk, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm_pp_kernel<T, EPI, MODE, D0, D1, D2, ABL>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
        if (e != hipSuccess) { pg_set_error("gemm_pp: set LDS attr: %s", hipGetErrorString(e)); return PG_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(512), PP_LDS, s, g);
    return pg_check_launch("gemm_pp");
}

template <typename T, int MODE, int D0, int D1, int D2, int ABL = 0>
int launch_pp_epi(const GemmArgs& g, int epi, int nblk, hipStream_t s) {
    if constexpr (ABL != 0) {                                // ablations: timing only, two epilogues are enough
        switch (epi) {
            case EPI_QKV: case EPI_GELU: return launch_pp<T, EPI_QKV, MODE, D0, D1, D2, ABL>(g, nblk, s);
            default: return launch_pp<T, EPI_RESID, MODE, D0, D1, D2, ABL>(g, nblk, s);
        }
    } else {
        switch (epi) {
            case EPI_QKV: return launch_pp<T, EPI_QKV, MODE, D0, D1, D2, 0>(g, nblk, s);
            case EPI_GELU: return launch_pp<T, EPI_GELU, MODE, D0, D1, D2, 0>(g, nblk, s);
            case EPI_RESID: return launch_pp<T, EPI_RESID, MODE, D0, D1, D2, 0>(g, nblk, s);
            case EPI_PATCH: return launch_pp<T, EPI_PATCH, MODE, D0, D1, D2, 0>(g, nblk, s);
            case EPI_F32: return launch_pp<T, EPI_F32, MODE, D0, D1, D2, 0>(g, nblk, s);
            case EPI_RESID_STAT: case EPI_QKV_LN: case EPI_GELU_LN: case EPI_GELU_X3:
                // the LayerNorm-fold epilogues (and the exact mode's fused fc1 epilogue) are built only for the production schedule
                // (ping-pong, DMA 4/4/0/0)
                if constexpr (MODE == 1 && D0 == 4 && D1 == 4 && D2 == 0) {
                    if (epi == EPI_GELU_X3) {
                        if constexpr (std::is_same<T, T_F16>::value) return launch_pp<T, EPI_GELU_X3, MODE, D0, D1, D2, 0>(g, nblk, s);
                        else { pg_set_error("gemm_pp: EPI_GELU_X3 exists for fp16 operands only (the exact mode)"); return PG_EINVAL; }
                    }
                    if (epi == EPI_RESID_STAT) return launch_pp<T, EPI_RESID_STAT, MODE, D0, D1, D2, 0>(g, nblk, s);
                    if (epi == EPI_QKV_LN) return launch_pp<T, EPI_QKV_LN, MODE, D0, D1, D2, 0>(g, nblk, s);
                    return launch_pp<T, EPI_GELU_LN, MODE, D0, D1, D2, 0>(g, nblk, s);
                } else {
                    pg_set_error("gemm_pp: epilogue %d exists only in variants 33 / 36 / 38 / 39", epi);
                    return PG_EINVAL;
                }
            default: pg_set_error("gemm_pp: bad epilogue %d", epi); return PG_EINVAL;
        }
    }
}

template <typename T>
int dispatch_pp(GemmArgs& g, int epi, int variant, int nblk, hipStream_t s) {
    g.gn = g.tilesN;                                         // N-fastest raster unless the variant says otherwise
    switch (variant) {
        case 33: return launch_pp_epi<T, 1, 4, 4, 0>(g, epi, nblk, s);    // ping-pong, DMA 4/4/0/0, N-fastest raster
        case 36: if (g.tilesN % 4 == 0) g.gn = 4; return launch_pp_epi<T, 1, 4, 4, 0>(g, epi, nblk, s);   // 33 + 8x4 super-tile raster (product)
#ifdef PIGEON_ABLATIONS                                                     // tools build only (python -m pigeon_amd.build --dev)
        case 30: return launch_pp_epi<T, 0, 2, 2, 2>(g, epi, nblk, s);    // persistent, free-running
        case 31: return launch_pp_epi<T, 1, 3, 3, 2>(g, epi, nblk, s);    // ping-pong, DMA 3/3/2/0
        case 34: return launch_pp_epi<T, 2, 4, 4, 0>(g, epi, nblk, s);    // phases without stagger
        case 37: if (g.tilesN % 4 == 0) g.gn = 4; return launch_pp_epi<T, 1, 3, 3, 2>(g, epi, nblk, s);   // 31 + 8x4 super-tile raster
        case 38: g.stagger = (g.K / BK) * 2600 + 6000; return launch_pp_epi<T, 1, 4, 4, 0>(g, epi, nblk, s);   // 33 + staggered start
        case 39: g.stagger = (g.K / BK) * 1300 + 3000; return launch_pp_epi<T, 1, 4, 4, 0>(g, epi, nblk, s);   // 33 + half-period stagger
        // (DMA schedules 8/0/0/0, 6/2/0/0, 5/3/0/0, 4/2/2/0 measured within +-3 % of 4/4/0/0 -- box-to-box noise -- and removed)
        // ablations of 33 (timing only, WRONG RESULTS by construction)
        case 40: return launch_pp_epi<T, 1, 4, 4, 0, 1>(g, epi, nblk, s);          // no DMA in the K loop
        case 41: return launch_pp_epi<T, 1, 4, 4, 0, 2>(g, epi, nblk, s);          // every DMA hits panel 0 (L2 resident)
        case 42: return launch_pp_epi<T, 1, 4, 4, 0, 4>(g, epi, nblk, s);          // no fragment ds_reads
        case 43: return launch_pp_epi<T, 1, 4, 4, 0, 5>(g, epi, nblk, s);          // MFMA + barriers only
        case 44: return launch_pp_epi<T, 1, 4, 4, 0, 8>(g, epi, nblk, s);          // 7 of 8 operand DMAs (-12.5 % bytes)
        case 45: return launch_pp_epi<T, 1, 4, 4, 0, 16>(g, epi, nblk, s);         // 6 of 8 operand DMAs (-25 % bytes)
#endif
        default: pg_set_error("gemm_pp: variant %d is not part of this build (product: 33, 36; others need -DPIGEON_ABLATIONS)", variant); return PG_EINVAL;
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

}  // namespace

int pg_gemm_pp_launch(int dtype, GemmArgs g, int epi, int variant, hipStream_t s) {
    if (g.N % PP_BN != 0 || g.K % (2 * BK) != 0) {
        pg_set_error("gemm_pp: N %% 256 or K %% 128 != 0 (N=%d K=%d)", g.N, g.K);
        return PG_EINVAL;
    }
    if ((int64_t)g.lda * 2 * PP_BM >= (1ll << 31) || (int64_t)g.ldw * 2 * PP_BN >= (1ll << 31)) {
        pg_set_error("gemm_pp: operand panel exceeds the 2 GB buffer-descriptor range");
        return PG_EINVAL;
    }
    g.tilesM = (g.M + PP_BM - 1) / PP_BM;
    g.tilesN = g.N / PP_BN;
    g.part_tiles = g.tilesM * g.tilesN;
    if (g.ex.parts < 1) g.ex.parts = 1;
    if (g.ex.parts > 1 && epi != EPI_F32) { pg_set_error("gemm_pp: parts > 1 exists for EPI_F32 only (epi = %d)", epi); return PG_EINVAL; }
    if ((int64_t)g.part_tiles * g.ex.parts >= (1ll << 30)) { pg_set_error("gemm_pp: too many tiles"); return PG_EINVAL; }
    g.ntiles = g.part_tiles * g.ex.parts;
    int cap = num_cus();
    if (pg_gemm_block_cap() > 0 && pg_gemm_block_cap() < cap) cap = pg_gemm_block_cap();   // tuning: share the chip between streams
    const int nblk = g.ntiles < cap ? g.ntiles : cap;
    if (dtype == PG_DTYPE_F16) return dispatch_pp<T_F16>(g, epi, variant, nblk, s);
    if (dtype == PG_DTYPE_BF16) return dispatch_pp<T_BF16>(g, epi, variant, nblk, s);
    pg_set_error("gemm_pp: operand dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16 (got %d)", dtype);
    return PG_EINVAL;
}
