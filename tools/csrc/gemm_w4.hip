// gemm_w4.hip -- persistent GEMM with ONE wave per SIMD: C[M,N] = A[M,K] * W[N,K]^T, fp16/bf16 operands, fp32 MFMA
// accumulation, the fused epilogues of gemm_pp.hip.  EXPERIMENTAL (variant 64, not the default): it matches the 8-wave
// ping-pong kernels but does not beat them -- the measurements that came out of building it are the point.
//
// Why it was built.  Round 2 calibrated the GEMMs of gemm_pp.hip / gemm_pp6.hip against hipBLASLt on the same box
// (profiles/r02/hipblaslt_probe.txt, epilogue_costs.txt): with a plain 16-bit-store epilogue the 8-wave ping-pong mainloop
// needs 1.65-1.72 us per 256x256x64 K tile per CU, hipBLASLt's hand-written kernel 1.45 us.  Its structure (read off its name
// and instruction mix, not its code): 256 x 256 x 64 tiles, FOUR waves per workgroup -- one per SIMD -- each owning a 128 x 128
// wave tile whose 256 accumulator registers live in AGPRs, v_mfma_f32_16x16x32, operands DMA'd straight into LDS, three
// barriers per K tile, MFMAs back to back with the LDS reads of the next k-step between them.  This file is that structure,
// written from scratch on this library's building blocks (buffer-descriptor DMA with the bank swizzle on the source address,
// lane-linear LDS image, slab-transposed epilogues, XCD-aware super-tile raster).
//
// What was measured (MI355X, 295 424 x 1024 x 4096 / x 1024, per 64 of K per CU, profiles/r02/w4_*.txt):
//   v_mfma_f32_32x32x16, fragment reads one per MFMA pair, 16 DMAs right behind the barrier   1.98 us
//   + reads in the first half of a k-step, 8 MFMAs before the barrier, DMAs in k-step 0       1.92 us
//   + v_mfma_f32_16x16x32 (this file)                                                          1.72 us   (ping-pong: 1.65-1.69)
//   ... without the operand DMA (timing-only ablation)                                         0.96 us   = 2.06 PFLOP/s chip-wide
//   ... without the fragment reads (timing-only ablation)                                      1.50 us
//   32-wide K tiles in four 32 KB stages, DMAs spread one per eight MFMAs                      1.92 us   (rejected)
// i.e. the MFMA stream itself is not the limit (tools/mfma_issue.hip: 1844 TFLOP/s for 16x16x32 at the 1400 W cap, 1505 for
// 32x32x16: the 16x16x32 form moves half as many accumulator registers per flop); the global -> LDS operand feed costs 44 % of
// the K-tile time, and with a single wave per SIMD every non-MFMA instruction takes ~6 cycles away from the matrix pipe
// (mfma_issue: 35.3 instead of 32 cycles per MFMA with one ds_read per two MFMAs) where the ping-pong kernels issue them from
// the other wave of the SIMD.
//
// LDS map (160 KB): stage 0 [0,64K) | stage 1 [64K,128K) | spare [128K,160K).  A stage = 256 A rows then 256 W rows of one
// 64-wide K tile (128-byte rows), 16-byte chunk c of row r stored at chunk c ^ ((r>>1)&7).  The epilogue's four per-wave
// slabs (32 rows x 132 floats = 16.5 KB each, 66 KB) start at 64K: stage 1 and the first 2 KB of the spare.  K/64 is even,
// so the last K tile of an output tile sits in stage 1 and stage 0 is free during the epilogue.
//
// K-tile stream and synchronisation.  One barrier per K tile: before it, every wave has waited for its own DMAs of the
// next K tile and for its last fragment reads of the current one; after it the stage of the current K tile is free.  The
// DMAs of K tile u + 1 are issued in k-step 0 of K tile u (one after every fourth MFMA) into the stage freed by the barrier
// that ended K tile u - 1, and have the rest of K tile u to land.  The stream continues across output tiles: the last K tile
// (stage 1) prefetches K tile 0 of the NEXT output tile into stage 0, where it lands under the epilogue (whose slabs live in
// stage 1).  Within a k-step (32 of K, 64 MFMAs) the 16 fragment reads of the next k-step sit between the first 32 MFMAs, so
// their latency is covered by the other 32; the k-step that carries the barrier issues 32 MFMAs BEFORE it (the matrix pipe
// works through them while the wave waits) and the next K tile's first fragment reads right behind it.
//
// v_mfma_f32_16x16x32 sums the k products of one instruction in another association than the 32x32x16 kernels (variants
// 8 / 36 / 56): outputs agree with those to fp32 rounding (checked with a tolerance, tools/gemm_pp_check.py), not bit for bit;
// the row-statistics partials use the same 64-column slices and summation order.
#include "gemm_epi.h"

#include <cstdlib>

namespace {

constexpr int W4_BM = 256, W4_BN = 256;
constexpr int W4_STAGE = (W4_BM + W4_BN) * ROWB;          // 64 KB
constexpr int W4_W_OFF = W4_BM * ROWB;
constexpr int W4_SLAB_OFF = W4_STAGE;                      // slabs overlay stage 1 (+ 2 KB of the spare)
constexpr int W4_SLAB_ROWF = 128 + 4;                      // padded slab row, floats (528 B: rows shift by 4 banks)
constexpr int W4_SLAB_BYTES = 32 * W4_SLAB_ROWF * 4;       // 16 896 B per wave
constexpr int W4_LDS = 160 * 1024;
static_assert(W4_SLAB_OFF + 4 * W4_SLAB_BYTES <= W4_LDS, "slabs must fit behind stage 0");

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_uniform, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_uniform, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
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
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

struct Tile4 {
    __amdgpu_buffer_rsrc_t ra, rw;
    int m0, n0;
};

// band / super-tile order of gemm_pp.hip make_tile: a persistent round gives every XCD one 8 x 4 super-tile
__device__ __forceinline__ Tile4 make_tile4(const GemmArgs& g, int L) {
    Tile4 c;
    const int gmax = g.gn >= 32 ? 1 : 32 / g.gn;
    const int band_sz = gmax * g.tilesN;
    const int band = L / band_sz, rem = L - band * band_sz;
    const int gm = min(gmax, g.tilesM - band * gmax);
    const int sup = rem / (gm * g.gn), rem2 = rem - sup * gm * g.gn;
    const int im = rem2 / g.gn, in = rem2 - im * g.gn;
    const int tm = band * gmax + im, tn = sup * g.gn + in;
    c.m0 = tm * W4_BM; c.n0 = tn * W4_BN;
    const int rows = min(W4_BM, g.M - c.m0);
    c.ra = make_rsrc(g.A + (int64_t)c.m0 * g.lda, (uint32_t)rows * (uint32_t)g.lda * 2u);
    c.rw = make_rsrc(g.W + (int64_t)c.n0 * g.ldw, (uint32_t)W4_BN * (uint32_t)g.ldw * 2u);
    return c;
}

// The wave's 16 DMAs of one K tile: d 0..7 = its eight 8-row groups of A (rows (wave + 4 d) * 8 ..), d 8..15 = of W.
// voffA / voffW: the lane's offset inside group d = 0 (the swizzle term only sees (row>>1)&7 = 4 (wave&1) + (lane>>4), the
// same for every d); the group advance (32 rows) goes into the VGPR offset for A -- the descriptor's bounds check, which
// clips the M tail, does not cover the SGPR soffset -- and into the soffset for W (N is a multiple of 256: no tail).
// The K advance (koff bytes) is an soffset for both: it never moves a valid row out of, or an invalid row into, the bound.
__device__ __forceinline__ void issue_dma4(const Tile4& c, char* stage, int wave, int voffA, int voffW, int sa, int sw, int koff) {
#pragma unroll
    for (int d = 0; d < 8; ++d) dma16(c.ra, stage + (wave + 4 * d) * 8 * ROWB, voffA + d * sa, koff);
#pragma unroll
    for (int d = 0; d < 8; ++d) dma16(c.rw, stage + W4_W_OFF + (wave + 4 * d) * 8 * ROWB, voffW, koff + d * sw);
}

// fragments of one k-step of 32: eight 16-row blocks of A and of W, 16 bytes (8 k-values) per lane each
template <typename T> struct Frag4 { typename T::v8 a[8], b[8]; };
typedef f32x4 Acc4[8][8];                                   // 64 blocks of 16 x 16, 4 registers per lane each = 256 AGPRs

// Which MFMA.  tools/mfma_issue.hip (profiles/r02/mfma_issue.txt), full chip, one wave per SIMD, accumulators in AGPRs,
// nothing but MFMAs: v_mfma_f32_32x32x16_f16 sustains 1505 TFLOP/s at the 1400 W cap, v_mfma_f32_16x16x32_f16 1844 -- the
// 16x16x32 form moves half as many accumulator registers per flop (4 in + 4 out per 16 384 flop against 16 + 16 per 32 768),
// and at the power cap that is clock.  It is also what hipBLASLt's kernel uses (MI16x16x1).  Its k = 32 products are summed
// inside one instruction, so the results differ in the last bits from the 32x32x16 kernels of this library (same fp32
// accumulation, different association): this kernel is checked against them with a tolerance, and bit-exactly against itself.
//
// The mainloop is written as inline asm in program order (MFMA, LDS fragment reads, waits), because hipcc cannot be left to
// allocate it: with 256 accumulator registers live across the K loop it keeps part of them in VGPRs at the loop boundary and
// the rest in AGPRs, copies 176 registers between the two files every K tile and spills 450 (first build of this kernel).
// "+a" pins every accumulator block to AGPRs for the whole tile; the fragment reads are asm so that they stay between the
// MFMAs where they are written (the compiler would hoist them in front); s_waitcnt is explicit.  The compiler's hazard
// recogniser does not look inside asm: the only MFMA hazards here are (1) an MFMA reading the accumulator written by the
// MFMA 64 instructions earlier (none) and (2) the first read of the accumulators after the last MFMA (the epilogue: padded
// by hand, see w4_drain).  With one wave per SIMD every non-MFMA instruction costs the matrix pipe ~6 cycles (mfma_issue:
// 35.3 instead of 32 cycles per MFMA with one ds_read per two MFMAs), so the loop carries as few of them as possible.
template <typename T> struct Mfma4;
template <> struct Mfma4<T_F16> {
    static __device__ __forceinline__ void acc(f32x4& c, const f16x8& a, const f16x8& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void init(f32x4& c, const f16x8& a, const f16x8& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    }
};
template <> struct Mfma4<T_BF16> {
    static __device__ __forceinline__ void acc(f32x4& c, const bf16x8& a, const bf16x8& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void init(f32x4& c, const bf16x8& a, const bf16x8& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    }
};

#ifndef W4_ABL
#define W4_ABL 0                                            // tools build: 1 = no DMA in the K loop, 2 = no fragment reads (timing only)
#endif
template <int OFF, typename V>
__device__ __forceinline__ void lds_read_b128(V& dst, uint32_t addr) {
#if W4_ABL == 2
    asm volatile("" : "+v"(dst) : "v"(addr));
#else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
#endif
}
__device__ __forceinline__ void asm_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// fragment r of a k-step: r 0..7 = A blocks (16 rows each), r 8..15 = W blocks; aA / aB = LDS byte addresses of the lane's row
// (lane & 15) in block 0 at this k-step's chunk (lane >> 4, swizzled)
template <int R, typename T>
__device__ __forceinline__ void read_frag(Frag4<T>& f, uint32_t aA, uint32_t aB) {
    if constexpr (R < 8) lds_read_b128<R * 16 * ROWB>(f.a[R], aA);
    else lds_read_b128<(R - 8) * 16 * ROWB>(f.b[R - 8], aB);
}
template <int R0, int CNT, typename T>
__device__ __forceinline__ void read_frags(Frag4<T>& f, uint32_t aA, uint32_t aB) {
    if constexpr (CNT > 0) { read_frag<R0>(f, aA, aB); read_frags<R0 + 1, CNT - 1>(f, aA, aB); }
}

// MFMA m (0..63) of a k-step: block (i, j) = (m / 8, m % 8); swapped operands (weights first): D[n][m'] -- a lane owns output
// row lane & 15 of block i and the 4 consecutive columns 4 (lane >> 4) .. of block j
template <int M, bool ZERO, typename T>
__device__ __forceinline__ void mma_one(Acc4& acc, const Frag4<T>& f) {
    constexpr int i = M / 8, j = M % 8;
    if constexpr (ZERO) Mfma4<T>::init(acc[i][j], f.b[j], f.a[i]);
    else Mfma4<T>::acc(acc[i][j], f.b[j], f.a[i]);
}
template <int M0, int CNT, bool ZERO, typename T>
__device__ __forceinline__ void mma_run(Acc4& acc, const Frag4<T>& f) {
    if constexpr (CNT > 0) { mma_one<M0, ZERO>(acc, f); mma_run<M0 + 1, CNT - 1, ZERO>(acc, f); }
}

// One DMA of the wave's 16 per K tile
template <int D>
__device__ __forceinline__ void dma_one(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, char* stage, int wave, int voffA,
                                        int voffW, int sa, int sw, int koff) {
#if W4_ABL != 1
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (D < 8) dma16(ra, stage + (wave + 4 * D) * 8 * ROWB, voffA + D * sa, koff);
    else dma16(rw, stage + W4_W_OFF + (wave + 4 * (D - 8)) * 8 * ROWB, voffW, koff + (D - 8) * sw);
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// k-step 0 of a K tile: 64 MFMAs on `fc`; the 16 fragment reads of k-step 1 (into `fn`) one after every second MFMA of the
// first half -- the wait at the end comes 32 MFMAs after the last read -- and the wave's 16 DMAs of the NEXT K tile of the
// stream, one after every fourth MFMA, into the other stage (freed by the barrier that ended the previous K tile)
// SCHED (round 3 experiment, env PIGEON_W4_SCHED): 0 = as described above (one DMA per four MFMAs over the whole k-step: the last
// one is issued 36 MFMAs before the wave waits for it); 1 = all sixteen DMAs in the first 32 MFMAs, two per quad next to the two
// fragment reads (one non-MFMA instruction per MFMA, as in hipBLASLt's loop), the second half of the k-step pure MFMAs: the last
// DMA then has 64 MFMAs (~1000 matrix-pipe cycles) to land.
template <int Q, bool ZERO, int SCHED, typename T>
__device__ __forceinline__ void kstep0_quad(Acc4& acc, const Frag4<T>& fc, Frag4<T>& fn, uint32_t aA, uint32_t aB,
                                            __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, char* stage, int wave, int voffA,
                                            int voffW, int sa, int sw, int koff) {
    if constexpr (Q < 16) {
        mma_one<4 * Q, ZERO>(acc, fc);
        if constexpr (SCHED == 0) dma_one<Q>(ra, rw, stage, wave, voffA, voffW, sa, sw, koff);
        else if constexpr (Q < 8) dma_one<2 * Q>(ra, rw, stage, wave, voffA, voffW, sa, sw, koff);
        mma_one<4 * Q + 1, ZERO>(acc, fc);
        if constexpr (Q < 8) read_frag<2 * Q>(fn, aA, aB);
        mma_one<4 * Q + 2, ZERO>(acc, fc);
        if constexpr (SCHED == 1 && Q < 8) dma_one<2 * Q + 1>(ra, rw, stage, wave, voffA, voffW, sa, sw, koff);
        mma_one<4 * Q + 3, ZERO>(acc, fc);
        if constexpr (Q < 8) read_frag<2 * Q + 1>(fn, aA, aB);
        kstep0_quad<Q + 1, ZERO, SCHED>(acc, fc, fn, aA, aB, ra, rw, stage, wave, voffA, voffW, sa, sw, koff);
    }
}
template <bool ZERO, int SCHED, typename T>
__device__ __forceinline__ void kstep4_dma(Acc4& acc, const Frag4<T>& fc, Frag4<T>& fn, uint32_t aA, uint32_t aB,
                                           __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, char* stage, int wave, int voffA,
                                           int voffW, int sa, int sw, int koff) {
    kstep0_quad<0, ZERO, SCHED>(acc, fc, fn, aA, aB, ra, rw, stage, wave, voffA, voffW, sa, sw, koff);
    asm_wait_lgkm0();
}

// k-step 1 of a K tile that is not the last of its output tile: the first 32 MFMAs are issued BEFORE the wait for the next K
// tile's DMAs and the barrier (the matrix pipe works through them while the wave sits in the barrier), the fragment reads of
// the next K tile's k-step 0 right behind the barrier, the other 32 MFMAs cover their latency
template <int M, typename T>
__device__ __forceinline__ void close_tail(Acc4& acc, const Frag4<T>& fc, Frag4<T>& fn, uint32_t aA, uint32_t aB) {
    if constexpr (M < 64) {
        mma_one<M, false>(acc, fc);
        if constexpr (M - 32 + 4 < 16) read_frag<M - 32 + 4>(fn, aA, aB);
        close_tail<M + 1>(acc, fc, fn, aA, aB);
    }
}
template <typename T>
__device__ __forceinline__ void kstep4_close(Acc4& acc, const Frag4<T>& fc, Frag4<T>& fn, uint32_t aA, uint32_t aB) {
    mma_run<0, 32, false>(acc, fc);
    wait_vm0();
    raw_barrier();
    read_frags<0, 4>(fn, aA, aB);
    close_tail<32>(acc, fc, fn, aA, aB);
    asm_wait_lgkm0();
}

// the epilogue's first read of the accumulators must not overtake the last MFMA: explicit wait states
__device__ __forceinline__ void w4_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// One K tile u < nt - 1 of the stream; f[0] holds the fragments of its k-step 0 on entry and of the next K tile's k-step 0 on
// exit.  Its k-step 0 issues the DMAs of K tile u + 1 (same output tile) into the other stage.  FIRST: u == 0 of an output
// tile, the first k-step accumulates from 0.  Branch-free: a branch around the asm would make hipcc merge 256 accumulator
// registers at the join and spill them.
template <typename T, bool FIRST, int SCHED>
__device__ __forceinline__ void ktile4(Acc4& acc, Frag4<T> (&f)[2], char* smem, int u, const Tile4& c, int wave,
                                       int voffA, int voffW, int sa, int sw, uint32_t baseA, uint32_t baseB) {
    const uint32_t so = (uint32_t)(u & 1) * W4_STAGE, sn = W4_STAGE - so;
    // k-step 1 reads chunk (4 + (lane >> 4)) ^ swz = chunk0 ^ 4: address ^ 64 (disjoint bits)
    kstep4_dma<FIRST, SCHED>(acc, f[0], f[1], (baseA + so) ^ 64u, (baseB + so) ^ 64u, c.ra, c.rw, smem + sn, wave, voffA, voffW, sa, sw,
                      (u + 1) * ROWB);
    kstep4_close<T>(acc, f[1], f[0], baseA + sn, baseB + sn);
}

// the last K tile of an output tile (u == nt - 1, stage 1; nt is even): its k-step 0 prefetches K tile 0 of the NEXT output
// tile into stage 0 -- it lands under the rest of this K tile and the epilogue -- and nothing is read past its k-step 1
template <typename T, int SCHED>
__device__ __forceinline__ void ktile4_last(Acc4& acc, Frag4<T> (&f)[2], char* smem, const Tile4& cn, int wave,
                                            int voffA, int voffW, int sa, int sw, uint32_t baseA, uint32_t baseB) {
    const uint32_t so = W4_STAGE;
    kstep4_dma<false, SCHED>(acc, f[0], f[1], (baseA + so) ^ 64u, (baseB + so) ^ 64u, cn.ra, cn.rw, smem, wave, voffA, voffW, sa, sw, 0);
    mma_run<0, 64, false>(acc, f[1]);
}

template <int EPI> constexpr bool w4_out16() { return EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool w4_ln() { return EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool w4_resid() { return EPI == EPI_RESID || EPI == EPI_RESID_STAT; }

struct Bias4 { f32x4 lo, hi, slo, shi; };

template <int EPI>
__device__ __forceinline__ void load_bias4(Bias4& b, const GemmArgs& g, int col) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    b.lo = z; b.hi = z; b.slo = z; b.shi = z;
    if (g.bias) { b.lo = *(const f32x4*)(g.bias + col); b.hi = *(const f32x4*)(g.bias + col + 4); }
    if constexpr (w4_ln<EPI>()) { b.slo = *(const f32x4*)(g.ex.colsum + col); b.shi = *(const f32x4*)(g.ex.colsum + col + 4); }
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov4(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the 8 consecutive lanes that hold one 64-column slice of an output row (same association order as gemm_pp.hip)
__device__ __forceinline__ float row8_sum4(float v) {
    v += dpp_mov4<0xB1>(v);
    v += dpp_mov4<0x4E>(v);
    v += dpp_mov4<0x141>(v);
    return v;
}

// Epilogue: per wave four 32-row x 128-column fp32 slabs transposed through LDS; on the row-major side a lane owns 8
// consecutive columns (16 lanes per row, 4 rows per store instruction, 8 instructions per slab).
template <typename T, int EPI>
__device__ __forceinline__ void epilogue4(Acc4& acc, const GemmArgs& g, char* smem, int wave, int lane, int row0,
                                          int col0, const Bias4& bias) {
    constexpr bool OUT16 = w4_out16<EPI>();
    constexpr bool LN = w4_ln<EPI>();
    constexpr bool RESID = w4_resid<EPI>();
    constexpr bool STAT = (EPI == EPI_RESID_STAT);
    constexpr int ROWPF = W4_SLAB_ROWF;
    constexpr int ESZ = OUT16 ? 2 : 4;
    const int lrow = lane & 31, lhalf = lane >> 5;           // row / 64-column slice of the statistics write-out
    const int l15 = lane & 15, lq = lane >> 4;               // MFMA side: row inside a 16-row block, column quad
    float* slab = (float*)(smem + W4_SLAB_OFF + wave * W4_SLAB_BYTES);
    const int rr = lane >> 4, cc = (lane & 15) * 8;
    const int col = col0 + cc;
    const float qsc = ((EPI == EPI_QKV || EPI == EPI_QKV_LN) && col < g.qcols) ? g.qscale : 1.f;

    int rv = g.M - row0; rv = rv < 0 ? 0 : (rv > 128 ? 128 : rv);

    rv = __builtin_amdgcn_readfirstlane(rv);   // descriptor stays in SGPRs (hipcc clamps with v_med3_i32, see gemm_pp6.hip rowstat_rsrc6)
    const uint32_t nbytes = rv > 0 ? (uint32_t)(((int64_t)(rv - 1) * g.ldc + 128) * ESZ) : 0u;
    __amdgpu_buffer_rsrc_t ro = make_rsrc((const char*)g.out + ((int64_t)row0 * g.ldc + col0) * ESZ, nbytes);
    const int voff = (rr * (int)g.ldc + cc) * ESZ;
    int rstep = 4 * (int)g.ldc * ESZ;                        // bytes between two store iterations (4 rows)
    int sstep = 32 * (int)g.ldc * ESZ;                       // bytes between two slabs
    asm volatile("" : "+s"(rstep), "+s"(sstep));
    __amdgpu_buffer_rsrc_t rx16 = ro, rrs = ro;
    float* stat_base = nullptr;
    if constexpr (STAT) {
        const uint32_t nb16 = rv > 0 ? (uint32_t)(((int64_t)(rv - 1) * g.ldc + 128) * 2) : 0u;
        rx16 = make_rsrc((const char*)g.ex.x16 + ((int64_t)row0 * g.ldc + col0) * 2, nb16);
        stat_base = g.ex.statpart + ((int64_t)(col0 / 64 + lhalf) * g.ex.stat_rows + row0) * 2;     // this lane's slice on the write-out
    }
    if constexpr (LN) rrs = make_rsrc((const char*)g.ex.rowstat + (int64_t)row0 * 8, (uint32_t)rv * 8u);

    // one slab ahead: residual rows (16 x 16 bytes per lane) and / or the rows' (rstd, mean*rstd)
    u32x4 xr[2][RESID ? 8 : 1][RESID ? 2 : 1];
    u32x2 rs[2][LN ? 8 : 1];
    auto fetch = [&](int i, int set) {
        if constexpr (RESID) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    xr[set][it][h] = __builtin_amdgcn_raw_buffer_load_b128(ro, voff + 16 * h, i * sstep + it * rstep, 0);
        }
        if constexpr (LN) {
#pragma unroll
            for (int it = 0; it < 8; ++it) rs[set][it] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (i * 32 + it * 4 + rr) * 8, 0, 0);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) fetch(i + 1, (i + 1) & 1);
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int j = 0; j < 8; ++j) *(f32x4*)(slab + (ib * 16 + l15) * ROWPF + j * 16 + 4 * lq) = acc[2 * i + ib][j];
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + rr;
            f32x4 lo = *(const f32x4*)(slab + r * ROWPF + cc);
            f32x4 hi = *(const f32x4*)(slab + r * ROWPF + cc + 4);
            // row offset in the VGPR offset, not an SGPR soffset (hipcc pads no wait states after a >64-bit buffer store with
            // a register soffset: gemm_pp.hip)
            const int ooff = voff + (i * sstep + it * rstep);
            if constexpr (OUT16) {
                if constexpr (LN) {
                    // each scalar through an asm move of its own (hipcc SLP-packs the fmas into v_pk_fma_f32 and drops the
                    // op_sel of the high half of the loaded pair: gemm_pp.hip)
                    float rstd, mrs;
                    asm("v_mov_b32 %0, %1" : "=v"(rstd) : "v"(rs[i & 1][it][0]));
                    asm("v_mov_b32 %0, %1" : "=v"(mrs) : "v"(rs[i & 1][it][1]));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lo[e] = fmaf(lo[e], rstd, fmaf(-mrs, bias.slo[e], bias.lo[e]));
                        hi[e] = fmaf(hi[e], rstd, fmaf(-mrs, bias.shi[e], bias.hi[e]));
                    }
                } else {
                    lo += bias.lo; hi += bias.hi;
                }
                if constexpr (EPI == EPI_QKV || EPI == EPI_QKV_LN) {
                    // qcols is a multiple of 8: a lane's 8 columns are all inside or all outside (x * 1.0f is exact)
                    lo *= qsc; hi *= qsc;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { lo[e] = quick_gelu(lo[e]); hi[e] = quick_gelu(hi[e]); }
                }
                u32x4 pk;
                pk[0] = pack16x2<T>(lo[0], lo[1]); pk[1] = pack16x2<T>(lo[2], lo[3]);
                pk[2] = pack16x2<T>(hi[0], hi[1]); pk[3] = pack16x2<T>(hi[2], hi[3]);
                __builtin_amdgcn_raw_buffer_store_b128(pk, ro, ooff, 0, 0);
            } else if constexpr (RESID) {
                f32x4 x = __builtin_bit_cast(f32x4, xr[i & 1][it][0]);
                f32x4 y = __builtin_bit_cast(f32x4, xr[i & 1][it][1]);
                x += lo + bias.lo;                           // same expression as epi_store_f32x4<EPI_RESID>
                y += hi + bias.hi;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), ro, ooff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), ro, ooff + 16, 0, 0);
                if constexpr (STAT) {
                    u32x4 h4;
                    h4[0] = pack16x2<T>(x[0], x[1]); h4[1] = pack16x2<T>(x[2], x[3]);
                    h4[2] = pack16x2<T>(y[0], y[1]); h4[3] = pack16x2<T>(y[2], y[3]);
                    __builtin_amdgcn_raw_buffer_store_b128(h4, rx16, ooff >> 1, 0, 0);
                    const float s1 = row8_sum4(((x[0] + x[1]) + (x[2] + x[3])) + ((y[0] + y[1]) + (y[2] + y[3])));
                    const float s2 = row8_sum4(((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) +
                                               ((y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3])));
                    // the row's two 64-column slices park their (sum, sum of squares) in the 4 padding floats of the slab row
                    if ((lane & 7) == 0) {
                        const int sl = (lane >> 3) & 1;
                        slab[r * ROWPF + 128 + 2 * sl] = s1;
                        slab[r * ROWPF + 128 + 2 * sl + 1] = s2;
                    }
                }
            } else {                                         // EPI_F32
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo + bias.lo), ro, ooff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi + bias.hi), ro, ooff + 16, 0, 0);
            }
        }
        wave_lds_fence();                                    // slab reads retired before the next slab overwrites it
        if constexpr (STAT) {
            // statpart[slot][row][2], slot = 64-column slice: lanes 0..31 write slice 0 of the 32 rows, lanes 32..63 slice 1
            if (i * 32 + lrow < rv)
                *(u32x2*)(stat_base + (i * 32 + lrow) * 2) = *(const u32x2*)(slab + lrow * ROWPF + 128 + 2 * lhalf);
            wave_lds_fence();
        }
    }
}

template <typename T, int EPI, int SCHED>
__global__ __launch_bounds__(256) void gemm_w4_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // per-lane DMA offset inside 8-row group 0 (bytes): row wave * 8 + lane / 8, swizzled chunk
    const int r0 = wave * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((r0 >> 1) & 7);
    const int voffA = r0 * (int)g.lda * 2 + ch * 16;
    const int voffW = r0 * (int)g.ldw * 2 + ch * 16;
    const int sa = 32 * (int)g.lda * 2, sw = 32 * (int)g.ldw * 2;       // bytes between two of the wave's 8-row groups
    const int l15 = lane & 15;
    const int xo0 = ((lane >> 4) ^ ((lane >> 1) & 7)) << 4;  // k-step 0's chunk (lane >> 4) of the lane's fragment row, swizzled
    const uint32_t smem0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t baseA = smem0 + (wm * 128 + l15) * ROWB + xo0;
    const uint32_t baseB = smem0 + W4_W_OFF + (wn * 128 + l15) * ROWB + xo0;
    const int ecc = (lane & 15) * 8;

    const int nt = g.K / BK;                                 // even, >= 2 (checked on the host)
    const int nblk = gridDim.x;
    int L = xcd_remap(blockIdx.x, nblk);
    if (L >= g.ntiles) return;
    xcd_stagger_wait(g.xcd_stagger_ticks);
    Tile4 c = make_tile4(g, L);
    issue_dma4(c, smem, wave, voffA, voffW, sa, sw, 0);      // K tile 0 of the first output tile -> stage 0
    Bias4 bias;
    load_bias4<EPI>(bias, g, c.n0 + wn * 128 + ecc);

    while (true) {
        Acc4 acc;                                            // not cleared: the first k-step of the tile runs with C = 0
        const int Ln = L + nblk;
        const bool more = Ln < g.ntiles;
        // the next output tile (or, for the last one, this tile again: an idle 64 KB prefetch instead of a branch around VMEM)
        const Tile4 cn = make_tile4(g, more ? Ln : L);

        wait_vm0();                                          // K tile 0 landed (and the previous epilogue's stores drained)
        wait_lgkm0();
        raw_barrier();                                       // ... for every wave; the previous epilogue's slabs are released
        Frag4<T> f[2];
        read_frags<0, 16>(f[0], baseA, baseB);
        asm_wait_lgkm0();
        ktile4<T, true, SCHED>(acc, f, smem, 0, c, wave, voffA, voffW, sa, sw, baseA, baseB);            // first k-step: C = 0
        for (int u = 1; u + 1 < nt; ++u) ktile4<T, false, SCHED>(acc, f, smem, u, c, wave, voffA, voffW, sa, sw, baseA, baseB);
        ktile4_last<T, SCHED>(acc, f, smem, cn, wave, voffA, voffW, sa, sw, baseA, baseB);
        w4_drain();
        // the next tile's bias (/ colsum): lands under the epilogue
        Bias4 bias_next;
        load_bias4<EPI>(bias_next, g, cn.n0 + wn * 128 + ecc);
        raw_barrier();                                       // every wave has finished reading stage 1: the slabs may overlay it

        const int row0 = c.m0 + wm * 128, col0 = c.n0 + wn * 128;
        epilogue4<T, EPI>(acc, g, smem, wave, lane, row0, col0, bias);
        if (!more) break;
        L = Ln;
        c = cn;
        bias = bias_next;
    }
}

static int w4_sched() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PIGEON_W4_SCHED"); v = e ? atoi(e) : 0; if (v < 0 || v > 1) v = 0; }
    return v;
}
template <typename T, int EPI, int SCHED>
int launch_w4s(const GemmArgs& g, int nblk, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm_w4_kernel<T, EPI, SCHED>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
        if (e != hipSuccess) { pg_set_error("gemm_w4: set LDS attr: %s", hipGetErrorString(e)); return PG_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(256), W4_LDS, s, g);
    return pg_check_launch("gemm_w4");
}
template <typename T, int EPI>
int launch_w4(const GemmArgs& g, int nblk, hipStream_t s) {
    // the schedule experiment is built for the two residual epilogues only (out-projection / fc2: where this kernel could pay)
    if constexpr (EPI == EPI_RESID || EPI == EPI_RESID_STAT) { if (w4_sched() == 1) return launch_w4s<T, EPI, 1>(g, nblk, s); }
    return launch_w4s<T, EPI, 0>(g, nblk, s);
}

int cus4() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <typename T>
int dispatch_w4(const GemmArgs& g, int epi, int nblk, hipStream_t s) {
    switch (epi) {
        case EPI_QKV: return launch_w4<T, EPI_QKV>(g, nblk, s);
        case EPI_GELU: return launch_w4<T, EPI_GELU>(g, nblk, s);
        case EPI_RESID: return launch_w4<T, EPI_RESID>(g, nblk, s);
        case EPI_F32: return launch_w4<T, EPI_F32>(g, nblk, s);
        case EPI_RESID_STAT: return launch_w4<T, EPI_RESID_STAT>(g, nblk, s);
        case EPI_QKV_LN: return launch_w4<T, EPI_QKV_LN>(g, nblk, s);
        case EPI_GELU_LN: return launch_w4<T, EPI_GELU_LN>(g, nblk, s);
        default: pg_set_error("gemm_w4: epilogue %d is not built (the patch embedding stays on variant 36)", epi); return PG_EINVAL;
    }
}

}  // namespace

bool pg_gemm_w4_supported(int epi, int N, int K) {
    return epi != EPI_PATCH && N % W4_BN == 0 && K % (2 * BK) == 0 && K >= 2 * BK;
}

int pg_gemm_w4_launch(int dtype, GemmArgs g, int epi, hipStream_t s) {
    if (!pg_gemm_w4_supported(epi, g.N, g.K)) { pg_set_error("gemm_w4: unsupported shape / epilogue (N=%d K=%d epi=%d)", g.N, g.K, epi); return PG_EINVAL; }
    if ((int64_t)g.lda * 2 * W4_BM >= (1ll << 31) || (int64_t)g.ldw * 2 * W4_BN >= (1ll << 31)) {
        pg_set_error("gemm_w4: operand panel exceeds the 2 GB buffer-descriptor range");
        return PG_EINVAL;
    }
    g.tilesM = (g.M + W4_BM - 1) / W4_BM;
    g.tilesN = g.N / W4_BN;
    g.ntiles = g.tilesM * g.tilesN;
    g.gn = (g.tilesN % 4 == 0) ? 4 : g.tilesN;
    const int nblk = g.ntiles < cus4() ? g.ntiles : cus4();
    if (dtype == PG_DTYPE_F16) return dispatch_w4<T_F16>(g, epi, nblk, s);
    if (dtype == PG_DTYPE_BF16) return dispatch_w4<T_BF16>(g, epi, nblk, s);
    pg_set_error("gemm_w4: operand dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16 (got %d)", dtype);
    return PG_EINVAL;
}
