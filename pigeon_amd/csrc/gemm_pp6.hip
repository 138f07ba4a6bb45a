// gemm_pp6.hip -- the persistent ping-pong GEMM of gemm_pp.hip with a 384 x 256 block tile (EXPERIMENTAL, variant 56).
//
// Why: every kernel of the path runs at the 1400 W package power cap and the operand DMA (global -> LDS) is the largest
// energy item of the GEMMs (no DMA -> 1.45x; 7 of 8 DMAs -> -3...-9 % time).  The bytes DMA'd per
// output are (BM + BN) / (BM * BN): 1/128 for 256 x 256, 1/153.6 for 384 x 256 -- the weight panel is fetched once per 384
// instead of once per 256 rows (-33 % weight bytes, activation bytes unchanged).  The register file bounds the tile: 8 waves
// x 64 lanes x 192 accumulator registers = 98 304 outputs is all that fits next to the fragments.
//
// Same structure as gemm_pp.hip (read its header first): one 512-thread block per CU walking tiles L0, L0+G, ...; waves
// 2 (M) x 4 (N), wave tile 192 x 64 = 6 x 2 MFMA blocks of 32 x 32; two wave groups one barrier apart, a k-step is
// {LOAD: 8 ds_read_b128 + share of the next K tile's DMA; barrier; MFMA: 12 MFMAs; barrier}; K tile 64, two 80 KB stages
// = the whole 160 KB of LDS (A rows [0,384) then W rows [384,640), 128-byte rows, chunk c of row r at c ^ ((r>>1)&7));
// the epilogue's per-wave transpose slabs (8 x 8.5 KB) overlay stage 1, which is free once every wave has left the
// mainloop (K/64 even).  A wave issues 6 + 4 DMAs per K tile; the lane's offset inside a 64-row DMA group does not depend
// on the group (the swizzle term only sees (row>>1)&7), so ONE VGPR offset per operand is kept and the group / K advance
// is added per DMA -- there are no registers to spare: 192 accumulators + 32 fragment registers + bias.
// The 16-bit-output epilogues (EPI_QKV, EPI_GELU and their LayerNorm-fold forms: QKV and fc1 are where the weight bytes are, N = 3072 /
// 4096, K = 1024) and, since round 3, the fp32 residual epilogue with row statistics (EPI_RESID_STAT: out-projection and fc2).  Same MFMA order over K as every other GEMM kernel of the library: results are bit-identical.
#include "gemm_epi.h"

namespace {

constexpr int P6_TM = 6;                                   // 32-row MFMA blocks per wave along M
constexpr int P6_BM = 2 * P6_TM * 32, P6_BN = 256;         // 384 x 256
constexpr int P6_STAGE = (P6_BM + P6_BN) * ROWB;           // 80 KB
constexpr int P6_W_OFF = P6_BM * ROWB;
constexpr int P6_SLAB_OFF = P6_STAGE;                      // slabs overlay stage 1
constexpr int P6_SLAB_ROWF = 64 + 4;
constexpr int P6_SLAB_BYTES = 32 * P6_SLAB_ROWF * 4;       // 8704 B per wave, 8 waves = 68 KB <= 80 KB
constexpr int P6_LDS = 2 * P6_STAGE;                       // 160 KB
constexpr int P6_NDMA = P6_TM + 4;                         // DMAs per wave per K tile (6 A + 4 W)
static_assert(8 * P6_SLAB_BYTES <= P6_STAGE, "slabs must fit into stage 1");

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

struct Tile6 {
    __amdgpu_buffer_rsrc_t ra, rw;
    int m0, n0;
    int sa, sw;                                              // bytes between two 64-row DMA groups of A / W
};

// same band / super-tile order as gemm_pp.hip make_tile (8 x 4 super-tiles per XCD round), in 384-row panels
__device__ __forceinline__ Tile6 make_tile6(const GemmArgs& g, int L) {
    Tile6 c;
    const int gmax = g.gn >= 32 ? 1 : 32 / g.gn;
    const int band_sz = gmax * g.tilesN;
    const int band = L / band_sz, rem = L - band * band_sz;
    const int gm = min(gmax, g.tilesM - band * gmax);
    const int sup = rem / (gm * g.gn), rem2 = rem - sup * gm * g.gn;
    const int im = rem2 / g.gn, in = rem2 - im * g.gn;
    const int tm = band * gmax + im, tn = sup * g.gn + in;
    c.m0 = tm * P6_BM; c.n0 = tn * P6_BN;
    const int rows = min(P6_BM, g.M - c.m0);
    c.ra = make_rsrc(g.A + (int64_t)c.m0 * g.lda, (uint32_t)rows * (uint32_t)g.lda * 2u);
    c.rw = make_rsrc(g.W + (int64_t)c.n0 * g.ldw, (uint32_t)P6_BN * (uint32_t)g.ldw * 2u);
    c.sa = 64 * (int)g.lda * 2; c.sw = 64 * (int)g.ldw * 2;
    return c;
}

// DMA d of a K tile: d 0..5 = this wave's six 8-row groups of A (rows (wave + 8 d) * 8 ..), d 6..9 = its four of W
template <int FROM, int CNT>
__device__ __forceinline__ void issue_dma6(const Tile6& c, char* stage, int wave, int voffA, int voffW, int soff) {
    // opaque: with a constant K offset (the peeled first K-tile pair, the next-tile prefetch) the ten per-lane offsets are
    // invariant across the persistent tile loop and hipcc hoists -- and then spills -- thirty of them
    asm volatile("" : "+s"(soff));
#pragma unroll
    for (int d = 0; d < P6_NDMA; ++d) {
        if (d < FROM || d >= FROM + CNT) continue;
        // group advance and K advance are added into the VGPR offset (one v_add with an SGPR operand per DMA): the SGPR
        // soffset is NOT part of the descriptor's bounds check, and the M tail relies on that check (rows past M read 0)
        if (d < P6_TM) dma16(c.ra, stage + (wave + 8 * d) * 8 * ROWB, voffA + (soff + d * c.sa), 0);
        else dma16(c.rw, stage + P6_W_OFF + (wave + 8 * (d - P6_TM)) * 8 * ROWB, voffW + (soff + (d - P6_TM) * c.sw), 0);
    }
}

// MFMA shape: v_mfma_f32_16x16x32 (round 2).  A k-step of 32 is computed in TWO halves of the wave's 192 rows, so that the
// phase structure of the 32x32x16 version is kept as it was -- four {LOAD; barrier; MFMA; barrier} phases per 64-wide K tile,
// 384 matrix-pipe cycles each -- with 8 more fragment registers: phase (s, half) reads the six 16-row A blocks of rows
// [96 half, 96 half + 96) at k-step s (and, in the first half, the four 16-column W blocks, kept for the second half) and
// runs 24 MFMAs into accumulator blocks [6 half, 6 half + 6) x [0, 4).
constexpr int P6_RB = 2 * P6_TM;                           // 16-row accumulator blocks per wave along M (12)
typedef f32x4 Acc6[P6_RB][4];
template <typename T> struct Frag6 { typename T::v8 a[P6_TM], b[4]; };

// Fragment reads as inline asm from ONE address register per operand: written as C++ loads, hipcc precomputes and keeps
// live an address VGPR per (stage, k-step, operand) -- 16 registers this kernel does not have (it spilled 73).  The
// k-step's chunk is an XOR on the address (disjoint bits, see ktile6), the 16-row blocks are immediates (<= 10 KB).
template <int OFF, typename V>
__device__ __forceinline__ void lds_read_b128(V& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <typename T, bool WITH_B>
__device__ __forceinline__ void load_frag6(Frag6<T>& f, uint32_t aA, uint32_t aB) {
    lds_read_b128<0 * 16 * ROWB>(f.a[0], aA); lds_read_b128<1 * 16 * ROWB>(f.a[1], aA); lds_read_b128<2 * 16 * ROWB>(f.a[2], aA);
    lds_read_b128<3 * 16 * ROWB>(f.a[3], aA); lds_read_b128<4 * 16 * ROWB>(f.a[4], aA);
    if constexpr (P6_TM > 5) lds_read_b128<5 * 16 * ROWB>(f.a[P6_TM > 5 ? 5 : 0], aA);
    if constexpr (WITH_B) {
        lds_read_b128<0 * 16 * ROWB>(f.b[0], aB); lds_read_b128<1 * 16 * ROWB>(f.b[1], aB);
        lds_read_b128<2 * 16 * ROWB>(f.b[2], aB); lds_read_b128<3 * 16 * ROWB>(f.b[3], aB);
    }
}

// 24 MFMAs of one phase; swapped operands (weights first): a lane owns output row lane & 15 of a 16 x 16 block and the 4
// consecutive columns 4 (lane >> 4) ..
// P6_EARLY (round 4): the barrier that ends an MFMA phase is ARRIVED AT before the phase's last P6_EARLY MFMAs are issued.  The
// phase stamps of the tools build (tools/stall_probe.py, profiles/r04/gemm_pp6_phase_stamps.txt) show every wave's operands in
// place when it asks for them (12-15 ns per K tile in vmcnt(0)) and LOAD + barrier = 145 ns against 227 ns of MFMAs -- yet a
// phase takes ~630 ns where two alternating MFMA sections would take 454: the matrix pipe idles across each of the 8 barriers of
// a K tile, between the last MFMA issue of one wave group and the first of the other (the s_barrier round trip).  MFMAs touch
// registers only -- the barrier orders LDS reads against the DMAs into the other stage, and schedules the two groups -- so the
// trailing MFMAs may follow the barrier: the partner group is released while they still feed the pipe.  Same MFMA order per
// accumulator: bit-identical.
#ifndef PG_P6_EARLY
#define PG_P6_EARLY 0
#endif
constexpr int P6_EARLY = PG_P6_EARLY;
template <typename T, bool ZERO, int HALF>
__device__ __forceinline__ void mma24(Acc6& acc, const Frag6<T>& f) {
#pragma unroll
    for (int i = 0; i < P6_TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (P6_EARLY > 0 && i * 4 + j == 4 * P6_TM - P6_EARLY) raw_barrier();
            if constexpr (ZERO) T::mfma16_init(acc[HALF * P6_TM + i][j], f.b[j], f.a[i]);
            else T::mfma16_acc(acc[HALF * P6_TM + i][j], f.b[j], f.a[i]);
        }
}

// One K tile in ping-pong form; the wave's 10 DMAs of the next K tile go 5 / 5 in the first two LOAD phases.
// baseA / baseB: LDS byte addresses of this lane's fragment row (lane & 15 of block 0) in STAGE 0 at k-step 0 (chunk
// ((lane >> 4) ^ sw) << 4); k-step s reads chunk (4 s + (lane >> 4)) ^ sw = chunk0 ^ (4 s), i.e. address ^ (s << 6) (disjoint
// bits).  The per-phase addresses are formed by asm (one v_add / v_xor each) so that they are NOT hoisted into live registers.
#ifdef PIGEON_ABLATIONS
// tools build: wall-clock ticks (100 MHz) block 0's waves 0 (leader group) and 4 (follower group) spend per K tile (a) in the
// `s_waitcnt vmcnt(0)` that waits for the NEXT K tile's operand DMAs and (b) in the barrier right behind it; [grp][0] = (a),
// [grp][1] = (b), [grp][2] = K tiles counted.  Read / reset with pg_dbg_stall_read (tools/stall_probe.py).
__device__ unsigned long long pg_dbg_stall[2][4];
// ... and, per phase of the ping-pong schedule (4 per K tile), where those two waves' time goes: [grp][0] LOAD (fragment ds_reads +
// DMA issue, up to the lgkmcnt wait), [1] the barrier in front of the MFMAs, [2] the 24 MFMAs (issue of the first to issue of the
// last + the setprio pair), [3] the barrier behind them, [4] phases counted.
__device__ unsigned long long pg_dbg_phase[2][8];
__device__ unsigned long long pg_dbg_vm[8][4];               // per wave of block 0: ticks in vmcnt(0), waits, waits > 200 ns
extern "C" int pg_dbg_vm_read(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(pg_dbg_vm), sizeof(unsigned long long) * 32) != hipSuccess) return PG_EHIP;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pg_dbg_vm), z, sizeof(z)) != hipSuccess) return PG_EHIP; }
    return PG_OK;
}
extern "C" int pg_dbg_stall_read(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pg_dbg_stall), sizeof(unsigned long long) * 8) != hipSuccess) return PG_EHIP;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pg_dbg_stall), z, sizeof(z)) != hipSuccess) return PG_EHIP; }
    return PG_OK;
}
extern "C" int pg_dbg_phase_read(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pg_dbg_phase), sizeof(unsigned long long) * 16) != hipSuccess) return PG_EHIP;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pg_dbg_phase), z, sizeof(z)) != hipSuccess) return PG_EHIP; }
    return PG_OK;
}
#endif

template <typename T, bool ZERO, int STAGE>
__device__ __forceinline__ void ktile6(Acc6& acc, char* smem, uint32_t baseA, uint32_t baseB, const Tile6& c,
                                       int wave, int voffA, int voffW, int soff_next, bool has_next) {
    char* nxt = smem + (STAGE ? 0 : P6_STAGE);
    Frag6<T> f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        // phase kk = (k-step s = kk >> 1, half = kk & 1)
#ifdef PIGEON_ABLATIONS
        const bool probe = blockIdx.x == 0 && (wave & 3) == 0;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (probe) t0 = __builtin_amdgcn_s_memrealtime();
#endif
        uint32_t aA, aB;
        asm volatile("v_add_u32 %0, %2, %1\n\tv_xor_b32 %0, %3, %0" : "=&v"(aA) : "v"(baseA),
                     "n"(STAGE * P6_STAGE + (kk & 1) * P6_TM * 16 * ROWB), "n"((kk >> 1) << 6));
        asm volatile("v_add_u32 %0, %2, %1\n\tv_xor_b32 %0, %3, %0" : "=&v"(aB) : "v"(baseB), "n"(STAGE * P6_STAGE), "n"((kk >> 1) << 6));
        if ((kk & 1) == 0) load_frag6<T, true>(f, aA, aB);
        else load_frag6<T, false>(f, aA, aB);
        if (has_next) {
            if (kk == 0) issue_dma6<0, 5>(c, nxt, wave, voffA, voffW, soff_next);
            if (kk == 1) issue_dma6<5, 5>(c, nxt, wave, voffA, voffW, soff_next);
#ifdef PIGEON_ABLATIONS
            if (kk == 3) {                                   // every wave of block 0: how long for ITS next-tile DMAs, and how often > 200 ns
                unsigned long long ta = 0;
                if (blockIdx.x == 0) ta = __builtin_amdgcn_s_memrealtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
                    const unsigned long long dtk = __builtin_amdgcn_s_memrealtime() - ta;
                    pg_dbg_vm[wave][0] += dtk; pg_dbg_vm[wave][1] += 1; if (dtk > 20) pg_dbg_vm[wave][2] += 1;
                }
            }
#else
            if (kk == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
        wait_lgkm0();
#ifdef PIGEON_ABLATIONS
        if (probe) t1 = __builtin_amdgcn_s_memrealtime();
#endif
        raw_barrier();
#ifdef PIGEON_ABLATIONS
        if (probe) t2 = __builtin_amdgcn_s_memrealtime();
#endif
        __builtin_amdgcn_s_setprio(1);
        if (kk == 0) mma24<T, ZERO, 0>(acc, f);
        if (kk == 1) mma24<T, ZERO, 1>(acc, f);
        if (kk == 2) mma24<T, false, 0>(acc, f);
        if (kk == 3) mma24<T, false, 1>(acc, f);
        __builtin_amdgcn_s_setprio(0);
#ifdef PIGEON_ABLATIONS
        if (probe) t3 = __builtin_amdgcn_s_memrealtime();
#endif
        if (P6_EARLY == 0) raw_barrier();
#ifdef PIGEON_ABLATIONS
        if (probe && (threadIdx.x & 63) == 0) {
            const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* d = pg_dbg_phase[wave >> 2];
            d[0] += t1 - t0; d[1] += t2 - t1; d[2] += t3 - t2; d[3] += t4 - t3; d[4] += 1;
        }
#endif
    }
}

struct Bias6 { f32x4 lo, hi; };

// HOFF: column distance between the lane's two f32x4 (4: eight consecutive columns; 32: the split halves of EPI_RESID_STAT)
template <int HOFF = 4>
__device__ __forceinline__ void load_bias6(Bias6& b, const GemmArgs& g, int col) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    b.lo = z; b.hi = z;
    if (g.bias) {
        b.lo = *(const f32x4*)(g.bias + col);
        b.hi = *(const f32x4*)(g.bias + col + HOFF);
    }
}
__device__ __forceinline__ void pin_bias6(Bias6& b) { asm volatile("" : "+v"(b.lo), "+v"(b.hi)); }

template <int EPI> constexpr bool ln6() { return EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool qkv6() { return EPI == EPI_QKV || EPI == EPI_QKV_LN; }

// LayerNorm-fold epilogues: (rstd, mean*rstd) of the lane's rows of one 32-row slab (rows rr + 8 it), eight registers;
// the slab's pair is fetched while the previous slab is processed (there is no room for a whole tile's worth: 48 registers)
struct RowStat6 { u32x2 v[4]; };
__device__ __forceinline__ void load_rowstat6(RowStat6& rs, __amdgpu_buffer_rsrc_t rrs, int rr, int slab) {
#pragma unroll
    for (int it = 0; it < 4; ++it) rs.v[it] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (rr + slab * 32 + it * 8) * 8, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rowstat_rsrc6(const GemmArgs& g, int row0) {
    int rvs = g.M - row0; rvs = rvs < 0 ? 0 : (rvs > P6_TM * 32 ? P6_TM * 32 : rvs);
    // wave-uniform, but hipcc clamps with v_med3_i32 (there is no scalar med3): the record count, and with it the whole
    // descriptor, would sit in VGPRs and every statistics load of the epilogue would be wrapped in a waterfall loop
    rvs = __builtin_amdgcn_readfirstlane(rvs);
    return make_rsrc((const char*)g.ex.rowstat + (int64_t)row0 * 8, (uint32_t)rvs * 8u);
}

// Round 2's epilogue of the 16-bit outputs (tools build: PIGEON_GEMM_PARK16=0; the product form is epilogue6_park16 below, same
// bits, -1 % per launch): per wave six 32-row x 64-column fp32 slabs transposed through LDS (gemm_pp.hip
// pp_epilogue, WIDE geometry: 8 lanes per row, 8 rows per store instruction, 4 instructions per slab).
// LN: colsum (cs) and the row statistics of slab 0 (rs0) were fetched right after the last MFMA phase.
template <typename T, int EPI, typename PREFETCH_DMA, typename PREFETCH_BIAS>
__device__ __forceinline__ void epilogue6(Acc6& acc, const GemmArgs& g, char* smem, int wave, int lane, int row0,
                                          int col0, const Bias6& bias, const Bias6& cs, const RowStat6& rs0,
                                          __amdgpu_buffer_rsrc_t rrs, PREFETCH_DMA&& prefetch_dma, PREFETCH_BIAS&& prefetch_bias,
                                          int dbg_iter = 0) {
    constexpr int ROWPF = P6_SLAB_ROWF;
    constexpr bool LN = ln6<EPI>();
    const int l15 = lane & 15, lq = lane >> 4;               // MFMA side: row inside a 16-row block, column quad
    float* slab = (float*)(smem + P6_SLAB_OFF + wave * P6_SLAB_BYTES);
    const int rr = lane >> 3, cc = (lane & 7) * 8;
    const int col = col0 + cc;
    const float qsc = (qkv6<EPI>() && col < g.qcols) ? g.qscale : 1.f;
    int rv = g.M - row0; rv = rv < 0 ? 0 : (rv > P6_TM * 32 ? P6_TM * 32 : rv);
    rv = __builtin_amdgcn_readfirstlane(rv);   // descriptor stays in SGPRs (hipcc clamps with v_med3_i32, see gemm_pp6.hip rowstat_rsrc6)
    const uint32_t nbytes = rv > 0 ? (uint32_t)(((int64_t)(rv - 1) * g.ldc + 64) * 2) : 0u;
    __amdgpu_buffer_rsrc_t ro = make_rsrc((const char*)g.out + ((int64_t)row0 * g.ldc + col0) * 2, nbytes);
    const int voff = (rr * (int)g.ldc + cc) * 2;
    int rstep = 8 * (int)g.ldc * 2;                          // bytes between two store iterations
    int sstep = 32 * (int)g.ldc * 2;                         // bytes between two slabs
    asm volatile("" : "+s"(rstep), "+s"(sstep));             // not hoisted into 24 SGPRs across the K loop
    prefetch_dma();
    RowStat6 rs[2];
    rs[0] = rs0;
#pragma unroll
    for (int i = 0; i < P6_TM; ++i) {
        if constexpr (LN) { if (i + 1 < P6_TM) load_rowstat6(rs[(i + 1) & 1], rrs, rr, i + 1); }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (ib * 16 + l15) * ROWPF + j * 16 + 4 * lq) = acc[2 * i + ib][j];
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + rr;
            f32x4 lo = *(const f32x4*)(slab + r * ROWPF + cc);
            f32x4 hi = *(const f32x4*)(slab + r * ROWPF + cc + 4);
            // row offset in the VGPR offset, not the SGPR soffset (hipcc pads no wait states after a >64-bit buffer store
            // with a register soffset: gemm_pp.hip)
            const int ooff = voff + (i * sstep + it * rstep);
            // LN: each scalar through an asm move of its own (hipcc SLP-packs the fmas into v_pk_fma_f32 and drops the op_sel of the
            // high half of the loaded pair: gemm_pp.hip); the arithmetic itself is gemm_epi.h epi16_finish (packed fp32 forms)
            float rstd = 0.f, mrs = 0.f;
            if constexpr (LN) {
                asm("v_mov_b32 %0, %1" : "=v"(rstd) : "v"(rs[i & 1].v[it][0]));
                asm("v_mov_b32 %0, %1" : "=v"(mrs) : "v"(rs[i & 1].v[it][1]));
            }
            const u32x4 pk = epi16_finish<T, EPI>(lo, hi, bias.lo, bias.hi, cs.lo, cs.hi, rstd, mrs, col0 < g.qcols, qsc);
            __builtin_amdgcn_raw_buffer_store_b128(pk, ro, ooff, 0, 0);
        }
        wave_lds_fence();                                    // slab reads retired before the next slab overwrites it
        PG_TS(g, dbg_iter, wave, 3 + i);
        if (i == 0) prefetch_bias();                         // next tile's bias (/ nothing else): 32 registers are free now
    }
}

// ---- 16-bit outputs: finish in the accumulator layout, transpose the 16-bit values (product epilogue since the end of round 3) --
// epilogue6 parks fp32 accumulators and finishes them after the read-back: 768 KB of LDS traffic per tile.  Here the arithmetic
// (the same gemm_epi.h epi16_finish on the same values: outputs are bit-identical) runs on the accumulators where they are -- a
// lane owns row rb * 16 + (lane & 15) and the 4 consecutive columns 16 j + 4 (lane >> 4) .. of each 16 x 16 block -- and only the
// packed 16-bit results go through the slab: 384 KB per tile, and the read-back feeds the stores directly.  The stores keep the
// slab form's geometry (8 rows x 128 bytes per instruction: full lines; tools/store_bw.hip: partial lines cost per request).
// Needs the row statistics of the lane's two rows per slab and the column constants of all four column blocks: blocks 0, 1 come
// as the tile's Bias6 (HOFF = 16), blocks 2, 3 and the colsum vectors are fetched right after the last MFMA phase.
constexpr int P6_P16_ROWB = 128 + 16;                      // bytes per 64-column 16-bit slab row (pad: the 16 rows of a b64 write hit 16 bank pairs)
static_assert(32 * P6_P16_ROWB <= P6_SLAB_BYTES, "16-bit slab must fit into the wave's slab");
struct RowStatP { u32x2 v[2]; };
__device__ __forceinline__ void load_rowstat_p(RowStatP& rs, __amdgpu_buffer_rsrc_t rrs, int l15, int slab) {
    rs.v[0] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (slab * 32 + l15) * 8, 0, 0);
    rs.v[1] = __builtin_amdgcn_raw_buffer_load_b64(rrs, (slab * 32 + 16 + l15) * 8, 0, 0);
}
template <typename T, int EPI, typename PREFETCH_DMA, typename PREFETCH_BIAS>
__device__ __forceinline__ void epilogue6_park16(Acc6& acc, const GemmArgs& g, char* smem, int wave, int lane, int row0, int col0,
                                                 const Bias6& b01, const Bias6& b23, const Bias6& s01, const Bias6& s23,
                                                 const RowStatP& rs0, __amdgpu_buffer_rsrc_t rrs,
                                                 PREFETCH_DMA&& prefetch_dma, PREFETCH_BIAS&& prefetch_bias, int dbg_iter = 0) {
    constexpr bool LN = ln6<EPI>();
    const int l15 = lane & 15, lq = lane >> 4;
    char* slab = smem + P6_SLAB_OFF + wave * P6_SLAB_BYTES;
    const int rr = lane >> 3, c8 = lane & 7;
    const bool q_strip = col0 < g.qcols;                     // wave-uniform: qcols is a multiple of 64
    const float qsc = (qkv6<EPI>() && q_strip) ? g.qscale : 1.f;
    int rv = g.M - row0; rv = rv < 0 ? 0 : (rv > P6_TM * 32 ? P6_TM * 32 : rv);
    rv = __builtin_amdgcn_readfirstlane(rv);
    const uint32_t nbytes = rv > 0 ? (uint32_t)(((int64_t)(rv - 1) * g.ldc + 64) * 2) : 0u;
    __amdgpu_buffer_rsrc_t ro = make_rsrc((const char*)g.out + ((int64_t)row0 * g.ldc + col0) * 2, nbytes);
    const int voff = (rr * (int)g.ldc + c8 * 8) * 2;
    int rstep = 8 * (int)g.ldc * 2;
    int sstep = 32 * (int)g.ldc * 2;
    asm volatile("" : "+s"(rstep), "+s"(sstep));
    prefetch_dma();
    RowStatP rs[2];
    rs[0] = rs0;
    char* wr = slab + l15 * P6_P16_ROWB + 8 * lq;            // this lane's 8 bytes of column block 0 in row block 0 of the slab
#pragma unroll
    for (int i = 0; i < P6_TM; ++i) {
        if constexpr (LN) { if (i + 1 < P6_TM) load_rowstat_p(rs[(i + 1) & 1], rrs, l15, i + 1); }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            float rstd = 0.f, mrs = 0.f;
            if constexpr (LN) {                              // each half through a move of its own (see epilogue6)
                asm("v_mov_b32 %0, %1" : "=v"(rstd) : "v"(rs[i & 1].v[ib][0]));
                asm("v_mov_b32 %0, %1" : "=v"(mrs) : "v"(rs[i & 1].v[ib][1]));
            }
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                const Bias6& bb = jp ? b23 : b01;
                const Bias6& ss = jp ? s23 : s01;
                const u32x4 pk = epi16_finish<T, EPI>(acc[2 * i + ib][2 * jp], acc[2 * i + ib][2 * jp + 1], bb.lo, bb.hi, ss.lo, ss.hi,
                                                      rstd, mrs, q_strip, qsc);
                u32x2 h0, h1;
                h0[0] = pk[0]; h0[1] = pk[1]; h1[0] = pk[2]; h1[1] = pk[3];
                *(u32x2*)(wr + ib * 16 * P6_P16_ROWB + (2 * jp) * 32) = h0;
                *(u32x2*)(wr + ib * 16 * P6_P16_ROWB + (2 * jp + 1) * 32) = h1;
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const u32x4 v = *(const u32x4*)(slab + (it * 8 + rr) * P6_P16_ROWB + c8 * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, ro, voff + (i * sstep + it * rstep), 0, 0);
        }
        wave_lds_fence();                                    // slab reads retired before the next slab overwrites it
        PG_TS(g, dbg_iter, wave, 3 + i);
        if (i == 0) prefetch_bias();
    }
}

// ---- EPI_RESID_STAT on 384 x 256 tiles (round 3: out-projection and fc2) ------------------------------------------------------
// X += acc + b (fp32, in place), the 16-bit copy of the new rows and the per-64-column (sum, sum of squares) partials -- the
// arithmetic is gemm_epi.h's (epi_resid4 / epi_copy16x4 / epi_stat8 / row8_sum), the geometry gemm_pp.hip's split halves (a lane
// holds columns 4k.. and 32 + 4k.. of its row, k = lane & 7): results are bit-identical to the 256 x 256 kernel and the tail kernel.
// Registers are the problem here (192 accumulators): the residual rows of a slab are 32 registers per lane.  Slab 0's rows are
// fetched right after the last MFMA phase, into the dead fragment registers; slab i + 1's rows are fetched AFTER slab i's
// accumulators have been parked in LDS, into the registers that frees, and land while slab i is processed (one slab ahead, as
// in gemm_pp.hip, but with nothing spare).
struct XRows6 { u32x4 v[4][2]; };
template <int IT0 = 0, int IT1 = 4>
__device__ __forceinline__ void fetch_xrows6(XRows6& x, __amdgpu_buffer_rsrc_t ro, int voff, int slab_off, int rstep) {
#pragma unroll
    for (int it = IT0; it < IT1; ++it)
#pragma unroll
        for (int h = 0; h < 2; ++h)       // (readfirstlane: keeps the row offset an SGPR soffset -- no waterfall loop; loads need no
            x.v[it][h] =                  //  bounds check on it: rows past M are read and dropped, the STORES carry it in the VGPR offset)
                __builtin_amdgcn_raw_buffer_load_b128(ro, voff + 128 * h, __builtin_amdgcn_readfirstlane(slab_off + it * rstep), 0);
}
constexpr int P6_X0_EARLY = 2;                              // store iterations of slab 0 whose residual rows are fetched before the epilogue
struct ResidCtx6 {
    __amdgpu_buffer_rsrc_t ro, rx16, rstat;
    int voff;
};
__device__ __forceinline__ ResidCtx6 make_resid6(const GemmArgs& g, int row0, int col0, int lane) {
    ResidCtx6 r;
    int rv = g.M - row0; rv = rv < 0 ? 0 : (rv > P6_TM * 32 ? P6_TM * 32 : rv);
    rv = __builtin_amdgcn_readfirstlane(rv);                 // descriptors stay in SGPRs (rowstat_rsrc6)
    const uint32_t nel = rv > 0 ? (uint32_t)((int64_t)(rv - 1) * g.ldc + 64) : 0u;
    r.ro = make_rsrc((const char*)g.out + ((int64_t)row0 * g.ldc + col0) * 4, nel * 4u);
    r.rx16 = make_rsrc((const char*)g.ex.x16 + ((int64_t)row0 * g.ldc + col0) * 2, nel * 2u);
    r.rstat = make_rsrc((const char*)(g.ex.statpart + ((int64_t)(col0 / 64) * g.ex.stat_rows + row0) * 2), (uint32_t)rv * 8u);
    r.voff = ((lane >> 3) * (int)g.ldc + (lane & 7) * 4) * 4;
    return r;
}
template <typename T, typename PREFETCH_DMA, typename PREFETCH_BIAS>
__device__ __forceinline__ void epilogue6_resid(Acc6& acc, const GemmArgs& g, char* smem, int wave, int lane, int row0, int col0,
                                                const Bias6& bias, const XRows6& x0, const ResidCtx6& rc,
                                                PREFETCH_DMA&& prefetch_dma, PREFETCH_BIAS&& prefetch_bias, int dbg_iter = 0) {
    constexpr int ROWPF = P6_SLAB_ROWF;
    const int l15 = lane & 15, lq = lane >> 4;
    float* slab = (float*)(smem + P6_SLAB_OFF + wave * P6_SLAB_BYTES);
    const int rr = lane >> 3, cc = (lane & 7) * 4;
    int rstep = 8 * (int)g.ldc * 4;                          // bytes between two store iterations
    int sstep = 32 * (int)g.ldc * 4;                         // bytes between two slabs
    asm volatile("" : "+s"(rstep), "+s"(sstep));             // not hoisted into SGPRs across the K loop
    prefetch_dma();
    XRows6 xr[2];
#pragma unroll
    for (int it = 0; it < P6_X0_EARLY; ++it) { xr[0].v[it][0] = x0.v[it][0]; xr[0].v[it][1] = x0.v[it][1]; }
    auto process = [&](int i, int it) {
        const int r = it * 8 + rr;
        const f32x4 lo = *(const f32x4*)(slab + r * ROWPF + cc);
        const f32x4 hi = *(const f32x4*)(slab + r * ROWPF + cc + 32);
        // row offset in the VGPR offset, not the SGPR soffset (bounds check of the M tail; hazard of >64-bit stores: gemm_pp.hip)
        const int ooff = rc.voff + (i * sstep + it * rstep);
        const f32x4 x = epi_resid4(__builtin_bit_cast(f32x4, xr[i & 1].v[it][0]), lo, bias.lo);
        const f32x4 y = epi_resid4(__builtin_bit_cast(f32x4, xr[i & 1].v[it][1]), hi, bias.hi);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), rc.ro, ooff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rc.ro, ooff + 128, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(epi_copy16x4<T>(x), rc.rx16, ooff >> 1, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(epi_copy16x4<T>(y), rc.rx16, (ooff >> 1) + 64, 0, 0);
        float s1, s2;
        epi_stat8(x, y, s1, s2);
        s1 = row8_sum(s1);
        s2 = row8_sum(s2);
        if ((lane & 7) == 0) { slab[r * ROWPF + 64] = s1; slab[r * ROWPF + 65] = s2; }
    };
#pragma unroll
    for (int i = 0; i < P6_TM; ++i) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (ib * 16 + l15) * ROWPF + j * 16 + 4 * lq) = acc[2 * i + ib][j];
        // Slab i's accumulators are parked: 32 registers are free.  The residual rows arrive in HALVES (16 registers each):
        // the first two store iterations of slab i + 1 now, its last two once the first two of slab i are done with theirs --
        // 48 residual registers live at the peak instead of 64 (next to 160 accumulators that is the difference between no
        // spill and a spilled piece, which costs a vmcnt(0) in the middle of the epilogue).  Slab 0: only its first two
        // iterations could be fetched before the epilogue, its last two come with this first batch.
        if (i == 0) fetch_xrows6<P6_X0_EARLY, 4>(xr[0], rc.ro, rc.voff, 0, rstep);
        if (i + 1 < P6_TM) fetch_xrows6<0, 2>(xr[(i + 1) & 1], rc.ro, rc.voff, (i + 1) * sstep, rstep);
        wave_lds_fence();
        process(i, 0);
        process(i, 1);
        if (i + 1 < P6_TM) fetch_xrows6<2, 4>(xr[(i + 1) & 1], rc.ro, rc.voff, (i + 1) * sstep, rstep);
        process(i, 2);
        process(i, 3);
        wave_lds_fence();                                    // slab reads retired, the row partials parked
        // one coalesced 256-byte store of the slab's 32 row partials, through a descriptor that ends at the tile's last valid row
        // (the slab offset rides in the instruction's immediate, which the bounds check covers): written as flat stores, hipcc keeps
        // six (32 i + lane) indices and their byte offsets live across the whole K loop and spills them (first build: 88 bytes)
        if (lane < 32)
            __builtin_amdgcn_raw_buffer_store_b64(*(const u32x2*)(slab + lane * ROWPF + 64), rc.rstat, lane * 8 + i * 256, 0, 0);
        wave_lds_fence();
        PG_TS(g, dbg_iter, wave, 3 + i);
        if (i == 0) prefetch_bias();
    }
}

template <typename T, int EPI, bool PARK>
__global__ __launch_bounds__(512) void gemm_pp6_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool follower = (wm == 1);

    // per-lane DMA offset inside a 64-row group (bytes): row (wave * 8 + lane / 8) of the group, swizzled chunk
    const int r0 = wave * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((r0 >> 1) & 7);
    const int voffA = r0 * (int)g.lda * 2 + ch * 16;
    const int voffW = r0 * (int)g.ldw * 2 + ch * 16;
    const int l15 = lane & 15;
    const int xo0 = ((lane >> 4) ^ ((lane >> 1) & 7)) << 4;  // k-step 0's chunk (lane >> 4) of the lane's fragment row, swizzled
    const uint32_t smem0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t baseA = smem0 + (wm * (P6_TM * 32) + l15) * ROWB + xo0;
    const uint32_t baseB = smem0 + P6_W_OFF + (wn * 64 + l15) * ROWB + xo0;
    constexpr bool RSTAT = (EPI == EPI_RESID_STAT);
    constexpr bool PARK16 = PARK && !RSTAT;                  // 16-bit outputs finished in the accumulator layout (epilogue6_park16)
    constexpr int BHOFF = RSTAT ? 32 : (PARK16 ? 16 : 4);
    const int ecc = RSTAT ? (lane & 7) * 4 : (PARK16 ? (lane >> 4) * 4 : (lane & 7) * 8);

    const int nt = g.K / BK;                                 // even, >= 4 (checked on the host)
    const int nblk = gridDim.x;
    int L = xcd_remap(blockIdx.x, nblk);
    if (L >= g.ntiles) return;
    xcd_stagger_wait(g.xcd_stagger_ticks);
    Tile6 c = make_tile6(g, L);
    issue_dma6<0, P6_NDMA>(c, smem, wave, voffA, voffW, 0);  // K tile 0 of the first output tile -> stage 0
    Bias6 bias;
    load_bias6<BHOFF>(bias, g, c.n0 + wn * 64 + ecc);
    pin_bias6(bias);
    // at the top of the next tile vmcnt(NST) must mean "the prefetched K tile 0 has landed": the DMAs are the oldest
    // operations of an epilogue, at least 5 x 4 stores (+ the bias / row-statistics loads) are younger than all of them
    constexpr int NST = (P6_TM - 1) * 4;
    bool first = true;
    int dbg_iter = 0;                                        // tile counter of the tools build's time stamps (dead code otherwise)

    while (true) {
        Acc6 acc;                                            // not cleared: the first k-step of the tile runs with C = 0
        if (first) {
            wait_vm0();
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_lgkm0();
        raw_barrier();                                       // B_0: K tile 0 visible, previous epilogue's slabs released
        PG_TS(g, dbg_iter, wave, 0);
        if (follower) raw_barrier();
        ktile6<T, true, 0>(acc, smem, baseA, baseB, c, wave, voffA, voffW, ROWB, true);
        ktile6<T, false, 1>(acc, smem, baseA, baseB, c, wave, voffA, voffW, 2 * ROWB, true);
        for (int t = 2; t < nt; t += 2) {
            ktile6<T, false, 0>(acc, smem, baseA, baseB, c, wave, voffA, voffW, (t + 1) * ROWB, true);
            ktile6<T, false, 1>(acc, smem, baseA, baseB, c, wave, voffA, voffW, (t + 2) * ROWB, t + 2 < nt);
        }
        PG_TS(g, dbg_iter, wave, 1);
        // LN epilogues: colsum of the lane's 8 columns and the row statistics of slab 0, fetched as soon as the 32 fragment
        // registers are dead (after the last MFMA phase); they land under the re-align barrier, the prefetch and slab 0's parking
        const int row0 = c.m0 + wm * (P6_TM * 32), col0 = c.n0 + wn * 64;
        Bias6 cs;
        RowStat6 rs0;
        Bias6 b23, cs23;
        RowStatP rsp;
        __amdgpu_buffer_rsrc_t rrs = c.ra;                   // placeholder for the plain epilogues (never dereferenced)
        if constexpr (PARK16) load_bias6<16>(b23, g, col0 + 32 + ecc);
        if constexpr (ln6<EPI>()) {
            cs.lo = *(const f32x4*)(g.ex.colsum + col0 + ecc);
            cs.hi = *(const f32x4*)(g.ex.colsum + col0 + ecc + BHOFF);
            rrs = rowstat_rsrc6(g, row0);
            if constexpr (PARK16) {
                cs23.lo = *(const f32x4*)(g.ex.colsum + col0 + 32 + ecc);
                cs23.hi = *(const f32x4*)(g.ex.colsum + col0 + 48 + ecc);
                load_rowstat_p(rsp, rrs, l15, 0);
            } else {
                load_rowstat6(rs0, rrs, lane >> 3, 0);
            }
        }
        ResidCtx6 rc;
        XRows6 x0;
        if constexpr (RSTAT) {                               // residual rows of slab 0 into the (dead) fragment registers
            rc = make_resid6(g, row0, col0, lane);
            fetch_xrows6<0, P6_X0_EARLY>(x0, rc.ro, rc.voff, 0, 8 * (int)g.ldc * 4);
        }
        if (!follower) raw_barrier();                        // re-align: every wave has left the mainloop
        // inline-asm MFMAs: hipcc pads no "matrix-pipe write -> VALU / LDS read" hazard for the accumulators (see gemm_pp.hip)
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        PG_TS(g, dbg_iter, wave, 2);

        L += nblk;
        const bool more = L < g.ntiles;
        Bias6 bias_next = bias;
        // unconditional on purpose (see gemm_pp.hip: a VMEM block under `if (more)` breaks the compiler's in-order vmcnt count)
        auto prefetch_dma = [&]() {
            c = make_tile6(g, more ? L : L - nblk);
            issue_dma6<0, P6_NDMA>(c, smem, wave, voffA, voffW, 0);
        };
        auto prefetch_bias = [&]() { load_bias6<BHOFF>(bias_next, g, c.n0 + wn * 64 + ecc); };
        if constexpr (RSTAT) epilogue6_resid<T>(acc, g, smem, wave, lane, row0, col0, bias, x0, rc, prefetch_dma, prefetch_bias, dbg_iter);
        else if constexpr (PARK16) epilogue6_park16<T, EPI>(acc, g, smem, wave, lane, row0, col0, bias, b23, cs, cs23, rsp, rrs, prefetch_dma, prefetch_bias, dbg_iter);
        else epilogue6<T, EPI>(acc, g, smem, wave, lane, row0, col0, bias, cs, rs0, rrs, prefetch_dma, prefetch_bias, dbg_iter);
        PG_TS(g, dbg_iter, wave, 9);
        ++dbg_iter;
        if (!more) break;
        pin_bias6(bias_next);
        bias = bias_next;
        first = false;
    }
}

template <typename T, int EPI, bool PARK = false>
int launch_pp6(const GemmArgs& g, int nblk, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm_pp6_kernel<T, EPI, PARK>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, P6_LDS);
        if (e != hipSuccess) { pg_set_error("gemm_pp6: set LDS attr: %s", hipGetErrorString(e)); return PG_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(512), P6_LDS, s, g);
    return pg_check_launch("gemm_pp6");
}

// 16-bit epilogues of the 384 x 256 kernel: finished in the accumulator layout, 16-bit slabs (epilogue6_park16).  Tools build:
// PIGEON_GEMM_PARK16=0 selects round 2's fp32-slab epilogue6 (A/B arm; same bits).
bool park16_enabled() {
#ifdef PIGEON_ABLATIONS
    static int v = -1;
    if (v < 0) { const char* e = getenv("PIGEON_GEMM_PARK16"); v = e ? (atoi(e) != 0) : 1; }
    return v != 0;
#else
    return true;
#endif
}

int cus6() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    }
    return n > 0 ? n : 256;
}

}  // namespace

// true if this (epilogue, shape) has a 384 x 256 kernel
bool pg_gemm_pp6_supported(int epi, int N, int K) {
    return (epi == EPI_QKV || epi == EPI_GELU || epi == EPI_QKV_LN || epi == EPI_GELU_LN || epi == EPI_RESID_STAT) && N % P6_BN == 0 &&
           K % (2 * BK) == 0 && K >= 4 * BK;
}

int pg_gemm_pp6_launch(int dtype, GemmArgs g, int epi, hipStream_t s) {
    if (!pg_gemm_pp6_supported(epi, g.N, g.K)) { pg_set_error("gemm_pp6: unsupported epilogue / shape (epi=%d N=%d K=%d)", epi, g.N, g.K); return PG_EINVAL; }
    if ((int64_t)g.lda * 2 * P6_BM >= (1ll << 31) || (int64_t)g.ldw * 2 * P6_BN >= (1ll << 31)) {
        pg_set_error("gemm_pp6: operand panel exceeds the 2 GB buffer-descriptor range");
        return PG_EINVAL;
    }
    g.tilesM = (g.M + P6_BM - 1) / P6_BM;
    g.tilesN = g.N / P6_BN;
    g.ntiles = g.tilesM * g.tilesN;
    g.gn = (g.tilesN % 4 == 0) ? 4 : g.tilesN;               // 8 x 4 super-tiles per XCD round (gemm_pp.hip variant 36)
    {   // pg_tune_gemm_raster: -1 = an XCD round walks ALL N tiles of 32 / tilesN row panels (the A panels cross the fabric once,
        // the weight panels stream from L2 / the Infinity Cache); n > 0 = n N tiles per group -- honoured only when n divides
        // tilesN (make_tile6's super-tile decode is a bijection only then; any other n keeps the default).  Raster only: results
        // do not move (tests/test_gpu_parity.py::test_gemm_raster_knob_changes_nothing).
        const int rg = pg_gemm_raster_gn();
        if (rg == -1 || rg >= g.tilesN) g.gn = g.tilesN;
        else if (rg > 0 && g.tilesN % rg == 0) g.gn = rg;
    }
    int cap = cus6();
    if (pg_gemm_block_cap() > 0 && pg_gemm_block_cap() < cap) cap = pg_gemm_block_cap();   // tuning: share the chip between streams
    const int nblk = g.ntiles < cap ? g.ntiles : cap;
    if ((epi == EPI_QKV_LN || epi == EPI_GELU_LN) && (!g.ex.colsum || !g.ex.rowstat)) { pg_set_error("gemm_pp6: LN epilogue needs colsum / rowstat"); return PG_EINVAL; }
    if (epi == EPI_RESID_STAT && (!g.ex.x16 || !g.ex.statpart || g.ex.ldx != g.ldc || !g.bias)) { pg_set_error("gemm_pp6: EPI_RESID_STAT needs bias, x16 / statpart and ldx == ldc"); return PG_EINVAL; }
    const bool park = park16_enabled();
#ifdef PIGEON_ABLATIONS
#define P6_OLD16(TT, E) launch_pp6<TT, E, false>(g, nblk, s)
#else
#define P6_OLD16(TT, E) launch_pp6<TT, E, true>(g, nblk, s)      /* the product library carries the production epilogue only */
#endif
#define P6_DISPATCH(TT)                                                          \
    switch (epi) {                                                               \
        case EPI_QKV: return park ? launch_pp6<TT, EPI_QKV, true>(g, nblk, s) : P6_OLD16(TT, EPI_QKV);                \
        case EPI_GELU: return park ? launch_pp6<TT, EPI_GELU, true>(g, nblk, s) : P6_OLD16(TT, EPI_GELU);              \
        case EPI_QKV_LN: return park ? launch_pp6<TT, EPI_QKV_LN, true>(g, nblk, s) : P6_OLD16(TT, EPI_QKV_LN);          \
        case EPI_RESID_STAT: return launch_pp6<TT, EPI_RESID_STAT>(g, nblk, s);  \
        default: return park ? launch_pp6<TT, EPI_GELU_LN, true>(g, nblk, s) : P6_OLD16(TT, EPI_GELU_LN);                 \
    }
    if (dtype == PG_DTYPE_F16) { P6_DISPATCH(T_F16) }
    if (dtype == PG_DTYPE_BF16) { P6_DISPATCH(T_BF16) }
#undef P6_DISPATCH
#undef P6_OLD16
    pg_set_error("gemm_pp6: operand dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16 (got %d)", dtype);
    return PG_EINVAL;
}
