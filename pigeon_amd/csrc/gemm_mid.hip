// gemm_mid.hip -- the GEMMs of a SMALL batch (round 6): one panorama = 4 images = 2 308 token rows is the reference's serving unit
// (models/super_guessr.py:462-466, one request at a time), and what the exact tier re-encodes when a single call must be settled.
//
// At that size the persistent kernels (gemm_pp.hip 256 x 256, gemm_pp6.hip 384 x 256) put 28 - 112 tiles on 256 CUs: fc2 is 28
// tiles of 64 K tiles each, 110 us of a 240 us layer with nine tenths of the chip idle.  Splitting K would fill the chip but changes
// the accumulation order -- a row's bits would then depend on the batch it rides in.  This kernel keeps the order and shrinks the
// tile instead: 128 x 128 block tiles, 4 waves (2 x 2) of 64 x 64 each, one tile per block, up to 2.2 x the blocks of a 256 x 256
// tiling (152 for fc2 / out-projection, 456 for QKV, 608 for fc1 at one panorama), operands through LDS by direct-to-LDS DMA in a
// THREE-stage ring (a K tile is 32 KB; two K tiles are in flight while one is consumed: a block has too few MFMAs per K tile
// -- 32 per wave, 0.25 us -- to hide a DMA's latency behind a single tile).
//
// RESULTS ARE BIT-IDENTICAL to the persistent kernels and to gemm_tail.hip (tests/test_gpu_parity.py::test_gemm_mid_*): every output
// element is the same chain of v_mfma_f32_16x16x32 over ascending k (weights as the first operand, first step onto 0), the accumulators
// go through an LDS slab into the same 8-columns-per-lane geometry (split halves for EPI_RESID_STAT), and the epilogue arithmetic,
// incl. the association order of the row statistics, is the ONE definition in gemm_epi.h.  Which kernel computes a row is a timing
// decision (pg_gemm_launch: a cost model of the shape); a row's value does not depend on where in a batch -- or in which batch -- it sits.
//
// L2 / fabric: block b runs on XCD b % 8; the XCDs get contiguous chunks of the logical tile order (xcd_remap), and that order walks
// the SMALLER operand slowest, so every XCD streams one eighth of the larger operand and the whole of the smaller one.
#include "gemm_epi.h"

namespace {

constexpr int MD_BM = 128, MD_BN = 128;
#ifndef PG_MID_STAGES
#define PG_MID_STAGES 3                                        // A/B knob (python -m pigeon_amd.build --variant s5 -DPG_MID_STAGES=5): 3 .. 5
#endif
constexpr int MD_STAGES = PG_MID_STAGES;
// A FIFTH wave that does nothing but issue the operand DMAs (second session of round 6).  tools/dma_rate_probe.hip: the 32 DMAs of a K
// tile (32 KB) take the CU's texture path 0.33 us whoever issues them, and a wave that issues VMEM instructions faster than the path
// takes them stalls IN ORDER -- with the four MFMA waves issuing their own 8 DMAs each, every K tile was 0.33 us of DMA issue followed
// by 0.25 us of fragment reads and MFMAs on a SIMD that has no second wave to switch to: the 0.58 us no ring depth and no software
// pipelining moved.  A producer wave takes the issue stall out of the MFMA waves' instruction stream (it shares SIMD 0 with wave 0;
// one wave issuing all 32 DMAs sustains 0.355 us per K tile).  Same MFMA chains, same epilogue: the same bits.
#ifndef PG_MID_PRODUCER
#define PG_MID_PRODUCER 1                                      // A/B knob: python -m pigeon_amd.build --variant noprod -DPG_MID_PRODUCER=0
#endif
constexpr bool MD_PROD = PG_MID_PRODUCER != 0;
constexpr int MD_THREADS = MD_PROD ? 320 : 256;
constexpr int MD_STAGE = (MD_BM + MD_BN) * ROWB;               // 32 KB
constexpr int MD_W_OFF = MD_BM * ROWB;
constexpr int MD_NDMA = (MD_BM + MD_BN) / 8 / 4;               // DMAs per wave per K tile (8 rows x 128 B each): 8
constexpr int MD_LDS = MD_STAGES * MD_STAGE;                   // 96 KB: one block per CU
constexpr int MD_ROWPF = 64 + 4;                               // slab row in floats (the persistent kernels' padding)
constexpr int MD_SLAB_BYTES = 32 * MD_ROWPF * 4;               // 8704 B per wave, overlaying stage 0 after the mainloop

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void md_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_uniform, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_uniform, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t md_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

// fragment reads as inline asm (gemm_pp6.hip lds_read_b128): written as C++ loads, hipcc may put `s_waitcnt vmcnt(0)` in front of an LDS
// read while a direct-to-LDS DMA is in flight -- it assumes the two can alias -- and the ring's prefetch would be serialised
template <int OFF, typename V>
__device__ __forceinline__ void md_lds_read(V& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

template <int EPI> constexpr bool md_out16() { return EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool md_ln() { return EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }

// the eight DMAs of one K tile of this wave: groups w, w + 4, .. (8 rows each); groups 0 .. 15 are A rows, 16 .. 31 W rows
__device__ __forceinline__ void md_issue(__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw, char* stage, int wave,
                                         const int (&voffA)[4], const int (&voffW)[4], int soff) {
#pragma unroll
    for (int d = 0; d < 4; ++d) md_dma16(ra, stage + (wave + 4 * d) * 8 * ROWB, voffA[d], soff);
#pragma unroll
    for (int d = 0; d < 4; ++d) md_dma16(rw, stage + MD_W_OFF + (wave + 4 * d) * 8 * ROWB, voffW[d], soff);
}

// One 32-row x 64-column slab of a wave: accumulator blocks -> LDS -> row-major pieces -> fused epilogue.  This is gemm_tail.hip's
// epilogue (which is pp_epilogue's): same geometry, same gemm_epi.h expressions.
template <typename T, int EPI, int H>
__device__ __forceinline__ void md_slab(const GemmArgs& g, float* slab, int lane, int row0, int col0, const f32x4 (&acc)[4][4]) {
    constexpr bool OUT16 = md_out16<EPI>();
    constexpr bool LN = md_ln<EPI>();
    constexpr bool STAT = (EPI == EPI_RESID_STAT);
    constexpr bool RESID = (EPI == EPI_RESID || EPI == EPI_RESID_STAT);
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (ib * 16 + l15) * MD_ROWPF + j * 16 + 4 * lq) = acc[2 * H + ib][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int HOFF = STAT ? 32 : 4;
    const int rr = lane >> 3, cc = STAT ? (lane & 7) * 4 : (lane & 7) * 8;
    const int col = col0 + cc;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 b_lo = zero4, b_hi = zero4, s_lo = zero4, s_hi = zero4;
    if (g.bias) { b_lo = *(const f32x4*)(g.bias + col); b_hi = *(const f32x4*)(g.bias + col + HOFF); }
    if constexpr (LN) { s_lo = *(const f32x4*)(g.ex.colsum + col); s_hi = *(const f32x4*)(g.ex.colsum + col + 4); }
    const float qsc = ((EPI == EPI_QKV || EPI == EPI_QKV_LN) && col < g.qcols) ? g.qscale : 1.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + rr;
        const int row = row0 + r;
        if (row >= g.M) continue;
        f32x4 lo = *(const f32x4*)(slab + r * MD_ROWPF + cc);
        f32x4 hi = *(const f32x4*)(slab + r * MD_ROWPF + cc + HOFF);
        if constexpr (OUT16) {
            float rstd = 0.f, mrs = 0.f;
            if constexpr (LN) {
                const u32x2 rs = *(const u32x2*)(g.ex.rowstat + (int64_t)row * 2);
                // (the asm moves: see gemm_pp.hip -- hipcc SLP-packs the fmas and broadcasts the wrong half otherwise)
                asm("v_mov_b32 %0, %1" : "=v"(rstd) : "v"(rs[0]));
                asm("v_mov_b32 %0, %1" : "=v"(mrs) : "v"(rs[1]));
            }
            *(u32x4*)((uint16_t*)g.out + (int64_t)row * g.ldc + col) =
                epi16_finish<T, EPI>(lo, hi, b_lo, b_hi, s_lo, s_hi, rstd, mrs, col0 < g.qcols, qsc);
        } else if constexpr (RESID) {
            float* p = (float*)g.out + (int64_t)row * g.ldc + col;
            const f32x4 x = epi_resid4(*(const f32x4*)p, lo, b_lo);
            const f32x4 y = epi_resid4(*(const f32x4*)(p + HOFF), hi, b_hi);
            *(f32x4*)p = x;
            *(f32x4*)(p + HOFF) = y;
            if constexpr (STAT) {
                uint16_t* p16 = (uint16_t*)g.ex.x16 + (int64_t)row * g.ldc + col;
                *(u32x2*)p16 = epi_copy16x4<T>(x);
                *(u32x2*)(p16 + HOFF) = epi_copy16x4<T>(y);
                float s1, s2;
                epi_stat8(x, y, s1, s2);
                s1 = row8_sum(s1);
                s2 = row8_sum(s2);
                if ((lane & 7) == 0) {
                    float* sp = g.ex.statpart + ((int64_t)(col0 / 64) * g.ex.stat_rows + row) * 2;
                    sp[0] = s1; sp[1] = s2;
                }
            }
        } else {                                             // EPI_F32
            float* p = (float*)g.out + (int64_t)row * g.ldc + col;
            *(f32x4*)p = lo + b_lo;
            *(f32x4*)(p + 4) = hi + b_hi;
        }
    }
    // the next slab of this wave overwrites the same LDS: its reads above have returned (their values were consumed) before any
    // later write of this wave is issued -- LDS operations of one wave execute in order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int EPI>
__global__ __launch_bounds__(MD_THREADS) void gemm_mid_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // logical tile id: XCD x (= blockIdx % 8) owns a contiguous chunk; g.gn != 0: row tile slowest (an XCD streams all of W and an
    // eighth of A), g.gn == 0: column tile slowest (all of A, an eighth of W) -- the host picks the smaller operand to replicate
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    if (g.gn) { tm = L / g.tilesN; tn = L - tm * g.tilesN; }
    else { tn = L / g.tilesM; tm = L - tn * g.tilesM; }
    const int m0 = tm * MD_BM, n0 = tn * MD_BN;
    const int rows = min(MD_BM, g.M - m0);
    const __amdgpu_buffer_rsrc_t ra = md_rsrc(g.A + (int64_t)m0 * g.lda, (uint32_t)rows * (uint32_t)g.lda * 2u);
    const __amdgpu_buffer_rsrc_t rw = md_rsrc(g.W + (int64_t)n0 * g.ldw, (uint32_t)MD_BN * (uint32_t)g.ldw * 2u);
    // per-lane DMA offsets (bytes) inside the operand panels: row (wave + 4 d) * 8 + (lane >> 3), logical 16-byte chunk
    // (lane & 7) ^ ((row >> 1) & 7) -- the swizzle on the SOURCE address, the LDS image is lane-linear (gemm_pp.hip)
    int voffA[4], voffW[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int r = (wave + 4 * d) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffA[d] = r * (int)g.lda * 2 + c * 16;
        voffW[d] = r * (int)g.ldw * 2 + c * 16;
    }
    const int nt = g.K / BK;
    if constexpr (MD_PROD) {
        static_assert(!MD_PROD || MD_STAGES == 3, "the producer form is written for the three-stage ring");
        if (wave == 4) {
            // THE PRODUCER: all 32 DMAs of a K tile -- row group gr (8 rows) of a panel = the lane offsets of group (gr & 1) + (gr >> 1)
            // x 16 rows, in the VGPR offset (the descriptor's bounds check, which zero-fills the A rows past M, looks at that one only)
            int pvA[2], pvW[2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int r = d * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                pvA[d] = r * (int)g.lda * 2 + c * 16;
                pvW[d] = r * (int)g.ldw * 2 + c * 16;
            }
            const int stepA = 16 * (int)g.lda * 2, stepW = 16 * (int)g.ldw * 2;
            auto issue = [&](int t, char* stage) {
#pragma unroll
                for (int gr = 0; gr < 16; ++gr) md_dma16(ra, stage + gr * 8 * ROWB, pvA[gr & 1] + (gr >> 1) * stepA, t * BK * 2);
#pragma unroll
                for (int gr = 0; gr < 16; ++gr) md_dma16(rw, stage + MD_W_OFF + gr * 8 * ROWB, pvW[gr & 1] + (gr >> 1) * stepW, t * BK * 2);
            };
            issue(0, smem);
            if (nt > 1) issue(1, smem + MD_STAGE);
            int freed = 2 * MD_STAGE;                        // where K tile t + 2 goes: stage (t + 2) % 3 == the stage of K tile t - 1
            for (int t = 0; t < nt; ++t) {
                // K tile t has landed when at most the 32 DMAs of tile t + 1 are outstanding (vmcnt retires in order)
                if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                // publishes K tile t; the MFMA waves have finished K tile t - 1
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < nt) issue(t + 2, smem + freed);
                freed += MD_STAGE;
                if (freed >= MD_LDS) freed = 0;
            }
            __builtin_amdgcn_s_barrier();                    // the MFMA waves have left the operand stages (their slabs overlay them)
            return;
        }
    } else {
#pragma unroll
        for (int t = 0; t < MD_STAGES - 1; ++t)
            if (t < nt) md_issue(ra, rw, smem + t * MD_STAGE, wave, voffA, voffW, t * BK * 2);
    }

    const int l15 = lane & 15, lq = lane >> 4;
    const int sw = (lane >> 1) & 7;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_base = lds0 + (wm * 64 + l15) * ROWB, b_base = lds0 + MD_W_OFF + (wn * 64 + l15) * ROWB;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int cur = 0;                                             // byte offset of the stage that holds K tile kt
    for (int kt = 0; kt < nt; ++kt) {
        // this wave's DMAs of K tile kt have landed when at most those of the tiles issued after it (kt + 1 .. kt + STAGES - 2, 8 each)
        // are outstanding (vmcnt retires in order)
        static_assert(MD_STAGES >= 3 && MD_STAGES <= 5, "the vmcnt ladder below covers 3 .. 5 stages");
        if constexpr (!MD_PROD) {
            const int ahead = min(MD_STAGES - 2, nt - 1 - kt);
            if (MD_STAGES >= 5 && ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * MD_NDMA) : "memory");
            else if (MD_STAGES >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MD_NDMA) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MD_NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                        // ... and everybody's; everybody has also left K tile kt - 1's stage
        __builtin_amdgcn_sched_barrier(0);
        if (!MD_PROD && kt + MD_STAGES - 1 < nt) {
            int nxt = cur - MD_STAGE;                        // stage (kt + STAGES - 1) % STAGES == the stage of K tile kt - 1
            if (nxt < 0) nxt += MD_LDS;
            md_issue(ra, rw, smem + nxt, wave, voffA, voffW, (kt + MD_STAGES - 1) * BK * 2);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t xo = (uint32_t)(((ks * 4 + lq) ^ sw) << 4) + (uint32_t)cur;
            const uint32_t aa = a_base + xo, ab = b_base + xo;
            typename T::v8 a[4], b[4];
            md_lds_read<0 * 16 * ROWB>(a[0], aa); md_lds_read<1 * 16 * ROWB>(a[1], aa);
            md_lds_read<2 * 16 * ROWB>(a[2], aa); md_lds_read<3 * 16 * ROWB>(a[3], aa);
            md_lds_read<0 * 16 * ROWB>(b[0], ab); md_lds_read<1 * 16 * ROWB>(b[1], ab);
            md_lds_read<2 * 16 * ROWB>(b[2], ab); md_lds_read<3 * 16 * ROWB>(b[3], ab);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // swapped operands (weights first) as in every GEMM kernel of the library
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(b[j], a[i], acc[i][j]);
        }
        cur += MD_STAGE;
        if (cur >= MD_LDS) cur = 0;
    }
    if constexpr (MD_PROD) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // (the producer wave joins this one and leaves)
    } else {
        __syncthreads();                                     // every wave has left the operand stages: the slabs may overlay them
    }
    float* slab = (float*)(smem + wave * MD_SLAB_BYTES);
    const int row0 = m0 + wm * 64, col0 = n0 + wn * 64;
    md_slab<T, EPI, 0>(g, slab, lane, row0, col0, acc);
    md_slab<T, EPI, 1>(g, slab, lane, row0 + 32, col0, acc);
}

template <typename T, int EPI>
int launch_mid(const GemmArgs& g, int nblk, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm_mid_kernel<T, EPI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, MD_LDS);
        if (e != hipSuccess) { pg_set_error("gemm_mid: set LDS attr: %s", hipGetErrorString(e)); return PG_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(MD_THREADS), MD_LDS, s, g);
    return pg_check_launch("gemm_mid");
}

template <typename T>
int dispatch_mid(const GemmArgs& g, int epi, int nblk, hipStream_t s) {
    switch (epi) {
        case EPI_QKV: return launch_mid<T, EPI_QKV>(g, nblk, s);
        case EPI_GELU: return launch_mid<T, EPI_GELU>(g, nblk, s);
        case EPI_RESID: return launch_mid<T, EPI_RESID>(g, nblk, s);
        case EPI_F32: return launch_mid<T, EPI_F32>(g, nblk, s);
        case EPI_RESID_STAT: return launch_mid<T, EPI_RESID_STAT>(g, nblk, s);
        case EPI_QKV_LN: return launch_mid<T, EPI_QKV_LN>(g, nblk, s);
        case EPI_GELU_LN: return launch_mid<T, EPI_GELU_LN>(g, nblk, s);
        default: pg_set_error("gemm_mid: epilogue %d not supported", epi); return PG_EINVAL;
    }
}

}  // namespace

bool pg_gemm_mid_supported(int epi, int N, int K) {
    return epi != EPI_PATCH && epi >= EPI_QKV && epi <= EPI_GELU_LN && N % MD_BN == 0 && K % BK == 0 && K >= BK;
}

// rows [m_begin, g.M) of the problem; every pointer in g is that of row 0 (m_begin > 0: the rows a persistent launch left over,
// pg_gemm_launch's tail split)
int pg_gemm_mid_launch(int dtype, GemmArgs g, int epi, hipStream_t s, int m_begin) {
    if (!pg_gemm_mid_supported(epi, g.N, g.K)) { pg_set_error("gemm_mid: unsupported epilogue / shape (epi=%d N=%d K=%d)", epi, g.N, g.K); return PG_EINVAL; }
    if (g.ex.parts > 1) { pg_set_error("gemm_mid: several products in one launch exist in gemm_pp.hip only"); return PG_EINVAL; }
    if (m_begin < 0 || m_begin >= g.M) return m_begin == g.M ? PG_OK : (pg_set_error("gemm_mid: m_begin = %d outside [0, %d)", m_begin, g.M), PG_EINVAL);
    if (m_begin > 0) {
        // the kernel indexes every row-major buffer by the row number: move the bases instead of teaching it an offset
        const bool out16 = epi == EPI_QKV || epi == EPI_GELU || epi == EPI_QKV_LN || epi == EPI_GELU_LN;
        if (g.ex.stat_rows <= 0) g.ex.stat_rows = g.M;       // the slices of statpart keep the stride of the WHOLE problem
        g.A += (int64_t)m_begin * g.lda;
        g.out = (char*)g.out + (int64_t)m_begin * g.ldc * (out16 ? 2 : 4);
        if (g.ex.rowstat) g.ex.rowstat += (int64_t)m_begin * 2;
        if (g.ex.x16) g.ex.x16 = (uint16_t*)g.ex.x16 + (int64_t)m_begin * g.ldc;
        if (g.ex.statpart) g.ex.statpart += (int64_t)m_begin * 2;
        g.M -= m_begin;
    }
    if ((int64_t)g.lda * 2 * MD_BM >= (1ll << 31) || (int64_t)g.ldw * 2 * MD_BN >= (1ll << 31)) {
        pg_set_error("gemm_mid: operand panel exceeds the 2 GB buffer-descriptor range");
        return PG_EINVAL;
    }
    g.tilesM = (g.M + MD_BM - 1) / MD_BM;
    g.tilesN = g.N / MD_BN;
    g.ntiles = g.tilesM * g.tilesN;
    g.gn = g.M >= g.N ? 1 : 0;                               // replicate the smaller operand over the XCDs (see the kernel)
    if (dtype == PG_DTYPE_F16) return dispatch_mid<T_F16>(g, epi, g.ntiles, s);
    if (dtype == PG_DTYPE_BF16) return dispatch_mid<T_BF16>(g, epi, g.ntiles, s);
    pg_set_error("gemm_mid: operand dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16 (got %d)", dtype);
    return PG_EINVAL;
}
