// gemm_tail.hip -- the last rows of a GEMM whose tile count does not fill the final round of the persistent kernels.
//
// The persistent kernels (gemm_pp.hip 256 x 256, gemm_pp6.hip 384 x 256) give every CU the same number of equal-time tiles,
// so a launch takes ceil(tiles / CUs) tile periods.  The benchmark batch (512 images x 577 tokens = 295 424 rows = 1154
// row panels of 256) makes 4616 tiles in out-proj / fc2: 18 full rounds on 256 CUs and a 19th round in which 8 CUs work and
// 248 wait for 45 - 125 us; the 384-row kernels lose 0.09 / 0.125 of a round the same way (37 instead of 36.09 rounds in
// QKV, 49 instead of 48.125 in fc1).  pg_gemm_launch therefore cuts such a problem at the last row panel that still makes
// whole rounds (294 912 rows here, for all four GEMMs) and hands the remaining few hundred rows to this kernel, which
// spreads them over the whole chip in tiles of 32 rows x 64 columns, one wave each (256 - 1024 waves).
//
// RESULTS ARE BIT-IDENTICAL to the persistent kernels (tests/test_gpu_parity.py::test_gemm_tail_*): every output element is
// the same chain of v_mfma_f32_16x16x32 over ascending k (same operand roles, first step onto 0), the accumulators go through
// an LDS slab into the same 8-columns-per-lane geometry, and the epilogue arithmetic (incl. the association order of the
// row statistics) is the ONE definition in gemm_epi.h that gemm_pp.hip pp_epilogue uses too.  So a row's value still does not depend on where in a batch it sits.
//
// No LDS staging of the operands: a wave reads its fragments straight from L2 / HBM into registers (16 rows x 64 contiguous
// bytes per load instruction), eight k-steps ahead.  A 512-row tail is 1 - 4 GFLOP; the kernel is latency-, not
// throughput-bound, and it is short (profiles/r02/gemm_tail.txt).
#include "gemm_epi.h"

namespace {

constexpr int TL_BM = 32, TL_BN = 64;
constexpr int TL_DEPTH = 8;                                  // k-steps (of 32) in flight per wave: 6 x 16 bytes per lane each (48 KB per wave)
constexpr int TL_ROWPF = 64 + 4;                             // slab row in floats (same padding as the persistent kernels)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tl_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}

template <typename T> struct FragTL { typename T::v8 a[2], b[4]; };

// fragment row lane & 15 of a 16-row block, k elements 8 (lane >> 4) .. of the k-step; rows past M fail the descriptor's
// bounds check (the k advance rides in the SGPR offset, which is not part of that check) and read 0
template <typename T>
__device__ __forceinline__ void load_frag_tl(FragTL<T>& f, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rw,
                                             const int (&va)[2], const int (&vw)[4], int kbytes) {
#pragma unroll
    for (int i = 0; i < 2; ++i) f.a[i] = __builtin_bit_cast(typename T::v8, __builtin_amdgcn_raw_buffer_load_b128(ra, va[i], kbytes, 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) f.b[j] = __builtin_bit_cast(typename T::v8, __builtin_amdgcn_raw_buffer_load_b128(rw, vw[j], kbytes, 0));
}

template <int EPI> constexpr bool tl_out16() { return EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool tl_ln() { return EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }

template <typename T, int EPI>
__global__ __launch_bounds__(64) void gemm_tail_kernel(GemmArgs g, int m_begin) {
    __shared__ __attribute__((aligned(16))) float slab[TL_BM * TL_ROWPF];
    constexpr bool OUT16 = tl_out16<EPI>();
    constexpr bool LN = tl_ln<EPI>();
    constexpr bool STAT = (EPI == EPI_RESID_STAT);
    constexpr bool RESID = (EPI == EPI_RESID || EPI == EPI_RESID_STAT);
    const int lane = threadIdx.x;
    // Block b runs on XCD b % 8 (round-robin dispatch).  XCD x gets the column tiles tn = x (mod 8): its L2 then holds one
    // eighth of W (1 MB for fc2) instead of all of it -- with the plain (tm, tn) order every XCD pulled the whole weight
    // matrix through the fabric, ~100 MB for a 512-row fc2 tail -- and walks them column-fastest, so neighbours share A rows.
    int tn, tm;
    if (g.tilesN % 8 == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = g.tilesN >> 3;
        tn = xcd + 8 * (idx % per); tm = idx / per;
    } else {
        tn = blockIdx.x % g.tilesN; tm = blockIdx.x / g.tilesN;
    }
    const int row0 = m_begin + tm * TL_BM, col0 = tn * TL_BN;
    const int l15 = lane & 15, lq = lane >> 4;
    const int rows = min(TL_BM, g.M - row0);
    const __amdgpu_buffer_rsrc_t ra = tl_rsrc(g.A + (int64_t)row0 * g.lda, (uint32_t)rows * (uint32_t)g.lda * 2u);
    const __amdgpu_buffer_rsrc_t rw = tl_rsrc(g.W + (int64_t)col0 * g.ldw, (uint32_t)TL_BN * (uint32_t)g.ldw * 2u);
    int va[2], vw[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) va[i] = (i * 16 + l15) * (int)g.lda * 2 + lq * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) vw[j] = (j * 16 + l15) * (int)g.ldw * 2 + lq * 16;

    const int nks = g.K / 32;                                // a multiple of TL_DEPTH (K % 256 == 0, checked on the host)
    FragTL<T> fr[TL_DEPTH];
#pragma unroll
    for (int d = 0; d < TL_DEPTH; ++d) {
        load_frag_tl<T>(fr[d], ra, rw, va, vw, d * 64);
        // in k order: if hipcc issues k-step 0 last here, the loop's first wait becomes vmcnt(0) for every iteration
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nks; ks += TL_DEPTH) {
#pragma unroll
        for (int d = 0; d < TL_DEPTH; ++d) {
            // swapped operands (weights first) as in the persistent kernels: a lane owns output row lane & 15 of a 16 x 16
            // block and the 4 consecutive columns 4 (lane >> 4) ..
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(fr[d].b[j], fr[d].a[i], acc[i][j]);
            // unconditional refill (the last DEPTH - 1 re-read the final k-step and are never used): a branch here would
            // make hipcc's in-order vmcnt bookkeeping conservative
            const int kn = min(ks + d + TL_DEPTH, nks - 1);
            load_frag_tl<T>(fr[d], ra, rw, va, vw, kn * 64);
            __builtin_amdgcn_sched_barrier(0);               // keep the ring in program order: k-step d waits for ITS six loads only
        }
    }

    // accumulators -> slab -> row-major pieces: lane (rr, cc) holds 8 consecutive columns of rows rr, rr + 8, rr + 16, rr + 24
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (ib * 16 + l15) * TL_ROWPF + j * 16 + 4 * lq) = acc[ib][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // EPI_RESID_STAT uses pp_epilogue's split-halves geometry (columns 4k.. and 32 + 4k.. per lane), everything else 8 consecutive
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
        f32x4 lo = *(const f32x4*)(slab + r * TL_ROWPF + cc);
        f32x4 hi = *(const f32x4*)(slab + r * TL_ROWPF + cc + HOFF);
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
}

template <typename T, int EPI>
int launch_tail(const GemmArgs& g, int m_begin, int nblk, hipStream_t s) {
    hipLaunchKernelGGL((gemm_tail_kernel<T, EPI>), dim3(nblk), dim3(64), 0, s, g, m_begin);
    return pg_check_launch("gemm_tail");
}

template <typename T>
int dispatch_tail(const GemmArgs& g, int epi, int m_begin, int nblk, hipStream_t s) {
    switch (epi) {
        case EPI_QKV: return launch_tail<T, EPI_QKV>(g, m_begin, nblk, s);
        case EPI_GELU: return launch_tail<T, EPI_GELU>(g, m_begin, nblk, s);
        case EPI_RESID: return launch_tail<T, EPI_RESID>(g, m_begin, nblk, s);
        case EPI_F32: return launch_tail<T, EPI_F32>(g, m_begin, nblk, s);
        case EPI_RESID_STAT: return launch_tail<T, EPI_RESID_STAT>(g, m_begin, nblk, s);
        case EPI_QKV_LN: return launch_tail<T, EPI_QKV_LN>(g, m_begin, nblk, s);
        case EPI_GELU_LN: return launch_tail<T, EPI_GELU_LN>(g, m_begin, nblk, s);
        default: pg_set_error("gemm_tail: epilogue %d not supported", epi); return PG_EINVAL;
    }
}

}  // namespace

bool pg_gemm_tail_supported(int epi, int N, int K) {
    return epi != EPI_PATCH && epi >= EPI_QKV && epi <= EPI_GELU_LN && N % TL_BN == 0 && K % (32 * TL_DEPTH) == 0;
}

// rows [m_begin, g.M) of the problem; every pointer in g is that of row 0
int pg_gemm_tail_launch(int dtype, GemmArgs g, int epi, int m_begin, hipStream_t s) {
    if (!pg_gemm_tail_supported(epi, g.N, g.K)) { pg_set_error("gemm_tail: unsupported epilogue / shape (epi=%d N=%d K=%d)", epi, g.N, g.K); return PG_EINVAL; }
    if (m_begin < 0 || m_begin >= g.M) return PG_OK;
    if ((int64_t)g.lda * 2 * TL_BM >= (1ll << 31) || (int64_t)g.ldw * 2 * TL_BN >= (1ll << 31)) {
        pg_set_error("gemm_tail: operand panel exceeds the 2 GB buffer-descriptor range");
        return PG_EINVAL;
    }
    g.tilesM = (g.M - m_begin + TL_BM - 1) / TL_BM;
    g.tilesN = g.N / TL_BN;
    g.ntiles = g.tilesM * g.tilesN;
    if (dtype == PG_DTYPE_F16) return dispatch_tail<T_F16>(g, epi, m_begin, g.ntiles, s);
    if (dtype == PG_DTYPE_BF16) return dispatch_tail<T_BF16>(g, epi, m_begin, g.ntiles, s);
    pg_set_error("gemm_tail: operand dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16 (got %d)", dtype);
    return PG_EINVAL;
}
