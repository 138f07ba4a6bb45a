// gemm_bf16.hip -- C[M,N] = A[M,K] * W[N,K]^T, 16-bit (fp16 or bf16) operands, fp32 accumulation on MFMA,
// fused epilogues.
//
// Replaces the nn.Linear / Conv2d library GEMMs the reference reaches through transformers
// (modeling_clip.py CLIPAttention.q/k/v/out_proj, CLIPMLP.fc1/fc2, CLIPVisionEmbeddings.patch_embedding),
// SURVEY.md section 2c rows K1,K4,K6,K7,K8 = 91% of the path's FLOPs.
//
// Design (gfx950 / CDNA4, wave64):
//   * v_mfma_f32_32x32x16_{f16,bf16}; both operands are K-contiguous in memory ([M,K] activations, [N,K] weights as
//     nn.Linear stores them) so a lane's fragment is one 16-byte ds_read_b128.
//   * K tile BK = 64 (128-byte rows in LDS).  Tiles are staged global->LDS with the direct-to-LDS DMA
//     (global_load_lds_dwordx4: 64 lanes x 16 B = 8 rows per wave instruction), no VGPR round trip.
//   * LDS image is lane-linear (a DMA constraint), so the bank-conflict swizzle is applied to the SOURCE
//     address: 16-byte chunk c of row r is stored at chunk c ^ ((r>>1)&7).  A ds_read_b128 lane group
//     (16 lanes) then touches 16 distinct 16-byte slots of the 256-byte bank row.
//   * Double-buffered LDS, one barrier per K tile: DMA of tile t+1 is in flight while tile t is multiplied.
//   * The MFMA is issued with the operands swapped (weights as "A", activations as "B") so a lane ends up
//     owning ONE output row m and quads of 4 consecutive columns n: float4-sized pieces that the epilogue
//     transposes through LDS into full-row 16-byte global stores (bf16: 8 columns per lane; fp32: 4).
//   * 1-D grid with a bijective XCD remap; the N tile index runs fastest inside an XCD's chunk so the blocks
//     that share an A row-panel run on the same L2, and the weight matrix stays L2/MALL resident.
//   * M tail: source rows are clamped to M-1 (reads stay in bounds), stores are masked.
#include "common.h"
#include "pigeon_internal.h"

#include "gemm_epi.h"

// Tile rasterisation.  Blocks are first remapped so every XCD owns a contiguous chunk of logical ids (hardware
// places block b on XCD b % 8), then logical ids walk the tile grid in groups of `gn` N-tiles: inside a group
// the N index runs fastest, then M, then the next group.  The gn weight panels of a group (gn x 256 x K) stay
// resident in the XCD's 4 MB L2 while the XCD walks down its rows; an A row-panel is fetched from HBM/MALL once
// per group and shared by the gn blocks that run side by side.
__device__ __forceinline__ void tile_coords(const GemmArgs& g, int& tm, int& tn) {
    const int wg = xcd_remap(blockIdx.x, g.ntiles);
    const int gsz = g.tilesM * g.gn;
    const int grp = wg / gsz, rem = wg - grp * gsz;
    const int gn_here = min(g.gn, g.tilesN - grp * g.gn);
    tm = rem / gn_here;
    tn = grp * g.gn + (rem - tm * gn_here);
}

// ---- LDS-staged epilogue: per wave, one 32 x WTN fp32 slab at a time -------------------------------------------
// The accumulators hold D[n][m] (operands swapped), i.e. lane owns row m = lane&31 and 4-column quads.  Each wave
// parks a 32-row slab in its private LDS region, then re-reads it row-major so that a lane stores 16 contiguous
// bytes (8 x 16-bit or 4 x fp32) and a wave instruction covers whole 128/256-byte row segments.
template <typename T, int EPI, int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void staged_epilogue(f32x16 (&acc)[TM][TN], const GemmArgs& g, char* smem, int wave, int lane,
                                                int row0 /* first row of this wave's tile */, int col0) {
    constexpr bool OUT16 = (EPI == EPI_QKV || EPI == EPI_GELU);
    constexpr int ROWPF = WTN + 4;                           // padded slab row, floats (272 B for WTN=64)
    const int lrow = lane & 31, lhalf = lane >> 5;
    float* slab = (float*)(smem + wave * (32 * ROWPF * 4));
    __syncthreads();                                         // every wave is done with the K-loop buffers
    constexpr int CPL = OUT16 ? 8 : 4;                       // columns per lane on the row-major side
    constexpr int LPR = WTN / CPL, RPI = 64 / LPR, ITS = 32 / RPI;
    const int rr = lane / LPR, cc = (lane % LPR) * CPL;
    const int col = col0 + cc;
    f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
        b_lo = *(const f32x4*)(g.bias + col);
        if (OUT16) b_hi = *(const f32x4*)(g.bias + col + 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                *(f32x4*)(slab + lrow * ROWPF + j * 32 + q * 8 + 4 * lhalf) = v;
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
            const int r = it * RPI + rr;
            const int row = row0 + i * 32 + r;
            const f32x4 lo = *(const f32x4*)(slab + r * ROWPF + cc);
            if (OUT16) {
                const f32x4 hi = *(const f32x4*)(slab + r * ROWPF + cc + 4);
                if (row < g.M) epi_store_bf16x8<T, EPI>(g, row, col, lo, hi, b_lo, b_hi);
            } else {
                if (row < g.M) epi_store_f32x4<EPI>(g, row, col, lo, b_lo);
            }
        }
        __syncthreads();
    }
}

// One K tile (BK = 64 = 4 k-steps of 16) of MFMAs for a wave's TM x TN tiles, fragments software-pipelined:
// the ds_reads of k-step kk+1 are issued before the MFMAs of kk.  a_ptr / b_ptr already include the lane's row.
template <typename T, int TM, int TN>
__device__ __forceinline__ void mma_ktile(f32x16 (&acc)[TM][TN], const char* a_ptr, const char* b_ptr, const int (&xoff)[4]) {
    typename T::v8 af[2][TM], bfr[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *(const typename T::v8*)(a_ptr + i * 32 * ROWB + xoff[0]);
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = *(const typename T::v8*)(b_ptr + j * 32 * ROWB + xoff[0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[(kk + 1) & 1][i] = *(const typename T::v8*)(a_ptr + i * 32 * ROWB + xoff[kk < 3 ? kk + 1 : 3]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[(kk + 1) & 1][j] = *(const typename T::v8*)(b_ptr + j * 32 * ROWB + xoff[kk < 3 ? kk + 1 : 3]);
        }
        __builtin_amdgcn_s_setprio(1);
        // swapped operands: D[n][m] -> lane owns row m = lane&31, columns n = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = T::mfma(bfr[kk & 1][j], af[kk & 1][i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    }
}

// Same K tile, but the next tile's direct-to-LDS DMAs are issued in NDMA/4-sized slices between the k-steps
// instead of all at once after the barrier: a DMA issue can stall its wave for 100+ cycles when the vector-memory
// queue is backed up, and spreading them lets the other wave of the SIMD keep the matrix pipe busy meanwhile.
template <typename T, int TM, int TN, int NDMA>
__device__ __forceinline__ void mma_ktile_dma(f32x16 (&acc)[TM][TN], const char* a_ptr, const char* b_ptr, const int (&xoff)[4],
                                              const uint16_t* const (&src)[NDMA], const int (&ldsoff)[NDMA], char* nxt,
                                              int64_t koff, bool do_dma) {
    static_assert(NDMA % 4 == 0, "DMA count must split over the 4 k-steps");
    constexpr int PER = NDMA / 4;
    typename T::v8 af[2][TM], bfr[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *(const typename T::v8*)(a_ptr + i * 32 * ROWB + xoff[0]);
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = *(const typename T::v8*)(b_ptr + j * 32 * ROWB + xoff[0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (kk < 3) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[(kk + 1) & 1][i] = *(const typename T::v8*)(a_ptr + i * 32 * ROWB + xoff[kk < 3 ? kk + 1 : 3]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[(kk + 1) & 1][j] = *(const typename T::v8*)(b_ptr + j * 32 * ROWB + xoff[kk < 3 ? kk + 1 : 3]);
        }
        if (do_dma) {
#pragma unroll
            for (int d = 0; d < PER; ++d) glds16(src[kk * PER + d] + koff, nxt + ldsoff[kk * PER + d]);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = T::mfma(bfr[kk & 1][j], af[kk & 1][i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    }
}

// ================================================================================================================
// Kernel A: BM x BN block tile, WM x WN waves, 2 LDS stages of (A|W), one __syncthreads per K tile.
// ================================================================================================================
template <typename T, int BM, int BN, int WM, int WN, int EPI, bool DIRECT, bool PIPE, int DBG = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int GROUPS = (BM + BN) / 8;        // 8-row DMA groups per stage
    constexpr int LPW = GROUPS / NW;             // DMA instructions per wave per stage
    static_assert(GROUPS % NW == 0, "stage must split evenly over waves");
    static_assert(BM % 16 == 0 && BN % 16 == 0, "tile alignment");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int tm, tn;
    tile_coords(g, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-lane DMA source pointers (advance by BK elements per K tile) ----
    const uint16_t* src[LPW];
    int ldsoff[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int grp = wave + i * NW;           // 8-row group inside the stage
        const int r = grp * 8 + (lane >> 3);     // row inside the stage image (A rows first, then W rows)
        const int pc = lane & 7;                 // physical 16-byte chunk this lane fills
        const int c = pc ^ ((r >> 1) & 7);       // logical chunk it must fetch (swizzle on the source)
        if (grp * 8 < BM) {
            int row = m0 + r;
            row = row < g.M ? row : g.M - 1;
            src[i] = g.A + (int64_t)row * g.lda + c * 8;
        } else {
            const int row = n0 + (r - BM);
            src[i] = g.W + (int64_t)row * g.ldw + c * 8;
        }
        ldsoff[i] = grp * 8 * ROWB;              // wave-uniform LDS base of this DMA
    }

    // ---- per-lane fragment read offsets ----
    const int lrow = lane & 31;
    const int lhalf = lane >> 5;
    const int sw = (lane >> 1) & 7;              // == ((row>>1)&7) for row = 32*j + lrow
    int xoff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xoff[kk] = ((kk * 2 + lhalf) ^ sw) << 4;
    const int a_base = (wm * WTM + lrow) * ROWB;
    const int b_base = (BM + wn * WTN + lrow) * ROWB;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = g.K / BK;

    // prologue: stage tile 0 into buffer 0
#pragma unroll
    for (int i = 0; i < LPW; ++i) glds16(src[i], smem + ldsoff[i]);

    for (int t = 0; t < nt; ++t) {
        // tile t has landed for every wave (the barrier's release carries vmcnt(0) for the in-flight DMA),
        // and every wave is done reading the buffer tile t+1 is about to overwrite.
        __syncthreads();
        const int cur = t & 1;
        if constexpr ((DBG & 4) != 0) {                 // DMA of tile t+1 interleaved with the k-steps of tile t
            mma_ktile_dma<T, TM, TN, LPW>(acc, smem + cur * STAGE + a_base, smem + cur * STAGE + b_base, xoff, src, ldsoff,
                                          smem + (cur ^ 1) * STAGE, (int64_t)(t + 1) * BK, t + 1 < nt);
            continue;
        }
        if (t + 1 < nt && !(DBG & 1)) {                 // DBG&1: ablation -- no DMA inside the loop (results are garbage)
            char* nxt = smem + (cur ^ 1) * STAGE;
#pragma unroll
            for (int i = 0; i < LPW; ++i) glds16(src[i] + (int64_t)(t + 1) * BK, nxt + ldsoff[i]);
        }
        const char* sb = smem + cur * STAGE;
        if (DBG & 2) {                                   // DBG&2: ablation -- MFMAs only, no ds_reads in the loop
            typename T::v8 f = *(const typename T::v8*)(smem + a_base);
            asm volatile("" : "+v"(f));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = T::mfma(f, f, acc[i][j]);
            continue;
        }
        if (PIPE) {
            mma_ktile<T, TM, TN>(acc, sb + a_base, sb + b_base, xoff);
            continue;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            typename T::v8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *(const typename T::v8*)(sb + a_base + i * 32 * ROWB + xoff[kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *(const typename T::v8*)(sb + b_base + j * 32 * ROWB + xoff[kk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = T::mfma(bfr[j], af[i], acc[i][j]);
        }
    }

    if (DIRECT) {
        // simple (slow) epilogue kept as a debugging fallback: scalar stores straight from the accumulators
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * WTM + i * 32 + lrow;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = n0 + wn * WTN + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    if (row < g.M) epi_store_scalar<T, EPI>(g, row, col, acc[i][j][r]);
                }
        }
        return;
    }
    staged_epilogue<T, EPI, TM, TN, WTM, WTN>(acc, g, smem, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
}

// ================================================================================================================
// Kernel B ("a3w2"): 256 x 256 tile, 8 waves (2 x 4), the whole 160 KB of LDS:
//     A ring  3 x 32 KB  -- activations stream from HBM/MALL (every A line is a compulsory miss for the first of
//                           the gn blocks that share it), so A is prefetched TWO K tiles ahead;
//     W ring  2 x 32 KB  -- weights hit the XCD's L2 (see tile_coords), one tile of lookahead is enough.
// vmcnt retires in issue order, so each iteration issues W(t+1) BEFORE A(t+2): at the top of iteration t the
// queue is [A(t) W(t) | A(t+1)] and `s_waitcnt vmcnt(4)` (this wave's 4 newest DMAs may stay in flight) is exactly
// "A(t) and W(t) have landed".  Raw s_barrier + counted waits: a __syncthreads() would drain vmcnt to 0.
// ================================================================================================================
template <typename T, int EPI>
__global__ __launch_bounds__(512) void gemm16_a3w2_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, WN = 4, NW = 8, WTM = 128, WTN = 64, TM = 4, TN = 2;
    constexpr int ASTG = BM * ROWB, WSTG = BN * ROWB;                 // 32 KB each
    constexpr int W_OFF = 3 * ASTG;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int tm, tn;
    tile_coords(g, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const uint16_t* srcA[4];
    const uint16_t* srcW[4];
    int ldsoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int grp = wave + i * NW;                                // 8-row group 0..31 inside a 256-row stage
        const int r = grp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int row = m0 + r;
        row = row < g.M ? row : g.M - 1;
        srcA[i] = g.A + (int64_t)row * g.lda + c * 8;
        srcW[i] = g.W + (int64_t)(n0 + r) * g.ldw + c * 8;
        ldsoff[i] = grp * 8 * ROWB;
    }
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int sw = (lane >> 1) & 7;
    int xoff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xoff[kk] = ((kk * 2 + lhalf) ^ sw) << 4;
    const int a_base = (wm * WTM + lrow) * ROWB;
    const int b_base = W_OFF + (wn * WTN + lrow) * ROWB;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = g.K / BK;
    // prologue: A(0), W(0), A(1)  (same relative order as the steady state)
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(srcA[i], smem + ldsoff[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(srcW[i], smem + W_OFF + ldsoff[i]);
    if (nt > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(srcA[i] + BK, smem + ASTG + ldsoff[i]);
    }

    int a_slot = 0;                                                   // t % 3
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // tile t visible to all; slots of t-1 are free
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) {
            char* wn_ = smem + W_OFF + ((t + 1) & 1) * WSTG;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(srcW[i] + (int64_t)(t + 1) * BK, wn_ + ldsoff[i]);
        }
        if (t + 2 < nt) {
            const int s2 = a_slot == 0 ? 2 : a_slot - 1;              // (t + 2) % 3
            char* an_ = smem + s2 * ASTG;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(srcA[i] + (int64_t)(t + 2) * BK, an_ + ldsoff[i]);
        }
        mma_ktile<T, TM, TN>(acc, smem + a_slot * ASTG + a_base, smem + (t & 1) * WSTG + b_base, xoff);
        a_slot = a_slot == 2 ? 0 : a_slot + 1;
    }
    staged_epilogue<T, EPI, TM, TN, WTM, WTN>(acc, g, smem, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
}

template <typename KFN>
static int launch_kernel(KFN kfn, bool& attr_set, size_t lds, dim3 grid, dim3 block, const GemmArgs& g, hipStream_t s) {
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { pg_set_error("gemm: set LDS attr (%zu B): %s", lds, hipGetErrorString(e)); return PG_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, block, lds, s, g);
    return pg_check_launch("gemm16");
}

static int finish_args(GemmArgs& g, int BM, int BN) {
    if (g.N % BN != 0 || g.K % BK != 0) { pg_set_error("gemm: N %% %d or K %% 64 != 0 (N=%d K=%d)", BN, g.N, g.K); return PG_EINVAL; }
    g.tilesM = (g.M + BM - 1) / BM;
    g.tilesN = g.N / BN;
    g.ntiles = g.tilesM * g.tilesN;
    if (g.gn <= 0) g.gn = 4;
    if (g.gn > g.tilesN) g.gn = g.tilesN;
    return PG_OK;
}

template <typename T, int BM, int BN, int WM, int WN, bool DIRECT, bool PIPE, int DBG = 0>
static int launch_cfg(const GemmArgs& g0, int epi, hipStream_t s) {
    GemmArgs g = g0;
    int rc = finish_args(g, BM, BN);
    if (rc) return rc;
    const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
    dim3 grid(g.ntiles), block(WM * WN * 64);
#define PG_LAUNCH(E) { static bool a = false; return launch_kernel(gemm_bf16_kernel<T, BM, BN, WM, WN, E, DIRECT, PIPE, DBG>, a, lds, grid, block, g, s); }
    switch (epi) {
        case EPI_QKV: PG_LAUNCH(EPI_QKV)
        case EPI_GELU: PG_LAUNCH(EPI_GELU)
        case EPI_RESID: PG_LAUNCH(EPI_RESID)
        case EPI_PATCH: PG_LAUNCH(EPI_PATCH)
        case EPI_F32: PG_LAUNCH(EPI_F32)
        default: pg_set_error("gemm: bad epilogue %d", epi); return PG_EINVAL;
    }
#undef PG_LAUNCH
}

template <typename T>
static int launch_a3w2(const GemmArgs& g0, int epi, hipStream_t s) {
    GemmArgs g = g0;
    int rc = finish_args(g, 256, 256);
    if (rc) return rc;
    const size_t lds = 5 * 256 * ROWB;                               // 160 KB: the whole LDS of a CU
    dim3 grid(g.ntiles), block(512);
#define PG_LAUNCH(E) { static bool a = false; return launch_kernel(gemm16_a3w2_kernel<T, E>, a, lds, grid, block, g, s); }
    switch (epi) {
        case EPI_QKV: PG_LAUNCH(EPI_QKV)
        case EPI_GELU: PG_LAUNCH(EPI_GELU)
        case EPI_RESID: PG_LAUNCH(EPI_RESID)
        case EPI_PATCH: PG_LAUNCH(EPI_PATCH)
        case EPI_F32: PG_LAUNCH(EPI_F32)
        default: pg_set_error("gemm: bad epilogue %d", epi); return PG_EINVAL;
    }
#undef PG_LAUNCH
}

// EPI_RESID_STAT on the 384 x 256 kernel (round 3): bit 0 = long-K GEMMs (fc2, K >= 2048), bit 1 = short-K ones (out-projection).
// Results are bit-identical either way; this only selects the tile shape.  Env PIGEON_GEMM_RESID6 (A/B).  Measured on the 512-image
// step (profiles/r03/pp6_resid_stat_ab.txt): fc2 2.224 -> 2.152 ms (-3.2 %; W is re-streamed through every XCD's L2 12 instead of 18
// times), out-projection 0.90 -> 0.976 ms (its epilogue is 43 % of a tile, first build): the default takes fc2 only, +0.9 % end to end.
#ifndef PG_DEFAULT_GEMM_RESID6
#define PG_DEFAULT_GEMM_RESID6 1
#endif
static bool resid6_enabled(int K) {
    static int mask = -1;
    if (mask < 0) { const char* e = getenv("PIGEON_GEMM_RESID6"); mask = e ? atoi(e) : PG_DEFAULT_GEMM_RESID6; if (mask < 0) mask = 0; }
    return (mask & (K >= 2048 ? 1 : 2)) != 0;
}
// cost model of the small-batch choice (pg_gemm_launch): microseconds per 64-wide K tile of one tile period
#define PG_MID_US_KT_P6 1.75     // 384 x 256 persistent tile on a mostly idle chip (2.4 GHz)
#define PG_MID_US_KT_PP 1.25     // 256 x 256
#define PG_TAIL_US 36.0          // gemm_tail.hip on a <= 768-row tail, either shape (profiles/r02/gemm_tail.txt, profiles/r06/step_kernel_stats.csv)
#define PG_MID_US_KT_MID 0.44    // 128 x 128 through the 3-stage ring with the producer wave (0.41 measured on fc2's 64 K tiles, profiles/r06/
                                 // gemm_three_sweep_producer.txt; 0.58 - 0.62 while the MFMA waves issued their own DMAs: gemm_mid_sweep.txt)
static bool use_pp6(int variant, int epi, int N, int K) {
    return variant == 56 && pg_gemm_pp6_supported(epi, N, K) && (epi != EPI_RESID_STAT || resid6_enabled(K));
}

// Modelled time (us) of one GEMM launch on `ncu` CUs through kernel `kind` (0 = 384 x 256 persistent, 1 = 256 x 256 persistent,
// 2 = gemm_mid.hip), for batches of up to ~64 images: the routing of pg_gemm_launch below.  A persistent launch is a sequence of
// rounds; a round's tile period grows with the share f of the CUs it keeps busy (the power cap: 2.4 GHz on an idle chip, ~1.7 GHz on
// a full one), c(f) = ci + (cf - ci) f^2 microseconds per 64-wide K tile, plus an epilogue the first round pays in full and the later
// ones partly (the next tile's operands are in flight under it).  Constants fitted to profiles/r06/gemm_three_sweep.txt (the model's
// four GEMM shapes x 1 .. 64 images x the three kernels): the pick is the measured best, or within 0.4 % of it, in all 36 cells.
// (Second session: gemm_mid's two constants refitted to gemm_three_sweep_producer.txt -- its producer wave; the pick is within 9 %
// of the best in every cell of the first file and of the two residual shapes of the second, tests/test_host_cpu.py.)
static double gemm_model_us(int kind, int M, int N, int K, int epi, int ncu) {
    const bool resid = epi == EPI_RESID || epi == EPI_RESID_STAT;
    const bool gelu = epi == EPI_GELU || epi == EPI_GELU_LN;
    const double kt = K / 64;
    if (kind == 2) {
        const int64_t tiles = (int64_t)((M + 127) / 128) * (N / 128);
        return (double)((tiles + ncu - 1) / ncu) * (kt * PG_MID_US_KT_MID + (resid ? 8.0 : 5.5));
    }
    const int bm = kind == 0 ? 384 : 256;
    const double ci = kind == 0 ? 1.6 : 1.1, cf = kind == 0 ? 3.0 : 1.8;
    const double e1 = kind == 0 ? (resid ? 7.0 : (gelu ? 9.0 : 6.0)) : (resid ? 12.0 : (gelu ? 9.0 : 7.0));
    const double e2 = kind == 0 ? 3.0 : 6.0;
    const int64_t tiles = (int64_t)((M + bm - 1) / bm) * (N / 256);
    const int64_t full = tiles / ncu;
    const double f = (double)(tiles % ncu) / (double)ncu;
    double t = 0.0;
    if (full > 0) t += kt * cf + e1 + (double)(full - 1) * (kt * cf + e2);
    if (f > 0.0) t += kt * (ci + (cf - ci) * f * f) + (full > 0 ? e2 : e1);
    return t;
}
#define PG_ROUTE_MAX_ROWS 40000   // the routing model is fitted up to 64 images (36 928 token rows); above that a variant means its kernel
static int route_max_rows() {     // (env PIGEON_GEMM_ROUTE_MAX_ROWS: experiments beyond the fitted range)
    static int v = -1;
    if (v < 0) { const char* e = getenv("PIGEON_GEMM_ROUTE_MAX_ROWS"); v = e ? atoi(e) : PG_ROUTE_MAX_ROWS; if (v < 0) v = 0; }
    return v;
}

// Which kernel a launch of the persistent variants takes: 0 = 384 x 256 persistent, 1 = 256 x 256 persistent, 2 = gemm_mid.hip; -1 = not
// a persistent variant / shape (the caller's dispatch decides).  A pure function of the shape, the knobs and the CU count.
static int gemm_route(int variant, int epi, int M, int N, int K) {
    const bool six = use_pp6(variant, epi, N, K);
    const bool pp = !six && (variant == 56 || (variant >= 30 && variant < 50)) && N % 256 == 0 && K % 128 == 0;
    if (!six && !pp) return -1;
    const int own = six ? 0 : 1;
    if (!(pg_gemm_mid_on() && M <= route_max_rows() && epi != EPI_PATCH && epi >= EPI_QKV && epi <= EPI_GELU_LN)) return own;
    int ncu = pg_num_cus();
    if (pg_gemm_block_cap() > 0 && pg_gemm_block_cap() < ncu) ncu = pg_gemm_block_cap();
    const double t_own = gemm_model_us(own, M, N, K, epi, ncu);
    const double t_pp = (six && pg_gemm_route_pp256()) ? gemm_model_us(1, M, N, K, epi, ncu) : 1e30;
    const double t_mid = pg_gemm_mid_supported(epi, N, K) ? gemm_model_us(2, M, N, K, epi, ncu) : 1e30;
    // the 256 x 256 kernel has to win by a margin: where the model calls it level with the 384 x 256 kernel (64 images) or with
    // gemm_mid (2 and 8 images) the encoder measured 1 - 3 % SLOWER with it in place (profiles/r06/latency_route_ab.txt: a
    // launch in a forward is not a launch in a loop of its own); where it wins by more, the encoder gains 3 - 9 %
    const bool pp_wins = t_pp < 0.90 * t_own && t_pp < 0.90 * t_mid;
    if (pp_wins) return 1;
    return t_mid < t_own ? 2 : own;
}
// (exported for the host-logic tests and tools: no launch, no device work)
extern "C" int pg_gemm_route(int variant, int epi, int M, int N, int K, int* kind) {
    if (!kind || M <= 0 || N <= 0 || K <= 0) { pg_set_error("gemm_route: bad argument"); return PG_EINVAL; }
    *kind = gemm_route(variant ? variant : pg_default_gemm_variant(), epi, M, N, K);
    return PG_OK;
}

template <typename T>
static int gemm_dispatch(GemmArgs& g, int epi, int variant, hipStream_t s) {
    switch (variant) {
        // the product library carries ONE one-tile-per-block kernel (variant 8: the fallback for shapes the persistent kernel
        // does not take, and the bit-exact reference the persistent kernels are tested against); the older tilings and the
        // timing-only ablations (21..23: wrong results by construction) exist only in the -DPIGEON_ABLATIONS tools build
        case 8: g.gn = 1 << 20; return launch_cfg<T, 256, 256, 2, 4, false, true, 4>(g, epi, s);   // 256x256, N-fastest raster, interleaved DMA
#ifdef PIGEON_ABLATIONS
        case 1: return launch_cfg<T, 256, 256, 2, 4, false, false>(g, epi, s);
        case 2: return launch_cfg<T, 128, 128, 2, 2, false, true>(g, epi, s);
        case 3: return launch_cfg<T, 256, 128, 4, 2, false, true>(g, epi, s);
        case 4: return launch_cfg<T, 256, 256, 2, 4, false, true>(g, epi, s);
        case 5: return launch_a3w2<T>(g, epi, s);
        case 6: g.gn = 1 << 20; return launch_cfg<T, 256, 256, 2, 4, false, true>(g, epi, s);   // variant 4, N-fastest raster
        case 7: g.gn = 1 << 20; return launch_a3w2<T>(g, epi, s);                                  // variant 5, N-fastest raster
        case 11: return launch_cfg<T, 256, 256, 2, 4, true, false>(g, epi, s);
        case 21: g.gn = 1 << 20; return launch_cfg<T, 256, 256, 2, 4, false, true, 1>(g, epi, s);   // no in-loop DMA
        case 22: g.gn = 1 << 20; return launch_cfg<T, 256, 256, 2, 4, false, true, 2>(g, epi, s);   // no ds_reads
        case 23: g.gn = 1 << 20; return launch_cfg<T, 256, 256, 2, 4, false, true, 3>(g, epi, s);   // neither
#endif
        default: pg_set_error("gemm: variant %d is not part of this build (product variants: 8, 33, 36, 56, 70; the rest needs the "
                              "-DPIGEON_ABLATIONS tools build, python -m pigeon_amd.build --dev)", variant); return PG_EINVAL;
    }
}

#ifdef PIGEON_ABLATIONS
static void* g_dbg_ts = nullptr;
// tools build: arm (buf != null) / disarm the PG_TS time stamps of the persistent kernels; buf = 2 * 16 * 8 * 12 uint64 on the device
extern "C" int pg_dbg_timestamps(void* buf) { g_dbg_ts = buf; return PG_OK; }
#endif

int pg_gemm_launch(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldc,
                   int M, int N, int K, int epi, float qscale, int qcols, const float* aux, int variant,
                   hipStream_t s, const PgGemmExtra* extra) {
    if (M <= 0) return PG_OK;
    GemmArgs g;
    g.A = (const uint16_t*)A; g.lda = lda; g.W = (const uint16_t*)W; g.ldw = ldw > 0 ? ldw : K; g.bias = bias; g.out = out; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.qscale = qscale; g.qcols = qcols; g.aux = aux;
    g.tilesM = 0; g.tilesN = 0; g.ntiles = 0; g.part_tiles = 0; g.gn = 0; g.stagger = 0; g.xcd_stagger_ticks = 0;
    if (extra) g.ex = *extra;
    if (g.ex.parts > 1) {                                    // several products in one launch: 256 x 256 persistent kernel, EPI_F32 only
        if (epi != EPI_F32 || N % 256 != 0 || K % 128 != 0) { pg_set_error("gemm: parts > 1 needs EPI_F32, N %% 256 == 0, K %% 128 == 0"); return PG_EINVAL; }
        return pg_gemm_pp_launch(dtype, g, epi, 36, s);
    }
    if ((epi == EPI_GELU || epi == EPI_RESID || epi >= EPI_RESID_STAT) && !bias) { pg_set_error("gemm: epilogue %d needs a bias", epi); return PG_EINVAL; }
    if (epi == EPI_RESID_STAT && (!g.ex.x16 || !g.ex.statpart || g.ex.ldx != ldc)) { pg_set_error("gemm: EPI_RESID_STAT needs x16 / statpart and ldx == ldc"); return PG_EINVAL; }
    if ((epi == EPI_QKV_LN || epi == EPI_GELU_LN) && (!g.ex.colsum || !g.ex.rowstat)) { pg_set_error("gemm: LN epilogue needs colsum / rowstat"); return PG_EINVAL; }
    if (epi == EPI_PATCH && !aux) { pg_set_error("gemm: patch epilogue needs aux"); return PG_EINVAL; }
    if (epi == EPI_GELU_X3 && (ldc != 3 * (int64_t)N || dtype != PG_DTYPE_F16 || N % 256 != 0 || K % 128 != 0)) {
        pg_set_error("gemm: EPI_GELU_X3 writes the fp16 triple [M][3N]: ldc == 3 N, fp16 operands, N %% 256 == 0, K %% 128 == 0 (ldc=%lld N=%d K=%d)",
                     (long long)ldc, N, K);
        return PG_EINVAL;
    }
    if (epi < EPI_QKV || epi > EPI_GELU_X3) { pg_set_error("gemm: bad epilogue %d", epi); return PG_EINVAL; }
    if ((lda % 8) || (ldc % 8) || (qcols % 8) || (g.ldw % 8) || g.ldw < K) { pg_set_error("gemm: lda/ldw/ldc/qcols must be multiples of 8, ldw >= K"); return PG_EINVAL; }
    if (variant == 0) variant = pg_default_gemm_variant();
    {
        // XCD stagger (gemm_epi.h): total spread = fraction x estimated tile period.  Tile periods measured on MI355X
        // (profiles/r02): 256x256 tiles 25 us + 1.63 us per 64-wide K tile (fp32 residual epilogues), 384x256 tiles
        // 8 us (+4 us with the GELU) + 2.44 us per K tile.
        const float f = pg_gemm_stagger_fraction();
        if (f > 0.f && M >= 256 * 64) {
            const bool six = use_pp6(variant, epi, N, K);
            const float period_us = six ? ((epi == EPI_GELU || epi == EPI_GELU_LN ? 12.f : 8.f) + 2.44f * (K / 64))
                                        : ((epi == EPI_RESID || epi == EPI_RESID_STAT ? 25.f : 10.f) + 1.63f * (K / 64));
            g.xcd_stagger_ticks = (int)(f * period_us * 100.f);              // 100 ticks per us
        }
    }
    if (g.ex.stat_rows <= 0) g.ex.stat_rows = M;
#ifdef PIGEON_ABLATIONS
    if (g_dbg_ts && epi != EPI_PATCH) { g.aux = (const float*)g_dbg_ts; g.stagger = -7; }
    else if (epi == EPI_RESID_STAT) { static const bool abl = getenv("PIGEON_EPI_ABL") != nullptr; if (abl) g.stagger = -11; }
#endif
    if (variant == 71) {                                     // the whole problem through the 128 x 128 kernel of small batches (tests, tools)
        if (!pg_gemm_mid_supported(epi, N, K)) { pg_set_error("gemm: variant 71 (gemm_mid) does not support epi=%d N=%d K=%d", epi, N, K); return PG_EINVAL; }
        return pg_gemm_mid_launch(dtype, g, epi, s);
    }
    if (variant == 70) {                                     // the whole problem through the small-tile tail kernel (tests, tools)
        if (!pg_gemm_tail_supported(epi, N, K)) { pg_set_error("gemm: variant 70 (gemm_tail) does not support epi=%d N=%d K=%d", epi, N, K); return PG_EINVAL; }
        return pg_gemm_tail_launch(dtype, g, epi, 0, s);
    }
    {
        // Small and middle batches (round 6).  The product variant (56) means "384 x 256 tiles where they exist, 256 x 256 elsewhere",
        // chosen for the 512-image step, where a launch is 12 - 50 rounds.  Up to ~64 images a launch is 1 - 7 rounds and what
        // decides is how the row panels of a tile shape fill whole rounds of the CUs: one panorama (2308 rows) is 7 panels of 384
        // or 10 of 256 or 19 of 128; 16 images leave the 384-row kernel a second round with 44 of 256 CUs busy.  All three kernels
        // produce the same bits for a row (tests/test_gpu_parity.py), so the choice is a timing decision, taken by gemm_model_us
        // above (pg_tune_gemm_mid(0) / PIGEON_GEMM_MID=0: the variant's own kernel, always).  Measured
        // (profiles/r06/gemm_three_sweep.txt, latency_route.txt): 16 images QKV 79.7 -> 64.6 us, fc2 123.5 -> 94.6; one panorama fc1
        // 38.7 -> 31.7.
        const int kind = gemm_route(variant, epi, M, N, K);
        if (kind == 2) return pg_gemm_mid_launch(dtype, g, epi, s);
        if (kind == 1 && use_pp6(variant, epi, N, K)) return pg_gemm_pp_launch(dtype, g, epi, 36, s);
    }
    {
        // Tail split: if the tiles do not fill the persistent kernel's last round and the rows beyond the last whole round are
        // few, the persistent kernel gets the rows that make whole rounds and a small-tile kernel the rest.  All three produce
        // the same bits for a row, so the cut changes timing only.  WHERE to cut is round 2's measurement (vit.hip, the MIN_K /
        // MIN_N thresholds: fc2 and fc1; for out-projection and QKV the extra launch costs what the 8-tile last round did --
        // measured again in round 6 with the cheaper tail kernel below, profiles/r06/tail_mid_ab.txt: still nothing end to end).
        // WHICH kernel takes the tail is a cost model: gemm_tail.hip (32 x 64 one-wave tiles) needs ~36 us for the benchmark
        // batch's 512 rows whatever the shape; gemm_mid.hip (round 6) does a K = 1024 tail in 15 - 19 us
        // (profiles/r06/gemm_mid_sweep.txt, the n = 1 column) and a K = 4096 one in 44: fc1's tail goes through it (the fc1
        // launch pair 2.257 -> 2.237 ms, 0.4347 -> 0.4384 of the MFMA peak on one box), fc2's stays.
        const bool six = use_pp6(variant, epi, N, K);
        const bool pp = !six && (variant == 56 || (variant >= 30 && variant < 50)) && N % 256 == 0 && K % 128 == 0;
        const int tail_max = pg_gemm_tail_rows();
        if ((six || pp) && tail_max > 0) {
            const int bm = six ? 384 : 256;
            int ncu = pg_num_cus();
            if (pg_gemm_block_cap() > 0 && pg_gemm_block_cap() < ncu) ncu = pg_gemm_block_cap();
            const int tilesN = N / 256;
            const int64_t ntiles = (int64_t)((M + bm - 1) / bm) * tilesN;
            const int64_t rounds = ntiles / ncu;
            if (rounds >= 1 && ntiles % ncu != 0) {
                const int64_t m_main = (rounds * ncu / tilesN) * bm;         // row panels that fit into `rounds` whole rounds
                if (m_main > 0 && m_main < M && M - m_main <= tail_max) {
                    const bool resid = epi == EPI_RESID || epi == EPI_RESID_STAT;
                    const double kt = K / 64;
                    // the round the cut removes (constants of the small-batch model above: a lower bound on a busy chip)
                    const double t_round = six ? kt * PG_MID_US_KT_P6 + (resid ? 14.0 : 9.0) : kt * PG_MID_US_KT_PP + (resid ? 20.0 : 8.0);
                    const int64_t tiles_m = (int64_t)((M - m_main + 127) / 128) * (N / 128);
                    const double t_mid = (double)((tiles_m + ncu - 1) / ncu) * (kt * PG_MID_US_KT_MID + (resid ? 6.0 : 5.0)) + 3.0;   // + a launch
                    const bool cut = K >= pg_gemm_tail_min_k() || N >= pg_gemm_tail_min_n();
                    const bool by_mid = cut && pg_gemm_mid_on() && pg_gemm_mid_supported(epi, N, K) && t_mid < PG_TAIL_US && t_mid < t_round;
                    const bool by_tail = cut && !by_mid && pg_gemm_tail_supported(epi, N, K);
                    if (by_mid || by_tail) {
                        GemmArgs gm = g;
                        gm.M = (int)m_main;
                        const int rc = six ? pg_gemm_pp6_launch(dtype, gm, epi, s) : pg_gemm_pp_launch(dtype, gm, epi, variant == 56 ? 36 : variant, s);
                        if (rc != PG_OK) return rc;
                        return by_mid ? pg_gemm_mid_launch(dtype, g, epi, s, (int)m_main) : pg_gemm_tail_launch(dtype, g, epi, (int)m_main, s);
                    }
                }
            }
        }
    }
    if (variant == 64) {                                     // one-wave-per-SIMD persistent kernel (gemm_w4.hip: archived, branch archive/kernel-generations-r04)
#if defined(PIGEON_ABLATIONS) && defined(PIGEON_OLD_GENERATIONS)
        if (pg_gemm_w4_supported(epi, N, K)) return pg_gemm_w4_launch(dtype, g, epi, s);
        variant = 36;
#else
        pg_set_error("gemm: variant 64 (gemm_w4) is experimental and not part of the product library (python -m pigeon_amd.build --dev)");
        return PG_EINVAL;
#endif
    }
    if (variant == 56) {                                     // 384 x 256 tiles where they exist, the product kernel elsewhere
        if (use_pp6(variant, epi, N, K)) return pg_gemm_pp6_launch(dtype, g, epi, s);
        variant = 36;
    }
    if (variant >= 30 && variant < 50) {
        // the persistent kernel needs N % 256 == 0 and an even number of K tiles; everything the model launches qualifies
        if (N % 256 == 0 && K % 128 == 0) return pg_gemm_pp_launch(dtype, g, epi, variant, s);
        variant = 8;
    }
    if (epi >= EPI_RESID_STAT) { pg_set_error("gemm: epilogue %d exists only in the persistent kernel (variants 30..49, N %% 256 == 0, K %% 128 == 0)", epi); return PG_EINVAL; }
    if (dtype == PG_DTYPE_F16) return gemm_dispatch<T_F16>(g, epi, variant, s);
    if (dtype == PG_DTYPE_BF16) return gemm_dispatch<T_BF16>(g, epi, variant, s);
    pg_set_error("gemm: operand dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16 (got %d)", dtype);
    return PG_EINVAL;
}
