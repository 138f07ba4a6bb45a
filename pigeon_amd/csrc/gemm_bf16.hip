// gemm_bf16.hip -- C[M,N] = A[M,K] * W[N,K]^T, bf16 operands, fp32 accumulation on MFMA, fused epilogues.
//
// Replaces the nn.Linear / Conv2d library GEMMs the reference reaches through transformers
// (modeling_clip.py CLIPAttention.q/k/v/out_proj, CLIPMLP.fc1/fc2, CLIPVisionEmbeddings.patch_embedding),
// SURVEY.md section 2c rows K1,K4,K6,K7,K8 = 91% of the path's FLOPs.
//
// Design (gfx950 / CDNA4, wave64):
//   * v_mfma_f32_32x32x16_bf16; both operands are K-contiguous in memory ([M,K] activations, [N,K] weights as
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

#define BK 64
#define ROWB 128   // bytes per LDS row (BK bf16)

struct GemmArgs {
    const uint16_t* A; int64_t lda;
    const uint16_t* W;            // [N][K]
    const float* bias;            // [N] or null
    void* out; int64_t ldc;
    int M, N, K;
    float qscale; int qcols;
    const float* aux;             // epi 3: position embedding [577][N]
    int tilesN, ntiles;
};

__device__ __forceinline__ void glds16(const void* gptr, void* lds_base_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_base_uniform, 16, 0, 0);
}

__device__ __forceinline__ float quick_gelu(float v) { return v / (1.0f + __expf(-1.702f * v)); }

// Apply the epilogue to 4 consecutive columns [col, col+4) of one output row (fp32-out epilogues).
template <int EPI>
__device__ __forceinline__ void epi_store_f32x4(const GemmArgs& g, int row, int col, f32x4 v, const f32x4& b4) {
    if (EPI == EPI_RESID) {
        float* p = (float*)g.out + (int64_t)row * g.ldc + col;
        f32x4 x = *(const f32x4*)p;
        x += v + b4;
        *(f32x4*)p = x;
    } else if (EPI == EPI_PATCH) {
        const int img = row / VIT_PATCHES, p = row - img * VIT_PATCHES;
        float* o = (float*)g.out + ((int64_t)img * VIT_TOKENS + 1 + p) * g.ldc + col;
        const f32x4 pos = *(const f32x4*)(g.aux + (int64_t)(1 + p) * g.N + col);
        *(f32x4*)o = v + pos;
    } else {  // EPI_F32
        *(f32x4*)((float*)g.out + (int64_t)row * g.ldc + col) = v + b4;
    }
}

// bf16-out epilogues on 8 consecutive columns.
template <int EPI>
__device__ __forceinline__ void epi_store_bf16x8(const GemmArgs& g, int row, int col, f32x4 lo, f32x4 hi,
                                                 const f32x4& b_lo, const f32x4& b_hi) {
    lo += b_lo; hi += b_hi;
    if (EPI == EPI_QKV) {
        if (col < g.qcols) { lo *= g.qscale; hi *= g.qscale; }   // qcols is a multiple of 8
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = quick_gelu(lo[e]); hi[e] = quick_gelu(hi[e]); }
    }
    u32x4 pk;
    pk[0] = pack_bf16x2(lo[0], lo[1]); pk[1] = pack_bf16x2(lo[2], lo[3]);
    pk[2] = pack_bf16x2(hi[0], hi[1]); pk[3] = pack_bf16x2(hi[2], hi[3]);
    *(u32x4*)((uint16_t*)g.out + (int64_t)row * g.ldc + col) = pk;
}

template <int EPI>
__device__ __forceinline__ void epi_store_scalar(const GemmArgs& g, int row, int col, float v) {
    if (EPI == EPI_QKV) {
        if (g.bias) v += g.bias[col];
        if (col < g.qcols) v *= g.qscale;
        ((uint16_t*)g.out)[(int64_t)row * g.ldc + col] = f32_to_bf16_bits(v);
    } else if (EPI == EPI_GELU) {
        v = quick_gelu(v + g.bias[col]);
        ((uint16_t*)g.out)[(int64_t)row * g.ldc + col] = f32_to_bf16_bits(v);
    } else if (EPI == EPI_RESID) {
        float* p = (float*)g.out + (int64_t)row * g.ldc + col;
        *p = *p + (v + g.bias[col]);
    } else if (EPI == EPI_PATCH) {
        const int img = row / VIT_PATCHES, p = row - img * VIT_PATCHES;
        float* o = (float*)g.out + ((int64_t)img * VIT_TOKENS + 1 + p) * g.ldc + col;
        *o = v + g.aux[(int64_t)(1 + p) * g.N + col];
    } else {
        if (g.bias) v += g.bias[col];
        ((float*)g.out)[(int64_t)row * g.ldc + col] = v;
    }
}

// BM x BN block tile, WM x WN waves, each wave owns (BM/WM) x (BN/WN) = TM x TN MFMA tiles of 32x32.
template <int BM, int BN, int WM, int WN, int EPI, bool DIRECT>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int GROUPS = (BM + BN) / 8;        // 8-row DMA groups per stage
    constexpr int LPW = GROUPS / NW;             // DMA instructions per wave per stage
    static_assert(GROUPS % NW == 0, "stage must split evenly over waves");
    static_assert(BM % 16 == 0 && BN % 16 == 0, "tile alignment");
    constexpr bool BF16_OUT = (EPI == EPI_QKV || EPI == EPI_GELU);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int wg = xcd_remap(blockIdx.x, g.ntiles);
    const int tm = wg / g.tilesN, tn = wg - tm * g.tilesN;
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
            src[i] = g.W + (int64_t)row * g.K + c * 8;
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
        if (t + 1 < nt) {
            char* nxt = smem + (cur ^ 1) * STAGE;
#pragma unroll
            for (int i = 0; i < LPW; ++i) glds16(src[i] + (int64_t)(t + 1) * BK, nxt + ldsoff[i]);
        }
        const char* sb = smem + cur * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *(const bf16x8*)(sb + a_base + i * 32 * ROWB + xoff[kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *(const bf16x8*)(sb + b_base + j * 32 * ROWB + xoff[kk]);
            // swapped operands: D[n][m] -> lane owns row m = lane&31, columns n = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
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
                    if (row < g.M) epi_store_scalar<EPI>(g, row, col, acc[i][j][r]);
                }
        }
        return;
    }

    // ---- LDS-staged epilogue: per wave, one 32 x WTN fp32 slab at a time ----
    constexpr int ROWPF = WTN + 4;                           // padded slab row, floats (272 B for WTN=64)
    float* slab = (float*)(smem + wave * (32 * ROWPF * 4));
    __syncthreads();                                         // every wave is done with the K-loop buffers
    if (BF16_OUT) {
        constexpr int LPR = WTN / 8, RPI = 64 / LPR, ITS = 32 / RPI;
        const int rr = lane / LPR, cc = (lane % LPR) * 8;
        const int col = n0 + wn * WTN + cc;
        f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) { b_lo = *(const f32x4*)(g.bias + col); b_hi = *(const f32x4*)(g.bias + col + 4); }
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
                const f32x4 lo = *(const f32x4*)(slab + r * ROWPF + cc);
                const f32x4 hi = *(const f32x4*)(slab + r * ROWPF + cc + 4);
                const int row = m0 + wm * WTM + i * 32 + r;
                if (row < g.M) epi_store_bf16x8<EPI>(g, row, col, lo, hi, b_lo, b_hi);
            }
            __syncthreads();
        }
    } else {
        constexpr int LPR = WTN / 4, RPI = 64 / LPR, ITS = 32 / RPI;
        const int rr = lane / LPR, cc = (lane % LPR) * 4;
        const int col = n0 + wn * WTN + cc;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) b4 = *(const f32x4*)(g.bias + col);
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
                const f32x4 v = *(const f32x4*)(slab + r * ROWPF + cc);
                const int row = m0 + wm * WTM + i * 32 + r;
                if (row < g.M) epi_store_f32x4<EPI>(g, row, col, v, b4);
            }
            __syncthreads();
        }
    }
}

template <int BM, int BN, int WM, int WN, bool DIRECT>
static int launch_cfg(const GemmArgs& g0, int epi, hipStream_t s) {
    GemmArgs g = g0;
    if (g.N % BN != 0 || g.K % BK != 0) { pg_set_error("gemm: N %% %d or K %% 64 != 0 (N=%d K=%d)", BN, g.N, g.K); return PG_EINVAL; }
    const int tilesM = (g.M + BM - 1) / BM;
    g.tilesN = g.N / BN;
    g.ntiles = tilesM * g.tilesN;
    const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
    dim3 grid(g.ntiles), block(WM * WN * 64);
#define PG_LAUNCH(E)                                                                                    \
    {                                                                                                   \
        auto kfn = gemm_bf16_kernel<BM, BN, WM, WN, E, DIRECT>;                                         \
        static bool attr_set = false;                                                                   \
        if (!attr_set) {                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) { pg_set_error("gemm: set LDS attr: %s", hipGetErrorString(e)); return PG_EHIP; } \
            attr_set = true;                                                                            \
        }                                                                                               \
        hipLaunchKernelGGL(kfn, grid, block, lds, s, g);                                                \
    }
    switch (epi) {
        case EPI_QKV: PG_LAUNCH(EPI_QKV) break;
        case EPI_GELU: PG_LAUNCH(EPI_GELU) break;
        case EPI_RESID: PG_LAUNCH(EPI_RESID) break;
        case EPI_PATCH: PG_LAUNCH(EPI_PATCH) break;
        case EPI_F32: PG_LAUNCH(EPI_F32) break;
        default: pg_set_error("gemm: bad epilogue %d", epi); return PG_EINVAL;
    }
#undef PG_LAUNCH
    return pg_check_launch("gemm_bf16");
}

int pg_gemm_launch(const void* A, int64_t lda, const void* W, const float* bias, void* out, int64_t ldc,
                   int M, int N, int K, int epi, float qscale, int qcols, const float* aux, int variant,
                   hipStream_t s) {
    if (M <= 0) return PG_OK;
    GemmArgs g;
    g.A = (const uint16_t*)A; g.lda = lda; g.W = (const uint16_t*)W; g.bias = bias; g.out = out; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.qscale = qscale; g.qcols = qcols; g.aux = aux; g.tilesN = 0; g.ntiles = 0;
    if ((epi == EPI_GELU || epi == EPI_RESID) && !bias) { pg_set_error("gemm: epilogue %d needs a bias", epi); return PG_EINVAL; }
    if (epi == EPI_PATCH && !aux) { pg_set_error("gemm: patch epilogue needs aux"); return PG_EINVAL; }
    if ((lda % 8) || (ldc % 8) || (qcols % 8)) { pg_set_error("gemm: lda/ldc/qcols must be multiples of 8"); return PG_EINVAL; }
    if (variant == 0) variant = pg_default_gemm_variant();
    switch (variant) {
        case 1: return launch_cfg<256, 256, 2, 4, false>(g, epi, s);
        case 2: return launch_cfg<128, 128, 2, 2, false>(g, epi, s);
        case 3: return launch_cfg<256, 128, 4, 2, false>(g, epi, s);
        case 11: return launch_cfg<256, 256, 2, 4, true>(g, epi, s);
        case 12: return launch_cfg<128, 128, 2, 2, true>(g, epi, s);
        default: pg_set_error("gemm: unknown variant %d", variant); return PG_EINVAL;
    }
}
