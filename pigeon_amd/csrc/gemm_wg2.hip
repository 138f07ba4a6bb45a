// gemm_wg2.hip -- EXPERIMENTAL, tools build only (-DPIGEON_ABLATIONS, variant 72).  NOT YET RUN ON A GPU (written at the end of
// round 2 after the GPU budget was spent): it compiles, its register / LDS budget is checked by tools/asm_audit.py, nothing else
// is claimed.  First thing to do with it: tools/gemm_pp_check.py --variants 72 (bit-compare against variant 36) on the dev library.
//
// Why: the time stamps of round 2 (tools/epi_timeline.py) show the persistent 8-wave kernels spending 15 % of
// the benchmark step in epilogues with idle matrix pipes -- the residual epilogue is bound by what one CU can pull from HBM
// (~50 GB/s), the 16-bit ones by VALU issue -- and there is no register room for a second accumulator set.  The structure that can
// hide an epilogue is TWO INDEPENDENT WORKGROUPS PER CU: while one stores, the other owns the matrix pipe, with no choreography
// between them (the hardware interleaves the two waves of a SIMD).
//
//   workgroup  = 4 waves (one per SIMD), block tile 128 x 256, wave tile 128 x 64 = 8 x 4 blocks of v_mfma_f32_16x16x32
//                (128 accumulator registers, <= 256 VGPRs in all so that two workgroups fit a CU)
//   K tile     = 32 (ONE k-step), LDS rows of 64 bytes; stage = A 128 rows + W 256 rows = 24 KB, THREE stages = 72 KB per
//                workgroup (two workgroups: 144 of 160 KB); the epilogue slabs (4 x 8.5 KB) overlay the stages
//   pipeline   = free running: per K tile { wait for my DMAs of this tile; barrier; issue the DMAs of tile + 2; twelve fragment
//                reads, then 32 MFMAs in two halves }, one barrier per 32 MFMAs per wave
//   operands   = direct-to-LDS DMA, one instruction = 16 rows x 64 bytes; bank swizzle on the source address: the 16-byte chunk c
//                of row r sits in slot c ^ ((r >> 2) & 1) (rows r and r + 4 share banks with a 64-byte pitch)
//   price      = the weight panel is fetched per 128 instead of 256 (384) rows: +50 % (+80 %) operand bytes per flop.  Whether the
//                hidden epilogue pays for that is the experiment.
//
// To check when it runs: the epilogue uses the packed-fp32 forms (ln_fold4 / quick_gelu4), which share the matrix pipe -- fine in
// the 8-wave kernels, whose epilogues have the CU to themselves, but here the other workgroup's MFMAs run next to them (37 instead
// of 5 cycles per packed instruction in tools/pipe_rate.hip); the scalar forms give the same bits.
//
// Same MFMA chain per output element as gemm_pp.hip (k-steps ascending, first step onto 0) and the epilogue expressions of
// gemm_tail.hip / pp_epilogue: results are meant to be bit-identical to variants 36 / 56 / 70.
#include "gemm_epi.h"

#ifdef PIGEON_ABLATIONS

namespace {

constexpr int G2_BM = 128, G2_BN = 256, G2_BK = 32;
constexpr int G2_ROWB = 64;                                 // bytes per LDS row (32 halfs)
constexpr int G2_A_BYTES = G2_BM * G2_ROWB;                 // 8 KB
constexpr int G2_STAGE = (G2_BM + G2_BN) * G2_ROWB;         // 24 KB
constexpr int G2_NSTAGE = 3;
constexpr int G2_LDS = G2_NSTAGE * G2_STAGE;                // 72 KB
constexpr int G2_ROWPF = 64 + 4;                            // slab row in floats
constexpr int G2_SLAB_BYTES = 32 * G2_ROWPF * 4;            // 8704 B per wave, 4 waves = 34 KB <= 72 KB
constexpr int G2_NDMA = 6;                                  // DMA instructions per wave per stage: 2 of A's 8 groups, 4 of W's 16

typedef __attribute__((address_space(3))) void g2_lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t g2_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void g2_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_uniform, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (g2_lds_void*)lds_wave_uniform, 16, voff, soff, 0, 0);
}

struct Tile2 {
    __amdgpu_buffer_rsrc_t ra, rw;
    int m0, n0;
};

// N-fastest raster inside bands of 8 row panels (the two workgroups of a CU and their neighbours share weight panels in L2)
__device__ __forceinline__ Tile2 make_tile2(const GemmArgs& g, int L) {
    Tile2 c;
    const int band_sz = 8 * g.tilesN;
    const int band = L / band_sz, rem = L - band * band_sz;
    const int gm = min(8, g.tilesM - band * 8);
    const int tn = rem / gm, im = rem - tn * gm;
    c.m0 = (band * 8 + im) * G2_BM; c.n0 = tn * G2_BN;
    int rows = g.M - c.m0; rows = rows > G2_BM ? G2_BM : rows;
    rows = __builtin_amdgcn_readfirstlane(rows);            // keep the descriptor in SGPRs (tools/asm_audit.py)
    c.ra = g2_rsrc(g.A + (int64_t)c.m0 * g.lda, (uint32_t)rows * (uint32_t)g.lda * 2u);
    c.rw = g2_rsrc(g.W + (int64_t)c.n0 * g.ldw, (uint32_t)G2_BN * (uint32_t)g.ldw * 2u);
    return c;
}

// One stage: wave w fetches A groups 2w, 2w + 1 and W groups 4w .. 4w + 3 (a group = 16 rows x 64 bytes = one instruction).
// voffA / voffW: the lane's offset inside group 0 (row lane >> 2, swizzled chunk); group advance = 16 rows.  The K advance rides
// in the SGPR offset (not bounds-checked; the M tail is guarded by the row part of the VGPR offset).
__device__ __forceinline__ void issue_dma2(const Tile2& c, char* stage, int wave, int voffA, int voffW, int stepA, int stepW, int soff) {
#pragma unroll
    for (int d = 0; d < 2; ++d) g2_dma16(c.ra, stage + (2 * wave + d) * 1024, voffA + (2 * wave + d) * stepA, soff);
#pragma unroll
    for (int d = 0; d < 4; ++d) g2_dma16(c.rw, stage + G2_A_BYTES + (4 * wave + d) * 1024, voffW + (4 * wave + d) * stepW, soff);
}

typedef f32x4 Acc2[8][4];

template <int CTRL>
__device__ __forceinline__ float g2_dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float g2_row8_sum(float v) {       // gemm_pp.hip row8_sum
    v += g2_dpp_mov<0xB1>(v);
    v += g2_dpp_mov<0x4E>(v);
    v += g2_dpp_mov<0x141>(v);
    return v;
}

template <int EPI> constexpr bool g2_out16() { return EPI == EPI_QKV || EPI == EPI_GELU || EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }
template <int EPI> constexpr bool g2_ln() { return EPI == EPI_QKV_LN || EPI == EPI_GELU_LN; }

// Epilogue of one 32 x 64 slab (rows row0 .., columns col0 ..) sitting in the wave's LDS slab: gemm_tail.hip's, verbatim.
template <typename T, int EPI>
__device__ __forceinline__ void slab_epilogue2(const GemmArgs& g, const float* slab, int lane, int row0, int col0) {
    constexpr bool OUT16 = g2_out16<EPI>();
    constexpr bool LN = g2_ln<EPI>();
    constexpr bool STAT = (EPI == EPI_RESID_STAT);
    constexpr bool RESID = (EPI == EPI_RESID || EPI == EPI_RESID_STAT);
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
        f32x4 lo = *(const f32x4*)(slab + r * G2_ROWPF + cc);
        f32x4 hi = *(const f32x4*)(slab + r * G2_ROWPF + cc + HOFF);
        if constexpr (OUT16) {
            if constexpr (LN) {
                const u32x2 rs = *(const u32x2*)(g.ex.rowstat + (int64_t)row * 2);
                float rstd, mrs;
                asm("v_mov_b32 %0, %1" : "=v"(rstd) : "v"(rs[0]));
                asm("v_mov_b32 %0, %1" : "=v"(mrs) : "v"(rs[1]));
                lo = ln_fold4(lo, rstd, mrs, s_lo, b_lo);
                hi = ln_fold4(hi, rstd, mrs, s_hi, b_hi);
            } else {
                lo += b_lo; hi += b_hi;
            }
            if constexpr (EPI == EPI_QKV || EPI == EPI_QKV_LN) {
                if (col0 < g.qcols) { lo *= qsc; hi *= qsc; }
            } else {
                lo = quick_gelu4(lo); hi = quick_gelu4(hi);
            }
            u32x4 pk;
            pk[0] = pack16x2<T>(lo[0], lo[1]); pk[1] = pack16x2<T>(lo[2], lo[3]);
            pk[2] = pack16x2<T>(hi[0], hi[1]); pk[3] = pack16x2<T>(hi[2], hi[3]);
            *(u32x4*)((uint16_t*)g.out + (int64_t)row * g.ldc + col) = pk;
        } else if constexpr (RESID) {
            float* p = (float*)g.out + (int64_t)row * g.ldc + col;
            f32x4 x = *(const f32x4*)p;
            f32x4 y = *(const f32x4*)(p + HOFF);
            x += lo + b_lo;
            y += hi + b_hi;
            *(f32x4*)p = x;
            *(f32x4*)(p + HOFF) = y;
            if constexpr (STAT) {
                u32x2 hx, hy;
                hx[0] = pack16x2<T>(x[0], x[1]); hx[1] = pack16x2<T>(x[2], x[3]);
                hy[0] = pack16x2<T>(y[0], y[1]); hy[1] = pack16x2<T>(y[2], y[3]);
                uint16_t* p16 = (uint16_t*)g.ex.x16 + (int64_t)row * g.ldc + col;
                *(u32x2*)p16 = hx;
                *(u32x2*)(p16 + HOFF) = hy;
                const float s1 = g2_row8_sum(((x[0] + x[1]) + (x[2] + x[3])) + ((y[0] + y[1]) + (y[2] + y[3])));
                const float s2 = g2_row8_sum(((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) +
                                             ((y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3])));
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
__global__ __launch_bounds__(256, 2) void gemm_wg2_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = the wave's column quarter (wn)
    const int l15 = lane & 15, lq = lane >> 4;

    // DMA: lane -> row lane >> 2 of a 16-row group, LDS slot lane & 3 holds logical chunk slot ^ ((row >> 2) & 1)
    const int drow = lane >> 2;
    const int dch = (lane & 3) ^ ((drow >> 2) & 1);
    const int voffA = drow * (int)g.lda * 2 + dch * 16, stepA = 16 * (int)g.lda * 2;
    const int voffW = drow * (int)g.ldw * 2 + dch * 16, stepW = 16 * (int)g.ldw * 2;
    // fragments: row lane & 15 of a 16-row block, logical chunk lane >> 4 (the k-step is the whole 64-byte row)
    const int xo = (lq ^ ((l15 >> 2) & 1)) << 4;
    const int a_off = l15 * G2_ROWB + xo;                            // + (half * 64 + i * 16) rows
    const int b_off = G2_A_BYTES + (wave * 64 + l15) * G2_ROWB + xo; // + j * 16 rows

    const int nkt = g.K / G2_BK;                                     // >= 4 (checked on the host)
    for (int L = blockIdx.x; L < g.ntiles; L += gridDim.x) {
        const Tile2 c = make_tile2(g, L);
        issue_dma2(c, smem, wave, voffA, voffW, stepA, stepW, 0);
        issue_dma2(c, smem + G2_STAGE, wave, voffA, voffW, stepA, stepW, G2_ROWB);
        Acc2 acc;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        int stage = 0, refill = 2;                                   // ring positions of tile kt and of tile kt + 2
        for (int kt = 0; kt < nkt; ++kt) {
            // my DMAs of tile kt have landed (those of kt + 1 may be in flight); after the barrier everybody's have, and everybody
            // has finished reading the stage that is refilled next (it held tile kt - 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G2_NDMA) : "memory");
            __builtin_amdgcn_s_barrier();
            // unconditional (the last two iterations re-fetch the final K tile into a stage nobody reads again): a branch around
            // VMEM would break the counted wait above
            const int kn = min(kt + 2, nkt - 1);
            issue_dma2(c, smem + refill * G2_STAGE, wave, voffA, voffW, stepA, stepW, kn * G2_ROWB);
            const char* st = smem + stage * G2_STAGE;
            // all twelve fragment reads first (hipcc otherwise serialises {one A read, wait, four MFMAs} to save registers and the
            // wave stalls on the LDS latency eight times per K tile); the second half's reads land under the first half's MFMAs
            typename T::v8 fa[2][4], fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *(const typename T::v8*)(st + b_off + j * 16 * G2_ROWB);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[h][i] = *(const typename T::v8*)(st + a_off + (h * 64 + i * 16) * G2_ROWB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[h * 4 + i][j] = T::mfma16(fb[j], fa[h][i], acc[h * 4 + i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage = stage == 2 ? 0 : stage + 1;
            refill = refill == 2 ? 0 : refill + 1;
        }
        // the stages become epilogue slabs: every DMA (incl. the two dummy refills) must have landed, every wave must have left the loop
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* slab = (float*)(smem + wave * G2_SLAB_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(slab + (ib * 16 + l15) * G2_ROWPF + j * 16 + 4 * lq) = acc[2 * i + ib][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            slab_epilogue2<T, EPI>(g, slab, lane, c.m0 + i * 32, c.n0 + wave * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // slab reads retired before the next slab overwrites it
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_s_barrier();                                // slabs released before the next tile's DMAs land on them
    }
}

template <typename T, int EPI>
int launch_wg2(const GemmArgs& g, int nblk, hipStream_t s) {
    static bool attr_set = false;
    auto kfn = gemm_wg2_kernel<T, EPI>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
        if (e != hipSuccess) { pg_set_error("gemm_wg2: set LDS attr: %s", hipGetErrorString(e)); return PG_EHIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, dim3(nblk), dim3(256), G2_LDS, s, g);
    return pg_check_launch("gemm_wg2");
}

template <typename T>
int dispatch_wg2(const GemmArgs& g, int epi, int nblk, hipStream_t s) {
    switch (epi) {
        case EPI_QKV: return launch_wg2<T, EPI_QKV>(g, nblk, s);
        case EPI_GELU: return launch_wg2<T, EPI_GELU>(g, nblk, s);
        case EPI_RESID: return launch_wg2<T, EPI_RESID>(g, nblk, s);
        case EPI_F32: return launch_wg2<T, EPI_F32>(g, nblk, s);
        case EPI_RESID_STAT: return launch_wg2<T, EPI_RESID_STAT>(g, nblk, s);
        case EPI_QKV_LN: return launch_wg2<T, EPI_QKV_LN>(g, nblk, s);
        case EPI_GELU_LN: return launch_wg2<T, EPI_GELU_LN>(g, nblk, s);
        default: pg_set_error("gemm_wg2: epilogue %d not supported", epi); return PG_EINVAL;
    }
}

}  // namespace

bool pg_gemm_wg2_supported(int epi, int N, int K) {
    return epi != EPI_PATCH && epi >= EPI_QKV && epi <= EPI_GELU_LN && N % G2_BN == 0 && K % G2_BK == 0 && K >= 4 * G2_BK;
}

int pg_gemm_wg2_launch(int dtype, GemmArgs g, int epi, hipStream_t s) {
    if (!pg_gemm_wg2_supported(epi, g.N, g.K)) { pg_set_error("gemm_wg2: unsupported epilogue / shape (epi=%d N=%d K=%d)", epi, g.N, g.K); return PG_EINVAL; }
    if ((int64_t)g.lda * 2 * G2_BM >= (1ll << 31) || (int64_t)g.ldw * 2 * G2_BN >= (1ll << 31)) {
        pg_set_error("gemm_wg2: operand panel exceeds the 2 GB buffer-descriptor range");
        return PG_EINVAL;
    }
    g.tilesM = (g.M + G2_BM - 1) / G2_BM;
    g.tilesN = g.N / G2_BN;
    g.ntiles = g.tilesM * g.tilesN;
    const int cap = 2 * pg_num_cus();                        // two workgroups per CU
    const int nblk = g.ntiles < cap ? g.ntiles : cap;
    if (dtype == PG_DTYPE_F16) return dispatch_wg2<T_F16>(g, epi, nblk, s);
    if (dtype == PG_DTYPE_BF16) return dispatch_wg2<T_BF16>(g, epi, nblk, s);
    pg_set_error("gemm_wg2: operand dtype must be PG_DTYPE_F16 or PG_DTYPE_BF16 (got %d)", dtype);
    return PG_EINVAL;
}

#endif  // PIGEON_ABLATIONS
