// dma_rate_probe.hip -- round 6, second session: what is a K tile of gemm_mid.hip made of?  Its K loop takes 0.58 us per 64-wide K tile
// (32 KB of operands per CU) against 0.21 us of MFMA issue, and overlapping the fragment reads with the MFMAs changed nothing
// (tools/experiments/gemm_mid_pipelined.patch).  This kernel is gemm_mid's operand traffic WITHOUT the arithmetic: the same blocks
// (one per CU, 4 waves, 96 KB of LDS in three stages), the same direct-to-LDS DMAs (8 per wave and K tile, 8 rows x 128 B each, rows one
// operand pitch apart, source-side swizzle), the same tile -> block mapping over the same two matrices (fc2 at one panorama: A 2 308 x
// 4 096, W 1 024 x 4 096 fp16), the same barrier per K tile -- and nothing between the waits.  It reports the K tile period as a
// function of how many K tiles are in flight (1 .. 3) and of how many blocks run (152 = fc2 at one panorama, 256 = a full chip).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate_probe.hip -o tools/bin/dma_rate_probe && tools/bin/dma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ROWB = 128;                       // bytes of one operand row in a K tile (64 fp16)
constexpr int STAGE = 256 * ROWB;               // 128 A rows + 128 W rows = 32 KB
constexpr int LDS = 3 * STAGE;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_wave_uniform, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_wave_uniform, 16, voff, soff, 0, 0);
}

// PRODUCER: ONE wave issues all 32 DMAs of a K tile (the other three only meet it at the barrier) -- can a single wave feed the CU?
template <int DEPTH, bool BARRIER, bool PRODUCER = false>
__global__ __launch_bounds__(256) void dma_only(const uint16_t* A, const uint16_t* W, int M, int K, int tilesN, int reps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // gemm_mid's mapping: XCD x = blockIdx % 8 owns a contiguous chunk of the tile order, row tile slowest
    const int per = (gridDim.x + 7) / 8;
    int L = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (L >= (int)gridDim.x) L = blockIdx.x;
    int tm = L / tilesN;
    const int tn = L - tm * tilesN;
    tm %= (M + 127) / 128;                          // more blocks than tiles: the row tiles wrap (the same panels, shared by more CUs)
    const int rows = min(128, M - tm * 128);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A + (int64_t)tm * 128 * K), (short)0, rows * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W + (int64_t)tn * 128 * K), (short)0, 128 * K * 2, 0x00020000);
    int voff[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int r = (wave + 4 * d) * 8 + (lane >> 3);
        voff[d] = r * K * 2 + ((lane & 7) ^ ((r >> 1) & 7)) * 16;
    }
    // producer form: row group gr (8 rows) of a panel = lane offset of group (gr & 1) + an SGPR offset (gr >> 1) * 16 rows
    int pvoff[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int r = d * 8 + (lane >> 3);
        pvoff[d] = r * K * 2 + ((lane & 7) ^ ((r >> 1) & 7)) * 16;
    }
    const int nt = K / 64;
    for (int rep = 0; rep < reps; ++rep) {
        int issued = 0;
        for (int kt = 0; kt < nt; ++kt) {
            // keep DEPTH K tiles in flight: before waiting for tile kt, tiles kt .. kt + DEPTH - 1 have been issued
            while (issued < nt && issued < kt + DEPTH) {
                char* st = smem + (issued % 3) * STAGE;
                if (PRODUCER) {
                    if (wave == 0) {
#pragma unroll
                        for (int gr = 0; gr < 16; ++gr) dma16(ra, st + gr * 8 * ROWB, pvoff[gr & 1], issued * 128 + (gr >> 1) * 16 * K * 2);
#pragma unroll
                        for (int gr = 0; gr < 16; ++gr) dma16(rw, st + 128 * ROWB + gr * 8 * ROWB, pvoff[gr & 1], issued * 128 + (gr >> 1) * 16 * K * 2);
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) dma16(ra, st + (wave + 4 * d) * 8 * ROWB, voff[d], issued * 128);
#pragma unroll
                    for (int d = 0; d < 4; ++d) dma16(rw, st + 128 * ROWB + (wave + 4 * d) * 8 * ROWB, voff[d], issued * 128);
                }
                ++issued;
            }
            const int ahead = issued - kt - 1;              // tiles issued after tile kt: their DMAs (8 per wave, 32 for a producer) may stay outstanding
            if (PRODUCER) {
                if (wave == 0) {
                    if (ahead >= 1) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // (two tiles ahead would be 64 > the counter's 63)
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            } else if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (BARRIER) __builtin_amdgcn_s_barrier();
        }
    }
    if (sink && tid == 0) sink[blockIdx.x] = ((volatile unsigned*)smem)[lane];
}

template <int DEPTH, bool BARRIER, bool PRODUCER = false>
static int run(const uint16_t* A, const uint16_t* W, int M, int K, int blocks, unsigned* sink) {
    auto k = dma_only<DEPTH, BARRIER, PRODUCER>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int reps = 40, nt = K / 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), LDS, 0, A, W, M, K, 8, reps, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    const double us_tile = best * 1e3 / (reps * nt);
    printf("  %s%d K tiles in flight, %s: %6.3f us per K tile and CU = %5.1f KB/us per CU, %5.2f TB/s over %d CUs\n", PRODUCER ? "ONE wave issues all 32 DMAs, " : "", DEPTH,
           BARRIER ? "barrier per K tile" : "no barrier        ", us_tile, 32.0 / us_tile, blocks * 32768.0 / us_tile / 1e6, blocks);
    return 0;
}

int main() {
    const int M = 2308, K = 4096, N = 1024;
    uint16_t *A, *W; unsigned* sink;
    CK(hipMalloc(&A, (size_t)(M + 256) * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(A, 0x11, (size_t)(M + 256) * K * 2)); CK(hipMemset(W, 0x22, (size_t)N * K * 2));
    printf("gemm_mid's operand DMAs alone (fc2 at one panorama: A 2308 x 4096, W 1024 x 4096 fp16; 128 x 128 tiles, 32 KB per K tile and CU)\n");
    for (int blocks : {152, 256, 32}) {
        printf("%d blocks (one per CU)%s:\n", blocks, blocks == 152 ? " -- the tiles of this shape" : blocks == 256 ? " -- a full chip (row tiles wrap)" : "");
        if (run<1, true>(A, W, M, K, blocks, sink)) return 1;
        if (run<2, true>(A, W, M, K, blocks, sink)) return 1;
        if (run<3, true>(A, W, M, K, blocks, sink)) return 1;
        if (run<3, false>(A, W, M, K, blocks, sink)) return 1;
        if (run<1, true, true>(A, W, M, K, blocks, sink)) return 1;
        if (run<2, true, true>(A, W, M, K, blocks, sink)) return 1;
    }
    return 0;
}
