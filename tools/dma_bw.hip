// dma_bw.hip -- microbenchmark: how many bytes per clock can ONE CU pull from L2 into LDS (direct-to-LDS DMA) or into
// VGPRs, with every CU of the chip doing the same?  This bounds the operand feed of the 256x256x64 GEMM tile
// (64 KB per K tile per CU = 32 B/clk at full MFMA rate).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_bw tools/dma_bw.hip && tools/bin/dma_bw
// Modes: 0 DMA, rows of 128 B at stride `ld` bytes (the GEMM pattern: 8 rows x 128 B per wave instruction, swizzled)
//        1 DMA, linear 1 KB per wave instruction
//        2 VGPR loads (buffer_load_dwordx4), GEMM pattern, results consumed by a cheap VALU op
//        3 DMA, GEMM pattern, unswizzled (lane c reads chunk c)
// Working set per CU: `panel_kb` KB re-read `iters` times (L2 resident when 32 * panel_kb <= ~3 MB per XCD); with
// share > 1, `share` neighbouring CUs of an XCD read the SAME panel (as GEMM tiles share operand panels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void bw_kernel(const char* base, long long panel_bytes, int ld, int iters, int share,
                                                 int nwaves, unsigned long long* cycles, unsigned int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave >= nwaves) return;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    // share > 0: `share` neighbouring blocks of an XCD read the same panel.  share == 0: GEMM-like sharing -- rows 0..255
    // come from an "A panel" shared by 4 blocks (8 per XCD), rows 256..511 from a "W panel" shared by 8 blocks (4 per XCD).
    const int panel = xcd * 64 + (share > 0 ? slot / share : slot / 4);
    const int panel2 = xcd * 64 + (share > 0 ? slot / share : 8 + slot % 4);
    const char* p = base + (long long)panel * panel_bytes;
    const char* p2 = base + (long long)panel2 * panel_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), (short)0, (int)panel_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p2), (short)0, (int)panel_bytes, 0x00020000);
    // one "K tile" = 512 rows x 128 B = 64 KB; a wave covers 8 rows per instruction -> 64 instructions per tile per CU
    const int per_wave = 64 / nwaves;
    int voff[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int grp = wave + i * nwaves;
        const int r = grp * 8 + (lane >> 3);
        int c = lane & 7;
        if (MODE == 0 || MODE == 2) c ^= (r >> 1) & 7;
        voff[i] = (MODE == 1) ? (grp * 1024 + lane * 16) : ((r & 255) * ld + c * 16);
    }
    const int ntile = MODE == 1 ? (int)(panel_bytes / 65536) : ld / 128;
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < ntile; ++t) {
            const int soff = MODE == 1 ? t * 65536 : t * 128;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i >= per_wave) break;
                const bool second = (MODE != 1) && (wave + i * nwaves >= 32);      // rows 256..511 (wave-uniform)
                if (MODE == 2) {
                    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(second ? rs2 : rs, voff[i], soff, 0);
                    acc ^= v;
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rs2 : rs, (lds_void*)(smem + ((t & 1) * 65536) + (wave + i * nwaves) * 1024), 16,
                                                             voff[i], soff, 0, 0);
                }
            }
            if (MODE != 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // keep one tile's worth in flight
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave == 0) cycles[blockIdx.x] = t1 - t0;
    if (MODE == 2) sink[blockIdx.x * 512 + tid] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    else if (tid == 0) sink[blockIdx.x] = ((unsigned int*)smem)[lane];
}

template <int MODE>
void run(const char* name, const char* base, long long panel_bytes, int ld, int iters, int share, int nwaves, int nblk,
         unsigned long long* dcyc, unsigned int* dsink) {
    CK(hipFuncSetAttribute((const void*)bw_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(bw_kernel<MODE>, dim3(nblk), dim3(512), 131072, 0, base, panel_bytes, ld, iters, share, nwaves, dcyc, dsink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> cyc(nblk);
    CK(hipMemcpy(cyc.data(), dcyc, nblk * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto c : cyc) avg += (double)c;
    avg /= nblk;
    const double bytes = (double)(ld == 2048 ? 2 * panel_bytes : panel_bytes) * iters;   // per CU (two 256-row panels when ld = 2048)
    printf("%-28s ld=%5d share=%d waves=%d: %7.1f GB/s/CU  %6.2f TB/s chip  %5.1f B/cycle/CU (cycle counter)  %.3f ms\n", name, ld,
           share, nwaves, bytes / (ms * 1e-3) / 1e9, bytes * nblk / (ms * 1e-3) / 1e12, bytes / avg, ms);
}

int main() {
    const int nblk = 256;
    const long long region = 512ll << 20;                        // room for 512 panels of up to 1 MB
    char* base;
    CK(hipMalloc((void**)&base, region));
    CK(hipMemset(base, 1, region));
    unsigned long long* dcyc; unsigned int* dsink;
    CK(hipMalloc((void**)&dcyc, nblk * 8));
    CK(hipMalloc((void**)&dsink, nblk * 512 * 4));
    // panel = 512 rows x ld bytes; ld = 2048 (K = 1024 fp16) -> 1 MB panel... keep it L2 resident: use ld = 256 (2 K tiles,
    // 128 KB per CU -> 4 MB per XCD: borderline) and ld = 128 (64 KB per CU -> 2 MB per XCD: resident)
    for (int share : {1, 4, 8}) {
        for (int nw : {8, 4}) {
            run<0>("DMA gemm-pattern swizzled", base, 512ll * 128, 128, 400, share, nw, nblk, dcyc, dsink);
            run<3>("DMA gemm-pattern linear", base, 512ll * 128, 128, 400, share, nw, nblk, dcyc, dsink);
            run<1>("DMA 1KB-linear", base, 65536, 128, 400, share, nw, nblk, dcyc, dsink);
            run<2>("VGPR loads gemm-pattern", base, 512ll * 128, 128, 400, share, nw, nblk, dcyc, dsink);
        }
    }
    // 2 KB row stride (K = 1024 fp16), 16 K tiles per 512 KB panel (256 rows)
    run<0>("DMA K=1024 panels, private", base, 256ll * 2048, 2048, 16, 1, 8, nblk, dcyc, dsink);
    run<2>("VGPR K=1024 panels, private", base, 256ll * 2048, 2048, 16, 1, 8, nblk, dcyc, dsink);
    run<0>("DMA K=1024 panels, GEMM sharing", base, 256ll * 2048, 2048, 16, 0, 8, nblk, dcyc, dsink);
    run<3>("DMA K=1024 GEMM sharing, linear", base, 256ll * 2048, 2048, 16, 0, 8, nblk, dcyc, dsink);
    run<2>("VGPR K=1024 panels, GEMM sharing", base, 256ll * 2048, 2048, 16, 0, 8, nblk, dcyc, dsink);
    run<0>("DMA K=1024 GEMM sharing, 4 waves", base, 256ll * 2048, 2048, 16, 0, 4, nblk, dcyc, dsink);
    return 0;
}
