// store_bw.hip -- microbenchmark: what does ONE CU's global-store path cost per wave instruction, with every CU of the chip
// storing at the same time (the situation of a persistent GEMM's epilogue)?  The fused epilogues all move about 35 GB/s per CU
// whatever their bytes per instruction: is the cost per byte, per instruction, or per cache-line
// request?  Each wave streams its own region of a 16-bit "output matrix" with one of the access shapes the epilogues use or
// could use:
//   0  b128 per lane,  8 rows x 128 B per instruction  (slab epilogue: 8 full lines)
//   1  b64  per lane,  8 rows x  64 B                  (the 16-bit copy of the residual epilogue: 8 half lines)
//   2  b64  per lane, 16 rows x  32 B                  (store straight from the MFMA accumulator layout: 16 quarter lines)
//   3  b128 per lane, 16 rows x  64 B                  (accumulator layout with permuted weight rows: 16 half lines)
//   4  b128 per lane,  4 rows x 256 B                  (two full lines per row)
//   5  b32  per lane,  8 rows x  32 B
//   6  b128 per lane, linear 1 KB (one row of 1024 contiguous bytes)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/store_bw tools/store_bw.hip && tools/bin/store_bw [row_stride_bytes=8192] [blocks=256] [passes=1]
// Prints clocks per wave instruction per CU and bytes per clock per CU (8 waves per CU storing, as in the epilogues).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int INSTR = 24;            // store instructions per wave and pass: one 384 x 256 tile's 16-bit epilogue (192 KB per CU)

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(char* base, long long region_bytes, int ld, int passes,
                                                    unsigned long long* cycles) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* p = base + ((long long)blockIdx.x * 8 + wave) * region_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)region_bytes, 0x00020000);
    // rows per instruction and the lane's (row, byte) inside it
    int rows, voff;
    if (MODE == 0) { rows = 8; voff = (lane >> 3) * ld + (lane & 7) * 16; }
    else if (MODE == 1) { rows = 8; voff = (lane >> 3) * ld + (lane & 7) * 8; }
    else if (MODE == 2) { rows = 16; voff = (lane & 15) * ld + (lane >> 4) * 8; }
    else if (MODE == 3) { rows = 16; voff = (lane & 15) * ld + (lane >> 4) * 16; }
    else if (MODE == 4) { rows = 4; voff = (lane >> 4) * ld + (lane & 15) * 16; }
    else if (MODE == 5) { rows = 8; voff = (lane >> 3) * ld + (lane & 7) * 4; }
    else { rows = 1; voff = lane * 16; }
    const int step = MODE == 6 ? 1024 : rows * ld;           // bytes between two instructions of a wave
    u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < passes; ++it) {
#pragma unroll 8
        for (int i = 0; i < INSTR; ++i) {
            const int off = voff + (it * INSTR + i) * step;  // every pass writes fresh lines, as every tile does
            if (MODE == 1 || MODE == 2) { u32x2 h; h[0] = v[0]; h[1] = v[1]; __builtin_amdgcn_raw_buffer_store_b64(h, rs, off, 0, 0); }
            else if (MODE == 5) __builtin_amdgcn_raw_buffer_store_b32(v[0], rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
        }
        v[0] += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int bytes_per_instr, char* base, long long region, int ld, int passes, int nblk, unsigned long long* dcyc) {
    std::vector<unsigned long long> h(nblk);
    double best = 1e30, sum = 0;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float ms_best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(store_kernel<MODE>, dim3(nblk), dim3(512), 0, 0, base, region, ld, passes, dcyc);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h.data(), dcyc, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        sum = 0;
        for (auto c : h) sum += (double)c;
        const double avg = sum / nblk;
        if (rep > 0 && avg < best) { best = avg; ms_best = ms; }
    }
    const double instr_per_cu = 8.0 * INSTR * passes;
    const double total_bytes = (double)nblk * instr_per_cu * bytes_per_instr;
    printf("%-44s %6.1f clk / wave instruction / CU   %5.1f B/clk/CU   %6.2f TB/s chip (wall clock)\n", name, best / instr_per_cu,
           instr_per_cu * bytes_per_instr / best, total_bytes / (ms_best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int ld = argc > 1 ? atoi(argv[1]) : 8192;          // fc1's output row: 4096 x 2 bytes
    const int nblk = argc > 2 ? atoi(argv[2]) : 256;
    const int passes = argc > 3 ? atoi(argv[3]) : 1;         // 1 = one tile's burst (50 MB chip-wide: L2 + MALL take it); more = sustained
    // a wave's region: passes x INSTR instructions of up to 16 rows
    const long long region = (long long)passes * INSTR * 16 * ld + 4096;
    char* base;
    CK(hipMalloc(&base, (size_t)nblk * 8 * region));
    CK(hipMemset(base, 0, (size_t)nblk * 8 * region));
    unsigned long long* dcyc;
    CK(hipMalloc(&dcyc, nblk * sizeof(unsigned long long)));
    printf("row stride %d B, %d blocks x 8 waves, %d store instructions per wave and pass, %d pass(es) over fresh lines\n", ld, nblk, INSTR, passes);
    run<0>("0 b128,  8 rows x 128 B (8 full lines)", 1024, base, region, ld, passes, nblk, dcyc);
    run<1>("1 b64,   8 rows x  64 B (8 half lines)", 512, base, region, ld, passes, nblk, dcyc);
    run<2>("2 b64,  16 rows x  32 B (16 quarter lines)", 512, base, region, ld, passes, nblk, dcyc);
    run<3>("3 b128, 16 rows x  64 B (16 half lines)", 1024, base, region, ld, passes, nblk, dcyc);
    run<4>("4 b128,  4 rows x 256 B (8 full lines)", 1024, base, region, ld, passes, nblk, dcyc);
    run<5>("5 b32,   8 rows x  32 B (8 quarter lines)", 256, base, region, ld, passes, nblk, dcyc);
    run<6>("6 b128, linear 1 KB (8 full lines, 1 row)", 1024, base, region, ld, passes, nblk, dcyc);
    return 0;
}
