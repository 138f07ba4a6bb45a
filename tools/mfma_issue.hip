// mfma_issue.hip -- can ONE wave per SIMD keep the matrix pipe full?  Independent MFMAs issued back to back from a single
// wave per SIMD (accumulators in AGPRs, as gemm_w4.hip) against two waves per SIMD, v_mfma_f32_32x32x16_f16 (8 passes) against
// v_mfma_f32_16x16x32_f16 (4 passes), with and without an LDS read between the MFMAs.  Reports shader cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_issue.hip -o tools/bin/mfma_issue && tools/bin/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int KIND, bool LDS, int THREADS>   // KIND 0: 16 x (32x32x16) per iteration; 1: 64 x (16x16x32) per iteration
__global__ __launch_bounds__(THREADS) void issue_loop(const uint16_t* in, float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int lane = threadIdx.x & 63;
    f16x8 a = *(const f16x8*)(in + (size_t)(threadIdx.x) * 8), b = *(const f16x8*)(in + (size_t)(threadIdx.x + 77) * 8);
    *(f16x8*)(lds + threadIdx.x % 1024 * 16) = a;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16;
    f16x8 f0 = a, f1 = b;
    long long t0 = 0, t1 = 0;
    if (KIND == 0) {
        constexpr int NC = THREADS == 256 ? 16 : 6;
        f32x16 acc[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc[i]) : "v"(a), "v"(b));
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i % NC]) : "v"(f0), "v"(f1));
                if (LDS && (i & 1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(0));
            }
            if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) s += acc[i][0] + acc[i][7];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + a[0];
    } else {
        constexpr int NC = THREADS == 256 ? 64 : 24;
        f32x4 acc[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(acc[i]) : "v"(a), "v"(b));
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i % NC]) : "v"(f0), "v"(f1));
                if (LDS && (i & 7) == 7) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(0));
            }
            if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        t1 = __builtin_readcyclecounter();
        float s = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) s += acc[i][0];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + a[0];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, bool LDS, int THREADS>
void run(const char* name, int blocks, uint16_t* d, float* o, long long* c) {
    const int threads = THREADS;
    const int iters = 40000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((issue_loop<KIND, LDS, THREADS>), dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long hc = 0; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    const double mfma_per_wave = (double)iters * (KIND == 0 ? 16 : 64);
    const int waves_per_simd = threads / 256;
    const double flop = (double)blocks * (threads / 64) * mfma_per_wave * (KIND == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32);
    printf("%-44s blocks=%3d waves/SIMD=%d: %7.3f ms %6.0f TF/s  cycles per MFMA per SIMD %.1f (ideal %d)  clock %.2f GHz\n", name, blocks,
           waves_per_simd, ms, flop / (ms * 1e-3) / 1e12, (double)hc / (mfma_per_wave * waves_per_simd), KIND == 0 ? 32 : 16,
           (double)hc / (ms * 1e-3) / 1e9);
}

int main() {
    uint16_t* d; float* o; long long* c;
    hipMalloc(&d, 1 << 20); hipMalloc(&o, 512 * 512 * 4); hipMalloc(&c, 16);
    uint16_t* h = (uint16_t*)malloc(1 << 20);
    srand(1);
    for (int i = 0; i < (1 << 19); ++i) { _Float16 v = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1); memcpy(&h[i], &v, 2); }
    hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
    for (int blocks : {1, 256}) {
        run<0, false, 256>("32x32x16, MFMA only", blocks, d, o, c);
        run<0, true, 256>("32x32x16, one ds_read_b128 per two MFMAs", blocks, d, o, c);
        run<1, false, 256>("16x16x32, MFMA only", blocks, d, o, c);
        run<1, true, 256>("16x16x32, one ds_read_b128 per eight MFMAs", blocks, d, o, c);
        run<0, false, 512>("32x32x16 (8 chains/wave), MFMA only", blocks, d, o, c);
        run<0, true, 512>("32x32x16 (8 chains/wave), ds_read per two MFMAs", blocks, d, o, c);
        run<1, false, 512>("16x16x32 (24 chains/wave), MFMA only", blocks, d, o, c);
        run<1, true, 512>("16x16x32 (24 chains/wave), ds_read per eight MFMAs", blocks, d, o, c);
    }
    return 0;
}
