// handshake_probe.hip -- VERDICT r05 item 3, measured at the level of the primitive: is an LDS-flag handshake between the two waves of a
// SIMD a cheaper way to alternate their MFMA phases than the CU-wide s_barrier the ping-pong GEMMs use (gemm_pp6.hip ktile6: eight
// barriers per K tile)?  The kernel is the ping-pong SKELETON of the 384 x 256 GEMM with nothing else in it: 8 waves per CU (one
// 512-thread block per CU, 256 blocks), two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD), four phases per "K tile", a
// phase = { LOAD stand-in: 10 ds_read_b128 + lgkmcnt(0) ; sync ; 24 x v_mfma_f32_16x16x32_f16 at s_setprio 1 ; sync }.  Variants of `sync`:
//   0  s_barrier / s_barrier, group 1 one barrier behind group 0                  (the product's schedule)
//   1  LDS flag per SIMD: wait until it is this wave's turn (s_sleep 1 between polls), MFMAs, pass the turn; ONE s_barrier per K tile
//      (what the real kernel would still need to hand the DMA'd stage over)
//   2  the same with a tight poll (no s_sleep)
//   3  no alternation at all: one s_barrier per K tile, both waves of a SIMD issue when they can (the free-running form)
// Reported: ms, TF/s, shader cycles and ns per phase PAIR (two MFMA phases = 2 x 384 matrix-pipe cycles ideal), effective clock.
//   hipcc --offload-arch=gfx950 -O3 tools/handshake_probe.hip -o tools/bin/handshake_probe && tools/bin/handshake_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int VAR>
__global__ __launch_bounds__(512) void skeleton(const uint16_t* in, float* out, int ktiles, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    __shared__ int turn[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, simd = wave & 3;
    for (int i = tid; i < 4096; i += 512) *(f16x8*)(lds + i * 16) = *(const f16x8*)(in + (size_t)((i * 37 + blockIdx.x) & 4095) * 8);
    if (tid < 4) turn[tid] = 0;
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (wave * 64 + lane) * 16;
    const uint32_t taddr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)turn + simd * 4;
    f16x8 a[6], b[4];
    f32x4 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int my_turn = grp;                                          // turn counter values at which THIS wave runs its MFMAs: grp, grp + 2, ...
    if (VAR == 0 && grp == 1) barrier();                        // group 1 runs one barrier behind
    const long long t0 = __builtin_readcyclecounter();
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            // LOAD stand-in: the fragment reads of a phase (6 A + 4 W blocks)
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[i]) : "v"(addr), "n"(0));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[i]) : "v"(addr), "n"(8192));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (VAR == 0) barrier();
            if (VAR == 1 || VAR == 2) {
                int v;
                while (true) {
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(taddr) : "memory");
                    if (__builtin_amdgcn_readfirstlane(v) == my_turn) break;
                    if (VAR == 1) __builtin_amdgcn_s_sleep(1);
                }
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i * 4 + j]) : "v"(b[j]), "v"(a[i]));
            __builtin_amdgcn_s_setprio(0);
            if (VAR == 0) barrier();
            if (VAR == 1 || VAR == 2) {
                my_turn += 1;                                    // the partner's turn value
                if (lane == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(taddr), "v"(my_turn) : "memory");
                my_turn += 1;                                    // ... and this wave's next one
            }
        }
        if (VAR != 0) barrier();                                 // the stage hand-over the real kernel would keep (once per K tile)
    }
    const long long t1 = __builtin_readcyclecounter();
    if (VAR == 0 && grp == 0) barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int VAR>
void run(const char* name, int blocks, uint16_t* d, float* o, long long* c) {
    const int ktiles = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((skeleton<VAR>), dim3(blocks), dim3(512), 0, 0, d, o, ktiles, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long hc = 0; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    const double pairs = (double)ktiles * 4;                       // per SIMD: 4 phases of each of its two waves = 4 phase pairs per K tile
    const double flop = (double)blocks * 8 * ktiles * 4 * 24 * 2.0 * 16 * 16 * 32;
    printf("%-58s blocks=%3d: %8.3f ms %6.0f TF/s  cycles per phase pair %.0f (768 ideal)  ns per phase pair %.0f  clock %.2f GHz\n", name, blocks, ms,
           flop / (ms * 1e-3) / 1e12, (double)hc / pairs, ms * 1e6 / pairs, (double)hc / (ms * 1e-3) / 1e9);
}

int main() {
    uint16_t* d; float* o; long long* c;
    hipMalloc(&d, 1 << 20); hipMalloc(&o, 512 * 512 * 4); hipMalloc(&c, 16);
    uint16_t* h = (uint16_t*)malloc(1 << 20);
    srand(1);
    for (int i = 0; i < (1 << 19); ++i) { _Float16 v = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1); memcpy(&h[i], &v, 2); }
    hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {1, 256}) {
            run<0>("0 s_barrier / s_barrier (the product's ping-pong)", blocks, d, o, c);
            run<1>("1 LDS turn flag per SIMD, s_sleep 1 between polls", blocks, d, o, c);
            run<2>("2 LDS turn flag per SIMD, tight poll", blocks, d, o, c);
            run<3>("3 free-running, one s_barrier per K tile", blocks, d, o, c);
        }
    return 0;
}
