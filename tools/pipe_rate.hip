// pipe_rate.hip -- issue-rate / co-issue microbenchmark for one gfx950 CU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/pipe_rate tools/pipe_rate.hip && tools/bin/pipe_rate
// One 8-wave block on one CU: waves 0..3 (one per SIMD) run job A, waves 4..7 run job B; each wave reports s_memtime
// cycles for ITER repetitions of a 32-instruction unrolled body.  Jobs: 0 idle, 1 v_exp_f32, 2 v_pk_mul_f32, 3 v_mul_f32,
// 4 MFMA 32x32x16 f16 (two independent accumulators), 5 v_cvt_pk_f16_f32, 6 v_max3_f32, 7 ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define ITER 2000

template <int JOB>
__device__ __forceinline__ float run_job(float seed, char* lds) {
    float acc = 0.f;
    if (JOB == 1) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7]));
        }
        for (int i = 0; i < 8; ++i) acc += v[i];
    } else if (JOB == 2) {
        f32x2 v[8];
        for (int i = 0; i < 8; ++i) v[i] = f32x2{seed * i, seed};
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v[u & 7]));
        }
        for (int i = 0; i < 8; ++i) acc += v[i][0] + v[i][1];
    } else if (JOB == 3) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[u & 7]));
        }
        for (int i = 0; i < 8; ++i) acc += v[i];
    } else if (JOB == 4) {
        f32x16 a = {}, b = {};
        f16x8 x, y;
        for (int i = 0; i < 8; ++i) { x[i] = (_Float16)seed; y[i] = (_Float16)(seed * 0.5f); }
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a, 0, 0, 0);
                b = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, b, 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) acc += a[i] + b[i];
    } else if (JOB == 5) {
        float v[8]; uint32_t w[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(w[u & 7]) : "v"(v[u & 7]));
        }
        for (int i = 0; i < 8; ++i) acc += (float)w[i];
    } else if (JOB == 6) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[u & 7]));
        }
        for (int i = 0; i < 8; ++i) acc += v[i];
    } else if (JOB == 8) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
        const uint32_t a = 0x3C003C00u;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(v[u & 7]) : "v"(a));
        }
        for (int i = 0; i < 8; ++i) acc += v[i];
    } else if (JOB == 9) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[u & 7]));
        }
        for (int i = 0; i < 8; ++i) acc += v[i];
    } else if (JOB == 10) {
        uint32_t v[8];
        for (int i = 0; i < 8; ++i) v[i] = 0x3C003C00u + i;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_pk_add_f16 %0, %0, %0" : "+v"(v[u & 7]));
        }
        for (int i = 0; i < 8; ++i) acc += (float)v[i];
    } else if (JOB == 11) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
        const uint32_t a = 0x3C003C00u;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(v[u & 7]) : "v"(a));
        }
        for (int i = 0; i < 8; ++i) acc += v[i];
    } else if (JOB == 7) {
        u32x4 r[8];
        const char* p = lds + (threadIdx.x & 63) * 16;
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) r[u & 7] = *(volatile u32x4*)(p + (u & 3) * 1024);
        }
        for (int i = 0; i < 8; ++i) acc += (float)r[i][0];
    }
    return acc;
}

// ---- attention-shaped wave program (no LDS, no memory): per iteration 8 MFMAs on two accumulators, a 16-deep max3 chain
// over the results, 32 exp, 16 cvt_pk, 16 dot2c, then 8 MFMAs consuming the packed values.  MODE 1 replaces the VALU part
// by nothing (MFMA only), MODE 2 drops the MFMAs (VALU only).  Run with 1..3 waves per SIMD to see how far the MFMA and
// VALU phases of different waves overlap.
template <int MODE, int PRIO = 0>
__global__ __launch_bounds__(768) void attn_like_kernel(float seed, long long* cyc, float* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f16x8 q, k;
    for (int i = 0; i < 8; ++i) { q[i] = (_Float16)(seed * 0.01f); k[i] = (_Float16)(seed * 0.02f); }
    f32x16 o0 = {}, o1 = {};
    float l0 = 0.f, l1 = 0.f;
    const f32x16 zero = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; ++it) {
        f32x16 s0 = zero, s1 = zero;
        asm volatile("" : "+v"(q), "+v"(k));                 // not loop-invariant for the compiler
        if (MODE != 2) {
            if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k, q, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q, k, s1, 0, 0, 0);
            }
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] = o0[r] * 0.5f; s1[r] = o1[r] * 0.25f; }
        }
        uint32_t pw[16];
        if (MODE != 1) {
            float m = s0[0];
#pragma unroll
            for (int r = 1; r < 15; r += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, s0[r]), s0[r + 1]);
#pragma unroll
            for (int r = 0; r < 16; r += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, s1[r]), s1[r + 1]);
            asm volatile("" : "+v"(m));
            l0 += m;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float e0 = __builtin_amdgcn_exp2f(s0[2 * r]), e1 = __builtin_amdgcn_exp2f(s0[2 * r + 1]);
                const float e2 = __builtin_amdgcn_exp2f(s1[2 * r]), e3 = __builtin_amdgcn_exp2f(s1[2 * r + 1]);
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const h2 a = {(_Float16)e0, (_Float16)e1}, b = {(_Float16)e2, (_Float16)e3};
                pw[r] = __builtin_bit_cast(uint32_t, a);
                pw[8 + r] = __builtin_bit_cast(uint32_t, b);
                const h2 one = {(_Float16)1.f, (_Float16)1.f};
                if (MODE == 3) { l0 += e0 + e1; l1 += e2 + e3; }
                else {
                    l0 = __builtin_amdgcn_fdot2(a, one, l0, false);
                    l1 = __builtin_amdgcn_fdot2(b, one, l1, false);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) pw[r] = __builtin_bit_cast(uint32_t, s0[r]) ^ __builtin_bit_cast(uint32_t, s1[r]);
        }
        if (MODE != 2) {
            if (PRIO == 1) __builtin_amdgcn_s_setprio(3);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4 pv = {pw[4 * u], pw[4 * u + 1], pw[4 * u + 2], pw[4 * u + 3]};
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k, __builtin_bit_cast(f16x8, pv), o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(q, __builtin_bit_cast(f16x8, pv), o1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] += __builtin_bit_cast(float, pw[r]); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = l0 + l1;
    for (int i = 0; i < 16; ++i) acc += o0[i] + o1[i];
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    sink[threadIdx.x] = acc;
}

template <int MODE, int PRIO = 0>
static void run_attn_like(const char* name) {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 128); hipMalloc(&sink, 4096);
    for (int wps = 1; wps <= 3; ++wps) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((attn_like_kernel<MODE, PRIO>), dim3(1), dim3(256 * wps), 0, 0, 1.0001f, cyc, sink);
        hipDeviceSynchronize();
        long long c[12];
        hipMemcpy(c, cyc, 8 * 4 * wps, hipMemcpyDeviceToHost);
        long long mx = 0;
        for (int i = 0; i < 4 * wps; ++i) mx = c[i] > mx ? c[i] : mx;
        printf("%-24s %d wave(s)/SIMD: first wave %7.1f ticks/iteration, slowest %7.1f -> %7.1f per wave-iteration per SIMD\n", name, wps,
               c[0] / 1000.0, mx / 1000.0, mx / 1000.0 / wps);
    }
    hipFree(cyc); hipFree(sink);
}

template <int JA, int JB>
__global__ __launch_bounds__(512) void rate_kernel(float seed, long long* cyc, uint32_t* hwid, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[8192];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) ((float*)lds)[i] = seed;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long t0 = __builtin_amdgcn_s_memtime();
    float r;
    if (wave < 4) r = run_job<JA>(seed, lds); else r = run_job<JB>(seed, lds);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        cyc[wave] = t1 - t0;
        uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hwid[wave] = id;
    }
    sink[threadIdx.x] = r;
}

template <int JA, int JB>
static void run(const char* name, double ops_a, double ops_b) {
    long long* cyc; uint32_t* hw; float* sink;
    hipMalloc(&cyc, 64); hipMalloc(&hw, 32); hipMalloc(&sink, 2048);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((rate_kernel<JA, JB>), dim3(1), dim3(512), 0, 0, 1.0001f, cyc, hw, sink);
    hipDeviceSynchronize();
    long long c[8]; uint32_t h[8];
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost); hipMemcpy(h, hw, 32, hipMemcpyDeviceToHost);
    printf("%-28s", name);
    printf(" A: %7.2f ticks/inst", (double)c[0] / ops_a);
    if (JB) printf("   B: %7.2f ticks/inst", (double)c[4] / ops_b);
    printf("   simd(w0,w4)=%u,%u ticks A %lld B %lld\n", (h[0] >> 4) & 3, (h[4] >> 4) & 3, c[0], c[4]);
    hipFree(cyc); hipFree(hw); hipFree(sink);
}

int main() {
    const double n32 = 32.0 * ITER;
    // s_memtime ticks at a constant 100 MHz on gfx9: also print a calibration against a known 4-cycle op (v_mul_f32)
    run<3, 0>("v_mul_f32 alone", n32, 1);
    run<1, 0>("v_exp_f32 alone", n32, 1);
    run<2, 0>("v_pk_mul_f32 alone", n32, 1);
    run<5, 0>("v_cvt_pk_f16_f32 alone", n32, 1);
    run<6, 0>("v_max3_f32 alone", n32, 1);
    run<4, 0>("mfma32x32x16 alone", n32, 1);
    run<7, 0>("ds_read_b128 alone", n32, 1);
    run<3, 3>("v_mul + v_mul", n32, n32);
    run<4, 4>("mfma + mfma", n32, n32);
    run<4, 3>("mfma + v_mul", n32, n32);
    run<4, 1>("mfma + v_exp", n32, n32);
    run<4, 2>("mfma + v_pk_mul", n32, n32);
    run<4, 7>("mfma + ds_read_b128", n32, n32);
    run<1, 3>("v_exp + v_mul", n32, n32);
    run<5, 4>("v_cvt_pk + mfma", n32, n32);
    run<6, 4>("v_max3 + mfma", n32, n32);
    run<8, 0>("v_dot2c_f32_f16 alone", n32, 1);
    run<8, 4>("v_dot2c + mfma", n32, n32);
    run<11, 4>("v_dot2_f32_f16 + mfma", n32, n32);
    run<9, 4>("v_fma_f32 + mfma", n32, n32);
    run<10, 4>("v_pk_add_f16 + mfma", n32, n32);
    run<2, 4>("v_pk_mul_f32 + mfma", n32, n32);
    run_attn_like<0>("attention-like");
    run_attn_like<3>("attn-like, v_add for dot2c");
    run_attn_like<3, 1>("v_add, MFMA phases prio 3");
    run_attn_like<3, 2>("v_add, VALU phase prio 3");
    run_attn_like<1>("attention-like MFMA only");
    run_attn_like<2>("attention-like VALU only");
    return 0;
}
