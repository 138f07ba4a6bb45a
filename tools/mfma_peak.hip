// mfma_peak.hip -- calibrates the MFMA ceiling and the sustained shader clock of THIS GPU under THIS kind of
// operand data (DVFS makes both data dependent).  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int DT>
__global__ __launch_bounds__(512) void mfma_loop(const uint16_t* in, float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x;
    f16x8 a = *(const f16x8*)(in + (size_t)(blockIdx.x * 512 + lane) * 8 % (1 << 20));
    f16x8 b = *(const f16x8*)(in + (size_t)(blockIdx.x * 512 + lane + 77) * 8 % (1 << 20));
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (DT == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

int main() {
    const int n = 1 << 20;
    std::vector<uint16_t> h(n);
    uint16_t *d; float* o; long long* c;
    hipMalloc(&d, n * 2); hipMalloc(&o, 2048 * 512 * 4); hipMalloc(&c, 16);
    for (int data = 0; data < 3; ++data) {
        srand(1);
        for (int i = 0; i < n; ++i) {
            if (data == 0) h[i] = 0;
            else if (data == 1) { _Float16 v = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1); memcpy(&h[i], &v, 2); }
            else { _Float16 v = (_Float16)(((rand() / (float)RAND_MAX) * 2 - 1) * 0.02f); memcpy(&h[i], &v, 2); }
        }
        hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
        for (int threads : {256, 512}) for (int dt : {0, 1}) {
            const int blocks = 256, iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (dt == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(threads), 0, 0, d, o, iters, c);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
            double flops = (double)blocks * (threads / 64) * iters * 8 * 2.0 * 32 * 32 * 16;
            printf("data=%s dtype=%s waves/SIMD=%d : %.3f ms  %.0f TF/s   shader clock %.3f GHz (cycles %lld, wall ticks %lld @100MHz)\n",
                   data == 0 ? "zero" : data == 1 ? "uniform[-1,1]" : "small(0.02)", dt == 0 ? "f16" : "bf16", threads / 256, ms,
                   flops / (ms * 1e-3) / 1e12, (double)hc[0] / ((double)hc[1] / 100e6) / 1e9, hc[0], hc[1]);
        }
    }
    return 0;
}
