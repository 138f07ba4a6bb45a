// mfma_fp8_probe.hip -- round 6, groundwork for the fp8-corrected exact tier: what the block-scaled fp8 MFMA
// of gfx950 computes, where a lane's operand bytes sit, what the E8M0 scale operands do, and how fast it issues next to the fp16 MFMA
// the GEMM kernels use.   hipcc --offload-arch=gfx950 -O3 tools/mfma_fp8_probe.hip -o tools/bin/mfma_fp8_probe && tools/bin/mfma_fp8_probe
//
//   v_mfma_scale_f32_16x16x128_f8f6f4  D[16][16] += A[16][128] . B[16][128]^T, A / B OCP e4m3 (cbsz = blgp = 0), one E8M0 scale byte per
//   lane and operand (2^(s - 127) per 32 consecutive k).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wave: lane l hands over the 32 bytes at a[l] / b[l] (the host decides what they mean) and gets its 4 accumulators back
__global__ void one_mfma(const i32x8* a, const i32x8* b, f32x4* d, int sa, int sb) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    d[l] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, 0, sa, 0, sb);
}

template <int KIND>
__global__ __launch_bounds__(256) void rate(const i32x8* in, float* out, int iters, long long* cyc) {
    const int l = threadIdx.x;
    i32x8 a = in[(blockIdx.x * 256 + l) & 1023], b = in[(blockIdx.x * 256 + l + 77) & 1023];
    const f16x8 ah = __builtin_bit_cast(f16x8, ((const int __attribute__((ext_vector_type(4)))*)&a)[0]);
    const f16x8 bh = __builtin_bit_cast(f16x8, ((const int __attribute__((ext_vector_type(4)))*)&b)[0]);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            else if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);   // e2m3 (fp6): 24 of the 32 bytes
            else if (KIND == 3) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);   // e2m1 (fp4): 16 of the 32 bytes
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static float e4m3(uint8_t v) {                                  // OCP e4m3fn: bias 7, no infinities, 0x7F / 0xFF = NaN
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    const float mag = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -mag : mag;
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    srand(7);
    for (auto* v : {&A, &B})
        for (auto& x : *v) { do { x = (uint8_t)(rand() & 0xFF); } while ((x & 0x7F) == 0x7F || ((x >> 3) & 15) > 10); }   // finite, moderate
    std::vector<double> ref(256, 0.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 128; ++k) s += (double)e4m3(A[i * 128 + k]) * e4m3(B[j * 128 + k]);
            ref[i * 16 + j] = s;
        }
    i32x8 *da, *db; f32x4* dd;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dd, 64 * 16));
    // hypothesis per operand: lane l = row (l & 15), bytes k = 32 (l >> 4) .. + 31   (H1)   or the two 16-byte halves 64 apart (H2)
    for (int hyp = 1; hyp <= 2; ++hyp) {
        std::vector<uint8_t> la(64 * 32), lb(64 * 32);
        for (int l = 0; l < 64; ++l)
            for (int t = 0; t < 32; ++t) {
                const int k = hyp == 1 ? 32 * (l >> 4) + t : (t < 16 ? 16 * (l >> 4) + t : 64 + 16 * (l >> 4) + (t - 16));
                la[l * 32 + t] = A[(l & 15) * 128 + k];
                lb[l * 32 + t] = B[(l & 15) * 128 + k];
            }
        CK(hipMemcpy(da, la.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, lb.data(), 64 * 32, hipMemcpyHostToDevice));
        for (int sc = 0; sc < 3; ++sc) {
            const int sa = sc == 0 ? 0x7F7F7F7F : (sc == 1 ? 127 - 11 : 0), sb = sc == 0 ? 0x7F7F7F7F : (sc == 1 ? 127 - 11 : 0);
            hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb);
            std::vector<float> d(256);
            CK(hipMemcpy(d.data(), dd, 64 * 16, hipMemcpyDeviceToHost));
            // D layouts tried: (a) first operand = rows: lane l holds D[4 (l >> 4) + r][l & 15]; (b) transposed
            for (int lay = 0; lay < 2; ++lay) {
                double worst = 0, scale_seen = 0;
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 4; ++r) {
                        const int i = lay == 0 ? 4 * (l >> 4) + r : (l & 15), j = lay == 0 ? (l & 15) : 4 * (l >> 4) + r;
                        const double want = ref[i * 16 + j];
                        if (std::fabs(want) > 1e-3) scale_seen = d[l * 4 + r] / want;
                        worst = std::fmax(worst, std::fabs(d[l * 4 + r] - want * (sc == 1 ? std::ldexp(1.0, -22) : 1.0)) / (std::fabs(want) * (sc == 1 ? std::ldexp(1.0, -22) : 1.0) + 1e-6));
                    }
                printf("k layout H%d, scales %s, D layout %s: worst relative deviation %.3e (last ratio result / reference %.6g)\n", hyp,
                       sc == 0 ? "0x7F both (2^0)" : (sc == 1 ? "116 both (2^-22 expected)" : "0 both"), lay == 0 ? "D[4(l>>4)+r][l&15]" : "D[l&15][4(l>>4)+r]",
                       worst, scale_seen);
            }
        }
    }
    // issue rate: 1024 blocks x 4 waves = FOUR waves per SIMD (one wave per SIMD does not keep the matrix pipe full: the first
    // record of this probe read 746 TFLOP/s for the fp16 instruction), 8 independent accumulators per wave
    std::vector<int> junk(1024 * 8);
    for (auto& x : junk) x = 0x38383838 ^ (rand() & 0x07070707);
    i32x8* din; float* dout; long long* dc;
    CK(hipMalloc(&din, 1024 * 32)); CK(hipMalloc(&dout, 1024 * 256 * 4)); CK(hipMalloc(&dc, 8));
    CK(hipMemcpy(din, junk.data(), 1024 * 32, hipMemcpyHostToDevice));
    for (int kind = 0; kind < 4; ++kind) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, din, dout, iters, dc);
            else if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(1024), dim3(256), 0, 0, din, dout, iters, dc);
            else if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(1024), dim3(256), 0, 0, din, dout, iters, dc);
            else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, din, dout, iters, dc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc; CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
        const double flop = 2.0 * 16 * 16 * (kind != 1 ? 128 : 32) * 8.0 * iters * 4 * 1024;
        printf("%s: %.1f shader cycles per instruction and wave (4 waves per SIMD), %.0f TFLOP/s on 256 CUs (%.2f ms)\n",
               kind == 0 ? "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3)" : kind == 2 ? "v_mfma_scale_f32_16x16x128_f8f6f4 (e2m3)" : kind == 3 ? "v_mfma_scale_f32_16x16x128_f8f6f4 (e2m1)" : "v_mfma_f32_16x16x32_f16              ", (double)cyc / (8.0 * iters), flop / (ms * 1e-3) / 1e12, ms);
    }
    return 0;
}
