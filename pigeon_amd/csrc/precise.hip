// precise.hip -- the kernels of the encoder's EXACT mode (pg_vit_forward_precise, vit.hip): a second, near-fp32 pass over the few
// panoramas whose geocell top-1 / top-2 logit margin is inside the error band of the 16-bit path (reference
// models/super_guessr.py:447-459: `torch.argmax(geocell_probs)` is fp32 end to end).
//
// Arithmetic.  fp32 everywhere the fast path is fp32 (residual stream, LayerNorm, softmax), and the GEMMs on the SAME persistent
// fp16 MFMA kernels (gemm_pp.hip, EPI_F32 / EPI_RESID / EPI_PATCH) with every fp32 operand split in two fp16 halves and the three
// significant partial products laid side by side along K:
//     x = hi + lo,  W = Wh + Wl          (hi = fp16(x), lo = fp16(x - hi): 22 significant bits together)
//     x . W  ~=  hi.Wh + lo.Wh + hi.Wl   (lo.Wl ~ 2^-24 of the product: dropped)
//     A' = [ hi | lo | hi * 2^-8 ]   (row of 3K fp16),   W' = [ Wh | Wh | Wl * 2^8 ]   (row of 3K fp16)
// so one K' = 3K GEMM with fp32 accumulation delivers the fp32-grade product (the 2^8 keeps Wl -- ~2^-12 |W| ~ 5e-6 for
// |W| ~ 0.02 -- out of fp16's subnormal range; lo needs no scale: its absolute error is bounded by the subnormal quantum 2^-24).
// 3x the MFMA work of the fast path per image, 1/5 of what fp32 MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) would cost.
// Attention (8.6 % of the FLOPs): round 4 ran it in plain fp32 on that fp32 MFMA (attention_f32_kernel, kept as the A/B arm); round 5
// runs it on split fp16 operands like the GEMMs (attention_x3_kernel).
//
// Kernels here (all HBM-bound streaming except attention):
//   im2col_x3_kernel   pixels (n,3,336,336) -> patch matrix triple [576 n][3 * 640] fp16
//   ln_x3_kernel       LayerNorm of fp32 rows -> triple [M][3 * 1024]
//   split_x3_kernel    fp32 [M][C] (optionally through QuickGELU) -> triple [M][3 * C]
//   sum_parts_kernel   (round 5) fixed-order sum of the K-split partial products of one GEMM (+ the fp32 residual row)
//   attention_x3_kernel / attention_f32_kernel  fp32 QKV [M][3072] -> fp32 O [M][1024], softmax(q k^T / 8) v per (image, head)
#include "common.h"
#include "pigeon_internal.h"
#include "x3.h"
#include <cstdlib>

// ---- LayerNorm -> triple; one wave per 1024-float row (layernorm_kernel's arithmetic, rowops.hip) ------------------------------
__global__ __launch_bounds__(256) void ln_x3_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, uint16_t* __restrict__ y, int64_t rows, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * VIT_HIDDEN;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) * (1.0f / VIT_HIDDEN);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / VIT_HIDDEN) + eps);
    uint16_t* yr = y + row * (3 * VIT_HIDDEN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 g4 = *(const f32x4*)(gamma + c);
        const f32x4 b4 = *(const f32x4*)(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
        x3_store4(yr, VIT_HIDDEN, c, o);
    }
}
int pg_x3_ln_launch(const float* x, const float* gamma, const float* beta, void* y3, int64_t rows, float eps, hipStream_t s) {
    if (rows <= 0) return PG_OK;
    hipLaunchKernelGGL(ln_x3_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, (uint16_t*)y3, rows, eps);
    return pg_check_launch("ln_x3");
}

// ---- fp32 [rows][C] -> triple [rows][3C], optionally through QuickGELU x * sigmoid(1.702 x) (modeling_clip.py QuickGELUActivation)
// with the accurate expf and an IEEE division (the fast path's v_exp / v_rcp forms are good to ~1e-6 relative, not to the last ulp)
template <bool GELU>
__global__ __launch_bounds__(256) void split_x3_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t rows, int C) {
    const int64_t vec_per_row = C / 4, total = rows * vec_per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vec_per_row;
        const int c = (int)(i - r * vec_per_row) * 4;
        f32x4 v = *(const f32x4*)(x + r * C + c);
        if (GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = x3_quick_gelu(v[e]);
        }
        x3_store4(y + r * (3 * (int64_t)C), C, c, v);
    }
}
int pg_x3_split_launch(const float* x, void* y3, int64_t rows, int C, int gelu, hipStream_t s) {
    if (rows <= 0) return PG_OK;
    if (C % 4) { pg_set_error("split_x3: C must be a multiple of 4"); return PG_EINVAL; }
    int64_t blocks = (rows * (C / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (gelu) hipLaunchKernelGGL(split_x3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, x, (uint16_t*)y3, rows, C);
    else hipLaunchKernelGGL(split_x3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, x, (uint16_t*)y3, rows, C);
    return pg_check_launch("split_x3");
}

// ---- fixed-order sum of the K-split partial products (vit.hip precise_gemm): dst = (RESID ? dst : 0) + ((p0 + p1) + p2 ...) -----
template <bool RESID>
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, int S, int64_t stride4, float* __restrict__ dst,
                                                        int64_t n4) {
    const f32x4* p = (const f32x4*)parts;
    f32x4* d = (f32x4*)dst;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 acc = p[i];
        for (int k = 1; k < S; ++k) acc += p[k * stride4 + i];
        d[i] = RESID ? d[i] + acc : acc;
    }
}
int pg_sum_parts_launch(const float* parts, int S, int64_t part_elems, float* dst, int64_t n, int resid, hipStream_t s) {
    if (n <= 0) return PG_OK;
    if ((n & 3) || (part_elems & 3) || S < 1) { pg_set_error("sum_parts: element counts must be multiples of 4"); return PG_EINVAL; }
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (resid) hipLaunchKernelGGL(sum_parts_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, parts, S, part_elems / 4, dst, n / 4);
    else hipLaunchKernelGGL(sum_parts_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, parts, S, part_elems / 4, dst, n / 4);
    return pg_check_launch("sum_parts");
}

// ---- im2col -> triple [576 n][3 * 640]; k order == Conv2d weight [1024,3,14,14] flattened, columns 588..639 zero ----------------
template <typename PIX, int PIXKIND>   // PIXKIND 0 fp32, 1 bf16 bits, 2 fp16 bits
__global__ __launch_bounds__(256) void im2col_x3_kernel(const PIX* __restrict__ pix, uint16_t* __restrict__ out) {
    const int img = blockIdx.x / 24, py = blockIdx.x % 24;
    const PIX* src = pix + (int64_t)img * 3 * VIT_IMG * VIT_IMG;
    uint16_t* dst = out + ((int64_t)img * VIT_PATCHES + py * 24) * (3 * VIT_PATCH_KPAD);
    for (int idx = threadIdx.x; idx < 42 * VIT_IMG; idx += 256) {
        const int line = idx / VIT_IMG, xcol = idx - line * VIT_IMG;   // line = c*14 + ky
        const int c = line / 14, ky = line - c * 14;
        const int px = xcol / 14, kx = xcol - px * 14;
        const PIX raw = src[((int64_t)c * VIT_IMG + py * 14 + ky) * VIT_IMG + xcol];
        float v;
        if (PIXKIND == 0) v = (float)raw;
        else if (PIXKIND == 2) v = f16_bits_to_f32((uint16_t)raw);
        else v = bf16_bits_to_f32((uint16_t)raw);
        const X3 t = x3_split(v);
        uint16_t* d = dst + px * (3 * VIT_PATCH_KPAD) + c * 196 + ky * 14 + kx;
        d[0] = t.hi; d[VIT_PATCH_KPAD] = t.lo; d[2 * VIT_PATCH_KPAD] = t.hs;
    }
    const int pad = VIT_PATCH_KPAD - VIT_PATCH_K;
    for (int idx = threadIdx.x; idx < 24 * 3 * pad; idx += 256) {
        const int px = idx / (3 * pad), rem = idx % (3 * pad), seg = rem / pad, k = rem % pad;
        dst[px * (3 * VIT_PATCH_KPAD) + seg * VIT_PATCH_KPAD + VIT_PATCH_K + k] = 0;
    }
}
int pg_x3_im2col_launch(const void* pixels, int pix_dtype, void* out3, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    dim3 grid(n_images * 24), block(256);
    if (pix_dtype == PG_DTYPE_F32) hipLaunchKernelGGL((im2col_x3_kernel<float, 0>), grid, block, 0, s, (const float*)pixels, (uint16_t*)out3);
    else if (pix_dtype == PG_DTYPE_BF16) hipLaunchKernelGGL((im2col_x3_kernel<uint16_t, 1>), grid, block, 0, s, (const uint16_t*)pixels, (uint16_t*)out3);
    else if (pix_dtype == PG_DTYPE_F16) hipLaunchKernelGGL((im2col_x3_kernel<uint16_t, 2>), grid, block, 0, s, (const uint16_t*)pixels, (uint16_t*)out3);
    else { pg_set_error("im2col_x3: unsupported pixel dtype %d", pix_dtype); return PG_EINVAL; }
    return pg_check_launch("im2col_x3");
}

// ---- fp32 attention on v_mfma_f32_32x32x2_f32 ------------------------------------------------------------------------------------
// Block = 4 waves = 128 queries of one (image, head); 5 blocks per (image, head).  A wave owns 32 queries.  K / V stream through
// LDS in 32-key tiles (double buffered, next tile's global loads in flight during the MFMAs).
//   S^T[key][query] = K . Q^T      A = K tile (lane: key l%32, d = 2t + l/32 from LDS, row stride 66 floats: conflict-free),
//                                  B = Q^T (32 registers per lane, loaded once: Q[query l%32][2t + l/32], scaled by 1/8)
//   accumulator layout of 32x32: lane holds query l%32 and keys 8j + 4(l/32) + i (register 4j + i) -> a lane owns ONE query:
//   row max / sum are in-lane over 16 keys + one exchange with lane^32.
//   O^T[d][query] += V^T . P^T     B = the lane's own probability register (key 8j + 4(l/32) + i of query l%32 -- exactly the
//                                  k index the B operand of step (j,i) wants from this lane), A = V[that key][d = l%32 (+32)]
//                                  from LDS (row stride 72 floats: the two lane halves land in different banks).
// Online softmax in base 2 on s * log2(e) (v_exp_f32; <= 1 ulp), keys past token 576 masked to -inf.
#define AF_KS 66
#define AF_VS 72
typedef __attribute__((ext_vector_type(16))) float af16;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n_images) {
    __shared__ float Ks[2][32 * AF_KS];
    __shared__ float Vs[2][32 * AF_VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb = blockIdx.x % 5, head = (blockIdx.x / 5) % VIT_HEADS, img = blockIdx.x / (5 * VIT_HEADS);
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t row0 = (int64_t)img * VIT_TOKENS;
    const float* base = qkv + row0 * (3 * VIT_HIDDEN) + head * VIT_HEAD_DIM;
    const int q = qb * 128 + wave * 32 + l32;
    const bool q_ok = q < VIT_TOKENS;
    const bool wave_ok = qb * 128 + wave * 32 < VIT_TOKENS;
    // Q^T operand registers: Q[q][2t + half] * (1/8) * log2(e)  (the 1/8 is exact; one rounding for log2(e))
    float qr[32];
    {
        const float* qp = base + (int64_t)(q_ok ? q : 0) * (3 * VIT_HIDDEN) + half;
#pragma unroll
        for (int t = 0; t < 32; ++t) qr[t] = q_ok ? (qp[2 * t] * 0.125f) * 1.4426950408889634f : 0.f;
    }
    // cooperative tile loads: thread -> key tid/16 (+16), d4 = (tid%16)*4
    const int lk = tid >> 4, ld4 = (tid & 15) * 4;
    f32x4 pk[2], pv[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int key = t * 32 + lk + 16 * h;
            if (key < VIT_TOKENS) {
                const float* p = base + (int64_t)key * (3 * VIT_HIDDEN) + ld4;
                pk[h] = *(const f32x4*)(p + VIT_HIDDEN);
                pv[h] = *(const f32x4*)(p + 2 * VIT_HIDDEN);
            } else { pk[h] = f32x4{0.f, 0.f, 0.f, 0.f}; pv[h] = pk[h]; }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* kd = &Ks[buf][(lk + 16 * h) * AF_KS + ld4];
            kd[0] = pk[h][0]; kd[1] = pk[h][1]; kd[2] = pk[h][2]; kd[3] = pk[h][3];   // stride 66: 8-byte aligned only
            *(f32x4*)&Vs[buf][(lk + 16 * h) * AF_VS + ld4] = pv[h];
        }
    };
    af16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, lsum = 0.f;
    const int NT = (VIT_TOKENS + 31) / 32;     // 19
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < NT; ++t) {
        const int buf = t & 1;
        if (t + 1 < NT) load_tile(t + 1);
        if (wave_ok) {
            af16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
            const float* kp = &Ks[buf][l32 * AF_KS + half];
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * kk], qr[kk], sc, 0, 0, 0);
            if (t == NT - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (key >= VIT_TOKENS) sc[r] = -INFINITY;
                }
            }
            float tm = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tm = fmaxf(tm, sc[r]);
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float mn = fmaxf(m, tm);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);      // m = -inf on the first tile -> 0
            m = mn;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - mn); ps += sc[r]; }
            lsum = lsum * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            const float* vp = &Vs[buf][(4 * half) * AF_VS + l32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = 8 * (r >> 2) + (r & 3);
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[krow * AF_VS], sc[r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[krow * AF_VS + 32], sc[r], o1, 0, 0, 0);
            }
        }
        if (t + 1 < NT) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (!q_ok) return;
    lsum += __shfl_xor(lsum, 32, 64);
    float* op = out + (row0 + q) * VIT_HIDDEN + head * VIT_HEAD_DIM + 4 * half;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 a = {o0[4 * j] / lsum, o0[4 * j + 1] / lsum, o0[4 * j + 2] / lsum, o0[4 * j + 3] / lsum};
        f32x4 b = {o1[4 * j] / lsum, o1[4 * j + 1] / lsum, o1[4 * j + 2] / lsum, o1[4 * j + 3] / lsum};
        *(f32x4*)(op + 8 * j) = a;
        *(f32x4*)(op + 32 + 8 * j) = b;
    }
}
// ---- the same attention on split-fp16 operands (round 5) -----------------------------------------------------------------------
// attention_f32_kernel spends 64 x v_mfma_f32_32x32x2_f32 (64 cycles each on a SIMD) per 32-key tile and wave; on a few images it is
// a quarter of the exact pass.  Here both products run on v_mfma_f32_32x32x16_f16 with every operand split in two fp16 halves and the
// three significant partial products accumulated in fp32 -- x.y ~= xh.yh + xl.yh + xh.yl, the arithmetic of the exact mode's GEMMs:
//   S^T += Kh.Qh^T + Kl.Qh^T + Kh.Ql^T     A = K tile rows (keys) from LDS (fp16 hi / lo images, 16-byte fragments: key l%32, dims
//                                          16 kk + 8 (l/32) ..+8), B = Q^T in registers (split once, scaled by log2(e) / 8 first)
//   O^T += Vh^T.Ph^T + Vl^T.Ph^T + Vh^T.Pl^T   B = the lane's own probabilities: in the 32x32 accumulator layout a lane holds keys
//                                          8 j + 4 (l/32) + i of query l%32, so for k-step s (keys 16 s ..) its eight k slots are
//                                          registers 8 s .. 8 s + 7 = keys {16 s + 4 h + i, 16 s + 8 + 4 h + i}; A = V^T rows (dims)
//                                          from an LDS image that is TRANSPOSED and stored in that very key order (pos_of), so a
//                                          fragment is one 16-byte read again.
// 24 MFMAs of 32 cycles per tile and wave instead of 64 of 64; 39 KB of LDS and ~130 VGPRs: three blocks per CU instead of two.
// Same online softmax (fp32, base 2), same masking of the keys past token 576.  Products lose the lo.lo term (2^-22 relative), as in
// the GEMMs; fp16 subnormal halves are not flushed by the MFMA (tests/test_gpu_precise.py).
#define AX_KS 72                                            // halves per K row in LDS (64 + 8): 144-byte rows, 16-byte aligned
#define AX_VS 40                                            // halves per V^T row (32 keys + 8): 80-byte rows
__device__ __forceinline__ int ax_pos_of(int key) {         // position of tile-local key 0..31 in a V^T row (see above)
    const int w = key & 15;
    return (key & 16) + ((w >> 2) & 1) * 8 + (w >> 3) * 4 + (w & 3);
}

// X3OUT (round 6): the output row is written as the split-fp16 triple the out-projection GEMM reads ([M][3 * 1024] fp16 through
// x3_store4: what split_x3_kernel would make of the fp32 row, bit for bit) instead of fp32 -- one launch and one fp32 round trip less
// per layer.
template <bool X3OUT>
__global__ __launch_bounds__(256, 3) void attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n_images) {
    __shared__ __attribute__((aligned(16))) _Float16 Kh[2][32 * AX_KS];
    __shared__ __attribute__((aligned(16))) _Float16 Kl[2][32 * AX_KS];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[2][64 * AX_VS];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[2][64 * AX_VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qb = blockIdx.x % 5, head = (blockIdx.x / 5) % VIT_HEADS, img = blockIdx.x / (5 * VIT_HEADS);
    const int half = lane >> 5, l32 = lane & 31;
    const int64_t row0 = (int64_t)img * VIT_TOKENS;
    const float* base = qkv + row0 * (3 * VIT_HIDDEN) + head * VIT_HEAD_DIM;
    const int q = qb * 128 + wave * 32 + l32;
    const bool q_ok = q < VIT_TOKENS;
    const bool wave_ok = qb * 128 + wave * 32 < VIT_TOKENS;
    // Q^T fragments: Q[q][16 kk + 8 half + e] * (1/8) * log2(e), split in fp16 halves
    f16x8 qh[4], ql[4];
    {
        const float* qp = base + (int64_t)(q_ok ? q : 0) * (3 * VIT_HIDDEN) + 8 * half;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = q_ok ? (qp[16 * kk + e] * 0.125f) * 1.4426950408889634f : 0.f;
                const _Float16 h = (_Float16)v;
                qh[kk][e] = h;
                ql[kk][e] = (_Float16)(v - (float)h);
            }
    }
    // cooperative tile loads: thread -> key tid/16 (+16), d4 = (tid%16)*4
    const int lk = tid >> 4, ld4 = (tid & 15) * 4;
    f32x4 pk[2], pv[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int key = t * 32 + lk + 16 * h;
            if (key < VIT_TOKENS) {
                const float* p = base + (int64_t)key * (3 * VIT_HIDDEN) + ld4;
                pk[h] = *(const f32x4*)(p + VIT_HIDDEN);
                pv[h] = *(const f32x4*)(p + 2 * VIT_HIDDEN);
            } else { pk[h] = f32x4{0.f, 0.f, 0.f, 0.f}; pv[h] = pk[h]; }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int key = lk + 16 * h;
            _Float16 kh4[4], kl4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 hh = (_Float16)pk[h][e];
                kh4[e] = hh; kl4[e] = (_Float16)(pk[h][e] - (float)hh);
            }
            *(u32x2*)&Kh[buf][key * AX_KS + ld4] = __builtin_bit_cast(u32x2, *(const f16x4_t*)kh4);
            *(u32x2*)&Kl[buf][key * AX_KS + ld4] = __builtin_bit_cast(u32x2, *(const f16x4_t*)kl4);
            const int pos = ax_pos_of(key);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 hh = (_Float16)pv[h][e];
                Vh[buf][(ld4 + e) * AX_VS + pos] = hh;
                Vl[buf][(ld4 + e) * AX_VS + pos] = (_Float16)(pv[h][e] - (float)hh);
            }
        }
    };
    af16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m = -INFINITY, lsum = 0.f;
    const int NT = (VIT_TOKENS + 31) / 32;     // 19
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < NT; ++t) {
        const int buf = t & 1;
        if (t + 1 < NT) load_tile(t + 1);
        if (wave_ok) {
            af16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.f;
            const _Float16* khp = &Kh[buf][l32 * AX_KS + 8 * half];
            const _Float16* klp = &Kl[buf][l32 * AX_KS + 8 * half];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f16x8 ah = *(const f16x8*)(khp + 16 * kk), al = *(const f16x8*)(klp + 16 * kk);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[kk], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[kk], sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[kk], sc, 0, 0, 0);
            }
            if (t == NT - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (key >= VIT_TOKENS) sc[r] = -INFINITY;
                }
            }
            float tm = sc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tm = fmaxf(tm, sc[r]);
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const float mn = fmaxf(m, tm);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);      // m = -inf on the first tile -> 0
            m = mn;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - mn); ps += sc[r]; }
            lsum = lsum * alpha + ps;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 ph, pl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 hh = (_Float16)sc[8 * s2 + e];
                    ph[e] = hh;
                    pl[e] = (_Float16)(sc[8 * s2 + e] - (float)hh);
                }
                const int voff = 16 * s2 + 8 * half;
                const f16x8 v0h = *(const f16x8*)&Vh[buf][l32 * AX_VS + voff], v0l = *(const f16x8*)&Vl[buf][l32 * AX_VS + voff];
                const f16x8 v1h = *(const f16x8*)&Vh[buf][(l32 + 32) * AX_VS + voff], v1l = *(const f16x8*)&Vl[buf][(l32 + 32) * AX_VS + voff];
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, ph, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l, ph, o0, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, pl, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, ph, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l, ph, o1, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, pl, o1, 0, 0, 0);
            }
        }
        if (t + 1 < NT) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (!q_ok) return;
    lsum += __shfl_xor(lsum, 32, 64);
    float* op = out + (row0 + q) * VIT_HIDDEN + head * VIT_HEAD_DIM + 4 * half;
    uint16_t* op3 = (uint16_t*)out + (row0 + q) * (3 * VIT_HIDDEN);
    const int c0 = head * VIT_HEAD_DIM + 4 * half;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 a = {o0[4 * j] / lsum, o0[4 * j + 1] / lsum, o0[4 * j + 2] / lsum, o0[4 * j + 3] / lsum};
        f32x4 b = {o1[4 * j] / lsum, o1[4 * j + 1] / lsum, o1[4 * j + 2] / lsum, o1[4 * j + 3] / lsum};
        if constexpr (X3OUT) {
            x3_store4(op3, VIT_HIDDEN, c0 + 8 * j, a);
            x3_store4(op3, VIT_HIDDEN, c0 + 32 + 8 * j, b);
        } else {
            *(f32x4*)(op + 8 * j) = a;
            *(f32x4*)(op + 32 + 8 * j) = b;
        }
    }
}

// PIGEON_EXACT_ATTN=f32 / pg_tune_exact_attention(1) select the fp32-MFMA kernel (the A/B arm and the checker of the split one)
static int g_exact_attn_f32 = -1;
extern "C" int pg_tune_exact_attention(int use_f32_mfma) {
    g_exact_attn_f32 = use_f32_mfma ? 1 : 0;
    return PG_OK;
}
static bool exact_attn_f32() {
    if (g_exact_attn_f32 < 0) { const char* e = getenv("PIGEON_EXACT_ATTN"); g_exact_attn_f32 = (e && e[0] == 'f') ? 1 : 0; }
    return g_exact_attn_f32 != 0;
}
int pg_attention_f32_launch(const float* qkv, float* out, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    if (exact_attn_f32()) hipLaunchKernelGGL(attention_f32_kernel, dim3(n_images * VIT_HEADS * 5), dim3(256), 0, s, qkv, out, n_images);
    else hipLaunchKernelGGL(attention_x3_kernel<false>, dim3(n_images * VIT_HEADS * 5), dim3(256), 0, s, qkv, out, n_images);
    return pg_check_launch("attention_f32");
}
// the same attention with the output written as the triple [M][3 * 1024] fp16 (attention_x3_kernel<true>); not available with the
// fp32-MFMA A/B arm selected -- the caller then runs pg_attention_f32_launch + pg_x3_split_launch
bool pg_attention_x3out_available() { return !exact_attn_f32(); }
int pg_attention_x3out_launch(const float* qkv, void* out3, int n_images, hipStream_t s) {
    if (n_images <= 0) return PG_OK;
    if (exact_attn_f32()) { pg_set_error("attention_x3out: not available with the fp32-MFMA attention selected"); return PG_ESTATE; }
    hipLaunchKernelGGL(attention_x3_kernel<true>, dim3(n_images * VIT_HEADS * 5), dim3(256), 0, s, qkv, (float*)out3, n_images);
    return pg_check_launch("attention_x3out");
}
