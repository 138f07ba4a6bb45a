// head.hip -- SuperGuessr geocell classification head, fp32 end to end.
//
// Replaces reference models/super_guessr.py:437 (mean over the 4 panorama panels), :447 (cell_layer Linear),
// :448 (softmax), :454 (argmax), :455 (index_select of the float64 centroids) and :459 (topk).
// fp32 is deliberate: the geocell argmax must be identical to the reference's.
//
// Kernel 1: logits[B,C] = mean_p(emb[B,P,1024]) . W[C,1024]^T + bias.   LDS-tiled fp32 FMA GEMM (64x64 tile,
//           BK=32, 4x4 outputs per thread); the whole op is ~2.6 GFLOP / 41 MB per batch = <0.2% of a step.
// Kernel 2: one block per row: max, sum-exp, probabilities, then k rounds of block-wide (value desc, index asc)
//           selection -> top-k probabilities/indices, argmax = first selection, centroid gather (float64).
#include "common.h"
#include "pigeon_internal.h"
#include <cfloat>

#define HT 64
#define HK 32

__global__ __launch_bounds__(256) void head_logits_kernel(const float* __restrict__ emb, int B, int P,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          int C, float* __restrict__ logits) {
    __shared__ float As[HK][HT + 1];   // [k][b]
    __shared__ float Ws[HK][HT + 1];   // [k][c]
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * HT, b0 = blockIdx.y * HT;
    const int tx = tid & 15, ty = tid >> 4;            // thread owns b = ty*4..+4, c = tx*4..+4
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    const float invP = 1.0f / (float)P;
    for (int k0 = 0; k0 < VIT_HIDDEN; k0 += HK) {
        // stage 64 x 32 of A (panel-averaged) and of W; thread loads 8 elements of each
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + i * 256;             // 0..2047
            const int r = idx >> 5, k = idx & 31;      // r: row in tile, k: 0..31 (coalesced along k)
            float a = 0.f;
            if (b0 + r < B) {
                const float* e = emb + ((int64_t)(b0 + r) * P) * VIT_HIDDEN + k0 + k;
                for (int p = 0; p < P; ++p) a += e[(int64_t)p * VIT_HIDDEN];
                a *= invP;
            }
            As[k][r] = a;
            Ws[k][r] = (c0 + r < C) ? W[(int64_t)(c0 + r) * VIT_HIDDEN + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HK; ++k) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = Ws[k][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b0 + ty * 4 + i;
        if (b >= B) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + tx * 4 + j;
            if (c < C) logits[(int64_t)b * C + c] = acc[i][j] + bias[c];
        }
    }
}

struct ValIdx { float v; int i; };
// (value desc, index asc); a NaN ranks above every number (torch.argmax / torch.topk semantics), lowest index first
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return (v > bv) || (v == bv && i < bi);
}
__device__ __forceinline__ ValIdx block_argbest(ValIdx x, ValIdx* red /*[4]*/) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(x.v, o, 64);
        const int oi = __shfl_xor(x.i, o, 64);
        if (better(ov, oi, x.v, x.i)) { x.v = ov; x.i = oi; }
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = x;
    __syncthreads();
    ValIdx r = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) if (better(red[w].v, red[w].i, r.v, r.i)) r = red[w];
    return r;
}

// one block per batch row; the row's C probabilities live in dynamic LDS (GLOBAL = false: C <= 38 400, every geocell set of the
// reference) or, for larger heads, in a row of a device scratch matrix (GLOBAL = true, round 4: same arithmetic and selection order,
// the k selection passes then stream the row from L2 / HBM instead of LDS)
template <bool GLOBAL>
__global__ __launch_bounds__(256) void head_row_kernel(const float* __restrict__ logits, int C, int k,
                                                       const double* __restrict__ centroids,
                                                       float* __restrict__ topk_val, int64_t* __restrict__ topk_idx,
                                                       int64_t* __restrict__ argmax_out, double* __restrict__ pred_llh,
                                                       float* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) float probs_lds[];
    float* probs = GLOBAL ? scratch + (int64_t)blockIdx.x * C : probs_lds;
    __shared__ ValIdx red[4];
    __shared__ float redf[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)b * C;
    // max
    float mx = -FLT_MAX;
    for (int c = tid; c < C; c += 256) { const float v = row[c]; probs[c] = v; mx = fmaxf(mx, v); }
    mx = wave_max(mx);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    __syncthreads();
    // sum of exp(x - max)
    float sm = 0.f;
    for (int c = tid; c < C; c += 256) { const float e = expf(probs[c] - mx); probs[c] = e; sm += e; }
    sm = wave_sum(sm);
    if ((tid & 63) == 0) redf[tid >> 6] = sm;
    __syncthreads();
    sm = (redf[0] + redf[1]) + (redf[2] + redf[3]);
    for (int c = tid; c < C; c += 256) probs[c] = probs[c] / sm;
    __syncthreads();
    // k rounds of selection
    for (int r = 0; r < k; ++r) {
        ValIdx best; best.v = -1.f; best.i = 0x7fffffff;
        for (int c = tid; c < C; c += 256) {
            const float v = probs[c];
            if (better(v, c, best.v, best.i)) { best.v = v; best.i = c; }
        }
        best = block_argbest(best, red);
        if (tid == 0) {
            if (best.i < 0 || best.i >= C) best.i = r < C ? r : C - 1;   // unreachable (k <= C, NaN ranks first); never index out of range
            topk_val[(int64_t)b * k + r] = best.v;
            topk_idx[(int64_t)b * k + r] = best.i;
            if (r == 0) {
                argmax_out[b] = best.i;
                pred_llh[2 * b] = centroids[2 * (int64_t)best.i];
                pred_llh[2 * b + 1] = centroids[2 * (int64_t)best.i + 1];
            }
            probs[best.i] = -2.f;                      // retire (probabilities are >= 0)
        }
        if (GLOBAL) __threadfence_block();             // the retirement is a global store: visible to the block's next pass
        __syncthreads();
    }
}

extern "C" int pg_head_forward(const float* emb, int B, int P, const float* W, const float* bias,
                               const double* centroids, int C, int k, float* logits, float* topk_val,
                               int64_t* topk_idx, int64_t* argmax, double* pred_llh, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B < 0) { pg_set_error("head: B = %d", B); return PG_EINVAL; }
    if (B == 0) return PG_OK;                              // an empty batch is a no-op: its (empty) buffers may be NULL
    if (!emb || !W || !bias || !centroids || !logits || !topk_val || !topk_idx || !argmax || !pred_llh) {
        pg_set_error("head: null pointer argument"); return PG_EINVAL;
    }
    if (P < 1 || C < 1 || k < 1 || k > C) { pg_set_error("head: bad P=%d C=%d k=%d", P, C, k); return PG_EINVAL; }
    const size_t lds = (size_t)C * sizeof(float);
    const bool big = lds > 150 * 1024;                     // > 38 400 geocells: probabilities in a stream-ordered device scratch
    dim3 g1((C + HT - 1) / HT, (B + HT - 1) / HT);
    hipLaunchKernelGGL(head_logits_kernel, g1, dim3(256), 0, s, emb, B, P, W, bias, C, logits);
    int rc = pg_check_launch("head_logits");
    if (rc) return rc;
    if (big) {
        float* scratch = nullptr;
        PG_HIP(hipMallocAsync((void**)&scratch, (size_t)B * C * sizeof(float), s));
        hipLaunchKernelGGL(head_row_kernel<true>, dim3(B), dim3(256), 0, s, logits, C, k, centroids, topk_val, topk_idx, argmax, pred_llh, scratch);
        rc = pg_check_launch("head_row (large C)");
        (void)hipFreeAsync(scratch, s);
        return rc;
    }
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        PG_HIP(hipFuncSetAttribute((const void*)head_row_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    hipLaunchKernelGGL(head_row_kernel<false>, dim3(B), dim3(256), lds, s, logits, C, k, centroids, topk_val, topk_idx, argmax, pred_llh,
                       (float*)nullptr);
    return pg_check_launch("head_row");
}

// ---- certainty of the top-1 (round 4) ----------------------------------------------------------------------------------------
// margin[b]  = logit(top-1) - logit(top-2) of row b (value desc, index asc: the same order as the selection above);
// sens[b]    = |mean_p emb[b]|_2 * |W[top1] - W[top2]|_2 / sqrt(1024): the size of the margin change a unit RELATIVE error of the
//              embedding in a random direction causes.  A caller holding a bound eps on the encoder's relative embedding error
//              (pigeon_amd/super_guessr.py: calibrated against the reference fixtures) calls the top-1 CERTAIN when
//              margin > kappa * eps * sens, and re-encodes the rest in the exact mode (pg_vit_forward_precise).
// One block per row; C == 1: margin = +inf, sens = 0, top2 = top1.
__global__ __launch_bounds__(256) void head_margin_kernel(const float* __restrict__ logits, int C, const float* __restrict__ emb, int P,
                                                          const float* __restrict__ W, float* __restrict__ margin,
                                                          float* __restrict__ sens, int64_t* __restrict__ top2) {
    __shared__ ValIdx red[4];
    __shared__ float redf[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (int64_t)b * C;
    ValIdx b1; b1.v = -FLT_MAX; b1.i = 0x7fffffff;
    bool any = false;
    for (int c = tid; c < C; c += 256) { const float v = row[c]; if (!any || better(v, c, b1.v, b1.i)) { b1.v = v; b1.i = c; any = true; } }
    if (!any) { b1.v = -FLT_MAX; b1.i = 0x7fffffff; }
    const ValIdx t1 = block_argbest(b1, red);
    ValIdx b2; b2.v = -FLT_MAX; b2.i = 0x7fffffff;
    any = false;
    for (int c = tid; c < C; c += 256) {
        if (c == t1.i) continue;
        const float v = row[c];
        if (!any || better(v, c, b2.v, b2.i)) { b2.v = v; b2.i = c; any = true; }
    }
    if (!any) { b2.v = -FLT_MAX; b2.i = 0x7fffffff; }
    __syncthreads();
    const ValIdx t2 = block_argbest(b2, red);
    const bool have2 = C > 1 && t2.i >= 0 && t2.i < C;
    // |mean_p e|^2 and |w1 - w2|^2: thread owns 4 columns
    float se = 0.f, sw = 0.f;
    {
        const int c0 = tid * 4;
        const float invP = 1.0f / (float)P;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = 0.f;
            for (int p = 0; p < P; ++p) a += emb[((int64_t)b * P + p) * VIT_HIDDEN + c0 + e];
            a *= invP;
            se += a * a;
            if (have2) { const float d = W[(int64_t)t1.i * VIT_HIDDEN + c0 + e] - W[(int64_t)t2.i * VIT_HIDDEN + c0 + e]; sw += d * d; }
        }
    }
    se = wave_sum(se); sw = wave_sum(sw);
    __syncthreads();
    if ((tid & 63) == 0) { redf[tid >> 6] = se; redf[4 + (tid >> 6)] = sw; }
    __syncthreads();
    if (tid == 0) {
        se = (redf[0] + redf[1]) + (redf[2] + redf[3]);
        sw = (redf[4] + redf[5]) + (redf[6] + redf[7]);
        margin[b] = have2 ? t1.v - t2.v : INFINITY;
        sens[b] = have2 ? sqrtf(se) * sqrtf(sw) * (1.0f / 32.0f) : 0.f;
        if (top2) top2[b] = have2 ? t2.i : t1.i;
    }
}

extern "C" int pg_head_margin(const float* logits, int B, int C, const float* emb, int P, const float* W, float* margin, float* sens,
                              int64_t* top2, void* stream) {
    if (B < 0) { pg_set_error("head_margin: B = %d", B); return PG_EINVAL; }
    if (B == 0) return PG_OK;
    if (!logits || !emb || !W || !margin || !sens) { pg_set_error("head_margin: null pointer argument"); return PG_EINVAL; }
    if (P < 1 || C < 1) { pg_set_error("head_margin: bad P=%d C=%d", P, C); return PG_EINVAL; }
    hipLaunchKernelGGL(head_margin_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, C, emb, P, W, margin, sens, top2);
    return pg_check_launch("head_margin");
}
