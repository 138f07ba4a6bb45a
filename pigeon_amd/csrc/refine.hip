// refine.hip -- ProtoRefiner: brute-force L2 nearest-prototype search over a CSR prototype bank, the
// within-cluster (FARTHEST member) step, the un-shifted temperature softmax and the haversine veto (float64 with the reference's fp32 promotion of the refined point).
//
// Replaces the reference's Python double loop models/proto_refiner.py:154-222 (one H2D copy + >=3 .item()
// syncs per (sample, candidate)) with two launches and no host round trips:
//
//   refine_candidates_kernel  grid = B x topk blocks (4 waves each).  Block (b, j) owns candidate cell
//       cand[b][j]: every wave streams whole 4 KB prototype rows (16 floats per lane as four 16-byte loads),
//       reduces sum((p-q)^2) across the wave with shuffles and keeps a running (min distance, first index)
//       -- proto_refiner.py:176-181 (`-cdist`, max, argmax => nearest, lowest index on ties).  Then, if the
//       winning prototype has count > 1, the same block gathers that cluster's member rows from the training
//       bank through member_idx and keeps the (MAX distance, first index) member -- :244-255 (the reference
//       takes argmax of the POSITIVE distances: the farthest member).  Empty cell -> score
//       -100000, prediction (0,0) -- :168-174.  This kernel is pure HBM streaming: 4096 B per row touched.
//   refine_select_kernel  one lane per query: probs = exp(score/T)/sum (no max shift, :355-357), final =
//       c_probs*probs (:192), argmax with torch semantics (first max, NaN counts as max), haversine veto with
//       the reference's dtypes (:198-205, geo_utils.py:40-55: initial point float64, refined point float32 through
//       deg2rad and cos, float64 after), final argmax (:219), outputs (:221-222).
#include "common.h"
#include "pigeon_internal.h"
#include <cmath>

#define RF_DIM 1024

__device__ __forceinline__ void load_row16(const float* __restrict__ p, int lane, f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(p + i * 256 + lane * 4);
}

__device__ __forceinline__ float row_sqdist(const float* __restrict__ row, int lane, const f32x4 (&q)[4]) {
    f32x4 v[4];
    load_row16(row, lane, v);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - q[i][e]; s = fmaf(d, d, s); }
    return wave_sum(s);
}

// "a beats b" for a (min value, lowest index) search in which NaN beats every number and the FIRST NaN wins: what
// torch.max / torch.argmax over `-distance` do on the reference side (proto_refiner.py:180-181, :254).
__device__ __forceinline__ bool nan_aware_less(float v, long long i, float bv, long long bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v < bv || (v == bv && i < bi);
}

// scratch layout per (b, j): [score, lng, lat, rows] -- rows = bank rows this (query, candidate) streamed (prototypes of the
// cell + members of the chosen cluster when count > 1): the algorithmic-bytes bookkeeping of the benchmark (4096 B each)
__global__ __launch_bounds__(256) void refine_candidates_kernel(pg_bank bank, const float* __restrict__ q, int P,
                                                                const int64_t* __restrict__ cand, int k, int topk,
                                                                float* __restrict__ scratch) {
    __shared__ float red_d[4];
    __shared__ long long red_i[4];
    __shared__ long long chosen;
    const int b = blockIdx.x / topk, j = blockIdx.x % topk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* outp = scratch + ((int64_t)b * topk + j) * 4;

    const int64_t cell = cand[(int64_t)b * k + j];
    int64_t s = 0, e = 0;
    if (cell >= 0 && cell < bank.num_cells) { s = bank.cell_off[cell]; e = bank.cell_off[cell + 1]; }
    if (e <= s) {                                           // empty geocell (or out-of-range id)
        if (tid == 0) { outp[0] = -100000.0f; outp[1] = 0.f; outp[2] = 0.f; outp[3] = 0.f; }
        return;
    }

    // query: mean over the P panels (proto_refiner.py:139-140), kept in registers
    f32x4 qv[4];
    {
        const float* qp = q + (int64_t)b * P * RF_DIM;
        load_row16(qp, lane, qv);
        for (int p = 1; p < P; ++p) {
            f32x4 t[4];
            load_row16(qp + (int64_t)p * RF_DIM, lane, t);
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] += t[i];
        }
        if (P > 1) {
            const float inv = 1.0f / (float)P;
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] *= inv;
        }
    }

    // ---- phase 1: nearest prototype of the cell (min distance, lowest index on ties) ----
    // torch.max / torch.argmax treat NaN as the maximum (first NaN wins): a NaN distance "wins" here too, so a NaN query
    // propagates NaN scores exactly like the reference instead of leaving the sentinel index behind
    float best = INFINITY; long long bi = 0x7fffffffffffffffLL;
    {
        // two rows in flight per wave (8 KB): with one 4 KB row per wave the kernel is latency-bound (10 resident waves per CU
        // keep ~40 KB in flight where HBM needs ~60 KB per CU to stay busy); rows of a wave are visited in ascending order, so
        // the (min, first index) result is what the single-row loop gives
        int64_t r = s + wave;
        for (; r + 4 < e; r += 8) {
            f32x4 v0[4], v1[4];
            load_row16(bank.proto_emb + r * RF_DIM, lane, v0);
            load_row16(bank.proto_emb + (r + 4) * RF_DIM, lane, v1);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float d0 = v0[i][e4] - qv[i][e4]; s0 = fmaf(d0, d0, s0);
                    const float d1 = v1[i][e4] - qv[i][e4]; s1 = fmaf(d1, d1, s1);
                }
            const float dd0 = sqrtf(wave_sum(s0)), dd1 = sqrtf(wave_sum(s1));
            if (nan_aware_less(dd0, r, best, bi)) { best = dd0; bi = r; }
            if (nan_aware_less(dd1, r + 4, best, bi)) { best = dd1; bi = r + 4; }
        }
        for (; r < e; r += 4) {
            const float d = sqrtf(row_sqdist(bank.proto_emb + r * RF_DIM, lane, qv));
            if (nan_aware_less(d, r, best, bi)) { best = d; bi = r; }
        }
    }
    if (lane == 0) { red_d[wave] = best; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        float bd = red_d[0]; long long bx = red_i[0];
        for (int w = 1; w < 4; ++w) {
            const float d = red_d[w]; const long long x = red_i[w];
            if (x == 0x7fffffffffffffffLL) continue;
            if (bx == 0x7fffffffffffffffLL || nan_aware_less(d, x, bd, bx)) { bd = d; bx = x; }
        }
        chosen = bx;
        outp[0] = -bd;                                      // score = max(-distance)
    }
    __syncthreads();
    const int64_t pid = chosen;
    if (pid < s || pid >= e) {                              // cannot happen (e > s); guard the gathers below anyway
        if (tid == 0) { outp[1] = 0.f; outp[2] = 0.f; outp[3] = 0.f; }
        return;
    }
    const int cnt = bank.proto_count[pid];
    if (cnt == 1) {                                         // proto_refiner.py:245-246
        if (tid == 0) { outp[1] = bank.proto_lnglat[2 * pid]; outp[2] = bank.proto_lnglat[2 * pid + 1]; outp[3] = (float)(e - s); }
        return;
    }

    // ---- phase 2: farthest member of the chosen cluster (max distance, lowest position on ties) ----
    const int64_t ms = bank.member_off[pid], me = bank.member_off[pid + 1];
    float far = -INFINITY; long long fi = 0x7fffffffffffffffLL;
    for (int64_t r = ms + wave; r < me; r += 4) {
        int64_t tr = bank.member_idx[r];
        tr = tr < 0 ? 0 : (tr >= bank.num_train ? bank.num_train - 1 : tr);      // never fault on a corrupt member list
        const float d = sqrtf(row_sqdist(bank.train_emb + tr * RF_DIM, lane, qv));
        if (nan_aware_less(-d, r, -far, fi)) { far = d; fi = r; }
    }
    __syncthreads();
    if (lane == 0) { red_d[wave] = far; red_i[wave] = fi; }
    __syncthreads();
    if (tid == 0) {
        float bd = red_d[0]; long long bx = red_i[0];
        for (int w = 1; w < 4; ++w) {
            const float d = red_d[w]; const long long x = red_i[w];
            if (x == 0x7fffffffffffffffLL) continue;
            if (bx == 0x7fffffffffffffffLL || nan_aware_less(-d, x, -bd, bx)) { bd = d; bx = x; }
        }
        float lng = 0.f, lat = 0.f;
        if (bx != 0x7fffffffffffffffLL) {
            int64_t tr = bank.member_idx[bx];
            tr = tr < 0 ? 0 : (tr >= bank.num_train ? bank.num_train - 1 : tr);
            lng = bank.train_lnglat[2 * tr]; lat = bank.train_lnglat[2 * tr + 1];
        }
        outp[1] = lng; outp[2] = lat; outp[3] = (float)((e - s) + (me - ms));
    }
}

// torch.argmax semantics over n <= 64 values: first maximum; a NaN is the maximum (first NaN wins).
__device__ __forceinline__ int argmax_torch(const float* v, int n) {
    int bi = 0; float bv = v[0];
    for (int i = 1; i < n; ++i) {
        const float x = v[i];
        if (bv != bv) break;                                // already NaN -> stays
        if (x != x || x > bv) { bv = x; bi = i; }
    }
    return bi;
}

// The veto distance with the reference's dtype promotion.  models/proto_refiner.py:198-202 calls
// haversine(initial_LLH, refined_LLH) (preprocessing/geo_utils.py:40-55) with x = the float64 initial prediction and
// y = torch.tensor(top_preds[...]) = a FLOAT32 tensor: torch.deg2rad(y) and torch.cos(y_rad[:,1]) are evaluated in fp32
// (deg2rad multiplies by pi/180 rounded to the tensor dtype), `y_rad - x_rad` and everything after promote to float64.
// cos of the fp32 latitude is taken as the correctly rounded fp32 value (double cos, rounded once): torch's CPU kernel
// (Sleef, <= 1 ulp) agrees with it except for rare last-bit cases, which no device libm could reproduce anyway.
__device__ __forceinline__ double haversine_km(double lng1, double lat1, float lng2, float lat2) {
    const double d2r = 0.017453292519943295769236907684886127134428718885417;   // M_PI / 180
    const double x0 = lng1 * d2r, x1 = lat1 * d2r;
    const float y0 = lng2 * (float)d2r, y1 = lat2 * (float)d2r;                  // fp32 deg2rad
    const double dl = (double)y0 - x0, dp = (double)y1 - x1;
    const double sp = sin(dp / 2), sl = sin(dl / 2);
    const float cy = (float)cos((double)y1);                                      // fp32 cos(lat of the refined point)
    const double a = sp * sp + cos(x1) * (double)cy * (sl * sl);
    const double c = 2 * asin(sqrt(a));
    return (6378137.0 * c) / 1000;
}

__global__ __launch_bounds__(64) void refine_select_kernel(const float* __restrict__ scratch, int B, int k, int topk,
                                                           const int64_t* __restrict__ cand,
                                                           const float* __restrict__ cand_prob,
                                                           const double* __restrict__ init_llh, float temperature,
                                                           double max_km, float* __restrict__ out_llh,
                                                           int64_t* __restrict__ out_cell, int32_t* __restrict__ out_choice) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* sc = scratch + (int64_t)b * topk * 4;
    float cp[64], fin[64];
    float sum = 0.f;
    for (int j = 0; j < topk; ++j) {
        const float ex = expf(sc[4 * j] / temperature);     // torch.exp(input / T), fp32
        fin[j] = ex;
        sum += ex;                                          // torch.sum, sequential for <= 64 elements
    }
    for (int j = 0; j < topk; ++j) {
        cp[j] = cand_prob ? cand_prob[(int64_t)b * k + j] : (j == 0 ? 1.0f : 0.0f);
        fin[j] = cp[j] * (fin[j] / sum);
    }
    const int refined = argmax_torch(fin, topk);
    const float rlng = sc[4 * refined + 1], rlat = sc[4 * refined + 2];
    const double dist = haversine_km(init_llh[2 * b], init_llh[2 * b + 1], rlng, rlat);
    int choice = refined;
    if (dist > max_km) choice = argmax_torch(cp, topk);    // veto: fall back to the geocell probabilities
    out_llh[2 * b] = sc[4 * choice + 1];
    out_llh[2 * b + 1] = sc[4 * choice + 2];
    out_cell[b] = cand[(int64_t)b * k + choice];
    out_choice[b] = choice;
}

extern "C" int pg_refine_forward(const pg_bank* bank, const float* q, int B, int P, const double* init_llh,
                                 const int64_t* cand, const float* cand_prob, int k, int topk, float temperature,
                                 double max_refine_km, float* scratch, float* out_llh, int64_t* out_cell,
                                 int32_t* out_choice, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B < 0) { pg_set_error("refine: B = %d", B); return PG_EINVAL; }
    if (B == 0) return PG_OK;                              // an empty batch is a no-op: its (empty) buffers may be NULL
    if (!bank || !q || !init_llh || !cand || !scratch || !out_llh || !out_cell || !out_choice) {
        pg_set_error("refine: null pointer argument"); return PG_EINVAL;
    }
    if (topk < 1 || topk > k || topk > 64 || P < 1) {
        pg_set_error("refine: need 1 <= topk <= min(k,64) and P >= 1 (topk=%d k=%d P=%d)", topk, k, P); return PG_EINVAL;
    }
    hipLaunchKernelGGL(refine_candidates_kernel, dim3((unsigned)B * topk), dim3(256), 0, s, *bank, q, P, cand, k, topk, scratch);
    int rc = pg_check_launch("refine_candidates");
    if (rc) return rc;
    hipLaunchKernelGGL(refine_select_kernel, dim3((B + 63) / 64), dim3(64), 0, s, scratch, B, k, topk, cand, cand_prob,
                       init_llh, temperature, max_refine_km, out_llh, out_cell, out_choice);
    return pg_check_launch("refine_select");
}
