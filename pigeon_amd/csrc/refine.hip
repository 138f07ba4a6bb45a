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
//
// Round 5: pg_refine_forward_ex evaluates n_eval >= topk candidates per query (the extra ones never take part in the selection) and
// leaves, per (query, candidate), what the certainty pass (certainty.hip) needs: the runner-up prototype and its distance, the two
// farthest members of the chosen cluster.  Same kernels, same arithmetic, same selection: `EXT` only widens the scratch record.
#include "common.h"
#include "pigeon_internal.h"
#include <cmath>

#define RF_DIM 1024
#define RF_SENT 0x7fffffffffffffffLL

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

// best and runner-up of a (min value, lowest index) search under nan_aware_less; both start at (+inf, sentinel)
struct Top2 {
    float d1, d2;
    long long i1, i2;
    __device__ __forceinline__ void clear() { d1 = d2 = INFINITY; i1 = i2 = RF_SENT; }
    __device__ __forceinline__ void push(float d, long long i) {
        if (i == RF_SENT) return;
        const bool first = i1 == RF_SENT || nan_aware_less(d, i, d1, i1);
        const bool second = i2 == RF_SENT || nan_aware_less(d, i, d2, i2);
        const float nd2 = first ? d1 : (second ? d : d2);
        const long long ni2 = first ? i1 : (second ? i : i2);
        d1 = first ? d : d1; i1 = first ? i : i1;
        d2 = nd2; i2 = ni2;
    }
};

// scratch layout per (b, j): [score, lng, lat, rows] -- rows = bank rows this (query, candidate) streamed (prototypes of the
// cell + members of the chosen cluster when count > 1): the algorithmic-bytes bookkeeping of the benchmark (4096 B each).
// EXT (pg_refine_forward_ex): 12 floats per (b, j), the four above plus
//   [4] d2     distance of the runner-up prototype of the cell (+inf: the cell has one prototype)
//   [5] pid1   bank row of the nearest prototype, [6] pid2 of the runner-up (int32 bit patterns; -1 = none)
//   [7] far1   distance of the farthest member of the chosen cluster, [8] far2 of the second farthest (-1: none / count == 1)
//   [9] t1, [10] t2  their training-bank rows (int32 bit patterns; -1 = none),   [11] count of the chosen prototype (int32 bits)
template <bool EXT>
__global__ __launch_bounds__(256) void refine_candidates_kernel(pg_bank bank, const float* __restrict__ q, int P,
                                                                const int64_t* __restrict__ cand, int k, int topk,
                                                                float* __restrict__ scratch) {
    constexpr int SC = EXT ? 12 : 4;
    __shared__ float red_d[8];                              // per wave: best, runner-up (plain arrays: no struct copies through LDS)
    __shared__ long long red_i[8];
    __shared__ long long chosen;
    const int b = blockIdx.x / topk, j = blockIdx.x % topk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* outp = scratch + ((int64_t)b * topk + j) * SC;

    const int64_t cell = cand[(int64_t)b * k + j];
    int64_t s = 0, e = 0;
    if (cell >= 0 && cell < bank.num_cells) { s = bank.cell_off[cell]; e = bank.cell_off[cell + 1]; }
    if (e <= s) {                                           // empty geocell (or out-of-range id)
        if (tid == 0) {
            outp[0] = -100000.0f; outp[1] = 0.f; outp[2] = 0.f; outp[3] = 0.f;
            if (EXT) {
                outp[4] = INFINITY; outp[5] = __int_as_float(-1); outp[6] = __int_as_float(-1); outp[7] = -1.f; outp[8] = -1.f;
                outp[9] = __int_as_float(-1); outp[10] = __int_as_float(-1); outp[11] = __int_as_float(0);
            }
        }
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
    Top2 near; near.clear();
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
            near.push(dd0, r);
            near.push(dd1, r + 4);
        }
        for (; r < e; r += 4) near.push(sqrtf(row_sqdist(bank.proto_emb + r * RF_DIM, lane, qv)), r);
    }
    if (lane == 0) { red_d[2 * wave] = near.d1; red_i[2 * wave] = near.i1; red_d[2 * wave + 1] = near.d2; red_i[2 * wave + 1] = near.i2; }
    __syncthreads();
    if (tid == 0) {
        Top2 m; m.clear();
        for (int x = 0; x < 8; ++x) m.push(red_d[x], red_i[x]);
        chosen = m.i1;
        outp[0] = -m.d1;                                    // score = max(-distance)
        if (EXT) {
            outp[4] = m.i2 == RF_SENT ? INFINITY : m.d2;
            outp[5] = __int_as_float((int)m.i1);
            outp[6] = __int_as_float(m.i2 == RF_SENT ? -1 : (int)m.i2);
        }
    }
    __syncthreads();
    const int64_t pid = chosen;
    if (pid < s || pid >= e) {                              // cannot happen (e > s); guard the gathers below anyway
        if (tid == 0) {
            outp[1] = 0.f; outp[2] = 0.f; outp[3] = 0.f;
            if (EXT) { outp[7] = -1.f; outp[8] = -1.f; outp[9] = __int_as_float(-1); outp[10] = __int_as_float(-1); outp[11] = __int_as_float(0); }
        }
        return;
    }
    const int cnt = bank.proto_count[pid];
    if (cnt == 1) {                                         // proto_refiner.py:245-246
        if (tid == 0) {
            outp[1] = bank.proto_lnglat[2 * pid]; outp[2] = bank.proto_lnglat[2 * pid + 1]; outp[3] = (float)(e - s);
            if (EXT) { outp[7] = -1.f; outp[8] = -1.f; outp[9] = __int_as_float(-1); outp[10] = __int_as_float(-1); outp[11] = __int_as_float(1); }
        }
        return;
    }

    // ---- phase 2: farthest member of the chosen cluster (max distance, lowest position on ties) ----
    const int64_t ms = bank.member_off[pid], me = bank.member_off[pid + 1];
    Top2 far; far.clear();                                  // keyed by -distance: (max distance, first position)
    for (int64_t r = ms + wave; r < me; r += 4) {
        int64_t tr = bank.member_idx[r];
        tr = tr < 0 ? 0 : (tr >= bank.num_train ? bank.num_train - 1 : tr);      // never fault on a corrupt member list
        far.push(-sqrtf(row_sqdist(bank.train_emb + tr * RF_DIM, lane, qv)), r);
    }
    __syncthreads();
    if (lane == 0) { red_d[2 * wave] = far.d1; red_i[2 * wave] = far.i1; red_d[2 * wave + 1] = far.d2; red_i[2 * wave + 1] = far.i2; }
    __syncthreads();
    if (tid == 0) {
        Top2 m; m.clear();
        for (int x = 0; x < 8; ++x) m.push(red_d[x], red_i[x]);
        float lng = 0.f, lat = 0.f;
        int64_t t1 = -1, t2 = -1;
        if (m.i1 != RF_SENT) {
            t1 = bank.member_idx[m.i1];
            t1 = t1 < 0 ? 0 : (t1 >= bank.num_train ? bank.num_train - 1 : t1);
            lng = bank.train_lnglat[2 * t1]; lat = bank.train_lnglat[2 * t1 + 1];
        }
        if (m.i2 != RF_SENT) {
            t2 = bank.member_idx[m.i2];
            t2 = t2 < 0 ? 0 : (t2 >= bank.num_train ? bank.num_train - 1 : t2);
        }
        outp[1] = lng; outp[2] = lat; outp[3] = (float)((e - s) + (me - ms));
        if (EXT) {
            outp[7] = m.i1 == RF_SENT ? -1.f : -m.d1; outp[8] = m.i2 == RF_SENT ? -1.f : -m.d2;
            outp[9] = __int_as_float((int)t1); outp[10] = __int_as_float((int)t2); outp[11] = __int_as_float(cnt);
        }
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

// n_eval records of SC floats per query in `scratch`; only the first topk take part (pg_refine_forward: n_eval = topk, SC = 4)
__global__ __launch_bounds__(64) void refine_select_kernel(const float* __restrict__ scratch, int B, int k, int topk, int n_eval, int SC,
                                                           const int64_t* __restrict__ cand,
                                                           const float* __restrict__ cand_prob,
                                                           const double* __restrict__ init_llh, float temperature,
                                                           double max_km, float* __restrict__ out_llh,
                                                           int64_t* __restrict__ out_cell, int32_t* __restrict__ out_choice,
                                                           int32_t* __restrict__ out_refined) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* sc = scratch + (int64_t)b * n_eval * SC;
    float cp[64], fin[64];
    float sum = 0.f;
    for (int j = 0; j < topk; ++j) {
        const float ex = expf(sc[SC * j] / temperature);    // torch.exp(input / T), fp32
        fin[j] = ex;
        sum += ex;                                          // torch.sum, sequential for <= 64 elements
    }
    for (int j = 0; j < topk; ++j) {
        cp[j] = cand_prob ? cand_prob[(int64_t)b * k + j] : (j == 0 ? 1.0f : 0.0f);
        fin[j] = cp[j] * (fin[j] / sum);
    }
    const int refined = argmax_torch(fin, topk);
    const float rlng = sc[SC * refined + 1], rlat = sc[SC * refined + 2];
    const double dist = haversine_km(init_llh[2 * b], init_llh[2 * b + 1], rlng, rlat);
    int choice = refined;
    if (dist > max_km) choice = argmax_torch(cp, topk);    // veto: fall back to the geocell probabilities
    out_llh[2 * b] = sc[SC * choice + 1];
    out_llh[2 * b + 1] = sc[SC * choice + 2];
    out_cell[b] = cand[(int64_t)b * k + choice];
    out_choice[b] = choice;
    if (out_refined) out_refined[b] = refined;              // the candidate picked BEFORE the veto
}

static int refine_launch(const pg_bank* bank, const float* q, int B, int P, const double* init_llh, const int64_t* cand,
                         const float* cand_prob, int k, int topk, int n_eval, bool ext, float temperature, double max_refine_km,
                         float* scratch, float* out_llh, int64_t* out_cell, int32_t* out_choice, int32_t* out_refined, hipStream_t s,
                         const char* who) {
    if (B < 0) { pg_set_error("%s: B = %d", who, B); return PG_EINVAL; }
    if (B == 0) return PG_OK;                              // an empty batch is a no-op: its (empty) buffers may be NULL
    if (!bank || !q || !init_llh || !cand || !scratch || !out_llh || !out_cell || !out_choice) {
        pg_set_error("%s: null pointer argument", who); return PG_EINVAL;
    }
    if (topk < 1 || topk > k || topk > 64 || P < 1 || n_eval < topk || n_eval > k) {
        pg_set_error("%s: need 1 <= topk <= min(k,64), topk <= n_eval <= k and P >= 1 (topk=%d n_eval=%d k=%d P=%d)", who, topk, n_eval, k, P);
        return PG_EINVAL;
    }
    if (ext && (bank->num_protos >= (1ll << 31) || bank->num_train >= (1ll << 31))) {
        pg_set_error("%s: bank rows do not fit the int32 row ids of the extended scratch record", who); return PG_EINVAL;
    }
    if (ext) hipLaunchKernelGGL(refine_candidates_kernel<true>, dim3((unsigned)B * n_eval), dim3(256), 0, s, *bank, q, P, cand, k, n_eval, scratch);
    else hipLaunchKernelGGL(refine_candidates_kernel<false>, dim3((unsigned)B * n_eval), dim3(256), 0, s, *bank, q, P, cand, k, n_eval, scratch);
    int rc = pg_check_launch("refine_candidates");
    if (rc) return rc;
    hipLaunchKernelGGL(refine_select_kernel, dim3((B + 63) / 64), dim3(64), 0, s, scratch, B, k, topk, n_eval, ext ? 12 : 4, cand, cand_prob,
                       init_llh, temperature, max_refine_km, out_llh, out_cell, out_choice, out_refined);
    return pg_check_launch("refine_select");
}

extern "C" int pg_refine_forward(const pg_bank* bank, const float* q, int B, int P, const double* init_llh,
                                 const int64_t* cand, const float* cand_prob, int k, int topk, float temperature,
                                 double max_refine_km, float* scratch, float* out_llh, int64_t* out_cell,
                                 int32_t* out_choice, void* stream) {
    return refine_launch(bank, q, B, P, init_llh, cand, cand_prob, k, topk, topk, false, temperature, max_refine_km, scratch, out_llh,
                         out_cell, out_choice, nullptr, (hipStream_t)stream, "refine");
}

extern "C" int pg_refine_forward_ex(const pg_bank* bank, const float* q, int B, int P, const double* init_llh,
                                    const int64_t* cand, const float* cand_prob, int k, int topk, int n_eval, float temperature,
                                    double max_refine_km, float* scratch12, float* out_llh, int64_t* out_cell,
                                    int32_t* out_choice, int32_t* out_refined, void* stream) {
    return refine_launch(bank, q, B, P, init_llh, cand, cand_prob, k, topk, n_eval, true, temperature, max_refine_km, scratch12, out_llh,
                         out_cell, out_choice, out_refined, (hipStream_t)stream, "refine_ex");
}
