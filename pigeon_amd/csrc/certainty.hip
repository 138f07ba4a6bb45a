// certainty.hip -- which of this path's DISCRETE outputs are certain to be the reference's (round 5).
//
// The reference computes everything in fp32 (models/super_guessr.py:447-459, models/proto_refiner.py:154-222); this path's embeddings
// carry the rounding of 16-bit MFMA operands (relative error eps ~ 3e-4).  Every discrete output downstream of the embedding is an
// argmax / argmin / top-k membership, i.e. the sign of a MARGIN m(e) >= 0 that is a smooth function of the panel-mean embedding e:
//
//   head     top-1 cell              m = logit(c0) - logit(cj)                       grad = W[c0] - W[cj]
//   refiner  winning candidate r     m = s_r - s_j,  s_j = log p_j - d_j / T         grad = W[cr] - W[cj] - (u_r - u_j) / T
//            (proto_refiner.py:187-192, 219: argmax of c_probs * softmax(-d / T); the softmax denominator is common to all j)
//            candidate-set boundary  m = log p_{topk-1} - log p_j  (j outside the set; it matters only if j could also win)
//            nearest prototype       m = d2 - d1  (:176-182)                          grad = u_2 - u_1
//            farthest member         m = far1 - far2  (:244-255)                      grad = v_1 - v_2
//   with d_j = |e - p_j| the distance to candidate j's nearest prototype and u_j = (e - p_j) / d_j.  The haversine veto (:198-205)
//   compares two points that are themselves picked from discrete sets: it has no margin of its own.
//
// An embedding error  delta = |e| (beta + r)  (beta: the calibrated systematic part, relative to |e|; r: the rest, relative RMS norm
// eps, direction unknown) moves a margin by  |e| grad.beta  +  N(0, (eps |e| |grad| / 32)^2)   (32 = sqrt(1024)).  Per decision
//   t = (m - |e| grad.beta) / (|e| |grad| / 32)
// is the largest eps, in units of one standard deviation, that the decision survives; a row's TOLERANCE is the minimum over its
// decisions, and the host calls the row certain when tolerance > kappa * eps (pigeon_amd/certainty.py: eps and beta are measured per
// set of weights against the exact encoder, kappa is a z-score).  Uncertain rows are re-encoded by pg_vit_forward_precise.
//
// Two kernels, one block (4 waves) per row; a wave evaluates one decision at a time: the gradient is a linear combination of at most
// four 4 KB rows (head weights, prototypes, training rows) and e, 16 columns per lane, reduced with shuffles.  ~100 decisions x
// <= 16 KB per row: a few hundred MB per 128-row batch from L2 / HBM, the cost of one refinement pass (tens of microseconds).
#include "common.h"
#include "pigeon_internal.h"
#include <cfloat>
#include <cmath>

#define CT_DIM 1024
#define CT_MAX_EVAL 96

namespace {

__device__ __forceinline__ void ld16(const float* __restrict__ p, int lane, f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(p + i * 256 + lane * 4);
}
__device__ __forceinline__ void zero16(f32x4 (&g)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// g += coef * row   (row == nullptr or coef == 0: nothing; both are wave-uniform)
__device__ __forceinline__ void axpy16(f32x4 (&g)[4], const float* __restrict__ row, float coef, int lane) {
    if (!row || coef == 0.f) return;
    f32x4 v[4];
    ld16(row, lane, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] += coef * v[i];
}
__device__ __forceinline__ void axpy16r(f32x4 (&g)[4], const f32x4 (&v)[4], float coef) {
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] += coef * v[i];
}
__device__ __forceinline__ float dot16(const f32x4 (&a)[4], const f32x4 (&b)[4]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fmaf(a[i][e], b[i][e], s);
    return wave_sum(s);
}
// mean over the P panels of row b (the head's and the refiner's query: super_guessr.py:437, proto_refiner.py:139-140)
__device__ __forceinline__ void panel_mean16(const float* __restrict__ emb, int P, int lane, f32x4 (&ev)[4]) {
    ld16(emb, lane, ev);
    for (int p = 1; p < P; ++p) {
        f32x4 t[4];
        ld16(emb + (int64_t)p * CT_DIM, lane, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) ev[i] += t[i];
    }
    if (P > 1) {
        const float inv = 1.0f / (float)P;
#pragma unroll
        for (int i = 0; i < 4; ++i) ev[i] *= inv;
    }
}

// t = (m - en * gb) / (en * sqrt(g2) / 32).  m is >= 0 by construction (the margin in favour of the decision taken); a NaN anywhere
// makes the decision uncertain (0); a zero gradient means no embedding error can move the margin (+inf).
__device__ __forceinline__ float tol_of(float m, float g2, float gb, float en) {
    if (!(m == m)) return 0.f;
    if (!(g2 > 0.f)) return (g2 == g2) ? INFINITY : 0.f;
    const float num = m - en * gb;
    const float t = num / (en * sqrtf(g2) * (1.0f / 32.0f));
    return (t == t) ? t : 0.f;
}

struct MinTol {
    float t; int code;
    __device__ __forceinline__ void take(float x, int c) { if (x < t) { t = x; code = c; } }
};
__device__ __forceinline__ MinTol block_min(MinTol v, float* red_t, int* red_c) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red_t[wave] = v.t; red_c[wave] = v.code; }
    __syncthreads();
    MinTol r; r.t = red_t[0]; r.code = red_c[0];
    for (int w = 1; w < 4; ++w) r.take(red_t[w], red_c[w]);
    return r;
}

// ---- head: is the top-1 cell certain?  decisions: c0 against every listed candidate, and against everything not listed ----------
// code: j >= 1 = the candidate that sets the tolerance, -1 = the cells beyond the list
__global__ __launch_bounds__(256) void head_certainty_kernel(const float* __restrict__ logits, int C, const float* __restrict__ emb, int P,
                                                             const float* __restrict__ W, const int64_t* __restrict__ topk_idx, int kx,
                                                             const float* __restrict__ beta, const float* __restrict__ wstats,
                                                             float* __restrict__ tol, int32_t* __restrict__ code,
                                                             float* __restrict__ margin, float* __restrict__ sens) {
    __shared__ float red_t[4];
    __shared__ int red_c[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 ev[4], bv[4];
    panel_mean16(emb + (int64_t)b * P * CT_DIM, P, lane, ev);
    const float en = sqrtf(dot16(ev, ev));
    if (beta) ld16(beta, lane, bv); else zero16(bv);
    const int64_t* idx = topk_idx + (int64_t)b * kx;
    const float* row = logits + (int64_t)b * C;
    const int64_t c0 = idx[0];
    MinTol best; best.t = INFINITY; best.code = 0;
    if (c0 < 0 || c0 >= C) {
        best.t = 0.f; best.code = -2;
        if (tid == 0) { if (margin) margin[b] = NAN; if (sens) sens[b] = 0.f; }     // never leave the report fields unwritten
    } else {
        const float l0 = row[c0];
        for (int j = 1 + wave; j < kx; j += 4) {
            const int64_t cj = idx[j];
            float t = 0.f, m = NAN, g2 = 0.f;
            if (cj >= 0 && cj < C) {
                m = l0 - row[cj];
                f32x4 g[4];
                zero16(g);
                axpy16(g, W + c0 * CT_DIM, 1.f, lane);
                axpy16(g, W + cj * CT_DIM, -1.f, lane);
                g2 = dot16(g, g);
                t = tol_of(m, g2, dot16(g, bv), en);
            }
            best.take(t, j);
            if (j == 1 && lane == 0) { if (margin) margin[b] = m; if (sens) sens[b] = en * sqrtf(g2) * (1.0f / 32.0f); }
        }
        if (wave == 0 && kx < C) {
            // cells beyond the list: logit <= the list's last, |W[c0] - W[c]| <= |W[c0]| + the largest row norm (wstats[0]), and the
            // systematic part (W[c0] - W[c]).beta <= W[c0].beta + max_c |W[c].beta| (wstats[1], computed once per calibration)
            const int64_t ck = idx[kx - 1];
            float t = 0.f;
            if (ck >= 0 && ck < C) {
                f32x4 g[4];
                zero16(g);
                axpy16(g, W + c0 * CT_DIM, 1.f, lane);
                const float gmax = sqrtf(dot16(g, g)) + wstats[0];
                t = tol_of(l0 - row[ck], gmax * gmax, dot16(g, bv) + wstats[1], en);
            }
            best.take(t, -1);
        }
        if (kx < 2 && lane == 0 && wave == 0) { if (margin) margin[b] = INFINITY; if (sens) sens[b] = 0.f; }
    }
    const MinTol r = block_min(best, red_t, red_c);
    if (tid == 0) { tol[b] = r.t; code[b] = r.code; }
}

// ---- refiner: are the refined cell and point certain?  (scratch12: the records pg_refine_forward_ex left) ------------------------
// code: 1000 + j  winner against candidate j of the set;  2000 + j  candidate j outside the set could enter and win;  2999 the cells
// beyond the evaluated ones could enter;  3000 / 3001 nearest prototype of the refined / the chosen candidate against every other
// prototype of the cell;  4000 / 4001 farthest member against every other member of the cluster;
// -9 the winning product underflows in fp32 (or an empty cell wins a set that is not all empty);  -8 refined / choice out of range;
// 0 nothing can change it
__global__ __launch_bounds__(256) void refine_certainty_kernel(pg_bank bank, const float* __restrict__ q, int P,
                                                               const int64_t* __restrict__ cand, const float* __restrict__ cand_prob,
                                                               int k, int topk, int n_eval, const float* __restrict__ scratch12,
                                                               const float* __restrict__ W, int C, const float* __restrict__ beta,
                                                               const float* __restrict__ wstats, float temperature,
                                                               const int32_t* __restrict__ refined, const int32_t* __restrict__ choice,
                                                               float* __restrict__ tol, int32_t* __restrict__ code) {
    __shared__ float L[CT_MAX_EVAL], S[CT_MAX_EVAL];
    __shared__ float red_t[4];
    __shared__ int red_c[4];
    __shared__ int flag_under;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* rec0 = scratch12 + (int64_t)b * n_eval * 12;
    const int64_t* cd = cand + (int64_t)b * k;
    const float invT = 1.0f / temperature;
    if (tid < n_eval) {
        const float p = cand_prob ? cand_prob[(int64_t)b * k + tid] : (tid == 0 ? 1.0f : 0.0f);
        const float l = logf(p);                             // differences of log-probabilities = differences of logits
        L[tid] = l;
        S[tid] = l + rec0[12 * tid] * invT;
    }
    const int r = refined[b], ch = choice[b];
    if (r < 0 || r >= topk || ch < 0 || ch >= topk) {        // not a record pg_refine_forward_ex wrote (block-uniform exit)
        if (tid == 0) { tol[b] = 0.f; code[b] = -8; }
        return;
    }
    if (tid == 0) {
        // the product the reference's argmax looks at (proto_refiner.py:187-192), in its own fp32 arithmetic
        float sum = 0.f;
        for (int j = 0; j < topk; ++j) sum += expf(rec0[12 * j] * invT);
        const float pr = cand_prob ? cand_prob[(int64_t)b * k + r] : (r == 0 ? 1.0f : 0.0f);
        const float fin = pr * (expf(rec0[12 * r] * invT) / sum);
        const bool empty_winner = __float_as_int(rec0[12 * r + 5]) < 0;
        // An empty cell (score -100000) wins only when every product of the set is 0 or NaN -- the underflow situation -- unless the
        // whole set is empty cells, in which case no embedding error can change anything.
        bool all_empty = true;
        for (int j = 0; j < topk; ++j) all_empty = all_empty && __float_as_int(rec0[12 * j + 5]) < 0;
        flag_under = (empty_winner && all_empty) ? 2 : ((!(fin >= 1e-30f) || empty_winner) ? 1 : 0);
    }
    __syncthreads();
    f32x4 ev[4], bv[4];
    panel_mean16(q + (int64_t)b * P * CT_DIM, P, lane, ev);
    const float en = sqrtf(dot16(ev, ev));
    if (beta) ld16(beta, lane, bv); else zero16(bv);

    auto wrow = [&](int64_t c) -> const float* { return (c >= 0 && c < C) ? W + c * CT_DIM : nullptr; };
    // tolerance of "candidate a stays ahead of candidate j in s = log p - d / T"
    auto pair_s = [&](int a, int j) -> float {
        const float* ra = rec0 + 12 * a;
        const float* rj = rec0 + 12 * j;
        const int pa = __float_as_int(ra[5]), pj = __float_as_int(rj[5]);
        const float da = -ra[0], dj = -rj[0];
        const float ia = (pa >= 0 && da > 0.f) ? invT / da : 0.f, ij = (pj >= 0 && dj > 0.f) ? invT / dj : 0.f;
        f32x4 g[4];
        zero16(g);
        axpy16(g, wrow(cd[a]), 1.f, lane);
        axpy16(g, wrow(cd[j]), -1.f, lane);
        axpy16(g, pa >= 0 ? bank.proto_emb + (int64_t)pa * CT_DIM : nullptr, ia, lane);
        axpy16(g, pj >= 0 ? bank.proto_emb + (int64_t)pj * CT_DIM : nullptr, -ij, lane);
        axpy16r(g, ev, ij - ia);
        return tol_of(S[a] - S[j], dot16(g, g), dot16(g, bv), en);
    };
    // every alternative to an argmin (sign +1: nearest prototype) / argmax (sign -1: farthest member) over rows [lo, hi) of `base`
    // (through `index` if given): the pick `win` against row j has margin m = sign (d_j - d_win) and gradient sign (u_j - u_win), u =
    // unit vector from the row to the query; |grad|^2 = 2 - 2 u_j.u_win, grad.beta = sign (u_j.beta - u_win.beta).  Distances are
    // recomputed here (the record holds the winner's and the runner-up's only); a NaN anywhere makes the pick uncertain.
    auto scan_alternatives = [&](const float* base, const int64_t* index, int64_t lo, int64_t hi, int64_t win, float sign, int cd_,
                                 MinTol& acc) {
        f32x4 w[4];
        ld16(base + win * CT_DIM, lane, w);
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = ev[i] - w[i];
        const float dw = sqrtf(dot16(w, w));
        const float iw = dw > 0.f ? 1.0f / dw : 0.f;
        const float wb = dot16(w, bv) * iw;
        for (int64_t j = lo + wave; j < hi; j += 4) {
            const int64_t row = index ? index[j] : j;
            if (row == win) continue;
            f32x4 l[4];
            ld16(base + row * CT_DIM, lane, l);
#pragma unroll
            for (int i = 0; i < 4; ++i) l[i] = ev[i] - l[i];
            const float dl = sqrtf(dot16(l, l));
            const float il = dl > 0.f ? 1.0f / dl : 0.f;
            const float g2 = (dl > 0.f ? 1.f : 0.f) + (dw > 0.f ? 1.f : 0.f) - 2.f * dot16(l, w) * il * iw;
            const float gb = sign * (dot16(l, bv) * il - wb);
            acc.take(tol_of(sign * (dl - dw), fmaxf(g2, 0.f), gb, en), cd_);
        }
    };

    MinTol best; best.t = INFINITY; best.code = 0;
    int task = 0;
    if (flag_under == 1) { best.t = 0.f; best.code = -9; }
    else if (flag_under == 0) {
        for (int j = 0; j < topk; ++j) {                     // the winner against the rest of the set
            if (j == r) continue;
            if ((task++ & 3) == wave) best.take(pair_s(r, j), 1000 + j);
        }
        for (int j = topk; j < n_eval; ++j) {                // a candidate outside the set: it must get in AND (unless it pushes r out) win
            if ((task++ & 3) != wave) continue;
            f32x4 g[4];
            zero16(g);
            axpy16(g, wrow(cd[topk - 1]), 1.f, lane);
            axpy16(g, wrow(cd[j]), -1.f, lane);
            const float t_in = tol_of(L[topk - 1] - L[j], dot16(g, g), dot16(g, bv), en);
            const float t = (r == topk - 1) ? t_in : fmaxf(t_in, pair_s(r, j));
            best.take(t, 2000 + j);
        }
        if (n_eval > topk && n_eval < C && (task++ & 3) == wave) {
            // cells beyond the evaluated ones: their log-probability is <= the last evaluated one's; nothing is known about their
            // prototypes, so getting into the set already counts (bounds as in head_certainty_kernel: wstats)
            f32x4 g[4];
            zero16(g);
            axpy16(g, wrow(cd[topk - 1]), 1.f, lane);
            const float gmax = sqrtf(dot16(g, g)) + wstats[0];
            best.take(tol_of(L[topk - 1] - L[n_eval - 1], gmax * gmax, dot16(g, bv) + wstats[1], en), 2999);
        }
        for (int which = 0; which < 2; ++which) {            // the discrete picks inside the refined / the finally chosen candidate
            const int x = which == 0 ? r : ch;
            if (which == 1 && ch == r) break;
            const float* rx = rec0 + 12 * x;
            // Round 6: the pick is checked against EVERY alternative -- all prototypes of the cell, all members of the
            // cluster -- not only against the runner-up of the record: a third prototype with a slightly larger margin whose
            // direction differs more from the winner's can have the smaller tolerance.  The rows are streamed again (<= 2 cells and
            // 2 clusters per query: a fraction of the refinement pass that just read them), one alternative per wave at a time.
            const int p1 = __float_as_int(rx[5]);
            const int64_t cx = cd[x];
            if (p1 >= 0 && cx >= 0 && cx < bank.num_cells)
                scan_alternatives(bank.proto_emb, nullptr, bank.cell_off[cx], bank.cell_off[cx + 1], p1, 1.f, 3000 + which, best);
            const int t1 = __float_as_int(rx[9]);
            if (t1 >= 0 && p1 >= 0)
                scan_alternatives(bank.train_emb, bank.member_idx, bank.member_off[p1], bank.member_off[p1 + 1], t1, -1.f, 4000 + which, best);
        }
    }
    const MinTol res = block_min(best, red_t, red_c);
    if (tid == 0) { tol[b] = res.t; code[b] = res.code; }
}

// emb[r] -= |emb[r]| * bias  (one wave per row of 1024; the norm's summation order is fixed: 16 columns per lane in ld16's order, then
// wave_sum's butterfly).  The systematic part of the 16-bit encoder's error, measured once per set of weights against the exact
// encoder (pigeon_amd/certainty.py, `debias`), taken out of every fast embedding before anything downstream reads it.
__global__ __launch_bounds__(256) void embedding_debias_kernel(float* __restrict__ emb, int64_t n, const float* __restrict__ bias) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float* row = emb + r * CT_DIM;
    f32x4 e[4], b[4];
    ld16(row, lane, e);
    ld16(bias, lane, b);
    const float norm = sqrtf(dot16(e, e));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = __fmaf_rn(-norm, b[i][c], e[i][c]);
        *(f32x4*)(row + i * 256 + lane * 4) = o;
    }
}

}  // namespace

extern "C" int pg_embedding_debias(float* emb, int64_t n, int dim, const float* bias, void* stream) {
    if (n < 0 || dim != CT_DIM) { pg_set_error("embedding_debias: n = %lld, dim = %d (must be %d)", (long long)n, dim, CT_DIM); return PG_EINVAL; }
    if (n == 0) return PG_OK;
    if (!emb || !bias) { pg_set_error("embedding_debias: null pointer argument"); return PG_EINVAL; }
    if ((n + 3) / 4 > 2147483647LL) { pg_set_error("embedding_debias: n = %lld rows exceed the launch grid", (long long)n); return PG_EINVAL; }
    hipLaunchKernelGGL(embedding_debias_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, emb, n, bias);
    return pg_check_launch("embedding_debias");
}

extern "C" int pg_head_certainty(const float* logits, int B, int C, const float* emb, int P, const float* W, const int64_t* topk_idx,
                                 int kx, const float* beta, const float* wstats, float* tol, int32_t* code, float* margin,
                                 float* sens, void* stream) {
    if (B < 0) { pg_set_error("head_certainty: B = %d", B); return PG_EINVAL; }
    if (B == 0) return PG_OK;
    if (!logits || !emb || !W || !topk_idx || !wstats || !tol || !code) { pg_set_error("head_certainty: null pointer argument"); return PG_EINVAL; }
    if (P < 1 || C < 1 || kx < 1 || kx > C) { pg_set_error("head_certainty: bad P=%d C=%d kx=%d", P, C, kx); return PG_EINVAL; }
    hipLaunchKernelGGL(head_certainty_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, C, emb, P, W, topk_idx, kx, beta,
                       wstats, tol, code, margin, sens);
    return pg_check_launch("head_certainty");
}

extern "C" int pg_refine_certainty(const pg_bank* bank, const float* q, int B, int P, const int64_t* cand, const float* cand_prob, int k,
                                   int topk, int n_eval, const float* scratch12, const float* W, int C, const float* beta,
                                   const float* wstats, float temperature, const int32_t* refined, const int32_t* choice, float* tol,
                                   int32_t* code, void* stream) {
    if (B < 0) { pg_set_error("refine_certainty: B = %d", B); return PG_EINVAL; }
    if (B == 0) return PG_OK;
    if (!bank || !q || !cand || !scratch12 || !W || !wstats || !refined || !choice || !tol || !code) {
        pg_set_error("refine_certainty: null pointer argument"); return PG_EINVAL;
    }
    if (topk < 1 || topk > 64 || n_eval < topk || n_eval > k || n_eval > CT_MAX_EVAL || P < 1 || C < 1 || !(temperature > 0.f)) {
        pg_set_error("refine_certainty: need 1 <= topk <= 64, topk <= n_eval <= min(k, %d), P >= 1, C >= 1, T > 0 (topk=%d n_eval=%d k=%d)",
                     CT_MAX_EVAL, topk, n_eval, k);
        return PG_EINVAL;
    }
    hipLaunchKernelGGL(refine_certainty_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, *bank, q, P, cand, cand_prob, k, topk, n_eval,
                       scratch12, W, C, beta, wstats, temperature, refined, choice, tol, code);
    return pg_check_launch("refine_certainty");
}
