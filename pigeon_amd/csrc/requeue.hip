// requeue.hip -- the device side of the DEFERRED exact tier (round 6).
//
// The reference computes the whole path in fp32 (models/super_guessr.py:447-459, models/proto_refiner.py:176-222); here a sample whose
// discrete outputs the 16-bit encoder cannot settle (certainty.hip) is re-encoded by pg_vit_forward_precise.  Until round 5 the host
// found those samples with `torch.nonzero` -- one host synchronisation per step -- and re-encoded them right away, 4 to 16 images at a
// time, a batch size at which every kernel of a 24-layer pass is latency-bound.  Now the uncertain rows are COMPACTED ON THE DEVICE
// into a circular queue (their pixels and the row of the result ring they belong to), the host learns the queue length one step late
// through pinned memory, and the exact tier runs on 40+ queued images at once; its results are scattered back into the result ring
// before anything is collected (the reference's evaluate loop concatenates at the end: training/train_eval_loop.py:98-112).
//
//   pg_requeue_append    per row: certain = head_tol > thr && refine_tol > thr (a NaN is not certain), cause code, and -- for the rows
//                        that are not certain -- a queue slot, in row order (one block, ballot prefix sums: deterministic)
//   pg_rows_to_slots     queue_rows[row_slot[r]] = src_rows[r]            (the pixels of the queued rows; 5.4 MB per fp32 panorama)
//   pg_requeue_take      the ring rows of `n_pad` queue slots from `head` on (-1 beyond the first n_valid: padding rows)
//   pg_scatter_rows      dst[dst_row[i]] = src[i]                          (patching the ring with the exact tier's results)
//   pg_head_wstats       [max_c |W[c]|, max_c |W[c].beta|]: the bounds pg_head_certainty / pg_refine_certainty use for the cells they
//                        do not visit one by one (was a torch matmul + norm on the host side of the ABI)
#include "common.h"
#include "pigeon_internal.h"

namespace {

// One block of 256 threads walks the rows in chunks of 256, in order.
__global__ void __launch_bounds__(256)
requeue_append_kernel(const float* __restrict__ head_tol, const float* __restrict__ refine_tol, const int32_t* __restrict__ refine_code,
                      int B, float thr, int force_all, int64_t dst_base, int64_t flushed, int64_t cap,
                      int64_t* __restrict__ counters, int64_t* __restrict__ slot_dst, int32_t* __restrict__ row_slot,
                      uint8_t* __restrict__ certain, int32_t* __restrict__ cause) {
    __shared__ int wave_cnt[4];
    __shared__ int64_t s_app, s_drop;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_app = counters ? counters[0] : 0; s_drop = counters ? counters[1] : 0; }
    __syncthreads();
    for (int base = 0; base < B; base += 256) {
        const int r = base + tid;
        bool flag = false;
        int why = 0;
        if (r < B) {
            const bool h_unc = force_all || !(head_tol[r] > thr);
            const bool r_unc = refine_tol ? !(refine_tol[r] > thr) : false;
            flag = h_unc || r_unc;
            why = h_unc ? 1 : (r_unc ? (refine_code ? refine_code[r] : 2) : 0);
            if (r_unc && !h_unc && why == 0) why = 2;                  // a flagged row always carries a non-zero cause
            if (certain) certain[r] = flag ? 0 : 1;
            if (cause) cause[r] = why;
        }
        const unsigned long long m = __ballot(flag);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wv] = __popcll(m);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wv; ++w) off += wave_cnt[w];
        const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (r < B && row_slot) {
            int slot = -1;
            if (flag && cap > 0) {
                const int64_t pos = s_app + off + before;              // position in the stream of everything ever queued
                if (pos - flushed < cap) {
                    slot = (int)(pos % cap);
                    slot_dst[slot] = dst_base + r;
                } else
                    slot = -2;                                         // does not fit: counted, never silently lost
            }
            row_slot[r] = slot;
        }
        __syncthreads();
        if (tid == 0 && cap > 0) {
            const int64_t room = cap - (s_app - flushed);
            const int64_t take = total < room ? total : (room > 0 ? room : 0);
            s_app += take;
            s_drop += total - take;
        }
        __syncthreads();
    }
    if (tid == 0 && counters) { counters[0] = s_app; counters[1] = s_drop; }
}

template <typename V>
__device__ __forceinline__ void copy_row(const char* __restrict__ src, char* __restrict__ dst, int64_t row_bytes) {
    const int64_t n = row_bytes / (int64_t)sizeof(V);
    const V* s = (const V*)src;
    V* d = (V*)dst;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

// grid (chunks, rows): block (x, r) copies its share of row r -- if the row has a destination
template <typename V>
__global__ void __launch_bounds__(256)
rows_to_slots_kernel(const char* __restrict__ src, int64_t row_bytes, const int32_t* __restrict__ row_slot, char* __restrict__ dst) {
    const int r = blockIdx.y;
    const int slot = row_slot[r];
    if (slot < 0) return;
    copy_row<V>(src + (int64_t)r * row_bytes, dst + (int64_t)slot * row_bytes, row_bytes);
}

template <typename V>
__global__ void __launch_bounds__(256)
scatter_rows_kernel(const char* __restrict__ src, int64_t row_bytes, const int64_t* __restrict__ dst_row, char* __restrict__ dst,
                    int64_t dst_rows, int64_t remap_wb, int64_t remap_b, int64_t remap_off) {
    const int i = blockIdx.y;
    int64_t d = dst_row[i];
    if (d < 0) return;
    if (remap_wb > 0) {                                                // ring row of the gathered layout -> row of a local-only array
        const int64_t slab = d / remap_wb, r = d % remap_wb - remap_off;
        if (r < 0 || r >= remap_b) return;
        d = slab * remap_b + r;
    }
    if (d >= dst_rows) return;
    copy_row<V>(src + (int64_t)i * row_bytes, dst + d * row_bytes, row_bytes);
}

__global__ void requeue_take_kernel(const int64_t* __restrict__ slot_dst, int64_t cap, int64_t head, int n_valid, int n_pad,
                                    int64_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) out[i] = i < n_valid ? slot_dst[(head + i) % cap] : -1;
}

// one wave per row of W: |W[c]|^2 and W[c].beta, atomic max of the (non-negative) results through their bit patterns
__global__ void __launch_bounds__(256)
head_wstats_kernel(const float* __restrict__ W, int C, const float* __restrict__ beta, unsigned int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float* row = W + (int64_t)c * 1024;
    float n2 = 0.f, wb = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 v = *(const f32x4*)(row + i * 256 + lane * 4);
        n2 = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], n2))));
        if (beta) {
            const f32x4 b = *(const f32x4*)(beta + i * 256 + lane * 4);
            wb = fmaf(v[0], b[0], fmaf(v[1], b[1], fmaf(v[2], b[2], fmaf(v[3], b[3], wb))));
        }
    }
    n2 = wave_sum(n2);
    wb = wave_sum(wb);
    if (lane == 0) {
        // NaN bit patterns compare above every finite one: a NaN weight poisons the bound, as it should
        atomicMax(out + 0, __float_as_uint(sqrtf(n2)));
        atomicMax(out + 1, __float_as_uint(fabsf(wb)));
    }
}

int vec_bytes(const void* a, const void* b, int64_t row_bytes) {
    const uintptr_t al = (uintptr_t)a | (uintptr_t)b | (uintptr_t)row_bytes;
    return (al & 15) == 0 ? 16 : (al & 7) == 0 ? 8 : (al & 3) == 0 ? 4 : 1;
}
dim3 row_grid(int64_t row_bytes, int vb, int rows) {
    int64_t chunks = (row_bytes / vb + 256 * 8 - 1) / (256 * 8);
    if (chunks < 1) chunks = 1;
    if (chunks > 256) chunks = 256;
    return dim3((unsigned)chunks, (unsigned)rows, 1);
}

}  // namespace

extern "C" int pg_requeue_append(const float* head_tol, const float* refine_tol, const int32_t* refine_code, int B, float thr,
                                 int force_all, int64_t dst_base, int64_t flushed, int64_t cap, int64_t* counters, int64_t* slot_dst,
                                 int32_t* row_slot, uint8_t* certain, int32_t* cause, void* stream) {
    if (B < 0 || cap < 0) { pg_set_error("requeue_append: B = %d, cap = %lld", B, (long long)cap); return PG_EINVAL; }
    if (B == 0) return PG_OK;
    if (!head_tol) { pg_set_error("requeue_append: null head_tol"); return PG_EINVAL; }
    if (cap > 0 && (!counters || !slot_dst || !row_slot)) { pg_set_error("requeue_append: a queue needs counters, slot_dst and row_slot"); return PG_EINVAL; }
    hipLaunchKernelGGL(requeue_append_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, head_tol, refine_tol, refine_code, B, thr,
                       force_all, dst_base, flushed, cap, counters, slot_dst, row_slot, certain, cause);
    return pg_check_launch("requeue_append_kernel");
}

extern "C" int pg_rows_to_slots(const void* src, int64_t row_bytes, const int32_t* row_slot, int B, void* dst, void* stream) {
    if (B < 0 || row_bytes < 0) { pg_set_error("rows_to_slots: bad size"); return PG_EINVAL; }
    if (B == 0 || row_bytes == 0) return PG_OK;
    if (!src || !row_slot || !dst) { pg_set_error("rows_to_slots: null argument"); return PG_EINVAL; }
    if (B > 65535) { pg_set_error("rows_to_slots: B = %d rows exceed the launch grid (65535)", B); return PG_EINVAL; }
    const int vb = vec_bytes(src, dst, row_bytes);
    const dim3 g = row_grid(row_bytes, vb, B);
    hipStream_t s = (hipStream_t)stream;
    if (vb == 16) hipLaunchKernelGGL(rows_to_slots_kernel<uint4>, g, dim3(256), 0, s, (const char*)src, row_bytes, row_slot, (char*)dst);
    else if (vb == 8) hipLaunchKernelGGL(rows_to_slots_kernel<uint2>, g, dim3(256), 0, s, (const char*)src, row_bytes, row_slot, (char*)dst);
    else if (vb == 4) hipLaunchKernelGGL(rows_to_slots_kernel<uint32_t>, g, dim3(256), 0, s, (const char*)src, row_bytes, row_slot, (char*)dst);
    else hipLaunchKernelGGL(rows_to_slots_kernel<uint8_t>, g, dim3(256), 0, s, (const char*)src, row_bytes, row_slot, (char*)dst);
    return pg_check_launch("rows_to_slots_kernel");
}

extern "C" int pg_scatter_rows(const void* src, int64_t row_bytes, const int64_t* dst_row, int n, void* dst, int64_t dst_rows,
                               int64_t remap_wb, int64_t remap_b, int64_t remap_off, void* stream) {
    if (n < 0 || row_bytes < 0 || dst_rows < 0) { pg_set_error("scatter_rows: bad size"); return PG_EINVAL; }
    if (n == 0 || row_bytes == 0) return PG_OK;
    if (!src || !dst_row || !dst) { pg_set_error("scatter_rows: null argument"); return PG_EINVAL; }
    if (n > 65535) { pg_set_error("scatter_rows: n = %d rows exceed the launch grid (65535)", n); return PG_EINVAL; }
    const int vb = vec_bytes(src, dst, row_bytes);
    const dim3 g = row_grid(row_bytes, vb, n);
    hipStream_t s = (hipStream_t)stream;
#define PG_SCATTER(V) hipLaunchKernelGGL(scatter_rows_kernel<V>, g, dim3(256), 0, s, (const char*)src, row_bytes, dst_row, (char*)dst, \
                                         dst_rows, remap_wb, remap_b, remap_off)
    if (vb == 16) PG_SCATTER(uint4); else if (vb == 8) PG_SCATTER(uint2); else if (vb == 4) PG_SCATTER(uint32_t); else PG_SCATTER(uint8_t);
#undef PG_SCATTER
    return pg_check_launch("scatter_rows_kernel");
}

extern "C" int pg_requeue_take(const int64_t* slot_dst, int64_t cap, int64_t head, int n_valid, int n_pad, int64_t* dst_out, void* stream) {
    if (n_pad < 0 || n_valid < 0 || n_valid > n_pad || cap <= 0 || head < 0) { pg_set_error("requeue_take: bad argument"); return PG_EINVAL; }
    if (n_pad == 0) return PG_OK;
    if (!slot_dst || !dst_out) { pg_set_error("requeue_take: null argument"); return PG_EINVAL; }
    hipLaunchKernelGGL(requeue_take_kernel, dim3((n_pad + 255) / 256), dim3(256), 0, (hipStream_t)stream, slot_dst, cap, head, n_valid, n_pad, dst_out);
    return pg_check_launch("requeue_take_kernel");
}

extern "C" int pg_head_wstats(const float* W, int C, const float* beta, float* out2, void* stream) {
    if (C < 0) { pg_set_error("head_wstats: C = %d", C); return PG_EINVAL; }
    if (!out2 || (C > 0 && !W)) { pg_set_error("head_wstats: null argument"); return PG_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    PG_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(float), s));
    if (C == 0) return PG_OK;
    hipLaunchKernelGGL(head_wstats_kernel, dim3((C + 3) / 4), dim3(256), 0, s, W, C, beta, (unsigned int*)out2);
    return pg_check_launch("head_wstats_kernel");
}
