// geo_proto.hip -- the steps AFTER / AROUND the hot path:
//
//   pg_proto_build        prototype construction: proto[p] = mean over its member training embeddings, the members being
//                         panel-averaged first when the training bank is (Ntr,4,1024).  Replaces
//                         ProtoRefiner._compute_protos_for_cell (reference models/proto_refiner.py:359-384:
//                         `embeddings.mean(dim=1)` then `.mean(dim=0)`), which the reference runs in 64 CPU processes over
//                         Arrow tables (:257-313).  fp32, rows added in torch's CPU reduction order (cascade of 16-row
//                         chunks, see proto_build_kernel) then divided by the count: bit-identical to the reference's.
//                         HBM-bound: 4 KB (16 KB with panels) read per member row, 4 KB written per prototype.
//   pg_haversine_matrix   (N,2) x (M,2) [lng,lat] degrees -> (N,M) km, reference preprocessing/geo_utils.py:58-74, with
//                         torch's dtype rules: x may be fp32 (deg2rad and cos(lat) evaluated in fp32, then promoted) or
//                         fp64, y is fp64 (SuperGuessr.lla_geocells), the result fp64.
//   pg_smooth_labels      exp(-(d - rowmin(d)) / constant), NaN/inf -> 0, reference preprocessing/utils.py:7-19.
//   Both feed the soft-label cross entropy at reference models/super_guessr.py:469-474.
#include "common.h"
#include "pigeon_internal.h"

#include <cmath>

// --------------------------------------------------------------------------------------------- prototype build
// torch's CPU reduction of a (cnt,1024) fp32 block over dim 0 (ATen SumKernel.cpp cascade_sum -> multi_row_sum) is NOT a
// plain left-to-right sum once cnt >= 16: rows are added in chunks of 2^level_power (16 for cnt <= 65536) into a level-0
// accumulator that starts from zero for every chunk; a finished chunk is added to level 1 (16 chunks of those to level 2,
// ...), the tail rows go to level 0, and the result is ((l0 + l1) + l2) + l3.  Reproduced exactly.
template <int PANELS>
__device__ __forceinline__ f32x4 proto_row(const float* __restrict__ train, const int64_t* __restrict__ member_idx, int64_t j,
                                           int64_t num_train, int c) {
    int64_t m = member_idx[j];
    m = m < 0 ? 0 : (m >= num_train ? num_train - 1 : m);         // indices are validated on the host; never fault
    const float* row = train + m * (int64_t)(PANELS * VIT_HIDDEN) + c;
    f32x4 r = *(const f32x4*)row;
    if (PANELS == 4) {                                            // embeddings.mean(dim=1): 4 < 16 rows -> sequential
        r = r + *(const f32x4*)(row + VIT_HIDDEN);
        r = r + *(const f32x4*)(row + 2 * VIT_HIDDEN);
        r = r + *(const f32x4*)(row + 3 * VIT_HIDDEN);
        r = r * 0.25f;                                             // / 4, exact
    }
    return r;
}

template <int PANELS>
__global__ __launch_bounds__(256) void proto_build_kernel(const float* __restrict__ train, const int64_t* __restrict__ member_off,
                                                          const int64_t* __restrict__ member_idx, float* __restrict__ proto,
                                                          int64_t num_train) {
    const int64_t p = blockIdx.x;
    const int c = threadIdx.x * 4;
    const int64_t j0 = member_off[p];
    const int64_t cnt = member_off[p + 1] - j0;
    int ceil_log2 = 1;
    while (ceil_log2 < 62 && (1ll << ceil_log2) < cnt) ++ceil_log2;             // utils::CeilLog2 (1 for cnt <= 2)
    const int level_power = ceil_log2 / 4 > 4 ? ceil_log2 / 4 : 4;
    const int64_t level_step = 1ll << level_power, level_mask = level_step - 1;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 a0 = zero, a1 = zero, a2 = zero, a3 = zero;
    int64_t i = 0;
    while (i + level_step <= cnt) {
        for (int64_t j = 0; j < level_step; ++j, ++i) a0 = a0 + proto_row<PANELS>(train, member_idx, j0 + i, num_train, c);
        a1 = a1 + a0; a0 = zero;
        if ((i & (level_mask << level_power)) != 0) continue;
        a2 = a2 + a1; a1 = zero;
        if ((i & (level_mask << (2 * level_power))) != 0) continue;
        a3 = a3 + a2; a2 = zero;
    }
    for (; i < cnt; ++i) a0 = a0 + proto_row<PANELS>(train, member_idx, j0 + i, num_train, c);
    f32x4 acc = ((a0 + a1) + a2) + a3;
    if (cnt > 0) {
        const float n = (float)cnt;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __fdiv_rn(acc[e], n);
    }
    *(f32x4*)(proto + p * VIT_HIDDEN + c) = acc;
}

extern "C" int pg_proto_build(const float* train_emb, int panels, int64_t num_train, const int64_t* member_off,
                              const int64_t* member_idx, int64_t num_protos, float* proto_emb, void* stream) {
    if (!train_emb || !member_off || !member_idx || !proto_emb) { pg_set_error("proto_build: null argument"); return PG_EINVAL; }
    if (panels != 1 && panels != 4) { pg_set_error("proto_build: panels must be 1 or 4 (got %d)", panels); return PG_EINVAL; }
    if (num_protos <= 0) return PG_OK;
    if (num_train <= 0) { pg_set_error("proto_build: empty training bank"); return PG_EINVAL; }
    if (num_protos > 0x7fffffffLL) { pg_set_error("proto_build: too many prototypes"); return PG_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    if (panels == 1)
        hipLaunchKernelGGL(proto_build_kernel<1>, dim3((unsigned)num_protos), dim3(256), 0, s, train_emb, member_off, member_idx, proto_emb, num_train);
    else
        hipLaunchKernelGGL(proto_build_kernel<4>, dim3((unsigned)num_protos), dim3(256), 0, s, train_emb, member_off, member_idx, proto_emb, num_train);
    return pg_check_launch("proto_build");
}

// --------------------------------------------------------------------------------------------- haversine matrix
// torch.deg2rad multiplies by the double constant pi/180 rounded to the tensor's dtype
#define DEG2RAD_D 0.017453292519943295769236907684886127134428718885417
template <typename X> struct Deg;
template <> struct Deg<double> {
    static __device__ __forceinline__ double rad(double v) { return v * DEG2RAD_D; }
    static __device__ __forceinline__ double cosv(double r) { return cos(r); }
};
template <> struct Deg<float> {
    static __device__ __forceinline__ float rad(float v) { return v * (float)DEG2RAD_D; }
    static __device__ __forceinline__ float cosv(float r) { return (float)cos((double)r); }   // correctly rounded fp32 cos
};

template <typename X>
__global__ __launch_bounds__(256) void haversine_matrix_kernel(const X* __restrict__ x, const double* __restrict__ y,
                                                               double* __restrict__ out, int N, int M) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const X xlng = Deg<X>::rad(x[2 * i]), xlat = Deg<X>::rad(x[2 * i + 1]);
    const double ylng = y[2 * j] * DEG2RAD_D, ylat = y[2 * j + 1] * DEG2RAD_D;
    const double dlng = (double)xlng - ylng, dlat = (double)xlat - ylat;
    const double p = (double)Deg<X>::cosv(xlat) * cos(ylat);
    const double s1 = sin(dlat / 2), s0 = sin(dlng / 2);
    const double a = s1 * s1 + p * (s0 * s0);
    const double c = 2 * asin(sqrt(a));
    out[(int64_t)i * M + j] = (6378137.0 * c) / 1000;
}

extern "C" int pg_haversine_matrix(const void* x, int x_dtype, const double* y, int N, int M, double* out, void* stream) {
    if (!x || !y || !out) { pg_set_error("haversine_matrix: null argument"); return PG_EINVAL; }
    if (N <= 0 || M <= 0) return PG_OK;
    if (N > 65535) { pg_set_error("haversine_matrix: at most 65535 rows per call"); return PG_EINVAL; }
    dim3 grid((M + 255) / 256, N);
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == PG_DTYPE_F64) hipLaunchKernelGGL(haversine_matrix_kernel<double>, grid, dim3(256), 0, s, (const double*)x, y, out, N, M);
    else if (x_dtype == PG_DTYPE_F32) hipLaunchKernelGGL(haversine_matrix_kernel<float>, grid, dim3(256), 0, s, (const float*)x, y, out, N, M);
    else { pg_set_error("haversine_matrix: x dtype must be PG_DTYPE_F32 or PG_DTYPE_F64"); return PG_EINVAL; }
    return pg_check_launch("haversine_matrix");
}

// pairwise form, reference preprocessing/geo_utils.py:40-55 (`haversine(x, y)`, rows paired): x is float64 (initial
// predictions / labels), y float32 or float64 -- the dtype rules of the refiner's veto call (models/proto_refiner.py:198-202).
template <typename Y>
__global__ __launch_bounds__(256) void haversine_pairs_kernel(const double* __restrict__ x, const Y* __restrict__ y,
                                                              double* __restrict__ out, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double xlng = x[2 * i] * DEG2RAD_D, xlat = x[2 * i + 1] * DEG2RAD_D;
    const Y ylng = Deg<Y>::rad(y[2 * i]), ylat = Deg<Y>::rad(y[2 * i + 1]);
    const double dlng = (double)ylng - xlng, dlat = (double)ylat - xlat;
    const double s1 = sin(dlat / 2), s0 = sin(dlng / 2);
    const double a = s1 * s1 + cos(xlat) * (double)Deg<Y>::cosv(ylat) * (s0 * s0);
    const double c = 2 * asin(sqrt(a));
    out[i] = (6378137.0 * c) / 1000;
}

extern "C" int pg_haversine_pairs(const double* x, const void* y, int y_dtype, int64_t N, double* out, void* stream) {
    if (!x || !y || !out) { pg_set_error("haversine_pairs: null argument"); return PG_EINVAL; }
    if (N <= 0) return PG_OK;
    dim3 grid((unsigned)((N + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == PG_DTYPE_F64) hipLaunchKernelGGL(haversine_pairs_kernel<double>, grid, dim3(256), 0, s, x, (const double*)y, out, N);
    else if (y_dtype == PG_DTYPE_F32) hipLaunchKernelGGL(haversine_pairs_kernel<float>, grid, dim3(256), 0, s, x, (const float*)y, out, N);
    else { pg_set_error("haversine_pairs: y dtype must be PG_DTYPE_F32 or PG_DTYPE_F64"); return PG_EINVAL; }
    return pg_check_launch("haversine_pairs");
}

// --------------------------------------------------------------------------------------------- label smoothing
// One block per row: row minimum (torch.min propagates NaN), then exp(-(d - min) / constant) with nan_to_num(0,0,0).
__global__ __launch_bounds__(256) void smooth_labels_kernel(const double* __restrict__ d, double* __restrict__ out, int M,
                                                            double constant) {
    __shared__ double smin[4];
    __shared__ int snan[4];
    const int64_t base = (int64_t)blockIdx.x * M;
    double m = INFINITY;
    int has_nan = 0;
    for (int j = threadIdx.x; j < M; j += 256) {
        const double v = d[base + j];
        if (v != v) has_nan = 1;
        m = v < m ? v : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double mo = __shfl_xor(m, o, 64);
        m = mo < m ? mo : m;
        has_nan |= __shfl_xor(has_nan, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = m; snan[threadIdx.x >> 6] = has_nan; }
    __syncthreads();
    m = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    has_nan = snan[0] | snan[1] | snan[2] | snan[3];
    if (has_nan) m = NAN;
    for (int j = threadIdx.x; j < M; j += 256) {
        const double adj = d[base + j] - m;
        double v = exp(-adj / constant);
        if (v != v || isinf(v)) v = 0.0;
        out[base + j] = v;
    }
}

extern "C" int pg_smooth_labels(const double* distances, int N, int M, double constant, double* out, void* stream) {
    if (!distances || !out) { pg_set_error("smooth_labels: null argument"); return PG_EINVAL; }
    if (N <= 0 || M <= 0) return PG_OK;
    hipLaunchKernelGGL(smooth_labels_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, distances, out, M, constant);
    return pg_check_launch("smooth_labels");
}
