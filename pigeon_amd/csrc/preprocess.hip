// preprocess.hip -- CLIP image preprocessing on the GPU: uint8 RGB (N,H,W,3) -> pixel_values (N,3,336,336), the step in
// FRONT of the hot path.
//
// Replaces `CLIPProcessor(images=pil_image, return_tensors='pt')` at reference models/clip_embedder.py:52,
// dataset_creation/finetune/embed_dataset.py:20, preprocessing/dataset_preprocessing.py:193,
// dataset_creation/benchmark/benchmark_dataset.py:99, i.e. (transformers 4.23.1 CLIPFeatureExtractor, reference env.yml:60)
//   resize shorter edge -> 336 with PIL BICUBIC ; centre crop 336x336 ; float32 / 255.0 ; (x - mean) / std ; HWC -> CHW
// bit for bit: Pillow resamples 8-bit images in FIXED POINT (libImaging/Resample.c: 22-bit coefficients, horizontal pass
// then vertical pass, the intermediate image rounded and clipped to uint8), so the same integer arithmetic on the GPU
// reproduces the uint8 result exactly; the float part takes only 3 x 256 distinct values and is a lookup table built with
// IEEE float32 divisions.  The 1.35 MB/image fp32 host->device stream of the reference becomes <= 1.2 MB of uint8 (640x640)
// and the ViT's im2col reads 16-bit pixels.
//
// Kernels (both HBM-bound byte streams, one block per image row):
//   prep_h_kernel   source row (W x 3 bytes) staged in LDS; thread xo of 336 accumulates its taps for R,G,B -> uint8 temp
//   prep_v_kernel   output row yo: thread xo accumulates the vertical taps over temp rows (coalesced 3-byte pixels),
//                   clips, looks up the normalised value and writes the three channel planes (coalesced along x)
// Only the rows / columns the 336x336 crop needs are ever computed (Pillow computes the full resized image, then crops:
// rows and columns are independent, so the cropped values are identical).
#include "common.h"
#include "pigeon_internal.h"

#include <cmath>
#include <vector>

#define PREP_BITS 22              // Pillow PRECISION_BITS = 32 - 8 - 2
#define PREP_SIZE 336
#define PREP_TROW (PREP_SIZE * 3) // bytes per temp row

struct pg_prep {
    int device = 0;
    int in_h = 0, in_w = 0, new_h = 0, new_w = 0, top = 0, left = 0;
    int ksize_h = 0, ksize_v = 0;
    int row0 = 0, nrows = 0;                     // source rows the vertical pass needs
    int32_t *bounds_h = nullptr, *kk_h = nullptr, *bounds_v = nullptr, *kk_v = nullptr;
    float* lut = nullptr;                        // [3][256]
};

// ---- host: Pillow's coefficient tables, in double, same expression order as Resample.c --------------------------------
static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Full-box precompute_coeffs + normalize_coeffs_8bpc for outputs [o0, o0 + PREP_SIZE) of an in_size -> out_size resize.
// in_size == out_size is Pillow's "pass not needed" case: identity taps (2^22 at the pixel itself reproduce it exactly).
static int make_coeffs(int in_size, int out_size, int o0, std::vector<int32_t>& bounds, std::vector<int32_t>& kk) {
    bounds.assign(PREP_SIZE * 2, 0);
    if (in_size == out_size) {
        kk.assign(PREP_SIZE, 1 << PREP_BITS);
        for (int i = 0; i < PREP_SIZE; ++i) { bounds[2 * i] = o0 + i; bounds[2 * i + 1] = 1; }
        return 1;
    }
    double scale, filterscale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    kk.assign((size_t)PREP_SIZE * ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int i = 0; i < PREP_SIZE; ++i) {
        const int xx = o0 + i;
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            w[x] = bicubic_filter((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            double v = w[x];
            if (ww != 0.0) v /= ww;
            kk[(size_t)i * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PREP_BITS)) : (int)(0.5 + v * (1 << PREP_BITS));
        }
        bounds[2 * i] = xmin;
        bounds[2 * i + 1] = xmax;
    }
    return ksize;
}

template <typename T>
static int upload(const std::vector<T>& v, T** dst) {
    PG_HIP(hipMalloc((void**)dst, v.size() * sizeof(T)));
    PG_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return PG_OK;
}

extern "C" int pg_prep_create(pg_prep** out, int device, int in_h, int in_w) {
    if (!out) { pg_set_error("prep_create: null argument"); return PG_EINVAL; }
    if (in_h < 1 || in_w < 1 || in_h > 16384 || in_w > 16384) {
        pg_set_error("prep_create: image size %dx%d out of range (1..16384)", in_h, in_w);
        return PG_EINVAL;
    }
    int n = 0;
    PG_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { pg_set_error("prep_create: device %d of %d", device, n); return PG_EINVAL; }
    PG_HIP(hipSetDevice(device));
    pg_prep* h = new pg_prep();
    h->device = device; h->in_h = in_h; h->in_w = in_w;
    // transformers 4.23.1 ImageFeatureExtractionMixin.resize(size=336, default_to_square=False)
    const int shortside = in_w <= in_h ? in_w : in_h, longside = in_w <= in_h ? in_h : in_w;
    int new_short = shortside, new_long = longside;
    if (shortside != PREP_SIZE) { new_short = PREP_SIZE; new_long = (int)((double)PREP_SIZE * longside / shortside); }
    h->new_w = in_w <= in_h ? new_short : new_long;
    h->new_h = in_w <= in_h ? new_long : new_short;
    if (h->new_w < PREP_SIZE || h->new_h < PREP_SIZE) {
        delete h;
        pg_set_error("prep_create: resized image %dx%d is smaller than the 336x336 crop", h->new_h, h->new_w);
        return PG_EINVAL;
    }
    h->top = (h->new_h - PREP_SIZE) / 2;
    h->left = (h->new_w - PREP_SIZE) / 2;
    std::vector<int32_t> bh, kh, bv, kv;
    h->ksize_h = make_coeffs(in_w, h->new_w, h->left, bh, kh);
    h->ksize_v = make_coeffs(in_h, h->new_h, h->top, bv, kv);
    h->row0 = bv[0];
    h->nrows = bv[2 * (PREP_SIZE - 1)] + bv[2 * (PREP_SIZE - 1) + 1] - h->row0;
    for (int i = 0; i < PREP_SIZE; ++i) bv[2 * i] -= h->row0;           // vertical taps index the temp image
    // ((v / 255.0f) - mean) / std in IEEE float32 (what numpy does for float32 arrays)
    std::vector<float> lut(3 * 256);
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float x = (float)v / 255.0f;
            volatile float y = x - mean[c];
            lut[c * 256 + v] = y / stdv[c];
        }
    int rc = upload(bh, &h->bounds_h);
    if (!rc) rc = upload(kh, &h->kk_h);
    if (!rc) rc = upload(bv, &h->bounds_v);
    if (!rc) rc = upload(kv, &h->kk_v);
    if (!rc) rc = upload(lut, &h->lut);
    if (rc) { pg_prep_destroy(h); return rc; }
    *out = h;
    return PG_OK;
}

extern "C" int pg_prep_destroy(pg_prep* h) {
    if (!h) return PG_OK;
    (void)hipFree(h->bounds_h); (void)hipFree(h->kk_h); (void)hipFree(h->bounds_v); (void)hipFree(h->kk_v); (void)hipFree(h->lut);
    delete h;
    return PG_OK;
}

extern "C" int pg_prep_geometry(const pg_prep* h, int32_t* out6) {
    if (!h || !out6) { pg_set_error("prep_geometry: null argument"); return PG_EINVAL; }
    out6[0] = h->new_h; out6[1] = h->new_w; out6[2] = h->top; out6[3] = h->left; out6[4] = h->row0; out6[5] = h->nrows;
    return PG_OK;
}

extern "C" int pg_prep_workspace_bytes(const pg_prep* h, int n_images, size_t* bytes) {
    if (!h || !bytes || n_images < 0) { pg_set_error("prep_workspace_bytes: bad argument"); return PG_EINVAL; }
    *bytes = (size_t)(n_images > 0 ? n_images : 1) * h->nrows * PREP_TROW + 256;
    return PG_OK;
}

// ---- device -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(384) void prep_h_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ tmp,
                                                     const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                     int ksize, int in_h, int in_w, int row0, int nrows) {
    extern __shared__ uint8_t srow[];
    const int n = blockIdx.y, r = blockIdx.x;
    const uint8_t* src = img + ((size_t)n * in_h + (row0 + r)) * (size_t)in_w * 3;
    const int nbytes = in_w * 3;
    for (int i = threadIdx.x; i < nbytes; i += blockDim.x) srow[i] = src[i];
    __syncthreads();
    const int xo = threadIdx.x;
    if (xo >= PREP_SIZE) return;
    const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
    const int32_t* k = kk + (size_t)xo * ksize;
    int s0 = 1 << (PREP_BITS - 1), s1 = s0, s2 = s0;
    const uint8_t* p = srow + xmin * 3;
    for (int x = 0; x < cnt; ++x) {
        const int w = k[x];
        s0 += (int)p[3 * x] * w; s1 += (int)p[3 * x + 1] * w; s2 += (int)p[3 * x + 2] * w;
    }
    uint8_t* o = tmp + ((size_t)n * nrows + r) * PREP_TROW + xo * 3;
    o[0] = (uint8_t)min(max(s0 >> PREP_BITS, 0), 255);
    o[1] = (uint8_t)min(max(s1 >> PREP_BITS, 0), 255);
    o[2] = (uint8_t)min(max(s2 >> PREP_BITS, 0), 255);
}

template <typename OUT>
__global__ __launch_bounds__(384) void prep_v_kernel(const uint8_t* __restrict__ tmp, OUT* __restrict__ out,
                                                     const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                     const float* __restrict__ lut, int ksize, int nrows) {
    __shared__ float slut[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) slut[i] = lut[i];
    __syncthreads();
    const int n = blockIdx.y, yo = blockIdx.x, xo = threadIdx.x;
    if (xo >= PREP_SIZE) return;
    const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
    const int32_t* k = kk + (size_t)yo * ksize;
    const uint8_t* p = tmp + ((size_t)n * nrows + ymin) * PREP_TROW + xo * 3;
    int s0 = 1 << (PREP_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < cnt; ++y) {
        const int w = k[y];
        const uint8_t* q = p + (size_t)y * PREP_TROW;
        s0 += (int)q[0] * w; s1 += (int)q[1] * w; s2 += (int)q[2] * w;
    }
    const int v0 = min(max(s0 >> PREP_BITS, 0), 255), v1 = min(max(s1 >> PREP_BITS, 0), 255), v2 = min(max(s2 >> PREP_BITS, 0), 255);
    const size_t plane = (size_t)PREP_SIZE * PREP_SIZE;
    OUT* o = out + (size_t)n * 3 * plane + (size_t)yo * PREP_SIZE + xo;
    if constexpr (sizeof(OUT) == 4) {
        o[0] = slut[v0]; o[plane] = slut[256 + v1]; o[2 * plane] = slut[512 + v2];
    } else {
        o[0] = f32_to_f16_bits(slut[v0]); o[plane] = f32_to_f16_bits(slut[256 + v1]); o[2 * plane] = f32_to_f16_bits(slut[512 + v2]);
    }
}

extern "C" int pg_prep_forward(pg_prep* h, const void* images_u8, int n_images, void* out, int out_dtype, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!h) { pg_set_error("prep_forward: null handle"); return PG_EINVAL; }
    if (n_images < 0) { pg_set_error("prep_forward: n_images = %d", n_images); return PG_EINVAL; }
    if (n_images == 0) return PG_OK;                       // an empty batch is a no-op: its (empty) buffers may be NULL
    if (!images_u8 || !out || !workspace) { pg_set_error("prep_forward: null argument"); return PG_EINVAL; }
    if (out_dtype != PG_DTYPE_F32 && out_dtype != PG_DTYPE_F16) { pg_set_error("prep_forward: out dtype must be F32 or F16"); return PG_EINVAL; }
    size_t need = 0;
    pg_prep_workspace_bytes(h, n_images, &need);
    if (workspace_bytes < need) { pg_set_error("prep_forward: workspace %zu < required %zu bytes", workspace_bytes, need); return PG_ENOMEM; }
    if (n_images > 65535) { pg_set_error("prep_forward: at most 65535 images per call"); return PG_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)h->in_w * 3;
    hipLaunchKernelGGL(prep_h_kernel, dim3(h->nrows, n_images), dim3(384), lds, s, (const uint8_t*)images_u8, (uint8_t*)workspace,
                       h->bounds_h, h->kk_h, h->ksize_h, h->in_h, h->in_w, h->row0, h->nrows);
    int rc = pg_check_launch("prep_h");
    if (rc) return rc;
    if (out_dtype == PG_DTYPE_F32)
        hipLaunchKernelGGL(prep_v_kernel<float>, dim3(PREP_SIZE, n_images), dim3(384), 0, s, (const uint8_t*)workspace, (float*)out,
                           h->bounds_v, h->kk_v, h->lut, h->ksize_v, h->nrows);
    else
        hipLaunchKernelGGL(prep_v_kernel<uint16_t>, dim3(PREP_SIZE, n_images), dim3(384), 0, s, (const uint8_t*)workspace, (uint16_t*)out,
                           h->bounds_v, h->kk_v, h->lut, h->ksize_v, h->nrows);
    return pg_check_launch("prep_v");
}
