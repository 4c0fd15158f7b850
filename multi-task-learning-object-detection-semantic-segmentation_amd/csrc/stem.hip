// Stem: Rescaling (x/127.5 - 1) fused into a dense 3x3 stride-2 SAME convolution with 3 input channels.
// Replaces `Rescaling` + the first Conv2D of both backbones (reference models.py:187,196 -> :65 and :622,628).
// HBM-bound (AI ~10 FLOP/B): a thread owns 4 output channels of one output pixel; the 8 (or 6) lanes of a pixel
// read the same 27 inputs (one broadcast transaction) and write one contiguous NHWC pixel.
#include "common.h"

int ssdseg_colsum(ssdseg_ctx* ctx, const float* part, int nparts, long long len, float* out);

namespace {

constexpr int CIN = 3;
constexpr int MAX_BLOCKS = 2048;

struct StemGeom {
    int n, h, w, cout, cv, ho, wo, pt, pl;
    long long npix;
    float in_scale, in_offset;
};

__device__ __forceinline__ void add4(float4& a, float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void fma4s(float4& acc, float a, float4 b) {
    acc.x = fmaf(a, b.x, acc.x); acc.y = fmaf(a, b.y, acc.y); acc.z = fmaf(a, b.z, acc.z); acc.w = fmaf(a, b.w, acc.w);
}

__device__ __forceinline__ float4 reduce_over_y(float4 v, float4* red) {
    __syncthreads();
    red[threadIdx.y * blockDim.x + threadIdx.x] = v;
    __syncthreads();
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.y == 0)
        for (int y = 0; y < (int)blockDim.y; ++y) add4(r, red[y * blockDim.x + threadIdx.x]);
    return r;
}

// loads the rescaled 3x3x3 window of output pixel (n, ho, wo); zero outside the image
__device__ __forceinline__ void load_window(const StemGeom& g, const float* __restrict__ x, int n, int ho, int wo, float win[27]) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 + kh - g.pt;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wi = wo * 2 + kw - g.pl;
            const bool in = hi >= 0 && hi < g.h && wi >= 0 && wi < g.w;
            const float* p = x + (((long long)n * g.h + hi) * g.w + wi) * CIN;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) win[(kh * 3 + kw) * CIN + ci] = in ? fmaf(p[ci], g.in_scale, g.in_offset) : 0.f;
        }
    }
}

__global__ void __launch_bounds__(512) stem_fwd_kernel(StemGeom g, const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ stats) {
    extern __shared__ float4 red[];
    const int cvi = threadIdx.x;  // blockDim.x == cv
    const int c0 = cvi * 4;
    float4 wk[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wk[t] = ld4(w + (long long)t * g.cout + c0);
    const float4 b4 = bias ? ld4(bias + c0) : f4(0.f);
    float4 ssum = f4(0.f), ssq = f4(0.f);
    for (long long pix = (long long)blockIdx.x * blockDim.y + threadIdx.y; pix < g.npix; pix += (long long)gridDim.x * blockDim.y) {
        const int wo = (int)(pix % g.wo);
        const long long r = pix / g.wo;
        const int ho = (int)(r % g.ho);
        const int n = (int)(r / g.ho);
        float win[27];
        load_window(g, x, n, ho, wo, win);
        float4 acc = b4;
#pragma unroll
        for (int t = 0; t < 27; ++t) fma4s(acc, win[t], wk[t]);
        st4(y + pix * g.cout + c0, acc);
        add4(ssum, acc);
        ssq.x = fmaf(acc.x, acc.x, ssq.x); ssq.y = fmaf(acc.y, acc.y, ssq.y); ssq.z = fmaf(acc.z, acc.z, ssq.z); ssq.w = fmaf(acc.w, acc.w, ssq.w);
    }
    if (stats != nullptr) {
        float4 a = reduce_over_y(ssum, red);
        float4 b = reduce_over_y(ssq, red);
        if (threadIdx.y == 0) {
            float* row = stats + (long long)blockIdx.x * 2 * g.cout;
            st4(row + c0, a);
            st4(row + g.cout + c0, b);
        }
    }
}

struct GCoef {
    const float* g;
    const float* y;
    const float* scale;
    const float* shift;
    const float* k1;
    const float* k0;
    int act;
};

// partial[blk][28][cout]: rows 0..26 = dW taps, row 27 = dbias
__global__ void __launch_bounds__(512) stem_bwd_weight_kernel(StemGeom g, const float* __restrict__ x, GCoef dy, float* __restrict__ part) {
    extern __shared__ float4 red[];
    const int cvi = threadIdx.x;
    const int c0 = cvi * 4;
    const bool aff = dy.scale != nullptr;
    float4 cs = f4(0.f), ct = f4(0.f), ck1 = f4(0.f), ck0 = f4(0.f);
    if (aff) { cs = ld4(dy.scale + c0); ct = ld4(dy.shift + c0); ck1 = ld4(dy.k1 + c0); ck0 = ld4(dy.k0 + c0); }
    float4 acc[28];
#pragma unroll
    for (int t = 0; t < 28; ++t) acc[t] = f4(0.f);
    for (long long pix = (long long)blockIdx.x * blockDim.y + threadIdx.y; pix < g.npix; pix += (long long)gridDim.x * blockDim.y) {
        const int wo = (int)(pix % g.wo);
        const long long r = pix / g.wo;
        const int ho = (int)(r % g.ho);
        const int n = (int)(r / g.ho);
        float win[27];
        load_window(g, x, n, ho, wo, win);
        float4 d = ld4(dy.g + pix * g.cout + c0);
        if (aff) d = gview_apply4(d, ld4(dy.y + pix * g.cout + c0), cs, ct, ck1, ck0, dy.act);
#pragma unroll
        for (int t = 0; t < 27; ++t) fma4s(acc[t], win[t], d);
        add4(acc[27], d);
    }
#pragma unroll
    for (int t = 0; t < 28; ++t) {
        float4 v = reduce_over_y(acc[t], red);
        if (threadIdx.y == 0) st4(part + ((long long)blockIdx.x * 28 + t) * g.cout + c0, v);
    }
}

int stem_geometry(int n, int h, int w, int cout, StemGeom* g, dim3* grid, dim3* block, size_t* lds) {
    g->n = n; g->h = h; g->w = w; g->cout = cout; g->cv = cout / 4;
    same_pad(h, 3, 2, 1, &g->ho, &g->pt);
    same_pad(w, 3, 2, 1, &g->wo, &g->pl);
    g->npix = (long long)n * g->ho * g->wo;
    int bx = g->cv;
    int by = 256 / bx;
    if (by < 1) by = 1;
    long long want = (g->npix + (long long)by * 4 - 1) / ((long long)by * 4);
    int gx = (int)(want < MAX_BLOCKS ? want : MAX_BLOCKS);
    if (gx < 1) gx = 1;
    *block = dim3(bx, by, 1);
    *grid = dim3(gx, 1, 1);
    *lds = (size_t)bx * by * sizeof(float4);
    return 0;
}

}  // namespace

extern "C" {

int ssdseg_stem_conv_parts(int n, int h, int w, int cout, int* nparts_host) {
    SSDSEG_ARG(n > 0 && h > 0 && w > 0, 1);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0 && cout <= 256, 4);
    SSDSEG_ARG(nparts_host != nullptr, 5);
    StemGeom g;
    dim3 grid, block;
    size_t lds;
    stem_geometry(n, h, w, cout, &g, &grid, &block, &lds);
    *nparts_host = (int)grid.x;
    return 0;
}

int ssdseg_stem_conv_fwd(ssdseg_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int n, int h, int wdt,
                         int cin, int cout, float in_scale, float in_offset, float* stats) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(x != nullptr, 2);
    SSDSEG_ARG(w != nullptr, 3);
    SSDSEG_ARG(y != nullptr, 5);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(cin == CIN, 9);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0 && cout <= 256, 10);
    StemGeom g;
    dim3 grid, block;
    size_t lds;
    stem_geometry(n, h, wdt, cout, &g, &grid, &block, &lds);
    g.in_scale = in_scale;
    g.in_offset = in_offset;
    SSDSEG_LAUNCH(ctx, 4.0 * ((double)n * h * wdt * CIN + (double)g.npix * cout + 27.0 * cout), 54.0 * g.npix * cout, stem_fwd_kernel, grid, block, lds, g, x, w, bias, y, stats);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_stem_conv_bwd_weight(ssdseg_ctx* ctx, const float* x, const ssdseg_gview* dy, float* dw, float* dbias, int n, int h,
                                int wdt, int cin, int cout, float in_scale, float in_offset) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(x != nullptr, 2);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 3);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 3);
    SSDSEG_ARG(dw != nullptr, 4);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(cin == CIN, 9);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0 && cout <= 256, 10);
    StemGeom g;
    dim3 grid, block;
    size_t lds;
    stem_geometry(n, h, wdt, cout, &g, &grid, &block, &lds);
    g.in_scale = in_scale;
    g.in_offset = in_offset;
    // partial slab [parts][28][cout] followed by the reduced [28][cout]
    const size_t part_floats = (size_t)grid.x * 28 * cout;
    void* ws;
    int rc = ssdseg_workspace(ctx, (part_floats + 28 * (size_t)cout) * sizeof(float), &ws);
    if (rc) return rc;
    float* part = (float*)ws;
    float* reduced = part + part_floats;
    GCoef gc{dy->g, dy->y, dy->scale, dy->shift, dy->k1, dy->k0, dy->act};
    SSDSEG_LAUNCH(ctx, 4.0 * ((double)n * h * wdt * CIN + (double)g.npix * cout + 27.0 * cout), 54.0 * g.npix * cout, stem_bwd_weight_kernel, grid, block, lds, g, x, gc, part);
    SSDSEG_LAUNCH_CHECK();
    rc = ssdseg_colsum(ctx, part, (int)grid.x, 28LL * cout, reduced);
    if (rc) return rc;
    SSDSEG_HIP(hipMemcpyAsync(dw, reduced, 27 * (size_t)cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    if (dbias) SSDSEG_HIP(hipMemcpyAsync(dbias, reduced + 27 * (size_t)cout, (size_t)cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

}  // extern "C"
