// Pooling / resize / head-tail kernels of the DeepLabV3+ and SSDLite heads (all HBM-bound elementwise or gather
// kernels, float4 over the channel axis, NHWC):
//   GlobalAveragePooling2D keepdims                (reference blocks.py:57)
//   UpSampling2D(bilinear), half-pixel centres     (reference blocks.py:61,104,129; semantics SURVEY.md App. B.5)
//   mask head tail: x4 bilinear -> Softmax -> weighted cross-entropy, fused (blocks.py:128-130 + losses.py:294-303)
//   SSD head gather: Reshape(-1,4) + Concatenate(axis=1) (blocks.py:155, models.py:256,271) and Softmax (models.py:259)
#include "common.h"

int ssdseg_colsum(ssdseg_ctx* ctx, const float* part, int nparts, long long len, float* out);

namespace {

__device__ __forceinline__ void add4(float4& a, float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void axpy4(float4& a, float s, float4 b) {
    a.x = fmaf(s, b.x, a.x); a.y = fmaf(s, b.y, a.y); a.z = fmaf(s, b.z, a.z); a.w = fmaf(s, b.w, a.w);
}

int ew_blocks(long long total, int threads = 256) {
    long long b = (total + threads - 1) / threads;
    return (int)(b < 8192 ? (b < 1 ? 1 : b) : 8192);
}

// ------------------------------------------------------------------------------------------------ GAP
// one block per (image, 64-channel-vector group); threads (cv, y) walk the pixels, fixed-order reduction over y
// Sum over the `hw` pixels of "image" blockIdx.x (= one of the `chunks` equal slices of a real image when the caller splits
// the reduction so that n * chunks blocks fill the chip): out[blockIdx.x][c] = mul * sum_p act(s*x + t).  x rows are `ldx` apart.
__global__ void __launch_bounds__(512) gap_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int act, float* __restrict__ out, int hw, int c,
                                                      float mul) {
    extern __shared__ float4 red[];
    const int cv = c / 4;
    const int cvi = blockIdx.y * blockDim.x + threadIdx.x;
    const int n = blockIdx.x;
    float4 acc = f4(0.f);
    const bool aff = scale != nullptr;
    if (cvi < cv) {
        float4 s = f4(0.f), t = f4(0.f);
        if (aff) { s = ld4(scale + cvi * 4); t = ld4(shift + cvi * 4); }
        for (int p = threadIdx.y; p < hw; p += blockDim.y) add4(acc, view_apply4(ld4(x + ((long long)n * hw + p) * ldx + cvi * 4), s, t, aff, act));
    }
    red[threadIdx.y * blockDim.x + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && cvi < cv) {
        float4 r = f4(0.f);
        for (int y = 0; y < (int)blockDim.y; ++y) add4(r, red[y * blockDim.x + threadIdx.x]);
        st4(out + (long long)n * c + cvi * 4, make_float4(r.x * mul, r.y * mul, r.z * mul, r.w * mul));
    }
}

// second stage of a split pixel sum: out[n][.] (row stride ldo) = sum_k part[n][k][.] in fixed order (+ previous contents)
__global__ void chunk_sum_kernel(const float* __restrict__ part, int chunks, int cv, float* __restrict__ out, int ldo, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * cv) return;
    const int img = i / cv, c4 = i - img * cv;
    float4 r = f4(0.f);
    for (int k = 0; k < chunks; ++k) add4(r, ld4(part + ((long long)(img * chunks + k) * cv + c4) * 4));
    float* o = out + (long long)img * ldo + c4 * 4;
    if (accumulate) add4(r, ld4(o));
    st4(o, r);
}

// host side of the (optionally split) pixel sum: out[n][c] (row stride ldo) = mul * sum over the hw pixels of each image
int pixel_sum(ssdseg_ctx* ctx, const float* x, int ldx, const float* scale, const float* shift, int act, float* out, int ldo, int n, int hw,
              int c, float mul, int accumulate, double cost_bytes) {
    const int cv = c / 4;
    const int bx = cv < 128 ? cv : 128;
    int chunks = 1;   // equal slices only (deterministic, no ragged tail): the largest divisor of hw that still leaves >= 64 pixels
    for (int k = 16; k >= 2; --k)
        if (hw % k == 0 && hw / k >= 64 && n * k <= 1024) { chunks = k; break; }
    const int hwc = hw / chunks;
    int by = 512 / bx;
    if (by > hwc) by = hwc;
    if (by < 1) by = 1;
    if (chunks == 1 && !accumulate && ldo == c) {
        SSDSEG_LAUNCH(ctx, cost_bytes, 0.0, gap_fwd_kernel, dim3(n, cdiv(cv, bx)), dim3(bx, by), (size_t)bx * by * sizeof(float4), x, ldx, scale,
                      shift, act, out, hw, c, mul);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    void* ws;
    int rc = ssdseg_workspace(ctx, (size_t)n * chunks * c * sizeof(float), &ws);
    if (rc) return rc;
    SSDSEG_LAUNCH(ctx, cost_bytes, 0.0, gap_fwd_kernel, dim3(n * chunks, cdiv(cv, bx)), dim3(bx, by), (size_t)bx * by * sizeof(float4), x, ldx,
                  scale, shift, act, (float*)ws, hwc, c, mul);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 4.0 * n * (chunks + 1) * c, 0.0, chunk_sum_kernel, dim3(cdiv(n * cv, 256)), dim3(256), 0, (const float*)ws, chunks, cv, out,
                  ldo, n, accumulate);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

__global__ void gap_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, int n, int hw, int cv, int accumulate) {
    const long long total = (long long)n * hw * cv;
    const float inv = 1.f / (float)hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cv);
        const long long img = i / ((long long)hw * cv);
        float4 v = ld4(g + (img * cv + c4) * 4);
        v = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
        if (accumulate) add4(v, ld4(dx + i * 4));
        st4(dx + i * 4, v);
    }
}

// ------------------------------------------------------------------------------------------------ bilinear
// tf.image.resize(bilinear, half_pixel_centers=True): src = (dst + 0.5) * (in/out) - 0.5, clamped to [0, in-1]
struct Lerp {
    int i0, i1;
    float f;
};
__device__ __forceinline__ Lerp lerp_of(int dst, int in_size, float inv_factor) {
    float src = ((float)dst + 0.5f) * inv_factor - 0.5f;
    src = fminf(fmaxf(src, 0.f), (float)(in_size - 1));
    Lerp l;
    l.i0 = (int)floorf(src);
    l.i1 = l.i0 + 1 < in_size ? l.i0 + 1 : in_size - 1;
    l.f = src - (float)l.i0;
    return l;
}
// weight with which input index `i` contributes to output index `dst`
__device__ __forceinline__ float lerp_weight(int dst, int i, int in_size, float inv_factor) {
    const Lerp l = lerp_of(dst, in_size, inv_factor);
    return (l.i0 == i ? 1.f - l.f : 0.f) + (l.i1 == i ? l.f : 0.f);
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                    int ldx, float* __restrict__ out, int ldo, int n, int h, int w, int cv, int fy, int fx, int pad) {
    const int ho = h * fy, wo = w * fx;
    const int hp = ho + 2 * pad, wp = wo + 2 * pad;       // pad = 1: the output is the interior of a bordered [n][ho+2][wo+2] tensor
    const long long total = (long long)n * ho * wo * cv;
    const bool aff = scale != nullptr;
    const float ify = 1.f / (float)fy, ifx = 1.f / (float)fx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * 4;
        long long r = i / cv;
        const int ox = (int)(r % wo); r /= wo;
        const int oy = (int)(r % ho);
        const long long img = r / ho;
        const Lerp ly = lerp_of(oy, h, ify), lx = lerp_of(ox, w, ifx);
        float4 s = f4(0.f), t = f4(0.f);
        if (aff) { s = ld4(scale + c0); t = ld4(shift + c0); }
        const float* base = x + img * h * w * ldx + c0;
        const float4 v00 = view_apply4(ld4(base + ((long long)ly.i0 * w + lx.i0) * ldx), s, t, aff, act);
        const float4 v01 = view_apply4(ld4(base + ((long long)ly.i0 * w + lx.i1) * ldx), s, t, aff, act);
        const float4 v10 = view_apply4(ld4(base + ((long long)ly.i1 * w + lx.i0) * ldx), s, t, aff, act);
        const float4 v11 = view_apply4(ld4(base + ((long long)ly.i1 * w + lx.i1) * ldx), s, t, aff, act);
        float4 top, bot, o;
        top.x = v00.x + (v01.x - v00.x) * lx.f; top.y = v00.y + (v01.y - v00.y) * lx.f; top.z = v00.z + (v01.z - v00.z) * lx.f; top.w = v00.w + (v01.w - v00.w) * lx.f;
        bot.x = v10.x + (v11.x - v10.x) * lx.f; bot.y = v10.y + (v11.y - v10.y) * lx.f; bot.z = v10.z + (v11.z - v10.z) * lx.f; bot.w = v10.w + (v11.w - v10.w) * lx.f;
        o.x = top.x + (bot.x - top.x) * ly.f; o.y = top.y + (bot.y - top.y) * ly.f; o.z = top.z + (bot.z - top.z) * ly.f; o.w = top.w + (bot.w - top.w) * ly.f;
        st4(out + ((img * hp + oy + pad) * wp + ox + pad) * ldo + c0, o);
    }
}

// x4 in both directions (the DeepLabV3+ decoder's up-sampling of the ASPP output): a thread owns one INPUT pixel's 4 x 4 block of
// outputs and one 4-channel vector.  Those sixteen outputs interpolate between the 3 x 3 inputs around it, which the thread loads
// (and activates) once -- 9 loads per 16 outputs where the kernel above does 64 -- and every output is formed by the same
// expression from the same operands: bit-identical.
__global__ void __launch_bounds__(256) bilinear_fwd_x4_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int act, int ldx, float* __restrict__ out, int ldo, int n, int h, int w, int cv, int pad) {
    const long long total = (long long)n * h * w * cv;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c0 = (int)(i % cv) * 4;
    long long r = i / cv;
    const int bx = (int)(r % w); r /= w;
    const int by = (int)(r % h);
    const long long img = r / h;
    const bool aff = scale != nullptr;
    float4 s = f4(0.f), t = f4(0.f);
    if (aff) { s = ld4(scale + c0); t = ld4(shift + c0); }
    const float* base = x + img * h * w * ldx + c0;
    float4 v[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int yy = by - 1 + dy;
        yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int xx = bx - 1 + dx;
            xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
            v[dy][dx] = view_apply4(ld4(base + ((long long)yy * w + xx) * ldx), s, t, aff, act);
        }
    }
    const int ho = h * 4, wo = w * 4;
    const int hp = ho + 2 * pad, wp = wo + 2 * pad;
    Lerp lx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lx[j] = lerp_of(bx * 4 + j, w, 0.25f);
#pragma unroll
    for (int jy = 0; jy < 4; ++jy) {
        const int oy = by * 4 + jy;
        const Lerp ly = lerp_of(oy, h, 0.25f);
        // rows i0, i1 of the source are rows (i - (by - 1)) of the cache; a clamped border row was loaded under its clamped index
        const int r0 = ly.i0 - (by - 1), r1 = ly.i1 - (by - 1);
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const int q0 = lx[jx].i0 - (bx - 1), q1 = lx[jx].i1 - (bx - 1);
            float4 v00, v01, v10, v11;
            // (compile-time indexed selects: the cache stays in registers)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (a == r0 && b == q0) v00 = v[a][b];
                    if (a == r0 && b == q1) v01 = v[a][b];
                    if (a == r1 && b == q0) v10 = v[a][b];
                    if (a == r1 && b == q1) v11 = v[a][b];
                }
            const float fx = lx[jx].f, fy = ly.f;
            float4 top, bot, o;
            top.x = v00.x + (v01.x - v00.x) * fx; top.y = v00.y + (v01.y - v00.y) * fx; top.z = v00.z + (v01.z - v00.z) * fx; top.w = v00.w + (v01.w - v00.w) * fx;
            bot.x = v10.x + (v11.x - v10.x) * fx; bot.y = v10.y + (v11.y - v10.y) * fx; bot.z = v10.z + (v11.z - v10.z) * fx; bot.w = v10.w + (v11.w - v10.w) * fx;
            o.x = top.x + (bot.x - top.x) * fy; o.y = top.y + (bot.y - top.y) * fy; o.z = top.z + (bot.z - top.z) * fy; o.w = top.w + (bot.w - top.w) * fy;
            st4(out + ((img * hp + oy + pad) * wp + bx * 4 + jx + pad) * ldo + c0, o);
        }
    }
}

// gather form of the transposed resize: each input pixel sums the outputs that referenced it (deterministic)
__global__ void bilinear_bwd_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dx, int ldx, int n, int h, int w, int cv,
                                    int fy, int fx, int accumulate) {
    const int ho = h * fy, wo = w * fx;
    const long long total = (long long)n * h * w * cv;
    const float ify = 1.f / (float)fy, ifx = 1.f / (float)fx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * 4;
        long long r = i / cv;
        const int ix = (int)(r % w); r /= w;
        const int iy = (int)(r % h);
        const long long img = r / h;
        // outputs whose source coordinate lies in (iy-1, iy+1): oy in ((iy-0.5)*fy - 0.5, (iy+1.5)*fy - 0.5)
        int oy0 = (iy == 0) ? 0 : (iy * fy - fy / 2 - fy), oy1 = (iy == h - 1) ? ho - 1 : (iy * fy + fy + fy / 2 + 1);
        int ox0 = (ix == 0) ? 0 : (ix * fx - fx / 2 - fx), ox1 = (ix == w - 1) ? wo - 1 : (ix * fx + fx + fx / 2 + 1);
        oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
        oy1 = oy1 > ho - 1 ? ho - 1 : oy1; ox1 = ox1 > wo - 1 ? wo - 1 : ox1;
        float4 acc = f4(0.f);
        for (int oy = oy0; oy <= oy1; ++oy) {
            const float wy = lerp_weight(oy, iy, h, ify);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const float wx = lerp_weight(ox, ix, w, ifx);
                if (wx == 0.f) continue;
                axpy4(acc, wy * wx, ld4(g + ((img * ho + oy) * wo + ox) * ldg + c0));
            }
        }
        float* p = dx + ((img * h + iy) * w + ix) * ldx + c0;
        if (accumulate) add4(acc, ld4(p));
        st4(p, acc);
    }
}

// x4 in both directions (the gradient of the decoder's up-sampled ASPP output, 614,400 x 256 -> 38,400 x 256 at batch 32): a thread
// owns a 2 x 2 block of INPUT pixels and one 4-channel vector.  The four pixels' supports (8 x 8 outputs each) overlap: their union
// is 12 x 12 outputs, every one of which is loaded ONCE and added to the up to four pixels it belongs to -- 36 loads per input
// pixel where the gather kernel above does 64 behind two lerp_weight evaluations each; the weights of the 12 rows / columns are
// formed once per thread by the same lerp_weight (borders and clamping included).  Fixed summation order (rows, then columns).
__global__ void __launch_bounds__(256) bilinear_bwd_x4_kernel(const float* __restrict__ g, int ldg, float* __restrict__ dx, int ldx, int n, int h, int w,
                                                             int cv, int accumulate) {
    const int hb = (h + 1) >> 1, wb = (w + 1) >> 1, ho = h * 4, wo = w * 4;
    const long long total = (long long)n * hb * wb * cv;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c0 = (int)(i % cv) * 4;
    long long r = i / cv;
    const int bx = (int)(r % wb); r /= wb;
    const int by = (int)(r % hb);
    const long long img = r / hb;
    const int iy0 = 2 * by, ix0 = 2 * bx;
    const int oy0 = 4 * iy0 - 2, ox0 = 4 * ix0 - 2;          // first output row / column of the union window (may be < 0)
    float wy[12][2], wx[12][2];
#pragma unroll
    for (int k = 0; k < 12; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = oy0 + k, ox = ox0 + k;
            wy[k][a] = (oy >= 0 && oy < ho && iy0 + a < h) ? lerp_weight(oy, iy0 + a, h, 0.25f) : 0.f;
            wx[k][a] = (ox >= 0 && ox < wo && ix0 + a < w) ? lerp_weight(ox, ix0 + a, w, 0.25f) : 0.f;
        }
    float4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f4(0.f);
    const float* base = g + img * ho * wo * (long long)ldg + c0;
#pragma unroll 2
    for (int ky = 0; ky < 12; ++ky) {
        const int oy = oy0 + ky;
        if (oy < 0 || oy >= ho) continue;
        const float* row = base + (long long)oy * wo * ldg;
        float4 v[12];
#pragma unroll
        for (int kx = 0; kx < 12; ++kx) {
            int ox = ox0 + kx;
            ox = ox < 0 ? 0 : (ox > wo - 1 ? wo - 1 : ox);      // (clamped columns carry weight 0)
            v[kx] = ld4(row + (long long)ox * ldg);
        }
#pragma unroll
        for (int kx = 0; kx < 12; ++kx)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) axpy4(acc[a][b], wy[ky][a] * wx[kx][b], v[kx]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (iy0 + a < h && ix0 + b < w) {
                float* p = dx + ((img * h + iy0 + a) * w + ix0 + b) * (long long)ldx + c0;
                float4 o = acc[a][b];
                if (accumulate) add4(o, ld4(p));
                st4(p, o);
            }
        }
}

// ------------------------------------------------------------------------------------------------ mask head (4 classes)
constexpr float KEPS = 1e-7f;  // tf.keras.backend.epsilon()

__device__ __forceinline__ float4 softmax4(float4 z) {
    const float m = fmaxf(fmaxf(z.x, z.y), fmaxf(z.z, z.w));
    float4 e = make_float4(expf(z.x - m), expf(z.y - m), expf(z.z - m), expf(z.w - m));
    const float inv = 1.f / (e.x + e.y + e.z + e.w);
    return make_float4(e.x * inv, e.y * inv, e.z * inv, e.w * inv);
}
__device__ __forceinline__ float4 up_logits(const float* __restrict__ logits, long long img, int h, int w, int oy, int ox, float ify, float ifx) {
    const Lerp ly = lerp_of(oy, h, ify), lx = lerp_of(ox, w, ifx);
    const float* base = logits + img * h * w * 4;
    const float4 v00 = ld4(base + ((long long)ly.i0 * w + lx.i0) * 4), v01 = ld4(base + ((long long)ly.i0 * w + lx.i1) * 4);
    const float4 v10 = ld4(base + ((long long)ly.i1 * w + lx.i0) * 4), v11 = ld4(base + ((long long)ly.i1 * w + lx.i1) * 4);
    float4 top, bot, o;
    top.x = v00.x + (v01.x - v00.x) * lx.f; top.y = v00.y + (v01.y - v00.y) * lx.f; top.z = v00.z + (v01.z - v00.z) * lx.f; top.w = v00.w + (v01.w - v00.w) * lx.f;
    bot.x = v10.x + (v11.x - v10.x) * lx.f; bot.y = v10.y + (v11.y - v10.y) * lx.f; bot.z = v10.z + (v11.z - v10.z) * lx.f; bot.w = v10.w + (v11.w - v10.w) * lx.f;
    o.x = top.x + (bot.x - top.x) * ly.f; o.y = top.y + (bot.y - top.y) * ly.f; o.z = top.z + (bot.z - top.z) * ly.f; o.w = top.w + (bot.w - top.w) * ly.f;
    return o;
}
__device__ __forceinline__ float clip_log(float p) { return logf(fminf(fmaxf(p, KEPS), 1.f - KEPS)); }
__device__ __forceinline__ float inside(float p) { return (p >= KEPS && p <= 1.f - KEPS) ? 1.f : 0.f; }

// dL/dp of one pixel.  mode 0: weighted cross-entropy, -w_c y_c / clip(p_c) inside the clip interval, 0 outside (App. B.6);
// mode 1 / 2: dice / dice_square (reference losses.py:204-216, 250-262): with the per-image sums I_c = sum y p, T_c = sum (y + p)
// [sum (y^2 + p^2)],  dL/dp_c = A_c y_c + B_c [2 p_c],  A_c = -2 w_c / (T_c + eps),  B_c = w_c (2 I_c + eps) / (T_c + eps)^2
// (cA, cB: written per image by mask_dice_final_kernel).
__device__ __forceinline__ float4 mask_dp(int mode, float4 cw, float4 y, float4 pr, float4 cA, float4 cB) {
    float4 dp;
    if (mode == 0) {
        dp.x = -cw.x * y.x / fminf(fmaxf(pr.x, KEPS), 1.f - KEPS) * inside(pr.x);
        dp.y = -cw.y * y.y / fminf(fmaxf(pr.y, KEPS), 1.f - KEPS) * inside(pr.y);
        dp.z = -cw.z * y.z / fminf(fmaxf(pr.z, KEPS), 1.f - KEPS) * inside(pr.z);
        dp.w = -cw.w * y.w / fminf(fmaxf(pr.w, KEPS), 1.f - KEPS) * inside(pr.w);
    } else {
        const float4 t = mode == 2 ? make_float4(2.f * pr.x, 2.f * pr.y, 2.f * pr.z, 2.f * pr.w) : f4(1.f);
        dp.x = fmaf(cA.x, y.x, cB.x * t.x);
        dp.y = fmaf(cA.y, y.y, cB.y * t.y);
        dp.z = fmaf(cA.z, y.z, cB.z * t.z);
        dp.w = fmaf(cA.w, y.w, cB.w * t.w);
    }
    return dp;
}

// dice / dice_square forward: probabilities (optional) and per-block partial sums (I_c, T_c); grid (blocks_per_image, n),
// partial[n][blocks_per_image][8]
__global__ void __launch_bounds__(256) mask_head_fwd_dice_kernel(const float* __restrict__ logits, int h, int w, int fy, int fx,
                                                                 const float* __restrict__ y_true, int squared, float* __restrict__ prob,
                                                                 float* __restrict__ partial) {
    __shared__ float red[8][256];
    const int ho = h * fy, wo = w * fx;
    const long long npix = (long long)ho * wo;
    const long long img = blockIdx.y;
    const float ify = 1.f / (float)fy, ifx = 1.f / (float)fx;
    float4 si = f4(0.f), st = f4(0.f);
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(p % wo), oy = (int)(p / wo);
        const float4 pr = softmax4(up_logits(logits, img, h, w, oy, ox, ify, ifx));
        const long long off = (img * npix + p) * 4;
        if (prob) st4(prob + off, pr);
        const float4 y = ld4(y_true + off);
        si.x = fmaf(y.x, pr.x, si.x); si.y = fmaf(y.y, pr.y, si.y); si.z = fmaf(y.z, pr.z, si.z); si.w = fmaf(y.w, pr.w, si.w);
        if (squared) {
            st.x += fmaf(y.x, y.x, pr.x * pr.x); st.y += fmaf(y.y, y.y, pr.y * pr.y); st.z += fmaf(y.z, y.z, pr.z * pr.z); st.w += fmaf(y.w, y.w, pr.w * pr.w);
        } else {
            st.x += y.x + pr.x; st.y += y.y + pr.y; st.z += y.z + pr.z; st.w += y.w + pr.w;
        }
    }
    const int t = threadIdx.x;
    red[0][t] = si.x; red[1][t] = si.y; red[2][t] = si.z; red[3][t] = si.w;
    red[4][t] = st.x; red[5][t] = st.y; red[6][t] = st.z; red[7][t] = st.w;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s)
#pragma unroll
            for (int v = 0; v < 8; ++v) red[v][t] += red[v][t + s];
        __syncthreads();
    }
    if (t < 8) partial[(img * gridDim.x + blockIdx.x) * 8 + t] = red[t][0];
}

// per image: I_c, T_c (partials summed in index order, double), the loss, and the backward coefficients coef[img] = (A_0..3, B_0..3)
__global__ void mask_dice_final_kernel(const float* __restrict__ partial, int nblk, int n, float4 cw, float* __restrict__ loss,
                                       float* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nblk; ++b)
#pragma unroll
        for (int v = 0; v < 8; ++v) s[v] += (double)partial[((long long)i * nblk + b) * 8 + v];
    const double w[4] = {cw.x, cw.y, cw.z, cw.w};
    const double eps = (double)KEPS;
    double l = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double den = s[4 + c] + eps, num = 2.0 * s[c] + eps;
        l += w[c] * (1.0 - num / den);
        if (coef != nullptr) {
            coef[i * 8 + c] = (float)(-2.0 * w[c] / den);
            coef[i * 8 + 4 + c] = (float)(w[c] * num / (den * den));
        }
    }
    if (loss != nullptr) loss[i] = (float)l;
}

// grid (blocks_per_image, n); partial[n][blocks_per_image] per-image loss partials
__global__ void __launch_bounds__(256) mask_head_fwd_kernel(const float* __restrict__ logits, int h, int w, int fy, int fx,
                                                            const float* __restrict__ y_true, float4 cw, float* __restrict__ prob,
                                                            float* __restrict__ partial) {
    __shared__ float red[256];
    const int ho = h * fy, wo = w * fx;
    const long long npix = (long long)ho * wo;
    const long long img = blockIdx.y;
    const float ify = 1.f / (float)fy, ifx = 1.f / (float)fx;
    float loss = 0.f;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(p % wo), oy = (int)(p / wo);
        const float4 pr = softmax4(up_logits(logits, img, h, w, oy, ox, ify, ifx));
        const long long off = (img * npix + p) * 4;
        if (prob) st4(prob + off, pr);
        if (y_true) {
            const float4 y = ld4(y_true + off);
            loss -= cw.x * y.x * clip_log(pr.x) + cw.y * y.y * clip_log(pr.y) + cw.z * y.z * clip_log(pr.z) + cw.w * y.w * clip_log(pr.w);
        }
    }
    if (partial) {
        red[threadIdx.x] = loss;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[img * gridDim.x + blockIdx.x] = red[0];
    }
}

__global__ void mask_loss_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ loss, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)partial[(long long)i * nblk + b];
    loss[i] = (float)s;
}

// dlogits(low res) = sum over the full-res pixels that interpolate from it of weight * dz, dz = softmax'(dL/dp)
__global__ void __launch_bounds__(256) mask_head_bwd_kernel(const float* __restrict__ logits, int n, int h, int w, int fy, int fx,
                                                            const float* __restrict__ y_true, float4 cw, float loss_scale,
                                                            float* __restrict__ dlogits, int mode, const float* __restrict__ coef) {
    const int ho = h * fy, wo = w * fx;
    const long long total = (long long)n * h * w;
    const float ify = 1.f / (float)fy, ifx = 1.f / (float)fx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ix = (int)(i % w);
        const int iy = (int)((i / w) % h);
        const long long img = i / ((long long)w * h);
        int oy0 = (iy == 0) ? 0 : (iy * fy - fy / 2 - fy), oy1 = (iy == h - 1) ? ho - 1 : (iy * fy + fy + fy / 2 + 1);
        int ox0 = (ix == 0) ? 0 : (ix * fx - fx / 2 - fx), ox1 = (ix == w - 1) ? wo - 1 : (ix * fx + fx + fx / 2 + 1);
        oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
        oy1 = oy1 > ho - 1 ? ho - 1 : oy1; ox1 = ox1 > wo - 1 ? wo - 1 : ox1;
        float4 acc = f4(0.f);
        const float4 cA = mode != 0 ? ld4(coef + img * 8) : f4(0.f), cB = mode != 0 ? ld4(coef + img * 8 + 4) : f4(0.f);
        for (int oy = oy0; oy <= oy1; ++oy) {
            const float wy = lerp_weight(oy, iy, h, ify);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const float wx = lerp_weight(ox, ix, w, ifx);
                if (wx == 0.f) continue;
                const float4 pr = softmax4(up_logits(logits, img, h, w, oy, ox, ify, ifx));
                const float4 y = ld4(y_true + ((img * ho + oy) * wo + ox) * 4);
                const float4 dp = mask_dp(mode, cw, y, pr, cA, cB);
                const float dot = dp.x * pr.x + dp.y * pr.y + dp.z * pr.z + dp.w * pr.w;
                const float wgt = wy * wx * loss_scale;
                acc.x = fmaf(wgt, pr.x * (dp.x - dot), acc.x);
                acc.y = fmaf(wgt, pr.y * (dp.y - dot), acc.y);
                acc.z = fmaf(wgt, pr.z * (dp.z - dot), acc.z);
                acc.w = fmaf(wgt, pr.w * (dp.w - dot), acc.w);
            }
        }
        st4(dlogits + i * 4, acc);
    }
}

// The same sum with each full-resolution pixel's dz computed ONCE per block instead of once per contributing low-resolution pixel
// (x16 up-sampling area: every dz has up to four takers, and the kernel above re-does the bilinear gather, the softmax and the
// one-hot read for each of them: 0.31 ms at 480x640, batch 32).  A block owns a TL x TL tile of low-resolution pixels, writes the
// dz of the (TL*F + F) x (TL*F + F) full-resolution pixels that can reach them into LDS, then every thread gathers its window.
// Each dz is the same expression and the window is walked in the same (oy, ox) order as above: results are bit-identical.
template <int F, int TL>
__global__ void __launch_bounds__(TL * TL) mask_head_bwd_tile_kernel(const float* __restrict__ logits, int n, int h, int w,
                                                                       const float* __restrict__ y_true, float4 cw, float loss_scale,
                                                                       float* __restrict__ dlogits, int mode, const float* __restrict__ coef) {
    constexpr int R = TL * F + F;                    // full-resolution rows / columns a tile can draw from (F/2 + F/2 beyond each side)
    extern __shared__ float4 dz[];                    // [R][R]
    const int ho = h * F, wo = w * F;
    const float inv = 1.f / (float)F;
    const int tiles_x = (w + TL - 1) / TL, tiles_y = (h + TL - 1) / TL;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const long long img = b / tiles_y;
    const int iy0 = ty * TL, ix0 = tx * TL;
    // low-res pixel i takes weight from the full-resolution indices whose source coordinate (o + 0.5) / F - 0.5 lies in (i - 1, i + 1):
    // o in [i*F - F/2, i*F + F + F/2 - 1]; a tile of TL pixels therefore draws from R = TL*F + F of them, starting at i0*F - F/2
    // (the candidate loops below run a little wider and skip the zero weights before touching LDS)
    const int oyb = iy0 * F - F / 2, oxb = ix0 * F - F / 2;
    const float4 cA = mode != 0 ? ld4(coef + img * 8) : f4(0.f), cB = mode != 0 ? ld4(coef + img * 8 + 4) : f4(0.f);
    for (int i = threadIdx.x; i < R * R; i += TL * TL) {
        const int ry = i / R, rx = i - ry * R;
        const int oy = oyb + ry, ox = oxb + rx;
        float4 v = f4(0.f);
        if (oy >= 0 && oy < ho && ox >= 0 && ox < wo) {
            const float4 pr = softmax4(up_logits(logits, img, h, w, oy, ox, inv, inv));
            const float4 y = ld4(y_true + ((img * ho + oy) * wo + ox) * 4);
            const float4 dp = mask_dp(mode, cw, y, pr, cA, cB);
            const float dot = dp.x * pr.x + dp.y * pr.y + dp.z * pr.z + dp.w * pr.w;
            v = make_float4(pr.x * (dp.x - dot), pr.y * (dp.y - dot), pr.z * (dp.z - dot), pr.w * (dp.w - dot));
        }
        dz[i] = v;
    }
    __syncthreads();
    const int ly = threadIdx.x / TL, lx = threadIdx.x - ly * TL;
    const int iy = iy0 + ly, ix = ix0 + lx;
    if (iy >= h || ix >= w) return;
    int oy0 = (iy == 0) ? 0 : (iy * F - F / 2 - F), oy1 = (iy == h - 1) ? ho - 1 : (iy * F + F + F / 2 + 1);
    int ox0 = (ix == 0) ? 0 : (ix * F - F / 2 - F), ox1 = (ix == w - 1) ? wo - 1 : (ix * F + F + F / 2 + 1);
    oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
    oy1 = oy1 > ho - 1 ? ho - 1 : oy1; ox1 = ox1 > wo - 1 ? wo - 1 : ox1;
    // the column weights of the window once per thread, not once per (row, column) -- the same lerp_weight values, the same products and
    // the same summation order (rows, then columns, zero weights skipped): bit-identical, ~130 weight evaluations per thread less
    constexpr int WMAX = 3 * F + 2;                   // candidate columns of a window: [i*F - F/2 - F, i*F + F + F/2 + 1]
    float wxs[WMAX];
#pragma unroll
    for (int k = 0; k < WMAX; ++k) wxs[k] = (ox0 + k <= ox1) ? lerp_weight(ox0 + k, ix, w, inv) : 0.f;
    float4 acc = f4(0.f);
    for (int oy = oy0; oy <= oy1; ++oy) {
        const float wy = lerp_weight(oy, iy, h, inv);
        if (wy == 0.f) continue;
        const float4* drow = dz + (oy - oyb) * R + (ox0 - oxb);
#pragma unroll
        for (int k = 0; k < WMAX; ++k) {
            const float wx = wxs[k];
            if (wx == 0.f) continue;
            const float4 d = drow[k];
            const float wgt = wy * wx * loss_scale;
            acc.x = fmaf(wgt, d.x, acc.x);
            acc.y = fmaf(wgt, d.y, acc.y);
            acc.z = fmaf(wgt, d.z, acc.z);
            acc.w = fmaf(wgt, d.w, acc.w);
        }
    }
    st4(dlogits + ((img * h + iy) * w + ix) * 4, acc);
}

// Larger factors (x8: ShuffleNetV2's 60 x 80 logits): a window holds (2F + F/2)^2 full-resolution pixels, so the gather is split over
// PARTS threads per low-resolution pixel (interleaved rows, partials folded in part order: deterministic, but not the summation
// order of the one-thread kernel) and all TL^2 * PARTS threads share the dz phase.
template <int F, int TL, int PARTS>
__global__ void __launch_bounds__(TL * TL * PARTS) mask_head_bwd_tile_split_kernel(const float* __restrict__ logits, int n, int h, int w,
                                                                                     const float* __restrict__ y_true, float4 cw, float loss_scale,
                                                                                     float* __restrict__ dlogits, int mode, const float* __restrict__ coef) {
    constexpr int R = TL * F + F, NT = TL * TL * PARTS;
    extern __shared__ float4 dz[];                    // [R][R] + [PARTS][TL * TL]
    float4* red = dz + R * R;
    const int ho = h * F, wo = w * F;
    const float inv = 1.f / (float)F;
    const int tiles_x = (w + TL - 1) / TL, tiles_y = (h + TL - 1) / TL;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const long long img = b / tiles_y;
    const int iy0 = ty * TL, ix0 = tx * TL;
    const int oyb = iy0 * F - F / 2, oxb = ix0 * F - F / 2;
    const float4 cA = mode != 0 ? ld4(coef + img * 8) : f4(0.f), cB = mode != 0 ? ld4(coef + img * 8 + 4) : f4(0.f);
    for (int i = threadIdx.x; i < R * R; i += NT) {
        const int ry = i / R, rx = i - ry * R;
        const int oy = oyb + ry, ox = oxb + rx;
        float4 v = f4(0.f);
        if (oy >= 0 && oy < ho && ox >= 0 && ox < wo) {
            const float4 pr = softmax4(up_logits(logits, img, h, w, oy, ox, inv, inv));
            const float4 y = ld4(y_true + ((img * ho + oy) * wo + ox) * 4);
            const float4 dp = mask_dp(mode, cw, y, pr, cA, cB);
            const float dot = dp.x * pr.x + dp.y * pr.y + dp.z * pr.z + dp.w * pr.w;
            v = make_float4(pr.x * (dp.x - dot), pr.y * (dp.y - dot), pr.z * (dp.z - dot), pr.w * (dp.w - dot));
        }
        dz[i] = v;
    }
    __syncthreads();
    const int pix = threadIdx.x % (TL * TL), part = threadIdx.x / (TL * TL);
    const int ly = pix / TL, lx = pix - ly * TL;
    const int iy = iy0 + ly, ix = ix0 + lx;
    const bool live = iy < h && ix < w;
    float4 acc = f4(0.f);
    if (live) {
        int oy0 = (iy == 0) ? 0 : (iy * F - F / 2 - F), oy1 = (iy == h - 1) ? ho - 1 : (iy * F + F + F / 2 + 1);
        int ox0 = (ix == 0) ? 0 : (ix * F - F / 2 - F), ox1 = (ix == w - 1) ? wo - 1 : (ix * F + F + F / 2 + 1);
        oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
        oy1 = oy1 > ho - 1 ? ho - 1 : oy1; ox1 = ox1 > wo - 1 ? wo - 1 : ox1;
        for (int oy = oy0 + part; oy <= oy1; oy += PARTS) {
            const float wy = lerp_weight(oy, iy, h, inv);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const float wx = lerp_weight(ox, ix, w, inv);
                if (wx == 0.f) continue;
                const float4 d = dz[(oy - oyb) * R + (ox - oxb)];
                const float wgt = wy * wx * loss_scale;
                acc.x = fmaf(wgt, d.x, acc.x);
                acc.y = fmaf(wgt, d.y, acc.y);
                acc.z = fmaf(wgt, d.z, acc.z);
                acc.w = fmaf(wgt, d.w, acc.w);
            }
        }
    }
    red[part * TL * TL + pix] = acc;
    __syncthreads();
    if (part == 0 && live) {
        float4 s = red[pix];
#pragma unroll
        for (int q = 1; q < PARTS; ++q) { const float4 r = red[q * TL * TL + pix]; s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w; }
        st4(dlogits + ((img * h + iy) * w + ix) * 4, s);
    }
}

// ------------------------------------------------------------------------------------------------ SSD head gather / softmax
// forward: out[b][off + r] = view(in)[b][r], r in [0, in_img_elems), channel of element = r % c (float4 granules)
// reverse: in_grad[b][r] = out_grad[b][off + r]
__global__ void head_gather_kernel(const float* __restrict__ src, const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                   float* __restrict__ dst, int b, int in_img_elems, int c, int out_off, int out_img_elems, int reverse) {
    const int per = in_img_elems / 4;
    const long long total = (long long)b * per;
    const bool aff = scale != nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / per;
        const int r = (int)(i % per) * 4;
        const long long dense = img * in_img_elems + r, strided = img * out_img_elems + out_off + r;
        if (!reverse) {
            float4 s = f4(0.f), t = f4(0.f);
            const int c0 = r % c;
            if (aff) { s = ld4(scale + c0); t = ld4(shift + c0); }
            st4(dst + strided, view_apply4(ld4(src + dense), s, t, aff, act));
        } else {
            st4(dst + dense, ld4(src + strided));
        }
    }
}

__global__ void softmax_rows4_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                     float* __restrict__ out, long long rows) {
    const bool aff = scale != nullptr;
    float4 s = f4(0.f), t = f4(0.f);
    if (aff) { s = ld4(scale); t = ld4(shift); }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long long)gridDim.x * blockDim.x)
        st4(out + i * 4, softmax4(view_apply4(ld4(x + i * 4), s, t, aff, act)));
}


// ------------------------------------------------------------------------------------------------ training metrics
// (reference metrics.py; per-image values, Keras averages them).  All float reductions are two-level in fixed order.

// soft Jaccard of the segmentation masks (metrics.py:35-47): per image and class  inter = sum t*p,  total = sum (t + p)
// over the full-resolution pixels; p = softmax(upsampled logits) when FROM_LOGITS (the training path never stores the
// probabilities) or the given probabilities.  grid (nblk, n) -> partial[n][nblk][8]
template <bool FROM_LOGITS>
__global__ void __launch_bounds__(256) mask_iou_partial_kernel(const float* __restrict__ src, int h, int w, int fy, int fx,
                                                               const float* __restrict__ y_true, float* __restrict__ partial) {
    __shared__ float red[256];
    const int ho = h * fy, wo = w * fx;
    const long long npix = (long long)ho * wo;
    const int img = blockIdx.y;
    const float ify = 1.f / (float)fy, ifx = 1.f / (float)fx;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const int oy = (int)(i / wo), ox = (int)(i - (long long)oy * wo);
        float4 pr;
        if (FROM_LOGITS) pr = softmax4(up_logits(src, img, h, w, oy, ox, ify, ifx));
        else pr = ld4(src + ((long long)img * npix + i) * 4);
        const float4 t = ld4(y_true + ((long long)img * npix + i) * 4);
        acc[0] = fmaf(t.x, pr.x, acc[0]); acc[1] = fmaf(t.y, pr.y, acc[1]); acc[2] = fmaf(t.z, pr.z, acc[2]); acc[3] = fmaf(t.w, pr.w, acc[3]);
        acc[4] += t.x + pr.x; acc[5] += t.y + pr.y; acc[6] += t.z + pr.z; acc[7] += t.w + pr.w;
    }
    for (int k = 0; k < 8; ++k) {
        __syncthreads();
        red[threadIdx.x] = acc[k];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[((long long)img * gridDim.x + blockIdx.x) * 8 + k] = red[0];
    }
}
__global__ void mask_iou_finish_kernel(const float* __restrict__ partial, int nblk, int n, float4 cw, float* __restrict__ out) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= n) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nblk; ++q)
        for (int k = 0; k < 8; ++k) s[k] += partial[((long long)img * nblk + q) * 8 + k];
    const float w[4] = {cw.x, cw.y, cw.z, cw.w};
    float m = 0.f;
    for (int c = 0; c < 4; ++c) m += s[c] / (s[4 + c] - s[c] + KEPS) * w[c];   // metrics.py:41-45
    out[img] = m;
}

// weighted "categorical accuracy" of the anchor labels (metrics.py:204-216): per class the number of anchors where
// one_hot(argmax p)[c] == y_true[c] (agreeing zeros count too), / #anchors, weighted sum.  One block per image, integer counts.
__global__ void __launch_bounds__(256) label_accuracy_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred, int a,
                                                             float4 cw, float* __restrict__ out) {
    __shared__ int red[4][256];
    const int img = blockIdx.x;
    int cnt[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < a; i += 256) {
        const float4 t = ld4(y_true + ((long long)img * a + i) * 4), p = ld4(y_pred + ((long long)img * a + i) * 4);
        int am = 0;   // first maximum, as tf.math.argmax
        float best = p.x;
        if (p.y > best) { best = p.y; am = 1; }
        if (p.z > best) { best = p.z; am = 2; }
        if (p.w > best) { best = p.w; am = 3; }
        cnt[0] += ((am == 0) ? 1.f : 0.f) == t.x;
        cnt[1] += ((am == 1) ? 1.f : 0.f) == t.y;
        cnt[2] += ((am == 2) ? 1.f : 0.f) == t.z;
        cnt[3] += ((am == 3) ? 1.f : 0.f) == t.w;
    }
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = cnt[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n = (float)a;
        out[img] = (((float)red[0][0] / n * cw.x + (float)red[1][0] / n * cw.y) + (float)red[2][0] / n * cw.z) + (float)red[3][0] / n * cw.w;
    }
}

// mean IoU of decoded predicted vs ground-truth boxes over the non-background anchors (metrics.py:76-171, including its
// conventions: widths clamped at 0, corners c -+ (w-1)/2, areas w*h, +1 in the intersection extents, epsilon in the
// denominator, 0/0 = NaN for an image without objects).  anchors: [a][4] = (cx, cy, w, h).  One block per image.
__global__ void __launch_bounds__(256) box_iou_kernel(const float* __restrict__ y_true, const float* __restrict__ y_pred,
                                                      const float* __restrict__ anchors, float4 sd, int a, float* __restrict__ out) {
    __shared__ float red[2][256];
    const int img = blockIdx.x;
    float s_iou = 0.f, s_nb = 0.f;
    for (int i = threadIdx.x; i < a; i += 256) {
        const float4 t = ld4(y_true + ((long long)img * a + i) * 4), p = ld4(y_pred + ((long long)img * a + i) * 4);
        const float4 an = ld4(anchors + (long long)i * 4);
        const float nb = (fabsf(t.x) + fabsf(t.y) + fabsf(t.z) + fabsf(t.w)) > 0.f ? 1.f : 0.f;
        float c[2][6];   // xmin, ymin, xmax, ymax, width, height of (pred, true)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 o = q == 0 ? p : t;
            const float cx = (o.x * sd.x * an.z + an.x) * nb, cy = (o.y * sd.y * an.w + an.y) * nb;
            const float wd = fmaxf(0.f, (expf(o.z * sd.z) - 1.f) * an.z) * nb, hg = fmaxf(0.f, (expf(o.w * sd.w) - 1.f) * an.w) * nb;
            c[q][0] = (cx - (wd - 1.f) / 2.f) * nb; c[q][1] = (cy - (hg - 1.f) / 2.f) * nb;
            c[q][2] = (cx + (wd - 1.f) / 2.f) * nb; c[q][3] = (cy + (hg - 1.f) / 2.f) * nb;
            c[q][4] = wd; c[q][5] = hg;
        }
        const float wi = fmaxf(0.f, fminf(c[1][2], c[0][2]) - fmaxf(c[1][0], c[0][0]) + 1.f) * nb;
        const float hi = fmaxf(0.f, fminf(c[1][3], c[0][3]) - fmaxf(c[1][1], c[0][1]) + 1.f) * nb;
        const float inter = wi * hi;
        s_iou += inter / (c[0][4] * c[0][5] + c[1][4] * c[1][5] - inter + KEPS);
        s_nb += nb;
    }
    red[0][threadIdx.x] = s_iou; red[1][threadIdx.x] = s_nb;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[img] = red[0][0] / red[1][0];
}

}  // namespace

extern "C" {

int ssdseg_gap_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int n, int hw, int c) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(out != nullptr, 3);
    SSDSEG_ARG(n > 0, 4);
    SSDSEG_ARG(hw > 0, 5);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 6);
    return pixel_sum(ctx, in->x, c, in->scale, in->shift, in->act, out, c, n, hw, c, 1.f / (float)hw, 0,
                     4.0 * ((double)n * hw * c + (double)n * c));
}

int ssdseg_gap_bwd(ssdseg_ctx* ctx, const float* g, float* dx, int n, int hw, int c, int accumulate) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(g != nullptr, 2);
    SSDSEG_ARG(dx != nullptr, 3);
    SSDSEG_ARG(n > 0 && hw > 0, 4);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 6);
    const long long total = (long long)n * hw * (c / 4);
    SSDSEG_LAUNCH(ctx, 4.0 * n * hw * c * (accumulate ? 2 : 1), 0.0, gap_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, g, dx, n, hw, c / 4,
                  accumulate);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

static int bilinear_fwd_impl(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, float* out, int ldo, int n, int h, int wdt, int c, int fy, int fx,
                             int pad) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldx >= c && ldx % 4 == 0, 3);
    SSDSEG_ARG(out != nullptr, 4);
    SSDSEG_ARG(ldo >= c && ldo % 4 == 0, 5);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 9);
    SSDSEG_ARG(fy >= 1 && fx >= 1, 10);
    const long long total = (long long)n * h * fy * wdt * fx * (c / 4);
    const char* bl = getenv("SSDSEG_BILINEAR");       // "gather": the general kernels (A/B runs, parity tests)
    if (fy == 4 && fx == 4 && !(bl != nullptr && !strcmp(bl, "gather"))) {
        const long long threads = (long long)n * h * wdt * (c / 4);
        SSDSEG_LAUNCH(ctx, 4.0 * ((double)n * h * wdt * c + 4.0 * total), 0.0, bilinear_fwd_x4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, in->x,
                      in->scale, in->shift, in->act, ldx, out, ldo, n, h, wdt, c / 4, pad);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    SSDSEG_LAUNCH(ctx, 4.0 * ((double)n * h * wdt * c + 4.0 * total), 0.0, bilinear_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, in->x,
                  in->scale, in->shift, in->act, ldx, out, ldo, n, h, wdt, c / 4, fy, fx, pad);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_bilinear_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, float* out, int ldo, int n, int h, int wdt, int c, int fy,
                        int fx) {
    return bilinear_fwd_impl(ctx, in, ldx, out, ldo, n, h, wdt, c, fy, fx, 0);
}

// the same values written into the INTERIOR of a bordered tensor out[n][h*fy + 2][w*fx + 2][ldo] (the border is left alone): the
// up-sampled ASPP output lands directly in the zero-bordered input copy the decoder's 3x3 conv kernels read (ssdseg_conv3x3_fwd_saved_from)
int ssdseg_bilinear_fwd_padded(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, float* out, int ldo, int n, int h, int wdt, int c, int fy,
                               int fx) {
    return bilinear_fwd_impl(ctx, in, ldx, out, ldo, n, h, wdt, c, fy, fx, 1);
}

int ssdseg_bilinear_bwd(ssdseg_ctx* ctx, const float* g, int ldg, float* dx, int ldx, int n, int h, int wdt, int c, int fy, int fx,
                        int accumulate) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(g != nullptr, 2);
    SSDSEG_ARG(ldg >= c && ldg % 4 == 0, 3);
    SSDSEG_ARG(dx != nullptr, 4);
    SSDSEG_ARG(ldx >= c && ldx % 4 == 0, 5);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 9);
    SSDSEG_ARG(fy >= 1 && fx >= 1, 10);
    if (h == 1 && wdt == 1)   // a 1x1 source feeds every output pixel with weight 1 (the ASPP pooling branch): a plain pixel sum
        return pixel_sum(ctx, g, ldg, nullptr, nullptr, SSDSEG_ACT_NONE, dx, ldx, n, fy * fx, c, 1.f, accumulate,
                         4.0 * ((double)n * c * (1 + fy * fx)));
    const char* bl = getenv("SSDSEG_BILINEAR");       // "gather": the general kernel (A/B runs, parity tests)
    if (fy == 4 && fx == 4 && !(bl != nullptr && !strcmp(bl, "gather"))) {
        const long long threads = (long long)n * ((h + 1) / 2) * ((wdt + 1) / 2) * (c / 4);
        SSDSEG_LAUNCH(ctx, 4.0 * ((double)n * h * wdt * c * (1 + fy * fx)), 0.0, bilinear_bwd_x4_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, g,
                      ldg, dx, ldx, n, h, wdt, c / 4, accumulate);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    const long long total = (long long)n * h * wdt * (c / 4);
    SSDSEG_LAUNCH(ctx, 4.0 * ((double)n * h * wdt * c * (1 + fy * fx)), 0.0, bilinear_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, g, ldg,
                  dx, ldx, n, h, wdt, c / 4, fy, fx, accumulate);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_mask_head_fwd(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx, const float* y_true,
                         const float* class_weights_host, float* prob, float* loss) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(logits != nullptr, 2);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 3);
    SSDSEG_ARG(c == 4, 6);   // the reference itself hard-codes depth 4 (layers.py:204, models.py:250-253)
    SSDSEG_ARG(fy >= 1 && fx >= 1, 7);
    SSDSEG_ARG((loss == nullptr) || (y_true != nullptr && class_weights_host != nullptr), 9);
    float cwh[4] = {0, 0, 0, 0};
    if (loss) memcpy(cwh, class_weights_host, sizeof(cwh));
    const long long npix = (long long)h * fy * wdt * fx;
    int nblk = (int)((npix + 256 * 8 - 1) / (256 * 8));
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    float* partial = nullptr;
    if (loss) {
        void* ws;
        int rc = ssdseg_workspace(ctx, (size_t)n * nblk * sizeof(float), &ws);
        if (rc) return rc;
        partial = (float*)ws;
    }
    SSDSEG_LAUNCH(ctx, 16.0 * n * npix * ((y_true ? 1 : 0) + (prob ? 1 : 0)), 0.0, mask_head_fwd_kernel, dim3(nblk, n), dim3(256), 0, logits, h,
                  wdt, fy, fx, loss ? y_true : nullptr, make_float4(cwh[0], cwh[1], cwh[2], cwh[3]), prob, partial);
    SSDSEG_LAUNCH_CHECK();
    if (loss) {
        SSDSEG_LAUNCH(ctx, 4.0 * n * nblk, 0.0, mask_loss_final_kernel, dim3(cdiv(n, 64)), dim3(64), 0, partial, nblk, loss, n);
        SSDSEG_LAUNCH_CHECK();
    }
    return 0;
}

static int mask_head_bwd_launch(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int fy, int fx, const float* y_true, const float* cwh,
                                float loss_scale, float* dlogits, int mode, const float* coef);

int ssdseg_mask_head_fwd_dice(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx, const float* y_true,
                              const float* class_weights_host, int squared, float* prob, float* loss, float* coef) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(logits != nullptr, 2);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 3);
    SSDSEG_ARG(c == 4, 6);
    SSDSEG_ARG(fy >= 1 && fx >= 1, 7);
    SSDSEG_ARG(y_true != nullptr, 9);
    SSDSEG_ARG(class_weights_host != nullptr, 10);
    SSDSEG_ARG(loss != nullptr || coef != nullptr, 13);
    float cwh[4];
    memcpy(cwh, class_weights_host, sizeof(cwh));
    const long long npix = (long long)h * fy * wdt * fx;
    int nblk = (int)((npix + 256 * 8 - 1) / (256 * 8));
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    void* ws;
    int rc = ssdseg_workspace(ctx, (size_t)n * nblk * 8 * sizeof(float), &ws);
    if (rc) return rc;
    SSDSEG_LAUNCH(ctx, 16.0 * n * npix * (1 + (prob ? 1 : 0)), 0.0, mask_head_fwd_dice_kernel, dim3(nblk, n), dim3(256), 0, logits, h, wdt, fy, fx,
                  y_true, squared ? 1 : 0, prob, (float*)ws);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 32.0 * n * nblk, 0.0, mask_dice_final_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (const float*)ws, nblk, n,
                  make_float4(cwh[0], cwh[1], cwh[2], cwh[3]), loss, coef);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_mask_head_bwd_dice(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx, const float* y_true,
                              const float* coef, int squared, float loss_scale, float* dlogits) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(logits != nullptr, 2);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 3);
    SSDSEG_ARG(c == 4, 6);
    SSDSEG_ARG(fy >= 1 && fx >= 1, 7);
    SSDSEG_ARG(y_true != nullptr, 9);
    SSDSEG_ARG(coef != nullptr, 10);
    SSDSEG_ARG(dlogits != nullptr, 13);
    const float zero[4] = {0.f, 0.f, 0.f, 0.f};
    return mask_head_bwd_launch(ctx, logits, n, h, wdt, fy, fx, y_true, zero, loss_scale, dlogits, squared ? 2 : 1, coef);
}

int ssdseg_mask_head_bwd(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int c, int fy, int fx, const float* y_true,
                         const float* class_weights_host, float loss_scale, float* dlogits) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(logits != nullptr, 2);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 3);
    SSDSEG_ARG(c == 4, 6);
    SSDSEG_ARG(fy >= 1 && fx >= 1, 7);
    SSDSEG_ARG(y_true != nullptr, 9);
    SSDSEG_ARG(class_weights_host != nullptr, 10);
    SSDSEG_ARG(dlogits != nullptr, 12);
    float cwh[4];
    memcpy(cwh, class_weights_host, sizeof(cwh));
    return mask_head_bwd_launch(ctx, logits, n, h, wdt, fy, fx, y_true, cwh, loss_scale, dlogits, 0, nullptr);
}

static int mask_head_bwd_launch(ssdseg_ctx* ctx, const float* logits, int n, int h, int wdt, int fy, int fx, const float* y_true, const float* cwh,
                                float loss_scale, float* dlogits, int mode, const float* coef) {
    const long long total = (long long)n * h * wdt;
    const char* mt = getenv("SSDSEG_MASK_BWD");       // "gather": the one-thread-per-pixel kernel (A/B runs, parity tests)
    if (fy == 4 && fx == 4 && !(mt != nullptr && !strcmp(mt, "gather"))) {
        constexpr int F = 4, TL = 16, R = TL * F + F;
        const long long blocks = (long long)n * cdiv(h, TL) * cdiv(wdt, TL);
        static bool configured = false;
        if (!configured) {
            SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_head_bwd_tile_kernel<F, TL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(R * R * sizeof(float4))));
            configured = true;
        }
        SSDSEG_LAUNCH(ctx, 16.0 * n * h * fy * wdt * fx, 0.0, (mask_head_bwd_tile_kernel<F, TL>), dim3((unsigned)blocks), dim3(TL * TL), R * R * sizeof(float4), logits, n,
                      h, wdt, y_true, make_float4(cwh[0], cwh[1], cwh[2], cwh[3]), loss_scale, dlogits, mode, coef);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    if (fy == 8 && fx == 8 && !(mt != nullptr && !strcmp(mt, "gather"))) {
        constexpr int F = 8, TL = 8, PARTS = 4, R = TL * F + F;
        constexpr size_t lds = (size_t)(R * R + PARTS * TL * TL) * sizeof(float4);
        const long long blocks = (long long)n * cdiv(h, TL) * cdiv(wdt, TL);
        static bool configured8 = false;
        if (!configured8) {
            SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_head_bwd_tile_split_kernel<F, TL, PARTS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured8 = true;
        }
        SSDSEG_LAUNCH(ctx, 16.0 * n * h * fy * wdt * fx, 0.0, (mask_head_bwd_tile_split_kernel<F, TL, PARTS>), dim3((unsigned)blocks), dim3(TL * TL * PARTS), lds, logits, n, h,
                      wdt, y_true, make_float4(cwh[0], cwh[1], cwh[2], cwh[3]), loss_scale, dlogits, mode, coef);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    SSDSEG_LAUNCH(ctx, 16.0 * n * h * fy * wdt * fx, 0.0, mask_head_bwd_kernel, dim3(ew_blocks(total, 256)), dim3(256), 0, logits, n, h, wdt,
                  fy, fx, y_true, make_float4(cwh[0], cwh[1], cwh[2], cwh[3]), loss_scale, dlogits, mode, coef);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_head_gather(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int b, int in_img_elems, int c, int out_off_elems,
                       int out_img_elems, int reverse) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(out != nullptr, 3);
    SSDSEG_ARG(b > 0, 4);
    SSDSEG_ARG(in_img_elems > 0 && in_img_elems % 4 == 0, 5);
    SSDSEG_ARG(c > 0 && c % 4 == 0 && in_img_elems % c == 0, 6);
    SSDSEG_ARG(out_off_elems >= 0 && out_off_elems % 4 == 0, 7);
    SSDSEG_ARG(out_img_elems >= out_off_elems + in_img_elems && out_img_elems % 4 == 0, 8);
    const long long total = (long long)b * (in_img_elems / 4);
    SSDSEG_LAUNCH(ctx, 8.0 * b * in_img_elems, 0.0, head_gather_kernel, dim3(ew_blocks(total)), dim3(256), 0, in->x, in->scale, in->shift,
                  in->act, out, b, in_img_elems, c, out_off_elems, out_img_elems, reverse);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_softmax_rows(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int rows, int c) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(out != nullptr, 3);
    SSDSEG_ARG(rows > 0, 4);
    SSDSEG_ARG(c == 4, 5);
    SSDSEG_LAUNCH(ctx, 32.0 * rows, 0.0, softmax_rows4_kernel, dim3(ew_blocks(rows)), dim3(256), 0, in->x, in->scale, in->shift, in->act, out,
                  (long long)rows);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}


int ssdseg_metric_mask_iou(ssdseg_ctx* ctx, const float* src, int n, int h, int wdt, int c, int fy, int fx, int from_logits,
                           const float* y_true, const float* class_weights_host, float* out) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(src != nullptr, 2);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 3);
    SSDSEG_ARG(c == 4, 6);
    SSDSEG_ARG(fy >= 1 && fx >= 1 && (from_logits || (fy == 1 && fx == 1)), 7);
    SSDSEG_ARG(y_true != nullptr, 10);
    SSDSEG_ARG(class_weights_host != nullptr, 11);
    SSDSEG_ARG(out != nullptr, 12);
    const long long npix = (long long)h * fy * wdt * fx;
    int nblk = (int)((npix + 256 * 8 - 1) / (256 * 8));
    if (nblk > 64) nblk = 64;
    if (nblk < 1) nblk = 1;
    void* ws;
    int rc = ssdseg_workspace(ctx, (size_t)n * nblk * 8 * sizeof(float), &ws);
    if (rc) return rc;
    const double bytes = 16.0 * n * npix * (from_logits ? 1.0 : 2.0);
    if (from_logits)
        SSDSEG_LAUNCH(ctx, bytes, 0.0, mask_iou_partial_kernel<true>, dim3(nblk, n), dim3(256), 0, src, h, wdt, fy, fx, y_true, (float*)ws);
    else
        SSDSEG_LAUNCH(ctx, bytes, 0.0, mask_iou_partial_kernel<false>, dim3(nblk, n), dim3(256), 0, src, h, wdt, fy, fx, y_true, (float*)ws);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 0.0, 0.0, mask_iou_finish_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (const float*)ws, nblk, n,
                  make_float4(class_weights_host[0], class_weights_host[1], class_weights_host[2], class_weights_host[3]), out);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_metric_label_accuracy(ssdseg_ctx* ctx, const float* y_true, const float* y_pred, int b, int a, int c,
                                 const float* class_weights_host, float* out) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(y_true != nullptr, 2);
    SSDSEG_ARG(y_pred != nullptr, 3);
    SSDSEG_ARG(b > 0 && a > 0, 4);
    SSDSEG_ARG(c == 4, 6);
    SSDSEG_ARG(class_weights_host != nullptr, 7);
    SSDSEG_ARG(out != nullptr, 8);
    SSDSEG_LAUNCH(ctx, 32.0 * b * a, 0.0, label_accuracy_kernel, dim3(b), dim3(256), 0, y_true, y_pred, a,
                  make_float4(class_weights_host[0], class_weights_host[1], class_weights_host[2], class_weights_host[3]), out);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_metric_box_iou(ssdseg_ctx* ctx, const float* y_true, const float* y_pred, const float* anchors_centroids, const float* stds4_host,
                          int b, int a, float* out) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(y_true != nullptr, 2);
    SSDSEG_ARG(y_pred != nullptr, 3);
    SSDSEG_ARG(anchors_centroids != nullptr, 4);
    SSDSEG_ARG(stds4_host != nullptr, 5);
    SSDSEG_ARG(b > 0 && a > 0, 6);
    SSDSEG_ARG(out != nullptr, 8);
    SSDSEG_LAUNCH(ctx, 32.0 * b * a, 0.0, box_iou_kernel, dim3(b), dim3(256), 0, y_true, y_pred, anchors_centroids,
                  make_float4(stds4_host[0], stds4_host[1], stds4_host[2], stds4_host[3]), a, out);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
