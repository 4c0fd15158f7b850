// ShuffleNetV2-only data-movement ops (reference models.py:480-505 channel shuffle, :573 split, :629 max-pool).
// All HBM-bound copies; the shuffle is an index permutation (a later round can fold it into the consumer's loads).
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace {

int ew_blocks(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 8192 ? (b < 1 ? 1 : b) : 8192);
}

struct PoolGeom {
    int n, h, w, cv, ho, wo, pt, pl;
};

__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)); }

// MaxPooling2D(3, strides=2, 'same'): padded cells never win
__global__ void maxpool_fwd_kernel(PoolGeom g, const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                   int act, float* __restrict__ out) {
    const long long total = (long long)g.n * g.ho * g.wo * g.cv;
    const bool aff = scale != nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % g.cv) * 4;
        long long r = i / g.cv;
        const int wo = (int)(r % g.wo); r /= g.wo;
        const int ho = (int)(r % g.ho);
        const long long img = r / g.ho;
        float4 s = f4(0.f), t = f4(0.f);
        if (aff) { s = ld4(scale + c0); t = ld4(shift + c0); }
        float4 m = f4(-INFINITY);
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 + kh - g.pt;
            if (hi < 0 || hi >= g.h) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 + kw - g.pl;
                if (wi < 0 || wi >= g.w) continue;
                m = max4(m, view_apply4(ld4(x + ((img * g.h + hi) * g.w + wi) * (long long)g.cv * 4 + c0), s, t, aff, act));
            }
        }
        st4(out + i * 4, m);
    }
}

// Backward in two passes (round 2; SSDSEG_MAXPOOL_BWD=scan keeps the one-pass kernel below for parity tests): every window's
// winner is found ONCE (first maximum of the row-major scan, strict '>': tap code kh * 3 + kw, one byte per channel), then every
// input pixel looks up the <= 4 windows that cover it -- 5 bytes per window and channel instead of nine float4 re-reads of the
// window (36 loads per input element in the one-pass form: 0.52 ms for ShuffleNetV2's 240 x 320 x 24 stem at batch 32).
__global__ void maxpool_argmax_kernel(PoolGeom g, const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                      int act, uchar4* __restrict__ code) {
    const long long total = (long long)g.n * g.ho * g.wo * g.cv;
    const bool aff = scale != nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % g.cv) * 4;
        long long r = i / g.cv;
        const int wo = (int)(r % g.wo); r /= g.wo;
        const int ho = (int)(r % g.ho);
        const long long img = r / g.ho;
        float4 s = f4(0.f), t = f4(0.f);
        if (aff) { s = ld4(scale + c0); t = ld4(shift + c0); }
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned arg[4] = {255u, 255u, 255u, 255u};
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 + kh - g.pt;
            if (hi < 0 || hi >= g.h) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 + kw - g.pl;
                if (wi < 0 || wi >= g.w) continue;
                const float4 v = view_apply4(ld4(x + ((img * g.h + hi) * g.w + wi) * (long long)g.cv * 4 + c0), s, t, aff, act);
                const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (ve[k] > best[k]) { best[k] = ve[k]; arg[k] = (unsigned)(kh * 3 + kw); }
            }
        }
        code[i] = make_uchar4((unsigned char)arg[0], (unsigned char)arg[1], (unsigned char)arg[2], (unsigned char)arg[3]);
    }
}

__global__ void maxpool_bwd_codes_kernel(PoolGeom g, const uchar4* __restrict__ code, const float* __restrict__ gout, float* __restrict__ dx) {
    const long long total = (long long)g.n * g.h * g.w * g.cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % g.cv);
        long long r = i / g.cv;
        const int wi0 = (int)(r % g.w); r /= g.w;
        const int hi0 = (int)(r % g.h);
        const long long img = r / g.h;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // the same window order as the one-pass kernel (ho outer, wo inner): the sums are bit-identical
        for (int ho = (hi0 + g.pt - 2 + 1) / 2; ho <= (hi0 + g.pt) / 2; ++ho) {
            const int kh = hi0 + g.pt - 2 * ho;
            if (ho < 0 || ho >= g.ho || kh < 0 || kh > 2) continue;
            for (int wo = (wi0 + g.pl - 2 + 1) / 2; wo <= (wi0 + g.pl) / 2; ++wo) {
                const int kw = wi0 + g.pl - 2 * wo;
                if (wo < 0 || wo >= g.wo || kw < 0 || kw > 2) continue;
                const long long o = ((img * g.ho + ho) * g.wo + wo) * (long long)g.cv + cq;
                const uchar4 cd = code[o];
                const float4 go = ld4(gout + o * 4);
                const unsigned mine = (unsigned)(kh * 3 + kw);
                if (cd.x == mine) acc[0] += go.x;
                if (cd.y == mine) acc[1] += go.y;
                if (cd.z == mine) acc[2] += go.z;
                if (cd.w == mine) acc[3] += go.w;
            }
        }
        st4(dx + i * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// gradient goes to the FIRST maximum of each window (row-major scan, strict '>'); gather form, deterministic
__global__ void maxpool_bwd_kernel(PoolGeom g, const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                   int act, const float* __restrict__ gout, float* __restrict__ dx) {
    const long long total = (long long)g.n * g.h * g.w * g.cv;
    const bool aff = scale != nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % g.cv) * 4;
        long long r = i / g.cv;
        const int wi0 = (int)(r % g.w); r /= g.w;
        const int hi0 = (int)(r % g.h);
        const long long img = r / g.h;
        float4 s = f4(0.f), t = f4(0.f);
        if (aff) { s = ld4(scale + c0); t = ld4(shift + c0); }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ho = (hi0 + g.pt - 2 + 1) / 2; ho <= (hi0 + g.pt) / 2; ++ho) {
            if (ho < 0 || ho >= g.ho || hi0 + g.pt - 2 * ho < 0 || hi0 + g.pt - 2 * ho > 2) continue;
            for (int wo = (wi0 + g.pl - 2 + 1) / 2; wo <= (wi0 + g.pl) / 2; ++wo) {
                if (wo < 0 || wo >= g.wo || wi0 + g.pl - 2 * wo < 0 || wi0 + g.pl - 2 * wo > 2) continue;
                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int arg[4] = {-1, -1, -1, -1};
                for (int kh = 0; kh < 3; ++kh) {
                    const int hi = ho * 2 + kh - g.pt;
                    if (hi < 0 || hi >= g.h) continue;
                    for (int kw = 0; kw < 3; ++kw) {
                        const int wi = wo * 2 + kw - g.pl;
                        if (wi < 0 || wi >= g.w) continue;
                        const float4 v = view_apply4(ld4(x + ((img * g.h + hi) * g.w + wi) * (long long)g.cv * 4 + c0), s, t, aff, act);
                        const float ve[4] = {v.x, v.y, v.z, v.w};
                        for (int k = 0; k < 4; ++k)
                            if (ve[k] > best[k]) { best[k] = ve[k]; arg[k] = hi * g.w + wi; }
                    }
                }
                const float4 go = ld4(gout + ((img * g.ho + ho) * g.wo + wo) * (long long)g.cv * 4 + c0);
                const float ge[4] = {go.x, go.y, go.z, go.w};
                for (int k = 0; k < 4; ++k)
                    if (arg[k] == hi0 * g.w + wi0) acc[k] += ge[k];
            }
        }
        st4(dx + i * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// out[m][j*groups + i] = in[m][i*(c/groups) + j]   (Reshape -> Permute(1,2,4,3) -> Reshape); inverse swaps the roles
__global__ void shuffle_kernel(const float* __restrict__ in, const float* __restrict__ scale, const float* __restrict__ shift, int act,
                               int ldi, float* __restrict__ out, int ldo, long long m, int c, int groups, int inverse) {
    const long long total = m * c;
    const int per = c / groups;
    const float lo = act_lo(act), hi = act_hi(act);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c;
        const int k = (int)(i % c);                 // output channel
        int src;
        if (!inverse) src = (k % groups) * per + k / groups;
        else src = (k % per) * groups + k / per;
        float v = in[r * ldi + src];
        if (scale != nullptr) v = fmaf(scale[src], v, shift[src]);   // the (lazy) BatchNorm + activation of the shuffled concat
        out[r * ldo + k] = fminf(fmaxf(v, lo), hi);
    }
}

__global__ void act_bwd_kernel(float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx, long long m, int cv, int act) {
    const long long total = m * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cv;
        const int c0 = (int)(i % cv) * 4;
        float4 gv = ld4(g + r * ldg + c0);
        const float4 xv = ld4(x + r * ldx + c0);
        gv.x *= act_mask(xv.x, act); gv.y *= act_mask(xv.y, act); gv.z *= act_mask(xv.z, act); gv.w *= act_mask(xv.w, act);
        st4(g + r * ldg + c0, gv);
    }
}

void pool_geom(int n, int h, int w, int c, PoolGeom* g) {
    g->n = n; g->h = h; g->w = w; g->cv = c / 4;
    same_pad(h, 3, 2, 1, &g->ho, &g->pt);
    same_pad(w, 3, 2, 1, &g->wo, &g->pl);
}


// out[m][j] = table[j] >= 0 ? act(s * in[m][table[j]] + t) : 0   (+ previous contents).  The general channel re-indexing
// behind the ShuffleNetV2 sizes whose branches are not multiples of 4 channels wide ('1x': 58, '2x': 122): inside a unit the
// branch tensors are zero-padded to the next multiple of 4, and split / shuffle / their gradients become table lookups
// between the packed (116-wide) and the padded (60 + 60) layouts.  One thread per output element, scalar accesses.
__global__ void channel_gather_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                      int ldi, float* __restrict__ out, int ldo, long long m, int c_out, const int* __restrict__ table,
                                      int accumulate) {
    const long long total = m * c_out;
    const bool aff = scale != nullptr;
    const float lo = act_lo(act), hi = act_hi(act);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % c_out);
        const long long row = i / c_out;
        const int src = table[j];
        float v = 0.f;
        if (src >= 0) {
            v = x[row * ldi + src];
            v = fminf(fmaxf(aff ? fmaf(scale[src], v, shift[src]) : v, lo), hi);
        }
        float* o = out + row * ldo + j;
        *o = accumulate ? *o + v : v;
    }
}

// dst[r][c] = src[r][c] for a rows x cols block of two row-major matrices with different leading dimensions (parameter
// packing / unpacking between the exact Keras shapes of the flat bucket and the zero-padded shapes the kernels use)
__global__ void copy2d_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int rows, int cols) {
    const int total = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        dst[(long long)r * ldd + c] = src[(long long)r * lds + c];
    }
}

// the same for a whole table of blocks in ONE launch: blockIdx.y = table row {dst, ldd, src, lds, rows, cols} (six 64-bit words),
// blockIdx.x strides over its elements
__global__ void __launch_bounds__(256) copy2d_batch_kernel(const long long* __restrict__ table) {
    const long long* e = table + 6 * (long long)blockIdx.y;
    float* __restrict__ dst = reinterpret_cast<float*>(e[0]);
    const float* __restrict__ src = reinterpret_cast<const float*>(e[2]);
    const int ldd = (int)e[1], lds = (int)e[3], rows = (int)e[4], cols = (int)e[5];
    const int total = rows * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / cols, c = i - r * cols;
        dst[(long long)r * ldd + c] = src[(long long)r * lds + c];
    }
}

}  // namespace

extern "C" {

int ssdseg_maxpool3x3s2_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, float* out, int n, int h, int wdt, int c) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(out != nullptr, 3);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 4);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 7);
    PoolGeom g;
    pool_geom(n, h, wdt, c, &g);
    const long long total = (long long)n * g.ho * g.wo * g.cv;
    SSDSEG_LAUNCH(ctx, 4.0 * ((double)n * h * wdt * c + 4.0 * total), 0.0, maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, g, in->x,
                  in->scale, in->shift, in->act, out);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_maxpool3x3s2_bwd(ssdseg_ctx* ctx, const ssdseg_view* in, const float* gout, float* dx, int n, int h, int wdt, int c) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(gout != nullptr, 3);
    SSDSEG_ARG(dx != nullptr, 4);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 5);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 8);
    PoolGeom g;
    pool_geom(n, h, wdt, c, &g);
    const long long total = (long long)n * h * wdt * g.cv;
    const char* e = getenv("SSDSEG_MAXPOOL_BWD");
    if (!(e != nullptr && !strcmp(e, "scan"))) {
        const long long ototal = (long long)n * g.ho * g.wo * g.cv;
        void* ws;
        int rc = ssdseg_workspace(ctx, (size_t)ototal * sizeof(uchar4), &ws);
        if (rc) return rc;
        SSDSEG_LAUNCH(ctx, 4.0 * n * h * wdt * c + 1.0 * n * g.ho * g.wo * c, 0.0, maxpool_argmax_kernel, dim3(ew_blocks(ototal)), dim3(256), 0, g, in->x,
                      in->scale, in->shift, in->act, (uchar4*)ws);
        SSDSEG_LAUNCH_CHECK();
        SSDSEG_LAUNCH(ctx, 4.0 * n * h * wdt * c + 5.0 * n * g.ho * g.wo * c, 0.0, maxpool_bwd_codes_kernel, dim3(ew_blocks(total)), dim3(256), 0, g,
                      (const uchar4*)ws, gout, dx);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    SSDSEG_LAUNCH(ctx, 4.0 * (2.0 * n * h * wdt * c + (double)n * g.ho * g.wo * c), 0.0, maxpool_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0,
                  g, in->x, in->scale, in->shift, in->act, gout, dx);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_channel_shuffle(ssdseg_ctx* ctx, const ssdseg_view* in, int ldi, float* out, int ldo, int m, int c, int groups,
                           int inverse) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldi >= c, 3);
    SSDSEG_ARG(out != nullptr && out != in->x, 4);
    SSDSEG_ARG(ldo >= c, 5);
    SSDSEG_ARG(m > 0, 6);
    SSDSEG_ARG(c > 0, 7);
    SSDSEG_ARG(groups > 0 && c % groups == 0, 8);
    const long long total = (long long)m * c;
    SSDSEG_LAUNCH(ctx, 8.0 * total, 0.0, shuffle_kernel, dim3(ew_blocks(total)), dim3(256), 0, in->x, in->scale, in->shift, in->act, ldi, out,
                  ldo, (long long)m, c, groups, inverse);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_act_bwd(ssdseg_ctx* ctx, float* g, int ldg, const float* x, int ldx, int m, int c, int act) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(g != nullptr, 2);
    SSDSEG_ARG(ldg >= c && ldg % 4 == 0, 3);
    SSDSEG_ARG(x != nullptr, 4);
    SSDSEG_ARG(ldx >= c && ldx % 4 == 0, 5);
    SSDSEG_ARG(m > 0, 6);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 7);
    const long long total = (long long)m * (c / 4);
    SSDSEG_LAUNCH(ctx, 12.0 * m * c, 0.0, act_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, g, ldg, x, ldx, (long long)m, c / 4, act);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}


int ssdseg_channel_gather(ssdseg_ctx* ctx, const ssdseg_view* in, int ldi, float* out, int ldo, long long m, int c_out,
                          const int32_t* table, int accumulate) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldi > 0, 3);
    SSDSEG_ARG(out != nullptr && out != in->x, 4);
    SSDSEG_ARG(ldo >= c_out, 5);
    SSDSEG_ARG(m > 0, 6);
    SSDSEG_ARG(c_out > 0, 7);
    SSDSEG_ARG(table != nullptr, 8);
    const long long total = m * c_out;
    SSDSEG_LAUNCH(ctx, 8.0 * total, 0.0, channel_gather_kernel, dim3(ew_blocks(total)), dim3(256), 0, in->x, in->scale, in->shift, in->act, ldi,
                  out, ldo, m, c_out, (const int*)table, accumulate);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_copy2d(ssdseg_ctx* ctx, float* dst, int ldd, const float* src, int lds, int rows, int cols) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(dst != nullptr, 2);
    SSDSEG_ARG(src != nullptr, 4);
    SSDSEG_ARG(rows > 0 && cols > 0 && ldd >= cols && lds >= cols, 6);
    SSDSEG_LAUNCH(ctx, 8.0 * rows * cols, 0.0, copy2d_kernel, dim3(ew_blocks((long long)rows * cols)), dim3(256), 0, dst, ldd, src, lds, rows, cols);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_copy2d_batch(ssdseg_ctx* ctx, const long long* table, int ncopies, int max_elems, long long total_floats) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(table != nullptr, 2);
    SSDSEG_ARG(ncopies > 0, 3);
    SSDSEG_ARG(max_elems > 0, 4);
    const int gx = (max_elems + 255) / 256 < 16 ? (max_elems + 255) / 256 : 16;
    SSDSEG_LAUNCH(ctx, 8.0 * (double)total_floats, 0.0, copy2d_batch_kernel, dim3(gx, ncopies, 1), dim3(256), 0, table);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
