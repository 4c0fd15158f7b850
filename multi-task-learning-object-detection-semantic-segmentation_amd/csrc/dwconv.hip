// Depthwise 3x3 convolution, NHWC fp32, TF SAME padding, stride 1|2, dilation >= 1 (stride 1 only when dilated).
// Replaces DepthwiseConv2D / the depthwise half of SeparableConv2D (reference models.py:88,236,242;
// blocks.py:33,38,43,122,152) forward and backward.
//
// HBM-bound (0.9-2.25 FLOP/B).  Mapping: one thread owns ONE 4-channel vector (16 B) for its whole life, so the
// 9 filter taps, the producer's BN scale/shift and the BN-stat / dW accumulators live in registers; consecutive
// lanes hold consecutive channel vectors, i.e. a wave reads whole contiguous NHWC pixels (C*4 bytes each).
// Each thread walks a grid-strided list of 1x4 output strips and slides a register window along W
// (18 loads per 4 outputs at stride 1 instead of 36); vertical reuse is left to L1/L2.
// BatchNorm(train) is fused on both sides: the producer's normalise+ReLU6 is applied on load (ssdseg_view),
// and this layer's per-channel (sum, sumsq) leave as one partial row per block (no atomics, deterministic).
#include "common.h"

namespace {

constexpr int TW = 4;           // outputs per strip
constexpr int MAX_BLOCKS = 1024;  // also the number of BN-stat / dW partial rows

struct DwGeom {
    int n, h, w, c, ho, wo, s, d, pt, pl;
    int cv;       // c / 4
    int wtiles;   // ceil(wo / TW)
    long long ntiles;
};

struct ViewDev {
    const float* x;
    const float* scale;
    const float* shift;
    int act;
};
struct GViewDev {
    const float* g;
    const float* y;
    const float* scale;
    const float* shift;
    const float* k1;
    const float* k0;
    int act;
};

struct ChanCoef {  // per-thread channel-vector constants (identity views: s = 1, t = k1 = k0 = 0)
    float4 s, t, k1, k0;
    float lo, hi;  // activation clamp of an input view
    bool affine;
};

__device__ __forceinline__ float4 load_view(const ViewDev& v, const ChanCoef& cc, long long off) {
    return view_affine4(ld4(v.x + off), cc.s, cc.t, cc.lo, cc.hi);
}
__device__ __forceinline__ float4 load_gview(const GViewDev& v, const ChanCoef& cc, long long off) {
    return gview_apply4(ld4(v.g + off), ld4(v.y + off), cc.s, cc.t, cc.k1, cc.k0, v.act);
}
// predicated forms: the load is unconditional (address clamped to the thread's own channel vector of pixel 0) and the result
// selected, so the compiler can issue a whole window of loads back to back instead of branch + wait per element
__device__ __forceinline__ float4 sel4(bool ok, float4 v) { return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 load_view_if(const ViewDev& v, const ChanCoef& cc, long long off, bool ok, int c0) {
    return sel4(ok, load_view(v, cc, ok ? off : (long long)c0));
}
__device__ __forceinline__ float4 load_gview_if(const GViewDev& v, const ChanCoef& cc, long long off, bool ok, int c0) {
    return sel4(ok, load_gview(v, cc, ok ? off : (long long)c0));
}
__device__ __forceinline__ void fma4(float4& acc, float4 a, float4 b) {
    acc.x = fmaf(a.x, b.x, acc.x); acc.y = fmaf(a.y, b.y, acc.y); acc.z = fmaf(a.z, b.z, acc.z); acc.w = fmaf(a.w, b.w, acc.w);
}
__device__ __forceinline__ void add4(float4& acc, float4 a) { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }

// block-level reduction over threadIdx.y of one float4 per thread; result valid for threadIdx.y == 0
__device__ __forceinline__ float4 reduce_over_y(float4 v, float4* red) {
    __syncthreads();
    red[threadIdx.y * blockDim.x + threadIdx.x] = v;
    __syncthreads();
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.y == 0) {
        for (int y = 0; y < (int)blockDim.y; ++y) add4(r, red[y * blockDim.x + threadIdx.x]);
    }
    return r;
}

// ------------------------------------------------------------------------------------------------ forward
// S: stride (1|2); DIL: 1 = dense taps with sliding window, 0 = runtime dilation (stride 1), direct gathers.
template <int S, int DIL>
__global__ void __launch_bounds__(512) dw_fwd_kernel(DwGeom gm, ViewDev in, const float* __restrict__ wgt,
                                                     float* __restrict__ y, float* __restrict__ stats) {
    extern __shared__ float4 red[];
    const BlockPos bpos = xcd_block_pos();   // (tile slot, channel group), neighbouring slots on the same XCD / L2
    const int cvi = bpos.y * blockDim.x + threadIdx.x;
    const bool active = cvi < gm.cv;
    const int c0 = cvi * 4;
    float4 wk[9];
    ChanCoef cc;
    cc.affine = in.scale != nullptr;
    cc.s = f4(1.f);
    cc.t = cc.k1 = cc.k0 = f4(0.f);
    cc.lo = act_lo(in.act);
    cc.hi = act_hi(in.act);
    if (active) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wk[t] = ld4(wgt + (long long)t * gm.c + c0);
        if (cc.affine) { cc.s = ld4(in.scale + c0); cc.t = ld4(in.shift + c0); }
    }
    float4 ssum = f4(0.f), ssq = f4(0.f);
    if (active) {
        for (long long tile = (long long)bpos.x * blockDim.y + threadIdx.y; tile < gm.ntiles;
             tile += (long long)gridDim.x * blockDim.y) {
            const int wt = (int)(tile % gm.wtiles);
            const long long r = tile / gm.wtiles;
            const int ho = (int)(r % gm.ho);
            const int n = (int)(r / gm.ho);
            const int wo0 = wt * TW;
            float4 out[TW];
#pragma unroll
            for (int j = 0; j < TW; ++j) out[j] = f4(0.f);
            const long long img = (long long)n * gm.h * gm.w;
            if (DIL == 1) {
                constexpr int WC = (TW - 1) * S + 3;
                const int wi0 = wo0 * S - gm.pl;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hi = ho * S + kh - gm.pt;
                    const bool rowok = hi >= 0 && hi < gm.h;
                    float4 row[WC];
#pragma unroll
                    for (int ci = 0; ci < WC; ++ci) {
                        const int wi = wi0 + ci;
                        row[ci] = load_view_if(in, cc, ((img + (long long)hi * gm.w + wi) * gm.c) + c0, rowok && wi >= 0 && wi < gm.w, c0);
                    }
#pragma unroll
                    for (int j = 0; j < TW; ++j)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) fma4(out[j], row[j * S + kw], wk[kh * 3 + kw]);
                }
            } else {
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hi = ho + kh * gm.d - gm.pt;
                    const bool rowok = hi >= 0 && hi < gm.h;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                        for (int j = 0; j < TW; ++j) {
                            const int wi = wo0 + j + kw * gm.d - gm.pl;
                            fma4(out[j], load_view_if(in, cc, ((img + (long long)hi * gm.w + wi) * gm.c) + c0,
                                                      rowok && wi >= 0 && wi < gm.w && wo0 + j < gm.wo, c0), wk[kh * 3 + kw]);
                        }
                    }
                }
            }
            const long long obase = (((long long)n * gm.ho + ho) * gm.wo + wo0) * gm.c + c0;
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                if (wo0 + j < gm.wo) {
                    st4(y + obase + (long long)j * gm.c, out[j]);
                    add4(ssum, out[j]);
                    fma4(ssq, out[j], out[j]);
                }
            }
        }
    }
    if (stats != nullptr) {
        float4 a = reduce_over_y(ssum, red);
        float4 b = reduce_over_y(ssq, red);
        if (threadIdx.y == 0 && active) {
            float* row = stats + (long long)bpos.x * 2 * gm.c;
            st4(row + c0, a);
            st4(row + gm.c + c0, b);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// One pass produces dx (gradient w.r.t. the activated input) and a per-block partial of dW.
// Each thread owns the output strip (n, ho, wo0..wo0+3) for dW and the input patch rows [ho*S, ho*S+S) x cols
// [wo0*S, wo0*S + 4*S) for dx; the 3x6 window of dy around the strip serves both.
template <int S, int DIL, int PT, int PL>
__global__ void __launch_bounds__(512) dw_bwd_kernel(DwGeom gm, ViewDev in, const float* __restrict__ wgt, GViewDev dy,
                                                     float* __restrict__ dx, float* __restrict__ dwpart, int accumulate) {
    extern __shared__ float4 red[];
    const BlockPos bpos = xcd_block_pos();   // (tile slot, channel group), neighbouring slots on the same XCD / L2
    const int cvi = bpos.y * blockDim.x + threadIdx.x;
    const bool active = cvi < gm.cv;
    const int c0 = cvi * 4;
    ChanCoef ci, co;  // input-side view coefficients, output-side gradient-view coefficients
    ci.affine = in.scale != nullptr;
    co.affine = dy.scale != nullptr;
    ci.s = co.s = f4(1.f);
    ci.t = ci.k1 = ci.k0 = co.t = co.k1 = co.k0 = f4(0.f);
    ci.lo = act_lo(in.act); ci.hi = act_hi(in.act);
    co.lo = co.hi = 0.f;
    if (!co.affine) { dy.y = dy.g; dy.act = SSDSEG_ACT_NONE; }   // identity gradient view, branch-free form
    if (active) {
        if (ci.affine) { ci.s = ld4(in.scale + c0); ci.t = ld4(in.shift + c0); }
        if (co.affine) { co.s = ld4(dy.scale + c0); co.t = ld4(dy.shift + c0); co.k1 = ld4(dy.k1 + c0); co.k0 = ld4(dy.k0 + c0); }
    }
    float4 dwacc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) dwacc[t] = f4(0.f);

    if (active) {
        for (long long tile = (long long)bpos.x * blockDim.y + threadIdx.y; tile < gm.ntiles;
             tile += (long long)gridDim.x * blockDim.y) {
            const int wt = (int)(tile % gm.wtiles);
            const long long r = tile / gm.wtiles;
            const int ho = (int)(r % gm.ho);
            const int n = (int)(r / gm.ho);
            const int wo0 = wt * TW;
            const long long oimg = (long long)n * gm.ho * gm.wo;
            const long long iimg = (long long)n * gm.h * gm.w;

            if (DIL == 1) {
                // ---- dy window rows ho-1..ho+1, cols wo0-1..wo0+TW
                float4 dyw[3][TW + 2];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int hh = ho - 1 + a;
#pragma unroll
                    for (int b = 0; b < TW + 2; ++b) {
                        const int ww = wo0 - 1 + b;
                        dyw[a][b] = load_gview_if(dy, co, (oimg + (long long)hh * gm.wo + ww) * gm.c + c0,
                                                  hh >= 0 && hh < gm.ho && ww >= 0 && ww < gm.wo, c0);
                    }
                }
                // ---- dx over the owned input patch
                if (dx != nullptr) {
                    float4 wk[9];
#pragma unroll
                    for (int t = 0; t < 9; ++t) wk[t] = ld4(wgt + (long long)t * gm.c + c0);
#pragma unroll
                    for (int ir = 0; ir < S; ++ir) {
                        const int hi = ho * S + ir;
                        if (hi >= gm.h) continue;
#pragma unroll
                        for (int ic = 0; ic < TW * S; ++ic) {
                            const int wi = wo0 * S + ic;
                            if (wi >= gm.w) continue;
                            float4 acc = f4(0.f);
#pragma unroll
                            for (int kh = 0; kh < 3; ++kh) {
                                // ho' = (hi + PT - kh) / S must be integral: (ir + PT - kh) % S == 0
                                if (((ir + PT - kh) % S + S) % S != 0) continue;
                                const int ra = ((ir + PT - kh) - (((ir + PT - kh) % S + S) % S)) / S + 1;  // window row, compile-time
#pragma unroll
                                for (int kw = 0; kw < 3; ++kw) {
                                    if (((ic + PL - kw) % S + S) % S != 0) continue;
                                    const int cb = ((ic + PL - kw) - (((ic + PL - kw) % S + S) % S)) / S + 1;  // window col
                                    if (ra >= 0 && ra < 3 && cb >= 0 && cb < TW + 2) fma4(acc, dyw[ra][cb], wk[kh * 3 + kw]);
                                }
                            }
                            float* p = dx + (iimg + (long long)hi * gm.w + wi) * gm.c + c0;
                            if (accumulate) add4(acc, ld4(p));
                            st4(p, acc);
                        }
                    }
                }
                // ---- dW: a-window row by row against the centre row of dy
                constexpr int WC = (TW - 1) * S + 3;
                const int wi0 = wo0 * S - PL;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hi = ho * S + kh - PT;
                    const bool rowok = hi >= 0 && hi < gm.h;
                    float4 row[WC];
#pragma unroll
                    for (int q = 0; q < WC; ++q) {
                        const int wi = wi0 + q;
                        row[q] = load_view_if(in, ci, (iimg + (long long)hi * gm.w + wi) * gm.c + c0, rowok && wi >= 0 && wi < gm.w, c0);
                    }
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int j = 0; j < TW; ++j) fma4(dwacc[kh * 3 + kw], row[j * S + kw], dyw[1][j + 1]);
                }
            } else {
                // dilated, stride 1, pads == dilation: direct gathers
                const int d = gm.d;
                float4 wk[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wk[t] = ld4(wgt + (long long)t * gm.c + c0);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int wq = wo0 + j;
                    if (wq >= gm.wo) continue;
                    const float4 dyc = load_gview(dy, co, (oimg + (long long)ho * gm.wo + wq) * gm.c + c0);
                    float4 acc = f4(0.f);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int hh = ho + gm.pt - kh * d;   // output row feeding dx at input row ho
                        const int hi = ho + kh * d - gm.pt;   // input row feeding dW from output row ho
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const int ww = wq + gm.pl - kw * d;
                            const int wi = wq + kw * d - gm.pl;
                            if (dx != nullptr && hh >= 0 && hh < gm.ho && ww >= 0 && ww < gm.wo)
                                fma4(acc, load_gview(dy, co, (oimg + (long long)hh * gm.wo + ww) * gm.c + c0), wk[kh * 3 + kw]);
                            if (hi >= 0 && hi < gm.h && wi >= 0 && wi < gm.w)
                                fma4(dwacc[kh * 3 + kw], load_view(in, ci, (iimg + (long long)hi * gm.w + wi) * gm.c + c0), dyc);
                        }
                    }
                    if (dx != nullptr) {
                        float* p = dx + (iimg + (long long)ho * gm.w + wq) * gm.c + c0;
                        if (accumulate) add4(acc, ld4(p));
                        st4(p, acc);
                    }
                }
            }
        }
    }
    // ---- block partial of dW: [gridDim.x][9][c]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float4 v = reduce_over_y(dwacc[t], red);
        if (threadIdx.y == 0 && active) st4(dwpart + ((long long)bpos.x * 9 + t) * gm.c + c0, v);
    }
}

struct DwLaunch {
    dim3 grid, block;
    size_t lds;
};

bool dw_geometry(int n, int h, int w, int c, int stride, int dilation, DwGeom* g, DwLaunch* l) {
    g->n = n; g->h = h; g->w = w; g->c = c; g->s = stride; g->d = dilation;
    same_pad(h, 3, stride, dilation, &g->ho, &g->pt);
    same_pad(w, 3, stride, dilation, &g->wo, &g->pl);
    g->cv = c / 4;
    g->wtiles = cdiv(g->wo, TW);
    g->ntiles = (long long)n * g->ho * g->wtiles;
    int bx = g->cv < 256 ? g->cv : 256;
    int by = 512 / bx;
    if (by > 64) by = 64;
    if (by < 1) by = 1;
    long long want = (g->ntiles + by - 1) / by;
    int gx = (int)(want < MAX_BLOCKS ? want : MAX_BLOCKS);
    if (gx < 1) gx = 1;
    gx = (gx + 7) & ~7;   // multiple of 8 for the XCD remap (surplus blocks find no tile and write zero partial rows)
    l->block = dim3(bx, by, 1);
    l->grid = dim3(gx, cdiv(g->cv, bx), 1);
    l->lds = (size_t)bx * by * sizeof(float4);
    return true;
}

}  // namespace

#include "dwconv_lds.h"
#include "dwconv_march.h"

#include <stdlib.h>

int ssdseg_colsum(ssdseg_ctx* ctx, const float* part, int nparts, long long len, float* out);  // bn.hip
// bn.hip: (dgamma, dbeta, k1, k0) from nparts partial rows of (sum mask*g, sum mask*g*xhat)
int ssdseg_bn_bwd_finalize_launch(ssdseg_ctx* ctx, const float* part, int nparts, int c, double count, const float* scale,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* k1, float* k0);

namespace {

// SSDSEG_DW_BWD=march|lds|reg forces one backward kernel family (A/B measurements); default: measured best per shape
// (read on every call, not cached: the parity tests flip it between calls)
int dw_bwd_choice() {
    const char* e = getenv("SSDSEG_DW_BWD");
    return !e ? 0 : (!strcmp(e, "march") ? 1 : (!strcmp(e, "lds") ? 2 : (!strcmp(e, "reg") ? 3 : 0)));
}

// forward kernel family: column-marching for every dense-tap conv whose tensors fit 32-bit byte offsets
// (SSDSEG_DW_FWD=lds keeps the LDS-tiled kernels for A/B measurements)
bool dw_fwd_use_march(int n, int h, int w, int c, int dilation) {
    const char* e = getenv("SSDSEG_DW_FWD");
    const bool lds = e != nullptr && !strcmp(e, "lds");
    const char* atr = getenv("SSDSEG_DW_ATROUS");   // "gather": direct-gather kernels for the dilated convs
    const bool atrous_ok = !(atr != nullptr && !strcmp(atr, "gather"));
    return !lds && (dilation == 1 || atrous_ok) && (long long)n * h * w * c < (1LL << 30);
}

struct BnFuse {   // BatchNorm-backward reduction of the layer feeding this depthwise conv, fused into its backward
    const float* mean;
    const float* invstd;
    float *dgamma, *dbeta, *k1, *k0;
};

int dw_bwd_impl(ssdseg_ctx* ctx, const ssdseg_view* in, const float* w, const ssdseg_gview* dy, float* dx, float* dw, int n, int h,
                int wdt, int c, int stride, int dilation, int accumulate, const BnFuse* bn, bool* bn_done) {
    DwGeom g;
    DwLaunch l;
    dw_geometry(n, h, wdt, c, stride, dilation, &g, &l);
    ViewDev v{in->x, in->scale, in->shift, in->act};
    GViewDev gv{dy->g, dy->y, dy->scale, dy->shift, dy->k1, dy->k0, dy->act};
    // algorithmic traffic, SURVEY.md 8(d): 4 * (2*X + Y + 18*C) -- read X, read dY, write dX, read W, write dW.  A BatchNorm-backward
    // gradient view is formed from TWO tensors (g and the raw forward output y): the second one is what this design reads on top
    // of 8(d)'s ideal and is reported separately (`view_bytes`), never inside the roofline's algorithmic bytes.
    const double cost_bytes = 4.0 * (2.0 * n * h * wdt * c + (double)n * g.ho * g.wo * c + 18.0 * c);
    ctx->timing_view_bytes = dy->scale != nullptr ? 4.0 * n * g.ho * g.wo * c : 0.0;
    const double cost_flops = 36.0 * n * g.ho * g.wo * c;
    const int choice = dw_bwd_choice();
    // atrous (stride 1, SAME: pad == dilation): dilation^2 interleaved dense convs through the same marching kernel
    // (SSDSEG_DW_ATROUS=gather keeps the direct-gather kernel: nine two-tensor gathers per output pixel, 0.5 TB/s)
    const char* atr = getenv("SSDSEG_DW_ATROUS");
    const bool atrous_march = dilation > 1 && !(atr != nullptr && !strcmp(atr, "gather")) && g.pt == dilation && g.pl == dilation &&
                              g.ho == h && g.wo == wdt;
    const bool march_ok = (dilation == 1 || atrous_march) && stride == 1 && (long long)n * h * wdt * c < (1LL << 30);   // 32-bit byte offsets
    if (bn_done) *bn_done = false;
    if (march_ok && (choice == 0 || choice == 1)) {
        MarchGeom mg;
        const MarchLaunch ml = march_geometry(n, h, wdt, c, &mg, dilation);
        const int nparts = (int)ml.grid.x;
        const bool fuse = bn != nullptr && dilation == 1 && dx != nullptr;   // (with accumulate: this conv is the LAST writer of dx)
        void* ws;
        int rc = ssdseg_partials(ctx, (size_t)nparts * 11 * c * sizeof(float), &ws);
        if (rc) return rc;
        float* part = (float*)ws;
        float* bnpart = part + (size_t)nparts * 9 * c;
        const bool wfull = wdt % MTW == 0 && dilation == 1;
#define DW_BWD_MARCH(BN_, WF_, AC_, DIL_)                                                                                                        \
    SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_bwd_march_kernel<BN_, WF_, AC_, DIL_>), ml.grid, ml.block, ml.lds, mg, v, w, gv, dx, part, accumulate, \
                  fuse ? bn->mean : (const float*)nullptr, fuse ? bn->invstd : (const float*)nullptr, fuse ? bnpart : (float*)nullptr)
        if (dilation > 1 && accumulate) DW_BWD_MARCH(false, false, true, true);
        else if (dilation > 1) DW_BWD_MARCH(false, false, false, true);
        else if (fuse && wfull && accumulate) DW_BWD_MARCH(true, true, true, false);
        else if (fuse && accumulate) DW_BWD_MARCH(true, false, true, false);
        else if (fuse && wfull) DW_BWD_MARCH(true, true, false, false);
        else if (fuse) DW_BWD_MARCH(true, false, false, false);
        else if (wfull && accumulate) DW_BWD_MARCH(false, true, true, false);
        else if (wfull) DW_BWD_MARCH(false, true, false, false);
        else if (accumulate) DW_BWD_MARCH(false, false, true, false);
        else DW_BWD_MARCH(false, false, false, false);
#undef DW_BWD_MARCH
        SSDSEG_LAUNCH_CHECK();
        rc = ssdseg_colsum(ctx, part, nparts, 9LL * c, dw);
        if (rc || !fuse) return rc;
        *bn_done = true;
        return ssdseg_bn_bwd_finalize_launch(ctx, bnpart, nparts, c, (double)n * h * wdt, in->scale, bn->mean, bn->invstd, bn->dgamma,
                                             bn->dbeta, bn->k1, bn->k0);
    }
    const bool march2_ok = dilation == 1 && stride == 2 && (long long)n * h * wdt * c < (1LL << 30);
    if (march2_ok && (choice == 0 || choice == 1)) {
        March2Geom mg;
        const MarchLaunch ml = march2_geometry(n, h, wdt, c, g.ho, g.wo, &mg);
        const int nparts = (int)ml.grid.x;
        const bool fuse = bn != nullptr && dx != nullptr;
        void* ws;
        int rc = ssdseg_partials(ctx, (size_t)nparts * 11 * c * sizeof(float), &ws);
        if (rc) return rc;
        float* part = (float*)ws;
        float* bnpart = part + (size_t)nparts * 9 * c;
#define DW_BWD_MARCH2(BN_, PT_, PL_, AC_)                                                                                                     \
    SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_bwd_march2_kernel<BN_, PT_, PL_, AC_>), ml.grid, ml.block, ml.lds, mg, v, w, gv, dx, part, \
                  accumulate, fuse ? bn->mean : (const float*)nullptr, fuse ? bn->invstd : (const float*)nullptr,                          \
                  fuse ? bnpart : (float*)nullptr)
#define DW_BWD_MARCH2_P(BN_, AC_)                             \
    do {                                                      \
        if (g.pt == 0 && g.pl == 0) DW_BWD_MARCH2(BN_, 0, 0, AC_); \
        else if (g.pt == 0) DW_BWD_MARCH2(BN_, 0, 1, AC_);    \
        else if (g.pl == 0) DW_BWD_MARCH2(BN_, 1, 0, AC_);    \
        else DW_BWD_MARCH2(BN_, 1, 1, AC_);                   \
    } while (0)
        if (fuse && accumulate) DW_BWD_MARCH2_P(true, true);
        else if (fuse) DW_BWD_MARCH2_P(true, false);
        else if (accumulate && dx != nullptr) DW_BWD_MARCH2_P(false, true);
        else DW_BWD_MARCH2_P(false, false);
#undef DW_BWD_MARCH2_P
#undef DW_BWD_MARCH2
        SSDSEG_LAUNCH_CHECK();
        rc = ssdseg_colsum(ctx, part, nparts, 9LL * c, dw);
        if (rc || !fuse) return rc;
        *bn_done = true;
        return ssdseg_bn_bwd_finalize_launch(ctx, bnpart, nparts, c, (double)n * h * wdt, in->scale, bn->mean, bn->invstd, bn->dgamma,
                                             bn->dbeta, bn->k1, bn->k0);
    }
    const LdsLaunch ll = stride == 1 ? lds_launch<1>(g) : lds_launch<2>(g);
    // measured on MI355X (profiles/): the fused LDS backward wins for stride 1 with few channel groups (big early layers,
    // decoder); with many channel groups or stride 2 its 256-VGPR footprint loses to the register-window kernel
    const bool use_lds = dilation == 1 && stride == 1 && (choice == 2 || (choice != 3 && c <= 160));
    const int nparts = use_lds ? (int)ll.grid.x : (int)l.grid.x;
    void* ws;
    size_t part_bytes = (size_t)nparts * 9 * c * sizeof(float);
    int rc = ssdseg_partials(ctx, part_bytes, &ws);
    if (rc) return rc;
    float* part = (float*)ws;
#define DW_BWD_LDS(S_, PT_, PL_) \
    SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_bwd_lds_kernel<S_, PT_, PL_>), ll.grid, dim3(256), ll.lds_bwd, g, v, w, gv, dx, part, accumulate)
#define DW_BWD_REG(S_, D_, PT_, PL_) \
    SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_bwd_kernel<S_, D_, PT_, PL_>), l.grid, l.block, l.lds, g, v, w, gv, dx, part, accumulate)
    if (dilation != 1) DW_BWD_REG(1, 0, 0, 0);
    else if (use_lds) DW_BWD_LDS(1, 1, 1);
    else if (stride == 1) DW_BWD_REG(1, 1, 1, 1);
    else if (g.pt == 0 && g.pl == 0) DW_BWD_REG(2, 1, 0, 0);
    else if (g.pt == 0 && g.pl == 1) DW_BWD_REG(2, 1, 0, 1);
    else if (g.pt == 1 && g.pl == 0) DW_BWD_REG(2, 1, 1, 0);
    else DW_BWD_REG(2, 1, 1, 1);
#undef DW_BWD_REG
#undef DW_BWD_LDS
    SSDSEG_LAUNCH_CHECK();
    return ssdseg_colsum(ctx, part, nparts, 9LL * c, dw);
}

}  // namespace

extern "C" {

int ssdseg_dwconv_parts(int n, int h, int w, int c, int stride, int dilation, int* nparts_host) {
    SSDSEG_ARG(n > 0 && h > 0 && w > 0, 1);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 4);
    SSDSEG_ARG(stride == 1 || stride == 2, 5);
    SSDSEG_ARG(dilation >= 1 && (dilation == 1 || stride == 1), 6);
    SSDSEG_ARG(nparts_host != nullptr, 7);
    DwGeom g;
    DwLaunch l;
    dw_geometry(n, h, w, c, stride, dilation, &g, &l);
    if (dw_fwd_use_march(n, h, w, c, dilation)) {
        March2Geom mg;
        *nparts_host = (int)march_fwd_geometry(n, h, w, c, g.ho, g.wo, stride, &mg, dilation).grid.x;   // column-marching kernels
    } else if (dilation == 1) {
        *nparts_host = (int)(stride == 1 ? lds_launch<1>(g) : lds_launch<2>(g)).grid.x;                  // LDS-tiled kernels
    } else {
        *nparts_host = (int)l.grid.x;                                                                   // gather kernels
    }
    return 0;
}

int ssdseg_dwconv_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, const float* w, float* y, int n, int h, int wdt, int c,
                      int stride, int dilation, float* stats) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr, 2);
    SSDSEG_ARG(w != nullptr, 3);
    SSDSEG_ARG(y != nullptr, 4);
    SSDSEG_ARG(n > 0, 5);
    SSDSEG_ARG(h > 0, 6);
    SSDSEG_ARG(wdt > 0, 7);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 8);
    SSDSEG_ARG(stride == 1 || stride == 2, 9);
    SSDSEG_ARG(dilation >= 1 && (dilation == 1 || stride == 1), 10);
    SSDSEG_ARG((in->scale == nullptr) == (in->shift == nullptr), 2);
    DwGeom g;
    DwLaunch l;
    dw_geometry(n, h, wdt, c, stride, dilation, &g, &l);
    ViewDev v{in->x, in->scale, in->shift, in->act};
    // algorithmic traffic: read X, write Y, read W
    const double cost_bytes = 4.0 * ((double)n * h * wdt * c + (double)n * g.ho * g.wo * c + 9.0 * c);
    const double cost_flops = 18.0 * n * g.ho * g.wo * c;
    if (dw_fwd_use_march(n, h, wdt, c, dilation)) {
        March2Geom mg;
        const MarchLaunch ml = march_fwd_geometry(n, h, wdt, c, g.ho, g.wo, stride, &mg, dilation);
        { const char* e = getenv("SSDSEG_DW_FWD_DEPTH"); mg.depth2 = !(e != nullptr && e[0] == '1'); }
#define DW_FWD_MARCH(S_, PT_, PL_, DIL_) \
    SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_fwd_march_kernel<S_, PT_, PL_, DIL_>), ml.grid, ml.block, ml.lds, mg, v, w, y, stats)
        if (dilation > 1) DW_FWD_MARCH(1, 1, 1, true);
        else if (stride == 1 && mg.depth2) SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_fwd_march_kernel<1, 1, 1, false, true>), ml.grid, ml.block, ml.lds, mg, v, w, y, stats);
        else if (stride == 1) DW_FWD_MARCH(1, 1, 1, false);
        else if (g.pt == 0 && g.pl == 0) DW_FWD_MARCH(2, 0, 0, false);
        else if (g.pt == 0) DW_FWD_MARCH(2, 0, 1, false);
        else if (g.pl == 0) DW_FWD_MARCH(2, 1, 0, false);
        else DW_FWD_MARCH(2, 1, 1, false);
#undef DW_FWD_MARCH
    } else if (dilation == 1 && stride == 1) {
        const LdsLaunch ll = lds_launch<1>(g);
        SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_fwd_lds_kernel<1>), ll.grid, dim3(256), ll.lds_fwd, g, v, w, y, stats);
    } else if (dilation == 1) {
        const LdsLaunch ll = lds_launch<2>(g);
        SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_fwd_lds_kernel<2>), ll.grid, dim3(256), ll.lds_fwd, g, v, w, y, stats);
    } else {
        SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (dw_fwd_kernel<1, 0>), l.grid, l.block, l.lds, g, v, w, y, stats);
    }
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_dwconv_bwd(ssdseg_ctx* ctx, const ssdseg_view* in, const float* w, const ssdseg_gview* dy, float* dx,
                      float* dw, int n, int h, int wdt, int c, int stride, int dilation, int accumulate) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr, 2);
    SSDSEG_ARG(w != nullptr, 3);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 4);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 4);
    SSDSEG_ARG(dw != nullptr, 6);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 7);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 10);
    SSDSEG_ARG(stride == 1 || stride == 2, 11);
    SSDSEG_ARG(dilation >= 1 && (dilation == 1 || stride == 1), 12);
    return dw_bwd_impl(ctx, in, w, dy, dx, dw, n, h, wdt, c, stride, dilation, accumulate, nullptr, nullptr);
}

int ssdseg_dwconv_bwd_bn(ssdseg_ctx* ctx, const ssdseg_view* in, const float* w, const ssdseg_gview* dy, float* dx, float* dw, int n,
                         int h, int wdt, int c, int stride, int dilation, int accumulate, const float* in_mean, const float* in_invstd,
                         float* in_dgamma, float* in_dbeta, float* in_k1, float* in_k0) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && in->scale != nullptr && in->shift != nullptr, 2);
    SSDSEG_ARG(w != nullptr, 3);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 4);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 4);
    SSDSEG_ARG(dx != nullptr, 5);
    SSDSEG_ARG(dw != nullptr, 6);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 7);
    SSDSEG_ARG(c > 0 && c % 4 == 0, 10);
    SSDSEG_ARG(stride == 1 || stride == 2, 11);
    SSDSEG_ARG(dilation >= 1 && (dilation == 1 || stride == 1), 12);
    SSDSEG_ARG(in_mean != nullptr && in_invstd != nullptr, 14);
    SSDSEG_ARG(in_k1 != nullptr && in_k0 != nullptr, 18);
    const BnFuse bn{in_mean, in_invstd, in_dgamma, in_dbeta, in_k1, in_k0};
    bool done = false;
    int rc = dw_bwd_impl(ctx, in, w, dy, dx, dw, n, h, wdt, c, stride, dilation, accumulate, &bn, &done);
    if (rc || done) return rc;
    // shapes without a fused kernel: the same reduction as a separate pass over (dx, x)
    return ssdseg_bn_bwd_reduce(ctx, dx, c, in->x, c, n * h * wdt, c, in->scale, in->shift, in_mean, in_invstd, in->act, in_dgamma, in_dbeta,
                                in_k1, in_k0);
}

}  // extern "C"
