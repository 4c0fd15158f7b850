// Input gradient of the NARROW dense 3x3 conv (the decoder's 256 -> 4 logits conv, reference blocks.py:127) with the BatchNorm
// backward sums of the layer in front fused (ssdseg_conv3x3_bwd_data_bn), DIRECT form for cin == 256, cout == 4.
//
// dx[q][c] = sum_tap sum_o dY[q - d(tap)][o] * W[tap][c][o]: 36 multiply-adds per value from a 4-channel tensor that fits the L2
// (9.8 MB at batch 32) -- a streaming layer: read the raw input of the BatchNorm in front (for its backward sums), write dx, 2 KB per
// pixel.  As a tap-expanded GEMM (gemm.hip: dz = shift(dY) materialised, K = 36, N = 256, transposing epilogue that also reads the raw
// input) it ran 0.53 + 0.03 ms = 2.3 TB/s of that traffic; this kernel: 0.365 ms = 3.5 TB/s.  Here a wave is one pixel's 64 channel quads: the nine dY vectors of a pixel are
// wave-uniform (scalar loads), the 36 x 4 weights of a lane stay in registers, a lane forms its four dx values, adds them into the
// BatchNorm sums (sum mask dx, sum mask dx xhat: the same expressions as the GEMM epilogue's) and stores 16 bytes -- a wave stores 1 KB.
#include "common.h"

// bn.hip
int ssdseg_bn_bwd_finalize_launch(ssdseg_ctx* ctx, const float* part, int nparts, int c, double count, const float* scale,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* k1, float* k0);

namespace {

// dymat[i] = the gradient view applied to pixel i's four channels (cout == 4: one float4 per pixel)
__global__ void __launch_bounds__(256) conv3n_dymat_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ gs,
                                                           const float* __restrict__ gt, const float* __restrict__ gk1, const float* __restrict__ gk0,
                                                           int act, float* __restrict__ out, int m) {
    const float4 s = ld4(gs), t = ld4(gt), k1 = ld4(gk1), k0 = ld4(gk0);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256)
        st4(out + (size_t)i * 4, gview_apply4(ld4(g + (size_t)i * 4), ld4(y + (size_t)i * 4), s, t, k1, k0, act));
}

struct C3NArgs {
    const float* dymat;  // [n][h][w][4]
    const float* w;      // [3][3][256][4]
    const float* x;      // raw input of the BatchNorm in front, [m][ldx]
    const float *xs, *xt, *mean, *istd;
    int xact;
    float* dx;           // [m][ldx]
    int ldx;
    float* part;         // [blocks][2][256]
    int n, h, w_;
};

constexpr int C3N_CIN = 256;

__global__ void __launch_bounds__(256) conv3n_bwd_bn_direct_kernel(C3NArgs p) {
    __shared__ float4 red[2][4][64];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c0 = 4 * lane;
    // wq[tap][o] = W[tap][c0 .. c0+3][o]
    float4 wq[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float4 r[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) r[cc] = ld4(p.w + ((size_t)tap * C3N_CIN + c0 + cc) * 4);
        wq[tap][0] = make_float4(r[0].x, r[1].x, r[2].x, r[3].x);
        wq[tap][1] = make_float4(r[0].y, r[1].y, r[2].y, r[3].y);
        wq[tap][2] = make_float4(r[0].z, r[1].z, r[2].z, r[3].z);
        wq[tap][3] = make_float4(r[0].w, r[1].w, r[2].w, r[3].w);
    }
    const float4 bs = ld4(p.xs + c0), bt = ld4(p.xt + c0), bm = ld4(p.mean + c0), bi = ld4(p.istd + c0);
    const float lo = act_lo(p.xact), hi = act_hi(p.xact);
    float4 esb = f4(0.f), esg = f4(0.f);
    // dY through the CONSTANT address space: the indices are wave-uniform (readfirstlane), so these become scalar loads into SGPRs
    // (another launch wrote the tensor; nothing in this kernel does) -- as vector loads the 2 x 9 float4 sat in 72 VGPRs
    typedef float c3n_f4 __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(4))) c3n_f4* dm = (const __attribute__((address_space(4))) c3n_f4*)(p.dymat);
    const int rows = p.n * p.h;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int img = row / p.h, yy = row - img * p.h;
        constexpr int PX = 4;                               // pixels per wave and trip: x0, x0 + 4, ... (their raw-input loads go out first; 2: 425 us, 5 / 8: no better than 4)
        for (int x0 = wave; x0 < p.w_; x0 += 4 * PX) {
            float4 xv[PX];
            bool pok[PX];
#pragma unroll
            for (int u = 0; u < PX; ++u) {
                pok[u] = x0 + 4 * u < p.w_;
                const size_t m = (size_t)row * p.w_ + (pok[u] ? x0 + 4 * u : x0);
                xv[u] = ld4(p.x + m * p.ldx + c0);
            }
#pragma unroll
            for (int u = 0; u < PX; ++u) {
                if (!pok[u]) break;
                const int xx = x0 + 4 * u;
                float4 acc = f4(0.f);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int sy = yy - (tap / 3 - 1), sx = xx - (tap % 3 - 1);      // wave-uniform
                    const bool ok = sy >= 0 && sy < p.h && sx >= 0 && sx < p.w_;
                    const int src = __builtin_amdgcn_readfirstlane(ok ? (img * p.h + sy) * p.w_ + sx : 0);
                    const c3n_f4 dl = dm[src];
                    const float keep = ok ? 1.f : 0.f;
                    const float4 d = make_float4(dl[0] * keep, dl[1] * keep, dl[2] * keep, dl[3] * keep);
                    acc.x = fmaf(d.x, wq[tap][0].x, acc.x); acc.y = fmaf(d.x, wq[tap][0].y, acc.y); acc.z = fmaf(d.x, wq[tap][0].z, acc.z); acc.w = fmaf(d.x, wq[tap][0].w, acc.w);
                    acc.x = fmaf(d.y, wq[tap][1].x, acc.x); acc.y = fmaf(d.y, wq[tap][1].y, acc.y); acc.z = fmaf(d.y, wq[tap][1].z, acc.z); acc.w = fmaf(d.y, wq[tap][1].w, acc.w);
                    acc.x = fmaf(d.z, wq[tap][2].x, acc.x); acc.y = fmaf(d.z, wq[tap][2].y, acc.y); acc.z = fmaf(d.z, wq[tap][2].z, acc.z); acc.w = fmaf(d.z, wq[tap][2].w, acc.w);
                    acc.x = fmaf(d.w, wq[tap][3].x, acc.x); acc.y = fmaf(d.w, wq[tap][3].y, acc.y); acc.z = fmaf(d.w, wq[tap][3].z, acc.z); acc.w = fmaf(d.w, wq[tap][3].w, acc.w);
                }
                const size_t m = (size_t)row * p.w_ + xx;
                st4(p.dx + m * p.ldx + c0, acc);
                const float4 x4 = xv[u];
                float4 mg;
                mg.x = (fmaf(bs.x, x4.x, bt.x) > lo && fmaf(bs.x, x4.x, bt.x) < hi) ? acc.x : 0.f;
                mg.y = (fmaf(bs.y, x4.y, bt.y) > lo && fmaf(bs.y, x4.y, bt.y) < hi) ? acc.y : 0.f;
                mg.z = (fmaf(bs.z, x4.z, bt.z) > lo && fmaf(bs.z, x4.z, bt.z) < hi) ? acc.z : 0.f;
                mg.w = (fmaf(bs.w, x4.w, bt.w) > lo && fmaf(bs.w, x4.w, bt.w) < hi) ? acc.w : 0.f;
                esb.x += mg.x; esb.y += mg.y; esb.z += mg.z; esb.w += mg.w;
                esg.x = fmaf(mg.x, (x4.x - bm.x) * bi.x, esg.x); esg.y = fmaf(mg.y, (x4.y - bm.y) * bi.y, esg.y);
                esg.z = fmaf(mg.z, (x4.z - bm.z) * bi.z, esg.z); esg.w = fmaf(mg.w, (x4.w - bm.w) * bi.w, esg.w);
            }
        }
    }
    red[0][wave][lane] = esb;
    red[1][wave][lane] = esg;
    __syncthreads();
    if (t < 128) {
        const int which = t >> 6;
        float4 a = f4(0.f);
        for (int k = 0; k < 4; ++k) {      // fixed order
            const float4 v = red[which][k][lane];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        st4(p.part + ((size_t)blockIdx.x * 2 + which) * C3N_CIN + c0, a);
    }
}

}  // namespace

extern "C" {

// (internal helpers of ssdseg_conv3x3_bwd_data_bn in gemm.hip, which declares them inside its extern "C" block)
bool ssdseg_conv3n_direct_takes(int cin, int cout, int ldx) {
    const char* e = getenv("SSDSEG_CONV3N_DIRECT");      // "0": the tap-expanded GEMM (A/B runs, a parity-test case)
    if (e != nullptr && e[0] == '0') return false;
    return cin == C3N_CIN && cout == 4 && ldx % 4 == 0;
}

int ssdseg_conv3n_bwd_bn_direct(ssdseg_ctx* ctx, const ssdseg_view* in, const ssdseg_gview* dy, const float* w, float* dx, int ldx, int n, int h,
                                int wdt, const float* in_mean, const float* in_invstd, float* in_dgamma, float* in_dbeta, float* in_k1,
                                float* in_k0) {
    const long long m = (long long)n * h * wdt;
    const int rows = n * h;
    const int blocks = rows < 2048 ? rows : 2048;
    void* ws;
    const size_t dyb = ((size_t)m * 4 * sizeof(float) + 255) & ~(size_t)255, pb = (size_t)blocks * 2 * C3N_CIN * sizeof(float);
    int rc = ssdseg_workspace(ctx, dyb + pb, &ws);
    if (rc) return rc;
    const float* dymat = dy->g;
    if (dy->scale != nullptr) {
        const int gb = (int)((m + 255) / 256 < 4096 ? (m + 255) / 256 : 4096);
        SSDSEG_LAUNCH(ctx, 48.0 * m, 0.0, conv3n_dymat_kernel, dim3(gb), dim3(256), 0, dy->g, dy->y, dy->scale, dy->shift, dy->k1, dy->k0, dy->act, (float*)ws, (int)m);
        SSDSEG_LAUNCH_CHECK();
        dymat = (const float*)ws;
    }
    C3NArgs a{};
    a.dymat = dymat; a.w = w; a.x = in->x; a.xs = in->scale; a.xt = in->shift; a.mean = in_mean; a.istd = in_invstd; a.xact = in->act;
    a.dx = dx; a.ldx = ldx; a.part = (float*)((char*)ws + dyb); a.n = n; a.h = h; a.w_ = wdt;
    // algorithmic (dense 3x3 backward-data): read dY, write dX, read W; the raw BatchNorm input is the fused reduction's
    // operand (view bytes, as for the GEMM epilogue)
    const double bytes = 4.0 * ((double)m * 4 + (double)m * C3N_CIN + 9.0 * C3N_CIN * 4), flops = 2.0 * m * 36 * C3N_CIN;
    ctx->timing_view_bytes = 4.0 * (double)m * C3N_CIN;
    SSDSEG_LAUNCH(ctx, bytes, flops, conv3n_bwd_bn_direct_kernel, dim3(blocks), dim3(256), 0, a);
    SSDSEG_LAUNCH_CHECK();
    return ssdseg_bn_bwd_finalize_launch(ctx, a.part, blocks, C3N_CIN, (double)m, in->scale, in_mean, in_invstd, in_dgamma, in_dbeta, in_k1, in_k0);
}

}  // extern "C"
