// Weight gradient of the dense 3x3 stride-1 SAME convolution (DeepLabV3+ decoder, reference blocks.py:117,127), all nine taps in
// ONE pass -- included by gemm.hip inside its anonymous namespace.
//
//   dW[tap][k][n] = sum_m a[m + off(tap)][k] * dy[m][n]        a = act(s*x + t) (zero outside the image), dy = gview(g, y)
//
// The first version ran nine shifted weight-gradient GEMMs, i.e. streamed the 1.3 GB (g, y) pair and the 0.75 GB input nine
// times (19 ms at 45 TFLOP/s, bound by HBM / L2).  Here a block of NINE waves owns one 32(k) x 32*WN(n) tile of every tap:
// wave t is tap t.  Per step the block stages 32 consecutive pixels of one image row -- dy[32][32*WN] and the activated input
// rows h-1, h, h+1 x 34 pixels x 32 k -- in LDS once; the nine waves read the SAME dy fragments (B operand) and their own
// shifted view of the input tile (A operand), so the gradient pair is read once per k tile instead of nine times.
// v_mfma_f32_32x32x2_f32 as everywhere else; register-staged prefetch of the next step; deterministic (fixed split of the
// pixel steps, partial slabs [split][9][K][N] folded by colsum).
#pragma once

constexpr int C9_PX = 32;              // pixels per step
constexpr int C9_KT = 32;              // input channels per block
constexpr int C9_XS = C9_KT + 4;       // LDS pixel stride of the input tile (floats, 16-byte aligned rows)
constexpr int C9_XW = C9_PX + 2;       // input pixels per row of the tile (one halo pixel each side)
constexpr int C9_THREADS = 576;        // nine waves

struct Conv9Args {
    const float* x;   // input view [n][h][w][ldx]
    const float* xs;
    const float* xt;
    int xact, ldx;
    const float* g;   // gradient view [n][h][w][N]
    const float* y;
    const float* gs;
    const float* gt;
    const float* gk1;
    const float* gk0;
    int gact;
    float* part;      // [splits][9][K][N]
    int n, h, w, K, N;
    int wchunks;          // ceil(w / 32)
    long long steps;      // n * h * wchunks
    long long steps_per_split;
};

template <int WN>
__global__ void __launch_bounds__(C9_THREADS) conv3_wgrad9_kernel(Conv9Args p) {
    constexpr int NT = 32 * WN;            // output channels per block
    constexpr int DS = NT + 4;             // LDS row stride of the dy tile
    constexpr int DV = C9_PX * (NT / 4);   // float4 per dy tile
    constexpr int DQ = (DV + C9_THREADS - 1) / C9_THREADS;
    constexpr int XV = 3 * C9_XW * (C9_KT / 4);   // float4 per input tile (816)
    constexpr int XQ = (XV + C9_THREADS - 1) / C9_THREADS;
    extern __shared__ float smem[];
    float* dyT = smem;                     // [32][DS]
    float* xT = smem + C9_PX * DS;         // [3][34][C9_XS]
    const int t = threadIdx.x;
    const int tap = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;
    const int dh = tap / 3 - 1, dw = tap % 3 - 1;
    const int n0 = blockIdx.x * NT, k0 = blockIdx.y * C9_KT;
    const long long s0 = (long long)blockIdx.z * p.steps_per_split;
    long long s1 = s0 + p.steps_per_split;
    if (s1 > p.steps) s1 = p.steps;

    const bool gaff = p.gs != nullptr;
    const float* yp = gaff ? p.y : p.g;                    // identity gradient view: y aliases g, act NONE
    const int gact = gaff ? p.gact : SSDSEG_ACT_NONE;
    const float xlo = act_lo(p.xact), xhi = act_hi(p.xact);

    // per-thread staging slots: the float4 column of every slot is fixed (576 % (NT/4) == 0 for NT in {32, 128}; 576 % 8 == 0)
    const int dn4 = t % (NT / 4);
    const bool dnok = n0 + dn4 * 4 < p.N;
    float4 cgs = f4(1.f), cgt = f4(0.f), cgk1 = f4(0.f), cgk0 = f4(0.f);
    if (gaff && dnok) { cgs = ld4(p.gs + n0 + dn4 * 4); cgt = ld4(p.gt + n0 + dn4 * 4); cgk1 = ld4(p.gk1 + n0 + dn4 * 4); cgk0 = ld4(p.gk0 + n0 + dn4 * 4); }
    const int xk4 = t & 7;
    const bool xkok = k0 + xk4 * 4 < p.K;
    float4 cxs = f4(1.f), cxt = f4(0.f);
    if (p.xs != nullptr && xkok) { cxs = ld4(p.xs + k0 + xk4 * 4); cxt = ld4(p.xt + k0 + xk4 * 4); }

    float4 sg[DQ], sy[DQ], sx[XQ];
    unsigned okd = 0, okx = 0;
    auto issue = [&](long long s) {
        const int wc = (int)(s % p.wchunks);
        const long long row = s / p.wchunks;               // img * h + hrow
        const int hrow = (int)(row % p.h);
        const long long img = row / p.h;
        const int w0 = wc * C9_PX;
        okd = okx = 0;
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            const int idx = t + C9_THREADS * q;
            const int px = idx / (NT / 4);
            const bool ok = idx < DV && dnok && w0 + px < p.w;
            const long long off = ok ? (row * p.w + w0 + px) * (long long)p.N + n0 + dn4 * 4 : 0;
            sg[q] = ld4(p.g + off);
            sy[q] = ld4(yp + off);
            okd |= (ok ? 1u : 0u) << q;
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + C9_THREADS * q;
            const int pxr = idx >> 3, r = pxr / C9_XW, c = pxr - r * C9_XW;
            const int hi = hrow + r - 1, wi = w0 + c - 1;
            const bool ok = idx < XV && xkok && hi >= 0 && hi < p.h && wi >= 0 && wi < p.w;
            const long long off = ok ? ((img * p.h + hi) * p.w + wi) * (long long)p.ldx + k0 + xk4 * 4 : 0;
            sx[q] = ld4(p.x + off);
            okx |= (ok ? 1u : 0u) << q;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            const int idx = t + C9_THREADS * q;
            if (idx < DV) {
                const int px = idx / (NT / 4);
                const float4 v = gview_apply4(sg[q], sy[q], cgs, cgt, cgk1, cgk0, gact);
                st4(dyT + px * DS + dn4 * 4, ((okd >> q) & 1u) ? v : f4(0.f));
            }
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + C9_THREADS * q;
            if (idx < XV) {
                const int pxr = idx >> 3;
                const float4 v = view_affine4(sx[q], cxs, cxt, xlo, xhi);
                st4(xT + pxr * C9_XS + xk4 * 4, ((okx >> q) & 1u) ? v : f4(0.f));
            }
        }
    };

    f32x16 acc[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    if (s0 < s1) issue(s0);
    // A operand of tap (dh, dw): a[hrow + dh][w0 + px + dw][k0 + li] = xT[dh + 1][px + dw + 1][li], px = 2*ps + hh
    const float* xa = xT + ((dh + 1) * C9_XW + dw + 1 + hh) * C9_XS + li;
    const float* yb = dyT + hh * DS + li;
    for (long long s = s0; s < s1; ++s) {
        __syncthreads();            // every wave is done with the previous tile
        commit();
        __syncthreads();
        issue(s + 1 < s1 ? s + 1 : s);   // unconditional (a dummy re-issue on the last step keeps the s_waitcnt placement exact)
        constexpr int PF = 2, NST = C9_PX / 2;
        float afr[PF + 1], bfr[PF + 1][WN];
        auto fetch = [&](int ps, int buf) {
            afr[buf] = xa[(2 * ps) * C9_XS];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) bfr[buf][nt] = yb[(2 * ps) * DS + nt * 32];
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) fetch(q, q);
#pragma unroll
        for (int ps = 0; ps < NST; ++ps) {
            if (ps + PF < NST) fetch(ps + PF, (ps + PF) % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) acc[nt] = mfma32(afr[ps % (PF + 1)], bfr[ps % (PF + 1)][nt], acc[nt]);
        }
    }

    // this block's partial: part[split][tap][k][n]; C/D layout col = lane&31 (n), row = (e&3) + 8*(e>>2) + 4*hh (k)
    float* out = p.part + ((long long)blockIdx.z * 9 + tap) * p.K * p.N;
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        const int n = n0 + nt * 32 + li;
        if (n < p.N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (k < p.K) out[(long long)k * p.N + n] = acc[nt][e];
            }
        }
    }
}

// Twelve-wave variant for 128-column tiles.  Nine waves over four SIMDs sit 3/2/2/2, so the SIMD with three tap-waves sets the
// pace and the matrix pipes are capped at 75 %.  Here the 9 taps x 4 column tiles = 36 (tap, tile) units are dealt three to a
// wave, twelve waves = three per SIMD: same staging, same LDS tiles, every unit still one 32x32 accumulator, but the MFMA
// work is even across the SIMDs (and a wave carries 48 accumulator registers instead of 64).
constexpr int C12_THREADS = 768;
__global__ void __launch_bounds__(C12_THREADS) conv3_wgrad12_kernel(Conv9Args p) {
    constexpr int NT = 128, DS = NT + 4;
    constexpr int DV = C9_PX * (NT / 4);
    constexpr int DQ = (DV + C12_THREADS - 1) / C12_THREADS;
    constexpr int XV = 3 * C9_XW * (C9_KT / 4);
    constexpr int XQ = (XV + C12_THREADS - 1) / C12_THREADS;
    extern __shared__ float smem[];
    float* dyT = smem;                     // [32][DS]
    float* xT = smem + C9_PX * DS;         // [3][34][C9_XS]
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * NT, k0 = blockIdx.y * C9_KT;
    const long long s0 = (long long)blockIdx.z * p.steps_per_split;
    long long s1 = s0 + p.steps_per_split;
    if (s1 > p.steps) s1 = p.steps;

    const bool gaff = p.gs != nullptr;
    const float* yp = gaff ? p.y : p.g;
    const int gact = gaff ? p.gact : SSDSEG_ACT_NONE;
    const float xlo = act_lo(p.xact), xhi = act_hi(p.xact);
    const int dn4 = t % (NT / 4);          // 768 % 32 == 0: the float4 column of every staging slot is fixed
    const bool dnok = n0 + dn4 * 4 < p.N;
    float4 cgs = f4(1.f), cgt = f4(0.f), cgk1 = f4(0.f), cgk0 = f4(0.f);
    if (gaff && dnok) { cgs = ld4(p.gs + n0 + dn4 * 4); cgt = ld4(p.gt + n0 + dn4 * 4); cgk1 = ld4(p.gk1 + n0 + dn4 * 4); cgk0 = ld4(p.gk0 + n0 + dn4 * 4); }
    const int xk4 = t & 7;
    const bool xkok = k0 + xk4 * 4 < p.K;
    float4 cxs = f4(1.f), cxt = f4(0.f);
    if (p.xs != nullptr && xkok) { cxs = ld4(p.xs + k0 + xk4 * 4); cxt = ld4(p.xt + k0 + xk4 * 4); }

    float4 sg[DQ], sy[DQ], sx[XQ];
    unsigned okd = 0, okx = 0;
    auto issue = [&](long long s) {
        const int wc = (int)(s % p.wchunks);
        const long long row = s / p.wchunks;
        const int hrow = (int)(row % p.h);
        const long long img = row / p.h;
        const int w0 = wc * C9_PX;
        okd = okx = 0;
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            const int idx = t + C12_THREADS * q;
            const int px = idx / (NT / 4);
            const bool ok = idx < DV && dnok && w0 + px < p.w;
            const long long off = ok ? (row * p.w + w0 + px) * (long long)p.N + n0 + dn4 * 4 : 0;
            sg[q] = ld4(p.g + off);
            sy[q] = ld4(yp + off);
            okd |= (ok ? 1u : 0u) << q;
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + C12_THREADS * q;
            const int pxr = idx >> 3, r = pxr / C9_XW, c = pxr - r * C9_XW;
            const int hi = hrow + r - 1, wi = w0 + c - 1;
            const bool ok = idx < XV && xkok && hi >= 0 && hi < p.h && wi >= 0 && wi < p.w;
            const long long off = ok ? ((img * p.h + hi) * p.w + wi) * (long long)p.ldx + k0 + xk4 * 4 : 0;
            sx[q] = ld4(p.x + off);
            okx |= (ok ? 1u : 0u) << q;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            const int idx = t + C12_THREADS * q;
            if (idx < DV) {
                const int px = idx / (NT / 4);
                const float4 v = gview_apply4(sg[q], sy[q], cgs, cgt, cgk1, cgk0, gact);
                st4(dyT + px * DS + dn4 * 4, ((okd >> q) & 1u) ? v : f4(0.f));
            }
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + C12_THREADS * q;
            if (idx < XV) {
                const int pxr = idx >> 3;
                const float4 v = view_affine4(sx[q], cxs, cxt, xlo, xhi);
                st4(xT + pxr * C9_XS + xk4 * 4, ((okx >> q) & 1u) ? v : f4(0.f));
            }
        }
    };

    // this wave's three (tap, column tile) units
    int utap[3], unt[3];
    const float* xa[3];
    const float* yb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = 3 * wave + j;
        utap[j] = u >> 2;
        unt[j] = u & 3;
        const int dh = utap[j] / 3 - 1, dw = utap[j] % 3 - 1;
        xa[j] = xT + ((dh + 1) * C9_XW + dw + 1 + hh) * C9_XS + li;
        yb[j] = dyT + hh * DS + li + unt[j] * 32;
    }
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    if (s0 < s1) issue(s0);
    for (long long s = s0; s < s1; ++s) {
        __syncthreads();
        commit();
        __syncthreads();
        issue(s + 1 < s1 ? s + 1 : s);
        constexpr int PF = 2, NST = C9_PX / 2;
        float afr[PF + 1][3], bfr[PF + 1][3];
        auto fetch = [&](int ps, int buf) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                afr[buf][j] = xa[j][(2 * ps) * C9_XS];
                bfr[buf][j] = yb[j][(2 * ps) * DS];
            }
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) fetch(q, q);
#pragma unroll
        for (int ps = 0; ps < NST; ++ps) {
            if (ps + PF < NST) fetch(ps + PF, (ps + PF) % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = mfma32(afr[ps % (PF + 1)][j], bfr[ps % (PF + 1)][j], acc[j]);
        }
    }

#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float* out = p.part + ((long long)blockIdx.z * 9 + utap[j]) * p.K * p.N;
        const int n = n0 + unt[j] * 32 + li;
        if (n < p.N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (k < p.K) out[(long long)k * p.N + n] = acc[j][e];
            }
        }
    }
}
