// LDS-tiled depthwise 3x3 kernels (dilation 1, stride 1|2) -- included by dwconv.hip after its helper definitions.
//
// Why LDS: the register-window kernels in dwconv.hip re-fetch every input ~4.5x through L1/L2 and, at 120-240 VGPRs,
// keep only 2-4 waves per SIMD in flight, so they are bound by load latency, not by HBM (measured 1.4-3.0 TB/s).  Here a
// 256-thread block stages a (TH*S+2) x (TW*S+2) x 32-channel halo tile with plain coalesced float4 loads -- 6-10
// independent loads per thread -- and computes from LDS (ds_read_b128).  Each input is fetched ~1.3x (halo) instead of
// ~4.5x.  The producer's BatchNorm+ReLU6 (or the BN-backward gradient view) is applied once per element on the way into LDS.
//   thread = (channel vector cv = t & 7, strip = t >> 3); 32 strips of TWT outputs cover the TH x TW output tile.
//   A block walks tiles bpos.x, +gridDim.x, ... so BN-stat / dW partials stay in registers until the end.
//   Software pipeline: the RAW loads of tile i+1 are issued before the LDS compute of tile i and only consumed at the top of
//   the next iteration, so HBM requests stay in flight through the compute phase.
#pragma once

namespace {

constexpr int LCV = 8;            // float4 channel vectors per block (32 channels = 128 contiguous bytes per pixel)
constexpr int LPS = LCV + 1;      // LDS pixel stride in float4 (one float4 of padding against bank conflicts)
constexpr int LTW = 16;           // output tile width

template <int S> struct LTile {
    static constexpr int TH = S == 1 ? 8 : 4;           // output tile height
    static constexpr int TWT = S == 1 ? 4 : 2;          // outputs per thread strip
    static constexpr int SPR = LTW / TWT;               // strips per tile row; TH * SPR == 32
    static constexpr int IH = (TH - 1) * S + 3;         // input halo tile
    static constexpr int IW = (LTW - 1) * S + 3;
    static constexpr int DH = TH + 2, DW = LTW + 2;     // dy tile of the backward: rows ho0-1..ho0+TH, cols wo0-1..wo0+TW
    static constexpr int NA = IH * IW * LCV, NAI = (NA + 255) / 256;   // staging loads per thread (compile-time trip counts)
    static constexpr int ND = DH * DW * LCV, NDI = (ND + 255) / 256;
};

struct TilePos {
    long long img;
    int ho0, wo0;
};
__device__ __forceinline__ TilePos tile_pos(long long tile, int tiles_w, int tiles_h, int th_size) {
    TilePos p;
    const int tw = (int)(tile % tiles_w);
    const long long r = tile / tiles_w;
    p.img = r / tiles_h;
    p.ho0 = (int)(r % tiles_h) * th_size;
    p.wo0 = tw * LTW;
    return p;
}

// sum over the 32 strips of a block (threads sharing cv); result valid for t < LCV
__device__ __forceinline__ float4 reduce_strips(float4 v, float4* red, int t) {
    __syncthreads();
    red[t] = v;
    __syncthreads();
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < LCV)
        for (int s = 0; s < 32; ++s) add4(r, red[s * LCV + t]);
    return r;
}

template <int S>
__global__ void __launch_bounds__(256) dw_fwd_lds_kernel(DwGeom gm, ViewDev in, const float* __restrict__ wgt, float* __restrict__ y,
                                                         float* __restrict__ stats) {
    using T = LTile<S>;
    extern __shared__ float4 lds[];                      // [IH*IW][LPS]
    const BlockPos bpos = xcd_block_pos();   // (tile slot, channel group), neighbouring slots on the same XCD / L2
    const int t = threadIdx.x, cv = t & 7, strip = t >> 3;
    const int srow = strip / T::SPR, scol = (strip % T::SPR) * T::TWT;
    const int c0 = bpos.y * (LCV * 4) + cv * 4;
    const bool cok = c0 < gm.c;
    const int cs = cok ? c0 : 0;                         // safe channel offset for clamped loads
    float4 wk[9];
    float4 vs = f4(1.f), vt = f4(0.f);
    const float lo = act_lo(in.act), hi = act_hi(in.act);
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = ld4(wgt + (long long)k * gm.c + cs);
    if (in.scale != nullptr) { vs = ld4(in.scale + cs); vt = ld4(in.shift + cs); }
    float4 ssum = f4(0.f), ssq = f4(0.f);
    const int tiles_w = (gm.wo + LTW - 1) / LTW, tiles_h = (gm.ho + T::TH - 1) / T::TH;
    const long long ntiles = (long long)gm.n * tiles_h * tiles_w;

    float4 stage[T::NAI];
    unsigned okmask = 0;
    auto fetch = [&](long long tile) {
        const TilePos tp = tile_pos(tile, tiles_w, tiles_h, T::TH);
        const int hi0 = tp.ho0 * S - gm.pt, wi0 = tp.wo0 * S - gm.pl;
        okmask = 0;
#pragma unroll
        for (int k = 0; k < T::NAI; ++k) {
            const int idx = t + 256 * k;      // this thread always handles its own channel vector: 256 % 8 == 0
            const int p = idx >> 3, py = p / T::IW, px = p - py * T::IW;
            const int hy = hi0 + py, wx = wi0 + px;
            const bool ok = idx < T::NA && cok && hy >= 0 && hy < gm.h && wx >= 0 && wx < gm.w;
            stage[k] = ld4(in.x + (ok ? ((tp.img * gm.h + hy) * gm.w + wx) * gm.c + c0 : (long long)cs));
            okmask |= (ok ? 1u : 0u) << k;
        }
    };
    if ((long long)bpos.x < ntiles) fetch(bpos.x);
    for (long long tile = bpos.x; tile < ntiles; tile += gridDim.x) {
        const TilePos tp = tile_pos(tile, tiles_w, tiles_h, T::TH);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < T::NAI; ++k) {
            const int idx = t + 256 * k;
            if (idx < T::NA) lds[(idx >> 3) * LPS + cv] = sel4((okmask >> k) & 1u, view_affine4(stage[k], vs, vt, lo, hi));
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        // ---- compute TWT outputs from LDS
        float4 out[T::TWT];
#pragma unroll
        for (int j = 0; j < T::TWT; ++j) out[j] = f4(0.f);
        constexpr int WC = (T::TWT - 1) * S + 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float4 row[WC];
            const float4* src = lds + ((srow * S + kh) * T::IW + scol * S) * LPS + cv;
#pragma unroll
            for (int q = 0; q < WC; ++q) row[q] = src[q * LPS];
#pragma unroll
            for (int j = 0; j < T::TWT; ++j)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) fma4(out[j], row[j * S + kw], wk[kh * 3 + kw]);
        }
        const int ho = tp.ho0 + srow;
#pragma unroll
        for (int j = 0; j < T::TWT; ++j) {
            const int wo = tp.wo0 + scol + j;
            if (cok && ho < gm.ho && wo < gm.wo) {
                st4(y + ((tp.img * gm.ho + ho) * gm.wo + wo) * gm.c + c0, out[j]);
                add4(ssum, out[j]);
                fma4(ssq, out[j], out[j]);
            }
        }
    }
    if (stats != nullptr) {
        const float4 a = reduce_strips(ssum, lds, t);
        const float4 b = reduce_strips(ssq, lds, t);
        if (t < LCV && cok) {
            float* row = stats + (long long)bpos.x * 2 * gm.c;
            st4(row + c0, a);
            st4(row + gm.c + c0, b);
        }
    }
}

// dx (gradient w.r.t. the activated input) + per-block partial of dW.  LDS holds the activated-input halo tile, the dy halo
// tile (dy formed from (g, y) by the gradient view on the way in) and -- to keep them out of the register file -- the 9
// filter taps and the 6 per-channel view coefficients of the block's 8 channel vectors.
template <int S, int PT, int PL>
__global__ void __launch_bounds__(256) dw_bwd_lds_kernel(DwGeom gm, ViewDev in, const float* __restrict__ wgt, GViewDev dy,
                                                         float* __restrict__ dx, float* __restrict__ dwpart, int accumulate) {
    using T = LTile<S>;
    constexpr int DW = T::DW;
    extern __shared__ float4 lds[];
    float4* ta = lds;                                    // [IH*IW][LPS] activated input
    float4* td = ta + T::IH * T::IW * LPS;               // [DH*DW][LPS] dy
    float4* cw = td + T::DH * DW * LPS;                  // [9][LCV] filter taps, then [6][LCV] coefficients
    float4* cc = cw + 9 * LCV;
    const BlockPos bpos = xcd_block_pos();   // (tile slot, channel group), neighbouring slots on the same XCD / L2
    const int t = threadIdx.x, cv = t & 7, strip = t >> 3;
    const int srow = strip / T::SPR, scol = (strip % T::SPR) * T::TWT;
    const int c0 = bpos.y * (LCV * 4) + cv * 4;
    const bool cok = c0 < gm.c;
    const int cs = cok ? c0 : 0;
    const bool gaff = dy.scale != nullptr;
    if (!gaff) { dy.y = dy.g; dy.act = SSDSEG_ACT_NONE; }   // identity gradient view, branch-free form (see common.h)
    const float ilo = act_lo(in.act), ihi = act_hi(in.act);
    if (t < 9 * LCV) cw[t] = ld4(wgt + (long long)(t >> 3) * gm.c + cs);
    if (t < LCV) {
        const bool iaff = in.scale != nullptr;
        cc[0 * LCV + t] = iaff ? ld4(in.scale + cs) : f4(1.f);
        cc[1 * LCV + t] = iaff ? ld4(in.shift + cs) : f4(0.f);
        cc[2 * LCV + t] = gaff ? ld4(dy.scale + cs) : f4(1.f);
        cc[3 * LCV + t] = gaff ? ld4(dy.shift + cs) : f4(0.f);
        cc[4 * LCV + t] = gaff ? ld4(dy.k1 + cs) : f4(0.f);
        cc[5 * LCV + t] = gaff ? ld4(dy.k0 + cs) : f4(0.f);
    }
    float4 dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = f4(0.f);

    const int tiles_w = (gm.wo + LTW - 1) / LTW, tiles_h = (gm.ho + T::TH - 1) / T::TH;
    const long long ntiles = (long long)gm.n * tiles_h * tiles_w;

    float4 sa[T::NAI], sg[T::NDI], sy[T::NDI];           // raw loads in flight across the compute phase
    unsigned oka = 0, okd = 0;
    auto fetch = [&](long long tile) {
        const TilePos tp = tile_pos(tile, tiles_w, tiles_h, T::TH);
        const int hi0 = tp.ho0 * S - PT, wi0 = tp.wo0 * S - PL;
        oka = okd = 0;
#pragma unroll
        for (int k = 0; k < T::NAI; ++k) {
            const int idx = t + 256 * k;
            const int p = idx >> 3, py = p / T::IW, px = p - py * T::IW;
            const int hy = hi0 + py, wx = wi0 + px;
            const bool ok = idx < T::NA && cok && hy >= 0 && hy < gm.h && wx >= 0 && wx < gm.w;
            sa[k] = ld4(in.x + (ok ? ((tp.img * gm.h + hy) * gm.w + wx) * gm.c + c0 : (long long)cs));
            oka |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < T::NDI; ++k) {
            const int idx = t + 256 * k;
            const int p = idx >> 3, py = p / DW, px = p - py * DW;
            const int hh = tp.ho0 - 1 + py, ww = tp.wo0 - 1 + px;
            const bool ok = idx < T::ND && cok && hh >= 0 && hh < gm.ho && ww >= 0 && ww < gm.wo;
            const long long o = ok ? ((tp.img * gm.ho + hh) * gm.wo + ww) * gm.c + c0 : (long long)cs;
            sg[k] = ld4(dy.g + o);
            sy[k] = ld4(dy.y + o);
            okd |= (ok ? 1u : 0u) << k;
        }
    };
    if ((long long)bpos.x < ntiles) fetch(bpos.x);
    for (long long tile = bpos.x; tile < ntiles; tile += gridDim.x) {
        const TilePos tp = tile_pos(tile, tiles_w, tiles_h, T::TH);
        __syncthreads();
        {
            const float4 is = cc[0 * LCV + cv], it = cc[1 * LCV + cv];
#pragma unroll
            for (int k = 0; k < T::NAI; ++k) {
                const int idx = t + 256 * k;
                if (idx < T::NA) ta[(idx >> 3) * LPS + cv] = sel4((oka >> k) & 1u, view_affine4(sa[k], is, it, ilo, ihi));
            }
            const float4 gs = cc[2 * LCV + cv], gt = cc[3 * LCV + cv], gk1 = cc[4 * LCV + cv], gk0 = cc[5 * LCV + cv];
#pragma unroll
            for (int k = 0; k < T::NDI; ++k) {
                const int idx = t + 256 * k;
                if (idx < T::ND) td[(idx >> 3) * LPS + cv] = sel4((okd >> k) & 1u, gview_apply4(sg[k], sy[k], gs, gt, gk1, gk0, dy.act));
            }
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);

        // ---- dW: activated-input window rows against this strip's dy values (dy tile row srow+1, cols scol+1+j)
        float4 dyc[T::TWT];
#pragma unroll
        for (int j = 0; j < T::TWT; ++j) dyc[j] = td[((srow + 1) * DW + scol + 1 + j) * LPS + cv];
        constexpr int WC = (T::TWT - 1) * S + 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float4 row[WC];
            const float4* src = ta + ((srow * S + kh) * T::IW + scol * S) * LPS + cv;
#pragma unroll
            for (int q = 0; q < WC; ++q) row[q] = src[q * LPS];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int j = 0; j < T::TWT; ++j) fma4(dwacc[kh * 3 + kw], row[j * S + kw], dyc[j]);
        }

        // ---- dx over the owned input patch: rows (ho0+srow)*S + ir, cols (wo0+scol)*S + ic
        if (dx != nullptr) {
#pragma unroll
            for (int ir = 0; ir < S; ++ir) {
                const int hy = (tp.ho0 + srow) * S + ir;
#pragma unroll
                for (int ic = 0; ic < T::TWT * S; ++ic) {
                    const int wx = (tp.wo0 + scol) * S + ic;
                    float4 acc = f4(0.f);
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        if (((ir + PT - kh) % S + S) % S != 0) continue;           // compile-time parity test
                        const int dr = (ir + PT - kh) / S;                         // output row = ho0 + srow + dr, dr in {-1,0,1}
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            if (((ic + PL - kw) % S + S) % S != 0) continue;
                            const int dc = (ic + PL - kw) / S;                     // output col = wo0 + scol + dc
                            if (dr < -1 || dr > 1) continue;
                            fma4(acc, td[((srow + 1 + dr) * DW + scol + 1 + dc) * LPS + cv], cw[(kh * 3 + kw) * LCV + cv]);
                        }
                    }
                    if (cok && hy < gm.h && wx < gm.w) {
                        float* p = dx + ((tp.img * gm.h + hy) * gm.w + wx) * gm.c + c0;
                        if (accumulate) add4(acc, ld4(p));
                        st4(p, acc);
                    }
                }
            }
        }
    }
    // ---- block partial of dW: [gridDim.x][9][c]
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float4 v = reduce_strips(dwacc[k], lds, t);
        if (t < LCV && cok) st4(dwpart + ((long long)bpos.x * 9 + k) * gm.c + c0, v);
    }
}

struct LdsLaunch {
    dim3 grid;
    size_t lds_fwd, lds_bwd;
};

template <int S>
LdsLaunch lds_launch(const DwGeom& g) {
    using T = LTile<S>;
    LdsLaunch l;
    const long long tiles = (long long)g.n * ((g.ho + T::TH - 1) / T::TH) * ((g.wo + LTW - 1) / LTW);
    const int cgroups = (g.c + LCV * 4 - 1) / (LCV * 4);
    long long gx = tiles < 1024 ? tiles : 1024;            // also the number of BN-stat / dW partial rows
    // spread the channel groups: keep about 4096 blocks in total when a layer has many channel groups
    if (gx * cgroups > 4096) gx = 4096 / cgroups < 64 ? 64 : 4096 / cgroups;
    if (gx > tiles) gx = tiles;
    if (gx < 1) gx = 1;
    gx = (gx + 7) & ~7LL;   // multiple of 8 for the XCD remap (surplus blocks find no tile and write zero partial rows)
    l.grid = dim3((unsigned)gx, cgroups, 1);
    l.lds_fwd = (size_t)T::IH * T::IW * LPS * sizeof(float4);
    l.lds_bwd = l.lds_fwd + (size_t)(T::DH * T::DW * LPS + 15 * LCV) * sizeof(float4);
    const size_t red = 256 * sizeof(float4);
    if (l.lds_fwd < red) l.lds_fwd = red;
    return l;
}

}  // namespace
