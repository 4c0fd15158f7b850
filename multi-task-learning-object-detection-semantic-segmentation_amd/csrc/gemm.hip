// Pointwise (1x1) convolution as GEMM on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// products, fp32 accumulate -- the only MFMA dtype that keeps 1e-3 rel through ~60 layers without operand
// splitting).  Replaces Conv2D 1x1 / the pointwise half of SeparableConv2D (reference models.py:65,110;
// blocks.py:28,58,70,109) forward, backward-data and backward-weight.
//
//   fwd        y[m][n]  = sum_k act(scale_k*x[m][k]+shift_k) * w[k][n]         + per-channel (sum,sumsq) partials
//   bwd_data   dx[m][k] = sum_n dy[m][n] * w[k][n]   (+ residual, + accumulate)  dy formed on load (BN backward)
//   bwd_weight dw[k][n] = sum_m a[m][k] * dy[m][n]                               split over m, fixed-order reduce
//
// fwd/bwd_data share one kernel ("rowA": the streamed operand is row-major with the reduction axis contiguous).
// Tile: 128 rows x (32*WN) cols per 256-thread block, BK = 32.  The streamed operand is staged through LDS as
// [128][32+4] so that each lane fetches 4 consecutive k with one conflict-free ds_read_b128; since a GEMM may
// visit k in any order, MFMA #jj of a group takes k = 8*kk + jj from lanes 0-31 and k = 8*kk + 4 + jj from lanes
// 32-63 (for A and B alike).  The weight tile sits in LDS as [32][BN+1] and is read with conflict-free ds_read_b32.
// The next tile's global loads are issued before the current tile's MFMAs (register prefetch).
#include "common.h"
#include <vector>

#include <stdlib.h>

// bn.hip: (dgamma, dbeta, k1, k0) from nparts partial rows of (sum mask*g, sum mask*g*xhat)
int ssdseg_bn_bwd_finalize_launch(ssdseg_ctx* ctx, const float* part, int nparts, int c, double count, const float* scale,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* k1, float* k0);
int ssdseg_colsum(ssdseg_ctx* ctx, const float* part, int nparts, long long len, float* out);

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int AS = BK + 4;

struct RowAArgs {
    const float* a0;   // x (fwd) | g (bwd_data)
    const float* a1;   // unused  | y
    const float* cs;   // per-reduction-channel coefficients (nullable -> identity)
    const float* ct;
    const float* ck1;
    const float* ck0;
    int act;
    int lda;
    const float* b;
    int ldb;
    float* out;
    int ldo;
    const float* residual;
    int ldr;
    int accumulate;
    float* stats;
    int I, R, J;
    // dense 3x3 stride-1 SAME convolution as implicit GEMM (CONV kernels): the reduction axis is (tap, channel) with
    // convC channels per tap, row m = (n, h, w) reads the streamed operand at (h + sign*(kh-1), w + sign*(kw-1))
    int convH, convW, convC, convSign;
    // stem (LD == 2): 3x3 stride-2 SAME conv on a 3-channel image as implicit GEMM with R = 27 = (kh, kw, ci); row m =
    // (n, ho, wo) over convH x convW OUTPUT pixels reads image (2*ho + kh - stemPt, 2*wo + kw - stemPl, ci) of a
    // stemH x stemW image, rescaled on load (x*stemScale + stemOffset; padding stays 0).  bias: optional, added in the epilogue.
    int stemH, stemW, stemPt, stemPl;
    float stemScale, stemOffset;
    const float* bias;
    // fused backward (NT > 0, MODE 1): the layer's INPUT view x[I][J] (a = act(xs*x + xt)) and the per-block partial slabs of
    // dW[J][R] = sum_m a[m][j] * dy[m][r]  ([gridDim.y][J][R]); the dy tile already sits in LDS for the dx product
    const float* xw;
    const float* xws;
    const float* xwt;
    int xwact, ldxw;
    float* wpart;
    // BNE (MODE 1): BatchNorm backward of the producer of this conv's INPUT, fused into the (LDS-transposed, float4) epilogue:
    // raw input bn_y[I][J] (ld ldby), that BN's scale/shift/mean/invstd/activation; bnpart: [gridDim.y][2][J] partial rows of
    // (sum mask*dx, sum mask*dx*xhat)
    const float* bn_y;
    const float* bn_s;
    const float* bn_t;
    const float* bn_mean;
    const float* bn_istd;
    int bn_act, ldby;
    float* bnpart;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// LD: how the streamed operand is addressed -- 0 plain row-major matrix, 1 dense 3x3 stride-1 gather, 2 stem gather
// NT > 0 (MODE 1, LD 0, WN 1, J <= 32, R <= 32*NT): fused backward of a pointwise conv with few input channels (the MBConv
// expand convs): the same pass over (g, y) that yields dx = dy * W^T also accumulates dW = a^T * dy, one 32x32 tile per
// 32-column chunk of dy, each wave over its own 32 rows -- so the 6x-wide gradient is read once instead of twice.
// BNE (MODE 1, LD 0, NT 0): float4 epilogue through LDS -- the 128 x BN tile leaves the accumulators in two 64-row halves,
// is re-read as rows of float4 (thread = fixed float4 column, several rows), added to residual / previous contents, stored
// with 16-byte lanes, and in the same pass reduced into the BatchNorm-backward sums of the layer that produced this conv's
// input (one extra read of that layer's raw output instead of a separate two-tensor reduction pass).
// EP: 0 = scalar epilogue straight from the accumulators, 1 = the float4 epilogue, 2 = float4 epilogue + BN backward sums
// occupancy targets (waves per SIMD; a block is 4 waves = one per SIMD): the widest tile used to land at 264-300 registers,
// i.e. ONE block per CU, where every barrier and every global->LDS hand-over idles the matrix pipe
// (measured: conv backward-data 77 -> 96 TFLOP/s at two blocks per CU, the HBM-bound early pointwise layers +20-50 %).
// OCC = 1 instantiations carry the target; OCC = 0 keeps the compiler's own allocation (accumulators in AGPRs, looser
// schedule), which is what the short-M late layers want: they are latency- not occupancy-bound (too few blocks to fill the
// chip twice anyway) and ran 10-25 % SLOWER with the tight allocation.  Dispatch: rows >= ROWA_OCC_ROWS.
constexpr int rowa_min_waves(int WN, int MODE, int NT) { return NT > 0 ? 2 : (WN >= 2 ? 2 : (MODE == 0 ? 4 : 3)); }
constexpr int ROWA_OCC_ROWS = 150000;
// (read on every call, like the other dispatch switches: the parity tests flip it between calls to force either instantiation)
inline long long occ_rows() { const char* e = getenv("SSDSEG_OCC_ROWS"); return e != nullptr ? atoll(e) : ROWA_OCC_ROWS; }
template <int WN, int MODE, int LD, int NT = 0, int EP = 0, int OCC = 0>
__global__ void __launch_bounds__(256, OCC ? rowa_min_waves(WN, MODE, NT) : 1) gemm_rowA_kernel(RowAArgs p) {
    constexpr bool CONV = LD == 1, STEM = LD == 2, FUSEW = NT > 0, F4 = EP >= 1, BNE = EP == 2;
    static_assert(!F4 || (LD == 0 && NT == 0), "float4 epilogue only for the plain pointwise GEMM");
    static_assert(!BNE || MODE == 1, "BN epilogue only for backward-data");
    static_assert(!FUSEW || (MODE == 1 && LD == 0 && WN == 1), "fused dW only for plain backward-data with one column tile");
    constexpr int BN = 32 * WN;
    constexpr int BS = BN + 1;
    extern __shared__ float smem[];
    float* As = smem;                 // [BM][AS]
    float* Bs = smem + BM * AS;       // [BK][BS]

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * BN;
    const int mtiles = (p.I + BM - 1) / BM;
    const int KT = (p.R + BK - 1) / BK;
    const bool affine = p.cs != nullptr;

    const int a_c4 = t & 7;
    const int a_r = t >> 3;
    float4 areg[4];
    float4 breg[WN];

    const float alo = act_lo(p.act), ahi = act_hi(p.act);
    const float* ya = affine ? p.a1 : p.a0;                       // identity gradient view: y aliases g, act NONE
    const int gact = affine ? p.act : SSDSEG_ACT_NONE;
    int m0 = 0;
    int rn[4], rh[4], rw[4];   // CONV: (image, row, col) of this thread's 4 staged rows, fixed for a row tile
    auto load_tiles = [&](int kt) {
        const int r = kt * BK + a_c4 * 4;
        const bool rin = r < p.R;
        int ch = r, dh = 0, dw = 0;   // channel inside the tap, spatial offset of the tap
        if (CONV) {
            const int tap = r / p.convC;
            ch = r - tap * p.convC;
            const int kh = tap / 3;
            dh = (kh - 1) * p.convSign;
            dw = (tap - kh * 3 - 1) * p.convSign;
        }
        float4 cs = f4(1.f), ct = f4(0.f), ck1 = f4(0.f), ck0 = f4(0.f);   // identity view unless per-channel coefficients exist
        if (affine && rin) {
            cs = ld4(p.cs + ch);
            ct = ld4(p.ct + ch);
            if (MODE == 1) { ck1 = ld4(p.ck1 + ch); ck0 = ld4(p.ck0 + ch); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + a_r + 32 * i;
            float4 v = f4(0.f);
            bool ok = rin && m < p.I;
            long long off = (long long)m * p.lda + r;
            if (CONV) {
                const int sh = rh[i] + dh, sw = rw[i] + dw;
                ok = ok && sh >= 0 && sh < p.convH && sw >= 0 && sw < p.convW;
                off = (((long long)rn[i] * p.convH + sh) * p.convW + sw) * p.lda + ch;
            }
            if (STEM) {
                // four scalar gathers: r .. r+3 are (kh, kw, ci) triples of a 3-channel pixel row, not 16-byte aligned
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rq = r + q, tap = rq / 3, ci = rq - tap * 3, kh = tap / 3, kw = tap - kh * 3;
                    const int hi = 2 * rh[i] + kh - p.stemPt, wi = 2 * rw[i] + kw - p.stemPl;
                    const bool okq = rq < p.R && m < p.I && hi >= 0 && hi < p.stemH && wi >= 0 && wi < p.stemW;
                    const long long o = okq ? (((long long)rn[i] * p.stemH + hi) * p.stemW + wi) * 3 + ci : 0;
                    const float x = p.a0[o];
                    e[q] = okq ? fmaf(x, p.stemScale, p.stemOffset) : 0.f;
                }
                areg[i] = make_float4(e[0], e[1], e[2], e[3]);
                continue;
            }
            // unconditional loads from a clamped address + select: the 4 row loads (and their twins for y) issue back to back
            if (!ok) off = 0;
            if (MODE == 0) v = view_affine4(ld4(p.a0 + off), cs, ct, alo, ahi);
            else v = gview_apply4(ld4(p.a0 + off), ld4(ya + off), cs, ct, ck1, ck0, gact);
            areg[i] = ok ? v : f4(0.f);
        }
#pragma unroll
        for (int q = 0; q < WN; ++q) {
            const int idx = t + 256 * q;
            float4 v = f4(0.f);
            if (MODE == 0) {
                const int rr = idx / (8 * WN), j4 = idx % (8 * WN);
                const int gr = kt * BK + rr, gj = j0 + j4 * 4;
                const bool bok = gr < p.R && gj < p.J;
                v = ld4(p.b + (bok ? (long long)gr * p.ldb + gj : 0));
                if (!bok) v = f4(0.f);
            } else {
                const int jj = idx >> 3, r4 = idx & 7;
                const int gr = kt * BK + r4 * 4, gj = j0 + jj;
                // CONV: w[tap][j][c] with r = tap*convC + c  ->  j*convC + r + tap*convC*(J-1)
                const long long tapoff = CONV ? (long long)(gr / p.convC) * p.convC * (p.J - 1) : 0;
                const bool bok = gr < p.R && gj < p.J;
                v = ld4(p.b + (bok ? (long long)gj * p.ldb + gr + tapoff : 0));
                if (!bok) v = f4(0.f);
            }
            breg[q] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(As + (a_r + 32 * i) * AS + a_c4 * 4, areg[i]);
#pragma unroll
        for (int q = 0; q < WN; ++q) {
            const int idx = t + 256 * q;
            if (MODE == 0) {
                const int rr = idx / (8 * WN), j4 = idx % (8 * WN);
                float* d = Bs + rr * BS + j4 * 4;
                d[0] = breg[q].x; d[1] = breg[q].y; d[2] = breg[q].z; d[3] = breg[q].w;
            } else {
                const int jj = idx >> 3, r4 = idx & 7;
                float* d = Bs + (r4 * 4) * BS + jj;
                d[0] = breg[q].x; d[BS] = breg[q].y; d[2 * BS] = breg[q].z; d[3 * BS] = breg[q].w;
            }
        }
    };

    float ssum[WN], ssq[WN];   // BN statistics of this block's rows, carried across its row tiles
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) ssum[nt] = ssq[nt] = 0.f;
    f32x16 wacc[FUSEW ? NT : 1];   // FUSEW: this wave's partial of dW[j][32*kt + ..] over its rows, carried across row tiles
#pragma unroll
    for (int q = 0; q < (FUSEW ? NT : 1); ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) wacc[q][e] = 0.f;
    float xop[FUSEW ? 16 : 1];     // FUSEW: A operand a[m0 + 32*wave + 2s + hh][j = li] of the dW products, per row tile
    // BNE: this thread's float4 column of the epilogue and its BatchNorm constants / running sums
    float4 ebs = f4(0.f), ebt = f4(0.f), ebm = f4(0.f), ebi = f4(0.f), esb = f4(0.f), esg = f4(0.f);
    const float bnlo = act_lo(p.bn_act), bnhi = act_hi(p.bn_act);
    if (BNE) {
        const int ec4 = t % (BN / 4), ej = j0 + ec4 * 4;
        if (ej < p.J) { ebs = ld4(p.bn_s + ej); ebt = ld4(p.bn_t + ej); ebm = ld4(p.bn_mean + ej); ebi = ld4(p.bn_istd + ej); }
    }

    // a block walks row tiles blockIdx.y, blockIdx.y + gridDim.y, ... so the number of BN partial rows stays small
    for (int mt = blockIdx.y; mt < mtiles; mt += gridDim.y) {
    m0 = mt * BM;
    if (CONV || STEM) {
        const int hw = p.convH * p.convW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + a_r + 32 * i;
            rn[i] = m / hw;
            const int rem = m - rn[i] * hw;
            rh[i] = rem / p.convW;
            rw[i] = rem - rh[i] * p.convW;
        }
    }
    f32x16 acc[WN], accb;
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;

    if (FUSEW) {
        const float xlo = act_lo(p.xwact), xhi = act_hi(p.xwact);
        const bool jok = li < p.J;
        const float xs = (p.xws != nullptr && jok) ? p.xws[li] : 1.f, xt = (p.xws != nullptr && jok) ? p.xwt[li] : 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int m = m0 + wave * 32 + 2 * s + hh;
            const bool ok = jok && m < p.I;
            const float v = p.xw[ok ? (long long)m * p.ldxw + li : 0];
            xop[s] = ok ? fminf(fmaxf(fmaf(xs, v, xt), xlo), xhi) : 0.f;
        }
    }
    const int kt0 = 0, kt1 = KT;
    load_tiles(kt0);
    // one 32-deep reduction step; `wtile` = which dW accumulator this step feeds (a compile-time constant in the fused loop)
    auto kstep = [&](int kt, f32x16& wtile) {
        __syncthreads();
        store_tiles();
        __syncthreads();
        if (kt + 1 < kt1) load_tiles(kt + 1);
        if (FUSEW) {
            // dW tile kt: rows of the reduction = this wave's 32 tile rows, B operand dy[row][32*kt + li] straight from As
            const float* dcol = As + (wave * 32 + hh) * AS + li;
#pragma unroll
            for (int s = 0; s < 16; ++s) wtile = mfma32(xop[s], dcol[(2 * s) * AS], wtile);
        }
        const float* arow = As + (wave * 32 + li) * AS + 4 * hh;
        const float* bcol = Bs + (4 * hh) * BS + li;
        // Operand fragments are read from LDS one 8-deep group AHEAD of the MFMAs that use them (two register sets).  Left
        // to itself the compiler put each ds_read right in front of its MFMA with an s_waitcnt lgkmcnt(0), so the matrix
        // pipe idled for an LDS round trip every 1-2 instructions (38-59 TFLOP/s on the compute-bound layers).
        float4 afr[2];
        float bfr[2][4][WN];
        auto fetch = [&](int kk, int buf) {
            afr[buf] = ld4(arow + kk * 8);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt) bfr[buf][jj][nt] = bcol[(kk * 8 + jj) * BS + nt * 32];
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) fetch(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);   // keep the reads of group kk+1 in front of the MFMAs of group kk
            const float av[4] = {afr[kk & 1].x, afr[kk & 1].y, afr[kk & 1].z, afr[kk & 1].w};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int nt = 0; nt < WN; ++nt) {
                    // one column tile: alternate between two accumulators so consecutive MFMAs are independent
                    if (WN == 1 && (jj & 1)) accb = mfma32(av[jj], bfr[kk & 1][jj][nt], accb);
                    else acc[nt] = mfma32(av[jj], bfr[kk & 1][jj][nt], acc[nt]);
                }
            }
        }
    };
    if (FUSEW) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
            if (kt < KT) kstep(kt, wacc[kt]);
    } else {
        for (int kt = kt0; kt < kt1; ++kt) kstep(kt, wacc[0]);
    }
    if (WN == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] += accb[e];
    }

    if (F4) {
        constexpr int CS = BN + 4;        // LDS row stride of the transposed tile (float4-aligned)
        constexpr int CV = BN / 4;        // float4 columns
        constexpr int RPP = 256 / CV;     // rows per pass of the 256 threads
        float* Cs = smem;                 // [64][CS], over As/Bs (all waves are past their last operand read after the barrier)
        const int er = t / CV, ec4 = t - er * CV;
        const int ej = j0 + ec4 * 4;
        const bool eact = er < RPP && ej < p.J;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __syncthreads();
            if ((wave >> 1) == h) {
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Cs[((wave & 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * CS + nt * 32 + li] = acc[nt][e];
            }
            __syncthreads();
            if (eact) {
                // Rows er, er + RPP, ... of this half, FOUR at a time: the global loads of a group (the raw input of the fused
                // BatchNorm sums, residual, previous dx) are issued together and the group is then consumed in row order -- the
                // summation order per thread is unchanged.  One row per trip left a single 16-byte load in flight per thread
                // (8 KB per CU): the project-conv backward GEMMs of blocks 0-3 ran at 2.2-3.3 TB/s of real traffic.
                // Only where the registers are there (<= 3 column tiles, the register-limited big-M instantiations): at 4-5 column
                // tiles the extra float4s spill (block-2 project conv +11 %), and the short-M instantiations would lose a wave per SIMD.
                constexpr int ITER = (64 + RPP - 1) / RPP;
                constexpr int EU = (OCC && WN <= 3) ? 4 : 1;
                if constexpr (EU == 1) {
                // one row per trip, the raw input of the fused BatchNorm sums fetched one row AHEAD (two loads in flight per thread
                // for four more registers)
                float4 ynext = f4(0.f);
                if (BNE && m0 + h * 64 + er < p.I) ynext = ld4(p.bn_y + (long long)(m0 + h * 64 + er) * p.ldby + ej);
                for (int row = er; row < 64; row += RPP) {
                    const int m = m0 + h * 64 + row;
                    if (m >= p.I) break;
                    const float4 yv = ynext;
                    if (BNE && row + RPP < 64 && m + RPP < p.I) ynext = ld4(p.bn_y + (long long)(m + RPP) * p.ldby + ej);
                    float4 v = ld4(Cs + row * CS + ec4 * 4);
                    if (MODE == 1 && p.residual) {
                        const float4 r4 = ld4(p.residual + (long long)m * p.ldr + ej);
                        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                    }
                    float* o = p.out + (long long)m * p.ldo + ej;
                    if (MODE == 1 && p.accumulate) {
                        const float4 o4 = ld4(o);
                        v.x += o4.x; v.y += o4.y; v.z += o4.z; v.w += o4.w;
                    }
                    st4(o, v);
                    if (!BNE) continue;
                    float4 mg;
                    mg.x = (fmaf(ebs.x, yv.x, ebt.x) > bnlo && fmaf(ebs.x, yv.x, ebt.x) < bnhi) ? v.x : 0.f;
                    mg.y = (fmaf(ebs.y, yv.y, ebt.y) > bnlo && fmaf(ebs.y, yv.y, ebt.y) < bnhi) ? v.y : 0.f;
                    mg.z = (fmaf(ebs.z, yv.z, ebt.z) > bnlo && fmaf(ebs.z, yv.z, ebt.z) < bnhi) ? v.z : 0.f;
                    mg.w = (fmaf(ebs.w, yv.w, ebt.w) > bnlo && fmaf(ebs.w, yv.w, ebt.w) < bnhi) ? v.w : 0.f;
                    esb.x += mg.x; esb.y += mg.y; esb.z += mg.z; esb.w += mg.w;
                    esg.x = fmaf(mg.x, (yv.x - ebm.x) * ebi.x, esg.x); esg.y = fmaf(mg.y, (yv.y - ebm.y) * ebi.y, esg.y);
                    esg.z = fmaf(mg.z, (yv.z - ebm.z) * ebi.z, esg.z); esg.w = fmaf(mg.w, (yv.w - ebm.w) * ebi.w, esg.w);
                }
                } else {
#pragma unroll
                for (int it0 = 0; it0 < ITER; it0 += EU) {
                    float4 cv[EU], yv4[EU], rv4[EU], ov4[EU];
                    bool rok[EU];
#pragma unroll
                    for (int u = 0; u < EU; ++u) {
                        const int row = er + (it0 + u) * RPP;
                        const int m = m0 + h * 64 + row;
                        rok[u] = (it0 + u) < ITER && row < 64 && m < p.I;
                        const long long mm = rok[u] ? m : 0;               // clamped address, value unused
                        if (BNE) yv4[u] = ld4(p.bn_y + mm * p.ldby + ej);
                        if (MODE == 1 && p.residual) rv4[u] = ld4(p.residual + mm * p.ldr + ej);
                        if (MODE == 1 && p.accumulate) ov4[u] = ld4(p.out + mm * p.ldo + ej);
                        cv[u] = ld4(Cs + (rok[u] ? row : 0) * CS + ec4 * 4);
                    }
#pragma unroll
                    for (int u = 0; u < EU; ++u) {
                        if (!rok[u]) continue;
                        const int m = m0 + h * 64 + er + (it0 + u) * RPP;
                        float4 v = cv[u];
                        if (MODE == 1 && p.residual) { v.x += rv4[u].x; v.y += rv4[u].y; v.z += rv4[u].z; v.w += rv4[u].w; }
                        if (MODE == 1 && p.accumulate) { v.x += ov4[u].x; v.y += ov4[u].y; v.z += ov4[u].z; v.w += ov4[u].w; }
                        st4(p.out + (long long)m * p.ldo + ej, v);
                        if (!BNE) continue;
                        const float4 yv = yv4[u];
                        float4 mg;
                        mg.x = (fmaf(ebs.x, yv.x, ebt.x) > bnlo && fmaf(ebs.x, yv.x, ebt.x) < bnhi) ? v.x : 0.f;
                        mg.y = (fmaf(ebs.y, yv.y, ebt.y) > bnlo && fmaf(ebs.y, yv.y, ebt.y) < bnhi) ? v.y : 0.f;
                        mg.z = (fmaf(ebs.z, yv.z, ebt.z) > bnlo && fmaf(ebs.z, yv.z, ebt.z) < bnhi) ? v.z : 0.f;
                        mg.w = (fmaf(ebs.w, yv.w, ebt.w) > bnlo && fmaf(ebs.w, yv.w, ebt.w) < bnhi) ? v.w : 0.f;
                        esb.x += mg.x; esb.y += mg.y; esb.z += mg.z; esb.w += mg.w;
                        esg.x = fmaf(mg.x, (yv.x - ebm.x) * ebi.x, esg.x); esg.y = fmaf(mg.y, (yv.y - ebm.y) * ebi.y, esg.y);
                        esg.z = fmaf(mg.z, (yv.z - ebm.z) * ebi.z, esg.z); esg.w = fmaf(mg.w, (yv.w - ebm.w) * ebi.w, esg.w);
                    }
                }
                }
            }
        }
        __syncthreads();   // the tile is consumed: the next row tile may overwrite As/Bs
    } else {
    // ---------------- epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        const int j = j0 + nt * 32 + li;
        if (j < p.J) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (m < p.I) {
                    float v = acc[nt][e];
                    if (STEM && p.bias) v += p.bias[j];
                    if (MODE == 1) {
                        if (p.residual) v += p.residual[(long long)m * p.ldr + j];
                        if (p.accumulate) v += p.out[(long long)m * p.ldo + j];
                    }
                    p.out[(long long)m * p.ldo + j] = v;
                }
            }
        }
    }
    }   // !F4
    if (MODE == 0 && p.stats != nullptr) {
        // per-channel (sum, sumsq) of this 128-row tile; padded rows are exactly zero
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { ssum[nt] += acc[nt][e]; ssq[nt] = fmaf(acc[nt][e], acc[nt][e], ssq[nt]); }
        }
    }
    }  // row-tile loop

    if (FUSEW) {
        // this block's dW partial slab [J][R]: sum the four waves' row-quarters tile by tile through LDS (fixed order)
        float* red = smem;   // [3 waves][16][64]
        float* slab = p.wpart + (long long)blockIdx.y * p.J * p.R;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) red[((wave - 1) * 16 + e) * 64 + lane] = wacc[kt][e];
            }
            __syncthreads();
            if (wave == 0) {
                const int n = kt * 32 + li;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = ((wacc[kt][e] + red[(0 * 16 + e) * 64 + lane]) + red[(1 * 16 + e) * 64 + lane]) + red[(2 * 16 + e) * 64 + lane];
                    const int k = (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (k < p.J && n < p.R) slab[(long long)k * p.R + n] = v;
                }
            }
        }
    }

    if (BNE) {
        // fold the threads that share a float4 column (fixed order) -> this block's partial row
        constexpr int CV = BN / 4, RPP = 256 / CV;
        float4* red4 = reinterpret_cast<float4*>(smem);   // [2][256]
        __syncthreads();
        red4[t] = esb;
        red4[256 + t] = esg;
        __syncthreads();
        if (t < CV && j0 + t * 4 < p.J) {
            float4 a = f4(0.f), b = f4(0.f);
            for (int k = 0; k < RPP; ++k) {
                const float4 x = red4[k * CV + t], y = red4[256 + k * CV + t];
                a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
                b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
            }
            st4(p.bnpart + ((long long)blockIdx.y * 2 + 0) * p.J + j0 + t * 4, a);
            st4(p.bnpart + ((long long)blockIdx.y * 2 + 1) * p.J + j0 + t * 4, b);
        }
    }
    if (MODE == 0 && p.stats != nullptr) {
        __syncthreads();
        float* red = smem;  // [4 waves][2][BN]
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            float s = ssum[nt], q = ssq[nt];
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (hh == 0) {
                red[(wave * 2 + 0) * BN + nt * 32 + li] = s;
                red[(wave * 2 + 1) * BN + nt * 32 + li] = q;
            }
        }
        __syncthreads();
        for (int idx = t; idx < 2 * BN; idx += 256) {
            const int which = idx / BN, jl = idx % BN;
            const int j = j0 + jl;
            if (j < p.J) {
                float v = red[(0 * 2 + which) * BN + jl] + red[(1 * 2 + which) * BN + jl] + red[(2 * 2 + which) * BN + jl] +
                          red[(3 * 2 + which) * BN + jl];
                p.stats[((long long)blockIdx.y * 2 + which) * p.J + j] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ bwd_weight
struct WGradArgs {
    const float* x;  // view over [M][K]
    const float* xs;
    const float* xt;
    int xact;
    int ldx;
    const float* g;  // gview over [M][N]
    const float* y;
    const float* gs;
    const float* gt;
    const float* gk1;
    const float* gk0;
    int gact;
    int ldy;
    float* part;  // [P][K][N]
    int M, K, N;
    int rows_per_split;
    // one tap of a dense 3x3 conv: row m = (n, h, w) reads x at (h + dh, w + dw); convH == 0 -> plain GEMM
    int convH, convW, dh, dw;
    // stem: x is the 3-channel image, row m = (n, ho, wo) over convH x convW output pixels, k = (kh, kw, ci) (see RowAArgs)
    int stem, stemH, stemW, stemPt, stemPl;
    float stemScale, stemOffset;
};

// reduction rows per wave per step: 64 rows per block step when the waves split the rows 4- or 2-way, 32 when all four waves
// sit along k (16-row steps left that shape with two barriers per 8 MFMAs: 2.5-3.2 TB/s of real traffic)
constexpr int rw_of(int wr) { return wr == 4 ? 16 : 32; }

// WI waves along the output rows (k), WR waves splitting the reduction rows (m); WI*WR == 4.
// occupancy targets where the raw-load staging would otherwise cost a wave per SIMD (184 registers for <4,1,3>: 2 waves instead of 3)
constexpr int wgrad_min_waves(int WI, int WR, int WN) { return (WI == 4 && WN <= 2) ? 3 : 1; }
template <int WI, int WR, int WN>
__global__ void __launch_bounds__(256, wgrad_min_waves(WI, WR, WN)) gemm_wgrad_kernel(WGradArgs p) {
    constexpr int RW = rw_of(WR);
    constexpr int BI = 32 * WI, BJ = 32 * WN, BRT = RW * WR;
    extern __shared__ float smem[];
    float* Xs = smem;              // [BRT][BI]
    float* Ys = smem + BRT * BI;   // [BRT][BJ]
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;
    const int wi = wave % WI, wr = wave / WI;
    const int i0 = blockIdx.y * BI;   // k offset
    const int j0 = blockIdx.x * BJ;   // n offset
    const int split = blockIdx.z;
    const long long mbeg = (long long)split * p.rows_per_split;
    long long mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;
    const bool xaff = p.xs != nullptr, gaff = p.gs != nullptr;
    const float xlo = act_lo(p.xact), xhi = act_hi(p.xact);
    const float* yptr = gaff ? p.y : p.g;                        // identity gradient view: y aliases g, act NONE
    const int yact = gaff ? p.gact : SSDSEG_ACT_NONE;

    constexpr int XV = BRT * BI / 4;   // float4 per X tile (== 256 * 2)
    constexpr int YV = BRT * BJ / 4;
    constexpr int XQ = (XV + 255) / 256, YQ = (YV + 255) / 256;
    // Staging in two halves (as in gemm_wres.h): load_tiles() only issues the RAW loads of the next step; store_tiles() -- one
    // MFMA phase and a barrier later -- applies the views and writes LDS.  With the view arithmetic inside load_tiles every
    // step waited for its global loads before the first MFMA.  The per-channel view coefficients sit in LDS (loaded once).
    // (RAW = false: the 96-column tiles of the 30x40 / 15x20 stages, where the extra staging registers cost a wave per SIMD)
    constexpr bool RAW = !(WI == 4 && WN == 3);
    float4 xraw[RAW ? XQ : 1], graw[RAW ? YQ : 1], yraw[RAW ? YQ : 1];
    unsigned xok = 0, yok = 0;   // bit q (stem: bit 4q + element): the slot holds real data
    float* Xc = smem + BRT * (BI + BJ);   // [2][BI]: scale, shift of the X view
    float* Yc = Xc + 2 * BI;              // [4][BJ]: scale, shift, k1, k0 of the gradient view
    for (int i = t; i < BI; i += 256) {
        const int k = i0 + i;
        const bool ok = xaff && k < p.K;
        Xc[i] = ok ? p.xs[k] : 1.f;
        Xc[BI + i] = ok ? p.xt[k] : 0.f;
    }
    for (int i = t; i < BJ; i += 256) {
        const int n = j0 + i;
        const bool ok = gaff && n < p.N;
        Yc[i] = ok ? p.gs[n] : 1.f;
        Yc[BJ + i] = ok ? p.gt[n] : 0.f;
        Yc[2 * BJ + i] = ok ? p.gk1[n] : 0.f;
        Yc[3 * BJ + i] = ok ? p.gk0[n] : 0.f;
    }
    // (made visible by the first barrier of the main loop)

    auto load_tiles = [&](long long mrow) {
        xok = yok = 0;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + 256 * q;
            float4 v = f4(0.f);
            if (idx < XV) {
                const int rr = idx / (BI / 4), c4 = idx % (BI / 4);
                const long long m = mrow + rr;
                const int k = i0 + c4 * 4;
                bool ok = m < mend && k < p.K;
                long long src = m;
                if (p.stem) {
                    const long long hw = (long long)p.convH * p.convW;
                    const long long img = m / hw;
                    const int rem = (int)(m - img * hw);
                    const int ho = rem / p.convW, wo = rem - ho * p.convW;
                    float e[4];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int rq = k + qq, tap = rq / 3, ci = rq - tap * 3, kh = tap / 3, kw = tap - kh * 3;
                        const int hi = 2 * ho + kh - p.stemPt, wi = 2 * wo + kw - p.stemPl;
                        const bool okq = m < mend && rq < p.K && hi >= 0 && hi < p.stemH && wi >= 0 && wi < p.stemW;
                        e[qq] = p.x[okq ? ((img * p.stemH + hi) * p.stemW + wi) * 3 + ci : 0];
                        xok |= (okq ? 1u : 0u) << (4 * q + qq);
                    }
                    xraw[q] = make_float4(e[0], e[1], e[2], e[3]);
                    continue;
                }
                if (p.convH > 0 && ok) {
                    const long long hw = (long long)p.convH * p.convW;
                    const long long img = m / hw;
                    const int rem = (int)(m - img * hw);
                    const int hy = rem / p.convW + p.dh, wx = rem % p.convW + p.dw;
                    ok = hy >= 0 && hy < p.convH && wx >= 0 && wx < p.convW;
                    src = (img * p.convH + hy) * p.convW + wx;
                }
                v = ld4(p.x + (ok ? src * p.ldx + k : 0));
                xok |= (ok ? 1u : 0u) << (p.stem ? 4 * q : q);
            }
            xraw[q] = v;
        }
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const int idx = t + 256 * q;
            float4 g4 = f4(0.f), y4 = f4(0.f);
            if (idx < YV) {
                const int rr = idx / (BJ / 4), c4 = idx % (BJ / 4);
                const long long m = mrow + rr;
                const int n = j0 + c4 * 4;
                const bool ok = m < mend && n < p.N;
                const long long o = ok ? m * p.ldy + n : 0;
                g4 = ld4(p.g + o);
                y4 = ld4(yptr + o);
                yok |= (ok ? 1u : 0u) << q;
            }
            graw[q] = g4;
            yraw[q] = y4;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + 256 * q;
            if (idx < XV) {
                const int c4 = idx % (BI / 4);
                float4 v;
                if (p.stem) {
                    v.x = ((xok >> (4 * q + 0)) & 1u) ? fmaf(xraw[q].x, p.stemScale, p.stemOffset) : 0.f;
                    v.y = ((xok >> (4 * q + 1)) & 1u) ? fmaf(xraw[q].y, p.stemScale, p.stemOffset) : 0.f;
                    v.z = ((xok >> (4 * q + 2)) & 1u) ? fmaf(xraw[q].z, p.stemScale, p.stemOffset) : 0.f;
                    v.w = ((xok >> (4 * q + 3)) & 1u) ? fmaf(xraw[q].w, p.stemScale, p.stemOffset) : 0.f;
                } else {
                    v = view_affine4(xraw[q], ld4(Xc + c4 * 4), ld4(Xc + BI + c4 * 4), xlo, xhi);
                    if (!((xok >> q) & 1u)) v = f4(0.f);
                }
                st4(Xs + idx * 4, v);
            }
        }
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const int idx = t + 256 * q;
            if (idx < YV) {
                const int c4 = idx % (BJ / 4);
                float4 v = gview_apply4(graw[q], yraw[q], ld4(Yc + c4 * 4), ld4(Yc + BJ + c4 * 4), ld4(Yc + 2 * BJ + c4 * 4),
                                        ld4(Yc + 3 * BJ + c4 * 4), yact);
                if (!((yok >> q) & 1u)) v = f4(0.f);
                st4(Ys + idx * 4, v);
            }
        }
    };

    // ---- the original staging (view arithmetic at load time), kept for the shapes where it measured faster
    float4 xreg[RAW ? 1 : XQ], yreg[RAW ? 1 : YQ];   // !RAW: transformed at load time

    // per-thread channel coefficients are fixed across steps when the tile width divides 256 float4 columns;
    // otherwise they are re-read per step (they sit in L1).
    auto load_tiles_t = [&](long long mrow) {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + 256 * q;
            float4 v = f4(0.f);
            if (idx < XV) {
                const int rr = idx / (BI / 4), c4 = idx % (BI / 4);
                const long long m = mrow + rr;
                const int k = i0 + c4 * 4;
                bool ok = m < mend && k < p.K;
                long long src = m;
                if (p.stem) {
                    const long long hw = (long long)p.convH * p.convW;
                    const long long img = m / hw;
                    const int rem = (int)(m - img * hw);
                    const int ho = rem / p.convW, wo = rem - ho * p.convW;
                    float e[4];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int rq = k + qq, tap = rq / 3, ci = rq - tap * 3, kh = tap / 3, kw = tap - kh * 3;
                        const int hi = 2 * ho + kh - p.stemPt, wi = 2 * wo + kw - p.stemPl;
                        const bool okq = m < mend && rq < p.K && hi >= 0 && hi < p.stemH && wi >= 0 && wi < p.stemW;
                        const float xv = p.x[okq ? ((img * p.stemH + hi) * p.stemW + wi) * 3 + ci : 0];
                        e[qq] = okq ? fmaf(xv, p.stemScale, p.stemOffset) : 0.f;
                    }
                    xreg[q] = make_float4(e[0], e[1], e[2], e[3]);
                    continue;
                }
                if (p.convH > 0 && ok) {
                    const long long hw = (long long)p.convH * p.convW;
                    const long long img = m / hw;
                    const int rem = (int)(m - img * hw);
                    const int hy = rem / p.convW + p.dh, wx = rem % p.convW + p.dw;
                    ok = hy >= 0 && hy < p.convH && wx >= 0 && wx < p.convW;
                    src = (img * p.convH + hy) * p.convW + wx;
                }
                {
                    const int kk = ok ? k : 0;
                    float4 s = f4(1.f), sh = f4(0.f);
                    if (xaff) { s = ld4(p.xs + kk); sh = ld4(p.xt + kk); }
                    v = view_affine4(ld4(p.x + (ok ? src * p.ldx + k : 0)), s, sh, xlo, xhi);
                    if (!ok) v = f4(0.f);
                }
            }
            xreg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const int idx = t + 256 * q;
            float4 v = f4(0.f);
            if (idx < YV) {
                const int rr = idx / (BJ / 4), c4 = idx % (BJ / 4);
                const long long m = mrow + rr;
                const int n = j0 + c4 * 4;
                {
                    const bool ok = m < mend && n < p.N;
                    const long long o = ok ? m * p.ldy + n : 0;
                    const int nn = ok ? n : 0;
                    float4 gs = f4(1.f), gt = f4(0.f), gk1 = f4(0.f), gk0 = f4(0.f);
                    if (gaff) { gs = ld4(p.gs + nn); gt = ld4(p.gt + nn); gk1 = ld4(p.gk1 + nn); gk0 = ld4(p.gk0 + nn); }
                    v = gview_apply4(ld4(p.g + o), ld4(yptr + o), gs, gt, gk1, gk0, yact);
                    if (!ok) v = f4(0.f);
                }
            }
            yreg[q] = v;
        }
    };
    auto store_tiles_t = [&]() {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = t + 256 * q;
            if (idx < XV) st4(Xs + idx * 4, xreg[q]);
        }
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const int idx = t + 256 * q;
            if (idx < YV) st4(Ys + idx * 4, yreg[q]);
        }
    };


    f32x16 acc[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    if (mbeg < mend) { if (RAW) load_tiles(mbeg); else load_tiles_t(mbeg); }
    for (long long mrow = mbeg; mrow < mend; mrow += BRT) {
        __syncthreads();
        if (RAW) store_tiles(); else store_tiles_t();
        __syncthreads();
        if (mrow + BRT < mend) { if (RAW) load_tiles(mrow + BRT); else load_tiles_t(mrow + BRT); }
        const float* xa = Xs + (wr * RW + hh) * BI + wi * 32 + li;
        const float* yb = Ys + (wr * RW + hh) * BJ + li;
        // fragments of step st + 2 are read while the MFMAs of step st run (see gemm_rowA_kernel: no per-MFMA LDS round trip)
        constexpr int PF = 2, NST = RW / 2;
        float afr[PF + 1], bfr[PF + 1][WN];
        auto fetch = [&](int st, int buf) {
            afr[buf] = xa[(2 * st) * BI];
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) bfr[buf][nt] = yb[(2 * st) * BJ + nt * 32];
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) fetch(q, q);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            if (st + PF < NST) fetch(st + PF, (st + PF) % (PF + 1));
            __builtin_amdgcn_sched_barrier(0);   // keep those reads in front of this step's MFMAs
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) acc[nt] = mfma32(afr[st % (PF + 1)], bfr[st % (PF + 1)][nt], acc[nt]);
        }
    }

    // reduce the WR reduction-waves through LDS, then write the split's partial tile
    float* out = p.part + (long long)split * p.K * p.N;
    if (WR > 1) {
        __syncthreads();
        float* red = smem;  // [WR-1][WI][WN][16][64]
        if (wr > 0) {
#pragma unroll
            for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[((((wr - 1) * WI + wi) * WN + nt) * 16 + e) * 64 + lane] = acc[nt][e];
        }
        __syncthreads();
        if (wr == 0) {
#pragma unroll
            for (int q = 1; q < WR; ++q)
#pragma unroll
                for (int nt = 0; nt < WN; ++nt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[nt][e] += red[((((q - 1) * WI + wi) * WN + nt) * 16 + e) * 64 + lane];
        }
    }
    if (wr == 0) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int n = j0 + nt * 32 + li;
            if (n < p.N) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = i0 + wi * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (k < p.K) out[(long long)k * p.N + n] = acc[nt][e];
                }
            }
        }
    }
}

// Column-tile width in 32-col MFMA tiles.  `row_blocks` = how many blocks the launch has per column tile.
// Among the widths that still give the chip >= 2 blocks per CU, take the one wasting the fewest padded columns (ties ->
// wider: more reuse of the streamed operand per block); if no width reaches that, take the one with the most blocks
// (the 15x20 / 8x10 layers have only 75 / 20 row tiles: a 160-wide tile would leave 180 of 256 CUs idle).
int pick_wn(int n, long long row_blocks) {
    int best = 1;
    long long best_pad = -1, best_blocks = -1;
    bool best_full = false;
    for (int wn = 1; wn <= 5; ++wn) {
        const long long tiles = (n + 32 * wn - 1) / (32 * wn);
        const long long pad = tiles * 32 * wn, blocks = tiles * row_blocks;
        const bool full = blocks >= 512;
        bool take;
        if (best_pad < 0) take = true;
        else if (full != best_full) take = full;
        else if (full) take = pad <= best_pad;
        else take = blocks > best_blocks || (blocks == best_blocks && pad <= best_pad);
        if (take) { best = wn; best_pad = pad; best_blocks = blocks; best_full = full; }
    }
    return best;
}

int rowA_wn(int rows, int cols);
// Backward-data GEMMs with a tiny reduction (<= 48 channels: the project convs of blocks 1-3, the decoder's backbone / logits convs)
// are all epilogue: 128 x 144 outputs per 128 x 24 inputs.  At 4-5 column tiles the float4 epilogue walks 11 rows per thread with
// one load in flight; at <= 3 it takes four rows at a time (gemm_rowA_kernel, EU) -- two 96- / 72-column tiles beat one of 144 / 160
// although the narrow operand is read twice (block-2 project conv 379 -> 277 us).  Only for the register-limited big-M instantiations.
int rowA_wn_bwd(int rows, int cols, int red) {
    const int wn = rowA_wn(rows, cols);
    const char* e = getenv("SSDSEG_ROWA_BWD_WN");         // (A/B runs) "0": no cap
    const int cap = e != nullptr ? atoi(e) : 3;
    return (cap >= 1 && rows >= occ_rows() && red <= 48 && wn > cap) ? cap : wn;
}
int rowA_wn(int rows, int cols) {
    const int wn = pick_wn(cols, cdiv(rows, BM));
    const char* e = getenv("SSDSEG_ROWA_WN_MAX");        // (A/B runs) widest column tile, in 32-column units
    if (e != nullptr && atoi(e) >= 1 && wn > atoi(e)) return atoi(e);
    return wn;
}

#include "gemm_wres.h"
#include "conv3_wgrad.h"
#include "conv3_tile.h"
#include "conv3_wgrad_tile.h"
#include "pw_tile.h"
#include "conv3_wino.h"
#include "conv3_wino4.h"
#include "conv3_wino_wgrad.h"
#include "pw_wgrad.h"

// ---- tile GEMM of the pointwise convs (pw_tile.h): shape -> (waves, column tile), launch
// SSDSEG_PW_TILE: "0" never, "1" every shape the kernel takes, unset: where it measured faster
int pw_tile_mode() {
    const char* e = getenv("SSDSEG_PW_TILE");
    return e == nullptr ? 2 : (e[0] == '0' ? 0 : 1);
}

template <int WAVES, int WN, int MODE, int WNW = 1>
int pwt_launch_inst(ssdseg_ctx* ctx, const PwTArgs& a, dim3 grid, double cost_bytes, double cost_flops) {
    const size_t lds = pwt_lds_floats(32 * WAVES / WNW, 32 * WN * WNW, a.cred) * sizeof(float);
    static size_t configured = 0;
    if (lds > configured) {
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_tile_kernel<WAVES, WN, MODE, WNW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    char kbuf[64];
    if (WNW == 1) snprintf(kbuf, sizeof(kbuf), "pw_tile_kernel<%d, %d, %d>", WAVES, WN, MODE);
    else snprintf(kbuf, sizeof(kbuf), "pw_tile_kernel<%d, %d, %d, %d>", WAVES, WN, MODE, WNW);
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (pw_tile_kernel<WAVES, WN, MODE, WNW>), grid, dim3(64 * WAVES), lds, a);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

// default dispatch rule, read off the per-layer A/B table (profiles/r02_pw_tile_vs_rowA_per_layer.txt)
// (round 2, one MI355X, batch 32).  Forward: every layer with >= 256 output channels (the expand convs of the 30x40 / 15x20
// stages, the ASPP branches and output conv, the decoder sepconv: 0.83-0.95 of the rowA time; project convs with <= 160 outputs
// lose 1.0-1.9x: too few, too narrow tiles) and the 36-column tap-expanded GEMM of the 256 -> 4 logits conv (614,400 rows).
// Input gradient: plain epilogues (no fused BatchNorm sums, no accumulation into an existing gradient: the tile kernel's scalar
// epilogue loses there) with a reduction of <= 384 channels and a wide side (>= 256) somewhere: decoder sepconv 0.76, encoder
// output conv 0.80, expand convs of blocks 7-10 0.84-0.93, ASPP atrous branches 0.93-0.98; long reductions onto few columns
// (expand convs of blocks 11-16: one column tile, 75-300 blocks) lose 1.1-1.4x.
// Round 2, second table (profiles/r02_pw_tile_short_m_per_layer.txt): with 64- / 32-column tiles the tile kernel also wins every
// FORWARD below 65,536 rows (project convs of the 30x40 / 15x20 stages 0.8, the SSD head and extra-feature-map convs 0.45-0.8) and the
// plain input gradients of <= 16k rows (expand convs of blocks 14-16: 0.75-0.85, extra feature maps 0.45).
bool pw_tile_default(int mode, long long rows, int cred, int nout, bool fused_bn, bool accumulate) {
    const bool pinned = getenv("SSDSEG_PWT_SMALL") != nullptr && getenv("SSDSEG_PWT_SMALL")[0] == '0';
    if (mode == 0) return nout >= 256 || (rows >= 500000 && cred >= 256) || (rows < 65536 && !pinned);
    if (!fused_bn && !accumulate && rows <= 16384 && !pinned) return true;
    return !fused_bn && !accumulate && cred <= 384 && (nout >= 256 || cred >= 256);
}

// can the tile kernel take this GEMM?  (reduction a multiple of 8 and at least two steps deep; 32-bit buffer offsets)
bool pw_tile_takes(long long rows, int lda, int cred, int nout) {
    return cred % 8 == 0 && cred >= 64 && nout % 4 == 0 && rows * (long long)lda * 4 < (1LL << 31) && (long long)nout * cred * 4 < (1LL << 31);
}

// Short-M layers (the 30x40 / 15x20 stages and the extra feature maps: 38,400 / 9,600 / 2,560 ... rows at batch 32): 128-row tiles
// give 300 / 75 / 20 row tiles for 256 CUs, so the column tile is what makes the blocks.  Measured per layer
// (profiles/r02_pw_tile_short_m_per_layer.txt; blocks of one or two waves lost everywhere: every block stages its own weight
// tile, and LDS then holds three waves per CU): 64-column tiles, 32 where 64 would pad more than a quarter, 32 for <= 16k rows.
// SSDSEG_PWT_SMALL=<ncols> overrides the width for A/B runs, "0" restores one tile of <= 160 columns.
constexpr int PWT_SHORT_ROWS = 65536;
int pwt_short_ncols(int mode, long long rows, int cred, int nout) {
    const char* e = getenv("SSDSEG_PWT_SMALL");
    if (e != nullptr) return atoi(e) >= 32 ? atoi(e) : 0;
    const int t64 = cdiv(nout, 64);
    const bool pad64 = cdiv((cdiv(nout, t64) + 3) / 4 * 4, 32) * 32 * t64 * 4 > nout * 5;
    if (mode == 0) {
        if (rows <= 16384) return nout >= 512 ? 0 : 32;
        if (cred <= 64) return 128;
        return (nout <= 64 || pad64) ? 32 : 64;
    }
    if (rows <= 16384) return 32;
    return nout <= 64 ? 64 : (pad64 ? 32 : 64);
}

// nparts_y: number of blocks along y == number of partial rows the caller sized its statistics table for (0: free choice)
template <int MODE>
int pw_tile_launch(ssdseg_ctx* ctx, PwTArgs a, int nparts_y, double view_bytes) {
    int ntiles = a.nout <= 160 ? 1 : cdiv(a.nout, 256);
    if (a.M < PWT_SHORT_ROWS) {
        const int nc = pwt_short_ncols(MODE, a.M, a.cred, a.nout);
        if (nc > 0) ntiles = cdiv(a.nout, nc);
    }
    a.ncols = (cdiv(a.nout, ntiles) + 3) / 4 * 4;
    int wn = cdiv(a.ncols, 32);
    const char* bige = getenv("SSDSEG_PW_TILE_BIG_ROWS");      // (A/B runs) rows from which the eight-wave 256-row blocks are used
    const bool big = a.M >= (bige != nullptr ? atoll(bige) : 65536);                       // >= 256 row tiles of 256 rows: eight-wave blocks, one per CU
    // column tiles of <= 160: four-wave blocks (two blocks per CU), and the input-gradient kernel whatever its size (its gradient
    // view staging and epilogue need ~60 registers more: 6- and 8-tile instantiations spill)
    if ((!big || MODE == 1) && wn > 5) {
        const int nt2 = cdiv(a.nout, 160);
        a.ncols = (cdiv(a.nout, nt2) + 3) / 4 * 4;
        wn = cdiv(a.ncols, 32);
    }
    // input gradient with 161 .. 256 output columns at >= 65,536 rows (the decoder sepconv): a 4 x 2 wave grid over a 128-row block
    // that spans ALL columns -- the two-tensor operand is streamed once, not once per 128-column tile (SSDSEG_PWT_GRID=0: off)
    const char* ge = getenv("SSDSEG_PWT_GRID");
    const bool grid42 = MODE == 1 && big && a.nout > 160 && a.nout <= 256 && !(ge != nullptr && ge[0] == '0');
    if (grid42) { a.ncols = (a.nout + 3) / 4 * 4; wn = cdiv(a.ncols, 64); }
    const int gx = cdiv(a.nout, a.ncols);
    const int bm = grid42 ? 128 : (big ? 256 : 128);
    const int mtiles = cdiv(a.M, bm);
    const char* pe = getenv("SSDSEG_PWT_PARTS");       // (A/B runs) blocks of a launch without a caller-sized statistics table
    const int cap = (pe != nullptr && atoi(pe) >= 64 ? atoi(pe) : 1024 * gx) / gx;
    int gy = nparts_y > 0 ? nparts_y : (mtiles < cap ? mtiles : cap);
    if (gy > mtiles && nparts_y == 0) gy = mtiles;
    if (gy < 1) gy = 1;
    a.a_bytes = (unsigned)((((long long)a.M - 1) * a.lda + a.cred) * 4);
    a.wt_bytes = (unsigned)((long long)a.nout * a.cred * 4);
    const dim3 grid(gx, gy, 1);
    const double cost_bytes = 4.0 * ((double)a.M * a.cred + (double)a.M * a.nout + (double)a.cred * a.nout);
    const double cost_flops = 2.0 * a.M * a.cred * a.nout;
    ctx->timing_view_bytes = view_bytes;
#define PWT_CASE(W, N) return pwt_launch_inst<W, N, MODE>(ctx, a, grid, cost_bytes, cost_flops)
    if constexpr (MODE == 1) {
        if (grid42) {
            if (wn <= 3) return pwt_launch_inst<8, 3, MODE, 2>(ctx, a, grid, cost_bytes, cost_flops);
            return pwt_launch_inst<8, 4, MODE, 2>(ctx, a, grid, cost_bytes, cost_flops);
        }
    }
    if (big) {
        if (wn <= 2) PWT_CASE(8, 2);
        if (wn <= 4) PWT_CASE(8, 4);
        if (wn == 5) PWT_CASE(8, 5);
        if constexpr (MODE == 0) {
            if (wn == 6) PWT_CASE(8, 6);
            PWT_CASE(8, 8);
        }
    }
    if (wn <= 1) PWT_CASE(4, 1);
    if (wn <= 2) PWT_CASE(4, 2);
    if (wn <= 3) PWT_CASE(4, 3);
    if (wn <= 4) PWT_CASE(4, 4);
    PWT_CASE(4, 5);
#undef PWT_CASE
}

// 1: the resident kernels where they measured faster (default); 0: SSDSEG_NO_WRES=1, general kernels everywhere (A/B
// measurements); 2: SSDSEG_WRES_FORCE=1, resident kernels for every shape that fits (the parity tests run that way)
int wres_mode() {
    // (read on every call, not cached: the parity tests flip these between calls)
    const int mode = (getenv("SSDSEG_NO_WRES") != nullptr && getenv("SSDSEG_NO_WRES")[0] == '1')
                                ? 0
                                : ((getenv("SSDSEG_WRES_FORCE") != nullptr && getenv("SSDSEG_WRES_FORCE")[0] == '1') ? 2 : 1);
    return mode;
}
bool wres_enabled() { return wres_mode() != 0; }

// row-tile slots per column tile: enough blocks to fill the chip (~8 per CU), few enough that the BN partial
// table stays short
int rowA_grid_y_wn(int rows, int cols, int wn);
int rowA_grid_y(int rows, int cols) { return rowA_grid_y_wn(rows, cols, rowA_wn(rows, cols)); }
int rowA_grid_y_wn(int rows, int cols, int wn) {
    const int ntiles = cdiv(cols, 32 * wn), mtiles = cdiv(rows, BM);
    const char* pe = getenv("SSDSEG_ROWA_PARTS");      // (A/B runs) cap of row-tile slots x column tiles
    // two blocks per CU for the 2-5 tile instantiations; the one-tile ones are compiled for four (three) blocks per CU
    // (rowa_min_waves) and want them: the stem conv ran 225 us on 2048 blocks, 316 us on 512
    const int cap = pe != nullptr && atoi(pe) >= 64 ? atoi(pe) : (wn == 1 ? 1024 : 512);
    // few tiles (the short-M stages): one row tile per block -- 75 row tiles dealt to 51 blocks is 2 rounds instead of 1
    if ((long long)mtiles * ntiles <= 2 * cap) return mtiles;
    int gy = cap / ntiles;
    if (gy < 1) gy = 1;
    if (mtiles <= gy) return mtiles;
    const int per = cdiv(mtiles, gy);       // row tiles per block, then as few blocks as carry them: an even deal
    return cdiv(mtiles, per);
}

// wt_pre (forward only, may be nullptr): the weights already transposed to [J][R] by ssdseg_transpose_batch
template <int MODE, int LD>
int launch_rowA(ssdseg_ctx* ctx, const RowAArgs& a0, const float* wt_pre = nullptr) {
    RowAArgs a = a0;
    int wn = (MODE == 1 && LD == 0) ? rowA_wn_bwd(a.I, a.J, a.R) : rowA_wn(a.I, a.J);
    const int nparts = rowA_grid_y(a.I, a.J);   // BN-statistics partial rows the caller allocated: fixed by (I, J) alone
    if (LD == 0 && pw_tile_mode() != 0 && pw_tile_takes(a.I, a.lda, a.R, a.J) && !(MODE == 0 && (a.residual != nullptr || a.accumulate != 0)) && (pw_tile_mode() == 1 || pw_tile_default(MODE, a.I, a.R, a.J, false, a.accumulate != 0 || a.residual != nullptr))) {
        PwTArgs t{};
        t.a0 = a.a0; t.a1 = (MODE == 1 && a.cs != nullptr) ? a.a1 : a.a0;
        t.cs = a.cs; t.ct = a.ct; t.ck1 = a.ck1; t.ck0 = a.ck0; t.act = a.act; t.lda = a.lda;
        t.out = a.out; t.ldo = a.ldo; t.residual = a.residual; t.ldr = a.ldr; t.accumulate = a.accumulate;
        t.stats = a.stats; t.M = a.I; t.cred = a.R; t.nout = a.J;
        t.wt = a.b;
        if (MODE == 0 && wt_pre != nullptr) {
            t.wt = wt_pre;
        } else if (MODE == 0) {
            // the reduction channel must be contiguous in the staged weight rows: W[k][n] -> Wt[n][k] (one small transpose per call)
            void* ws;
            int rc = ssdseg_workspace(ctx, (size_t)a.R * a.J * sizeof(float), &ws);
            if (rc) return rc;
            SSDSEG_LAUNCH(ctx, 8.0 * a.R * a.J, 0.0, conv3_transpose_w_kernel, dim3(cdiv(a.J, 32), cdiv(a.R, 32), 1), dim3(256), 0, a.b, (float*)ws, a.R, a.J);
            SSDSEG_LAUNCH_CHECK();
            t.wt = (const float*)ws;
        }
        return pw_tile_launch<MODE>(ctx, t, a.stats != nullptr ? nparts : 0, (MODE == 1 && a.cs != nullptr) ? 4.0 * a.I * a.R : 0.0);
    }
    // Backward-data of the 30x40-stage expand convs (38400 rows = 300 row tiles, 384-576 reduction channels of a TWO-tensor
    // gradient view, 64-96 output columns): the default picks 32-column tiles to have 600-900 blocks, and every column tile
    // re-reads the whole (g, y) operand from the fabric (measured 5 TB/s of L2-level reads for 1.4 TB/s algorithmic).  One
    // column tile spanning all outputs reads it once; with >= 256 row tiles there is still a block per CU (109 -> 88 us).
    // Not for the 15x20 stage (75 row tiles: 2x slower) and no gain for the forward (single-tensor operand).
    int grid_y = (MODE == 1 && LD == 0) ? rowA_grid_y_wn(a.I, a.J, wn) : nparts;   // (no statistics table in backward-data: free choice)
    if (MODE == 1 && LD == 0 && a.cs != nullptr && a.R >= 256) {
        const int wide = cdiv(a.J, 32), mtiles = cdiv(a.I, BM);
        if (wide > wn && wide <= 3 && mtiles >= 256 && a.I < ROWA_OCC_ROWS) { wn = wide; grid_y = mtiles < 2048 ? mtiles : 2048; }
    }
    dim3 grid(cdiv(a.J, 32 * wn), grid_y);
    size_t lds = (size_t)(BM * AS + BK * (32 * wn + 1)) * sizeof(float);
    size_t red = (size_t)(4 * 2 * 32 * wn) * sizeof(float);
    if (red > lds) lds = red;
    // algorithmic: read the streamed operand and the weights once, write the output once.  The dense 3x3 conv
    // (LD == 1) streams its input ONCE per 8(d) -- X + Y + W, not the nine-fold im2col operand the implicit GEMM walks (a.R =
    // 9 * channels).  The second tensor of a BatchNorm-backward gradient view (raw y next to g) is reported as `view_bytes`.
    const double red_once = LD == 1 ? (double)a.convC : (double)a.R;
    const double cost_bytes = 4.0 * ((double)a.I * red_once + (double)a.I * a.J + (double)a.R * a.J);
    const double cost_flops = 2.0 * a.I * a.R * a.J;
    const double view_bytes = (MODE == 1 && a.cs != nullptr) ? 4.0 * a.I * red_once : 0.0;
    ctx->timing_view_bytes = view_bytes;
    const bool occ = a.I >= occ_rows();
    char kbuf[64];
    // Measured per layer on MI355X (profiles/r01_wres_vs_general_per_layer.txt): the resident kernel wins when every wave
    // streams several row tiles (>= ~500k rows: the 240x320 and 120x160 stages at batch 32) and loses 10-30 % below that,
    // where its once-per-block weight load is not amortised; its 5-tile backward variant needs > 256 registers.
    if (LD == 0 && wres_lds_bytes(a.R, wn) > 0 && (wres_mode() == 2 || (wres_mode() == 1 && a.I >= 500000 && !(MODE == 1 && wn >= 4)))) {
        // whole weight slab resident in LDS, barrier-free per-wave streaming (gemm_wres.h); same grid, same partial rows
        const size_t wl = wres_lds_bytes(a.R, wn);
        snprintf(kbuf, sizeof(kbuf), "gemm_wres_kernel<%d, %d, 0>", wn, MODE);
        const char* wname = ctx->timing ? ssdseg_intern(kbuf) : "";
        constexpr int WM = LD == 0 ? MODE : 0;   // (only instantiated for LD == 0)
        switch (wn) {
            case 1: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<1, WM, 0>), grid, dim3(256), wl, a); break;
            case 2: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<2, WM, 0>), grid, dim3(256), wl, a); break;
            case 3: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<3, WM, 0>), grid, dim3(256), wl, a); break;
            case 4: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<4, WM, 0>), grid, dim3(256), wl, a); break;
            default: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<5, WM, 0>), grid, dim3(256), wl, a); break;
        }
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    // wide, 16-byte-aligned outputs leave through the LDS-transposed float4 epilogue (16 B per lane instead of 4)
    const char* f4env = getenv("SSDSEG_NO_F4_EPILOGUE");
    if (LD == 0 && wn >= 2 && !(f4env != nullptr && f4env[0] == '1') && a.J % 4 == 0 && a.ldo % 4 == 0 &&
        ((uintptr_t)a.out & 15) == 0 && (a.residual == nullptr || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 15) == 0))) {
        const size_t cs = (size_t)64 * (32 * wn + 4) * sizeof(float);
        if (cs > lds) lds = cs;
        snprintf(kbuf, sizeof(kbuf), "gemm_rowA_kernel<%d, %d, 0, 0, 1, %d>", wn, MODE, (int)occ);
        const char* fname = ctx->timing ? ssdseg_intern(kbuf) : "";
        constexpr int FM = LD == 0 ? MODE : 0;   // (only instantiated for LD == 0)
        switch (wn) {
            case 2:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<2, FM, 0, 0, 1, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<2, FM, 0, 0, 1, 0>), grid, dim3(256), lds, a);
            break;
            case 3:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<3, FM, 0, 0, 1, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<3, FM, 0, 0, 1, 0>), grid, dim3(256), lds, a);
            break;
            case 4:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<4, FM, 0, 0, 1, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<4, FM, 0, 0, 1, 0>), grid, dim3(256), lds, a);
            break;
            default:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<5, FM, 0, 0, 1, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, fname, cost_bytes, cost_flops, (gemm_rowA_kernel<5, FM, 0, 0, 1, 0>), grid, dim3(256), lds, a);
            break;
        }
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    snprintf(kbuf, sizeof(kbuf), "gemm_rowA_kernel<%d, %d, %d, 0, 0, %d>", wn, MODE, LD, (int)occ);   // = the symbol rocprofv3 shows
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    switch (wn) {
        case 1:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, MODE, LD, 0, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, MODE, LD, 0, 0, 0>), grid, dim3(256), lds, a);
            break;
        case 2:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<2, MODE, LD, 0, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<2, MODE, LD, 0, 0, 0>), grid, dim3(256), lds, a);
            break;
        case 3:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<3, MODE, LD, 0, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<3, MODE, LD, 0, 0, 0>), grid, dim3(256), lds, a);
            break;
        case 4:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<4, MODE, LD, 0, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<4, MODE, LD, 0, 0, 0>), grid, dim3(256), lds, a);
            break;
        default:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<5, MODE, LD, 0, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<5, MODE, LD, 0, 0, 0>), grid, dim3(256), lds, a);
            break;
    }
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

template <int WI, int WR>
int launch_wgrad_wn(ssdseg_ctx* ctx, const WGradArgs& a, int wn, dim3 grid) {
    size_t lds = ((size_t)(rw_of(WR) * WR) * (32 * WI + 32 * wn) + 2 * 32 * WI + 4 * 32 * wn) * sizeof(float);   // tiles + view coefficients
    size_t red = (size_t)(WR - 1) * WI * wn * 16 * 64 * sizeof(float);
    if (red > lds) lds = red;
    const double share = 1.0 / ((double)grid.x * grid.y);   // every (k-tile, n-tile) block column re-reads its operands
    const double cost_bytes = 4.0 * ((double)a.M * a.K + (double)a.M * a.N + (double)a.K * a.N);   // 8(d): read X, read dY, write dW
    const double cost_flops = 2.0 * a.M * a.K * a.N;
    ctx->timing_view_bytes = a.gs != nullptr ? 4.0 * a.M * a.N : 0.0;
    (void)share;
    char kbuf[64];
    snprintf(kbuf, sizeof(kbuf), "gemm_wgrad_kernel<%d, %d, %d>%s", WI, WR, wn, a.convH > 0 ? " [conv3x3 tap]" : "");
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    switch (wn) {
        case 1: SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_wgrad_kernel<WI, WR, 1>), grid, dim3(256), lds, a); break;
        case 2: SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_wgrad_kernel<WI, WR, 2>), grid, dim3(256), lds, a); break;
        case 3: SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_wgrad_kernel<WI, WR, 3>), grid, dim3(256), lds, a); break;
        case 4: SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_wgrad_kernel<WI, WR, 4>), grid, dim3(256), lds, a); break;
        default: SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_wgrad_kernel<WI, WR, 5>), grid, dim3(256), lds, a); break;
    }
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

// partial slabs of a split weight gradient (splits * K * N floats, written once and read once by the column sum) are kept below this
// fraction of the layer's operand traffic M * (K + N)  (SSDSEG_WGRAD_SLAB_FRAC, read per call: A/B runs)
double slab_fraction() {
    const char* e = getenv("SSDSEG_WGRAD_SLAB_FRAC");
    const double f = e != nullptr ? atof(e) : 0.5;
    return f > 0.0 ? f : 0.5;
}

// ---- pointwise weight gradient, row-naming form (pw_wgrad.h).  SSDSEG_PW_WGRAD=0: gemm_wgrad_kernel for everything.
template <int JX, int JY, int WK, int WN>
int pw_wgrad_launch(ssdseg_ctx* ctx, PwWgArgs a, float* dw) {
    constexpr int G = 8 / (WK * WN), KT = 32 * JX * WK, NT = 32 * JY * WN, MS = pww_ms(KT, NT);
    const int ktiles = cdiv(a.K, KT), ntiles = cdiv(a.N, NT);
    const long long steps = ((long long)a.M + MS - 1) / MS;
    // blocks: two per CU; every split >= 4 steps; partial slabs (splits * K * N, written and re-read) below half the operand traffic
    long long splits = (2LL * ctx->num_cus + (long long)ktiles * ntiles - 1) / ((long long)ktiles * ntiles);
    const long long cap_steps = (steps + 3) / 4;
    const long long cap_traffic = (long long)((double)a.M * (a.K + a.N) * slab_fraction() / ((double)a.K * a.N));
    if (splits > cap_steps) splits = cap_steps;
    if (splits > cap_traffic) splits = cap_traffic;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const long long sps = (steps + splits - 1) / splits;
    splits = (steps + sps - 1) / sps;
    a.rows_per_split = (int)(sps * MS);
    const long long slabs = splits;
    const size_t pb = (size_t)slabs * a.K * a.N * sizeof(float);
    if (pb >= ((size_t)1 << 31)) return -1;
    float* part = dw;
    if (slabs > 1) {
        void* ws;
        int rc = ssdseg_partials(ctx, pb, &ws);
        if (rc) return rc;
        part = (float*)ws;
    }
    a.part = part;
    a.part_bytes = (unsigned)pb;
    const size_t lds = pww_lds_bytes(KT, NT, G);
    static bool configured = false;      // (per instantiation) dynamic LDS beyond 64 KiB has to be announced once
    if (lds > 64 * 1024 && !configured) {
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_kernel<JX, JY, WK, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const double cost_bytes = 4.0 * ((double)a.M * a.K + (double)a.M * a.N + (double)a.K * a.N);   // 8(d): read X, read dY, write dW
    const double cost_flops = 2.0 * a.M * a.K * a.N;
    ctx->timing_view_bytes = a.gs != nullptr ? 4.0 * a.M * a.N : 0.0;
    char kbuf[64];
    snprintf(kbuf, sizeof(kbuf), "pw_wgrad_kernel<%d, %d, %d, %d>", JX, JY, WK, WN);
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (pw_wgrad_kernel<JX, JY, WK, WN>), dim3(ntiles, ktiles, (unsigned)splits), dim3(512), lds, a);
    SSDSEG_LAUNCH_CHECK();
    if (slabs > 1) return ssdseg_colsum(ctx, part, (int)slabs, (long long)a.K * a.N, dw);
    return 0;
}

bool pw_wgrad_enabled() {
    const char* e = getenv("SSDSEG_PW_WGRAD");
    return !(e != nullptr && e[0] == '0');
}

// -> 0 launched, < 0 not taken (the caller falls back), > 0 error
int pw_wgrad_try(ssdseg_ctx* ctx, const WGradArgs& w, float* dw) {
    if (!pw_wgrad_enabled() || w.stem || w.convH > 0 || w.K % 4 != 0 || w.N % 4 != 0 || w.ldx % 4 != 0 || w.ldy % 4 != 0) return -1;
    // the pipelined issue() prefetches one step (<= 64 rows) past the end of a split and forms m0 * ld * 4 in 32 bits
    if (((long long)w.M + 64) * w.ldx * 4 >= (1LL << 31) || ((long long)w.M + 64) * w.ldy * 4 >= (1LL << 31)) return -1;
    PwWgArgs a{};
    a.x = w.x; a.xs = w.xs; a.xt = w.xt; a.xact = w.xact; a.ldx = w.ldx;
    a.g = w.g; a.y = w.y; a.gs = w.gs; a.gt = w.gt; a.gk1 = w.gk1; a.gk0 = w.gk0; a.gact = w.gact; a.ldy = w.ldy;
    a.M = w.M; a.K = w.K; a.N = w.N;
    a.x_bytes = (unsigned)((((long long)w.M - 1) * w.ldx + w.K) * 4);
    a.g_bytes = (unsigned)((((long long)w.M - 1) * w.ldy + w.N) * 4);
    // tile = (32 JX WK) x (32 JY WN), chosen per layer among the instantiated shapes (K = 160 on a 256-row tile wastes 37 % of the
    // MFMAs, three 64-row tiles 17 %; a 24 -> 144 layer on 64-column tiles reads x three times)
    struct Shape { int kt, nt; int (*launch)(ssdseg_ctx*, PwWgArgs, float*); };
    static const Shape shapes[] = {
        {256, 128, &pw_wgrad_launch<2, 2, 4, 2>}, {128, 256, &pw_wgrad_launch<2, 2, 2, 4>}, {128, 128, &pw_wgrad_launch<2, 2, 2, 2>},
        {256, 64, &pw_wgrad_launch<2, 2, 4, 1>},  {64, 128, &pw_wgrad_launch<2, 2, 1, 2>},  {128, 64, &pw_wgrad_launch<2, 2, 2, 1>},
        {256, 32, &pw_wgrad_launch<4, 1, 2, 1>},  {64, 64, &pw_wgrad_launch<2, 2, 1, 1>},   {32, 128, &pw_wgrad_launch<1, 4, 1, 1>},
        {128, 32, &pw_wgrad_launch<4, 1, 1, 1>},  {32, 64, &pw_wgrad_launch<1, 2, 1, 1>},   {64, 32, &pw_wgrad_launch<2, 1, 1, 1>},
        {32, 32, &pw_wgrad_launch<1, 1, 1, 1>}};
    // estimated time of a shape = max(padded MFMA work at ~110 TFLOP/s, operand traffic at ~4.5 TB/s): every n-tile re-reads the x
    // columns of its k-tile and vice versa, so small tiles cost traffic and large ones padding
    const Shape* best = nullptr;
    double best_t = 0.0;
    for (const Shape& sh : shapes) {
        const double tk = cdiv(w.K, sh.kt), tn = cdiv(w.N, sh.nt);
        const double t_mfma = 2.0 * w.M * (tk * sh.kt) * (tn * sh.nt) / 110e12;
        const double t_hbm = 4.0 * w.M * ((double)w.K * tn + (double)w.N * tk) / 4.5e12;
        const double t = t_mfma > t_hbm ? t_mfma : t_hbm;
        if (best == nullptr || t < best_t * 0.999) { best = &sh; best_t = t; }      // (listed largest first: ties keep the larger tile)
    }
    return best->launch(ctx, a, dw);
}

// picks the tile shape / split count for dw[k][n] = sum_m x[m][k]*dy[m][n], launches, reduces the split partials
int wgrad_run(ssdseg_ctx* ctx, WGradArgs a, float* dw) {
    {
        const int rc = pw_wgrad_try(ctx, a, dw);
        if (rc >= 0) return rc;
    }
    const int m = a.M, k = a.K, n = a.N;
    const int wi = k <= 32 ? 1 : (k <= 64 ? 2 : 4);
    const int wr = 4 / wi;
    const int brt = rw_of(wr) * wr;
    const int itiles = cdiv(k, 32 * wi);
    long long steps = ((long long)m + brt - 1) / brt;
    // split the reduction rows so that (a) the chip is full, (b) every block still does >= 4 steps and (c) the partial
    // slabs (splits*k*n floats, written then re-read) stay below half of the operand traffic m*(k+n)
    long long max_splits = (steps + 3) / 4;
    const long long traffic_cap = (long long)((double)m * (k + n) * slab_fraction() / ((double)k * n));
    if (max_splits > traffic_cap) max_splits = traffic_cap < 1 ? 1 : traffic_cap;
    int wn = pick_wn(n, (long long)itiles * max_splits);
    // the 4-way row-split shape stages 64 x (32*wn) of (g, y) per step: beyond 3 column tiles it needs > 256 VGPRs (1 wave/SIMD)
    if (wi == 1 && wn > 3) wn = 3;
    const int jtiles = cdiv(n, 32 * wn);
    // target blocks per CU: the long-M (HBM-bound) layers want more, shorter splits in flight; the short-M ones fewer, longer
    // splits (half the partial slabs, prologue / epilogue amortised over more steps)
    const char* el = getenv("SSDSEG_WGRAD_BPC_LONG");   // (read per call: tests / A-B runs flip them inside one process)
    const char* es = getenv("SSDSEG_WGRAD_BPC");
    const long long bpc_long = el != nullptr ? atoll(el) : 4;
    const long long bpc_short = es != nullptr ? atoll(es) : 2;
    const long long bpc = m >= ROWA_OCC_ROWS ? bpc_long : bpc_short;
    long long want = (bpc * ctx->num_cus + (long long)itiles * jtiles - 1) / ((long long)itiles * jtiles);
    long long splits = want < 1 ? 1 : (want > max_splits ? max_splits : want);
    if (splits > 65535) splits = 65535;
    long long steps_per_split = (steps + splits - 1) / splits;
    splits = (steps + steps_per_split - 1) / steps_per_split;
    a.rows_per_split = (int)(steps_per_split * brt);
    float* part = dw;
    if (splits > 1) {
        void* ws;
        int rc = ssdseg_partials(ctx, (size_t)splits * k * n * sizeof(float), &ws);
        if (rc) return rc;
        part = (float*)ws;
    }
    a.part = part;
    dim3 grid(jtiles, itiles, (unsigned)splits);
    int rc;
    if (wi == 1) rc = launch_wgrad_wn<1, 4>(ctx, a, wn, grid);
    else if (wi == 2) rc = launch_wgrad_wn<2, 2>(ctx, a, wn, grid);
    else rc = launch_wgrad_wn<4, 1>(ctx, a, wn, grid);
    if (rc) return rc;
    if (splits > 1) return ssdseg_colsum(ctx, part, (int)splits, (long long)k * n, dw);
    return 0;
}


// ------------------------------------------------------------------------------------------------ narrow 3x3 conv
// Dense 3x3 conv with very few output channels (the 256 -> 4 mask-logits conv at 120x160: 75 % of the implicit-GEMM tile
// is padding and every input pixel is gathered nine times: 1.2 / 1.4 / 0.35 ms for fwd / dW / dx).  Rewritten over
// TAP-EXPANDED columns, everything heavy becomes a pointwise GEMM that reads the wide tensor exactly once:
//   fwd : z[m][tap*co + o] = sum_c a[m][c] W[tap][c][o]   (GEMM, N = 9*co)     y[m][o] = sum_tap z[m + d(tap)][tap*co + o]
//   bwd : dz[m][tap*co + o] = dy[m - d(tap)][o]            (shifted copy)       dx = dz * W2^T,  dW2 = a^T * dz  (GEMMs)
// with d(tap) = (kh - 1, kw - 1) and W2[c][tap*co + o] = W[tap][c][o].
constexpr int C3N_MAX_COUT = 8;
__device__ __forceinline__ void add4(float4& acc, float4 a) { acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }

bool conv3_narrow(int cin, int cout) {
    const char* e = getenv("SSDSEG_CONV3_NARROW");   // "0": the implicit-GEMM kernels (A/B measurements, parity tests)
    return !(e != nullptr && e[0] == '0') && cout <= C3N_MAX_COUT && cin >= 4 * cout;
}

__global__ void conv3n_pack_w_kernel(const float* __restrict__ w, float* __restrict__ w2, int cin, int cout, int reverse) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over [9][cin][cout]
    if (i >= 9 * cin * cout) return;
    const int o = i % cout, c = (i / cout) % cin, tap = i / (cout * cin);
    const int j = c * 9 * cout + tap * cout + o;
    if (reverse) const_cast<float*>(w)[i] = w2[j];   // dW2 -> dW
    else w2[j] = w[i];
}

// y[m][o] = sum_tap z[m + d(tap)][tap*co + o]; optional BN statistics: one partial row (sum, sumsq per channel) per block
__global__ void __launch_bounds__(256) conv3n_tapsum_kernel(const float* __restrict__ z, float* __restrict__ y, int n, int h, int w, int cv,
                                                            float* __restrict__ stats) {
    __shared__ float4 red[2][256];
    // (32-bit index arithmetic: the launcher guarantees n*h*w*9*cv < 2^31 -- 64-bit divisions cost more than the kernel's traffic)
    const int total = n * h * w * cv;
    const int ldz = 9 * cv * 4;
    float4 ssum = f4(0.f), ssq = f4(0.f);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c4 = i % cv;
        int r = i / cv;
        const int x = r % w; r /= w;
        const int yy = r % h;
        const long long img = r / h;
        float4 acc = f4(0.f);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int sy = yy + tap / 3 - 1, sx = x + tap % 3 - 1;
            if (sy >= 0 && sy < h && sx >= 0 && sx < w) add4(acc, ld4(z + ((img * h + sy) * w + sx) * ldz + (tap * cv + c4) * 4));
        }
        st4(y + (long long)i * 4, acc);
        add4(ssum, acc);
        ssq.x = fmaf(acc.x, acc.x, ssq.x); ssq.y = fmaf(acc.y, acc.y, ssq.y); ssq.z = fmaf(acc.z, acc.z, ssq.z); ssq.w = fmaf(acc.w, acc.w, ssq.w);
    }
    if (stats == nullptr) return;
    red[0][threadIdx.x] = ssum; red[1][threadIdx.x] = ssq;   // thread t always has channel vector t % cv (256 % cv == 0)
    __syncthreads();
    for (int off = 128; off >= cv; off >>= 1) {
        if ((int)threadIdx.x < off) { add4(red[0][threadIdx.x], red[0][threadIdx.x + off]); add4(red[1][threadIdx.x], red[1][threadIdx.x + off]); }
        __syncthreads();
    }
    if ((int)threadIdx.x < cv) {
        st4(stats + ((long long)blockIdx.x * 2 + 0) * cv * 4 + threadIdx.x * 4, red[0][threadIdx.x]);
        st4(stats + ((long long)blockIdx.x * 2 + 1) * cv * 4 + threadIdx.x * 4, red[1][threadIdx.x]);
    }
}

// dz[m][tap*co + o] = dy[m - d(tap)][o] (0 outside the image), dy formed from the gradient view on the way
__global__ void __launch_bounds__(256) conv3n_shift_kernel(const float* __restrict__ g, const float* __restrict__ yv, const float* __restrict__ gs,
                                                           const float* __restrict__ gt, const float* __restrict__ gk1,
                                                           const float* __restrict__ gk0, int gact, float* __restrict__ dz, int n, int h, int w,
                                                           int cv) {
    const int total = n * h * w * 9 * cv;      // < 2^31 (launcher)
    const bool aff = gs != nullptr;
    const float* yp = aff ? yv : g;
    const int act = aff ? gact : SSDSEG_ACT_NONE;
    float4 s = f4(1.f), t = f4(0.f), k1 = f4(0.f), k0 = f4(0.f);
    const int cfix = threadIdx.x % cv;           // 256 % cv == 0 and the grid stride is a multiple of 256: a thread keeps its channel vector
    if (aff) { s = ld4(gs + cfix * 4); t = ld4(gt + cfix * 4); k1 = ld4(gk1 + cfix * 4); k0 = ld4(gk0 + cfix * 4); }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c4 = cfix;
        int r = i / cv;
        const int tap = r % 9; r /= 9;
        const int x = r % w; r /= w;
        const int yy = r % h;
        const long long img = r / h;
        const int sy = yy - (tap / 3 - 1), sx = x - (tap % 3 - 1);
        float4 v = f4(0.f);
        if (sy >= 0 && sy < h && sx >= 0 && sx < w) {
            const long long o = (((img * h + sy) * w + sx) * cv + c4) * 4;
            v = gview_apply4(ld4(g + o), ld4(yp + o), s, t, k1, k0, act);
        }
        st4(dz + (long long)i * 4, v);
    }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// scratch for a composite call: `bytes` for the caller plus head-room for the nested launches, which then allocate BEHIND it
// (ssdseg_workspace honours ctx->ws_reserved); returns the caller's region
int conv3n_scratch(ssdseg_ctx* ctx, size_t bytes, void** out) {
    return ssdseg_workspace(ctx, bytes + bytes / 2 + ((size_t)64 << 20), out);
}


// ---- halo-tile 3x3 conv (conv3_tile.h): geometry and launch
bool conv3_tile_enabled() {
    const char* e = getenv("SSDSEG_CONV3_TILE");   // "0": the implicit-GEMM kernels (A/B measurements, parity tests)
    return !(e != nullptr && e[0] == '0');
}
bool conv3_tile_fwd_ok(int cin, int cout) { return conv3_tile_enabled() && !conv3_narrow(cin, cout) && cin % C3T_KC == 0; }
// 32-bit buffer offsets: the streamed tensor (row stride ld) has to stay below 2^31 bytes
bool conv3_tile_fits(int n, int h, int w, int ld) { return (long long)n * h * w * ld * 4 < (1LL << 31); }

struct Conv3TGeom {
    int tiles_h, tiles_w, mtiles, ntiles_n, ncols, wn;
};
Conv3TGeom conv3t_geometry(int n, int h, int w, int nout) {
    Conv3TGeom g;
    g.tiles_h = cdiv(h, C3T_ROWS);
    g.tiles_w = cdiv(w, C3T_COLS);
    g.mtiles = n * g.tiles_h * g.tiles_w;
    g.ntiles_n = cdiv(nout, 160);
    g.ncols = (cdiv(nout, g.ntiles_n) + 3) / 4 * 4;
    g.wn = cdiv(g.ncols, 32);
    return g;
}

template <int WN>
int conv3t_launch_wn(ssdseg_ctx* ctx, const Conv3TArgs& a, const Conv3TGeom& g, double cost_bytes, double cost_flops) {
    const size_t lds = conv3t_lds_floats(WN, a.cred) * sizeof(float);
    static size_t configured = 0;   // dynamic LDS beyond 64 KiB has to be announced once per kernel
    if (lds > configured) {
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_tile_kernel<WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    char kbuf[64];
    snprintf(kbuf, sizeof(kbuf), "conv3_tile_kernel<%d>%s", WN, a.flip ? " [bwd_data]" : " [fwd]");
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (conv3_tile_kernel<WN>), dim3((unsigned)(g.mtiles * g.ntiles_n)), dim3(C3T_THREADS), lds, a);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int conv3t_launch(ssdseg_ctx* ctx, Conv3TArgs a) {
    const Conv3TGeom g = conv3t_geometry(a.n, a.h, a.w, a.nout);
    a.tiles_h = g.tiles_h; a.tiles_w = g.tiles_w; a.ntiles_n = g.ntiles_n; a.ncols = g.ncols;
    a.in_bytes = (unsigned)((((long long)a.n * a.h * a.w - 1) * a.ldi + a.cred) * 4);
    a.wt_bytes = (unsigned)((long long)9 * a.nout * a.cred * 4);
    const double m = (double)a.n * a.h * a.w;
    const double cost_bytes = 4.0 * (m * a.cred + m * a.nout + 9.0 * a.cred * a.nout);   // X + Y + W
    const double cost_flops = 18.0 * m * a.cred * a.nout;
    switch (g.wn) {
        case 1: return conv3t_launch_wn<1>(ctx, a, g, cost_bytes, cost_flops);
        case 2: return conv3t_launch_wn<2>(ctx, a, g, cost_bytes, cost_flops);
        case 3: return conv3t_launch_wn<3>(ctx, a, g, cost_bytes, cost_flops);
        case 4: return conv3t_launch_wn<4>(ctx, a, g, cost_bytes, cost_flops);
        default: return conv3t_launch_wn<5>(ctx, a, g, cost_bytes, cost_flops);
    }
}

// ---- Winograd F(2x2, 3x3) form (conv3_wino.h).  SSDSEG_CONV3_WINOGRAD=0: the direct halo-tile kernels.
int conv3_wino_mode() {      // 0 off, 1 forced (any size; parity tests), 2 automatic
    const char* e = getenv("SSDSEG_CONV3_WINOGRAD");
    if (!conv3_tile_enabled()) return 0;
    return e == nullptr || e[0] == '\0' ? 2 : (e[0] == '0' ? 0 : 1);
}
// automatic: where the 16 transformed GEMMs fill the chip -- >= 64 output channels, a few hundred pixel tiles
bool conv3_wino_takes(int n, int h, int w, int cred, int nout) {
    const int mode = conv3_wino_mode();
    if (mode == 0 || cred % C3T_KC != 0 || wino_lds_floats(cred) * sizeof(float) > (size_t)160 * 1024) return false;
    return mode == 1 || (nout >= 64 && (long long)n * cdiv(h, C3T_ROWS) * cdiv(w, C3T_COLS) >= 256);
}

// a: in / view / out / stats / shape as for conv3t_launch; w = the layer's [3][3][cin][cout] weights; mode 0 forward, 1 input gradient
int conv3_wino4_launch(ssdseg_ctx* ctx, const Conv3TArgs& a, const float* w, int cin, int cout, int mode);
bool conv3_wino4_takes(int n, int h, int w, int cred, int nout);
int conv3_wino_launch(ssdseg_ctx* ctx, Conv3TArgs a, const float* w, int cin, int cout, int mode) {
    if (conv3_wino4_takes(a.n, a.h, a.w, a.cred, a.nout)) return conv3_wino4_launch(ctx, a, w, cin, cout, mode);
    void* ws;
    const int npad = cdiv(a.nout, WINO_NT) * WINO_NT;
    const size_t ubytes = (size_t)16 * a.cred * npad * sizeof(float);      // U[cred / 8][16][npad][8]
    int rc = ssdseg_workspace(ctx, ubytes, &ws);
    if (rc) return rc;
    SSDSEG_LAUNCH(ctx, 4.0 * (9 + 16) * cin * cout, 0.0, conv3_wino_weights_kernel, dim3(cdiv(cout, 32), cdiv(cin, 32)), dim3(256), 0, w, (float*)ws, cin, cout, mode);
    SSDSEG_LAUNCH_CHECK();
    a.wt = (const float*)ws;
    a.tiles_h = cdiv(a.h, C3T_ROWS); a.tiles_w = cdiv(a.w, C3T_COLS); a.ntiles_n = cdiv(a.nout, WINO_NT); a.ncols = WINO_NT;
    if (a.in_hp == 0) { a.in_hp = a.h; a.in_wp = a.w; }
    // (the zero-bordered copy is entered at its pixel (1, 1): the last byte the kernel may touch is that of image pixel (h-1, w-1))
    a.in_bytes = (unsigned)(((((long long)(a.n - 1) * a.in_hp + a.h - 1) * a.in_wp + a.w - 1) * a.ldi + a.cred) * 4);
    a.wt_bytes = (unsigned)ubytes;
    const size_t lds = wino_lds_floats(a.cred) * sizeof(float);
    static size_t configured = 0;
    if (lds > configured) {
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    const double m = (double)a.n * a.h * a.w;
    const double cost_bytes = 4.0 * (m * a.cred + m * a.nout + 9.0 * a.cred * a.nout);   // X + Y + W
    // flops EXECUTED on the MFMA pipe: 16 positions x (m / 4) tiles x 2 cred nout = 8 m cred nout -- 16/36 of the direct convolution's
    // 18 m cred nout (bench.py reports that figure beside it as `direct_equivalent`; the roofline fraction uses the executed ones)
    const double cost_flops = 8.0 * m * a.cred * a.nout;
    const int mtiles = a.n * a.tiles_h * a.tiles_w;
    const bool with_view = a.cs != nullptr || a.act != SSDSEG_ACT_NONE;
    char kbuf[64];
    snprintf(kbuf, sizeof(kbuf), "conv3_wino_kernel<%s> [%s]", with_view ? "true" : "false", mode ? "bwd_data" : "fwd");   // symbol as rocprofv3 spells it + role
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    if (with_view)
        SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, conv3_wino_kernel<true>, dim3((unsigned)(mtiles * a.ntiles_n)), dim3(C3T_THREADS), lds, a);
    else
        SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, conv3_wino_kernel<false>, dim3((unsigned)(mtiles * a.ntiles_n)), dim3(C3T_THREADS), lds, a);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

// ---- Winograd F(4x4, 3x3) form (conv3_wino4.h): forward / input gradient of the layers the F(2x2) form takes, where a tile
// geometry exists.  SSDSEG_CONV3_F4=0: never, =1: wherever it fits (parity tests); unset: the large layers.
struct Wino4Geom { int tr, tc, trs, tiles_h, tiles_w; };
bool wino4_geometry(int h, int w, Wino4Geom* g) {
    long long best = -1;
    for (int tc = 1; tc <= 11; ++tc)
        for (int tr = 1; tr * tc <= 32; ++tr) {
            if (4 * tr * (4 * tc + 2) > 2 * W4_THREADS) continue;      // strips x four 16-byte chunks: two tasks per thread
            int trs = 24 * W4_RL;
            while ((trs & 15) != (tc & 15)) ++trs;
            if (tr * trs + 15 > W4_QP_MAX) continue;            // (+ the round-up of the plane stride to 1 mod 16)
            const long long blocks = (long long)cdiv(h, 4 * tr) * cdiv(w, 4 * tc);
            // fewest blocks (every block costs a full 32-row MFMA pass); then the squarer patch (less halo)
            const long long key = blocks * 1024 + (tr > tc ? tr - tc : tc - tr);
            if (best < 0 || key < best) { best = key; *g = Wino4Geom{tr, tc, trs, cdiv(h, 4 * tr), cdiv(w, 4 * tc)}; }
        }
    return best >= 0;
}
bool conv3_wino4_takes(int n, int h, int w, int cred, int nout) {
    const char* e = getenv("SSDSEG_CONV3_F4");
    if (e != nullptr && e[0] == '0') return false;
    if (!conv3_wino_takes(n, h, w, cred, nout) || cred % 16 != 0 || wino4_lds_floats(cred) * sizeof(float) > (size_t)160 * 1024) return false;
    Wino4Geom g;
    if (!wino4_geometry(h, w, &g)) return false;
    if ((long long)36 * cred * cdiv(nout, W4_NT) * W4_NT * 4 >= (1LL << 31)) return false;
    if (e != nullptr && e[0] == '1') return true;
    // automatic: where every CU gets a few work items (pixel tile x 32-channel tile) -- the decoder conv of the full-size models
    return (long long)n * g.tiles_h * g.tiles_w * cdiv(nout, W4_NT) >= 2048;
}

int conv3_wino4_launch(ssdseg_ctx* ctx, const Conv3TArgs& a, const float* w, int cin, int cout, int mode) {
    Wino4Geom g;
    if (!wino4_geometry(a.h, a.w, &g)) return SSDSEG_EINVAL(6);
    void* ws;
    const int npad = cdiv(a.nout, W4_NT) * W4_NT;
    const size_t ubytes = (size_t)36 * a.cred * npad * sizeof(float);      // U[cred / 8][36][npad][8]
    int rc = ssdseg_workspace(ctx, ubytes, &ws);
    if (rc) return rc;
    SSDSEG_LAUNCH(ctx, 4.0 * (9 + 36) * cin * cout, 0.0, conv3_wino4_weights_kernel, dim3(cdiv(cout, 32), cdiv(cin, 32)), dim3(256), 0, w, (float*)ws, cin, cout, mode);
    SSDSEG_LAUNCH_CHECK();
    Wino4Args p{};
    p.in = a.in; p.cs = a.cs; p.ct = a.ct; p.act = a.act; p.ldi = a.ldi;
    p.u = (const float*)ws;
    p.out = a.out; p.ldo = a.ldo; p.accumulate = a.accumulate; p.stats = a.stats;
    p.n = a.n; p.h = a.h; p.w = a.w; p.cred = a.cred; p.nout = a.nout; p.npad = npad;
    p.tr = g.tr; p.tc = g.tc; p.trs = g.trs;
    p.qps = g.tr * g.trs;
    while ((p.qps & 15) != 1) ++p.qps; p.tiles_h = g.tiles_h; p.tiles_w = g.tiles_w; p.ntiles_n = npad / W4_NT;
    {
        const char* ge = getenv("SSDSEG_W4_GROUP");      // (A/B runs) channel tiles per group of the work order
        p.group = ge != nullptr ? atoi(ge) : 2;
        if (p.group < 1 || p.ntiles_n % p.group != 0) p.group = 1;
    }
    p.in_hp = a.in_hp ? a.in_hp : a.h; p.in_wp = a.in_hp ? a.in_wp : a.w;
    p.in_bytes = (unsigned)(((((long long)(a.n - 1) * p.in_hp + a.h - 1) * p.in_wp + a.w - 1) * a.ldi + a.cred) * 4);
    p.u_bytes = (unsigned)ubytes;
    {
        const long long ob = (((long long)a.n * a.h * a.w - 1) * a.ldo + a.nout) * 4;
        p.out_bytes = ob < (1LL << 31) ? (unsigned)ob : 0u;
        if (ob >= (1LL << 31) && !p.accumulate) p.accumulate = 2;      // 32-bit buffer offsets do not reach: plain stores
        const char* fe = getenv("SSDSEG_W4_PLAIN_STORES");              // (parity tests) that path at any size
        if (fe != nullptr && fe[0] == '1' && !p.accumulate) p.accumulate = 2;
    }
    p.trace = nullptr;
    const size_t lds = wino4_lds_floats(a.cred) * sizeof(float);
    static size_t configured = 0;
    if (lds > configured) {
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wino4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wino4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    const double m = (double)a.n * a.h * a.w;
    const double cost_bytes = 4.0 * (m * a.cred + m * a.nout + 9.0 * a.cred * a.nout);   // X + Y + W
    // flops EXECUTED on the MFMA pipe: 36 positions x (m / 16) tiles x 2 cred nout = 4.5 m cred nout -- a quarter of the direct sum's 18
    const double cost_flops = 4.5 * m * a.cred * a.nout;
    const int mtiles = a.n * g.tiles_h * g.tiles_w;
    // persistent blocks: one per CU (154 KB of LDS each), a multiple of 8 so that every XCD walks one contiguous run of the items
    const long long items = (long long)mtiles * p.ntiles_n;
    int nblocks = ctx->num_cus;
    if (const char* e = getenv("SSDSEG_W4_BLOCKS")) nblocks = atoi(e) > 0 ? atoi(e) : nblocks;
    if (nblocks > items) nblocks = (int)items;
    if (nblocks >= 8 && items % 8 == 0) nblocks -= nblocks % 8;
    if (getenv("SSDSEG_W4_TRACE") != nullptr) { SSDSEG_HIP(hipMalloc((void**)&p.trace, (size_t)nblocks * 64 * 4 * 8)); SSDSEG_HIP(hipMemset(p.trace, 0, (size_t)nblocks * 64 * 4 * 8)); }
    const bool with_view = a.cs != nullptr || a.act != SSDSEG_ACT_NONE;
    char kbuf[64];
    snprintf(kbuf, sizeof(kbuf), "conv3_wino4_kernel<%s> [%s]", with_view ? "true" : "false", mode ? "bwd_data" : "fwd");
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    if (with_view)
        SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, conv3_wino4_kernel<true>, dim3((unsigned)nblocks), dim3(W4_THREADS), lds, p);
    else
        SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, conv3_wino4_kernel<false>, dim3((unsigned)nblocks), dim3(W4_THREADS), lds, p);
    SSDSEG_LAUNCH_CHECK();
    if (p.trace != nullptr) {       // measurement only: synchronous, prints to stderr
        std::vector<unsigned long long> h((size_t)nblocks * 64 * 4);
        SSDSEG_HIP(hipStreamSynchronize(ctx->stream));
        SSDSEG_HIP(hipMemcpy(h.data(), p.trace, h.size() * 8, hipMemcpyDeviceToHost));
        SSDSEG_HIP(hipFree(p.trace));
        const int per = (int)((items + nblocks - 1) / nblocks) < 64 ? (int)((items + nblocks - 1) / nblocks) : 64;
        for (int b : {0, 1, nblocks / 2, nblocks - 1}) {
            double loop = 0, epi = 0;
            for (int k = 0; k < per; ++k) {
                loop += (double)(h[((size_t)b * 64 + k) * 4 + 1] - h[((size_t)b * 64 + k) * 4 + 0]);
                epi += (double)(h[((size_t)b * 64 + k) * 4 + 2] - h[((size_t)b * 64 + k) * 4 + 1]);
            }
            const double gap = per > 1 ? ((double)(h[((size_t)b * 64 + per - 1) * 4 + 0] - h[((size_t)b * 64) * 4 + 0]) - (loop - (double)(h[((size_t)b * 64 + per - 1) * 4 + 1] - h[((size_t)b * 64 + per - 1) * 4 + 0])) ) / (per - 1) : 0;
            fprintf(stderr, "w4 trace block %3d: %d items, loop %.0f clk/item (%.0f per 16-channel step), loop end -> item end %.0f, loop end -> next loop start %.0f\n", b, per,
                    loop / per, loop / per / (a.cred / 16), epi / per, gap);
        }
    }
    return 0;
}

// weight gradient in the Winograd form (conv3_wino_wgrad.h): h even, w a multiple of 32, everything below 2^31 bytes
bool conv3_wino_wgrad_takes(int n, int h, int w, int cin, int cout) {
    const int mode = conv3_wino_mode();
    if (mode == 0 || h % 2 != 0 || w % 2 != 0) return false;
    if ((long long)n * (h + 2) * (w + 2) * cin * 4 >= (1LL << 31) || (long long)n * h * w * cout * 4 >= (1LL << 31)) return false;
    return mode == 1 || (long long)n * (h / 2) * cdiv(w, 32) >= 512;
}

// xsaved != nullptr: the zero-bordered activated input already exists (written by ssdseg_conv3x3_fwd_saved), `in` is not read
int conv3_wino_wgrad_launch(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* dy, float* dw, int n, int h, int w, int cin, int cout,
                            const float* xsaved = nullptr) {
    WinoWgArgs a{};
    a.n = n; a.h = h; a.w = w; a.cin = cin; a.cout = cout;
    a.cpatches = cdiv(cin, WWG_KT); a.npatches = cdiv(cout, WWG_NT);
    a.strips = cdiv(w, 32);
    a.wrem = w - 32 * (a.strips - 1);
    a.steps = n * (h / 2) * a.strips;
    const int patches = a.cpatches * a.npatches;
    // one block per CU (100 KB of LDS, 8 waves): patches x steps units dealt evenly (WinoWgArgs); >= 4 steps per block
    long long span = ((long long)patches * a.steps + ctx->num_cus - 1) / ctx->num_cus;
    if (span < 4) span = 4;
    if (span > a.steps) span = a.steps;
    a.span = (int)span;
    a.full = a.steps / a.span;
    a.tail = a.steps - a.full * a.span;
    a.slots = a.full + (a.tail > 0 ? (a.tail + a.span - 1) / a.span + 1 : 0);
    const int nblocks = a.full * patches + (int)(((long long)patches * a.tail + a.span - 1) / a.span);
    const size_t xpb = xsaved != nullptr ? 0 : align256((size_t)n * (h + 2) * (w + 2) * cin * sizeof(float));
    const size_t pb = (size_t)patches * a.slots * 16 * WWG_KT * WWG_NT * sizeof(float);
    SSDSEG_ARG(pb < ((size_t)1 << 31), 9);
    void* ws;
    int rc = ssdseg_workspace(ctx, xpb + pb, &ws);
    if (rc) return rc;
    float* xp = (float*)ws;
    a.xp = xsaved != nullptr ? xsaved : xp; a.dy = dy; a.part = (float*)((char*)ws + xpb);
    a.xp_bytes = (unsigned)((size_t)n * (h + 2) * (w + 2) * cin * sizeof(float));
    a.dy_bytes = (unsigned)((size_t)n * h * w * cout * sizeof(float));
    a.part_bytes = (unsigned)pb;
    const double m = (double)n * h * w;
    if (xsaved == nullptr) {
        const long long tot4 = (long long)n * (h + 2) * (w + 2) * (cin / 4);
        SSDSEG_LAUNCH(ctx, 8.0 * m * cin, 0.0, conv3_pad_view_kernel, dim3((unsigned)((tot4 + 255) / 256 < 16384 ? (tot4 + 255) / 256 : 16384)), dim3(256), 0, in->x,
                      in->scale, in->shift, in->act, ldx, xp, n, h, w, cin, 0);
        SSDSEG_LAUNCH_CHECK();
    }
    static bool configured = false;   // dynamic LDS beyond 64 KiB has to be announced once
    if (!configured) {
        SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wino_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WWG_LDS_BYTES));
        configured = true;
    }
    const double cost_bytes = 4.0 * (m * cin + m * cout + 9.0 * cin * cout);   // X + dY + dW
    const double cost_flops = 8.0 * m * cin * cout;                             // executed MFMA flops: 16/36 of the direct form's 18 m cin cout
    SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, conv3_wino_wgrad_kernel, dim3((unsigned)nblocks), dim3(WWG_THREADS), WWG_LDS_BYTES, a);
    SSDSEG_LAUNCH_CHECK();
    const long long cn = (long long)cin * cout;
    SSDSEG_LAUNCH(ctx, 4.0 * cn * (16.0 * a.slots + 9.0), 0.0, conv3_wino_wgrad_finalize_kernel, dim3((unsigned)((cn + 255) / 256)), dim3(256), 0, (const float*)a.part, dw,
                  cin, cout, a.npatches, a.full, a.tail, a.span, a.slots);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

// Wt[n][k] = W[k][n] for a whole table of matrices in ONE launch: blockIdx.y = matrix, blockIdx.x strides over its 32x32 tiles.
// table[mat] = {source pointer, destination pointer, k, n} as four 64-bit words.
__global__ void __launch_bounds__(256) transpose_batch_kernel(const long long* __restrict__ table) {
    __shared__ float tile[32][33];
    const long long* e = table + 4 * (long long)blockIdx.y;
    const float* __restrict__ w = reinterpret_cast<const float*>(e[0]);
    float* __restrict__ wt = reinterpret_cast<float*>(e[1]);
    const int k = (int)e[2], n = (int)e[3];
    const int tn = (n + 31) / 32, tk = (k + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int t = blockIdx.x; t < tn * tk; t += gridDim.x) {
        const int c0 = (t / tn) * 32, n0 = (t % tn) * 32;
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, j = n0 + tx;
            tile[r][tx] = (c < k && j < n) ? w[(long long)c * n + j] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int j = n0 + r, c = c0 + tx;
            if (j < n && c < k) wt[(long long)j * k + c] = tile[tx][r];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int ssdseg_pwconv_parts(int m, int n, int* nparts_host) {
    SSDSEG_ARG(m > 0, 1);
    SSDSEG_ARG(n > 0, 2);
    SSDSEG_ARG(nparts_host != nullptr, 3);
    *nparts_host = rowA_grid_y(m, n);
    return 0;
}

int ssdseg_pwconv_fwd_wt(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, const float* wt, float* y, int ldy, int m,
                         int k, int n, float* stats) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldx >= k && ldx % 4 == 0, 3);
    SSDSEG_ARG(w != nullptr, 4);
    SSDSEG_ARG(y != nullptr, 5);
    SSDSEG_ARG(ldy >= n, 6);
    SSDSEG_ARG(m > 0, 7);
    SSDSEG_ARG(k > 0 && k % 4 == 0, 8);
    SSDSEG_ARG(n > 0 && n % 4 == 0, 9);
    RowAArgs a{};
    a.a0 = in->x; a.cs = in->scale; a.ct = in->shift; a.act = in->act; a.lda = ldx;
    a.b = w; a.ldb = n;
    a.out = y; a.ldo = ldy;
    a.stats = stats;
    a.I = m; a.R = k; a.J = n;
    return launch_rowA<0, 0>(ctx, a, wt);
}

int ssdseg_pwconv_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int ldy, int m, int k,
                      int n, float* stats) {
    return ssdseg_pwconv_fwd_wt(ctx, in, ldx, w, nullptr, y, ldy, m, k, n, stats);
}

int ssdseg_pwconv_wt_floats(int m, int ldx, int k, int n, int* floats_host) {
    SSDSEG_ARG(m > 0, 1);
    SSDSEG_ARG(k > 0 && n > 0, 3);
    SSDSEG_ARG(floats_host != nullptr, 5);
    const bool tile = pw_tile_mode() != 0 && pw_tile_takes(m, ldx, k, n) && (pw_tile_mode() == 1 || pw_tile_default(0, m, k, n, false, false));
    *floats_host = tile ? k * n : 0;
    return 0;
}

int ssdseg_transpose_batch(ssdseg_ctx* ctx, const long long* table, int nmat, int max_tiles, long long total_floats) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(table != nullptr, 2);
    SSDSEG_ARG(nmat > 0, 3);
    SSDSEG_ARG(max_tiles > 0, 4);
    const int gx = max_tiles < 64 ? max_tiles : 64;
    SSDSEG_LAUNCH(ctx, 8.0 * (double)total_floats, 0.0, transpose_batch_kernel, dim3(gx, nmat, 1), dim3(256), 0, table);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_pwconv_bwd_data(ssdseg_ctx* ctx, const ssdseg_gview* dy, int ldy, const float* w, float* dx, int ldx, int m,
                           int k, int n, const float* residual, int ldr, int accumulate) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 2);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 2);
    SSDSEG_ARG(ldy >= n && ldy % 4 == 0, 3);
    SSDSEG_ARG(w != nullptr, 4);
    SSDSEG_ARG(dx != nullptr, 5);
    SSDSEG_ARG(ldx >= k, 6);
    SSDSEG_ARG(m > 0, 7);
    SSDSEG_ARG(k > 0 && k % 4 == 0, 8);
    SSDSEG_ARG(n > 0 && n % 4 == 0, 9);
    SSDSEG_ARG(residual == nullptr || ldr >= k, 11);
    RowAArgs a{};
    a.a0 = dy->g; a.a1 = dy->y; a.cs = dy->scale; a.ct = dy->shift; a.ck1 = dy->k1; a.ck0 = dy->k0; a.act = dy->act;
    a.lda = ldy;
    a.b = w; a.ldb = n;
    a.out = dx; a.ldo = ldx;
    a.residual = residual; a.ldr = ldr; a.accumulate = accumulate;
    a.I = m; a.R = n; a.J = k;
    return launch_rowA<1, 0>(ctx, a);
}

// dx and dW of a pointwise conv in ONE pass over the gradient (models.py:65-67 backward).  Shapes the fused kernel covers:
// k <= 32 input channels, n <= 192 output channels (MobileNetV2 expand convs of blocks 1-6, where the 6x-wide gradient
// is the whole cost); everything else runs the two separate kernels -- same results either way.
int ssdseg_pwconv_bwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, int ldy, const float* w, float* dx,
                      int lddx, float* dw, int m, int k, int n, const float* residual, int ldr, int accumulate) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldx >= k && ldx % 4 == 0, 3);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 4);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 4);
    SSDSEG_ARG(ldy >= n && ldy % 4 == 0, 5);
    SSDSEG_ARG(w != nullptr, 6);
    SSDSEG_ARG(dx != nullptr, 7);
    SSDSEG_ARG(lddx >= k, 8);
    SSDSEG_ARG(dw != nullptr, 9);
    SSDSEG_ARG(m > 0, 10);
    SSDSEG_ARG(k > 0 && k % 4 == 0, 11);
    SSDSEG_ARG(n > 0 && n % 4 == 0, 12);
    SSDSEG_ARG(residual == nullptr || ldr >= k, 14);
    // Measured per layer on MI355X (profiles/r01_wres_vs_general_per_layer.txt): the fused pass wins on the big early layers
    // (blocks 1-3 at batch 32: 760 -> 642, 492 -> 462, 456 -> 354 us against wgrad + bwd_data), is even at 153,600 rows and
    // loses with a single 32-column chunk, where the per-tile set-up dominates.  Its 16*NT accumulator registers cap the
    // occupancy at 1-2 waves per SIMD, which is why it stops at ~3.4 TB/s.  SSDSEG_PW_FUSED=1 forces it for every shape it
    // supports (k <= 32, n <= 192); the parity tests run both ways.
    const char* fenv = getenv("SSDSEG_PW_FUSED");   // "1": every supported shape, "0": never, unset: where it measured faster
    const bool force_fused = fenv != nullptr && fenv[0] == '1', never_fused = fenv != nullptr && fenv[0] == '0';
    const bool fused = !never_fused && k <= 32 && n <= 192 && (force_fused || (n > 32 && m >= 500000));
    if (!fused) {
        // dW is off the critical path (nothing reads it before the optimizer): it runs on the side stream, concurrently with
        // the backward-data GEMM and whatever follows it on the main stream
        const bool side = ssdseg_side_begin(ctx);
        int rc = ssdseg_pwconv_bwd_weight(ctx, in, ldx, dy, ldy, dw, m, k, n);
        if (side) ssdseg_side_end(ctx);
        if (rc) return rc;
        return ssdseg_pwconv_bwd_data(ctx, dy, ldy, w, dx, lddx, m, k, n, residual, ldr, accumulate);
    }
    RowAArgs a{};
    a.a0 = dy->g; a.a1 = dy->y; a.cs = dy->scale; a.ct = dy->shift; a.ck1 = dy->k1; a.ck0 = dy->k0; a.act = dy->act;
    a.lda = ldy;
    a.b = w; a.ldb = n;
    a.out = dx; a.ldo = lddx;
    a.residual = residual; a.ldr = ldr; a.accumulate = accumulate;
    a.I = m; a.R = n; a.J = k;
    a.xw = in->x; a.xws = in->scale; a.xwt = in->shift; a.xwact = in->act; a.ldxw = ldx;
    const int mtiles = cdiv(m, BM);
    const int nt = cdiv(n, 32);
    const size_t wl = wres_enabled() ? wres_lds_bytes(n, 1) : 0;
    // two resident blocks per CU (112 + 16*NT registers each); every block walks >= 1 row tile.  (Round 3: three blocks per CU for
    // NT <= 3 -- 168 registers, 12 waves per CU -- left the block-1 expand backward at 0.642 ms: not short of waves in flight.)
    int gy = 2 * ctx->num_cus;
    if (gy > mtiles) gy = mtiles;
    void* ws;
    int rc = ssdseg_partials(ctx, (size_t)gy * k * n * sizeof(float), &ws);
    if (rc) return rc;
    a.wpart = (float*)ws;
    const dim3 grid(1, gy, 1);
    const size_t lds = (size_t)(BM * AS + BK * 33) * sizeof(float);
    const double cost_bytes = 4.0 * ((double)m * n + 2.0 * m * k + 2.0 * k * n);   // 8(d): read dY, read X, write dX, read W, write dW
    const double cost_flops = 4.0 * m * k * n;
    ctx->timing_view_bytes = dy->scale != nullptr ? 4.0 * m * n : 0.0;
    if (wl > 0) {
        char wbuf[64];
        snprintf(wbuf, sizeof(wbuf), "gemm_wres_kernel<1, 1, %d>", nt > 6 ? 6 : nt);   // NT > 0: fused dW
        const char* wname = ctx->timing ? ssdseg_intern(wbuf) : "";
        switch (nt) {
            case 1: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<1, 1, 1>), grid, dim3(256), wl, a); break;
            case 2: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<1, 1, 2>), grid, dim3(256), wl, a); break;
            case 3: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<1, 1, 3>), grid, dim3(256), wl, a); break;
            case 4: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<1, 1, 4>), grid, dim3(256), wl, a); break;
            case 5: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<1, 1, 5>), grid, dim3(256), wl, a); break;
            default: SSDSEG_LAUNCH_NAMED(ctx, wname, cost_bytes, cost_flops, (gemm_wres_kernel<1, 1, 6>), grid, dim3(256), wl, a); break;
        }
        SSDSEG_LAUNCH_CHECK();
        if (gy == 1) {
            SSDSEG_HIP(hipMemcpyAsync(dw, a.wpart, (size_t)k * n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            return 0;
        }
        return ssdseg_colsum(ctx, a.wpart, gy, (long long)k * n, dw);
    }
    const bool occ = a.I >= occ_rows();
    char fbuf[64];
    snprintf(fbuf, sizeof(fbuf), "gemm_rowA_kernel<1, 1, 0, %d, 0, %d>", nt > 6 ? 6 : nt, (int)occ);   // NT > 0: fused dW
    const char* kname = ctx->timing ? ssdseg_intern(fbuf) : "";
    switch (nt) {
        case 1:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 1, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 1, 0, 0>), grid, dim3(256), lds, a);
            break;
        case 2:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 2, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 2, 0, 0>), grid, dim3(256), lds, a);
            break;
        case 3:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 3, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 3, 0, 0>), grid, dim3(256), lds, a);
            break;
        case 4:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 4, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 4, 0, 0>), grid, dim3(256), lds, a);
            break;
        case 5:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 5, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 5, 0, 0>), grid, dim3(256), lds, a);
            break;
        default:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 6, 0, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 6, 0, 0>), grid, dim3(256), lds, a);
            break;
    }
    SSDSEG_LAUNCH_CHECK();
    if (gy == 1) {
        SSDSEG_HIP(hipMemcpyAsync(dw, a.wpart, (size_t)k * n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    return ssdseg_colsum(ctx, a.wpart, gy, (long long)k * n, dw);
}

// dx + dW of a pointwise conv plus the BatchNorm backward of the layer feeding it (the depthwise BN in front of a project conv):
// the backward-data kernel's float4 epilogue reduces sum(mask*dx), sum(mask*dx*xhat) while it stores dx.  Only valid when this
// conv is the ONLY consumer of that BatchNorm's output.  Falls back to the separate reduction pass for unaligned tensors.
// dx = dy * w^T plus the BatchNormalization backward of the layer that feeds this conv (sums in the GEMM epilogue); shared by
// ssdseg_pwconv_bwd_bn and the tap-expanded form of the narrow 3x3 conv (ssdseg_conv3x3_bwd_data_bn)
static int pwconv_bwd_data_bn(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, int ldy, const float* w, float* dx,
                              int lddx, int m, int k, int n, const float* in_mean, const float* in_invstd, float* in_dgamma,
                              float* in_dbeta, float* in_k1, float* in_k0) {
    int rc = 0;
    const bool aligned = lddx % 4 == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)in->x & 15) == 0;
    const char* benv = getenv("SSDSEG_NO_BN_EPILOGUE");
    if (!aligned || (benv != nullptr && benv[0] == '1')) {
        rc = ssdseg_pwconv_bwd_data(ctx, dy, ldy, w, dx, lddx, m, k, n, nullptr, 0, 0);
        if (rc) return rc;
        return ssdseg_bn_bwd_reduce(ctx, dx, lddx, in->x, ldx, m, k, in->scale, in->shift, in_mean, in_invstd, in->act, in_dgamma, in_dbeta,
                                    in_k1, in_k0);
    }
    if (pw_tile_mode() != 0 && pw_tile_takes(m, ldy, n, k) && (pw_tile_mode() == 1 || pw_tile_default(1, m, n, k, true, false))) {
        // tile GEMM with the BatchNorm-backward sums taken in its epilogue (pw_tile.h)
        PwTArgs t{};
        t.a0 = dy->g; t.a1 = dy->scale != nullptr ? dy->y : dy->g;
        t.cs = dy->scale; t.ct = dy->shift; t.ck1 = dy->k1; t.ck0 = dy->k0; t.act = dy->act; t.lda = ldy;
        t.wt = w;
        t.out = dx; t.ldo = lddx;
        t.M = m; t.cred = n; t.nout = k;
        const int mt = cdiv(m, m >= 65536 ? 256 : 128);
        const char* pe = getenv("SSDSEG_PWT_PARTS");
        const int capb = pe != nullptr && atoi(pe) >= 64 ? atoi(pe) : 1024;
        const int gy = mt < capb ? mt : capb;
        void* ws;
        rc = ssdseg_workspace(ctx, (size_t)gy * 2 * k * sizeof(float), &ws);
        if (rc) return rc;
        t.bn_y = in->x; t.ldby = ldx; t.bn_s = in->scale; t.bn_t = in->shift; t.bn_mean = in_mean; t.bn_istd = in_invstd; t.bn_act = in->act;
        t.bnpart = (float*)ws;
        rc = pw_tile_launch<1>(ctx, t, gy, 4.0 * ((dy->scale != nullptr ? (double)m * n : 0.0) + (double)m * k));
        if (rc) return rc;
        return ssdseg_bn_bwd_finalize_launch(ctx, t.bnpart, gy, k, (double)m, in->scale, in_mean, in_invstd, in_dgamma, in_dbeta, in_k1, in_k0);
    }
    RowAArgs a{};
    a.a0 = dy->g; a.a1 = dy->y; a.cs = dy->scale; a.ct = dy->shift; a.ck1 = dy->k1; a.ck0 = dy->k0; a.act = dy->act;
    a.lda = ldy;
    a.b = w; a.ldb = n;
    a.out = dx; a.ldo = lddx;
    a.I = m; a.R = n; a.J = k;
    const int wn = rowA_wn_bwd(m, k, n);
    const int nparts = rowA_grid_y_wn(m, k, wn);
    void* ws;
    rc = ssdseg_workspace(ctx, (size_t)nparts * 2 * k * sizeof(float), &ws);
    if (rc) return rc;
    a.bn_y = in->x; a.ldby = ldx; a.bn_s = in->scale; a.bn_t = in->shift; a.bn_mean = in_mean; a.bn_istd = in_invstd; a.bn_act = in->act;
    a.bnpart = (float*)ws;
    const dim3 grid(cdiv(k, 32 * wn), nparts, 1);
    size_t lds = (size_t)(BM * AS + BK * (32 * wn + 1)) * sizeof(float);
    const size_t cs = (size_t)64 * (32 * wn + 4) * sizeof(float);
    if (cs > lds) lds = cs;
    // 8(d): read dY, write dX, read W; the raw input tensor read by the fused BatchNorm-backward epilogue replaces that BN's own
    // reduction pass and is counted with the gradient view's second tensor as `view_bytes`
    const double cost_bytes = 4.0 * ((double)m * n + (double)m * k + (double)k * n);
    const double cost_flops = 2.0 * m * k * n;
    ctx->timing_view_bytes = 4.0 * ((dy->scale != nullptr ? (double)m * n : 0.0) + (double)m * k);
    const bool occ = m >= occ_rows();
    char kbuf[64];
    snprintf(kbuf, sizeof(kbuf), "gemm_rowA_kernel<%d, 1, 0, 0, 2, %d>", wn, (int)occ);
    const char* kname = ctx->timing ? ssdseg_intern(kbuf) : "";
    switch (wn) {
        case 1:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 0, 2, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<1, 1, 0, 0, 2, 0>), grid, dim3(256), lds, a);
            break;
        case 2:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<2, 1, 0, 0, 2, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<2, 1, 0, 0, 2, 0>), grid, dim3(256), lds, a);
            break;
        case 3:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<3, 1, 0, 0, 2, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<3, 1, 0, 0, 2, 0>), grid, dim3(256), lds, a);
            break;
        case 4:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<4, 1, 0, 0, 2, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<4, 1, 0, 0, 2, 0>), grid, dim3(256), lds, a);
            break;
        default:
            if (occ) SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<5, 1, 0, 0, 2, 1>), grid, dim3(256), lds, a);
            else SSDSEG_LAUNCH_NAMED(ctx, kname, cost_bytes, cost_flops, (gemm_rowA_kernel<5, 1, 0, 0, 2, 0>), grid, dim3(256), lds, a);
            break;
    }
    SSDSEG_LAUNCH_CHECK();
    return ssdseg_bn_bwd_finalize_launch(ctx, a.bnpart, nparts, k, (double)m, in->scale, in_mean, in_invstd, in_dgamma, in_dbeta, in_k1, in_k0);
}

int ssdseg_pwconv_bwd_bn(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, int ldy, const float* w, float* dx,
                         int lddx, float* dw, int m, int k, int n, const float* in_mean, const float* in_invstd, float* in_dgamma,
                         float* in_dbeta, float* in_k1, float* in_k0) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && in->scale != nullptr && in->shift != nullptr, 2);
    SSDSEG_ARG(ldx >= k && ldx % 4 == 0, 3);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 4);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 4);
    SSDSEG_ARG(ldy >= n && ldy % 4 == 0, 5);
    SSDSEG_ARG(w != nullptr, 6);
    SSDSEG_ARG(dx != nullptr, 7);
    SSDSEG_ARG(lddx >= k, 8);
    SSDSEG_ARG(dw != nullptr, 9);
    SSDSEG_ARG(m > 0, 10);
    SSDSEG_ARG(k > 0 && k % 4 == 0, 11);
    SSDSEG_ARG(n > 0 && n % 4 == 0, 12);
    SSDSEG_ARG(in_mean != nullptr && in_invstd != nullptr, 13);
    SSDSEG_ARG(in_k1 != nullptr && in_k0 != nullptr, 17);
    // dW first, on the side stream (it only reads)
    const bool side = ssdseg_side_begin(ctx);
    int rc = ssdseg_pwconv_bwd_weight(ctx, in, ldx, dy, ldy, dw, m, k, n);
    if (side) ssdseg_side_end(ctx);
    if (rc) return rc;
    return pwconv_bwd_data_bn(ctx, in, ldx, dy, ldy, w, dx, lddx, m, k, n, in_mean, in_invstd, in_dgamma, in_dbeta, in_k1, in_k0);
}

int ssdseg_pwconv_bwd_weight(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, int ldy, float* dw,
                             int m, int k, int n) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldx >= k && ldx % 4 == 0, 3);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 4);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 4);
    SSDSEG_ARG(ldy >= n && ldy % 4 == 0, 5);
    SSDSEG_ARG(dw != nullptr, 6);
    SSDSEG_ARG(m > 0, 7);
    SSDSEG_ARG(k > 0 && k % 4 == 0, 8);
    SSDSEG_ARG(n > 0 && n % 4 == 0, 9);
    WGradArgs a{};
    a.x = in->x; a.xs = in->scale; a.xt = in->shift; a.xact = in->act; a.ldx = ldx;
    a.g = dy->g; a.y = dy->y; a.gs = dy->scale; a.gt = dy->shift; a.gk1 = dy->k1; a.gk0 = dy->k0; a.gact = dy->act;
    a.ldy = ldy;
    a.M = m; a.K = k; a.N = n;
    return wgrad_run(ctx, a, dw);
}

// ------------------------------------------------------------------------------------------------ stem (K1 + K2)
// Rescaling + Conv2D 3x3 stride 2 SAME on the 3-channel image (reference models.py:187,196 -> :65; ShuffleNetV2 :622,628)
// as implicit GEMM [n*ho*wo, 27] x [27, cout] through the same MFMA kernels: the im2col gather happens in the LDS
// loader, the output leaves the accumulators as full 128-byte row segments, BN statistics come from the epilogue.
// conv3n.hip: the narrow 3x3 conv's input gradient + fused BatchNorm sums as a streaming kernel (cin == 256, cout == 4)
bool ssdseg_conv3n_direct_takes(int cin, int cout, int ldx);
int ssdseg_conv3n_bwd_bn_direct(ssdseg_ctx* ctx, const ssdseg_view* in, const ssdseg_gview* dy, const float* w, float* dx, int ldx, int n, int h,
                                int wdt, const float* in_mean, const float* in_invstd, float* in_dgamma, float* in_dbeta, float* in_k1,
                                float* in_k0);
// stem.hip: the direct streaming kernel for <= 64 output channels (SSDSEG_STEM_DIRECT=0: the implicit GEMM below)
bool ssdseg_stem_direct_takes(int cout);
bool ssdseg_stem_direct_eligible(int cout);
int ssdseg_stem_direct_blocks(int n, int h);
int ssdseg_stem_direct_fwd(ssdseg_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int n, int h, int wdt, int cout,
                           float in_scale, float in_offset, float* stats);

static void stem_geometry(int h, int wdt, int* ho, int* wo, int* pt, int* pl) {
    same_pad(h, 3, 2, 1, ho, pt);
    same_pad(wdt, 3, 2, 1, wo, pl);
}

int ssdseg_stem_conv_parts(int n, int h, int w, int cout, int* nparts_host) {
    SSDSEG_ARG(n > 0 && h > 0 && w > 0, 1);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 4);
    SSDSEG_ARG(nparts_host != nullptr, 5);
    int ho, wo, pt, pl;
    stem_geometry(h, w, &ho, &wo, &pt, &pl);
    // (sized for whichever forward kernel may run: the direct one writes one row per block, the implicit GEMM one per grid row)
    const int a = rowA_grid_y(n * ho * wo, cout), b = ssdseg_stem_direct_eligible(cout) ? ssdseg_stem_direct_blocks(n, h) : 0;
    *nparts_host = a > b ? a : b;
    return 0;
}

int ssdseg_stem_conv_fwd(ssdseg_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int n, int h, int wdt,
                         int cin, int cout, float in_scale, float in_offset, float* stats) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(x != nullptr, 2);
    SSDSEG_ARG(w != nullptr, 3);
    SSDSEG_ARG(bias == nullptr || stats == nullptr, 4);   // a biased conv is never followed by BatchNormalization here
    SSDSEG_ARG(y != nullptr, 5);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(cin == 3, 9);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 10);
    const bool direct = ssdseg_stem_direct_takes(cout) && (long long)n * h * wdt * 12 < (1LL << 31);      // (32-bit buffer offsets into the image)
    if (stats != nullptr) {      // the table is sized for the larger of the two kernels' row counts: the rows this launch does not write are zero
        int nparts = 0, ho_, wo_, pt_, pl_;
        int rc = ssdseg_stem_conv_parts(n, h, wdt, cout, &nparts);
        if (rc) return rc;
        stem_geometry(h, wdt, &ho_, &wo_, &pt_, &pl_);
        const int mine = direct ? ssdseg_stem_direct_blocks(n, h) : rowA_grid_y(n * ho_ * wo_, cout);
        if (nparts > mine) SSDSEG_HIP(hipMemsetAsync(stats + (size_t)mine * 2 * cout, 0, (size_t)(nparts - mine) * 2 * cout * sizeof(float), ctx->stream));
    }
    if (direct) return ssdseg_stem_direct_fwd(ctx, x, w, bias, y, n, h, wdt, cout, in_scale, in_offset, stats);
    RowAArgs a{};
    a.a0 = x; a.act = SSDSEG_ACT_NONE;
    a.b = w; a.ldb = cout;
    a.out = y; a.ldo = cout;
    a.stats = stats;
    a.bias = bias;
    stem_geometry(h, wdt, &a.convH, &a.convW, &a.stemPt, &a.stemPl);
    a.stemH = h; a.stemW = wdt; a.stemScale = in_scale; a.stemOffset = in_offset;
    a.I = n * a.convH * a.convW; a.R = 27; a.J = cout;
    return launch_rowA<0, 2>(ctx, a);
}

int ssdseg_stem_conv_bwd_weight(ssdseg_ctx* ctx, const float* x, const ssdseg_gview* dy, float* dw, float* dbias, int n, int h,
                                int wdt, int cin, int cout, float in_scale, float in_offset) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(x != nullptr, 2);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 3);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 3);
    SSDSEG_ARG(dbias == nullptr || dy->scale == nullptr, 5);   // the bias gradient is the plain column sum of g
    SSDSEG_ARG(dw != nullptr, 4);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(cin == 3, 9);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 10);
    WGradArgs a{};
    a.x = x; a.xact = SSDSEG_ACT_NONE;
    a.g = dy->g; a.y = dy->y; a.gs = dy->scale; a.gt = dy->shift; a.gk1 = dy->k1; a.gk0 = dy->k0; a.gact = dy->act;
    a.ldy = cout;
    a.stem = 1;
    stem_geometry(h, wdt, &a.convH, &a.convW, &a.stemPt, &a.stemPl);
    a.stemH = h; a.stemW = wdt; a.stemScale = in_scale; a.stemOffset = in_offset;
    a.M = n * a.convH * a.convW; a.K = 27; a.N = cout;
    int rc = wgrad_run(ctx, a, dw);
    if (rc || !dbias) return rc;
    // dbias[c] = sum_m g[m][c]: per-block channel sums, then the fixed-order fold
    int nparts = 0;
    rc = ssdseg_channel_stats_parts(a.M, cout, &nparts);
    if (rc) return rc;
    void* ws;
    rc = ssdseg_workspace(ctx, ((size_t)nparts * 2 * cout + 2 * (size_t)cout) * sizeof(float), &ws);
    if (rc) return rc;
    float* part = (float*)ws;
    float* both = part + (size_t)nparts * 2 * cout;
    rc = ssdseg_channel_stats(ctx, dy->g, cout, a.M, cout, part);
    if (rc) return rc;
    rc = ssdseg_colsum(ctx, part, nparts, 2LL * cout, both);
    if (rc) return rc;
    SSDSEG_HIP(hipMemcpyAsync(dbias, both, (size_t)cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ dense 3x3 (K6)
int ssdseg_conv3x3_parts(int n, int h, int w, int cin, int cout, int* nparts_host) {
    SSDSEG_ARG(n > 0 && h > 0 && w > 0, 1);
    SSDSEG_ARG(cin > 0 && cin % 4 == 0, 4);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 5);
    SSDSEG_ARG(nparts_host != nullptr, 6);
    // the halo-tile kernel writes one partial row per 8 x 32 pixel tile, the implicit-GEMM kernels one per row-tile slot
    if (conv3_tile_fwd_ok(cin, cout) && conv3_wino4_takes(n, h, w, cin, cout)) {      // (one row per block of 4x4-pixel tiles)
        Wino4Geom g;
        wino4_geometry(h, w, &g);
        *nparts_host = n * g.tiles_h * g.tiles_w;
        return 0;
    }
    *nparts_host = conv3_tile_fwd_ok(cin, cout) ? conv3t_geometry(n, h, w, cout).mtiles : rowA_grid_y(n * h * w, cout);
    return 0;
}

int ssdseg_conv3x3_fwd(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int n, int h, int wdt, int cin,
                       int cout, float* stats) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldx >= cin && ldx % 4 == 0, 3);
    SSDSEG_ARG(w != nullptr, 4);
    SSDSEG_ARG(y != nullptr, 5);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(cin > 0 && cin % 4 == 0, 9);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 10);
    if (conv3_narrow(cin, cout)) {
        const long long m = (long long)n * h * wdt;
        const int nc = 9 * cout;
        const size_t wb = align256((size_t)cin * nc * sizeof(float)), zb = align256((size_t)m * nc * sizeof(float));
        void* ws;
        int rc = conv3n_scratch(ctx, wb + zb, &ws);
        if (rc) return rc;
        float* w2 = (float*)ws;
        float* z = (float*)((char*)ws + wb);
        SSDSEG_LAUNCH(ctx, 8.0 * 9 * cin * cout, 0.0, conv3n_pack_w_kernel, dim3(cdiv(9 * cin * cout, 256)), dim3(256), 0, w, w2, cin, cout, 0);
        SSDSEG_LAUNCH_CHECK();
        ctx->ws_reserved += wb + zb;
        rc = ssdseg_pwconv_fwd(ctx, in, ldx, w2, z, nc, (int)m, cin, nc, nullptr);
        ctx->ws_reserved -= wb + zb;
        if (rc) return rc;
        int nparts = 0;
        ssdseg_conv3x3_parts(n, h, wdt, cin, cout, &nparts);
        const int cv = cout / 4;
        SSDSEG_ARG((long long)m * cv < (1LL << 31), 6);   // 32-bit element indices in conv3n_tapsum_kernel
        int blocks = stats != nullptr ? nparts : (int)((m * cv + 255) / 256 < 4096 ? (m * cv + 255) / 256 : 4096);
        SSDSEG_LAUNCH(ctx, 4.0 * m * (nc + cout), 0.0, conv3n_tapsum_kernel, dim3(blocks), dim3(256), 0, (const float*)z, y, n, h, wdt, cv, stats);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    if (conv3_tile_fwd_ok(cin, cout) && conv3_tile_fits(n, h, wdt, ldx)) {
        // weights with the reduction channel contiguous: W[tap][c][n] -> Wt[tap][n][c] (2.8 MB for the decoder conv, ~3 us)
        void* ws;
        int rc = ssdseg_workspace(ctx, (size_t)9 * cin * cout * sizeof(float), &ws);
        if (rc) return rc;
        SSDSEG_LAUNCH(ctx, 8.0 * 9 * cin * cout, 0.0, conv3_transpose_w_kernel, dim3(cdiv(cout, 32), cdiv(cin, 32), 9), dim3(256), 0, w, (float*)ws, cin, cout);
        SSDSEG_LAUNCH_CHECK();
        if (conv3_wino_takes(n, h, wdt, cin, cout)) {
            Conv3TArgs t{};
            t.in = in->x; t.cs = in->scale; t.ct = in->shift; t.act = in->act; t.ldi = ldx;
            t.out = y; t.ldo = cout; t.accumulate = 0;
            t.stats = stats;
            t.n = n; t.h = h; t.w = wdt; t.cred = cin; t.nout = cout; t.flip = 0;
            return conv3_wino_launch(ctx, t, w, cin, cout, 0);
        }
        Conv3TArgs t{};
        t.in = in->x; t.cs = in->scale; t.ct = in->shift; t.act = in->act; t.ldi = ldx;
        t.wt = (const float*)ws;
        t.out = y; t.ldo = cout; t.accumulate = 0;
        t.stats = stats;
        t.n = n; t.h = h; t.w = wdt; t.cred = cin; t.nout = cout; t.flip = 0;
        return conv3t_launch(ctx, t);
    }
    RowAArgs a{};
    a.a0 = in->x; a.cs = in->scale; a.ct = in->shift; a.act = in->act; a.lda = ldx;
    a.b = w; a.ldb = cout;
    a.out = y; a.ldo = cout;
    a.stats = stats;
    a.I = n * h * wdt; a.R = 9 * cin; a.J = cout;
    a.convH = h; a.convW = wdt; a.convC = cin; a.convSign = 1;
    return launch_rowA<0, 1>(ctx, a);
}

// ---- forward that SAVES its activated input for the weight gradient (large Winograd layers; include/ssdseg.h)
int ssdseg_conv3x3_saved_floats(int n, int h, int w, int cin, int cout, long long* floats_host) {
    SSDSEG_ARG(n > 0 && h > 0 && w > 0, 1);
    SSDSEG_ARG(cin > 0 && cin % 4 == 0, 4);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 5);
    SSDSEG_ARG(floats_host != nullptr, 6);
    const bool both = !conv3_narrow(cin, cout) && getenv("SSDSEG_CONV3_WGRAD") == nullptr && getenv("SSDSEG_CONV3_SAVED") == nullptr &&
                      conv3_wino_takes(n, h, w, cin, cout) && conv3_wino_wgrad_takes(n, h, w, cin, cout);
    *floats_host = both ? (long long)n * (h + 2) * (w + 2) * cin : 0;
    return 0;
}

int ssdseg_conv3x3_fwd_saved(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int n, int h, int wdt, int cin, int cout,
                             float* stats, float* xsaved) {
    return ssdseg_conv3x3_fwd_saved_from(ctx, in, ldx, w, y, n, h, wdt, cin, cout, stats, xsaved, 0);
}

int ssdseg_conv3x3_fwd_saved_from(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const float* w, float* y, int n, int h, int wdt, int cin,
                                  int cout, float* stats, float* xsaved, int c_from) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(c_from >= 0 && c_from < cin && c_from % 4 == 0, 13);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldx >= cin && ldx % 4 == 0, 3);
    SSDSEG_ARG(w != nullptr, 4);
    SSDSEG_ARG(y != nullptr, 5);
    SSDSEG_ARG(xsaved != nullptr, 12);
    long long need = 0;
    int rc = ssdseg_conv3x3_saved_floats(n, h, wdt, cin, cout, &need);
    if (rc) return rc;
    SSDSEG_ARG(need > 0, 6);     // only for shapes ssdseg_conv3x3_saved_floats reports a size for
    const long long tot4 = (long long)n * (h + 2) * (wdt + 2) * ((cin - c_from) / 4);
    SSDSEG_LAUNCH(ctx, 8.0 * n * h * wdt * (cin - c_from), 0.0, conv3_pad_view_kernel, dim3((unsigned)((tot4 + 255) / 256 < 16384 ? (tot4 + 255) / 256 : 16384)), dim3(256), 0,
                  in->x, in->scale, in->shift, in->act, ldx, xsaved, n, h, wdt, cin, c_from);
    SSDSEG_LAUNCH_CHECK();
    Conv3TArgs t{};
    t.in = xsaved + ((long long)(wdt + 2) + 1) * cin;      // pixel (1, 1) of image 0 of the zero-bordered copy
    t.act = SSDSEG_ACT_NONE; t.ldi = cin;
    t.in_hp = h + 2; t.in_wp = wdt + 2;
    t.out = y; t.ldo = cout; t.accumulate = 0;
    t.stats = stats;
    t.n = n; t.h = h; t.w = wdt; t.cred = cin; t.nout = cout; t.flip = 0;
    return conv3_wino_launch(ctx, t, w, cin, cout, 0);
}

int ssdseg_conv3x3_bwd_weight_saved(ssdseg_ctx* ctx, const float* xsaved, const float* dy, float* dw, int n, int h, int wdt, int cin, int cout) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(xsaved != nullptr, 2);
    SSDSEG_ARG(dy != nullptr, 3);
    SSDSEG_ARG(dw != nullptr, 4);
    long long need = 0;
    int rc = ssdseg_conv3x3_saved_floats(n, h, wdt, cin, cout, &need);
    if (rc) return rc;
    SSDSEG_ARG(need > 0, 5);
    return conv3_wino_wgrad_launch(ctx, nullptr, cin, dy, dw, n, h, wdt, cin, cout, xsaved);
}

int ssdseg_conv3x3_bwd_data_bn(ssdseg_ctx* ctx, const ssdseg_view* in, const ssdseg_gview* dy, const float* w, float* dx, int ldx, int n,
                               int h, int wdt, int cin, int cout, const float* in_mean, const float* in_invstd, float* in_dgamma,
                               float* in_dbeta, float* in_k1, float* in_k0) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && in->scale != nullptr && in->shift != nullptr, 2);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 3);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 3);
    SSDSEG_ARG(w != nullptr, 4);
    SSDSEG_ARG(dx != nullptr, 5);
    SSDSEG_ARG(ldx >= cin && ldx % 4 == 0, 6);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 7);
    SSDSEG_ARG(cin > 0 && cin % 4 == 0, 10);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 11);
    SSDSEG_ARG(in_mean != nullptr && in_invstd != nullptr, 12);
    SSDSEG_ARG(in_k1 != nullptr && in_k0 != nullptr, 16);
    const long long m = (long long)n * h * wdt;
    if (conv3_narrow(cin, cout) && ssdseg_conv3n_direct_takes(cin, cout, ldx) && m < (1LL << 29))      // conv3n.hip: the streaming form (256 -> 4)
        return ssdseg_conv3n_bwd_bn_direct(ctx, in, dy, w, dx, ldx, n, h, wdt, in_mean, in_invstd, in_dgamma, in_dbeta, in_k1, in_k0);
    if (conv3_narrow(cin, cout)) {
        // tap-expanded form: dz[m][tap][o] = dy[m - d(tap)][o], then dx = dz * W2^T is a pointwise GEMM whose float4 epilogue holds the dx
        // tile and reads the matching tile of the raw input -- that BN's sums ride there (the 256 -> 4 logits conv of the decoder:
        // one pass over the 614,400 x 256 gradient and its raw tensor less)
        const int nc = 9 * cout, cv = cout / 4;
        const size_t wb = align256((size_t)cin * nc * sizeof(float)), zb = align256((size_t)m * nc * sizeof(float));
        void* ws;
        int rc = conv3n_scratch(ctx, wb + zb, &ws);
        if (rc) return rc;
        float* w2 = (float*)ws;
        float* dz = (float*)((char*)ws + wb);
        SSDSEG_LAUNCH(ctx, 8.0 * 9 * cin * cout, 0.0, conv3n_pack_w_kernel, dim3(cdiv(9 * cin * cout, 256)), dim3(256), 0, w, w2, cin, cout, 0);
        SSDSEG_LAUNCH_CHECK();
        const long long tot = m * 9 * cv;
        SSDSEG_ARG(tot < (1LL << 31), 6);       // 32-bit element indices in conv3n_shift_kernel
        SSDSEG_LAUNCH(ctx, 4.0 * m * (nc + (dy->scale ? 2.0 : 1.0) * cout), 0.0, conv3n_shift_kernel, dim3((unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192)),
                      dim3(256), 0, dy->g, dy->y, dy->scale, dy->shift, dy->k1, dy->k0, dy->act, dz, n, h, wdt, cv);
        SSDSEG_LAUNCH_CHECK();
        ssdseg_gview idv{};
        idv.g = dz;
        ctx->ws_reserved += wb + zb;
        rc = pwconv_bwd_data_bn(ctx, in, ldx, &idv, nc, w2, dx, ldx, (int)m, cin, nc, in_mean, in_invstd, in_dgamma, in_dbeta, in_k1, in_k0);
        ctx->ws_reserved -= wb + zb;
        return rc;
    }
    int rc = ssdseg_conv3x3_bwd_data(ctx, dy, w, dx, ldx, n, h, wdt, cin, cout, 0);
    if (rc) return rc;
    return ssdseg_bn_bwd_reduce(ctx, dx, ldx, in->x, ldx, (int)m, cin, in->scale, in->shift, in_mean, in_invstd, in->act, in_dgamma, in_dbeta, in_k1, in_k0);
}

int ssdseg_conv3x3_bwd_data(ssdseg_ctx* ctx, const ssdseg_gview* dy, const float* w, float* dx, int ldx, int n, int h, int wdt,
                            int cin, int cout, int accumulate) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 2);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 2);
    SSDSEG_ARG(w != nullptr, 3);
    SSDSEG_ARG(dx != nullptr, 4);
    SSDSEG_ARG(ldx >= cin, 5);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(cin > 0 && cin % 4 == 0, 9);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 10);
    if (conv3_narrow(cin, cout)) {
        const long long m = (long long)n * h * wdt;
        const int nc = 9 * cout, cv = cout / 4;
        const size_t wb = align256((size_t)cin * nc * sizeof(float)), zb = align256((size_t)m * nc * sizeof(float));
        void* ws;
        int rc = conv3n_scratch(ctx, wb + zb, &ws);
        if (rc) return rc;
        float* w2 = (float*)ws;
        float* dz = (float*)((char*)ws + wb);
        SSDSEG_LAUNCH(ctx, 8.0 * 9 * cin * cout, 0.0, conv3n_pack_w_kernel, dim3(cdiv(9 * cin * cout, 256)), dim3(256), 0, w, w2, cin, cout, 0);
        SSDSEG_LAUNCH_CHECK();
        const long long tot = m * 9 * cv;
        SSDSEG_ARG(tot < (1LL << 31), 6);       // 32-bit element indices in conv3n_shift_kernel
        SSDSEG_LAUNCH(ctx, 4.0 * m * (nc + (dy->scale ? 2.0 : 1.0) * cout), 0.0, conv3n_shift_kernel, dim3((unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192)),
                      dim3(256), 0, dy->g, dy->y, dy->scale, dy->shift, dy->k1, dy->k0, dy->act, dz, n, h, wdt, cv);
        SSDSEG_LAUNCH_CHECK();
        ssdseg_gview idv{};
        idv.g = dz;
        ctx->ws_reserved += wb + zb;
        rc = ssdseg_pwconv_bwd_data(ctx, &idv, nc, w2, dx, ldx, (int)m, cin, nc, nullptr, 0, accumulate);
        ctx->ws_reserved -= wb + zb;
        return rc;
    }
    if (conv3_tile_enabled() && dy->scale == nullptr && cout % C3T_KC == 0 && conv3_tile_fits(n, h, wdt, cout)) {
        // dx[p][c] = sum_{tap, n} dy[p - d(tap)][n] W[tap][c][n]: the forward loop over the mirrored taps; W's native layout already has
        // the reduction channel (n) contiguous.  (A BatchNorm gradient view is materialised by the caller first: nine taps would
        // each re-form it -- ssdseg_gview_materialize.)
        if (conv3_wino_takes(n, h, wdt, cout, cin)) {
            Conv3TArgs t{};
            t.in = dy->g; t.act = SSDSEG_ACT_NONE; t.ldi = cout;
            t.out = dx; t.ldo = ldx; t.accumulate = accumulate;
            t.n = n; t.h = h; t.w = wdt; t.cred = cout; t.nout = cin; t.flip = 0;
            return conv3_wino_launch(ctx, t, w, cin, cout, 1);
        }
        Conv3TArgs t{};
        t.in = dy->g; t.act = SSDSEG_ACT_NONE; t.ldi = cout;
        t.wt = w;
        t.out = dx; t.ldo = ldx; t.accumulate = accumulate;
        t.n = n; t.h = h; t.w = wdt; t.cred = cout; t.nout = cin; t.flip = 1;
        return conv3t_launch(ctx, t);
    }
    RowAArgs a{};
    a.a0 = dy->g; a.a1 = dy->y; a.cs = dy->scale; a.ct = dy->shift; a.ck1 = dy->k1; a.ck0 = dy->k0; a.act = dy->act;
    a.lda = cout;
    a.b = w; a.ldb = cout;
    a.out = dx; a.ldo = ldx;
    a.accumulate = accumulate;
    a.I = n * h * wdt; a.R = 9 * cout; a.J = cin;
    a.convH = h; a.convW = wdt; a.convC = cout; a.convSign = -1;   // dx(h,w) gathers dy(h-(kh-1), w-(kw-1))
    return launch_rowA<1, 1>(ctx, a);
}

int ssdseg_conv3x3_bwd_weight(ssdseg_ctx* ctx, const ssdseg_view* in, int ldx, const ssdseg_gview* dy, float* dw, int n, int h,
                              int wdt, int cin, int cout) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(in != nullptr && in->x != nullptr && ((in->scale == nullptr) == (in->shift == nullptr)), 2);
    SSDSEG_ARG(ldx >= cin && ldx % 4 == 0, 3);
    SSDSEG_ARG(dy != nullptr && dy->g != nullptr, 4);
    SSDSEG_ARG(dy->scale == nullptr || (dy->y && dy->shift && dy->k1 && dy->k0), 4);
    SSDSEG_ARG(dw != nullptr, 5);
    SSDSEG_ARG(n > 0 && h > 0 && wdt > 0, 6);
    SSDSEG_ARG(cin > 0 && cin % 4 == 0, 9);
    SSDSEG_ARG(cout > 0 && cout % 4 == 0, 10);
    if (conv3_narrow(cin, cout)) {
        const long long m = (long long)n * h * wdt;
        const int nc = 9 * cout, cv = cout / 4;
        const size_t wb = align256((size_t)cin * nc * sizeof(float)), zb = align256((size_t)m * nc * sizeof(float));
        void* ws;
        int rc = conv3n_scratch(ctx, wb + zb, &ws);
        if (rc) return rc;
        float* dw2 = (float*)ws;
        float* dz = (float*)((char*)ws + wb);
        const long long tot = m * 9 * cv;
        SSDSEG_ARG(tot < (1LL << 31), 6);       // 32-bit element indices in conv3n_shift_kernel
        SSDSEG_LAUNCH(ctx, 4.0 * m * (nc + (dy->scale ? 2.0 : 1.0) * cout), 0.0, conv3n_shift_kernel, dim3((unsigned)((tot + 255) / 256 < 8192 ? (tot + 255) / 256 : 8192)),
                      dim3(256), 0, dy->g, dy->y, dy->scale, dy->shift, dy->k1, dy->k0, dy->act, dz, n, h, wdt, cv);
        SSDSEG_LAUNCH_CHECK();
        WGradArgs a{};
        a.x = in->x; a.xs = in->scale; a.xt = in->shift; a.xact = in->act; a.ldx = ldx;
        a.g = dz; a.ldy = nc;
        a.M = (int)m; a.K = cin; a.N = nc;
        ctx->ws_reserved += wb + zb;
        ssdseg_defer_hold(ctx, +1);           // dw2 is scratch, repacked right below: its column sum cannot wait for the flush
        rc = wgrad_run(ctx, a, dw2);
        ssdseg_defer_hold(ctx, -1);
        ctx->ws_reserved -= wb + zb;
        if (rc) return rc;
        SSDSEG_LAUNCH(ctx, 8.0 * 9 * cin * cout, 0.0, conv3n_pack_w_kernel, dim3(cdiv(9 * cin * cout, 256)), dim3(256), 0, (const float*)dw, dw2, cin, cout, 1);
        SSDSEG_LAUNCH_CHECK();
        return 0;
    }
    const char* c3env = getenv("SSDSEG_CONV3_WGRAD");   // "taps": the nine shifted GEMMs (A/B measurements, parity tests)
    if (c3env == nullptr && dy->scale == nullptr && conv3_wino_wgrad_takes(n, h, wdt, cin, cout))
        return conv3_wino_wgrad_launch(ctx, in, ldx, dy->g, dw, n, h, wdt, cin, cout);
    if (c3env == nullptr && conv3_tile_enabled() && dy->scale == nullptr && conv3_tile_fits(n, h, wdt, ldx) && conv3_tile_fits(n, h, wdt, cout)) {
        // halo-tile form (conv3_wgrad_tile.h): 64 x 64 (k, n) tiles of all nine taps, one image row x 32 columns per step
        Wg3TArgs a{};
        a.x = in->x; a.xs = in->scale; a.xt = in->shift; a.xact = in->act; a.ldx = ldx;
        a.g = dy->g;
        a.n = n; a.h = h; a.w = wdt; a.K = cin; a.N = cout;
        a.ktiles = cdiv(cin, W3T_KT); a.ntiles = cdiv(cout, W3T_NT);
        a.strips = cdiv(wdt, W3T_COLS);
        a.steps = n * a.strips * h;
        const int tiles = a.ktiles * a.ntiles;
        int splits = 2 * ctx->num_cus / tiles;                // two blocks per CU (67 KB of LDS, <= 256 registers each)
        if (splits > a.steps / 4) splits = a.steps / 4;
        if (splits < 1) splits = 1;
        a.steps_per_split = (a.steps + splits - 1) / splits;
        splits = (a.steps + a.steps_per_split - 1) / a.steps_per_split;
        a.x_bytes = (unsigned)((((long long)n * h * wdt - 1) * ldx + cin) * 4);
        a.g_bytes = (unsigned)((long long)n * h * wdt * cout * 4);
        void* ws;
        int rc = ssdseg_partials(ctx, (size_t)splits * 9 * cin * cout * sizeof(float), &ws);
        if (rc) return rc;
        a.part = (float*)ws;
        static bool configured = false;   // dynamic LDS beyond 64 KiB has to be announced once
        if (!configured) {
            SSDSEG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgrad_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3T_LDS_BYTES));
            configured = true;
        }
        const double m = (double)n * h * wdt;
        const double cost_bytes = 4.0 * (m * cin + m * cout + 9.0 * cin * cout);   // X + dY + dW
        const double cost_flops = 18.0 * m * cin * cout;
        SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, conv3_wgrad_tile_kernel, dim3((unsigned)(tiles * splits)), dim3(W3T_THREADS), W3T_LDS_BYTES, a);
        SSDSEG_LAUNCH_CHECK();
        if (splits == 1) {
            SSDSEG_HIP(hipMemcpyAsync(dw, a.part, (size_t)9 * cin * cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            return 0;
        }
        return ssdseg_colsum(ctx, a.part, splits, 9LL * cin * cout, dw);
    }
    if (!(c3env != nullptr && !strcmp(c3env, "taps"))) {
        // all nine taps in one pass (conv3_wgrad.h)
        Conv9Args a{};
        a.x = in->x; a.xs = in->scale; a.xt = in->shift; a.xact = in->act; a.ldx = ldx;
        a.g = dy->g; a.y = dy->y; a.gs = dy->scale; a.gt = dy->shift; a.gk1 = dy->k1; a.gk0 = dy->k0; a.gact = dy->act;
        a.n = n; a.h = h; a.w = wdt; a.K = cin; a.N = cout;
        a.wchunks = cdiv(wdt, C9_PX);
        a.steps = (long long)n * h * a.wchunks;
        const int wn = cout > 32 ? 4 : 1;
        const int gx = cdiv(cout, 32 * wn), gy = cdiv(cin, C9_KT);
        long long splits = (2LL * ctx->num_cus) / ((long long)gx * gy);
        if (splits > a.steps / 8) splits = a.steps / 8;
        if (splits < 1) splits = 1;
        a.steps_per_split = (a.steps + splits - 1) / splits;
        splits = (a.steps + a.steps_per_split - 1) / a.steps_per_split;
        void* ws;
        int rc = ssdseg_partials(ctx, (size_t)splits * 9 * cin * cout * sizeof(float), &ws);
        if (rc) return rc;
        a.part = (float*)ws;
        const dim3 grid(gx, gy, (unsigned)splits);
        const size_t lds = (size_t)(C9_PX * (32 * wn + 4) + 3 * C9_XW * C9_XS) * sizeof(float);
        const double m = (double)n * h * wdt;
        const double cost_bytes = 4.0 * (m * cin + m * cout + 9.0 * cin * cout);   // 8(d): X + dY + dW
        ctx->timing_view_bytes = dy->scale != nullptr ? 4.0 * m * cout : 0.0;
        const double cost_flops = 18.0 * m * cin * cout;
        const char* w12 = getenv("SSDSEG_CONV3_WGRAD");   // "nine": the nine-wave kernel for the 128-column tiles as well
        if (wn == 4 && !(w12 != nullptr && !strcmp(w12, "nine")))
            SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, conv3_wgrad12_kernel, grid, dim3(C12_THREADS), lds, a);
        else if (wn == 4) SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (conv3_wgrad9_kernel<4>), grid, dim3(C9_THREADS), lds, a);
        else SSDSEG_LAUNCH(ctx, cost_bytes, cost_flops, (conv3_wgrad9_kernel<1>), grid, dim3(C9_THREADS), lds, a);
        SSDSEG_LAUNCH_CHECK();
        if (splits == 1) {
            SSDSEG_HIP(hipMemcpyAsync(dw, a.part, (size_t)9 * cin * cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            return 0;
        }
        return ssdseg_colsum(ctx, a.part, (int)splits, 9LL * cin * cout, dw);
    }
    for (int tap = 0; tap < 9; ++tap) {
        WGradArgs a{};
        a.x = in->x; a.xs = in->scale; a.xt = in->shift; a.xact = in->act; a.ldx = ldx;
        a.g = dy->g; a.y = dy->y; a.gs = dy->scale; a.gt = dy->shift; a.gk1 = dy->k1; a.gk0 = dy->k0; a.gact = dy->act;
        a.ldy = cout;
        a.M = n * h * wdt; a.K = cin; a.N = cout;
        a.convH = h; a.convW = wdt; a.dh = tap / 3 - 1; a.dw = tap % 3 - 1;
        int rc = wgrad_run(ctx, a, dw + (size_t)tap * cin * cout);
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
