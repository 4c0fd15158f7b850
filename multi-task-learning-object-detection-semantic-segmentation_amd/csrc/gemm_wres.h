// "Weights-resident" pointwise GEMM for the layers whose whole weight slab fits in LDS (<= ~26 KiB: every 1x1 conv of
// MobileNetV2 blocks 0-6, i.e. exactly the HBM-bound ones) -- included by gemm.hip inside its anonymous namespace.
//
// Why a second shape: the general kernel stages a 128-row tile cooperatively and needs two __syncthreads() per 32-deep
// step, so a block has exactly one 16-32 KiB tile in flight and everything waits for the slowest wave; with the fused-dW
// accumulators it drops to one block per CU and 1.3-1.5 TB/s.  Here the weights are loaded into LDS ONCE per block and
// every wave owns its own 32-row tiles and its own [32][36] staging slab, so the streaming loop has no block barrier at
// all: each wave keeps its next 4-8 KiB in flight while it computes, waves drift freely, and 8-12 waves per CU cover HBM
// latency.  Same MFMA scheme as gemm_rowA_kernel (32x32x2 f32, k = 8*kk + jj | 8*kk + 4 + jj per half-wave).
//   MODE 0  y = act(s*x + t) * W            (+ BN statistics of y)
//   MODE 1  dx = dy * W^T (+ residual, + accumulate), dy formed from (g, y) on the way in
//   NT > 0  (MODE 1, WN 1) additionally dW = a^T * dy: one 32x32 accumulator tile per 32-column chunk of dy.
//   (Round 3 also built the fused form with y RECOMPUTED from the narrow input tile and the resident weights instead of read -- 43 %
//   less HBM traffic for 8-16 more MFMAs and one slab round trip per chunk: 558-592 us against 513-522 for the block-1 expand
//   conv.  The kernel was never short of bytes; it was short of loads in flight, see D2 below.  Removed again.)
#pragma once

template <int WN, int MODE, int NT>
__global__ void __launch_bounds__(256, 2) gemm_wres_kernel(RowAArgs p) {   // two blocks per CU: see rowa_min_waves
    constexpr bool FUSEW = NT > 0;
    static_assert(!FUSEW || (MODE == 1 && WN == 1), "fused dW only for backward-data with one column tile");
    constexpr int BN = 32 * WN;
    constexpr int BS = BN + 1;
    extern __shared__ float smem[];
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;
    const int KT = (p.R + BK - 1) / BK;
    const int RP = KT * BK;                       // reduction length padded to whole steps
    float* Ws = smem;                             // [RP][BS] resident weights (zero padded)
    float* Cs = Ws + RP * BS;                     // [4][RP] view coefficients of the streamed operand (s, t, k1, k0)
    float* As = Cs + 4 * RP + wave * (32 * AS);   // this wave's [32][AS] staging slab
    const int j0 = blockIdx.x * BN;
    const bool affine = p.cs != nullptr;

    // ---- once per block: weights and coefficients into LDS
    if (MODE == 0) {
        for (int idx = t; idx < RP * BN; idx += 256) {
            const int r = idx / BN, jl = idx - r * BN;
            const bool ok = r < p.R && j0 + jl < p.J;
            Ws[r * BS + jl] = ok ? p.b[(long long)r * p.ldb + j0 + jl] : 0.f;
        }
    } else {
        for (int idx = t; idx < RP * BN; idx += 256) {
            const int jl = idx / RP, r = idx - jl * RP;
            const bool ok = r < p.R && j0 + jl < p.J;
            Ws[r * BS + jl] = ok ? p.b[(long long)(j0 + jl) * p.ldb + r] : 0.f;
        }
    }
    for (int r = t; r < RP; r += 256) {
        const bool ok = affine && r < p.R;
        Cs[0 * RP + r] = ok ? p.cs[r] : 1.f;
        Cs[1 * RP + r] = ok ? p.ct[r] : 0.f;
        Cs[2 * RP + r] = (ok && MODE == 1) ? p.ck1[r] : 0.f;
        Cs[3 * RP + r] = (ok && MODE == 1) ? p.ck0[r] : 0.f;
    }
    __syncthreads();

    const float alo = act_lo(p.act), ahi = act_hi(p.act);
    const float* ya = affine ? p.a1 : p.a0;                       // identity gradient view: y aliases g, act NONE
    const int gact = affine ? p.act : SSDSEG_ACT_NONE;
    const int a_c4 = lane & 7, a_r = lane >> 3;                   // staging: rows a_r + 8*i, float4 column a_c4 of the step
    const int tiles = (p.I + 31) / 32;
    const int tstride = gridDim.y * 4;

    // Raw loads in flight.  The fused form with 2-4 column chunks keeps TWO steps in flight (D2): with one, a wave waited ~3 us of
    // every ~5 us chunk step for loads it had issued one step -- ~1 us of MFMAs -- earlier (round 3: the kernel was bound by that
    // latency, not by HBM bytes: recomputing y, -43 % traffic, changed nothing; nor did a third block per CU).  Buffer of global step
    // q is q & 1; NT = KT is a compile-time constant in the fused form, so with the tile loop unrolled by two every index is static.
    // (register budget: 256 at two blocks per CU.  NT = 5 has room for the second step OR for the input tile one tile ahead, not for
    // both -- 248 registers either way; measured equal within the box-to-box noise, the tile-ahead form is kept)
    constexpr bool D2 = FUSEW && NT >= 2 && NT <= 4;
    constexpr bool XPRE = FUSEW;
    constexpr int NB = D2 ? 2 : 1;
    float4 sg[NB][4], sy[NB][MODE == 1 ? 4 : 1];
    unsigned sok[NB];
    auto issue = [&](int b, int mt, int kt) {       // b: compile-time after unrolling (registers, not scratch: checked in the resource usage)
        const int r = kt * BK + a_c4 * 4;
        sok[b] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mt * 32 + a_r + 8 * i;
            const bool ok = r < p.R && m < p.I;
            const long long off = ok ? (long long)m * p.lda + r : 0;
            sg[b][i] = ld4(p.a0 + off);
            if (MODE == 1) sy[b][i] = ld4(ya + off);
            sok[b] |= (ok ? 1u : 0u) << i;
        }
    };
    auto commit = [&](int b, int kt) {   // transform the landed step and write it to this wave's slab
        const int r = kt * BK + a_c4 * 4;
        const float4 cs = ld4(Cs + 0 * RP + r), ct = ld4(Cs + 1 * RP + r);
        float4 ck1 = f4(0.f), ck0 = f4(0.f);
        if (MODE == 1) { ck1 = ld4(Cs + 2 * RP + r); ck0 = ld4(Cs + 3 * RP + r); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v;
            if (MODE == 0) v = view_affine4(sg[b][i], cs, ct, alo, ahi);
            else v = gview_apply4(sg[b][i], sy[b][i], cs, ct, ck1, ck0, gact);
            st4(As + (a_r + 8 * i) * AS + a_c4 * 4, ((sok[b] >> i) & 1u) ? v : f4(0.f));
        }
    };

    float ssum[WN], ssq[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) ssum[nt] = ssq[nt] = 0.f;
    f32x16 wacc[FUSEW ? NT : 1];
#pragma unroll
    for (int q = 0; q < (FUSEW ? NT : 1); ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) wacc[q][e] = 0.f;

    int mt = blockIdx.y * 4 + wave;
    // the fused form's input tile a[m0 + 2s + hh][j = li] (the A operand of the dW products) is loaded ONE TILE AHEAD: read at the top
    // of its own tile it was sixteen loads the wave waited a full memory latency for, once per tile, with nothing else to do
    float xraw[FUSEW ? 16 : 1];
    auto xfetch = [&](int tile) {
#pragma unroll
        for (int s = 0; s < (FUSEW ? 16 : 1); ++s) {
            const int m = tile * 32 + 2 * s + hh;
            const bool ok = li < p.J && m < p.I;
            xraw[s] = p.xw[ok ? (long long)m * p.ldxw + li : 0];
        }
    };
    if (XPRE) xfetch(mt < tiles ? mt : 0);
    issue(0, mt < tiles ? mt : 0, 0);
    if (D2) issue(NB - 1, mt < tiles ? mt : 0, 1);          // (NT >= 2: the second step is chunk 1 of the same tile)
    // tiles of this wave.  With two buffers and NT odd the parity of a tile's first step alternates: two tiles per trip of the outer
    // loop, the inner one fully unrolled, so that every buffer index is a constant.
    constexpr int UN = (D2 && (NT & 1)) ? 2 : 1;
    while (mt < tiles) {
#pragma unroll
      for (int un = 0; un < UN; ++un) {
        if (mt >= tiles) break;
        const int par = D2 ? ((un * NT) & 1) : 0;
        const int m0 = mt * 32;
        const int mnext = mt + tstride < tiles ? mt + tstride : mt;   // dummy re-issue of this tile when there is no next one
        float xop[FUSEW ? 16 : 1];
        if (FUSEW) {
            const float xlo = act_lo(p.xwact), xhi = act_hi(p.xwact);
            const bool jok = li < p.J;
            const float xs = (p.xws != nullptr && jok) ? p.xws[li] : 1.f, xt = (p.xws != nullptr && jok) ? p.xwt[li] : 0.f;
            if (!XPRE) xfetch(mt);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int m = m0 + 2 * s + hh;
                const bool ok = jok && m < p.I;
                xop[s] = ok ? fminf(fmaxf(fmaf(xs, xraw[s], xt), xlo), xhi) : 0.f;
            }
            if (XPRE) xfetch(mnext);                                // lands during this tile's NT chunk steps
        }
        f32x16 acc[WN];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

        auto kstep = [&](int kt, f32x16& wtile) {
            const int buf = D2 ? ((par + kt) & 1) : 0;
            commit(buf, kt);
            // the loads of the step NB ahead go out now, into the registers just consumed -- unconditionally (a branch here would
            // make the s_waitcnt placement conservative)
            const int ahead = kt + NB;
            const bool wrap = ahead >= KT;
            issue(buf, wrap ? mnext : mt, wrap ? ahead - KT : ahead);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // slab written by all lanes before any lane reads it
            __builtin_amdgcn_wave_barrier();
            if (FUSEW) {
                const float* dcol = As + hh * AS + li;               // dy[row 2s + hh][32*kt + li]
#pragma unroll
                for (int s = 0; s < 16; ++s) wtile = mfma32(xop[s], dcol[(2 * s) * AS], wtile);
            }
            const float* arow = As + li * AS + 4 * hh;
            const float* bcol = Ws + (kt * BK + 4 * hh) * BS + li;
            // fragments one 8-deep group ahead of their MFMAs (two register sets), as in gemm_rowA_kernel
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
                    for (int nt = 0; nt < WN; ++nt) acc[nt] = mfma32(av[jj], bfr[kk & 1][jj][nt], acc[nt]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // all reads of the slab done before the next commit
            __builtin_amdgcn_wave_barrier();
        };
        if (FUSEW) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
                if (kt < KT) kstep(kt, wacc[kt]);          // (NT == KT in the fused form)
        } else {
            for (int kt = 0; kt < KT; ++kt) kstep(kt, wacc[0]);
        }

        // ---------------- epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int j = j0 + nt * 32 + li;
            if (j < p.J) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (m < p.I) {
                        float v = acc[nt][e];
                        if (MODE == 1) {
                            if (p.residual) v += p.residual[(long long)m * p.ldr + j];
                            if (p.accumulate) v += p.out[(long long)m * p.ldo + j];
                        }
                        p.out[(long long)m * p.ldo + j] = v;
                    }
                }
            }
        }
        if (MODE == 0 && p.stats != nullptr) {   // padded rows / columns are exactly zero
#pragma unroll
            for (int nt = 0; nt < WN; ++nt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { ssum[nt] += acc[nt][e]; ssq[nt] = fmaf(acc[nt][e], acc[nt][e], ssq[nt]); }
            }
        }
        mt += tstride;
      }
    }

    // ---------------- block-level tails (the only barriers after the set-up): dW slab, BN statistics
    float* red = Cs + 4 * RP;   // the four staging slabs, free now: 4 * 32 * AS floats
    if (FUSEW) {
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
    if (MODE == 0 && p.stats != nullptr) {
        __syncthreads();
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
                const float v = red[(0 * 2 + which) * BN + jl] + red[(1 * 2 + which) * BN + jl] + red[(2 * 2 + which) * BN + jl] +
                                red[(3 * 2 + which) * BN + jl];
                p.stats[((long long)blockIdx.y * 2 + which) * p.J + j] = v;
            }
        }
    }
}

// LDS bytes of the resident kernel for a reduction length r and column-tile width wn (0 when it does not fit the budget)
inline size_t wres_lds_bytes(int r, int wn) {
    const int rp = cdiv(r, BK) * BK;
    const size_t stage = (size_t)4 * 32 * AS;                        // also >= the 12 KiB / 8*BN floats the tails need
    const size_t bytes = ((size_t)rp * (32 * wn + 1) + 4 * (size_t)rp + stage) * sizeof(float);
    // 64 KiB: two blocks per CU.  (52 KiB -- three blocks -- until round 3: the decoder's 144 -> 48 backbone conv needs 62.6 KiB and ran
    // as the row-tile GEMM at 190 us; resident: 149 us.  No layer of the models sits between 52 and 64 KiB otherwise.)  SSDSEG_WRES_LDS_KB: A/B runs
    static const size_t cap = getenv("SSDSEG_WRES_LDS_KB") != nullptr ? (size_t)atoi(getenv("SSDSEG_WRES_LDS_KB")) * 1024 : (size_t)64 * 1024;
    return bytes <= cap ? bytes : 0;
}
