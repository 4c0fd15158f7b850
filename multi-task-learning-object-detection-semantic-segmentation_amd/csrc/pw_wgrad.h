// Weight gradient of a pointwise (1x1) convolution:  dW[k][n] = sum_m view(x)[m][k] * gview(g, y)[m][n]   (reference blocks.py: every
// Conv2D(1x1)) -- included by gemm.hip inside its anonymous namespace, after pw_tile.h.
//
// The reduction index m (the pixel) is the STRIDED index of both NHWC operands.  gemm_wgrad_kernel stages [pixel][channel] tiles
// and reads its MFMA operands from them one float at a time -- two ds_read_b32 and some address arithmetic per MFMA -- and
// re-forms the views with 64-bit addressing and per-element branches: 20-70 TFLOP/s, 1-3 TB/s on the narrow layers.
// Here the MFMA's freedom to NAME its rows does the transposition (as in conv3_wino_wgrad.h): a lane is (group r of JX input
// channels | group r' of JY output channels, pixel parity kk); ONE ds_read of JX floats of x and ONE of JY floats of dy feed
//     acc[jx][jy] += mfma_32x32x2( x.comp[jx], dy.comp[jy] )         JX * JY MFMAs per pair of reads
// where MFMA row r MEANS input channel JX r + jx and column r' MEANS output channel JY r' + jy; the instruction's two k-slots
// are the two pixels of the pair.  Block = eight waves = (G row groups) x (WK x WN wave tiles of 32 JX x 32 JY): wide layers
// use G = 1 and a 256 x 128 / 128 x 256 block tile, narrow ones a tile that covers ALL their channels (each operand element
// is then read from HBM exactly once) with the eight waves splitting the pixels (their partial tiles are folded through LDS in a
// fixed order at the end); every split writes its own partial slab, the fixed-order column sum (ssdseg_colsum) adds them.
// Staging: raw buffer loads (32-bit offsets, hardware range check) one step ahead, views applied once per element while
// committing to LDS ([pixel][channel], the natural layout: no swizzle needed -- a wave's read is one contiguous run), two LDS
// buffers, one barrier per step.  Rows past the end of a split are zeroed on the dy side only (a zero factor is enough).
#pragma once
// store form of the JY == 4 epilogue: 3 (default) one 16-byte store with an immediate soffset | 0 two 8-byte stores | 1, 2: the
// two forms of scripts/dbg/b128_experiment.sh (1 reproduces the round-2 wrong results)
#ifndef PWW_STORE
#define PWW_STORE 3
#endif

struct PwWgArgs {
    const float* x;      // [M][ldx] raw input
    const float* xs;     // view: act(xs * x + xt); nullptr = identity
    const float* xt;
    int xact, ldx;
    const float* g;      // [M][ldy] incoming gradient
    const float* y;      // gradient view: gs * mask(gs * y + gt) * g + gk1 * y + gk0; gs == nullptr = identity
    const float* gs;
    const float* gt;
    const float* gk1;
    const float* gk0;
    int gact, ldy;
    float* part;         // [splits][K][N]
    int M, K, N;
    int rows_per_split;  // multiple of the step's rows
    unsigned x_bytes, g_bytes, part_bytes;
};

template <int N> struct pww_vec;
template <> struct pww_vec<1> { typedef float type; };
template <> struct pww_vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct pww_vec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int N> __device__ __forceinline__ float pww_comp(const typename pww_vec<N>::type& v, int i) { return v[i]; }
template <> __device__ __forceinline__ float pww_comp<1>(const float& v, int) { return v; }

constexpr int pww_ms(int kt, int nt) { return (kt + nt) > 256 ? 16 : ((kt + nt) > 128 ? 32 : 64); }     // rows per step: <= 24 KB per buffer
constexpr size_t pww_lds_bytes(int kt, int nt, int g) {
    const size_t loop = (2 * (size_t)pww_ms(kt, nt) * (kt + nt) + 2 * (size_t)kt + 4 * (size_t)nt) * sizeof(float);
    const size_t fold = g > 1 ? (size_t)(g / 2) * kt * nt * sizeof(float) : 0;      // the upper half of the row groups parks its tile
    return loop > fold ? loop : fold;
}

template <int JX, int JY, int WK, int WN>
__global__ void __launch_bounds__(512, 2) pw_wgrad_kernel(PwWgArgs p) {
    constexpr int G = 8 / (WK * WN);
    constexpr int KT = 32 * JX * WK, NT = 32 * JY * WN;
    constexpr int MS = pww_ms(KT, NT);
    constexpr int XV = MS * KT / 4, YV = MS * NT / 4;         // float4 slots per step
    constexpr int XQ = (XV + 511) / 512, YQ = (YV + 511) / 512;
    constexpr int BUF_F = MS * (KT + NT);
    constexpr int PAIRS = MS / 2 / G;                          // pixel pairs per wave and step
    static_assert(G * WK * WN == 8 && PAIRS >= 1 && XV % 64 == 0 && YV % 64 == 0, "eight waves; whole waves of staging slots");
    typedef typename pww_vec<JX>::type xvec;
    typedef typename pww_vec<JY>::type yvec;
    extern __shared__ float smem[];
    float* xc = smem + 2 * BUF_F;      // [2][KT]: scale, shift of the x view
    float* yc = xc + 2 * KT;           // [4][NT]: scale, shift, k1, k0 of the gradient view

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, r = lane & 31, kk = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wn = wave_u % WN, wk = (wave_u / WN) % WK, grp = wave_u / (WN * WK);
    const int k0 = blockIdx.y * KT, n0 = blockIdx.x * NT;
    const int split = blockIdx.z;
    const int mbeg = split * p.rows_per_split;
    int mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;

    const bool xaff = p.xs != nullptr || p.xact != SSDSEG_ACT_NONE, gaff = p.gs != nullptr;
    const float xlo = act_lo(p.xact), xhi = act_hi(p.xact);
    for (int i = t; i < KT; i += 512) {
        const bool ok = p.xs != nullptr && k0 + i < p.K;
        xc[i] = ok ? p.xs[k0 + i] : 1.f;
        xc[KT + i] = ok ? p.xt[k0 + i] : 0.f;
    }
    for (int i = t; i < NT; i += 512) {
        const bool ok = gaff && n0 + i < p.N;
        yc[i] = ok ? p.gs[n0 + i] : 1.f;
        yc[NT + i] = ok ? p.gt[n0 + i] : 0.f;
        yc[2 * NT + i] = ok ? p.gk1[n0 + i] : 0.f;
        yc[3 * NT + i] = ok ? p.gk0[n0 + i] : 0.f;
    }

    // ---- staging slots (fixed per thread): slot -> (row of the step, 4-channel quad); the step's first row rides in the scalar offset
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gaff ? p.y : p.g), 0, p.g_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned xgo[XQ], ygo[YQ];
    int xrow[XQ], yrow[YQ], xcq[XQ], ycq[YQ];
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
        const int slot = t + 512 * q;
        xrow[q] = slot / (KT / 4);
        xcq[q] = slot % (KT / 4);
        const bool ok = slot < XV && k0 + 4 * xcq[q] < p.K;
        xgo[q] = ok ? (unsigned)(((long long)xrow[q] * p.ldx + k0 + 4 * xcq[q]) * 4) : OOB;
    }
#pragma unroll
    for (int q = 0; q < YQ; ++q) {
        const int slot = t + 512 * q;
        yrow[q] = slot / (NT / 4);
        ycq[q] = slot % (NT / 4);
        const bool ok = slot < YV && n0 + 4 * ycq[q] < p.N;
        ygo[q] = ok ? (unsigned)(((long long)yrow[q] * p.ldy + n0 + 4 * ycq[q]) * 4) : OOB;
    }
    float4 xreg[XQ], greg[YQ], yreg[YQ];
    // rows of the step beyond the end of the split: their loads are pushed out of range (zeros) by adding 2^31 to the offset
    auto issue = [&](int m0) __attribute__((always_inline)) {
        const int xso = m0 * p.ldx * 4, yso = m0 * p.ldy * 4;      // (the launcher keeps M * ld * 4 below 2^31)
#pragma unroll
        for (int q = 0; q < XQ; ++q) xreg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, xgo[q], xso, 0));
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            const unsigned off = ygo[q] | ((m0 + yrow[q] < mend) ? 0u : OOB);
            greg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rg, off, yso, 0));
            if (gaff) yreg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, yso, 0));
        }
    };
    auto commit = [&](int m0, int boff) __attribute__((always_inline)) {
        float* bx = smem + boff;
        float* by = smem + boff + MS * KT;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            if (XV % 512 == 0 || t + 512 * q < XV) {
                float4 v = xreg[q];
                if (xaff) v = view_affine4(v, ld4(xc + 4 * xcq[q]), ld4(xc + KT + 4 * xcq[q]), xlo, xhi);
                st4(bx + (t + 512 * q) * 4, v);
            }
        }
#pragma unroll
        for (int q = 0; q < YQ; ++q) {
            if (YV % 512 == 0 || t + 512 * q < YV) {
                float4 v = greg[q];
                if (gaff) {
                    v = gview_apply4(greg[q], yreg[q], ld4(yc + 4 * ycq[q]), ld4(yc + NT + 4 * ycq[q]), ld4(yc + 2 * NT + 4 * ycq[q]),
                                     ld4(yc + 3 * NT + 4 * ycq[q]), p.gact);
                    const float keep = (m0 + yrow[q] < mend) ? 1.f : 0.f;      // gview(0, 0) = k0, not 0
                    v = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
                }
                st4(by + (t + 512 * q) * 4, v);
            }
        }
    };

    f32x16 acc[JX][JY];
#pragma unroll
    for (int a = 0; a < JX; ++a)
#pragma unroll
        for (int b = 0; b < JY; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // lane bases (floats from the buffer start): row of the pair = 2 * (grp + G * i) + kk
    const int xlb = (2 * grp + kk) * KT + wk * 32 * JX + JX * r;
    const int ylb = MS * KT + (2 * grp + kk) * NT + wn * 32 * JY + JY * r;
    auto compute = [&](int boff) __attribute__((always_inline)) {
        const float* buf = smem + boff;
        xvec xv[2];
        yvec yv[2];
        xv[0] = *reinterpret_cast<const xvec*>(buf + xlb);
        yv[0] = *reinterpret_cast<const yvec*>(buf + ylb);
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int c = i & 1;
            if (i + 1 < PAIRS) {
                xv[c ^ 1] = *reinterpret_cast<const xvec*>(buf + xlb + (i + 1) * 2 * G * KT);
                yv[c ^ 1] = *reinterpret_cast<const yvec*>(buf + ylb + (i + 1) * 2 * G * NT);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < JX; ++a)
#pragma unroll
                for (int b = 0; b < JY; ++b) acc[a][b] = mfma32(pww_comp<JX>(xv[c], a), pww_comp<JY>(yv[c], b), acc[a][b]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- pipeline: buffer (s & 1) holds step s; the loads of step s+1 fly during the MFMAs of step s.  One loop body, one exit.
    const int nsteps = (mend - mbeg + MS - 1) / MS;
    issue(mbeg);
    __syncthreads();                    // coefficient tables visible
    commit(mbeg, 0);
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
        const int mn = mbeg + (s + 1) * MS;      // past the end for the last step: every row zeroed / out of range, never consumed
        issue(mn);
        compute(cur * BUF_F);
        commit(mn, (cur ^ 1) * BUF_F);
        __syncthreads();
        cur ^= 1;
    }

    // ---- the G row groups of the block hold partial sums of the SAME tile: fold them through LDS in log2(G) rounds (upper half
    // writes, lower half adds: a fixed order), so that a block leaves ONE slab -- the column sum over the slabs reads G times less
    if (G > 1) {
        constexpr int WT = JX * JY * 16 * 64;                  // floats of one wave's accumulators
        const int wpos = wk * WN + wn;
#pragma unroll
        for (int h = G / 2; h >= 1; h >>= 1) {
            __syncthreads();                                    // the operand buffers (first round) / the previous round's slots are free
            if (grp >= h && grp < 2 * h) {
                float* dst = smem + ((grp - h) * (WK * WN) + wpos) * WT + lane;
#pragma unroll
                for (int a = 0; a < JX; ++a)
#pragma unroll
                    for (int b = 0; b < JY; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) dst[((a * JY + b) * 16 + e) * 64] = acc[a][b][e];
            }
            __syncthreads();
            if (grp < h) {
                const float* src = smem + (grp * (WK * WN) + wpos) * WT + lane;
#pragma unroll
                for (int a = 0; a < JX; ++a)
#pragma unroll
                    for (int b = 0; b < JY; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[a][b][e] += src[((a * JY + b) * 16 + e) * 64];
            }
        }
        if (grp != 0) return;
    }

    // ---- partial slab of this split.  C/D layout: column = lane & 31 (group r' of output channels), row = (e & 3) + 8 * (e >> 2) + 4 * kk
    // (group r of input channels): element (k, n) = (kw0 + JX row + a, nw0 + JY col + b); a lane's JY outputs are adjacent in memory.
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, p.part_bytes, 0x00020000);
    const int kw0 = k0 + wk * 32 * JX, nw0 = n0 + wn * 32 * JY;
    const int nn = nw0 + JY * r;
    const int slab = split * p.K;
    const unsigned lane_off = (unsigned)(((long long)(slab + kw0 + 4 * JX * kk) * p.N + nn) * 4);
    const int krow = kw0 + 4 * JX * kk;
#pragma unroll
    for (int a = 0; a < JX; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dr = JX * ((e & 3) + 8 * (e >> 2)) + a;
            const bool ok = krow + dr < p.K && nn < p.N;          // N is a multiple of 4 and nn of JY: the lane's JY outputs are in range together
            const unsigned off = ok ? lane_off : OOB;
            const int so = dr * p.N * 4;
            if (JY == 1) {
                const float v1 = acc[a][0][e];      // (through a scalar: a bit_cast of the vector element stores element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rp, off, so, 0);
            } else if (JY == 2) {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v;
                v.x = acc[a][0][e]; v.y = acc[a][JY > 1 ? 1 : 0][e];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rp, off, so, 0);
            } else if (PWW_STORE == 0) {
                // two 8-byte stores (no more than 64 bits of data per store: outside the hazard below)
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v, w2;
                v.x = acc[a][0][e]; v.y = acc[a][JY > 1 ? 1 : 0][e]; w2.x = acc[a][JY > 2 ? 2 : 0][e]; w2.y = acc[a][JY > 3 ? 3 : 0][e];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rp, off, so, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, w2), rp, off, so + 8, 0);
            } else {
                // ONE 16-byte store of the lane's four outputs, with the row offset folded into the VECTOR offset (soffset 0).
                // Round 2 had the row offset in an SGPR soffset and saw wrong values in the last four lanes of every sixteen.  Cause
                // (round 3; scripts/micro/store_x4_hazard.hip, profiles/r03_store_x4_hazard.txt): on gfx950 a buffer store of more
                // than 64 bits reads its data registers late -- a VALU write of those registers needs TWO wait states after the
                // store, an SGPR soffset buys ONE.  hipcc inserts the two when soffset is an immediate and nothing when it is a
                // register (LLVM GCNHazardRecognizer::createsVALUHazard exempts MUBUF stores with a register soffset), and here
                // the compiler recycles a data register for the next row's index in the very next slot (`buffer_store_dwordx4
                // v[0:3], v18, s[28:31], s2 offen` / `v_or_b32 v0, 2, v70`).  With an immediate soffset the compiler keeps its own
                // two wait states; scripts/check_store_hazard.py scans the built library for the register-soffset pattern.
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 v;
                v.x = acc[a][0][e]; v.y = acc[a][JY > 1 ? 1 : 0][e]; v.z = acc[a][JY > 2 ? 2 : 0][e]; v.w = acc[a][JY > 3 ? 3 : 0][e];
                if (PWW_STORE == 1) {            // the round-2 form, kept for the experiment only: HAZARDOUS
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rp, off, so, 0);
                } else if (PWW_STORE == 2) {     // the same store followed by one wait state, as inline assembly: exact
                    const unsigned long long pp = (unsigned long long)p.part;
                    u4 rs;
                    rs.x = (unsigned)pp; rs.y = (unsigned)(pp >> 32) & 0xffffu; rs.z = p.part_bytes; rs.w = 0x00020000u;
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n s_nop 1" : : "v"(__builtin_bit_cast(u4, v)), "v"(off), "s"(rs), "s"(so) : "memory");
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rp, ok ? lane_off + (unsigned)so : OOB, 0, 0);
                }
            }
        }
}
