// Dense 3x3 stride-1 SAME convolution (DeepLabV3+ decoder, reference blocks.py:117) forward and input gradient in the
// WINOGRAD F(2x2, 3x3) form, fp32 throughout -- included by gemm.hip inside its anonymous namespace.
//
// The halo-tile kernel (conv3_tile.h) runs the nine taps as nine MFMA passes over an LDS patch and sits at 0.85 of the fp32
// MFMA peak: the arithmetic itself is what is left.  With 2x2 output tiles,
//     Y = A^T [ sum_c (G w_c G^T) .* (B^T d_c B) ] A,       B^T, G, A^T the 4x4 / 4x3 / 2x4 matrices below,
// the reduction over input channels becomes SIXTEEN independent GEMMs (one per position of the 4x4 transformed tile) over a
// quarter of the rows: 16 * (M / 4) * K * N multiply-adds instead of 9 * M * K * N -- 2.25x fewer MFMAs.  B^T and A^T hold only
// 0 / +-1 and G 0 / +-1 / +-1/2: the transforms are additions (and exact halvings); measured against float64 the result's
// error is ~2x that of the direct fp32 sum (7e-7 vs 3e-7 of the output scale at K = 304; tests hold it to 1e-5).
//
// One block = eight waves = an 8-row x 32-column patch of output pixels (4 x 16 = 64 Winograd tiles) x 64 output channels x
// all 16 positions: 64 x 64 x 16 accumulators = 128 registers per lane; wave v owns positions 2v, 2v+1.  Per step of 8 input
// channels a block
//   stages   the (8+2) x (32+2) input halo (view applied while staging, zero padding after the view) and the 16 x 64 x 8
//            slice of the transformed weights U = G w G^T (pre-computed per launch by conv3_wino_weights_kernel) -> LDS;
//   forms    V = B^T d B for its 64 tiles x 8 channels: thread (a, 4-channel quad, tile) reads the two patch rows that row a
//            of B^T combines (8 ds_read_b128), 32 additions, 4 ds_write_b128 -- the transformed input never exists in HBM;
//   runs     per wave 2 positions x (2 x 2) 32x32 tiles x 4 MFMAs on 16-byte fragments (channel pairs (j, 4+j) as in
//            conv3_tile.h): 8 ds_read_b128 per 32 v_mfma_f32_32x32x2_f32.
// Three-deep software pipeline, ONE barrier per step: in iteration s the global loads of patch s+2 / weights s+1 are in
// flight, patch s+1 (staged in iteration s-1) is transformed into V[(s+1)&1], the MFMAs consume V[s&1], U[s&1].
// Epilogue: the accumulators go through LDS once (two halves of 32 output channels), each thread applies A^T . A to the 16
// positions of its (tile, channel) pairs, writes the 2x2 outputs and -- forward -- accumulates the BatchNorm partial sums of
// the block (one partial row per pixel tile, fixed order, no atomics), exactly like the halo-tile kernel.
//
//   forward:   in = x (raw + view), U[k][n][c] from w[i][j][c][n]
//   backward:  in = dy,             U[k][c][n] from w[2-i][2-j][c][n]   (the mirrored, transposed filter)
#pragma once

constexpr int WINO_NT = 64;                          // output channels per block
constexpr int WINO_TILES = 64;                       // 2x2 output tiles per block (4 rows x 16 columns of tiles)
constexpr int WINO_V_F = 16 * WINO_TILES * C3T_KC;   // floats of one V buffer  [k][tile][8]
constexpr int WINO_U_F = 16 * WINO_NT * C3T_KC;      // floats of one U buffer  [k][n][8]
constexpr int WINO_MS_LD = 33;                       // epilogue: M rows of 32 channels, padded (the two lane halves sit 4 tiles apart)
constexpr size_t wino_lds_floats(int cred) { return 2 * (size_t)(C3T_PATCH_F + WINO_V_F + WINO_U_F) + 2 * (size_t)(cred + 16); }
static_assert(16 * WINO_TILES * WINO_MS_LD + 2 * 2 * 16 * 32 <= 2 * (C3T_PATCH_F + WINO_V_F + WINO_U_F), "epilogue staging fits the operand buffers");

__global__ void __launch_bounds__(C3T_THREADS, 2) conv3_wino_kernel(Conv3TArgs p) {
    constexpr int PSLOTS = C3T_PIX * 2;                // float4 slots of a patch (680)
    constexpr int PQ = (PSLOTS + C3T_THREADS - 1) / C3T_THREADS;
    constexpr int USLOTS = 16 * WINO_NT * 2;           // float4 slots of a weight slice (2048)
    constexpr int UQ = USLOTS / C3T_THREADS;
    extern __shared__ float smem[];
    // buffer offsets (floats from smem; integers, so that every access stays a DS instruction through the buffer swaps)
    constexpr int RAW0 = 0, RAW1 = C3T_PATCH_F, V0 = 2 * C3T_PATCH_F, V1 = V0 + WINO_V_F, U0 = V1 + WINO_V_F, U1 = U0 + WINO_U_F;
    float* coef = smem + U1 + WINO_U_F;                // [2][cred + 16]: scale, shift of the input view (+ spare steps)
    const int cld = p.cred + 16;

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;

    const unsigned total = gridDim.x;
    unsigned L = blockIdx.x;
    if ((total & 7u) == 0u) L = (L & 7u) * (total >> 3) + (L >> 3);      // column tiles + neighbouring pixel tiles on one XCD
    const int ntile = (int)(L % (unsigned)p.ntiles_n);
    const int mtile = (int)(L / (unsigned)p.ntiles_n);
    const int tw = mtile % p.tiles_w;
    const int th = (mtile / p.tiles_w) % p.tiles_h;
    const int img = mtile / (p.tiles_w * p.tiles_h);
    const int h0 = th * C3T_ROWS, w0 = tw * C3T_COLS;
    const int n0 = ntile * WINO_NT;

    const bool affine = p.cs != nullptr;
    const float alo = act_lo(p.act), ahi = act_hi(p.act);
    for (int i = t; i < cld; i += C3T_THREADS) {
        coef[i] = (affine && i < p.cred) ? p.cs[i] : 1.f;
        coef[cld + i] = (affine && i < p.cred) ? p.ct[i] : 0.f;
    }

    // ---- staging slots (fixed per thread): raw buffer loads, 32-bit offsets, hardware range check (offset 2^31 -> zeros)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.wt_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned pgo[PQ], ugo[UQ];
    int plo[PQ], ulo[UQ];
    unsigned inimg = 0;      // bit q: patch slot q is an image pixel (else zero padding -- NOT act(shift))
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int slot = t + C3T_THREADS * q;
        const int pix = slot >> 1, chunk = slot & 1;
        const int prow = pix / C3T_PW, pcol = pix - prow * C3T_PW;
        const int gh = h0 - 1 + prow, gw = w0 - 1 + pcol;
        const bool ok = slot < PSLOTS && gh >= 0 && gh < p.h && gw >= 0 && gw < p.w;
        pgo[q] = ok ? (unsigned)(((((long long)img * p.h + gh) * p.w + gw) * p.ldi + 4 * chunk) * 4) : OOB;
        plo[q] = slot < PSLOTS ? pix * C3T_KC + 4 * (chunk ^ ((pix >> 3) & 1)) : -1;
        inimg |= (ok ? 1u : 0u) << q;
    }
#pragma unroll
    for (int q = 0; q < UQ; ++q) {
        const int slot = t + C3T_THREADS * q;
        const int chunk = slot & 1, rowi = slot >> 1;          // rowi = k * 64 + nn
        const int nn = rowi & (WINO_NT - 1), k = rowi / WINO_NT;
        const bool ok = n0 + nn < p.nout;
        ugo[q] = ok ? (unsigned)((((long long)k * p.nout + n0 + nn) * p.cred + 4 * chunk) * 4) : OOB;
        ulo[q] = rowi * C3T_KC + 4 * (chunk ^ ((nn >> 3) & 1));
    }
    const int pchunk = t & 1;

    float4 preg[PQ], ureg[UQ];
    auto issue_raw = [&](int s) {
        const int soff = s * C3T_KC * 4;
#pragma unroll
        for (int q = 0; q < PQ; ++q) preg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rin, pgo[q], soff, 0));
    };
    auto issue_u = [&](int s) {
        const int soff = s * C3T_KC * 4;
#pragma unroll
        for (int q = 0; q < UQ; ++q) ureg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rwt, ugo[q], soff, 0));
    };
    auto commit_raw = [&](int s, int boff) {
        float* buf = smem + boff;
        const int c0 = s * C3T_KC + 4 * pchunk;
        const float4 cs = ld4(coef + c0), ct = ld4(coef + cld + c0);
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (plo[q] >= 0) st4(buf + plo[q], ((inimg >> q) & 1u) ? view_affine4(preg[q], cs, ct, alo, ahi) : f4(0.f));
        }
    };
    auto commit_u = [&](int boff) {
        float* buf = smem + boff;
#pragma unroll
        for (int q = 0; q < UQ; ++q) st4(buf + ulo[q], ureg[q]);
    };

    // ---- input transform: thread = (row a of B^T: wave >> 1, 4-channel quad: wave & 1, tile: lane).
    //   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:  row a combines patch rows (i0, i1) = (0,2) (1,2) (2,1) (1,3), sign - + - -
    const int ta = wave >> 1, tq = wave & 1;
    const int ttr = lane >> 4, ttc = lane & 15;
    const int ti0 = ta == 0 ? 0 : (ta == 2 ? 2 : 1);
    const int ti1 = ta == 0 ? 2 : (ta == 1 ? 2 : (ta == 2 ? 1 : 3));
    const float tsgn = ta == 1 ? 1.f : -1.f;
    int tro[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int px0 = (2 * ttr + ti0) * C3T_PW + 2 * ttc + j, px1 = (2 * ttr + ti1) * C3T_PW + 2 * ttc + j;
        tro[j] = px0 * C3T_KC + 4 * (tq ^ ((px0 >> 3) & 1));
        tro[4 + j] = px1 * C3T_KC + 4 * (tq ^ ((px1 >> 3) & 1));
    }
    const int two = (4 * ta * WINO_TILES + lane) * C3T_KC + 4 * (tq ^ ((lane >> 3) & 1));       // + b * 64 * 8 per column position
    auto transform = [&](int roff, int voff) {
        const float* raw = smem + roff;
        float* v = smem + voff;
        float4 r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 d0 = ld4(raw + tro[j]), d1 = ld4(raw + tro[4 + j]);
            r[j] = make_float4(fmaf(tsgn, d1.x, d0.x), fmaf(tsgn, d1.y, d0.y), fmaf(tsgn, d1.z, d0.z), fmaf(tsgn, d1.w, d0.w));
        }
        st4(v + two + 0 * WINO_TILES * C3T_KC, make_float4(r[0].x - r[2].x, r[0].y - r[2].y, r[0].z - r[2].z, r[0].w - r[2].w));
        st4(v + two + 1 * WINO_TILES * C3T_KC, make_float4(r[1].x + r[2].x, r[1].y + r[2].y, r[1].z + r[2].z, r[1].w + r[2].w));
        st4(v + two + 2 * WINO_TILES * C3T_KC, make_float4(r[2].x - r[1].x, r[2].y - r[1].y, r[2].z - r[1].z, r[2].w - r[1].w));
        st4(v + two + 3 * WINO_TILES * C3T_KC, make_float4(r[1].x - r[3].x, r[1].y - r[3].y, r[1].z - r[3].z, r[1].w - r[3].w));
    };

    // ---- MFMA fragments: wave owns positions k = 2*wave + kk; rows = tiles (rb * 32 + li), columns = channels (cb * 32 + li)
    const int foff = li * C3T_KC + 4 * (hh ^ ((li >> 3) & 1));
    f32x16 acc[2][2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[kk][rb][cb][e] = 0.f;

    auto compute = [&](int voff, int uoff) {
        const float* v = smem + voff;
        const float* u = smem + uoff;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = 2 * wave + kk;
            const float4 a0 = ld4(v + (k * WINO_TILES + 0) * C3T_KC + foff), a1 = ld4(v + (k * WINO_TILES + 32) * C3T_KC + foff);
            const float4 b0 = ld4(u + (k * WINO_NT + 0) * C3T_KC + foff), b1 = ld4(u + (k * WINO_NT + 32) * C3T_KC + foff);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const float4 a = rb ? a1 : a0;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const float4 b = cb ? b1 : b0;
                    acc[kk][rb][cb] = mfma32(a.x, b.x, acc[kk][rb][cb]);
                    acc[kk][rb][cb] = mfma32(a.y, b.y, acc[kk][rb][cb]);
                    acc[kk][rb][cb] = mfma32(a.z, b.z, acc[kk][rb][cb]);
                    acc[kk][rb][cb] = mfma32(a.w, b.w, acc[kk][rb][cb]);
                }
            }
        }
    };

    // ---- pipeline.  Invariant at the top of iteration s: V[s&1] = B^T d B of step s, U[s&1] = weights of step s, raw[(s+1)&1] =
    // patch of step s+1, all visible.  Loads / commits / transforms past the last step run on harmless data (range-checked
    // loads, spare coefficient entries) into buffers nobody reads: no conditionals around memory operations in the loop.
    const int S = p.cred / C3T_KC;
    issue_raw(0);
    issue_u(0);
    __syncthreads();            // coef[] visible
    commit_raw(0, RAW0);
    commit_u(U0);
    issue_raw(1);
    __syncthreads();
    transform(RAW0, V0);
    commit_raw(1, RAW1);
    __syncthreads();
    int rcur = RAW0, rnxt = RAW1, vcur = V0, vnxt = V1, ucur = U0, unxt = U1;
    for (int s = 0; s < S; ++s) {
        issue_raw(s + 2);
        issue_u(s + 1);
        transform(rnxt, vnxt);          // step s+1
        compute(vcur, ucur);            // step s
        commit_raw(s + 2, rcur);        // last read by transform(s), one barrier ago
        commit_u(unxt);                 // last read by compute(s-1)
        __syncthreads();
        int x;
        x = rcur; rcur = rnxt; rnxt = x;
        x = vcur; vcur = vnxt; vnxt = x;
        x = ucur; ucur = unxt; unxt = x;
    }

    // ---- epilogue.  C/D layout of a 32x32 accumulator: column = lane & 31 (output channel), row = (e & 3) + 8 * (e >> 2) + 4 * hh (tile).
    //   A^T = [1 1 1 0; 0 1 -1 -1]:  y[i][j] = sum_ab A^T[i][a] m[a][b] A^T[j][b]
    float* ms = smem;                                             // [16][64 tiles][33]
    float* red = smem + 16 * WINO_TILES * WINO_MS_LD;             // [2 halves][2][16 groups][32]
    const int ecol = t & 31, eg = t >> 5;                         // thread = (channel of the half, tile group): tiles eg, eg+16, eg+32, eg+48
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int tile = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    ms[((2 * wave + kk) * WINO_TILES + tile) * WINO_MS_LD + li] = acc[kk][rb][cb][e];
                }
        __syncthreads();
        const int jl = cb * 32 + ecol, j = n0 + jl;
        const bool jok = j < p.nout;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tile = eg + 16 * r;
            const int tr = tile >> 4, tc = tile & 15;
            float m[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) m[k] = ms[(k * WINO_TILES + tile) * WINO_MS_LD + ecol];
            float c0[4], c1[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                c0[b] = m[0 + b] + m[4 + b] + m[8 + b];
                c1[b] = m[4 + b] - m[8 + b] - m[12 + b];
            }
            const float y00 = c0[0] + c0[1] + c0[2], y01 = c0[1] - c0[2] - c0[3];
            const float y10 = c1[0] + c1[1] + c1[2], y11 = c1[1] - c1[2] - c1[3];
            const int oh = h0 + 2 * tr, ow = w0 + 2 * tc;
            const float yv[4] = {y00, y01, y10, y11};
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int hh2 = oh + (o >> 1), ww2 = ow + (o & 1);
                if (jok && hh2 < p.h && ww2 < p.w) {
                    float* dst = p.out + (((long long)img * p.h + hh2) * p.w + ww2) * p.ldo + j;
                    float val = yv[o];
                    if (p.accumulate) val += *dst;
                    *dst = val;
                    ssum += yv[o];
                    ssq = fmaf(yv[o], yv[o], ssq);
                }
            }
        }
        if (p.stats != nullptr) {
            red[((cb * 2 + 0) * 16 + eg) * 32 + ecol] = ssum;
            red[((cb * 2 + 1) * 16 + eg) * 32 + ecol] = ssq;
        }
        __syncthreads();          // ms is rewritten by the next half; red complete after the second
    }
    if (p.stats != nullptr && t < 2 * WINO_NT) {
        const int which = t / WINO_NT, jl = t - which * WINO_NT;
        const int cb = jl >> 5, col = jl & 31, j = n0 + jl;
        if (j < p.nout) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) v += red[((cb * 2 + which) * 16 + g) * 32 + col];
            p.stats[((long long)mtile * 2 + which) * p.nout + j] = v;
        }
    }
}

// U = G w G^T per (input channel, output channel) pair, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]; written with the reduction
// channel contiguous.  mode 0 (forward): U[k][n][c] from w[i][j][c][n];  mode 1 (input gradient): U[k][c][n] from
// w[2-i][2-j][c][n].  One 32 x 32 (c, n) tile per block through LDS, so reads (n contiguous) and writes are both coalesced.
__global__ void __launch_bounds__(256) conv3_wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int cin, int cout, int mode) {
    __shared__ float tile[9][32][33];
    const int c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int tap = 0; tap < 9; ++tap)
        for (int r = ty; r < 32; r += 8) {
            const int c = c0 + r, n = n0 + tx;
            tile[tap][r][tx] = (c < cin && n < cout) ? w[((long long)tap * cin + c) * cout + n] : 0.f;
        }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        // forward: this thread writes (n = n0 + r, c = c0 + tx); gradient: (c = c0 + r, n = n0 + tx)
        const int cl = mode == 0 ? tx : r, nl = mode == 0 ? r : tx;
        const int c = c0 + cl, n = n0 + nl;
        if (c >= cin || n >= cout) continue;
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i][j] = mode == 0 ? tile[i * 3 + j][cl][nl] : tile[(2 - i) * 3 + (2 - j)][cl][nl];
        float gw[4][3];       // G w
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gw[0][j] = g[0][j];
            gw[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
            gw[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
            gw[3][j] = g[2][j];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float o[4] = {gw[a][0], 0.5f * (gw[a][0] + gw[a][1] + gw[a][2]), 0.5f * (gw[a][0] - gw[a][1] + gw[a][2]), gw[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = 4 * a + b;
                if (mode == 0) u[((long long)k * cout + n) * cin + c] = o[b];
                else u[((long long)k * cin + c) * cout + n] = o[b];
            }
        }
    }
}
