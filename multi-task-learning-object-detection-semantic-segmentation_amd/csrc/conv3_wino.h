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
// all 16 positions: 64 x 64 x 16 accumulators = 128 registers per lane.  Wave = (row a of the 4x4 position grid, half of the
// tiles); it owns positions 4a .. 4a+3 for its 32 tiles.  Per step of 8 input channels
//   the block stages  the (8+2) x (32+2) input halo (view applied while staging, zero padding after the view) and the 16 x 64 x 8
//                     slice of the transformed weights U = G w G^T (pre-computed per launch by conv3_wino_weights_kernel) in LDS;
//   each lane forms   row a of V = B^T d B for ITS tile and 4-channel quad from the two patch rows that row a of B^T combines:
//                     8 ds_read_b128, 32 additions -- and those four float4 ARE the A fragments (row = tile, 16-byte fragments,
//                     channel pairs (j, 4+j) as in conv3_tile.h) of the positions the wave owns: the transformed input exists
//                     neither in HBM nor in LDS, a wave multiplies what it transformed;
//   each wave runs    4 positions x 2 column blocks x 4 MFMAs against 8 ds_read_b128 of U.
// Software pipeline, ONE barrier per step, placed between the MFMAs of positions 4a+2 and 4a+3 (three weight buffers make that
// legal): in iteration s the global loads of patch s+2 / weights s+1 are in flight, patch s+1 (staged in iteration s-1) is
// transformed into next step's A registers in the shadow of the MFMAs of step s.
// Epilogue: each wave applies the column half of A^T . A in registers (it holds all four b of its row a), the row half sums over
// the four a through LDS once; the threads write the 2x2 outputs and -- forward -- the BatchNorm partial sums of the block (one
// partial row per pixel tile, fixed order, no atomics), exactly like the halo-tile kernel.
//
//   forward:   in = x (raw + view), U from w[i][j][c][n], reduction over c
//   backward:  in = dy,             U from w[2-i][2-j][c][n] (the mirrored, transposed filter), reduction over n
#pragma once

constexpr int WINO_NT = 64;                          // output channels per block
constexpr int WINO_TILES = 64;                       // 2x2 output tiles per block (4 rows x 16 columns of tiles)
constexpr int WINO_PIXP = 344;                       // patch pixels per 4-channel plane, padded: the two planes sit 8 sixteen-byte slots apart
constexpr int WINO_RAW_F = 2 * WINO_PIXP * 4;        // floats of one patch buffer  [quad][pixel][4]
constexpr int WINO_U_F = 16 * WINO_NT * C3T_KC;      // floats of one U buffer      [k][n][8]
constexpr int WINO_MS_LD = 33;                       // epilogue rows of 32 channels, padded (the two lane halves sit 4 tiles apart)
constexpr int WINO_MS_F = 4 * 2 * WINO_TILES * 2 * WINO_MS_LD;     // [a][j][tile][cb][33]
constexpr int WINO_RED_F = 2 * 2 * 8 * 32;           // [cb][sum | sumsq][tile group][32]
constexpr size_t wino_lds_floats(int cred) {
    const size_t loop = 2 * (size_t)WINO_RAW_F + 3 * (size_t)WINO_U_F + 2 * (size_t)(cred + 16), epi = (size_t)WINO_MS_F + WINO_RED_F;
    return loop > epi ? loop : epi;
}

template <int V> struct wino_const { static constexpr int value = V; };

// VIEW: the input carries a BatchNorm / activation view (else the staged values are stored as loaded: an identity view costs no
// VALU work -- every vector instruction in the loop takes issue slots from the MFMAs, measured ~4 cycles each)
template <bool VIEW>
__global__ void __launch_bounds__(C3T_THREADS, 2) conv3_wino_kernel(Conv3TArgs p) {
    constexpr int PSLOTS = C3T_PIX * 2;                // float4 slots of a patch (680)
    constexpr int PQ = (PSLOTS + C3T_THREADS - 1) / C3T_THREADS;
    constexpr int USLOTS = 16 * WINO_NT * 2;           // float4 slots of a weight slice (2048)
    constexpr int UQ = USLOTS / C3T_THREADS;          // 4 per thread
    extern __shared__ float smem[];
    // buffer offsets (floats from smem; integers, so that every access stays a DS instruction through the buffer swaps)
    constexpr int RAW0 = 0, RAW1 = WINO_RAW_F, U0 = 2 * WINO_RAW_F, U1 = U0 + WINO_U_F, U2 = U1 + WINO_U_F;
    float* coef = smem + U2 + WINO_U_F;                // [2][cred + 16]: scale, shift of the input view (+ spare steps)
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

    // patch pixel (row, col) of plane `quad` -> LDS float offset.  Columns 16..31 swap places pairwise: the sixteen lanes of a
    // transform read sit two pixels apart, so lanes c and c + 16 would share a 16-byte slot; with the swap one of them moves to
    // the neighbouring (odd / even) slot and every group of sixteen lanes covers sixteen distinct slots.
    auto raw_off = [](int prow, int pcol, int quad) { return (quad * WINO_PIXP + prow * C3T_PW + (pcol ^ ((pcol >> 4) & 1))) * 4; };

    // ---- staging slots (fixed per thread): raw buffer loads, 32-bit offsets, hardware range check (offset 2^31 -> zeros)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int npad = p.ntiles_n * WINO_NT;
    unsigned pgo[PQ];
    int plo[PQ];
    unsigned inimg = 0;      // bit q: patch slot q is an image pixel (else zero padding -- NOT act(shift))
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int slot = t + C3T_THREADS * q;
        const int pix = slot >> 1, chunk = slot & 1;
        const int prow = pix / C3T_PW, pcol = pix - prow * C3T_PW;
        const int gh = h0 - 1 + prow, gw = w0 - 1 + pcol;
        const bool ok = slot < PSLOTS && gh >= 0 && gh < p.h && gw >= 0 && gw < p.w;
        pgo[q] = ok ? (unsigned)(((((long long)img * p.in_hp + gh) * p.in_wp + gw) * p.ldi + 4 * chunk) * 4) : OOB;
        // a thread without a second slot writes to the four spare pixels at the end of the plane: the stores stay unconditional
        // (a branch around them would also skip the load's s_waitcnt, and the compiler would re-insert it -- counting the DMAs -- later)
        plo[q] = slot < PSLOTS ? raw_off(prow, pcol, chunk) : (chunk * WINO_PIXP + C3T_PIX + (t & 3)) * 4;
        inimg |= (ok ? 1u : 0u) << q;
    }
    // weights by LDS-DMA (buffer_load ... lds: no registers, no ds_write): the LDS image is lane-linear -- slot = t + 512 q lands
    // at 16 * slot bytes, row = slot >> 1 = k * 64 + nn, LDS half = slot & 1 -- so the swizzle goes on the SOURCE half.  A
    // thread's four slots sit 256 rows = 4 positions apart: one offset register.
    const int urow = t >> 1, unn = urow & (WINO_NT - 1);
    const unsigned ugo0 = (n0 + unn < p.nout) ? (unsigned)((((long long)(urow >> 6) * npad + n0 + unn) * C3T_KC + 4 * ((t & 1) ^ ((unn >> 3) & 1))) * 4) : OOB;
    const unsigned uqstep = (unsigned)(4 * npad * C3T_KC * 4);       // bytes between a thread's slots (4 positions)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int pchunk = t & 1;
    const int ustep = 16 * npad * C3T_KC * 4;          // bytes of one step of U[step][k][npad][8]

    float4 preg[PQ];
    auto issue_raw = [&](int s) {
        const int soff = s * C3T_KC * 4;
#pragma unroll
        for (int q = 0; q < PQ; ++q) preg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rin, pgo[q], soff, 0));
    };
    const pwt_i32x4 rwt = pwt_make_rsrc(p.wt, p.wt_bytes);
    auto issue_u = [&](int s, int boff) {      // weights of step s -> LDS buffer at float offset boff, asynchronously
        const int soff = s * ustep;
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((boff + (wave_u * 64 + C3T_THREADS * q) * 4) * 4);
            lds_dma16(rwt, dst, ugo0 + q * uqstep, soff);
        }
    };
    auto commit_raw = [&](int s, int boff) {
        float* buf = smem + boff;
        if (VIEW) {
            const int c0 = s * C3T_KC + 4 * pchunk;
            const float4 cs = ld4(coef + c0), ct = ld4(coef + cld + c0);
#pragma unroll
            for (int q = 0; q < PQ; ++q) {
                const float keep = ((inimg >> q) & 1u) ? 1.f : 0.f;      // zero padding AFTER the view, branch-free
                const float4 v = view_affine4(preg[q], cs, ct, alo, ahi);
                st4(buf + plo[q], make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep));
            }
        } else {
#pragma unroll
            for (int q = 0; q < PQ; ++q) st4(buf + plo[q], preg[q]);       // padding slots loaded zeros (range-checked offset)
        }
    };
    // ---- roles.  Wave = (row a of the 4x4 transformed tile: wave >> 1, half of the block's tiles: wave & 1); lane = (tile of that
    // half: lane & 31, 4-channel quad: lane >> 5).  The lane forms V[a][0..3] = (B^T d B)[a][.] of ITS tile and quad in registers
    //   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:  row a combines patch rows (i0, i1) = (0,2) (1,2) (2,1) (1,3), sign - + - -
    // and that float4 IS the A fragment of the 16-byte-fragment MFMA scheme for position k = 4a + b (row = tile, lanes 0-31 channels
    // 0-3, lanes 32-63 channels 4-7): the transformed input never touches LDS, the wave multiplies what it transformed.
    const int ta = wave >> 1, rbk = wave & 1;
    const int tile = rbk * 32 + li, ttr = tile >> 4, ttc = tile & 15;
    const int ti0 = ta == 0 ? 0 : (ta == 2 ? 2 : 1);
    const int ti1 = ta == 0 ? 2 : (ta == 1 ? 2 : (ta == 2 ? 1 : 3));
    const float tsgn = ta == 1 ? 1.f : -1.f;
    int tro[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tro[j] = raw_off(2 * ttr + ti0, 2 * ttc + j, hh);
        tro[4 + j] = raw_off(2 * ttr + ti1, 2 * ttc + j, hh);
    }
    auto transform_load = [&](int roff, float4* r) {
        const float* raw = smem + roff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 d0 = ld4(raw + tro[j]), d1 = ld4(raw + tro[4 + j]);
            r[j] = make_float4(fmaf(tsgn, d1.x, d0.x), fmaf(tsgn, d1.y, d0.y), fmaf(tsgn, d1.z, d0.z), fmaf(tsgn, d1.w, d0.w));
        }
    };
    auto transform_finish = [&](const float4* r, float4* v) {
        v[0] = make_float4(r[0].x - r[2].x, r[0].y - r[2].y, r[0].z - r[2].z, r[0].w - r[2].w);
        v[1] = make_float4(r[1].x + r[2].x, r[1].y + r[2].y, r[1].z + r[2].z, r[1].w + r[2].w);
        v[2] = make_float4(r[2].x - r[1].x, r[2].y - r[1].y, r[2].z - r[1].z, r[2].w - r[1].w);
        v[3] = make_float4(r[1].x - r[3].x, r[1].y - r[3].y, r[1].z - r[3].z, r[1].w - r[3].w);
        // the values are not used before the next step: keep the compiler from sinking the arithmetic down to the barrier
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(v[b].x), "+v"(v[b].y), "+v"(v[b].z), "+v"(v[b].w));
    };

    // ---- MFMAs: positions k = 4a + b, rows = the wave's 32 tiles, columns = output channels cb * 32 + li
    const int fbase = 4 * ta * WINO_NT * C3T_KC + li * C3T_KC + 4 * (hh ^ ((li >> 3) & 1));
    f32x16 acc[4][2];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][cb][e] = 0.f;

    auto bfrag = [&](int uoff, int b, float4* bf) {
        const float* u = smem + uoff + fbase;
        bf[0] = ld4(u + b * WINO_NT * C3T_KC);
        bf[1] = ld4(u + b * WINO_NT * C3T_KC + 32 * C3T_KC);
    };
    auto mfmas = [&](const float4 a, const float4* bf, int b) {
        acc[b][0] = mfma32(a.x, bf[0].x, acc[b][0]);
        acc[b][1] = mfma32(a.x, bf[1].x, acc[b][1]);
        acc[b][0] = mfma32(a.y, bf[0].y, acc[b][0]);
        acc[b][1] = mfma32(a.y, bf[1].y, acc[b][1]);
        acc[b][0] = mfma32(a.z, bf[0].z, acc[b][0]);
        acc[b][1] = mfma32(a.z, bf[1].z, acc[b][1]);
        acc[b][0] = mfma32(a.w, bf[0].w, acc[b][0]);
        acc[b][1] = mfma32(a.w, bf[1].w, acc[b][1]);
    };

    // ---- pipeline.  Top of iteration s: acur = transformed input of step s (registers), U[s % 3] = weights of step s, raw[(s+1)&1]
    // = patch of step s+1, all visible.  The ONE barrier of a step sits INSIDE its MFMAs: positions b = 0..2 before it (the global
    // loads issued at the top have three MFMA blocks to land), b = 3 (fragments already read) after it, so the matrix pipe has
    // queued work on both sides while the slowest wave commits.  That leaves readers of U[s % 3] (b = 3 of iteration s)
    // unordered against the commits of iteration s+1 -- which therefore go to a THIRD weight buffer, last read two barriers ago.  Loads / commits / transforms past the last step run
    // on harmless data (range-checked loads, spare coefficient entries) into buffers nobody reads: no conditionals around
    // memory operations in the loop.
    const int S = p.cred / C3T_KC;
    float4 tr[4], av[2][4], bf0[2], bf1[2];
    issue_raw(0);
    issue_u(0, U0);
    __syncthreads();            // coef[] visible
    commit_raw(0, RAW0);
    issue_raw(1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // the DMA of step 0 has landed (the two patch loads of step 1 may still fly)
    __syncthreads();
    transform_load(RAW0, tr);
    transform_finish(tr, av[0]);
    commit_raw(1, RAW1);
    __syncthreads();
    int ucur = U0, unxt = U1, ufar = U2;
    bfrag(ucur, 0, bf0);
    // one step; PH = s & 1 selects the patch buffers and the A register sets at compile time (the loop is unrolled by two)
    auto step = [&](auto PH, int s) {
        constexpr int ph = decltype(PH)::value;
        constexpr int rcur = ph ? RAW1 : RAW0, rnxt = ph ? RAW0 : RAW1;
        // every block of 8 MFMAs runs on fragments read one block earlier; bf0 of position 0 was read at the end of the last step
        issue_u(s + 1, unxt);           // U[(s+1) % 3]: last read by the last MFMA block of step s-2, two barriers ago
        issue_raw(s + 2);
        bfrag(ucur, 1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(av[ph][0], bf0, 0);
        __builtin_amdgcn_sched_barrier(0);
        transform_load(rnxt, tr);       // step s+1, between two MFMA blocks: the other wave of the SIMD has the matrix pipe meanwhile
        transform_finish(tr, av[ph ^ 1]);
        bfrag(ucur, 2, bf0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(av[ph][1], bf1, 1);
        __builtin_amdgcn_sched_barrier(0);
        bfrag(ucur, 3, bf1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(av[ph][2], bf0, 2);
        __builtin_amdgcn_sched_barrier(0);
        commit_raw(s + 2, rcur);        // last read by the transform of step s-1, one barrier ago (its wait covers the DMA issued first)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        bfrag(unxt, 0, bf0);            // position 0 of step s+1 (committed before the barrier)
        __builtin_amdgcn_sched_barrier(0);
        mfmas(av[ph][3], bf1, 3);
        __builtin_amdgcn_sched_barrier(0);
        const int x = ucur;
        ucur = unxt; unxt = ufar; ufar = x;
    };
    int s = 0;
    for (; s + 2 <= S; s += 2) {
        step(wino_const<0>{}, s);
        step(wino_const<1>{}, s + 1);
    }
    if (s < S) step(wino_const<0>{}, s);
    __syncthreads();            // every wave is done with the operand buffers: the epilogue reuses them

    // ---- epilogue.  C/D layout of a 32x32 accumulator: column = lane & 31 (output channel), row = (e & 3) + 8 * (e >> 2) + 4 * hh (tile).
    //   A^T = [1 1 1 0; 0 1 -1 -1]:  y[i][j] = sum_ab A^T[i][a] m[a][b] A^T[j][b].  The wave holds all four b of its row a: the
    //   column half of the transform happens in registers (c_j = sum_b A^T[j][b] m[a][b]), the sum over a goes through LDS once.
    float* ms = smem;                                             // [a][j][tile][cb][33]
    float* red = smem + WINO_MS_F;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tl = rbk * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            const float m0 = acc[0][cb][e], m1 = acc[1][cb][e], m2 = acc[2][cb][e], m3 = acc[3][cb][e];
            ms[(((ta * 2 + 0) * WINO_TILES + tl) * 2 + cb) * WINO_MS_LD + li] = m0 + m1 + m2;
            ms[(((ta * 2 + 1) * WINO_TILES + tl) * 2 + cb) * WINO_MS_LD + li] = m1 - m2 - m3;
        }
    __syncthreads();
    const int ecol = t & 31, ecb = (t >> 5) & 1, eg = t >> 6;     // thread = (channel: ecb * 32 + ecol, tile group: tiles eg, eg + 8, ...)
    const int jl = ecb * 32 + ecol, j = n0 + jl;
    const bool jok = j < p.nout;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int tl = eg + 8 * r;
        const int tr2 = tl >> 4, tc2 = tl & 15;
        float c[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) c[a][jj] = ms[(((a * 2 + jj) * WINO_TILES + tl) * 2 + ecb) * WINO_MS_LD + ecol];
        const float yv[4] = {c[0][0] + c[1][0] + c[2][0], c[0][1] + c[1][1] + c[2][1], c[1][0] - c[2][0] - c[3][0], c[1][1] - c[2][1] - c[3][1]};
        const int oh = h0 + 2 * tr2, ow = w0 + 2 * tc2;
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
        red[((ecb * 2 + 0) * 8 + eg) * 32 + ecol] = ssum;
        red[((ecb * 2 + 1) * 8 + eg) * 32 + ecol] = ssq;
        __syncthreads();
        if (t < 2 * WINO_NT) {
            const int which = t / WINO_NT, jl2 = t - which * WINO_NT;
            const int cb = jl2 >> 5, col = jl2 & 31, j2 = n0 + jl2;
            if (j2 < p.nout) {
                float v = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) v += red[((cb * 2 + which) * 8 + g) * 32 + col];
                p.stats[((long long)mtile * 2 + which) * p.nout + j2] = v;
            }
        }
    }
}

// U = G w G^T per (input channel, output channel) pair, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], stored in the order the conv kernel
// streams it: U[step = r / 8][k][o (padded to whole 64-channel blocks)][r % 8], r the reduction channel, o the output channel -- a
// block's slice of one step is 16 contiguous 2 KB pieces.  mode 0 (forward): r = c, o = n, from w[i][j][c][n];  mode 1 (input
// gradient): r = n, o = c, from w[2-i][2-j][c][n].  One 32 x 32 (c, n) tile per block through LDS.
__global__ void __launch_bounds__(256) conv3_wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int cin, int cout, int mode) {
    const int opad = (mode == 0 ? (cout + WINO_NT - 1) / WINO_NT : (cin + WINO_NT - 1) / WINO_NT) * WINO_NT;
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
                const int rr = mode == 0 ? c : n, oo = mode == 0 ? n : c;
                u[((((long long)(rr >> 3) * 16 + k) * opad + oo) << 3) + (rr & 7)] = o[b];
            }
        }
    }
}
