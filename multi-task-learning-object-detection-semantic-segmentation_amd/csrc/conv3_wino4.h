// Dense 3x3 stride-1 SAME convolution (DeepLabV3+ decoder, reference blocks.py:117) forward and input gradient in the
// WINOGRAD F(4x4, 3x3) form, fp32 throughout -- included by gemm.hip inside its anonymous namespace.
//
//     Y = A^T [ sum_c (G w_c G^T) .* (B^T d_c B) ] A        B^T 6x6, G 6x3, A^T 4x6 (Lavin & Gray's matrices, below)
// 4x4 output pixels per tile from a 6x6 input tile: the reduction over input channels becomes THIRTY-SIX independent GEMMs
// over a sixteenth of the rows -- 36/16 = 2.25 multiply-adds per output and reduction channel against 4 for F(2x2, 3x3)
// (conv3_wino.h) and 9 for the direct sum.  The price is in the constants (B^T up to 5, A^T up to 8): measured against float64
// at the decoder's 304 channels the result is 1.2e-5 of the output scale off (F(2x2): 8e-7, direct fp32: 4e-7;
// scripts/study/winograd_f4x4_error.py) -- inside the 1e-3 the parity statement allows, tests hold it to 5e-5.
//
// One block = FOUR waves = TR x TC tiles (<= 32: the MFMA's rows; 3 x 10 for a 120 x 160 image = 12 x 40 output pixels)
// x 32 output channels x all 36 positions; ONE block per CU, one wave per SIMD with the whole register file:
// wave = (ra, rb) owns the 3 x 3 quadrant a in 3ra.., b in 3rb.. of the 6 x 6 position grid for all 32 tiles -- 9 x 16 = 144
// accumulator registers per lane (64 channels would be 288: more than the 256 AGPRs the MFMAs can address -- tried, 1500 spills).
// Per step of 16 input channels
//   every thread     owns two STRIP tasks: the six vertically adjacent pixels (one column of one tile row) that a column of
//                    B^T d combines, 16 bytes (four channels) of them.  It loads them (6 buffer_load_b128, issued a step
//                    ahead; the four lanes of a pixel read 64 contiguous bytes), applies the input view, forms all six rows a
//                    of B^T d (48 multiply-adds) and stores them to LDS -- the ROW half of the input transform is done once
//                    per pixel column, not once per position;
//   each lane        (tile, 4-channel quad) reads the five columns its quadrant combines for row a (5 ds_read_b128), forms
//                    V[a][3rb..3rb+2] (28 multiply-adds) -- and those three float4 ARE the A fragments of three positions
//                    (row = tile, lanes 0-31 channels 0-3, lanes 32-63 channels 4-7 of an 8-channel half, as in conv3_wino.h);
//   the B fragments  (transformed weights U = G w G^T, laid out [8-channel step][position][channel][8] by
//                    conv3_wino4_weights_kernel) come STRAIGHT from global memory into registers, 16 bytes per lane, a wave's
//                    read of one position is 1 KB contiguous: a rolling ring of 18 float4, each reloaded for the next step
//                    right after the MFMAs that consumed it (a whole step of latency cover).  U never touches LDS;
//   each wave runs   3 rows x 2 halves x 3 positions x 4 = 72 v_mfma_f32_32x32x2_f32 against 30 + 12 LDS instructions.
// Two LDS buffers of row-transformed strips, ONE barrier per step (before the last of its six MFMA blocks).
// Work items are ordered in groups of TWO 32-channel tiles, pixel-tile-major inside a group, and dealt to the XCDs in contiguous
// runs: an XCD keeps two slices of U (2 x 1.4 MB at 304 reduction channels) in its L2 and the two channel tiles of a pixel tile are
// taken by two of its CUs back to back, so the second one's input strips are L2 hits (channel-tile-major, one slice per XCD: every
// XCD streamed the whole input -- 7.8 GB of HBM traffic per launch instead of 5.0, 4 % slower; profiles/r03_w4_group_ab.txt).
// Epilogue: the 36 x 32 x 32 accumulators go through LDS once; thread = (channel, 4 tiles) gathers the 36 values, applies
// A^T . A in registers and writes 4 x 4 pixels (+ forward: BatchNorm partial sums, one row per pixel tile, fixed order, no atomics).
//
//   forward:   in = x (raw + view), U from w[i][j][c][n], reduction over c
//   backward:  in = dy,             U from w[2-i][2-j][c][n] (the mirrored, transposed filter), reduction over n
#pragma once

constexpr int W4_THREADS = 256;
constexpr int W4_NT = 32;                            // output channels per block
constexpr int W4_RL = 12;                            // float4 slots of one (row a, column phase) line: pixel columns 4q + phase, q <= 11
constexpr int W4_AROW_F = 4 * W4_RL * 4;             // floats between rows a of a strip (192)
constexpr int W4_QP_MAX = 1040;                      // float4 slots of one 4-channel plane (>= TR * TRS, == 1 mod 16)
constexpr int W4_VB_F = 4 * W4_QP_MAX * 4;           // floats of one buffer of row-transformed strips (16 channels = 4 planes)
constexpr int W4_SPARE_F = 1024;                     // where threads without a strip store
constexpr int W4_MS_LD = 33;
constexpr int W4_MS_F = 36 * 8 * W4_MS_LD;           // epilogue staging, one round: [position][8 tiles][33]
constexpr int W4_RED_F = 2 * 8 * 32;                 // [sum | sumsq][tile group][32] (in the spare area)
constexpr size_t wino4_lds_floats(int cred) {
    static_assert(W4_MS_F <= W4_VB_F && W4_RED_F <= W4_SPARE_F, "the epilogue staging lives in buffer 1 / the spare area");
    return 2 * (size_t)W4_VB_F + W4_SPARE_F + 2 * (size_t)(cred + 16);
}

struct Wino4Args {
    const float* in;     // [n][in_hp][in_wp][ldi], entered at image pixel (0, 0)
    const float* cs;     // view of the input: act(cs*x + ct); nullptr = identity
    const float* ct;
    int act, ldi;
    const float* u;      // U[cred / 8][36][npad][8]
    float* out;          // [n][h][w][ldo]
    int ldo, accumulate;
    float* stats;        // forward: [mtiles][2][nout] partial (sum, sumsq); may be nullptr
    int n, h, w;
    int cred, nout, npad;
    int tr, tc;          // tiles per block (rows, columns), tr * tc <= 32, tc <= 11
    int trs;             // float4 slots between tile rows in LDS (>= 24 * W4_RL, == tc mod 16)
    int qps;             // float4 slots between 4-channel planes (>= tr * trs, == 1 mod 16, <= W4_QP_MAX)
    int tiles_h, tiles_w, ntiles_n;
    int group;           // channel tiles per group of the work order (1 or 2; divides ntiles_n)
    unsigned in_bytes, u_bytes, out_bytes;
    int in_hp, in_wp;
    unsigned long long* trace;   // measurement only (SSDSEG_W4_TRACE): per block and item, the clock at loop start / loop end / item end
};

// float4 as two PACKED pairs.  What scripts/micro/mfma_filler.hip measured on gfx950 (profiles/r03_mfma_f32_filler_cost.txt): beside
// v_mfma_f32_32x32x2_f32 NOTHING of the vector ALU hides -- a loop of chained MFMAs runs 64 cycles per MFMA, and every v_fma_f32 /
// v_add_u32 / v_mov_b32 placed between two of them adds ~4.4 cycles (one wave per SIMD; ~2.2 with two), plus ~10 cycles per MFMA ->
// VALU -> MFMA turn-around, whatever the instruction; a v_pk_fma_f32 costs the same slot as a v_fma_f32 and does twice the work.
// (LDS reads, scalar instructions and s_nop are nearly free; buffer loads ~8 cycles when at most one sits in a gap.)  The fp32 MFMA
// and the vector ALU behave as ONE pipe.  Hence: (1) the transforms are v_pk_*_f32, and in INLINE ASSEMBLY because hipcc splits
// packed f32 instructions back into single-lane ones wherever an MFMA is near (a peephole tuned for the 16-bit MFMAs, whose shadow
// does hide single-lane work); (2) the vector work of a block of 12 MFMAs sits in ONE cluster behind the block, not spread over
// its gaps (six turn-arounds per step instead of seventy-two).
typedef float w4v2 __attribute__((ext_vector_type(2)));
struct w4q { w4v2 lo, hi; };
__device__ __forceinline__ w4q w4_from(float4 v) { return w4q{w4v2{v.x, v.y}, w4v2{v.z, v.w}}; }
__device__ __forceinline__ float4 w4_to(w4q v) { return make_float4(v.lo[0], v.lo[1], v.hi[0], v.hi[1]); }
#define W4_PK3(name, text)                                                                                        \
    __device__ __forceinline__ w4q name(w4q a, w4q b) {                                                             \
        w4q d;                                                                                                      \
        asm(text : "=v"(d.lo) : "v"(a.lo), "v"(b.lo));                                                             \
        asm(text : "=v"(d.hi) : "v"(a.hi), "v"(b.hi));                                                             \
        return d;                                                                                                   \
    }
W4_PK3(w4_add, "v_pk_add_f32 %0, %1, %2")                                         // a + b
W4_PK3(w4_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")               // a - b
W4_PK3(w4_fma4, "v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]")                 // 4 a + b
W4_PK3(w4_fmam4, "v_pk_fma_f32 %0, %1, -4.0, %2 op_sel_hi:[1,0,1]")               // -4 a + b
W4_PK3(w4_fma2, "v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]")                 // 2 a + b
W4_PK3(w4_fmam2, "v_pk_fma_f32 %0, %1, -2.0, %2 op_sel_hi:[1,0,1]")               // -2 a + b
#undef W4_PK3
// one column of B^T d from the six values down it (B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0;
// 0 4 0 -5 0 1]): rows 0..2 from d0..d4, rows 3..5 from d1..d5; six packed-pair operations (12 instructions) each
__device__ __forceinline__ void w4_bt_lo(w4q d0, w4q d1, w4q d2, w4q d3, w4q d4, w4q* o) {
    const w4q t1 = w4_fmam4(d2, d4), t2 = w4_fmam4(d1, d3);       // d4 - 4 d2, d3 - 4 d1
    o[0] = w4_fma4(d0, w4_sub(t1, d2));                           // 4 d0 - 5 d2 + d4
    o[1] = w4_add(t1, t2);
    o[2] = w4_sub(t1, t2);
}
__device__ __forceinline__ void w4_bt_hi(w4q d1, w4q d2, w4q d3, w4q d4, w4q d5, w4q* o) {
    const w4q t3 = w4_sub(d4, d2), t4 = w4_sub(d3, d1);
    o[0] = w4_fma2(t4, t3);                                       // (d4 - d2) + 2 (d3 - d1)
    o[1] = w4_fmam2(t4, t3);
    o[2] = w4_fmam4(t4, w4_sub(d5, d3));                          // 4 d1 - 5 d3 + d5
}

template <bool VIEW>
__global__ void __launch_bounds__(W4_THREADS, 1) conv3_wino4_kernel(Wino4Args p) {
    extern __shared__ float smem[];
    float* coef = smem + 2 * W4_VB_F + W4_SPARE_F;     // [2][cred + 16]: scale, shift of the input view
    const int cld = p.cred + 16;

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;

    // PERSISTENT blocks, one per CU: a block walks its share of the work items (pixel tile x 32-channel tile), and the first loads
    // of the next item are in flight while the current one leaves through its epilogue.  Work order: see `setup` below; one
    // contiguous run per XCD (blocks go to the XCDs round-robin), the blocks of an XCD interleaved within its run.
    const int mtiles = p.n * p.tiles_h * p.tiles_w;
    const int total = mtiles * p.ntiles_n;
    const int nx = ((gridDim.x & 7u) == 0u && (total & 7) == 0) ? 8 : 1;
    const int per_x = total / nx, nbx = (int)gridDim.x / nx;
    const int item_base = ((int)blockIdx.x % nx) * per_x;
    int item_l = (int)blockIdx.x / nx;                 // index within the XCD's run; the block's items: item_l, item_l + nbx, ...
    const int ntl = p.tr * p.tc;                       // tiles in use (MFMA rows beyond them compute on tile 0's data, never written)
    const int PC = 4 * p.tc + 2;                       // patch columns
    const int M = p.cred / 16;                         // 16-channel steps

    const bool affine = p.cs != nullptr;
    const float alo = act_lo(p.act), ahi = act_hi(p.act);
    if (VIEW) {
        for (int i = t; i < cld; i += W4_THREADS) {
            coef[i] = (affine && i < p.cred) ? p.cs[i] : 1.f;
            coef[cld + i] = (affine && i < p.cred) ? p.ct[i] : 0.f;
        }
    }

    // ---- strip role: two tasks per thread and 16-channel step, task = (strip, 16-byte chunk c of the pixel's 64 bytes): the four
    // lanes of a pixel read one contiguous 64-byte piece -- a load instruction touches 16 cache lines, not 64 (measured: the
    // line look-ups of 16-byte pieces scattered over 64 pixels were what the loop waited for)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int sc = t & 3;
    int sstr[2], sspc[2];
    bool ssv[2];
    int swo[2][2];      // [task][buffer]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int strip = (t >> 2) + 64 * j;
        ssv[j] = strip < p.tr * PC;
        sstr[j] = ssv[j] ? strip / PC : 0;
        sspc[j] = ssv[j] ? strip - sstr[j] * PC : 0;
        // LDS slot of (quad, tile row, row a, pixel column x): quad * QPS + tr * TRS + (4a + (x & 3)) * RL + (x >> 2): the lanes of a
        // fragment read (consecutive tiles: columns 4 apart) and of a strip store (four quads of consecutive columns; QPS = 1 mod 16)
        // both spread over the banks
        const int v = (sc * p.qps + sstr[j] * p.trs + (sspc[j] & 3) * W4_RL + (sspc[j] >> 2)) * 4;
        swo[j][0] = ssv[j] ? v : 2 * W4_VB_F + (t & 3) * 4;
        swo[j][1] = ssv[j] ? v + W4_VB_F : 2 * W4_VB_F + (t & 3) * 4;
    }
    // per work item: where it is, and the global offsets of the thread's strips / weight fragments
    struct Item { int img, h0, w0, n0, mtile; };
    unsigned pgb[2];        // byte offset of the strip's top pixel (row h0 - 1 + 4 str; may lie above the image: wraps, never used then)
    unsigned inimg[2];      // bit r: row r of the strip is an image pixel (else zero padding)
    unsigned ubo;
    const int rowpitch = p.in_wp * p.ldi * 4;
    auto setup = [&](int il, Item& it) {
        // work order: channel tiles in GROUPS of p.group (1 or 2), pixel-tile-major inside a group -- items W and W + 1 are the two
        // channel tiles of one pixel tile, taken by two CUs of the same XCD at about the same time: the second one's input strips
        // come from that XCD's L2.  An XCD then holds p.group 1.4 MB slices of U instead of one and streams the input of
        // 1 / (8 / (ntiles_n / group)) of the pixel tiles: HBM reads of the input drop from ntiles_n x to ntiles_n / group x.
        const int W = item_base + il;
        const int gsz = p.group * mtiles;
        const int grp = W / gsz, rem = W - grp * gsz;
        it.mtile = rem / p.group;
        const int ntile = grp * p.group + (rem - it.mtile * p.group);
        const int tw = it.mtile % p.tiles_w;
        const int th = (it.mtile / p.tiles_w) % p.tiles_h;
        it.img = it.mtile / (p.tiles_w * p.tiles_h);
        it.h0 = th * 4 * p.tr; it.w0 = tw * 4 * p.tc;
        it.n0 = ntile * W4_NT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gh0 = it.h0 - 1 + 4 * sstr[j], gw = it.w0 - 1 + sspc[j];
            const bool cok = ssv[j] && gw >= 0 && gw < p.w;
            pgb[j] = (unsigned)(((((long long)it.img * p.in_hp + gh0) * p.in_wp + gw) * p.ldi + 4 * sc) * 4);
            inimg[j] = 0u;
#pragma unroll
            for (int r = 0; r < 6; ++r) inimg[j] |= ((cok && gh0 + r >= 0 && gh0 + r < p.h) ? 1u : 0u) << r;
        }
        ubo = it.n0 + li < p.nout ? (unsigned)(((it.n0 + li) * 8 + hh * 4) * 4) : OOB;
    };
    float4 sreg[2][6];
    auto issue_strip = [&](int m, int j) {
        const int soff = (m < M ? m : M - 1) * 64;
        // (one offset register per strip: the rows are formed on the way, a row outside the image takes the out-of-range offset.
        // NOT in the scalar offset: that one is added after the range check, and the top row of the top tiles wraps below zero.)
#pragma unroll
        for (int r = 0; r < 6; ++r)
            sreg[j][r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rin, ((inimg[j] >> r) & 1u) ? pgb[j] + (unsigned)(r * rowpitch) : OOB, soff, 0));
    };
    // the row half of the input transform down a strip: half 0 = rows 0..2 of B^T d, half 1 = rows 3..5 (dealt to two MFMA blocks);
    // computed into registers here, stored to LDS by the caller (one ds_write_b128 per MFMA gap)
    auto commit_compute = [&](int m, int j, int half, w4q* o) {
        w4q d[6];
        if (VIEW) {
            const int c0 = (m < M ? m : M - 1) * 16 + 4 * sc;
            const float4 cs = ld4(coef + c0), ct = ld4(coef + cld + c0);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const float keep = ((inimg[j] >> r) & 1u) ? 1.f : 0.f;      // zero padding AFTER the view
                const float4 v = view_affine4(sreg[j][r], cs, ct, alo, ahi);
                d[r] = w4_from(make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 6; ++r) d[r] = w4_from(sreg[j][r]);
        }
        if (half == 0) w4_bt_lo(d[0], d[1], d[2], d[3], d[4], o);
        else w4_bt_hi(d[1], d[2], d[3], d[4], d[5], o);
    };
    auto commit_store = [&](int boff, int half, int i, const w4q* o) { st4(smem + boff + (3 * half + i) * W4_AROW_F, w4_to(o[i])); };
    auto commit_strip = [&](int m, int j, int boff, int half) {      // (outside the step loop)
        w4q o[3];
        commit_compute(m, j, half, o);
#pragma unroll
        for (int i = 0; i < 3; ++i) commit_store(boff, half, i, o);
    };

    // ---- MFMA role: wave = quadrant (ra, rb) of the 6 x 6 position grid; lane = (tile, 4-channel quad of an 8-channel half step)
    const int ra = __builtin_amdgcn_readfirstlane(wave >> 1), rb = __builtin_amdgcn_readfirstlane(wave & 1);
    const int T = li < ntl ? li : 0;
    const int ttr = T / p.tc, ttc = T - ttr * p.tc;
    int rbase[2];       // [8-channel half kh]: quad 2 kh + hh
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) rbase[kh] = ((2 * kh + hh) * p.qps + ttr * p.trs + ttc) * 4 + 3 * ra * W4_AROW_F;
    const int upos = p.npad * 32;                      // bytes of one position of one 8-channel step
    const int ustep = 36 * upos;
    const int ubase = (ra * 18 + rb * 3) * upos;

    f32x16 acc[9];
    float4 bq[6][3];                                    // [block r = 2 ai + kh][bi]
    auto load_b = [&](int m, int r, int bi) {
        const int soff = (2 * (m < M ? m : M - 1) + (r & 1)) * ustep + ubase + ((r >> 1) * 6 + bi) * upos;
        bq[r][bi] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ru, ubo, soff, 0));
    };
    // the loop over the 16-channel steps of one work item, instantiated per column half RB of the quadrant (compile-time LDS offsets
    // and transform constants).  On entry: the strips of step 0 are in buffer 0 and visible, those of step 1 and the weights of
    // step 0 are in flight (in that order).
    auto run = [&](auto RBc) {
        constexpr int RB = decltype(RBc)::value;
        // the five pixel columns j = RB .. RB + 4 of the tile's six: float offsets of (phase j & 3, quad column j >> 2)
        auto read1 = [&](int boff, int r, int jj, w4q* e) {
            const float* src = smem + boff + rbase[r & 1] + (r >> 1) * W4_AROW_F;
            const int j = RB + jj;
            e[jj] = w4_from(ld4(src + ((j & 3) * W4_RL + (j >> 2)) * 4));
        };
        // rows b of B^T applied along the columns: V[a][3RB + 0..2]
        auto finish_row = [&](const w4q* e, w4q* v) {
            if constexpr (RB == 0) w4_bt_lo(e[0], e[1], e[2], e[3], e[4], v);      // e = d0..d4
            else w4_bt_hi(e[0], e[1], e[2], e[3], e[4], v);                        // e = d1..d5
        };

        w4q ev[5], av[2][3], cw[3];
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) read1(0, 0, jj, ev);
        finish_row(ev, av[0]);
        commit_compute(1, 0, 0, cw);          // what block 0 of the first step stores

        // One step = 16 input channels = six blocks of 12 MFMAs (the quadrant's three rows x two 8-channel halves).  The fp32 MFMA
        // shares its pipe with the vector ALU (see the note at w4q above), so a block is laid out by hand, gap by gap, every gap
        // fenced: memory instructions (cheap beside MFMAs) one or two per gap, ALL vector arithmetic in one cluster behind the last
        // MFMA.  Block r: gaps 0..2 store the strip rows the previous block's cluster computed (blocks 0..3: half a strip task each
        // of step m + 1, into buffer `nxt`, last read before the barrier of step m - 1); gaps 3..5 read the five columns of block
        // r + 1's row; the weight fragments of a column tile are re-loaded for the next step right after its four MFMAs (gaps 3, 7,
        // 11); blocks 1 and 3 re-issue their strip task's six loads for step m + 2 (its registers were consumed by the cluster of
        // the block before); the cluster turns the columns into block r + 1's fragments and transforms the strip half block r + 1
        // stores.  The ONE barrier of the step sits in gap 3 of block 5, before its reads -- the first of `nxt`.
        // PH = m & 1 selects the buffers at compile time (the loop is unrolled by two).
        auto step = [&](auto PH, int m) {
            constexpr int ph = decltype(PH)::value;
            constexpr int cur = ph ? W4_VB_F : 0, nxt = ph ? 0 : W4_VB_F;
            auto block = [&](auto RC) {
                constexpr int r = decltype(RC)::value;
                constexpr bool stores = r < 4, issues = r == 1 || r == 3;
                constexpr int tj = r >> 1;                           // the strip task of blocks 0..3
                auto mf = [&](int g) {
                    const int bi = g >> 2, k = (r >> 1) * 3 + bi;
                    const w4q a = av[r & 1][bi];
                    const float4 b = bq[r][bi];
                    const float af = (g & 3) == 0 ? a.lo[0] : (g & 3) == 1 ? a.lo[1] : (g & 3) == 2 ? a.hi[0] : a.hi[1];
                    const float bf = (g & 3) == 0 ? b.x : (g & 3) == 1 ? b.y : (g & 3) == 2 ? b.z : b.w;
                    acc[k] = mfma32(af, bf, acc[k]);
                };
                auto issue1 = [&](int rr) {
                    const int soff = (m + 2 < M ? m + 2 : M - 1) * 64;
                    sreg[tj][rr] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rin, ((inimg[tj] >> rr) & 1u) ? pgb[tj] + (unsigned)(rr * rowpitch) : OOB, soff, 0));
                };
                auto rd = [&](int jj) {
                    if constexpr (r < 5) read1(cur, r + 1, jj, ev);
                    else read1(nxt, 0, jj, ev);
                };
#define W4_FENCE __builtin_amdgcn_sched_barrier(0)
                W4_FENCE;
                mf(0); if constexpr (stores) commit_store(swo[tj][ph ^ 1], r & 1, 0, cw); W4_FENCE;
                mf(1); if constexpr (stores) commit_store(swo[tj][ph ^ 1], r & 1, 1, cw); if constexpr (issues) issue1(0); W4_FENCE;
                mf(2); if constexpr (stores) commit_store(swo[tj][ph ^ 1], r & 1, 2, cw); if constexpr (issues) issue1(1); W4_FENCE;
                mf(3); load_b(m + 1, r, 0);
                if constexpr (r == 5) __syncthreads();               // strips of step m + 1 visible, buffer `cur` free
                rd(0); W4_FENCE;
                mf(4); rd(1); rd(2); W4_FENCE;
                mf(5); rd(3); rd(4); W4_FENCE;
                mf(6); if constexpr (issues) issue1(2); W4_FENCE;
                mf(7); load_b(m + 1, r, 1); W4_FENCE;
                mf(8); if constexpr (issues) issue1(3); W4_FENCE;
                mf(9); if constexpr (issues) issue1(4); W4_FENCE;
                mf(10); if constexpr (issues) issue1(5); W4_FENCE;
                mf(11); load_b(m + 1, r, 2);
                finish_row(ev, av[(r + 1) & 1]);
                if constexpr (r < 3) commit_compute(m + 1, (r + 1) >> 1, (r + 1) & 1, cw);
                if constexpr (r == 5) commit_compute(m + 2, 0, 0, cw);
                W4_FENCE;
#undef W4_FENCE
            };
            block(wino_const<0>{}); block(wino_const<1>{}); block(wino_const<2>{});
            block(wino_const<3>{}); block(wino_const<4>{}); block(wino_const<5>{});
        };
        int m = 0;
        for (; m + 2 <= M; m += 2) {
            step(wino_const<0>{}, m);
            step(wino_const<1>{}, m + 1);
        }
        if (m < M) step(wino_const<0>{}, m);
    };

    // ---- epilogue pieces.  C/D layout of a 32x32 accumulator: column = lane & 31 (output channel), row = (e & 3) + 8 * (e >> 2) + 4 * hh
    // (tile).  Four rounds of eight tiles (e >> 2 = round) through a 38 KB staging area in buffer 1 -- buffer 0 is free for the next
    // item's first strips meanwhile.   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]:  y = A^T m A per (tile, channel)
    float* ms = smem + W4_VB_F;                                   // [position][8 tiles][33]
    float* red = smem + 2 * W4_VB_F;                              // (the spare area) [sum | sumsq][tile group][32]
    float ssum = 0.f, ssq = 0.f;
    auto epilogue_round = [&](auto QC, const Item& it) {
        constexpr int q = decltype(QC)::value;
        // (the thread's coordinates pass through an empty asm: everything derived from them -- LDS addresses, the tile's row and
        // column by integer division -- is recomputed here, per round, instead of being hoisted out of the item loop and held in
        // registers, or spilled, across the step loop: measured 130 spills, reloaded behind s_waitcnt vmcnt(0))
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int ecol = tt & 31, eg = tt >> 5;                   // thread = (channel ecol, tile eg of the round)
        const int wli = tt & 31, whh = (tt >> 5) & 1;
        {
            float* wdst = ms + (4 * whh) * W4_MS_LD + wli;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int pos = (3 * ra + k / 3) * 6 + 3 * rb + k % 3;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) wdst[(pos * 8 + e4) * W4_MS_LD] = acc[k][4 * q + e4];
            }
        }
        __syncthreads();
        const int j = it.n0 + ecol;
        const bool jok = j < p.nout;
        const int tl = 8 * q + eg;
        const int tr2 = tl / p.tc, tc2 = tl - tr2 * p.tc;
        float c[4][6];      // A^T m: rows i, columns b
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float mm[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) mm[a] = ms[((a * 6 + b) * 8 + eg) * W4_MS_LD + ecol];
            const float pp = mm[1] + mm[2], qq = mm[1] - mm[2], rr = mm[3] + mm[4], uu = mm[3] - mm[4];
            c[0][b] = mm[0] + pp + rr;
            c[1][b] = fmaf(2.f, uu, qq);
            c[2][b] = fmaf(4.f, rr, pp);
            c[3][b] = fmaf(8.f, uu, qq) + mm[5];
        }
        const int oh = it.h0 + 4 * tr2, ow = it.w0 + 4 * tc2;
        // stores: ONE 32-bit lane offset per round (the tile's first pixel, this thread's channel), the pixel of the 4 x 4 patch in the
        // scalar offset, pixels outside the image take the out-of-range offset -- no branch and no 64-bit address per store (sixteen
        // exec-masked regions of a dozen instructions each were a seventh of the epilogue; with one wave per SIMD every instruction
        // of the epilogue is exposed).  Accumulating launches (a gradient added to an existing one) keep the plain path.
        const bool tok = jok && tl < ntl;
        const unsigned obase = (unsigned)(((((long long)it.img * p.h + oh) * p.w + ow) * p.ldo + j) * 4);
        const int opix = p.ldo * 4, orow = p.w * opix;
        float yv[4][4];
        bool okp[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pp = c[i][1] + c[i][2], qq = c[i][1] - c[i][2], rr = c[i][3] + c[i][4], uu = c[i][3] - c[i][4];
            yv[i][0] = c[i][0] + pp + rr; yv[i][1] = fmaf(2.f, uu, qq); yv[i][2] = fmaf(4.f, rr, pp); yv[i][3] = fmaf(8.f, uu, qq) + c[i][5];
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) okp[i][jx] = tok && oh + i < p.h && ow + jx < p.w;
        }
        if (p.accumulate) {      // (1: add to what is there; 2: plain store through a 64-bit address -- outputs beyond 2 GiB)
            for (int i = 0; i < 4; ++i)
                for (int jx = 0; jx < 4; ++jx)
                    if (okp[i][jx]) {
                        float* dst = p.out + (((long long)it.img * p.h + oh + i) * p.w + ow + jx) * p.ldo + j;
                        *dst = p.accumulate == 1 ? *dst + yv[i][jx] : yv[i][jx];
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jx = 0; jx < 4; ++jx)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv[i][jx]), rout, okp[i][jx] ? obase : OOB, i * orow + jx * opix, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const float ys = okp[i][jx] ? yv[i][jx] : 0.f;
                ssum += ys;
                ssq = fmaf(ys, ys, ssq);
            }
        __syncthreads();            // the staging area is free for the next round
    };

    // the loads in flight when the step loop is entered, in the ORDER the loop keeps them in (strips of step 1, task 0 then task 1, then
    // the weight fragments block by block): the compiler's s_waitcnt counts are the minimum over all ways into the loop, and it may
    // reorder independent loads -- hence the fences
    auto first_strips = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        issue_strip(1, 0);
        __builtin_amdgcn_sched_barrier(0);
        issue_strip(1, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto first_weights = [&]() {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bi = 0; bi < 3; ++bi) load_b(0, r, bi);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- the block's items
    const int n_items = item_l < per_x ? (per_x - item_l + nbx - 1) / nbx : 0;
    if (n_items == 0) return;
    Item cur_it;
    setup(item_l, cur_it);
    issue_strip(0, 0);
    issue_strip(0, 1);
    __syncthreads();                // coef[] visible
#pragma unroll
    for (int j = 0; j < 2; ++j) { commit_strip(0, j, swo[j][0], 0); commit_strip(0, j, swo[j][0], 1); }
    first_strips();
    first_weights();
    // (the column half RB of the wave's quadrant is a compile-time constant of everything below: ONE branch per kernel.  Inside the
    // item loop it made the register allocator store and reload the 120 registers of loads in flight around it, per item, behind
    // s_waitcnt vmcnt(0).)
    auto items = [&](auto RBc) {
    for (int k = 0; k < n_items; ++k) {
#pragma unroll
        for (int kk = 0; kk < 9; ++kk)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[kk][e] = 0.f;
        __syncthreads();            // the strips of step 0 (buffer 0) are visible
        if (p.trace != nullptr && t == 0) p.trace[((long long)blockIdx.x * 64 + (k & 63)) * 4 + 0] = clock64();
        run(RBc);
        if (p.trace != nullptr && t == 0) p.trace[((long long)blockIdx.x * 64 + (k & 63)) * 4 + 1] = clock64();
        __syncthreads();            // every wave is done with the strip buffers
        const Item done = cur_it;
        const bool more = k + 1 < n_items;
        if (more) {                 // the next item's first loads fly while this one leaves
            item_l += nbx;
            setup(item_l, cur_it);
            issue_strip(0, 0);
            issue_strip(0, 1);
        }
        ssum = 0.f; ssq = 0.f;
        epilogue_round(wino_const<0>{}, done);
        epilogue_round(wino_const<1>{}, done);
        if (more) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { commit_strip(0, j, swo[j][0], 0); commit_strip(0, j, swo[j][0], 1); }
            first_strips();
        }
        epilogue_round(wino_const<2>{}, done);
        epilogue_round(wino_const<3>{}, done);
        if (more) first_weights();      // (L2 hits; 72 registers the epilogue needs for itself)
        if (p.stats != nullptr) {
            int tt = t;
            asm volatile("" : "+v"(tt));
            red[(0 * 8 + (tt >> 5)) * 32 + (tt & 31)] = ssum;
            red[(1 * 8 + (tt >> 5)) * 32 + (tt & 31)] = ssq;
            __syncthreads();
            if (tt < 2 * W4_NT) {
                const int which = tt / W4_NT, col = tt - which * W4_NT, j2 = done.n0 + col;
                if (j2 < p.nout) {
                    float v = 0.f;
#pragma unroll
                    for (int g = 0; g < 8; ++g) v += red[(which * 8 + g) * 32 + col];
                    p.stats[((long long)done.mtile * 2 + which) * p.nout + j2] = v;
                }
            }
        }
        if (p.trace != nullptr && t == 0) p.trace[((long long)blockIdx.x * 64 + (k & 63)) * 4 + 2] = clock64();
    }
    };
    if (rb == 0) items(wino_const<0>{});
    else items(wino_const<1>{});
}

// U = G w G^T per (input channel, output channel) pair, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1],
// stored in the order the conv kernel streams it: U[step = r / 8][position 6a + b][o (padded to whole 64-channel blocks)][r % 8], r the
// reduction channel, o the output channel.  mode 0 (forward): r = c, o = n, from w[i][j][c][n];  mode 1 (input gradient): r = n, o = c,
// from w[2-i][2-j][c][n].  One 32 x 32 (c, n) tile per block through LDS.
__global__ void __launch_bounds__(256) conv3_wino4_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int cin, int cout, int mode) {
    const int opad = (mode == 0 ? (cout + W4_NT - 1) / W4_NT : (cin + W4_NT - 1) / W4_NT) * W4_NT;
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
        float gw[6][3];       // G w
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float s02 = g[0][j] + g[2][j];
            gw[0][j] = 0.25f * g[0][j];
            gw[1][j] = (-1.f / 6.f) * (s02 + g[1][j]);
            gw[2][j] = (-1.f / 6.f) * (s02 - g[1][j]);
            gw[3][j] = (1.f / 24.f) * g[0][j] + (1.f / 12.f) * g[1][j] + (1.f / 6.f) * g[2][j];
            gw[4][j] = (1.f / 24.f) * g[0][j] - (1.f / 12.f) * g[1][j] + (1.f / 6.f) * g[2][j];
            gw[5][j] = g[2][j];
        }
        const int rr = mode == 0 ? c : n, oo = mode == 0 ? n : c;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const float s02 = gw[a][0] + gw[a][2];
            const float o[6] = {0.25f * gw[a][0],
                                (-1.f / 6.f) * (s02 + gw[a][1]),
                                (-1.f / 6.f) * (s02 - gw[a][1]),
                                (1.f / 24.f) * gw[a][0] + (1.f / 12.f) * gw[a][1] + (1.f / 6.f) * gw[a][2],
                                (1.f / 24.f) * gw[a][0] - (1.f / 12.f) * gw[a][1] + (1.f / 6.f) * gw[a][2],
                                gw[a][2]};
#pragma unroll
            for (int b = 0; b < 6; ++b) u[((((long long)(rr >> 3) * 36 + a * 6 + b) * opad + oo) << 3) + (rr & 7)] = o[b];
        }
    }
}
