// Weight gradient of the dense 3x3 stride-1 SAME convolution (DeepLabV3+ decoder, reference blocks.py:117) in the WINOGRAD
// F(2x2, 3x3) form, fp32 throughout -- included by gemm.hip inside its anonymous namespace, after conv3_wino.h.
//
// With Y = A^T [ sum_c U_c .* V_c ] A per 2x2 output tile (U = G w G^T, V = B^T d B, conv3_wino.h), the gradient of the loss
// with respect to U is, per position k of the 4x4 grid, a plain GEMM over the TILES:
//     dU[k][c][n] = sum_tiles V[k][tile][c] * Z[k][tile][n],      Z = A dY A^T  (4x4 from the 2x2 tile of the output gradient)
//     dw          = G^T dU G
// sixteen (cin x tiles) x (tiles x cout) products instead of nine (cin x pixels) x (pixels x cout): 2.25x fewer MFMAs than the
// halo-tile kernel (conv3_wgrad_tile.h), which already runs at 0.78 of the fp32 MFMA peak.
//
// The reduction index of those GEMMs (the tile) is the STRIDED index of the NHWC tensors, so both operands would have to be
// transposed on the way to the MFMA; instead the MFMA's freedom to NAME its rows is used.  A lane is (pair r of channels,
// tile parity kk); it reads eight-byte pieces [pixel][2r, 2r+1] of the staged patches, forms V (for its two input channels)
// and Z (for its two output channels) of ITS tile in registers, and issues, per position,
//     acc[j][j'] += mfma_32x32x2( V.comp[j], Z.comp[j'] ),     j, j' in {0, 1}
// where MFMA row r MEANS channel 2r + j and column r' MEANS output channel 2r' + j'; the two k-slots of the instruction are the
// two tiles of the pair.  No operand ever goes through LDS in transformed form, no transposition happens anywhere: 4 MFMAs per
// (V, Z) register pair.  One block = eight waves = all 16 positions of a 64 x 64 (cin, cout) patch = 128 accumulator registers
// per lane; wave = (row a of the position grid, half bh of its columns): positions 4a + 2bh, 4a + 2bh + 1.  The minus signs of
// A's last row / column are left out of Z and applied by the finalize kernel (positions with a == 3 or b == 3 flip sign).
//
// Staging: per step a strip of 16 tiles (2 output rows x 32 columns): the 4 x 34 pixel halo of the input (64 channels) and the
// 2 x 32 pixels of dY (64 channels), both by LDS-DMA in their natural [pixel][channel] layout, double-buffered, one barrier
// per step (64 MFMAs per wave).  That needs plain, in-range, already-activated input: the caller first materialises
// act(scale * x + shift) into a ZERO-BORDERED copy [n][h+2][w+2][cin] (conv3_pad_view_kernel, ~0.35 ms for the decoder conv) --
// applying the BatchNorm view inside the loop would cost more vector instructions than there are MFMAs.
// The reduction over tiles is split across blocks (partials per split, summed in a fixed order by the finalize kernel).
// Shapes: h even, w even (a row's last strip may be partial: its surplus dY columns are loaded as zeros); else the caller keeps the
// halo-tile kernel.
#pragma once

constexpr int WWG_KT = 64, WWG_NT = 64;               // channels of the patch: input (rows), output (columns)
constexpr int WWG_TS = 16;                            // tiles per step (32 output columns x 2 rows)
constexpr int WWG_XPIX = 4 * 34, WWG_YPIX = 2 * 32;   // staged pixels per step
constexpr int WWG_X_F = WWG_XPIX * WWG_KT;            // floats of the input patch   [4][34][64]
constexpr int WWG_Y_F = WWG_YPIX * WWG_NT;            // floats of the dY patch      [2][32][64]
constexpr int WWG_BUF_F = WWG_X_F + WWG_Y_F;
constexpr size_t WWG_LDS_BYTES = 2 * (size_t)WWG_BUF_F * sizeof(float);   // 102,400
constexpr int WWG_THREADS = 512;

struct WinoWgArgs {
    const float* xp;     // [n][h+2][w+2][cin]  activated input with a zero border
    const float* dy;     // [n][h][w][cout]
    float* part;         // [patch][slots][16][64][64]: partial dU per (patch, range of steps)
    int n, h, w, cin, cout;
    int cpatches, npatches;
    int strips;          // ceil(w / 32)
    int wrem;            // output columns of a row's LAST strip (32 when w is a multiple of 32): the dY loads of the others read zeros
    int steps;           // n * (h / 2) * strips
    // Even split of patches x steps over the CUs (one 100 KB block per CU): blocks 0 .. full * patches - 1 take `span` steps of one
    // patch each (chunk-major: the blocks of a chunk read the same pixels); the remaining `tail` = steps - full * span steps of all
    // patches form one patch-major sequence of patches * tail units cut into pieces of `span` units -- a tail block works on up to
    // three patches in turn.  The decoder conv: 20 patches x 9600 steps on 256 CUs = 750 units per block (12 full chunks = 240
    // blocks + 16 tail blocks) instead of 240 blocks of 800 steps and 16 idle CUs.
    int span, full, tail, slots;
    unsigned xp_bytes, dy_bytes, part_bytes;
};

// slot of the partial a tail block `tb` writes for patch `patch` (the finalize kernel walks the same numbering)
__device__ __forceinline__ int wwg_tail_first_block(int patch, int tail, int span) { return (int)(((long long)patch * tail) / span); }

typedef float wwg_f2 __attribute__((ext_vector_type(2)));
typedef unsigned wwg_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wwg_f2 wwg_ld2(const float* p) { return *reinterpret_cast<const wwg_f2*>(p); }

__global__ void __launch_bounds__(WWG_THREADS, 2) conv3_wino_wgrad_kernel(WinoWgArgs p) {
    extern __shared__ float smem[];
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, r = lane & 31, kk = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int a = wave_u >> 1, bh = wave_u & 1;
    const unsigned L = xcd_logical_id(blockIdx.x, gridDim.x);
    const int patches = p.cpatches * p.npatches;
    const int nfull = p.full * patches;
    const bool is_tail = (int)L >= nfull;
    // units of this block: a full block owns steps [chunk * span, + span) of ONE patch; a tail block owns units [u, uend) of the
    // patch-major tail sequence (unit = patch * tail + step - full * span)
    int u = is_tail ? ((int)L - nfull) * p.span : 0;
    const int uend = is_tail ? (u + p.span < patches * p.tail ? u + p.span : patches * p.tail) : p.span;
    const pwt_i32x4 rx = pwt_make_rsrc(p.xp, p.xp_bytes);
    const pwt_i32x4 ry = pwt_make_rsrc(p.dy, p.dy_bytes);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, p.part_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const int wp = p.w + 2;

  // one pass = one (patch, step range): accumulators are zeroed at its top and stored at its bottom, nothing but `u` lives across
  while (u < uend) {
    const int patch = is_tail ? u / p.tail : (int)(L % (unsigned)patches);
    const int s0 = is_tail ? p.full * p.span + (u - patch * p.tail) : (int)(L / (unsigned)patches) * p.span;
    const int seg = is_tail ? (uend - u < p.tail - (u - patch * p.tail) ? uend - u : p.tail - (u - patch * p.tail)) : p.span;
    const int s1 = s0 + seg;
    const int slot = is_tail ? p.full + ((int)L - nfull) - wwg_tail_first_block(patch, p.tail, p.span) : (int)(L / (unsigned)patches);
    u += seg;
    const int c0 = (patch / p.npatches) * WWG_KT, n0 = (patch % p.npatches) * WWG_NT;

    f32x16 acc[2][2][2];      // [position of the half][j: input channel of the pair][j': output channel of the pair]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][j][jj][e] = 0.f;

    // ---- DMA slots.  Input patch: 136 pixels x 16 quads = 34 wave-instructions (wave w issues w, w+8, ...); dY: 64 x 16 = 16.
    unsigned xgo[5], ygo[2], ygl[2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int pix = 4 * (wave + 8 * i) + (lane >> 4), quad = lane & 15;
        const int prow = pix / 34, pcol = pix - prow * 34;
        const bool ok = pix < WWG_XPIX && c0 + 4 * quad < p.cin;
        xgo[i] = ok ? (unsigned)(((prow * wp + pcol) * p.cin + c0 + 4 * quad) * 4) : OOB;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pix = 4 * (wave + 8 * i) + (lane >> 4), quad = lane & 15;
        const int prow = pix >> 5, pcol = pix & 31;
        const bool ok = n0 + 4 * quad < p.cout;
        ygo[i] = ok ? (unsigned)(((prow * p.w + pcol) * p.cout + n0 + 4 * quad) * 4) : OOB;
        ygl[i] = pcol < p.wrem ? ygo[i] : OOB;       // last strip of a row: columns beyond the image contribute Z = 0 (whatever V holds there)
    }
    // step -> byte offsets of its strip in the two tensors, advanced incrementally (all wave-uniform)
    const int hrows = p.h >> 1, nstrips = p.strips;      // (locals: a lambda that touches `p` itself can push the whole argument struct into scratch)
    int strip = s0 % p.strips, trow = (s0 / p.strips) % hrows;
    int xs = (int)((((long long)(s0 / (p.strips * hrows)) * (p.h + 2) + 2 * trow) * wp + 32 * strip) * p.cin * 4);
    int ys = (int)((((long long)(s0 / (p.strips * hrows)) * p.h + 2 * trow) * p.w + 32 * strip) * p.cout * 4);
    const int xs_strip = 32 * p.cin * 4, ys_strip = 32 * p.cout * 4;
    const int xs_row = (2 * wp - 32 * (p.strips - 1)) * p.cin * 4, ys_row = (2 * p.w - 32 * (p.strips - 1)) * p.cout * 4;     // last strip -> first strip, two rows down
    const int xs_img = xs_row + 2 * wp * p.cin * 4;       // ... and over the two border rows into the next image (dY has none)
    // the step the offsets point at -> buffer BUF, asynchronously; then advance (selects on values, no branches).  A macro, not a
    // lambda: with three call sites the closure of a by-reference lambda stays in scratch memory together with everything it
    // captures (the inline asm's memory clobber keeps the compiler from promoting it back to registers)
#define WWG_ISSUE(BUF)                                                                                                  \
    do {                                                                                                                \
        const int bb_ = __builtin_amdgcn_readfirstlane(BUF) * WWG_BUF_F;                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 5; ++i_) {                                                              \
            if (i_ < 4 || wave_u < 2) lds_dma16(rx, (unsigned)((bb_ + 64 * (wave_u + 8 * i_) * 4) * 4), xgo[i_], xs);   \
        }                                                                                                               \
        const bool wrap_ = strip + 1 == nstrips, wrap2_ = wrap_ && trow + 1 == hrows;                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                \
            lds_dma16(ry, (unsigned)((bb_ + WWG_X_F + 64 * (wave_u + 8 * i_) * 4) * 4), wrap_ ? ygl[i_] : ygo[i_], ys); \
        xs += wrap_ ? (wrap2_ ? xs_img : xs_row) : xs_strip;                                                            \
        ys += wrap_ ? ys_row : ys_strip;                                                                                \
        strip = wrap_ ? 0 : strip + 1;                                                                                  \
        trow = wrap_ ? (wrap2_ ? 0 : trow + 1) : trow;                                                                  \
    } while (0)

    // ---- roles.  B^T rows:  a -> patch rows (i0, i1), sign:  (0,2,-) (1,2,+) (2,1,-) (1,3,-)            [conv3_wino.h]
    //              A rows (sign of the last one folded into the finalize):  a -> dY rows (rA, rB), beta:  (0,0,0) (0,1,+1) (0,1,-1) (1,1,0)
    // Column combinations, written so that both halves run the same instructions:
    //   u = row combination at patch columns (o0, o1, o2) = bh ? (2, 1, 3) : (0, 2, 1)
    //   first position  (b = 0 | 2):  V = u0 - u1            (r0 - r2 | r2 - r1)         Z = e0 + phi * e1,  phi = bh ? -1 : 0
    //   second position (b = 1 | 3):  V = u1 + tau * u2      (r2 + r1 | r1 - r3)         Z = psi * e0 + e1,  psi = bh ? 0 : 1   (b = 3: sign in the finalize)
    const int i0 = a == 0 ? 0 : (a == 2 ? 2 : 1);
    const int i1 = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
    const float xsgn = a == 1 ? 1.f : -1.f;
    const int rA = a == 3 ? 1 : 0, rB = (a == 1 || a == 2) ? 1 : rA;
    const float beta = a == 1 ? 1.f : (a == 2 ? -1.f : 0.f);
    const int o0 = bh ? 2 : 0, o1 = bh ? 1 : 2, o2 = bh ? 3 : 1;
    const float tau = bh ? -1.f : 1.f, phi = bh ? -1.f : 0.f, psi = bh ? 0.f : 1.f;
    // lane bases (floats from the buffer start): tile t = 2 * pair + kk -> patch column 2t = 4 * pair + 2kk
    int xb[2][3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int o = m == 0 ? o0 : (m == 1 ? o1 : o2);
        xb[0][m] = (i0 * 34 + 2 * kk + o) * WWG_KT + 2 * r;
        xb[1][m] = (i1 * 34 + 2 * kk + o) * WWG_KT + 2 * r;
    }
    const int yA = WWG_X_F + (rA * 32 + 2 * kk) * WWG_NT + 2 * r, yB = WWG_X_F + (rB * 32 + 2 * kk) * WWG_NT + 2 * r;

    auto form = [&](const float* buf, int pair, wwg_f2& v0, wwg_f2& v1, wwg_f2& z0, wwg_f2& z1) __attribute__((always_inline)) {
        const int po = pair * 4 * WWG_KT;      // WWG_KT == WWG_NT
        wwg_f2 u[3], e[2];
#pragma unroll
        for (int m = 0; m < 3; ++m) u[m] = wwg_ld2(buf + xb[0][m] + po) + xsgn * wwg_ld2(buf + xb[1][m] + po);
#pragma unroll
        for (int c = 0; c < 2; ++c) e[c] = wwg_ld2(buf + yA + po + c * WWG_NT) + beta * wwg_ld2(buf + yB + po + c * WWG_NT);
        v0 = u[0] - u[1]; v1 = u[1] + tau * u[2];
        z0 = e[0] + phi * e[1]; z1 = psi * e[0] + e[1];
    };
    auto compute = [&](int boff) __attribute__((always_inline)) {
        const float* buf = smem + boff;
        wwg_f2 v0[2], v1[2], z0[2], z1[2];
        form(buf, 0, v0[0], v1[0], z0[0], z1[0]);
#pragma unroll
        for (int pair = 0; pair < WWG_TS / 2; ++pair) {
            const int c = pair & 1, nx = c ^ 1;
            // the next pair's reads and additions go out ahead of this pair's MFMAs; the fences keep the scheduler from hoisting
            // ALL the reads of the step to the top (160 registers of them)
            if (pair + 1 < WWG_TS / 2) form(buf, pair + 1, v0[nx], v1[nx], z0[nx], z1[nx]);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0][0] = mfma32(v0[c].x, z0[c].x, acc[0][0][0]);
            acc[0][0][1] = mfma32(v0[c].x, z0[c].y, acc[0][0][1]);
            acc[0][1][0] = mfma32(v0[c].y, z0[c].x, acc[0][1][0]);
            acc[0][1][1] = mfma32(v0[c].y, z0[c].y, acc[0][1][1]);
            acc[1][0][0] = mfma32(v1[c].x, z1[c].x, acc[1][0][0]);
            acc[1][0][1] = mfma32(v1[c].x, z1[c].y, acc[1][0][1]);
            acc[1][1][0] = mfma32(v1[c].y, z1[c].x, acc[1][1][0]);
            acc[1][1][1] = mfma32(v1[c].y, z1[c].y, acc[1][1][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- pipeline: buffer (s & 1) holds step s; its DMA was issued one step earlier.  ONE loop body, one exit (two bodies or an
    // early exit make the compiler carry a second copy of the 128 accumulator registers across the join); the host never
    // launches an empty split.  DMAs past the last step are issued anyway (range-checked: they read zeros or live data into a
    // buffer nobody reads) so that no branch surrounds them.
    const int nsteps = s1 - s0;
    WWG_ISSUE(0);
    WWG_ISSUE(1);
    // a wave issues 6 or 7 DMA instructions per step: at most 6 outstanding == all of step 0 (and, for the 7-instruction waves, one of step 1) landed
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < nsteps; ++s) {
        compute(cur * WWG_BUF_F);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // step s+1 has landed
        __syncthreads();                                         // ... for everybody, and everybody is done reading buffer `cur`
        WWG_ISSUE(cur);                                          // step s+2
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing in flight into LDS when the pass ends (a wave only ever
                                                               // writes its own LDS slots, and everybody is past the last barrier)
#undef WWG_ISSUE

    // ---- partial dU of this pass.  C/D layout: column = lane & 31 (pair r' of output channels), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    // (pair r of input channels): patch element (2 row + j, 2 col + j'); the two j' of a lane are adjacent in memory.  Compact
    // [64][64] patches (rows beyond cin / columns beyond cout hold products of zeros and are never read).
    // Buffer stores: one 32-bit lane offset, the (e, j) part of the address in the scalar offset (64-bit pointers for 64 stores
    // would not fit beside the accumulators).
    const int col = lane & 31, hh = lane >> 5;
    const unsigned lane_off = (unsigned)(((8 * hh) * WWG_NT + 2 * col) * 4);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int k = 4 * a + 2 * bh + b;
        const int kbase = (int)(((((long long)patch * p.slots + slot) * 16 + k) * WWG_KT) * WWG_NT * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dr = 2 * ((e & 3) + 8 * (e >> 2)) + j;
                const float lo = acc[b][j][0][e], hi2 = acc[b][j][1][e];
                wwg_f2 v;
                v.x = lo; v.y = hi2;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wwg_u2, v), rp, lane_off, kbase + dr * WWG_NT * 4, 0);
            }
    }
  }   // next (patch, step range) of a tail block
}

// dw[i][j][c][n] = sum_ab G[a][i] G[b][j] s(a) s(b) sum_slot part[patch][slot][4a+b][c % 64][n % 64],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
// s(3) = -1 (the signs of A's last row / column the main kernel left out), s = +1 otherwise.  Slots are summed in index order:
// the `full` chunk partials, then the tail pieces of this patch.
__global__ void __launch_bounds__(256) conv3_wino_wgrad_finalize_kernel(const float* __restrict__ part, float* __restrict__ dw, int cin, int cout, int npatches,
                                                                        int full, int tail, int span, int slots) {
    const long long cn = (long long)cin * cout;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cn) return;
    const int c = (int)(i / cout), n = (int)(i - (long long)c * cout);
    const int patch = (c / WWG_KT) * npatches + n / WWG_NT;
    const int local = (c % WWG_KT) * WWG_NT + n % WWG_NT;
    int nslots = full;
    if (tail > 0) nslots += (int)((((long long)patch + 1) * tail - 1) / span) - wwg_tail_first_block(patch, tail, span) + 1;
    const float* base = part + (long long)patch * slots * 16 * (WWG_KT * WWG_NT) + local;
    float u[4][4];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float s = 0.f;
        for (int sp = 0; sp < nslots; ++sp) s += base[((long long)sp * 16 + k) * (WWG_KT * WWG_NT)];
        u[k >> 2][k & 3] = ((k >> 2) == 3) != ((k & 3) == 3) ? -s : s;
    }
    float tmp[3][4];      // G^T u
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        tmp[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
        tmp[1][b] = 0.5f * (u[1][b] - u[2][b]);
        tmp[2][b] = u[3][b] + 0.5f * (u[1][b] + u[2][b]);
    }
#pragma unroll
    for (int ii = 0; ii < 3; ++ii) {
        dw[(ii * 3 + 0) * cn + i] = tmp[ii][0] + 0.5f * (tmp[ii][1] + tmp[ii][2]);
        dw[(ii * 3 + 1) * cn + i] = 0.5f * (tmp[ii][1] - tmp[ii][2]);
        dw[(ii * 3 + 2) * cn + i] = tmp[ii][3] + 0.5f * (tmp[ii][1] + tmp[ii][2]);
    }
}

// act(scale * x + shift) of the [n][h][w][ldx] input (channels c_from .. cin-1) into the zero-bordered [n][h+2][w+2][cin] copy; the
// channels below c_from already hold their values (written there by their producer: ssdseg_bilinear_fwd_padded) over a border that
// was zero when the buffer was made
__global__ void __launch_bounds__(256) conv3_pad_view_kernel(const float* __restrict__ x, const float* __restrict__ cs, const float* __restrict__ ct, int act, int ldx,
                                                             float* __restrict__ xp, int n, int h, int w, int cin, int c_from) {
    const int cv = (cin - c_from) >> 2, q0 = c_from >> 2, cvall = cin >> 2;
    const long long total = (long long)n * (h + 2) * (w + 2) * cv;
    const float lo = act_lo(act), hi = act_hi(act);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int q = q0 + (int)(i % cv);
        long long pix = i / cv;
        const long long pixel = pix;
        const int pw = (int)(pix % (w + 2));
        pix /= (w + 2);
        const int ph = (int)(pix % (h + 2)), img = (int)(pix / (h + 2));
        float4 v = f4(0.f);
        if (ph >= 1 && ph <= h && pw >= 1 && pw <= w) {
            v = ld4(x + (((long long)img * h + ph - 1) * w + pw - 1) * ldx + 4 * q);
            const float4 s = cs != nullptr ? ld4(cs + 4 * q) : f4(1.f), tt = cs != nullptr ? ld4(ct + 4 * q) : f4(0.f);
            v = view_affine4(v, s, tt, lo, hi);
        }
        st4(xp + (pixel * cvall + q) * 4, v);
    }
}
