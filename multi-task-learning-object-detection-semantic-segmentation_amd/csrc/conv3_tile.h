// Dense 3x3 stride-1 SAME convolution (DeepLabV3+ decoder, reference blocks.py:117) forward and input gradient as a
// HALO-TILE kernel -- included by gemm.hip inside its anonymous namespace.
//
// The implicit-GEMM form (gemm_rowA_kernel<.., LD = 1>) gathers every input pixel nine times from global memory, once per
// filter tap, and re-applies the BatchNorm view each time: 13.9 GB fetched per launch for 1.38 GB of tensors, ~180 VALU
// instructions of address / view arithmetic per 64 MFMAs, 0.56-0.65 of the fp32 MFMA peak (profiles/r01_*).  Here a block of
// EIGHT waves owns an 8-row x 32-column patch of output pixels and 32*WN output channels; per step of 8 input channels it
// stages ONCE
//     the (8+2) x (32+2) halo of the input, view applied while staging (one affine + clamp per element, not nine), and
//     the 9 x (32*WN) x 8 slice of the weights, reduction channel contiguous,
// into LDS, and the nine taps are nine shifted READS of that patch.  Both operand fragments are 16-byte LDS reads of four
// consecutive reduction channels: lanes 0-31 take channels 0-3, lanes 32-63 channels 4-7 of the step, so MFMA #j of a tap
// multiplies the channel pair (j, 4 + j) -- any pairing of the reduction index is a valid GEMM order.  Per tap and wave:
// 1 + WN ds_read_b128, 4 * WN v_mfma_f32_32x32x2_f32, nothing else.  Two LDS buffers, ONE barrier per step: the global loads
// of step s+1 are issued before the MFMAs of step s and written to the other buffer after them.
// LDS rows are 32 bytes (one pixel / one output channel x 8 reduction channels); the two 16-byte halves of a row are swapped
// on odd 8-row groups (`^ ((row >> 3) & 1)`), which makes every ds_read_b128 lane group hit 16 distinct 4-bank slots.
//
//   MODE 0 (forward):   out[p][n] = sum_{tap, c} a[p + d(tap)][c] * W[tap][c][n]     weights pre-transposed to [tap][n][c]
//   MODE 1 (backward):  dx[p][c]  = sum_{tap, n} dy[p - d(tap)][n] * W[tap][c][n]    = the same loop with the taps mirrored and
//                                                                                      W read in its native layout
// Forward also emits the BatchNorm partial sums (sum, sum of squares per output channel) of its tile: one partial row per
// pixel tile, fixed order, no atomics.
#pragma once

constexpr int C3T_ROWS = 8;                       // output rows per block = waves per block
constexpr int C3T_COLS = 32;                      // output columns per block = MFMA rows per wave
constexpr int C3T_KC = 8;                         // reduction channels per step
constexpr int C3T_THREADS = 64 * C3T_ROWS;
constexpr int C3T_PW = C3T_COLS + 2;              // patch width (pixels)
constexpr int C3T_PIX = (C3T_ROWS + 2) * C3T_PW;  // patch pixels (340)
constexpr int C3T_PATCH_F = C3T_PIX * C3T_KC;     // floats per patch buffer

struct Conv3TArgs {
    const float* in;     // [n][h][w][ldi] raw input (forward) / materialised dy (backward)
    const float* cs;     // view of the input: act(cs*x + ct); nullptr = identity
    const float* ct;
    int act, ldi;
    const float* wt;     // [9][nout][cred], reduction channel contiguous
    float* out;          // [n][h][w][ldo]
    int ldo, accumulate;
    float* stats;        // forward: [mtiles][2][nout] partial (sum, sumsq); may be nullptr
    int n, h, w;
    int cred, nout;      // reduction channels (multiple of 8), output channels
    int tiles_h, tiles_w, ntiles_n;
    int ncols;           // output channels per column tile (<= 32*WN)
    int flip;            // backward: patch tap t uses the weights of tap 8 - t
    unsigned in_bytes, wt_bytes;   // extents for the buffer descriptors (both < 2^31)
    int in_hp, in_wp;              // conv3_wino_kernel only: rows per image / pixels per row of the INPUT buffer (h, w; h+2, w+2 for the zero-bordered copy)
};

constexpr size_t conv3t_lds_floats(int wn, int cred) { return 2 * (size_t)(C3T_PATCH_F + 9 * 32 * wn * C3T_KC) + 2 * (size_t)cred; }

template <int WN>
__global__ void __launch_bounds__(C3T_THREADS, 2) conv3_tile_kernel(Conv3TArgs p) {
    constexpr int BN = 32 * WN;
    constexpr int W_F = 9 * BN * C3T_KC;               // floats per weight buffer
    constexpr int BUF_F = C3T_PATCH_F + W_F;
    constexpr int PSLOTS = C3T_PIX * 2;                // float4 slots of a patch (680)
    constexpr int PQ = (PSLOTS + C3T_THREADS - 1) / C3T_THREADS;
    constexpr int WSLOTS = 9 * BN * 2;                 // float4 slots of a weight slice
    constexpr int WQ = (WSLOTS + C3T_THREADS - 1) / C3T_THREADS;
    extern __shared__ float smem[];
    float* coef = smem + 2 * BUF_F;                    // [2][cred]: scale, shift of the input view

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;

    // XCD-aware logical block id: the column tiles of one pixel tile and spatially neighbouring pixel tiles run on one XCD
    // (they share the halo and the input patch through that XCD's L2); speed only, any order is correct
    const unsigned total = gridDim.x;
    unsigned L = blockIdx.x;
    if ((total & 7u) == 0u) L = (L & 7u) * (total >> 3) + (L >> 3);
    const int ntile = (int)(L % (unsigned)p.ntiles_n);
    const int mtile = (int)(L / (unsigned)p.ntiles_n);
    const int tw = mtile % p.tiles_w;
    const int th = (mtile / p.tiles_w) % p.tiles_h;
    const int img = mtile / (p.tiles_w * p.tiles_h);
    const int h0 = th * C3T_ROWS, w0 = tw * C3T_COLS;
    const int n0 = ntile * p.ncols;

    const bool affine = p.cs != nullptr;
    const float alo = act_lo(p.act), ahi = act_hi(p.act);
    for (int i = t; i < p.cred; i += C3T_THREADS) {
        coef[i] = affine ? p.cs[i] : 1.f;
        coef[p.cred + i] = affine ? p.ct[i] : 0.f;
    }

    // ---- staging slots of this thread (fixed for the whole tile).  Global reads are raw buffer loads: a 32-bit byte offset per
    // slot (no 64-bit address arithmetic in the loop, the step's channel offset rides in the scalar offset) and hardware range
    // checking -- a slot outside the image / beyond the last output channel carries offset 2^31, beyond num_records, and
    // loads zeros.
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.wt_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned pgo[PQ], wgo[WQ];
    int plo[PQ], wlo[WQ];
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
    for (int q = 0; q < WQ; ++q) {
        const int slot = t + C3T_THREADS * q;
        const int chunk = slot & 1, rowi = slot >> 1;
        const int nn = rowi % BN, tap = rowi / BN;
        const int gtap = p.flip ? 8 - tap : tap;
        const bool ok = slot < WSLOTS && nn < p.ncols && n0 + nn < p.nout;
        wgo[q] = ok ? (unsigned)((((long long)gtap * p.nout + n0 + nn) * p.cred + 4 * chunk) * 4) : OOB;
        wlo[q] = slot < WSLOTS ? C3T_PATCH_F + rowi * C3T_KC + 4 * (chunk ^ ((nn >> 3) & 1)) : -1;
    }
    const int pchunk = t & 1;   // (C3T_THREADS is even: every slot of a thread has the same 4-channel half)

    float4 preg[PQ], wreg[WQ];
    auto issue = [&](int s) {
        const int soff = s * C3T_KC * 4;
#pragma unroll
        for (int q = 0; q < PQ; ++q) preg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rin, pgo[q], soff, 0));
#pragma unroll
        for (int q = 0; q < WQ; ++q) wreg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rwt, wgo[q], soff, 0));
    };
    auto commit = [&](int s, float* buf) {
        const int c0 = s * C3T_KC + 4 * pchunk;
        const float4 cs = ld4(coef + c0), ct = ld4(coef + p.cred + c0);
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            if (plo[q] >= 0) st4(buf + plo[q], ((inimg >> q) & 1u) ? view_affine4(preg[q], cs, ct, alo, ahi) : f4(0.f));
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            if (wlo[q] >= 0) st4(buf + wlo[q], wreg[q]);
        }
    };

    // ---- fragment addresses (floats, relative to a buffer): A per tap (the swizzle bit depends on the shifted pixel), B one base
    int aoff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int pix = (wave + tap / 3) * C3T_PW + li + tap % 3;
        aoff[tap] = pix * C3T_KC + 4 * (hh ^ ((pix >> 3) & 1));
    }
    const int boff = C3T_PATCH_F + li * C3T_KC + 4 * (hh ^ ((li >> 3) & 1));

    f32x16 acc[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    // One step = 9 taps x WN column tiles = 9*WN units of four MFMAs (channel pairs (j, 4+j), j = 0..3, into ONE accumulator: the
    // 32x32x2 fp32 MFMA issues back to back on a dependent accumulator).  The 16-byte B fragment of unit u+1 -- and, at a tap
    // boundary, the A fragment of the next tap -- is read while the four MFMAs of unit u run (256 cycles >> LDS latency), so only
    // two fragment sets are live: 16 registers instead of 2 * (4 + 4*WN).
    auto compute = [&](const float* buf) {
        float4 afr[2], bfr[2];
        afr[0] = ld4(buf + aoff[0]);
        bfr[0] = ld4(buf + boff);
#pragma unroll
        for (int u = 0; u < 9 * WN; ++u) {
            const int tap = u / WN, nt = u - tap * WN;
            if (u + 1 < 9 * WN) {
                const int tap1 = (u + 1) / WN, nt1 = (u + 1) - tap1 * WN;
                bfr[(u + 1) & 1] = ld4(buf + boff + (tap1 * BN + nt1 * 32) * C3T_KC);
                if (nt1 == 0) afr[tap1 & 1] = ld4(buf + aoff[tap1]);
            }
            __builtin_amdgcn_sched_barrier(0);   // the next unit's reads stay in front of this unit's MFMAs
            const float4 a = afr[tap & 1], b = bfr[u & 1];
            acc[nt] = mfma32(a.x, b.x, acc[nt]);
            acc[nt] = mfma32(a.y, b.y, acc[nt]);
            acc[nt] = mfma32(a.z, b.z, acc[nt]);
            acc[nt] = mfma32(a.w, b.w, acc[nt]);
        }
    };

    const int S = p.cred / C3T_KC;
    float* buf0 = smem;
    float* buf1 = smem + BUF_F;
    issue(0);
    __syncthreads();            // coef[] visible
    commit(0, buf0);
    __syncthreads();
    for (int s = 0; s < S; s += 2) {
        if (s + 1 < S) issue(s + 1);
        compute(buf0);
        if (s + 1 < S) commit(s + 1, buf1);
        __syncthreads();
        if (s + 1 >= S) break;
        if (s + 2 < S) issue(s + 2);
        compute(buf1);
        if (s + 2 < S) commit(s + 2, buf0);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane & 31 (output channel), row = (e & 3) + 8 * (e >> 2) + 4 * hh (pixel column of the tile)
    const int oh = h0 + wave;
    const bool rowok = oh < p.h;
    float ssum[WN], ssq[WN];
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        ssum[nt] = ssq[nt] = 0.f;
        const int jl = nt * 32 + li, j = n0 + jl;
        const bool jok = jl < p.ncols && j < p.nout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ow = w0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            const bool ok = rowok && ow < p.w && jok;
            float v = acc[nt][e];
            if (ok) {
                float* o = p.out + (((long long)img * p.h + oh) * p.w + ow) * p.ldo + j;
                if (p.accumulate) v += *o;
                *o = v;
                ssum[nt] += acc[nt][e];
                ssq[nt] = fmaf(acc[nt][e], acc[nt][e], ssq[nt]);
            }
        }
    }
    if (p.stats != nullptr) {
        float* red = smem;   // [8 waves][2][BN] (the operand buffers are dead: every wave passed the last barrier)
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
        for (int idx = t; idx < 2 * BN; idx += C3T_THREADS) {
            const int which = idx / BN, jl = idx - which * BN;
            const int j = n0 + jl;
            if (jl < p.ncols && j < p.nout) {
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < C3T_ROWS; ++wv) v += red[(wv * 2 + which) * BN + jl];
                p.stats[((long long)mtile * 2 + which) * p.nout + j] = v;
            }
        }
    }
}

// W[tap][c][n] -> Wt[tap][n][c] (forward: the reduction channel must be contiguous in the staged weight rows)
__global__ void __launch_bounds__(256) conv3_transpose_w_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin, int cout) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        tile[r][tx] = (c < cin && n < cout) ? w[((long long)tap * cin + c) * cout + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        if (n < cout && c < cin) wt[((long long)tap * cout + n) * cin + c] = tile[tx][r];
    }
}
