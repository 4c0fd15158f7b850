// Weight gradient of the dense 3x3 stride-1 SAME convolution (reference blocks.py:117) -- halo-tile form, included by gemm.hip
// inside its anonymous namespace.
//
//   dW[kh][kw][c][n] = sum_p a[p + (kh-1, kw-1)][c] * dy[p][n]  =  sum_q a[row + kh - 1][q][c] * dy[row][q - (kw - 1)][n]
//
// (second form: the column sum re-indexed by the INPUT column q = p + kw - 1).  With it the nine taps of one reduction step
// need only THREE a-fragments (input rows row-1, row, row+1 at column q) and THREE dy-fragments (columns q+1, q, q-1 of row
// `row`): 6 LDS reads per 9 MFMAs instead of the 18 of the one-wave-per-tap kernels (conv3_wgrad.h), and the halo moves to dy
// (one extra column each side) and to a (one extra row above and below).
//
// Block = 4 waves = a 64 (input channels) x 64 (output channels) tile of all nine taps: wave w owns the 32 x 32 (k, n) sub-tile
// (w >> 1, w & 1) of every tap -- nine accumulators, 144 registers.  One step = one image row x 32 columns of pixels: the block
// stages a[3 rows][32 cols][64 ch] (through registers: the view -- BatchNorm affine + ReLU6, zero outside the image -- is
// applied once, while staging) and dy[34 cols][64 ch] (global -> LDS directly, buffer_load ... lds: the caller hands over a
// materialised gradient, so no arithmetic is due on the way and no staging registers are held); every wave then runs 16
// reduction steps (pixel pairs) x 9 MFMAs = 144 v_mfma_f32_32x32x2_f32 between two barriers (48 in conv3_wgrad.h).  Two LDS
// buffers (67 KB per block), one barrier per step, TWO blocks per CU: while one block stages, the other owns the matrix pipes.
// Register budget matters more than usual here: a spill reload shares the vector-memory counter with the prefetch loads, so a
// single reload in the loop waits for the whole prefetch (an eight-wave / two-row-step version with 36 staging registers
// spilled 30-60 registers and ran 102 TFLOP/s).
// The reduction over pixels is split across blocks in contiguous step ranges; partial slabs [split][9][K][N] are folded in a
// fixed order by colsum (deterministic).  The 20 (k, n) tiles of one split get consecutive XCD-aware block ids: they read the
// same pixels at the same time through one XCD's L2.
#pragma once

constexpr int W3T_KT = 64, W3T_NT = 64;          // input / output channels per block
constexpr int W3T_COLS = 32;                     // pixels per step (one image row)
constexpr int W3T_DW = 36;                       // dy columns staged per step: 34 used (one halo column each side), padded to whole waves
constexpr int W3T_THREADS = 256;
constexpr int W3T_A_F = 3 * W3T_COLS * W3T_KT;   // floats of the a patch (6144)
constexpr int W3T_D_F = W3T_DW * W3T_NT;         // floats of the dy patch (2304)
constexpr int W3T_BUF_F = W3T_A_F + W3T_D_F;
constexpr size_t W3T_LDS_BYTES = 2 * (size_t)W3T_BUF_F * sizeof(float);

struct Wg3TArgs {
    const float* x;      // [n][h][w][ldx] raw input
    const float* xs;     // view act(xs*x + xt); nullptr = identity
    const float* xt;
    int xact, ldx;
    const float* g;      // [n][h][w][N] dy (identity gradient view: the caller materialises BatchNorm views first)
    float* part;         // [splits][9][K][N]
    int n, h, w, K, N;
    int ktiles, ntiles;
    int strips;          // ceil(w / 32)
    int steps;           // n * strips * h
    int steps_per_split;
    unsigned x_bytes, g_bytes;
};

// bijective XCD-aware renumbering: blocks that share an XCD (b % 8 equal) get a contiguous range of logical ids
__device__ __forceinline__ unsigned xcd_logical_id(unsigned b, unsigned total) {
    const unsigned q = total >> 3, r = total & 7u, x = b & 7u;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

__global__ void __launch_bounds__(W3T_THREADS, 2) conv3_wgrad_tile_kernel(Wg3TArgs p) {
    constexpr int AQ = W3T_A_F / 4 / W3T_THREADS;      // float4 slots of a per thread (6)
    constexpr int DV = W3T_D_F / 4;                    // float4 slots of dy (576 = 9 waves)
    constexpr int DQ = (DV + W3T_THREADS - 1) / W3T_THREADS;
    static_assert(DV % 64 == 0 && W3T_A_F / 4 % W3T_THREADS == 0, "staging slots come in whole waves / whole passes");
    extern __shared__ float smem[];
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;
    const int kt = wave >> 1, nt = wave & 1;

    const unsigned L = xcd_logical_id(blockIdx.x, gridDim.x);
    const int tiles = p.ktiles * p.ntiles;
    const int tile = (int)(L % (unsigned)tiles);
    const int split = (int)(L / (unsigned)tiles);
    const int k0 = (tile / p.ntiles) * W3T_KT, n0 = (tile % p.ntiles) * W3T_NT;
    const int s0 = split * p.steps_per_split;
    int s1 = s0 + p.steps_per_split;
    if (s1 > p.steps) s1 = p.steps;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, p.g_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;    // beyond num_records (< 2^31): the load returns zeros / the store is dropped

    // staging slots: a slot = (pixel, float4 channel column); 16 float4 per pixel for both tensors, so the channel column of a
    // thread is the same for all its slots (256 % 16 == 0) and the view coefficients live in registers
    const int c4 = t & 15;
    const bool akok = k0 + c4 * 4 < p.K, dnok = n0 + c4 * 4 < p.N;
    float4 cxs = f4(1.f), cxt = f4(0.f);
    if (p.xs != nullptr && akok) { cxs = ld4(p.xs + k0 + c4 * 4); cxt = ld4(p.xt + k0 + c4 * 4); }
    const float xlo = act_lo(p.xact), xhi = act_hi(p.xact);
    const int pixq = t >> 4;                 // pixel of slot q = pixq + 16*q

    int st_row, st_strip, st_img;            // position of the NEXT step to issue (wave-uniform; advanced by increments)
    {
        st_row = s0 % p.h;
        const int rest = s0 / p.h;
        st_strip = rest % p.strips;
        st_img = rest / p.strips;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    float4 areg[AQ];
    unsigned aok = 0;
    auto issue = [&](float* nextbuf) {
        const int r0 = st_row, w0 = st_strip * W3T_COLS;
        const int pix0 = (st_img * p.h + r0) * p.w + w0;       // first pixel of the step (element offsets < 2^29: checked by the launcher)
#pragma unroll
        for (int q = 0; q < DQ; ++q) {
            const int dcol = pixq + 16 * q;                    // 0..35 (slots beyond 36 columns do not exist: see below)
            const int gw = w0 - 1 + dcol;
            const bool ok = dnok && dcol < W3T_COLS + 2 && gw >= 0 && gw < p.w;
            const unsigned off = ok ? (unsigned)((pix0 + dcol - 1) * p.N + n0 + c4 * 4) * 4u : OOB;
            if (wave_u * 64 + W3T_THREADS * q < DV) {          // wave-uniform: the last pass exists for wave 0 only
                float* dst = nextbuf + W3T_A_F + (wave_u * 64 + W3T_THREADS * q) * 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
            }
        }
        aok = 0;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int pix = pixq + 16 * q;                     // 0..95: (patch row, column)
            const int prow = pix >> 5, col = pix & 31;
            const int gh = r0 - 1 + prow, gw = w0 + col;
            const bool ok = akok && gh >= 0 && gh < p.h && gw < p.w;
            const unsigned off = ok ? (unsigned)((pix0 + (prow - 1) * p.w + col) * p.ldx + k0 + c4 * 4) * 4u : OOB;
            areg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            aok |= (ok ? 1u : 0u) << q;
        }
        if (++st_row == p.h) {
            st_row = 0;
            if (++st_strip == p.strips) { st_strip = 0; ++st_img; }
        }
    };
    auto commit = [&](float* buf) {
#pragma unroll
        for (int q = 0; q < AQ; ++q)
            st4(buf + (t + W3T_THREADS * q) * 4, ((aok >> q) & 1u) ? view_affine4(areg[q], cxs, cxt, xlo, xhi) : f4(0.f));
    };

    f32x16 acc[9];
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;

    // reduction step j of a staged tile: pixel pair (columns 2*j + hh)
    //   A[kh] = a[kh][col][k0 + 32*kt + li]               (patch row 0 is image row r0 - 1)
    //   B[kw] = dy[col + 2 - kw][n0 + 32*nt + li]         (halo column 0 is image column w0 - 1)
    const int abase = hh * W3T_KT + kt * 32 + li;
    const int bbase = W3T_A_F + hh * W3T_NT + nt * 32 + li;
    auto compute = [&](const float* buf) {
        float af[2][3], bf[2][3];
        auto fetch = [&](int j, int set) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) af[set][kh] = buf[abase + (kh * W3T_COLS + 2 * j) * W3T_KT];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) bf[set][kw] = buf[bbase + (2 * j + 2 - kw) * W3T_NT];
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < W3T_COLS / 2; ++j) {
            if (j + 1 < W3T_COLS / 2) fetch(j + 1, (j + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = mfma32(af[j & 1][kh], bf[j & 1][kw], acc[kh * 3 + kw]);
        }
    };

    // One copy of the step body, the two buffers selected by pointer swap; issue() / commit() are unconditional -- the last step
    // prefetches one step beyond the split's range into the buffer nobody will read (range-checked loads: harmless).  With the
    // pair under `if`s, or with the body replicated per buffer, the compiler's wait-count pass saw loads pending at the loop
    // header and put counted vmcnt waits into issue() that at run time waited for the just-started LDS-DMA (and the replicated
    // bodies made the register allocator spill).
    if (s0 < s1) {
        float* cur = smem;
        float* nxt = smem + W3T_BUF_F;
        issue(cur);
        commit(cur);
        __syncthreads();            // (waits for the LDS-DMA of every wave as well: vmcnt(0) in front of the barrier)
        for (int s = s0; s < s1; ++s) {
            issue(nxt);             // nxt was last read in step s - 1, before the barrier every wave has passed
            compute(cur);
            commit(nxt);
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
    }

    // partial slab of this split: C/D layout col = lane & 31 (n), row = (e & 3) + 8 * (e >> 2) + 4 * hh (k).  Raw buffer stores:
    // sixteen 32-bit element offsets per lane (rows / columns outside [K) x [N) carry an out-of-range offset and are dropped by
    // the hardware range check), the tap's slab offset rides in the scalar offset -- no 64-bit address per element.
    float* slab = p.part + (long long)split * 9 * p.K * p.N;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (unsigned)(9LL * p.K * p.N * 4), 0x00020000);
    const int n = n0 + nt * 32 + li;
    unsigned eoff[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = k0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        eoff[e] = (k < p.K && n < p.N) ? (unsigned)((k * p.N + n) * 4) : OOB;
    }
    const int tapbytes = p.K * p.N * 4;
#pragma unroll
    for (int u = 0; u < 9; ++u) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[u][e];   // (a bit_cast applied directly to the vector element stored element 0 sixteen times)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, eoff[e], u * tapbytes, 0);
        }
    }
}
