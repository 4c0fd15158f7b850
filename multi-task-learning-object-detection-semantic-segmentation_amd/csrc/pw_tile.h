// Pointwise (1x1) convolution forward / input gradient as a double-buffered tile GEMM in the style of conv3_tile.h -- included by
// gemm.hip inside its anonymous namespace.  (reference models.py:65,110; blocks.py:28,58,70,109: Conv2D 1x1 and the pointwise half
// of SeparableConv2D)
//
//   MODE 0:  out[m][n] = sum_k act(cs_k * x[m][k] + ct_k) * W[k][n]          weights read from a transposed copy Wt[n][k]
//   MODE 1:  out[m][k] = sum_n dy[m][n] * W[k][n]   (+ residual, + accumulate)  dy = BatchNorm-backward gradient view of (g, y)
//            formed while staging; W in its native layout (the reduction index n is contiguous)
//
// Differences from gemm_rowA_kernel (which stays for shapes this kernel does not take): BOTH operands sit in LDS with the
// reduction index contiguous -- 128-byte rows of 32 reduction channels, the eight 16-byte chunks of a row XOR-swizzled by
// (row >> 1) & 7 so that every ds_read_b128 lane group hits 16 distinct 4-bank slots -- so a fragment of four consecutive
// reduction channels is ONE ds_read_b128 for either operand (the B fragment was four ds_read_b32); two LDS buffers and ONE
// barrier per 32-deep step (were two); global reads are raw buffer loads with per-slot 32-bit offsets and hardware range
// checking (rows beyond M / columns beyond N read zeros: no branches, no 64-bit address arithmetic in the loop); weights go
// global -> LDS directly (buffer_load ... lds; the swizzle is applied to the SOURCE chunk each lane fetches, the LDS image a
// wave writes stays lane-linear).  A block walks row tiles blockIdx.y, blockIdx.y + gridDim.y, ... so the BatchNorm partial
// tables stay short.  Per (k-group, column tile): one ds_read_b128 + four back-to-back v_mfma_f32_32x32x2_f32.
#pragma once

constexpr int PWT_KC = 32;                       // reduction channels per step (one 128-byte LDS row)

struct PwTArgs {
    const float* a0;     // MODE 0: x [M][lda]; MODE 1: g [M][lda]
    const float* a1;     // MODE 1: raw y of the gradient view (== a0 for identity views)
    const float* cs;     // per-reduction-channel coefficients, nullptr = identity
    const float* ct;
    const float* ck1;
    const float* ck0;
    int act, lda;
    const float* wt;     // [nout][cred], reduction index contiguous
    float* out;
    int ldo;
    const float* residual;
    int ldr, accumulate;
    float* stats;        // MODE 0: [gridDim.y][2][nout] partial (sum, sumsq); may be nullptr
    int M, cred, nout;
    int ncols;           // output columns per column tile (<= 32*WN)
    // MODE 1, fused BatchNorm backward of the producer of this conv's INPUT: raw input bn_y [M][ldby], that BN's scale / shift /
    // mean / invstd / activation; bnpart [gridDim.y][2][nout] partial rows of (sum mask*dx, sum mask*dx*xhat); nullptr = off
    const float* bn_y;
    const float* bn_s;
    const float* bn_t;
    const float* bn_mean;
    const float* bn_istd;
    int bn_act, ldby;
    float* bnpart;
    unsigned a_bytes, wt_bytes;
};

// global -> LDS copy of 16 bytes per lane (buffer_load_dwordx4 ... lds), written as inline asm ON PURPOSE: through the builtin the
// compiler's wait-count pass orders every later ds_read behind the pending LDS write (it cannot tell the two buffers apart) and
// put `s_waitcnt vmcnt(<loads issued after the DMA>)` in front of the first fragment read of every step -- i.e. the prefetch was
// waited for right after it was issued.  An asm statement is invisible to that pass; completion is OUR job: the DMA is older
// than the step's register loads (the counter is in-order), commit() ends with an explicit vmcnt(0), and the barrier follows.
// lds_byte_addr: wave-uniform LDS address of lane 0's 16 bytes (lane l lands at +16*l); M0 is saved and restored in the statement.
typedef int pwt_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pwt_i32x4 pwt_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    pwt_i32x4 r;
    r.x = (int)(unsigned)(a & 0xffffffffull);
    r.y = (int)(unsigned)((a >> 32) & 0xffffull);     // stride 0
    r.z = (int)bytes;                                    // num_records (bytes): offsets beyond it read zeros
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void lds_dma16(pwt_i32x4 rsrc, unsigned lds_byte_addr, unsigned voffset, int soffset) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}

// (rows, cols): 32 * waves-along-M, 32 * WN * waves-along-N
constexpr size_t pwt_lds_floats(int rows, int cols, int cred) { return 2 * (size_t)(rows + cols) * PWT_KC + 4 * (size_t)((cred + 31) / 32 * 32 + PWT_KC); }

// WNW: waves along the columns.  WNW = 1: every wave owns 32 rows of the block and all 32 * WN columns.  WNW = 2: the eight waves form a
// 4 x 2 grid over a 128-row x (2 * 32 * WN)-column block tile, each wave 32 rows x 32 * WN columns -- for the input gradient of a
// 256 -> 256 conv (the decoder sepconv) the block then spans ALL output columns with 64 accumulator registers per wave, and the
// two-tensor gradient view is streamed ONCE instead of once per 128-column tile.
template <int WAVES, int WN, int MODE, int WNW = 1>
__global__ void __launch_bounds__(64 * WAVES, 2) pw_tile_kernel(PwTArgs p) {
    constexpr int T = 64 * WAVES;
    constexpr int WM = WAVES / WNW;
    static_assert(WM * WNW == WAVES, "wave grid");
    constexpr int BM = 32 * WM, BN = 32 * WN * WNW;
    constexpr int A_F = BM * PWT_KC, B_F = BN * PWT_KC, BUF_F = A_F + B_F;
    constexpr int AQ = BM * 8 / T;                      // float4 slots of A per thread (4)
    constexpr int BSLOTS = BN * 8;
    constexpr int BQ = (BSLOTS + T - 1) / T;
    static_assert(BM * 8 % T == 0 && BSLOTS % 64 == 0, "staging slots come in whole passes / whole waves");
    extern __shared__ float smem[];
    const int credp = (p.cred + 31) / 32 * 32 + PWT_KC; // one spare step: the loop prefetches (and commits) one step beyond the reduction
    float* coef = smem + 2 * BUF_F;                     // [4][credp]: cs, ct, ck1, ck0 (identity values where absent)

    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63, li = lane & 31, hh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave % WM, wnw = wave / WM;          // this wave's 32 rows / its group of 32 * WN columns
    const int n0 = blockIdx.x * p.ncols;
    const int mtiles = (p.M + BM - 1) / BM;
    const int S = (p.cred + PWT_KC - 1) / PWT_KC;

    const bool affine = p.cs != nullptr;
    for (int i = t; i < credp; i += T) {
        const bool ok = affine && i < p.cred;
        coef[i] = ok ? p.cs[i] : 1.f;
        coef[credp + i] = ok ? p.ct[i] : 0.f;
        coef[2 * credp + i] = (ok && MODE == 1) ? p.ck1[i] : 0.f;
        coef[3 * credp + i] = (ok && MODE == 1) ? p.ck0[i] : 0.f;
    }
    const float alo = act_lo(p.act), ahi = act_hi(p.act);
    const int gact = affine ? p.act : SSDSEG_ACT_NONE;  // identity gradient view: mask == 1

    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a0), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MODE == 1 ? p.a1 : p.a0), 0, p.a_bytes, 0x00020000);
    const pwt_i32x4 rwt = pwt_make_rsrc(p.wt, p.wt_bytes);
    constexpr unsigned OOB = 0x80000000u;

    // staging slots: slot = (row, 16-byte chunk of the step's 128 bytes); a thread's chunk is the same for all its slots
    const int chunk = t & 7, srow = t >> 3;             // rows srow + (T/8)*q
    int alds[AQ];                                        // LDS float offsets of the A slots (swizzled)
#pragma unroll
    for (int q = 0; q < AQ; ++q) {
        const int row = srow + (T / 8) * q;
        alds[q] = row * PWT_KC + 4 * (chunk ^ ((row >> 1) & 7));
    }
    // B by LDS-DMA: lane-linear LDS image (slot = t + T*q  <->  row = slot >> 3, LDS chunk = slot & 7), swizzle on the SOURCE chunk
    unsigned bgo[BQ];
    int bsch[BQ];
#pragma unroll
    for (int q = 0; q < BQ; ++q) {
        const int slot = t + T * q;
        const int row = slot >> 3, lchunk = slot & 7;
        bsch[q] = lchunk ^ ((row >> 1) & 7);
        const bool ok = slot < BSLOTS && row < p.ncols && n0 + row < p.nout;
        bgo[q] = ok ? (unsigned)(((long long)(n0 + row) * p.cred + 4 * bsch[q]) * 4) : OOB;
    }

    // fragment addresses (floats, relative to a buffer): (row li of the wave's / column tile's 32 rows, k-group g, half hh)
    int fo[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) fo[g] = li * PWT_KC + 4 * ((2 * g + hh) ^ ((li >> 1) & 7));
    const int abase = wm * 32 * PWT_KC;
    const int bbase = wnw * WN * 32 * PWT_KC;

    float ssum[WN], ssq[WN];          // MODE 0: BatchNorm statistics of this block's rows, carried across its row tiles
    float esb[WN], esg[WN];           // MODE 1 + bnpart: running sums of the fused BatchNorm backward
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) ssum[nt] = ssq[nt] = esb[nt] = esg[nt] = 0.f;
    const bool bne = MODE == 1 && p.bnpart != nullptr;
    const float bnlo = act_lo(p.bn_act), bnhi = act_hi(p.bn_act);

    __syncthreads();                  // coef[] visible

    for (int mt = blockIdx.y; mt < mtiles; mt += gridDim.y) {
        const int m0 = mt * BM;
        unsigned ago[AQ];
        unsigned rowok = 0;
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int m = m0 + srow + (T / 8) * q;
            const bool ok = m < p.M;
            ago[q] = ok ? (unsigned)(((long long)m * p.lda + 4 * chunk) * 4) : OOB;
            rowok |= (ok ? 1u : 0u) << q;
        }
        float4 areg[AQ], yreg[MODE == 1 ? AQ : 1];
        auto issue = [&](int s, float* nextbuf) {
            const int soff = s * PWT_KC * 4;
            // the last step of a reduction that is not a multiple of 32: channels beyond it get bit 31 set in their offset (out of
            // range -> zeros); an OR, not a select, so that no branch is built around the loads
            const unsigned koob = (s * PWT_KC + 4 * chunk < p.cred) ? 0u : OOB;
#pragma unroll
            for (int q = 0; q < BQ; ++q) {
                // whole passes are unconditional (a compile-time fact); only a partial last pass asks which waves still have slots
                if (T * (q + 1) <= BSLOTS || wave_u * 64 + T * q < BSLOTS) {
                    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(((int)(nextbuf - smem) + A_F + (wave_u * 64 + T * q) * 4) * 4);
                    const unsigned boob = (s * PWT_KC + 4 * bsch[q] < p.cred) ? 0u : OOB;
                    lds_dma16(rwt, dst, bgo[q] | boob, soff);
                }
            }
#pragma unroll
            for (int q = 0; q < AQ; ++q) {
                const unsigned off = ago[q] | koob;
                areg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra0, off, soff, 0));
                if (MODE == 1) yreg[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra1, off, soff, 0));
            }
        };
        auto commit = [&](int s, float* buf) {
            const int c0 = s * PWT_KC + 4 * chunk;
            const float4 cs = ld4(coef + c0), ct = ld4(coef + credp + c0);
            float4 k1 = f4(0.f), k0 = f4(0.f);
            if (MODE == 1) { k1 = ld4(coef + 2 * credp + c0); k0 = ld4(coef + 3 * credp + c0); }
            const bool kok = c0 < p.cred;
#pragma unroll
            for (int q = 0; q < AQ; ++q) {
                float4 v;
                if (MODE == 0) v = view_affine4(areg[q], cs, ct, alo, ahi);
                else v = gview_apply4(areg[q], yreg[q], cs, ct, k1, k0, gact);
                // branch-free zeroing of rows beyond M / channels beyond the reduction (the loads returned zeros there, so v is finite):
                // as a select the compiler built exec-masked branches around the view arithmetic, and on their skip path its
                // wait-count bookkeeping lost track of the completed loads -- issue() then waited vmcnt(0) on the in-flight LDS-DMA
                const float keep = (kok && ((rowok >> q) & 1u)) ? 1.f : 0.f;
                v.x *= keep; v.y *= keep; v.z *= keep; v.w *= keep;
                st4(buf + alds[q], v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the step's LDS-DMA (issued before the register loads above) has landed
        };

        f32x16 acc[WN];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

        // one step = 4 k-groups x WN column tiles = 4*WN units of four MFMAs into one accumulator; the fragment of the next unit is
        // read while the current unit's MFMAs run (conv3_tile.h)
        auto compute = [&](const float* buf) {
            float4 afr[2], bfr[2];
            afr[0] = ld4(buf + abase + fo[0]);
            bfr[0] = ld4(buf + A_F + bbase + fo[0]);
#pragma unroll
            for (int u = 0; u < 4 * WN; ++u) {
                const int g = u / WN, nt = u - g * WN;
                if (u + 1 < 4 * WN) {
                    const int g1 = (u + 1) / WN, nt1 = (u + 1) - g1 * WN;
                    bfr[(u + 1) & 1] = ld4(buf + A_F + bbase + nt1 * 32 * PWT_KC + fo[g1]);
                    if (nt1 == 0) afr[g1 & 1] = ld4(buf + abase + fo[g1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 a = afr[g & 1], b = bfr[u & 1];
                acc[nt] = mfma32(a.x, b.x, acc[nt]);
                acc[nt] = mfma32(a.y, b.y, acc[nt]);
                acc[nt] = mfma32(a.z, b.z, acc[nt]);
                acc[nt] = mfma32(a.w, b.w, acc[nt]);
            }
        };

        float* cur = smem;
        float* nxt = smem + BUF_F;
        issue(0, cur);
        commit(0, cur);
        __syncthreads();
        for (int s = 0; s < S; ++s) {
            issue(s + 1, nxt);        // unconditional: the step beyond the reduction reads zeros (range-checked) into the idle buffer
            compute(cur);
            commit(s + 1, nxt);
            __syncthreads();
            float* tmp = cur; cur = nxt; nxt = tmp;
        }

        // ---- epilogue: C/D layout col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * hh
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int jl = (wnw * WN + nt) * 32 + li, j = n0 + jl;
            const bool jok = jl < p.ncols && j < p.nout;
            float bs = 0.f, bt = 0.f, bm = 0.f, bi = 0.f;
            if (bne && jok) { bs = p.bn_s[j]; bt = p.bn_t[j]; bm = p.bn_mean[j]; bi = p.bn_istd[j]; }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (jok && m < p.M) {
                    float v = acc[nt][e];
                    if (MODE == 0) {
                        ssum[nt] += v;
                        ssq[nt] = fmaf(v, v, ssq[nt]);
                    } else {
                        if (p.residual) v += p.residual[(long long)m * p.ldr + j];
                        if (p.accumulate) v += p.out[(long long)m * p.ldo + j];
                        if (bne) {
                            const float yv = p.bn_y[(long long)m * p.ldby + j];
                            const float z = fmaf(bs, yv, bt);
                            const float mg = (z > bnlo && z < bnhi) ? v : 0.f;
                            esb[nt] += mg;
                            esg[nt] = fmaf(mg, (yv - bm) * bi, esg[nt]);
                        }
                    }
                    p.out[(long long)m * p.ldo + j] = v;
                }
            }
        }
        // (the idle buffer may still receive the dummy prefetch of step S: the next row tile re-issues step 0 into `cur` only
        // after this barrier, and every wave's loads were waited for in commit())
        __syncthreads();
    }

    // ---- per-block partial rows (fixed order): forward BatchNorm statistics / fused BatchNorm-backward sums
    float* s0 = MODE == 0 ? p.stats : p.bnpart;
    if (s0 != nullptr) {
        float* red = smem;            // [WM][2][BN]: a column is summed over the WM waves that hold rows of it, in wave order
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            float a = MODE == 0 ? ssum[nt] : esb[nt], b = MODE == 0 ? ssq[nt] : esg[nt];
            a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 32, 64);
            if (hh == 0) {
                red[(wm * 2 + 0) * BN + (wnw * WN + nt) * 32 + li] = a;
                red[(wm * 2 + 1) * BN + (wnw * WN + nt) * 32 + li] = b;
            }
        }
        __syncthreads();
        for (int idx = t; idx < 2 * BN; idx += T) {
            const int which = idx / BN, jl = idx - which * BN;
            const int j = n0 + jl;
            if (jl < p.ncols && j < p.nout) {
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < WM; ++wv) v += red[(wv * 2 + which) * BN + jl];
                s0[((long long)blockIdx.y * 2 + which) * p.nout + j] = v;
            }
        }
    }
}
