// Column-marching depthwise 3x3 backward (stride 1, dilation 1) -- included by dwconv.hip after its helper definitions.
//
// Why a third shape: the register-window and LDS-tiled backward kernels carry a float4 channel vector per thread and end up
// at 256 VGPRs (+AGPR spills), i.e. 1-2 waves per SIMD; with so little in flight they sit at 1.4-2.8 TB/s although neither
// HBM (FETCH_SIZE == algorithmic after the XCD remap) nor the L1/TA path is saturated.  Here a thread owns ONE channel of a
// 4-column output strip and marches down the rows of a row chunk with a rolling 3-row window of dy in registers:
//   * every dy / x row is loaded once per thread (horizontal halo 6/4, shared with the neighbouring strip through L1);
//   * ~120 VGPRs -> 4 waves per SIMD; the next row's 18 loads are issued before the current row's arithmetic into the OTHER
//     of two staging register sets (the row loop is unrolled by two), so no s_waitcnt sits between issue and the next step;
//   * lanes run along channels (then strips), so a wave still reads whole contiguous NHWC pixel segments; addresses are
//     one uniform base + a 32-bit byte offset per lane (saddr form: one v_add per load);
//   * the per-channel dW taps -- and, fused, the BatchNorm-backward sums of the layer feeding this one -- stay in
//     registers for the whole march and leave as one partial row per block (deterministic two-level reduction).
// BN fusion: dx is the gradient w.r.t. the ACTIVATED input a = act(s*x + t); the producer's BatchNorm backward needs
// sum(mask*dx) and sum(mask*dx*xhat) over exactly the elements this kernel writes, and x is already in registers.
#pragma once

namespace {

constexpr int MTW = 4;        // output columns per thread
constexpr int MARCH_MAX_THREADS = 512;
constexpr int MWC = MTW + 2;  // window columns

struct MarchGeom {
    int n, h, w, c;
    int rows;      // rows per chunk
    int chunks;    // row chunks per image
    int wstrips;   // ceil(w / MTW)
    int cb;        // channels per block (<= 256)
    int spb;       // strips per block
    int sblocks;   // spatial blocks = n * dil^2 * chunks * ceil(wstrips / spb)
    int sgroups;   // ceil(wstrips / spb)
    int dil;       // dilation (DIL kernels): the conv splits into dil^2 independent dense 3x3 convs over the sub-grids
                   // (row % dil, col % dil) -- taps at +-dil are neighbours at +-1 there; rows / wstrips then refer to the
                   // LARGEST sub-grid (ceil(h/dil) x ceil(w/dil)), smaller ones leave their surplus strips / rows idle
};

__device__ __forceinline__ float ldg_b(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

struct MarchStage {   // raw loads of one step: dy row i+1 (g, y) and x row i
    float g[MWC], y[MWC], x[MWC];
};

// WFULL: w % MTW == 0, the four owned columns always exist (only the two halo columns are conditional)
// ACC: dx += result (fan-out taps) -- a template flag, not a run-time branch around every store
// DIL: atrous conv as dil^2 interleaved dense convs (see MarchGeom::dil); same march, strided addressing
template <bool BNFUSE, bool WFULL, bool ACC, bool DIL = false>
__global__ void __launch_bounds__(MARCH_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) dw_bwd_march_kernel(MarchGeom gm, ViewDev in, const float* __restrict__ wgt, GViewDev dy,
                                                            float* __restrict__ dx, float* __restrict__ dwpart, int /*accumulate*/,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ bnpart) {
    extern __shared__ float mred[];   // [11][blockDim.x]
    const BlockPos bpos = xcd_block_pos();
    const int t = threadIdx.x;
    const int sp = t / gm.cb, cl = t - sp * gm.cb;
    const int ch = bpos.y * gm.cb + cl;
    // spatial block -> (image, row chunk, strip group)
    int sb = bpos.x;
    const int sg = sb % gm.sgroups; sb /= gm.sgroups;
    const int rc = sb % gm.chunks; sb /= gm.chunks;
    // sub-grid (ga, gb) of a dilated conv: its own height / width, origin and strides; the dense conv is the 1x1 case
    const int dil = DIL ? gm.dil : 1;
    const int sub = DIL ? sb % (dil * dil) : 0;
    const int img = DIL ? sb / (dil * dil) : sb;
    const int ga = sub / dil, gb = sub - ga * dil;
    const int hh = DIL ? (gm.h - ga + dil - 1) / dil : gm.h;
    const int ww = DIL ? (gm.w - gb + dil - 1) / dil : gm.w;
    const int ws = sg * gm.spb + sp;
    const bool active = sp < gm.spb && ch < gm.c && ws < gm.wstrips && bpos.x < gm.sblocks &&
                        (!DIL || (ws * MTW < ww && rc * gm.rows < hh));
    const int chs = ch < gm.c ? ch : 0;   // safe channel for clamped addresses

    const bool iaff = in.scale != nullptr, gaff = dy.scale != nullptr;
    if (!gaff) { dy.y = dy.g; dy.act = SSDSEG_ACT_NONE; }   // identity gradient view, branch-free form (see common.h)
    const float ilo = act_lo(in.act), ihi = act_hi(in.act);
    const float glo = act_lo(dy.act), ghi = act_hi(dy.act);
    const float is = iaff ? in.scale[chs] : 1.f, it = iaff ? in.shift[chs] : 0.f;
    const float gs = gaff ? dy.scale[chs] : 1.f, gt = gaff ? dy.shift[chs] : 0.f;
    const float gk1 = gaff ? dy.k1[chs] : 0.f, gk0 = gaff ? dy.k0[chs] : 0.f;
    float mu = 0.f, istd = 0.f;   // xhat = (x - mu) * istd: the subtraction first (exact for x == mu), not x*istd - mu*istd
    if (BNFUSE) { mu = mean[chs]; istd = invstd[chs]; }
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = wgt[k * gm.c + chs];

    float dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = 0.f;
    float bsum = 0.f, bxh = 0.f;

    if (active) {
        const int w0 = ws * MTW;
        const int r0 = rc * gm.rows;
        const int r1 = r0 + gm.rows < hh ? r0 + gm.rows : hh;
        // 32-bit BYTE offsets from the (uniform) tensor bases: the launcher guarantees n*h*w*c*4 < 2^32
        // element (row, col) of the sub-grid at ibase + row*rstride + col*cstride
        const unsigned ibase = ((((unsigned)img * gm.h + ga) * gm.w + gb) * gm.c + ch) * 4u;
        const unsigned rstride = (unsigned)dil * gm.w * gm.c * 4u;
        const unsigned cstride = (unsigned)dil * gm.c * 4u;
        bool cok[MWC];
        unsigned coff[MWC];
#pragma unroll
        for (int a = 0; a < MWC; ++a) {
            const int col = w0 - 1 + a;
            cok[a] = (WFULL && a >= 1 && a <= MTW) ? true : (col >= 0 && col < ww);
            coff[a] = (unsigned)(cok[a] ? col : 0) * cstride;
        }
        // issue the raw loads of one step: dy row `drow` (clamped into the image; masked later) and x row `xrow` (always valid)
        auto issue = [&](int drow, int xrow, MarchStage& s) {
            const int dclamp = drow < 0 ? 0 : (drow >= hh ? hh - 1 : drow);
            const unsigned db = ibase + (unsigned)dclamp * rstride, xb = ibase + (unsigned)xrow * rstride;
#pragma unroll
            for (int a = 0; a < MWC; ++a) {
                s.g[a] = ldg_b(dy.g, db + coff[a]);
                s.y[a] = ldg_b(dy.y, db + coff[a]);
                s.x[a] = ldg_b(in.x, xb + coff[a]);
            }
        };
        auto make_dy = [&](int row, const float (&g)[MWC], const float (&y)[MWC], float (&d)[MWC]) {
            // a row outside the image contributes nothing: fold that into the three per-row coefficients instead of six selects
            const bool rok = row >= 0 && row < hh;
            const float gsr = rok ? gs : 0.f, k1r = rok ? gk1 : 0.f, k0r = rok ? gk0 : 0.f;
#pragma unroll
            for (int a = 0; a < MWC; ++a) {
                const float z = fmaf(gs, y[a], gt);
                const float m = (z > glo && z < ghi) ? gsr : 0.f;
                const float v = fmaf(m, g[a], fmaf(k1r, y[a], k0r));
                d[a] = (WFULL && a >= 1 && a <= MTW) ? v : (cok[a] ? v : 0.f);
            }
        };
        float wa[MWC], wb[MWC], wc[MWC];   // the rolling dy window; which array is row i-1 / i / i+1 rotates with the step
        {
            MarchStage p;
            issue(r0 - 1, r0, p);
            make_dy(r0 - 1, p.g, p.y, wa);
            issue(r0, r0, p);
            make_dy(r0, p.g, p.y, wb);
        }
        // one row: consume `cur` (dy row i+1, x row i), refill `nxt` for row i+1.  dm / d0 hold dy rows i-1 / i, dp receives i+1.
        auto step = [&](int i, MarchStage& cur, MarchStage& nxt, const float (&dm)[MWC], const float (&d0)[MWC], float (&dp)[MWC]) {
            float xa[MWC];
            make_dy(i + 1, cur.g, cur.y, dp);
#pragma unroll
            for (int a = 0; a < MWC; ++a) {
                const float z = fminf(fmaxf(fmaf(is, cur.x[a], it), ilo), ihi);
                xa[a] = (WFULL && a >= 1 && a <= MTW) ? z : (cok[a] ? z : 0.f);
            }
            // next step's loads go out before this step's arithmetic.  Unconditional (the last step re-reads row r1-1 into
            // registers nobody consumes): a branch here makes the compiler's s_waitcnt placement merge the "no new loads"
            // path and wait for the new loads immediately (vmcnt(1) instead of vmcnt(18)), i.e. no overlap at all
            issue(i + 2, i + 1 < r1 ? i + 1 : r1 - 1, nxt);
            // ---- dx[i][w0 + j] = sum_{kh,kw} dy[i - kh + 1][w0 + j - kw + 1] * w[kh][kw]
            const unsigned ob = ibase + (unsigned)i * rstride;
#pragma unroll
            for (int j = 0; j < MTW; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    acc = fmaf(dp[j - kw + 2], wk[0 * 3 + kw], acc);
                    acc = fmaf(d0[j - kw + 2], wk[1 * 3 + kw], acc);
                    acc = fmaf(dm[j - kw + 2], wk[2 * 3 + kw], acc);
                }
                if (WFULL || cok[j + 1]) {
                    if (dx != nullptr) {
                        float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(dx) + (ob + coff[j + 1]));
                        if (ACC) acc += *p;
                        *p = acc;
                    }
                    if (BNFUSE) {
                        const float mg = (xa[j + 1] > ilo && xa[j + 1] < ihi) ? acc : 0.f;
                        bsum += mg;
                        bxh = fmaf(mg, (cur.x[j + 1] - mu) * istd, bxh);
                    }
                }
            }
            // ---- dW[kh][kw] += sum_j a[i][w0 + j + kw - 1] * dy[i - kh + 1][w0 + j]
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                for (int j = 0; j < MTW; ++j) {
                    dwacc[0 * 3 + kw] = fmaf(xa[j + kw], dp[j + 1], dwacc[0 * 3 + kw]);
                    dwacc[1 * 3 + kw] = fmaf(xa[j + kw], d0[j + 1], dwacc[1 * 3 + kw]);
                    dwacc[2 * 3 + kw] = fmaf(xa[j + kw], dm[j + 1], dwacc[2 * 3 + kw]);
                }
            }
        };
        MarchStage sa, sb2;
        issue(r0 + 1, r0, sa);
        int i = r0;
        // two rows per trip (the two staging sets are back in their roles at the back edge); the three-array window is rotated
        // by copying -- unrolling six rows to make that free as well costs 90 more VGPRs (214), i.e. half the occupancy
        for (; i + 1 < r1; i += 2) {
            step(i, sa, sb2, wa, wb, wc);
            step(i + 1, sb2, sa, wb, wc, wa);
#pragma unroll
            for (int a = 0; a < MWC; ++a) { const float t0 = wa[a]; wa[a] = wc[a]; wc[a] = wb[a]; wb[a] = t0; }
        }
        if (i < r1) step(i, sa, sb2, wa, wb, wc);
    }

    // ---- block partials: sum over the block's strips (fixed order), one row per spatial block
    const int nth = blockDim.x;
#pragma unroll
    for (int k = 0; k < 9; ++k) mred[k * nth + t] = dwacc[k];
    if (BNFUSE) { mred[9 * nth + t] = bsum; mred[10 * nth + t] = bxh; }
    __syncthreads();
    if (sp == 0 && ch < gm.c) {
        const int nv = BNFUSE ? 11 : 9;
        for (int k = 0; k < nv; ++k) {
            float s = 0.f;
            for (int q = 0; q < gm.spb; ++q) s += mred[k * nth + q * gm.cb + cl];
            if (k < 9) dwpart[((long long)bpos.x * 9 + k) * gm.c + ch] = s;
            else bnpart[((long long)bpos.x * 2 + (k - 9)) * gm.c + ch] = s;
        }
    }
}

struct MarchLaunch {
    dim3 grid, block;
    size_t lds;
};

// Block shape of the marching kernels: `cb` channels x `spb` ADJACENT column strips.  Strips of one block march in step, so
// the horizontal halo between them is served by L1; between blocks it is not (neighbouring blocks drift apart by more rows
// than L2 holds: PMC FETCH_SIZE showed the full 6/4 halo for one-strip blocks, 1.05x for the eight-strip blocks of c = 32).
// So: channel chunks of 64 (one wave = one strip's 256-byte pixel segment, line-aligned because c % 64 == 0), as many
// strips as fit in MARCH_MAX_THREADS, the count chosen to waste the fewest strip slots in the last group of a row.
inline long long march_min_waves() {   // row chunking stops once the launch has this many waves (A/B: SSDSEG_MARCH_WAVES)
    static const long long v = getenv("SSDSEG_MARCH_WAVES") ? atoll(getenv("SSDSEG_MARCH_WAVES")) : 4096;
    return v;
}
inline void march_split(int c, int wstrips, int* cb_out, int* spb_out, int* cchunks_out) {
    static const int env_cb = getenv("SSDSEG_MARCH_CB") ? atoi(getenv("SSDSEG_MARCH_CB")) : 0;        // A/B switches
    static const int env_mt = getenv("SSDSEG_MARCH_MAXT") ? atoi(getenv("SSDSEG_MARCH_MAXT")) : 0;
    const int maxt = (env_mt >= 64 && env_mt <= MARCH_MAX_THREADS) ? env_mt : 256;   // 512-thread blocks measured 3-5 % slower
    int unit = env_cb > 0 ? env_cb : 64;
    // (c = 144 -- blocks 2 / 3 at 120 x 160 -- runs as ONE 144-channel chunk: one strip of 2.25 waves per block.  Round 3 tried three
    // 48-channel chunks x five strips (full waves, shared halos): block-2 backward 326 -> 420 us, forward 190 -> 216 us -- 192-byte
    // channel segments are not line-aligned and every pixel row is then walked by three blocks.  Not kept.)
    int cb, cchunks;
    if (env_cb >= 0 && c > unit && c % unit == 0) { cb = unit; cchunks = c / unit; }
    else { cchunks = cdiv(c, 256); cb = cdiv(c, cchunks); }
    int maxspb = maxt / cb;
    if (maxspb < 1) maxspb = 1;
    if (maxspb > wstrips) maxspb = wstrips;
    int best = 1, bestwaste = 1 << 30;
    for (int sp = maxspb; sp >= 1; --sp) {
        const int waste = cdiv(wstrips, sp) * sp - wstrips;
        // fewer strips per block only if it saves strip slots AND keeps at least half of the halo sharing
        if (waste < bestwaste && (sp * 2 > maxspb || bestwaste == (1 << 30))) { best = sp; bestwaste = waste; }
    }
    *cb_out = cb; *spb_out = best; *cchunks_out = cchunks;
}

// geometry for an n x h x w x c tensor: channel chunks of <= 256, as many strips per block as fit in 256 threads, row chunks
// sized so that the launch has >= ~4096 waves (16 per CU) where the layer is big enough
inline MarchLaunch march_geometry(int n, int h, int w, int c, MarchGeom* g, int dil = 1) {
    g->n = n; g->h = h; g->w = w; g->c = c; g->dil = dil;
    const int subs = dil * dil;
    h = cdiv(h, dil); w = cdiv(w, dil);   // the largest sub-grid
    n *= subs;
    g->wstrips = cdiv(w, MTW);
    int cchunks;
    march_split(c, g->wstrips, &g->cb, &g->spb, &cchunks);
    g->sgroups = cdiv(g->wstrips, g->spb);
    const int threads = ((g->cb * g->spb + 63) / 64) * 64;
    const long long waves_per_chunkrow = (long long)n * g->sgroups * cchunks * (threads / 64);
    int rows = h;
    while (rows > 8 && waves_per_chunkrow * cdiv(h, rows) < march_min_waves()) rows = (rows + 1) / 2;
    g->rows = rows;
    g->chunks = cdiv(h, rows);
    g->sblocks = n * g->chunks * g->sgroups;
    MarchLaunch l;
    l.grid = dim3((unsigned)((g->sblocks + 7) & ~7), cchunks, 1);   // multiple of 8 for the XCD remap
    l.block = dim3(threads, 1, 1);
    l.lds = (size_t)11 * threads * sizeof(float);
    return l;
}

// ------------------------------------------------------------------------------------------------ stride 2
// Same march over OUTPUT rows r: a thread owns one channel of 2 output columns (wo0, wo0+1) = 4 input columns
// wi0 .. wi0+3 with wi0 = 2*wo0 - PL, and finalises the two input rows 2r - PT, 2r - PT + 1 per step:
//   dx[2r-PT  ][wi0+j] = sum_{kh in {0,2}} sum_kw dy[r - kh/2][..] w[kh][kw]      (taps of matching parity only)
//   dx[2r-PT+1][wi0+j] =                   sum_kw dy[r       ][..] w[1 ][kw]
//   dW[0|1][kw] += a[2r-PT+(0|1)][wi0+2q+kw] * dy[r][wo0+q],   dW[2][kw] += a[2r-PT][wi0+2q+kw] * dy[r-1][wo0+q]
// (the kh = 2 tap of output row r-1 reads input row 2r-PT, i.e. the first row of THIS step: dy row r-1 is kept).
template <bool ACC>
struct March2Stage {   // raw loads of one step: dy row r (3 columns of g, y) and x rows 2r-PT, 2r-PT+1 (5 columns each)
    float g[3], y[3], x0[5], x1[5];
    float o0[ACC ? 4 : 1], o1[ACC ? 4 : 1];   // ACC: previous contents of the 2 x 4 dx elements this step finalises
};

struct March2Geom {
    int n, h, w, c, ho, wo;
    int rows, chunks, wstrips, cb, spb, sblocks, sgroups;   // as MarchGeom, over the OUTPUT rows / 2-column output strips
    int dil;                                                // forward DIL kernels: see MarchGeom::dil
    int depth2;                                             // forward: loads two rows ahead (three stage sets) instead of one
};

// ACC: dx += result.  The old values are fetched WITH the step's other loads (one step ahead); read inline at the store they
// put a load -> add -> store round trip on every element (2.6 TB/s where the overwriting variant runs 5.4)
template <bool BNFUSE, int PT, int PL, bool ACC = false>
__global__ void __launch_bounds__(MARCH_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) dw_bwd_march2_kernel(March2Geom gm, ViewDev in, const float* __restrict__ wgt, GViewDev dy,
                                                             float* __restrict__ dx, float* __restrict__ dwpart, int /*accumulate*/,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             float* __restrict__ bnpart) {
    extern __shared__ float mred[];   // [11][blockDim.x]
    const BlockPos bpos = xcd_block_pos();
    const int t = threadIdx.x;
    const int sp = t / gm.cb, cl = t - sp * gm.cb;
    const int ch = bpos.y * gm.cb + cl;
    int sb = bpos.x;
    const int sg = sb % gm.sgroups; sb /= gm.sgroups;
    const int rc = sb % gm.chunks;
    const int img = sb / gm.chunks;
    const int ws = sg * gm.spb + sp;
    const bool active = sp < gm.spb && ch < gm.c && ws < gm.wstrips && bpos.x < gm.sblocks;
    const int chs = ch < gm.c ? ch : 0;

    const bool iaff = in.scale != nullptr, gaff = dy.scale != nullptr;
    if (!gaff) { dy.y = dy.g; dy.act = SSDSEG_ACT_NONE; }
    const float ilo = act_lo(in.act), ihi = act_hi(in.act);
    const float glo = act_lo(dy.act), ghi = act_hi(dy.act);
    const float is = iaff ? in.scale[chs] : 1.f, it = iaff ? in.shift[chs] : 0.f;
    const float gs = gaff ? dy.scale[chs] : 1.f, gt = gaff ? dy.shift[chs] : 0.f;
    const float gk1 = gaff ? dy.k1[chs] : 0.f, gk0 = gaff ? dy.k0[chs] : 0.f;
    float mu = 0.f, istd = 0.f;
    if (BNFUSE) { mu = mean[chs]; istd = invstd[chs]; }
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = wgt[k * gm.c + chs];
    float dwacc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dwacc[k] = 0.f;
    float bsum = 0.f, bxh = 0.f;

    if (active) {
        const int wo0 = ws * 2, wi0 = 2 * wo0 - PL;
        const int r0 = rc * gm.rows;
        const int r1 = r0 + gm.rows < gm.ho ? r0 + gm.rows : gm.ho;
        // 32-bit byte offsets (launcher: n*h*w*c*4 < 2^32)
        const unsigned ibase = (((unsigned)img * gm.h * gm.w) * gm.c + ch) * 4u;     // input-resolution tensors (x, dx)
        const unsigned obase = (((unsigned)img * gm.ho * gm.wo) * gm.c + ch) * 4u;   // output-resolution tensors (g, y)
        const unsigned irow = (unsigned)gm.w * gm.c * 4u, orow = (unsigned)gm.wo * gm.c * 4u;
        bool dok[3], xok[5];
        unsigned doff[3], xoff[5];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int col = wo0 - 1 + a;
            dok[a] = col >= 0 && col < gm.wo;
            doff[a] = (unsigned)(dok[a] ? col : 0) * gm.c * 4u;
        }
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int col = wi0 + b;
            xok[b] = col >= 0 && col < gm.w;
            xoff[b] = (unsigned)(xok[b] ? col : 0) * gm.c * 4u;
        }
        auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
        auto issue = [&](int r, March2Stage<ACC>& s) {   // rows clamped into the tensors; masked when consumed
            const unsigned db = obase + (unsigned)clampi(r, gm.ho - 1) * orow;
            const unsigned xb0 = ibase + (unsigned)clampi(2 * r - PT, gm.h - 1) * irow;
            const unsigned xb1 = ibase + (unsigned)clampi(2 * r - PT + 1, gm.h - 1) * irow;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s.g[a] = ldg_b(dy.g, db + doff[a]);
                s.y[a] = ldg_b(dy.y, db + doff[a]);
            }
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                s.x0[b] = ldg_b(in.x, xb0 + xoff[b]);
                s.x1[b] = ldg_b(in.x, xb1 + xoff[b]);
            }
            if (ACC) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s.o0[j] = ldg_b(dx, xb0 + xoff[j]);
                    s.o1[j] = ldg_b(dx, xb1 + xoff[j]);
                }
            }
        };
        auto make_dy = [&](int row, const float (&g)[3], const float (&y)[3], float (&d)[3]) {
            const bool rok = row >= 0 && row < gm.ho;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float z = fmaf(gs, y[a], gt);
                const float m = (z > glo && z < ghi) ? gs : 0.f;
                const float v = fmaf(m, g[a], fmaf(gk1, y[a], gk0));
                d[a] = (rok && dok[a]) ? v : 0.f;
            }
        };
        float dprev[3], dcur[3];
        {
            March2Stage<ACC> p;
            issue(r0 - 1, p);
            make_dy(r0 - 1, p.g, p.y, dprev);
        }
        auto step = [&](int r, March2Stage<ACC>& cur, March2Stage<ACC>& nxt) {
            make_dy(r, cur.g, cur.y, dcur);
            const int hi0 = 2 * r - PT, hi1 = hi0 + 1;
            const bool rok0 = hi0 >= 0 && hi0 < gm.h, rok1 = hi1 < gm.h;   // hi1 >= 0 always
            float xa0[5], xa1[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const float z0 = fminf(fmaxf(fmaf(is, cur.x0[b], it), ilo), ihi);
                const float z1 = fminf(fmaxf(fmaf(is, cur.x1[b], it), ilo), ihi);
                xa0[b] = (rok0 && xok[b]) ? z0 : 0.f;
                xa1[b] = (rok1 && xok[b]) ? z1 : 0.f;
            }
            issue(r + 1, nxt);   // unconditional: see the stride-1 kernel
            // ---- dx rows hi0 (kh in {0, 2}) and hi1 (kh = 1); column j: j even -> kw in {0, 2}, j odd -> kw = 1
            float acc0[4], acc1[4];
            // window index of dy column wo: wo - (wo0 - 1)
            acc0[0] = fmaf(dcur[1], wk[0], fmaf(dcur[0], wk[2], fmaf(dprev[1], wk[6], dprev[0] * wk[8])));
            acc0[1] = fmaf(dcur[1], wk[1], dprev[1] * wk[7]);
            acc0[2] = fmaf(dcur[2], wk[0], fmaf(dcur[1], wk[2], fmaf(dprev[2], wk[6], dprev[1] * wk[8])));
            acc0[3] = fmaf(dcur[2], wk[1], dprev[2] * wk[7]);
            acc1[0] = fmaf(dcur[1], wk[3], dcur[0] * wk[5]);
            acc1[1] = dcur[1] * wk[4];
            acc1[2] = fmaf(dcur[2], wk[3], dcur[1] * wk[5]);
            acc1[3] = dcur[2] * wk[4];
            const unsigned ob0 = ibase + (unsigned)(rok0 ? hi0 : 0) * irow, ob1 = ibase + (unsigned)(rok1 ? hi1 : 0) * irow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (rok0 && xok[j]) {
                    if (dx != nullptr) {
                        float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(dx) + (ob0 + xoff[j]));
                        if (ACC) acc0[j] += cur.o0[j];
                        *p = acc0[j];
                    }
                    if (BNFUSE) {
                        const float mg = (xa0[j] > ilo && xa0[j] < ihi) ? acc0[j] : 0.f;
                        bsum += mg;
                        bxh = fmaf(mg, (cur.x0[j] - mu) * istd, bxh);
                    }
                }
                if (rok1 && xok[j]) {
                    if (dx != nullptr) {
                        float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(dx) + (ob1 + xoff[j]));
                        if (ACC) acc1[j] += cur.o1[j];
                        *p = acc1[j];
                    }
                    if (BNFUSE) {
                        const float mg = (xa1[j] > ilo && xa1[j] < ihi) ? acc1[j] : 0.f;
                        bsum += mg;
                        bxh = fmaf(mg, (cur.x1[j] - mu) * istd, bxh);
                    }
                }
            }
            // ---- dW over the two owned outputs q (dy window index q + 1)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    dwacc[0 * 3 + kw] = fmaf(xa0[2 * q + kw], dcur[q + 1], dwacc[0 * 3 + kw]);
                    dwacc[1 * 3 + kw] = fmaf(xa1[2 * q + kw], dcur[q + 1], dwacc[1 * 3 + kw]);
                    dwacc[2 * 3 + kw] = fmaf(xa0[2 * q + kw], dprev[q + 1], dwacc[2 * 3 + kw]);
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) dprev[a] = dcur[a];
        };
        March2Stage<ACC> sa, sb2;
        issue(r0, sa);
        int r = r0;
        for (; r + 1 < r1; r += 2) {
            step(r, sa, sb2);
            step(r + 1, sb2, sa);
        }
        if (r < r1) step(r, sa, sb2);
    }

    const int nth = blockDim.x;
#pragma unroll
    for (int k = 0; k < 9; ++k) mred[k * nth + t] = dwacc[k];
    if (BNFUSE) { mred[9 * nth + t] = bsum; mred[10 * nth + t] = bxh; }
    __syncthreads();
    if (sp == 0 && ch < gm.c) {
        const int nv = BNFUSE ? 11 : 9;
        for (int k = 0; k < nv; ++k) {
            float s = 0.f;
            for (int q = 0; q < gm.spb; ++q) s += mred[k * nth + q * gm.cb + cl];
            if (k < 9) dwpart[((long long)bpos.x * 9 + k) * gm.c + ch] = s;
            else bnpart[((long long)bpos.x * 2 + (k - 9)) * gm.c + ch] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
// The same march for the forward conv: S = 1: 4 output columns per thread, rolling 3-row window of the activated input;
// S = 2: 2 output columns (5 input columns), two new input rows per output row, the third is next step's first.
// BN statistics (sum, sumsq of the raw output) stay in registers and leave as one partial row per block.
// `gm` is a March2Geom in both cases (h, w input; ho, wo output; strips over output columns).
// DIL (S == 1, PT == PL == 1): atrous conv as dil^2 interleaved dense convs over the sub-grids, as in the backward kernel
template <int S, int PT, int PL, bool DIL = false, bool D2 = false>
__global__ void __launch_bounds__(MARCH_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) dw_fwd_march_kernel(March2Geom gm, ViewDev in, const float* __restrict__ wgt, float* __restrict__ y,
                                                            float* __restrict__ stats) {
    constexpr int OC = S == 1 ? 4 : 2;        // output columns per thread
    constexpr int IC = (OC - 1) * S + 3;      // input columns per thread (6 | 5)
    constexpr int NR = S;                     // new input rows per step
    extern __shared__ float mred[];           // [2][blockDim.x]
    const BlockPos bpos = xcd_block_pos();
    const int t = threadIdx.x;
    const int sp = t / gm.cb, cl = t - sp * gm.cb;
    const int ch = bpos.y * gm.cb + cl;
    static_assert(!DIL || (S == 1 && PT == 1 && PL == 1), "dilated march: stride 1, SAME");
    int sb = bpos.x;
    const int sg = sb % gm.sgroups; sb /= gm.sgroups;
    const int rc = sb % gm.chunks; sb /= gm.chunks;
    const int dil = DIL ? gm.dil : 1;
    const int sub = DIL ? sb % (dil * dil) : 0;
    const int img = DIL ? sb / (dil * dil) : sb;
    const int ga = sub / dil, gb = sub - ga * dil;
    // height / width of this sub-grid (input == output for the dilated case); the dense conv is the 1x1 grid
    const int hh = DIL ? (gm.h - ga + dil - 1) / dil : gm.h, ww = DIL ? (gm.w - gb + dil - 1) / dil : gm.w;
    const int hho = DIL ? hh : gm.ho, wwo = DIL ? ww : gm.wo;
    const int ws = sg * gm.spb + sp;
    const bool active = sp < gm.spb && ch < gm.c && ws < gm.wstrips && bpos.x < gm.sblocks &&
                        (!DIL || (ws * OC < wwo && rc * gm.rows < hho));
    const int chs = ch < gm.c ? ch : 0;
    const bool iaff = in.scale != nullptr;
    const float ilo = act_lo(in.act), ihi = act_hi(in.act);
    const float is = iaff ? in.scale[chs] : 1.f, it = iaff ? in.shift[chs] : 0.f;
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = wgt[k * gm.c + chs];
    float ssum = 0.f, ssq = 0.f;

    if (active) {
        const int wo0 = ws * OC, wi0 = wo0 * S - PL;
        const int r0 = rc * gm.rows;
        const int r1 = r0 + gm.rows < hho ? r0 + gm.rows : hho;
        const unsigned ibase = ((((unsigned)img * gm.h + ga) * gm.w + gb) * gm.c + ch) * 4u;
        const unsigned obase = ((((unsigned)img * gm.ho + ga) * gm.wo + gb) * gm.c + ch) * 4u;
        const unsigned irow = (unsigned)dil * gm.w * gm.c * 4u, orow = (unsigned)dil * gm.wo * gm.c * 4u, ocol = (unsigned)dil * gm.c * 4u;
        bool xok[IC];
        unsigned xoff[IC];
#pragma unroll
        for (int b = 0; b < IC; ++b) {
            const int col = wi0 + b;
            xok[b] = col >= 0 && col < ww;
            xoff[b] = (unsigned)(xok[b] ? col : 0) * ocol;   // (input column stride == output column stride: dil * c floats)
        }
        struct Stage { float x[NR][IC]; };
        // the NR new input rows of output row r: S=1: row r + 1 - PT (= r+1-1 .. window rows r-PT, r-PT+1, r-PT+2);
        // S=2: rows 2r - PT + 1, 2r - PT + 2  (row 2r - PT is the previous step's last row)
        auto first_new = [&](int r) { return S == 1 ? r + 2 - PT : 2 * r - PT + 1; };
        auto issue = [&](int r, Stage& s) {
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                int row = first_new(r) + q;
                row = row < 0 ? 0 : (row > hh - 1 ? hh - 1 : row);
                const unsigned rb = ibase + (unsigned)row * irow;
#pragma unroll
                for (int b = 0; b < IC; ++b) s.x[q][b] = ldg_b(in.x, rb + xoff[b]);
            }
        };
        auto activate = [&](int row, const float (&raw)[IC], float (&a)[IC]) {
            const bool rok = row >= 0 && row < hh;
#pragma unroll
            for (int b = 0; b < IC; ++b) {
                const float z = fminf(fmaxf(fmaf(is, raw[b], it), ilo), ihi);
                a[b] = (rok && xok[b]) ? z : 0.f;
            }
        };
        // window rows: S=1: (r-PT, r-PT+1) carried, r-PT+2 new; S=2: (2r-PT) carried, 2r-PT+1, 2r-PT+2 new
        float wa[3][IC];
        {
            Stage p;
            // prologue: the carried rows of the first step, fetched through the same clamped loader
            if (S == 1) {
                // rows r0-PT and r0-PT+1 are "new rows" of the virtual steps r0-2 and r0-1
                issue(r0 - 2, p); activate(r0 - PT, p.x[0], wa[0]);
                issue(r0 - 1, p); activate(r0 - PT + 1, p.x[0], wa[1]);
            } else {
                issue(r0 - 1, p); activate(2 * r0 - PT, p.x[NR - 1], wa[0]);   // second new row of step r0-1 == 2*r0 - PT
            }
        }
        auto step = [&](int r, Stage& cur, Stage& nxt) {
            if (S == 1) {
                activate(r + 2 - PT, cur.x[0], wa[2]);
            } else {
                activate(2 * r - PT + 1, cur.x[0], wa[1]);
                activate(2 * r - PT + 2, cur.x[NR - 1], wa[2]);
            }
            issue(r + 1, nxt);   // unconditional (see the backward kernel)
            const unsigned ob = obase + (unsigned)r * orow + (unsigned)wo0 * ocol;
#pragma unroll
            for (int j = 0; j < OC; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) acc = fmaf(wa[kh][j * S + kw], wk[kh * 3 + kw], acc);
                if (wo0 + j < wwo) {
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(y) + (ob + (unsigned)j * ocol)) = acc;
                    ssum += acc;
                    ssq = fmaf(acc, acc, ssq);
                }
            }
            if (S == 1) {
#pragma unroll
                for (int b = 0; b < IC; ++b) { wa[0][b] = wa[1][b]; wa[1][b] = wa[2][b]; }
            } else {
#pragma unroll
                for (int b = 0; b < IC; ++b) wa[0][b] = wa[2][b];
            }
        };
        // loads run TWO rows ahead of the arithmetic (three stage sets in rotation): with one row ahead a wave had 1.5 KB in flight,
        // 24-48 KB per CU -- at the edge of what 5 TB/s x ~2 us of loaded HBM latency asks for (SSDSEG_DW_FWD_DEPTH=1: the old depth)
        if (D2) {
            auto step2 = [&](int r, Stage& cur, Stage& far) {      // `far` receives row r + 2
                if (S == 1) {
                    activate(r + 2 - PT, cur.x[0], wa[2]);
                } else {
                    activate(2 * r - PT + 1, cur.x[0], wa[1]);
                    activate(2 * r - PT + 2, cur.x[NR - 1], wa[2]);
                }
                issue(r + 3, far);
                const unsigned ob = obase + (unsigned)r * orow + (unsigned)wo0 * ocol;
#pragma unroll
                for (int j = 0; j < OC; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc = fmaf(wa[kh][j * S + kw], wk[kh * 3 + kw], acc);
                    if (wo0 + j < wwo) {
                        *reinterpret_cast<float*>(reinterpret_cast<char*>(y) + (ob + (unsigned)j * ocol)) = acc;
                        ssum += acc;
                        ssq = fmaf(acc, acc, ssq);
                    }
                }
                if (S == 1) {
#pragma unroll
                    for (int b = 0; b < IC; ++b) { wa[0][b] = wa[1][b]; wa[1][b] = wa[2][b]; }
                } else {
#pragma unroll
                    for (int b = 0; b < IC; ++b) wa[0][b] = wa[2][b];
                }
            };
            Stage s0, s1, s2;
            issue(r0, s0);
            issue(r0 + 1, s1);
            issue(r0 + 2, s2);
            int r = r0;
            for (; r + 2 < r1; r += 3) {
                step2(r, s0, s0);          // consumes s0 (row r), refills it with row r + 3
                step2(r + 1, s1, s1);
                step2(r + 2, s2, s2);
            }
            if (r < r1) step2(r, s0, s0);
            if (r + 1 < r1) step2(r + 1, s1, s1);
        } else {
            Stage sa, sb2;
            issue(r0, sa);
            int r = r0;
            for (; r + 1 < r1; r += 2) {
                step(r, sa, sb2);
                step(r + 1, sb2, sa);
            }
            if (r < r1) step(r, sa, sb2);
        }
    }

    if (stats != nullptr) {
        const int nth = blockDim.x;
        mred[t] = ssum;
        mred[nth + t] = ssq;
        __syncthreads();
        if (sp == 0 && ch < gm.c) {
            float s = 0.f, q = 0.f;
            for (int k = 0; k < gm.spb; ++k) { s += mred[k * gm.cb + cl]; q += mred[nth + k * gm.cb + cl]; }
            stats[((long long)bpos.x * 2 + 0) * gm.c + ch] = s;
            stats[((long long)bpos.x * 2 + 1) * gm.c + ch] = q;
        }
    }
}

// forward geometry: strips of OC output columns
inline MarchLaunch march_fwd_geometry(int n, int h, int w, int c, int ho, int wo, int stride, March2Geom* g, int dil = 1) {
    g->n = n; g->h = h; g->w = w; g->c = c; g->ho = ho; g->wo = wo; g->dil = dil;
    n *= dil * dil;                            // (dil > 1: stride 1, ho == h, wo == w; the largest sub-grid sets the geometry)
    ho = cdiv(ho, dil); wo = cdiv(wo, dil);
    g->wstrips = cdiv(wo, stride == 1 ? 4 : 2);
    int cchunks;
    march_split(c, g->wstrips, &g->cb, &g->spb, &cchunks);
    g->sgroups = cdiv(g->wstrips, g->spb);
    const int threads = ((g->cb * g->spb + 63) / 64) * 64;
    const long long waves_per_chunkrow = (long long)n * g->sgroups * cchunks * (threads / 64);
    int rows = ho;
    while (rows > 8 && waves_per_chunkrow * cdiv(ho, rows) < 4096) rows = (rows + 1) / 2;
    g->rows = rows;
    g->chunks = cdiv(ho, rows);
    g->sblocks = n * g->chunks * g->sgroups;
    MarchLaunch l;
    l.grid = dim3((unsigned)((g->sblocks + 7) & ~7), cchunks, 1);
    l.block = dim3(threads, 1, 1);
    l.lds = (size_t)2 * threads * sizeof(float);
    return l;
}

inline MarchLaunch march2_geometry(int n, int h, int w, int c, int ho, int wo, March2Geom* g) {
    g->n = n; g->h = h; g->w = w; g->c = c; g->ho = ho; g->wo = wo; g->dil = 1;
    g->wstrips = cdiv(wo, 2);
    int cchunks;
    march_split(c, g->wstrips, &g->cb, &g->spb, &cchunks);
    g->sgroups = cdiv(g->wstrips, g->spb);
    const int threads = ((g->cb * g->spb + 63) / 64) * 64;
    const long long waves_per_chunkrow = (long long)n * g->sgroups * cchunks * (threads / 64);
    int rows = ho;
    while (rows > 8 && waves_per_chunkrow * cdiv(ho, rows) < 4096) rows = (rows + 1) / 2;
    g->rows = rows;
    g->chunks = cdiv(ho, rows);
    g->sblocks = n * g->chunks * g->sgroups;
    MarchLaunch l;
    l.grid = dim3((unsigned)((g->sblocks + 7) & ~7), cchunks, 1);
    l.block = dim3(threads, 1, 1);
    l.lds = (size_t)11 * threads * sizeof(float);
    return l;
}

}  // namespace
