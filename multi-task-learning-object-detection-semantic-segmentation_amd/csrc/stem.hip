// Stem convolution forward (reference models.py:187-206 MobileNetV2: Rescaling + Conv2D 3x3 stride 2, 3 -> 32, no bias, BatchNorm
// statistics; models.py:622,628 ShuffleNetV2: 3 -> 24 with bias), DIRECT form for <= 64 output channels.
//
// The layer reads 12 bytes and writes 4 * cout bytes per output pixel with 27 multiply-adds per output value: a streaming kernel.  As an
// implicit GEMM (gemm.hip, LD = 2: K = 27 padded to 32, 128-row tiles, LDS staging, transposing epilogue) it ran 2.2 TB/s of
// algorithmic traffic at 480 x 640 x 32 (0.27 ms).  Here a thread is (output pixel, four output channels): the cout / 4 lanes of a pixel
// sit next to each other (in whole groups of four lanes, see QP), so a wave's store is one contiguous run of 64 x 16 bytes, the nine 12-byte input reads of a pixel are the
// same addresses for all its lanes (one request), and the 27 x 4 weights of the thread live in registers for the whole kernel.  The
// input rescale (x * s + o, zero padding AFTER it) is folded into the weights: sum_valid (x s + o) w = sum_valid x (s w) + sum_valid o w.
// A block walks whole output rows (no per-pixel division); BatchNorm partial sums: one row per block, fixed order.
#include "common.h"

namespace {


struct StemArgs {
    const float* x;      // [n][h][w][3]
    const float* w;      // [3][3][3][cout]
    const float* bias;   // [cout] or nullptr
    float* y;            // [n][ho][wo][cout]
    float* stats;        // [blocks][2][cout] or nullptr
    int n, h, w_, ho, wo, pt, pl, cout;
    float scale, offset;
    unsigned x_bytes;
};

template <int Q>   // channel quads per pixel: cout == 4 * Q
__global__ void __launch_bounds__(256) stem_fwd_direct_kernel(StemArgs p) {
    // lanes per pixel: Q rounded up to whole groups of four lanes.  The nine 12-byte reads of a pixel are the same addresses for all its
    // lanes; with 6 lanes per pixel (24 channels, ShuffleNetV2) every other aligned group of four lanes holds two different addresses and
    // the kernel ran 243 us against 170 for 32 channels (88 for 16); with 8 lanes per pixel, two of them idle, it takes the 32-channel
    // time (scripts/dbg/stem_time.py, profiles/r03_stem_lanes_per_pixel.txt -- also what did NOT help: nine wide loads per pixel,
    // rows staged in LDS, more pixels in flight)
    constexpr int QP = (Q + 3) / 4 * 4;
    constexpr int PPB = 256 / QP;                     // pixels per block iteration
    __shared__ float red[2][PPB][4 * Q];
    __shared__ float wo_s[9][4 * Q];                  // offset * sum_c w[tap][c][.]: what a tap adds through the rescale's offset
    const int t = threadIdx.x;
    const int pix = t / QP;
    const bool lane_on = pix < PPB && t % QP < Q;
    const int q = t % QP < Q ? t % QP : Q - 1;        // (idle lanes compute on the last quad's weights, store nothing)
    // weights of this thread's four channels with the input scale folded in: ws[tap][c] = scale * w
    float4 ws[27];
    float4 wo_all = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(p.w + (size_t)(tap * 3 + c) * p.cout + 4 * q);
            ws[tap * 3 + c] = make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s = make_float4(s.x * p.offset, s.y * p.offset, s.z * p.offset, s.w * p.offset);
        if (pix == 0 && lane_on) *reinterpret_cast<float4*>(&wo_s[tap][4 * q]) = s;
        wo_all.x += s.x; wo_all.y += s.y; wo_all.z += s.z; wo_all.w += s.w;
    }
    __syncthreads();
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias != nullptr) b4 = *reinterpret_cast<const float4*>(p.bias + 4 * q);
    float4 su = make_float4(0.f, 0.f, 0.f, 0.f), sq = su;
    const int rows = p.n * p.ho;
    // the image through a raw buffer descriptor: a tap outside the image takes an out-of-range offset and reads 0 -- no branch around any
    // load (with a branch per tap they went out tap by tap: nine memory round trips per pixel).  Dword loads: this hipcc lowers
    // __builtin_amdgcn_raw_buffer_load_b96 / _b64 to ONE dword load, and 12-byte global loads from clamped coordinates measured
    // slower (64-bit addresses).  TWO pixels per thread and trip: 54 loads in flight at 2 waves per SIMD (the 108 weight registers).
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int img = row / p.ho, oy = row - img * p.ho;
        const int iy0 = 2 * oy - p.pt;
        const unsigned img_off = (unsigned)((size_t)img * p.h * p.w_ * 12);
        float* yrow = p.y + (size_t)row * p.wo * p.cout;
        for (int ox0 = pix; ox0 < p.wo && lane_on; ox0 += 2 * PPB) {
            float a[2][27];
            unsigned invalid[2] = {0u, 0u};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ox = ox0 + u * PPB;
                const int ix0 = 2 * ox - p.pl;
                const bool pok = ox < p.wo;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int iy = iy0 + tap / 3, ix = ix0 + tap % 3;
                    const bool ok = pok && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w_;
                    invalid[u] |= ok ? 0u : (1u << tap);
                    const unsigned off = ok ? img_off + (unsigned)((iy * p.w_ + ix) * 12) : OOB;
#pragma unroll
                    for (int c = 0; c < 3; ++c) a[u][tap * 3 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 4 * c, 0));
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ox = ox0 + u * PPB;
                if (ox >= p.wo) break;
                float4 acc = make_float4(b4.x + wo_all.x, b4.y + wo_all.y, b4.z + wo_all.z, b4.w + wo_all.w);
                if (invalid[u] != 0u) {      // border pixel: the padding contributes nothing -- not even the offset term
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap)
                        if ((invalid[u] >> tap) & 1u) {
                            const float4 o = *reinterpret_cast<const float4*>(&wo_s[tap][4 * q]);
                            acc.x -= o.x; acc.y -= o.y; acc.z -= o.z; acc.w -= o.w;
                        }
                }
#pragma unroll
                for (int k = 0; k < 27; ++k) {
                    acc.x = fmaf(a[u][k], ws[k].x, acc.x); acc.y = fmaf(a[u][k], ws[k].y, acc.y);
                    acc.z = fmaf(a[u][k], ws[k].z, acc.z); acc.w = fmaf(a[u][k], ws[k].w, acc.w);
                }
                *reinterpret_cast<float4*>(yrow + (size_t)ox * p.cout + 4 * q) = acc;
                su.x += acc.x; su.y += acc.y; su.z += acc.z; su.w += acc.w;
                sq.x = fmaf(acc.x, acc.x, sq.x); sq.y = fmaf(acc.y, acc.y, sq.y); sq.z = fmaf(acc.z, acc.z, sq.z); sq.w = fmaf(acc.w, acc.w, sq.w);
            }
        }
    }
    if (p.stats == nullptr) return;
    if (lane_on) {
        *reinterpret_cast<float4*>(&red[0][pix][4 * q]) = su;
        *reinterpret_cast<float4*>(&red[1][pix][4 * q]) = sq;
    }
    __syncthreads();
    if (t < 2 * p.cout) {
        const int which = t / p.cout, c = t - which * p.cout;
        float s = 0.f;
        for (int k = 0; k < PPB; ++k) s += red[which][k][c];       // fixed order
        p.stats[((size_t)blockIdx.x * 2 + which) * p.cout + c] = s;
    }
}

inline void stem_pad(int size, int* out, int* before) {      // TF SAME, kernel 3, stride 2
    *out = (size + 1) / 2;
    const int total = (*out - 1) * 2 + 3 - size;
    *before = (total > 0 ? total : 0) / 2;
}

}  // namespace

// (internal helpers of ssdseg_stem_conv_fwd / ssdseg_stem_conv_parts in gemm.hip, which declares them inside its extern "C" block)
extern "C" {

// the direct kernel CAN take the layer (sizes the statistics table, whatever the environment says) / DOES take it
// (SSDSEG_STEM_DIRECT=0: never -- A/B runs and a parity-test case of the implicit-GEMM form)
bool ssdseg_stem_direct_eligible(int cout) { return cout % 4 == 0 && cout >= 4 && cout <= 64; }
bool ssdseg_stem_direct_takes(int cout) {
    const char* e = getenv("SSDSEG_STEM_DIRECT");
    if (e != nullptr && e[0] == '0') return false;
    return ssdseg_stem_direct_eligible(cout);
}

// blocks of the launch == partial rows of its statistics table
int ssdseg_stem_direct_blocks(int n, int h) {
    const int rows = n * ((h + 1) / 2);
    return rows < 2048 ? rows : 2048;
}

int ssdseg_stem_direct_fwd(ssdseg_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int n, int h, int wdt, int cout,
                           float in_scale, float in_offset, float* stats) {
    StemArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats;
    a.n = n; a.h = h; a.w_ = wdt; a.cout = cout; a.scale = in_scale; a.offset = in_offset;
    stem_pad(h, &a.ho, &a.pt);
    stem_pad(wdt, &a.wo, &a.pl);
    a.x_bytes = (unsigned)((size_t)n * h * wdt * 12);
    const double m = (double)n * a.ho * a.wo;
    const double bytes = 4.0 * ((double)n * h * wdt * 3 + m * cout + 27.0 * cout), flops = 2.0 * m * 27 * cout;
    const dim3 grid((unsigned)ssdseg_stem_direct_blocks(n, h)), block(256);
    switch (cout / 4) {
#define STEM_CASE(Q) case Q: SSDSEG_LAUNCH(ctx, bytes, flops, stem_fwd_direct_kernel<Q>, grid, block, 0, a); break;
        STEM_CASE(1) STEM_CASE(2) STEM_CASE(3) STEM_CASE(4) STEM_CASE(5) STEM_CASE(6) STEM_CASE(7) STEM_CASE(8)
        STEM_CASE(9) STEM_CASE(10) STEM_CASE(11) STEM_CASE(12) STEM_CASE(13) STEM_CASE(14) STEM_CASE(15) STEM_CASE(16)
#undef STEM_CASE
        default: return -1;
    }
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
