// Detection losses with batch-global hard-negative mining, exact top-k selection, dice losses.
//   confidence_loss   reference losses.py:52-172   (softmax cross-entropy, keep all positives + the k = min(3*P, BG)
//                                                   background anchors of the WHOLE batch with the highest loss)
//   localization_loss reference losses.py:5-49     (smooth-L1 over non-background anchors / #non-background)
//   dice, dice_square reference losses.py:175-264  (API surface)
// Integer/selection work is exact: the k-th largest value is found by an MSB-first radix select on order-preserving
// 32-bit keys and ties are broken by the lower index, which is tf.math.top_k's rule.
// All float reductions are two-level with a fixed order (no float atomics).
#include "common.h"

namespace {

constexpr float KEPS = 1e-7f;
constexpr int BPI = 16;  // blocks per image in the per-anchor passes

__device__ __forceinline__ float clipf(float p) { return fminf(fmaxf(p, KEPS), 1.f - KEPS); }
__device__ __forceinline__ float logcr(float p) { return (float)log((double)p); }
__device__ __forceinline__ float insidef(float p) { return (p >= KEPS && p <= 1.f - KEPS) ? 1.f : 0.f; }

__device__ __forceinline__ unsigned order_key(float v) {
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // larger float <=> larger unsigned key
}

template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
    // deterministic tree over 256 threads; result in v[] of thread 0
    for (int k = 0; k < NV; ++k) {
        __syncthreads();
        red[threadIdx.x] = v[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        v[k] = red[0];
    }
}

struct AnchorTerms {
    float ce, is_bg, not_bg, sl1, loc_nb;
};

__device__ __forceinline__ AnchorTerms anchor_terms(float4 yl, float4 p, float4 yb, float4 pb) {
    AnchorTerms t;
    t.is_bg = yl.x;
    t.not_bg = fabsf(yl.x - 1.f);
    // The per-anchor cross-entropy is the KEY of the batch-global hard-negative selection (losses.py:127-135): two background
    // anchors whose probabilities differ in the last bits are ranked by the float32 value of log().  logf() is a <= 1 ulp
    // approximation whose last bit differs between math libraries (ocml here, NumPy / Eigen elsewhere), which would make the
    // selected SET library-dependent.  The key is therefore defined as the correctly rounded float32 logarithm, obtained as
    // float(log(double(p))) -- reproducible by any host with an IEEE double log (the oracle does the same).
    t.ce = -(yl.x * logcr(clipf(p.x)) + yl.y * logcr(clipf(p.y)) + yl.z * logcr(clipf(p.z)) + yl.w * logcr(clipf(p.w)));
    t.loc_nb = (fabsf(yb.x) + fabsf(yb.y) + fabsf(yb.z) + fabsf(yb.w)) > 0.f ? 1.f : 0.f;
    const float e[4] = {yb.x - pb.x, yb.y - pb.y, yb.z - pb.z, yb.w - pb.w};
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = fabsf(e[k]);
        s += a < 1.f ? e[k] * e[k] * 0.5f : a - 0.5f;
    }
    t.sl1 = s;
    return t;
}

// pass 1: per-anchor cross-entropy, background loss vector, per-block partial sums, batch-global counts
__global__ void __launch_bounds__(256) det_prep_kernel(const float* __restrict__ yl, const float* __restrict__ pl, const float* __restrict__ yb,
                                                       const float* __restrict__ pb, int a, float* __restrict__ bgval,
                                                       float* __restrict__ partial, int* __restrict__ counts) {
    __shared__ float red[256];
    const int img = blockIdx.y;
    const int chunk = (a + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(a, i0 + chunk);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // pos_loss, npos, loc_sum, nloc
    int nbg = 0, npos = 0;
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const long long o = ((long long)img * a + i) * 4;
        const AnchorTerms t = anchor_terms(ld4(yl + o), ld4(pl + o), ld4(yb + o), ld4(pb + o));
        bgval[(long long)img * a + i] = t.ce * t.is_bg;
        acc[0] += t.ce * t.not_bg;
        acc[1] += t.not_bg;
        acc[2] += t.sl1 * t.loc_nb;
        acc[3] += t.loc_nb;
        nbg += t.is_bg != 0.f;
        npos += t.not_bg != 0.f;
    }
    block_sum<4>(acc, red);
    if (threadIdx.x == 0) {
        float* row = partial + ((long long)img * gridDim.x + blockIdx.x) * 4;
        row[0] = acc[0]; row[1] = acc[1]; row[2] = acc[2]; row[3] = acc[3];
    }
    // integer counts: atomics are order-independent
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nbg += __shfl_xor(nbg, o, 64);
        npos += __shfl_xor(npos, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], nbg);
        atomicAdd(&counts[1], npos);
    }
}

// per-image totals in fixed order
__global__ void det_image_stats_kernel(const float* __restrict__ partial, int nblk, int b, float* __restrict__ img_stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * 4) return;
    const int img = i / 4, k = i % 4;
    float s = 0.f;
    for (int q = 0; q < nblk; ++q) s += partial[((long long)img * nblk + q) * 4 + k];
    img_stats[i] = s;
}

// ---- exact top-k over a few 1e5 values: MSB-first radix select (four 8-bit passes) + one ordered tie pass, every pass a
// multi-block kernel.  All cross-block traffic is integer (histogram counts, equal-element counts), so the result does not
// depend on scheduling: mask[i] = 1 for the k largest values, ties at the k-th value resolved lowest index first.
constexpr int TK_THREADS = 256;
constexpr int TK_MAXB = 512;
struct TopkState {
    int hist[4][256];   // hist[p][b]: keys matching the prefix chosen by passes < p whose byte p (from the top) is b
    int eq[TK_MAXB];    // per block of the tie pass: elements equal to the k-th key inside the block's index range
};

__device__ __forceinline__ int topk_k(int k_host, const int* __restrict__ counts) {
    if (counts == nullptr) return k_host;
    const int nbg = counts[0], npos = counts[1];   // mining: k = min(3 * #positives, #background)  (losses.py:113)
    return nbg == 0 ? 0 : min(3 * npos, nbg);
}

// Every block re-derives the selection from the finished histograms of the earlier passes (256 ints each: cheaper than a
// one-block kernel between passes): prefix = the top 8*npass bits of the k-th largest key, krem = its rank inside that bucket.
__device__ __forceinline__ void topk_select(const TopkState* __restrict__ st, int npass, int k, int* lh, unsigned* s_prefix, int* s_krem) {
    const int t = threadIdx.x;
    if (t == 0) { *s_prefix = 0u; *s_krem = k; }
    for (int p = 0; p < npass; ++p) {
        __syncthreads();
        lh[t] = st->hist[p][t];
        __syncthreads();
        if (t == 0) {
            int krem = *s_krem, bin = 255;
            for (; bin > 0; --bin) {
                if (lh[bin] >= krem) break;
                krem -= lh[bin];
            }
            *s_prefix |= (unsigned)bin << (24 - 8 * p);
            *s_krem = krem;
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(TK_THREADS) topk_hist_kernel(const float* __restrict__ v, int n, int k_host, const int* __restrict__ counts,
                                                               TopkState* __restrict__ st, int pass) {
    __shared__ int lh[256];
    __shared__ int bins[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_krem;
    const int t = threadIdx.x;
    const int k = topk_k(k_host, counts);
    if (k <= 0 || k >= n) return;   // uniform: the mask kernel handles the trivial cases
    topk_select(st, pass, k, lh, &s_prefix, &s_krem);
    const unsigned prefix = s_prefix;
    const int shift = 24 - 8 * pass;
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    bins[t] = 0;
    __syncthreads();
    for (int i = blockIdx.x * TK_THREADS + t; i < n; i += gridDim.x * TK_THREADS) {
        const unsigned key = order_key(v[i]);
        if ((key & himask) == prefix) atomicAdd(&bins[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    if (bins[t] != 0) atomicAdd(&st->hist[pass][t], bins[t]);
}

// tie pass: block b owns the contiguous index range [b*chunk, (b+1)*chunk), thread t a contiguous slice of it
// PHASE 0: count the elements equal to the k-th key per block; PHASE 1: write the mask
template <int PHASE>
__global__ void __launch_bounds__(TK_THREADS) topk_tie_kernel(const float* __restrict__ v, int n, int k_host, const int* __restrict__ counts,
                                                              TopkState* __restrict__ st, unsigned char* __restrict__ mask) {
    __shared__ int lh[256];
    __shared__ int scan[TK_THREADS];
    __shared__ unsigned s_prefix;
    __shared__ int s_krem;
    const int t = threadIdx.x;
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int b0 = min(n, (int)blockIdx.x * chunk), b1 = min(n, b0 + chunk);
    const int k = topk_k(k_host, counts);
    if (k <= 0 || k >= n) {
        if (PHASE == 1)
            for (int i = b0 + t; i < b1; i += TK_THREADS) mask[i] = (k >= n && k > 0) ? 1 : 0;
        return;
    }
    topk_select(st, 4, k, lh, &s_prefix, &s_krem);
    const unsigned kth = s_prefix;     // key of the k-th largest element
    const int take_equal = s_krem;     // how many elements equal to it are selected, lowest indices first
    const int sub = (b1 - b0 + TK_THREADS - 1) / TK_THREADS;
    const int i0 = min(b1, b0 + t * sub), i1 = min(b1, i0 + sub);
    int eq = 0;
    for (int i = i0; i < i1; ++i) eq += order_key(v[i]) == kth;
    scan[t] = eq;
    __syncthreads();
    for (int off = 1; off < TK_THREADS; off <<= 1) {   // inclusive Hillis-Steele scan
        const int add = t >= off ? scan[t - off] : 0;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    if (PHASE == 0) {
        if (t == TK_THREADS - 1) st->eq[blockIdx.x] = scan[t];
        return;
    }
    int before = 0;                    // equal elements in the blocks in front of this one
    for (int q = t; q < (int)blockIdx.x; q += TK_THREADS) before += st->eq[q];
    __syncthreads();
    lh[t] = before;
    __syncthreads();
    for (int off = TK_THREADS / 2; off > 0; off >>= 1) {
        if (t < off) lh[t] += lh[t + off];
        __syncthreads();
    }
    int rank = lh[0] + scan[t] - eq;   // equal elements before this thread's slice
    for (int i = i0; i < i1; ++i) {
        const unsigned key = order_key(v[i]);
        unsigned char m = key > kth;
        if (key == kth) { m = rank < take_equal; ++rank; }
        mask[i] = m;
    }
}

// host side: `state` = TopkState scratch (device), zeroed here
int topk_mask_launch(ssdseg_ctx* ctx, const float* v, int n, int k_host, const int* counts, TopkState* state, unsigned char* mask) {
    SSDSEG_HIP(hipMemsetAsync(state, 0, sizeof(TopkState), ctx->stream));
    int nb = cdiv(n, TK_THREADS * 4);
    if (nb > TK_MAXB) nb = TK_MAXB;
    if (nb < 1) nb = 1;
    for (int pass = 0; pass < 4; ++pass) {
        SSDSEG_LAUNCH(ctx, 4.0 * n, 0.0, topk_hist_kernel, dim3(nb), dim3(TK_THREADS), 0, v, n, k_host, counts, state, pass);
        SSDSEG_LAUNCH_CHECK();
    }
    SSDSEG_LAUNCH(ctx, 4.0 * n, 0.0, topk_tie_kernel<0>, dim3(nb), dim3(TK_THREADS), 0, v, n, k_host, counts, state, mask);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 5.0 * n, 0.0, topk_tie_kernel<1>, dim3(nb), dim3(TK_THREADS), 0, v, n, k_host, counts, state, mask);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

// pass 2: gradients and the kept-background loss
__global__ void __launch_bounds__(256) det_final_kernel(const float* __restrict__ yl, const float* __restrict__ pl, const float* __restrict__ yb,
                                                        const float* __restrict__ pb, int a, const unsigned char* __restrict__ mask,
                                                        const float* __restrict__ img_stats, float loss_scale, float* __restrict__ d_logits,
                                                        float* __restrict__ d_boxes, float* __restrict__ partial2) {
    __shared__ float red[256];
    const int img = blockIdx.y;
    const int chunk = (a + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * chunk, i1 = min(a, i0 + chunk);
    const float inv_conf = 1.f / fmaxf(img_stats[img * 4 + 1], 1.f);
    const float inv_loc = 1.f / fmaxf(img_stats[img * 4 + 3], 1.f);
    float acc[1] = {0.f};
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const long long o = ((long long)img * a + i) * 4;
        const float4 y = ld4(yl + o), p = ld4(pl + o), ybx = ld4(yb + o), pbx = ld4(pb + o);
        const AnchorTerms t = anchor_terms(y, p, ybx, pbx);
        const float keep = (float)mask[(long long)img * a + i];
        acc[0] += t.ce * t.is_bg * keep;
        if (d_logits) {
            const float sel = (t.not_bg + t.is_bg * keep) * inv_conf * loss_scale;
            const float4 yi = make_float4(y.x * insidef(p.x), y.y * insidef(p.y), y.z * insidef(p.z), y.w * insidef(p.w));
            const float tot = yi.x + yi.y + yi.z + yi.w;
            st4(d_logits + o, make_float4(sel * (p.x * tot - yi.x), sel * (p.y * tot - yi.y), sel * (p.z * tot - yi.z), sel * (p.w * tot - yi.w)));
        }
        if (d_boxes) {
            const float s = t.loc_nb * inv_loc * loss_scale;
            const float e[4] = {ybx.x - pbx.x, ybx.y - pbx.y, ybx.z - pbx.z, ybx.w - pbx.w};
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ab = fabsf(e[k]);
                d[k] = s * (ab < 1.f ? -e[k] : (e[k] > 0.f ? -1.f : (e[k] < 0.f ? 1.f : 0.f)));
            }
            st4(d_boxes + o, make_float4(d[0], d[1], d[2], d[3]));
        }
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) partial2[(long long)img * gridDim.x + blockIdx.x] = acc[0];
}

__global__ void det_finish_kernel(const float* __restrict__ img_stats, const float* __restrict__ partial2, int nblk, int b,
                                  float* __restrict__ conf_loss, float* __restrict__ loc_loss) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= b) return;
    float bg = 0.f;
    for (int q = 0; q < nblk; ++q) bg += partial2[(long long)img * nblk + q];
    const float* s = img_stats + img * 4;
    if (conf_loss) conf_loss[img] = (s[0] + bg) / fmaxf(s[1], 1.f);
    if (loc_loss) loc_loss[img] = s[2] / fmaxf(s[3], 1.f);
}

// ---- dice / dice_square: per image (intersection_c, total_c) over the pixels
__global__ void __launch_bounds__(256) dice_partial_kernel(const float* __restrict__ y, const float* __restrict__ p, int hw, int squared,
                                                           float* __restrict__ partial) {
    __shared__ float red[256];
    const int img = blockIdx.y;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const long long o = ((long long)img * hw + i) * 4;
        const float4 a = ld4(y + o), b = ld4(p + o);
        if (squared == 2) {   // weighted cross-entropy on probabilities (losses.py:294-299): acc[c] = -sum y*log(clip p)
            acc[0] -= a.x * logf(clipf(b.x)); acc[1] -= a.y * logf(clipf(b.y)); acc[2] -= a.z * logf(clipf(b.z)); acc[3] -= a.w * logf(clipf(b.w));
            continue;
        }
        acc[0] += a.x * b.x; acc[1] += a.y * b.y; acc[2] += a.z * b.z; acc[3] += a.w * b.w;
        if (squared) { acc[4] += a.x * a.x + b.x * b.x; acc[5] += a.y * a.y + b.y * b.y; acc[6] += a.z * a.z + b.z * b.z; acc[7] += a.w * a.w + b.w * b.w; }
        else { acc[4] += a.x + b.x; acc[5] += a.y + b.y; acc[6] += a.z + b.z; acc[7] += a.w + b.w; }
    }
    block_sum<8>(acc, red);
    if (threadIdx.x == 0)
        for (int k = 0; k < 8; ++k) partial[((long long)img * gridDim.x + blockIdx.x) * 8 + k] = acc[k];
}

__global__ void dice_finish_kernel(const float* __restrict__ partial, int nblk, int n, float4 cw, int mode, float* __restrict__ loss) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= n) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nblk; ++q)
        for (int k = 0; k < 8; ++k) s[k] += partial[((long long)img * nblk + q) * 8 + k];
    const float w[4] = {cw.x, cw.y, cw.z, cw.w};
    float l = 0.f;
    for (int c = 0; c < 4; ++c) l += (mode == 2 ? s[c] : (1.f - (2.f * s[c] + KEPS) / (s[4 + c] + KEPS))) * w[c];
    loss[img] = l;
}

}  // namespace

extern "C" {

int ssdseg_topk_mask(ssdseg_ctx* ctx, const float* values, int n, int k, uint8_t* mask) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(values != nullptr, 2);
    SSDSEG_ARG(n > 0, 3);
    SSDSEG_ARG(k >= 0, 4);
    SSDSEG_ARG(mask != nullptr, 5);
    void* ws;
    int rc = ssdseg_workspace(ctx, sizeof(TopkState), &ws);
    if (rc) return rc;
    return topk_mask_launch(ctx, values, n, k, nullptr, (TopkState*)ws, mask);
}

int ssdseg_det_loss(ssdseg_ctx* ctx, const float* y_labels, const float* p_labels, const float* y_boxes, const float* p_boxes, int b,
                    int a, int c, float loss_scale, float* conf_loss, float* loc_loss, float* d_logits, float* d_boxes,
                    uint8_t* keep_mask) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(y_labels && p_labels && y_boxes && p_boxes, 2);
    SSDSEG_ARG(b > 0, 6);
    SSDSEG_ARG(a > 0, 7);
    SSDSEG_ARG(c == 4, 8);   // quirk Q2: the reference only works with 4 classes (models.py:250-253 vs :265-268)
    const long long n = (long long)b * a;
    SSDSEG_ARG(n < (1LL << 30), 6);
    // workspace: bgval[n] f32 | partial[b][BPI][4] | img_stats[b][4] | partial2[b][BPI] | counts[2] i32 | top-k state | mask[n] u8
    const size_t o_bg = 0, o_part = o_bg + (size_t)n * 4, o_stats = o_part + (size_t)b * BPI * 16, o_p2 = o_stats + (size_t)b * 16,
                 o_cnt = o_p2 + (size_t)b * BPI * 4, o_tk = o_cnt + 16, o_mask = o_tk + sizeof(TopkState), total = o_mask + (size_t)n;
    void* ws;
    int rc = ssdseg_workspace(ctx, total, &ws);
    if (rc) return rc;
    char* base = (char*)ws;
    float* bgval = (float*)(base + o_bg);
    float* partial = (float*)(base + o_part);
    float* img_stats = (float*)(base + o_stats);
    float* partial2 = (float*)(base + o_p2);
    int* counts = (int*)(base + o_cnt);
    unsigned char* mask = (unsigned char*)(base + o_mask);
    SSDSEG_HIP(hipMemsetAsync(counts, 0, 16, ctx->stream));
    const double pass_bytes = 64.0 * n;
    SSDSEG_LAUNCH(ctx, pass_bytes, 0.0, det_prep_kernel, dim3(BPI, b), dim3(256), 0, y_labels, p_labels, y_boxes, p_boxes, a, bgval, partial,
                  counts);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 0.0, 0.0, det_image_stats_kernel, dim3(cdiv(b * 4, 64)), dim3(64), 0, partial, BPI, b, img_stats);
    SSDSEG_LAUNCH_CHECK();
    rc = topk_mask_launch(ctx, bgval, (int)n, 0, (const int*)counts, (TopkState*)(base + o_tk), mask);
    if (rc) return rc;
    SSDSEG_LAUNCH(ctx, pass_bytes + 32.0 * n, 0.0, det_final_kernel, dim3(BPI, b), dim3(256), 0, y_labels, p_labels, y_boxes, p_boxes, a, mask,
                  img_stats, loss_scale, d_logits, d_boxes, partial2);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 0.0, 0.0, det_finish_kernel, dim3(cdiv(b, 64)), dim3(64), 0, img_stats, partial2, BPI, b, conf_loss, loc_loss);
    SSDSEG_LAUNCH_CHECK();
    if (keep_mask) SSDSEG_HIP(hipMemcpyAsync(keep_mask, mask, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int ssdseg_dice_loss(ssdseg_ctx* ctx, const float* y_true, const float* p, int n, int hw, int c, const float* class_weights_host,
                     int squared, float* loss) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(y_true != nullptr, 2);
    SSDSEG_ARG(p != nullptr, 3);
    SSDSEG_ARG(n > 0 && hw > 0, 4);
    SSDSEG_ARG(c == 4, 6);
    SSDSEG_ARG(class_weights_host != nullptr, 7);
    SSDSEG_ARG(loss != nullptr, 9);
    int nblk = (hw + 2047) / 2048;
    if (nblk > 64) nblk = 64;
    void* ws;
    int rc = ssdseg_workspace(ctx, (size_t)n * nblk * 8 * sizeof(float), &ws);
    if (rc) return rc;
    SSDSEG_LAUNCH(ctx, 32.0 * n * hw, 0.0, dice_partial_kernel, dim3(nblk, n), dim3(256), 0, y_true, p, hw, squared, (float*)ws);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 0.0, 0.0, dice_finish_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (const float*)ws, nblk, n,
                  make_float4(class_weights_host[0], class_weights_host[1], class_weights_host[2], class_weights_host[3]), squared, loss);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
