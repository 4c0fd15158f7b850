// Anchor-side integer / index kernels (this file is compiled with -ffp-contract=off so that every IoU is the same
// IEEE sequence of +,-,*,/ as the NumPy restatement -> index decisions are bit-exact):
//   ssdseg_encode_targets  DataEncoderDecoder._encode_ground_truth_labels_boxes   reference datacoder.py:205-300
//   ssdseg_decode_boxes    DecodeBoxesCentroidsOffsets.call                        reference layers.py:58-79
//   ssdseg_combined_nms    NonMaximumSuppression.call -> tf.image.combined_non_max_suppression + repack  layers.py:141-162
//   ssdseg_seg_suppress    SegmentationSuppression.call                             reference layers.py:203-210
// One workgroup per image (encode) or per (class, image) (NMS); arg-max reductions carry (value, lowest index).
#include "common.h"

namespace {

constexpr int GMAX = 64;      // ground-truth boxes per image the encoder accepts
constexpr int ENC_T = 1024;

struct Best {
    float v;
    int i;
};
__device__ __forceinline__ bool better(float v, int i, float v2, int i2) { return v > v2 || (v == v2 && i < i2); }

// block-wide arg-max with lowest-index tie break (tf.math.argmax returns the first maximum)
template <int T>
__device__ __forceinline__ Best block_argmax(float v, int i, float* rv, int* ri) {
    __syncthreads();
    rv[threadIdx.x] = v;
    ri[threadIdx.x] = i;
    __syncthreads();
    for (int s = T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float v2 = rv[threadIdx.x + s];
            const int i2 = ri[threadIdx.x + s];
            if (better(v2, i2, rv[threadIdx.x], ri[threadIdx.x])) { rv[threadIdx.x] = v2; ri[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    Best b{rv[0], ri[0]};
    return b;
}

// IoU with the reference's pixel-index (+1) convention (datacoder.py:206-220), operation order as written there
__device__ __forceinline__ float iou_plus1(float ax0, float ay0, float ax1, float ay1, float area_a, float gx0, float gy0, float gx1, float gy1,
                                           float area_g) {
    const float ix0 = fmaxf(ax0, gx0), iy0 = fmaxf(ay0, gy0), ix1 = fminf(ax1, gx1), iy1 = fminf(ay1, gy1);
    const float inter = fmaxf(0.f, ix1 - ix0 + 1.f) * fmaxf(0.f, iy1 - iy0 + 1.f);
    return inter / (area_a + area_g - inter);
}

__global__ void __launch_bounds__(ENC_T) encode_kernel(const float* __restrict__ anchors, int a, const float* __restrict__ gt,
                                                       const int* __restrict__ gt_count, int gmax, int c, float thr, float4 stds,
                                                       float* __restrict__ labels, float* __restrict__ boxes, int* __restrict__ match) {
    __shared__ float s_g[GMAX][6];      // label, xmin, ymin, xmax, ymax, area
    __shared__ int s_best_anchor[GMAX];
    __shared__ int s_valid[GMAX];
    __shared__ float rv[ENC_T];
    __shared__ int ri[ENC_T];
    const int img = blockIdx.x, t = threadIdx.x;
    int G = gt_count[img];
    if (G > gmax) G = gmax;
    if (t < G) {
        const float* p = gt + ((long long)img * gmax + t) * 5;
        s_g[t][0] = p[0]; s_g[t][1] = p[1]; s_g[t][2] = p[2]; s_g[t][3] = p[3]; s_g[t][4] = p[4];
        s_g[t][5] = (p[3] - p[1] + 1.f) * (p[4] - p[2] + 1.f);   // (xmax-xmin+1)*(ymax-ymin+1), datacoder.py:206
    }
    __syncthreads();
    // step 1: best anchor of every ground-truth box (datacoder.py:230-231)
    for (int g = 0; g < G; ++g) {
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int d = t; d < a; d += ENC_T) {
            const float4 A = ld4(anchors + (long long)d * 4);
            const float area_a = (A.w - A.y + 1.f) * (A.z - A.x + 1.f);   // (ymax-ymin+1)*(xmax-xmin+1), datacoder.py:112
            const float v = iou_plus1(A.x, A.y, A.z, A.w, area_a, s_g[g][1], s_g[g][2], s_g[g][3], s_g[g][4], s_g[g][5]);
            if (better(v, d, bv, bi)) { bv = v; bi = d; }
        }
        const Best b = block_argmax<ENC_T>(bv, bi, rv, ri);
        if (t == 0) { s_best_anchor[g] = b.i; s_valid[g] = b.v > 0.f; }
    }
    __syncthreads();
    // steps 2+3: best ground truth of every anchor, merge, last-writer-wins (datacoder.py:236-298)
    for (int d = t; d < a; d += ENC_T) {
        const float4 A = ld4(anchors + (long long)d * 4);
        const float area_a = (A.w - A.y + 1.f) * (A.z - A.x + 1.f);
        float bv = -1.f;
        int bg = -1, m1 = -1;
        for (int g = 0; g < G; ++g) {
            const float v = iou_plus1(A.x, A.y, A.z, A.w, area_a, s_g[g][1], s_g[g][2], s_g[g][3], s_g[g][4], s_g[g][5]);
            if (v > bv) { bv = v; bg = g; }
            if (s_valid[g] && s_best_anchor[g] == d) m1 = g;      // ascending g: ends at the largest member of S1
        }
        const int s2 = (G > 0 && bv > thr) ? bg : -1;
        const bool s2_in_s1 = s2 >= 0 && s_valid[s2] && s_best_anchor[s2] == d;
        const int fin = (s2 >= 0 && !s2_in_s1) ? s2 : m1;
        float* lab = labels + ((long long)img * a + d) * c;
        float4 off = f4(0.f);
        int cls = 0;
        if (fin >= 0) {
            cls = (int)s_g[fin][0];
            const float acx = (A.z + A.x) / 2.f, acy = (A.w + A.y) / 2.f, aw = A.z - A.x + 1.f, ah = A.w - A.y + 1.f;
            const float gcx = (s_g[fin][3] + s_g[fin][1]) / 2.f, gcy = (s_g[fin][4] + s_g[fin][2]) / 2.f;
            const float gw = s_g[fin][3] - s_g[fin][1] + 1.f, gh = s_g[fin][4] - s_g[fin][2] + 1.f;
            off.x = (gcx - acx) / aw / stds.x;                 // datacoder.py:266-269
            off.y = (gcy - acy) / ah / stds.y;
            off.z = logf(gw / aw + 1.f) / stds.z;
            off.w = logf(gh / ah + 1.f) / stds.w;
        }
        for (int k = 0; k < c; ++k) lab[k] = (k == cls) ? 1.f : 0.f;
        st4(boxes + ((long long)img * a + d) * 4, off);
        if (match) match[(long long)img * a + d] = fin;
    }
}

__global__ void decode_kernel(const float* __restrict__ offsets, const float* __restrict__ cent, long long total, int a, float4 stds,
                              float* __restrict__ corners) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const float4 o = ld4(offsets + i * 4);
        const float4 d = ld4(cent + (i % a) * 4);   // center_x, center_y, width, height
        const float cx = o.x * stds.x * d.z + d.x;
        const float cy = o.y * stds.y * d.w + d.y;
        const float w = (expf(o.z * stds.z) - 1.f) * d.z;
        const float h = (expf(o.w * stds.w) - 1.f) * d.w;
        // (ymin, xmin, ymax, xmax): the order tf.image.combined_non_max_suppression wants (layers.py:73-79)
        st4(corners + i * 4, make_float4(cy - (h - 1.f) / 2.f, cx - (w - 1.f) / 2.f, cy + (h - 1.f) / 2.f, cx + (w - 1.f) / 2.f));
    }
}

// TF's NMS IoU: raw coordinates, no +1, degenerate boxes never overlap
__device__ __forceinline__ float iou_tf(float4 p, float4 q) {
    const float py0 = fminf(p.x, p.z), px0 = fminf(p.y, p.w), py1 = fmaxf(p.x, p.z), px1 = fmaxf(p.y, p.w);
    const float qy0 = fminf(q.x, q.z), qx0 = fminf(q.y, q.w), qy1 = fmaxf(q.x, q.z), qx1 = fmaxf(q.y, q.w);
    const float area_p = (py1 - py0) * (px1 - px0), area_q = (qy1 - qy0) * (qx1 - qx0);
    if (area_p <= 0.f || area_q <= 0.f) return 0.f;
    const float iy0 = fmaxf(py0, qy0), ix0 = fmaxf(px0, qx0), iy1 = fminf(py1, qy1), ix1 = fminf(px1, qx1);
    const float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
    return inter / (area_p + area_q - inter);
}

constexpr int NMS_T = 1024;
// grid (classes, images): greedy NMS of one class of one image; "pick the best remaining, drop everything that
// overlaps it" is the same selection as TF's score-ordered scan, in at most max_per_class rounds.
__global__ void __launch_bounds__(NMS_T) nms_class_kernel(const float* __restrict__ corners, const float* __restrict__ probs, int a, int c,
                                                          int max_per_class, float iou_thr, float score_thr, int* __restrict__ sel) {
    extern __shared__ unsigned char alive[];
    __shared__ float rv[NMS_T];
    __shared__ int ri[NMS_T];
    const int cls = blockIdx.x, img = blockIdx.y, t = threadIdx.x;
    const float* sc = probs + (long long)img * a * c + cls;
    const float* bx = corners + (long long)img * a * 4;
    for (int i = t; i < a; i += NMS_T) alive[i] = sc[(long long)i * c] > score_thr;
    __syncthreads();
    int* out = sel + ((long long)img * c + cls) * max_per_class;
    for (int r = 0; r < max_per_class; ++r) {
        float bv = -1.f;
        int bi = 0x7fffffff;
        for (int i = t; i < a; i += NMS_T) {
            if (alive[i]) {
                const float v = sc[(long long)i * c];
                if (better(v, i, bv, bi)) { bv = v; bi = i; }
            }
        }
        const Best b = block_argmax<NMS_T>(bv, bi, rv, ri);
        if (b.i == 0x7fffffff) {                       // nothing left
            if (t == 0) for (int q = r; q < max_per_class; ++q) out[q] = -1;
            return;
        }
        if (t == 0) out[r] = b.i;
        const float4 kb = ld4(bx + (long long)b.i * 4);
        for (int i = t; i < a; i += NMS_T) {
            if (alive[i] && (i == b.i || iou_tf(ld4(bx + (long long)i * 4), kb) > iou_thr)) alive[i] = 0;
        }
        __syncthreads();
    }
}

// one thread per image: merge the per-class picks, order by (score desc, anchor asc, class asc), keep max_total
__global__ void nms_merge_kernel(const float* __restrict__ corners, const float* __restrict__ probs, const int* __restrict__ sel, int b, int a,
                                 int c, int max_per_class, int max_total, float* __restrict__ out, int* __restrict__ valid) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= b) return;
    float* o = out + (long long)img * max_total * 6;
    int taken = 0;
    float last_s = 0.f;
    int last_i = -1, last_c = -1;
    const int ncand = c * max_per_class;
    for (; taken < max_total; ++taken) {
        // selection "sort": next best candidate strictly after (last_s, last_i, last_c) in the total order
        float bs = -1.f;
        int bi = 0x7fffffff, bc = 0x7fffffff;
        for (int q = 0; q < ncand; ++q) {
            const int idx = sel[(long long)img * ncand + q];
            if (idx < 0) continue;
            const int cls = q / max_per_class;
            const float s = probs[((long long)img * a + idx) * c + cls];
            if (taken > 0) {
                const bool after = s < last_s || (s == last_s && (idx > last_i || (idx == last_i && cls > last_c)));
                if (!after) continue;
            }
            if (s > bs || (s == bs && (idx < bi || (idx == bi && cls < bc)))) { bs = s; bi = idx; bc = cls; }
        }
        if (bc == 0x7fffffff) break;
        const float4 k = ld4(corners + ((long long)img * a + bi) * 4);   // ymin, xmin, ymax, xmax
        float* row = o + taken * 6;
        row[0] = (float)bc; row[1] = bs; row[2] = k.y; row[3] = k.x; row[4] = k.w; row[5] = k.z;   // label, prob, xmin, ymin, xmax, ymax
        last_s = bs; last_i = bi; last_c = bc;
    }
    if (valid) valid[img] = taken;
    for (int r = taken; r < max_total; ++r)
        for (int k = 0; k < 6; ++k) o[r * 6 + k] = 0.f;
}

__global__ void seg_present_kernel(const float* __restrict__ mask, long long npix, int* __restrict__ flags) {
    int local = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const float4 p = ld4(mask + i * 4);
        int am = 0;                      // tf.math.argmax: first maximum
        float m = p.x;
        if (p.y > m) { m = p.y; am = 1; }
        if (p.z > m) { m = p.z; am = 2; }
        if (p.w > m) { m = p.w; am = 3; }
        local |= 1 << am;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local |= __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicOr(flags, local);
}

__global__ void seg_apply_kernel(const float* __restrict__ probs, const int* __restrict__ flags, long long rows, float* __restrict__ out) {
    const int f = flags[0];
    const float4 m = make_float4((f & 1) ? 1.f : 0.f, (f & 2) ? 1.f : 0.f, (f & 4) ? 1.f : 0.f, (f & 8) ? 1.f : 0.f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long long)gridDim.x * blockDim.x) {
        const float4 p = ld4(probs + i * 4);
        st4(out + i * 4, make_float4(p.x * m.x, p.y * m.y, p.z * m.z, p.w * m.w));
    }
}

int ew_blocks(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 4096 ? (b < 1 ? 1 : b) : 4096);
}

}  // namespace

extern "C" {

int ssdseg_encode_targets(ssdseg_ctx* ctx, const float* anchors_corners, int a, const float* gt, const int32_t* gt_count, int b, int gmax,
                          int c, float iou_threshold, const float* stds4_host, float* labels, float* boxes, int32_t* match) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(anchors_corners != nullptr, 2);
    SSDSEG_ARG(a > 0, 3);
    SSDSEG_ARG(gt != nullptr, 4);
    SSDSEG_ARG(gt_count != nullptr, 5);
    SSDSEG_ARG(b > 0, 6);
    SSDSEG_ARG(gmax > 0 && gmax <= GMAX, 7);
    SSDSEG_ARG(c > 1, 8);
    SSDSEG_ARG(stds4_host != nullptr, 10);
    SSDSEG_ARG(labels != nullptr, 11);
    SSDSEG_ARG(boxes != nullptr, 12);
    SSDSEG_LAUNCH(ctx, 4.0 * ((double)b * a * (c + 4) + 4.0 * a), 0.0, encode_kernel, dim3(b), dim3(ENC_T), 0, anchors_corners, a, gt, gt_count,
                  gmax, c, iou_threshold, make_float4(stds4_host[0], stds4_host[1], stds4_host[2], stds4_host[3]), labels, boxes, match);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_decode_boxes(ssdseg_ctx* ctx, const float* offsets, const float* anchors_centroids, int b, int a, const float* stds4_host,
                        float* corners) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(offsets != nullptr, 2);
    SSDSEG_ARG(anchors_centroids != nullptr, 3);
    SSDSEG_ARG(b > 0, 4);
    SSDSEG_ARG(a > 0, 5);
    SSDSEG_ARG(stds4_host != nullptr, 6);
    SSDSEG_ARG(corners != nullptr, 7);
    const long long total = (long long)b * a;
    SSDSEG_LAUNCH(ctx, 32.0 * total, 0.0, decode_kernel, dim3(ew_blocks(total)), dim3(256), 0, offsets, anchors_centroids, total, a,
                  make_float4(stds4_host[0], stds4_host[1], stds4_host[2], stds4_host[3]), corners);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_combined_nms(ssdseg_ctx* ctx, const float* corners, const float* probs, int b, int a, int c, int max_per_class, int max_total,
                        float iou_threshold, float score_threshold, float* out, int32_t* valid) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(corners != nullptr, 2);
    SSDSEG_ARG(probs != nullptr, 3);
    SSDSEG_ARG(b > 0, 4);
    SSDSEG_ARG(a > 0 && a <= 150000, 5);      // one byte of LDS per anchor
    SSDSEG_ARG(c > 0 && c <= 64, 6);
    SSDSEG_ARG(max_per_class > 0, 7);
    SSDSEG_ARG(max_total > 0, 8);
    SSDSEG_ARG(out != nullptr, 11);
    void* ws;
    int rc = ssdseg_workspace(ctx, (size_t)b * c * max_per_class * sizeof(int), &ws);
    if (rc) return rc;
    int* sel = (int*)ws;
    SSDSEG_LAUNCH(ctx, 4.0 * b * a * (4 + c), 0.0, nms_class_kernel, dim3(c, b), dim3(NMS_T), (size_t)((a + 15) / 16 * 16), corners, probs, a, c,
                  max_per_class, iou_threshold, score_threshold, sel);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 0.0, 0.0, nms_merge_kernel, dim3(cdiv(b, 64)), dim3(64), 0, corners, probs, (const int*)sel, b, a, c, max_per_class,
                  max_total, out, valid);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_seg_suppress(ssdseg_ctx* ctx, const float* mask_prob, int n_pixels_total, int c, const float* probs, int rows, float* probs_out) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(mask_prob != nullptr, 2);
    SSDSEG_ARG(n_pixels_total > 0, 3);
    SSDSEG_ARG(c == 4, 4);    // depth hard-coded to 4 in the reference too (layers.py:204)
    SSDSEG_ARG(probs != nullptr, 5);
    SSDSEG_ARG(rows > 0, 6);
    SSDSEG_ARG(probs_out != nullptr, 7);
    void* ws;
    int rc = ssdseg_workspace(ctx, 16, &ws);
    if (rc) return rc;
    SSDSEG_HIP(hipMemsetAsync(ws, 0, 16, ctx->stream));
    SSDSEG_LAUNCH(ctx, 16.0 * n_pixels_total, 0.0, seg_present_kernel, dim3(ew_blocks(n_pixels_total)), dim3(256), 0, mask_prob,
                  (long long)n_pixels_total, (int*)ws);
    SSDSEG_LAUNCH_CHECK();
    SSDSEG_LAUNCH(ctx, 32.0 * rows, 0.0, seg_apply_kernel, dim3(ew_blocks(rows)), dim3(256), 0, probs, (const int*)ws, (long long)rows, probs_out);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
