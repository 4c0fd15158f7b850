// Device-side expansion of a COMPACT training batch (the step on the input side of the hot path).
//
// The reference's tf.data map (DataEncoderDecoder.read_and_encode, reference datacoder.py:302-347) produces, per sample, a
// float32 image, a float32 one-hot mask and -- through _encode_ground_truth_labels_boxes (:177-300) -- the encoded anchors;
// with a random horizontal flip applied to all three consistently (:337-345, boxes x -> W - x :202-203, quirk Q8).  Handing those
// float tensors to the GPU costs 285 MB per batch-32 step over PCIe (118 MB images, 157 MB one-hot masks, 10 MB anchors).
// Here the host hands over what the files contain -- uint8 pixels, uint8 class indices, the (label, box) rows -- 39 MB, and
//   ssdseg_expand_inputs   casts the pixels to float32 (tf.cast :327), one-hots the class index (tf.one_hot :332: an index >=
//                          depth gives an all-zero row) and mirrors both left-right where the sample's flip flag is set
//                          (tf.image.flip_left_right :341-342), writing straight into the engine's input / target buffers;
//   ssdseg_flip_gt_boxes   mirrors the ground-truth rows of the flagged samples (xmin' = W - xmax, xmax' = W - xmin :202-203),
// after which ssdseg_encode_targets (boxes.hip) runs on the device as before.  Byte / integer work throughout: results are
// bit-identical to the host path (tests/test_gpu_input_pipeline.py).
#include "common.h"

namespace {

// one thread per OUTPUT pixel: 3 bytes + 1 byte in, 3 + c floats out (c <= 8; the one-hot row as float4 stores when c == 4)
__global__ void __launch_bounds__(256) expand_inputs_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ midx, const uint8_t* __restrict__ flip,
                                                            float* __restrict__ out_img, float* __restrict__ out_mask, int h, int w, int c,
                                                            long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % w);
        const long long row = i / w;                 // (image, y)
        const int n = (int)(row / h);
        const bool f = flip != nullptr && flip[n] != 0;
        const long long src = row * w + (f ? w - 1 - x : x);
        if (img != nullptr) {
            const uint8_t* p = img + src * 3;
            float* o = out_img + i * 3;
            o[0] = (float)p[0]; o[1] = (float)p[1]; o[2] = (float)p[2];
        }
        if (midx != nullptr) {
            const int k = midx[src];
            float* o = out_mask + i * c;
            if (c == 4) {
                st4(o, make_float4(k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f, k == 3 ? 1.f : 0.f));
            } else {
                for (int j = 0; j < c; ++j) o[j] = k == j ? 1.f : 0.f;
            }
        }
    }
}

__global__ void flip_gt_boxes_kernel(float* __restrict__ gt, const int* __restrict__ count, const uint8_t* __restrict__ flip, int b, int gmax,
                                     float width) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // over b * gmax rows
    if (i >= b * gmax) return;
    const int n = i / gmax, g = i - n * gmax;
    if (flip[n] == 0 || g >= count[n]) return;
    float* r = gt + (long long)i * 5;                          // (label, xmin, ymin, xmax, ymax)
    const float xmin = r[1], xmax = r[3];
    r[1] = width - xmax;
    r[3] = width - xmin;
}

}  // namespace

extern "C" {

int ssdseg_expand_inputs(ssdseg_ctx* ctx, const uint8_t* images_u8, const uint8_t* mask_index_u8, const uint8_t* flip, float* images_f32,
                         float* mask_onehot, int b, int h, int w, int c) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(images_u8 != nullptr || mask_index_u8 != nullptr, 2);
    SSDSEG_ARG(images_u8 == nullptr || images_f32 != nullptr, 5);
    SSDSEG_ARG(mask_index_u8 == nullptr || mask_onehot != nullptr, 6);
    SSDSEG_ARG(b > 0 && h > 0 && w > 0, 7);
    SSDSEG_ARG(c > 0 && c <= 8, 10);
    const long long total = (long long)b * h * w;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const double bytes = (double)total * ((images_u8 ? 3.0 + 12.0 : 0.0) + (mask_index_u8 ? 1.0 + 4.0 * c : 0.0));
    SSDSEG_LAUNCH(ctx, bytes, 0.0, expand_inputs_kernel, dim3(blocks), dim3(256), 0, images_u8, mask_index_u8, flip, images_f32, mask_onehot, h, w, c,
                  total);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

int ssdseg_flip_gt_boxes(ssdseg_ctx* ctx, float* gt, const int32_t* gt_count, const uint8_t* flip, int b, int gmax, float image_width) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(gt != nullptr, 2);
    SSDSEG_ARG(gt_count != nullptr, 3);
    SSDSEG_ARG(flip != nullptr, 4);
    SSDSEG_ARG(b > 0 && gmax > 0, 5);
    SSDSEG_LAUNCH(ctx, 40.0 * b * gmax, 0.0, flip_gt_boxes_kernel, dim3(cdiv(b * gmax, 256)), dim3(256), 0, gt, gt_count, flip, b, gmax, image_width);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
