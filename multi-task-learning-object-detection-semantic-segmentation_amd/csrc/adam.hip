// Adam as Keras 2.13 applies it (NB03#cell14: tf.keras.optimizers.Adam(1e-4); semantics SURVEY.md App. B.10):
//   m += (g - m)(1 - b1);  v += (g^2 - v)(1 - b2);  p -= lr*sqrt(1 - b2^t)/(1 - b1^t) * m / (sqrt(v) + eps)
// One fused multi-tensor pass over the flat parameter bucket (HBM-bound: 4 streams read, 3 written).
#include <math.h>

#include "common.h"

namespace {
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n4,
                            size_t n, float alpha, float one_minus_b1, float one_minus_b2, float eps, float gscale) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = ld4(p + 4 * i), gg = ld4(g + 4 * i), mm = ld4(m + 4 * i), vv = ld4(v + 4 * i);
        float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ge[k] * gscale;
            me[k] += (gk - me[k]) * one_minus_b1;
            ve[k] += (gk * gk - ve[k]) * one_minus_b2;
            pe[k] -= alpha * me[k] / (sqrtf(ve[k]) + eps);
        }
        st4(p + 4 * i, pp); st4(m + 4 * i, mm); st4(v + 4 * i, vv);
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float gk = g[i] * gscale;
        m[i] += (gk - m[i]) * one_minus_b1;
        v[i] += (gk * gk - v[i]) * one_minus_b2;
        p[i] -= alpha * m[i] / (sqrtf(v[i]) + eps);
    }
}
}  // namespace

extern "C" int ssdseg_adam_step(ssdseg_ctx* ctx, float* params, const float* grads, float* m, float* v, size_t count, float lr,
                                float beta1, float beta2, float eps, int step, float grad_scale) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(params && grads && m && v, 2);
    SSDSEG_ARG(step >= 1, 11);
    if (count == 0) return 0;
    { int jrc = ssdseg_join(ctx); if (jrc) return jrc; }   // weight gradients may still be in flight on the side stream
    const double alpha = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    const size_t n4 = count / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    SSDSEG_LAUNCH(ctx, 28.0 * (double)count, 0.0, adam_kernel, dim3((unsigned)blocks), dim3(256), 0, params, grads, m, v, n4, count, (float)alpha,
                       1.f - beta1, 1.f - beta2, eps, grad_scale);
    SSDSEG_LAUNCH_CHECK();
    return 0;
}
