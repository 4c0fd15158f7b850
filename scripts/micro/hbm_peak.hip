// Micro-benchmark: HBM streaming ceilings of this chip for the access mixes of the depthwise kernels -- plain float4 copy
// (1 read : 1 write, the forward conv's mix) and 3 reads : 1 write (the backward conv with a BatchNorm gradient view: x, g, y
// in, dx out) -- grid-strided, 256-thread blocks, sizes well beyond the 256 MiB Infinity Cache.
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_peak hbm_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) copy1(const float4* __restrict__ a, float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
__global__ void __launch_bounds__(256) mix31(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                             float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i], z = c[i];
        o[i] = make_float4(x.x + y.x * z.x, x.y + y.y * z.y, x.z + y.z * z.z, x.w + y.w * z.w);
    }
}

int main() {
    const size_t n = (size_t)64 << 20;   // float4 elements: 1 GiB per tensor
    float4 *a, *b, *c, *o;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&o, n * 16);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int kind = 0; kind < 2; ++kind) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) copy1<<<blocks, 256>>>(a, o, n);
                else mix31<<<blocks, 256>>>(a, b, c, o, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double bytes = (kind == 0 ? 2.0 : 4.0) * n * 16;
            printf("%s, %d blocks: %.3f ms, %.2f TB/s\n", kind == 0 ? "copy (1 read : 1 write)" : "3 reads : 1 write", blocks, best, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
