// Micro-benchmark: what a register-only loop of v_mfma_f32_32x32x2_f32 sustains on this chip (no memory traffic at all), to put
// the GEMM / conv kernels' TFLOP/s next to an ACHIEVABLE peak rather than the 157.3 TFLOP/s data-sheet figure (2.4 GHz boost).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip     run: ./mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 2;   // 256-thread blocks = one wave per SIMD each
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * blocks_per_cu, iters = 20000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nacc = 1; nacc <= 4; nacc *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (nacc == 1) mfma_loop<1><<<blocks, 256>>>(out, iters, 1.f, 1.f);
            else if (nacc == 2) mfma_loop<2><<<blocks, 256>>>(out, iters, 1.f, 1.f);
            else mfma_loop<4><<<blocks, 256>>>(out, iters, 1.f, 1.f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * nacc * 4096.0;
            if (rep == 1)
                printf("%s, %d CUs, %d waves/SIMD, %d independent accumulators per wave: %.1f ms, %.1f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
                       prop.gcnArchName, cus, blocks_per_cu, nacc, ms, flops / (ms * 1e-3) / 1e12,
                       2.4e9 * (ms * 1e-3) / ((double)iters * 8 * nacc * blocks_per_cu));
        }
    }
    return 0;
}
