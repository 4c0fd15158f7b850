// Micro-benchmark: what does an instruction placed in the shadow of a v_mfma_f32_32x32x2_f32 cost on gfx950, one wave per SIMD?
// The F(4x4,3x3) conv kernel (csrc/conv3_wino4.h) runs ~4.3 fillers per MFMA and its ablations price every one of them at full
// cost; this takes the kernel out: a register-only loop of chained MFMAs (4 per accumulator, as in the kernel) with NF fillers of
// one KIND after each, order pinned by sched_barrier.  Cycles per MFMA from s_memtime (shader clock).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_filler mfma_filler.hip     run: ./mfma_filler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: v_fma_f32   1: v_add_u32   2: v_pk_fma_f32   3: ds_read_b128 (results never waited for inside the loop)   4: v_mov_b32
//      5: v_fma_f32 whose inputs are MFMA operands of the NEXT mfma (dependent operand, like the kernel's fragments)
template <int NF, int KIND, int NACC, int EVERY>
__global__ void __launch_bounds__(256) filler_loop(float* out, unsigned long long* clk, int iters, float a, float b) {
    __shared__ float lds[256 * 4 * 8];
    for (int i = threadIdx.x; i < 256 * 4 * 8; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float f[16];
    unsigned u[16];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) { f[i] = a * (float)(i + threadIdx.x); u[i] = threadIdx.x + i; }
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{f[i], f[i + 8]};
    const unsigned laddr = (unsigned)(size_t)(lds) + threadIdx.x * 16;
    f32x4 sink[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f32x4 wdat = f32x4{a, b, a, b};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 20, 0x00020000);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4 * NACC; ++r) {
            float av = a;
            if (KIND == 5) av = f[r & 15];
            acc[r / 4] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[r / 4], 0, 0, 0);
            if ((r % EVERY) == EVERY - 1) {
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                const int j = (r * NF + k) & 15;
                if (KIND == 0 || KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[j]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[(j + 1) & 15]));
                if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7]) : "v"(p[(j + 1) & 7]), "v"(p[(j + 2) & 7]));
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(sink[k & 3]) : "v"(laddr));
                if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(u[j]) : "v"(u[(j + 1) & 15]));
                if (KIND == 6) asm volatile("ds_write_b128 %0, %1" :: "v"(laddr), "v"(wdat) : "memory");
                if (KIND == 7) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(sink[k & 3]) : "v"(laddr & 0xffff0u), "s"(rsrc));
                if (KIND == 8) asm volatile("s_nop 0");
                if (KIND == 9) asm volatile("s_add_u32 %0, %0, 1" : "+s"(iters));
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 3 || KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (KIND == 9) iters -= 4 * NACC / EVERY * NF;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i] + (float)u[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    if (KIND == 3 || KIND == 7) s += sink[0][0] + sink[1][1] + sink[2][2] + sink[3][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NF, int KIND, int EVERY = 1>
void run(const char* what, int blocks, int per_cu, float* out, unsigned long long* clk) {
    constexpr int NACC = 3;
    const int iters = 4000;
    filler_loop<NF, KIND, NACC, EVERY><<<blocks, 256>>>(out, clk, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    filler_loop<NF, KIND, NACC, EVERY><<<blocks, 256>>>(out, clk, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    // cycles of SIMD time per MFMA: a wave's loop cycles / its MFMAs / waves sharing the SIMD
    const double per = (double)h[0] / ((double)iters * 4 * NACC) / per_cu;
    const double nf = (double)NF / EVERY;
    printf("%d wave(s)/SIMD  %-18s %2d after every %d MFMA: %7.1f SIMD cycles per MFMA (%+6.1f over 64.0; %5.1f per filler), %.2f ms\n", per_cu, what, NF, EVERY, per, per - 64.0,
           nf > 0 ? (per - 64.0) / nf : 0.0, ms);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    float* out;
    unsigned long long* clk;
    hipMalloc(&out, (size_t)prop.multiProcessorCount * 4 * 256 * sizeof(float) + (2 << 20));
    hipMalloc(&clk, (size_t)prop.multiProcessorCount * 4 * 8);
    printf("%s, %d CUs, chains of 4 v_mfma_f32_32x32x2_f32 on 3 accumulators; fillers pinned by sched_barrier\n", prop.gcnArchName, prop.multiProcessorCount);
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int blocks = prop.multiProcessorCount * per_cu;      // 256-thread blocks: one wave per SIMD each
        run<0, 0>("none", blocks, per_cu, out, clk);
        run<1, 0>("v_fma_f32", blocks, per_cu, out, clk);
        run<2, 0>("v_fma_f32", blocks, per_cu, out, clk);
        run<4, 0>("v_fma_f32", blocks, per_cu, out, clk);
        run<8, 0>("v_fma_f32", blocks, per_cu, out, clk);
        run<16, 0>("v_fma_f32", blocks, per_cu, out, clk);
        run<16, 0, 4>("v_fma_f32", blocks, per_cu, out, clk);
        run<48, 0, 12>("v_fma_f32", blocks, per_cu, out, clk);
        run<4, 5>("v_fma->mfma operand", blocks, per_cu, out, clk);
        run<4, 1>("v_add_u32", blocks, per_cu, out, clk);
        run<4, 2>("v_pk_fma_f32", blocks, per_cu, out, clk);
        run<8, 2>("v_pk_fma_f32", blocks, per_cu, out, clk);
        run<4, 4>("v_mov_b32", blocks, per_cu, out, clk);
        run<1, 3>("ds_read_b128", blocks, per_cu, out, clk);
        run<2, 3>("ds_read_b128", blocks, per_cu, out, clk);
        run<4, 3>("ds_read_b128", blocks, per_cu, out, clk);
        run<8, 3, 4>("ds_read_b128", blocks, per_cu, out, clk);
        run<1, 6>("ds_write_b128", blocks, per_cu, out, clk);
        run<2, 6>("ds_write_b128", blocks, per_cu, out, clk);
        run<1, 7>("buffer_load_dwordx4", blocks, per_cu, out, clk);
        run<2, 7>("buffer_load_dwordx4", blocks, per_cu, out, clk);
        run<4, 8>("s_nop 0", blocks, per_cu, out, clk);
        run<4, 9>("s_add_u32", blocks, per_cu, out, clk);
    }
    return 0;
}
