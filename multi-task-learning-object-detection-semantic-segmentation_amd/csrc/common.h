// Internal helpers shared by the HIP translation units of libssdseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ssdseg.h"

struct ssdseg_timing;  // per-kernel HIP-event timing registry (runtime.hip)

struct ssdseg_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;
    void* workspace;
    size_t workspace_bytes;
    size_t ws_reserved;   // front part of the workspace held by a composite entry point while it calls others (runtime.hip)
    int num_cus;
    ssdseg_timing* timing;  // non-null while kernel timing is enabled
    // bytes of the SECOND tensor of a BatchNorm-backward gradient view (the raw forward output y next to g) read by the next
    // launch: not part of SURVEY.md 8(d)'s algorithmic bytes (read X, read dY, write dX), reported separately as `view_bytes`.
    // Set by a launcher right before SSDSEG_LAUNCH*, consumed (and cleared) by ssdseg_timing_begin.
    double timing_view_bytes;
    // Side stream for work that is off the critical path of the backward pass (weight gradients: nothing reads dW before the
    // optimizer).  ssdseg_side_begin() makes the side stream wait for everything queued so far and redirects `stream` /
    // `workspace` to it; ssdseg_side_end() restores them; ssdseg_join() makes the main stream wait for the side work and is
    // called by every entry point that synchronises, copies or reads gradients.
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
    hipEvent_t ev_mark;   // ssdseg_ctx_side_mark / _wait_mark: a point on the side stream the main stream can wait for without joining it
    bool mark_set;
    void* side_workspace;
    size_t side_workspace_bytes;
    bool side_ok, side_on, side_pending;
    // copy stream (created on first use): host -> device staging uploads that overlap the running step (runtime.hip)
    hipStream_t copy_stream;
    hipEvent_t ev_copy_fork, ev_copy_join;
    // RCCL communicator of this rank (comm.hip; ncclComm_t behind a void* so that only comm.hip needs the RCCL header)
    void* comm;
    int comm_rank, comm_world;
    // deferred column sums of weight-gradient partial slabs (bn.hip: ssdseg_colsum_defer / ssdseg_colsum_flush)
    struct ssdseg_defer* defer;
};

extern "C" {
bool ssdseg_side_begin(ssdseg_ctx* ctx);   // true when launches are now redirected to the side stream
void ssdseg_side_end(ssdseg_ctx* ctx);
int ssdseg_join(ssdseg_ctx* ctx);
}

// Brackets one kernel launch with HIP events on the ctx stream when timing is enabled (bench.py's roofline leg);
// `bytes`/`flops` are the ALGORITHMIC traffic / work of this launch (formulas: DESIGN.md "Kernels").
void ssdseg_timing_begin(ssdseg_ctx* ctx, const char* kernel, double bytes, double flops);
void ssdseg_timing_end(ssdseg_ctx* ctx);

// stable copy of a kernel-symbol string built at run time (template instances whose arguments are not literals)
const char* ssdseg_intern(const char* name);

#define SSDSEG_LAUNCH_NAMED(ctx, name, bytes, flops, kernel, grid, block, lds, ...)   \
    do {                                                                              \
        if ((ctx)->timing) ssdseg_timing_begin((ctx), (name), (bytes), (flops));      \
        hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);     \
        if ((ctx)->timing) ssdseg_timing_end((ctx));                                  \
    } while (0)

#define SSDSEG_LAUNCH(ctx, bytes, flops, kernel, grid, block, lds, ...)               \
    do {                                                                              \
        if ((ctx)->timing) ssdseg_timing_begin((ctx), #kernel, (bytes), (flops));     \
        hipLaunchKernelGGL(kernel, grid, block, lds, (ctx)->stream, __VA_ARGS__);     \
        if ((ctx)->timing) ssdseg_timing_end((ctx));                                  \
    } while (0)

void ssdseg_set_error(const char* fmt, ...);
int ssdseg_hip_fail(hipError_t e, const char* what);
// workspace of at least `bytes` (grows with hipMalloc)
int ssdseg_workspace(ssdseg_ctx* ctx, size_t bytes, void** out);
// storage for the partial slabs / rows a later ssdseg_colsum reads: the stream's workspace, or -- while column sums are deferred
// (ssdseg_colsum_defer) -- a persistent arena whose regions live until the flush (bn.hip)
int ssdseg_partials(ssdseg_ctx* ctx, size_t bytes, void** out);
// launches the pending column sums of a deferred backward pass as ONE kernel on the ctx stream (called by ssdseg_join once the
// side stream has been joined); no-op when nothing is pending
int ssdseg_colsum_flush(ssdseg_ctx* ctx);
void ssdseg_defer_destroy(ssdseg_ctx* ctx);
// A composite entry point whose nested weight-gradient call writes a SCRATCH result it consumes right away (the narrow 3x3 conv's
// tap-expanded dW2, repacked into dW by the next kernel) brackets that call with hold(+1) / hold(-1): column sums recorded in
// between are launched at once, as without deferral.
void ssdseg_defer_hold(ssdseg_ctx* ctx, int delta);

#define SSDSEG_HIP(call)                                      \
    do {                                                      \
        hipError_t _e = (call);                               \
        if (_e != hipSuccess) return ssdseg_hip_fail(_e, #call); \
    } while (0)

#define SSDSEG_ARG(cond, n)                                               \
    do {                                                                  \
        if (!(cond)) {                                                    \
            ssdseg_set_error("%s: invalid argument %d (%s)", __func__, (n), #cond); \
            return SSDSEG_EINVAL(n);                                      \
        }                                                                 \
    } while (0)

#define SSDSEG_LAUNCH_CHECK()                                   \
    do {                                                        \
        hipError_t _e = hipGetLastError();                      \
        if (_e != hipSuccess) return ssdseg_hip_fail(_e, __func__); \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// TF "SAME" geometry
static inline void same_pad(int in, int k, int s, int d, int* out, int* before) {
    int o = (in + s - 1) / s;
    int keff = (k - 1) * d + 1;
    int total = (o - 1) * s + keff - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Keras ReLU(max_value) (App. B.4) as a clamp to [lo, hi] -- branch-free on the (wave-uniform) activation code:
//   NONE [-inf, +inf] | RELU [0, +inf] | RELU6 [0, 6] | ZERO (max_value = 0.0, quirk Q1) [0, 0]
__device__ __forceinline__ float act_lo(int act) { return act == SSDSEG_ACT_NONE ? -INFINITY : 0.f; }
__device__ __forceinline__ float act_hi(int act) { return act == SSDSEG_ACT_RELU6 ? 6.f : (act == SSDSEG_ACT_ZERO ? 0.f : INFINITY); }
__device__ __forceinline__ float act_apply(float z, int act) { return fminf(fmaxf(z, act_lo(act)), act_hi(act)); }
// derivative of the activation at pre-activation z: 1 strictly inside (lo, hi), else 0
__device__ __forceinline__ float act_mask(float z, int act) { return (z > act_lo(act) && z < act_hi(act)) ? 1.f : 0.f; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }

// a = act(scale*x + shift) on a 4-channel vector; has_affine == false -> act(x)
__device__ __forceinline__ float4 view_apply4(float4 x, float4 s, float4 t, bool has_affine, int act) {
    // identity affine instead of a branch (fma(1, x, 0) == x exactly)
    if (!has_affine) { s = make_float4(1.f, 1.f, 1.f, 1.f); t = make_float4(0.f, 0.f, 0.f, 0.f); }
    x.x = fmaf(s.x, x.x, t.x); x.y = fmaf(s.y, x.y, t.y); x.z = fmaf(s.z, x.z, t.z); x.w = fmaf(s.w, x.w, t.w);
    x.x = act_apply(x.x, act); x.y = act_apply(x.y, act); x.z = act_apply(x.z, act); x.w = act_apply(x.w, act);
    return x;
}

// same with the affine always applied (callers substitute scale = 1, shift = 0 for identity views at set-up time, so the
// per-element path has no branch at all)
// (two packed fmas + four v_med3 = 6 vector instructions; as fminf(fmaxf(fmaf())) per component the compiler emits 12 -- and in
// the MFMA kernels every vector instruction of the staging path takes an issue slot from the matrix pipe.  med3(v, lo, hi) is
// the clamp for lo <= hi; built with -ffp-contract=fast, s * x + t on the 2-vectors IS the fused multiply-add)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 view_affine4(float4 x, float4 s, float4 t, float lo, float hi) {
    const f32x2 a = f32x2{s.x, s.y} * f32x2{x.x, x.y} + f32x2{t.x, t.y};
    const f32x2 b = f32x2{s.z, s.w} * f32x2{x.z, x.w} + f32x2{t.z, t.w};
    return make_float4(__builtin_amdgcn_fmed3f(a.x, lo, hi), __builtin_amdgcn_fmed3f(a.y, lo, hi), __builtin_amdgcn_fmed3f(b.x, lo, hi),
                       __builtin_amdgcn_fmed3f(b.y, lo, hi));
}

// dy = s*mask(s*y+t)*g + k1*y + k0 on a 4-channel vector.  Identity gradient views are expressed as s = 1, t = k1 = k0 = 0,
// act = NONE and y aliased to g (mask == 1 for finite values), so there is no per-element branch either.
// (packed: z = s*y + t, base = k1*y + k0 and the final fma are three pairs of v_pk_fma; the mask selects s or 0 per component)
__device__ __forceinline__ float4 gview_apply4(float4 g, float4 y, float4 s, float4 t, float4 k1, float4 k0, int act) {
    const float lo = act_lo(act), hi = act_hi(act);
    const f32x2 z0 = f32x2{s.x, s.y} * f32x2{y.x, y.y} + f32x2{t.x, t.y}, z1 = f32x2{s.z, s.w} * f32x2{y.z, y.w} + f32x2{t.z, t.w};
    const f32x2 b0 = f32x2{k1.x, k1.y} * f32x2{y.x, y.y} + f32x2{k0.x, k0.y}, b1 = f32x2{k1.z, k1.w} * f32x2{y.z, y.w} + f32x2{k0.z, k0.w};
    const f32x2 m0 = f32x2{(z0.x > lo && z0.x < hi) ? s.x : 0.f, (z0.y > lo && z0.y < hi) ? s.y : 0.f};
    const f32x2 m1 = f32x2{(z1.x > lo && z1.x < hi) ? s.z : 0.f, (z1.y > lo && z1.y < hi) ? s.w : 0.f};
    const f32x2 r0 = m0 * f32x2{g.x, g.y} + b0, r1 = m1 * f32x2{g.z, g.w} + b1;
    return make_float4(r0.x, r0.y, r1.x, r1.y);
}

// XCD-aware block -> work mapping.  Workgroups are handed to the 8 XCDs round-robin in launch order and every XCD has its
// own 4 MiB L2, so blocks b, b+1 (neighbouring tiles) never share an L2 and a 3x3 stencil's halo rows are fetched from HBM
// once per XCD (measured with FETCH_SIZE: 5x the algorithmic bytes on the register-window depthwise backward).  Remapped,
// XCD k walks the contiguous range [k*total/8, (k+1)*total/8) of the (x fastest, then y) logical grid, so spatial neighbours
// are resident on the same L2 at the same time.  A speed-only affinity: any mapping is correct.  Needs total % 8 == 0
// (launchers round grid.x up to a multiple of 8), otherwise the identity.
struct BlockPos {
    int x, y;
};
__device__ __forceinline__ BlockPos xcd_block_pos() {
    BlockPos p;
    const unsigned total = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    unsigned logical = lin;
    if ((total & 7u) == 0u) logical = (lin & 7u) * (total >> 3) + (lin >> 3);
    p.y = (int)(logical / gridDim.x);
    p.x = (int)(logical - (unsigned)p.y * gridDim.x);
    return p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif
