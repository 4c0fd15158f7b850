// compile-only harness for conv3_wino4.h (register / spill / ISA inspection in seconds instead of the minute gemm.hip takes):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -I include -I <csrc> -c scripts/dbg/wino4_standalone.hip -o /tmp/w4.o -save-temps
#include "common.h"
namespace {
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
template <int V> struct wino_const { static constexpr int value = V; };
#include "conv3_wino4.h"
}
void wino4_standalone_launch(Wino4Args p, hipStream_t st) {
    hipLaunchKernelGGL(conv3_wino4_kernel<false>, dim3(1), dim3(W4_THREADS), 0, st, p);
    hipLaunchKernelGGL(conv3_wino4_kernel<true>, dim3(1), dim3(W4_THREADS), 0, st, p);
}
