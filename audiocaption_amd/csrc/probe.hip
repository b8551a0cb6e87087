// Diagnostic, not on the hot path: the matrix rate this part SUSTAINS on the conv tier's instruction
// (v_mfma_f32_32x32x16_bf16, 8 independent accumulators per wave, two 4-wave workgroups per CU, no memory traffic).
// bench.py runs it next to the timed region so that `roofline` can say how much of the gap to the nominal 2.5 PFLOP/s is
// the clock the part holds while its matrix pipes are busy (~1.7 of 2.4 GHz) and how much is the kernel.
#include "ac_common.h"

namespace {

typedef __bf16 pb_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256, 2) void mfma_bf16_probe_kernel(float* out, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  pb_bf16x8 a[8], b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.001f * (float)((threadIdx.x + e) & 31));
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[m][e] = (__bf16)(0.01f * (float)(m + e + (threadIdx.x & 7)));
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b, acc[m], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_mfma_bf16_probe(float* out, int blocks, int iters, void* stream) {
  if (!out || blocks <= 0 || iters <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return ac_check_launch();
}
