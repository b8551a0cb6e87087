// Development probe: how fast is the matrix pipe on the instruction mix of the fp16 conv tier (two fp16 MFMAs per
// f32 product) vs a mix where the lo product runs on the MX-scaled fp8 / fp4 MFMA (K = 64 per instruction)?
// 8 independent accumulators per wave (the conv kernel's MW = 8), 2 workgroups of 4 waves per CU, ~0.3 s per variant
// so that the clock settles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  f32x16 acc[8];
  for (int m = 0; m < 8; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  f16x8 a[8], bh, bl;
  i32x8 a8[8], b8;
  for (int e = 0; e < 8; ++e) { bh[e] = (_Float16)(0.001f * (threadIdx.x + e)); bl[e] = (_Float16)(0.0001f * e); b8[e] = 0x38383838; }
  for (int m = 0; m < 8; ++m)
    for (int e = 0; e < 8; ++e) { a[m][e] = (_Float16)(0.01f * (m + e + (threadIdx.x & 7))); a8[m][e] = 0x38383838 + m; }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    // one "64-channel tap" of 8 pixel tiles: 4 k-steps of 16
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], bh, acc[m], 0, 0, 0);
    if (MODE == 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], bl, acc[m], 0, 0, 0);
    } else if (MODE == 1) {   // fp8 e4m3 x fp8 e4m3, scales 2^0
#pragma unroll
      for (int m = 0; m < 8; ++m)
        acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[m], b8, acc[m], 0, 0, 0, 127, 0, 127);
    } else if (MODE == 2) {   // fp4 x fp4
#pragma unroll
      for (int m = 0; m < 8; ++m)
        acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[m], b8, acc[m], 4, 4, 0, 127, 0, 127);
    } else if (MODE == 3) {   // fp6 x fp6
#pragma unroll
      for (int m = 0; m < 8; ++m)
        acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[m], b8, acc[m], 2, 2, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
  for (int m = 0; m < 8; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* tag, float* o) {
  const int blocks = 512, iters = 60000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // algorithmic work per iteration per wave: 8 tiles x 32x32 outputs x 64 k x 2 flop (one f32-grade product)
  const double flop = (double)blocks * 4 * iters * 8.0 * 32 * 32 * 64 * 2;
  printf("%-34s %8.1f ms  %7.1f TFLOP/s algorithmic (hi + lo product = one f32-grade product)\n", tag, ms, flop / ms / 1e9);
}
int main() {
  float* o; hipMalloc(&o, 512 * 256 * 4);
  run<0>("fp16 hi + fp16 lo (today)", o);
  run<1>("fp16 hi + MX fp8 lo", o);
  run<3>("fp16 hi + MX fp6 lo", o);
  run<2>("fp16 hi + MX fp4 lo", o);
  run<0>("fp16 hi + fp16 lo (again)", o);
  return 0;
}
