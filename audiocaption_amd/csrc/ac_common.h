// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the audio-captioning path.
// Wave size is 64 everywhere; nothing here compiles for another target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AC_WAVE 64

// Error codes of the C ABI (include/audiocaption_hip.h).
#define AC_OK 0
#define AC_ERR_ARG (-1)
#define AC_ERR_LAUNCH (-2)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact f32 (k-ordered fmaf chain).
// Lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the result register r
// of lane l is D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int ac_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AC_OK : AC_ERR_LAUNCH;
}
