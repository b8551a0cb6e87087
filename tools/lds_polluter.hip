// Development probe: co-runner kernels to run beside other kernels - (a) one that only scribbles over its own LDS, (b) one
// that occupies a chosen number of VGPRs per wave and keeps the matrix cores busy without touching memory.
//   hipcc -shared -fPIC --offload-arch=gfx950 -o libpollute.so lds_polluter.hip
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern "C" __global__ void k_pollute(float* sink, int lds_words, int iters, unsigned pattern) {
  extern __shared__ unsigned sm[];
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) sm[i] = pattern ^ (unsigned)(i * 2654435761u + it);
    __syncthreads();
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) acc += sm[(i * 7) % lds_words];
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = 1.f;
}
#define BURN(NAME, TOPREG)                                                                                      \
  extern "C" __global__ __launch_bounds__(256) void NAME(float* sink, int iters) {                              \
    extern __shared__ unsigned sm[];                                                                            \
    f32x16 acc[4];                                                                                              \
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;                                   \
    bf16x8 a, b;                                                                                                \
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }              \
    asm volatile("v_mov_b32 " TOPREG ", 0" ::: TOPREG);                                                         \
    for (int it = 0; it < iters; ++it) {                                                                        \
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);      \
      if ((it & 63) == 0) { sm[threadIdx.x] = it; __syncthreads(); }                                            \
    }                                                                                                           \
    float s = 0.f;                                                                                              \
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];                                    \
    if (s == 12345.f) sink[0] = s;                                                                              \
  }
BURN(k_burn216, "v215")
BURN(k_burn224, "v223")
BURN(k_burn232, "v231")
BURN(k_burn248, "v247")
BURN(k_burn128, "v127")
extern "C" __global__ __launch_bounds__(256) void k_valu224(float* sink, int iters) {
  extern __shared__ unsigned sm[];
  float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
  asm volatile("v_mov_b32 v223, 0" ::: "v223");
  for (int it = 0; it < iters * 8; ++it) { a = a * b + b; c = c * b + a; d = d * b + c; if ((it & 511) == 0) { sm[threadIdx.x] = it; __syncthreads(); } }
  if (a + c + d == 12345.f) sink[0] = a;
}
extern "C" int pollute(float* sink, int blocks, int threads, int lds_bytes, int iters, unsigned pattern, void* stream) {
  static bool set = false;
  if (!set) { (void)hipFuncSetAttribute((const void*)k_pollute, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
  hipLaunchKernelGGL(k_pollute, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, sink, lds_bytes / 4, iters, pattern);
  return (int)hipGetLastError();
}
extern "C" int burn(float* sink, int vgprs, int blocks, int lds_bytes, int iters, void* stream) {
  void (*k)(float*, int) = vgprs == 1224 ? k_valu224 : vgprs == 216 ? k_burn216 : vgprs == 224 ? k_burn224 : vgprs == 232 ? k_burn232 : vgprs == 248 ? k_burn248 : k_burn128;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, sink, iters);
  return (int)hipGetLastError();
}
