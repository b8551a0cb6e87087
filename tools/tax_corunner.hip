// Development probe (tools/tax_probe.py): co-runner kernels with a CHOSEN register / LDS footprint that stay resident for `spin`
// shader cycles - asleep, issuing vector FMAs, or streaming a buffer - to find what lets a workgroup share a CU with the
// one-wave-per-SIMD conv kernels (448 of a SIMD's 512 registers per lane, 83-95 KB of the 160 KB LDS).
#include <hip/hip_runtime.h>
#define SPIN(NAME, TOPREG)                                                                                       \
  extern "C" __global__ __launch_bounds__(256) void NAME(float* buf, int spin, int mode) {                       \
    extern __shared__ unsigned sm[];                                                                             \
    asm volatile("v_mov_b32 " TOPREG ", 0" ::: TOPREG);                                                          \
    const long long t0 = clock64();                                                                              \
    float a = threadIdx.x, b = 1.0001f, c = 0.5f;                                                                \
    const float4* src = (const float4*)buf + (size_t)blockIdx.x * 4096 + threadIdx.x;                            \
    float4 s4 = {0.f, 0.f, 0.f, 0.f};                                                                            \
    int it = 0;                                                                                                  \
    while (clock64() - t0 < spin) {                                                                              \
      if (mode == 0) __builtin_amdgcn_s_sleep(8);                                                                \
      else if (mode == 1) { for (int i = 0; i < 64; ++i) { a = a * b + c; c = c * b + a; } }                     \
      else if (mode == 2) { const float4 v = src[(it & 15) * 256]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; ++it; }  \
      else {  /* 3: weight streaming - 8 independent 16-byte loads per thread in flight over a 1 MB window every workgroup shares */ \
        const float4* w = (const float4*)buf + threadIdx.x + ((it * 8) & 255) * 256;                              \
        float4 v[8];                                                                                             \
        for (int u = 0; u < 8; ++u) v[u] = w[u * 256];                                                           \
        for (int u = 0; u < 8; ++u) { s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w; }           \
        ++it;                                                                                                    \
      }                                                                                                          \
    }                                                                                                            \
    if (a + c + s4.x + s4.y + s4.z + s4.w == 12345.f) buf[0] = a;                                                \
    if (spin < 0) sm[threadIdx.x] = 1;                                                                           \
  }
SPIN(k_spin24, "v23")
SPIN(k_spin56, "v55")
SPIN(k_spin64, "v63")
SPIN(k_spin72, "v71")
SPIN(k_spin96, "v95")
SPIN(k_spin128, "v127")
extern "C" int corun(float* buf, int vgprs, int blocks, int threads, int lds_bytes, int spin, int mode, void* stream) {
  void (*k)(float*, int, int) = vgprs <= 24 ? k_spin24 : vgprs <= 56 ? k_spin56 : vgprs <= 64 ? k_spin64 : vgprs <= 72 ? k_spin72 : vgprs <= 96 ? k_spin96 : k_spin128;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, buf, spin, mode);
  return (int)hipGetLastError();
}
