// Development probe: effective shader clock under light vs heavy load (s_memtime vs the 100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_clock(long long* out, int n) {
  long long c0 = clock64(), w0 = wall_clock64();
  float a = threadIdx.x, b = 1.0001f;
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)a; }
}
__global__ void k_heavy(float* out, int n) {
  float a = threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < n; ++i) { a = a * b + b; c = c * b + a; d = d * b + c; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + c + d;
}
int main() {
  long long* d; hipMalloc(&d, 64); float* o; hipMalloc(&o, 256 * 8 * 256 * 4 * 4);
  long long h[3];
  auto measure = [&](const char* tag, int blocks) {
    hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(64), 0, 0, d, 200);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-40s shader cycles %8lld wall ticks %6lld -> %.0f MHz (wall clock 100 MHz)  %.2f ns per dependent v_fma\n", tag, h[0], h[1],
           (double)h[0] / (double)h[1] * 100.0, (double)h[1] * 10.0 / (200.0 * 64));
  };
  measure("cold, 1 block", 1);
  for (int i = 0; i < 5; ++i) measure("light, back-to-back, 16 blocks", 16);
  hipLaunchKernelGGL(k_heavy, dim3(256 * 8), dim3(256), 0, 0, o, 2000000);
  measure("right after 1 heavy kernel", 16);
  for (int r = 0; r < 3; ++r) {
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_clock, dim3(16), dim3(64), 0, 0, d, 20);
    measure("after 200 light kernels", 16);
  }
  hipDeviceSynchronize();
  return 0;
}
