// Development probe: per-launch time of dependent chains of tiny kernels on one stream (eager and hipGraph).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty(float* p) {}
__global__ void k_touch(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
// one dependent global round trip per thread
__global__ void k_load1(const float* in, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i % n] = in[(i * 97) % n] + 1.f;
}
// `depth` dependent round trips (pointer chase through an index array)
__global__ void k_chase(const int* idx, float* out, int depth) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = i;
  for (int d = 0; d < depth; ++d) j = idx[j];
  out[i] = (float)j;
}
// LDS + barrier + reduce, no global loads except store
__global__ void k_lds(float* out) {
  __shared__ float s[512];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = 0;
  for (int i = 0; i < 8; ++i) a += s[(threadIdx.x + i * 64) & 511];
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

// straight-line code of growing size vs the same work in a loop: cost of instruction fetch on a cold I-cache
template <int N>
__global__ void k_straight(float* out) {
  float a = threadIdx.x, b = 1.0001f;
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ void k_loop(float* out, int n) {
  float a = threadIdx.x, b = 1.0001f;
#pragma unroll 1
  for (int i = 0; i < n; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <typename F>
float run(const char* name, F launch, hipStream_t s, int n) {
  for (int i = 0; i < 20; ++i) launch();
  hipStreamSynchronize(s);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, s);
  for (int i = 0; i < n; ++i) launch();
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // graph
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 200; ++i) launch();
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s); hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
  hipEventRecord(b, s);
  hipEventSynchronize(b);
  float msg; hipEventElapsedTime(&msg, a, b);
  printf("%-28s eager %6.2f us/launch   graph %6.2f us/kernel\n", name, ms * 1000 / n, msg * 1000 / 2000);
  return ms;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 1 << 20;
  float *a, *b; int* idx;
  CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&idx, N * 4));
  std::vector<int> h(N);
  for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 7919 + 12345) % N);
  CK(hipMemcpy(idx, h.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(a, 0, N * 4));
  const int n = 2000;
  run("empty 16x512", [&] { hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 0, s, a); }, s, n);
  run("empty 16x256", [&] { hipLaunchKernelGGL(k_empty, dim3(16), dim3(256), 0, s, a); }, s, n);
  run("empty 274x512", [&] { hipLaunchKernelGGL(k_empty, dim3(274), dim3(512), 0, s, a); }, s, n);
  run("empty 16x512 lds32K", [&] { hipLaunchKernelGGL(k_empty, dim3(16), dim3(512), 33792, s, a); }, s, n);
  run("touch", [&] { hipLaunchKernelGGL(k_touch, dim3(16), dim3(512), 0, s, a); }, s, n);
  run("load1 16x512", [&] { hipLaunchKernelGGL(k_load1, dim3(16), dim3(512), 0, s, a, b, N); }, s, n);
  run("chase depth1 16x512", [&] { hipLaunchKernelGGL(k_chase, dim3(16), dim3(512), 0, s, idx, b, 1); }, s, n);
  run("chase depth2 16x512", [&] { hipLaunchKernelGGL(k_chase, dim3(16), dim3(512), 0, s, idx, b, 2); }, s, n);
  run("chase depth4 16x512", [&] { hipLaunchKernelGGL(k_chase, dim3(16), dim3(512), 0, s, idx, b, 4); }, s, n);
  run("chase depth8 16x512", [&] { hipLaunchKernelGGL(k_chase, dim3(16), dim3(512), 0, s, idx, b, 8); }, s, n);
  run("lds+barrier 16x512", [&] { hipLaunchKernelGGL(k_lds, dim3(16), dim3(512), 0, s, b); }, s, n);
  run("straight 256 fma (2KB)", [&] { hipLaunchKernelGGL(k_straight<256>, dim3(16), dim3(64), 0, s, b); }, s, n);
  run("straight 1024 fma (8KB)", [&] { hipLaunchKernelGGL(k_straight<1024>, dim3(16), dim3(64), 0, s, b); }, s, n);
  run("straight 2048 fma (16KB)", [&] { hipLaunchKernelGGL(k_straight<2048>, dim3(16), dim3(64), 0, s, b); }, s, n);
  run("straight 4096 fma (32KB)", [&] { hipLaunchKernelGGL(k_straight<4096>, dim3(16), dim3(64), 0, s, b); }, s, n);
  run("loop 1024 fma", [&] { hipLaunchKernelGGL(k_loop, dim3(16), dim3(64), 0, s, b, 1024); }, s, n);
  run("loop 4096 fma", [&] { hipLaunchKernelGGL(k_loop, dim3(16), dim3(64), 0, s, b, 4096); }, s, n);
  // alternate two big kernels (evicting each other?)
  run("alternate 2x straight 2048", [&] { hipLaunchKernelGGL(k_straight<2048>, dim3(16), dim3(64), 0, s, b); hipLaunchKernelGGL(k_straight<2047>, dim3(16), dim3(64), 0, s, b); }, s, n / 2);
  return 0;
}
