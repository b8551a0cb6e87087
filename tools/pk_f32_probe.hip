// Development probe (standalone): do packed-f32 VALU instructions give the right answer while ANOTHER kernel keeps the
// matrix cores of the same SIMDs busy?   hipcc --offload-arch=gfx950 -O2 -o pk_probe pk_f32_probe.hip && ./pk_probe
// Found while chasing decode results that changed beside the F(2,3) conv kernels (csrc/decoder.hip, row_gemv256).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_burn(float* sink, int iters) {
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)((threadIdx.x + e) & 7); b[e] = (__bf16)1.0f; }
  asm volatile("v_mov_b32 v223, 0" ::: "v223");
  for (int it = 0; it < iters; ++it)
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
  float s = 0.f;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 12345.f) sink[0] = s;
}

// MODE 0: v_pk_fma_f32 (both halves from registers)   1: v_pk_fma_f32 with the scalar operand broadcast (op_sel_hi 1,0,1)
//      2: two v_fma_f32                                3: v_pk_add_f32      4: v_pk_mul_f32
// every thread runs a chain whose exact result is known: acc += w * x with w, x small integers -> exact in f32
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_test(const float* w, unsigned* bad, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f32x2 wv = {w[(tid * 2) & 1023], w[(tid * 2 + 1) & 1023]};
  unsigned errs = 0;
  for (int rep = 0; rep < iters; ++rep) {
    f32x2 acc = {0.f, 0.f};
    float ref0 = 0.f, ref1 = 0.f;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) {
      const float x = (float)((k * 7 + rep) & 15);
      f32x2 xx = {x, x};
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(wv), "v"(xx));
      else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(wv), "v"(xx));
      else if (MODE == 2) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(wv[0]), "v"(x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[1]) : "v"(wv[1]), "v"(x)); }
      else if (MODE == 3) { f32x2 p = {wv[0] * x, wv[1] * x}; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p)); }
      else { f32x2 p; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(wv), "v"(xx)); acc[0] += p[0]; acc[1] += p[1]; }
      ref0 += wv[0] * x; ref1 += wv[1] * x;     // integers < 2^24: exact whatever the instruction
    }
    // the reference chain above is compiled to ordinary VALU code of the same wave: compare against a closed form too
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += (float)((k * 7 + rep) & 15);
    if (acc[0] != wv[0] * s || acc[1] != wv[1] * s) ++errs;
  }
  if (errs) atomicAdd(bad, errs);
}

template <int MODE>
void run(const char* name, const float* w, unsigned* bad, float* sink, hipStream_t s1, hipStream_t s2, bool with_burner) {
  hipMemsetAsync(bad, 0, 4, s1);
  hipStreamSynchronize(s1);
  for (int r = 0; r < 20; ++r) {
    if (with_burner) hipLaunchKernelGGL(k_burn, dim3(4096), dim3(256), 0, s2, sink, 3000);
    for (int j = 0; j < 20; ++j) hipLaunchKernelGGL(k_test<MODE>, dim3(64), dim3(256), 0, s1, w, bad, 200);
    hipDeviceSynchronize();
  }
  unsigned h = 0;
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%-46s %s: wrong chains %u of %u\n", name, with_burner ? "beside the MFMA burner" : "alone                 ", h, 20u * 20u * 64u * 256u * 200u);
}

int main() {
  float* w; unsigned* bad; float* sink;
  hipMalloc(&w, 4096); hipMalloc(&bad, 4); hipMalloc(&sink, 64);
  float hw[1024];
  for (int i = 0; i < 1024; ++i) hw[i] = (float)((i * 37) % 61 - 30);
  hipMemcpy(w, hw, 4096, hipMemcpyHostToDevice);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  for (int b = 0; b < 2; ++b) {
    run<0>("v_pk_fma_f32", w, bad, sink, s1, s2, b);
    run<1>("v_pk_fma_f32 op_sel_hi:[1,0,1] (broadcast x)", w, bad, sink, s1, s2, b);
    run<2>("2 x v_fma_f32", w, bad, sink, s1, s2, b);
    run<3>("v_pk_add_f32", w, bad, sink, s1, s2, b);
    run<4>("v_pk_mul_f32", w, bad, sink, s1, s2, b);
  }
  return 0;
}
