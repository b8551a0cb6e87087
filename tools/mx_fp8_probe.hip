// Development probe: operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands.
// Hypothesis: lane l supplies row/col (l & 31) and the 32 consecutive k of block (l >> 5), bytes in k order inside the
// 8 VGPRs; scale_a / scale_b: E8M0 (value 2^(s - 127)) for the lane's 32-k block, byte selected by opsel.
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int HYP>
__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D) {
  // A [32 rows][64 k] bytes, B [32 cols][64 k] bytes (i.e. B^T), sa [32 rows][2 blocks], sb [32 cols][2 blocks]
  const int l = threadIdx.x, r = l & 31, kb = l >> 5;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    // HYP 0: lane half kb holds k = 32 kb + 4 i + byte;  HYP 1: k = 16 kb + 32 (i / 4) + 4 (i % 4) + byte
    const int k0 = HYP == 0 ? kb * 32 + 4 * i : 16 * kb + 32 * (i / 4) + 4 * (i % 4);
    a[i] = *(const int*)(A + r * 64 + k0);
    b[i] = *(const int*)(B + r * 64 + k0);
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int scale_a = sa[r * 2 + kb], scale_b = sb[r * 2 + kb];
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
    D[row * 32 + col] = c[i];
  }
}

static float e4m3_to_float(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + (float)m / 8.f, e - 7);
  if (e == 15 && m == 7) x = NAN;
  return s ? -x : x;
}

int main() {
  unsigned char hA[32 * 64], hB[32 * 64], hsa[64], hsb[64];
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) {
    do { hA[i] = rand() & 0xff; } while ((hA[i] & 0x7f) == 0x7f);
    do { hB[i] = rand() & 0xff; } while ((hB[i] & 0x7f) == 0x7f);
  }
  for (int i = 0; i < 64; ++i) { hsa[i] = 127 + (rand() % 7) - 3; hsb[i] = 127 + (rand() % 5) - 2; }
  unsigned char *dA, *dB, *dsa, *dsb; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
  float hD[32 * 32];
  for (int hyp = 0; hyp < 2; ++hyp) {
  if (hyp == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
  else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double worst = 0, big = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int kk = 0; kk < 64; ++kk)
        ref += (double)e4m3_to_float(hA[i * 64 + kk]) * ldexp(1.0, hsa[i * 2 + kk / 32] - 127) *
               (double)e4m3_to_float(hB[j * 64 + kk]) * ldexp(1.0, hsb[j * 2 + kk / 32] - 127);
      worst = fmax(worst, fabs(ref - hD[i * 32 + j]));
      big = fmax(big, fabs(ref));
    }
  printf("hypothesis %d: max |ref - mfma| = %.4g (max |ref| %.4g) -> %s\n", hyp, worst, big, worst < 1e-3 * big ? "LAYOUT HYPOTHESIS HOLDS" : "MISMATCH");
  }
  printf("D[0][0..3] = %g %g %g %g\n", hD[0], hD[1], hD[2], hD[3]);
  return 0;
}
