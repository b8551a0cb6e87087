// Y[M,N] = act(X[M,K] * W[N,K]^T + bias) on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).
// Both operands are K-contiguous (PyTorch nn.Linear layout), so every torch F.linear on the hot path
// (GRU input projections, attn_proj, cross-attention K/V, per-step decoder projections, classifier;
// reference rnn_encoder.py:41, transformer_decoder.py:86-101) maps to this file.
//
// Two shapes of the same contraction:
//  * gemm_tiled:  64x64 block tile, 4 waves (one 32x32 MFMA tile each), operands staged through LDS
//                 in 32-deep K slabs - for M in the hundreds/thousands (whole-sequence projections).
//  * gemm_skinny: 32x32 block tile, the 4 waves split K and reduce through LDS - for the decode
//                 step (M = batch <= 128) where latency, not throughput, is what matters.
// K-order trick shared with the conv kernel: within a group of 8 consecutive k, lanes 0-31 feed
// k = 0..3 and lanes 32-63 feed k = 4..7 over four MFMAs, so each operand fragment is ONE 16-byte
// read per lane per four MFMAs.
#include "ac_common.h"

namespace {

constexpr int LDS_STRIDE = 36;

struct GemmParams {
  const float* X;
  const float* W;
  const float* bias;
  float* Y;
  int M, N, K;
  long ldx, ldw, ldy;
  int relu;
};

__global__ __launch_bounds__(256) void gemm_tiled_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float sA[64 * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sB[64 * LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int abase = (wm * 32 + (lane & 31)) * LDS_STRIDE + half * 4;
  const int bbase = (wn * 32 + (lane & 31)) * LDS_STRIDE + half * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += 32) {
    float4 ra[2], rb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
      ra[u] = (m0 + row < p.M) ? *(const float4*)(p.X + (size_t)(m0 + row) * p.ldx + k0 + c4 * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[u] = (n0 + row < p.N) ? *(const float4*)(p.W + (size_t)(n0 + row) * p.ldw + k0 + c4 * 4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();  // previous slab fully consumed
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
      *(float4*)(sA + row * LDS_STRIDE + c4 * 4) = ra[u];
      *(float4*)(sB + row * LDS_STRIDE + c4 * 4) = rb[u];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 a = *(const f32x4*)(sA + abase + g * 8);
      const f32x4 b = *(const f32x4*)(sB + bbase + g * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma32(a[s], b[s], acc);
    }
  }
  const int gn = n0 + wn * 32 + (lane & 31);
  if (gn < p.N) {
    const float bv = p.bias ? p.bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (gm < p.M) {
        float y = acc[r] + bv;
        if (p.relu) y = fmaxf(y, 0.f);
        p.Y[(size_t)gm * p.ldy + gn] = y;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmParams p) {
  __shared__ float red[4][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int kw = p.K >> 2;  // K per wave (K % 32 == 0 so kw % 8 == 0)
  const int gm_l = m0 + (lane & 31), gn_l = n0 + (lane & 31);
  const bool mv = gm_l < p.M, nv = gn_l < p.N;
  const float* xa = p.X + (size_t)(mv ? gm_l : 0) * p.ldx + wave * kw + half * 4;
  const float* wb = p.W + (size_t)(nv ? gn_l : 0) * p.ldw + wave * kw + half * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
  for (int k = 0; k < kw; k += 8) {
    f32x4 a = *(const f32x4*)(xa + k);
    f32x4 b = *(const f32x4*)(wb + k);
    if (!mv) a = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!nv) b = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(a[s], b[s], acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
    red[wave][i * 33 + (lane & 31)] = acc[r];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = tid + u * 256, i = e >> 5, j = e & 31;
    const int gm = m0 + i, gn = n0 + j;
    if (gm < p.M && gn < p.N) {
      float y = (red[0][i * 33 + j] + red[1][i * 33 + j]) + (red[2][i * 33 + j] + red[3][i * 33 + j]);
      if (p.bias) y += p.bias[gn];
      if (p.relu) y = fmaxf(y, 0.f);
      p.Y[(size_t)gm * p.ldy + gn] = y;
    }
  }
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_linear(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                         long ldx, long ldw, long ldy, int relu, void* stream) {
  if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return AC_ERR_ARG;
  if (K % 32 || ldx % 4 || ldw % 4) return AC_ERR_ARG;
  if (((uintptr_t)X & 15) || ((uintptr_t)W & 15)) return AC_ERR_ARG;
  GemmParams p;
  p.X = X; p.W = W; p.bias = bias; p.Y = Y; p.M = M; p.N = N; p.K = K;
  p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  if (M <= 128) {
    dim3 grid((N + 31) / 32, (M + 31) / 32);
    hipLaunchKernelGGL(gemm_skinny_kernel, grid, dim3(256), 0, s, p);
  } else {
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    hipLaunchKernelGGL(gemm_tiled_kernel, grid, dim3(256), 0, s, p);
  }
  return ac_check_launch();
}
