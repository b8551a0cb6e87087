// Pieces shared by the F(4,3) split-bf16 conv kernels (csrc/conv3x3_wino43.hip, csrc/conv3x3_block1_w4.hip): the operand
// split, the input transform of a row quad, the output transform + BN + ReLU of one accumulator register.
#pragma once
#include "ac_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// x (4 floats) -> packed hi (2 dwords) and lo (2 dwords) bf16 quadruples: hi = RNE(x), lo = RNE(x - hi)
__device__ __forceinline__ void split_bf16x4(const f32x4 x, u32x2& hi, u32x2& lo) {
  hi.x = cvt_pk_bf16(x[0], x[1]);
  hi.y = cvt_pk_bf16(x[2], x[3]);
  const float h0 = __builtin_bit_cast(float, hi.x << 16), h1 = __builtin_bit_cast(float, hi.x & 0xffff0000u);
  const float h2 = __builtin_bit_cast(float, hi.y << 16), h3 = __builtin_bit_cast(float, hi.y & 0xffff0000u);
  lo.x = cvt_pk_bf16(x[0] - h0, x[1] - h1);
  lo.y = cvt_pk_bf16(x[2] - h2, x[3] - h3);
}

__device__ __forceinline__ f32x4 vfma(float a, const f32x4 x, const f32x4 y) {   // a * x + y, one rounding per element
  f32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = __builtin_fmaf(a, x[j], y[j]);
  return r;
}

// F(4,3) input transform: position pos (0..5) of the six rows d0..d5 of a quad (d[r] only touched where pos needs it)
__device__ __forceinline__ f32x4 w4_transform(int pos, const f32x4& d0, const f32x4& d1, const f32x4& d2, const f32x4& d3,
                                              const f32x4& d4, const f32x4& d5) {
  if (pos == 0) return vfma(4.f, d0, vfma(-5.f, d2, d4));
  if (pos == 5) return vfma(4.f, d1, vfma(-5.f, d3, d5));
  if (pos == 1 || pos == 2) {
    const f32x4 t1 = vfma(-4.f, d2, d4), t2 = vfma(-4.f, d1, d3);
    return pos == 1 ? t1 + t2 : t1 - t2;
  }
  const f32x4 t3 = d4 - d2, u = d3 - d1;
  return pos == 3 ? vfma(2.f, u, t3) : vfma(-2.f, u, t3);
}

// F(4,3) output transform of the six position sums of one (pixel, channel) + BatchNorm (scale, shift) + ReLU: four rows
__device__ __forceinline__ void w4_outputs(float m0, float m1, float m2, float m3, float m4, float m5, float sc, float sh,
                                           float (&y)[4]) {
  const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y[0] = fmaxf(fmaf((m0 + s12) + s34, sc, sh), 0.f);
  y[1] = fmaxf(fmaf(__builtin_fmaf(2.f, d34, d12), sc, sh), 0.f);
  y[2] = fmaxf(fmaf(__builtin_fmaf(4.f, s34, s12), sc, sh), 0.f);
  y[3] = fmaxf(fmaf(__builtin_fmaf(8.f, d34, d12) + m5, sc, sh), 0.f);
}

struct FastDiv4 {   // x mod d for 0 <= x < 2^23 with one reciprocal
  int d;
  float inv;
  __device__ __forceinline__ explicit FastDiv4(int d_) : d(d_), inv(1.0f / (float)d_) {}
  __device__ __forceinline__ int mod(int x) const {
    int q = (int)((float)x * inv);
    int r = x - q * d;
    if (r < 0) r += d;
    if (r >= d) r -= d;
    return r;
  }
};

// Dead block: every one of its `nrows` rows from `row0` lies in the padding of its clip(s) - beyond the geometry's H valid
// rows, or (ragged batches) beyond the need_mul * clip_frames[b] + need_add rows the clip's own length can bring to an
// output frame.  Such a block convolves nothing and stores zeros.
__device__ __forceinline__ bool w4_rows_live(int row0, int nrows, int rows_total, int Hp, int H, const int* clip_frames,
                                             int need_mul, int need_add) {
  bool live = false;
  if (row0 < rows_total) {
    const int r_end = row0 + nrows < rows_total ? row0 + nrows : rows_total;
    int b = row0 / Hp;
    for (int base = b * Hp; base < r_end; base += Hp, ++b) {
      const int lo = (row0 > base ? row0 : base) - base;
      int lim = H;
      if (clip_frames) {
        const int need = need_mul * clip_frames[b] + need_add;
        lim = need < lim ? need : lim;
      }
      live = live || lo < lim;
    }
  }
  return live;
}

}  // namespace
