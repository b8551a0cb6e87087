// Cnn14 conv stack, Winograd form: F(2x2, 3x3) convolution + eval BatchNorm + ReLU (+ 2x2 average
// pooling / mean over the 2 mel columns) on the exact-f32 matrix cores of gfx950.
//
// Same contract, layouts and epilogue modes as csrc/conv3x3.hip (reference ConvBlock.forward,
// cnn_encoder.py:59-75); 2.25x fewer multiplications: every 2x2 output tile needs 16 products per
// (cin, cout) pair instead of 36,
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        d = 4x4 input tile, g = 3x3 filter,
// and the 16 element-wise products become 16 independent GEMMs over the channels
//     M_p[tile, cout] = sum_cin V_p[tile, cin] * U_p[cin, cout],   p = (i, j) in 4x4,
// which is what the MFMAs run.  The filter transform U = G g G^T is done once on the host (in f64,
// rounded to f32).  The input transform has only +-1 coefficients and is done ON THE FLY from the raw
// halo patch in LDS while the A fragments are built (8 b128 reads + 8 float4 adds give the fragments of
// the four positions of one transform column), so nothing transformed is ever stored.  The output
// transform (+-1 again) runs in registers in the epilogue: a 2x2 Winograd tile IS a pooling window.
//
// Work decomposition: block = 64 tiles (256 output pixels) x 64 output channels, 4 waves, each wave owns
// ONE 32-tile x 32-channel MFMA tile for ALL 16 positions (16 x 16 = 256 accumulator registers, one wave
// per SIMD).  Per 32-channel chunk: 4 stages (transform column j), each streaming the 4 weight slabs
// U[(i, j)][64][32] through a double-buffered LDS ring; the next chunk's halo patch is prefetched into
// VGPRs during the last stage.
#include <stdlib.h>

#include "ac_common.h"

// Development builds only (-DAC_WINO_ABLATE=mask): bit0 no weight ring, bit1 no barriers, bit2 no transform, bit3 no patch
// reload, bit4 no epilogue - isolates what each phase of the kernel costs.  0 in the product library.
#ifndef AC_WINO_ABLATE
#define AC_WINO_ABLATE 0
#endif

namespace {

constexpr int kAblate = AC_WINO_ABLATE;

constexpr int LDS_STRIDE = 36;
constexpr int MAX_NPIX = 130 * 4;  // W = 2: (128 + 2) x (2 + 2)
// float4 per thread that cover one 32-channel halo patch.  Blocks spanning the full image width never load
// the two outer patch columns (always zero padding: cleared once), so 340 pixels is the largest load.
constexpr int PATCH_LD4 = (340 * 8 + 255) / 256;

struct WinoParams {
  const float* in;
  const float* upk;    // [Cin/32][4 j][4 i][Cout][32]
  const float* scale;
  const float* shift;
  float* out;
  int rows_total, Hp, H, W, Cin, Cout;
  int tct_log2;        // log2(tile columns per block), tile = 2x2 output pixels
  int mt_cols, MT, NT;
  int Hp_out, H_out, W_out;
  int map_mode;
};

enum { MODE_FULL = 0, MODE_POOL = 1, MODE_MEANW = 2 };

// Input transform B^T d B, per index x in 0..3 (rows and columns alike):
//   x=0: d0 - d2   x=1: d1 + d2   x=2: d2 - d1   x=3: d1 - d3
template <int MODE>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(WinoParams p) {
  __shared__ __attribute__((aligned(16))) float sA[MAX_NPIX * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sU[2][4 * 64 * LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5;

  int m_tile, n_tile;
  {
    const int bid = blockIdx.x;
    if (p.map_mode == 1) {
      const int xcd = bid & 7, seq = bid >> 3;
      n_tile = xcd + 8 * (seq / p.MT);
      m_tile = seq % p.MT;
    } else if (p.map_mode == 2) {
      const int xcd = bid & 7, seq = bid >> 3;
      n_tile = seq % p.NT;
      m_tile = (seq / p.NT) * 8 + xcd;
      if (m_tile >= p.MT) return;
    } else {
      n_tile = bid % p.NT;
      m_tile = bid / p.NT;
    }
  }
  const int TCT = 1 << p.tct_log2;       // tile columns
  const int TRT = 64 >> p.tct_log2;      // tile rows
  const int PW = 2 * TCT + 2, PH = 2 * TRT + 2;
  const int row0 = (m_tile / p.mt_cols) * (2 * TRT);
  const int col0 = (m_tile % p.mt_cols) * (2 * TCT);

  // A-operand row of this lane: tile t = wm*32 + (lane & 31); d[0][0] of the tile sits at patch (2*trow, 2*tcol)
  int pbase;
  {
    const int t = wm * 32 + (lane & 31);
    const int trow = t >> p.tct_log2, tcol = t & (TCT - 1);
    pbase = ((2 * trow) * PW + 2 * tcol) * LDS_STRIDE + half * 4;
  }
  const int rowoff = PW * LDS_STRIDE;
  const int nbase = (wn * 32 + (lane & 31)) * LDS_STRIDE + half * 4;

  f32x16 acc[16];  // acc[i*4 + j] = M_(i,j)
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  const int rc0 = row0 % p.Hp;
  const bool all_pad = (rc0 >= p.H && rc0 + 2 * TRT <= p.Hp) || row0 >= p.rows_total;

  const int nchunk = p.Cin >> 5;
  if (!all_pad) {
    // ---- halo patch: global -> VGPR (prefetch) and VGPR -> LDS ----
    f32x4 preg[PATCH_LD4];
    const bool full_w = p.mt_cols == 1;          // outer patch columns are padding for every block
    const int PWL = full_w ? PW - 2 : PW;        // patch columns actually loaded
    const int NLOAD = PH * PWL * 8;
    if (full_w) {
      for (int idx = tid; idx < PH * 16; idx += 256) {
        const int pr = idx >> 4, side = (idx >> 3) & 1, c4 = idx & 7;
        *(f32x4*)(sA + (pr * PW + (side ? PW - 1 : 0)) * LDS_STRIDE + c4 * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    auto patch_load = [&](int c) {
#pragma unroll
      for (int u = 0; u < PATCH_LD4; ++u) {
        const int idx = tid + u * 256;
        const int pix = idx >> 3, c4 = idx & 7;
        const int pr = pix / PWL, pc = pix - pr * PWL + (full_w ? 1 : 0);
        const int gr = row0 - 1 + pr, gc = col0 - 1 + pc;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < NLOAD && gr >= 0 && gr < p.rows_total && gc >= 0 && gc < p.W)
          v = *(const f32x4*)(p.in + ((size_t)gr * p.W + gc) * p.Cin + c * 32 + c4 * 4);
        preg[u] = v;
      }
    };
    auto patch_store = [&]() {
#pragma unroll
      for (int u = 0; u < PATCH_LD4; ++u) {
        const int idx = tid + u * 256;
        const int pix = idx >> 3, c4 = idx & 7;
        const int pr = pix / PWL, pc = pix - pr * PWL + (full_w ? 1 : 0);
        if (idx < NLOAD) *(f32x4*)(sA + (pr * PW + pc) * LDS_STRIDE + c4 * 4) = preg[u];
      }
    };
    // ---- weight stage (chunk c, column j): 4 slabs [64][32] -> 8 float4 per thread ----
    const float* ubase = p.upk + (size_t)n_tile * 64 * 32;
    const size_t pos_stride = (size_t)p.Cout * 32;  // between (j, i) slabs
    f32x4 ureg[8];
    auto u_load = [&](int stage /* c*4 + j */) {
      const float* src = ubase + (size_t)stage * 4 * pos_stride;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = tid + u * 256;          // 0..2047: i = idx >> 9, within-slab float4 = idx & 511
        ureg[u] = *(const f32x4*)(src + (size_t)(idx >> 9) * pos_stride + (size_t)(idx & 511) * 4);
      }
    };
    auto u_store = [&](float* dst) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = tid + u * 256;
        const int i = idx >> 9, f4 = idx & 511;
        *(f32x4*)(dst + (i * 64 + (f4 >> 3)) * LDS_STRIDE + (f4 & 7) * 4) = ureg[u];
      }
    };

    patch_load(0);
    u_load(0);
    patch_store();
    u_store(sU[0]);
    __syncthreads();

    const int total = nchunk * 4;
    int buf = 0;
    // One 8-channel group of transform column jj needs 8 raw patch reads (rows 0..3 at the two transform
    // columns CA/CB), 4 weight fragments, and the transform e[r] = x[r] +- y[r], v = row combinations of e.
    auto reads_a = [&](int jj, int g, f32x4 (&x)[4], f32x4 (&y)[4]) {
      const int ca = (jj == 0 ? 0 : (jj == 2 ? 2 : 1)) * LDS_STRIDE;
      const int cb = (jj == 0 ? 2 : (jj == 1 ? 2 : (jj == 2 ? 1 : 3))) * LDS_STRIDE;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[r] = *(const f32x4*)(sA + pbase + r * rowoff + ca + g * 8);
        y[r] = *(const f32x4*)(sA + pbase + r * rowoff + cb + g * 8);
      }
    };
    auto reads_b = [&](const float* u, int g, f32x4 (&b)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = *(const f32x4*)(u + i * 64 * LDS_STRIDE + nbase + g * 8);
    };
    auto transform = [&](int jj, const f32x4 (&x)[4], const f32x4 (&y)[4], f32x4 (&v)[4]) {
      f32x4 e[4];
      if (kAblate & 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x[r];
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = (jj == 1) ? (x[r] + y[r]) : (x[r] - y[r]);
      v[0] = e[0] - e[2];
      v[1] = e[1] + e[2];
      v[2] = e[2] - e[1];
      v[3] = e[1] - e[3];
    };
    // One wave per SIMD issues in order, so the matrix pipe only stays busy if fewer than 64 cycles of other
    // work sit between two MFMAs and no wait is reached before its data has landed.  The 16 (column j, group g)
    // steps of a chunk are software-pipelined and pinned with scheduling barriers:
    //   segment B: MFMAs 1-8 of the step, the LDS reads of the NEXT step issued behind MFMAs 1-6
    //   segment C: MFMAs 9-16, the 32 transform adds of the next step spread between them
    // Across a stage boundary only the patch reads are prefetched (the weight ring flips at the barrier).
    f32x4 vc[4], bc[4];
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
      {
        f32x4 x[4], y[4];
        reads_a(0, 0, x, y);
        reads_b(sU[buf], 0, bc);
        transform(0, x, y, vc);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int st = c * 4 + j;
        if (!(kAblate & 1)) u_load(st + 1 < total ? st + 1 : st);
        if (j == 3 && c + 1 < nchunk && !(kAblate & 8)) patch_load(c + 1);
        const float* ucur = sU[buf];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 xn[4], yn[4], bn[4], vn[4];
          const bool same_stage = g < 3;
          const bool next_stage = g == 3 && j < 3;
          __builtin_amdgcn_sched_barrier(0);
          if (same_stage) { reads_a(j, g + 1, xn, yn); reads_b(ucur, g + 1, bn); }
          if (next_stage) reads_a(j + 1, 0, xn, yn);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i * 4 + j] = mfma32(vc[i][s], bc[i][s], acc[i * 4 + j]);
          if (same_stage || next_stage) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (same_stage) transform(j, xn, yn, vn);
          if (next_stage) transform(j + 1, xn, yn, vn);
#pragma unroll
          for (int i = 2; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[i * 4 + j] = mfma32(vc[i][s], bc[i][s], acc[i * 4 + j]);
          if (same_stage || next_stage) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (same_stage) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { vc[i] = vn[i]; bc[i] = bn[i]; }
          }
          if (next_stage) {
#pragma unroll
            for (int i = 0; i < 4; ++i) vc[i] = vn[i];
          }
        }
        if (!(kAblate & 1)) u_store(sU[buf ^ 1]);
        if (!(kAblate & 2)) __syncthreads();
        buf ^= 1;
        if (j < 3) reads_b(sU[buf], 0, bc);
        if (j == 3 && c + 1 < nchunk && !(kAblate & 8)) {
          patch_store();
          if (!(kAblate & 2)) __syncthreads();
        }
      }
    }
  }

  // ---- epilogue: output transform Y = A^T M A, BN, ReLU, pool / store ----
  if (kAblate & 16) {
    if (acc[0][0] == 12345.678f) p.out[0] = 1.f;  // keep the accumulators live
    return;
  }
  const int ch = n_tile * 64 + wn * 32 + (lane & 31);
  const float sc = p.scale[ch], sh = p.shift[ch];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float m[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) m[q] = acc[q][r];
    // rows of A^T M: t0 = M0 + M1 + M2, t1 = M1 - M2 - M3 (per column)
    float t0[4], t1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
      t1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
    }
    float y[4];
    y[0] = t0[0] + t0[1] + t0[2];
    y[1] = t0[1] - t0[2] - t0[3];
    y[2] = t1[0] + t1[1] + t1[2];
    y[3] = t1[1] - t1[2] - t1[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = fmaxf(fmaf(y[e], sc, sh), 0.f);
    const int t = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    const int trow = t >> p.tct_log2, tcol = t & (TCT - 1);
    const int wy = row0 + 2 * trow, wx = col0 + 2 * tcol;
    if (MODE == MODE_FULL) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gr = wy + (e >> 1), gc = wx + (e & 1);
        if (gr < p.rows_total) {
          const bool valid = (gr % p.Hp) < p.H;
          p.out[((size_t)gr * p.W + gc) * p.Cout + ch] = valid ? y[e] : 0.f;
        }
      }
    } else if (MODE == MODE_POOL) {
      const int orow = wy >> 1, ocol = wx >> 1;
      if (wy < p.rows_total) {
        const bool valid = (orow % p.Hp_out) < p.H_out;
        const float o = 0.25f * ((y[0] + y[1]) + (y[2] + y[3]));
        p.out[((size_t)orow * p.W_out + ocol) * p.Cout + ch] = valid ? o : 0.f;
      }
    } else {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int gr = wy + dy;
        if (gr < p.rows_total) {
          const int b = gr / p.Hp, h = gr - b * p.Hp;
          if (h < p.H) p.out[((size_t)b * p.H + h) * p.Cout + ch] = 0.5f * (y[2 * dy] + y[2 * dy + 1]);
        }
      }
    }
  }
}

template <int MODE>
int launch_wino(const WinoParams& p, hipStream_t s) {
  unsigned grid;
  if (p.map_mode == 2) grid = (unsigned)(((p.MT + 7) / 8) * 8 * p.NT);
  else grid = (unsigned)(p.MT * p.NT);
  hipLaunchKernelGGL((conv3x3_wino_kernel<MODE>), dim3(grid), dim3(256), 0, s, p);
  return ac_check_launch();
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_conv3x3_bn_relu_winograd(const float* in, const float* upk, const float* scale,
                                           const float* shift, float* out, int B, int Hp, int H, int W, int Cin,
                                           int Cout, int mode, int map_mode, void* stream) {
  if (!in || !upk || !scale || !shift || !out) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || (Hp & 1) || W < 2 || (W & (W - 1)) || Cin % 32 || Cout % 64) return AC_ERR_ARG;
  if (mode < 0 || mode > 2) return AC_ERR_ARG;
  if (mode == MODE_MEANW && W != 2) return AC_ERR_ARG;
  WinoParams p;
  p.in = in; p.upk = upk; p.scale = scale; p.shift = shift; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TCT = (W / 2) < 16 ? (W / 2) : 16;
  int l2 = 0;
  while ((1 << l2) < TCT) ++l2;
  p.tct_log2 = l2;
  const int TRT = 64 / TCT;
  p.mt_cols = (W / 2) / TCT;
  p.MT = ((p.rows_total + 2 * TRT - 1) / (2 * TRT)) * p.mt_cols;
  p.NT = Cout / 64;
  p.Hp_out = Hp / 2; p.H_out = H / 2; p.W_out = W / 2;
  if (map_mode < 0) map_mode = (p.NT % 8 == 0 && p.NT >= 8) ? 1 : 2;
  if (map_mode == 1 && p.NT % 8 != 0) return AC_ERR_ARG;
  p.map_mode = map_mode;
  hipStream_t s = (hipStream_t)stream;
  if (mode == MODE_FULL) return launch_wino<MODE_FULL>(p, s);
  if (mode == MODE_POOL) return launch_wino<MODE_POOL>(p, s);
  return launch_wino<MODE_MEANW>(p, s);
}
