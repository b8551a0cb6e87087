// Decode-step projections for WIDE row batches (a greedy chain over several submissions, a beam search over grouped
// batches: 200 ... 1000 rows per step), gfx950.
//
//   Y[M, N] = act(P(X)[M, K] W[N, K]^T + bias)        (the F.linear calls of nn.TransformerDecoderLayer and the classifier:
//                                                      reference transformer_decoder.py:92-101)
// with the producer P of the A rows fused in, as in csrc/decoder.hip's dec_gemm_kernel:
//   0 PACKED: A = rows that a previous kernel left ALREADY split into bf16 planes in fragment order (the attention kernel, the
//             epilogue of the first feed-forward product, ac_dec_wide_pack): no staging, no LDS, no barrier before the products
//   1 EMBED : A = E[tok[r][t]] * sqrt(d) + pe[t]                       (transformer_decoder.py:89-91)
//   2 ADDLN : A = LayerNorm(X + Y2) * g + b                            (the post-LN residual join of the previous sub-layer)
//
// Why a second kernel family.  The step kernels of csrc/decoder.hip are built for <= 64 ... 128 rows: 16 x 16 output tiles
// on the exact-f32 matrix instruction (the weights are re-read once per 16 rows) and one workgroup per ROW for the attention
// sub-layers (three 256 x 256 matrix-vector products per row on the vector ALUs).  At 256 rows a step is 98 us of kernels
// that occupy every CU of the chip (25 CU.ms), at 768 rows 260 us - and beside the next batch's encoder every one of those
// CU.us is taken from the one-workgroup-per-CU conv kernels.  Here a workgroup owns 32 rows x 64 (32) columns, the weights
// are re-read once per 32 rows and a projection is a few dozen short-lived workgroups.
//
// Arithmetic: f32-grade on the bf16 matrix cores.  Both operands are split into THREE bf16 planes, x = x0 + x1 + x2 with
// x0 = RNE(x), x1 = RNE(x - x0), x2 = RNE(x - x0 - x1): 24 significant bits, the split of an f32 value is exact.  A product is
// the six plane products of order <= 2 (x0 w0, x0 w1, x1 w0, x0 w2, x1 w1, x2 w0: the dropped ones are below 2^-24 of
// |x| |w|), each a v_mfma_f32_32x32x16_bf16 with f32 accumulation: 192 matrix-pipe cycles per 16 k against 512 for the
// f32 instruction (v_mfma_f32_32x32x2_f32), and an error below the rounding of an f32 dot product (measured against f64
// on the classifier's shape: 2e-7 of the largest output, a CPU f32 matmul 3e-7; the two-plane split of the conv tiers 5e-6).
// The three second-order products and the three larger ones run as two independent accumulator chains.
//
// Layout.  A matrix [rows][K] is packed (ac_dec_wide_pack, or by the producing kernel) in MFMA fragment order,
//   P[tile of 32 rows][k step of 16][plane][lane][8 bf16],   lane l = row (l & 31), k = 8 (l >> 5) .. + 7,
// a wave's fragment is one contiguous 1 KiB read - weights (rows = output columns) and PACKED activations alike.  Operand
// order (weights, rows): a lane ends with four consecutive COLUMNS of one row per register quad -> 16-byte stores.
// Producers 1 / 2 stage their 32 rows through LDS planes [row][k] with a row pitch of 2 K + 16 bytes (the 16 lanes of every
// ds_read_b128 group hit 16 different 16-byte bank slots); every load of the 32 rows is requested before the first LayerNorm
// reduction.  The four waves split the tile's columns (halves) and K (halves / quarters); the K parts meet in LDS.
#include "ac_common.h"
#include "ac_wino43.h"   // bf16x8, u32x2, cvt_pk_bf16
#include "../../include/audiocaption_hip.h"

namespace {

constexpr int WD_D = 256;               // K of the fused producers (= d_model of the shapes the wide route covers)
constexpr int WD_RS = WD_D * 2 + 16;    // bytes between rows of an LDS plane
constexpr int WD_PLANE = 32 * WD_RS;
constexpr int WD_RING = 8;              // k steps of 16 in flight per wave

struct WideParams {
  const float* X; long ldx;             // producer 2: residual rows; producer 0: the fragment pack of the rows
  const float* Y2; long ldy2;
  const float* ln_w; const float* ln_b;
  const int* tok; long tok_stride; int t;
  const float* emb; const float* pe; float emb_scale;
  float* xout; long ldxo;
  const unsigned char* Wp; const float* bias;
  float* Y; long ldy;
  int M, N, K, relu, ntb, vec;
};

__device__ __forceinline__ float bf_lo(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf_hi(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// four floats -> three packed bf16 quadruples (2 dwords each)
__device__ __forceinline__ void split3_bf16x4(const f32x4 x, u32x2& p0, u32x2& p1, u32x2& p2) {
  p0.x = cvt_pk_bf16(x[0], x[1]);
  p0.y = cvt_pk_bf16(x[2], x[3]);
  const float r0 = x[0] - bf_lo(p0.x), r1 = x[1] - bf_hi(p0.x), r2 = x[2] - bf_lo(p0.y), r3 = x[3] - bf_hi(p0.y);
  p1.x = cvt_pk_bf16(r0, r1);
  p1.y = cvt_pk_bf16(r2, r3);
  p2.x = cvt_pk_bf16(r0 - bf_lo(p1.x), r1 - bf_hi(p1.x));
  p2.y = cvt_pk_bf16(r2 - bf_lo(p1.y), r3 - bf_hi(p1.y));
}

// W [N][K] f32 (row pitch ldw) -> fragment pack.  One thread per (tile, k step, lane): 8 values, three 16-byte stores.
__global__ void wide_pack_kernel(const float* W, long ldw, int N, int K, unsigned char* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kst = K >> 4;
  const size_t total = (size_t)((N + 31) / 32) * kst * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const size_t g = i >> 6;
  const int ks = (int)(g % kst), nt = (int)(g / kst);
  const int n = nt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  if (n < N) {
    a = *(const f32x4*)(W + (size_t)n * ldw + k);
    b = *(const f32x4*)(W + (size_t)n * ldw + k + 4);
  }
  u32x2 a0, a1, a2, b0, b1, b2;
  split3_bf16x4(a, a0, a1, a2);
  split3_bf16x4(b, b0, b1, b2);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned char* dst = out + (g * 3) * 1024 + (size_t)lane * 16;
  *(u32x4*)(dst) = (u32x4){a0.x, a0.y, b0.x, b0.y};
  *(u32x4*)(dst + 1024) = (u32x4){a1.x, a1.y, b1.x, b1.y};
  *(u32x4*)(dst + 2048) = (u32x4){a2.x, a2.y, b2.x, b2.y};
}

#ifdef AC_WIDE_STAMPS   // development (tools/wide_stamps.py): phase timestamps (100 MHz) of wave 0 of the LAST workgroup
__device__ unsigned long long g_wide_stamps[16];
#define WIDE_STAMP(k) do { if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && threadIdx.x == 0) g_wide_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WIDE_STAMP(k) do { } while (0)
#endif

// the six plane products of one k step: second-order terms into `lo`, the rest into `hi`
__device__ __forceinline__ void six_products(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x16& hi, f32x16& lo) {
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2], x[0], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[0], hi, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], x[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[1], hi, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[2], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], hi, 0, 0, 0);
}

// bias + ReLU + store of one register quad (row gm, columns gn .. gn + 3); SPLIT: into the fragment pack of the [M][N] result
template <int SPLIT>
__device__ __forceinline__ void wide_store(const WideParams& p, f32x4 y, const f32x4 bq, int gm, int gn) {
  if (gm >= p.M || gn >= p.N) return;
  if (p.vec) {
    y += bq;
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
    }
    if (SPLIT) {
      u32x2 s0, s1, s2;
      split3_bf16x4(y, s0, s1, s2);
      unsigned char* d = (unsigned char*)p.Y + ((size_t)(gm >> 5) * (p.N >> 4) + (gn >> 4)) * 3072 +
                         (size_t)((gm & 31) + 32 * ((gn >> 3) & 1)) * 16 + (gn & 7) * 2;
      *(u32x2*)(d) = s0;
      *(u32x2*)(d + 1024) = s1;
      *(u32x2*)(d + 2048) = s2;
    } else {
      *(f32x4*)(p.Y + (size_t)gm * p.ldy + gn) = y;
    }
  } else {   // rows of Y that are not 16-byte aligned (a vocabulary that is no multiple of 4): element stores
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (gn + e < p.N) {
        float ye = y[e] + (p.bias ? p.bias[gn + e] : 0.f);
        if (p.relu) ye = fmaxf(ye, 0.f);
        p.Y[(size_t)gm * p.ldy + gn + e] = ye;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// producer 0: both operands are fragment packs.  Waves = WN column tiles x (4 / WN) K parts, 8 NG k steps per wave.
// ---------------------------------------------------------------------------------------------------------------------
template <int WN, int NG>
__global__ __launch_bounds__(256) void dec_wide_packed_kernel(WideParams p) {
  constexpr int KP = 4 / WN;
  __shared__ __attribute__((aligned(16))) float red[(KP - 1) * WN * 4 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wk = wave / WN;
  const int kst = p.K >> 4;
  const int nt32_total = (p.N + 31) >> 5;
  int tile = blockIdx.x * WN + wn;
  tile = tile < nt32_total ? tile : nt32_total - 1;
  WIDE_STAMP(0);
  const unsigned char* wp = p.Wp + ((size_t)tile * kst + wk * (WD_RING * NG)) * 3072 + (size_t)lane * 16;
  const unsigned char* xp = (const unsigned char*)p.X + ((size_t)blockIdx.y * kst + wk * (WD_RING * NG)) * 3072 + (size_t)lane * 16;
  bf16x8 wr[WD_RING][3], xr[WD_RING][3];
#pragma unroll
  for (int i = 0; i < WD_RING; ++i)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      xr[i][pl] = *(const bf16x8*)(xp + (size_t)i * 3072 + pl * 1024);
      wr[i][pl] = *(const bf16x8*)(wp + (size_t)i * 3072 + pl * 1024);
    }
  const int n0 = (blockIdx.x * WN + wn) * 32;
  const int gm = blockIdx.y * 32 + (lane & 31);
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int gn = n0 + 8 * q + 4 * (lane >> 5);
    bq[q] = (p.bias && p.vec && gn < p.N) ? *(const f32x4*)(p.bias + gn) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  WIDE_STAMP(1);
  f32x16 hi, lo;
#pragma unroll
  for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < WD_RING; ++i) {
      bf16x8 w[3], x[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) { w[pl] = wr[i][pl]; x[pl] = xr[i][pl]; }
      if (g + 1 < NG) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          xr[i][pl] = *(const bf16x8*)(xp + (size_t)((g + 1) * WD_RING + i) * 3072 + pl * 1024);
          wr[i][pl] = *(const bf16x8*)(wp + (size_t)((g + 1) * WD_RING + i) * 3072 + pl * 1024);
        }
      }
      six_products(w, x, hi, lo);
    }
  hi += lo;
  WIDE_STAMP(4);
  // ---- the K parts meet in LDS; the wk = 0 waves finish the tile ----
  f32x4* rb = (f32x4*)red + lane;
  if (wk > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      rb[(((wk - 1) * WN + wn) * 4 + q) * 64] = (f32x4){hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]};
  }
  __syncthreads();
  WIDE_STAMP(6);
  if (wk == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 y = (f32x4){hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]};
#pragma unroll
      for (int k = 1; k < KP; ++k) y += rb[(((k - 1) * WN + wn) * 4 + q) * 64];
      wide_store<0>(p, y, bq[q], gm, n0 + 8 * q + 4 * (lane >> 5));
    }
  }
  WIDE_STAMP(7);
}

// ---------------------------------------------------------------------------------------------------------------------
// producers 1 / 2 (K = 256): the 32 A rows are produced once per workgroup into LDS planes; waves = 2 column tiles x 2 K halves;
// ntb consecutive 64-column groups re-use the planes.  SPLIT: the result leaves as a fragment pack (input of a producer-0 launch).
// ---------------------------------------------------------------------------------------------------------------------
template <int PRO, int SPLIT>
__global__ __launch_bounds__(256) void dec_wide_fused_kernel(WideParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  float* red = (float*)(wsm + 3 * WD_PLANE);   // [2 buffers][column half][4 register quads][64 lanes] float4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int m0 = blockIdx.y * 32;
  constexpr int kst = WD_D / 16;
  const int nt32_total = (p.N + 31) >> 5;
  WIDE_STAMP(0);

  // ---- every global load of the producer first (2 x 16 rows per wave: one row per 16-lane group, lane (grp, sub) holds the
  // row's float4 columns sub, sub + 16, sub + 32, sub + 48), then the weights of the first column group ----
  constexpr int NF = WD_D / 64;
  const int grp = lane >> 4, sub = lane & 15;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 v[2][NF], v2[2][NF], gw[NF], gb[NF];
  bool ok[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = m0 + it * 16 + wave * 4 + grp;
    ok[it] = r < p.M;
    const int rr = ok[it] ? r : 0;
    if (PRO == 1) {
      const int w = p.tok[(size_t)rr * p.tok_stride + p.t];
      const f32x4* e4 = (const f32x4*)(p.emb + (size_t)w * WD_D);
#pragma unroll
      for (int i = 0; i < NF; ++i) v[it][i] = e4[sub + 16 * i];
    } else {
      const f32x4* x4 = (const f32x4*)(p.X + (size_t)rr * p.ldx);
      const f32x4* y4 = (const f32x4*)(p.Y2 + (size_t)rr * p.ldy2);
#pragma unroll
      for (int i = 0; i < NF; ++i) { v[it][i] = x4[sub + 16 * i]; v2[it][i] = y4[sub + 16 * i]; }
    }
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    if (PRO == 1) {
      gw[i] = ((const f32x4*)(p.pe + (size_t)p.t * WD_D))[sub + 16 * i];
    } else {
      gw[i] = ((const f32x4*)p.ln_w)[sub + 16 * i];
      gb[i] = ((const f32x4*)p.ln_b)[sub + 16 * i];
    }
  }
  auto w_tile = [&](int nt) {
    int tile = (blockIdx.x * p.ntb + nt) * 2 + wn;
    return tile < nt32_total ? tile : nt32_total - 1;          // a tile beyond N: its outputs are never stored
  };
  const unsigned char* wbase = p.Wp + (size_t)(wk * WD_RING) * 3072 + (size_t)lane * 16;
  bf16x8 wr[WD_RING][3];
  {
    const unsigned char* s = wbase + (size_t)w_tile(0) * kst * 3072;
#pragma unroll
    for (int i = 0; i < WD_RING; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) wr[i][pl] = *(const bf16x8*)(s + (size_t)i * 3072 + pl * 1024);
  }
  WIDE_STAMP(1);

  // ---- produce, split, stage ----
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = it * 16 + wave * 4 + grp, r = m0 + row;
    if (PRO == 1) {
#pragma unroll
      for (int i = 0; i < NF; ++i) v[it][i] = v[it][i] * p.emb_scale + gw[i];
    } else {
#pragma unroll
      for (int i = 0; i < NF; ++i) v[it][i] += v2[it][i];
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NF; ++i) sm += (v[it][i][0] + v[it][i][1]) + (v[it][i][2] + v[it][i][3]);
      const float mean = row16_sum(sm) * (1.0f / WD_D);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dl = v[it][i][e] - mean; q = fmaf(dl, dl, q); }
      const float rstd = rsqrtf(row16_sum(q) * (1.0f / WD_D) + 1e-5f);
#pragma unroll
      for (int i = 0; i < NF; ++i) v[it][i] = (v[it][i] - mean) * rstd * gw[i] + gb[i];
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const f32x4 vv = ok[it] ? v[it][i] : zero4;
      u32x2 s0, s1, s2;
      split3_bf16x4(vv, s0, s1, s2);
      unsigned char* d = wsm + row * WD_RS + (sub + 16 * i) * 8;
      *(u32x2*)(d) = s0;
      *(u32x2*)(d + WD_PLANE) = s1;
      *(u32x2*)(d + 2 * WD_PLANE) = s2;
      if (blockIdx.x == 0 && p.xout && ok[it]) *(f32x4*)(p.xout + (size_t)r * p.ldxo + (sub + 16 * i) * 4) = v[it][i];
    }
  }
  WIDE_STAMP(2);
  __syncthreads();
  WIDE_STAMP(3);

  const unsigned char* xb = wsm + (lane & 31) * WD_RS + (lane >> 5) * 16 + wk * WD_RING * 32;   // this lane's B fragments
  const int gm = m0 + (lane & 31);
  for (int nt = 0; nt < p.ntb; ++nt) {
    const int n0 = ((blockIdx.x * p.ntb + nt) * 2 + wn) * 32;
    if (n0 - 32 * wn >= p.N) break;          // the whole 64-column group lies beyond N (uniform over the workgroup)
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gn = n0 + 8 * q + 4 * (lane >> 5);
      bq[q] = (p.bias && p.vec && gn < p.N) ? *(const f32x4*)(p.bias + gn) : zero4;
    }
    const unsigned char* snext = wbase + (size_t)w_tile(nt + 1 < p.ntb ? nt + 1 : nt) * kst * 3072;
    f32x16 hi, lo;
#pragma unroll
    for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
    bf16x8 xf[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) xf[0][pl] = *(const bf16x8*)(xb + pl * WD_PLANE);
#pragma unroll
    for (int i = 0; i < WD_RING; ++i) {
      if (i + 1 < WD_RING) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xf[(i + 1) & 1][pl] = *(const bf16x8*)(xb + pl * WD_PLANE + (i + 1) * 32);
      }
      bf16x8 w[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[pl] = wr[i][pl];
      if (p.ntb > 1) {   // uniform: the slot is free, request the same k step of the next column group (the last group: itself)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wr[i][pl] = *(const bf16x8*)(snext + (size_t)i * 3072 + pl * 1024);
      }
      six_products(w, xf[i & 1], hi, lo);
    }
    hi += lo;
    if (nt == 0) WIDE_STAMP(4);
    // ---- the two K halves meet in LDS (two buffers: the next group's partials may land while this one is read) ----
    f32x4* rb = (f32x4*)red + (size_t)((nt & 1) * 2 + wn) * 4 * 64 + lane;
    if (wk == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rb[q * 64] = (f32x4){hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]};
    }
    __syncthreads();
    if (nt == 0) WIDE_STAMP(6);
    if (wk == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 y = (f32x4){hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]} + rb[q * 64];
        wide_store<SPLIT>(p, y, bq[q], gm, n0 + 8 * q + 4 * (lane >> 5));
      }
    }
  }
  WIDE_STAMP(7);
}

constexpr size_t WD_FUSED_LDS = 3 * (size_t)WD_PLANE + 2 * 2 * 4 * 64 * 16;

template <int PRO, int SPLIT>
int launch_fused(const WideParams& p, hipStream_t s) {
  static AcLdsAttr attr;
  const int e = ac_allow_lds((const void*)dec_wide_fused_kernel<PRO, SPLIT>, (int)WD_FUSED_LDS, &attr);
  if (e != AC_OK) return e;
  dim3 grid((p.N + 64 * p.ntb - 1) / (64 * p.ntb), (p.M + 31) / 32);
  hipLaunchKernelGGL((dec_wide_fused_kernel<PRO, SPLIT>), grid, dim3(256), WD_FUSED_LDS, s, p);
  return ac_check_launch();
}

template <int WN, int NG>
int launch_packed(const WideParams& p, hipStream_t s) {
  dim3 grid((p.N + 32 * WN - 1) / (32 * WN), (p.M + 31) / 32);
  hipLaunchKernelGGL((dec_wide_packed_kernel<WN, NG>), grid, dim3(256), 0, s, p);
  return ac_check_launch();
}

}  // namespace

#ifdef AC_WIDE_STAMPS
extern "C" int ac_wide_stamps_read(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wide_stamps), sizeof(g_wide_stamps)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" long ac_dec_wide_packed_floats(int N, int K) {
  if (N <= 0 || K <= 0 || K % 16) return -1;
  return (long)((size_t)((N + 31) / 32) * (K / 16) * 3072 / 4);
}

extern "C" int ac_dec_wide_pack(const float* W, long ldw, int N, int K, float* out, void* stream) {
  if (!W || !out || N <= 0 || K <= 0 || K % 16 || ldw % 4 || ((uintptr_t)W & 15) || ((uintptr_t)out & 15)) return AC_ERR_ARG;
  const size_t total = (size_t)((N + 31) / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(wide_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                     (unsigned char*)out);
  return ac_check_launch();
}

extern "C" int ac_dec_wide_gemm(int producer, const float* X, long ldx, const float* Y2, long ldy2, const float* ln_w,
                                const float* ln_b, const int* tok, long tok_stride, int t, const float* emb, const float* pe,
                                float emb_scale, float* xout, long ldxo, const float* Wp, const float* bias, float* Y, long ldy,
                                int M, int N, int K, int relu, int ntb, int split_out, void* stream) {
  if (!Wp || !Y || M <= 0 || N <= 0 || producer < 0 || producer > 2 || ((uintptr_t)Wp & 15)) return AC_ERR_ARG;
  const bool vec = N % 4 == 0 && (split_out || ldy % 4 == 0) && !((uintptr_t)Y & 15) && !(bias && ((uintptr_t)bias & 15));
  if (split_out && (producer == 0 || N % 16 || !vec)) return AC_ERR_ARG;
  if (producer == 0 && (!X || ((uintptr_t)X & 15) || ntb != 1 || (K != 256 && K != 512 && K != 1024))) return AC_ERR_ARG;
  if (producer != 0 && (K != WD_D || ntb < 1 || (xout && (ldxo % 4 || ((uintptr_t)xout & 15))))) return AC_ERR_ARG;
  if (producer == 1 && (!tok || !emb || !pe || t < 0 || ((uintptr_t)emb & 15) || ((uintptr_t)pe & 15))) return AC_ERR_ARG;
  if (producer == 2 && (!X || !Y2 || !ln_w || !ln_b || ldx % 4 || ldy2 % 4 || ((uintptr_t)X & 15) || ((uintptr_t)Y2 & 15) ||
                        ((uintptr_t)ln_w & 15) || ((uintptr_t)ln_b & 15)))
    return AC_ERR_ARG;
  WideParams p;
  p.X = X; p.ldx = ldx; p.Y2 = Y2; p.ldy2 = ldy2; p.ln_w = ln_w; p.ln_b = ln_b;
  p.tok = tok; p.tok_stride = tok_stride; p.t = t; p.emb = emb; p.pe = pe; p.emb_scale = emb_scale;
  p.xout = xout; p.ldxo = ldxo; p.Wp = (const unsigned char*)Wp; p.bias = bias; p.Y = Y; p.ldy = ldy;
  p.M = M; p.N = N; p.K = K; p.relu = relu; p.ntb = ntb; p.vec = vec ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (producer == 0) {
    if (K == 256) return launch_packed<2, 1>(p, s);
    if (K == 512) return launch_packed<1, 1>(p, s);
    return launch_packed<1, 2>(p, s);
  }
  if (producer == 1) return split_out ? launch_fused<1, 1>(p, s) : launch_fused<1, 0>(p, s);
  return split_out ? launch_fused<2, 1>(p, s) : launch_fused<2, 0>(p, s);
}
