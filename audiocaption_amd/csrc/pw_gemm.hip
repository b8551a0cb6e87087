// 1x1 convolution / linear layer with STATIC weights on the split-bf16 matrix path, activation-stationary:
//   y[M][N] = act((x[M][K] .* gate[m / gate_rows][K]) W[N][K]^T + bias) + beta * y
// for the matrix-bound 1x1 convolutions of the EfficientNet-B2 encoder (MBConv expand / project / head at 8 k - 32 k
// rows against 88 ... 2112 channels; efficientnet_pytorch==0.7.1 _expand_conv / _project_conv / _conv_head with the
// BatchNorm folded in, call sites hf_wrapper.py:229-241) and the decoder's teacher-forced projections.
//
// Why not ac_gemm_bf16x3 (csrc/train.hip): that kernel splits BOTH f32 operands into bf16 hi + lo while staging them
// into LDS, per 64 x 64 or 128 x 128 output tile - ~150 VALU instructions per thread and 32-k chunk next to 24 MFMAs,
// repeated for every tile that touches an operand element (N / 128 times per activation).  Here
//   * the weights are split and laid out in MFMA FRAGMENT ORDER once, when the model is packed
//     ([K/16 k-steps][N/32 tiles][hi, lo][64 lanes][8 bf16]): a wave's B fragment is one contiguous 1 KiB read straight
//     from L2 into registers - no LDS, no conversion, the next k-step's fragments requested under the current MFMAs;
//   * a workgroup owns 32 MW rows and a 256-column group: its four waves walk ALL its rows for two 32-column tiles each
//     (1 x 4 wave grid: a weight fragment feeds 3 MW MFMAs), so an activation is split N / 256 times, not N / 128, by
//     ~40 VALU instructions per thread and chunk next to 24 MFMAs (MW = 2; 128-row tiles measured slower everywhere);
//   * global loads run two 32-k steps ahead of their use (two register sets of activations, a ring of four weight
//     k-steps): at 12-24 MFMAs per wave and step a single step does not cover the L2 latency;
//   * the accumulator is the TRANSPOSED tile (weights as the MFMA row operand): a lane ends up with 4 consecutive
//     channels of one row, so bias / swish / residual work on 16-byte words and the stores are 16 bytes wide.
// Arithmetic: x = hi + lo (bf16, RNE), products lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, f32 accumulation -
// 2^-16 relative operand error, the same as the split-bf16 conv tier and ac_gemm_bf16x3.
#include "ac_common.h"
#include "ac_drop.h"
#include <stdlib.h>

namespace {

typedef __bf16 pg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pg_u32x2 __attribute__((ext_vector_type(2)));
constexpr int PG_ROW = 40;   // bf16 per LDS row: 32 k + 8 pad (80 B: a lane's 16-byte fragment reads stay conflict-light)

struct PwgP {
  const float* x; const pg_bf16x8* wf; const float* bias; float* y; const float* gate;
  long ldx, ldy, row0;                                     // row strides in floats; row0: dropout index of row 0
  int M, N, K, KC, NT32, act, gate_rows;
  float beta;
  Drop drop;                                               // inverted dropout on the output (training forward), off: thresh 0
};

__device__ __forceinline__ unsigned pg_cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// 4 floats -> 4 bf16 hi (2 dwords) + 4 bf16 lo
__device__ __forceinline__ void pg_split4(const f32x4 x, pg_u32x2& hi, pg_u32x2& lo) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned h = pg_cvt_pk_bf16(x[2 * i], x[2 * i + 1]);
    const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
    hi[i] = h;
    lo[i] = pg_cvt_pk_bf16(x[2 * i] - h0, x[2 * i + 1] - h1);
  }
}

// ---- epilogue: lane = row m0 + 32 a + (lane & 31), registers 4 g .. 4 g + 3 = channels 8 g + 4 (lane >> 5) + 0..3 ----
template <int MW>
__device__ __forceinline__ void pw_epilogue(const PwgP& p, f32x16 (&acc)[MW][2], int m0, int nt0, bool on0, bool on1, int lane) {
  if (!on0) return;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (b == 1 && !on1) break;
    const int nb = (nt0 + b) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int a = 0; a < MW; ++a) {
      const int m = m0 + a * 32 + (lane & 31);
      if (m >= p.M) continue;
      float* yr = p.y + (long)m * p.ldy;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g;
        if (n >= p.N) continue;                       // N % 4 == 0: a quad is inside or outside as a whole
        f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (p.act == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = ac_swish_fast(v[q]);
        } else if (p.act == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if (p.drop.thresh != 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] *= p.drop.mask((uint64_t)(p.row0 + m) * (uint64_t)p.N + (uint64_t)(n + q));
        }
        if (p.beta != 0.f) v += p.beta * *(const f32x4*)(yr + n);
        *(f32x4*)(yr + n) = v;
      }
    }
  }
}

template <int MW>
__global__ __launch_bounds__(256, 2) void pw_bf16x3_kernel(PwgP p) {
  constexpr int BM = 32 * MW;
  constexpr int NI = BM / 32;                      // float4 staging items per thread and chunk
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][2][BM * PG_ROW];   // [buffer][hi, lo]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * BM;
  const int nt0 = blockIdx.y * 8 + wave * 2;       // this wave's two 32-column tiles
  const bool on0 = nt0 < p.NT32, on1 = nt0 + 1 < p.NT32;
  // ---- staging: item j of this thread = float4 (row, 4 k) of the chunk ----
  const float* xrow[NI];
  const float* grow[NI];
  bool rok[NI];
  int sdst[NI];
  const int kq = tid & 7;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (tid >> 3) + 32 * j;
    const int m = m0 + row;
    rok[j] = m < p.M;
    const int mc = rok[j] ? m : p.M - 1;
    xrow[j] = p.x + (long)mc * p.ldx + kq * 4;
    grow[j] = p.gate ? p.gate + (long)(mc / p.gate_rows) * p.K + kq * 4 : nullptr;
    sdst[j] = row * PG_ROW + kq * 4;
  }
  // Two register sets of staged activations: chunk c + 2 is requested at the top of step c and split into LDS at the end
  // of step c + 1, so a global load has two steps of MFMAs to land (a step is only 12 MW MFMAs per wave).
  f32x4 preA[NI], pgA[NI], preB[NI], pgB[NI];
  auto request = [&](int c, f32x4 (&pre)[NI], f32x4 (&pg)[NI]) {
    const bool kok = c < p.KC && c * 32 + kq * 4 < p.K;   // K % 4 == 0: a quad is inside or outside as a whole
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      pg[j] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (kok && rok[j]) {
        pre[j] = *(const f32x4*)(xrow[j] + c * 32);
        if (p.gate) pg[j] = *(const f32x4*)(grow[j] + c * 32);
      }
    }
  };
  auto commit = [&](int buf, const f32x4 (&pre)[NI], const f32x4 (&pg)[NI]) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      pg_u32x2 hi, lo;
      pg_split4(pre[j] * pg[j], hi, lo);
      *(pg_u32x2*)(sA[buf][0] + sdst[j]) = hi;
      *(pg_u32x2*)(sA[buf][1] + sdst[j]) = lo;
    }
  };
  // ---- weight fragments: (k-step kk, tile nt, plane) = 64 lanes x 16 B at ((kk * NT32 + nt) * 2 + plane) * 64; a ring
  // of four k-steps in registers, k-step kk + 3 requested while kk runs ----
  const pg_bf16x8* wbase = p.wf + (size_t)min(nt0, p.NT32 - 1) * 2 * 64 + lane;
  const size_t w_kstep = (size_t)p.NT32 * 2 * 64;
  const int w_t1 = on1 ? 2 * 64 : 0;
  const int nks = p.KC * 2;
  auto w_load = [&](int kk, pg_bf16x8 (&w)[2][2]) {
    const pg_bf16x8* q = wbase + (size_t)min(kk, nks - 1) * w_kstep;
    w[0][0] = q[0];
    w[0][1] = q[64];
    w[1][0] = q[w_t1];
    w[1][1] = q[w_t1 + 64];
  };

  f32x16 acc[MW][2];
#pragma unroll
  for (int a = 0; a < MW; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frag = (lane & 31) * PG_ROW + 8 * (lane >> 5);
  pg_bf16x8 w0[2][2], w1[2][2], w2[2][2], w3[2][2];
  auto kstep = [&](int buf, int ks, const pg_bf16x8 (&wc)[2][2]) {
    pg_bf16x8 ah[MW], al[MW];
#pragma unroll
    for (int a = 0; a < MW; ++a) {
      ah[a] = *(const pg_bf16x8*)(sA[buf][0] + a * 32 * PG_ROW + frag + ks * 16);
      al[a] = *(const pg_bf16x8*)(sA[buf][1] + a * 32 * PG_ROW + frag + ks * 16);
    }
#pragma unroll
    for (int a = 0; a < MW; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // weights as the row operand: acc = D[n][m]
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][0], al[a], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][1], ah[a], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][0], ah[a], acc[a][b], 0, 0, 0);
      }
  };
  request(0, preA, pgA);
  request(1, preB, pgB);
  if (on0) {
    w_load(0, w0);
    w_load(1, w1);
    w_load(2, w2);
  }
  commit(0, preA, pgA);
  __syncthreads();
  for (int c = 0; c < p.KC; c += 2) {
    // ---- step c (LDS buffer 0): chunk c + 2 -> set A, chunk c + 1 (set B) -> buffer 1 ----
    request(c + 2, preA, pgA);
    if (on0) {
      w_load(2 * c + 3, w3);
      kstep(0, 0, w0);
      w_load(2 * c + 4, w0);
      kstep(0, 1, w1);
    }
    if (c + 1 >= p.KC) break;
    commit(1, preB, pgB);          // buffer 1: its readers finished before the previous barrier
    __syncthreads();
    // ---- step c + 1 (LDS buffer 1): chunk c + 3 -> set B, chunk c + 2 (set A) -> buffer 0 ----
    request(c + 3, preB, pgB);
    if (on0) {
      w_load(2 * c + 5, w1);
      kstep(1, 0, w2);
      w_load(2 * c + 6, w2);
      kstep(1, 1, w3);
    }
    if (c + 2 < p.KC) {
      commit(0, preA, pgA);
      __syncthreads();
    }
  }
  pw_epilogue<MW>(p, acc, m0, nt0, on0, on1, lane);
}

// ---- long K with few workgroups (K >= 512 and at most one workgroup per CU: the decoder's 256 x 1024 second FFN matrix at
// a thousand rows, EfficientNet's 1248 / 2112 -> 208 / 352 projections at 8 k rows): a 32-k step is then a serialised
// chain - split, LDS write, barrier, fragment reads, 12 MFMAs - of ~1500 clocks with nothing else on the CU to hide it.
// Same kernel with 128-k LDS buffers: one barrier per EIGHT k-steps (48 MFMAs per wave), 32-row tiles. ----
constexpr int PG_ROWL = 136;   // bf16 per LDS row: 128 k + 8 pad

__global__ __launch_bounds__(256, 2) void pw_bf16x3_longk_kernel(PwgP p) {
  constexpr int NI = 4;                            // float4 staging items per thread and 128-k chunk (32 rows x 32 quads)
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][2][32 * PG_ROWL];   // [buffer][hi, lo]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 32;
  const int nt0 = blockIdx.y * 8 + wave * 2;
  const bool on0 = nt0 < p.NT32, on1 = nt0 + 1 < p.NT32;
  // staging: item j = float4 (row, k quad kq + 8 j... ) - a wave covers 8 rows x 32 consecutive floats per item
  const int row = tid >> 3, kq = tid & 7;
  const int m = m0 + row;
  const bool rok = m < p.M;
  const int mc = rok ? m : p.M - 1;
  const float* xrow = p.x + (long)mc * p.ldx + kq * 4;
  const float* grow = p.gate ? p.gate + (long)(mc / p.gate_rows) * p.K + kq * 4 : nullptr;
  const int sdst = row * PG_ROWL + kq * 4;
  const int KC4 = (p.K + 127) / 128;
  f32x4 preA[NI], pgA[NI], preB[NI], pgB[NI];
  auto request = [&](int c, f32x4 (&pre)[NI], f32x4 (&pg)[NI]) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int k = c * 128 + j * 32 + kq * 4;
      pre[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      pg[j] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (c < KC4 && k < p.K && rok) {
        pre[j] = *(const f32x4*)(xrow + c * 128 + j * 32);
        if (p.gate) pg[j] = *(const f32x4*)(grow + c * 128 + j * 32);
      }
    }
  };
  auto commit = [&](int buf, const f32x4 (&pre)[NI], const f32x4 (&pg)[NI]) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      pg_u32x2 hi, lo;
      pg_split4(pre[j] * pg[j], hi, lo);
      *(pg_u32x2*)(sA[buf][0] + sdst + j * 32) = hi;
      *(pg_u32x2*)(sA[buf][1] + sdst + j * 32) = lo;
    }
  };
  const pg_bf16x8* wbase = p.wf + (size_t)min(nt0, p.NT32 - 1) * 2 * 64 + lane;
  const size_t w_kstep = (size_t)p.NT32 * 2 * 64;
  const int w_t1 = on1 ? 2 * 64 : 0;
  const int nks = p.KC * 2;                        // k-steps the packed weights hold (K rounded up to 32)
  auto w_load = [&](int kk, pg_bf16x8 (&w)[2][2]) {
    const pg_bf16x8* q = wbase + (size_t)min(kk, nks - 1) * w_kstep;
    w[0][0] = q[0];
    w[0][1] = q[64];
    w[1][0] = q[w_t1];
    w[1][1] = q[w_t1 + 64];
  };
  f32x16 acc[1][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][b][r] = 0.f;
  const int frag = (lane & 31) * PG_ROWL + 8 * (lane >> 5);
  pg_bf16x8 w0[2][2], w1[2][2], w2[2][2], w3[2][2];
  auto kstep = [&](int buf, int ks, int kk, const pg_bf16x8 (&wc)[2][2]) {
    if (kk >= nks) return;                          // beyond the packed k-steps (K % 128 != 0): zeros anyway
    const pg_bf16x8 ah = *(const pg_bf16x8*)(sA[buf][0] + frag + ks * 16);
    const pg_bf16x8 al = *(const pg_bf16x8*)(sA[buf][1] + frag + ks * 16);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][0], al, acc[0][b], 0, 0, 0);
      acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][1], ah, acc[0][b], 0, 0, 0);
      acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[b][0], ah, acc[0][b], 0, 0, 0);
    }
  };
  // one 128-k chunk from LDS buffer `buf`: eight k-steps, the weight ring three k-steps ahead.  (A ring of eight - every
  // fragment requested a whole chunk ahead - and LDS fragment reads one k-step ahead were measured: no gain, then 256 VGPRs
  // and spills.)
  auto chunk = [&](int buf, int c) {
    if (!on0) return;
    const int k0 = 8 * c;
    w_load(k0 + 3, w3); kstep(buf, 0, k0 + 0, w0);
    w_load(k0 + 4, w0); kstep(buf, 1, k0 + 1, w1);
    w_load(k0 + 5, w1); kstep(buf, 2, k0 + 2, w2);
    w_load(k0 + 6, w2); kstep(buf, 3, k0 + 3, w3);
    w_load(k0 + 7, w3); kstep(buf, 4, k0 + 4, w0);
    w_load(k0 + 8, w0); kstep(buf, 5, k0 + 5, w1);
    w_load(k0 + 9, w1); kstep(buf, 6, k0 + 6, w2);
    w_load(k0 + 10, w2); kstep(buf, 7, k0 + 7, w3);
  };
  request(0, preA, pgA);
  request(1, preB, pgB);
  if (on0) {
    w_load(0, w0);
    w_load(1, w1);
    w_load(2, w2);
  }
  commit(0, preA, pgA);
  __syncthreads();
  for (int c = 0; c < KC4; c += 2) {
    request(c + 2, preA, pgA);
    chunk(0, c);
    if (c + 1 >= KC4) break;
    commit(1, preB, pgB);
    __syncthreads();
    request(c + 3, preB, pgB);
    chunk(1, c + 1);
    if (c + 2 < KC4) {
      commit(0, preA, pgA);
      __syncthreads();
    }
  }
  pw_epilogue<1>(p, acc, m0, nt0, on0, on1, lane);
}

// ---- weight packing: W(n, k) = w[n * s_n + k * s_k] f32 -> fragment order, split into bf16 hi / lo (RNE); K, N padded
// with zeros.  One launch packs a whole table of layers (blockIdx.y = layer): the training step repacks every weight
// after each optimiser update. ----
struct PwPackDesc {
  const float* w; __bf16* out;
  long s_n, s_k;
  int N, K;
};
// One thread = one lane's fragment of one (k-group, column tile): 8 consecutive k of one column -> 16 bytes of the hi plane
// and 16 of the lo plane (the first form wrote one bf16 per thread: 454 us per training step for the 10.7 M parameters).
__device__ __forceinline__ void pw_pack_items(const PwPackDesc& d, long first, long stride) {
  typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
  const int NT32 = (d.N + 31) / 32, KS = (d.K + 31) / 32 * 2;
  const long items = (long)KS * NT32 * 64;
  for (long it = first; it < items; it += stride) {
    const int lane = (int)(it & 63);
    const long t = it >> 6;
    const int nt = (int)(t % NT32), kk = (int)(t / NT32);
    const int n = nt * 32 + (lane & 31), k0 = kk * 16 + 8 * (lane >> 5);
    float v[8];
    if (n < d.N && k0 + 8 <= d.K && d.s_k == 1 && ((((uintptr_t)(d.w + (long)n * d.s_n + k0)) & 15) == 0)) {
      const f32x4 a = *(const f32x4*)(d.w + (long)n * d.s_n + k0), b = *(const f32x4*)(d.w + (long)n * d.s_n + k0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (n < d.N && k0 + e < d.K) ? d.w[(long)n * d.s_n + (long)(k0 + e) * d.s_k] : 0.f;
    }
    pk_bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 h = (__bf16)v[e];
      hi[e] = h;
      lo[e] = (__bf16)(v[e] - (float)h);
    }
    *(pk_bf16x8*)(d.out + ((t * 2) * 64 + lane) * 8) = hi;
    *(pk_bf16x8*)(d.out + ((t * 2 + 1) * 64 + lane) * 8) = lo;
  }
}
__global__ void pw_pack_kernel(const PwPackDesc* table) {
  const PwPackDesc d = table[blockIdx.y];
  pw_pack_items(d, blockIdx.x * 256L + threadIdx.x, gridDim.x * 256L);
}
__global__ void pw_pack_one_kernel(PwPackDesc d) {
  pw_pack_items(d, blockIdx.x * 256L + threadIdx.x, gridDim.x * 256L);
}

}  // namespace

extern "C" {

// bytes of the packed weights of an N x K layer
long ac_pw_gemm_packed_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (long)((K + 31) / 32) * 2 * ((N + 31) / 32) * 2 * 64 * 16;
}

int ac_pw_gemm_pack_strided(const float* w, long s_n, long s_k, void* wfrag, int N, int K, void* stream) {
  if (!w || !wfrag || N <= 0 || K <= 0) return AC_ERR_ARG;
  PwPackDesc d;
  d.w = w; d.out = (__bf16*)wfrag; d.s_n = s_n; d.s_k = s_k; d.N = N; d.K = K;
  const long total = ac_pw_gemm_packed_bytes(N, K) / 32;   // items of 8 hi + 8 lo values
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pw_pack_one_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d);
  return ac_check_launch();
}

int ac_pw_gemm_pack(const float* w, void* wfrag, int N, int K, void* stream) {
  return ac_pw_gemm_pack_strided(w, K, 1, wfrag, N, K, stream);
}

// table: `count` records of 40 bytes in DEVICE memory - {const float* w, void* wfrag, long s_n, long s_k, int N, int K}
// (the layout of PwPackDesc); one launch packs them all.
int ac_pw_gemm_pack_table(const void* table, int count, void* stream) {
  if (!table || count <= 0 || count > 65535) return AC_ERR_ARG;
  hipLaunchKernelGGL(pw_pack_kernel, dim3(256, (unsigned)count), dim3(256), 0, (hipStream_t)stream, (const PwPackDesc*)table);
  return ac_check_launch();
}

int ac_pw_gemm_bf16x3_ex(const float* x, long ldx, const void* wfrag, const float* bias, float* y, long ldy, long M, int N,
                         int K, int act, float beta, const float* gate, int gate_rows, float drop_p,
                         unsigned long long drop_seed, const unsigned long long* seed_dev, long row0, void* stream) {
  if (!x || !wfrag || !y || M <= 0 || M > 2147483647L || N <= 0 || K <= 0 || (K & 3) || (N & 3) || act < 0 || act > 2 ||
      ldx < K || ldy < N || (ldx & 3) || (ldy & 3) || drop_p < 0.f || drop_p >= 1.f ||
      (gate && gate_rows <= 0) || ((uintptr_t)x & 15) || ((uintptr_t)wfrag & 15) || ((uintptr_t)y & 15) ||
      (bias && ((uintptr_t)bias & 15)) || (gate && ((uintptr_t)gate & 15)))
    return AC_ERR_ARG;
  PwgP p;
  p.x = x; p.wf = (const pg_bf16x8*)wfrag; p.bias = bias; p.y = y; p.gate = gate;
  p.ldx = ldx; p.ldy = ldy; p.row0 = row0; p.drop = make_drop(drop_p, drop_seed, seed_dev);
  p.M = (int)M; p.N = N; p.K = K; p.KC = (K + 31) / 32; p.NT32 = (N + 31) / 32; p.act = act; p.gate_rows = gate ? gate_rows : 1;
  p.beta = beta;
  const unsigned gy = (unsigned)((p.NT32 + 7) / 8);
  hipStream_t st = (hipStream_t)stream;
  // rows per workgroup: 32 (most workgroups, the shortest dependent chain per step) up to 16 k rows, 64 above (a weight
  // fragment then feeds 6 MFMAs; measured per shape with tools/pw_gemm_bench.py)
  int mw = M >= 16384 ? 2 : 1;
  {
    static int forced = -1;
    if (forced < 0) {
      const char* e = getenv("AUDIOCAPTION_PW_MW");
      forced = e ? atoi(e) : 0;
    }
    if (forced == 1 || forced == 2) mw = forced;
  }
  static int longk = -1;   // AUDIOCAPTION_PW_LONGK=0 / 1 forces the 128-k variant off / on (where K >= 256)
  if (longk < 0) {
    const char* e = getenv("AUDIOCAPTION_PW_LONGK");
    longk = e ? 2 + atoi(e) : 0;
  }
  const bool use_long = longk == 3 ? K >= 256 : (longk == 2 ? false : ((K >= 512 && (M + 31) / 32 * gy <= 320) || (K >= 256 && (M + 31) / 32 * gy <= 200)));
  if (use_long) hipLaunchKernelGGL(pw_bf16x3_longk_kernel, dim3((unsigned)((M + 31) / 32), gy), dim3(256), 0, st, p);
  else if (mw == 2) hipLaunchKernelGGL(pw_bf16x3_kernel<2>, dim3((unsigned)((M + 63) / 64), gy), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(pw_bf16x3_kernel<1>, dim3((unsigned)((M + 31) / 32), gy), dim3(256), 0, st, p);
  return ac_check_launch();
}

int ac_pw_gemm_bf16x3(const float* x, const void* wfrag, const float* bias, float* y, long M, int N, int K, int act,
                      float beta, const float* gate, int gate_rows, void* stream) {
  return ac_pw_gemm_bf16x3_ex(x, K, wfrag, bias, y, N, M, N, K, act, beta, gate, gate_rows, 0.f, 0ull, nullptr, 0, stream);
}

}  // extern "C"
