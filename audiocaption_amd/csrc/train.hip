// Training-step kernels (forward with saved activations, backward, loss, optimiser) on gfx950.
//
// The reference trains the GRU + Transformer decoder on top of the frozen Cnn14 with scheduled sampling
// (base.py:131-137,152-170, transformer_model.py:34-57): step t runs the decoder on a (N, t+1) prefix and keeps
// the logit of the last position.  Here every prefix pass lives in one ROW SPACE (row = one token position of
// one pass, 256 floats per activation), so the forward of step t works on the row range of pass t while the
// whole backward - the passes are independent once the sampled tokens are fixed - runs ONCE over all rows:
// ~90 launches for 21 passes instead of 21 x 90.
//
// Dropout masks are a counter hash (seed, element index): forward and backward regenerate them, nothing is
// stored, and the CPU oracle reproduces them bit for bit (oracle/train_path.py).
#include "ac_common.h"
#include "ac_drop.h"

namespace {

// =========================================================================================================
// General f32 MFMA GEMM:  C[M][N] (ldc) = epilogue( sum_k A(m,k) B(k,n) ) with arbitrary element strides, so the
// same kernel serves X W^T (forward), dY W (input gradients) and dY^T X (weight gradients, split-K with
// atomic accumulation).  64x64 tile, 4 waves x (32x32 via v_mfma_f32_32x32x2_f32), K chunks of 16 through LDS.
// =========================================================================================================
struct GemmP {
  const float* A; long sam, sak;
  const float* B; long sbk, sbn;
  float* C; long ldc;
  int M, N, K;
  const float* bias;
  int relu;       // activation: 0 none, 1 ReLU, 2 swish (x * sigmoid(x)), 3 sigmoid
  const float* a_scale; int a_rows;   // optional per-(row group, k) multiplier on A: A(m,k) *= a_scale[(m / a_rows) * K + k]
  float beta;
  int splitk;     // > 1: K is cut into gridDim.z slices, results atomically added to C (beta must be 1)
  Drop drop;      // dropout on the output, element index = (row0 + m) * N + n
  long row0;
};

constexpr int GT = 64, GK = 16, GLD = GT + 1;  // odd pitch: k-fast global loads scatter over all 32 LDS banks when stored k-major
constexpr int GPF = 6;   // K chunks in flight per thread: most of these GEMMs are short (K = 256) and alone on the chip,
                         // so what bounds them is the ~2 us global-load latency per dependent round, not bandwidth

__global__ __launch_bounds__(256) void gemm_general_kernel(GemmP p) {
  __shared__ float As[2][GK][GLD];
  __shared__ float Bs[2][GK][GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * GT, n0 = blockIdx.y * GT;   // M tiles on x: row counts reach millions (grid y <= 65535)
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    const int per = ((p.K + p.splitk - 1) / p.splitk + GK - 1) / GK * GK;
    kbeg = blockIdx.z * per;
    kend = min(p.K, kbeg + per);
    if (kbeg >= kend) return;
  }
  // element -> (row, k) assignment follows whichever stride is 1 so that global loads coalesce
  const bool a_kfast = (p.sak == 1), b_nfast = (p.sbn == 1);
  int am[4], ak[4], bn[4], bk[4], agrp[4];
  const float* ap[4];
  const float* bp[4];
  bool aok[4], bok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * 256;
    am[i] = a_kfast ? (e >> 4) : (e & 63);
    agrp[i] = p.a_scale ? min(m0 + am[i], p.M - 1) / p.a_rows : 0;
    ak[i] = a_kfast ? (e & 15) : (e >> 6);
    bn[i] = b_nfast ? (e & 63) : (e >> 4);
    bk[i] = b_nfast ? (e >> 6) : (e & 15);
    aok[i] = m0 + am[i] < p.M;
    bok[i] = n0 + bn[i] < p.N;
    // out-of-range rows / columns / k are loaded from a clamped (valid) address and zeroed afterwards: the loads stay
    // unconditional, so all of them are in flight at once (a predicated load costs a branch region each)
    ap[i] = p.A + (long)min(m0 + am[i], p.M - 1) * p.sam;
    bp[i] = p.B + (long)min(n0 + bn[i], p.N - 1) * p.sbn;
  }
  float ar[GPF][4], br[GPF][4];
  const int nchunks = (kend - kbeg + GK - 1) / GK;
  auto load = [&](int c, float (&a)[4], float (&b)[4]) {
    const int k0 = kbeg + c * GK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ka = k0 + ak[i], kb = k0 + bk[i];
      float va = ap[i][(long)min(ka, kend - 1) * p.sak];
      const float vb = bp[i][(long)min(kb, kend - 1) * p.sbk];
      if (p.a_scale) va *= p.a_scale[(long)agrp[i] * p.K + min(ka, kend - 1)];
      a[i] = (aok[i] && ka < kend) ? va : 0.f;
      b[i] = (bok[i] && kb < kend) ? vb : 0.f;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int u = 0; u < GPF; ++u)
    if (u < nchunks) load(u, ar[u], br[u]);
  for (int c0 = 0; c0 < nchunks; c0 += GPF) {
#pragma unroll
    for (int u = 0; u < GPF; ++u) {
      const int c = c0 + u;
      if (c < nchunks) {
        const int buf = u & 1;   // GPF is even, so the parity of c is the parity of u
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          As[buf][ak[i]][am[i]] = ar[u][i];
          Bs[buf][bk[i]][bn[i]] = br[u][i];
        }
        if (c + GPF < nchunks) load(c + GPF, ar[u], br[u]);
        lds_barrier();
#pragma unroll
        for (int kk = 0; kk < GK; kk += 2) {
          const float a = As[buf][kk + (lane >> 5)][wm + (lane & 31)];
          const float b = Bs[buf][kk + (lane >> 5)][wn + (lane & 31)];
          acc = mfma32(a, b, acc);
        }
      }
    }
  }
  const int n = n0 + wn + (lane & 31);
  if (n >= p.N) return;
  const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= p.M) continue;
    float* c = p.C + (long)m * p.ldc + n;
    if (p.splitk > 1) {
      atomicAdd(c, acc[r]);
    } else {
      float v = acc[r] + bias;
      if (p.relu == 1) v = fmaxf(v, 0.f);
      else if (p.relu == 2) v = ac_swish_exact(v);
      else if (p.relu == 3) v = ac_sigmoid_exact(v);
      v *= p.drop.mask((uint64_t)(p.row0 + m) * (uint64_t)p.N + (uint64_t)n);
      if (p.beta != 0.f) v += p.beta * *c;
      *c = v;
    }
  }
}

// Large x W^T products (both operands contiguous along k, 16-byte aligned rows): 128x128 workgroup tile, each of the
// 4 waves a 2x2 block of 32x32 MFMA tiles.  Tiles sit in LDS row-major ([row][k], pitch 36 floats: a ds_read_b128 of 16
// consecutive rows touches every bank group exactly twice, the minimum), so one 16-byte LDS read gives a lane the 4
// consecutive k of its row that feed 4 MFMAs (lane l: k offset 4 * (l >> 5) of every 8-k group); global loads and LDS
// stores are 16 bytes wide as well.  The next K chunk is fetched into registers while the current one is multiplied.
constexpr int NT_K = 32, NT_P = NT_K + 4;
// TM = 2: 128x128 tile (2x2 MFMA tiles per wave) for the largest products; TM = 1: 64x64 (one MFMA tile per wave) when
// 128x128 tiles would leave CUs idle.
template <int TM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmP p) {
  constexpr int NT_T = 64 * TM, NL = 2 * TM;
  __shared__ __attribute__((aligned(16))) float As[NT_T][NT_P];
  __shared__ __attribute__((aligned(16))) float Bs[NT_T][NT_P];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * NT_T, n0 = blockIdx.y * NT_T;
  const int wm = (wave & 1) * 32 * TM, wn = (wave >> 1) * 32 * TM;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    const int per = ((p.K + p.splitk - 1) / p.splitk + NT_K - 1) / NT_K * NT_K;
    kbeg = blockIdx.z * per;
    kend = min(p.K, kbeg + per);
    if (kbeg >= kend) return;
  }
  // thread -> NL float4 of each operand tile per chunk: row = (tid >> 3) + 32 * i, k4 = tid & 7
  const int lr = tid >> 3, lk = (tid & 7) * 4;
  const float* ap[NL];
  const float* bp[NL];
  const float* gp[NL];
  bool aok[NL], bok[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int am = m0 + lr + 32 * i, bn = n0 + lr + 32 * i;
    aok[i] = am < p.M;
    bok[i] = bn < p.N;
    ap[i] = p.A + (long)min(am, p.M - 1) * p.sam + lk;
    bp[i] = p.B + (long)min(bn, p.N - 1) * p.sbn + lk;
    gp[i] = p.a_scale ? p.a_scale + (long)(min(am, p.M - 1) / p.a_rows) * p.K + lk : nullptr;
  }
  f32x4 ra[NL], rb[NL];
  auto load = [&](int k0) {
    const bool kok = k0 + lk < kend;                       // K % 4 == 0: a float4 is inside or outside as a whole
    const int kc = kok ? k0 : kbeg;                        // clamped (valid) address, zeroed below
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      f32x4 va = *(const f32x4*)(ap[i] + kc);
      const f32x4 vb = *(const f32x4*)(bp[i] + kc);
      if (gp[i]) va *= *(const f32x4*)(gp[i] + kc);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      ra[i] = (kok && aok[i]) ? va : z;
      rb[i] = (kok && bok[i]) ? vb : z;
    }
  };
  f32x16 acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  load(kbeg);
  const int frow = lane & 31, fk = 4 * (lane >> 5);
  for (int k0 = kbeg; k0 < kend; k0 += NT_K) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      *(f32x4*)(&As[lr + 32 * i][lk]) = ra[i];
      *(f32x4*)(&Bs[lr + 32 * i][lk]) = rb[i];
    }
    lds_barrier();
    if (k0 + NT_K < kend) load(k0 + NT_K);
#pragma unroll
    for (int g8 = 0; g8 < NT_K; g8 += 8) {
      f32x4 a[TM], b[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        a[i] = *(const f32x4*)(&As[wm + 32 * i + frow][g8 + fk]);
        b[i] = *(const f32x4*)(&Bs[wn + 32 * i + frow][g8 + fk]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = mfma32(a[i][q], b[j][q], acc[i][j]);
    }
    lds_barrier();
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int n = n0 + wn + 32 * j + (lane & 31);
    if (n >= p.N) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float* c = p.C + (long)m * p.ldc + n;
        if (p.splitk > 1) {
          atomicAdd(c, acc[i][j][r]);
        } else {
          float v = acc[i][j][r] + bias;
          if (p.relu == 1) v = fmaxf(v, 0.f);
          else if (p.relu == 2) v = ac_swish_exact(v);
          else if (p.relu == 3) v = ac_sigmoid_exact(v);
          v *= p.drop.mask((uint64_t)(p.row0 + m) * (uint64_t)p.N + (uint64_t)n);
          if (p.beta != 0.f) v += p.beta * *c;
          *c = v;
        }
      }
  }
}

// x W^T flavour for launches too small to fill the chip (the forward GEMMs of one decoder pass are <= 44 64x64 tiles
// on 256 CUs, and a CU needs >= 3.7 us of f32 MFMA for one 64x64x256 tile): 32x32 tiles, one per workgroup, so 4x
// more CUs work; the 4 waves of a workgroup take one K-quarter each and are reduced through LDS.  Both operands are
// contiguous along k, so there is no LDS staging: lane l reads 4 consecutive k of row (l & 31) at k offset
// 4 * (l >> 5) inside every 8-k group - one 16-byte load feeds 4 MFMAs - with one exposed load latency and no barrier
// in the K loop.  Needs K % 32 == 0 and 16-byte aligned rows.
constexpr int GS = 4, KT = 32;
__global__ __launch_bounds__(256) void gemm_kk_kernel(GemmP p) {
  __shared__ float red[GS - 1][KT][KT + 1];
  const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const int m0 = blockIdx.x * KT, n0 = blockIdx.y * KT;
  const int kper = p.K / GS;                 // multiple of 8
  const int kb = kq * kper;
  const float* a = p.A + (long)min(m0 + (lane & 31), p.M - 1) * p.sam + kb + 4 * (lane >> 5);
  const float* b = p.B + (long)min(n0 + (lane & 31), p.N - 1) * p.sbn + kb + 4 * (lane >> 5);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int ST = 8;                      // 8-k groups per stage (64 k)
  f32x4 va[2][ST], vb[2][ST];
  const int ngroups = kper / 8;
  auto load = [&](int g0, f32x4 (&xa)[ST], f32x4 (&xb)[ST]) {
#pragma unroll
    for (int g = 0; g < ST; ++g) {
      const int gg = min(g0 + g, ngroups - 1);   // clamped (valid) address; unused groups are skipped below
      xa[g] = *(const f32x4*)(a + 8 * gg);
      xb[g] = *(const f32x4*)(b + 8 * gg);
    }
  };
  load(0, va[0], vb[0]);
  for (int g0 = 0; g0 < ngroups; g0 += 2 * ST) {
    if (g0 + ST < ngroups) load(g0 + ST, va[1], vb[1]);
#pragma unroll
    for (int g = 0; g < ST; ++g)
      if (g0 + g < ngroups) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma32(va[0][g][j], vb[0][g][j], acc);
      }
    if (g0 + 2 * ST < ngroups) load(g0 + 2 * ST, va[0], vb[0]);
#pragma unroll
    for (int g = 0; g < ST; ++g)
      if (g0 + ST + g < ngroups) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma32(va[1][g][j], vb[1][g][j], acc);
      }
  }
  const int col = lane & 31;
  if (kq > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[kq - 1][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][col] = acc[r];
  }
  __syncthreads();
  if (kq != 0) return;
  const int n = n0 + col;
  if (n >= p.N) return;
  const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int m = m0 + row;
    if (m >= p.M) continue;
    float v = ((acc[r] + red[0][row][col]) + (red[1][row][col] + red[2][row][col])) + bias;
    if (p.relu == 1) v = fmaxf(v, 0.f);
    else if (p.relu == 2) v = ac_swish_exact(v);
    else if (p.relu == 3) v = ac_sigmoid_exact(v);
    v *= p.drop.mask((uint64_t)(p.row0 + m) * (uint64_t)p.N + (uint64_t)n);
    float* c = p.C + (long)m * p.ldc + n;
    if (p.beta != 0.f) v += p.beta * *c;
    *c = v;
  }
}

// =========================================================================================================
// Split-bf16 GEMM ("bf16x3"): the same contract as gemm_general_kernel - C = epilogue(sum_k A(m,k) B(k,n)), each operand
// contiguous along k OR along its row dimension - at 5x the f32 matrix rate.  Every f32 operand is split when it is
// staged into LDS, x = hi + lo (bf16 each, RNE; 16 significant bits together), and a product is three bf16 MFMAs
// (lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16, f32 accumulation): 2^-16 relative operand error instead of exact
// f32, the arithmetic of the split-bf16 conv tier.  It serves the backward GEMMs of the training step (dY W, dY^T X
// with split-K atomics; 7 392 rows) and the teacher-forced forward GEMMs, which is where the decoder's matrix work is.
//
// Workgroup tile (64 TM) x (64 TN), 2x2 waves, each wave TM x TN MFMA tiles of 32x32; K chunks of 32.  LDS holds, per
// operand, a hi and a lo plane of [row][32 k + 8 pad] bf16 (80-byte rows: a lane's fragment - 8 consecutive k of its
// row - is one 16-byte read).  Waves 0-1 stage A, waves 2-3 stage B; a thread's share of the NEXT chunk (8 float4) is
// requested right after the barrier and lands while the 24 (TM = TN = 2) MFMAs of the current chunk run.
//   operand contiguous along k     : item = (row, 8-k group): two 16-byte loads, one 16-byte LDS store per plane
//   operand contiguous along rows  : item = (4 rows, 8-k group): eight 16-byte loads (4 rows at one k each), transposed
//                                    in registers, one 16-byte LDS store per row and plane
// =========================================================================================================
typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gb_u32x4 __attribute__((ext_vector_type(4)));
constexpr int GB_ROW = 40;   // bf16 elements per LDS row

__device__ __forceinline__ unsigned gb_cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// 8 floats -> 8 bf16 hi (4 dwords) + 8 bf16 lo
__device__ __forceinline__ void gb_split8(const float (&x)[8], gb_u32x4& hi, gb_u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned h = gb_cvt_pk_bf16(x[2 * i], x[2 * i + 1]);
    const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
    hi[i] = h;
    lo[i] = gb_cvt_pk_bf16(x[2 * i] - h0, x[2 * i + 1] - h1);
  }
}

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmP p) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int ROWS = BM > BN ? BM : BN;          // staging threads are laid out for the larger operand tile
  __shared__ __attribute__((aligned(16))) __bf16 sA[2][BM * GB_ROW];   // [hi, lo]
  __shared__ __attribute__((aligned(16))) __bf16 sB[2][BN * GB_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    const int per = ((p.K + p.splitk - 1) / p.splitk + 31) / 32 * 32;
    kbeg = blockIdx.z * per;
    kend = min(p.K, kbeg + per);
    if (kbeg >= kend) return;
  }
  // ---- staging role of this thread: operand (A: waves 0-1, B: waves 2-3), layout, items ----
  const bool isB = tid >= 128;
  const int u = tid & 127;
  const float* base = isB ? p.B : p.A;
  const long s_row = isB ? p.sbn : p.sam, s_k = isB ? p.sbk : p.sak;
  const int r0 = isB ? n0 : m0, rmax = isB ? p.N : p.M, rows = isB ? BN : BM;
  const bool kfast = s_k == 1;
  __bf16* dst_hi = isB ? sB[0] : sA[0];
  __bf16* dst_lo = isB ? sB[1] : sA[1];
  (void)ROWS;
  // optional per-(row group, k) multiplier on A (the squeeze-excite gate of EfficientNet's projection convs), k-fast A only
  const bool scaled = !isB && p.a_scale != nullptr;
  float pre[8][4];   // scalar registers only (every index below is a compile-time constant after unrolling)
  float gate[8][4];
  auto ld4 = [&](float (&d)[4], const float* src, bool ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *(const float4*)src;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  };
  auto request = [&](int k0) {
    if (kfast) {
      // items (row, kg): idx = u + 128 j, row = idx / 4, kg = idx % 4; rows / 32 items per thread
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = u + 128 * j, row = idx >> 2, kg = idx & 3;
        const bool ok = row < rows && r0 + row < rmax;
        const int gr = min(r0 + row, rmax - 1), gk = k0 + kg * 8;
        const float* src = base + (long)gr * s_row;
        ld4(pre[2 * j], src + min(gk, kend - 4), ok && gk < kend);
        ld4(pre[2 * j + 1], src + min(gk + 4, kend - 4), ok && gk + 4 < kend);
        if (scaled) {
          const float* gs = p.a_scale + (long)(gr / p.a_rows) * p.K;
          ld4(gate[2 * j], gs + min(gk, kend - 4), ok && gk < kend);
          ld4(gate[2 * j + 1], gs + min(gk + 4, kend - 4), ok && gk + 4 < kend);
        }
      }
    } else {
      // item (4 rows, kg): rg = u % (rows / 4) (consecutive lanes -> consecutive rows), kg = u / (rows / 4): one item per
      // thread for a 128-row tile, the threads u < 64 for a 64-row tile
      const int rgs = rows >> 2;
      const int rg = u % rgs, kg = u / rgs;
      const int grow = r0 + rg * 4;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int gk = k0 + kg * 8 + kk;
        const bool in = kg < 4 && gk < kend && grow < rmax;       // rmax % 4 == 0 for row-contiguous operands (launcher)
        ld4(pre[kk], base + (long)min(gk, kend - 1) * s_k + min(grow, rmax - 4), in);
      }
    }
  };
  auto commit = [&]() {
    if (kfast) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = u + 128 * j, row = idx >> 2, kg = idx & 3;
        if (row >= rows) continue;
        float x[8] = {pre[2 * j][0], pre[2 * j][1], pre[2 * j][2], pre[2 * j][3],
                      pre[2 * j + 1][0], pre[2 * j + 1][1], pre[2 * j + 1][2], pre[2 * j + 1][3]};
        if (scaled) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { x[e] *= gate[2 * j][e]; x[4 + e] *= gate[2 * j + 1][e]; }
        }
        gb_u32x4 hi, lo;
        gb_split8(x, hi, lo);
        *(gb_u32x4*)(dst_hi + row * GB_ROW + kg * 8) = hi;
        *(gb_u32x4*)(dst_lo + row * GB_ROW + kg * 8) = lo;
      }
    } else {
      const int rgs = rows >> 2;
      const int rg = u % rgs, kg = u / rgs;
      if (kg < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x[8] = {pre[0][i], pre[1][i], pre[2][i], pre[3][i], pre[4][i], pre[5][i], pre[6][i], pre[7][i]};
          gb_u32x4 hi, lo;
          gb_split8(x, hi, lo);
          *(gb_u32x4*)(dst_hi + (rg * 4 + i) * GB_ROW + kg * 8) = hi;
          *(gb_u32x4*)(dst_lo + (rg * 4 + i) * GB_ROW + kg * 8) = lo;
        }
      }
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int frag = (lane & 31) * GB_ROW + 8 * (lane >> 5);
  const int nchunks = (kend - kbeg + 31) / 32;
  request(kbeg);
  for (int c = 0; c < nchunks; ++c) {
    if (c > 0) lds_barrier();          // every wave is done with the previous chunk's tiles
    commit();
    lds_barrier();
    if (c + 1 < nchunks) request(kbeg + (c + 1) * 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      gb_bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        ah[a] = *(const gb_bf16x8*)(sA[0] + ((wm * TM + a) * 32) * GB_ROW + frag + ks * 16);
        al[a] = *(const gb_bf16x8*)(sA[1] + ((wm * TM + a) * 32) * GB_ROW + frag + ks * 16);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        bh[b] = *(const gb_bf16x8*)(sB[0] + ((wn * TN + b) * 32) * GB_ROW + frag + ks * 16);
        bl[b] = *(const gb_bf16x8*)(sB[1] + ((wn * TN + b) * 32) * GB_ROW + frag + ks * 16);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
  }
  // ---- epilogue: the same as gemm_general_kernel's ----
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + (wn * TN + b) * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float* cp = p.C + (long)m * p.ldc + n;
        if (p.splitk > 1) {
          atomicAdd(cp, acc[a][b][r]);
        } else {
          float v = acc[a][b][r] + bias;
          if (p.relu == 1) v = fmaxf(v, 0.f);
          else if (p.relu == 2) v = ac_swish_fast(v);
          else if (p.relu == 3) v = ac_sigmoid_fast(v);
          v *= p.drop.mask((uint64_t)(p.row0 + m) * (uint64_t)p.N + (uint64_t)n);
          if (p.beta != 0.f) v += p.beta * *cp;
          *cp = v;
        }
      }
    }
  }
}

// ---- elementwise dropout (GRU inter-layer, Cnn14 block outputs): y = x * mask(idx0 + i) -----------------------
__global__ void dropout_kernel(const float* x, float* y, long n, Drop d, long idx0) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = x[i] * d.mask((uint64_t)(idx0 + i));
}

// g *= scale where the (post-ReLU, post-dropout) activation h is positive (FFN hidden backward)
__global__ void mask_pos_scale_kernel(float* g, const float* h, long n, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    g[i] = h[i] > 0.f ? g[i] * scale : 0.f;
}

// ---- prefix tokens of one scheduled-sampling pass (transformer_model.py:44-52) --------------------------------
// word[row0 + n*L + l] = use_cap ? cap[n][l] : (l == 0 ? start : seq[n][l-1])
__global__ void build_prefix_kernel(const long long* cap, int cap_ld, const int* seq, int seq_ld, const int* use_cap,
                                    int t, int start_idx, int* word, long row0, int N, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * L) return;
  const int n = i / L, l = i % L;
  int w;
  if (use_cap[t]) w = (int)cap[(long)n * cap_ld + l];
  else w = l == 0 ? start_idx : seq[(long)n * seq_ld + l - 1];
  word[row0 + i] = w;
}

// ---- embedding + positional encoding with both dropouts (transformer_decoder.py:88-90) ------------------------
// x[row][c] = dropB( dropA(E[word[row]][c]) * sqrt(d) + pe[pos[row]][c] )
__global__ void embed_fwd_kernel(const float* emb, const float* pe, const int* word, const int* pos, float* x,
                                 long row0, long rows, int d, float sqrt_d, Drop da, Drop db) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= rows * d) return;
  const long row = row0 + i / d;
  const int c = (int)(i % d);
  const uint64_t idx = (uint64_t)row * d + c;
  const float e = emb[(long)word[row] * d + c] * da.mask(idx);
  x[row * d + c] = (e * sqrt_d + pe[(long)pos[row] * d + c]) * db.mask(idx);
}
__global__ void embed_bwd_kernel(const float* dx, const int* word, float* demb, long rows, int d, float sqrt_d,
                                 Drop da, Drop db) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= rows * d) return;
  const long row = i / d;
  const int c = (int)(i % d);
  const uint64_t idx = (uint64_t)i;
  const float g = dx[i] * db.mask(idx) * sqrt_d * da.mask(idx);
  if (g != 0.f) atomicAdd(demb + (long)word[row] * d + c, g);
}

// =========================================================================================================
// y = LayerNorm(res + drop(x)) over 256 columns, one wave per row; `pre` (= res + drop(x)) is kept for backward.
// =========================================================================================================
constexpr int DM = 256;

__global__ __launch_bounds__(256) void dropadd_ln_fwd_kernel(const float* x, const float* res, const float* gamma,
                                                             const float* beta, float* pre, float* y, long row0,
                                                             long rows, long xmod, Drop d, float eps) {
  const int lane = threadIdx.x & 63;
  const long r = row0 + blockIdx.x * 4L + (threadIdx.x >> 6);
  if (r >= row0 + rows) return;
  const long xr = xmod > 0 ? r % xmod : r;   // x may be shared by `replicas` of xmod rows (decoder memory)
  const f32x4 xv = *(const f32x4*)(x + xr * DM + lane * 4);
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = xv[j] * d.mask((uint64_t)r * DM + lane * 4 + j);
  if (res) {
    const f32x4 rv = *(const f32x4*)(res + r * DM + lane * 4);
    v += rv;
  }
  *(f32x4*)(pre + r * DM + lane * 4) = v;
  const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / DM);
  const f32x4 c = v - mean;
  const float var = wave_sum(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]) * (1.0f / DM);
  const float rstd = rsqrtf(var + eps);
  const f32x4 g = *(const f32x4*)(gamma + lane * 4), b = *(const f32x4*)(beta + lane * 4);
  *(f32x4*)(y + r * DM + lane * 4) = c * rstd * g + b;
}

// dy -> dpre (written to dres, or added to it when accumulate != 0), dsub = dpre * mask [* (relu_src > 0)] -> dx,
// dgamma / dbeta accumulated with one atomic per column per block (LN_ROWS rows per block).
constexpr int LN_ROWS = 32;
__global__ __launch_bounds__(256) void dropadd_ln_bwd_kernel(const float* dy, const float* pre, const float* gamma,
                                                             float* dx, float* dres, int accumulate,
                                                             const float* relu_src, long relu_mod, float* dgamma,
                                                             float* dbeta, long rows, Drop d, float eps) {
  __shared__ float sg[4][DM], sb[4][DM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 g = *(const f32x4*)(gamma + lane * 4);
  f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < LN_ROWS; i += 4) {
    const long r = blockIdx.x * (long)LN_ROWS + i;
    if (r >= rows) break;
    const f32x4 v = *(const f32x4*)(pre + r * DM + lane * 4);
    const f32x4 dyv = *(const f32x4*)(dy + r * DM + lane * 4);
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / DM);
    const f32x4 c = v - mean;
    const float var = wave_sum(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]) * (1.0f / DM);
    const float rstd = rsqrtf(var + eps);
    const f32x4 xh = c * rstd;
    const f32x4 dxh = dyv * g;
    const float m1 = wave_sum(dxh[0] + dxh[1] + dxh[2] + dxh[3]) * (1.0f / DM);
    const float m2 = wave_sum(dxh[0] * xh[0] + dxh[1] * xh[1] + dxh[2] * xh[2] + dxh[3] * xh[3]) * (1.0f / DM);
    const f32x4 dp = (dxh - m1 - xh * m2) * rstd;
    ag += dyv * xh;
    ab += dyv;
    if (dres) {
      f32x4 o = dp;
      if (accumulate) o += *(const f32x4*)(dres + r * DM + lane * 4);
      *(f32x4*)(dres + r * DM + lane * 4) = o;
    }
    if (dx) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = dp[j] * d.mask((uint64_t)r * DM + lane * 4 + j);
      if (relu_src) {
        const f32x4 a = *(const f32x4*)(relu_src + (relu_mod > 0 ? r % relu_mod : r) * DM + lane * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] > 0.f ? o[j] : 0.f;
      }
      *(f32x4*)(dx + r * DM + lane * 4) = o;
    }
  }
  *(f32x4*)(&sg[wave][lane * 4]) = ag;
  *(f32x4*)(&sb[wave][lane * 4]) = ab;
  __syncthreads();
  const int c = threadIdx.x;
  atomicAdd(dgamma + c, (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]));
  atomicAdd(dbeta + c, (sb[0][c] + sb[1][c]) + (sb[2][c] + sb[3][c]));
}

// =========================================================================================================
// Multi-head attention over short sequences, one workgroup per (sequence, head); head_dim 64.
// Sequence s has L[s] queries at rows qrow0[s].. and Tk keys at rows krow0[s]..;  self-attention: key j is
// visible to query i iff j <= i and word[krow0 + j] != pad (causal + tgt_key_padding_mask,
// transformer_decoder.py:92-97); cross-attention: iff j < klen[s] (memory_key_padding_mask).
// P (softmax, before dropout) is stored for backward at P[(s * nhead + h) * LMAX * TKMAX + i * TKMAX + j].
// =========================================================================================================
constexpr int HD = 64, HDP = HD + 1;   // LDS row pitch of the Q/K/V tiles
constexpr int ATT_LDS_MAX = 160 * 1024;

struct AttnP {
  const float* q; long ldq;
  const float* k; long ldk;
  const float* v; long ldv;
  float* o; long ldo;
  float* P; int pl, ptk;         // P strides: rows pl (>= max L), cols ptk (>= max Tk)
  const int* qrow0; const int* qlen;
  const int* krow0; const int* klen;   // klen: number of key rows (self: = qlen)
  const int* kvalid;                   // cross: valid keys; self: null
  const int* word; int pad_idx;        // self: key padding from the tokens; cross: null
  int causal;
  int seq0;                            // first sequence of this launch
  float scale;
  int lmax, tkmax;                     // launch-wide maxima of qlen / klen (size the LDS carve-up)
  Drop drop;                           // on P; index = ((s*nhead + h) * pl + i) * ptk + j
  // backward only
  const float* dout; long lddo;
  float* dq; long lddq;
  float* dk; long lddk;
  float* dv; long lddv;
};

__global__ __launch_bounds__(256) void attn_seq_fwd_kernel(AttnP p) {
  extern __shared__ float att_smem[];
  const int PP = p.tkmax + 1;
  float (*sq)[HDP] = (float (*)[HDP])att_smem;
  float (*sk)[HDP] = sq + p.lmax;
  float (*sv)[HDP] = sk + p.tkmax;
  float* sp_ = (float*)(sv + p.tkmax);
#define sp(i, j) sp_[(i) * PP + (j)]
  const int s = p.seq0 + blockIdx.x, h = blockIdx.y, nh = gridDim.y, tid = threadIdx.x;
  const int L = p.qlen[s], Tk = p.klen[s];
  const long q0 = p.qrow0[s], k0 = p.krow0[s];
  for (int e = tid; e < L * HD; e += 256) sq[e / HD][e % HD] = p.q[(q0 + e / HD) * p.ldq + h * HD + e % HD];
  for (int e = tid; e < Tk * HD; e += 256) {
    sk[e / HD][e % HD] = p.k[(k0 + e / HD) * p.ldk + h * HD + e % HD];
    sv[e / HD][e % HD] = p.v[(k0 + e / HD) * p.ldv + h * HD + e % HD];
  }
  __syncthreads();
  const int nvalid = p.kvalid ? p.kvalid[s] : Tk;
  for (int e = tid; e < L * Tk; e += 256) {
    const int i = e / Tk, j = e % Tk;
    bool ok = j < nvalid;
    if (p.causal && j > i) ok = false;
    if (p.word && p.word[k0 + j] == p.pad_idx) ok = false;
    float a = 0.f;
#pragma unroll 16
    for (int c = 0; c < HD; ++c) a = fmaf(sq[i][c], sk[j][c], a);
    sp(i, j) = ok ? a * p.scale : -INFINITY;
  }
  __syncthreads();
  // softmax: one wave per query row
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < L; i += 4) {
    float m = -INFINITY;
    for (int j = lane; j < Tk; j += 64) m = fmaxf(m, sp(i, j));
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < Tk; j += 64) {
      const float e = expf(sp(i, j) - m);
      sp(i, j) = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    const long pb = ((long)(s * nh + h) * p.pl + i) * p.ptk;
    for (int j = lane; j < Tk; j += 64) {
      const float pr = sp(i, j) * inv;
      p.P[pb + j] = pr;
      sp(i, j) = pr * p.drop.mask((uint64_t)(pb + j));
    }
  }
  __syncthreads();
  for (int e = tid; e < L * HD; e += 256) {
    const int i = e / HD, c = e % HD;
    float a = 0.f;
    for (int j = 0; j < Tk; ++j) a = fmaf(sp(i, j), sv[j][c], a);
    p.o[(q0 + i) * p.ldo + h * HD + c] = a;
  }
}

__global__ __launch_bounds__(256) void attn_seq_bwd_kernel(AttnP p) {
  extern __shared__ float att_smem[];
  const int PP = p.tkmax + 1;
  float (*sq)[HDP] = (float (*)[HDP])att_smem;
  float (*sk)[HDP] = sq + p.lmax;
  float (*sv)[HDP] = sk + p.tkmax;
  float (*sdo)[HDP] = sv + p.tkmax;
  float* sp_ = (float*)(sdo + p.lmax);   // dropped P, then dP, then dS
  float* spd_ = sp_ + p.lmax * PP;       // P
#define spd(i, j) spd_[(i) * PP + (j)]
  const int s = p.seq0 + blockIdx.x, h = blockIdx.y, nh = gridDim.y, tid = threadIdx.x;
  const int L = p.qlen[s], Tk = p.klen[s];
  const long q0 = p.qrow0[s], k0 = p.krow0[s];
  for (int e = tid; e < L * HD; e += 256) {
    sq[e / HD][e % HD] = p.q[(q0 + e / HD) * p.ldq + h * HD + e % HD];
    sdo[e / HD][e % HD] = p.dout[(q0 + e / HD) * p.lddo + h * HD + e % HD];
  }
  for (int e = tid; e < Tk * HD; e += 256) {
    sk[e / HD][e % HD] = p.k[(k0 + e / HD) * p.ldk + h * HD + e % HD];
    sv[e / HD][e % HD] = p.v[(k0 + e / HD) * p.ldv + h * HD + e % HD];
  }
  for (int e = tid; e < L * Tk; e += 256) {
    const int i = e / Tk, j = e % Tk;
    const long pi = ((long)(s * nh + h) * p.pl + i) * p.ptk + j;
    const float pr = p.P[pi];
    spd(i, j) = pr;
    sp(i, j) = pr * p.drop.mask((uint64_t)pi);
  }
  __syncthreads();
  // dV[j][c] = sum_i Pdrop[i][j] dO[i][c]
  for (int e = tid; e < Tk * HD; e += 256) {
    const int j = e / HD, c = e % HD;
    float a = 0.f;
    for (int i = 0; i < L; ++i) a = fmaf(sp(i, j), sdo[i][c], a);
    p.dv[(k0 + j) * p.lddv + h * HD + c] = a;
  }
  __syncthreads();
  // dPdrop[i][j] = dO[i] . V[j]; dP = dPdrop * mask
  for (int e = tid; e < L * Tk; e += 256) {
    const int i = e / Tk, j = e % Tk;
    float a = 0.f;
#pragma unroll 16
    for (int c = 0; c < HD; ++c) a = fmaf(sdo[i][c], sv[j][c], a);
    const long pi = ((long)(s * nh + h) * p.pl + i) * p.ptk + j;
    sp(i, j) = a * p.drop.mask((uint64_t)pi);
  }
  __syncthreads();
  // dS = P * (dP - sum_j dP P)
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = wave; i < L; i += 4) {
    float dot = 0.f;
    for (int j = lane; j < Tk; j += 64) dot += sp(i, j) * spd(i, j);
    dot = wave_sum(dot);
    for (int j = lane; j < Tk; j += 64) sp(i, j) = spd(i, j) * (sp(i, j) - dot) * p.scale;
  }
  __syncthreads();
  for (int e = tid; e < L * HD; e += 256) {
    const int i = e / HD, c = e % HD;
    float a = 0.f;
    for (int j = 0; j < Tk; ++j) a = fmaf(sp(i, j), sk[j][c], a);
    p.dq[(q0 + i) * p.lddq + h * HD + c] = a;
  }
  for (int e = tid; e < Tk * HD; e += 256) {
    const int j = e / HD, c = e % HD;
    float a = 0.f;
    for (int i = 0; i < L; ++i) a = fmaf(sp(i, j), sq[i][c], a);
    p.dk[(k0 + j) * p.lddk + h * HD + c] = a;
  }
}

// ---- row gather / scatter-add through an index list (classifier on the last position of each pass) ---------
__global__ void gather_rows_kernel(const float* src, const int* index, float* dst, long nrows, int C) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= nrows * C) return;
  dst[i] = src[(long)index[i / C] * C + i % C];
}
__global__ void scatter_add_rows_kernel(const float* src, const int* index, float* dst, long nrows, int C) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= nrows * C) return;
  dst[(long)index[i / C] * C + i % C] += src[i];   // the index list has no duplicates
}

// out[r] = sum_t x[t * n + r]  (gradient of an activation shared by `reps` replicas)
__global__ void sum_replicas_kernel(const float* x, float* out, long n, int reps) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int t = 0; t < reps; ++t) a += x[(long)t * n + i];
    out[i] = a;
  }
}
// Cnn14 head in train mode: attn[b][h][c] = mean_w x[(b*Hp + h)][w][c], h < H (cnn_encoder.py:443-444)
__global__ void rows_mean_w_kernel(const float* x, float* out, int B, int Hp, int Hv, int W, int C) {
  const long n = (long)B * Hv * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long bh = i / C;
    const int h = (int)(bh % Hv);
    const long b = bh / Hv;
    const float* src = x + ((b * Hp + h) * W) * (long)C + c;
    float a = 0.f;
    for (int w = 0; w < W; ++w) a += src[(long)w * C];
    out[i] = a / (float)W;
  }
}
// dst[b][c][r] = src[b][r][c]
__global__ void transpose_kernel(const float* src, float* dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const float* s = src + (long)blockIdx.z * rows * cols;
  float* d = dst + (long)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? s[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) d[(long)c * rows + r] = tile[threadIdx.x][i];
  }
}

// SpecAugment on the bn0-normalised log-mel (the reference masks BEFORE bn0, cnn_encoder.py:423-429, so a masked bin
// becomes bn0(0) = shift[mel]): x [B*rows_per_clip][F], stripes [B][n_time + n_freq][2] = (begin, length).
__global__ void specaug_kernel(float* x, const int* stripes, const float* fill, int B, int rows_per_clip, int T, int F,
                               int n_time, int n_freq) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * T * F) return;
  const int f = (int)(i % F);
  const int t = (int)((i / F) % T);
  const int b = (int)(i / ((long)F * T));
  const int* st = stripes + (long)b * (n_time + n_freq) * 2;
  bool hit = false;
  for (int k = 0; k < n_time; ++k) hit |= t >= st[2 * k] && t < st[2 * k] + st[2 * k + 1];
  for (int k = 0; k < n_freq; ++k) hit |= f >= st[2 * (n_time + k)] && f < st[2 * (n_time + k)] + st[2 * (n_time + k) + 1];
  if (hit) x[((long)b * rows_per_clip + t) * F + f] = fill ? fill[f] : 0.f;
}

// ---- column sums (bias gradients): out[n] += sum_m X[m][n] ---------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, long ld, float* out, long M, int N) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;
  __shared__ float sh[4][64];
  float a = 0.f;
  if (n < N)
    for (long m = blockIdx.y * 4L + part; m < M; m += gridDim.y * 4L) a += x[m * ld + n];
  sh[part][threadIdx.x & 63] = a;
  __syncthreads();
  if (part == 0 && n < N) atomicAdd(out + n, (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]));
}

// ---- greedy choice of the next token (base.py:206-209): argmax over the vocabulary, first index on ties -------
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* logit, long ld, int V, int* out, long out_ld) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const float* row = logit + blockIdx.x * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float x = row[v];
    if (x > best) { best = x; bi = v; }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = sv[threadIdx.x + s];
      const int oi = si[threadIdx.x + s];
      if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = o; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x * out_ld] = si[0];
}

// =========================================================================================================
// Label-smoothing cross entropy (loss.py:51-74): row (n, t) contributes -sum_v q_v log_softmax(z)_v when
// t < tgt_len[n], q = confidence on the target, smoothing / (V - 1) elsewhere; mean over the valid rows.
// dlogit = (softmax - q) * gscale on valid rows, 0 elsewhere.  row_loss holds the per-row terms.
// =========================================================================================================
__global__ __launch_bounds__(256) void xent_kernel(const float* logit, const long long* tgt, long tgt_ld,
                                                   const int* tgt_len, int T, int V, float smoothing,
                                                   float* row_loss, float* dlogit, float gscale,
                                                   const float* gscale_dev, int N) {
  __shared__ float red[4];
  const int row = blockIdx.x, n = row / T, t = row % T;
  if (gscale <= 0.f) {  // "mean" with the count taken from tgt_len on the device (graph-replayable)
    int cnt = 0;
    for (int i = 0; i < N; ++i) cnt += min(tgt_len[i], T);
    gscale = 1.0f / (float)cnt;
  }
  const float* z = logit + (long)row * V;
  const bool valid = t < tgt_len[n];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (!valid) {
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    if (dlogit)
      for (int v = threadIdx.x; v < V; v += 256) dlogit[(long)row * V + v] = 0.f;
    return;
  }
  float m = -INFINITY;
  for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, z[v]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f, sz = 0.f;
  for (int v = threadIdx.x; v < V; v += 256) {
    se += expf(z[v] - m);
    sz += z[v];
  }
  se = wave_sum(se);
  if (lane == 0) red[wave] = se;
  __syncthreads();
  se = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  sz = wave_sum(sz);
  if (lane == 0) red[wave] = sz;
  __syncthreads();
  sz = (red[0] + red[1]) + (red[2] + red[3]);
  const float lse = m + logf(se);
  const int target = (int)tgt[(long)n * tgt_ld + t];
  const float off = smoothing / (float)(V - 1), conf = 1.0f - smoothing;
  if (threadIdx.x == 0) {
    // -sum_v q_v (z_v - lse) = lse - off * (sum z - z_tgt) - conf * z_tgt   (sum q = 1)
    row_loss[row] = lse - off * (sz - z[target]) - conf * z[target];
  }
  if (dlogit) {
    if (gscale_dev) gscale *= gscale_dev[0];
    const float inv = 1.0f / se;
    for (int v = threadIdx.x; v < V; v += 256) {
      const float q = v == target ? conf : off;
      dlogit[(long)row * V + v] = (expf(z[v] - m) * inv - q) * gscale;
    }
  }
}
__global__ void sum_scale_kernel(const float* x, long n, float scale, float* out, const int* tgt_len, int N, int T) {
  __shared__ float red[4];
  if (scale <= 0.f) {
    int cnt = 0;
    for (int i = 0; i < N; ++i) cnt += min(tgt_len[i], T);
    scale = 1.0f / (float)cnt;
  }
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) a += x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// =========================================================================================================
// GRU layer, training flavour: the forward also keeps (r, z, n, W_hn h + b_hn) per step; the backward walks the
// steps of each (clip, direction) in reverse with dh in LDS and emits the gate gradients, from which the weight
// gradients are plain GEMMs over all (clip, step) rows.
// =========================================================================================================
constexpr int H = 256;
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Same residency scheme as csrc/gru.hip's inference kernel (the first 96 k of every column in registers, the next 48 k
// in LDS, the rest streamed per step) and the same k-ascending accumulation order: bit-identical outputs.
constexpr int GRU_KREG = 96, GRU_KLDS = 48;
__global__ __launch_bounds__(768) void gru_train_fwd_kernel(const float* gx, const float* whhT, const float* bhh,
                                                            const int* lens, float* out, float* save, int T) {
  __shared__ __attribute__((aligned(16))) float sh[H];
  __shared__ float sg[3 * H];
  extern __shared__ __attribute__((aligned(16))) float swl[];   // [GRU_KLDS / 4][3H][4]
  const int n = threadIdx.x;
  const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
  int len = lens[b];
  len = len < 0 ? 0 : (len > T ? T : len);
  const float* W = whhT + (size_t)dir * H * 3 * H + (size_t)n * 4;
  const float bias = bhh[dir * 3 * H + n];
  float4 wreg[GRU_KREG / 4];
  if (len > 0) {
#pragma unroll
    for (int q = 0; q < GRU_KREG / 4; ++q) wreg[q] = *(const float4*)(W + (size_t)q * 3 * H * 4);
#pragma unroll
    for (int q = 0; q < GRU_KLDS / 4; ++q)
      *(float4*)(swl + ((size_t)q * 3 * H + n) * 4) = *(const float4*)(W + (size_t)(GRU_KREG / 4 + q) * 3 * H * 4);
  }
  if (n < H) sh[n] = 0.f;
  __syncthreads();
  for (int step = 0; step < len; ++step) {
    const int t = dir ? (len - 1 - step) : step;
    const size_t cell = ((size_t)b * T + t) * 2 + dir;
    const float* gxp = gx + cell * 3 * H;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (n < H) { gr = gxp[n]; gz = gxp[H + n]; gn = gxp[2 * H + n]; }
    float acc = bias;
#pragma unroll
    for (int q = 0; q < GRU_KREG / 4; ++q) {
      const float4 hv = *(const float4*)(sh + 4 * q);
      acc = fmaf(wreg[q].x, hv.x, acc);
      acc = fmaf(wreg[q].y, hv.y, acc);
      acc = fmaf(wreg[q].z, hv.z, acc);
      acc = fmaf(wreg[q].w, hv.w, acc);
    }
#pragma unroll
    for (int q = 0; q < GRU_KLDS / 4; ++q) {
      const float4 hv = *(const float4*)(sh + GRU_KREG + 4 * q);
      const float4 wv = *(const float4*)(swl + ((size_t)q * 3 * H + n) * 4);
      acc = fmaf(wv.x, hv.x, acc);
      acc = fmaf(wv.y, hv.y, acc);
      acc = fmaf(wv.z, hv.z, acc);
      acc = fmaf(wv.w, hv.w, acc);
    }
#pragma unroll 7
    for (int k = GRU_KREG + GRU_KLDS; k < H; k += 4) {
      const float4 hv = *(const float4*)(sh + k);
      const float4 wv = *(const float4*)(W + (size_t)(k >> 2) * 3 * H * 4);
      acc = fmaf(wv.x, hv.x, acc);
      acc = fmaf(wv.y, hv.y, acc);
      acc = fmaf(wv.z, hv.z, acc);
      acc = fmaf(wv.w, hv.w, acc);
    }
    sg[n] = acc;
    __syncthreads();
    if (n < H) {
      const float r = sigmoidf_(gr + sg[n]);
      const float z = sigmoidf_(gz + sg[H + n]);
      const float ghn = sg[2 * H + n];
      const float c = tanhf(gn + r * ghn);
      const float hn = (1.0f - z) * c + z * sh[n];
      sh[n] = hn;
      out[((size_t)b * T + t) * 2 * H + dir * H + n] = hn;
      float* sv = save + cell * 4 * H;
      sv[n] = r; sv[H + n] = z; sv[2 * H + n] = c; sv[3 * H + n] = ghn;
    }
    __syncthreads();
  }
  if (n < H)
    for (int t = len; t < T; ++t) out[((size_t)b * T + t) * 2 * H + dir * H + n] = 0.f;
}

// whh: [2][3H][H] (nn.GRU's own layout, so the transposed product reads it coalesced along k)
__global__ __launch_bounds__(768) void gru_bwd_kernel(const float* dout, const float* out, const float* save,
                                                      const float* whh, const int* lens, float* dgx, float* dgh,
                                                      float* hprev_out, int T) {
  __shared__ float sdh[H];
  __shared__ float sdg[3 * H];
  __shared__ float spart[3][H];
  extern __shared__ float swb[];   // [GRU_KLDS][3][H]: rows 96..143 of every gate block of W_hh
  const int tid = threadIdx.x;
  const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
  int len = lens[b];
  len = len < 0 ? 0 : (len > T ? T : len);
  const float* W = whh + (size_t)dir * 3 * H * H;
  const int part = tid >> 8, k = tid & 255;
  // same residency as the forward: this thread's column of W_hh^T (row index n of its gate block) is kept in
  // registers for n < 96, in LDS for the next 48, and streamed for the rest
  const float* wp = W + (size_t)(part * H) * H + k;
  float wreg[GRU_KREG];
  if (len > 0) {
#pragma unroll
    for (int n = 0; n < GRU_KREG; ++n) wreg[n] = wp[(size_t)n * H];
#pragma unroll
    for (int j = 0; j < GRU_KLDS; ++j) swb[(j * 3 + part) * H + k] = wp[(size_t)(GRU_KREG + j) * H];
  }
  if (tid < H) sdh[tid] = 0.f;
  __syncthreads();
  for (int step = len - 1; step >= 0; --step) {
    const int t = dir ? (len - 1 - step) : step;
    const size_t cell = ((size_t)b * T + t) * 2 + dir;
    float direct = 0.f;
    if (tid < H) {
      const int n = tid;
      const float* sv = save + cell * 4 * H;
      const float r = sv[n], z = sv[H + n], c = sv[2 * H + n], ghn = sv[3 * H + n];
      float hp = 0.f;
      if (step > 0) {
        const int tp = dir ? t + 1 : t - 1;
        hp = out[((size_t)b * T + tp) * 2 * H + dir * H + n];
      }
      const float dht = dout[((size_t)b * T + t) * 2 * H + dir * H + n] + sdh[n];
      const float dn = dht * (1.0f - z) * (1.0f - c * c);
      const float dz = dht * (hp - c) * z * (1.0f - z);
      const float dr = dn * ghn * r * (1.0f - r);
      const float dghn = dn * r;
      direct = dht * z;
      float* gxo = dgx + cell * 3 * H;
      float* gho = dgh + cell * 3 * H;
      gxo[n] = dr; gxo[H + n] = dz; gxo[2 * H + n] = dn;
      gho[n] = dr; gho[H + n] = dz; gho[2 * H + n] = dghn;
      hprev_out[cell * H + n] = hp;
      sdg[n] = dr; sdg[H + n] = dz; sdg[2 * H + n] = dghn;
    }
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int n = 0; n < GRU_KREG; ++n) a = fmaf(sdg[part * H + n], wreg[n], a);
#pragma unroll
    for (int j = 0; j < GRU_KLDS; ++j) a = fmaf(sdg[part * H + GRU_KREG + j], swb[(j * 3 + part) * H + k], a);
#pragma unroll 8
    for (int n = GRU_KREG + GRU_KLDS; n < H; ++n) a = fmaf(sdg[part * H + n], wp[(size_t)n * H], a);
    spart[part][k] = a;
    __syncthreads();
    if (tid < H) sdh[tid] = direct + (spart[0][tid] + spart[1][tid]) + spart[2][tid];
    __syncthreads();
  }
  // padded steps carry no gradient
  for (int t = len; t < T; ++t) {
    const size_t cell = ((size_t)b * T + t) * 2 + dir;
    dgx[cell * 3 * H + tid] = 0.f;
    dgh[cell * 3 * H + tid] = 0.f;
    if (tid < H) hprev_out[cell * H + tid] = 0.f;
  }
}

// =========================================================================================================
// Optimiser (run.py:122-126, torch.optim.Adam with weight_decay = L2 added to the gradient):
//   sumsq -> total norm -> clip coefficient min(1, max_norm / (norm + 1e-6)) -> Adam, all on flat buffers.
// =========================================================================================================
// Deterministic: every workgroup leaves its partial in norm_state[AC_NORM_PARTIALS ..]; the LAST one to finish (ticket
// counter behind them) adds them up in index order.  An atomicAdd per workgroup would make the norm - and through the
// clip coefficient every parameter - differ in the last bit between the ranks of a data-parallel job.
constexpr int AC_NORM_PARTIALS = 4, AC_NORM_MAXBLOCKS = 1024, AC_NORM_TICKET = AC_NORM_PARTIALS + AC_NORM_MAXBLOCKS;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long n, float* st) {
  __shared__ float red[4];
  __shared__ bool last;
  float a = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) a = fmaf(g[i], g[i], a);
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    st[AC_NORM_PARTIALS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __threadfence();
    const unsigned t = atomicAdd((unsigned*)(st + AC_NORM_TICKET), 1u);
    last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float s = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) s += ((volatile float*)st)[AC_NORM_PARTIALS + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    st[0] += (red[0] + red[1]) + (red[2] + red[3]);
    *(unsigned*)(st + AC_NORM_TICKET) = 0u;   // ready for the next call
  }
}
// norm_state[0] = sum of squares in, [1] = total norm out, [2] = clip coefficient out, [3] = 1 when the gradient is
// not finite: the update is then skipped on the device, the reference's `if not torch.isnan(loss)` (run.py:123)
// without a host synchronisation
__global__ void clip_coef_kernel(float* norm_state, float max_norm, float grad_div) {
  const float norm = sqrtf(norm_state[0]) / grad_div;
  norm_state[1] = norm;
  const bool bad = !(norm == norm) || norm > 3.0e38f;
  float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.0f;
  norm_state[2] = bad ? 0.f : (c < 1.0f ? c : 1.0f) / grad_div;
  norm_state[3] = bad ? 1.f : 0.f;
}
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, const float* norm_state, float lr,
                            float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, const int* step_dev) {
  const float coef = norm_state ? norm_state[2] : 1.0f;
  if (norm_state && norm_state[3] != 0.f) return;   // non-finite gradient: leave parameters and moments untouched
  if (step_dev != nullptr) {   // device-side step counter: only APPLIED updates advance the bias correction
    const float t = (float)(*step_dev + 1);
    bc1 = 1.0f - powf(b1, t);
    bc2_sqrt = sqrtf(1.0f - powf(b2, t));
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * coef;
    const float pi = p[i];
    gi = fmaf(wd, pi, gi);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
__global__ void adam_commit_kernel(int* step_dev, const float* norm_state) {
  if (!(norm_state && norm_state[3] != 0.f)) *step_dev += 1;
}
// stochastic weight averaging (train_util.py:233-253 with torch's default avg_fn): avg += (p - avg) / (n_averaged + 1)
__global__ void swa_kernel(float* avg, const float* p, long n, float inv) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float a = avg[i];
    avg[i] = a + (p[i] - a) * inv;
  }
}
__global__ void scale_kernel(float* x, long n, const float* norm_state) {
  const float coef = norm_state[2];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= coef;
}

inline int grid_for(long n, int block = 256, int cap = 65535 * 4) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// ============================================== C ABI ====================================================
// C[M][N] (row pitch ldc) = 0: the starting point of a split-K product with beta == 0.  (A kernel, not hipMemset2DAsync:
// inside a captured graph the 2-D memset node was not replayed and the slices accumulated onto the previous iteration.)
__global__ void zero2d_kernel(float* C, long ldc, int M, int N) {
  const long n = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    C[(i / N) * ldc + i % N] = 0.f;
}
static inline void zero2d(float* C, long ldc, int M, int N, hipStream_t s) {
  hipLaunchKernelGGL(zero2d_kernel, dim3(grid_for((long)M * N)), dim3(256), 0, s, C, ldc, M, N);
}

extern "C" {

int ac_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M, int N,
            int K, const float* bias, int relu, float beta, int splitk, float drop_p, unsigned long long drop_seed,
            const unsigned long long* seed_dev, long row0, const float* a_scale, int a_rows, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || relu < 0 || relu > 3 || (a_scale && a_rows <= 0)) return AC_ERR_ARG;
  if (splitk > 1 && (bias || relu || drop_p > 0.f || (beta != 1.0f && beta != 0.0f))) return AC_ERR_ARG;
  if (splitk > 1 && beta == 0.0f) {   // the slices accumulate with atomics: start from zeros
    zero2d(C, ldc, M, N, (hipStream_t)stream);
    beta = 1.0f;
  }
  GemmP p;
  p.A = A; p.sam = sam; p.sak = sak; p.B = B; p.sbk = sbk; p.sbn = sbn; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.bias = bias; p.relu = relu; p.beta = beta;
  p.a_scale = a_scale; p.a_rows = a_rows;
  p.splitk = splitk < 1 ? 1 : splitk;
  p.drop = make_drop(drop_p, drop_seed, seed_dev);
  p.row0 = row0;
  dim3 grid((M + GT - 1) / GT, (N + GT - 1) / GT, p.splitk);
  if (grid.y > 65535) return AC_ERR_ARG;
  const bool small = p.splitk == 1 && grid.x * grid.y <= 256 && K >= 128;
  const bool kk_ok = sak == 1 && sbk == 1 && K % 4 == 0 && sam % 4 == 0 && sbn % 4 == 0 && ((uintptr_t)A & 15) == 0 &&
                     ((uintptr_t)B & 15) == 0 && (!a_scale || ((uintptr_t)a_scale & 15) == 0);
  const long nt_tiles = (long)((M + 127) / 128) * ((N + 127) / 128) * p.splitk;      // 128x128 tiles
  const long nt_tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64) * p.splitk;
  if (small && !a_scale && sak == 1 && sbk == 1 && K % 32 == 0 && sam % 4 == 0 && sbn % 4 == 0 &&
      ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0)
    hipLaunchKernelGGL(gemm_kk_kernel, dim3((M + KT - 1) / KT, (N + KT - 1) / KT), dim3(256), 0, (hipStream_t)stream, p);
  else if (kk_ok && N >= 96 && nt_tiles >= 512)
    hipLaunchKernelGGL(gemm_nt_kernel<2>, dim3((M + 127) / 128, (N + 127) / 128, p.splitk), dim3(256), 0,
                       (hipStream_t)stream, p);
  else if (kk_ok && N >= 48 && nt_tiles64 >= 128)
    hipLaunchKernelGGL(gemm_nt_kernel<1>, dim3((M + 63) / 64, (N + 63) / 64, p.splitk), dim3(256), 0,
                       (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(gemm_general_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  return ac_check_launch();
}

// Whether the split-bf16 kernel can take an operand: contiguous along k or along its row dimension, the other stride a
// multiple of 4 floats, 16-byte aligned base; k-contiguous operands need K % 4 == 0, row-contiguous ones rows % 4 == 0.
static bool gb_operand_ok(const float* base, long s_row, long s_k, int rows, int K) {
  if (((uintptr_t)base & 15) != 0) return false;
  if (s_k == 1) return s_row % 4 == 0 && K % 4 == 0;
  if (s_row == 1) return s_k % 4 == 0 && rows % 4 == 0;
  return false;
}

int ac_gemm_bf16x3(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M,
                   int N, int K, const float* bias, int relu, float beta, int splitk, float drop_p,
                   unsigned long long drop_seed, const unsigned long long* seed_dev, long row0, const float* a_scale,
                   int a_rows, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || relu < 0 || relu > 3 || (a_scale && a_rows <= 0)) return AC_ERR_ARG;
  if (splitk > 1 && (bias || relu || drop_p > 0.f || (beta != 1.0f && beta != 0.0f))) return AC_ERR_ARG;
  // small or oddly laid out products: the exact-f32 kernels (a split-bf16 tile would be mostly padding / latency)
  const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64) * (splitk < 1 ? 1 : splitk);
  if ((double)M * N * K < 3.0e7 || t64 < 200 || !gb_operand_ok(A, sam, sak, M, K) || !gb_operand_ok(B, sbn, sbk, N, K) ||
      (a_scale && (sak != 1 || ((uintptr_t)a_scale & 15) != 0)))
    return ac_gemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias, relu, beta, splitk, drop_p, drop_seed, seed_dev, row0,
                   a_scale, a_rows, stream);
  if (splitk > 1 && beta == 0.0f) {
    zero2d(C, ldc, M, N, (hipStream_t)stream);
    beta = 1.0f;
  }
  GemmP p;
  p.A = A; p.sam = sam; p.sak = sak; p.B = B; p.sbk = sbk; p.sbn = sbn; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.bias = bias; p.relu = relu; p.beta = beta;
  p.a_scale = a_scale; p.a_rows = a_rows;
  p.splitk = splitk < 1 ? 1 : splitk;
  p.drop = make_drop(drop_p, drop_seed, seed_dev);
  p.row0 = row0;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * p.splitk;
  static const int force = getenv("AC_GB_TILE") ? atoi(getenv("AC_GB_TILE")) : 0;   // development: 22 / 21 / 12 / 11
  // 64x64 tiles fill the chip at the step's row counts (85-120 TFLOP/s); 128x128 only pays from ~450 such tiles on
  const int tile = force ? force : (t128 >= 448 ? 22 : 11);
  const int bm = tile / 10 * 64, bn = tile % 10 * 64;
  dim3 grid((M + bm - 1) / bm, (N + bn - 1) / bn, p.splitk);
  if (grid.y > 65535) return AC_ERR_ARG;
  if (tile == 22) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (tile == 21) hipLaunchKernelGGL((gemm_bf16x3_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (tile == 12) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
  return ac_check_launch();
}

int ac_dropout(const float* x, float* y, long n, float p, unsigned long long seed, const unsigned long long* seed_dev,
               long idx0, void* stream) {
  if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return AC_ERR_ARG;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n,
                     make_drop(p, seed, seed_dev), idx0);
  return ac_check_launch();
}

int ac_mask_pos_scale(float* g, const float* h, long n, float scale, void* stream) {
  if (!g || !h || n <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(mask_pos_scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, h, n, scale);
  return ac_check_launch();
}

int ac_build_prefix(const long long* cap, int cap_ld, const int* seq, int seq_ld, const int* use_cap, int t,
                    int start_idx, int* word, long row0, int N, int L, void* stream) {
  if (!cap || !seq || !use_cap || !word || N <= 0 || L <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(build_prefix_kernel, dim3((N * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, cap, cap_ld, seq,
                     seq_ld, use_cap, t, start_idx, word, row0, N, L);
  return ac_check_launch();
}

int ac_embed_fwd(const float* emb, const float* pe, const int* word, const int* pos, float* x, long row0, long rows,
                 int d, float pa, unsigned long long seed_a, float pb, unsigned long long seed_b,
                 const unsigned long long* seed_dev, void* stream) {
  if (!emb || !pe || !word || !pos || !x || rows <= 0 || d <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(rows * d)), dim3(256), 0, (hipStream_t)stream, emb, pe, word, pos, x,
                     row0, rows, d, sqrtf((float)d), make_drop(pa, seed_a, seed_dev), make_drop(pb, seed_b, seed_dev));
  return ac_check_launch();
}

int ac_embed_bwd(const float* dx, const int* word, float* demb, long rows, int d, float pa, unsigned long long seed_a,
                 float pb, unsigned long long seed_b, const unsigned long long* seed_dev, void* stream) {
  if (!dx || !word || !demb || rows <= 0 || d <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(rows * d)), dim3(256), 0, (hipStream_t)stream, dx, word, demb, rows, d,
                     sqrtf((float)d), make_drop(pa, seed_a, seed_dev), make_drop(pb, seed_b, seed_dev));
  return ac_check_launch();
}

int ac_dropadd_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* pre, float* y,
                      long row0, long rows, long xmod, int d, float p, unsigned long long seed,
                      const unsigned long long* seed_dev, float eps, void* stream) {
  if (!x || !gamma || !beta || !pre || !y || rows <= 0 || d != DM || xmod < 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(dropadd_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta,
                     pre, y, row0, rows, xmod, make_drop(p, seed, seed_dev), eps);
  return ac_check_launch();
}

int ac_dropadd_ln_bwd(const float* dy, const float* pre, const float* gamma, float* dx, float* dres, int accumulate,
                      const float* relu_src, long relu_mod, float* dgamma, float* dbeta, long rows, int d, float p,
                      unsigned long long seed, const unsigned long long* seed_dev, float eps, void* stream) {
  if (!dy || !pre || !gamma || !dgamma || !dbeta || rows <= 0 || d != DM || relu_mod < 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(dropadd_ln_bwd_kernel, dim3((rows + LN_ROWS - 1) / LN_ROWS), dim3(256), 0, (hipStream_t)stream, dy, pre,
                     gamma, dx, dres, accumulate, relu_src, relu_mod, dgamma, dbeta, rows, make_drop(p, seed, seed_dev), eps);
  return ac_check_launch();
}

static int attn_args_ok(int nseq, int nhead, int head_dim, int lmax, int tkmax, int pl, int ptk) {
  return nseq > 0 && nhead > 0 && head_dim == HD && lmax > 0 && tkmax > 0 && pl >= lmax && ptk >= tkmax;
}
static size_t attn_lds_bytes(int lmax, int tkmax, bool bwd) {
  size_t f = (size_t)(lmax + 2 * tkmax) * HDP + (size_t)lmax * (tkmax + 1);
  if (bwd) f += (size_t)lmax * HDP + (size_t)lmax * (tkmax + 1);
  return f * sizeof(float);
}
static int attn_allow_lds(const void* kern, AcLdsAttr* done) { return ac_allow_lds(kern, ATT_LDS_MAX, done); }

int ac_attn_seq_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o, long ldo,
                    float* P, int pl, int ptk, const int* qrow0, const int* qlen, const int* krow0, const int* klen,
                    const int* kvalid, const int* word, int pad_idx, int causal, int seq0, int nseq, int nhead,
                    int head_dim, int lmax, int tkmax, float drop_p, unsigned long long seed,
                    const unsigned long long* seed_dev, void* stream) {
  if (!q || !k || !v || !o || !P || !qrow0 || !qlen || !krow0 || !klen ||
      !attn_args_ok(nseq, nhead, head_dim, lmax, tkmax, pl, ptk))
    return AC_ERR_ARG;
  AttnP p = {};
  p.q = q; p.ldq = ldq; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv; p.o = o; p.ldo = ldo;
  p.P = P; p.pl = pl; p.ptk = ptk; p.qrow0 = qrow0; p.qlen = qlen; p.krow0 = krow0; p.klen = klen;
  p.kvalid = kvalid; p.word = word; p.pad_idx = pad_idx; p.causal = causal; p.seq0 = seq0;
  p.scale = 1.0f / sqrtf((float)head_dim);
  p.drop = make_drop(drop_p, seed, seed_dev);
  p.lmax = lmax; p.tkmax = tkmax;
  const size_t lds = attn_lds_bytes(lmax, tkmax, false);
  if (lds > (size_t)ATT_LDS_MAX) return AC_ERR_ARG;
  static AcLdsAttr allowed;   // per device
  if (attn_allow_lds((const void*)attn_seq_fwd_kernel, &allowed) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(attn_seq_fwd_kernel, dim3(nseq, nhead), dim3(256), lds, (hipStream_t)stream, p);
  return ac_check_launch();
}

int ac_attn_seq_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const float* P, int pl,
                    int ptk, const float* dout, long lddo, float* dq, long lddq, float* dk, long lddk, float* dv,
                    long lddv, const int* qrow0, const int* qlen, const int* krow0, const int* klen, int seq0, int nseq,
                    int nhead, int head_dim, int lmax, int tkmax, float drop_p, unsigned long long seed,
                    const unsigned long long* seed_dev, void* stream) {
  if (!q || !k || !v || !P || !dout || !dq || !dk || !dv || !qrow0 || !qlen || !krow0 || !klen ||
      !attn_args_ok(nseq, nhead, head_dim, lmax, tkmax, pl, ptk))
    return AC_ERR_ARG;
  AttnP p = {};
  p.q = q; p.ldq = ldq; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv;
  p.P = const_cast<float*>(P); p.pl = pl; p.ptk = ptk; p.qrow0 = qrow0; p.qlen = qlen; p.krow0 = krow0; p.klen = klen;
  p.seq0 = seq0;
  p.scale = 1.0f / sqrtf((float)head_dim);
  p.drop = make_drop(drop_p, seed, seed_dev);
  p.dout = dout; p.lddo = lddo; p.dq = dq; p.lddq = lddq; p.dk = dk; p.lddk = lddk; p.dv = dv; p.lddv = lddv;
  p.lmax = lmax; p.tkmax = tkmax;
  const size_t lds = attn_lds_bytes(lmax, tkmax, true);
  if (lds > (size_t)ATT_LDS_MAX) return AC_ERR_ARG;
  static AcLdsAttr allowed;   // per device
  if (attn_allow_lds((const void*)attn_seq_bwd_kernel, &allowed) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(attn_seq_bwd_kernel, dim3(nseq, nhead), dim3(256), lds, (hipStream_t)stream, p);
  return ac_check_launch();
}

int ac_gather_rows(const float* src, const int* index, float* dst, long nrows, int C, void* stream) {
  if (!src || !index || !dst || nrows <= 0 || C <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(nrows * C)), dim3(256), 0, (hipStream_t)stream, src, index, dst, nrows, C);
  return ac_check_launch();
}

int ac_scatter_add_rows(const float* src, const int* index, float* dst, long nrows, int C, void* stream) {
  if (!src || !index || !dst || nrows <= 0 || C <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for(nrows * C)), dim3(256), 0, (hipStream_t)stream, src, index, dst,
                     nrows, C);
  return ac_check_launch();
}

int ac_specaug(float* x, const int* stripes, const float* fill, int B, int rows_per_clip, int T, int F, int n_time,
               int n_freq, void* stream) {
  if (!x || !stripes || B <= 0 || T <= 0 || T > rows_per_clip || F <= 0 || n_time < 0 || n_freq < 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(specaug_kernel, dim3(grid_for((long)B * T * F)), dim3(256), 0, (hipStream_t)stream, x, stripes, fill, B,
                     rows_per_clip, T, F, n_time, n_freq);
  return ac_check_launch();
}

int ac_sum_replicas(const float* x, float* out, long n, int reps, void* stream) {
  if (!x || !out || n <= 0 || reps <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(sum_replicas_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, n, reps);
  return ac_check_launch();
}

int ac_rows_mean_w(const float* x, float* out, int B, int Hp, int H, int W, int C, void* stream) {
  if (!x || !out || B <= 0 || Hp <= 0 || H <= 0 || H > Hp || W <= 0 || C <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(rows_mean_w_kernel, dim3(grid_for((long)B * H * C)), dim3(256), 0, (hipStream_t)stream, x, out, B, Hp, H,
                     W, C);
  return ac_check_launch();
}

int ac_transpose(const float* src, float* dst, int batch, int rows, int cols, void* stream) {
  if (!src || !dst || batch <= 0 || rows <= 0 || cols <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(32, 8), 0, (hipStream_t)stream,
                     src, dst, rows, cols);
  return ac_check_launch();
}

int ac_colsum(const float* x, long ld, float* out, long M, int N, void* stream) {
  if (!x || !out || M <= 0 || N <= 0) return AC_ERR_ARG;
  long gy = (M + 255) / 256;
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (int)gy), dim3(256), 0, (hipStream_t)stream, x, ld, out, M, N);
  return ac_check_launch();
}

int ac_argmax_rows(const float* logit, long ld, int rows, int V, int* out, long out_ld, void* stream) {
  if (!logit || !out || rows <= 0 || V <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logit, ld, V, out, out_ld);
  return ac_check_launch();
}

int ac_label_smoothing_loss(const float* logit, const long long* tgt, long tgt_ld, const int* tgt_len, int N, int T, int V,
                            float smoothing, float inv_count, float* row_loss, float* loss, float* dlogit, float gscale,
                            const float* gscale_dev, void* stream) {
  if (!logit || !tgt || !tgt_len || !row_loss || !loss || N <= 0 || T <= 0 || V <= 1) return AC_ERR_ARG;
  hipLaunchKernelGGL(xent_kernel, dim3(N * T), dim3(256), 0, (hipStream_t)stream, logit, tgt, tgt_ld, tgt_len, T, V,
                     smoothing, row_loss, dlogit, gscale, gscale_dev, N);
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_loss, (long)N * T, inv_count, loss,
                     tgt_len, N, T);
  return ac_check_launch();
}

int ac_gru_layer_train(const float* gx, const float* whhT, const float* bhh, const int* lens, float* out, float* save,
                       int B, int T, int hidden, void* stream) {
  if (!gx || !whhT || !bhh || !lens || !out || !save || B <= 0 || T <= 0 || hidden != H) return AC_ERR_ARG;
  const size_t lds = (size_t)GRU_KLDS * 3 * H * sizeof(float);
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)gru_train_fwd_kernel, (int)lds, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(gru_train_fwd_kernel, dim3(2 * B), dim3(768), lds, (hipStream_t)stream, gx, whhT, bhh, lens, out, save, T);
  return ac_check_launch();
}

int ac_gru_layer_bwd(const float* dout, const float* out, const float* save, const float* whh, const int* lens,
                     float* dgx, float* dgh, float* hprev, int B, int T, int hidden, void* stream) {
  if (!dout || !out || !save || !whh || !lens || !dgx || !dgh || !hprev || B <= 0 || T <= 0 || hidden != H)
    return AC_ERR_ARG;
  const size_t lds = (size_t)GRU_KLDS * 3 * H * sizeof(float);
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)gru_bwd_kernel, (int)lds, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(2 * B), dim3(768), lds, (hipStream_t)stream, dout, out, save, whh, lens, dgx, dgh,
                     hprev, T);
  return ac_check_launch();
}

int ac_grad_sumsq(const float* g, long n, float* norm_state, void* stream) {
  if (!g || !norm_state || n <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, g, n, norm_state);
  return ac_check_launch();
}

int ac_clip_coef(float* norm_state, float max_norm, float grad_div, void* stream) {
  if (!norm_state || grad_div <= 0.f) return AC_ERR_ARG;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, norm_state, max_norm, grad_div);
  return ac_check_launch();
}

int ac_scale_by_coef(float* x, long n, const float* norm_state, void* stream) {
  if (!x || !norm_state || n <= 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, x, n, norm_state);
  return ac_check_launch();
}

int ac_swa_update(float* avg, const float* p, long n, int n_averaged, void* stream) {
  if (!avg || !p || n <= 0 || n_averaged < 0) return AC_ERR_ARG;
  hipLaunchKernelGGL(swa_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, avg, p, n,
                     1.0f / (float)(n_averaged + 1));
  return ac_check_launch();
}

int ac_adam_step(float* p, const float* g, float* m, float* v, long n, const float* norm_state, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, const int* step_dev, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || (step < 1 && !step_dev)) return AC_ERR_ARG;
  const float bc1 = 1.0f - powf(beta1, (float)(step < 1 ? 1 : step));
  const float bc2 = 1.0f - powf(beta2, (float)(step < 1 ? 1 : step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, norm_state,
                     lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), step_dev);
  return ac_check_launch();
}

int ac_adam_commit(int* step_dev, const float* norm_state, void* stream) {
  if (!step_dev) return AC_ERR_ARG;
  hipLaunchKernelGGL(adam_commit_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, norm_state);
  return ac_check_launch();
}

}  // extern "C"
