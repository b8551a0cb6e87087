// Cnn14 conv stack on gfx950: 3x3 / stride 1 / pad 1 convolution + eval-mode BatchNorm + ReLU
// (+ 2x2 average pooling, or the final mean over the 2 mel columns) as ONE implicit-GEMM kernel
// on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces reference ConvBlock.forward (cnn_encoder.py:59-75) and the pooling / mean / transpose
// glue of Cnn14Encoder.forward (cnn_encoder.py:431-444).
//
// Data layout (chosen for the GEMM, not inherited from NCHW):
//   activations  [B * Hp][W][C]  fp32, channels-last.  H = time, W = mel.  Every clip owns Hp >= H+1
//                physical rows; rows h >= H are ZERO.  They are the conv's vertical zero padding
//                (one shared pad row separates consecutive clips) and make Hp_k = 2 * Hp_{k+1}, so a
//                2x2 pooling window never straddles two clips and M-tiles may span clips freely.
//   weights      [Cin/32][9 taps][Cout][32]  (packed once on the host side from OIHW)
//   BN           folded to per-channel scale/shift applied to the f32 accumulator in the epilogue.
//
// GEMM view: M = B*Hp*W output pixels, N = Cout, K = 9*Cin.  A 256-thread block owns a 128-pixel
// (TR rows x TC cols, TC = min(W,16)) by BN-channel tile; per 32-channel K-chunk it stages the
// (TR+2)x(TC+2) input halo patch in LDS once and re-uses it for all 9 taps, streaming the 9 weight
// slabs through a double-buffered LDS ring (global -> VGPR issued before the MFMAs of the current
// tap, VGPR -> LDS after them).  The row index i of every 32x32 MFMA tile is laid out as
// i = 4*window + 2*dy + dx, so the four pixels of a pooling window sit in four consecutive
// accumulator registers of one lane and pooling costs three adds in the epilogue.
#include "ac_common.h"

namespace {

constexpr int LDS_STRIDE = 36;       // 32 channels + 4 pad floats: 144-byte rows keep b128 reads aligned
constexpr int MAX_NPIX = 66 * 4;     // largest halo patch: W = 2 -> (64+2) x (2+2)

struct ConvParams {
  const float* in;
  const float* wpk;
  const float* scale;
  const float* shift;
  float* out;
  int rows_total, Hp, H, W, Cin, Cout;
  int tc_log2, mt_cols, MT, NT;
  int Hp_out, H_out, W_out;
  int map_mode;
  // fused first layer (conv3x3_gw_kernel<..., FUSE1>): the 1-channel input and conv1's weights / folded BN
  int m_valid;   // one-tap GEMM use: number of valid "pixels" (rows of the GEMM), 0 = all
  const float* in1;
  const float* w1;
  const float* sc1;
  const float* sh1;
  // fp16 tier: *ovf is OR-ed with 1 when a value stored as fp16 exceeded the fp16 range (65504); may be null.
  // out32: the POOL epilogue of the fp16 tier writes f32 (the block feeding a split-bf16 block of the mixed tier).
  unsigned* ovf = nullptr;
  int out32 = 0;
};

// fp16 range guard of the "f16x2" tier: every value about to be stored as fp16 goes through track(); one atomic per
// wave at the end, and only when something overflowed (never on healthy activations).
struct F16Guard {
  float mx = 0.f;
  // post-ReLU values: y >= 0.  Written so that a NaN sticks (fmaxf would drop it) and trips the flag in commit().
  __device__ __forceinline__ void track(float y) { mx = (y <= mx) ? mx : y; }
  __device__ __forceinline__ void commit(unsigned* flag) const {
    if (flag != nullptr && !(mx <= 65504.f)) atomicOr(flag, 1u);
  }
};

enum { MODE_FULL = 0, MODE_POOL = 1, MODE_MEANW = 2, MODE_LINEAR = 3 };  // LINEAR: FULL without the ReLU (1-tap GEMM use)

// ---- block -> (m_tile, n_tile); block b runs on XCD b % 8 (speed only, never correctness) ----
__device__ __forceinline__ bool conv_block_map(const ConvParams& p, int& m_tile, int& n_tile) {
  const int bid = blockIdx.x;
  if (p.map_mode == 1) {         // weight-heavy layers: one XCD streams one weight column slab
    const int xcd = bid & 7, seq = bid >> 3;
    n_tile = xcd + 8 * (seq / p.MT);
    m_tile = seq % p.MT;
  } else if (p.map_mode == 2) {  // activation-heavy layers: blocks sharing a halo patch share an L2
    const int xcd = bid & 7, seq = bid >> 3;
    n_tile = seq % p.NT;
    m_tile = (seq / p.NT) * 8 + xcd;
    if (m_tile >= p.MT) return false;
  } else if (p.map_mode == 3) {  // NT in {1,2,4,8}: every XCD owns ONE weight column slab (L2-resident),
    const int xcd = bid & 7, seq = bid >> 3;  //          the 8/NT XCDs of a slab interleave the pixel tiles
    const int per = 8 / p.NT;
    n_tile = xcd % p.NT;
    m_tile = seq * per + xcd / p.NT;
    if (m_tile >= p.MT) return false;
  } else {
    n_tile = bid % p.NT;
    m_tile = bid / p.NT;
  }
  return true;
}

// Window order inside a 32-pixel MFMA tile.  A ds_read_b128 is serviced in 16-lane groups
// {0-3,12-15,20-27} / {4-11,16-19,28-31} (per 32-lane half), i.e. windows q in {0,3,5,6} / {1,2,4,7}.
// Mapping those to window POSITIONS 0-3 / 4-7 makes each hardware group read one compact half of the tile
// (2x8, 4x4 or 8x2 pixels), which a suitable patch row pitch then spreads over all 16 bank slots.
__device__ __forceinline__ int window_pos(int q) { return (0x73261540 >> (4 * q)) & 7; }  // [0,4,5,1,6,2,3,7]

// x mod d (and x / d) for 0 <= x < 2^23 with a reciprocal computed once per wave: the epilogues need the row inside the
// clip of 16-64 output rows per wave, and an integer division by the run-time Hp costs ~25 instructions each.
struct FastDiv {
  int d;
  float inv;
  __device__ __forceinline__ explicit FastDiv(int d_) : d(d_), inv(1.0f / (float)d_) {}
  __device__ __forceinline__ int div(int x, int& rem) const {
    int q = (int)((float)x * inv);
    int r = x - q * d;
    if (r < 0) { r += d; --q; }
    if (r >= d) { r -= d; ++q; }
    rem = r;
    return q;
  }
  __device__ __forceinline__ int mod(int x) const { int r; div(x, r); return r; }
};

// ---- epilogue: BN scale/shift, ReLU, pooling, zero rows; lanes 0..31 store 32 consecutive channels ----
// acc[m][n] = 32x32 tile (m-th pixel tile, n-th channel tile of this wave), MFMA row i = 4*window + 2*dy + dx
// A wave owns MW pixel tiles (wm * MW + m of the block's four) and NTW channel tiles (wn * NTW + n).
// OUT16: the FULL / POOL outputs are stored as fp16 (the activations of the "f16x2" tier live in HBM as fp16: the
// consumer would round them to fp16 anyway, so the numbers are the same and the traffic is half).
template <int BN, int MODE, int MW = 2, int NTW = BN / 64, bool OUT16 = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, const f32x16 (&acc)[MW][NTW], int n_tile, int row0,
                                              int col0, int wm, int wn, int lane) {
  const int half = lane >> 5;
  const int TC = 1 << p.tc_log2;
  const int QR2 = 32 >> p.tc_log2;
  const FastDiv by_hp(p.Hp), by_hp_out(MODE == MODE_POOL ? p.Hp_out : 1);
  F16Guard guard;
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int ch = n_tile * BN + (wn * NTW + n) * 32 + (lane & 31);
    const float sc = p.scale[ch], sh = p.shift[ch];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        // window q = 2*rq + half of this MFMA tile; registers 4*rq + (2*dy + dx)
        const int q = window_pos(2 * rq + half);
        const int qc = q & ((TC >> 1) - 1);
        const int qr = q >> (p.tc_log2 - 1);
        const int wy = row0 + (MW * wm + m) * QR2 + 2 * qr;  // physical row of the window's dy = 0
        const int wx = col0 + 2 * qc;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = fmaf(acc[m][n][4 * rq + e], sc, sh);
          if (MODE != MODE_LINEAR) y[e] = fmaxf(y[e], 0.f);
        }
        if (MODE == MODE_FULL || MODE == MODE_LINEAR) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int gr = wy + (e >> 1), gc = wx + (e & 1);
            if (gr < p.rows_total && (p.m_valid == 0 || gr * p.W + gc < p.m_valid)) {
              const bool valid = by_hp.mod(gr) < p.H;
              const size_t o = ((size_t)gr * p.W + gc) * p.Cout + ch;
              if (OUT16) {
                guard.track(valid ? y[e] : 0.f);   // padding rows are stored as zeros: they cannot overflow
                ((_Float16*)p.out)[o] = (_Float16)(valid ? y[e] : 0.f);
              } else p.out[o] = valid ? y[e] : 0.f;
            }
          }
        } else if (MODE == MODE_POOL) {
          const int orow = wy >> 1, ocol = wx >> 1;
          if (wy < p.rows_total) {
            const bool valid = by_hp_out.mod(orow) < p.H_out;
            const float o = 0.25f * ((y[0] + y[1]) + (y[2] + y[3]));
            const size_t oi = ((size_t)orow * p.W_out + ocol) * p.Cout + ch;
            if (OUT16 && !p.out32) {
              guard.track(valid ? o : 0.f);
              ((_Float16*)p.out)[oi] = (_Float16)(valid ? o : 0.f);
            } else p.out[oi] = valid ? o : 0.f;
          }
        } else {  // MODE_MEANW: W == 2, mean over the two mel columns, dense (B, H, Cout) output
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            const int gr = wy + dy;
            if (gr < p.rows_total) {
              int h;
              const int b = by_hp.div(gr, h);
              if (h < p.H) p.out[((size_t)b * p.H + h) * p.Cout + ch] = 0.5f * (y[2 * dy] + y[2 * dy + 1]);
            }
          }
        }
      }
    }
  }
  if (OUT16) guard.commit(p.ovf);
}

template <int BN, int MODE>
__global__ __launch_bounds__(256, 1) void conv3x3_mfma_kernel(ConvParams p) {
  constexpr int NTW = BN / 64;  // 32-wide MFMA column tiles per wave
  __shared__ __attribute__((aligned(16))) float sA[MAX_NPIX * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float sB[2][BN * LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5;

  int m_tile, n_tile;
  if (!conv_block_map(p, m_tile, n_tile)) return;
  const int TC = 1 << p.tc_log2;
  const int TR = 128 >> p.tc_log2;
  const int PW = TC + 2, PH = TR + 2;
  const int NPIX = PW * PH;
  const int row0 = (m_tile / p.mt_cols) * TR;
  const int col0 = (m_tile % p.mt_cols) * TC;

  // ---- lane geometry: pixel of MFMA row i = 4*q + 2*dy + dx ----
  const int QR2 = 32 >> p.tc_log2;  // rows covered by one 32-pixel MFMA tile
  int ty0[2], tx0;                  // top-left pixel (inside the tile) of this lane's A row, dy = dx = 0
  int pbase[2];
  {
    const int i = lane & 31;
    const int q = window_pos(i >> 2), dy = (i >> 1) & 1, dx = i & 1;
    const int qc = q & ((TC >> 1) - 1);
    const int qr = q >> (p.tc_log2 - 1);
    tx0 = 2 * qc;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      ty0[m] = (2 * wm + m) * QR2 + 2 * qr;
      pbase[m] = ((ty0[m] + dy) * PW + tx0 + dx) * LDS_STRIDE + half * 4;
    }
  }
  int nbase[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) nbase[n] = ((wn * NTW + n) * 32 + (lane & 31)) * LDS_STRIDE + half * 4;

  f32x16 acc[2][NTW];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // tiles lying entirely in one clip's zero rows only have to write zeros
  const int rc0 = row0 % p.Hp;
  const bool all_pad = (rc0 >= p.H && rc0 + TR <= p.Hp) || row0 >= p.rows_total;

  const int nchunk = p.Cin >> 5;
  constexpr int BLD = BN * 8 / 256;  // float4 weight loads per thread per tap
  if (!all_pad) {
    // weight slab of flattened step it = chunk*9 + tap: [BN][32] floats, contiguous
    const float* wsrc = p.wpk + (size_t)n_tile * BN * 32 + (size_t)tid * 4;
    const size_t slab_stride = (size_t)p.Cout * 32;
    const int total = nchunk * 9;
    auto stage_patch = [&](int c) {
      for (int idx = tid; idx < NPIX * 8; idx += 256) {
        const int pix = idx >> 3, c4 = idx & 7;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int gr = row0 - 1 + pr, gc = col0 - 1 + pc;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr >= 0 && gr < p.rows_total && gc >= 0 && gc < p.W)
          v = *(const float4*)(p.in + ((size_t)gr * p.W + gc) * p.Cin + c * 32 + c4 * 4);
        *(float4*)(sA + pix * LDS_STRIDE + c4 * 4) = v;
      }
    };
    stage_patch(0);
#pragma unroll
    for (int u = 0; u < BLD; ++u) {
      const int idx = tid + u * 256;
      *(float4*)(&sB[0][(idx >> 3) * LDS_STRIDE + (idx & 7) * 4]) = *(const float4*)(wsrc + (size_t)u * 1024);
    }
    __syncthreads();
    int tap = 0, c = 0;
#pragma unroll 1
    for (int it = 0; it < total; ++it) {
      // global -> VGPR for the next slab is issued before this slab's MFMAs and lands in LDS after them
      const int nxt = it + 1 < total ? it + 1 : it;
      f32x4 breg[BLD];
#pragma unroll
      for (int u = 0; u < BLD; ++u) breg[u] = *(const f32x4*)(wsrc + (size_t)nxt * slab_stride + (size_t)u * 1024);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMA block (hipcc sinks them otherwise)
      const int ky = tap / 3, kx = tap - 3 * ky;
      const float* a_tap = sA + (ky * PW + kx) * LDS_STRIDE;
      const float* b_cur = sB[it & 1];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 a[2], b[NTW];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = *(const f32x4*)(a_tap + pbase[m] + g * 8);
#pragma unroll
        for (int n = 0; n < NTW; ++n) b[n] = *(const f32x4*)(b_cur + nbase[n] + g * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[m][n] = mfma32(a[m][s], b[n][s], acc[m][n]);
      }
      float* b_nxt = sB[(it + 1) & 1];
#pragma unroll
      for (int u = 0; u < BLD; ++u) {
        const int idx = tid + u * 256;
        *(f32x4*)(b_nxt + (idx >> 3) * LDS_STRIDE + (idx & 7) * 4) = breg[u];
      }
      __syncthreads();
      if (++tap == 9) {
        tap = 0;
        if (++c < nchunk) {
          stage_patch(c);
          __syncthreads();
        }
      }
    }
  }

  conv_epilogue<BN, MODE>(p, acc, n_tile, row0, col0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant of the same implicit GEMM for the 1e-3-logit precision tier:
// every f32 operand is split into two bf16 numbers, x = hi + lo (hi = RNE(x), lo = RNE(x - hi), 16
// significant bits together), and the product is accumulated in f32 as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16.  Three bf16 MFMAs replace eight f32 MFMAs (K = 16): 5.3x the matrix rate of
// the exact-f32 path at ~2^-16 relative operand error.  Activations stay f32 in HBM: the halo patch is
// split once when it is staged (and then re-used by 9 taps x all channel tiles), weights are split
// offline.  LDS holds a hi and a lo plane of bf16 rows: 32 channels = 64 bytes + 16 pad (80-byte rows keep
// the 16-byte fragment reads aligned and spread over the banks).
// ------------------------------------------------------------------------------------------------
// pad (in 16-byte slots) that brings a patch row of (TC + 2) pixels x 5 slots to a pitch of 8 mod 16 (TC >= 8),
// 4 mod 8 (TC = 4) or 2 mod 4 (TC = 2)
// Column tiles (COLT kernels, W = 2 or 4): 32 consecutive rows of ONE column form an MFMA tile, so a ds_read_b128 walks the
// patch at a stride of one patch row; an odd number of 16-byte slots per row spreads 16 lanes over the 16 bank slots.
__host__ __device__ __forceinline__ int colt_row_pad_slots(int TC) { return (((TC + 2) * 5) & 1) ? 0 : 1; }

__host__ __device__ __forceinline__ int patch_row_pad_slots(int TC) {
  const int base = (TC + 2) * 5;
  return TC >= 8 ? ((8 - base) & 15) : (TC == 4 ? ((4 - base) & 7) : ((2 - base) & 3));
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int BROW = 40;  // bf16 elements per LDS row (32 + 8 pad)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// x (4 floats) -> packed hi (2 dwords) and lo (2 dwords) bf16 quadruples
__device__ __forceinline__ void split_bf16x4(const f32x4 x, u32x2& hi, u32x2& lo) {
  hi.x = cvt_pk_bf16(x[0], x[1]);
  hi.y = cvt_pk_bf16(x[2], x[3]);
  const float h0 = __builtin_bit_cast(float, hi.x << 16), h1 = __builtin_bit_cast(float, hi.x & 0xffff0000u);
  const float h2 = __builtin_bit_cast(float, hi.y << 16), h3 = __builtin_bit_cast(float, hi.y & 0xffff0000u);
  lo.x = cvt_pk_bf16(x[0] - h0, x[1] - h1);
  lo.y = cvt_pk_bf16(x[2] - h2, x[3] - h3);
}

template <int BN, int MODE>
__global__ __launch_bounds__(256, 1) void conv3x3_bf16x3_kernel(ConvParams p) {
  constexpr int NTW = BN / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5;

  int m_tile, n_tile;
  if (!conv_block_map(p, m_tile, n_tile)) return;
  const int TC = 1 << p.tc_log2;
  const int TR = 128 >> p.tc_log2;
  const int PW = TC + 2, PH = TR + 2;
  const int NPIX = PW * PH;
  const int row0 = (m_tile / p.mt_cols) * TR;
  const int col0 = (m_tile % p.mt_cols) * TC;

  // Patch rows are PW pixels of 80 bytes (5 bank slots of 16 bytes) plus a pad chosen so that the row pitch
  // is = 8 (TC >= 8), 4 (TC = 4) or 2 (TC = 2) slots mod 16: together with window_pos() the 16 lanes of
  // every ds_read_b128 group then hit 16 different slots (conflict-free A fragments).
  const int PITCH = PW * BROW + patch_row_pad_slots(TC) * 8;  // bf16 elements per patch row
  const int PLANE = PH * PITCH;
  // LDS carve: patch hi | patch lo | weights [2 buffers][hi, lo][BN rows]
  __bf16* sAh = (__bf16*)dsm_raw;
  __bf16* sAl = sAh + PLANE;
  __bf16* sBq = sAl + PLANE;
  constexpr int BPLANE = BN * BROW;  // elements of one weight plane

  const int QR2 = 32 >> p.tc_log2;
  int pbase[2];
  {
    const int i = lane & 31;
    const int q = window_pos(i >> 2), dy = (i >> 1) & 1, dx = i & 1;
    const int qc = q & ((TC >> 1) - 1);
    const int qr = q >> (p.tc_log2 - 1);
#pragma unroll
    for (int m = 0; m < 2; ++m)
      pbase[m] = ((2 * wm + m) * QR2 + 2 * qr + dy) * PITCH + (2 * qc + dx) * BROW + half * 8;
  }
  int nbase[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) nbase[n] = ((wn * NTW + n) * 32 + (lane & 31)) * BROW + half * 8;

  f32x16 acc[2][NTW];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int rc0 = row0 % p.Hp;
  const bool all_pad = (rc0 >= p.H && rc0 + TR <= p.Hp) || row0 >= p.rows_total;
  const int nchunk = p.Cin >> 5;
  constexpr int BLD = 2 * BN * 4 / 256;  // 16-byte weight loads per thread per tap (hi + lo planes)
  if (!all_pad) {
    // packed weights: [Cin/32][9][2 planes][Cout][32] bf16; one (chunk, tap, plane) slab = Cout x 64 bytes
    const unsigned char* wsrc = (const unsigned char*)p.wpk + (size_t)n_tile * BN * 64;
    const size_t plane_bytes = (size_t)p.Cout * 64;
    const int total = nchunk * 9;
    auto w_load = [&](int it, u32x4 (&reg)[BLD]) {
#pragma unroll
      for (int u = 0; u < BLD; ++u) {
        const int idx = tid + u * 256;               // 16-byte unit: plane = idx / (BN*4), row = (idx / 4) % BN
        const int plane = idx / (BN * 4), rem = idx - plane * (BN * 4);
        reg[u] = *(const u32x4*)(wsrc + ((size_t)it * 2 + plane) * plane_bytes + (size_t)rem * 16);
      }
    };
    auto w_store = [&](int buf, const u32x4 (&reg)[BLD]) {
#pragma unroll
      for (int u = 0; u < BLD; ++u) {
        const int idx = tid + u * 256;
        const int plane = idx / (BN * 4), rem = idx - plane * (BN * 4);
        *(u32x4*)(sBq + (buf * 2 + plane) * BPLANE + (rem >> 2) * BROW + (rem & 3) * 8) = reg[u];
      }
    };
    auto stage_patch = [&](int c) {
      for (int idx = tid; idx < NPIX * 8; idx += 256) {
        const int pix = idx >> 3, c4 = idx & 7;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int gr = row0 - 1 + pr, gc = col0 - 1 + pc;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr >= 0 && gr < p.rows_total && gc >= 0 && gc < p.W)
          v = *(const f32x4*)(p.in + ((size_t)gr * p.W + gc) * p.Cin + c * 32 + c4 * 4);
        u32x2 hi, lo;
        split_bf16x4(v, hi, lo);
        *(u32x2*)(sAh + pr * PITCH + pc * BROW + c4 * 4) = hi;
        *(u32x2*)(sAl + pr * PITCH + pc * BROW + c4 * 4) = lo;
      }
    };
    u32x4 breg[BLD];
    stage_patch(0);
    w_load(0, breg);
    w_store(0, breg);
    __syncthreads();
#ifdef AC_ABL_NOFRAG
    bf16x8 ah[2], al[2], wh[NTW], wl[NTW];
#endif
    int tap = 0, c = 0;
#pragma unroll 1
    for (int it = 0; it < total; ++it) {
#ifndef AC_ABL_NORING
      w_load(it + 1 < total ? it + 1 : it, breg);
#endif
      __builtin_amdgcn_sched_barrier(0);
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int aoff = ky * PITCH + kx * BROW;
      const __bf16* bh = sBq + ((it & 1) * 2 + 0) * BPLANE;
      const __bf16* bl = sBq + ((it & 1) * 2 + 1) * BPLANE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // two K = 16 steps per 32-channel chunk
#ifndef AC_ABL_NOFRAG
        bf16x8 ah[2], al[2], wh[NTW], wl[NTW];
#endif
#ifdef AC_ABL_NOFRAG  // development build: fragments read once (tap 0 addresses), isolates the MFMA issue rate
        const int aoff_ = 0;
        if (it == 0)
#else
        const int aoff_ = aoff;
#endif
        {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            ah[m] = *(const bf16x8*)(sAh + aoff_ + pbase[m] + ks * 16);
            al[m] = *(const bf16x8*)(sAl + aoff_ + pbase[m] + ks * 16);
          }
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            wh[n] = *(const bf16x8*)(bh + nbase[n] + ks * 16);
            wl[n] = *(const bf16x8*)(bl + nbase[n] + ks * 16);
          }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], wh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wh[n], acc[m][n], 0, 0, 0);
          }
      }
#ifndef AC_ABL_NORING
      w_store((it + 1) & 1, breg);
      __syncthreads();
#endif
      if (++tap == 9) {
        tap = 0;
        if (++c < nchunk) {
          stage_patch(c);
          __syncthreads();
        }
      }
    }
  }
  conv_epilogue<BN, MODE>(p, acc, n_tile, row0, col0, wm, wn, lane);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  const f16x2v h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, h);
}

// ---- epilogue of the column-tile kernels: m-tile mt = (row group mt / TC, column mt % TC) holds 32 consecutive rows
// of one column; lane l owns channel l % 32 and rows 8 (r / 4) + 4 (l / 32) + r % 4 of the tile (MFMA output layout),
// so vertically adjacent pixels are adjacent registers and horizontally adjacent ones are the same register of the
// neighbouring m-tile: 2x2 pooling and the mean over the two mel columns stay in registers.  fp16 outputs (f32 for
// MEANW), WM = 1 (a wave owns all four m-tiles of the block).
template <int BN, int MODE, int TC, int MW, bool OUT32 = false>
__device__ __forceinline__ void conv_epilogue_cols(const ConvParams& p, const f32x16 (&acc)[MW][1], int n_tile, int row0,
                                                   int col0, int wn, int lane) {
  const int half = lane >> 5;
  const int ch = n_tile * BN + wn * 32 + (lane & 31);
  const float sc = p.scale[ch], sh = p.shift[ch];
  _Float16* out16 = (_Float16*)p.out;
  const FastDiv by_hp(p.Hp), by_hp_out(MODE == MODE_POOL ? p.Hp_out : 1);
  F16Guard guard;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 8 * (r >> 2) + 4 * half + (r & 3);
    float y[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) y[m] = fmaxf(fmaf(acc[m][0][r], sc, sh), 0.f);
    if (MODE == MODE_FULL && OUT32) {   // split-bf16 tier: f32 activations, 32 lanes store 128 contiguous bytes
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const int gr = row0 + (m / TC) * 32 + row, gc = col0 + m % TC;
        if (gr < p.rows_total) p.out[((size_t)gr * p.W + gc) * p.Cout + ch] = by_hp.mod(gr) < p.H ? y[m] : 0.f;
      }
    } else if (MODE == MODE_FULL) {
      // Lane pairs (channels 2j, 2j+1) trade values so that every lane stores ONE 4-byte word: the even lane both
      // channels of row r, the odd lane both channels of row r + 1 (r even: the rows are adjacent registers) - half the
      // store instructions of one 2-byte store per value.
      if (r & 1) continue;
      const bool odd = lane & 1;
      const int myrow = row + (odd ? 1 : 0);
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const float a = y[m], b = fmaxf(fmaf(acc[m][0][r + 1], sc, sh), 0.f);
        guard.track(a);
        guard.track(b);
        const float pa = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));
        const float pb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, b), 0xB1, 0xF, 0xF, true));
        const int gr = row0 + (m / TC) * 32 + myrow, gc = col0 + m % TC;
        if (gr < p.rows_total) {
          const bool valid = by_hp.mod(gr) < p.H;
          const unsigned word = valid ? cvt_pk_f16(odd ? pb : a, odd ? b : pa) : 0u;
          *(unsigned*)(out16 + ((size_t)gr * p.W + gc) * p.Cout + (ch & ~1)) = word;
        }
      }
    } else if (MODE == MODE_MEANW) {   // TC == 2: m = 2 rg + column
#pragma unroll
      for (int rg = 0; rg < MW / 2; ++rg) {
        const int gr = row0 + rg * 32 + row;
        if (gr < p.rows_total) {
          int h;
          const int b = by_hp.div(gr, h);
          if (h < p.H) p.out[((size_t)b * p.H + h) * p.Cout + ch] = 0.5f * (y[2 * rg] + y[2 * rg + 1]);
        }
      }
    }
  }
  if (MODE == MODE_POOL) {   // m = TC rg + column; windows = rows (2j, 2j+1) x columns (2 oc, 2 oc + 1)
#pragma unroll
    for (int rg = 0; rg < MW / TC; ++rg) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int row = 8 * (r >> 2) + 4 * half + (r & 3);
        const int gr = row0 + rg * 32 + row;
        if (gr >= p.rows_total) continue;
        const int orow = gr >> 1;
        const bool valid = by_hp_out.mod(orow) < p.H_out;
#pragma unroll
        for (int oc = 0; oc < TC / 2; ++oc) {
          float o = 0.f;
#pragma unroll
          for (int dm = 0; dm < 2; ++dm)
#pragma unroll
            for (int dr = 0; dr < 2; ++dr) o += fmaxf(fmaf(acc[rg * TC + 2 * oc + dm][0][r + dr], sc, sh), 0.f);
          const size_t oi = ((size_t)orow * p.W_out + (col0 >> 1) + oc) * p.Cout + ch;
          o = valid ? 0.25f * o : 0.f;
          if (OUT32 || p.out32) {
            p.out[oi] = o;
          } else {
            guard.track(o);
            out16[oi] = (_Float16)o;
          }
        }
      }
    }
  }
  if (MODE != MODE_MEANW && !OUT32) guard.commit(p.ovf);
}

// ------------------------------------------------------------------------------------------------
// bf16x3, weights straight from L2 ("gw"): the B fragments never touch LDS.  Weights are pre-packed in MFMA
// fragment order, [Cin/32][9 taps][2 k-steps][Cout/32][hi, lo][64 lanes][8 bf16], so a wave's fragment is
// one contiguous 1 KiB read; the fragments of tap t+1 are requested into VGPRs while tap t runs on the
// matrix cores.  LDS then holds only the split halo patch: no weight ring, no per-tap barrier (2 barriers
// per 32-channel chunk = per 216 MFMAs of a wave), half the LDS fragment reads.
//
// PREC = 0: split-bf16, three products per f32 product (x_lo w_hi + x_hi w_lo + x_hi w_hi), 2^-16 operand error.
// PREC = 1 ("f16x2"): activations rounded ONCE to fp16 (RNE, 2^-12 relative), weights split into fp16 hi + lo
//   (2^-22; the host scales every output channel by a power of two so that the lo parts are normal numbers and
//   folds the inverse into the BN scale), two products per f32 product on v_mfma_f32_32x32x16_f16, f32
//   accumulation.  One LDS plane instead of two.  More accurate than the TF32 convolutions (2^-11 on BOTH operands)
//   the reference itself runs with on its GPUs (torch.backends.cudnn.allow_tf32 defaults to True).
//
// Instances in use (the other template parameters are explained at the top of the kernel body):
//   <128|64, MODE, 0, 2, 128, false, 9, 0>   split-bf16 tier, 2x2 waves                    (ac_conv3x3_bn_relu_bf16x3_gw)
//   <128,    MODE, 1, 1, 256, false, 9, 8|4> fp16 tier, column tiles, 256-pixel blocks: W >= 8 / W = 4 (blocks 2-5)
//   <128,    MODE, 1, 1, 128, false, 9, 2>   fp16 tier, column tiles: W = 2 (block 6)       (ac_conv3x3_bn_relu_f16x2_gw)
//   <128,    MODE, 1, 1, 128, false, 9, 0>   fp16 tier, 2x2-window tiles, 1x4 waves (AC_GW_COLT8=0 / AC_GW_NO_COLT)
//   <64,     POOL, 1, 2, 256, true,  9, 0>   fp16 tier, block 1 with conv1 fused            (ac_conv3x3_block1_f16x2)
//   <64,     MODE, 1, 2, 256|128, false, 9, 0> fp16 tier, Cout = 64 without the fusion
//   <64,     FULL|LINEAR, 0, 2, 128, false, 1, 0> one-tap GEMM: large linear layers          (ac_linear_bf16x3)
// ------------------------------------------------------------------------------------------------
template <int BN, int MODE, int PREC, int WM, int BM, bool FUSE1, int TAPS, int COLT>
__global__ __launch_bounds__(256, BM == 128 ? 3 : 2) void conv3x3_gw_kernel(ConvParams p) {
  // COLT = 2 or 4 (the layers with W = 2 / 4 mel columns, PREC 1, WM 1): COLUMN tiles - an MFMA tile is 32 consecutive
  // rows of one column instead of 2x2 windows over all columns.  With so few columns a third (W = 2) or a sixth (W = 4)
  // of the window tiles' products multiply the zero padding left and right of the image; a column tile knows which
  // taps fall outside (kx = 0 for the first column, kx = 2 for the last) and skips their MFMAs and fragment reads.
  // TAPS = 9: the 3x3 convolution.  TAPS = 1 (PREC 0 only): a plain GEMM - rows are the "pixels" of a 2-wide image, no
  // halo - through the same staging / fragment / epilogue machinery: the big f32 linear layers of the path
  // (ac_linear_bf16x3).  With one tap a chunk is only 12-24 MFMAs per wave, so the next chunk's rows are requested
  // into registers before the MFMAs of the current one (the 9-tap kernels hide that latency behind 216 MFMAs).
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  // FUSE1 (block 1 of the f16x2 tier: Cin = Cout = 64, W = 64, 256-pixel blocks): the input of this convolution is
  // itself conv1 + BN + ReLU of the 1-channel log-mel.  Instead of reading it from HBM (0.5 GB written by a separate
  // kernel, 0.7 GB read back with the halo) the workgroup computes its 18x18x64 patch from a 20x20 patch of the
  // log-mel: 72 FMAs with wave-uniform (scalar-register) weights per staged fp16 item on the vector ALUs, under the other workgroups' MFMAs.
  // wave grid WM (pixel tiles) x WN (channel tiles); WM = 1 makes every wave walk all 128 pixels of the block for
  // 32 channels: half the weight-fragment bytes per MFMA (the L1/L2 stream that limits the two-product tier)
  // BM = pixels per block (128, or 256 for the wide early layers: MW doubles, the halo overhead shrinks)
  constexpr int WN = 4 / WM, MW = (BM / 32) / WM, NTW = (BN / 32) / WN;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;

  int m_tile, n_tile;
  if (!conv_block_map(p, m_tile, n_tile)) return;
  const int TC = 1 << p.tc_log2;
  const int TR = BM >> p.tc_log2;
  const int PW = TC + 2 * HALO, PH = TR + 2 * HALO;
  const int NPIX = PW * PH;
  const int row0 = (m_tile / p.mt_cols) * TR;
  const int col0 = (m_tile % p.mt_cols) * TC;
  const bool at_left = col0 == 0, at_right = col0 + TC == p.W;   // column tiles: which edge columns touch the padding
  const int PITCH = PW * BROW + (COLT ? colt_row_pad_slots(TC) : patch_row_pad_slots(TC)) * 8;
  const int PLANE = PH * PITCH;
  __bf16* sAh = (__bf16*)dsm_raw;
  __bf16* sAl = sAh + PLANE;   // PREC 0: the lo plane

  const int QR2 = 32 >> p.tc_log2;
  int pbase[MW];
  {
    const int i = lane & 31;
    const int q = window_pos(i >> 2), dy = (i >> 1) & 1, dx = i & 1;
    const int qc = q & ((TC >> 1) - 1);
    const int qr = q >> (p.tc_log2 - 1);
#pragma unroll
    for (int m = 0; m < MW; ++m)
      pbase[m] = COLT ? ((m / (COLT ? COLT : 1)) * 32 + i) * PITCH + (m % (COLT ? COLT : 1)) * BROW + half * 8
                      : ((MW * wm + m) * QR2 + 2 * qr + dy) * PITCH + (2 * qc + dx) * BROW + half * 8;
  }

  f32x16 acc[MW][NTW];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int rc0 = row0 % p.Hp;
  const bool all_pad = (rc0 >= p.H && rc0 + TR <= p.Hp) || row0 >= p.rows_total;
  const int nchunk = p.Cin >> 5;
  if (!all_pad) {
    // fragment (it = chunk*9 + tap, ks, n-tile nt32, plane) = 1 KiB at ((it*2 + ks)*NT32 + nt32)*2 + plane
    const int NT32 = p.Cout >> 5;
    const bf16x8* wf = (const bf16x8*)p.wpk + (size_t)(n_tile * (BN / 32) + wn * NTW) * 2 * 64 + lane;
    const size_t ks_stride = (size_t)NT32 * 2 * 64;  // in bf16x8 units
    auto w_load = [&](int it, bf16x8 (&w)[2][NTW][2]) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) w[ks][n][pl] = wf[((size_t)it * 2 + ks) * ks_stride + (n * 2 + pl) * 64];
    };
    // Patch staging descriptors, computed ONCE per workgroup: which halo pixel / channel quad each of this thread's
    // (up to GW_MAXIT) items is, as a global element offset and an LDS offset.  The per-chunk staging loop is then a
    // load, the hi/lo split and two LDS stores per item - the divisions and bounds tests used to be redone for every
    // 32-channel chunk and made the kernel's VALU time comparable to its MFMA time.
    // An item is 16 bytes of one pixel: 4 f32 channels (PREC 0) or 8 fp16 channels (PREC 1, copied as they are).
    constexpr int IPP = PREC == 0 ? 8 : 4;          // items per pixel and 32-channel chunk
    // >= ceil(NPIX * IPP / 256); 256-pixel blocks: TC = 16 -> 324 pixels, 8 -> 340, 4 -> 396, 2 -> 520
    constexpr int GW_MAXIT = PREC == 0 ? (BM == 128 ? 9 : 11) : (BM == 128 ? 5 : (COLT == 2 ? 9 : 7));
    constexpr unsigned GW_NONE = 0xffffffffu, GW_OOB = 0x80000000u;
    unsigned goff[GW_MAXIT], loff[GW_MAXIT];
#pragma unroll
    for (int j = 0; j < GW_MAXIT; ++j) {
      const int idx = FUSE1 ? ((tid & 63) + j * 64) * IPP + (tid >> 6) : tid + j * 256;   // FUSE1: octet = wave
      loff[j] = GW_NONE;
      goff[j] = 0;
      if (idx < NPIX * IPP) {
        const int pix = idx / IPP, ci = idx % IPP;
        const int pr = pix / PW, pc = pix - pr * PW;
        const int gr = row0 - HALO + pr, gc = col0 - HALO + pc;
        bool ok = gr >= 0 && gr < p.rows_total && gc >= 0 && gc < p.W;
        if (FUSE1) ok = ok && (gr % p.Hp) < p.H;   // conv1 writes zeros on the padding rows of a clip
        if (TAPS == 1) ok = ok && gr * p.W + gc < p.m_valid;
        goff[j] = ok ? (unsigned)(((size_t)gr * p.W + gc) * p.Cin + ci * (32 / IPP)) : 0u;
        if (FUSE1) goff[j] = (unsigned)(pr * 21 + pc);   // top-left of the pixel's 3x3 window in the log-mel patch
        loff[j] = (unsigned)(pr * PITCH + pc * BROW + ci * (32 / IPP)) | (ok ? 0u : GW_OOB);
      }
    }
    auto stage_patch = [&](int c) {
#pragma unroll
      for (int j = 0; j < GW_MAXIT; ++j) {
        if (loff[j] == GW_NONE) continue;
        const unsigned lo_off = loff[j] & 0x7fffffffu;
        if (PREC == 0) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (!(loff[j] & GW_OOB)) v = *(const f32x4*)(p.in + (size_t)goff[j] + c * 32);
          u32x2 hi, lo;
          split_bf16x4(v, hi, lo);
          *(u32x2*)(sAh + lo_off) = hi;
          *(u32x2*)(sAl + lo_off) = lo;
        } else {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (!(loff[j] & GW_OOB)) v = *(const u32x4*)((const _Float16*)p.in + (size_t)goff[j] + c * 32);
          *(u32x4*)(sAh + lo_off) = v;
        }
      }
    };
    // PREC 1 has the registers to request the NEXT chunk's patch while the 9 taps of this one run (the fp16 items
    // are copied as they are): the global-load latency leaves the two-barrier window between chunks.
    u32x4 pre[PREC == 1 ? GW_MAXIT : 1];
    auto patch_request = [&](int c) {
#pragma unroll
      for (int j = 0; j < GW_MAXIT; ++j) {
        pre[j] = u32x4{0u, 0u, 0u, 0u};
        if (loff[j] != GW_NONE && !(loff[j] & GW_OOB)) pre[j] = *(const u32x4*)((const _Float16*)p.in + (size_t)goff[j] + c * 32);
      }
    };
    auto patch_commit = [&]() {
#pragma unroll
      for (int j = 0; j < GW_MAXIT; ++j)
        if (loff[j] != GW_NONE) *(u32x4*)(sAh + (loff[j] & 0x7fffffffu)) = pre[j];
    };
    // TAPS 1: the next chunk's f32 rows wait in registers while this chunk's MFMAs run
    constexpr int PRE1 = (TAPS == 1 && PREC == 0) ? 4 : 1;   // 128 rows x 8 items / 256 threads
    f32x4 pre1[PRE1];
    auto rows_request = [&](int c) {
#pragma unroll
      for (int j = 0; j < PRE1; ++j) {
        pre1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (loff[j] != GW_NONE && !(loff[j] & GW_OOB)) pre1[j] = *(const f32x4*)(p.in + (size_t)goff[j] + c * 32);
      }
    };
    auto rows_commit = [&]() {
#pragma unroll
      for (int j = 0; j < PRE1; ++j) {
        if (loff[j] == GW_NONE) continue;
        u32x2 hi, lo;
        split_bf16x4(pre1[j], hi, lo);
        const unsigned lo_off = loff[j] & 0x7fffffffu;
        *(u32x2*)(sAh + lo_off) = hi;
        *(u32x2*)(sAl + lo_off) = lo;
      }
    };
    const int total = nchunk * TAPS;
    bf16x8 wc[2][NTW][2], wnx[2][NTW][2];
    w_load(0, wc);
    float* s1 = (float*)(sAh + PLANE);   // FUSE1: [TR + 4][21] log-mel rows row0-2.., columns col0-2..
    // conv1 + BN + ReLU of channel chunk c1 for this thread's patch items.  With FUSE1 an item is (pixel = lane + 64 j,
    // channel octet = wave), so the 72 weights and 16 BN terms of an octet are wave-uniform: they sit in SGPRs and the
    // multiply-adds take them as scalar operands (the same 9-term fmaf chain per channel as conv_first_kernel).
    F16Guard guard1;
    auto conv1_patch = [&](int c1) {
      const int ch0 = c1 * 32 + __builtin_amdgcn_readfirstlane(tid >> 6) * 8;
      const float* __restrict__ w1 = p.w1 + ch0 * 9;
      const float* __restrict__ sc1 = p.sc1 + ch0;
      const float* __restrict__ sh1 = p.sh1 + ch0;
      float w1r[8][9], sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[e] = sc1[e];
        sh[e] = sh1[e];
#pragma unroll
        for (int t = 0; t < 9; ++t) w1r[e][t] = w1[e * 9 + t];
      }
#pragma unroll
      for (int j = 0; j < GW_MAXIT; ++j) {
        if (loff[j] == GW_NONE) continue;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (!(loff[j] & GW_OOB)) {
          float x[9];
#pragma unroll
          for (int t = 0; t < 9; ++t) x[t] = s1[goff[j] + (t / 3) * 21 + (t % 3)];
          float y[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) a = fmaf(x[t], w1r[e][t], a);
            y[e] = fmaxf(fmaf(a, sc[e], sh[e]), 0.f);
            guard1.track(y[e]);
          }
          v[0] = cvt_pk_f16(y[0], y[1]);
          v[1] = cvt_pk_f16(y[2], y[3]);
          v[2] = cvt_pk_f16(y[4], y[5]);
          v[3] = cvt_pk_f16(y[6], y[7]);
        }
        *(u32x4*)(sAh + (loff[j] & 0x7fffffffu)) = v;
      }
    };
    if (FUSE1) {
      const int S1W = TC + 4, S1H = TR + 4;
      for (int idx = tid; idx < S1W * S1H; idx += 256) {
        const int pr = idx / S1W, pc = idx - pr * S1W;
        const int gr = row0 - 2 + pr, gc = col0 - 2 + pc;
        float v = 0.f;
        if (gr >= 0 && gr < p.rows_total && gc >= 0 && gc < p.W) v = p.in1[(size_t)gr * p.W + gc];
        s1[pr * 21 + pc] = v;
      }
      __syncthreads();
      conv1_patch(0);
    } else {
      stage_patch(0);
    }
    __syncthreads();
    int tap = 0, c = 0;
#pragma unroll 1
    for (int it = 0; it < total; ++it) {
      if (PREC == 1 && !FUSE1 && tap == 0 && c + 1 < nchunk) patch_request(c + 1);
      if (TAPS == 1 && PREC == 0 && c + 1 < nchunk) rows_request(c + 1);
      w_load(it + 1 < total ? it + 1 : it, wnx);
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int aoff = TAPS == 9 ? ky * PITCH + kx * BROW : 0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[MW], al[MW];
        constexpr bool LATE_A = PREC == 0 && MW >= 8;   // split-bf16 with 8 tiles: fragments are read tile by tile below
#pragma unroll
        for (int m = 0; m < MW; ++m) {
          if (LATE_A) continue;
          if (COLT && ((kx == 0 && m % (COLT ? COLT : 1) == 0 && at_left) ||
                       (kx == 2 && m % (COLT ? COLT : 1) == COLT - 1 && at_right))) continue;
          ah[m] = *(const bf16x8*)(sAh + aoff + pbase[m] + ks * 16);
          if (PREC == 0) al[m] = *(const bf16x8*)(sAl + aoff + pbase[m] + ks * 16);
        }
#pragma unroll
        for (int m = 0; m < MW; ++m) {
          // column tile whose tap reads the zero padding beside the image: nothing to add
          if (COLT && ((kx == 0 && m % (COLT ? COLT : 1) == 0 && at_left) ||
                       (kx == 2 && m % (COLT ? COLT : 1) == COLT - 1 && at_right))) continue;
          if (LATE_A) {
            ah[m] = *(const bf16x8*)(sAh + aoff + pbase[m] + ks * 16);
            al[m] = *(const bf16x8*)(sAl + aoff + pbase[m] + ks * 16);
          }
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            if (PREC == 0) {
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], wc[ks][n][0], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wc[ks][n][1], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wc[ks][n][0], acc[m][n], 0, 0, 0);
            } else {
              const f16x8 a = __builtin_bit_cast(f16x8, ah[m]);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, wc[ks][n][1]), acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(f16x8, wc[ks][n][0]), acc[m][n], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) wc[ks][n][pl] = wnx[ks][n][pl];
      if (++tap == TAPS) {
        tap = 0;
        if (++c < nchunk) {
          __syncthreads();  // every wave is done with the patch of the previous chunk
          if (FUSE1) conv1_patch(c);
          else if (TAPS == 1 && PREC == 0) rows_commit();
          else if (PREC == 1) patch_commit();
          else stage_patch(c);
          __syncthreads();
        }
      }
    }
    if (FUSE1) guard1.commit(p.ovf);
  }
  if constexpr (COLT != 0) conv_epilogue_cols<BN, MODE, COLT, MW, PREC == 0>(p, acc, n_tile, row0, col0, wn, lane);
  else conv_epilogue<BN, MODE, MW, NTW, PREC == 1>(p, acc, n_tile, row0, col0, wm, wn, lane);
}

template <int BN, int MODE, int PREC = 0, int WM = 2, int BM = 128, bool FUSE1 = false, int TAPS = 9, int COLT = 0>
int launch_conv_gw(ConvParams p, hipStream_t s) {
  const int TC = 1 << p.tc_log2, TR = BM >> p.tc_log2;
  p.MT = ((p.rows_total + TR - 1) / TR) * p.mt_cols;
  unsigned grid;
  if (p.map_mode == 2) grid = (unsigned)(((p.MT + 7) / 8) * 8 * p.NT);
  else if (p.map_mode == 3) grid = (unsigned)(((p.MT + 8 / p.NT - 1) / (8 / p.NT)) * 8);
  else grid = (unsigned)(p.MT * p.NT);
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  const int pitch = (TC + 2 * HALO) * BROW + (COLT ? colt_row_pad_slots(TC) : patch_row_pad_slots(TC)) * 8;
  // hi + lo planes (bf16) or one fp16 plane (+ the log-mel patch of the fused first layer)
  const size_t lds = (size_t)(TR + 2 * HALO) * pitch * 2 * (PREC == 0 ? 2 : 1) + (FUSE1 ? (TR + 4) * 21 * 4 : 0);
  hipLaunchKernelGGL((conv3x3_gw_kernel<BN, MODE, PREC, WM, BM, FUSE1, TAPS, COLT>), dim3(grid), dim3(256), lds, s, p);
  return ac_check_launch();
}

template <int BN, int MODE>
int launch_conv_bf16x3(const ConvParams& p, hipStream_t s) {
  unsigned grid;
  if (p.map_mode == 2) grid = (unsigned)(((p.MT + 7) / 8) * 8 * p.NT);
  else grid = (unsigned)(p.MT * p.NT);
  const int TC = 1 << p.tc_log2, TR = 128 >> p.tc_log2;
  const int pitch = (TC + 2) * BROW + patch_row_pad_slots(TC) * 8;
  const size_t lds = ((size_t)(TR + 2) * pitch * 2 + (size_t)4 * BN * BROW) * 2;  // bytes (bf16 elements x 2)
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)conv3x3_bf16x3_kernel<BN, MODE>, 160 * 1024, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_bf16x3_kernel<BN, MODE>), dim3(grid), dim3(256), lds, s, p);
  return ac_check_launch();
}

// ------------------------------------------------------------------------------------------------
// First conv of block 1: Cin = 1, K = 9 - not a GEMM.  HBM-write bound (64 channels out per pixel).
// 16 lanes cover the 64 output channels of one pixel (float4 each), so a wave stores 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------
struct ConvFirstParams {
  const float* in;    // [rows_total][W]
  const float* w;     // [64][9]  (OIHW with I = 1)
  const float* scale;
  const float* shift;
  float* out;         // [rows_total][W][64]
  int rows_total, Hp, H, W;
  unsigned* ovf = nullptr;   // fp16 output: range guard flag (see F16Guard), may be null
};

template <bool OUT16>
__global__ __launch_bounds__(256) void conv_first_kernel(ConvFirstParams p) {
  constexpr int RT = 4;  // rows per block
  __shared__ float patch[RT + 2][64 + 2];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * RT;
  for (int idx = tid; idx < (RT + 2) * 66; idx += 256) {
    const int pr = idx / 66, pc = idx - pr * 66;
    const int gr = row0 - 1 + pr, gc = pc - 1;
    float v = 0.f;
    if (gr >= 0 && gr < p.rows_total && gc >= 0 && gc < p.W) v = p.in[(size_t)gr * p.W + gc];
    patch[pr][pc] = v;
  }
  const int cg = tid & 15, ps = tid >> 4;
  float w[4][9], sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = p.scale[cg * 4 + j];
    sh[j] = p.shift[cg * 4 + j];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[j][t] = p.w[(cg * 4 + j) * 9 + t];
  }
  __syncthreads();
  F16Guard guard;
  for (int px = ps; px < RT * p.W; px += 16) {
    const int r = px / p.W, c = px - r * p.W;
    const int gr = row0 + r;
    if (gr >= p.rows_total) break;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((gr % p.Hp) < p.H) {
      float x[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) x[t] = patch[r + t / 3][c + t % 3];
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) s = fmaf(x[t], w[j][t], s);
        a[j] = fmaxf(fmaf(s, sc[j], sh[j]), 0.f);
        if (OUT16) guard.track(a[j]);
      }
      o = make_float4(a[0], a[1], a[2], a[3]);
    }
    const size_t oi = ((size_t)gr * p.W + c) * 64 + cg * 4;
    if (OUT16) {
      u32x2 h;
      h.x = cvt_pk_f16(o.x, o.y);
      h.y = cvt_pk_f16(o.z, o.w);
      *(u32x2*)((_Float16*)p.out + oi) = h;
    } else {
      *(float4*)(p.out + oi) = o;
    }
  }
  if (OUT16) guard.commit(p.ovf);
}

template <int BN, int MODE>
int launch_conv(const ConvParams& p, hipStream_t s) {
  unsigned grid;
  if (p.map_mode == 2) grid = (unsigned)(((p.MT + 7) / 8) * 8 * p.NT);
  else grid = (unsigned)(p.MT * p.NT);
  hipLaunchKernelGGL((conv3x3_mfma_kernel<BN, MODE>), dim3(grid), dim3(256), 0, s, p);
  return ac_check_launch();
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_conv3x3_bn_relu(const float* in, const float* wpk, const float* scale, const float* shift,
                                  float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                  int map_mode, void* stream) {
  if (!in || !wpk || !scale || !shift || !out) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || W < 2 || (W & (W - 1)) || Cin % 32 || Cout % 64) return AC_ERR_ARG;
  if (mode < 0 || mode > 2) return AC_ERR_ARG;
  if (mode == MODE_POOL && (Hp & 1)) return AC_ERR_ARG;
  if (mode == MODE_MEANW && W != 2) return AC_ERR_ARG;
  ConvParams p;
  p.m_valid = 0;
  p.in = in; p.wpk = wpk; p.scale = scale; p.shift = shift; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TC = W < 16 ? W : 16;
  int l2 = 0;
  while ((1 << l2) < TC) ++l2;
  p.tc_log2 = l2;
  const int TR = 128 / TC;
  p.mt_cols = W / TC;
  p.MT = ((p.rows_total + TR - 1) / TR) * p.mt_cols;
  const int BN = (Cout % 128 == 0) ? 128 : 64;
  p.NT = Cout / BN;
  p.Hp_out = Hp / 2; p.H_out = H / 2; p.W_out = W / 2;
  if (map_mode < 0) map_mode = (p.NT % 8 == 0) ? 1 : 2;  // default: see the kernel's mapping comment
  if (map_mode == 1 && p.NT % 8 != 0) return AC_ERR_ARG;
  p.map_mode = map_mode;
  hipStream_t s = (hipStream_t)stream;
  if (BN == 128) {
    if (mode == MODE_FULL) return launch_conv<128, MODE_FULL>(p, s);
    if (mode == MODE_POOL) return launch_conv<128, MODE_POOL>(p, s);
    return launch_conv<128, MODE_MEANW>(p, s);
  } else {
    if (mode == MODE_FULL) return launch_conv<64, MODE_FULL>(p, s);
    if (mode == MODE_POOL) return launch_conv<64, MODE_POOL>(p, s);
    return launch_conv<64, MODE_MEANW>(p, s);
  }
}

extern "C" int ac_conv3x3_bn_relu_bf16x3(const float* in, const void* wpk, const float* scale, const float* shift,
                                         float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                         int map_mode, void* stream) {
  if (!in || !wpk || !scale || !shift || !out) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || W < 2 || (W & (W - 1)) || Cin % 32 || Cout % 64) return AC_ERR_ARG;
  if (mode < 0 || mode > 2) return AC_ERR_ARG;
  if (mode == MODE_POOL && (Hp & 1)) return AC_ERR_ARG;
  if (mode == MODE_MEANW && W != 2) return AC_ERR_ARG;
  ConvParams p;
  p.m_valid = 0;
  p.in = in; p.wpk = (const float*)wpk; p.scale = scale; p.shift = shift; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TC = W < 16 ? W : 16;
  int l2 = 0;
  while ((1 << l2) < TC) ++l2;
  p.tc_log2 = l2;
  const int TR = 128 / TC;
  p.mt_cols = W / TC;
  p.MT = ((p.rows_total + TR - 1) / TR) * p.mt_cols;
  const int BN = (Cout % 128 == 0) ? 128 : 64;
  p.NT = Cout / BN;
  p.Hp_out = Hp / 2; p.H_out = H / 2; p.W_out = W / 2;
  if (map_mode < 0) map_mode = (p.NT % 8 == 0) ? 1 : 2;
  if (map_mode == 1 && p.NT % 8 != 0) return AC_ERR_ARG;
  p.map_mode = map_mode;
  hipStream_t s = (hipStream_t)stream;
  if (BN == 128) {
    if (mode == MODE_FULL) return launch_conv_bf16x3<128, MODE_FULL>(p, s);
    if (mode == MODE_POOL) return launch_conv_bf16x3<128, MODE_POOL>(p, s);
    return launch_conv_bf16x3<128, MODE_MEANW>(p, s);
  } else {
    if (mode == MODE_FULL) return launch_conv_bf16x3<64, MODE_FULL>(p, s);
    if (mode == MODE_POOL) return launch_conv_bf16x3<64, MODE_POOL>(p, s);
    return launch_conv_bf16x3<64, MODE_MEANW>(p, s);
  }
}

static int conv_gw_dispatch(int prec, const float* in, const void* wfrag, const float* scale, const float* shift,
                            float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode, int map_mode,
                            int out_f32, unsigned* overflow_flag, void* stream) {
  if (!in || !wfrag || !scale || !shift || !out) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || W < 2 || (W & (W - 1)) || Cin % 32 || Cout % 64) return AC_ERR_ARG;
  if (mode < 0 || mode > 2) return AC_ERR_ARG;
  if (mode == MODE_POOL && (Hp & 1)) return AC_ERR_ARG;
  if (mode == MODE_MEANW && W != 2) return AC_ERR_ARG;
  if (out_f32 && (prec != 1 || mode != MODE_POOL)) return AC_ERR_ARG;
  // the staging descriptors of this kernel hold 32-bit element offsets into `in`
  if ((unsigned long long)B * Hp * W * Cin >= (1ull << 32)) return AC_ERR_ARG;
  ConvParams p;
  p.ovf = prec == 1 ? overflow_flag : nullptr;
  p.out32 = out_f32;
  p.m_valid = 0;
  p.in = in; p.wpk = (const float*)wfrag; p.scale = scale; p.shift = shift; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  const int TC = W < 16 ? W : 16;
  int l2 = 0;
  while ((1 << l2) < TC) ++l2;
  p.tc_log2 = l2;
  const int TR = 128 / TC;
  p.mt_cols = W / TC;
  p.MT = ((p.rows_total + TR - 1) / TR) * p.mt_cols;
  const int BN = (Cout % 128 == 0) ? 128 : 64;
  p.NT = Cout / BN;
  p.Hp_out = Hp / 2; p.H_out = H / 2; p.W_out = W / 2;
  if (map_mode < 0) map_mode = (p.NT % 8 == 0) ? 1 : 2;
  if (map_mode == 1 && p.NT % 8 != 0) return AC_ERR_ARG;
  if (map_mode == 3 && (p.NT > 8 || (8 % p.NT) != 0)) return AC_ERR_ARG;
  p.map_mode = map_mode;
  hipStream_t s = (hipStream_t)stream;
  if (prec == 0) {
    // The 2- and 4-column layers (blocks 5-6) as COLUMN tiles, like the fp16 tier: the taps that read the zero padding
    // beside the image (a third / a sixth of all products) are skipped, and a wave walks four 32-row tiles per weight
    // fragment (1x4 wave grid) instead of two
    static const bool colt0 = getenv("AC_GW_NO_COLT0") == nullptr;
    if (BN == 128 && colt0 && TC == 2) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 0, 1, 128, false, 9, 2>(p, s);
      if (mode == MODE_MEANW) return launch_conv_gw<128, MODE_MEANW, 0, 1, 128, false, 9, 2>(p, s);
    }
    if (BN == 128 && colt0 && TC == 4) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 0, 1, 128, false, 9, 4>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<128, MODE_POOL, 0, 1, 128, false, 9, 4>(p, s);
    }
    // 8 and more columns: 256-pixel column-tile blocks (8 columns x 32 rows), a wave walks eight tiles per weight
    // fragment - what took the fp16 tier from 5.08 to 4.32 ms, for the split-bf16 tier
    static const bool colt0_8 = getenv("AC_GW_NO_COLT0_8") == nullptr;
    if (BN == 128 && colt0 && colt0_8 && W >= 8 && mode != MODE_MEANW) {
      p.tc_log2 = 3;
      p.mt_cols = W / 8;
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 0, 1, 256, false, 9, 8>(p, s);
      return launch_conv_gw<128, MODE_POOL, 0, 1, 256, false, 9, 8>(p, s);
    }
    if (BN == 128) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<128, MODE_POOL>(p, s);
      return launch_conv_gw<128, MODE_MEANW>(p, s);
    } else {
      if (mode == MODE_FULL) return launch_conv_gw<64, MODE_FULL>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<64, MODE_POOL>(p, s);
      return launch_conv_gw<64, MODE_MEANW>(p, s);
    }
  }
  static const int bm256 = getenv("AC_GW_BM256") ? atoi(getenv("AC_GW_BM256")) : 1;   // bit 0: Cout 64, bit 1: Cout >= 128
  if (BN == 128) {
    if (TC == 16 && (bm256 & 2)) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 2, 256>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<128, MODE_POOL, 1, 2, 256>(p, s);
    }
    static const bool colt = getenv("AC_GW_NO_COLT") == nullptr;   // column tiles for the 2- and 4-column layers
    if (colt && TC == 2) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 1, 128, false, 9, 2>(p, s);
      if (mode == MODE_MEANW) return launch_conv_gw<128, MODE_MEANW, 1, 1, 128, false, 9, 2>(p, s);
    }
    static const bool wide1 = getenv("AC_GW_BM256W1") != nullptr;   // 256-pixel blocks, 1x4 waves, window tiles (W >= 16)
    if (wide1 && TC == 16) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 1, 256>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<128, MODE_POOL, 1, 1, 256>(p, s);
    }
    // Layers of 8 and more columns (up to AC_GW_COLT8, default all; 0: off) run as 256-pixel column-tile blocks of
    // 8 columns x 32 rows: a wave walks eight 32-row tiles per weight fragment - half the weight-fragment bytes per MFMA
    // of the 128-pixel blocks, which is what those were waiting for (conv stack 4.60 -> 4.33 ms)
    static const int colt8 = getenv("AC_GW_COLT8") ? atoi(getenv("AC_GW_COLT8")) : 64;
    if (colt8 > 0 && W >= 8 && W <= colt8 && mode != MODE_MEANW) {
      p.tc_log2 = 3;
      p.mt_cols = W / 8;
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 1, 256, false, 9, 8>(p, s);
      return launch_conv_gw<128, MODE_POOL, 1, 1, 256, false, 9, 8>(p, s);
    }
    static const bool colt2_256 = getenv("AC_GW_COLT2_256") != nullptr;
    if (colt && TC == 2 && colt2_256) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 1, 256, false, 9, 2>(p, s);
      if (mode == MODE_MEANW) return launch_conv_gw<128, MODE_MEANW, 1, 1, 256, false, 9, 2>(p, s);
    }
    static const bool colt4_256 = getenv("AC_GW_COLT4_128") == nullptr;   // 4-column layers: 256-pixel blocks (64 rows)
    if (colt && TC == 4 && colt4_256) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 1, 256, false, 9, 4>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<128, MODE_POOL, 1, 1, 256, false, 9, 4>(p, s);
    }
    if (colt && TC == 4) {
      if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 1, 128, false, 9, 4>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<128, MODE_POOL, 1, 1, 128, false, 9, 4>(p, s);
    }
    if (mode == MODE_FULL) return launch_conv_gw<128, MODE_FULL, 1, 1>(p, s);
    if (mode == MODE_POOL) return launch_conv_gw<128, MODE_POOL, 1, 1>(p, s);
    return launch_conv_gw<128, MODE_MEANW, 1, 1>(p, s);
  } else {
    if (TC == 16 && (bm256 & 1)) {
      if (mode == MODE_FULL) return launch_conv_gw<64, MODE_FULL, 1, 2, 256>(p, s);
      if (mode == MODE_POOL) return launch_conv_gw<64, MODE_POOL, 1, 2, 256>(p, s);
    }
    if (mode == MODE_FULL) return launch_conv_gw<64, MODE_FULL, 1>(p, s);
    if (mode == MODE_POOL) return launch_conv_gw<64, MODE_POOL, 1>(p, s);
    return launch_conv_gw<64, MODE_MEANW, 1>(p, s);
  }
}

extern "C" int ac_conv3x3_bn_relu_bf16x3_gw(const float* in, const void* wfrag, const float* scale, const float* shift,
                                            float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                            int map_mode, void* stream) {
  return conv_gw_dispatch(0, in, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, 0, nullptr, stream);
}

extern "C" int ac_conv3x3_bn_relu_f16x2_gw(const void* in, const void* wfrag, const float* scale, const float* shift,
                                           void* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                           int map_mode, int out_f32, unsigned int* overflow_flag, void* stream) {
  return conv_gw_dispatch(1, (const float*)in, wfrag, scale, shift, (float*)out, B, Hp, H, W, Cin, Cout, mode, map_mode,
                          out_f32, overflow_flag, stream);
}

// Block 1 of the f16x2 tier in one kernel: conv1 (Cin = 1) + BN + ReLU computed into the patch, conv2 + BN + ReLU +
// 2x2 pool on the matrix cores.  in1 [B*Hp][64] f32 log-mel, out [B*Hp/2][32][64] fp16.
extern "C" int ac_conv3x3_block1_f16x2(const float* in1, const float* w1, const float* scale1, const float* shift1,
                                       const void* wfrag2, const float* scale2, const float* shift2, void* out,
                                       int B, int Hp, int H, int W, unsigned int* overflow_flag, void* stream) {
  if (!in1 || !w1 || !scale1 || !shift1 || !wfrag2 || !scale2 || !shift2 || !out) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || (Hp & 1) || W != 64) return AC_ERR_ARG;
  ConvParams p;
  p.ovf = overflow_flag;
  p.m_valid = 0;
  p.in = nullptr; p.wpk = (const float*)wfrag2; p.scale = scale2; p.shift = shift2; p.out = (float*)out;
  p.in1 = in1; p.w1 = w1; p.sc1 = scale1; p.sh1 = shift1;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W; p.Cin = 64; p.Cout = 64;
  p.tc_log2 = 4;
  p.mt_cols = W / 16;
  p.MT = 0;   // set by the launcher from the block height
  p.NT = 1;
  p.Hp_out = Hp / 2; p.H_out = H / 2; p.W_out = W / 2;
  p.map_mode = 2;
  return launch_conv_gw<64, MODE_POOL, 1, 2, 256, true>(p, (hipStream_t)stream);
}

// Y[M][N] = act(X[M][K] W^T + bias) on the split-bf16 matrix path: the 1-tap instance of the kernel above over a
// 2-pixel-wide "image" of M/2 rows, 64-channel column tiles (more workgroups: M is only a few thousand rows).
// wfrag = W split and packed like the conv weights with one tap.
extern "C" int ac_linear_bf16x3(const float* X, const void* wfrag, const float* ones, const float* bias, float* Y,
                                int M, int N, int K, int relu, void* stream) {
  if (!X || !wfrag || !ones || !bias || !Y) return AC_ERR_ARG;
  if (M <= 0 || K % 32 || N % 64) return AC_ERR_ARG;
  if ((unsigned long long)(M + 1) * K >= (1ull << 32)) return AC_ERR_ARG;   // 32-bit staging offsets
  ConvParams p;
  p.m_valid = 0;
  p.in = X; p.wpk = (const float*)wfrag; p.scale = ones; p.shift = bias; p.out = Y;
  p.in1 = nullptr; p.w1 = nullptr; p.sc1 = nullptr; p.sh1 = nullptr;
  p.rows_total = (M + 1) / 2; p.Hp = p.rows_total + 1; p.H = p.rows_total; p.W = 2; p.Cin = K; p.Cout = N;
  p.m_valid = M;
  p.tc_log2 = 1;
  p.mt_cols = 1;
  p.MT = 0;   // set by the launcher
  p.NT = N / 64;
  p.Hp_out = 1; p.H_out = 1; p.W_out = 1;
  p.map_mode = (p.NT % 8 == 0) ? 1 : 2;
  hipStream_t s = (hipStream_t)stream;
  return relu ? launch_conv_gw<64, MODE_FULL, 0, 2, 128, false, 1>(p, s)
              : launch_conv_gw<64, MODE_LINEAR, 0, 2, 128, false, 1>(p, s);
}

extern "C" int ac_conv3x3_first(const float* in, const float* w, const float* scale, const float* shift,
                                float* out, int B, int Hp, int H, int W, void* stream) {
  if (!in || !w || !scale || !shift || !out || B <= 0 || Hp <= H || W != 64) return AC_ERR_ARG;
  ConvFirstParams p;
  p.in = in; p.w = w; p.scale = scale; p.shift = shift; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W;
  const unsigned grid = (unsigned)((p.rows_total + 3) / 4);
  hipLaunchKernelGGL(conv_first_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return ac_check_launch();
}

extern "C" int ac_conv3x3_first_f16(const float* in, const float* w, const float* scale, const float* shift,
                                    void* out, int B, int Hp, int H, int W, unsigned int* overflow_flag, void* stream) {
  if (!in || !w || !scale || !shift || !out || B <= 0 || Hp <= H || W != 64) return AC_ERR_ARG;
  ConvFirstParams p;
  p.ovf = overflow_flag;
  p.in = in; p.w = w; p.scale = scale; p.shift = shift; p.out = (float*)out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W;
  const unsigned grid = (unsigned)((p.rows_total + 3) / 4);
  hipLaunchKernelGGL(conv_first_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return ac_check_launch();
}
