// Conv block 1 of Cnn14 at f32 grade in ONE kernel: conv1 (1 -> 64 channels) + BN + ReLU computed straight into the
// staging of conv2 (64 -> 64) + BN + ReLU + 2x2 average pool, conv2 as the 1-D Winograd F(4,3) along time on split-bf16
// operands of csrc/conv3x3_wino43.hip.  The 64-channel intermediate (1.07 GB per 64 ten-second clips) never exists in
// HBM.  Replaces ConvBlock.forward of conv_block1 + F.avg_pool2d (cnn_encoder.py:59-75, :431-432).
//
// Geometry.  W = 64 mel columns, Cin = Cout = 64: K is only four 16-channel steps, so a workgroup's life would be mostly
// prologue and epilogue - the kernel is PERSISTENT: min(tiles, CUs) workgroups of four 512-register waves walk the tiles,
// and the staging of a tile's first K step runs under the MFMAs of the previous tile's last step.  A tile is 2 row quads
// x 64 columns (full width: the halo columns are LDS columns that stay zero) x 64 channels; wave w owns channel tile
// w & 1 and quad w >> 1 as two MFMA tiles of 32 columns x 6 positions = 192 accumulators.
//
// Staging (FORM 1: conv1 on the vector ALUs).  Thread t = (channel quad t & 3, column t >> 2) holds the 12 x 3 log-mel values around its column for
// both quads of the tile in registers (loaded a tile ahead).  Per K step it evaluates conv1 + BN + ReLU for its 4
// channels on the 10 rows 4 q0 - 1 .. 4 q0 + 8 (the same 9-term fmaf chain, BN and zero rows as conv_first_kernel,
// csrc/conv3x3.hip: bit-identical to the two-kernel form), transforms the two quads (F(4,3): 6 positions each), splits
// into bf16 hi + lo and stores the planes - one row or two pieces per MFMA group.  FORM 0: the rows come from a
// 64-channel activation tensor in HBM instead (the unfused reference form of the same kernel; tests).
//
// Staging (FORM 2: conv1 on the matrix cores, the default).  The 9-term chain above is 1760 of the 4700 vector
// instructions of a tile - the kernel was bound by them, not by its 432 MFMAs.  conv1 is a GEMM too: [32 channels] x
// [K = 16: tap (ky, kx) in slot 4 ky + kx, BN shift in slot 12 against a constant 1] x [32 columns of one row], on the same
// split-bf16 operands (hi x lo + lo x hi + hi x hi) as conv2, BN scale folded into the taps before the split.  Wave w
// computes, once per channel tile ct (32 channels = two K steps of conv2), the six rows of quad w >> 1 for columns
// 32 (w & 1) ..: 18 MFMAs, +8 % on the matrix pipe.  A lane then HOLDS what the staging needs - four groups of four
// consecutive channels of one column over the six rows of its quad: ReLU, F(4,3) transform, split, plane stores as before;
// groups 0-1 (the next K step) at once, groups 2-3 (the step after) one step later, ONE piece per MFMA group (a wave per
// SIMD hides ~5 vector instructions per MFMA gap; more are paid in full).  The B operand of row r is two packed log-mel
// rows (columns c - 1 .. c + 1; seven unaligned 16-byte loads per lane, requested three groups before the step); lanes
// 32-63 supply ky = 2 and the constant.  A row's accumulator is read a whole MFMA group after its last product.
// Not bit-identical to conv_first_kernel any more: conv1 now has the split-bf16 grade of every other layer of the tier.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "ac_common.h"
#include "ac_drop.h"
#include "ac_wino43.h"

namespace {

#ifndef B1_RING
#define B1_RING 3     // conv2 weight fragment ring (18 % B1_RING == 0): a group's pair is requested B1_RING - 1 groups ahead
#endif

#ifndef B1_RING_MFC
#define B1_RING_MFC 6   // FORM 2 without dropout has the registers for five groups of distance (477 vs 485 us)
#endif
#ifndef B1_SGB_MEM
#define B1_SGB_MEM 0x0b0   // one LDS or memory instruction per gap
#endif
#ifndef B1_SGB    // development: > 0 = ask the scheduler for (1 MFMA, B1_SGB vector instructions, 1 LDS, 1 load) x 9 per MFMA group
#define B1_SGB 0
#endif
#ifndef B1_KO    // development: 1 no staging (conv1 / transform / split / plane stores), 2 no MFMAs, 4 no epilogue stores,
                 // FORM 2: 8 no plane stores, 16 no transform / split / plane stores, 32 no conv1 products, 64 no row split
#define B1_KO 0
#endif
#ifdef B1_CLK
__device__ unsigned long long b1_clk[8];
#endif

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct B1Params {
  const float* in;       // FORM 1, 2: [B*Hp][64] log-mel after bn0; FORM 0: [B*Hp][64][64] conv1 output
  const float* w1;       // [64][9]
  const float* sc1;
  const float* sh1;
  const void* wpk;       // conv2, F(4,3) pack [4][18][2][2 (hi, lo)][64 lanes][8] bf16
  const float* scale;
  const float* shift;
  float* out;            // [B*Hp/2][32][64]
  int rows_total, Hp, H;
  int tiles;             // row blocks of 8 rows
  const int* clip_frames;
  int need_mul, need_add;
  Drop drop;
};

constexpr int B1_W = 64, B1_COLP = 3;
constexpr int B1_HALF = (((B1_W + 2) * B1_COLP * 16 + 127) / 128) * 128 + 64;   // 3264
constexpr int B1_PLANE = 2 * B1_HALF;
constexpr int B1_VBUF = 12 * B1_PLANE;                                         // 78336
constexpr int B1_WT = 2 * B1_VBUF;                                             // conv1 table [64][12] floats behind the planes
constexpr int B1_LDS = B1_WT + 4096;                                                // FORM 1: 3072 bytes; FORM 2: A operands of conv1
static_assert(B1_LDS <= 160 * 1024, "LDS");

// FORM 0: conv2 on a 64-channel input in HBM; 1: conv1 on the vector ALUs inside the staging; 2: conv1 on the matrix cores
template <int FORM, bool DROP>
__global__ __launch_bounds__(256, 1) void block1_w4_kernel(B1Params p) {
  constexpr bool FUSED = FORM == 1, MFC = FORM == 2;
  constexpr int HALF = B1_HALF, PLANE = B1_PLANE, VBUF = B1_VBUF, COLP = B1_COLP;
  extern __shared__ __attribute__((aligned(128))) unsigned char dsm_raw[];
  unsigned char* sV = dsm_raw;
  float* wt = (float*)(dsm_raw + B1_WT);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int cg = wave & 1, wq = wave >> 1;   // channel tile, quad of this wave's two MFMA tiles (columns 0-31, 32-63)
  const int nblk = (int)gridDim.x;

  auto tile_live = [&](int t) { return w4_rows_live(8 * t, 8, p.rows_total, p.Hp, p.H, p.clip_frames, p.need_mul, p.need_add); };
  auto zero_tile = [&](int t) {   // 4 pooled rows x 32 columns x 64 channels
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    float* o = p.out + (size_t)t * 4 * 32 * 64;
    const long left = ((long)p.rows_total / 2 - 4L * t) * 32 * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = (k * 256 + tid) * 4;
      if (i < left) *(f32x4*)(o + i) = z;
    }
  };
  auto next_live = [&](int t) {   // the next live tile of this workgroup after t; dead ones on the way are stored as zeros
    t += nblk;
    while (t < p.tiles && !tile_live(t)) { zero_tile(t); t += nblk; }
    return t;
  };
  int tile = next_live((int)blockIdx.x - nblk);
  if (tile >= p.tiles) return;

  // ---- once per workgroup: zero halo columns of both plane buffers, conv1 table ----
  {
    constexpr int NZ = 2 * 24 * 2 * COLP;   // buffers x (planes x k-halves) x 2 columns x COLP slots
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < NZ; i += 256) {
      const int slot = i % COLP, col = (i / COLP) & 1, hh = (i / (2 * COLP)) % 24, b = i / (48 * COLP);
      *(f32x4*)(sV + b * VBUF + hh * HALF + ((col ? B1_W + 1 : 0) * COLP + slot) * 16) = z;
    }
    if (FUSED) {
      for (int i = tid; i < 64 * 12; i += 256) {
        const int ch = i / 12, k = i % 12;
        wt[i] = k < 9 ? p.w1[ch * 9 + k] : (k == 9 ? p.sc1[ch] : (k == 10 ? p.sh1[ch] : 0.f));
      }
    }
    if (MFC && tid < 128) {
      // A operand of conv1 for channel tile ct = tid >> 6, lane (i, h): channel 32 ct + i, slots 8 h .. 8 h + 7 =
      // (ky = 2 h, kx = 0..3), (ky = 2 h + 1, kx = 0..3); kx = 3 and ky = 3 are zeros but slot 12 (ky = 3, kx = 0): the BN
      // shift.  Taps times the BN scale, split into bf16 hi + lo: [ct][hi, lo][lane] 16 bytes
      const int ct = tid >> 6, l = tid & 63, ch = 32 * ct + (l & 31), h = l >> 5;
      const float sc = p.sc1[ch];
      f32x4 v[2];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int ky = 2 * h + q / 4, kx = q % 4;
        float w = 0.f;
        if (ky < 3 && kx < 3) w = p.w1[ch * 9 + ky * 3 + kx] * sc;
        if (ky == 3 && kx == 0) w = p.sh1[ch];
        v[q / 4][q % 4] = w;
      }
      u32x2 hi0, lo0, hi1, lo1;
      split_bf16x4(v[0], hi0, lo0);
      split_bf16x4(v[1], hi1, lo1);
      unsigned* dst = (unsigned*)(dsm_raw + B1_WT) + (ct * 2 * 64 + l) * 4;
      dst[0] = hi0.x; dst[1] = hi0.y; dst[2] = hi1.x; dst[3] = hi1.y;
      dst[256] = lo0.x; dst[257] = lo0.y; dst[258] = lo1.x; dst[259] = lo1.y;
    }
  }

  // A fragment of MFMA tile m (columns 32 m ..), tap kx, plane (pos, hl): lane (i, half) reads item (column 32 m + i + kx,
  // quad wq), k-half `half`
  unsigned pb[2];
  {
    const int i = lane & 31;
#pragma unroll
    for (int m = 0; m < 2; ++m) pb[m] = (unsigned)(half * HALF + ((32 * m + i) * COLP + wq) * 16);
  }
  // staging: channel quad cq of column scol, both quads of the tile
  const int cq = tid & 3, scol = tid >> 2;
  unsigned lofs[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) lofs[j] = (unsigned)((cq >> 1) * HALF + ((scol + 1) * COLP + j) * 16 + (cq & 1) * 8);

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 4 * 18 * 4096, 0x00020000);
  const unsigned wvoff = (unsigned)(cg * 2048 + lane * 16);
  auto w_load = [&](int s, int gi, bf16x8 (&w)[2]) {
    const unsigned soff = (unsigned)((s & 3) * 18 + gi) * 4096u;
    w[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, soff, 0));
    w[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + 1024u, soff, 0));
  };

  // input descriptor: FUSED [rows][64] floats (16.8 MB at 64 ten-second clips), else [rows][64][64] rebased per tile
  const size_t in_elems = (size_t)p.rows_total * 64 * (FORM ? 1 : 64);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)(in_elems * 4 < 0x7fffffffull ? in_elems * 4 : 0x7fffffffull), 0x00020000);

  // ---- FUSED: the 12 x 3 log-mel values of rows 8 t - 2 .. 8 t + 9, columns scol - 1 .. scol + 1 ----
  auto load_x = [&](int t, float (&x)[12][3]) {
    const int cl = scol == 0 ? 0 : scol - 1, cr = scol == 63 ? 63 : scol + 1;   // clamped: the image border reads as zero below
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int gr = 8 * t - 2 + j;
      const unsigned off = (unsigned)(gr * 256);                   // rows above / below the batch: out of range -> zeros
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off + (unsigned)(cl * 4), 0, 0));
      const float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off + (unsigned)(scol * 4), 0, 0));
      const float c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off + (unsigned)(cr * 4), 0, 0));
      x[j][0] = scol == 0 ? 0.f : a;
      x[j][1] = b;
      x[j][2] = scol == 63 ? 0.f : c;
    }
  };
  // one dword of each 128-byte line of those rows, by the first lanes of wave 0, into a register nobody reads: the lines
  // are in L2 when load_x asks for them two steps later
  float touch = 0.f;
  auto touch_x = [&](int t) {
    if (tid < 24) {
      const unsigned off = (unsigned)(((8 * t - 2) * 64 + tid * 32) * 4);
      touch = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    }
  };
  // rows 8 t - 1 + r, r = 0..9, that hold data: inside the batch and below the clip's H valid rows (the others are the zero
  // rows of conv1's OUTPUT: conv2's vertical padding, not relu(shift))
  auto row_mask = [&](int t) {
    unsigned m = 0;
    for (int r = 0; r < 10; ++r) {
      const int gr = 8 * t - 1 + r;
      if (gr >= 0 && gr < p.rows_total && (gr % p.Hp) < p.H) m |= 1u << r;
    }
    return m;
  };

  f32x16 acc[6][2];
  constexpr int RING = MFC && !DROP ? B1_RING_MFC : B1_RING, AH = RING - 1;   // weight pairs are requested AH groups ahead
  bf16x8 wr[RING][2];
  bf16x8 af[2][2][2];
  float xc[12][3];                // FUSED: log-mel values around the thread's column (the tile being STAGED)
  unsigned mask_c = 0;
  f32x4 c1[10];                   // conv1 rows of the step being staged (FUSED) / raw rows, set 0 (!FUSED)
  f32x4 c1b[10];                  // !FUSED: raw rows, set 1
  f32x4 w1r[4][3];                // FUSED: this step's conv1 taps of the thread's 4 channels: w1r[e] = taps 0-3, 4-7, (8, sc, sh, -)

  // one conv1 row (FUSED): channels 16 s + 4 cq + e of row r of the tile from x[r .. r + 2][0 .. 2]
  auto conv1_row = [&](const float (&x)[12][3], unsigned mask, int r) {
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(x[r + t / 3][t % 3], w1r[e][t / 4][t % 4], a);
      y[e] = fmaxf(fmaf(a, w1r[e][2][1], w1r[e][2][2]), 0.f);
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return (mask >> r) & 1u ? y : z;
  };
  auto taps_load = [&](int s) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 3; ++k) w1r[e][k] = *(const f32x4*)(wt + ((s & 3) * 16 + cq * 4 + e) * 12 + 4 * k);
  };
  // !FUSED: raw rows of step s of tile t into set `st`
  auto rows_request = [&](int t, int s, f32x4 (&dst)[10]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const long gr = 8L * t - 1 + r;
      const unsigned off = (unsigned)(((gr * 64 + scol) * 64 + (s & 3) * 16 + cq * 4) * 4);
      dst[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    }
  };
  auto piece = [&](unsigned char* buf, const f32x4 (&d)[10], int j, int pos) {
    const f32x4 v = w4_transform(pos, d[4 * j], d[4 * j + 1], d[4 * j + 2], d[4 * j + 3], d[4 * j + 4], d[4 * j + 5]);
    u32x2 hi, lo;
    split_bf16x4(v, hi, lo);
    unsigned char* dst = buf + lofs[j] + (2 * pos) * PLANE;
    *(u32x2*)dst = hi;
    *(u32x2*)(dst + PLANE) = lo;
  };
  // staging work of MFMA group gi: planes of the NEXT step (S) into `buf`; FUSED: from x / mask; !FUSED: from raw row set d
  auto stage_unit = [&](unsigned char* buf, int gi, int S, const float (&x)[12][3], unsigned mask, f32x4 (&d)[10]) {
    // rows 0..4, two pieces, two pieces, row 5, two pieces, rows 6..8, two, two, row 9, two
    constexpr int ROW_AT[16] = {0, 1, 2, 3, 4, -1, -1, 5, -1, 6, 7, 8, -1, -1, 9, -1};
    constexpr int PIECE_AT[16] = {-1, -1, -1, -1, -1, 0, 2, -1, 4, -1, -1, -1, 6, 8, -1, 10};   // first of two pieces (j * 6 + pos)
    if (gi >= 16 || (B1_KO & 1)) return;
    if (FUSED && gi == 0) taps_load(S);
    if (FUSED && ROW_AT[gi] >= 0) d[ROW_AT[gi]] = conv1_row(x, mask, ROW_AT[gi]);
    if (PIECE_AT[gi] >= 0) {
      piece(buf, d, PIECE_AT[gi] / 6, PIECE_AT[gi] % 6);
      piece(buf, d, (PIECE_AT[gi] + 1) / 6, (PIECE_AT[gi] + 1) % 6);
    }
  };
  auto a_load = [&](const unsigned char* buf, int gi, bf16x8 (&a)[2][2]) {
    const int kx = gi / 6, q = gi % 6;
    const unsigned char* vh = buf + (2 * q) * PLANE + kx * COLP * 16;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      a[m][0] = *(const bf16x8*)(vh + pb[m]);
      a[m][1] = *(const bf16x8*)(vh + PLANE + pb[m]);
    }
  };

  // ---- FORM 2: conv1 on the matrix cores.  This wave's conv1 rows: quad j1, columns 32 m1 .. + 31; lane (i, half) ----
  const int m1 = wave & 1, j1 = wave >> 1;
  const int c1col = 32 * m1 + (lane & 31);
  unsigned lofs1[2];
#pragma unroll
  for (int gh = 0; gh < 2; ++gh) lofs1[gh] = (unsigned)(gh * HALF + ((c1col + 1) * COLP + j1) * 16 + half * 8);
  f32x4 xr[7];                    // log-mel rows 8 t - 2 + 4 j1 + 2 half + k, columns c1col - 1 .. + 2, as loaded
  u32x2 rph[7], rpl[7];           // the same rows as bf16 (columns c - 1, c | c + 1, zero), hi and lo
  f32x4 d1[4][6];                 // conv1 + BN + ReLU: channels 32 ct + 8 g + 4 half .. + 3 of rows 4 j1 - 1 + r of the tile
  f32x16 c1a[2];                  // the rows in flight (row r in c1a[r & 1]: read a whole MFMA group after its last product)
  bf16x8 a1h, a1l;
  auto x_request = [&](int t) {
    const int row0 = 8 * t - 2 + 4 * j1 + 2 * half;
    const int cadj = c1col == 0 ? 0 : c1col - 1;   // column 0: no load from in front of the buffer; shifted below
#pragma unroll
    for (int k = 0; k < 7; ++k)   // rows above / below the batch: out of range -> zeros
      xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(((row0 + k) * 64 + cadj) * 4), 0, 0));
  };
  auto x_convert = [&](int k) {   // row k of the seven
    const bool first = c1col == 0, last = c1col == 63;
    if (B1_KO & 64) {   // development: no split of the log-mel rows
      rph[k].x = __builtin_bit_cast(unsigned, xr[k][0]); rph[k].y = __builtin_bit_cast(unsigned, xr[k][1]);
      rpl[k].x = __builtin_bit_cast(unsigned, xr[k][2]); rpl[k].y = __builtin_bit_cast(unsigned, xr[k][3]);
      return;
    }
    const float e0 = first ? 0.f : xr[k][0], e1 = first ? xr[k][0] : xr[k][1];
    const float e2 = last ? 0.f : (first ? xr[k][1] : xr[k][2]);
    const unsigned h0 = cvt_pk_bf16(e0, e1), h1 = cvt_pk_bf16(e2, 0.f);
    const float f0 = __builtin_bit_cast(float, h0 << 16), f1 = __builtin_bit_cast(float, h0 & 0xffff0000u);
    const float f2 = __builtin_bit_cast(float, h1 << 16);
    rph[k].x = h0; rph[k].y = h1;
    rpl[k].x = cvt_pk_bf16(e0 - f0, e1 - f1);
    rpl[k].y = cvt_pk_bf16(e2 - f2, 0.f);
  };
  auto a1_load = [&](int ct) {
    const unsigned char* t = dsm_raw + B1_WT + (ct * 2 * 64 + lane) * 16;
    a1h = *(const bf16x8*)t;
    a1l = *(const bf16x8*)(t + 1024);
  };
  // product `which` (hi x lo, lo x hi, hi x hi) of conv1 row r: lanes 0-31 supply taps ky = 0, 1 (rows r, r + 1 of the
  // lane's seven), lanes 32-63 ky = 2 (row r of THEIR seven, two rows further down) and the constant 1 of the BN shift
  auto c1_mfma = [&](int which, int r) {
    f32x16& a = c1a[r & 1];
    if (B1_KO & 32) {   // development: no conv1 products (one cheap dependent instruction per call instead)
      if (which == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = __builtin_bit_cast(float, rph[r].x);
      } else {
        a[which] += __builtin_bit_cast(float, rpl[r + 1].y);
      }
      return;
    }
    if (which == 0) {
      const u32x4 bl = {rpl[r].x, rpl[r].y, half ? 0u : rpl[r + 1].x, half ? 0u : rpl[r + 1].y};
      f32x16 z;
#pragma unroll
      for (int k = 0; k < 16; ++k) z[k] = 0.f;
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, __builtin_bit_cast(bf16x8, bl), z, 0, 0, 0);
    } else {
      const u32x4 bh = {rph[r].x, rph[r].y, half ? 0x3f80u : rph[r + 1].x, half ? 0u : rph[r + 1].y};
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(which == 1 ? a1l : a1h, __builtin_bit_cast(bf16x8, bh), a, 0, 0, 0);
    }
  };
  auto c1_post = [&](int r, unsigned mask) {   // ReLU; rows outside the clip are the zero rows of conv1's output
    const float cap = (mask >> (4 * j1 + r)) & 1u ? __builtin_inff() : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) d1[g][r][e] = __builtin_amdgcn_fmed3f(c1a[r & 1][4 * g + e], 0.f, cap);
  };
  auto piece1 = [&](unsigned char* buf, int g, int pos) {
    const f32x4 v = w4_transform(pos, d1[g][0], d1[g][1], d1[g][2], d1[g][3], d1[g][4], d1[g][5]);
    u32x2 hi, lo;
    split_bf16x4(v, hi, lo);
    unsigned char* dst = buf + lofs1[g & 1] + (2 * pos) * PLANE;
    if (B1_KO & 8) {
      asm volatile("" :: "v"(hi), "v"(lo), "v"(dst));
      return;
    }
    *(u32x2*)dst = hi;
    *(u32x2*)(dst + PLANE) = lo;
  };
  // vector work of MFMA group gi in a step that runs conv1 (channel tile ct; groups 0-1 = the next K step into `buf`) ...
  auto stage_a = [&](unsigned char* buf, int gi, int ct, unsigned mask) {
    if (B1_KO & 1) return;
    if (gi == 0) a1_load(ct);
    if (gi == 0) {   // (spread over groups 0-2, or moved to the end of the step in front: spills / no faster)
#pragma unroll
      for (int k = 0; k < 7; ++k) x_convert(k);
    }
    if (gi >= 2 && gi <= 7) c1_post(gi - 2, mask);       // its products were issued in group gi - 1
    if (gi >= 8 && gi <= 15 && !(B1_KO & 16)) {   // twelve pieces over eight groups: 2, 1, 2, 1, ...
      const int p0 = 3 * ((gi - 8) >> 1) + 2 * (gi & 1), np = gi & 1 ? 1 : 2;
#pragma unroll
      for (int q = p0; q < p0 + np; ++q) piece1(buf, q & 1, q >> 1);
    }
  };
  // ... and in the step after it (groups 2-3 = the K step after the next)
  auto stage_b = [&](unsigned char* buf, int gi) {
    if (B1_KO & 1) return;
    // one piece per group: a group's five or six vector instructions per MFMA are what one wave per SIMD hides (two pieces
    // in every other group: +390 cycles per step)
    if (gi < 12 && !(B1_KO & 16)) piece1(buf, 2 + (gi & 1), gi >> 1);
  };

  // ---- prologue of the first tile: its inputs, step 0 staged into buffer 0 ----
  lds_barrier();   // conv1 table, zero columns
  static_assert(18 % RING == 0, "ring positions are static");
#pragma unroll
  for (int g0 = 0; g0 < AH; ++g0) w_load(0, g0, wr[g0]);
  if (FUSED) {
    load_x(tile, xc);
    mask_c = row_mask(tile);
  } else if (MFC) {
    x_request(tile);
    mask_c = row_mask(tile);
  } else {
    rows_request(tile, 0, c1);
    rows_request(tile, 1, c1b);
  }
  if (MFC) {
    a1_load(0);
#pragma unroll
    for (int k = 0; k < 7; ++k) x_convert(k);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int k = 0; k < 3; ++k) c1_mfma(k, r);
      c1_post(r, mask_c);
    }
#pragma unroll
    for (int gi = 8; gi < 16; ++gi) stage_a(sV, gi, 0, mask_c);
  } else {
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) stage_unit(sV, gi, 0, xc, mask_c, c1);
  }
  lds_barrier();
  a_load(sV, 0, af[0]);

  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)p.scale, 0, 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc((void*)p.shift, 0, 256, 0x00020000);
  const FastDiv4 by_hp_out(p.Hp / 2);
  const int chw = cg * 32 + 4 * half;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

#ifdef B1_CLK
  unsigned long long clk_loop = 0, clk_epi = 0, clk_n = 0, clk_even = 0, clk_s1 = 0;
  const unsigned long long clk_t0 = __builtin_readcyclecounter(), clk_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
  while (tile < p.tiles) {
#ifdef B1_CLK
    const unsigned long long clk_a = __builtin_readcyclecounter();
#endif
    const int next = next_live(tile);   // >= p.tiles: none (its staging below then works on zeros nobody reads)
    // ---- four K steps; step s stages step s + 1 (step 3: step 0 of the next tile) ----
    auto step = [&](auto S_) {
      constexpr int s = decltype(S_)::value;
      const unsigned char* cur = sV + (s & 1) * VBUF;
      unsigned char* nxt = sV + ((s + 1) & 1) * VBUF;
#pragma unroll
      for (int gi = 0; gi < 18; ++gi) {
        const int q = gi % 6;
        if (gi + 1 < 18) a_load(cur, gi + 1, af[(gi + 1) & 1]);
        else a_load(nxt, 0, af[0]);                      // behind the barrier of group 16
        if (gi + AH < 18) w_load(s, gi + AH, wr[(gi + AH) % RING]);
        else w_load(s + 1, gi + AH - 18, wr[(gi + AH) % RING]);
        // operand order (weights, pixels): D rows = channels, columns = pixels (16-byte stores in the epilogue); the first
        // products of a tile start from zero instead of the previous tile's sums
        if (!(B1_KO & 2)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (s == 0 && gi < 6) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % RING][0], af[gi & 1][m][1], z, 0, 0, 0);
          } else {
            acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % RING][0], af[gi & 1][m][1], acc[q][m], 0, 0, 0);
          }
        }
        // conv1 row gi - 1 rides along, one product behind each pair
        const bool C1ROW = MFC && (s & 1) && gi >= 1 && gi <= 6 && !(B1_KO & 1);
        if (C1ROW) c1_mfma(0, gi - 1);
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % RING][1], af[gi & 1][m][0], acc[q][m], 0, 0, 0);
        if (C1ROW) c1_mfma(1, gi - 1);
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % RING][0], af[gi & 1][m][0], acc[q][m], 0, 0, 0);
        if (C1ROW) c1_mfma(2, gi - 1);
        } else {
#pragma unroll
          for (int m = 0; m < 2; ++m)
            asm volatile("" :: "v"(af[gi & 1][m][0]), "v"(af[gi & 1][m][1]), "v"(wr[gi % RING][0]), "v"(wr[gi % RING][1]));
          if (s == 0 && gi < 6) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;
          }
        }
        // staging of the next step: steps 0-2 from this tile's inputs, step 3 from the next tile's
        if (MFC) {
          // steps 1 / 3 run conv1 for channel tile 1 of this tile / 0 of the next one and stage its first K step; steps
          // 2 / 0 stage the second; the log-mel rows are requested three groups before the end of the step in front
          if (s & 1) stage_a(nxt, gi, s == 1 ? 1 : 0, mask_c);
          else stage_b(nxt, gi);
          if (!(s & 1) && gi == 13) x_request(s == 0 ? tile : next);
          if (s == 0 && gi == 16) touch_x(next);
          if (s == 2 && gi == 16) mask_c = row_mask(next);
        } else if (FUSED) {
          // steps 0-2 stage this tile's next step, step 3 the first step of the next tile: its log-mel values replace this
          // tile's once step 2 has staged the last of them (the lines were touched two steps earlier: L2 hits)
          stage_unit(nxt, gi, s + 1, xc, mask_c, c1);
          if (s == 0 && gi == 16) touch_x(next);
          if (s == 2 && gi == 16) {
            load_x(next, xc);
            mask_c = row_mask(next);
          }
        } else {
          // raw rows: step k lives in set k & 1.  Set (s + 1) & 1 is staged here; set s & 1 was consumed a step ago and
          // takes the step after the next one (a whole step of latency)
          if (((s + 1) & 1) == 0) stage_unit(nxt, gi, s + 1, xc, mask_c, c1);
          else stage_unit(nxt, gi, s + 1, xc, mask_c, c1b);
          if (gi == 0) {
            const int t2 = s + 2 < 4 ? tile : next;
            if ((s & 1) == 0) rows_request(t2, s + 2, c1);
            else rows_request(t2, s + 2, c1b);
          }
        }
        if (gi == 16) lds_barrier();
        if (B1_SGB > 0 && MFC) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, B1_SGB, 0);
            __builtin_amdgcn_sched_group_barrier(B1_SGB_MEM, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    step(std::integral_constant<int, 0>{});
#ifdef B1_CLK
    const unsigned long long clk_s0 = __builtin_readcyclecounter();
    clk_even += clk_s0 - clk_a;
#endif
    step(std::integral_constant<int, 1>{});
#ifdef B1_CLK
    clk_s1 += __builtin_readcyclecounter() - clk_s0;
#endif
    step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{});

#ifdef B1_CLK
    const unsigned long long clk_b = __builtin_readcyclecounter();
#endif
    // ---- epilogue: output transform, BN, ReLU, 2x2 pool.  Lane l owns PIXEL column 32 m + l % 32 of quad wq and, in
    // register quad g, channels chw + 8 g .. + 3; the two columns of a pooling window sit in lanes l, l ^ 1.  Stores go
    // through a descriptor rebased to the tile's 32 KB of output and cut at the end of the tensor (no 64-bit addresses,
    // no bounds test per store) ----
    {
      f32x4 sc4[4], sh4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        sc4[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsc, (unsigned)(chw * 4 + 32 * g), 0, 0));
        sh4[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsh, (unsigned)(chw * 4 + 32 * g), 0, 0));
      }
      const int qg = 2 * tile + wq;
      const int hp0 = by_hp_out.mod(2 * qg);
      const long left = ((long)(p.rows_total / 2) - 4L * tile) * (32 * 64 * 4);
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.out + (size_t)tile * (4 * 32 * 64)), 0, (int)(left < 32768 ? (left < 0 ? 0 : left) : 32768), 0x00020000);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int col = 32 * m + (lane & 31);
        const bool valid = hp0 + (col & 1) < p.H / 2;
        const unsigned ob = (unsigned)((((2 * wq + (col & 1)) * 32 + (col >> 1)) * 64 + chw) * 4);   // byte offset in the tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 t0, t1, o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float y[4];
            w4_outputs(acc[0][m][r], acc[1][m][r], acc[2][m][r], acc[3][m][r], acc[4][m][r], acc[5][m][r], sc4[g][e],
                       sh4[g][e], y);
            t0[e] = y[0] + y[1];
            t1[e] = y[2] + y[3];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = t0[e] + dpp_mov<DPP_QUAD_XOR1>(t0[e]), b = t1[e] + dpp_mov<DPP_QUAD_XOR1>(t1[e]);
            o[e] = 0.25f * ((col & 1) ? b : a);
          }
          if (!valid) o = zero4;
          if (DROP) {
            const size_t oi = (size_t)tile * (4 * 32 * 64) + ob / 4 + 8 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] *= p.drop.mask(oi + e);
          }
          if (!(B1_KO & 4))
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ro, ob + 32u * g, 0, 0);
        }
      }
    }
    asm volatile("" :: "v"(touch));
#ifdef B1_CLK
    clk_loop += clk_b - clk_a;
    clk_epi += __builtin_readcyclecounter() - clk_b;
    ++clk_n;
#endif
    tile = next;
  }
#ifdef B1_CLK
  if (tid == 0) {
    atomicAdd(&b1_clk[0], clk_loop);
    atomicAdd(&b1_clk[1], clk_epi);
    atomicAdd(&b1_clk[2], __builtin_readcyclecounter() - clk_t0);
    atomicAdd(&b1_clk[3], __builtin_amdgcn_s_memrealtime() - clk_rt0);
    atomicAdd(&b1_clk[4], clk_n);
    atomicAdd(&b1_clk[5], clk_even);
    atomicAdd(&b1_clk[6], clk_s1);
  }
#endif
}

template <int FORM, bool DROP>
int launch_b1(const B1Params& p, hipStream_t s) {
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)block1_w4_kernel<FORM, DROP>, 160 * 1024, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  static int cus_of[64];   // compute units per device (one persistent workgroup each), 0 = not asked yet
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return AC_ERR_LAUNCH;
  if (dev >= 0 && dev < 64) cus = __atomic_load_n(&cus_of[dev], __ATOMIC_RELAXED);
  if (!cus) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return AC_ERR_LAUNCH;
    if (cus <= 0) cus = 256;
    if (dev >= 0 && dev < 64) __atomic_store_n(&cus_of[dev], cus, __ATOMIC_RELAXED);
  }
  const unsigned grid = (unsigned)(p.tiles < cus ? p.tiles : cus);
  hipLaunchKernelGGL((block1_w4_kernel<FORM, DROP>), dim3(grid), dim3(256), (size_t)B1_LDS, s, p);
  return ac_check_launch();
}

}  // namespace

static int b1_dispatch(int form, const float* in, const float* w1, const float* sc1, const float* sh1, const void* wfrag2,
                       const float* scale2, const float* shift2, float* out, int B, int Hp, int H, const int* clip_frames,
                       int need_mul, int need_add, void* stream, Drop drop) {
  if (!in || !wfrag2 || !scale2 || !shift2 || !out || (form && (!w1 || !sc1 || !sh1))) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || (Hp & 7) || H < 2) return AC_ERR_ARG;   // 8-row tiles inside a clip
  if (((unsigned long long)B * Hp + 4096) * 256 >= (1ull << 31)) return AC_ERR_ARG;   // 32-bit byte offsets into [rows][64] floats
  if (!form && (unsigned long long)B * Hp * 64 * 64 * 4 >= (1ull << 31)) return AC_ERR_ARG;   // unfused form: one descriptor
  B1Params p;
  p.in = in; p.w1 = w1; p.sc1 = sc1; p.sh1 = sh1; p.wpk = wfrag2; p.scale = scale2; p.shift = shift2; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H;
  p.tiles = p.rows_total / 8;
  p.clip_frames = clip_frames; p.need_mul = need_mul; p.need_add = need_add;
  p.drop = drop;
  const hipStream_t st = (hipStream_t)stream;
  if (drop.thresh != 0) return form == 2 ? launch_b1<2, true>(p, st) : form == 1 ? launch_b1<1, true>(p, st) : AC_ERR_ARG;
  return form == 2 ? launch_b1<2, false>(p, st) : form == 1 ? launch_b1<1, false>(p, st) : launch_b1<0, false>(p, st);
}

#ifdef B1_CLK
extern "C" int ac_b1_clk_read(unsigned long long* out5, int reset) {
  if (hipMemcpyFromSymbol(out5, HIP_SYMBOL(b1_clk), 56) != hipSuccess) return -2;   // out5: seven words
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(b1_clk), z, 64) != hipSuccess) return -2; }
  return 0;
}
#endif

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_conv3x3_block1_wino43(const float* in1, const float* w1, const float* scale1, const float* shift1,
                                        const void* wfrag2, const float* scale2, const float* shift2, float* out, int B,
                                        int Hp, int H, const int* clip_frames, int need_mul, int need_add, float drop_p,
                                        unsigned long long drop_seed, const unsigned long long* seed_dev, void* stream) {
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return AC_ERR_ARG;
  return b1_dispatch(1, in1, w1, scale1, shift1, wfrag2, scale2, shift2, out, B, Hp, H, clip_frames, need_mul, need_add,
                     stream, make_drop(drop_p, drop_seed, seed_dev));
}

extern "C" int ac_conv3x3_block1_wino43_mfma(const float* in1, const float* w1, const float* scale1, const float* shift1,
                                             const void* wfrag2, const float* scale2, const float* shift2, float* out, int B,
                                             int Hp, int H, const int* clip_frames, int need_mul, int need_add, float drop_p,
                                             unsigned long long drop_seed, const unsigned long long* seed_dev, void* stream) {
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return AC_ERR_ARG;
  return b1_dispatch(2, in1, w1, scale1, shift1, wfrag2, scale2, shift2, out, B, Hp, H, clip_frames, need_mul, need_add,
                     stream, make_drop(drop_p, drop_seed, seed_dev));
}

extern "C" int ac_conv3x3_block1_conv2_wino43(const float* in64, const void* wfrag2, const float* scale2, const float* shift2,
                                              float* out, int B, int Hp, int H, void* stream) {
  return b1_dispatch(0, in64, nullptr, nullptr, nullptr, wfrag2, scale2, shift2, out, B, Hp, H, nullptr, 0, 0, stream,
                     make_drop(0.f, 0, nullptr));
}
