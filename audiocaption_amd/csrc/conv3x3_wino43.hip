// Cnn14 conv stack, f32-grade tier, second generation: 3x3 convolution + eval BatchNorm + ReLU (+ 2x2 average pooling) as a
// 1-D Winograd F(4,3) along the TIME axis on split-bf16 operands, ONE 512-register wave per SIMD.
//
// Same contract, layouts and epilogue modes 0 / 1 as csrc/conv3x3_wino1d.hip (reference ConvBlock.forward,
// cnn_encoder.py:59-75; pooling glue of Cnn14Encoder.forward, cnn_encoder.py:431-444).  f32 activations [B*Hp][W][C] in and
// out, Hp % 4 == 0.
//
// Arithmetic.  For an output row QUAD (4j .. 4j+3) of one mel column w the six input rows d0..d5 = rows 4j-1 .. 4j+4 give
//     V0 = 4 d0 - 5 d2 + d4          V1 = (d4 - 4 d2) + (d3 - 4 d1)      V2 = (d4 - 4 d2) - (d3 - 4 d1)
//     V3 = (d4 - d2) + 2 (d3 - d1)   V4 = (d4 - d2) - 2 (d3 - d1)        V5 = 4 d1 - 5 d3 + d5          (f32)
//     U_p = sum_ky G[p][ky] g_ky,  G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
//     M_p[quad, w, cout] = sum_kx sum_cin V_p[quad, w + kx - 1, cin] U_p,kx[cout, cin]            (6 positions x 3 mel taps)
//     y0 = M0+M1+M2+M3+M4   y1 = (M1-M2) + 2 (M3-M4)   y2 = (M1+M2) + 4 (M3+M4)   y3 = (M1-M2) + 8 (M3-M4) + M5
// i.e. 18 products per (cin, cout) and row quad instead of 36: HALF the multiplications of the direct form (F(2,3): 2/3).
// Every product runs on split-bf16 operands (x = hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, f32
// accumulation), so a f32 product costs 3 / 2 = 1.5 bf16 MFMA products.  V is transformed in f32 BEFORE the split and U in
// f64 before the split, so the larger transform constants of F(4,3) do not amplify the 2^-17 operand error: greedy logits
// within 3e-5 of the fp32 CPU reference, the same as F(2,3) and the direct split form (tests/wino_split_emulation.py).
//
// Work decomposition.  Six accumulator sets per (pixel tile, channel tile) do not fit two waves per SIMD, so a workgroup is
// FOUR waves, one per SIMD, each with up to 512 registers: a wave owns 32 output channels x MW = 2 MFMA tiles of 32
// (quad, column) pixels x 6 positions = 192 accumulator registers, and every weight fragment pair it fetches from L2
// feeds 6 MFMAs.  (MW = 1, 96 accumulators and <= 256 registers: half-size workgroups, two to a CU - the layer whose K
// loop is shorter than a workgroup's prologue + epilogue; three tiles per wave, 288 accumulators: hipcc keeps every MFMA
// accumulator in the 256 AGPRs and shuffles the rest through VGPRs, +50 %: not kept.)  A workgroup covers
// the FULL image width (W = 32, 16, 8, 4 = blocks 2 .. 5) x PQ = MW * 32 / W quads x 128 channels; the two halo columns
// beside the image are LDS columns that stay zero.  The K loop runs in 16-channel steps over double-buffered V planes in
// LDS ([position][hi | lo][k-half][column][quad] items of 8 channels = 16 bytes, column pitch chosen so that the 16 lanes
// of every ds_read_b128 group land in 16 different bank slots).  One wave per SIMD means nothing hides a stall but the
// wave's own instruction stream, so the step is one straight-line block: the raw rows of steps s + 2 and s + 3 are
// requested together in the middle of every even step (full 128-byte lines; a step or two of latency), the rows of step
// s + 1 are transformed, split and stored a piece per MFMA group, the A fragments of a group are read two groups ahead,
// the weight fragments eight groups ahead in a register ring, and the single barrier of a step sits before its last two
// groups so that the first fragments of the next step are read under their MFMAs.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "ac_common.h"
#include "ac_drop.h"
#include "ac_wino43.h"

namespace {

#ifndef W4_KO   // development (tools/conv_bench.py): 1 no weight loads, 2 no A reads, 4 no plane stores, 8 no row loads, 32 no MFMAs
#define W4_KO 0
#endif

#ifndef W4_RING     // weight fragment ring: a group's pair is requested W4_RING - 1 groups ahead (18 % W4_RING == 0)
#define W4_RING 9
#endif
#ifndef W4_PRO2     // prologue: request the rows of the first two steps together
#define W4_PRO2 1
#endif
#ifndef W4_ADEPTH   // A fragments are read this many groups ahead (1 or 2)
#define W4_ADEPTH 2
#endif
#ifndef W4_ROWPAIR  // two tiles per wave: the rows of TWO consecutive K steps are requested together (two register sets), so that
#define W4_ROWPAIR 1 //   both 64-byte halves of a pixel's 128-byte line are asked for back to back - see `rows_request`
#endif

#ifdef W4_CLK   // development: shader-clock cycles of the prologue / K loop / epilogue, 100 MHz ticks of the whole block, count
__device__ unsigned long long w4_clk[8];
#define W4_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define W4_STAMP(var)
#endif

constexpr int KS = 16;   // input channels per K step = one MFMA k-step
constexpr int W4_OUT_TILE = 4 * 32 * 144;   // epilogue: a wave's LDS tile of 4 rows x 32 pixels x (128 + 16) bytes

struct W4Params {
  const float* in;
  const void* wpk;    // [Cin/16][18 = 3 kx x 6 p][Cout/32][2 (hi, lo)][64 lanes][8] bf16
  const float* scale;
  const float* shift;
  float* out;
  int rows_total, Hp, H, W, Cin, Cout;
  int MT, NT;
  int Hp_out, H_out, W_out;
  int map_mode;
  const int* clip_frames;   // ragged batches: see csrc/conv3x3_wino1d.hip
  int need_mul, need_add;
  Drop drop;
};

enum { MODE_FULL = 0, MODE_POOL = 1, MODE_MEANW = 2 };   // MEANW: mean over the two mel columns of block 6 (column tiles)

// block -> (row block, channel tile); block b runs on XCD b % 8 (speed only).  1: an XCD streams one weight column slab
// (NT % 8 == 0); 3: NT in {1, 2, 4, 8}: the channel tiles of a row block sit on 8 / NT ... XCDs each owning one slab, row
// blocks dealt round-robin; 4 (default for those NT): the same with a contiguous range of row blocks per XCD;
// 2: the channel tiles of a row block back to back on one XCD.  Fabric reads per launch, B = 64, FETCH_SIZE x 2 in MiB
// (profiles/r05_w4_fetch.txt): mode 3 with rows requested a step apart 604 / 928 / 500 / 792 / 378 / 722 / 346 / 672 for
// b2c1 .. b5c2, mode 4 with paired row requests 328 / 652 / 330 / 658 / 330 / 658 / 330 / 656 (input 256 / 512 / 128 / 256 /
// 64 / 128 / 32 / 64 MiB, read once by each of the NT XCD groups; weights 0.6 ... 72 MiB, once per XCD and round)
__device__ __forceinline__ bool w4_block_map(const W4Params& p, int& m_tile, int& n_tile) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, seq = bid >> 3;
  if (p.map_mode == 1) {
    n_tile = xcd + 8 * (seq / p.MT);
    m_tile = seq % p.MT;
  } else if (p.map_mode == 2) {
    n_tile = seq % p.NT;
    m_tile = (seq / p.NT) * 8 + xcd;
    if (m_tile >= p.MT) return false;
  } else if (p.map_mode == 3) {
    const int per = 8 / p.NT;
    n_tile = xcd % p.NT;
    m_tile = seq * per + xcd / p.NT;
    if (m_tile >= p.MT) return false;
  } else if (p.map_mode == 4) {   // like 3, but an XCD owns a CONTIGUOUS range of row blocks: the halo rows two neighbouring
    const int per = 8 / p.NT;     //   blocks share (2 of the 10 a block reads at W = 32) are then hits in that XCD's L2
    const int mtg = (p.MT + per - 1) / per;
    n_tile = xcd % p.NT;
    m_tile = (xcd / p.NT) * mtg + seq;
    if (seq >= mtg || m_tile >= p.MT) return false;
  } else {
    n_tile = bid % p.NT;
    m_tile = bid / p.NT;
  }
  return true;
}

template <int TC, int MW>
struct W4Geom {
  static_assert(TC == 32 || TC == 16 || TC == 8 || TC == 4 || TC == 2, "full-width blocks of 32, 16, 8, 4 or 2 mel columns");
  static_assert(MW == 1 || MW == 2, "one or two MFMA tiles per wave");
  // TC = 2 (conv block 6): COLUMN tiles - tile m is column m of 32 quads, so the taps that read the zero padding beside the
  // image (kx = 0 of column 0, kx = 2 of column 1: a third of the products) are skipped at compile time, and no halo
  // column is staged
  static constexpr bool COLT = TC == 2;
  static_assert(!COLT || MW == 2, "column tiles: one tile per column");
  static constexpr int QT = COLT ? 32 : 32 / TC;   // quads of an MFMA tile: row i = (quad i / TC, column i % TC); COLT: quad i
  static constexpr int PQ = COLT ? 32 : MW * QT;   // quads of a block
  static constexpr int LCOLS = COLT ? 2 : TC + 2;  // columns of a plane in LDS (with the two zero halo columns)
  // column pitch in 16-byte slots: the 16 lanes of a ds_read_b128 group ({0-3, 12-15, 20-27} ...) read column i % TC,
  // quad i / TC: TC = 32 any odd pitch, 16: pitch % 4 == 2, 8 and 4: pitch % 8 == 4 put them in 16 different slots; column
  // tiles read 32 consecutive quads of one column: any pitch
  static constexpr int COLP = COLT ? 32 : TC == 32 ? (PQ | 1) : TC == 16 ? (PQ % 4 == 2 ? PQ : PQ + 2) : (PQ % 8 == 4 ? PQ : PQ + 4);
  static constexpr int HALF = ((LCOLS * COLP * 16 + 127) / 128) * 128 + 64;   // bytes of one k-half of a plane (= 64 mod 128:
  static constexpr int PLANE = 2 * HALF;      //   the 8-byte stores of channels 0-7 / 8-15 of four items hit different banks)
  static constexpr int VBUF = 12 * PLANE;     // 6 positions x (hi, lo)
  static constexpr int NITEM = PQ * TC * 4;   // staging items (quad, column, channel quad) of a step: 256 (MW 2) or 128
  static_assert(NITEM == 256 || NITEM == 128, "one item per thread (MW = 1: per thread of waves 0 and 1)");
  static constexpr int NPIECE = 6;
};

// MW = 1: HALF-SIZE workgroups - 96 accumulators and <= 256 registers per wave, so TWO workgroups share a CU and one's
// prologue / epilogue (8 k + 6-14 k cycles: as long as the 4-step K loop of conv1 of block 2) runs under the other's K loop.
// A weight fragment pair then feeds 3 MFMAs instead of 6: twice the weight bytes through the L1 per product, which is why
// the long-K layers keep MW = 2 (see w4_dispatch).
template <int MODE, int TC, int MW>
__global__ __launch_bounds__(256, MW == 1 ? 2 : 1) void conv3x3_w4_kernel(W4Params p) {
  using G = W4Geom<TC, MW>;
  constexpr int QT = G::QT, PQ = G::PQ, COLP = G::COLP, HALF = G::HALF, PLANE = G::PLANE, VBUF = G::VBUF;
  constexpr int NPIECE = G::NPIECE;
  extern __shared__ __attribute__((aligned(128))) unsigned char dsm_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  int m_tile, n_tile;
  if (!w4_block_map(p, m_tile, n_tile)) return;
  W4_STAMP(clk_t0);
#ifdef W4_CLK
  const unsigned long long clk_rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long clk_t1 = clk_t0;
#endif
  const int quad0 = m_tile * PQ;
  unsigned char* sV = dsm_raw;   // two buffers of VBUF bytes

  // A fragment of tile m, tap kx, plane (p, hl): lane (i, half) reads item (column i % TC + kx, quad m QT + i / TC), k-half
  // `half`
  unsigned pb[MW];
  {
    const int i = lane & 31;
#pragma unroll
    for (int m = 0; m < MW; ++m)
      pb[m] = G::COLT ? (unsigned)(half * HALF + (m * COLP + i) * 16)    // column m; tap kx reads column m + kx - 1
                      : (unsigned)(half * HALF + ((i % TC) * COLP + m * QT + i / TC) * 16);
  }

  f32x16 acc[6][MW];
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;

  const int row0 = 4 * quad0;
  const bool live = w4_rows_live(row0, 4 * PQ, p.rows_total, p.Hp, p.H, p.clip_frames, p.need_mul, p.need_add);
  const bool all_pad = !live;
  const int nstep = p.Cin / KS;
  if (!all_pad) {
    // Input window of this block: rows 4 quad0 - 1 .. 4 (quad0 + PQ), through a buffer descriptor REBASED to the block's
    // first row (offsets stay small whatever the batch; bytes beyond the tensor, the row above the batch and lanes
    // parked on 0x80000000 read as zero)
    const int R0 = row0 > 0 ? row0 - 1 : 0;
    const int first = row0 > 0 ? 0 : 1;   // rows to subtract: relative row of (quad, r) = 4 quad + r - first
    const size_t row_elems = (size_t)p.W * p.Cin;
    const size_t left = ((size_t)p.rows_total - R0) * row_elems * 4;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + (size_t)R0 * row_elems), 0, (int)(left < 0x7fffffffull ? left : 0x7fffffffull), 0x00020000);
    const int NT32 = p.Cout >> 5;
    const unsigned g_bytes = (unsigned)NT32 * 2048u;   // one (step, kx, p) group: NT32 x (hi, lo) x 1 KiB
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wpk, 0, (int)((unsigned)nstep * 18u * g_bytes), 0x00020000);
    const unsigned wvoff = (unsigned)((n_tile * 4 + wave) * 2048 + lane * 16);
    auto w_load = [&](int s, int gi, bf16x8 (&w)[2]) {
      if (W4_KO & 1) { if (s | gi) return; }
      const unsigned soff = (unsigned)(s * 18 + gi) * g_bytes;
      w[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, soff, 0));
      w[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + 1024u, soff, 0));
    };

    // Staging items.  Item it = (channel quad it & 3, quad (it >> 2) % PQ, column (it >> 2) / PQ).  Thread t owns item t
    // whole (MW = 1: the 128 items belong to waves 0 and 1; waves 2 and 3 stage nothing).
    const unsigned row_bytes = (unsigned)row_elems * 4u;
    const bool stager = tid < G::NITEM;   // wave-uniform
    unsigned vbA, lofsA;
    {
      auto item = [&](int it, unsigned& vb, unsigned& lofs) {
        const int cq = it & 3, rest = it >> 2, quad = rest % PQ, c = rest / PQ;
        vb = (unsigned)(((4 * quad - first) * p.W + c) * p.Cin * 4 + cq * 16);
        lofs = (unsigned)((cq >> 1) * HALF + ((c + (G::COLT ? 0 : 1)) * COLP + quad) * 16 + (cq & 1) * 8);
      };
      item(stager ? tid : 0, vbA, lofsA);
    }

    auto run = [&]() {
      // PAIR (two tiles per wave): a K step is 16 channels = 64 bytes of a pixel's channel vector, HALF a 128-byte line; asked
      // for a step apart, the second half found the line gone from L1 and L2 (an XCD's 32 workgroups pull 5 MB through a 4 MB
      // L2 per step) and the fabric delivered every input line twice (tools/traffic_calib.hip: FETCH_SIZE of this pattern =
      // 2x the bytes; profiles/r05_traffic_calibration.txt).  So the rows of steps s + 2 and s + 3 are requested TOGETHER in
      // every even step s, into two register sets: the rows of step r live in set r & 1 (`rwp`).
      constexpr bool PAIR = W4_ROWPAIR && MW == 2;
      static_assert(!PAIR || W4_PRO2, "the paired form starts from the two-step prologue");
      f32x4 preA[6], rwp[PAIR ? 6 : 1];   // preA = set 0, rwp = set 1
      auto rows_request_pair = [&](int s) {   // rows of steps s (set 0) and s + 1 (set 1): the two halves of a line back to back
        if (W4_KO & 8) { if (s > 1) return; }
        const unsigned cs = (unsigned)(s * KS * 4);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          preA[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vbA + cs + (unsigned)r * row_bytes, 0, 0));
          rwp[PAIR ? r : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vbA + cs + (unsigned)(KS * 4) + (unsigned)r * row_bytes, 0, 0));
        }
      };
      auto rows_request = [&](int s) {
        if (W4_KO & 8) { if (s) return; }
        if (!stager) return;
        const unsigned cs = (unsigned)(s * KS * 4);
#pragma unroll
        for (int r = 0; r < 6; ++r)
          preA[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vbA + cs + (unsigned)r * row_bytes, 0, 0));
      };
      auto store_piece = [&](unsigned char* buf, unsigned lofs, int pos, const f32x4 v) {
        u32x2 hi, lo;
        split_bf16x4(v, hi, lo);
        unsigned char* dst = buf + lofs + (2 * pos) * PLANE;
        *(u32x2*)dst = hi;
        *(u32x2*)(dst + PLANE) = lo;
      };
      // piece k of a step: position k of the thread's item
      auto commit_piece_set1 = [&](unsigned char* buf, int k) {   // PAIR: piece k of the rows in set 1
        if (W4_KO & 4) return;
        store_piece(buf, lofsA, k, w4_transform(k, rwp[0], rwp[PAIR ? 1 : 0], rwp[PAIR ? 2 : 0], rwp[PAIR ? 3 : 0],
                                                rwp[PAIR ? 4 : 0], rwp[PAIR ? 5 : 0]));
      };
      auto commit_piece = [&](unsigned char* buf, int k) {
        if (W4_KO & 4) return;
        if (!stager) return;
        store_piece(buf, lofsA, k, w4_transform(k, preA[0], preA[1], preA[2], preA[3], preA[4], preA[5]));
      };
      // column tiles: tap kx of tile (= column) m reads column m + kx - 1, which exists for (m, kx) in {(0,1), (0,2), (1,0), (1,1)}
      auto tap_valid = [](int m, int kx) { return !G::COLT || (m + kx - 1 >= 0 && m + kx - 1 <= 1); };
      auto a_load = [&](const unsigned char* buf, int gi, bf16x8 (&a)[MW][2]) {
        if (W4_KO & 2) { if (gi) return; }
        const int kx = gi / 6, q = gi % 6;
        const unsigned char* vh = buf + (2 * q) * PLANE;
        const int koff = (G::COLT ? kx - 1 : kx) * COLP * 16;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
          if (!tap_valid(m, kx)) continue;
          a[m][0] = *(const bf16x8*)(vh + ((int)pb[m] + koff));
          a[m][1] = *(const bf16x8*)(vh + PLANE + ((int)pb[m] + koff));
        }
      };

      // ---- prologue: zero halo columns of both buffers, step 0 into buffer 0, rows of step 1 requested ----
      {
        constexpr int NZ = G::COLT ? 0 : 2 * 12 * 2 * 2 * COLP;   // 16-byte slots: buffers x planes x k-halves x 2 columns x COLP
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < NZ; i += 256) {
          const int slot = i % COLP, col = (i / COLP) & 1, hh = (i / (2 * COLP)) % 24, b = i / (48 * COLP);
          *(f32x4*)(sV + b * VBUF + hh * HALF + ((col ? TC + 1 : 0) * COLP + slot) * 16) = z;
        }
      }
      constexpr int RING = MW == 1 ? 6 : W4_RING, AH = RING - 1, NA = W4_ADEPTH + 1, LASTG = 17 - W4_ADEPTH;   // MW = 1: 256 registers
      static_assert(18 % RING == 0 && 18 % NA == 0 && NPIECE <= LASTG, "ring positions are static; staging ends before the barrier");
      bf16x8 wr[RING][2];   // ring of weight fragments: group gi lives in wr[gi % RING], requested RING - 1 groups ahead
#if W4_PRO2
      // the rows of steps 0 AND 1 are requested back to back before anything else (one exposed HBM round trip instead of
      // two: step 0's staging below waits for the first set, step 0 of the K loop finds the second one landed), the
      // weight fragments behind them (L2 hits, and the vector-memory counter retires in order)
      f32x4 pre2[6];
      if (PAIR) rows_request_pair(0);
      else rows_request(0);
      if (MW == 2 && !PAIR) {
        const unsigned cs = (unsigned)(KS * 4);
#pragma unroll
        for (int r = 0; r < 6; ++r)
          pre2[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, vbA + cs + (unsigned)r * row_bytes, 0, 0));
      }
#pragma unroll
      for (int g0 = 0; g0 < AH; ++g0) w_load(0, g0, wr[g0]);
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) commit_piece(sV, k);
      if (PAIR) {
        // step 0 stages set 1 (rows of step 1) and requests the rows of steps 2 and 3
      } else if (MW == 2) {
#pragma unroll
        for (int r = 0; r < 6; ++r) preA[r] = pre2[r];
      } else {
        rows_request(1);
      }
#else
#pragma unroll
      for (int g0 = 0; g0 < AH; ++g0) w_load(0, g0, wr[g0]);
      rows_request(0);
#pragma unroll
      for (int k = 0; k < NPIECE; ++k) commit_piece(sV, k);
      rows_request(1);   // nstep >= 2
#endif
      lds_barrier();     // LDS only: the rows just requested stay in flight
      bf16x8 af[NA][MW][2];   // A fragments (tile, hi | lo) of the current group and the next W4_ADEPTH ones
#pragma unroll
      for (int g0 = 0; g0 < W4_ADEPTH; ++g0) a_load(sV, g0, af[g0]);

      // ---- one K step.  MORE: a step follows (its planes are staged here; the one after it is requested) ----
      auto step = [&](int s, auto MORE_, auto ODD_) {   // ODD (PAIR only): s & 1 as a compile-time value
        constexpr bool MORE = decltype(MORE_)::value;
        constexpr bool ODD = decltype(ODD_)::value;
        const unsigned char* cur = sV + (s & 1) * VBUF;
        unsigned char* nxt = sV + ((s + 1) & 1) * VBUF;
#pragma unroll
        for (int gi = 0; gi < 18; ++gi) {
          const int q = gi % 6;
          if (gi + W4_ADEPTH < 18) a_load(cur, gi + W4_ADEPTH, af[(gi + W4_ADEPTH) % NA]);
          else if (MORE) a_load(nxt, gi + W4_ADEPTH - 18, af[(gi + W4_ADEPTH) % NA]);   // behind the barrier of group LASTG
          if (gi + AH < 18) w_load(s, gi + AH, wr[(gi + AH) % RING]);
          else if (MORE) w_load(s + 1, gi + AH - 18, wr[(gi + AH) % RING]);
          if (!(W4_KO & 32)) {
            // operand order (weights, pixels): D rows = channels, columns = pixels, so that a lane ends up with four
            // CONSECUTIVE channels of one pixel per register quad (16-byte stores in the epilogue)
            const int kxg = gi / 6;
#pragma unroll
            for (int m = 0; m < MW; ++m)
              if (tap_valid(m, kxg))
                acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % RING][0], af[gi % NA][m][1], acc[q][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MW; ++m)
              if (tap_valid(m, kxg))
                acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % RING][1], af[gi % NA][m][0], acc[q][m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MW; ++m)
              if (tap_valid(m, kxg))
                acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[gi % RING][0], af[gi % NA][m][0], acc[q][m], 0, 0, 0);
          } else {
#pragma unroll
            for (int m = 0; m < MW; ++m)
              if (tap_valid(m, gi / 6))
                asm volatile("" :: "v"(af[gi % NA][m][0]), "v"(af[gi % NA][m][1]), "v"(wr[gi % RING][0]), "v"(wr[gi % RING][1]));
          }
          if (MORE) {
            if (PAIR) {   // even steps stage set 1 (rows of the odd step that follows), then ask for the next two steps' rows
              if (gi < NPIECE) { if (ODD) commit_piece(nxt, gi); else commit_piece_set1(nxt, gi); }
              if (gi == NPIECE && !ODD) rows_request_pair(s + 2);   // past the last step: lands in registers nobody reads
            } else {
              if (gi < NPIECE) commit_piece(nxt, gi);
              if (gi == NPIECE) rows_request(s + 2);   // past the last step: lands in registers nobody reads
            }
          }
          if (gi == LASTG && !(W4_KO & 16)) lds_barrier();   // step s + 1 is complete in `nxt`; the fragments of the groups left are in registers
          __builtin_amdgcn_sched_barrier(0);
        }
      };
#ifdef W4_CLK
      clk_t1 = __builtin_readcyclecounter();
#endif
      if (PAIR) {   // nstep is even (Cin % 32 == 0: checked by the dispatcher)
#pragma unroll 1
        for (int s = 0; s + 2 < nstep; s += 2) {
          step(s, std::true_type{}, std::false_type{});
          step(s + 1, std::true_type{}, std::true_type{});
        }
        step(nstep - 2, std::true_type{}, std::false_type{});
        step(nstep - 1, std::false_type{}, std::true_type{});
      } else {
#pragma unroll 1
        for (int s = 0; s + 1 < nstep; ++s) step(s, std::true_type{}, std::false_type{});
        step(nstep - 1, std::false_type{}, std::false_type{});
      }
    };
    run();
  }

  W4_STAMP(clk_t2);
  // ---- epilogue: output transform, BN, ReLU, pooling, zero rows.  Lane l owns PIXEL l % 32 of each tile (column i % TC,
  // quad i / TC) and, in register quad g, the four consecutive channels 8 g + 4 (l / 32) .. + 3 of the wave's 32: one
  // 16-byte store per (tile, g, row).  The two columns of a pooling window sit in lanes l, l ^ 1. ----
  const int chw = n_tile * 128 + wave * 32 + 4 * half;
  const FastDiv4 by_hp(p.Hp), by_hp_out(MODE == MODE_POOL ? p.Hp_out : 1);
  const int pi = lane & 31;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 sc4[4], sh4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    sc4[g] = all_pad ? zero4 : *(const f32x4*)(p.scale + chw + 8 * g);   // dead blocks store zeros
    sh4[g] = all_pad ? zero4 : *(const f32x4*)(p.shift + chw + 8 * g);
  }
  if (MODE == MODE_MEANW) {
    // conv block 6, second conv: mean over the two mel columns = the wave's two (column) tiles, same lane, same register;
    // out (B, H, Cout) dense (cnn_encoder.py:443: torch.mean(x, dim=3))
    static_assert(MODE != MODE_MEANW || G::COLT, "the mean over mel is the two-column form's epilogue");
    const int qg = quad0 + pi, gr = 4 * qg;
    const int bclip = gr / p.Hp, h = gr - bclip * p.Hp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 mean[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float ya[4], yb[4];
        w4_outputs(acc[0][0][r], acc[1][0][r], acc[2][0][r], acc[3][0][r], acc[4][0][r], acc[5][0][r], sc4[g][e], sh4[g][e], ya);
        w4_outputs(acc[0][1][r], acc[1][1][r], acc[2][1][r], acc[3][1][r], acc[4][1][r], acc[5][1][r], sc4[g][e], sh4[g][e], yb);
#pragma unroll
        for (int j = 0; j < 4; ++j) mean[j][e] = 0.5f * (ya[j] + yb[j]);
      }
      if (gr < p.rows_total) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (h + j < p.H && !(W4_KO & 64)) *(f32x4*)(p.out + ((size_t)bclip * p.H + h + j) * p.Cout + chw + 8 * g) = mean[j];
      }
    }
  } else {
#pragma unroll
  for (int m = 0; m < MW; ++m) {
    const int col = G::COLT ? m : pi % TC;
    const int qg = quad0 + (G::COLT ? pi : m * QT + pi / TC);
    const int gr = 4 * qg;
    f32x4 y[4][4];   // y[g][j][e]: channel quad g, output row j, channel chw + 8 g + e
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        float yy[4];
        w4_outputs(acc[0][m][r], acc[1][m][r], acc[2][m][r], acc[3][m][r], acc[4][m][r], acc[5][m][r], sc4[g][e], sh4[g][e], yy);
#pragma unroll
        for (int j = 0; j < 4; ++j) y[g][j][e] = yy[j];
      }
    // The values go to memory through a wave-private LDS tile ([row][pixel], 144-byte pixel pitch): lane (pixel, half)
    // writes its channel quads, lane (P = l / 8, q = l % 8) reads quad q of pixel 8 k + P back, so that a store instruction
    // writes the wave's 128 contiguous bytes of 8 pixels.  Straight from the accumulator layout every lane's 16 bytes were
    // a memory request of their own (a lane's neighbour in memory sits 32 lanes away): 64 requests per instruction, and a
    // conv1 layer's 128 KB per workgroup took 9 k cycles to drain (tools/w4_variants.py, EXPERIMENTS.md).
    unsigned char* wt = dsm_raw + wave * W4_OUT_TILE;
    constexpr int PITCH = 144;
    const int P8 = lane >> 3, q8 = lane & 7;
    if (MODE == MODE_FULL) {
      const int h = by_hp.mod(gr);   // Hp % 4 == 0: the four rows of a quad belong to one clip
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(wt + (j * 32 + pi) * PITCH + (2 * g + half) * 16) = h + j < p.H ? y[g][j] : zero4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = 8 * k + P8;
        const int pcol = G::COLT ? m : px % TC;
        const int pgr = 4 * (quad0 + (G::COLT ? px : m * QT + px / TC));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v = *(const f32x4*)(wt + (j * 32 + px) * PITCH + q8 * 16);
          const size_t oi = ((size_t)(pgr + j) * p.W + pcol) * p.Cout + n_tile * 128 + wave * 32 + 4 * q8;
          if (p.drop.thresh != 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= p.drop.mask(oi + e);
          }
          if (pgr < p.rows_total && !(W4_KO & 64)) *(f32x4*)(p.out + oi) = v;
        }
      }
    } else {   // a row quad is two pooled rows: the even lane of a column pair holds the first, the odd lane the second
      const bool valid = by_hp_out.mod(2 * qg) + (col & 1) < p.H_out;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t0 = y[g][0] + y[g][1], t1 = y[g][2] + y[g][3];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = t0[e] + dpp_mov<DPP_QUAD_XOR1>(t0[e]), b = t1[e] + dpp_mov<DPP_QUAD_XOR1>(t1[e]);
          o[e] = 0.25f * ((col & 1) ? b : a);
        }
        *(f32x4*)(wt + pi * PITCH + (2 * g + half) * 16) = valid ? o : zero4;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = 8 * k + P8;
        const int pcol = px % TC, pqg = quad0 + m * QT + px / TC;
        f32x4 o = *(const f32x4*)(wt + px * PITCH + q8 * 16);
        const size_t oi = ((size_t)(2 * pqg + (pcol & 1)) * p.W_out + (pcol >> 1)) * p.Cout + n_tile * 128 + wave * 32 + 4 * q8;
        if (p.drop.thresh != 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= p.drop.mask(oi + e);
        }
        if (4 * pqg < p.rows_total && !(W4_KO & 64)) *(f32x4*)(p.out + oi) = o;
      }
    }
  }
  }
#ifdef W4_CLK
  __builtin_amdgcn_s_waitcnt(0);
  if (tid == 0) {
    const unsigned long long t3 = __builtin_readcyclecounter();
    atomicAdd(&w4_clk[0], clk_t1 - clk_t0);
    atomicAdd(&w4_clk[1], clk_t2 - clk_t1);
    atomicAdd(&w4_clk[2], t3 - clk_t2);
    atomicAdd(&w4_clk[3], __builtin_amdgcn_s_memrealtime() - clk_rt0);
    atomicAdd(&w4_clk[4], 1ull);
  }
#endif
}

template <int MODE, int TC, int MW>
int launch_w4(W4Params p, hipStream_t s) {
  using G = W4Geom<TC, MW>;
  const int quads = p.rows_total / 4;
  p.MT = (quads + G::PQ - 1) / G::PQ;
  unsigned grid;
  if (p.map_mode == 2) grid = (unsigned)(((p.MT + 7) / 8) * 8 * p.NT);
  else if (p.map_mode == 3 || p.map_mode == 4) grid = (unsigned)(((p.MT + 8 / p.NT - 1) / (8 / p.NT)) * 8);
  else grid = (unsigned)(p.MT * p.NT);
  constexpr size_t lds = (size_t)2 * G::VBUF > (size_t)4 * W4_OUT_TILE ? (size_t)2 * G::VBUF : (size_t)4 * W4_OUT_TILE;
  static_assert(lds <= 160 * 1024, "V planes exceed the LDS");
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)conv3x3_w4_kernel<MODE, TC, MW>, 160 * 1024, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_w4_kernel<MODE, TC, MW>), dim3(grid), dim3(256), lds, s, p);
  return ac_check_launch();
}

template <int MODE, int TC>
int launch_w4_mw(const W4Params& p, int mw, hipStream_t s) {
  if (mw == 1) return launch_w4<MODE, TC, 1>(p, s);
  return launch_w4<MODE, TC, 2>(p, s);
}

}  // namespace

// workgroups of a launch with MW tiles per wave
static long w4_grid(int rows_total, int W, int Cout, int mw) {
  const int pq = W == 2 ? 32 : mw * (32 / W);
  return (long)((rows_total / 4 + pq - 1) / pq) * (Cout / 128);
}

static int w4_dispatch(const float* in, const void* wfrag, const float* scale, const float* shift, float* out, int B, int Hp,
                       int H, int W, int Cin, int Cout, int mode, int map_mode, int mw, const int* clip_frames, int need_mul,
                       int need_add, void* stream, Drop drop) {
  if (!in || !wfrag || !scale || !shift || !out) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || (Hp & 3) || !(W == 32 || W == 16 || W == 8 || W == 4 || W == 2) || Cin % 32 || Cin < 32 || Cout % 128)
    return AC_ERR_ARG;   // (Cin % 32: the K loop walks its 16-channel steps in pairs)
  if (W == 2 ? (mode != MODE_FULL && mode != MODE_MEANW) : (mode != MODE_FULL && mode != MODE_POOL)) return AC_ERR_ARG;
  if (mode == MODE_MEANW && drop.thresh != 0) return AC_ERR_ARG;   // dropout sits BEFORE the mean over mel: use mode 0
  if ((unsigned long long)B * Hp >= (1ull << 23)) return AC_ERR_ARG;          // the epilogue finds a row's clip offset with FastDiv4::mod (float reciprocal: exact below 2^23 rows); callers chunk clips beyond it
  if ((unsigned long long)(Cin / 16) * 18 * (Cout / 32) * 2048 >= (1ull << 31)) return AC_ERR_ARG;   // packed weights: one descriptor
  W4Params p;
  p.in = in; p.wpk = wfrag; p.scale = scale; p.shift = shift; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.MT = 0;
  p.NT = Cout / 128;
  p.Hp_out = Hp / 2; p.H_out = H / 2; p.W_out = W / 2;
#ifndef W4_MW1_STEPS   // layers of up to this many 16-channel K steps run as half-size workgroups (tiles_per_wave = 0).
#define W4_MW1_STEPS 4 // Measured at B = 64 (profiles/r05_w4_kloop_experiments.txt): conv1 of block 2 (4 steps) 374 -> 324-340 us;
#endif                 // 8 steps (b2c2, b3c1) +5 %, 16 (b3c2, b4c1) +13 %, longer +20 %: twice the weight bytes per product through the L1
#ifndef W4_DEFAULT_MAP
#define W4_DEFAULT_MAP 4
#endif
  if (map_mode < 0) map_mode = (p.NT % 8 == 0 && p.NT > 8) ? 1 : ((p.NT == 1 || p.NT == 2 || p.NT == 4 || p.NT == 8) ? W4_DEFAULT_MAP : 2);
  if (map_mode == 1 && p.NT % 8 != 0) return AC_ERR_ARG;
  if ((map_mode == 3 || map_mode == 4) && !(p.NT == 1 || p.NT == 2 || p.NT == 4 || p.NT == 8)) return AC_ERR_ARG;
  if (map_mode > 4) return AC_ERR_ARG;
  p.map_mode = map_mode;
  p.clip_frames = clip_frames; p.need_mul = need_mul; p.need_add = need_add;
  p.drop = drop;
  if (mw != 1 && mw != 2) {
    // half-size workgroups (two per CU) where a workgroup's prologue + epilogue is a large part of its life: short K loops
    static const int mw1_max_steps = getenv("AUDIOCAPTION_W4_MW1_STEPS") ? atoi(getenv("AUDIOCAPTION_W4_MW1_STEPS")) : W4_MW1_STEPS;
    mw = (W != 2 && Cin / KS <= mw1_max_steps) ? 1 : 2;
  }
  if (W == 2) mw = 2;   // column tiles
  hipStream_t s = (hipStream_t)stream;
#define W4_CASE(TCV)                                                                     \
  if (W == TCV) return mode == MODE_FULL ? launch_w4_mw<MODE_FULL, TCV>(p, mw, s) : launch_w4_mw<MODE_POOL, TCV>(p, mw, s);
  W4_CASE(32) W4_CASE(16) W4_CASE(8) W4_CASE(4)
#undef W4_CASE
  if (W == 2) return mode == MODE_FULL ? launch_w4<MODE_FULL, 2, 2>(p, s) : launch_w4<MODE_MEANW, 2, 2>(p, s);
  return AC_ERR_ARG;
}

#ifdef W4_CLK
extern "C" int ac_w4_clk_read(unsigned long long* out5, int reset) {
  if (hipMemcpyFromSymbol(out5, HIP_SYMBOL(w4_clk), 40) != hipSuccess) return -2;
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(w4_clk), z, 64) != hipSuccess) return -2; }
  return 0;
}
#endif

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_conv3x3_bn_relu_wino43(const float* in, const void* wfrag, const float* scale, const float* shift,
                                         float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                         int map_mode, int tiles_per_wave, const int* clip_frames, int need_mul,
                                         int need_add, void* stream) {
  return w4_dispatch(in, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, tiles_per_wave, clip_frames,
                     need_mul, need_add, stream, make_drop(0.f, 0, nullptr));
}

extern "C" int ac_conv3x3_bn_relu_wino43_drop(const float* in, const void* wfrag, const float* scale, const float* shift,
                                              float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                              int map_mode, float drop_p, unsigned long long drop_seed,
                                              const unsigned long long* seed_dev, void* stream) {
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return AC_ERR_ARG;
  return w4_dispatch(in, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, 0, nullptr, 0, 0, stream,
                     make_drop(drop_p, drop_seed, seed_dev));
}

extern "C" long ac_conv3x3_wino43_workgroups(int B, int Hp, int W, int Cout) {
  if (B <= 0 || Hp <= 0 || (Hp & 3) || !(W == 32 || W == 16 || W == 8 || W == 4 || W == 2) || Cout % 128) return 0;
  return w4_grid(B * Hp, W, Cout, 2);
}
