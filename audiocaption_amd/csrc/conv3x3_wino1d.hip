// Cnn14 conv stack, f32-grade tier with FEWER matrix instructions: 3x3 convolution + eval BatchNorm + ReLU (+ 2x2
// average pooling / mean over the 2 mel columns) as a 1-D Winograd F(2,3) along the TIME axis on split-bf16 operands.
//
// Same contract, layouts and epilogue modes as csrc/conv3x3.hip (reference ConvBlock.forward, cnn_encoder.py:59-75;
// pooling / mean glue of Cnn14Encoder.forward, cnn_encoder.py:431-444).  f32 activations [B*Hp][W][C] in and out.
//
// Arithmetic.  For an output row pair (2j, 2j+1) of one mel column w the four input rows d0..d3 = rows 2j-1 .. 2j+2 give
//     V0 = d0 - d2    V1 = d1 + d2    V2 = d2 - d1    V3 = d1 - d3                       (input transform, +-1 only, f32)
//     U0 = g0         U1 = (g0 + g1 + g2) / 2         U2 = (g0 - g1 + g2) / 2    U3 = g2  (rows ky of the filter, f64 offline)
//     M_p[pair, w, cout] = sum_kx sum_cin V_p[pair, w + kx - 1, cin] U_p,kx[cout, cin]    (4 positions x 3 mel taps)
//     y(2j) = M0 + M1 + M2        y(2j+1) = M1 - M2 - M3                                  (output transform, epilogue)
// i.e. 12 products per (cin, cout) and row pair instead of 18: 1.5x fewer multiplications than the direct form.  Every
// product runs on split-bf16 operands like the "bf16x3" tier (x = hi + lo, hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_bf16, f32 accumulation: 2^-16 relative operand error), so a f32 product costs 3 / 1.5 = TWO bf16
// MFMA products - what the fp16-activation tier pays - while the activations stay f32 in HBM and the result stays
// f32-grade (greedy logits within 3e-5 of the fp32 CPU reference: tests/wino_split_emulation.py, tests/test_gpu_model.py).
// The 2-D form F(2x2,3x3) (1.33 products) was priced and dropped: its 16 accumulator sets per output tile cap a CU at
// 64 x 64 (tile, channel) outputs in flight, and at that size the transformed operands (4 KB of fresh fragments per
// three MFMAs) exceed what LDS and the L1 can deliver to the matrix cores.
//
// Work decomposition.  Every wave owns 32 output channels x two MFMA tiles of 32 output-row pairs x 4 positions = 128
// accumulator registers; two waves per SIMD.  The K loop runs in 16-channel steps over a DOUBLE-BUFFERED set of V planes
// in LDS: while the matrix cores consume step s, the raw rows of step s + 1 (requested into registers at the start of
// step s) are transformed, split and stored into the other buffer, piecewise between the MFMA groups; one workgroup
// barrier per step.  The 12 (kx, p) weight fragments of a step come straight from L2 in MFMA fragment order
// (pre-transformed, pre-split), requested two groups ahead; the A fragments of a group one group ahead.
//   * W = 2 / W = 4 (blocks 5-6: long K, few pixels): 512-thread workgroups of 256 pixels (64 pairs x 2 columns, 32 pairs
//     x 4 columns) x 128 channels; a tile is 32 pairs of ONE column, so the kx taps that read the zero padding beside the
//     image are skipped (a third / a sixth of the work) and the outer halo columns are not staged.
//   * W >= 8 (blocks 2-4: short K, many pixels): 256-thread workgroups of 128 pixels (16 pairs x 4 columns) x 128
//     channels, TWO per CU, so that one's prologue / epilogue (a third of a workgroup's life at K = 64...256) runs under
//     the other's MFMAs; a tile is 16 pairs x 2 columns.
//   * Cout = 64 (conv2 of block 1, W = 64): the same 256-thread form over 8 pairs x 16 columns x 64 channels - waves =
//     2 channel groups x 2 column halves, a tile is 8 pairs x 4 columns (18 x 18 staged positions for 16 x 16 outputs:
//     the smallest halo of all forms; 78 KB of planes, two workgroups per CU).
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "ac_common.h"
#include "ac_drop.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Development only (tools/w1_knockout.py): -DW1_KO=<bits> removes one ingredient of the K loop at a time to price it
// (results are then wrong).  1 weight loads, 2 A-fragment reads, 4 plane stores, 8 patch loads, 16 barrier, 32 MFMAs.
// 64 / 128 / 256: the weight loads / A-fragment reads / patch loads of the K loop are still ISSUED (same addresses, same
// traffic) but into scratch registers nothing waits for, the MFMAs and the transform run on the prologue's registers:
// prices the waiting for operands separately from the moving of them.  512 / 1024: the weight requests keep their number
// and size but all hit the same 16 KB (L1) / the same 0.8 MB (L2): prices where the weights come from.
#ifndef W1_KO
#define W1_KO 0
#endif
// Tuning switches of the same tool: raw-row register sets of the wide form, depth of the weight ring (groups ahead + 1),
// MFMA / VALU interleaving hints.
#ifndef W1_NSET
#define W1_NSET 1
#endif
#ifndef W1_RING
#define W1_RING 3
#endif
#ifndef W1_SGB
#define W1_SGB 0
#endif
#ifndef W1_PRIO
#define W1_PRIO 0
#endif

#ifdef W1_CLK   // development: shader-clock cycles and 100 MHz ticks spent between kernel entry and the end of the K loop
__device__ unsigned long long w1_clk[4];
#endif

constexpr int KS = 16;          // input channels per K step (one MFMA k-step); the packed weights come in chunks of 32
constexpr int BROW = KS;        // bf16 elements per (pair, column) item in LDS: 32 bytes = 2 bank slots

struct W1Params {
  const float* in;
  const void* wpk;    // [Cin/32][3 kx][4 p][2 k-steps][Cout/32][2 (hi, lo)][64 lanes][8] bf16
  const float* scale;
  const float* shift;
  float* out;
  int rows_total, Hp, H, W, Cin, Cout;
  int mt_cols, MT, NT;
  int Hp_out, H_out, W_out;
  int map_mode;
  // ragged batches: clip b needs only its first need_mul * clip_frames[b] + need_add output rows of this layer (the rows
  // beyond can never reach an output frame the temporal encoder reads); null = every row of the geometry
  const int* clip_frames;
  int need_mul, need_add;
  // few-workgroup launches (single clips): blockIdx.y walks slices of `ksteps` K steps each and stores its transformed,
  // not yet normalised sums to partial[slice][row][col][Cout]; w1_finish_kernel adds the slices in order and applies the
  // epilogue.  partial == nullptr: one slice, epilogue in this kernel.
  float* partial;
  int ksteps;
  // train-mode forward of the frozen network (F.dropout after every conv block, cnn_encoder.py:431-442): the mask of the
  // output element with linear index i of the output buffer is the counter hash of csrc/ac_drop.h - what ac_dropout over
  // that buffer would apply; thresh 0 = off
  Drop drop;
};

enum { MODE_FULL = 0, MODE_POOL = 1, MODE_MEANW = 2 };

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

// x mod d / x div d for 0 <= x < 2^23 with one reciprocal per wave (an integer division by a run-time value is ~25
// instructions; the epilogue needs the row inside the clip for every stored row)
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

// block -> (m_tile, n_tile); block b runs on XCD b % 8 (speed only).  map_mode 1: one XCD streams one weight column
// slab (weight-heavy layers, NT % 8 == 0); 2: the channel tiles of a pixel tile share an L2; 3: NT in {1, 2, 4}: every
// XCD owns one weight slab; 4: row blocks (below)
__device__ __forceinline__ bool block_map(const W1Params& p, int& m_tile, int& n_tile) {
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
  } else if (p.map_mode == 4) {
    // every column block and channel tile of one ROW block back to back on one XCD: the row block's input rows (with
    // the column halos its blocks share) reach that L2 once
    const int G = p.mt_cols * p.NT;
    const int within = seq % G;
    n_tile = within % p.NT;
    m_tile = ((seq / G) * 8 + xcd) * p.mt_cols + within / p.NT;
    if (m_tile >= p.MT) return false;
  } else {
    n_tile = bid % p.NT;
    m_tile = bid / p.NT;
  }
  return true;
}

// Dead block: every one of its `nrows` rows from `row0` lies in the padding of its clip(s) - beyond the geometry's H valid
// rows, or (ragged batches) beyond the rows the clip's own length can bring to an output frame.  Such a block convolves
// nothing and stores zeros.
__device__ __forceinline__ bool w1_block_live(const W1Params& p, int row0, int nrows) {
  bool live = false;
  if (row0 < p.rows_total) {
    const int r_end = row0 + nrows < p.rows_total ? row0 + nrows : p.rows_total;
    int b = row0 / p.Hp;
    for (int base = b * p.Hp; base < r_end; base += p.Hp, ++b) {
      const int lo = (row0 > base ? row0 : base) - base;
      int lim = p.H;
      if (p.clip_frames) {
        const int need = p.need_mul * p.clip_frames[b] + p.need_add;
        lim = need < lim ? need : lim;
      }
      live = live || lo < lim;
    }
  }
  return live;
}

// TC: columns of the block (4, or 2 for the 2-column layers).  FULLW: the block spans the whole image width (W == TC):
// its two outer halo columns are never read (every tap that would is skipped), so they are not staged.  WIDE: the
// 256-thread / 128-pixel form for W >= 8 (TC = 4, not FULLW).
//
// V planes in LDS, one per (position, hi | lo), items of 16 channels = 32 bytes = two 16-byte bank slots.
//   !WIDE: [pair][staged column], pair pitch an odd number of slots;  WIDE: [staged column][pair], column pitch an odd
//   number of slots - either way the 16 lanes of every ds_read_b128 group (16 consecutive pairs of one column, or 8
//   pairs of one column + 8 of the next) land in 16 different slots.
// CW: output channels per workgroup.  256 (wide form only, 512 threads = eight channel groups over the SAME 128 pixels, one
// workgroup per CU): the row loads, the transform / split and the LDS stores of a pixel tile then feed twice the MFMAs -
// that staging chain, not operand traffic, is what the kernel loses time to (tools/w1_mix_probe.hip: +9 ... 14 %).
template <int TC, bool FULLW, bool WIDE, int CW = 128>
struct W1Geom {
  static_assert(!WIDE || ((TC == 4 || TC == 16) && !FULLW), "the wide forms are 4 or 16 columns with halo");
  static constexpr bool C64 = TC == 16;                    // the 64-channel form (conv2 of block 1): see the kernel
  static_assert(CW == 128 || (CW == 256 && WIDE && TC == 4), "256-channel workgroups: the 4-column wide form only");
  static constexpr int THREADS = WIDE ? (CW == 256 ? 512 : 256) : 512;
  static constexpr int PR = WIDE ? (C64 ? 8 : 16) : 128 / TC;   // pair rows per block
  static constexpr int COFF = FULLW ? 1 : 0;
  static constexpr int PWS = TC + 2 - 2 * COFF;            // staged columns: image columns col0 - 1 + COFF ..
  static constexpr int PAIR_PITCH = WIDE ? BROW : PWS * BROW + 8;
  static constexpr int COL_PITCH = WIDE ? PR * BROW + 8 : BROW;
  static constexpr int PLANE = WIDE ? PWS * COL_PITCH : PR * PAIR_PITCH;   // one (position, hi | lo) plane
  static constexpr int VBUF = 8 * PLANE;
  // staging item = NP vertically adjacent pairs x one staged column x one channel quad: 2 NP + 2 input rows
  static constexpr int NP = FULLW ? 1 : (C64 ? 4 : 2);
  static constexpr int NROW = 2 * NP + 2;
  static constexpr int NITEM = (PR / NP) * PWS * (KS / 4);
  static_assert(NITEM <= THREADS, "one staging item per thread");
  static constexpr int NSET = WIDE ? W1_NSET : 1;                // register sets of raw rows in flight (see the kernel)
};

template <int MODE, int TC, bool FULLW, bool WIDE, int CW = 128>
__global__ __launch_bounds__((W1Geom<TC, FULLW, WIDE, CW>::THREADS), 2) void conv3x3_w1_kernel(W1Params p) {
  using G = W1Geom<TC, FULLW, WIDE, CW>;
  constexpr int PR = G::PR, MW = 2, COFF = G::COFF, PWS = G::PWS, PLANE = G::PLANE, VBUF = G::VBUF;
  constexpr int PAIR_PITCH = G::PAIR_PITCH, COL_PITCH = G::COL_PITCH;
  constexpr int NP = G::NP, NROW = G::NROW, NITEM = G::NITEM;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler: scalar branches below
  constexpr bool C64 = G::C64;
  const int wn = C64 ? (wave & 1) : (CW == 256 ? wave : (wave & 3)), wg = C64 ? (wave >> 1) : (CW == 256 ? 0 : (wave >> 2));
  const int half = lane >> 5;

  int m_tile, n_tile;
  if (!block_map(p, m_tile, n_tile)) return;
#ifdef W1_CLK
  const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int pair0 = (m_tile / p.mt_cols) * PR;
  const int col0 = (m_tile % p.mt_cols) * TC;
  const bool at_left = col0 == 0, at_right = col0 + TC == p.W;
  __bf16* sV = (__bf16*)dsm_raw;            // two buffers of VBUF elements

  // this wave's two tiles.  !WIDE: 32 pairs of one column each - TC = 4: columns 2 wg, 2 wg + 1; TC = 2: columns 0, 1 of
  // pair group wg.  WIDE (wg = 0): tile m = 16 pairs x columns 2 m, 2 m + 1; MFMA row i = pair i & 15 of column i >> 4
  // C64: tile m of column half wg = 8 pairs x columns 8 wg + 4 m .. + 3; MFMA row i = pair i & 7 of column i >> 3
  const int mrow = (WIDE || TC == 4) ? 0 : wg * 32;
  const int mcol0 = C64 ? 8 * wg : ((!WIDE && TC == 4) ? 2 * wg : 0);
  int pbase[MW];
  {
    const int i = lane & 31;
    const int pr = C64 ? (i & 7) : (WIDE ? (i & 15) : mrow + i), dc = C64 ? (i >> 3) : (WIDE ? (i >> 4) : 0);
#pragma unroll
    for (int m = 0; m < MW; ++m)
      pbase[m] = pr * PAIR_PITCH + (mcol0 + (C64 ? 4 * m : (WIDE ? 2 * m : m)) + dc - COFF) * COL_PITCH + half * 8;
  }

  f32x16 acc[4][MW];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;

  const int row0 = 2 * pair0;
  const int rc0 = row0 % p.Hp;
  (void)rc0;
  const bool live = w1_block_live(p, row0, 2 * PR);
  const bool all_pad = !live;
  const int nstep_all = p.Cin / KS;
  const int sbeg = p.partial ? (int)blockIdx.y * p.ksteps : 0;                                   // even
  const int nstep = p.partial ? (sbeg + p.ksteps < nstep_all ? sbeg + p.ksteps : nstep_all) : nstep_all;   // end of this slice
  if (!all_pad) {
    // Buffer descriptors (wave-uniform): out-of-range offsets read as zero, so the rows above / below the batch and the
    // threads without a staging item need no branches.
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.in, 0, (int)((unsigned)p.rows_total * (unsigned)p.W * (unsigned)p.Cin * 4u), 0x00020000);
    const int NT32 = p.Cout >> 5;
    const unsigned ks_bytes = (unsigned)NT32 * 2048u;   // one k-step of a (chunk, group): NT32 x (hi, lo) x 1 KiB
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wpk, 0, (int)((unsigned)(p.Cin / 32) * 24u * ks_bytes), 0x00020000);
    // weight fragment of (K step s = 2 chunk + k-step, group gi = kx * 4 + p, plane) for this wave's 32 channels
    const unsigned wvoff = (unsigned)((n_tile * (C64 ? 2 : CW / 32) + wn) * 2048 + lane * 16);
    typedef int ko_i32x4 __attribute__((ext_vector_type(4)));
    bool fire = false;   // W1_KO & (64 | 128 | 256): true inside the K loop
    const ko_i32x4 srd_w = {(int)(unsigned)(uintptr_t)p.wpk, (int)(((uintptr_t)p.wpk >> 32) & 0xffff),
                            (int)((unsigned)(p.Cin / 32) * 24u * ks_bytes), 0x00020000};
    const ko_i32x4 srd_in = {(int)(unsigned)(uintptr_t)p.in, (int)(((uintptr_t)p.in >> 32) & 0xffff),
                             (int)((unsigned)p.rows_total * (unsigned)p.W * (unsigned)p.Cin * 4u), 0x00020000};
    auto forget_buf = [&](const ko_i32x4& srd, unsigned voff, unsigned soff) {
      ko_i32x4 scratch;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(scratch) : "v"(voff), "s"(srd), "s"(soff) : "memory");
    };
    auto forget_lds = [&](const __bf16* ptr) {
      ko_i32x4 scratch;
      const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(const char*)ptr;
      asm volatile("ds_read_b128 %0, %1" : "=v"(scratch) : "v"(a) : "memory");
    };
    auto w_load = [&](int s, int gi, bf16x8 (&w)[2]) {
      unsigned soff = (unsigned)(((s >> 1) * 12 + gi) * 2 + (s & 1)) * ks_bytes;
      if (W1_KO & 512) soff = (unsigned)(gi & 1) * ks_bytes;   // every request hits the same 2 x 8 KB: weights from the L1
      if (W1_KO & 1024) soff = (unsigned)((gi * 8 + (s & 7)) % 96) * ks_bytes;   // 96 blocks: 0.8 MB per channel tile (L2)
      if ((W1_KO & 64) && fire) {
        forget_buf(srd_w, wvoff, soff);
        forget_buf(srd_w, wvoff + 1024u, soff);
        return;
      }
      w[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, soff, 0));
      w[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff + 1024u, soff, 0));
    };

    // Staging descriptor of this thread's item, once per workgroup: byte offset of its first input row (row 2 pair - 1:
    // "negative" for the very first row of the batch -> out of range -> zeros; the rows below it wrap back into range)
    // and its LDS offset.  Threads without an item / columns outside the image: an offset that stays out of range.
    unsigned vbase;
    unsigned lofs = 0;       // LDS element offset of the item's first pair inside a plane
    const bool has_item = tid < NITEM;
    {
      const int q = tid & 3, pc = (tid >> 2) % PWS, pr = NP * ((tid >> 2) / PWS);
      const int gr0 = 2 * (pair0 + pr) - 1, gc = col0 - 1 + COFF + pc;
      const bool ok = has_item && gc >= 0 && gc < p.W;
      vbase = ok ? (unsigned)((gr0 * p.W + gc) * p.Cin + q * 4) * 4u : 0x80000000u;
      lofs = (unsigned)(pr * PAIR_PITCH + pc * COL_PITCH + q * 4);
    }
    const unsigned row_bytes = (unsigned)(p.W * p.Cin) * 4u;
    // Raw rows of the step(s) ahead.  WIDE keeps TWO sets: the rows of step s + 2 are requested at the top of step s and
    // transformed during step s + 1, i.e. the request has a whole step plus four groups (~3000 cycles) to come back from
    // HBM; with one set (the 512-thread forms, whose registers are spoken for) it has four groups (~770 cycles).
    constexpr int NSET = G::NSET;
    f32x4 pre[NSET][NROW];
    auto patch_request = [&](int s, auto SET_) {
      constexpr int st = decltype(SET_)::value;
      const unsigned v0 = vbase + (unsigned)(s * KS * 4);
      if ((W1_KO & 256) && fire) {
#pragma unroll
        for (int r = 0; r < NROW; ++r) forget_buf(srd_in, v0 + (unsigned)r * row_bytes, 0u);
        return;
      }
#pragma unroll
      for (int r = 0; r < NROW; ++r)
        pre[st][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, v0 + (unsigned)r * row_bytes, 0, 0));
    };
    // one of the 4 NP (pair, position) pieces of the item: transform, split, two 8-byte LDS stores
    auto commit_piece = [&](__bf16* buf, int piece, auto SET_) {
      constexpr int st = decltype(SET_)::value;
      if (!has_item) return;
      const int e = piece >> 2, q = piece & 3;
      const f32x4 d0 = pre[st][2 * e], d1 = pre[st][2 * e + 1], d2 = pre[st][2 * e + 2], d3 = pre[st][2 * e + 3];
      const f32x4 v = q == 0 ? d0 - d2 : (q == 1 ? d1 + d2 : (q == 2 ? d2 - d1 : d1 - d3));
      u32x2 hi, lo;
      split_bf16x4(v, hi, lo);
      __bf16* dst = buf + lofs + e * PAIR_PITCH + (2 * q) * PLANE;
      *(u32x2*)dst = hi;
      *(u32x2*)(dst + PLANE) = lo;
    };

    // prologue: step 0 into buffer 0
    constexpr int RING = W1_RING;   // 12 % RING == 0: the ring position of a group is static
    static_assert(12 % RING == 0, "ring");
    bf16x8 wr[RING][2];   // ring of weight fragments: group gi lives in wr[gi % RING], requested RING - 1 groups ahead
#pragma unroll
    for (int g0 = 0; g0 < RING - 1; ++g0) w_load(sbeg, g0, wr[g0]);
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, NSET - 1>;
    patch_request(sbeg, I0{});
    if (NSET == 2 && sbeg + 1 < nstep) patch_request(sbeg + 1, I1{});
#pragma unroll
    for (int piece = 0; piece < 4 * NP; ++piece) commit_piece(sV, piece, I0{});   // sbeg is even: buffer 0
    __syncthreads();

    // SL / SR (compile time): this wave's tile 0 is the first / its tile 1 the last column of the IMAGE, whose kx = 0 /
    // kx = 2 products read the zero padding beside it: nothing to add.  (Kept out of run-time branches: an MFMA under a
    // branch makes the compiler copy its accumulator.)
    auto k_loop = [&](auto SL_, auto SR_) {
      constexpr bool SL = decltype(SL_)::value, SR = decltype(SR_)::value;
      auto a_load = [&](const __bf16* buf, int gi, bf16x8 (&a)[MW][2]) {
        const int kx = gi >> 2, q = gi & 3;
        const __bf16* vh = buf + (2 * q) * PLANE + kx * COL_PITCH;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
          if ((kx == 0 && m == 0 && SL) || (kx == 2 && m == 1 && SR)) continue;
          if ((W1_KO & 128) && fire) {
            forget_lds(vh + pbase[m]);
            forget_lds(vh + PLANE + pbase[m]);
            continue;
          }
          a[m][0] = *(const bf16x8*)(vh + pbase[m]);
          a[m][1] = *(const bf16x8*)(vh + PLANE + pbase[m]);
        }
      };
      bf16x8 af[2][MW][2];   // A fragments (tile, hi | lo) of the current and the next group
      // one K step.  NXT_: the register set that holds the raw rows of step s + 1 (NSET == 2: requested a step ago; the
      // set they leave behind at the end of this step is the one step s + 3 will use, the other one is free NOW for s + 2)
      auto step = [&](int s, auto NXT_) {
        constexpr int nx = decltype(NXT_)::value;
        using Other = std::integral_constant<int, NSET == 2 ? 1 - nx : 0>;
        const __bf16* cur = sV + (s & 1) * VBUF;
        __bf16* nxt = sV + ((s + 1) & 1) * VBUF;
        const bool more = s + 1 < nstep;
        if (!(W1_KO & 8)) {
          if (NSET == 2) { if (s + 2 < nstep) patch_request(s + 2, Other{}); }
          else if (more) patch_request(s + 1, NXT_);
        }
        if (!(W1_KO & 2) || s == sbeg) a_load(cur, 0, af[0]);
        if (W1_KO & (64 | 128 | 256)) {
          if (s == sbeg) a_load(cur, 1, af[1]);   // both fragment sets hold real data before the loads turn into requests only
          fire = true;
        }
#pragma unroll
        for (int gi = 0; gi < 12; ++gi) {
          const int kx = gi >> 2, q = gi & 3;
          // the next group's fragments are requested before this group's MFMAs; weights two groups ahead (the ring
          // position of a group is static: 12 % 3 == 0)
          if (gi + 1 < 12 && (!(W1_KO & 2) || (s == sbeg && gi == 0))) a_load(cur, gi + 1, af[(gi + 1) & 1]);
          if (!(W1_KO & 1) || (s == sbeg && gi == 0)) {
            constexpr int AH = RING - 1;
            if (gi + AH < 12) w_load(s, gi + AH, wr[(gi + AH) % RING]);
            else if (more) w_load(s + 1, gi + AH - 12, wr[(gi + AH) % RING]);
          }
          if (W1_PRIO) __builtin_amdgcn_s_setprio(W1_PRIO);   // the wave that has its operands issues its MFMAs first
#pragma unroll
          for (int m = 0; m < MW; ++m) {
            if ((kx == 0 && m == 0 && SL) || (kx == 2 && m == 1 && SR)) continue;
            const bf16x8 ah = af[gi & 1][m][0], al = af[gi & 1][m][1];
            if (W1_KO & 32) {   // keep the operands alive without the matrix instructions
              asm volatile("" :: "v"(ah), "v"(al), "v"(wr[gi % RING][0]), "v"(wr[gi % RING][1]));
              continue;
            }
            acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wr[gi % RING][0], acc[q][m], 0, 0, 0);
            acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wr[gi % RING][1], acc[q][m], 0, 0, 0);
            acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wr[gi % RING][0], acc[q][m], 0, 0, 0);
          }
          if (W1_PRIO) __builtin_amdgcn_s_setprio(0);
          // the next step's planes, a piece per group
          if (more) {
            constexpr int PPG = NP > 2 ? NP / 2 : 1, FIRST = 12 - 4 * NP / PPG;   // NP = 4: two pieces per group
            if (gi >= FIRST && !(W1_KO & 4)) {
#pragma unroll
              for (int k = 0; k < PPG; ++k) commit_piece(nxt, (gi - FIRST) * PPG + k, NXT_);
            }
          }
          if (W1_SGB) {   // the group's requests first, then its VALU / LDS-store work spread between its MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // DS reads
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM reads
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, W1_SGB, 0);   // a few VALU
            }
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   // DS writes
          }
          __builtin_amdgcn_sched_barrier(0);   // keep every group's requests and pieces inside the group
        }
        if (!(W1_KO & 16)) lds_barrier();   // step s + 1 is complete in `nxt`; every wave is done reading `cur` (weight requests stay in flight)
      };
      if (NSET == 2) {
#pragma unroll 1
        for (int s = sbeg; s < nstep; s += 2) {
          step(s, I1{});                        // step 1's rows went into set 1 in the prologue, step 2's go into set 0, ...
          if (s + 1 < nstep) step(s + 1, I0{});
        }
      } else {
#pragma unroll 1
        for (int s = sbeg; s < nstep; ++s) step(s, I0{});
      }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (WIDE) {
      k_loop(F_{}, F_{});                       // two-column tiles: the halo columns beside the image are staged as zeros
    } else if (TC == 2) {
      k_loop(T_{}, T_{});                       // both columns of the image in every wave
    } else {
      const bool sl = at_left && wg == 0, sr = at_right && wg == 1;   // wave-uniform
      if (sl) k_loop(T_{}, F_{});
      else if (sr) k_loop(F_{}, T_{});
      else k_loop(F_{}, F_{});
    }
  }

#ifdef W1_CLK
  if (tid == 0) {
    atomicAdd(&w1_clk[0], __builtin_readcyclecounter() - clk0);
    atomicAdd(&w1_clk[1], __builtin_amdgcn_s_memrealtime() - rt0);
    atomicAdd(&w1_clk[2], 1ull);
  }
#endif
  // ---- epilogue: output transform, BN, ReLU, pooling / mean, zero rows.  Lane l owns channel l % 32 and the MFMA rows
  // i = 8 (r / 4) + 4 (l / 32) + r % 4 of each tile: both rows of a pair, and the two columns of a pooling window / of the
  // last layer's mean - the wave's two tiles (!WIDE), registers r and r + 8 of one tile (WIDE) - sit in this lane ----
  const int ch = n_tile * (C64 ? 64 : CW) + wn * 32 + (lane & 31);
  const float sc = all_pad ? 0.f : p.scale[ch], sh = all_pad ? 0.f : p.shift[ch];   // dead blocks store zeros
  const FastDiv by_hp(p.Hp), by_hp_out(MODE == MODE_POOL ? p.Hp_out : 1);
#pragma unroll
  for (int r = 0; r < (C64 ? 4 : (WIDE ? 8 : 16)); ++r) {
    const int i = C64 ? 4 * half + r : 8 * (r >> 2) + 4 * half + (r & 3);   // WIDE: r < 8 -> i < 16 = the pair; C64: i < 8
    const int prow = pair0 + mrow + i;                                    // global pair index
    // y[row of the pair][column slot]: !WIDE: slot = tile; WIDE: slot = 2 tile + (column inside the tile); C64: slot =
    // 4 tile + (column inside the tile), register 4 c + r of the tile
    constexpr int NS = C64 ? 8 : (WIDE ? 4 : 2);
    float y0[NS], y1[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      const int m = C64 ? sl >> 2 : (WIDE ? sl >> 1 : sl), rr = C64 ? r + 4 * (sl & 3) : (WIDE ? r + 8 * (sl & 1) : r);
      const float m0 = acc[0][m][rr], m1 = acc[1][m][rr], m2 = acc[2][m][rr], m3 = acc[3][m][rr];
      if (p.partial) {   // this slice's share of the two output rows, before BatchNorm
        y0[sl] = (m0 + m1) + m2;
        y1[sl] = (m1 - m2) - m3;
      } else {
        y0[sl] = fmaxf(fmaf((m0 + m1) + m2, sc, sh), 0.f);
        y1[sl] = fmaxf(fmaf((m1 - m2) - m3, sc, sh), 0.f);
      }
    }
    const int gr = 2 * prow;
    if (gr >= p.rows_total) continue;
    if (p.partial) {
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        float* o = p.partial + (((size_t)blockIdx.y * p.rows_total + gr) * p.W + col0 + mcol0 + sl) * p.Cout + ch;
        o[0] = all_pad ? 0.f : y0[sl];
        o[(size_t)p.W * p.Cout] = all_pad ? 0.f : y1[sl];
      }
      continue;
    }
    if (MODE == MODE_FULL) {
      const int h = by_hp.mod(gr);   // Hp is even: both rows of a pair belong to one clip
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const size_t oi = ((size_t)gr * p.W + col0 + mcol0 + sl) * p.Cout + ch;
        float* o = p.out + oi;
        o[0] = (h < p.H ? y0[sl] : 0.f) * p.drop.mask(oi);
        o[(size_t)p.W * p.Cout] = (h + 1 < p.H ? y1[sl] : 0.f) * p.drop.mask(oi + (size_t)p.W * p.Cout);
      }
    } else if (MODE == MODE_POOL) {   // a row pair IS a pooled row; column slots (0, 1) and (2, 3) are pooled columns
      const bool valid = by_hp_out.mod(prow) < p.H_out;
#pragma unroll
      for (int oc = 0; oc < NS / 2; ++oc) {
        const float o = 0.25f * ((y0[2 * oc] + y1[2 * oc]) + (y0[2 * oc + 1] + y1[2 * oc + 1]));
        const size_t oi = ((size_t)prow * p.W_out + ((col0 + mcol0) >> 1) + oc) * p.Cout + ch;
        p.out[oi] = (valid ? o : 0.f) * p.drop.mask(oi);
      }
    } else {   // MEANW: TC == 2
      int h;
      const int b = by_hp.div(gr, h);
      if (h < p.H) p.out[((size_t)b * p.H + h) * p.Cout + ch] = 0.5f * (y0[0] + y0[1]);
      if (h + 1 < p.H) p.out[((size_t)b * p.H + h + 1) * p.Cout + ch] = 0.5f * (y1[0] + y1[1]);
    }
  }
}

// Adds the K slices of a split launch in slice order (deterministic) and applies the epilogue of conv3x3_w1_kernel: one
// thread per 4 channels of one output element (FULL: a pixel; POOL: a pooled pixel; MEANW: a (clip, row)).
template <int MODE>
__global__ __launch_bounds__(256) void w1_finish_kernel(W1Params p, int slices, int block_rows) {
  const int c4n = p.Cout / 4;
  const long n_out = MODE == MODE_FULL ? (long)p.rows_total * p.W * c4n
                   : MODE == MODE_POOL ? (long)(p.rows_total / 2) * p.W_out * c4n : (long)p.rows_total * c4n;
  const long slice_stride = (long)p.rows_total * p.W * p.Cout;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_out; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long e = i / c4n;
    const f32x4 sc = *(const f32x4*)(p.scale + c), sh = *(const f32x4*)(p.shift + c);
    auto pixel = [&](long row, int col) {   // BN + ReLU of the summed slices of one full-resolution pixel
      if (!w1_block_live(p, (int)(row / block_rows) * block_rows, block_rows)) return (f32x4){0.f, 0.f, 0.f, 0.f};   // as the conv kernel stores
      const float* q = p.partial + ((size_t)row * p.W + col) * p.Cout + c;
      f32x4 v = *(const f32x4*)q;
      for (int k = 1; k < slices; ++k) v += *(const f32x4*)(q + k * slice_stride);
      f32x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
      return y;
    };
    if (MODE == MODE_FULL) {
      const int col = (int)(e % p.W);
      const long row = e / p.W;
      const f32x4 y = pixel(row, col);
      const bool valid = (int)(row % p.Hp) < p.H;
      *(f32x4*)(p.out + ((size_t)row * p.W + col) * p.Cout + c) = valid ? y : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else if (MODE == MODE_POOL) {
      const int pc = (int)(e % p.W_out);
      const long prow = e / p.W_out;
      const f32x4 a = pixel(2 * prow, 2 * pc), b = pixel(2 * prow + 1, 2 * pc), c0 = pixel(2 * prow, 2 * pc + 1),
                  d = pixel(2 * prow + 1, 2 * pc + 1);
      const bool valid = (int)(prow % p.Hp_out) < p.H_out;
      const f32x4 o = 0.25f * ((a + b) + (c0 + d));   // the kernel's order: (y0[2 oc] + y1[2 oc]) + (y0[2 oc + 1] + y1[2 oc + 1])
      *(f32x4*)(p.out + ((size_t)prow * p.W_out + pc) * p.Cout + c) = valid ? o : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
      const int h = (int)(e % p.Hp);
      const long b = e / p.Hp;
      if (h < p.H) *(f32x4*)(p.out + ((size_t)b * p.H + h) * p.Cout + c) = 0.5f * (pixel(e, 0) + pixel(e, 1));
    }
  }
}

// K slices of a launch with `wgs` workgroups over `nstep` K steps: as many as fill ~256 CUs, at least 4 steps (even) each
static inline int w1_slices(long wgs, int nstep, int* ksteps) {
  int S = 1;
  if (wgs <= 64 && nstep >= 16) {
    S = (int)(256 / wgs);
    if (S > nstep / 4) S = nstep / 4;
  }
  int k = (nstep + S - 1) / S;
  k += k & 1;
  *ksteps = k;
  return (nstep + k - 1) / k;
}

template <int MODE, int TC, bool FULLW, bool WIDE, int CW = 128>
int launch_w1(W1Params p, hipStream_t s) {
  using G = W1Geom<TC, FULLW, WIDE, CW>;
  const int pairs = p.rows_total / 2;
  p.MT = ((pairs + G::PR - 1) / G::PR) * p.mt_cols;
  unsigned grid;
  if (p.map_mode == 2) grid = (unsigned)(((p.MT + 7) / 8) * 8 * p.NT);
  else if (p.map_mode == 3) grid = (unsigned)(((p.MT + 8 / p.NT - 1) / (8 / p.NT)) * 8);
  else if (p.map_mode == 4) grid = (unsigned)(((p.MT / p.mt_cols + 7) / 8) * 8 * p.mt_cols * p.NT);
  else grid = (unsigned)(p.MT * p.NT);
  constexpr size_t lds = (size_t)2 * G::VBUF * 2;   // two buffers of 4 positions x (hi, lo) planes, bf16
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)conv3x3_w1_kernel<MODE, TC, FULLW, WIDE, CW>, 160 * 1024, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  int slices = 1;
  if (p.partial) {
    slices = w1_slices(grid, p.Cin / KS, &p.ksteps);
    if (slices == 1) p.partial = nullptr;
  }
  hipLaunchKernelGGL((conv3x3_w1_kernel<MODE, TC, FULLW, WIDE, CW>), dim3(grid, slices), dim3(G::THREADS), lds, s, p);
  if (ac_check_launch() != 0) return AC_ERR_LAUNCH;
  if (slices > 1) {
    const long n_out = (MODE == MODE_FULL ? (long)p.rows_total * p.W : MODE == MODE_POOL ? (long)(p.rows_total / 2) * p.W_out
                                                                                         : (long)p.rows_total) * (p.Cout / 4);
    hipLaunchKernelGGL(w1_finish_kernel<MODE>, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, p, slices, 2 * G::PR);
    return ac_check_launch();
  }
  return 0;
}

// workgroups of a launch (before slicing): the same arithmetic as launch_w1
static long w1_grid(int rows_total, int W, int Cout) {
  const bool c64 = Cout == 64;
  const int pr = c64 ? 8 : (W >= 8 ? 16 : 128 / W), cols = c64 ? W / 16 : (W == 2 ? 1 : W / 4), nt = c64 ? 1 : Cout / 128;
  return (long)(((rows_total / 2) + pr - 1) / pr) * cols * nt;
}

}  // namespace

#ifdef W1_CLK
extern "C" int ac_w1_clk_read(unsigned long long* out3, int reset) {
  if (hipMemcpyFromSymbol(out3, HIP_SYMBOL(w1_clk), 24) != hipSuccess) return -2;
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(w1_clk), z, 32) != hipSuccess) return -2; }
  return 0;
}
#endif

static int w1_dispatch(const float* in, const void* wfrag, const float* scale, const float* shift, float* out, int B,
                       int Hp, int H, int W, int Cin, int Cout, int mode, int map_mode, const int* clip_frames,
                       int need_mul, int need_add, float* workspace, long workspace_floats, void* stream,
                       Drop drop = make_drop(0.f, 0, nullptr)) {
  if (!in || !wfrag || !scale || !shift || !out) return AC_ERR_ARG;
  const bool c64 = Cout == 64;   // conv2 of block 1: the 16-column form
  if (B <= 0 || Hp <= H || (Hp & 1) || W < 2 || (W != 2 && (W & 3)) || Cin % 32 || (Cout % 128 && !c64)) return AC_ERR_ARG;
  if (c64 && (W % 16 || mode == MODE_MEANW)) return AC_ERR_ARG;
  if (mode < 0 || mode > 2) return AC_ERR_ARG;
  if (mode == MODE_MEANW && W != 2) return AC_ERR_ARG;
  // 32-bit BYTE offsets into the input: the staging addresses are (row, column, channel) * 4 through one buffer descriptor,
  // and lanes without an item are parked on 0x80000000 + their row / step offsets, which must stay out of range
  if (((unsigned long long)B * Hp + 16) * W * Cin * 4 >= (1ull << 31)) return AC_ERR_ARG;
  W1Params p;
  p.in = in; p.wpk = wfrag; p.scale = scale; p.shift = shift; p.out = out;
  p.rows_total = B * Hp; p.Hp = Hp; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.mt_cols = c64 ? W / 16 : (W == 2 ? 1 : W / 4);
  p.MT = 0;
  p.NT = c64 ? 1 : Cout / 128;
  p.Hp_out = Hp / 2; p.H_out = H / 2; p.W_out = W / 2;
  const bool map_mode_auto = map_mode < 0;
  if (map_mode < 0) map_mode = (p.NT % 8 == 0) ? 1 : ((p.NT == 1 || p.NT == 2 || p.NT == 4) ? 3 : 2);
  if (map_mode == 1 && p.NT % 8 != 0) return AC_ERR_ARG;
  if (map_mode == 3 && !(p.NT == 1 || p.NT == 2 || p.NT == 4 || p.NT == 8)) return AC_ERR_ARG;
  p.map_mode = map_mode;
  p.clip_frames = clip_frames; p.need_mul = need_mul; p.need_add = need_add;
  p.partial = nullptr; p.ksteps = 0;
  p.drop = drop;
  if (drop.thresh != 0 && mode == MODE_MEANW) return AC_ERR_ARG;   // dropout sits BEFORE the mean over mel: use mode 0
  if (workspace && drop.thresh == 0) {   // K slices for launches of a few workgroups, if the caller's workspace holds them
    int ksteps;
    const int slices = w1_slices(w1_grid(p.rows_total, W, Cout), Cin / KS, &ksteps);
    if (slices > 1 && (long)slices * p.rows_total * W * Cout <= workspace_floats) p.partial = workspace;
  }
  hipStream_t s = (hipStream_t)stream;
  if (c64) {
    if (mode == MODE_FULL) return launch_w1<MODE_FULL, 16, false, true>(p, s);
    return launch_w1<MODE_POOL, 16, false, true>(p, s);
  }
  if (W == 2) {
    if (mode == MODE_FULL) return launch_w1<MODE_FULL, 2, true, false>(p, s);
    if (mode == MODE_MEANW) return launch_w1<MODE_MEANW, 2, true, false>(p, s);
    return AC_ERR_ARG;   // W = 2 is never pooled
  }
  if (W == 4) {
    if (mode == MODE_FULL) return launch_w1<MODE_FULL, 4, true, false>(p, s);
    if (mode == MODE_POOL) return launch_w1<MODE_POOL, 4, true, false>(p, s);
    return AC_ERR_ARG;
  }
  // 256-channel workgroups (one staging per 256 channels) when that still leaves two rounds of workgroups for the chip;
  // AUDIOCAPTION_W1_CW=128 keeps 128 everywhere (development)
  static const bool cw128 = getenv("AUDIOCAPTION_W1_CW") && !strcmp(getenv("AUDIOCAPTION_W1_CW"), "128");
  if (!cw128 && !p.partial && Cout % 256 == 0 && w1_grid(p.rows_total, W, Cout) / 2 >= 512) {
    p.NT = Cout / 256;
    if (map_mode_auto) p.map_mode = (p.NT % 8 == 0) ? 1 : ((p.NT == 1 || p.NT == 2 || p.NT == 4) ? 3 : 2);
    if (!((p.map_mode == 1 && p.NT % 8 != 0) || (p.map_mode == 3 && !(p.NT == 1 || p.NT == 2 || p.NT == 4 || p.NT == 8)))) {
      if (mode == MODE_FULL) return launch_w1<MODE_FULL, 4, false, true, 256>(p, s);
      if (mode == MODE_POOL) return launch_w1<MODE_POOL, 4, false, true, 256>(p, s);
    }
    p.NT = Cout / 128;
    p.map_mode = map_mode;
  }
  if (mode == MODE_FULL) return launch_w1<MODE_FULL, 4, false, true>(p, s);
  if (mode == MODE_POOL) return launch_w1<MODE_POOL, 4, false, true>(p, s);
  return AC_ERR_ARG;
}

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_conv3x3_bn_relu_wino1d(const float* in, const void* wfrag, const float* scale, const float* shift,
                                         float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                         int map_mode, const int* clip_frames, int need_mul, int need_add,
                                         void* stream) {
  return w1_dispatch(in, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, clip_frames, need_mul, need_add,
                     nullptr, 0, stream);
}

extern "C" long ac_conv3x3_wino1d_splitk_floats(int B, int Hp, int W, int Cin, int Cout) {
  if (B <= 0 || Hp <= 0 || W < 2 || Cin % 32 || Cin <= 0 || Cout <= 0) return 0;
  int ksteps;
  const int slices = w1_slices(w1_grid(B * Hp, W, Cout), Cin / KS, &ksteps);
  return slices > 1 ? (long)slices * B * Hp * W * Cout : 0;
}

extern "C" int ac_conv3x3_bn_relu_wino1d_splitk(const float* in, const void* wfrag, const float* scale,
                                                const float* shift, float* out, int B, int Hp, int H, int W, int Cin,
                                                int Cout, int mode, int map_mode, const int* clip_frames,
                                                int need_mul, int need_add, float* workspace,
                                                long workspace_floats, void* stream) {
  if (!workspace || workspace_floats <= 0) return AC_ERR_ARG;
  return w1_dispatch(in, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, clip_frames, need_mul, need_add,
                     workspace, workspace_floats, stream);
}


extern "C" int ac_conv3x3_bn_relu_wino1d_drop(const float* in, const void* wfrag, const float* scale, const float* shift,
                                              float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                              int map_mode, float drop_p, unsigned long long drop_seed,
                                              const unsigned long long* seed_dev, void* stream) {
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return AC_ERR_ARG;
  return w1_dispatch(in, wfrag, scale, shift, out, B, Hp, H, W, Cin, Cout, mode, map_mode, nullptr, 0, 0, nullptr, 0, stream,
                     make_drop(drop_p, drop_seed, seed_dev));
}
