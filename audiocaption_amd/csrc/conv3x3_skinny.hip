// Cnn14 conv blocks 5-6 for FEW PIXELS (single clips, training at the reference's per-GPU batch of 4): 3x3 convolution + eval
// BatchNorm + ReLU (+ 2x2 average pooling / mean over the two mel columns) as a K-SLICED direct convolution on split-bf16
// operands whose only job is to STREAM THE WEIGHTS ONCE at memory bandwidth.
//
// Same contract, layouts and epilogue modes as csrc/conv3x3_wino1d.hip (reference ConvBlock.forward, cnn_encoder.py:59-75;
// pooling glue of Cnn14Encoder.forward, cnn_encoder.py:431-444): f32 activations [B*Hp][W][C] in and out, W = 2 or 4.
//
// Why another kernel.  conv2 of block 6 has 2048 x 2048 x 9 weights: 151 MB as bf16 hi + lo, against 64 pixels x 2048
// channels of work for one clip.  The Winograd kernels are built for the opposite regime (weights from L2, re-used by
// thousands of pixel tiles): their K-sliced F(2,3) form moves 201 MB (12 transformed taps) with two weight fragments in
// flight per wave and reaches 0.7 TB/s - 280 us for that layer at one clip, 390 us at four, 0.65 ms of a 1.0 ms single-clip
// encoder.  Here: the direct form (9 taps: the fewest weight bytes), a workgroup = 128 output channels x ALL pixels of up
// to eight 32-pixel MFMA tiles x one slice of the (input channel) loop; a wave streams its 32 channels' fragments
// (1 KiB per instruction, straight from the fragment-ordered pack of ac_conv3x3_bn_relu_bf16x3_gw) through a register ring
// EIGHT groups deep - 16 KB in flight per wave, 64 KB per CU - while each fragment pair feeds 3 x (pixel tiles) MFMAs
// (hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_bf16, f32 accumulate: the "bf16x3" tier's arithmetic, 2^-16 operand
// error).  The input patch of a 32-channel chunk (all pixels + halo, split into bf16 hi | lo once) is double-buffered in
// LDS.  Every slice stores its raw sums to workspace[slice][pixel][Cout]; skinny_finish_kernel adds the slices IN ORDER
// (deterministic) and applies BN / ReLU / pool / mean / dropout.
#include "ac_common.h"
#include "ac_drop.h"
#include "ac_wino43.h"   // bf16x8, split_bf16x4
#include <type_traits>

namespace {

constexpr int SK_PITCH = 144;   // bytes of a patch pixel in LDS: 32 channels hi (64 B) | lo (64 B) + 16 (bank spread)

struct SkinnyParams {
  const float* in;
  const void* wpk;     // [Cin/32][9 taps][2 k-steps][Cout/32][2 (hi, lo)][64 lanes][8] bf16 (pack_conv_weight_bf16x3_frag)
  float* partial;      // [slices][rows_total * W][Cout]
  int rows_total, W, Cin, Cout;
  int chunks_per_slice, nchunk;
};

enum { SK_FULL = 0, SK_POOL = 1, SK_MEANW = 2 };

// grid (Cout / 128, slices, m tiles); 256 threads = 4 waves x 32 channels
template <int TC, int MT>
__global__ __launch_bounds__(256, MT >= 4 ? 1 : 2) void conv3x3_skinny_kernel(SkinnyParams p) {
  // weight fragment ring: a group's (hi, lo) pair is requested SK_RING - 1 groups ahead - 8 groups of 6 / 12 MFMAs, 5 groups of 24
  constexpr int SK_RING = MT == 8 ? 6 : 9;   // divides the 18 groups of a chunk: ring positions are static
  static_assert(18 % SK_RING == 0, "ring positions are static");
  constexpr bool APRE = MT < 8;   // pixel fragments read a group ahead (eight tiles: the 24 MFMAs of a group cover the read)
  constexpr int RT = 32 / TC;              // rows of an MFMA tile
  constexpr int R = MT * RT;               // output rows of this workgroup
  constexpr int PW = TC + 2, NP = (R + 2) * PW;   // patch: rows row0 - 1 .. row0 + R, columns -1 .. TC
  constexpr int PBUF = ((NP * SK_PITCH + 127) / 128) * 128;
  constexpr int NITEM = NP * 8;            // (patch pixel, channel quad) staging items of a chunk
  constexpr int IPT = (NITEM + 255) / 256;
  extern __shared__ __attribute__((aligned(128))) unsigned char sk_lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tile = blockIdx.x, slice = blockIdx.y, row0 = blockIdx.z * R;
  const int c0 = slice * p.chunks_per_slice;
  const int c1 = c0 + p.chunks_per_slice < p.nchunk ? c0 + p.chunks_per_slice : p.nchunk;
  const int NT32 = p.Cout >> 5;

  // input through a buffer descriptor rebased to the patch's first row: rows outside the batch read as zero
  const int prow0 = row0 - 1;
  const int rbase = prow0 > 0 ? prow0 : 0;
  const size_t row_elems = (size_t)p.W * p.Cin;
  const size_t left = ((size_t)p.rows_total - rbase) * row_elems * 4;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in + (size_t)rbase * row_elems), 0,
                                                                       (int)(left < 0x7fffffffull ? left : 0x7fffffffull), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)((size_t)p.nchunk * 18 * NT32 * 2048),
                                                                      0x00020000);
  const unsigned wv = (unsigned)((n_tile * 4 + wave) * 2048 + lane * 16);
  const unsigned g_bytes = (unsigned)NT32 * 2048u;   // one (chunk, tap, k-step) group
  auto w_load = [&](int gg, bf16x8 (&w)[2]) {         // gg = (chunk - c0) * 18 + tap * 2 + ks, counted from the slice's start
    const unsigned soff = (unsigned)(c0 * 18 + gg) * g_bytes;
    w[0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wv, soff, 0));
    w[1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wv + 1024u, soff, 0));
  };

  // staging items: it = tid + 256 k -> (patch pixel it >> 3, channel quad it & 7)
  unsigned gofs[IPT], lofs[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int it = tid + 256 * k;
    const int pp = it >> 3, cq = it & 7;
    const int pr = pp / PW, pc = pp % PW;
    const int grow = prow0 + pr, gcol = pc - 1;
    const bool ok = it < NITEM && gcol >= 0 && gcol < TC && grow >= 0;
    gofs[k] = ok ? (unsigned)((((grow - rbase) * p.W + gcol) * p.Cin + cq * 4) * 4) : 0x80000000u;   // parked: reads zero
    lofs[k] = (unsigned)(pp * SK_PITCH + cq * 8);
  }
  // The rows of chunk r live in register set (r - c0) & 1: requested a whole chunk before they are split and stored
  // (chunk r - 1 stages them while it multiplies), so that their latency never shows.
  f32x4 raw[2][IPT];
  auto rows_request = [&](int c, auto SET_) {
    constexpr int SET = decltype(SET_)::value;
    const unsigned cs = (unsigned)(c * 32 * 4);
#pragma unroll
    for (int k = 0; k < IPT; ++k) raw[SET][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, gofs[k] + cs, 0, 0));
  };
  auto commit = [&](unsigned char* buf, int k, auto SET_) {
    constexpr int SET = decltype(SET_)::value;
    if (tid + 256 * k < NITEM) {
      u32x2 hi, lo;
      split_bf16x4(raw[SET][k], hi, lo);
      *(u32x2*)(buf + lofs[k]) = hi;
      *(u32x2*)(buf + lofs[k] + 64) = lo;
    }
  };

  // B fragment (pixels) of tile m: lane (i, half) reads pixel (row m RT + i / TC, column i % TC) + tap offset, 8 channels
  unsigned pb[MT];
  {
    const int i = lane & 31;
#pragma unroll
    for (int m = 0; m < MT; ++m) pb[m] = (unsigned)(((m * RT + i / TC) * PW + i % TC) * SK_PITCH + half * 16);
  }
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int ngroups = (c1 - c0) * 18;
  bf16x8 wr[SK_RING][2];
  rows_request(c0, S0{});
  if (c0 + 1 < c1) rows_request(c0 + 1, S1{});
#pragma unroll
  for (int g0 = 0; g0 < SK_RING - 1; ++g0)
    if (g0 < ngroups) w_load(g0, wr[g0]);
#pragma unroll
  for (int k = 0; k < IPT; ++k) commit(sk_lds, k, S0{});
  lds_barrier();

  bf16x8 a[APRE ? 2 : 1][MT][2];   // pixel fragments of the current group (and the next one)
  auto a_load = [&](const unsigned char* buf, int g, bf16x8 (&dst)[MT][2]) {
    const int tap = g >> 1, ks = g & 1, ky = tap / 3, kx = tap % 3;
    const unsigned toff = (unsigned)((ky * PW + kx) * SK_PITCH + ks * 32);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      dst[m][0] = *(const bf16x8*)(buf + pb[m] + toff);
      dst[m][1] = *(const bf16x8*)(buf + pb[m] + toff + 64);
    }
  };
  // one chunk; PAR = (c - c0) & 1 at compile time: buffer and register set of the chunk
  auto chunk = [&](int c, auto PAR_) {
    constexpr int PAR = decltype(PAR_)::value;
    using CUR = std::integral_constant<int, PAR>;
    using NXT = std::integral_constant<int, PAR ^ 1>;
    const unsigned char* cur = sk_lds + PAR * PBUF;
    unsigned char* nxt = sk_lds + (PAR ^ 1) * PBUF;
    const bool more = c + 1 < c1;
    if (APRE) a_load(cur, 0, a[0]);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      const int gg = (c - c0) * 18 + g;
      if (gg + SK_RING - 1 < ngroups) w_load(gg + SK_RING - 1, wr[(g + SK_RING - 1) % SK_RING]);
      if (APRE) { if (g + 1 < 18) a_load(cur, g + 1, a[(g + 1) & 1]); }
      else a_load(cur, g, a[0]);
      // this chunk's rows were stored a chunk ago: its register set takes the rows of chunk c + 2
      if (g == 0 && c + 2 < c1) rows_request(c + 2, CUR{});
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[g % SK_RING][0], a[APRE ? (g & 1) : 0][m][1], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[g % SK_RING][1], a[APRE ? (g & 1) : 0][m][0], acc[m], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[g % SK_RING][0], a[APRE ? (g & 1) : 0][m][0], acc[m], 0, 0, 0);
      if (more && g + 1 < 18 && g < IPT) commit(nxt, g, NXT{});   // the next chunk's patch, an item per group (IPT <= 17)
      __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
  };
  static_assert(IPT <= 17, "one staging item per MFMA group");
  for (int c = c0; c < c1; c += 2) {
    chunk(c, S0{});
    if (c + 1 < c1) chunk(c + 1, S1{});
  }

  // raw sums -> workspace[slice][pixel][channel]: lane (pixel i, half) holds channels 8 g + 4 half + (0..3) of the wave's 32
  float* ws = p.partial + (size_t)slice * p.rows_total * p.W * p.Cout;
  const int i = lane & 31;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int grow = row0 + m * RT + i / TC, gcol = i % TC;
    if (grow < p.rows_total) {
      float* q = ws + ((size_t)grow * p.W + gcol) * p.Cout + n_tile * 128 + wave * 32 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
        *(f32x4*)(q + 8 * g) = v;
      }
    }
  }
}

struct SkinnyFinish {
  const float* partial; const float* scale; const float* shift; float* out;
  int rows_total, Hp, H, W, Cout, slices;
  Drop drop;
};

// one thread per 4 channels of one output element (FULL: a pixel; POOL: a pooled pixel; MEANW: a (clip, row))
template <int MODE>
__global__ __launch_bounds__(256) void skinny_finish_kernel(SkinnyFinish p) {
  const int c4n = p.Cout / 4;
  const int W_out = p.W / 2, Hp_out = p.Hp / 2, H_out = p.H / 2;
  const long n_out = MODE == SK_FULL ? (long)p.rows_total * p.W * c4n
                   : MODE == SK_POOL ? (long)(p.rows_total / 2) * W_out * c4n : (long)p.rows_total * c4n;
  const long slice_stride = (long)p.rows_total * p.W * p.Cout;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_out; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long e = i / c4n;
    const f32x4 sc = *(const f32x4*)(p.scale + c), sh = *(const f32x4*)(p.shift + c);
    auto pixel = [&](long row, int col) {   // BN + ReLU of the summed slices of one full-resolution pixel
      const float* q = p.partial + ((size_t)row * p.W + col) * p.Cout + c;
      f32x4 v = *(const f32x4*)q;
      for (int k = 1; k < p.slices; ++k) v += *(const f32x4*)(q + k * slice_stride);
      f32x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
      return y;
    };
    if (MODE == SK_FULL) {
      const int col = (int)(e % p.W);
      const long row = e / p.W;
      f32x4 y = (int)(row % p.Hp) < p.H ? pixel(row, col) : zero4;
      const size_t oi = ((size_t)row * p.W + col) * p.Cout + c;
      if (p.drop.thresh != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] *= p.drop.mask(oi + j);
      }
      *(f32x4*)(p.out + oi) = y;
    } else if (MODE == SK_POOL) {
      const int pc = (int)(e % W_out);
      const long prow = e / W_out;
      f32x4 o = zero4;
      if ((int)(prow % Hp_out) < H_out) {
        const f32x4 a = pixel(2 * prow, 2 * pc), b = pixel(2 * prow + 1, 2 * pc), c2 = pixel(2 * prow, 2 * pc + 1),
                    d = pixel(2 * prow + 1, 2 * pc + 1);
        o = 0.25f * ((a + b) + (c2 + d));
      }
      const size_t oi = ((size_t)prow * W_out + pc) * p.Cout + c;
      if (p.drop.thresh != 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] *= p.drop.mask(oi + j);
      }
      *(f32x4*)(p.out + oi) = o;
    } else {   // mean over the two mel columns, out (B, H, Cout) dense (cnn_encoder.py:443)
      const int h = (int)(e % p.Hp);
      const long b = e / p.Hp;
      if (h < p.H) *(f32x4*)(p.out + ((size_t)b * p.H + h) * p.Cout + c) = 0.5f * (pixel(e, 0) + pixel(e, 1));
    }
  }
}

// pixel tiles per workgroup (2, 4 or 8), K slices and m tiles of a geometry; false: not a launch for this kernel
struct SkinnyPlan { int mt, slices, chunks_per_slice, mtiles; };
inline bool skinny_plan(int rows_total, int W, int Cin, int Cout, SkinnyPlan* pl) {
  if ((W != 2 && W != 4) || Cin % 32 || Cout % 128 || rows_total <= 0) return false;
  const long px = (long)rows_total * W;
  if (px > 2048) return false;                     // beyond that the pixel tiles re-use the weights enough for the Winograd kernels
  const int mt = px <= 64 ? 2 : (px <= 128 ? 4 : 8);
  const int rows_per = mt * (32 / W);
  const int mtiles = (rows_total + rows_per - 1) / rows_per;
  const int nchunk = Cin / 32, nt = Cout / 128;
  int slices = 512 / (nt * mtiles);                // ~two workgroups' worth of slices per CU: short tails
  if (slices > nchunk / 2) slices = nchunk / 2;    // at least two chunks per slice
  if (slices < 1) slices = 1;
  const int cps = (nchunk + slices - 1) / slices;
  pl->mt = mt; pl->chunks_per_slice = cps; pl->slices = (nchunk + cps - 1) / cps; pl->mtiles = mtiles;
  return true;
}

template <int TC, int MT>
int launch_skinny(const SkinnyParams& p, const SkinnyPlan& pl, hipStream_t s) {
  constexpr int R = MT * (32 / TC), NP = (R + 2) * (TC + 2);
  constexpr size_t lds = (size_t)2 * (((NP * SK_PITCH + 127) / 128) * 128);
  static_assert(lds <= 160 * 1024, "patch buffers exceed the LDS");
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)conv3x3_skinny_kernel<TC, MT>, 160 * 1024, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL((conv3x3_skinny_kernel<TC, MT>), dim3(p.Cout / 128, pl.slices, pl.mtiles), dim3(256), lds, s, p);
  return ac_check_launch();
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" long ac_conv3x3_skinny_workspace_floats(int B, int Hp, int W, int Cin, int Cout) {
  SkinnyPlan pl;
  if (B <= 0 || Hp <= 0 || !skinny_plan(B * Hp, W, Cin, Cout, &pl)) return 0;
  return (long)pl.slices * B * Hp * W * Cout;
}

extern "C" int ac_conv3x3_bn_relu_skinny(const float* in, const void* wfrag, const float* scale, const float* shift, float* out,
                                         int B, int Hp, int H, int W, int Cin, int Cout, int mode, float* workspace,
                                         long workspace_floats, float drop_p, unsigned long long drop_seed,
                                         const unsigned long long* seed_dev, void* stream) {
  if (!in || !wfrag || !scale || !shift || !out || !workspace) return AC_ERR_ARG;
  if (B <= 0 || Hp <= H || (Hp & 1) || mode < 0 || mode > 2) return AC_ERR_ARG;
  if ((mode == SK_MEANW && W != 2) || (mode == SK_POOL && W != 4)) return AC_ERR_ARG;
  if (!(drop_p >= 0.f) || drop_p >= 1.f || (drop_p > 0.f && mode == SK_MEANW)) return AC_ERR_ARG;   // dropout sits before the mean over mel
  SkinnyPlan pl;
  if (!skinny_plan(B * Hp, W, Cin, Cout, &pl)) return AC_ERR_ARG;
  if ((long)pl.slices * B * Hp * W * Cout > workspace_floats) return AC_ERR_ARG;
  if ((unsigned long long)(Cin / 32) * 18 * (Cout / 32) * 2048 >= (1ull << 31)) return AC_ERR_ARG;   // packed weights: one descriptor
  hipStream_t s = (hipStream_t)stream;
  SkinnyParams p;
  p.in = in; p.wpk = wfrag; p.partial = workspace;
  p.rows_total = B * Hp; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.chunks_per_slice = pl.chunks_per_slice; p.nchunk = Cin / 32;
  int rc;
  if (W == 2) rc = pl.mt == 2 ? launch_skinny<2, 2>(p, pl, s) : pl.mt == 4 ? launch_skinny<2, 4>(p, pl, s) : launch_skinny<2, 8>(p, pl, s);
  else rc = pl.mt == 2 ? launch_skinny<4, 2>(p, pl, s) : pl.mt == 4 ? launch_skinny<4, 4>(p, pl, s) : launch_skinny<4, 8>(p, pl, s);
  if (rc != AC_OK) return rc;
  SkinnyFinish f;
  f.partial = workspace; f.scale = scale; f.shift = shift; f.out = out;
  f.rows_total = B * Hp; f.Hp = Hp; f.H = H; f.W = W; f.Cout = Cout; f.slices = pl.slices;
  f.drop = make_drop(drop_p, drop_seed, seed_dev);
  const long n_out = (mode == SK_FULL ? (long)f.rows_total * W : mode == SK_POOL ? (long)(f.rows_total / 2) * (W / 2) : (long)f.rows_total) *
                     (Cout / 4);
  const unsigned blocks = (unsigned)((n_out + 255) / 256 < 4096 ? (n_out + 255) / 256 : 4096);
  if (mode == SK_FULL) hipLaunchKernelGGL(skinny_finish_kernel<SK_FULL>, dim3(blocks), dim3(256), 0, s, f);
  else if (mode == SK_POOL) hipLaunchKernelGGL(skinny_finish_kernel<SK_POOL>, dim3(blocks), dim3(256), 0, s, f);
  else hipLaunchKernelGGL(skinny_finish_kernel<SK_MEANW>, dim3(blocks), dim3(256), 0, s, f);
  return ac_check_launch();
}
