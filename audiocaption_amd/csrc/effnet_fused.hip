// EfficientNet-B2 MBConv on gfx950, fused: expand 1x1 convolution (+ BN + swish) -> depthwise k x k convolution
// (+ BN + swish) -> squeeze sums in ONE kernel, so that the EXPANDED tensor (6x the block input: 788 MB for the first
// stride-2 block at 128 clips, 296 MB for each of its successors) never exists in HBM.  The unfused chain
// (ac_pointwise_conv / ac_gemm_bf16x3 -> ac_effnet_depthwise, csrc/effnet.hip) writes it once and reads it once; every
// kernel of this encoder is HBM-bound, so those two passes were ~40 % of the encoder's traffic.
//
// Arithmetic replaced: MBConvBlock.forward of efficientnet_pytorch==0.7.1 up to the squeeze (un-vendored; the
// reference's call sites are hf_wrapper.py:229-241 / cnn_encoder.py:798-805, its restatement of the construction
// eff_latent_encoder.py:74-186).  Checked against oracle/effb2_path.py and the independent-witness fixture tests/golden/g11_effb2.npz.
//
// Work decomposition.  A workgroup owns one clip, a band of R output time rows over ALL mel columns, and a group of
// 32-channel chunks of the expanded tensor (the expand and the depthwise convolution are independent per expanded
// channel).  It stages the band's input rows [(R-1) S + k][F][Cin] in LDS once, then per chunk:
//   expand   x W_e^T on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32; weights as the row operand so that a lane
//            ends up with 16 channels of ONE position and stores them to LDS as four 16-byte words), + bias, swish.
//            A wave owns up to 3 position tiles and walks k ONCE for all of them: every weight word is loaded once per
//            chunk and wave, the next k group's words are requested before the current group's MFMAs.  Positions
//            outside the image are written as zeros (they are the depthwise convolution's zero padding: swish(bias)
//            would be wrong there); the mel padding columns of the LDS tile are zeroed once;
//   depthwise (position, channel quad) items straight from the LDS tile (row / column of the next item follow from the
//            previous one without a division), BN, swish, 16-byte stores of the 32-channel row segment
//            (128 B contiguous per position), squeeze sums through LDS atomics -> one global atomic per channel per
//            workgroup (as ac_effnet_depthwise does).
// LDS tile of the expanded chunk: [row][column'][32 channels] with 128-byte position rows, the 16-byte slot of channel
// quad q of position p stored at slot q ^ (p & 7): the depthwise reads (8 lanes = the 8 quads of one position, the next
// 8 lanes the next output position) sweep all 64 banks once per two positions, the expand writes (lane = position,
// fixed quad) are spread by the XOR.  Rows are padded to a multiple of 8 positions (the XOR term then does not depend
// on the tap row); with stride 2 the columns are stored de-interleaved (even columns, then odd ones), so that the
// outputs fo, fo + 1 of a tap read NEIGHBOURING positions as they do with stride 1.
// With y == NULL only the squeeze sums are produced.
#include "ac_common.h"
#include <stdlib.h>
#include <stdio.h>

namespace {

__device__ __forceinline__ float swish_fast(float v) { return ac_swish_fast(v); }

struct EdP {
  const float* x; const float* we; const float* be; const float* wd; const float* scale; const float* shift;
  float* y; float* pool;
  int T, F, To, Fo, Cin, Cmid, pb, R, Rin, Fq, MT, cpg, nchunks;   // Fq: positions per row of the LDS tile (% 8 == 0)
  float pool_scale;
};

#ifndef ED_UNROLL3
#define ED_UNROLL3 1
#endif
constexpr int ED_MAXJ = 3;    // position tiles per wave in the expand phase (MT <= 12)

// physical column of logical padded column c in an LDS row: stride 2 de-interleaves (even columns first)
template <int S>
__device__ __forceinline__ int ed_col(int c, int Fq) { return S == 1 ? c : (c & 1) * (Fq >> 1) + (c >> 1); }

template <int K, int S>
__global__ __launch_bounds__(256, 2) void expand_dw_kernel(EdP p) {
  extern __shared__ __attribute__((aligned(16))) float ed_smem[];
  const int tid = threadIdx.x;
  const int rt = tid, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.y;
  const int to0 = blockIdx.x * p.R;
  const int ti0 = to0 * S - p.pb;                    // input row of the band's first tap
  const int PX = p.Cin + 4;
  const int ES = p.Rin * p.Fq * 32;                   // floats per buffer of the expanded tile
  float* xs = ed_smem;                                // [MT * 32 positions][PX]
  float* es0 = xs + p.MT * 32 * PX;                   // [Rin][Fq][32]
  float* spool = es0 + ES;                            // [32]
  const int npos = p.Rin * p.F;
  {
    // the band's input rows: requests of four items are in flight before the first is stored
    const int C4 = p.Cin >> 2;
    const int items = p.MT * 32 * C4;
    const float* xb = p.x + (long)b * p.T * p.F * p.Cin;
    for (int it0 = tid; it0 < items; it0 += 256 * 4) {
      f32x4 v[4];
      int off[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + 256 * u;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        off[u] = -1;
        if (it < items) {
          const int pos = it / C4, c4 = it - pos * C4;
          off[u] = pos * PX + c4 * 4;
          if (pos < npos) {
            const int i = pos / p.F;
            const int t = ti0 + i;
            if (t >= 0 && t < p.T) v[u] = *(const f32x4*)(xb + ((long)t * p.F + (pos - i * p.F)) * p.Cin + c4 * 4);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (off[u] >= 0) *(f32x4*)(xs + off[u]) = v[u];
    }
    const int en = p.Rin * p.Fq * 8;
    for (int it = tid; it < en; it += 256) ((f32x4*)es0)[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < 32) spool[tid] = 0.f;
  }

  // ---- expand: this wave's position tiles and where their lanes write ----
  int e_wr[ED_MAXJ];          // float offset of the position's row in a buffer, or -1: no write (beyond the band)
  bool e_ok[ED_MAXJ];         // inside the image (else zeros are written)
  int e_sw[ED_MAXJ];          // position & 7 (slot XOR)
  const float* xa[ED_MAXJ];
#pragma unroll
  for (int j = 0; j < ED_MAXJ; ++j) {
    const int mt = wave + 4 * j;
    const int pos = mt * 32 + l31;
    e_wr[j] = -1;
    e_ok[j] = false;
    e_sw[j] = 0;
    xa[j] = xs + (min(mt, p.MT - 1) * 32 + l31) * PX + 4 * half;
    if (mt < p.MT && pos < npos) {
      const int i = pos / p.F, f = pos - i * p.F;
      const int t = ti0 + i;
      const int pp = i * p.Fq + ed_col<S>(f + p.pb, p.Fq);
      e_wr[j] = pp * 32;
      e_sw[j] = pp & 7;
      e_ok[j] = t >= 0 && t < p.T;
    }
  }
  const int njobs = wave < p.MT ? (p.MT - wave + 3) / 4 : 0;
  const int ngroups = p.Cin >> 3;
  // expand chunk ch into buffer es: es[pos][32 channels] = swish(x[pos] . W_e[32 ch ..] + bias), zero outside the image
  auto expand = [&](int ch, float* es) {
    if (njobs == 0) return;
    const int c0 = ch * 32;
    const int crow = min(c0 + l31, p.Cmid - 1);
    const float* wrow = p.we + (long)crow * p.Cin + 4 * half;
    f32x4 bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cb = c0 + 8 * g + 4 * half;
      bias[g] = cb < p.Cmid ? *(const f32x4*)(p.be + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x16 acc[ED_MAXJ];
#pragma unroll
    for (int j = 0; j < ED_MAXJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 wb = *(const f32x4*)wrow;
    for (int gq = 0; gq < ngroups; ++gq) {
      const f32x4 wnx = *(const f32x4*)(wrow + 8 * min(gq + 1, ngroups - 1));
#pragma unroll
      for (int j = 0; j < ED_MAXJ; ++j) {
        if (j < njobs) {
          const f32x4 va = *(const f32x4*)(xa[j] + 8 * gq);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[j] = mfma32(wb[q], va[q], acc[j]);
        }
      }
      wb = wnx;
    }
#pragma unroll
    for (int j = 0; j < ED_MAXJ; ++j) {
      if (e_wr[j] < 0) continue;
      float* e = es + e_wr[j];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
        v += bias[g];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = e_ok[j] ? swish_fast(v[q]) : 0.f;
        *(f32x4*)(e + (((2 * g + half) ^ e_sw[j]) << 2)) = v;
      }
    }
  };

  // ---- depthwise: thread = (channel quad cq, slot), items o = slot + 32 n over the band's outputs; (row, column) of
  // the next item follow from the previous one without a division ----
  const int cq = rt & 7, slot = rt >> 3;
  const int nout = p.R * p.Fo;
  const int r_first = slot / p.Fo, fo_first = slot - r_first * p.Fo;
  const int q32 = 32 / p.Fo, m32 = 32 - q32 * p.Fo;
  int koff[K];                // position offset of tap column kf
#pragma unroll
  for (int kf = 0; kf < K; ++kf) koff[kf] = S == 1 ? kf : (kf & 1) * (p.Fq >> 1) + (kf >> 1);
  // depthwise + BN + swish + squeeze sums of chunk ch from buffer es; sums into sp[32]
  auto depthwise = [&](int ch, const float* es, float* sp) {
    const int c0 = ch * 32;
    const int c = c0 + cq * 4;
    if (c >= p.Cmid) return;
    f32x4 w[K][K];
#pragma unroll
    for (int kt = 0; kt < K; ++kt)
#pragma unroll
      for (int kf = 0; kf < K; ++kf) w[kt][kf] = *(const f32x4*)(p.wd + (long)(kt * K + kf) * p.Cmid + c);
    const f32x4 sc = *(const f32x4*)(p.scale + c), sh = *(const f32x4*)(p.shift + c);
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    int r = r_first, fo = fo_first;
#pragma unroll(K == 3 ? ED_UNROLL3 : 1)
    for (int o = slot; o < nout; o += 32) {
      const int to = to0 + r;
      if (to >= p.To) break;
      // stride 2: columns 2 fo + kf live at (kf & 1) * Fq/2 + fo + (kf >> 1)
      const int pe = (r * S) * p.Fq + fo;
      int eoff[K];          // float offset of tap column kf in the window's first row, slot XOR applied
#pragma unroll
      for (int kf = 0; kf < K; ++kf) {
        const int pp = pe + koff[kf];
        eoff[kf] = pp * 32 + ((cq ^ (pp & 7)) << 2);
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < K; ++kt)
#pragma unroll
        for (int kf = 0; kf < K; ++kf) acc += *(const f32x4*)(es + eoff[kf] + kt * p.Fq * 32) * w[kt][kf];
      f32x4 v = acc * sc + sh;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = swish_fast(v[q]);
      psum += v;
      if (p.y) *(f32x4*)(p.y + (((long)b * p.To + to) * p.Fo + fo) * p.Cmid + c) = v;
      fo += m32;
      r += q32;
      if (fo >= p.Fo) {
        fo -= p.Fo;
        ++r;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(&sp[cq * 4 + q], psum[q]);
  };

  __syncthreads();
  const int chunk0 = blockIdx.z * p.cpg;
  const int chunk1 = min(p.nchunks, chunk0 + p.cpg);
  for (int ch = chunk0; ch < chunk1; ++ch) {
    expand(ch, es0);
    __syncthreads();
    depthwise(ch, es0, spool);
    __syncthreads();
    if (tid < 32) {
      const int cc = ch * 32 + tid;
      if (cc < p.Cmid) atomicAdd(p.pool + (long)b * p.Cmid + cc, spool[tid] * p.pool_scale);
      spool[tid] = 0.f;      // read again only after the next chunk's first barrier
    }
  }
}

int ed_lds_limit() {
  static int limit = 0;
  if (!limit) {
    const char* e = getenv("AUDIOCAPTION_EDW_LDS_KB");
    int kb = e ? atoi(e) : 76;                       // two workgroups per CU inside the 160 KB
    if (kb < 16) kb = 16;
    if (kb > 156) kb = 156;
    limit = kb * 1024;
  }
  return limit;
}

template <int K, int S>
int launch_expand_dw(EdP p, int B, int groups, size_t lds, hipStream_t st) {
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)expand_dw_kernel<K, S>, 156 * 1024, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  dim3 grid((p.To + p.R - 1) / p.R, B, groups);
  hipLaunchKernelGGL((expand_dw_kernel<K, S>), grid, dim3(256), lds, st, p);
  return ac_check_launch();
}

}  // namespace

extern "C" int ac_effnet_expand_depthwise(const float* x, const float* we, const float* be, const float* wd,
                                          const float* scale, const float* shift, float* y, float* pool, float pool_scale,
                                          int B, int T, int F, int Cin, int Cmid, int k, int stride, int pad_before,
                                          int pad_after, void* stream) {
  if (!x || !we || !be || !wd || !scale || !shift || !pool || B <= 0 || T <= 0 || F <= 0 || Cin <= 0 || Cmid <= 0 ||
      (Cin & 7) || (Cmid & 3) || (k != 3 && k != 5) || (stride != 1 && stride != 2) || pad_before < 0 || pad_after < 0 ||
      B > 65535 || ((uintptr_t)x & 15) || ((uintptr_t)we & 15) || ((uintptr_t)be & 15) || ((uintptr_t)wd & 15) ||
      ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15) || (y && ((uintptr_t)y & 15)))
    return AC_ERR_ARG;
  EdP p;
  p.x = x; p.we = we; p.be = be; p.wd = wd; p.scale = scale; p.shift = shift; p.y = y; p.pool = pool;
  p.T = T; p.F = F; p.Cin = Cin; p.Cmid = Cmid; p.pb = pad_before; p.pool_scale = pool_scale;
  p.To = (T + pad_before + pad_after - k) / stride + 1;
  p.Fo = (F + pad_before + pad_after - k) / stride + 1;
  if (p.To <= 0 || p.Fo <= 0) return AC_ERR_ARG;
  // positions per LDS row: the padded width, even for the stride-2 de-interleave, rounded up to a multiple of 8
  p.Fq = (F + pad_before + pad_after + 7) / 8 * 8;
  auto lds_bytes = [&](int R) {
    const long Rin = (long)(R - 1) * stride + k;
    const long MT = (Rin * F + 31) / 32;
    return (MT * 32 * (Cin + 4) + Rin * p.Fq * 32 + 32) * 4;
  };
  auto fits = [&](int R) {
    const long Rin = (long)(R - 1) * stride + k;
    return lds_bytes(R) <= ed_lds_limit() && (Rin * F + 31) / 32 <= 4 * ED_MAXJ;
  };
  if (!fits(1)) return AC_ERR_ARG;                   // the caller keeps the two-kernel chain for such a layer
  // band height: the cheapest per output row among those that fit.  Model: a wave's MFMA chain over its
  // ceil(MT / 4) position tiles plus the depthwise items of a thread, both per 32-channel chunk.
  int R = 1;
  double best = 1e30;
  for (int r = 1; r <= p.To && fits(r); ++r) {
    const long Rin = (long)(r - 1) * stride + k;
    const long MT = (Rin * F + 31) / 32;
    const double cost = ((double)((MT + 3) / 4) * ((Cin / 8) * 4 * 64 + 500) + (double)((r * p.Fo + 31) / 32) * (k * k * 50 + 250) +
                         600.0) / r;
    if (cost <= best) {
      best = cost;
      R = r;
    }
  }
  p.R = R;
  p.Rin = (R - 1) * stride + k;
  p.MT = (p.Rin * F + 31) / 32;
  p.nchunks = (Cmid + 31) / 32;
  const long bands = (p.To + R - 1) / R;
  long groups = (2048 + bands * B - 1) / (bands * B);   // enough workgroups for 256 CUs x 2, several rounds
  if (groups > p.nchunks) groups = p.nchunks;
  if (groups < 1) groups = 1;
  p.cpg = (int)((p.nchunks + groups - 1) / groups);
  groups = (p.nchunks + p.cpg - 1) / p.cpg;
  const size_t lds = (size_t)lds_bytes(R);
  if (getenv("AUDIOCAPTION_EDW_VERBOSE")) fprintf(stderr, "edw: F %d Cin %d Cmid %d k %d s %d -> R %d Rin %d MT %d Fq %d lds %zu bands %ld groups %ld cpg %d\n", F, Cin, Cmid, k, stride, R, p.Rin, p.MT, p.Fq, lds, bands, groups, p.cpg);
  hipStream_t st = (hipStream_t)stream;
  if (k == 3 && stride == 1) return launch_expand_dw<3, 1>(p, B, (int)groups, lds, st);
  if (k == 3) return launch_expand_dw<3, 2>(p, B, (int)groups, lds, st);
  if (stride == 1) return launch_expand_dw<5, 1>(p, B, (int)groups, lds, st);
  return launch_expand_dw<5, 2>(p, B, (int)groups, lds, st);
}
