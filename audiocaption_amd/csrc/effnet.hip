// EfficientNet-B2 audio encoder pieces on gfx950: everything of an MBConv block that
// is not a 1x1 convolution (those are ac_gemm with BatchNorm folded into the weights and the swish / squeeze-excite
// gate / residual in its prologue and epilogue).
//
// The arithmetic being replaced is efficientnet_pytorch==0.7.1's EfficientNet.extract_features (un-vendored; call
// sites hf_wrapper.py:229-241, cnn_encoder.py:798-805), whose construction the reference restates in
// eff_latent_encoder.py:74-186.  Activations are channels-last [clip][time][mel][channel] fp32 (time is the
// reference's W axis, mel its H axis): every kernel here is HBM-bound, so threads walk channels fastest (float4) and
// each activation is read once; the squeeze-excite mean is accumulated by the depthwise kernel that produces the
// tensor (LDS atomics per workgroup, one global atomic per channel per workgroup).
#include "ac_common.h"
#include <stdlib.h>
#include <string.h>

namespace {

__device__ __forceinline__ float swishf(float v) { return ac_swish_fast(v); }

// ---- AmplitudeToDB(top_db): clamp at (max over the whole batch) - top_db (torchaudio packs the batch axis) ----------
__global__ __launch_bounds__(256) void block_max_kernel(const float* x, long n, float* partial) {
  __shared__ float red[4];
  float m = -INFINITY;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void clamp_top_db_kernel(float* x, long n, const float* partial, int nparts, float top_db) {
  __shared__ float red[4];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, partial[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  const float lo = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - top_db;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) x[i] = fmaxf(x[i], lo);
}

// ---- stem: 3x3 stride-2 convolution of the 1-channel log-mel + BN + swish ------------------------------------------
// x [B][T][F], w [C][3 (mel)][3 (time)], y [B][To][Fo][C]; static "same" padding (pb before, pa after) on both axes.
// One thread per output position: the 9 input taps are loaded once and reused for all C (= 32) channels, the weights
// sit in LDS (broadcast reads), the C outputs leave as float4 stores.
__global__ __launch_bounds__(256) void stem_kernel(const float* x, const float* w, const float* scale, const float* shift,
                                                   float* y, int B, int T, int F, int To, int Fo, int C, int pb) {
  extern __shared__ float sw[];   // w [C][9] | scale [C] | shift [C]
  for (int i = threadIdx.x; i < C * 9; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < C; i += 256) {
    sw[C * 9 + i] = scale[i];
    sw[C * 10 + i] = shift[i];
  }
  __syncthreads();
  const long pos = blockIdx.x * 256L + threadIdx.x;
  const long npos = (long)B * To * Fo;
  if (pos >= npos) return;
  const int fo = (int)(pos % Fo);
  const long r = pos / Fo;
  const int to = (int)(r % To);
  const int b = (int)(r / To);
  const float* xb = x + (long)b * T * F;
  float tap[9];
#pragma unroll
  for (int kf = 0; kf < 3; ++kf) {
    const int f = fo * 2 - pb + kf;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int t = to * 2 - pb + kt;
      tap[kf * 3 + kt] = (f >= 0 && f < F && t >= 0 && t < T) ? xb[(long)t * F + f] : 0.f;
    }
  }
  float* yo = y + pos * C;
  for (int c = 0; c < C; c += 4) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 9; ++q) acc = fmaf(tap[q], sw[(c + j) * 9 + q], acc);
      v[j] = swishf(acc * sw[C * 9 + c + j] + sw[C * 10 + c + j]);
    }
    *(f32x4*)(yo + c) = v;
  }
}

// ---- depthwise k x k convolution (stride s) + BN + swish, and the squeeze-excite channel sums ---------------------
// x [B][T][F][C] -> y [B][To][Fo][C]; w [k (time)][k (mel)][C]; pool [B][C] += sum over positions of y.
struct DwP {
  const float* x; const float* w; const float* scale; const float* shift;
  float* y; float* pool;
  int T, F, To, Fo, C, k, s, pb, pos_per_block;
  float pool_scale;
};

// A thread owns one channel group (4 channels) and walks output rows `to`; inside a row it slides a K x K window of
// float4 along the mel axis, so every input element is loaded K / S times instead of K*K / S^2 and the K*K weights
// stay in registers.  The squeeze sums accumulate in registers and reach LDS once per thread.
template <int K, int S>
__global__ __launch_bounds__(256) void depthwise_kernel(DwP p) {
  extern __shared__ float spool[];   // [C]
  const int b = blockIdx.y;
  const int C4 = p.C >> 2;
  const int lanes = C4 < 256 ? (256 / C4) * C4 : 256;   // threads in use
  const int rstep = C4 < 256 ? 256 / C4 : 1;             // output rows handled per sweep
  for (int c = threadIdx.x; c < p.C; c += 256) spool[c] = 0.f;
  __syncthreads();
  const int row0 = blockIdx.x * p.pos_per_block;         // pos_per_block = output rows (time) per workgroup here
  const int row1 = min(p.To, row0 + p.pos_per_block);
  const float* xb = p.x + (long)b * p.T * p.F * p.C;
  float* yb = p.y + (long)b * p.To * p.Fo * p.C;
  if ((int)threadIdx.x < lanes) {
    const int cg0 = C4 < 256 ? threadIdx.x % C4 : threadIdx.x;
    const int rofs = C4 < 256 ? threadIdx.x / C4 : 0;
    for (int cg = cg0; cg < C4; cg += 256) {
      const int c = cg * 4;
      const f32x4 sc = *(const f32x4*)(p.scale + c), sh = *(const f32x4*)(p.shift + c);
      f32x4 w[K][K];
#pragma unroll
      for (int kt = 0; kt < K; ++kt)
#pragma unroll
        for (int kf = 0; kf < K; ++kf) w[kt][kf] = *(const f32x4*)(p.w + (long)(kt * K + kf) * p.C + c);
      f32x4 psum = {0.f, 0.f, 0.f, 0.f};
      for (int to = row0 + rofs; to < row1; to += rstep) {
        const float* rowp[K];
        bool rok[K];
#pragma unroll
        for (int kt = 0; kt < K; ++kt) {
          const int t = to * S - p.pb + kt;
          rok[kt] = t >= 0 && t < p.T;
          rowp[kt] = xb + (long)(rok[kt] ? t : 0) * p.F * p.C + c;
        }
        f32x4 win[K][K];
        auto col = [&](int f, int slot) {
          const bool fok = f >= 0 && f < p.F;
#pragma unroll
          for (int kt = 0; kt < K; ++kt) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (fok && rok[kt]) v = *(const f32x4*)(rowp[kt] + (long)f * p.C);
            win[kt][slot] = v;
          }
        };
#pragma unroll
        for (int kf = 0; kf < K; ++kf) col(-p.pb + kf, kf);
        for (int fo = 0; fo < p.Fo; ++fo) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < K; ++kt)
#pragma unroll
            for (int kf = 0; kf < K; ++kf) acc += win[kt][kf] * w[kt][kf];
          f32x4 v = acc * sc + sh;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = swishf(v[j]);
          psum += v;
          *(f32x4*)(yb + ((long)to * p.Fo + fo) * p.C + c) = v;
          // slide the window by S columns
#pragma unroll
          for (int kf = 0; kf + S < K; ++kf)
#pragma unroll
            for (int kt = 0; kt < K; ++kt) win[kt][kf] = win[kt][kf + S];
#pragma unroll
          for (int q = 0; q < S; ++q) col((fo + 1) * S - p.pb + K - S + q, K - S + q);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&spool[c + j], psum[j]);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) atomicAdd(p.pool + (long)b * p.C + c, spool[c] * p.pool_scale);
}

// The same layer for stride 1 and a NARROW mel axis (F = 2 ... 16: 19 of EfficientNet-B2's 23 depthwise layers): lane = channel group of 4, wave = a chunk of consecutive output rows (time); a thread keeps the K input
// rows x F columns of its window in registers and SLIDES ALONG TIME - one new input row (F 16-byte loads, requested a row
// ahead) per F output positions, instead of re-reading K rows per output row.  A wave instruction reads / writes 64 channel
// groups = 1 KiB contiguous.  The squeeze sums of a workgroup's four chunks meet in LDS: one atomic per channel and workgroup.
template <int K, int F, int HALVES>
__global__ __launch_bounds__(256) void depthwise_rows_kernel(DwP p, int lc) {
  // HALVES = 2 (F = 16): a thread owns HALF of the output columns and keeps the F / 2 + (K - 1) / 2 input
  // columns they read - the window fits the registers at 1.25-1.5x the loads
  constexpr int PAD = (K - 1) / 2, NO = F / HALVES, NW = HALVES == 1 ? F : NO + PAD;
  __shared__ __attribute__((aligned(16))) float spart[4][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, C4 = p.C >> 2;
  const int cg = blockIdx.x * 64 + lane;
  const int hv = HALVES == 1 ? 0 : (int)(blockIdx.z % HALVES);
  const int chunk = (int)(blockIdx.z / HALVES) * 4 + wave;
  const int o0 = hv * NO;                               // first output column of this thread
  const int cw0 = HALVES == 1 ? 0 : (hv == 0 ? 0 : o0 - PAD);   // first input column of its window
  const int to0 = chunk * lc, to1 = min(p.To, to0 + lc);
  const bool act = cg < C4 && to0 < to1;
  const int c = (act ? cg : 0) * 4;
  const float* xb = p.x + (long)b * p.T * F * p.C + c;
  float* yb = p.y + (long)b * p.To * F * p.C + c;
  f32x4 psum = {0.f, 0.f, 0.f, 0.f};
  if (act) {
    const f32x4 sc = *(const f32x4*)(p.scale + c), sh = *(const f32x4*)(p.shift + c);
    f32x4 w[K][K];
#pragma unroll
    for (int kt = 0; kt < K; ++kt)
#pragma unroll
      for (int kf = 0; kf < K; ++kf) w[kt][kf] = *(const f32x4*)(p.w + (long)(kt * K + kf) * p.C + c);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_row = [&](int t, f32x4 (&dst)[NW]) {   // input row t, columns cw0 .. cw0 + NW - 1 (zeros outside the image)
      const bool ok = t >= 0 && t < p.T;
      const float* rp = xb + ((long)(ok ? t : 0) * F + cw0) * p.C;
#pragma unroll
      for (int j = 0; j < NW; ++j) dst[j] = ok ? *(const f32x4*)(rp + (long)j * p.C) : zero4;
    };
    f32x4 win[K][NW], nxt[NW];
#pragma unroll
    for (int kt = 0; kt + 1 < K; ++kt) load_row(to0 - p.pb + kt, win[kt + 1]);   // rows of the first window but its last, shifted below
    load_row(to0 - p.pb + K - 1, nxt);
    for (int to = to0; to < to1; ++to) {
#pragma unroll
      for (int kt = 0; kt + 1 < K; ++kt)
#pragma unroll
        for (int j = 0; j < NW; ++j) win[kt][j] = win[kt + 1][j];
#pragma unroll
      for (int j = 0; j < NW; ++j) win[K - 1][j] = nxt[j];
      if (to + 1 < to1) load_row(to + 1 - p.pb + K - 1, nxt);   // in flight under this row's arithmetic
#pragma unroll
      for (int oo = 0; oo < NO; ++oo) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < K; ++kt)
#pragma unroll
          for (int kf = 0; kf < K; ++kf) {
            // input column of tap kf for output column o0 + oo ("same" padding of an odd kernel at stride 1), as a window
            // slot: HALVES == 1 or the first half: slot = column; second half: slot = column - (o0 - PAD)
            const int rel = oo - PAD + kf;                    // column - o0
            const int slot0 = rel;                            // first half / whole row: o0 == cw0 == 0
            const int slot1 = rel + PAD;                      // second half: cw0 = o0 - PAD
            if (HALVES == 1 || hv == 0) {
              if (slot0 >= 0 && slot0 < NW) acc += win[kt][slot0 < 0 ? 0 : (slot0 < NW ? slot0 : 0)] * w[kt][kf];
            } else {
              if (slot1 >= 0 && slot1 < NW && o0 + rel < F) acc += win[kt][slot1 < NW ? slot1 : 0] * w[kt][kf];
            }
          }
        f32x4 v = acc * sc + sh;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = swishf(v[j]);
        psum += v;
        *(f32x4*)(yb + ((long)to * F + o0 + oo) * p.C) = v;
      }
    }
  }
  *(f32x4*)(&spart[wave][lane * 4]) = psum;
  __syncthreads();
  if (wave == 0 && cg < C4) {
    const f32x4 t = (*(const f32x4*)(&spart[0][lane * 4]) + *(const f32x4*)(&spart[1][lane * 4])) +
                    (*(const f32x4*)(&spart[2][lane * 4]) + *(const f32x4*)(&spart[3][lane * 4]));
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(p.pool + (long)b * p.C + cg * 4 + j, t[j] * p.pool_scale);
  }
}

// The same sliding window for the FIRST stage's two 3 x 3 layers (F = 32 mel columns, 32 or 16 channels = 8 or 4 channel
// groups: far fewer than a wave's lanes): a wave covers a whole row - lane = (channel group cg = lane % C4W, column segment
// seg = lane / C4W of NO = F * C4W / 64 columns); a thread keeps its NO + 2 window columns of the 3 rows in registers and
// slides along time.  A wave instruction reads 64 / C4W separate runs of C4W * 16 contiguous bytes (whole 128-byte lines at 8
// channel groups).  223 -> 97 us (32 channels) / 133 -> 54 us (16) at 128 clips against the sliding-along-mel form.  (For the
// 5 x 5 layers at 8 mel columns x 288 channels - blocks of 8 channel groups over blockIdx.x, one output column per thread, 254
// registers - the form takes 165 us against 146 for the sliding-along-mel kernel: not routed.)
template <int K, int C4W, int NO>
__global__ __launch_bounds__(256) void depthwise_rowseg_kernel(DwP p, int lc) {
  constexpr int PAD = (K - 1) / 2, NSEG = 64 / C4W, F = NSEG * NO, NW = NO + K - 1;
  __shared__ __attribute__((aligned(16))) float spart[4][64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, C4 = p.C >> 2;
  const int cgl = lane % C4W, seg = lane / C4W;
  const int cg = blockIdx.x * C4W + cgl;                // blockIdx.x: blocks of C4W channel groups
  const int chunk = (int)blockIdx.z * 4 + wave;
  const int o0 = seg * NO;                              // first output column of this thread; its window starts at o0 - PAD
  const int to0 = chunk * lc, to1 = min(p.To, to0 + lc);
  const bool act = to0 < to1 && cg < C4;
  const int c = (cg < C4 ? cg : 0) * 4;
  const float* xb = p.x + (long)b * p.T * F * p.C + c;
  float* yb = p.y + (long)b * p.To * F * p.C + c;
  f32x4 psum = {0.f, 0.f, 0.f, 0.f};
  if (act) {
    const f32x4 sc = *(const f32x4*)(p.scale + c), sh = *(const f32x4*)(p.shift + c);
    f32x4 w[K][K];
#pragma unroll
    for (int kt = 0; kt < K; ++kt)
#pragma unroll
      for (int kf = 0; kf < K; ++kf) w[kt][kf] = *(const f32x4*)(p.w + (long)(kt * K + kf) * p.C + c);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_row = [&](int t, f32x4 (&dst)[NW]) {   // input row t, columns o0 - PAD .. o0 + NO - 1 + PAD (zeros outside the image)
      const bool ok = t >= 0 && t < p.T;
      const float* rp = xb + ((long)(ok ? t : 0) * F + o0 - PAD) * p.C;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int col = o0 - PAD + j;
        dst[j] = (ok && col >= 0 && col < F) ? *(const f32x4*)(rp + (long)j * p.C) : zero4;
      }
    };
    f32x4 win[K][NW], nxt[NW];
#pragma unroll
    for (int kt = 0; kt + 1 < K; ++kt) load_row(to0 - p.pb + kt, win[kt + 1]);   // rows of the first window but its last, shifted below
    load_row(to0 - p.pb + K - 1, nxt);
    for (int to = to0; to < to1; ++to) {
#pragma unroll
      for (int kt = 0; kt + 1 < K; ++kt)
#pragma unroll
        for (int j = 0; j < NW; ++j) win[kt][j] = win[kt + 1][j];
#pragma unroll
      for (int j = 0; j < NW; ++j) win[K - 1][j] = nxt[j];
      if (to + 1 < to1) load_row(to + 1 - p.pb + K - 1, nxt);   // in flight under this row's arithmetic
#pragma unroll
      for (int oo = 0; oo < NO; ++oo) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < K; ++kt)
#pragma unroll
          for (int kf = 0; kf < K; ++kf) acc += win[kt][oo + kf] * w[kt][kf];   // the order of depthwise_rows_kernel
        f32x4 v = acc * sc + sh;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = swishf(v[j]);
        psum += v;
        *(f32x4*)(yb + ((long)to * F + o0 + oo) * p.C) = v;
      }
    }
  }
  *(f32x4*)(&spart[wave][lane * 4]) = psum;
  __syncthreads();
  if (threadIdx.x < C4W && blockIdx.x * C4W + threadIdx.x < C4) {
    // squeeze sums: the workgroup's 4 row chunks x NSEG column segments of channel group threadIdx.x of this block
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int wv = 0; wv < 4; ++wv)
      for (int sg = 0; sg < NSEG; ++sg) t += *(const f32x4*)(&spart[wv][(sg * C4W + (int)threadIdx.x) * 4]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      atomicAdd(p.pool + (long)b * p.C + (blockIdx.x * C4W + threadIdx.x) * 4 + j, t[j] * p.pool_scale);
  }
}

// ---- squeeze-excite gate: g[b][c] = sigmoid(W2 swish(W1 mean[b] + b1) + b2) -----------------------------------------
// pool [B][C] sums, w1 [S][C], w2 [C][S]; one workgroup per clip.
__global__ __launch_bounds__(256) void se_gate_kernel(const float* pool, float inv_count, const float* w1, const float* b1,
                                                      const float* w2, const float* b2, float* gate, int C, int S) {
  extern __shared__ float sm[];   // mean [C] | squeezed [S]
  float* mean = sm;
  float* sq = sm + C;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) mean[c] = pool[(long)b * C + c] * inv_count;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = wave; s < S; s += 4) {
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a = fmaf(w1[(long)s * C + c], mean[c], a);
    a = wave_sum(a);
    if (lane == 0) sq[s] = swishf(a + b1[s]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = b2[c];
    for (int s = 0; s < S; ++s) a = fmaf(w2[(long)c * S + s], sq[s], a);
    gate[(long)b * C + c] = ac_sigmoid_fast(a);
  }
}

// ---- the same gate with the second weight matrix TRANSPOSED (w2t [S][C]) and both phases bandwidth-shaped -----------
// One workgroup per clip.  Phase 1: a wave owns squeezed units s, s + 4, ...: 16-byte loads of w1[s] across the lanes
// (two rows in flight), DPP wave sum.  Phase 2: a thread owns channel quads; for every s one coalesced 16-byte load of
// w2t[s] (four in flight), sq[s] broadcast from LDS.  se_gate_kernel above walks w2 [C][S] with one dependent scalar
// load per (channel, s) - an S-long latency chain per thread - and the two-GEMM form it was replaced by costs two ~20 us
// launches per block for 128 x S x C products of a few MFLOP.
__global__ __launch_bounds__(256) void se_gate_t_kernel(const float* pool, float inv_count, const float* w1, const float* b1,
                                                        const float* w2t, const float* b2, float* gate, int C, int S) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // mean [C] | squeezed [S]
  float* mean = sm;
  float* sq = sm + C;
  const int b = blockIdx.x;
  const int C4 = C >> 2;
  for (int c4 = threadIdx.x; c4 < C4; c4 += 256)
    ((f32x4*)mean)[c4] = ((const f32x4*)(pool + (long)b * C))[c4] * inv_count;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // four squeezed units per wave and pass (s = wave + 4 j): their weight rows are all requested before the first FMA
  for (int s0 = wave; s0 < S; s0 += 16) {
    const f32x4* r[4];
    f32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = (const f32x4*)(w1 + (long)min(s0 + 4 * j, S - 1) * C);
      a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c4 = lane; c4 < C4; c4 += 64) {
      const f32x4 m = ((const f32x4*)mean)[c4];
      f32x4 wv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = r[j][c4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += wv[j] * m;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = wave_sum((a[j][0] + a[j][1]) + (a[j][2] + a[j][3]));
      if (lane == 0 && s0 + 4 * j < S) sq[s0 + 4 * j] = swishf(t + b1[s0 + 4 * j]);
    }
  }
  __syncthreads();
  for (int c4 = threadIdx.x; c4 < C4; c4 += 256) {
    f32x4 a = ((const f32x4*)b2)[c4];
    const f32x4* col = (const f32x4*)w2t + c4;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = col[(long)(s + j) * C4];
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v[j] * sq[s + j];
    }
    for (; s < S; ++s) a += col[(long)s * C4] * sq[s];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ac_sigmoid_fast(a[j]);
    ((f32x4*)(gate + (long)b * C))[c4] = a;
  }
}

// ---- 1x1 convolution = thin GEMM y[M][N] = act((x .* gate) W^T + b) (+ y), M = clips * positions in the millions,
// K, N <= 2112.  HBM-bound: every wave owns 32 rows and reads its A fragments straight from global memory exactly
// once per pass (lane l: 4 consecutive k of row l & 31 at k offset 4 * (l >> 5) of every 8-k group - one 16-byte load
// feeds 4 MFMAs per output tile), the weights (<= 3 MB, L2-resident) the same way; up to 4 output tiles of 32 columns
// are kept in accumulators per pass.  No LDS, no barriers; the squeeze-excite gate multiplies A on the way in.
struct PwP {
  const float* x; const float* w; const float* bias; float* y;
  const float* gate; int gate_rows;
  long M; int N, K, act;
  float beta;
};

template <int NT>
__global__ __launch_bounds__(256) void pointwise_kernel(PwP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m0 = ((long)blockIdx.x * 4 + wave) * 32;
  if (m0 >= p.M) return;
  const long row = min(m0 + (lane & 31), p.M - 1);
  const int koff = 4 * (lane >> 5);
  const float* a = p.x + row * p.K + koff;
  const float* g = p.gate ? p.gate + (row / p.gate_rows) * p.K + koff : nullptr;
  const int ngroups = p.K >> 3;
  for (int n0 = 0; n0 < p.N; n0 += 32 * NT) {
    f32x16 acc[NT];
    const float* b[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      b[t] = p.w + (long)min(n0 + t * 32 + (lane & 31), p.N - 1) * p.K + koff;
    }
#pragma unroll 2
    for (int gq = 0; gq < ngroups; ++gq) {
      f32x4 va = *(const f32x4*)(a + 8 * gq);
      if (g) va *= *(const f32x4*)(g + 8 * gq);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 vb = *(const f32x4*)(b[t] + 8 * gq);
        // weights as the MFMA's row operand, activations as its column operand: the accumulator is the TRANSPOSED
        // tile, lane = output row m, registers 4g..4g+3 = four consecutive channels -> 16-byte stores
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t] = mfma32(vb[j], va[j], acc[t]);
      }
    }
    const long m = m0 + (lane & 31);
    if (m < p.M) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int gr = 0; gr < 4; ++gr) {
          const int n = n0 + t * 32 + 8 * gr + 4 * (lane >> 5);
          if (n >= p.N) continue;                       // N % 4 == 0: a quad is inside or outside as a whole
          f32x4 v = {acc[t][4 * gr], acc[t][4 * gr + 1], acc[t][4 * gr + 2], acc[t][4 * gr + 3]};
          if (p.bias) v += *(const f32x4*)(p.bias + n);
          if (p.act == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = swishf(v[j]);
          } else if (p.act == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          float* c = p.y + m * p.N + n;
          if (p.beta != 0.f) v += p.beta * *(const f32x4*)c;
          *(f32x4*)c = v;
        }
      }
    }
  }
}

}  // namespace

extern "C" {

int ac_pointwise_conv(const float* x, const float* w, const float* bias, float* y, long M, int N, int K, int act, float beta,
                      const float* gate, int gate_rows, void* stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 3) || act < 0 || act > 2 ||
      (gate && gate_rows <= 0) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15) ||
      (bias && ((uintptr_t)bias & 15)))
    return AC_ERR_ARG;
  PwP p;
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.gate = gate; p.gate_rows = gate_rows;
  p.M = M; p.N = N; p.K = K; p.act = act; p.beta = beta;
  const long blocks = (M + 127) / 128;
  if (blocks > 2147483647L) return AC_ERR_ARG;
  dim3 grid((unsigned)blocks);
  hipStream_t s = (hipStream_t)stream;
  if (N <= 32) hipLaunchKernelGGL(pointwise_kernel<1>, grid, dim3(256), 0, s, p);
  else if (N <= 64) hipLaunchKernelGGL(pointwise_kernel<2>, grid, dim3(256), 0, s, p);
  else if (N <= 96) hipLaunchKernelGGL(pointwise_kernel<3>, grid, dim3(256), 0, s, p);
  else if (N <= 128) hipLaunchKernelGGL(pointwise_kernel<4>, grid, dim3(256), 0, s, p);
  else if (N <= 160 || (N > 256 && N <= 320)) hipLaunchKernelGGL(pointwise_kernel<5>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(pointwise_kernel<6>, grid, dim3(256), 0, s, p);
  return ac_check_launch();
}

int ac_top_db_clamp(float* x, long n, float top_db, float* scratch, int scratch_floats, void* stream) {
  if (!x || !scratch || n <= 0 || scratch_floats < 1) return AC_ERR_ARG;
  int nparts = (int)((n + 255) / 256);
  if (nparts > scratch_floats) nparts = scratch_floats;
  if (nparts > 1024) nparts = 1024;
  hipLaunchKernelGGL(block_max_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, x, n, scratch);
  long g = (n + 255) / 256;
  if (g > 65535) g = 65535;
  hipLaunchKernelGGL(clamp_top_db_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, scratch, nparts, top_db);
  return ac_check_launch();
}

int ac_effnet_stem(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int T, int F,
                   int C, int pad_before, int pad_after, void* stream) {
  if (!x || !w || !scale || !shift || !y || B <= 0 || T <= 0 || F <= 0 || C <= 0) return AC_ERR_ARG;
  const int To = (T + pad_before + pad_after - 3) / 2 + 1, Fo = (F + pad_before + pad_after - 3) / 2 + 1;
  if (To <= 0 || Fo <= 0) return AC_ERR_ARG;
  if (C & 3) return AC_ERR_ARG;
  const long n = (long)B * To * Fo;
  hipLaunchKernelGGL(stem_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)C * 11 * sizeof(float),
                     (hipStream_t)stream, x, w, scale, shift, y, B, T, F, To, Fo, C, pad_before);
  return ac_check_launch();
}

int ac_effnet_depthwise(const float* x, const float* w, const float* scale, const float* shift, float* y, float* pool,
                        float pool_scale, int B, int T, int F, int C, int k, int stride, int pad_before, int pad_after,
                        void* stream) {
  if (!x || !w || !scale || !shift || !y || !pool || B <= 0 || T <= 0 || F <= 0 || C <= 0 || (C & 3) ||
      (k != 3 && k != 5) || (stride != 1 && stride != 2) || C > 8192)
    return AC_ERR_ARG;
  DwP p;
  p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.y = y; p.pool = pool;
  p.T = T; p.F = F; p.C = C; p.k = k; p.s = stride; p.pb = pad_before; p.pool_scale = pool_scale;
  p.To = (T + pad_before + pad_after - k) / stride + 1;
  p.Fo = (F + pad_before + pad_after - k) / stride + 1;
  if (p.To <= 0 || p.Fo <= 0) return AC_ERR_ARG;
  // output rows per workgroup: every thread (one channel group) should walk >= 2 rows when there are enough of them
  const int C4 = C / 4;
  const int rstep = C4 < 256 ? 256 / C4 : 1;
  static int mult = 0;
  if (!mult) {
    const char* e = getenv("AUDIOCAPTION_DW_ROWS");
    mult = e ? atoi(e) : 4;   // measured at 128 clips: 2 -> 4 rows per thread -6...-16 us on the 63 x 4 and 32 x 2 stages, 8 and 16 slower
    if (mult < 1) mult = 1;
  }
  // stride 1, "same" padding, a narrow mel axis: the rows-in-registers form (AUDIOCAPTION_DW_ROWS_KERNEL=0: the form below)
  static const bool rows_kernel = !(getenv("AUDIOCAPTION_DW_ROWS_KERNEL") && !strcmp(getenv("AUDIOCAPTION_DW_ROWS_KERNEL"), "0"));
  if (rows_kernel && stride == 1 && k == 3 && pad_before == 1 && p.Fo == F && p.To == T && F == 32 && (C == 32 || C == 16)) {
    const int lc = 16;
    dim3 g(1, B, ((p.To + lc - 1) / lc + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (C == 32) hipLaunchKernelGGL((depthwise_rowseg_kernel<3, 8, 4>), g, dim3(256), 0, st, p, lc);
    else hipLaunchKernelGGL((depthwise_rowseg_kernel<3, 4, 2>), g, dim3(256), 0, st, p, lc);
    return ac_check_launch();
  }
  if (rows_kernel && stride == 1 && pad_before == (k - 1) / 2 && p.Fo == F && p.To == T &&
      (F == 2 || F == 4 || ((F == 8 || F == 16) && k == 3))) {
    // (k = 5 at F = 8, 288 channels = 72 channel groups: a 6-column window + 25 weights do not fit 256 registers; with the
    // weights in LDS the form takes 160 us, as depthwise_rowseg_kernel<5, 8, 1> 165 us, against 141-146 for the one below)
    const int lc = T >= 48 ? 16 : 8;                        // output rows per wave: 4 (2) halo rows per 16 / 8
    const int halves = F == 16 ? 2 : 1;
    dim3 g((C4 + 63) / 64, B, (((p.To + lc - 1) / lc + 3) / 4) * halves);
    hipStream_t st = (hipStream_t)stream;
    if (k == 5 && F == 2) hipLaunchKernelGGL((depthwise_rows_kernel<5, 2, 1>), g, dim3(256), 0, st, p, lc);
    else if (k == 5) hipLaunchKernelGGL((depthwise_rows_kernel<5, 4, 1>), g, dim3(256), 0, st, p, lc);
    else if (F == 2) hipLaunchKernelGGL((depthwise_rows_kernel<3, 2, 1>), g, dim3(256), 0, st, p, lc);
    else if (F == 4) hipLaunchKernelGGL((depthwise_rows_kernel<3, 4, 1>), g, dim3(256), 0, st, p, lc);
    else if (F == 8) hipLaunchKernelGGL((depthwise_rows_kernel<3, 8, 1>), g, dim3(256), 0, st, p, lc);
    else hipLaunchKernelGGL((depthwise_rows_kernel<3, 16, 2>), g, dim3(256), 0, st, p, lc);
    return ac_check_launch();
  }
  int rpb = rstep * mult;
  if (rpb > p.To) rpb = p.To;
  p.pos_per_block = rpb;
  dim3 grid((p.To + rpb - 1) / rpb, B);
  const size_t lds = (size_t)C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (k == 3 && stride == 1) hipLaunchKernelGGL((depthwise_kernel<3, 1>), grid, dim3(256), lds, st, p);
  else if (k == 3) hipLaunchKernelGGL((depthwise_kernel<3, 2>), grid, dim3(256), lds, st, p);
  else if (stride == 1) hipLaunchKernelGGL((depthwise_kernel<5, 1>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((depthwise_kernel<5, 2>), grid, dim3(256), lds, st, p);
  return ac_check_launch();
}

int ac_effnet_se_gate(const float* pool, float inv_count, const float* w1, const float* b1, const float* w2,
                      const float* b2, float* gate, int B, int C, int S, void* stream) {
  if (!pool || !w1 || !b1 || !w2 || !b2 || !gate || B <= 0 || C <= 0 || S <= 0 || C + S > 12000) return AC_ERR_ARG;
  hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(256), (size_t)(C + S) * sizeof(float), (hipStream_t)stream, pool,
                     inv_count, w1, b1, w2, b2, gate, C, S);
  return ac_check_launch();
}

int ac_effnet_se_gate_t(const float* pool, float inv_count, const float* w1, const float* b1, const float* w2t,
                        const float* b2, float* gate, int B, int C, int S, void* stream) {
  if (!pool || !w1 || !b1 || !w2t || !b2 || !gate || B <= 0 || C <= 0 || S <= 0 || (C & 3) || C + S > 12000 ||
      ((uintptr_t)pool & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2t & 15) || ((uintptr_t)b2 & 15) || ((uintptr_t)gate & 15))
    return AC_ERR_ARG;
  hipLaunchKernelGGL(se_gate_t_kernel, dim3(B), dim3(256), (size_t)(C + S) * sizeof(float), (hipStream_t)stream, pool,
                     inv_count, w1, b1, w2t, b2, gate, C, S);
  return ac_check_launch();
}

}  // extern "C"
