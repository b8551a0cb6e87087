// Log-mel front-end for the Cnn14 / EfficientNet-B2 audio encoders on gfx950.
//
// Replaces, in ONE HBM pass over the waveform, what the reference computes with
//   torchaudio MelSpectrogram (reflect-padded, centred, periodic-Hann STFT power -> mel matmul)
//   + AmplitudeToDB + the transposes + bn0          (reference cnn_encoder.py:418-429)
// and writes the result directly in the row-padded channels-last layout the conv stack reads.
//
// One 64-lane wave owns one STFT frame: the N real samples are packed into an N/2-point complex
// sequence, transformed by three Stockham radix passes held entirely in registers + LDS
// (N=1024: 8x8x8, N=512: 4x8x8), unpacked to the N/2+1 one-sided bins, squared, reduced over each
// mel filter's support (the filterbank is triangular: <= ~70 non-zero bins per filter) and
// converted to dB.  HBM traffic is the waveform read (frames overlap 3.2x, served by L2) plus the
// 64 x T output; nothing of the (B, 513, T) spectrogram ever reaches memory.
#include "ac_common.h"

namespace {

struct LogmelParams {
  const float* wav;
  int B, L;
  int hop, T, rows_per_clip;
  const float* window;    // [N]
  const float2* twiddle;  // [N]  exp(-2*pi*i*n/N)
  const float* melfb;     // [N/2+1][64]
  const int* mel_lo;      // [64] first non-zero bin of each filter
  const int* mel_hi;      // [64] last non-zero bin (inclusive)
  const float* scale;     // [64] or null (bn0 folded: y = x*scale + shift)
  const float* shift;
  float* out;
  long stride_b, stride_t, stride_m;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_negi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)

template <int R>
__device__ __forceinline__ void dft(float2 (&u)[R]);

template <>
__device__ __forceinline__ void dft<4>(float2 (&u)[4]) {
  float2 a0 = cadd(u[0], u[2]), a1 = csub(u[0], u[2]);
  float2 a2 = cadd(u[1], u[3]), a3 = mul_negi(csub(u[1], u[3]));
  u[0] = cadd(a0, a2);
  u[2] = csub(a0, a2);
  u[1] = cadd(a1, a3);
  u[3] = csub(a1, a3);
}

template <>
__device__ __forceinline__ void dft<8>(float2 (&u)[8]) {
  const float h = 0.70710678118654752440f;
  float2 a0 = cadd(u[0], u[4]), a4 = csub(u[0], u[4]);
  float2 a1 = cadd(u[1], u[5]), a5 = csub(u[1], u[5]);
  float2 a2 = cadd(u[2], u[6]), a6 = csub(u[2], u[6]);
  float2 a3 = cadd(u[3], u[7]), a7 = csub(u[3], u[7]);
  a5 = make_float2(h * (a5.x + a5.y), h * (a5.y - a5.x));    // * (1 - i)/sqrt2
  a6 = mul_negi(a6);                                          // * (-i)
  a7 = make_float2(h * (a7.y - a7.x), -h * (a7.x + a7.y));   // * (-1 - i)/sqrt2
  float2 b0 = cadd(a0, a2), b2 = csub(a0, a2);
  float2 b1 = cadd(a1, a3), b3 = mul_negi(csub(a1, a3));
  float2 b4 = cadd(a4, a6), b6 = csub(a4, a6);
  float2 b5 = cadd(a5, a7), b7 = mul_negi(csub(a5, a7));
  u[0] = cadd(b0, b1);
  u[4] = csub(b0, b1);
  u[2] = cadd(b2, b3);
  u[6] = csub(b2, b3);
  u[1] = cadd(b4, b5);
  u[5] = csub(b4, b5);
  u[3] = cadd(b6, b7);
  u[7] = csub(b6, b7);
}

// One Stockham pass of radix R over an M-point sequence; Ns = product of the radices already done.
// Butterfly j (0 <= j < M/R) reads in[j + r*M/R], applies the twiddle exp(-2 pi i r k/(Ns R)) with
// k = j mod Ns, and writes X[r] to out[(j - k) * R + k + r * Ns].  The R - 1 twiddles of a lane are the same for every
// frame (j = lane): they live in registers (tw[r - 1]).
template <int M, int R, int Ns>
__device__ __forceinline__ void stockham_store(float2 (&u)[R], float2* out, const float2 (&tw)[R - 1], int j) {
  const int k = j % Ns;
  if (Ns > 1) {
#pragma unroll
    for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[r - 1]);
  }
  dft<R>(u);
  const int j0 = (j - k) * R + k;
#pragma unroll
  for (int r = 0; r < R; ++r) out[j0 + r * Ns] = u[r];
}

template <int M, int R, int Ns>
__device__ __forceinline__ void stockham_pass(const float2* in, float2* out, const float2 (&tw)[R - 1], int lane) {
  if (lane < M / R) {
    float2 u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = in[lane + r * (M / R)];
    stockham_store<M, R, Ns>(u, out, tw, lane);
  }
}

constexpr int MEL_CSR = 2048;   // non-zero filterbank weights of all 64 filters (slaney 50-14000 Hz at 1024: 1055; htk at 512: 546)

template <int N>
__global__ __launch_bounds__(256, 4) void logmel_kernel(LogmelParams p) {
  constexpr int M = N / 2;
  constexpr int R0 = (N == 1024) ? 8 : 4;  // first radix; the remaining two passes are radix 8
  constexpr int WAVES = 4;
  constexpr int NU = M / 64 + 1;           // bins a lane unpacks: lane + 64 i, i < NU (the last one only for lane 0)
  __shared__ float s_w[MEL_CSR];           // filterbank weights, filter by filter (lane l: s_w[woff .. woff + hi - lo])
  __shared__ float2 s_buf[WAVES][2][M];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- per-lane constants, the same for every frame: window samples, twiddles of the three passes and of the unpacking,
  // this lane's filter (support and weights).  Nothing below the frame loop reads a table from global memory. ----
  float win[2 * R0];
#pragma unroll
  for (int r = 0; r < R0; ++r) {
    const int n = 2 * (lane + r * (M / R0));
    win[2 * r] = p.window[n];
    win[2 * r + 1] = p.window[n + 1];
  }
  float2 tw1[7], tw2[7], twu[NU];
  {
    const int k1 = lane % R0, k2 = lane % (R0 * 8);
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      tw1[r - 1] = p.twiddle[(r * k1 * (M / (R0 * 8))) * (N / M)];
      tw2[r - 1] = p.twiddle[(r * k2 * (M / (R0 * 8 * 8))) * (N / M)];
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int k = lane + 64 * i;
      twu[i] = k < M ? p.twiddle[k] : make_float2(-1.f, 0.f);
    }
  }
  const float2 tw0[R0 - 1] = {};   // first pass: Ns = 1, no twiddles
  const int mlo = p.mel_lo[lane], mhi = p.mel_hi[lane];
  int woff = 0;
  {
    // exclusive prefix sum of the support lengths over the 64 lanes (filters)
    const int cnt = mhi >= mlo ? mhi - mlo + 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    woff = incl - cnt;
    const int total_w = __shfl(incl, 63, 64);
    if (wave == 0 && total_w <= MEL_CSR)
      for (int k = mlo; k <= mhi; ++k) s_w[woff + k - mlo] = p.melfb[k * 64 + lane];
    if (total_w > MEL_CSR) woff = -1;   // a filterbank denser than the table: weights straight from memory
  }
  const float bn_s = p.scale ? p.scale[lane] : 1.f, bn_t = p.scale ? p.shift[lane] : 0.f;
  __syncthreads();

  float2* bufA = s_buf[wave][0];
  float2* bufB = s_buf[wave][1];
  float* pw = (float*)bufB;   // the power spectrum (M + 1 floats) reuses the ping-pong half that is free after the last pass
  // A frame belongs to ONE wave: its passes only need the wave's own LDS writes to have landed (the lanes run in lock
  // step), not a workgroup barrier - the four waves of a workgroup drift apart and hide each other's latencies.
#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  const long total = (long)p.B * p.rows_per_clip;

  // the samples of a frame (reflect padding at both ends), two per radix leg; requested one frame AHEAD of their use
  auto frame_of = [&](long slot, int& b, int& t, bool& in_range) {
    in_range = slot < total;
    b = in_range ? (int)(slot / p.rows_per_clip) : 0;
    t = in_range ? (int)(slot % p.rows_per_clip) : 0;
    return in_range && t < p.T;
  };
  auto samples = [&](int b, int t, float2 (&u)[R0]) {
    const float* x = p.wav + (long)b * p.L;
    const int g0 = t * p.hop - N / 2;
#pragma unroll
    for (int r = 0; r < R0; ++r) {
      const int n = 2 * (lane + r * (M / R0));
      int g = g0 + n, g1 = g0 + n + 1;
      g = g < 0 ? -g : (g >= p.L ? 2 * (p.L - 1) - g : g);
      g1 = g1 < 0 ? -g1 : (g1 >= p.L ? 2 * (p.L - 1) - g1 : g1);
      u[r] = make_float2(x[g], x[g1]);
    }
  };
  float2 nxt[R0];
  int nb, nt;
  bool n_in;
  const long step = (long)gridDim.x * WAVES;
  long slot = (long)blockIdx.x * WAVES + wave;
  bool n_act = frame_of(slot, nb, nt, n_in);
  if (n_act) samples(nb, nt, nxt);

  for (long slot0 = (long)blockIdx.x * WAVES; slot0 < total; slot0 += step) {
    const int b = nb, t = nt;
    const bool in_range = n_in, active = n_act;
    float2 u[R0];
#pragma unroll
    for (int r = 0; r < R0; ++r) u[r] = make_float2(nxt[r].x * win[2 * r], nxt[r].y * win[2 * r + 1]);
    slot += step;
    n_act = frame_of(slot, nb, nt, n_in);
    if (n_act) samples(nb, nt, nxt);   // in flight under this frame's transform

    if (active) stockham_store<M, R0, 1>(u, bufA, tw0, lane);   // pass 0 on the windowed samples
    WAVE_LDS_SYNC();
    if (active) stockham_pass<M, 8, R0>(bufA, bufB, tw1, lane);
    WAVE_LDS_SYNC();
    if (active) stockham_pass<M, 8, R0 * 8>(bufB, bufA, tw2, lane);
    WAVE_LDS_SYNC();
    if (active) {
      // ---- unpack the packed-real transform to bins 0..M and take the power ----
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int k = lane + 64 * i;
        if (k <= M) {
          const float2 zk = bufA[k & (M - 1)];
          const float2 zc = bufA[(M - k) & (M - 1)];
          const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
          const float2 o = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
          const float2 xk = cadd(e, cmul(twu[i], o));
          pw[k] = xk.x * xk.x + xk.y * xk.y;
        }
      }
    }
    WAVE_LDS_SYNC();
    if (in_range) {
      float v = 0.f;
      if (active) {
        float s = 0.f;
        if (woff >= 0) {
          const float* wl = s_w + woff - mlo;
          for (int k = mlo; k <= mhi; ++k) s = fmaf(pw[k], wl[k], s);
        } else {
          for (int k = mlo; k <= mhi; ++k) s = fmaf(pw[k], p.melfb[k * 64 + lane], s);
        }
        v = 10.0f * log10f(fmaxf(s, 1e-10f));
        if (p.scale) v = fmaf(v, bn_s, bn_t);
      }
      p.out[b * p.stride_b + t * p.stride_t + lane * p.stride_m] = v;
    }
    WAVE_LDS_SYNC();
  }
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_logmel(const float* wav, int B, int L, int n_fft, int hop, const float* window,
                         const float* twiddle, const float* melfb, const int* mel_lo, const int* mel_hi,
                         const float* scale, const float* shift, float* out, int rows_per_clip,
                         long stride_b, long stride_t, long stride_m, void* stream) {
  if (!wav || !out || B <= 0 || L <= n_fft / 2 || hop <= 0) return AC_ERR_ARG;
  if (n_fft != 1024 && n_fft != 512) return AC_ERR_ARG;
  LogmelParams p;
  p.wav = wav; p.B = B; p.L = L; p.hop = hop; p.T = L / hop + 1;
  p.rows_per_clip = rows_per_clip;
  if (rows_per_clip < p.T) return AC_ERR_ARG;
  p.window = window; p.twiddle = (const float2*)twiddle; p.melfb = melfb;
  p.mel_lo = mel_lo; p.mel_hi = mel_hi; p.scale = scale; p.shift = shift; p.out = out;
  p.stride_b = stride_b; p.stride_t = stride_t; p.stride_m = stride_m;
  const long total = (long)B * rows_per_clip;
  long blocks = (total + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipStream_t s = (hipStream_t)stream;
  if (n_fft == 1024)
    hipLaunchKernelGGL(logmel_kernel<1024>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(logmel_kernel<512>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  return ac_check_launch();
}
