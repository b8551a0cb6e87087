// Length-aware bidirectional GRU layer + masked mean pooling on gfx950.
//
// Replaces, per layer, what the reference gets from pack_padded_sequence -> nn.GRU ->
// pad_packed_sequence (rnn_encoder.py:34-49, model_util.py:10-27): the input projections of all time
// steps are one MFMA GEMM (ac_linear, N = 2 directions x 3 gates x H); this file is the recurrence.
//
// Clips never interact inside the recurrence, so each (clip, direction) pair is ONE persistent
// workgroup that walks its own valid steps (forward: 0..len-1, reverse: len-1..0 - i.e. the reverse
// direction starts at the clip's own last valid frame, exactly what packing does) with the hidden
// state in LDS; no inter-workgroup synchronisation exists anywhere.  The 768x256 recurrent matrix is
// streamed from L2 every step in a packed copy [k/4][column][4] (ac_gru_pack_whh): thread n reads 4
// consecutive k of its column with one 16-byte load, consecutive threads consecutive 16-byte words.  Steps t >= len are written as zeros (pad_packed_sequence semantics).
#include "ac_common.h"

namespace {

struct GruParams {
  const float* gx;     // [B][T][2][3H]  x W_ih^T + b_ih, gate order r, z, n
  const float* whhT;   // [2][H/4][3H][4] W_hh packed by ac_gru_pack_whh
  const float* bhh;    // [2][3H]
  const int* lens;     // [B]
  float* out;          // [B][T][2H]
  int B, T;
};

constexpr int H = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// The 786 KB recurrent matrix cannot stay on a CU, and re-streaming all of it from L2 every step runs at the CU's
// L1 fill rate (~50 B/clk: 6.7 us per step).  So the part that fits stays resident for the whole sequence: the first
// GRU_KREG k of every column in registers (24 float4 per thread), the next GRU_KLDS k in LDS (144 KB); only the last
// 44 % is streamed per step.  The accumulation order (k ascending) is unchanged.
constexpr int GRU_KREG = 96, GRU_KLDS = 48;
__global__ __launch_bounds__(768) void gru_layer_kernel(GruParams p) {
  __shared__ __attribute__((aligned(16))) float sh[H];
  __shared__ float sg[3 * H];
  extern __shared__ __attribute__((aligned(16))) float swl[];   // [GRU_KLDS / 4][3H][4]
  const int n = threadIdx.x;
  const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
  int len = p.lens[b];
  len = len < 0 ? 0 : (len > p.T ? p.T : len);
  const float* W = p.whhT + (size_t)dir * H * 3 * H + (size_t)n * 4;
  const float bias = p.bhh[dir * 3 * H + n];
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
    // the input-side gate pre-activations of this step are requested before the matrix-vector product
    const float* gxp = p.gx + (((size_t)b * p.T + t) * 2 + dir) * 3 * H;
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
      const float4 wv = *(const float4*)(W + (size_t)(k >> 2) * 3 * H * 4);   // k..k+3 of this thread's column
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
      const float c = tanhf(gn + r * sg[2 * H + n]);
      const float hn = (1.0f - z) * c + z * sh[n];
      sh[n] = hn;
      p.out[((size_t)b * p.T + t) * 2 * H + dir * H + n] = hn;
    }
    __syncthreads();
  }
  if (n < H)
    for (int t = len; t < p.T; ++t) p.out[((size_t)b * p.T + t) * 2 * H + dir * H + n] = 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// The recurrence of ONE (clip, direction) split over TWO workgroups (two CUs).  A CU cannot hold the 786 KB recurrent
// matrix (the single-workgroup kernel above re-streams 44 % of it from L2 every step, which is its whole critical
// path), but half of it fits in the registers of one CU: workgroup `half` owns the hidden units [128 half, 128 half +
// 128) - the 384 gate rows r_j, z_j, n_j of those units, two threads per row with 128 k each in VGPRs - and never reads
// a weight again after the prologue.  Per step the two halves trade their 128 new hidden values through L2 as 8-byte
// {tag, value} granules (one relaxed agent-scope atomic store each, the value IS the flag; the consumer's 128 polling
// threads re-read their own granule until the tag is this step's - cdna_hip_programming.md section 6 Guideline 16, form
// R2: no fences, correct for any workgroup -> XCD placement).  Slots alternate with the step parity: a producer cannot
// reach step s + 2 before it has consumed the partner's step s + 1, which the partner published after consuming the
// producer's step s.
// Partners are paired by START ORDER, not by block index (HIP promises no dispatch order): every workgroup draws a
// ticket, tickets 2p and 2p + 1 form pair p.  At any time at most one started workgroup is without its partner, and that
// partner is the next workgroup to start - all other resident workgroups are complete pairs that finish on their own,
// so the scheme cannot deadlock however the blocks are dispatched; a spin that outlasts GRU_SPIN_TICKS all the same
// (another process holding the GPU ...) raises the error word instead of hanging.
// The 64-clip bench batch uses all 256 CUs this way (the single-workgroup kernel only 128): 175 -> ~60 us per layer.
// ---------------------------------------------------------------------------------------------------------------------
struct GruSplitParams {
  const float* gx;     // [B][T][2][3H]
  const float* whh;    // [2][3H][H]   nn.GRU weight_hh of both directions, UNPACKED
  const float* bhh;    // [2][3H]
  const int* lens;     // [B]
  float* out;          // [B][T][2H]
  unsigned long long* xch;   // [2B pairs][2 halves][2 slots][128] granules, zeroed before every launch
  unsigned* ticket;          // zeroed before every launch
  unsigned* error;           // set to 1 when a partner never showed up
  int B, T;
};
constexpr int HH = H / 2;                      // hidden units per workgroup
constexpr long long GRU_SPIN_TICKS = 200000000;   // 2 s of the 100 MHz wall clock

__global__ __launch_bounds__(768) void gru_layer_split_kernel(GruSplitParams p) {
  __shared__ __attribute__((aligned(16))) float sh[H];
  __shared__ float sg[3 * HH];
  __shared__ unsigned s_ticket;
  const int n = threadIdx.x;
  if (n == 0) s_ticket = atomicAdd(p.ticket, 1u);
  __syncthreads();
  const int pair = (int)(s_ticket >> 1), half = (int)(s_ticket & 1u);
  const int b = pair >> 1, dir = pair & 1;
  if (b >= p.B) return;
  int len = p.lens[b];
  len = len < 0 ? 0 : (len > p.T ? p.T : len);
  // thread n: local gate row rr = n / 2 (gate g = rr / 128 of hidden unit j = rr % 128), k half kh = n % 2
  const int rr = n >> 1, kh = n & 1;
  const int g = rr / HH, j = rr - g * HH;
  const int row = g * H + half * HH + j;                 // row of W_hh / b_hh of this direction
  float4 w[HH / 4];
  if (len > 0) {
    const float4* wp = (const float4*)(p.whh + ((size_t)dir * 3 * H + row) * H + kh * HH);
#pragma unroll
    for (int q = 0; q < HH / 4; ++q) w[q] = wp[q];
  }
  const float bias = p.bhh[dir * 3 * H + row];
  if (n < H) sh[n] = 0.f;
  unsigned long long* mine = p.xch + ((size_t)(pair * 2 + half) * 2) * HH;
  unsigned long long* theirs = p.xch + ((size_t)(pair * 2 + (1 - half)) * 2) * HH;
  __syncthreads();
  for (int step = 0; step < len; ++step) {
    const int t = dir ? (len - 1 - step) : step;
    const float* gxp = p.gx + (((size_t)b * p.T + t) * 2 + dir) * 3 * H + half * HH;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (n < HH) { gr = gxp[n]; gz = gxp[H + n]; gn = gxp[2 * H + n]; }
    float acc = 0.f;
    const float4* hp = (const float4*)(sh + kh * HH);
#pragma unroll
    for (int q = 0; q < HH / 4; ++q) {
      const float4 hv = hp[q];
      acc = fmaf(w[q].x, hv.x, acc);
      acc = fmaf(w[q].y, hv.y, acc);
      acc = fmaf(w[q].z, hv.z, acc);
      acc = fmaf(w[q].w, hv.w, acc);
    }
    acc += dpp_mov<DPP_QUAD_XOR1>(acc);                  // the two k halves of a row sit in neighbouring lanes
    if (kh == 0) sg[rr] = acc + bias;
    __syncthreads();
    const unsigned tag = (unsigned)step + 1u;
    if (n < HH) {
      const float r = sigmoidf_(gr + sg[n]);
      const float z = sigmoidf_(gz + sg[HH + n]);
      const float c = tanhf(gn + r * sg[2 * HH + n]);
      const float hn = (1.0f - z) * c + z * sh[half * HH + n];
      __hip_atomic_store(mine + (step & 1) * HH + n, ((unsigned long long)tag << 32) | __float_as_uint(hn),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh[half * HH + n] = hn;
      p.out[((size_t)b * p.T + t) * 2 * H + dir * H + half * HH + n] = hn;
    } else if (n < 2 * HH && step + 1 < len) {
      // the partner's half of h_t (not needed after the last step)
      const int jj = n - HH;
      unsigned long long* gq = theirs + (step & 1) * HH + jj;
      unsigned long long x = __hip_atomic_load(gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)(x >> 32) != tag) {
        const long long t0 = wall_clock64();
        do {
          __builtin_amdgcn_s_sleep(1);
          x = __hip_atomic_load(gq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (wall_clock64() - t0 > GRU_SPIN_TICKS) {
            atomicOr(p.error, 1u);
            break;
          }
        } while ((unsigned)(x >> 32) != tag);
      }
      sh[(1 - half) * HH + jj] = __uint_as_float((unsigned)x);
    }
    __syncthreads();
  }
  if (n < HH)
    for (int t = len; t < p.T; ++t) p.out[((size_t)b * p.T + t) * 2 * H + dir * H + half * HH + n] = 0.f;
}

// fc_emb[b][c] = sum_{t < len[b]} x[b][t][c] / len[b]   (model_util.py:41-63 mean_with_lens)
__global__ void mean_lens_kernel(const float* x, const int* lens, float* out, int T, int C) {
  const int b = blockIdx.x;
  const int len = lens[b];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int t = 0; t < len && t < T; ++t) s += x[((size_t)b * T + t) * C + c];
    out[(size_t)b * C + c] = s / (float)len;
  }
}

// max over valid steps (model_util.py:65-81 max_with_lens) + mean, Cnn14's own fc_emb input
__global__ void maxmean_lens_kernel(const float* x, const int* lens, float* out, int T, int C) {
  const int b = blockIdx.x;
  const int len = lens[b];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f, m = -INFINITY;
    for (int t = 0; t < len && t < T; ++t) {
      const float v = x[((size_t)b * T + t) * C + c];
      s += v;
      m = fmaxf(m, v);
    }
    out[(size_t)b * C + c] = m + s / (float)len;
  }
}

// packed[d][k/4][n][k%4] = whh[d][n][k]
__global__ void gru_pack_whh_kernel(const float* whh, float* packed, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int kk = i & 3, n = (i >> 2) % (3 * H), k4 = ((i >> 2) / (3 * H)) % (H / 4), d = i / (3 * H * H);
  packed[i] = whh[((size_t)d * 3 * H + n) * H + k4 * 4 + kk];
}

}  // namespace

extern "C" int ac_gru_pack_whh(const float* whh, float* packed, int hidden, void* stream) {
  if (!whh || !packed || hidden != H) return AC_ERR_ARG;
  const int total = 2 * 3 * H * H;
  hipLaunchKernelGGL(gru_pack_whh_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, whh, packed, total);
  return ac_check_launch();
}

// C ABI: see include/audiocaption_hip.h
extern "C" int ac_gru_layer(const float* gx, const float* whhT, const float* bhh, const int* lens, float* out,
                            int B, int T, int hidden, void* stream) {
  if (!gx || !whhT || !bhh || !lens || !out || B <= 0 || T <= 0 || hidden != H) return AC_ERR_ARG;
  GruParams p;
  p.gx = gx; p.whhT = whhT; p.bhh = bhh; p.lens = lens; p.out = out; p.B = B; p.T = T;
  const size_t lds = (size_t)GRU_KLDS * 3 * H * sizeof(float);
  static bool allowed = false;
  if (!allowed) {
    if (hipFuncSetAttribute((const void*)gru_layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return AC_ERR_LAUNCH;
    allowed = true;
  }
  hipLaunchKernelGGL(gru_layer_kernel, dim3(2 * B), dim3(768), lds, (hipStream_t)stream, p);
  return ac_check_launch();
}

// Workspace of ac_gru_layer_split in bytes: [error word, sticky][pad to 64][ticket][pad to 128][granules [2B][2][2][128] x 8].
// The error word sits at offset 0 whatever B is, so a workspace sized for a large batch can serve a small one.
extern "C" long ac_gru_split_workspace_bytes(int B) {
  if (B <= 0) return AC_ERR_ARG;
  return 128 + (long)B * 2 * 2 * 2 * HH * 8;
}

extern "C" int ac_gru_layer_split(const float* gx, const float* whh, const float* bhh, const int* lens, float* out,
                                  void* workspace, int B, int T, int hidden, void* stream) {
  if (!gx || !whh || !bhh || !lens || !out || !workspace || B <= 0 || T <= 0 || hidden != H) return AC_ERR_ARG;
  const size_t gran = (size_t)B * 2 * 2 * 2 * HH * 8;
  GruSplitParams p;
  p.gx = gx; p.whh = whh; p.bhh = bhh; p.lens = lens; p.out = out; p.B = B; p.T = T;
  p.error = (unsigned*)workspace;
  p.ticket = (unsigned*)((char*)workspace + 64);
  p.xch = (unsigned long long*)((char*)workspace + 128);
  // granules and the ticket start from zero on EVERY launch (a memset node when captured); the error word is sticky
  if (hipMemsetAsync((char*)workspace + 64, 0, 64 + gran, (hipStream_t)stream) != hipSuccess) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(gru_layer_split_kernel, dim3(4 * B), dim3(768), 0, (hipStream_t)stream, p);
  return ac_check_launch();
}

extern "C" int ac_mean_with_lens(const float* x, const int* lens, float* out, int B, int T, int C, int add_max,
                                 void* stream) {
  if (!x || !lens || !out || B <= 0 || T <= 0 || C <= 0) return AC_ERR_ARG;
  if (add_max)
    hipLaunchKernelGGL(maxmean_lens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, lens, out, T, C);
  else
    hipLaunchKernelGGL(mean_lens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, lens, out, T, C);
  return ac_check_launch();
}
