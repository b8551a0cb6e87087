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
// The recurrence of ONE (clip, direction) split over FOUR workgroups of 256 threads.  A CU cannot hold the 786 KB recurrent
// matrix (the single-workgroup kernel above re-streams 44 % of it from L2 every step, which is its whole critical
// path); a quarter of it fits in the registers of four waves: workgroup `part` owns the hidden units [64 part, 64 part +
// 64); thread (unit j = tid / 4, k quarter kq = tid % 4) holds the r, z and n rows of its unit for k in [64 kq, 64 kq + 64)
// - 192 weights in VGPRs, read once - so the three gate sums of a unit are complete after a 4-lane DPP reduction and the
// lane kq = 0 computes the new hidden value straight away (no gate round trip through LDS).  Per step the four parts
// trade their 64 new hidden values through L2 as 8-byte {tag, value} granules (one relaxed agent-scope atomic store each,
// the value IS the flag; the 192 lanes kq != 0 each poll ONE granule of another part until the tag is this step's -
// cdna_hip_programming.md section 6 Guideline 16, form R2: no fences, correct for any workgroup -> XCD placement).  Slots
// alternate with the step parity: a producer cannot reach step s + 2 before it has consumed every partner's step s + 1,
// which they published after consuming the producer's step s.
// Partners are grouped by START ORDER, not by block index (HIP promises no dispatch order): every workgroup draws a
// ticket, tickets 4g .. 4g + 3 form group g.  At any time at most one group is incomplete, and its missing members are the
// next workgroups to start - all other resident workgroups belong to complete groups that finish on their own, so the
// scheme cannot deadlock however the blocks are dispatched; a spin that outlasts GRU_SPIN_TICKS all the same (another
// process holding the GPU ...) raises the error word instead of hanging.
// One wave per SIMD and <= 242 VGPRs on purpose: a workgroup fits the slot one conv workgroup of the encoder leaves on
// a CU, two fit a CU: the 512 workgroups of a 64-clip batch are resident together on the 256 CUs.
// ---------------------------------------------------------------------------------------------------------------------
struct GruSplitParams {
  const float* gx;     // [B][T][2][3H]
  const float* whh;    // [2][3H][H]   nn.GRU weight_hh of both directions, UNPACKED
  const float* bhh;    // [2][3H]
  const int* lens;     // [B]
  float* out;          // [B][T][2H]
  float* save;         // optional [B][T][2][4H]: r, z, n, W_hn h + b_hn per cell, for the backward pass (training)
  unsigned long long* xch;   // [2B groups][4 parts][2 slots][64] granules, zeroed before every launch
  unsigned* ticket;          // zeroed before every launch
  unsigned* error;           // set to 1 when a partner never showed up
  int B, T;
};
constexpr int GQ = 4, HQ = H / GQ;              // parts per (clip, direction); hidden units per part
constexpr int HQP = HQ + 4;                      // LDS pitch of a quarter of h: the four k quarters hit different banks
constexpr long long GRU_SPIN_TICKS = 200000000;   // 2 s of the 100 MHz wall clock

__global__ void gru_split_reset_kernel(unsigned long long* w, unsigned words) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < words) w[i] = 0ull;
}

__global__ __launch_bounds__(256) void gru_layer_split_kernel(GruSplitParams p) {
  __shared__ __attribute__((aligned(16))) float sh[GQ * HQP];
  __shared__ unsigned s_ticket;
  const int n = threadIdx.x;
  if (n == 0) s_ticket = atomicAdd(p.ticket, 1u);
  __syncthreads();
  const int group = (int)(s_ticket >> 2), part = (int)(s_ticket & 3u);
  const int b = group >> 1, dir = group & 1;
  if (b >= p.B) return;
  int len = p.lens[b];
  len = len < 0 ? 0 : (len > p.T ? p.T : len);
  const int j = n >> 2, kq = n & 3;
  const int unit = part * HQ + j;                          // hidden unit of this thread
  float4 w[3][HQ / 4];
  if (len > 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float4* wp = (const float4*)(p.whh + ((size_t)dir * 3 * H + g * H + unit) * H + kq * HQ);
#pragma unroll
      for (int q = 0; q < HQ / 4; ++q) w[g][q] = wp[q];
    }
  }
  const float br = p.bhh[dir * 3 * H + unit], bz = p.bhh[dir * 3 * H + H + unit], bn = p.bhh[dir * 3 * H + 2 * H + unit];
  for (int i = n; i < GQ * HQP; i += 256) sh[i] = 0.f;
  unsigned long long* mine = p.xch + ((size_t)(group * GQ + part) * 2) * HQ;
  // the lanes kq = 1..3 of unit j fetch value j of part (part + kq) % 4
  const int other = (part + kq) & 3;
  unsigned long long* theirs = p.xch + ((size_t)(group * GQ + other) * 2) * HQ;
  __syncthreads();
  for (int step = 0; step < len; ++step) {
    const int t = dir ? (len - 1 - step) : step;
    const float* gxp = p.gx + (((size_t)b * p.T + t) * 2 + dir) * 3 * H + unit;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (kq == 0) { gr = gxp[0]; gz = gxp[H]; gn = gxp[2 * H]; }
    float ar = 0.f, az = 0.f, an = 0.f;
    const float4* hp = (const float4*)(sh + kq * HQP);
#pragma unroll
    for (int q = 0; q < HQ / 4; ++q) {
      const float4 hv = hp[q];
      ar = fmaf(w[0][q].x, hv.x, ar); ar = fmaf(w[0][q].y, hv.y, ar); ar = fmaf(w[0][q].z, hv.z, ar); ar = fmaf(w[0][q].w, hv.w, ar);
      az = fmaf(w[1][q].x, hv.x, az); az = fmaf(w[1][q].y, hv.y, az); az = fmaf(w[1][q].z, hv.z, az); az = fmaf(w[1][q].w, hv.w, az);
      an = fmaf(w[2][q].x, hv.x, an); an = fmaf(w[2][q].y, hv.y, an); an = fmaf(w[2][q].z, hv.z, an); an = fmaf(w[2][q].w, hv.w, an);
    }
    // the four k quarters of a unit sit in four neighbouring lanes
    ar += dpp_mov<DPP_QUAD_XOR1>(ar); ar += dpp_mov<DPP_QUAD_XOR2>(ar);
    az += dpp_mov<DPP_QUAD_XOR1>(az); az += dpp_mov<DPP_QUAD_XOR2>(az);
    an += dpp_mov<DPP_QUAD_XOR1>(an); an += dpp_mov<DPP_QUAD_XOR2>(an);
    const float hold = sh[part * HQP + j];
    __syncthreads();                                   // every wave has read h(t-1): it may be overwritten now
    const unsigned tag = (unsigned)step + 1u;
    if (kq == 0) {
      const float r = sigmoidf_(gr + ar + br);
      const float z = sigmoidf_(gz + az + bz);
      const float ghn = an + bn;
      const float c = tanhf(gn + r * ghn);
      const float hn = (1.0f - z) * c + z * hold;
      if (p.save) {
        float* sv = p.save + (((size_t)b * p.T + t) * 2 + dir) * 4 * H + unit;
        sv[0] = r; sv[H] = z; sv[2 * H] = c; sv[3 * H] = ghn;
      }
      __hip_atomic_store(mine + (step & 1) * HQ + j, ((unsigned long long)tag << 32) | __float_as_uint(hn),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh[part * HQP + j] = hn;
      p.out[((size_t)b * p.T + t) * 2 * H + dir * H + unit] = hn;
    }
    // A SEPARATE statement, not an else-branch: publishers (kq = 0) and pollers share waves, and a wave that entered the
    // polling side first would spin while its own publishing lanes are masked off - every part waiting for every other.
    // The empty asm is a compiler barrier for memory operations: nothing else stops hipcc from placing the polling loop -
    // relaxed atomics on other addresses - ahead of the publishing store (it did, in a multi-sequence variant of this kernel).
    // Pinned twice: __atomic_signal_fence is the language-level statement (atomic operations of this thread are not moved
    // across a seq_cst signal fence: no instruction, compiler ordering only), the empty asm is the same for every other
    // memory operation.  tests/test_cpu_host.py::test_split_gru_publishes_before_it_polls checks the order in the ISA of the
    // library that ships (the sc1 granule store precedes the first sc1 granule load of the loop body).
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    asm volatile("" ::: "memory");
    if (kq != 0 && step + 1 < len) {
      // another part's value j of h(t) (not needed after the last step)
      unsigned long long* gq = theirs + (step & 1) * HQ + j;
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
      sh[other * HQP + j] = __uint_as_float((unsigned)x);
    }
    __syncthreads();
  }
  if (kq == 0)
    for (int t = len; t < p.T; ++t) p.out[((size_t)b * p.T + t) * 2 * H + dir * H + unit] = 0.f;
}

// fc_emb[b][c] = sum_{t < len[b]} x[b][t][c] / len[b]   (model_util.py:41-63 mean_with_lens).  One thread per (clip, channel),
// channels of a clip over blockIdx.y (one workgroup per clip walked 1408 channels in 6 passes of 32 dependent loads: 63 us for
// 128 EfficientNet clips); t ascending as before: the same bits.
__global__ void mean_lens_kernel(const float* x, const int* lens, float* out, int T, int C) {
  const int b = blockIdx.x, c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int len = lens[b];
  float s = 0.f;
  for (int t = 0; t < len && t < T; ++t) s += x[((size_t)b * T + t) * C + c];
  out[(size_t)b * C + c] = s / (float)len;
}

// max over valid steps (model_util.py:65-81 max_with_lens) + mean, Cnn14's own fc_emb input
__global__ void maxmean_lens_kernel(const float* x, const int* lens, float* out, int T, int C) {
  const int b = blockIdx.x, c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int len = lens[b];
  float s = 0.f, m = -INFINITY;
  for (int t = 0; t < len && t < T; ++t) {
    const float v = x[((size_t)b * T + t) * C + c];
    s += v;
    m = fmaxf(m, v);
  }
  out[(size_t)b * C + c] = m + s / (float)len;
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
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)gru_layer_kernel, (int)lds, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(gru_layer_kernel, dim3(2 * B), dim3(768), lds, (hipStream_t)stream, p);
  return ac_check_launch();
}

// Workspace of ac_gru_layer_split in bytes: [error word, sticky][pad to 64][ticket][pad to 128][granules [2B][4][2][64] x 8].
// The error word sits at offset 0 whatever B is, so a workspace sized for a large batch can serve a small one.
extern "C" long ac_gru_split_workspace_bytes(int B) {
  if (B <= 0) return AC_ERR_ARG;
  return 128 + (long)B * 2 * GQ * 2 * HQ * 8;
}

extern "C" int ac_gru_layer_split(const float* gx, const float* whh, const float* bhh, const int* lens, float* out,
                                  float* save, void* workspace, int B, int T, int hidden, void* stream) {
  if (!gx || !whh || !bhh || !lens || !out || !workspace || B <= 0 || T <= 0 || hidden != H) return AC_ERR_ARG;
  const size_t gran = (size_t)B * 2 * GQ * 2 * HQ * 8;
  GruSplitParams p;
  p.gx = gx; p.whh = whh; p.bhh = bhh; p.lens = lens; p.out = out; p.save = save; p.B = B; p.T = T;
  p.error = (unsigned*)workspace;
  p.ticket = (unsigned*)((char*)workspace + 64);
  p.xch = (unsigned long long*)((char*)workspace + 128);
  // Granules and the ticket start from zero on EVERY launch; the error word is sticky.  By a KERNEL, not hipMemsetAsync: as
  // a memset node of a captured graph the clearing was not reliably ordered before the kernel node that follows it in
  // replays (rocm 7.2: with the host synchronised before a replay the tickets carried on from the previous launch, every
  // workgroup took itself for a surplus one and returned - a stale layer output; with the frozen Cnn14 of the next iteration
  // running beside the step the clearing landed in the middle of the recurrence - partner timeouts).
  {
    const unsigned words = (unsigned)((64 + gran) / 8);
    hipLaunchKernelGGL(gru_split_reset_kernel, dim3((words + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long*)((char*)workspace + 64), words);
  }
  hipLaunchKernelGGL(gru_layer_split_kernel, dim3(2 * GQ * B), dim3(256), 0, (hipStream_t)stream, p);
  return ac_check_launch();
}

extern "C" int ac_mean_with_lens(const float* x, const int* lens, float* out, int B, int T, int C, int add_max,
                                 void* stream) {
  if (!x || !lens || !out || B <= 0 || T <= 0 || C <= 0) return AC_ERR_ARG;
  if (add_max)
    hipLaunchKernelGGL(maxmean_lens_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, lens, out, T, C);
  else
    hipLaunchKernelGGL(mean_lens_kernel, dim3(B, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, lens, out, T, C);
  return ac_check_launch();
}
