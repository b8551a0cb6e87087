// Greedy decoding as ONE persistent launch: a ROW of the batch is decoded by a CLUSTER of four workgroups for all of its
// steps (reference: CaptionModel.stepwise_forward / sample_next_word / stepwise_process_step, base.py:152-218;
// TransformerDecoder.forward, transformer_decoder.py:80-103; TransformerModel.prepare_decoder_input, transformer_model.py:34-57).
//
// Why.  The launch-per-stage chain (csrc/decoder.hip: 10 launches per step replayed from a HIP graph) costs ~90 us per step
// at ANY row count: every stage is a dependent launch of a few microseconds of fill / first-touch latency, and both attention
// phases of a layer re-read their keys and values from L2 (4.7 us each).  When nothing else runs on the GPU (the blocking
// model() call, single clips) that latency is the whole cost of the decode.
//
// How.  Part p of a cluster owns HEAD p of both attention sub-layers and QUARTER p of the feed-forward hidden units and of
// the vocabulary, for every layer, and keeps in LDS what it would otherwise re-read each step: its head's slice of the
// projected audio memory (K, V of every layer) and its head's self-attention cache.  Per step and layer:
//   q, k, v of head p      = x W_in[head p]^T                    (full row x in LDS; 192 x 256 weights streamed from L2)
//   self attention of head p (keys in LDS)  ->  PARTIAL out-projection: the head's 64 context values x W_o[:, head p]^T
//   EXCHANGE 1: the four partial 256-vectors are summed by everybody (same order: parts 0..3) -> + bias, + x, LayerNorm 1
//   cross query of head p, cross attention of head p (memory in LDS) -> partial out-projection
//   EXCHANGE 2 -> LayerNorm 2
//   hidden quarter p = relu(x W_1[quarter p]^T + b) -> partial second projection W_2[:, quarter p]
//   EXCHANGE 3 -> LayerNorm 3
// and after the last layer the logits of vocabulary quarter p, its (max, arg-max, sum of exponentials), EXCHANGE 4 of those
// three numbers -> token, log-probability, <end> bookkeeping, computed identically by the four parts.  Seven exchanges per
// step for two layers.  An exchange is the split GRU kernel's hand-off (csrc/gru.hip): 8-byte {tag, value} granules written
// with one relaxed agent-scope store each and polled by the thread that needs them - the value IS the flag, so no fence
// and no assumption about which XCD a workgroup runs on; two slots alternate (a part cannot publish exchange e + 2 before
// it has read every partner's e + 1, which they published after reading its e).  Clusters form by START ORDER (a ticket
// counter), so the scheme cannot deadlock however the blocks are dispatched; a spin that outlasts 2 s raises the error
// word and the launch unwinds.  Every part computes the row's LayerNorms and the token redundantly from identical inputs
// in identical order: the four stay bit-identical without exchanging more.
//
// The reference stops a batch when every row has emitted <end> (base.py:206-211).  A cluster cannot know the other rows'
// state without a grid barrier; it reads the global counters of the PREVIOUS step without waiting (part 0, shared through
// exchange 4) and stops one or two steps after the reference would have; cluster_finalize_kernel then restores the
// reference's initial values in the columns it would not have written.
//
// d_model 256, 4 heads of 64, dim_ff 1024, <= 8 layers, max_len <= 32, the LDS must hold the row's memory slices
// (nlayers x (Tm + max_len) x 512 bytes + 12 KB <= 160 KB); other shapes are the launch chain's (AC_ERR_ARG here).
#include "ac_common.h"
#include <stdlib.h>
#include "../../include/audiocaption_hip.h"

namespace {

constexpr int CD = 256, CHD = 64, CFF = 1024, CFQ = 256, CPARTS = 4;
constexpr int CMAXL = 32;                         // max_len bound (scores of the self attention: one 8-lane group per key)
constexpr long long C_SPIN_TICKS = 200000000;     // 2 s of the 100 MHz wall clock (AUDIOCAPTION_CLUSTER_TIMEOUT_US overrides)
constexpr size_t C_LAYER_FLOATS = (size_t)CD * 192 + 64 * CD + (size_t)CD * 64 + 64 * CD + (size_t)CD * CFQ + (size_t)CFQ * CD;
constexpr size_t C_OFF_QKV = 0, C_OFF_O = (size_t)CD * 192, C_OFF_CQ = C_OFF_O + 64 * CD, C_OFF_CO = C_OFF_CQ + (size_t)CD * 64,
                 C_OFF_W1 = C_OFF_CO + 64 * CD, C_OFF_W2 = C_OFF_W1 + (size_t)CD * CFQ;

struct ClusterLayer {
  const float *bqkv, *bo, *bcq, *bco, *b1, *b2, *n1w, *n1b, *n2w, *n2b, *n3w, *n3b;
};

struct ClusterParams {
  const float* pk;        // [4 parts][part_floats]: per layer qkv [256][192], o [64][256], cq [256][64], co [64][256], w1 [256][256],
  size_t part_floats;     //   w2 [256][256]; then the classifier quarter in column blocks [256][256] ... [256][rest]
  const float* emb; const float* pe;
  ClusterLayer L[AC_MAX_LAYERS];
  int nlayers, V, VQ, NLC;   // vocabulary, columns of a quarter, the same padded to 64
  const float* memkv; const int* mem_len;
  int B, Tm, max_len, start_idx, end_idx, pad_idx;
  int stop_rows;            // B: stop when every row has emitted <end> (base.py:206-211); -1: run all max_len steps
  int64_t* seq; float* logit; float* logprob; float* embed; int* cnt;
  unsigned long long* prog; // [CMAXL]: per step, (clusters that finished it) << 32 | (rows still unfinished after it): ONE word,
                            //   so that a reader who sees every cluster arrived also sees every row's contribution
  unsigned long long* xch;  // [B][2 slots][4 parts][256] granules
  unsigned* ticket; unsigned* error;
  long long spin_ticks;     // how long a part polls for a partner's granule before it raises the error word (100 MHz ticks)
  int fault;                // development / tests: the workgroup that draws ticket fault - 1 returns at once (a lost partner); 0: off
};

__device__ __forceinline__ long long c_wall_clock() { return (long long)__builtin_amdgcn_s_memrealtime(); }

#ifdef AC_CLUSTER_STAMPS   // development (tools/cluster_stamps.py): 100 MHz timestamps of row 0 / part 0 at the stage boundaries of step 5
__device__ unsigned long long g_cluster_stamps[64];
#define C_STAMP(k) do { if (row == 0 && part == 0 && tid == 0 && t == 5) g_cluster_stamps[k] = (unsigned long long)c_wall_clock(); } while (0)
extern "C" int ac_cluster_stamps_read(unsigned long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_cluster_stamps), sizeof(g_cluster_stamps)) == hipSuccess ? 0 : -2;
}
#else
#define C_STAMP(k) do { } while (0)
#endif

// ---- matrix-vector product from a row-major [K][NL] blob: y[c] = sum_k W[k][c] x[k].  Thread (column quad cq, k slice ks); a
// wave instruction reads whole rows (NL = 256: one 1 KiB row).  The weights do not depend on the activations, so a product
// comes in two halves: `issue` requests the thread's first 16 rows (64 registers) BEFORE the stage that produces x - they land
// under that stage's exchange / attention / LayerNorm - and `finish` streams the rest double-buffered (16 rows being
// multiplied, 16 in flight) and reduces the k slices through `red` ([<= 5][NL] floats, free again on return).  Returns
// y[tid] for tid < NL; all 256 threads call both halves. ----
// Weights are read through ONE buffer descriptor per part (the blob is < 4 GiB): a load is descriptor + 32-bit per-thread
// offset + scalar offset of the (stage, row) - no 64-bit address arithmetic per load (with global loads hipcc hoisted ~450
// address registers out of the step loop and spilled them).
typedef __amdgpu_buffer_rsrc_t CRsrc;
__device__ __forceinline__ f32x4 c_wload(CRsrc rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

template <int NL, int K>
struct CGemv {
  static constexpr int NCQ = NL / 4, KS = 256 / NCQ, ROWS = (K + KS - 1) / KS;
  static_assert(NL % 4 == 0 && NL <= 256 && KS >= 1 && ROWS >= 16 && ROWS % 4 == 0 && K % 4 == 0, "column quads over the 256 threads");

  // woff: byte offset of the [K][NL] matrix inside the part's blob
  __device__ static __forceinline__ void issue(CRsrc rs, unsigned woff, int tid, f32x4 (&a)[16]) {
    const int cq = tid % NCQ, ks = tid / NCQ;
    if (ks < KS) {
      const unsigned voff = (unsigned)((ks * ROWS * NL + 4 * cq) * 4);   // every slice has at least 16 rows
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = c_wload(rs, voff, woff + (unsigned)(u * NL * 4));
    }
  }

  __device__ static __forceinline__ void fma16(const f32x4 (&w)[16], const float* xk, f32x4& acc) {
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
      const f32x4 xv = *(const f32x4*)(xk + 4 * u4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32x4 wv = w[4 * u4 + e];
        acc[0] = fmaf(wv[0], xv[e], acc[0]); acc[1] = fmaf(wv[1], xv[e], acc[1]);
        acc[2] = fmaf(wv[2], xv[e], acc[2]); acc[3] = fmaf(wv[3], xv[e], acc[3]);
      }
    }
  }

  __device__ static __forceinline__ float finish(CRsrc rs, unsigned woff, const float* x, float* red, int tid, f32x4 (&a)[16]) {
    const int cq = tid % NCQ, ks = tid / NCQ;
    if (ks < KS) {
      const int k0 = ks * ROWS;
      const int nr = K - k0 < ROWS ? K - k0 : ROWS;
      const int nb = nr >> 4;
      const unsigned voff = (unsigned)((k0 * NL + 4 * cq) * 4);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      f32x4 b[16];
#pragma unroll 1
      for (int i = 0; i < nb; i += 2) {
        if (i + 1 < nb) {
#pragma unroll
          for (int u = 0; u < 16; ++u) b[u] = c_wload(rs, voff, woff + (unsigned)(((i + 1) * 16 + u) * NL * 4));
        }
        fma16(a, x + k0 + i * 16, acc);
        if (i + 2 < nb) {
#pragma unroll
          for (int u = 0; u < 16; ++u) a[u] = c_wload(rs, voff, woff + (unsigned)(((i + 2) * 16 + u) * NL * 4));
        }
        if (i + 1 < nb) fma16(b, x + k0 + (i + 1) * 16, acc);
      }
      for (int k = nb * 16; k < nr; ++k) {   // NL = 192: 52 rows per slice
        const f32x4 wv = c_wload(rs, voff, woff + (unsigned)(k * NL * 4));
        const float xv = x[k0 + k];
        acc[0] = fmaf(wv[0], xv, acc[0]); acc[1] = fmaf(wv[1], xv, acc[1]);
        acc[2] = fmaf(wv[2], xv, acc[2]); acc[3] = fmaf(wv[3], xv, acc[3]);
      }
      *(f32x4*)(red + ks * NL + 4 * cq) = acc;
    }
    __syncthreads();
    float y = 0.f;
    if (tid < NL) {
#pragma unroll
      for (int s = 0; s < KS; ++s) y += red[s * NL + tid];
    }
    __syncthreads();
    return y;
  }
};

struct ClusterCtx {
  unsigned long long* base;     // the cluster's granules [2 slots][4 parts][256]
  unsigned* error;
  int part;
  unsigned seqno;
  long long spin_ticks;
};

// Publish `val` (threads with `active`) and collect the four parts' values of this exchange: x4[q * 256 + tid] = part q's value
// for thread tid (LDS, [4][256]; written and read back by the same thread - other threads' entries are visible after the
// barrier this function ends with).  Returns false when a partner never answered (the error word is set; the caller unwinds).
__device__ __forceinline__ bool cluster_exchange(ClusterCtx& c, float val, bool active, int tid, float* x4) {
  c.seqno += 1u;
  const unsigned tag = c.seqno;
  const int slot = (int)(tag & 1u);
  bool ok = true;
  if (active) {
    __hip_atomic_store(c.base + (size_t)(slot * 4 + c.part) * 256 + tid, ((unsigned long long)tag << 32) | __float_as_uint(val),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // publishers and pollers are the same threads: the order of the two statements is what the hand-off lives on (csrc/gru.hip)
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  asm volatile("" ::: "memory");
  if (active) {
    x4[c.part * 256 + tid] = val;
    unsigned long long x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int other = (c.part + 1 + i) & 3;
      x[i] = __hip_atomic_load(c.base + (size_t)(slot * 4 + other) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int other = (c.part + 1 + i) & 3;
      if ((unsigned)(x[i] >> 32) != tag) {
        const long long t0 = c_wall_clock();
        do {
          __builtin_amdgcn_s_sleep(1);
          x[i] = __hip_atomic_load(c.base + (size_t)(slot * 4 + other) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (c_wall_clock() - t0 > c.spin_ticks) { ok = false; break; }
        } while ((unsigned)(x[i] >> 32) != tag);
      }
      x4[other * 256 + tid] = __uint_as_float((unsigned)x[i]);
    }
    if (!ok) atomicOr(c.error, 1u);
  }
  return !__syncthreads_or(ok ? 0 : 1);
}

// LayerNorm over the 256 values of a row, one per thread (eps 1e-5, like nn.LayerNorm); red8: 8 floats of LDS
__device__ __forceinline__ float cluster_ln(float v, const float* w, const float* b, float* red8, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const float s1 = wave_sum(v);
  if (lane == 0) red8[wave] = s1;
  __syncthreads();
  const float mean = ((red8[0] + red8[1]) + (red8[2] + red8[3])) * (1.0f / CD);
  const float dl = v - mean;
  const float s2 = wave_sum(dl * dl);
  if (lane == 0) red8[4 + wave] = s2;
  __syncthreads();
  const float rstd = rsqrtf(((red8[4] + red8[5]) + (red8[6] + red8[7])) * (1.0f / CD) + 1e-5f);
  const float y = dl * rstd * w[tid] + b[tid];
  __syncthreads();   // red8 is reused by the next LayerNorm
  return y;
}

// Single-query attention of one head over `nkeys` keys held in LDS (Kh, Vh: [nkeys][64]): scores by 8-lane groups (32 keys per
// pass), softmax by the first wave, context by (channel, key quarter).  Key j is masked when j >= klen or kmask[j] != 0.
// q: [64] in LDS; sc: [>= nkeys] scratch; part4: [4][64] scratch; returns context channel tid for tid < 64.
__device__ __forceinline__ float cluster_attn(const float* q, const float* Kh, const float* Vh, int nkeys, int klen,
                                              const unsigned char* kmask, float* sc, float* part4, float* red8, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int sub = tid & 7;
  float qv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qv[e] = q[sub * 8 + e];
  for (int j0 = 0; j0 < nkeys; j0 += 32) {
    const int j = j0 + (tid >> 3);
    float s = 0.f;
    if (j < nkeys) {
      const float* kp = Kh + j * CHD + sub * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(qv[e], kp[e], s);
    }
    s += dpp_mov<DPP_QUAD_XOR1>(s);
    s += dpp_mov<DPP_QUAD_XOR2>(s);
    s += dpp_mov<DPP_ROW_HALF_MIRROR>(s);
    if (sub == 0 && j < nkeys) {
      const bool masked = (j >= klen) || (kmask && kmask[j]);
      sc[j] = masked ? -INFINITY : s * 0.125f;   // 1 / sqrt(64)
    }
  }
  __syncthreads();
  if (wave == 0) {
    float m = -INFINITY;
    for (int j = lane; j < nkeys; j += 64) m = fmaxf(m, sc[j]);
    m = wave_max(m);
    float den = 0.f;
    for (int j = lane; j < nkeys; j += 64) {
      const float e = expf(sc[j] - m);
      sc[j] = e;
      den += e;
    }
    den = wave_sum(den);
    if (lane == 0) red8[0] = den;
  }
  __syncthreads();
  {
    const int d = lane;
    float o = 0.f;
    for (int j = wave; j < nkeys; j += 4) o = fmaf(sc[j], Vh[j * CHD + d], o);
    part4[wave * CHD + d] = o;
  }
  __syncthreads();
  float ctx = 0.f;
  if (tid < CHD) ctx = ((part4[tid] + part4[CHD + tid]) + (part4[2 * CHD + tid] + part4[3 * CHD + tid])) / red8[0];
  __syncthreads();
  return ctx;
}

// batch i (rows 16 i .. 16 i + 15) of classifier column block `blk` ([256][256] floats), this lane's column quad
__device__ __forceinline__ void cls_issue(CRsrc rs, unsigned cls_off, int blk, int i, int lane, f32x4 (&buf)[16]) {
  const unsigned soff = cls_off + (unsigned)((blk * (CD * 256) + i * 16 * 256) * 4);
#pragma unroll
  for (int u = 0; u < 16; ++u) buf[u] = c_wload(rs, (unsigned)(lane * 16), soff + (unsigned)(u * 1024));
}

__device__ __forceinline__ void argmax_merge_c(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ __launch_bounds__(256) void cluster_init_kernel(ClusterParams p, unsigned xch_words) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < (unsigned)(p.B * p.max_len)) { p.seq[i] = p.end_idx; p.logprob[i] = 0.f; }
  if (i < (unsigned)p.max_len) p.cnt[i] = 0;
  if (i < (unsigned)CMAXL) p.prog[i] = 0ull;
  if (i == 0) { *p.ticket = 0u; *p.error = 0u; }   // the error word reports THIS call (a stale flag would re-decode every later batch)
  for (unsigned w = i; w < xch_words; w += gridDim.x * 256u) p.xch[w] = 0ull;
}

// unfinished_cnt[t] out of the progress words, and the reference's initial values in the columns of steps it would not have
// executed (every row had emitted <end> one step earlier, base.py:167)
__global__ void cluster_finalize_kernel(int64_t* seq, float* logprob, int* cnt, const unsigned long long* prog, int B, int max_len,
                                        int end_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * max_len) return;
  const int t = i % max_len;
  bool dead = false;
  for (int u = 0; u < t; ++u) dead = dead || (unsigned)prog[u] == 0u;
  if (dead) { seq[i] = end_idx; logprob[i] = 0.f; }
  if (i < max_len) {
    bool d2 = false;
    for (int u = 0; u < i; ++u) d2 = d2 || (unsigned)prog[u] == 0u;
    cnt[i] = d2 ? 0 : (int)(unsigned)prog[i];
  }
}

__global__ __launch_bounds__(256, 1) void greedy_cluster_kernel(ClusterParams p) {   // one workgroup per CU: 64 rows resident at once
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ unsigned s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_ticket = atomicAdd(p.ticket, 1u);
  __syncthreads();
  const int row = (int)(s_ticket >> 2), part = (int)(s_ticket & 3u);
  if (row >= p.B) return;
  if (p.fault && s_ticket == (unsigned)(p.fault - 1)) return;   // injected fault: this part never answers its partners
  const int nl = p.nlayers, Tm = p.Tm, L = p.max_len;
  // ---- LDS carving (floats) ----
  float* xs = lds;                    // [256] the row entering a sub-layer
  float* qkv = xs + CD;               // [192] q, k, v of this head / [64] cross query
  float* ctxs = qkv + 192;            // [64]
  float* hid = ctxs + CHD;            // [256] hidden quarter
  float* red = hid + CFQ;             // [5 * 256] matrix-vector partial sums
  float* sc = red + 5 * CD;           // [max(Tm, 32)] scores
  const int nsc = ((Tm > CMAXL ? Tm : CMAXL) + 3) & ~3;
  float* part4 = sc + nsc;            // [4][64]
  float* red8 = part4 + 4 * CHD;      // [8] + [8] spare
  float* x4 = red8 + 16;              // [4][256] the four parts' values of an exchange
  float* lg = x4 + 4 * CD;            // [NLC] this quarter's logits of the step
  float* selfK = lg + p.NLC;          // [nl][L][64]
  float* selfV = selfK + (size_t)nl * L * CHD;
  float* memK = selfV + (size_t)nl * L * CHD;   // [nl][Tm][64]
  float* memV = memK + (size_t)nl * Tm * CHD;
  unsigned char* kmask = (unsigned char*)(memV + (size_t)nl * Tm * CHD);   // [L + 1] token == pad
  // ---- this head's slice of the projected audio memory: memkv [layer][B * Tm][K 256 | V 256] ----
  for (int i = tid; i < nl * Tm * 16; i += 256) {
    const int c4 = i & 15, j = (i >> 4) % Tm, l = (i >> 4) / Tm;
    const float* src = p.memkv + ((size_t)l * p.B * Tm + (size_t)row * Tm + j) * (2 * CD) + part * CHD + 4 * c4;
    *(f32x4*)(memK + ((size_t)l * Tm + j) * CHD + 4 * c4) = *(const f32x4*)src;
    *(f32x4*)(memV + ((size_t)l * Tm + j) * CHD + 4 * c4) = *(const f32x4*)(src + CD);
  }
  if (tid == 0) kmask[0] = p.start_idx == p.pad_idx ? 1 : 0;
  int mlen = p.mem_len[row];
  mlen = mlen < 0 ? 0 : (mlen > Tm ? Tm : mlen);
  ClusterCtx cx;
  cx.base = p.xch + (size_t)row * 2 * CPARTS * 256;   // slot s of part q at base + (s * 4 + q) * 256
  cx.error = p.error;
  cx.part = part;
  cx.seqno = 0u;
  cx.spin_ticks = p.spin_ticks;
  const CRsrc wrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pk + (size_t)part * p.part_floats), 0, (int)(p.part_floats * 4), 0x00020000);
  const unsigned cls_off = (unsigned)((size_t)nl * C_LAYER_FLOATS * 4);
  int tok = p.start_idx, unfinished = 1;
  const float emb_scale = 16.0f;   // sqrt(d_model)
  f32x4 wpre[16];                  // the first 16 weight rows of the NEXT matrix-vector product (CGemv::issue)
  CGemv<192, CD>::issue(wrs, (unsigned)(C_OFF_QKV * 4), tid, wpre);
  __syncthreads();
  for (int t = 0; t < L; ++t) {
    // ---- x = E[tok] * sqrt(d) + pe[t] (transformer_decoder.py:89-91) ----
    float x = p.emb[(size_t)tok * CD + tid] * emb_scale + p.pe[(size_t)t * CD + tid];
    xs[tid] = x;
    __syncthreads();
    C_STAMP(0);
    for (int l = 0; l < nl; ++l) {
      const ClusterLayer& Ly = p.L[l];
      const unsigned lw = (unsigned)((size_t)l * C_LAYER_FLOATS * 4);   // byte offset of the layer's matrices in the blob
      // ---- self attention of head `part` (wpre: the first rows of the q, k, v weights, requested a stage ago) ----
      {
        const float y = CGemv<192, CD>::finish(wrs, lw + (unsigned)(C_OFF_QKV * 4), xs, red, tid, wpre);
        CGemv<CD, CHD>::issue(wrs, lw + (unsigned)(C_OFF_O * 4), tid, wpre);   // the out-projection's slice lands under the attention
        C_STAMP(1 + l * 12);
        if (tid < 192) {
          const int which = tid >> 6, c = tid & 63;
          const float v = y + Ly.bqkv[which * CD + part * CHD + c];
          if (which == 0) qkv[c] = v;
          else if (which == 1) selfK[((size_t)l * L + t) * CHD + c] = v;
          else selfV[((size_t)l * L + t) * CHD + c] = v;
        }
        __syncthreads();
        const float ctx = cluster_attn(qkv, selfK + (size_t)l * L * CHD, selfV + (size_t)l * L * CHD, t + 1, t + 1, kmask, sc,
                                       part4, red8, tid);
        if (tid < CHD) ctxs[tid] = ctx;
        __syncthreads();
        C_STAMP(2 + l * 12);
      }
      {
        const float partial = CGemv<CD, CHD>::finish(wrs, lw + (unsigned)(C_OFF_O * 4), ctxs, red, tid, wpre);
        CGemv<CHD, CD>::issue(wrs, lw + (unsigned)(C_OFF_CQ * 4), tid, wpre);   // the cross query's weights land under the exchange + LayerNorm
        C_STAMP(3 + l * 12);
        if (!cluster_exchange(cx, partial, true, tid, x4)) return;
        C_STAMP(4 + l * 12);
        const float v = x + (((x4[tid] + x4[CD + tid]) + (x4[2 * CD + tid] + x4[3 * CD + tid])) + Ly.bo[tid]);
        x = cluster_ln(v, Ly.n1w, Ly.n1b, red8, tid);
        xs[tid] = x;
        __syncthreads();
      }
      // ---- cross attention of head `part` over the audio memory ----
      {
        C_STAMP(5 + l * 12);
        const float y = CGemv<CHD, CD>::finish(wrs, lw + (unsigned)(C_OFF_CQ * 4), xs, red, tid, wpre);
        CGemv<CD, CHD>::issue(wrs, lw + (unsigned)(C_OFF_CO * 4), tid, wpre);
        C_STAMP(6 + l * 12);
        if (tid < CHD) qkv[tid] = y + Ly.bcq[part * CHD + tid];
        __syncthreads();
        const float ctx = cluster_attn(qkv, memK + (size_t)l * Tm * CHD, memV + (size_t)l * Tm * CHD, Tm, mlen, nullptr, sc, part4,
                                       red8, tid);
        if (tid < CHD) ctxs[tid] = ctx;
        __syncthreads();
        C_STAMP(7 + l * 12);
        const float partial = CGemv<CD, CHD>::finish(wrs, lw + (unsigned)(C_OFF_CO * 4), ctxs, red, tid, wpre);
        CGemv<CFQ, CD>::issue(wrs, lw + (unsigned)(C_OFF_W1 * 4), tid, wpre);
        if (!cluster_exchange(cx, partial, true, tid, x4)) return;
        C_STAMP(8 + l * 12);
        const float v = x + (((x4[tid] + x4[CD + tid]) + (x4[2 * CD + tid] + x4[3 * CD + tid])) + Ly.bco[tid]);
        x = cluster_ln(v, Ly.n2w, Ly.n2b, red8, tid);
        xs[tid] = x;
        __syncthreads();
      }
      // ---- feed forward: hidden quarter `part` ----
      {
        C_STAMP(9 + l * 12);
        const float h = CGemv<CFQ, CD>::finish(wrs, lw + (unsigned)(C_OFF_W1 * 4), xs, red, tid, wpre);
        CGemv<CD, CFQ>::issue(wrs, lw + (unsigned)(C_OFF_W2 * 4), tid, wpre);
        hid[tid] = fmaxf(h + Ly.b1[part * CFQ + tid], 0.f);
        __syncthreads();
        C_STAMP(10 + l * 12);
        const float partial = CGemv<CD, CFQ>::finish(wrs, lw + (unsigned)(C_OFF_W2 * 4), hid, red, tid, wpre);
        // what comes next: the following layer's q, k, v weights, or the classifier's first column block
        if (l + 1 < nl) CGemv<192, CD>::issue(wrs, lw + (unsigned)((C_LAYER_FLOATS + C_OFF_QKV) * 4), tid, wpre);
        else if ((p.NLC >> 10) > 0) cls_issue(wrs, cls_off, __builtin_amdgcn_readfirstlane(wave), 0, lane, wpre);
        C_STAMP(11 + l * 12);
        if (!cluster_exchange(cx, partial, true, tid, x4)) return;
        C_STAMP(12 + l * 12);
        const float v = x + (((x4[tid] + x4[CD + tid]) + (x4[2 * CD + tid] + x4[3 * CD + tid])) + Ly.b2[tid]);
        x = cluster_ln(v, Ly.n3w, Ly.n3b, red8, tid);
        xs[tid] = x;
        __syncthreads();
      }
    }
    C_STAMP(40);
    if (part == 0) p.embed[((size_t)row * L + t) * CD + tid] = x;
    // ---- logits of vocabulary quarter `part`: columns part * VQ + [0, VQ) ----
    float bestv = -INFINITY;
    int besti = 0x7fffffff;
    {
      float* lrow = p.logit + ((size_t)row * L + t) * p.V;
      // Groups of FOUR full blocks of 256 columns, one block per WAVE: lane l owns the column quad 4 l .. 4 l + 3 of its wave's
      // block over all 256 k - a wave instruction reads one contiguous 1 KiB row, no k slices, no reduction, one
      // double-buffered stream of sixteen 16-row batches (wpre = the first batch of group 0, requested a stage ago).
      const int nfull = p.NLC >> 8, ngroups = nfull >> 2;
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      for (int g = 0; g < ngroups; ++g) {
        const int blk = 4 * g + wv;
        if (g > 0) cls_issue(wrs, cls_off, blk, 0, lane, wpre);
        f32x4 b[16];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int i = 0; i < 16; i += 2) {
          cls_issue(wrs, cls_off, blk, i + 1, lane, b);
          CGemv<CD, CD>::fma16(wpre, xs + i * 16, acc);
          if (i + 2 < 16) cls_issue(wrs, cls_off, blk, i + 2, lane, wpre);
          CGemv<CD, CD>::fma16(b, xs + (i + 1) * 16, acc);
        }
        const int cb = blk << 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cq = cb + 4 * lane + e, col = part * p.VQ + cq;
          const bool valid = cq < p.VQ && col < p.V;
          lg[cq] = valid ? acc[e] : -INFINITY;
          if (valid) argmax_merge_c(bestv, besti, acc[e], col);
        }
        const int col0 = part * p.VQ + cb + 4 * lane;
        if (cb + 4 * lane + 3 < p.VQ && col0 + 3 < p.V && ((((size_t)row * L + t) * p.V + col0) & 3) == 0) {
          *(f32x4*)(lrow + col0) = acc;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (cb + 4 * lane + e < p.VQ && col0 + e < p.V) lrow[col0 + e] = acc[e];
        }
      }
      auto keep = [&](int cb, int nlb, float y) {
        const int cq = cb + tid, col = part * p.VQ + cq;
        const bool valid = tid < nlb && cq < p.VQ && col < p.V;
        if (tid < nlb) lg[cq] = valid ? y : -INFINITY;
        if (valid) {
          lrow[col] = y;
          argmax_merge_c(bestv, besti, y, col);
        }
      };
      // full blocks that do not fill a group of four, then the last partial block: all 256 threads on one block, k in slices
      for (int blk = 4 * ngroups; blk < nfull; ++blk) {
        const unsigned bw = cls_off + (unsigned)((size_t)blk * 256 * CD * 4);
        CGemv<256, CD>::issue(wrs, bw, tid, wpre);
        keep(blk << 8, 256, CGemv<256, CD>::finish(wrs, bw, xs, red, tid, wpre));
      }
      const int cb = nfull << 8;
      const int rest = p.NLC - cb;   // 0, 64, 128 or 192: uniform over the workgroup
      const unsigned rw = cls_off + (unsigned)((size_t)cb * CD * 4);
      if (rest == 192) { CGemv<192, CD>::issue(wrs, rw, tid, wpre); keep(cb, 192, CGemv<192, CD>::finish(wrs, rw, xs, red, tid, wpre)); }
      else if (rest == 128) { CGemv<128, CD>::issue(wrs, rw, tid, wpre); keep(cb, 128, CGemv<128, CD>::finish(wrs, rw, xs, red, tid, wpre)); }
      else if (rest == 64) { CGemv<64, CD>::issue(wrs, rw, tid, wpre); keep(cb, 64, CGemv<64, CD>::finish(wrs, rw, xs, red, tid, wpre)); }
      // the first rows of the NEXT step's first product land under the arg-max exchange and the embedding lookup
      CGemv<192, CD>::issue(wrs, (unsigned)(C_OFF_QKV * 4), tid, wpre);
    }
    C_STAMP(41);
    // (max, arg max) and sum of exponentials of the quarter (base.py:214-218: log_softmax + max)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bestv, o, 64);
      const int oi = __shfl_xor(besti, o, 64);
      argmax_merge_c(bestv, besti, ov, oi);
    }
    if (lane == 0) { red[wave] = bestv; red[8 + wave] = __int_as_float(besti); }
    __syncthreads();   // also: every logit of the quarter is in lg
    bestv = red[0]; besti = __float_as_int(red[8]);
#pragma unroll
    for (int k = 1; k < 4; ++k) argmax_merge_c(bestv, besti, red[k], __float_as_int(red[8 + k]));
    float se = 0.f;
    for (int c = tid; c < p.NLC; c += 256) se += expf(lg[c] - bestv);   // exp(-inf) = 0 for the padding columns
    se = wave_sum(se);
    __syncthreads();
    if (lane == 0) red[16 + wave] = se;
    __syncthreads();
    se = (red[16] + red[17]) + (red[18] + red[19]);
    // ---- exchange 4: thread 0 the quarter's max, 1 its arg max, 2 its exponential sum, 3 "the reference has stopped" ----
    float mine = 0.f;
    if (tid == 0) mine = bestv;
    else if (tid == 1) mine = __int_as_float(besti);
    else if (tid == 2) mine = se;
    else if (tid == 3) {
      int stop = 0;
      if (part == 0 && t > 0) {
        const unsigned long long pr = __hip_atomic_load(p.prog + t - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stop = ((int)(unsigned)(pr >> 32) == p.stop_rows && (unsigned)pr == 0u) ? 1 : 0;
      }
      mine = __int_as_float(stop);
    }
    C_STAMP(42);
    if (!cluster_exchange(cx, mine, tid < 4, tid, x4)) return;
    C_STAMP(43);
    float gm = x4[0];
    int gi = __float_as_int(x4[1]);
#pragma unroll
    for (int q = 1; q < 4; ++q) argmax_merge_c(gm, gi, x4[q * CD], __float_as_int(x4[q * CD + 1]));
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) tot += x4[q * CD + 2] * expf(x4[q * CD] - gm);
    const int stop = __float_as_int(x4[3]);   // part 0's word
    __syncthreads();
    // ---- stepwise_process_step (base.py:202-211) ----
    const int unf = unfinished && (gi != p.end_idx);
    const int w = unf ? gi : p.end_idx;
    unfinished = unf;
    tok = w;
    if (tid == 0) {
      kmask[t + 1] = (w == p.pad_idx) ? 1 : 0;
      if (part == 0) {
        p.seq[(size_t)row * L + t] = w;
        p.logprob[(size_t)row * L + t] = -logf(tot);
        __hip_atomic_fetch_add(p.prog + t, (1ull << 32) | (unsigned long long)(unf ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    C_STAMP(44);
    if (stop) break;   // every row had emitted <end> before this step: the reference's loop is over (same decision in all parts)
  }
}

__global__ void cluster_pack_kernel(const float* W, long ldw, int r0, int c0, int nvalid_rows, int K, int NL, int ldo, int col0,
                                    float* out) {
  // out[k][col0 + c] = W[r0 + c][c0 + k] for c < NL, k < K (zero where r0 + c >= nvalid_rows)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * NL) return;
  const int c = i % NL, k = i / NL;
  const float v = (r0 + c < nvalid_rows) ? W[(size_t)(r0 + c) * ldw + c0 + k] : 0.f;
  out[(size_t)k * ldo + col0 + c] = v;
}

inline int cluster_vq(int V) { return (V + 3) / 4; }
inline int cluster_nlc(int V) { return (cluster_vq(V) + 63) / 64 * 64; }
inline size_t cluster_part_floats(const ac_trm_weights* w) { return (size_t)w->nlayers * C_LAYER_FLOATS + (size_t)CD * cluster_nlc(w->vocab); }

inline int cluster_shape_ok(const ac_trm_weights* w) {
  return w && w->d_model == CD && w->nhead == 4 && w->dim_ff == CFF && w->nlayers >= 1 && w->nlayers <= AC_MAX_LAYERS &&
         w->vocab >= 4 && 3 * cluster_vq(w->vocab) < w->vocab &&   // every quarter owns a column (V = 5, 6, 9: the last one would not)
         cluster_nlc(w->vocab) <= 8192;
}

inline size_t cluster_lds_bytes(int nlayers, int Tm, int max_len, int NLC) {
  const int nsc = ((Tm > CMAXL ? Tm : CMAXL) + 3) & ~3;
  const size_t floats = CD + 192 + CHD + CFQ + 5 * CD + nsc + 4 * CHD + 16 + 4 * CD + NLC + (size_t)2 * nlayers * max_len * CHD +
                        (size_t)2 * nlayers * Tm * CHD;
  return floats * 4 + ((max_len + 1 + 15) & ~15);
}

}  // namespace

// C ABI: see include/audiocaption_hip.h
extern "C" long ac_trm_cluster_pack_floats(const ac_trm_weights* w) {
  if (!cluster_shape_ok(w)) return -1;
  return (long)(4 * cluster_part_floats(w));
}

extern "C" int ac_trm_cluster_pack(const ac_trm_weights* w, float* out, void* stream) {
  if (!cluster_shape_ok(w) || !out) return AC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t pf = cluster_part_floats(w);
  auto pack = [&](const float* W, long ldw, int r0, int c0, int nrows, int K, int NL, int ldo, int col0, float* dst) {
    hipLaunchKernelGGL(cluster_pack_kernel, dim3((unsigned)((K * NL + 255) / 256)), dim3(256), 0, s, W, ldw, r0, c0, nrows, K, NL, ldo,
                       col0, dst);
    return ac_check_launch();
  };
  const int V = w->vocab, VQ = cluster_vq(V), NLC = cluster_nlc(V);
  for (int part = 0; part < 4; ++part) {
    float* pb = out + (size_t)part * pf;
    for (int l = 0; l < w->nlayers; ++l) {
      const ac_trm_layer& L = w->layer[l];
      float* lb = pb + (size_t)l * C_LAYER_FLOATS;
      for (int which = 0; which < 3; ++which)   // q, k, v rows of head `part`: columns 64 which .. + 63 of the [256][192] blob
        if (pack(L.sa_in_w, CD, which * CD + part * CHD, 0, 3 * CD, CD, CHD, 192, which * CHD, lb + C_OFF_QKV) != AC_OK) return AC_ERR_LAUNCH;
      if (pack(L.sa_out_w, CD, 0, part * CHD, CD, CHD, CD, CD, 0, lb + C_OFF_O) != AC_OK) return AC_ERR_LAUNCH;
      if (pack(L.ca_in_w, CD, part * CHD, 0, CD, CD, CHD, CHD, 0, lb + C_OFF_CQ) != AC_OK) return AC_ERR_LAUNCH;   // query rows of in_proj
      if (pack(L.ca_out_w, CD, 0, part * CHD, CD, CHD, CD, CD, 0, lb + C_OFF_CO) != AC_OK) return AC_ERR_LAUNCH;
      if (pack(L.l1_w, CD, part * CFQ, 0, CFF, CD, CFQ, CFQ, 0, lb + C_OFF_W1) != AC_OK) return AC_ERR_LAUNCH;
      if (pack(L.l2_w, CFF, 0, part * CFQ, CD, CFQ, CD, CD, 0, lb + C_OFF_W2) != AC_OK) return AC_ERR_LAUNCH;
    }
    float* cb = pb + (size_t)w->nlayers * C_LAYER_FLOATS;
    for (int c0 = 0; c0 < NLC; c0 += 256) {
      const int nlb = NLC - c0 < 256 ? NLC - c0 : 256;
      // rows of the classifier beyond this quarter (or beyond V) are zero columns of the block
      const int last = part * VQ + VQ < V ? part * VQ + VQ : V;
      if (pack(w->cls_w, CD, part * VQ + c0, 0, last, CD, nlb, nlb, 0, cb + (size_t)c0 * CD) != AC_OK) return AC_ERR_LAUNCH;
    }
  }
  return AC_OK;
}

extern "C" long ac_trm_cluster_workspace_bytes(int B) {
  if (B <= 0) return AC_ERR_ARG;
  return 512 + (long)B * 2 * CPARTS * 256 * 8;
}

extern "C" int ac_trm_greedy_cluster(const ac_trm_weights* w, const float* cluster_pk, const float* memkv, const int* mem_len,
                                     int B, int Tm, int max_len, int start_idx, int end_idx, int pad_idx, int64_t* seq,
                                     float* logit, float* logprob, float* embed, int* unfinished_cnt, void* workspace,
                                     int early_stop, void* stream) {
  if (!cluster_shape_ok(w) || !cluster_pk || !memkv || !mem_len || !seq || !logit || !logprob || !embed || !unfinished_cnt ||
      !workspace)
    return AC_ERR_ARG;
  if (B <= 0 || Tm <= 0 || max_len <= 0 || max_len > CMAXL || max_len > w->max_pos) return AC_ERR_ARG;
  if ((long)B * max_len > 1024L * 256) return AC_ERR_ARG;   // cluster_init_kernel: one thread per output element
  const size_t lds = cluster_lds_bytes(w->nlayers, Tm, max_len, cluster_nlc(w->vocab));
  if (lds > 159 * 1024) return AC_ERR_ARG;   // (+ the kernel's 16 bytes of static LDS)
  hipStream_t s = (hipStream_t)stream;
  ClusterParams p;
  p.pk = cluster_pk; p.part_floats = cluster_part_floats(w);
  p.emb = w->emb; p.pe = w->pe;
  for (int l = 0; l < w->nlayers; ++l) {
    const ac_trm_layer& L = w->layer[l];
    p.L[l] = {L.sa_in_b, L.sa_out_b, L.ca_in_b, L.ca_out_b, L.l1_b, L.l2_b, L.n1_w, L.n1_b, L.n2_w, L.n2_b, L.n3_w, L.n3_b};
  }
  p.nlayers = w->nlayers; p.V = w->vocab; p.VQ = cluster_vq(w->vocab); p.NLC = cluster_nlc(w->vocab);
  p.memkv = memkv; p.mem_len = mem_len; p.B = B; p.Tm = Tm; p.max_len = max_len;
  p.start_idx = start_idx; p.end_idx = end_idx; p.pad_idx = pad_idx;
  p.stop_rows = early_stop ? B : -1;
  p.seq = seq; p.logit = logit; p.logprob = logprob; p.embed = embed; p.cnt = unfinished_cnt;
  {   // read per call (tests shorten the bound and inject a lost partner; a captured graph keeps what it was captured with)
    const char* e = getenv("AUDIOCAPTION_CLUSTER_TIMEOUT_US");
    const long long us = e ? atoll(e) : 0;
    p.spin_ticks = us > 0 ? us * 100 : C_SPIN_TICKS;
    const char* f = getenv("AUDIOCAPTION_CLUSTER_FAULT");
    p.fault = f ? atoi(f) : 0;
  }
  p.error = (unsigned*)workspace;
  p.ticket = (unsigned*)((char*)workspace + 64);
  p.prog = (unsigned long long*)((char*)workspace + 128);
  p.xch = (unsigned long long*)((char*)workspace + 512);
  const unsigned xch_words = (unsigned)((size_t)B * 2 * CPARTS * 256);
  // tickets, arrival counters, granules and the outputs' initial values by a KERNEL (a memset node of a captured graph was
  // not reliably ordered before the kernel that follows it: csrc/gru.hip)
  {
    unsigned blocks = (xch_words + 255) / 256;
    const unsigned need = (unsigned)((B * max_len + 255) / 256);
    if (blocks < need) blocks = need;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(cluster_init_kernel, dim3(blocks), dim3(256), 0, s, p, xch_words);
    if (ac_check_launch() != AC_OK) return AC_ERR_LAUNCH;
  }
  static AcLdsAttr lds_attr;   // per device
  if (ac_allow_lds((const void*)greedy_cluster_kernel, 159 * 1024, &lds_attr) != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(greedy_cluster_kernel, dim3(4 * B), dim3(256), lds, s, p);
  if (ac_check_launch() != AC_OK) return AC_ERR_LAUNCH;
  hipLaunchKernelGGL(cluster_finalize_kernel, dim3((B * max_len + 255) / 256), dim3(256), 0, s, seq, logprob, unfinished_cnt, p.prog,
                     B, max_len, end_idx);
  return ac_check_launch();
}
