// Autoregressive Transformer decoder on gfx950: memory preparation, the per-position decoder step with
// a self-attention KV cache, on-device greedy search and the device half of beam search.
//
// Replaces reference TransformerDecoder.forward (transformer_decoder.py:80-103), the decode loops
// CaptionModel.stepwise_forward / sample_next_word / beam_search (base.py:152-325) and
// TransformerModel.prepare_decoder_input (transformer_model.py:34-86).
//
// What is different from the reference's schedule (results are the same function of the inputs):
//  * the reference re-runs the decoder on the WHOLE prefix every step and recomputes attn_proj and
//    the cross-attention K/V projections of the audio memory on every call; here the memory side is
//    computed once per batch (ac_trm_memory) and each step processes only the new position against
//    cached self-attention K/V (the causal mask makes position t independent of later tokens);
//  * the greedy loop never returns to the host: argmax, log-prob, <end> bookkeeping and the
//    "every clip finished" test live in one kernel per step, so a whole decode is a fixed launch
//    sequence on one stream.
// Projections run on the f32 matrix cores through ac_linear; softmax / LayerNorm reductions are
// 64-lane wavefront shuffles.
#include "ac_common.h"
#include "../../include/audiocaption_hip.h"
#include <stdlib.h>
#include <string.h>

extern "C" int ac_linear(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K,
                         long ldx, long ldw, long ldy, int relu, void* stream);
// csrc/decoder_wide.hip: the projections of a step over >= AUDIOCAPTION_DEC_WIDE_MIN rows
extern "C" long ac_dec_wide_packed_floats(int N, int K);
extern "C" int ac_dec_wide_pack(const float* W, long ldw, int N, int K, float* out, void* stream);
extern "C" int ac_dec_wide_gemm(int producer, const float* X, long ldx, const float* Y2, long ldy2, const float* ln_w,
                                const float* ln_b, const int* tok, long tok_stride, int t, const float* emb, const float* pe,
                                float emb_scale, float* xout, long ldxo, const float* Wp, const float* bias, float* Y, long ldy,
                                int M, int N, int K, int relu, int ntb, int split_out, void* stream);

// DEC_REGCAP: registers per lane the decode chain's kernels may use (development switch: -DDEC_REGCAP=64 builds them to fit
// beside a one-wave-per-SIMD conv workgroup, which leaves 64 of a SIMD's 512 registers - tools/tax_probe.py)
#ifndef DEC_REGCAP
#define DEC_REGCAP 0
#endif
#if DEC_REGCAP
#define DEC_CAP __attribute__((amdgpu_waves_per_eu(512 / DEC_REGCAP, 8)))
#define DEC_ROW_BOUNDS __launch_bounds__(256) DEC_CAP
#else
#define DEC_CAP
#define DEC_ROW_BOUNDS __launch_bounds__(256, 2)
#endif

namespace {

// ---------------------------------------------------------------------------------------------
// x[r][:] = E[tok[r][t]] * sqrt(d) + pe[t]                     (transformer_decoder.py:89-91)
// ---------------------------------------------------------------------------------------------
__global__ void embed_pe_kernel(const int* tok, long tok_stride, int t, const float* emb, const float* pe,
                                float scale, float* x, int d) {
  const int r = blockIdx.x;
  const int w = tok[(size_t)r * tok_stride + t];
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    x[(size_t)r * d + c] = emb[(size_t)w * d + c] * scale + pe[(size_t)t * d + c];
}

// ---------------------------------------------------------------------------------------------
// out = LayerNorm(x + y) * w + b, one wave per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float* x, const float* y, const float* w,
                                                            const float* b, float* out, int rows, int d,
                                                            long ldx, long ldy, long ldo) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  const float* yr = y ? y + (size_t)row * ldy : nullptr;
  float v[16];  // d <= 1024
  float s = 0.f;
  int n = 0;
  for (int c = lane; c < d; c += 64, ++n) {
    v[n] = xr[c] + (yr ? yr[c] : 0.f);
    s += v[n];
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
  for (int i = 0; i < n; ++i) {
    const float dlt = v[i] - mean;
    q = fmaf(dlt, dlt, q);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
  n = 0;
  for (int c = lane; c < d; c += 64, ++n) out[(size_t)row * ldo + c] = (v[n] - mean) * rstd * w[c] + b[c];
}

// ---------------------------------------------------------------------------------------------
// Single-query multi-head attention for one decode position.  grid (rows, heads), one wave each.
//   self  : keys 0..t from the KV cache (the new k/v row is appended here), key j masked when
//           key_mask[r][j] != 0                      (tgt_key_padding_mask, transformer_model.py:55)
//   cross : keys 0..Tm-1 of the projected audio memory, key j masked when j >= key_len[r / row_div]
//           (memory_key_padding_mask, transformer_decoder.py:94)
// ---------------------------------------------------------------------------------------------
constexpr int MAX_KEYS = 1024;

struct AttnParams {
  const float* q; long ldq;
  const float* K; const float* V;        // key j of kv-row R at K + R*row_stride + j*key_stride
  long row_stride, key_stride;
  int row_div, nkeys;
  const int* key_len;                     // per kv-row valid length or null
  const unsigned char* key_mask; long mask_stride;  // per row or null
  const float* new_k; const float* new_v; long ld_new;  // appended at key index nkeys-1 when non-null
  float* Kw; float* Vw;                   // writable cache base (same geometry as K/V) for the append
  float* out; long ldo;
  int hd; float scale;
  // wide route (csrc/decoder_wide.hip): the context rows leave as three bf16 planes in MFMA fragment order instead, the A
  // operand of the out-projection launch: pack[row / 32][k step of 16][plane][lane = row % 32 + 32 * (k % 16 / 8)][k % 8]
  unsigned char* out_pk = nullptr; int pk_kst = 0;
};

__global__ __launch_bounds__(64) DEC_CAP void attn_step_kernel(AttnParams p) {
  __shared__ float sc[MAX_KEYS];
  const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const int kr = r / p.row_div;
  const bool act = lane < p.hd;
  const size_t hoff = (size_t)h * p.hd;
  if (p.new_k && act) {
    const size_t dst = (size_t)kr * p.row_stride + (size_t)(p.nkeys - 1) * p.key_stride + hoff + lane;
    p.Kw[dst] = p.new_k[(size_t)r * p.ld_new + hoff + lane];
    p.Vw[dst] = p.new_v[(size_t)r * p.ld_new + hoff + lane];
  }
  __syncthreads();
  const float* qp = p.q + (size_t)r * p.ldq + hoff;
  const float* Kb = p.K + (size_t)kr * p.row_stride + hoff;
  const float* Vb = p.V + (size_t)kr * p.row_stride + hoff;
  const int klen = p.key_len ? p.key_len[kr] : p.nkeys;
  if (lane < 32 && lane >= p.nkeys) sc[lane] = 0.f;  // the unrolled value loop reads sc[0..31]
  // scores: a wave-instruction covers FOUR keys (16 lanes x 16 bytes = one 256-byte K row each), so a
  // fragment costs 8 cache-line requests; the 4-element partial dots are reduced over the 16-lane groups
  if (p.hd == 64) {
    const int ks = lane >> 4, d4 = lane & 15;
    const f32x4 q4 = *(const f32x4*)(qp + 4 * d4);
#pragma unroll 4
    for (int j0 = 0; j0 < p.nkeys; j0 += 4) {
      const int j = j0 + ks;
      float s = 0.f;
      if (j < p.nkeys) {
        const f32x4 kv = *(const f32x4*)(Kb + (size_t)j * p.key_stride + 4 * d4);
        s = (q4[0] * kv[0] + q4[1] * kv[1]) + (q4[2] * kv[2] + q4[3] * kv[3]);
      }
      s = row16_sum(s);
      if (d4 == 0 && j < p.nkeys) {
        const bool masked = (j >= klen) || (p.key_mask && p.key_mask[(size_t)r * p.mask_stride + j]);
        sc[j] = masked ? -INFINITY : s * p.scale;
      }
    }
  } else {
    for (int j = lane; j < p.nkeys; j += 64) {
      const float* kp = Kb + (size_t)j * p.key_stride;
      float s = 0.f;
      for (int c = 0; c < p.hd; ++c) s = fmaf(qp[c], kp[c], s);
      const bool masked = (j >= klen) || (p.key_mask && p.key_mask[(size_t)r * p.mask_stride + j]);
      sc[j] = masked ? -INFINITY : s * p.scale;
    }
  }
  // the value rows do not depend on the softmax: request them now (coalesced, one row per instruction) so
  // their latency overlaps the score reductions
  constexpr int VPRE = 32;
  float vpre[VPRE];
#pragma unroll
  for (int j = 0; j < VPRE; ++j) vpre[j] = (act && j < p.nkeys) ? Vb[(size_t)j * p.key_stride + lane] : 0.f;
  __syncthreads();
  float m = -INFINITY;
  for (int j = lane; j < p.nkeys; j += 64) m = fmaxf(m, sc[j]);
  m = wave_max(m);
  float den = 0.f;
  for (int j = lane; j < p.nkeys; j += 64) {
    const float e = expf(sc[j] - m);
    sc[j] = e;
    den += e;
  }
  den = wave_sum(den);
  __syncthreads();
  if (act) {
    // lane d owns output channel d
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
    for (int j = 0; j < VPRE; j += 4) {
      o0 = fmaf(sc[j], vpre[j], o0);  // sc[j] = 0 and vpre[j] = 0 beyond nkeys
      o1 = fmaf(sc[j + 1], vpre[j + 1], o1);
      o2 = fmaf(sc[j + 2], vpre[j + 2], o2);
      o3 = fmaf(sc[j + 3], vpre[j + 3], o3);
    }
    for (int j = VPRE; j < p.nkeys; ++j) o0 = fmaf(sc[j], Vb[(size_t)j * p.key_stride + lane], o0);
    const float val = ((o0 + o1) + (o2 + o3)) / den;
    if (p.out_pk) {
      const int c = (int)hoff + lane;
      unsigned short* d = (unsigned short*)(p.out_pk + ((size_t)(r >> 5) * p.pk_kst + (c >> 4)) * 3072 +
                                            (size_t)((r & 31) + 32 * ((c >> 3) & 1)) * 16 + (c & 7) * 2);
      // RNE to bf16, three times over the exact remainders (the split of csrc/decoder_wide.hip, one value at a time)
      auto rne = [](float x) { const unsigned u = __builtin_bit_cast(unsigned, x); return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; };
      const unsigned h0 = rne(val);
      const float r1 = val - __builtin_bit_cast(float, h0);
      const unsigned h1 = rne(r1);
      const float r2 = r1 - __builtin_bit_cast(float, h1);
      const unsigned h2 = rne(r2);
      d[0] = (unsigned short)(h0 >> 16);
      d[512] = (unsigned short)(h1 >> 16);
      d[1024] = (unsigned short)(h2 >> 16);
    } else {
      p.out[(size_t)r * p.ldo + hoff + lane] = val;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// One attention sub-layer of a decode step for ONE row per workgroup (d_model 256, 4 heads of 64): everything after the
// QKV projection is row-local, so the four launches attention -> out-projection -> residual + LayerNorm -> next
// projection become one.  Wave h runs head h of the single-query attention (the same code path as attn_step_kernel);
// the 256-float context then stays in LDS; the out-projection is a matrix-vector product against a TRANSPOSED copy of
// the weights (WoT[k][n]: wave w takes k in [64 w, 64 w + 64), lane l the four outputs 4 l .. 4 l + 3, so a wave
// instruction reads one contiguous 1 KiB row and 16 such loads are in flight per lane), reduced over the waves in LDS;
// then the residual join + LayerNorm, and for the self-attention sub-layer the cross-attention query projection of the
// fresh row.  A workgroup streams 256 KiB per projection from L2 (64 workgroups: 16 MB per launch, ~2 us), which
// replaces two 5.5-7 us launches whose cost was launch boundary + first-touch latency, not arithmetic.
// ---------------------------------------------------------------------------------------------
constexpr int ROW_D = 256, ROW_H = 4;

struct RowParams {
  AttnParams a;                          // attention geometry; a.out is unused
  const float* res; long ldres;          // residual input rows
  const float* WoT; const float* bo;     // out-projection, transposed [k][n]
  const float* ln_w; const float* ln_b;
  float* xout; long ldxo;                // LayerNorm(res + attention output)
  const float* WqT; const float* bq;     // optional next projection of the normalised row (self: the cross query)
  float* qout; long ldqo;
};

constexpr int ATT_KPRE = 8, ATT_VPRE = 32;
struct AttnLoads {            // what a wave has requested for one attention phase (one head of one row)
  f32x4 kpre[ATT_KPRE];       // key rows of the first 32 keys: lane (ks = l / 16, d4 = l % 16) holds dims 4 d4 .. + 3 of key 4 i + ks
  float vpre[ATT_VPRE];       // value column `lane` of the first 32 keys
  unsigned char mpre[ATT_KPRE];   // key_mask of those keys (0 without a mask)
  int klen;                   // keys of this row that exist (key_len, or all)
};

// Request everything head h of row r reads from memory - NOT the query: it may not exist yet.  The newest key / value
// (this step's projection, which also goes into the cache for the steps to come) is read from where the projection left it
// instead of being stored, fenced and read back.
__device__ __forceinline__ void attn_issue_k(const AttnParams& p, int r, int h, int lane, f32x4 (&kpre)[ATT_KPRE]) {
  const int kr = r / p.row_div;
  const size_t hoff = (size_t)h * 64;
  const float* Kb = p.K + (size_t)kr * p.row_stride + hoff;
  const float* nk = p.new_k ? p.new_k + (size_t)r * p.ld_new + hoff : nullptr;
  const int newest = p.new_k ? p.nkeys - 1 : -1;
  const int ks = lane >> 4, d4 = lane & 15;
  // branch-free: a key beyond the last one re-reads the last one (its score is never used), so that the number of loads in
  // flight is known to the compiler and a later wait can leave the younger ones in flight
  const int last = p.nkeys - 1;
#pragma unroll
  for (int i = 0; i < ATT_KPRE; ++i) {
    const int j = 4 * i + ks < last ? 4 * i + ks : last;
    const float* src = j == newest ? nk + 4 * d4 : Kb + (size_t)j * p.key_stride + 4 * d4;
    kpre[i] = *(const f32x4*)src;
  }
}

__device__ __forceinline__ void attn_issue_v(const AttnParams& p, int r, int h, int lane, float (&vpre)[ATT_VPRE]) {
  const int kr = r / p.row_div;
  const size_t hoff = (size_t)h * 64;
  const float* Vb = p.V + (size_t)kr * p.row_stride + hoff;
  const float* nk = p.new_k ? p.new_k + (size_t)r * p.ld_new + hoff : nullptr;
  const float* nv = p.new_k ? p.new_v + (size_t)r * p.ld_new + hoff : nullptr;
  const int newest = p.new_k ? p.nkeys - 1 : -1;
  const int last = p.nkeys - 1;
#pragma unroll
  for (int j = 0; j < ATT_VPRE; ++j) {   // branch-free like the keys; columns beyond the last key are zeroed below
    const int jj = j < last ? j : last;
    const float* src = jj == newest ? nv + lane : Vb + (size_t)jj * p.key_stride + lane;
    vpre[j] = *src;
  }
#pragma unroll
  for (int j = 0; j < ATT_VPRE; ++j) vpre[j] = j < p.nkeys ? vpre[j] : 0.f;
  if (p.new_k) {   // the cache rows of this step, for the steps to come (nobody in this launch reads them back)
    const size_t dst = (size_t)kr * p.row_stride + (size_t)newest * p.key_stride + hoff + lane;
    p.Kw[dst] = nk[lane];
    p.Vw[dst] = nv[lane];
  }
}

__device__ __forceinline__ void attn_issue(const AttnParams& p, int r, int h, int lane, AttnLoads& L) {
  attn_issue_k(p, r, h, lane, L.kpre);
  {   // the mask bytes and the row's key count travel with the keys (they used to be a second round trip inside the score loop)
    const int ks = lane >> 4, last = p.nkeys - 1;
#pragma unroll
    for (int i = 0; i < ATT_KPRE; ++i) {
      const int j = 4 * i + ks < last ? 4 * i + ks : last;
      L.mpre[i] = p.key_mask ? p.key_mask[(size_t)r * p.mask_stride + j] : (unsigned char)0;
    }
    L.klen = p.key_len ? p.key_len[r / p.row_div] : p.nkeys;
  }
  attn_issue_v(p, r, h, lane, L.vpre);
}

// Head h of row r, one wave, from the requested rows; returns output channel `lane` (< 64) of the head.  The arithmetic
// of attn_step_kernel.
__device__ __forceinline__ float attn_compute(const AttnParams& p, int r, int h, int lane, float* sc, const AttnLoads& L) {
  const int kr = r / p.row_div;
  const size_t hoff = (size_t)h * 64;
  const float* qp = p.q + (size_t)r * p.ldq + hoff;
  const float* Kb = p.K + (size_t)kr * p.row_stride + hoff;
  const float* Vb = p.V + (size_t)kr * p.row_stride + hoff;
  const float* nk = p.new_k ? p.new_k + (size_t)r * p.ld_new + hoff : nullptr;
  const float* nv = p.new_k ? p.new_v + (size_t)r * p.ld_new + hoff : nullptr;
  const int newest = p.new_k ? p.nkeys - 1 : -1;
  const int klen = L.klen;
  const int ks = lane >> 4, d4 = lane & 15;
  const f32x4 q4 = *(const f32x4*)(qp + 4 * d4);
  if (lane < 32 && lane >= p.nkeys) sc[lane] = 0.f;
#pragma unroll 4
  for (int j0 = 0; j0 < p.nkeys; j0 += 4) {
    const int j = j0 + ks;
    float s = 0.f;
    unsigned char mk = 0;
    if (j < p.nkeys) {
      f32x4 kv;
      if (j0 < 4 * ATT_KPRE) {
#pragma unroll
        for (int i = 0; i < ATT_KPRE; ++i)
          if (j0 == 4 * i) { kv = L.kpre[i]; mk = L.mpre[i]; }
      } else {
        kv = *(const f32x4*)(j == newest ? nk + 4 * d4 : Kb + (size_t)j * p.key_stride + 4 * d4);
        mk = p.key_mask ? p.key_mask[(size_t)r * p.mask_stride + j] : (unsigned char)0;
      }
      s = (q4[0] * kv[0] + q4[1] * kv[1]) + (q4[2] * kv[2] + q4[3] * kv[3]);
    }
    s = row16_sum(s);
    if (d4 == 0 && j < p.nkeys) {
      const bool masked = (j >= klen) || mk;
      sc[j] = masked ? -INFINITY : s * p.scale;
    }
  }
  __syncthreads();
  float m = -INFINITY;
  for (int j = lane; j < p.nkeys; j += 64) m = fmaxf(m, sc[j]);
  m = wave_max(m);
  float den = 0.f;
  for (int j = lane; j < p.nkeys; j += 64) {
    const float e = expf(sc[j] - m);
    sc[j] = e;
    den += e;
  }
  den = wave_sum(den);
  __syncthreads();
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
  for (int j = 0; j < ATT_VPRE; j += 4) {
    o0 = fmaf(sc[j], L.vpre[j], o0);
    o1 = fmaf(sc[j + 1], L.vpre[j + 1], o1);
    o2 = fmaf(sc[j + 2], L.vpre[j + 2], o2);
    o3 = fmaf(sc[j + 3], L.vpre[j + 3], o3);
  }
  for (int j = ATT_VPRE; j < p.nkeys; ++j) o0 = fmaf(sc[j], j == newest ? nv[lane] : Vb[(size_t)j * p.key_stride + lane], o0);
  return ((o0 + o1) + (o2 + o3)) / den;
}

__device__ __forceinline__ float attn_head_row(const AttnParams& p, int r, int h, int lane, float* sc, bool active) {
  (void)active;
  AttnLoads L;
  attn_issue(p, r, h, lane, L);
  return attn_compute(p, r, h, lane, sc, L);
}

// Matrix-vector product against a transposed 256 x 256 matrix, split over the 4 waves of the workgroup: wave w owns
// k in [64 w, 64 w + 64), lane l the outputs 4 l .. 4 l + 3, so a wave instruction reads one contiguous 1 KiB row; 16 rows
// are in flight per lane.  Every thread gets y[tid] = bias[tid] + sum_k WT[k][tid] x[k]; x in LDS, part = [4][256] scratch.
// (Register budget on purpose: in the throughput schedule this kernel has to fit into the slot ONE finished conv
// workgroup leaves on a CU - 4 waves of <= 240 VGPRs.  Variants with 32 + 32 rows in flight over 8 waves, or with the
// rows requested before the attention (256 VGPRs), were ~1 us faster alone and cost the overlapped pipeline 0.7-0.8 ms
// per step, because their workgroups waited for whole CUs.)
__device__ __forceinline__ float row_gemv256(const float* WT, const float* bias, const float* x, float* part, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const f32x4* wp = (const f32x4*)(WT + (size_t)(wave * 64) * ROW_D) + lane;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#if defined(DEC_PK_PROBE)
  // Development (tools/pk_rootcause.py, packed-f32 builds only): the same product with every weight row of a 16-row batch
  // LANDED before the first multiply-add touches it.  DEC_PK_PROBE >= 1: explicit s_waitcnt vmcnt(0) lgkmcnt(0) between the
  // loads and the arithmetic (inline asm: invisible to the compiler's own wait-count pass, which keeps its partial
  // vmcnt(N) waits as well); 2: a workgroup-scope fence on top.
  for (int k0 = 0; k0 < 64; k0 += 16) {
    f32x4 wv[16];
    float xv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      wv[k] = wp[(size_t)(k0 + k) * (ROW_D / 4)];
      xv[k] = x[wave * 64 + k0 + k];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#if DEC_PK_PROBE == 2
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#endif
#if DEC_PK_PROBE == 3   // 3: sixteen idle issue slots between the landed loads and the first packed multiply-add
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#if DEC_PK_PROBE == 4   // 4: the packed instructions never read a register a LOAD wrote: every row goes through a plain v_mov first
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("v_mov_b32 %0, %1" : "=v"(wv[k][e]) : "v"(wv[k][e]));
      asm volatile("v_mov_b32 %0, %1" : "=v"(xv[k]) : "v"(xv[k]));
#endif
      asm volatile("" : "+v"(wv[k]));   // the arithmetic below cannot be hoisted above the wait
      acc[0] = fmaf(wv[k][0], xv[k], acc[0]);
      acc[1] = fmaf(wv[k][1], xv[k], acc[1]);
      acc[2] = fmaf(wv[k][2], xv[k], acc[2]);
      acc[3] = fmaf(wv[k][3], xv[k], acc[3]);
    }
  }
#else
#pragma unroll 16
  for (int k = 0; k < 64; ++k) {
    const f32x4 wv = wp[(size_t)k * (ROW_D / 4)];
    const float xv = x[wave * 64 + k];
    acc[0] = fmaf(wv[0], xv, acc[0]);
    acc[1] = fmaf(wv[1], xv, acc[1]);
    acc[2] = fmaf(wv[2], xv, acc[2]);
    acc[3] = fmaf(wv[3], xv, acc[3]);
  }
#endif
  *(f32x4*)(part + wave * ROW_D + 4 * lane) = acc;
  __syncthreads();
  const float y = ((part[tid] + part[ROW_D + tid]) + (part[2 * ROW_D + tid] + part[3 * ROW_D + tid])) + (bias ? bias[tid] : 0.f);
  __syncthreads();   // part is reused by the next projection
  return y;
}

#ifdef AC_ROW_STAMPS   // development (tools/row_stamps.py): phase timestamps (100 MHz) of workgroup 0 of dec_row2_kernel
__device__ unsigned long long g_row_stamps[16];
#define ROW_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_row_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int ac_row_stamps_read(unsigned long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_row_stamps), sizeof(g_row_stamps)) == hipSuccess ? 0 : -2;
}
#else
#define ROW_STAMP(k) do { } while (0)
#endif

__global__ DEC_ROW_BOUNDS void dec_row_kernel(RowParams p) {
  __shared__ float sc[ROW_H][MAX_KEYS];
  __shared__ __attribute__((aligned(16))) float sx[ROW_D];
  __shared__ __attribute__((aligned(16))) float part[4 * ROW_D];
  __shared__ float red[8];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float resv = p.res[(size_t)r * p.ldres + tid];
  sx[tid] = attn_head_row(p.a, r, wave, lane, sc[wave], true);
  __syncthreads();
  const float v = resv + row_gemv256(p.WoT, p.bo, sx, part, tid);
  // LayerNorm over the 256 values (one per thread)
  const float s1 = wave_sum(v);
  if (lane == 0) red[wave] = s1;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / ROW_D);
  const float dl = v - mean;
  const float s2 = wave_sum(dl * dl);
  if (lane == 0) red[4 + wave] = s2;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / ROW_D) + 1e-5f);
  const float xn = dl * rstd * p.ln_w[tid] + p.ln_b[tid];
  p.xout[(size_t)r * p.ldxo + tid] = xn;
  if (p.WqT) {
    sx[tid] = xn;
    __syncthreads();
    p.qout[(size_t)r * p.ldqo + tid] = row_gemv256(p.WqT, p.bq, sx, part, tid);
  }
}

// Both attention sub-layers of a decoder layer for one row in ONE launch: self attention -> out-projection -> LN1 -> cross
// query projection (dec_row_kernel with WqT) and then, with the query still in LDS, cross attention -> out-projection ->
// LN2 (dec_row_kernel on the audio memory).  Everything after the QKV projection is row-local, so the second launch only
// bought a kernel boundary and a global round trip of the 256-float query.  The same functions in the same order: the
// same bits as the two launches.
__global__ DEC_ROW_BOUNDS void dec_row2_kernel(RowParams p1, RowParams p2) {
  __shared__ float sc[ROW_H][MAX_KEYS];
  __shared__ __attribute__((aligned(16))) float sx[ROW_D];
  __shared__ __attribute__((aligned(16))) float sq[ROW_D];
  __shared__ __attribute__((aligned(16))) float part[4 * ROW_D];
  __shared__ float red[8];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float xn;
  ROW_STAMP(0);
  const float resv = p1.res[(size_t)r * p1.ldres + tid];
  sx[tid] = attn_head_row(p1.a, r, wave, lane, sc[wave], true);
  // (Requesting the cross attention's rows up here as well - they depend on nothing computed in this kernel - was tried:
  // held in registers they make the compiler serialise the matrix-vector products' loads (2.3 -> 14 us each); parked in
  // LDS the cross phase drops from 4.2 to 1.9 us and the self phase grows by as much, waiting for them to land.)
  {
    __syncthreads();
    ROW_STAMP(1);
    const float v = resv + row_gemv256(p1.WoT, p1.bo, sx, part, tid);
    ROW_STAMP(2);
    const float s1 = wave_sum(v);
    if (lane == 0) red[wave] = s1;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / ROW_D);
    const float dl = v - mean;
    const float s2 = wave_sum(dl * dl);
    if (lane == 0) red[4 + wave] = s2;
    __syncthreads();
    const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / ROW_D) + 1e-5f);
    xn = dl * rstd * p1.ln_w[tid] + p1.ln_b[tid];
    p1.xout[(size_t)r * p1.ldxo + tid] = xn;
    sx[tid] = xn;
    __syncthreads();
    ROW_STAMP(3);
    sq[tid] = row_gemv256(p1.WqT, p1.bq, sx, part, tid);
    __syncthreads();
    ROW_STAMP(4);
  }
  AttnParams a2 = p2.a;
  a2.q = sq;            // the cross query of this row, still in LDS
  a2.ldq = 0;
  sx[tid] = attn_head_row(a2, r, wave, lane, sc[wave], true);
  __syncthreads();
  ROW_STAMP(5);
  const float v = xn + row_gemv256(p2.WoT, p2.bo, sx, part, tid);
  ROW_STAMP(6);
  const float s1 = wave_sum(v);
  __syncthreads();      // red[] of the first half has been read by every thread
  if (lane == 0) red[wave] = s1;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) * (1.0f / ROW_D);
  const float dl = v - mean;
  const float s2 = wave_sum(dl * dl);
  if (lane == 0) red[4 + wave] = s2;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) * (1.0f / ROW_D) + 1e-5f);
  p2.xout[(size_t)r * p2.ldxo + tid] = dl * rstd * p2.ln_w[tid] + p2.ln_b[tid];
  ROW_STAMP(7);
}

// WT[k][n] = W[n][k] for a d x d matrix (rows n of W may be a slice of a taller matrix: ldw)
__global__ void transpose_sq_kernel(const float* W, long ldw, int d, float* WT) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) tile[i][threadIdx.x] = W[(size_t)(by + i) * ldw + bx + threadIdx.x];
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) WT[(size_t)(bx + i) * d + by + threadIdx.x] = tile[threadIdx.x][i];
}

// ---------------------------------------------------------------------------------------------
// Decode-step projection  Y[R, N] = act(A[R, K] W[N, K]^T + bias)  with the producer of A fused in:
//   PRO_PLAIN : A = X                                   (rows of a previous kernel's output)
//   PRO_EMBED : A = E[tok[r][t]] * sqrt(d) + pe[t]       (transformer_decoder.py:89-91)
//   PRO_ADDLN : A = LayerNorm(X + Y2) * g + b            (the post-LN residual join of the previous sub-layer)
// For the fused producers the 32-row A tile is built once per block in LDS (K = d_model) and the blocks of
// column tile 0 also write it out (xout): it is the residual stream for the next join / the `embed` output.
// Tile 32 x 32, the four waves split K and reduce through LDS: the step is latency-, not throughput-bound.
// ---------------------------------------------------------------------------------------------
enum { PRO_PLAIN = 0, PRO_EMBED = 1, PRO_ADDLN = 2 };
constexpr int DEC_MAX_D = 512;

struct DecGemmParams {
  const float* X; long ldx;          // PLAIN: A rows; ADDLN: residual input
  const float* Y2; long ldy2;        // ADDLN: sub-layer output to add
  const float* ln_w; const float* ln_b;
  const int* tok; long tok_stride; int t;  // EMBED
  const float* emb; const float* pe; float emb_scale;
  float* xout; long ldxo;            // where column-tile-0 blocks store the produced A rows (may be null)
  const float* Wp; const float* bias;  // Wp: fragment-packed weights [ceil(N/16)][K/16][64][4]
  float* Y; long ldy;
  int M, N, K, relu;
  int ntb;  // consecutive 16-column tiles per block (1 unless K fits one chunk)
};

#ifdef AC_DEC_STAMPS  // development probe only (tools/dec_probe.hip): phase timestamps of block (0,0), wave 0
__device__ long long g_dec_stamps[16];
#define DEC_STAMP(k) do { if (blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0) g_dec_stamps[k] = clock64(); } while (0)
#else
#define DEC_STAMP(k) do { } while (0)
#endif

constexpr int DEC_WAVES = 4;   // waves per block: they split K (and the 16 producer rows)
constexpr int DEC_KC = 512;    // K chunk staged in LDS at a time
constexpr int DEC_T = 16;      // output tile: 16 rows x 16 columns (v_mfma_f32_16x16x4_f32)

// A decode step is ~0.36 GFLOP spread over 18 dependent launches: what matters is how many CUs each
// launch reaches and how few cache-line requests each block issues, not MFMA efficiency.  Hence small
// 16 x 16 tiles (64 ... 1092 blocks per projection) and weights pre-packed in MFMA fragment order
// (ac_trm_pack_step_weights):
//   Wp[n_tile][k_group][lane][4],  16 columns x 16 k per group, lane = (n % 16) + 16 * ((k % 16) / 4),
//   element = k % 4: a wave's B fragment is ONE contiguous 1 KiB read.
__global__ void pack_frag_kernel(const float* W, long ldw, int N, int K, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the packed matrix
  const int kgs = K >> 4;
  const size_t total = (size_t)((N + DEC_T - 1) / DEC_T) * kgs * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const size_t g = i >> 6;
  const int kg = (int)(g % kgs), nt = (int)(g / kgs);
  const int n = nt * DEC_T + (lane & 15), k = kg * 16 + (lane >> 4) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n < N) v = *(const f32x4*)(W + (size_t)n * ldw + k);
  *(f32x4*)(out + i * 4) = v;
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  // lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; result reg r: D[(l >> 4) * 4 + r][l & 15]
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int PRO>
__global__ __launch_bounds__(64 * DEC_WAVES) DEC_CAP void dec_gemm_kernel(DecGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float* sX = dsm;   // [16][KC + 4] A tile: the fused producer's rows, or a coalesced copy of X
  float* red = dsm + DEC_T * ((p.K < DEC_KC ? p.K : DEC_KC) + 4);  // [DEC_WAVES][16*17] split-K partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4;           // which 4 of the 16 k of a group this lane feeds
  const int m0 = blockIdx.y * DEC_T;
  const int nt0 = blockIdx.x * p.ntb;  // first of the ntb consecutive column tiles of this block
  const int KC = p.K < DEC_KC ? p.K : DEC_KC;
  const int ldsx = KC + 4;
  const int kwc = KC / DEC_WAVES;     // k range of one wave inside a chunk (multiple of 16)
  const int nsteps = kwc >> 4;        // MFMA groups of 16 k per wave per chunk (<= 8)
  const size_t wtile = (size_t)(p.K >> 4) * 64;  // float4 per packed column tile
  const f32x4* wp0 = (const f32x4*)p.Wp + (size_t)nt0 * wtile + (size_t)wave * nsteps * 64 + lane;
  DEC_STAMP(0);
  // the weight fragments of the first chunk do not depend on the producer: request them first
  f32x4 b0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) b0[u] = wp0[(size_t)(u < nsteps ? u : 0) * 64];

  if (PRO != PRO_PLAIN) {
    // Fused producer of the 16 A rows.  A wave owns 4 rows, ONE ROW PER 16-LANE GROUP: lane (g = lane >> 4,
    // s = lane & 15) holds the row's float4 columns s, s + 16, ... so that loads are 16-byte, a group reads
    // 256 contiguous bytes, and the LayerNorm reductions are 4 DPP steps inside the group (no cross-row
    // traffic, all four rows of the wave advance together).
    constexpr int NF = DEC_MAX_D / 64;  // float4 per lane: K = d_model <= 512 -> up to 8
    const int nf = p.K >> 6;
    const int grp = lane >> 4, sub = lane & 15;
    const int row = wave * 4 + grp, r = m0 + row;
    const bool ok = r < p.M;
    f32x4 v[NF], gw[NF], gb[NF];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (PRO == PRO_EMBED) {
      const int w = ok ? p.tok[(size_t)r * p.tok_stride + p.t] : 0;
      const f32x4* e4 = (const f32x4*)(p.emb + (size_t)w * p.K);
      const f32x4* p4 = (const f32x4*)(p.pe + (size_t)p.t * p.K);
#pragma unroll
      for (int i = 0; i < NF; ++i)
        v[i] = (ok && i < nf) ? e4[sub + 16 * i] * p.emb_scale + p4[sub + 16 * i] : zero4;
    } else {
      const f32x4* x4 = (const f32x4*)(p.X + (size_t)(ok ? r : 0) * p.ldx);
      const f32x4* y4 = (const f32x4*)(p.Y2 + (size_t)(ok ? r : 0) * p.ldy2);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        gw[i] = i < nf ? ((const f32x4*)p.ln_w)[sub + 16 * i] : zero4;
        gb[i] = i < nf ? ((const f32x4*)p.ln_b)[sub + 16 * i] : zero4;
        v[i] = (ok && i < nf) ? x4[sub + 16 * i] + y4[sub + 16 * i] : zero4;
      }
      DEC_STAMP(1);
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NF; ++i) sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      const float mean = row16_sum(sm) / (float)p.K;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NF; ++i)
        if (i < nf) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float dl = v[i][e] - mean; q = fmaf(dl, dl, q); }
        }
      const float rstd = rsqrtf(row16_sum(q) / (float)p.K + 1e-5f);
#pragma unroll
      for (int i = 0; i < NF; ++i) v[i] = (v[i] - mean) * rstd * gw[i] + gb[i];
    }
#pragma unroll
    for (int i = 0; i < NF; ++i)
      if (i < nf) {
        *(f32x4*)(sX + row * ldsx + (sub + 16 * i) * 4) = ok ? v[i] : zero4;
        if (blockIdx.x == 0 && p.xout && ok) *(f32x4*)(p.xout + (size_t)r * p.ldxo + (sub + 16 * i) * 4) = v[i];
      }
  }
  DEC_STAMP(2);
  const float* xa = sX + (lane & 15) * ldsx + wave * kwc + kq * 4;
  // PRO_PLAIN: the X chunk is copied into the LDS tile with coalesced 16-byte loads (row = 4*KC bytes);
  // chunk c+1 is requested into registers before the MFMAs of chunk c
  constexpr int XLD = DEC_T * (DEC_KC / 4) / (64 * DEC_WAVES);  // float4 per thread per chunk (8)
  const int c4n = KC >> 2;
  f32x4 xr[XLD];
  auto x_load = [&](int c0) {
#pragma unroll
    for (int u = 0; u < XLD; ++u) {
      const int idx = tid + u * 64 * DEC_WAVES;
      const int row = idx / (DEC_KC / 4), c4 = idx % (DEC_KC / 4);  // constants: shifts
      const bool okr = c4 < c4n && m0 + row < p.M;
      xr[u] = okr ? *(const f32x4*)(p.X + (size_t)(m0 + row) * p.ldx + c0 + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto x_store = [&]() {
#pragma unroll
    for (int u = 0; u < XLD; ++u) {
      const int idx = tid + u * 64 * DEC_WAVES;
      const int row = idx / (DEC_KC / 4), c4 = idx % (DEC_KC / 4);
      if (c4 < c4n) *(f32x4*)(sX + row * ldsx + c4 * 4) = xr[u];
    }
  };
  if (PRO == PRO_PLAIN) x_load(0);
  // ntb consecutive column tiles re-use the A tile (single-chunk K only; the launcher enforces it)
  for (int nt = 0; nt < p.ntb; ++nt) {
    const int n0 = (nt0 + nt) * DEC_T;
    if (n0 >= p.N) break;
    const f32x4* wp = wp0 + (size_t)nt * wtile;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < p.K; c0 += KC) {
      f32x4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        b[u] = (c0 == 0 && nt == 0) ? b0[u] : wp[((size_t)(c0 >> 4) + (u < nsteps ? u : 0)) * 64];
      if (PRO == PRO_PLAIN) {
        if (c0 > 0) __syncthreads();
        if (nt == 0) {
          x_store();
          if (c0 + KC < p.K) x_load(c0 + KC);
        }
      }
      if (nt == 0) __syncthreads();
      DEC_STAMP(3);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < nsteps) {
          const f32x4 a = *(const f32x4*)(xa + 16 * u);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = mfma16(a[s], b[u][s], acc);
        }
      }
    }
    DEC_STAMP(4);
    if (nt > 0) __syncthreads();  // the previous tile's partials have been consumed
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * (DEC_T * 17) + (kq * 4 + r) * 17 + (lane & 15)] = acc[r];
    __syncthreads();
    {
      const int i = tid >> 4, jn = tid & 15;  // 256 threads = the 16 x 16 outputs
      const int gm = m0 + i, gn = n0 + jn;
      if (gm < p.M && gn < p.N) {
        const int o = i * 17 + jn;
        float y = (red[o] + red[DEC_T * 17 + o]) + (red[2 * DEC_T * 17 + o] + red[3 * DEC_T * 17 + o]);
        if (p.bias) y += p.bias[gn];
        if (p.relu) y = fmaxf(y, 0.f);
        p.Y[(size_t)gm * p.ldy + gn] = y;
      }
    }
  }
  DEC_STAMP(5);
}

template <int PRO>
int launch_dec_gemm(const DecGemmParams& p_in, hipStream_t s) {
  DecGemmParams p = p_in;
  if (p.K % 64 || (PRO != PRO_PLAIN && p.K > DEC_MAX_D) || (p.K > DEC_KC && p.K % DEC_KC)) return AC_ERR_ARG;
  const int KC = p.K < DEC_KC ? p.K : DEC_KC;
  if (p.ntb < 1) return AC_ERR_ARG;
  if (p.K > DEC_KC) p.ntb = 1;   // several column tiles per block re-use ONE staged A tile: single-chunk K only
  // the A tile is [16][KC + 4]: sized by the launch's K chunk, so that the K = 256 projections fit 7 workgroups to a CU
  const size_t lds = ((size_t)DEC_T * (KC + 4) + (size_t)DEC_WAVES * DEC_T * 17) * sizeof(float);
  const int ntiles = (p.N + DEC_T - 1) / DEC_T;
  dim3 grid((ntiles + p.ntb - 1) / p.ntb, (p.M + DEC_T - 1) / DEC_T);
  hipLaunchKernelGGL((dec_gemm_kernel<PRO>), grid, dim3(64 * DEC_WAVES), lds, s, p);
  return ac_check_launch();
}

// ---------------------------------------------------------------------------------------------
// greedy pick for step t (base.py:214-218 log_softmax + max, :161-167, :202-208)
// ---------------------------------------------------------------------------------------------
struct PickParams {
  const float* logit; long ldl;  // row b at logit + b*ldl (already offset to step t)
  int V, t, max_len, end_idx, pad_idx;
  int64_t* seq; float* logprob;
  int* tok; unsigned char* mask;  // [B][max_len+1]
  int* unfinished;                // [B]
  int* cnt;                       // [max_len]
};

__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

constexpr int PICK_MAXV = 16384;  // logits of a row are held in registers: <= 64 per thread

__global__ __launch_bounds__(256) DEC_CAP void greedy_pick_kernel(PickParams p) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float ssum[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (p.t > 0 && p.cnt[p.t - 1] == 0) return;  // the reference loop has already stopped (base.py:167)
  const float* row = p.logit + (size_t)b * p.ldl;
  // one pass over memory: all loads in flight at once, max / arg-max / exp-sum from registers
  float x[PICK_MAXV / 256];
#pragma unroll
  for (int i = 0; i < PICK_MAXV / 256; ++i) {
    const int c = tid + 256 * i;
    x[i] = c < p.V ? row[c] : -INFINITY;
  }
  float v = -INFINITY;
  int idx = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < PICK_MAXV / 256; ++i) argmax_merge(v, idx, x[i], tid + 256 * i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    argmax_merge(v, idx, ov, oi);
  }
  if (lane == 0) { sv[wave] = v; si[wave] = idx; }
  __syncthreads();
  v = sv[0]; idx = si[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) argmax_merge(v, idx, sv[k], si[k]);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PICK_MAXV / 256; ++i)
    if (tid + 256 * i < p.V) s += expf(x[i] - v);
  s = wave_sum(s);
  if (lane == 0) ssum[wave] = s;
  __syncthreads();
  if (tid == 0) {
    const float tot = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
    const int prev = p.t == 0 ? 1 : p.unfinished[b];
    const int unf = prev && (idx != p.end_idx);
    const int w = unf ? idx : p.end_idx;
    p.unfinished[b] = unf;
    p.seq[(size_t)b * p.max_len + p.t] = w;
    p.logprob[(size_t)b * p.max_len + p.t] = -logf(tot);
    p.tok[(size_t)b * (p.max_len + 1) + p.t + 1] = w;
    p.mask[(size_t)b * (p.max_len + 1) + p.t + 1] = (w == p.pad_idx) ? 1 : 0;
    if (unf) atomicAdd(&p.cnt[p.t], 1);
  }
}

__global__ void greedy_init_kernel(int64_t* seq, float* logprob, int* tok, unsigned char* mask, int* unfinished,
                                   int* cnt, int B, int max_len, int start_idx, int end_idx, int pad_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * max_len) { seq[i] = end_idx; logprob[i] = 0.f; }
  if (i < B * (max_len + 1)) {
    const int c = i % (max_len + 1);
    tok[i] = c == 0 ? start_idx : end_idx;
    mask[i] = c == 0 ? (start_idx == pad_idx) : 0;
  }
  if (i < B) unfinished[i] = 1;
  if (i < max_len) cnt[i] = 0;
}

// ---------------------------------------------------------------------------------------------
// beam search, device half (base.py:282-289)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max256(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// lp[r][:] = log_softmax(log_softmax(logit[r]) / temp) + cum[r]
__global__ __launch_bounds__(256) void beam_logprob_kernel(const float* logit, const float* cum, float temp,
                                                           float* lp, int V) {
  __shared__ float sh[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* row = logit + (size_t)r * V;
  float m = -INFINITY;
  for (int c = tid; c < V; c += 256) m = fmaxf(m, row[c]);
  m = block_max256(m, sh);
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += expf(row[c] - m);
  s = block_sum256(s, sh);
  const float lse1 = m + logf(s);
  const float inv_t = 1.0f / temp;
  float m2 = -INFINITY;
  for (int c = tid; c < V; c += 256) m2 = fmaxf(m2, (row[c] - lse1) * inv_t);
  m2 = block_max256(m2, sh);
  float s2 = 0.f;
  for (int c = tid; c < V; c += 256) s2 += expf((row[c] - lse1) * inv_t - m2);
  s2 = block_sum256(s2, sh);
  const float lse2 = m2 + logf(s2);
  const float cr = cum[r];
  for (int c = tid; c < V; c += 256) lp[(size_t)r * V + c] = cr + ((row[c] - lse1) * inv_t - lse2);
}

// per clip: the `beam` largest entries of lp[clip*beam .. +nrows][V] flattened (lowest index wins ties).
// One pass over the scores: every thread keeps its own best TOPK_LOCAL candidates in registers; the workgroup then
// picks the winners from the 256 x TOPK_LOCAL survivors in LDS (the global top `beam` is contained in them because each
// thread's list holds its own best `beam`).  beam > TOPK_LOCAL falls back to one full scan per winner.
constexpr int TOPK_LOCAL = 4;
__device__ __forceinline__ bool cand_better(float v, int i, float ov, int oi) { return v > ov || (v == ov && i < oi); }

__global__ __launch_bounds__(256) void beam_topk_kernel(const float* lp, int beam, int nrows, int V,
                                                        float* top_val, int* top_idx) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int chosen[64];
  __shared__ float cv[256 * TOPK_LOCAL];
  __shared__ int ci[256 * TOPK_LOCAL];
  const int clip = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* base = lp + (size_t)clip * beam * V;
  const int n = nrows * V;
  const bool local = beam <= TOPK_LOCAL;
  if (local) {
    float bv[TOPK_LOCAL];
    int bi[TOPK_LOCAL];
#pragma unroll
    for (int j = 0; j < TOPK_LOCAL; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
    for (int c = tid; c < n; c += 256) {
      float v = base[c];
      int idx = c;
      if (cand_better(v, idx, bv[TOPK_LOCAL - 1], bi[TOPK_LOCAL - 1])) {
#pragma unroll
        for (int j = 0; j < TOPK_LOCAL; ++j)   // insertion into the sorted list (best first)
          if (cand_better(v, idx, bv[j], bi[j])) {
            const float tv = bv[j]; const int ti = bi[j];
            bv[j] = v; bi[j] = idx; v = tv; idx = ti;
          }
      }
    }
#pragma unroll
    for (int j = 0; j < TOPK_LOCAL; ++j) { cv[tid * TOPK_LOCAL + j] = bv[j]; ci[tid * TOPK_LOCAL + j] = bi[j]; }
    __syncthreads();
  }
  const int ncand = local ? 256 * TOPK_LOCAL : n;
  for (int k = 0; k < beam; ++k) {
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int c = tid; c < ncand; c += 256) {
      const float x = local ? cv[c] : base[c];
      const int xi = local ? ci[c] : c;
      bool skip = false;
      for (int j = 0; j < k; ++j) skip |= (chosen[j] == xi);
      if (!skip) argmax_merge(v, idx, x, xi);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      argmax_merge(v, idx, ov, oi);
    }
    __syncthreads();
    if (lane == 0) { sv[wave] = v; si[wave] = idx; }
    __syncthreads();
    if (tid == 0) {
      v = sv[0]; idx = si[0];
      for (int w = 1; w < 4; ++w) argmax_merge(v, idx, sv[w], si[w]);
      chosen[k] = idx;
      top_val[clip * beam + k] = v;
      top_idx[clip * beam + k] = idx;
    }
    __syncthreads();
  }
}

// The two kernels above as ONE pass per row (beam <= 8, V <= 256 NV): the logits of a row live in registers (one global
// read instead of five sweeps), lp = log_softmax(log_softmax(x) / temp) + cum is never written - the row only publishes
// its own best `beam` candidates (the clip's best `beam` are among the per-row best `beam`), picked by `beam` rounds of a
// workgroup arg-max over the registers.  Same arithmetic in the same order as beam_logprob_kernel (the second maximum
// is the image of the first: the map is increasing), same tie rule as beam_topk_kernel: identical results.
template <int NV>
__global__ __launch_bounds__(256) void beam_row_topk_kernel(const float* logit, const float* cum, float temp, int beam, int V,
                                                            float* cand_val, int* cand_idx) {
  __shared__ float sh[4];
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int s_win;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logit + (size_t)r * V;
  float x[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) x[k] = tid + 256 * k < V ? row[tid + 256 * k] : -INFINITY;
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < NV; ++k) m = fmaxf(m, x[k]);
  m = block_max256(m, sh);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (tid + 256 * k < V) s += expf(x[k] - m);
  s = block_sum256(s, sh);
  const float lse1 = m + logf(s);
  const float inv_t = 1.0f / temp;
  const float m2 = (m - lse1) * inv_t;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k)
    if (tid + 256 * k < V) s2 += expf((x[k] - lse1) * inv_t - m2);
  s2 = block_sum256(s2, sh);
  const float lse2 = m2 + logf(s2);
  const float cr = cum[r];
#pragma unroll
  for (int k = 0; k < NV; ++k) x[k] = tid + 256 * k < V ? cr + ((x[k] - lse1) * inv_t - lse2) : -INFINITY;
  unsigned taken = 0u;
  const int flat0 = (r % beam) * V;           // index of the row's first entry in the clip's flattened scores
  for (int j = 0; j < beam; ++j) {
    float v = -INFINITY;
    int idx = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (!((taken >> k) & 1u) && tid + 256 * k < V) argmax_merge(v, idx, x[k], tid + 256 * k);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      argmax_merge(v, idx, ov, oi);
    }
    __syncthreads();
    if (lane == 0) { sv[wave] = v; si[wave] = idx; }
    __syncthreads();
    if (tid == 0) {
      v = sv[0]; idx = si[0];
      for (int w = 1; w < 4; ++w) argmax_merge(v, idx, sv[w], si[w]);
      s_win = idx;
      cand_val[(size_t)r * beam + j] = v;
      cand_idx[(size_t)r * beam + j] = flat0 + idx;
    }
    __syncthreads();
    const int win = s_win;
    if ((win & 255) == tid) taken |= 1u << (win >> 8);
  }
}

// per clip: the `beam` best of the nrows x beam row candidates (<= 64: one per lane), lowest flattened index wins ties
__global__ __launch_bounds__(64) void beam_merge_kernel(const float* cand_val, const int* cand_idx, int beam, int nrows,
                                                       float* top_val, int* top_idx) {
  const int clip = blockIdx.x, lane = threadIdx.x;
  const int n = nrows * beam;
  float v = lane < n ? cand_val[(size_t)clip * beam * beam + lane] : -INFINITY;
  int idx = lane < n ? cand_idx[(size_t)clip * beam * beam + lane] : 0x7fffffff;
  for (int j = 0; j < beam; ++j) {
    float bv = v;
    int bi = idx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      argmax_merge(bv, bi, ov, oi);
    }
    if (lane == 0) {
      top_val[clip * beam + j] = bv;
      top_idx[clip * beam + j] = bi;
    }
    if (idx == bi) { v = -INFINITY; idx = 0x7fffffff; }
  }
}

// Per-clip beam bookkeeping of base.py:290-323 on the device (one workgroup per clip, thread 0 does the serial part):
// re-gather the token rows by the chosen previous beams, append the new words, record the beams that END this step
// (in beam order, score = logprob / (t + 1)), apply the -1000 trick to their cumulative scores and retire the clip
// when its number of finished beams EQUALS the beam size (the reference's '==').  A retired clip keeps its rows.
__global__ __launch_bounds__(64) void beam_update_kernel(const float* top_val, const int* top_idx, const int* tok_in,
                                                        int* tok_out, unsigned char* mask_out, float* cum, int* active,
                                                        int* done_cnt, int* done_seq, float* done_score, int* src_row,
                                                        int* n_active, int beam, int V, int max_len, int t, int end_idx,
                                                        int pad_idx, int cap) {
  __shared__ int s_src[64];
  __shared__ int s_word[64];
  const int clip = blockIdx.x, tid = threadIdx.x;
  const int ld = max_len + 1;
  const bool act = active[clip] != 0;
  if (tid < beam) {
    const int flat = top_idx[clip * beam + tid];
    s_src[tid] = act ? clip * beam + flat / V : clip * beam + tid;
    s_word[tid] = flat % V;
    src_row[clip * beam + tid] = s_src[tid];
  }
  __syncthreads();
  // token rows (and the key-padding mask the decoder step reads)
  for (int e = tid; e < beam * ld; e += 64) {
    const int k = e / ld, c = e % ld;
    int v = tok_in[(size_t)s_src[k] * ld + c];
    if (act && c == t + 1) v = s_word[k];
    tok_out[(size_t)(clip * beam + k) * ld + c] = v;
    mask_out[(size_t)(clip * beam + k) * ld + c] = v == pad_idx ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0 && act) {
    int cnt = done_cnt[clip];
    for (int k = 0; k < beam; ++k) {
      const float v = top_val[clip * beam + k];
      const bool is_end = s_word[k] == end_idx || t == max_len - 1;
      if (is_end) {
        if (cnt < cap) {
          int* dst = done_seq + ((size_t)clip * cap + cnt) * max_len;
          const int* row = tok_out + (size_t)(clip * beam + k) * ld;
          for (int c = 0; c < max_len; ++c) dst[c] = c <= t ? row[c + 1] : end_idx;
          done_score[(size_t)clip * cap + cnt] = v / (float)(t + 1);
        }
        ++cnt;
      }
      cum[clip * beam + k] = v - (is_end ? 1000.0f : 0.0f);
    }
    done_cnt[clip] = cnt;
    if (cnt == beam) {
      active[clip] = 0;
      atomicSub(n_active, 1);
    }
  }
}

// dst[l][r][0..t][:] = src[l][src_row[r]][0..t][:] for both K and V caches
__global__ void cache_gather_kernel(const float* src, float* dst, const int* src_row, int R, int max_len, int t,
                                    int d, size_t set_stride /* floats between K and V sets and layers */, int nsets) {
  const int r = blockIdx.x, set = blockIdx.y;
  const int sr = src_row[r];
  const size_t n = (size_t)(t + 1) * d;
  const float* s = src + set * set_stride + (size_t)sr * max_len * d;
  float* o = dst + set * set_stride + (size_t)r * max_len * d;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) o[i] = s[i];
}

// ---------------------------------------------------------------------------------------------
// workspace carving
// ---------------------------------------------------------------------------------------------
// The wide route (steps over many rows, csrc/decoder_wide.hip) covers the reference's decoder width: d_model 256 (its fused
// producers stage a 256-wide row; heads of <= 64) and a feed-forward width of 256, 512 or 1024.
inline bool wide_shape_ok(const ac_trm_weights* w) {
  return w->d_model == 256 && (w->dim_ff == 256 || w->dim_ff == 512 || w->dim_ff == 1024);
}

struct Ws {
  float *x, *x2, *qkv, *att, *tmp, *ff, *q2, *lg;
  float *attp, *ffp;        // wide route: fragment packs of the attention context rows / the feed-forward hidden rows
  float* cache[2];  // [2 (K,V)][nlayers][R][max_len][d] each
  int *tok, *unfinished;
  unsigned char* mask;
  size_t cache_set_stride;  // R*max_len*d
  size_t total;
};

inline size_t align4(size_t n) { return (n + 3) & ~(size_t)3; }

Ws carve(const ac_trm_weights* w, int R, int max_len, float* base) {
  Ws s;
  const size_t d = w->d_model;
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += align4(n); return p; };
  s.x = take(R * d);
  s.x2 = take(R * d);
  s.qkv = take(R * 3 * d);
  s.att = take(R * d);
  s.tmp = take(R * d);
  s.ff = take((size_t)R * w->dim_ff);
  s.q2 = take(R * d);
  s.lg = take((size_t)R * w->vocab);
  const size_t r32 = (size_t)(R + 31) / 32 * 32;
  s.attp = take(wide_shape_ok(w) ? r32 * d * 3 / 2 : 0);
  s.ffp = take(wide_shape_ok(w) ? r32 * (size_t)w->dim_ff * 3 / 2 : 0);
  s.cache_set_stride = (size_t)R * max_len * d;
  for (int i = 0; i < 2; ++i) s.cache[i] = take(2 * (size_t)w->nlayers * s.cache_set_stride);
  s.tok = (int*)take((size_t)R * (max_len + 1));
  s.unfinished = (int*)take(R);
  s.mask = (unsigned char*)take(((size_t)R * (max_len + 1) + 3) / 4);
  s.total = off;
  return s;
}

#define AC_TRY(expr)            \
  do {                          \
    int _e = (expr);            \
    if (_e != AC_OK) return _e; \
  } while (0)

int check_weights(const ac_trm_weights* w) {
  if (!w || w->nlayers < 1 || w->nlayers > AC_MAX_LAYERS) return AC_ERR_ARG;
  if (w->d_model % 32 || w->d_model > 1024 || w->dim_ff % 32 || w->attn_emb_dim % 32) return AC_ERR_ARG;
  if (w->nhead < 1 || w->d_model % w->nhead || w->d_model / w->nhead > 64) return AC_ERR_ARG;
  return AC_OK;
}

int launch_ln(const float* x, const float* y, const float* w, const float* b, float* out, int rows, int d,
              long ldx, long ldy, long ldo, hipStream_t s) {
  hipLaunchKernelGGL(add_layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, y, w, b, out, rows, d, ldx,
                     ldy, ldo);
  return ac_check_launch();
}

// One decoder position for R rows.  tokens/mask: [R][tok_stride]; the input token is column t.
// cache: active KV cache set.  Returns through `fin` the operands of the LAST residual join
// (embed = LayerNorm(fin.x + fin.y) with the last layer's norm3), which the caller fuses into the
// classifier projection; 8 launches per layer: the residual joins and the embedding never run alone.
// Offsets (in floats) of the fragment-packed step weights inside ac_trm_weights::step_pk, fixed order:
// per layer sa_in, sa_out, ca_q, ca_out, l1, l2; then the classifier.
struct PackLayout {
  size_t sa_in, sa_out, ca_q, ca_out, l1, l2;
  size_t sa_outT, ca_qT, ca_outT;   // transposed [k][n] copies for the fused per-row sub-layer kernel
  size_t w_sa_in, w_sa_out, w_ca_q, w_ca_out, w_l1, w_l2;   // three-plane bf16 packs of the wide route (csrc/decoder_wide.hip)
};
inline size_t packed_floats(int N, int K) { return (size_t)((N + DEC_T - 1) / DEC_T) * DEC_T * K; }
inline size_t pack_layout(const ac_trm_weights* w, PackLayout* L /* [nlayers] or null */, size_t* cls_off,
                          size_t* wcls_off = nullptr) {
  const int d = w->d_model, ff = w->dim_ff;
  const bool wide = wide_shape_ok(w);
  auto wf = [&](int N, int K) { return wide ? (size_t)ac_dec_wide_packed_floats(N, K) : (size_t)0; };
  size_t off = 0;
  for (int l = 0; l < w->nlayers; ++l) {
    PackLayout t;
    t.sa_in = off; off += packed_floats(3 * d, d);
    t.sa_out = off; off += packed_floats(d, d);
    t.ca_q = off; off += packed_floats(d, d);
    t.ca_out = off; off += packed_floats(d, d);
    t.l1 = off; off += packed_floats(ff, d);
    t.l2 = off; off += packed_floats(d, ff);
    t.sa_outT = off; off += (size_t)d * d;
    t.ca_qT = off; off += (size_t)d * d;
    t.ca_outT = off; off += (size_t)d * d;
    t.w_sa_in = off; off += wf(3 * d, d);
    t.w_sa_out = off; off += wf(d, d);
    t.w_ca_q = off; off += wf(d, d);
    t.w_ca_out = off; off += wf(d, d);
    t.w_l1 = off; off += wf(ff, d);
    t.w_l2 = off; off += wf(d, ff);
    if (L) L[l] = t;
  }
  if (cls_off) *cls_off = off;
  off += packed_floats(w->vocab, d);
  if (wcls_off) *wcls_off = off;
  off += wf(w->vocab, d);
  return off;
}

struct StepOut {
  const float* x;
  const float* y;
  const float* ln_w;
  const float* ln_b;
};

// AUDIOCAPTION_DEC_WIDE_MIN=n (default 0 = never): a step over n rows or more takes the wide route.  Opt-in, because it does
// not pay on this part: stand-alone it is 145 / 202 us per step at 256 / 768 rows against 115 / 193 for the narrow route
// (22 launches per step instead of 10, each ~3 us of launch overhead + a 3-5 us critical path, although a projection
// occupies 32-128 workgroups for 3-5 us instead of every CU for 7-18), and beside the next batches' encoders the step
// costs what its DURATION is, not its CU time: headline 13.85 k vs 14.10 k clips/s, EffB2-Trm 20.65 k vs 20.55 k
// (EXPERIMENTS.md, round 6).
inline bool wide_route(const ac_trm_weights* w, int R) {
  const char* e = getenv("AUDIOCAPTION_DEC_WIDE_MIN");   // read per call (tests switch it); a captured graph keeps its route
  const int wide_min = e ? atoi(e) : 0;
  return wide_min > 0 && R >= wide_min && wide_shape_ok(w) && w->d_model / w->nhead <= 64 && w->d_model % w->nhead == 0;
}

// AUDIOCAPTION_DEC_HYBRID=1 (default off): from 512 rows on (a beam search over grouped batches) the QKV projection with the
// residual join in its prologue and the classifier take their wide twins (csrc/decoder_wide.hip).  Stand-alone those are the
// two launches of the narrow route furthest from their arithmetic (20 and 30 + 5 us at 768 rows against 11 and 27); inside the
// EffB2-Trm pipeline the wide classifier's one-workgroup-per-CU blocks take 43 us beside the encoder and the step does not move
// (20.41 vs 20.41 k clips/s, 30 s / beam 4: 7.70 vs 7.70 k): opt-in, kept for the record.
inline bool hybrid_route(const ac_trm_weights* w, int R) {
  const char* e = getenv("AUDIOCAPTION_DEC_HYBRID");
  return e && !strcmp(e, "1") && R >= 512 && wide_shape_ok(w);
}

int decoder_step(const ac_trm_weights* w, const float* memkv, const int* mem_len, int R, int row_div, int Tm,
                 int max_len, int t, const int* tok, const unsigned char* mask, long tok_stride, float* cache,
                 const Ws& ws, StepOut* fin, hipStream_t s) {
  const int d = w->d_model, hd = d / w->nhead;
  const float scale = 1.0f / sqrtf((float)hd);
  if (t >= w->max_pos || d > DEC_MAX_D || d % 64 || w->dim_ff % 64 || !w->step_pk) return AC_ERR_ARG;
  PackLayout PL[AC_MAX_LAYERS];
  pack_layout(w, PL, nullptr);
  const float* pk = w->step_pk;
  const size_t Rm = (size_t)(R / row_div) * Tm;  // memory rows
  float* xa = ws.x;    // residual stream, ping-pong: a join reads one and writes the other
  float* xb = ws.x2;
  DecGemmParams g;
  g.tok = tok; g.tok_stride = tok_stride; g.t = t; g.emb = w->emb; g.pe = w->pe; g.emb_scale = sqrtf((float)d);
  // column tiles per block of the single-chunk projections: two from 512 rows on (beam search over grouped batches: 768
  // rows - the A tile is staged once for both, EffB2-Trm 19.15 -> 19.40 k clips/s; at 256 rows and below no difference).  The
  // arithmetic of an output does not depend on it.  AUDIOCAPTION_DEC_NTB overrides (development).
  static const int dev_ntb = getenv("AUDIOCAPTION_DEC_NTB") ? atoi(getenv("AUDIOCAPTION_DEC_NTB")) : 0;
  g.M = R; g.ntb = dev_ntb > 0 ? dev_ntb : (R >= 512 ? 2 : 1);
  // pending join carried into the next projection: x_next = LayerNorm(jx + jy) * jw + jb
  const float *jx = nullptr, *jy = nullptr, *jw = nullptr, *jb = nullptr;
  // the reference's decoder shape (d_model 256 = 4 heads of 64) takes the fused per-row sub-layer kernel: 5 launches per
  // layer instead of 8; any other shape the general 8-launch sequence
  // AUDIOCAPTION_DEC_ROW=gemm: the general 18-launch sequence (attn_step_kernel + dec_gemm_kernel) instead of the per-row
  // kernels.  (The row kernels once gave co-runner-dependent results: built with packed-f32 VALU instructions their
  // matrix-vector products broke beside MFMA-heavy kernels of another stream - see audiocaption_amd/build.py; every mode is
  // now held to tests/test_gpu_model.py::test_decode_is_bit_stable_beside_matrix_heavy_kernels.)
  const char* dec_row = getenv("AUDIOCAPTION_DEC_ROW");   // read per call (tests compare the sequences; a captured graph keeps its own)
  const bool no_rows = dec_row && !strcmp(dec_row, "gemm");
  const bool fused = d == ROW_D && w->nhead == ROW_H && t + 1 <= MAX_KEYS && Tm <= MAX_KEYS && !no_rows;
  if (wide_route(w, R)) {
    // ---- the wide route: 8 launches per layer, every projection a few dozen 32-row workgroups on the bf16 matrix cores
    // (csrc/decoder_wide.hip), the attention sub-layers one wave per (row, head).  Rows that feed a projection without a
    // residual join in front (attention contexts, feed-forward hidden rows) travel as bf16-plane fragment packs ----
    const float emb_scale = sqrtf((float)d);
    auto fused = [&](int pro, const float* X, const float* Y2, const float* lw, const float* lb, float* xout, size_t wp,
                     const float* bias, float* Y, long ldy, int N, int relu, int split_out) {
      return ac_dec_wide_gemm(pro, X, d, Y2, d, lw, lb, tok, tok_stride, t, w->emb, w->pe, emb_scale, xout, d, pk + wp, bias, Y,
                              ldy, R, N, d, relu, 1, split_out, (void*)s);
    };
    auto packed = [&](const float* Xs, size_t wp, const float* bias, float* Y, int N, int K) {
      return ac_dec_wide_gemm(0, Xs, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0.f, nullptr, 0, pk + wp,
                              bias, Y, N, R, N, K, 0, 1, 0, (void*)s);
    };
    for (int l = 0; l < w->nlayers; ++l) {
      const ac_trm_layer& L = w->layer[l];
      AC_TRY(fused(l == 0 ? 1 : 2, jx, jy, jw, jb, xa, PL[l].w_sa_in, L.sa_in_b, ws.qkv, 3 * d, 3 * d, 0, 0));
      AttnParams a;
      float* Kc = cache + (size_t)(2 * l) * ws.cache_set_stride;
      float* Vc = cache + (size_t)(2 * l + 1) * ws.cache_set_stride;
      a.q = ws.qkv; a.ldq = 3 * d;
      a.K = Kc; a.V = Vc; a.Kw = Kc; a.Vw = Vc;
      a.row_stride = (long)max_len * d; a.key_stride = d; a.row_div = 1; a.nkeys = t + 1;
      a.key_len = nullptr; a.key_mask = mask; a.mask_stride = tok_stride;
      a.new_k = ws.qkv + d; a.new_v = ws.qkv + 2 * d; a.ld_new = 3 * d;
      a.out = nullptr; a.ldo = 0; a.hd = hd; a.scale = scale;
      a.out_pk = (unsigned char*)ws.attp; a.pk_kst = d / 16;
      hipLaunchKernelGGL(attn_step_kernel, dim3(R, w->nhead), dim3(64), 0, s, a);
      AC_TRY(ac_check_launch());
      AC_TRY(packed(ws.attp, PL[l].w_sa_out, L.sa_out_b, ws.tmp, d, d));
      AC_TRY(fused(2, xa, ws.tmp, L.n1_w, L.n1_b, xb, PL[l].w_ca_q, L.ca_in_b, ws.q2, d, d, 0, 0));
      const float* mk = memkv + (size_t)l * Rm * 2 * d;
      a.q = ws.q2; a.ldq = d;
      a.K = mk; a.V = mk + d; a.Kw = nullptr; a.Vw = nullptr;
      a.row_stride = (long)Tm * 2 * d; a.key_stride = 2 * d; a.row_div = row_div; a.nkeys = Tm;
      a.key_len = mem_len; a.key_mask = nullptr; a.mask_stride = 0;
      a.new_k = nullptr; a.new_v = nullptr; a.ld_new = 0;
      hipLaunchKernelGGL(attn_step_kernel, dim3(R, w->nhead), dim3(64), 0, s, a);
      AC_TRY(ac_check_launch());
      AC_TRY(packed(ws.attp, PL[l].w_ca_out, L.ca_out_b, ws.tmp, d, d));
      AC_TRY(fused(2, xb, ws.tmp, L.n2_w, L.n2_b, xa, PL[l].w_l1, L.l1_b, ws.ffp, 0, w->dim_ff, 1, 1));
      AC_TRY(packed(ws.ffp, PL[l].w_l2, L.l2_b, ws.tmp, d, w->dim_ff));
      jx = xa; jy = ws.tmp; jw = L.n3_w; jb = L.n3_b;
      float* tsw = xa; xa = xb; xb = tsw;
    }
    fin->x = jx; fin->y = jy; fin->ln_w = jw; fin->ln_b = jb;
    return AC_OK;
  }
  for (int l = 0; l < w->nlayers; ++l) {
    const ac_trm_layer& L = w->layer[l];
    // ---- self attention: QKV projection with the layer input produced in its prologue ----
    g.Wp = pk + PL[l].sa_in; g.bias = L.sa_in_b; g.Y = ws.qkv; g.ldy = 3 * d; g.N = 3 * d; g.K = d; g.relu = 0;
    g.xout = xa; g.ldxo = d;
    if (l == 0) {
      AC_TRY(launch_dec_gemm<PRO_EMBED>(g, s));
    } else if (hybrid_route(w, R)) {
      AC_TRY(ac_dec_wide_gemm(2, jx, d, jy, d, jw, jb, nullptr, 0, 0, nullptr, nullptr, 0.f, xa, d, pk + PL[l].w_sa_in, L.sa_in_b,
                              ws.qkv, 3 * d, R, 3 * d, d, 0, 1, 0, (void*)s));
    } else {
      g.X = jx; g.ldx = d; g.Y2 = jy; g.ldy2 = d; g.ln_w = jw; g.ln_b = jb;
      AC_TRY(launch_dec_gemm<PRO_ADDLN>(g, s));
    }
    AttnParams a;
    float* Kc = cache + (size_t)(2 * l) * ws.cache_set_stride;
    float* Vc = cache + (size_t)(2 * l + 1) * ws.cache_set_stride;
    a.q = ws.qkv; a.ldq = 3 * d;
    a.K = Kc; a.V = Vc; a.Kw = Kc; a.Vw = Vc;
    a.row_stride = (long)max_len * d; a.key_stride = d; a.row_div = 1; a.nkeys = t + 1;
    a.key_len = nullptr; a.key_mask = mask; a.mask_stride = tok_stride;
    a.new_k = ws.qkv + d; a.new_v = ws.qkv + 2 * d; a.ld_new = 3 * d;
    a.out = ws.att; a.ldo = d; a.hd = hd; a.scale = scale;
    if (fused) {
      // ---- self attention + out-projection + LN1 + cross query projection: one launch, one row per workgroup ----
      RowParams rp;
      rp.a = a;
      rp.res = xa; rp.ldres = d;
      rp.WoT = pk + PL[l].sa_outT; rp.bo = L.sa_out_b; rp.ln_w = L.n1_w; rp.ln_b = L.n1_b;
      rp.xout = xb; rp.ldxo = d;
      rp.WqT = pk + PL[l].ca_qT; rp.bq = L.ca_in_b; rp.qout = ws.q2; rp.ldqo = d;
      static const bool two_launches = getenv("AUDIOCAPTION_DEC_ROW") && !strcmp(getenv("AUDIOCAPTION_DEC_ROW"), "split");
      const RowParams rp_self = rp;
      if (two_launches) {
        hipLaunchKernelGGL(dec_row_kernel, dim3(R), dim3(256), 0, s, rp);
        AC_TRY(ac_check_launch());
      }
      // ---- cross attention + out-projection + LN2 ----
      const float* mkf = memkv + (size_t)l * Rm * 2 * d;
      rp.a.q = ws.q2; rp.a.ldq = d;
      rp.a.K = mkf; rp.a.V = mkf + d; rp.a.Kw = nullptr; rp.a.Vw = nullptr;
      rp.a.row_stride = (long)Tm * 2 * d; rp.a.key_stride = 2 * d; rp.a.row_div = row_div; rp.a.nkeys = Tm;
      rp.a.key_len = mem_len; rp.a.key_mask = nullptr; rp.a.mask_stride = 0;
      rp.a.new_k = nullptr; rp.a.new_v = nullptr; rp.a.ld_new = 0;
      rp.res = xb; rp.ldres = d;
      rp.WoT = pk + PL[l].ca_outT; rp.bo = L.ca_out_b; rp.ln_w = L.n2_w; rp.ln_b = L.n2_b;
      rp.xout = xa; rp.ldxo = d;
      rp.WqT = nullptr; rp.bq = nullptr; rp.qout = nullptr; rp.ldqo = 0;
      if (two_launches) hipLaunchKernelGGL(dec_row_kernel, dim3(R), dim3(256), 0, s, rp);
      else hipLaunchKernelGGL(dec_row2_kernel, dim3(R), dim3(256), 0, s, rp_self, rp);   // both sub-layers, one launch
      AC_TRY(ac_check_launch());
      // ---- feed forward on the materialised x2 ----
      g.X = xa; g.ldx = d; g.Wp = pk + PL[l].l1; g.bias = L.l1_b; g.Y = ws.ff; g.ldy = w->dim_ff; g.N = w->dim_ff;
      g.K = d; g.relu = 1; g.xout = nullptr;
      AC_TRY(launch_dec_gemm<PRO_PLAIN>(g, s));
      g.X = ws.ff; g.ldx = w->dim_ff; g.Wp = pk + PL[l].l2; g.bias = L.l2_b; g.Y = ws.tmp; g.ldy = d;
      g.N = d; g.K = w->dim_ff; g.relu = 0; g.xout = nullptr;
      AC_TRY(launch_dec_gemm<PRO_PLAIN>(g, s));   // K = dim_ff may exceed one chunk: the launcher then takes one tile per block
      jx = xa; jy = ws.tmp; jw = L.n3_w; jb = L.n3_b;
      float* tsw = xa; xa = xb; xb = tsw;
      continue;
    }
    hipLaunchKernelGGL(attn_step_kernel, dim3(R, w->nhead), dim3(64), 0, s, a);
    AC_TRY(ac_check_launch());
    g.X = ws.att; g.ldx = d; g.Wp = pk + PL[l].sa_out; g.bias = L.sa_out_b; g.Y = ws.tmp; g.ldy = d;
    g.N = d; g.K = d; g.relu = 0; g.xout = nullptr;
    AC_TRY(launch_dec_gemm<PRO_PLAIN>(g, s));
    // ---- cross attention: query projection of x1 = LN1(x + self_attn) ----
    g.X = xa; g.ldx = d; g.Y2 = ws.tmp; g.ldy2 = d; g.ln_w = L.n1_w; g.ln_b = L.n1_b;
    g.Wp = pk + PL[l].ca_q; g.bias = L.ca_in_b; g.Y = ws.q2; g.ldy = d; g.N = d; g.K = d; g.relu = 0;
    g.xout = xb; g.ldxo = d;
    AC_TRY(launch_dec_gemm<PRO_ADDLN>(g, s));
    const float* mk = memkv + (size_t)l * Rm * 2 * d;
    a.q = ws.q2; a.ldq = d;
    a.K = mk; a.V = mk + d; a.Kw = nullptr; a.Vw = nullptr;
    a.row_stride = (long)Tm * 2 * d; a.key_stride = 2 * d; a.row_div = row_div; a.nkeys = Tm;
    a.key_len = mem_len; a.key_mask = nullptr; a.mask_stride = 0;
    a.new_k = nullptr; a.new_v = nullptr; a.ld_new = 0;
    hipLaunchKernelGGL(attn_step_kernel, dim3(R, w->nhead), dim3(64), 0, s, a);
    AC_TRY(ac_check_launch());
    g.X = ws.att; g.ldx = d; g.Wp = pk + PL[l].ca_out; g.bias = L.ca_out_b; g.Y = ws.tmp; g.ldy = d;
    g.N = d; g.K = d; g.relu = 0; g.xout = nullptr;
    AC_TRY(launch_dec_gemm<PRO_PLAIN>(g, s));
    // ---- feed forward on x2 = LN2(x1 + cross_attn) ----
    g.X = xb; g.ldx = d; g.Y2 = ws.tmp; g.ldy2 = d; g.ln_w = L.n2_w; g.ln_b = L.n2_b;
    g.Wp = pk + PL[l].l1; g.bias = L.l1_b; g.Y = ws.ff; g.ldy = w->dim_ff; g.N = w->dim_ff; g.K = d; g.relu = 1;
    g.xout = xa; g.ldxo = d;
    AC_TRY(launch_dec_gemm<PRO_ADDLN>(g, s));
    g.X = ws.ff; g.ldx = w->dim_ff; g.Wp = pk + PL[l].l2; g.bias = L.l2_b; g.Y = ws.tmp; g.ldy = d;
    g.N = d; g.K = w->dim_ff; g.relu = 0; g.xout = nullptr;
    AC_TRY(launch_dec_gemm<PRO_PLAIN>(g, s));
    // x3 = LN3(x2 + ff) is produced by the next consumer (next layer's QKV or the caller's projection)
    jx = xa; jy = ws.tmp; jw = L.n3_w; jb = L.n3_b;
    float* tswap = xa; xa = xb; xb = tswap;  // next layer writes its input into the other buffer
  }
  fin->x = jx; fin->y = jy; fin->ln_w = jw; fin->ln_b = jb;
  return AC_OK;
}

// logits[R, V] = LayerNorm(fin.x + fin.y) W_cls^T, the normalised rows also stored to xout (= `embed`)
int classifier_step(const ac_trm_weights* w, const StepOut& fin, int R, float* xout, long ldxo, float* logit,
                    long ldl, hipStream_t s, float* scratch = nullptr) {
  // From 512 rows on (beam search over grouped batches: 768 rows x 4368 words) the 16 x 16-tile projection re-reads the
  // classifier once per 16 rows: 47 us per step; the residual join as its own launch + the tiled exact-f32 GEMM (ac_gemm)
  // take 4 + 30.  Another summation order than the projection's (last bits of the logits; not a precision change).
  // AUDIOCAPTION_DEC_CLS_GEMM=0: the projection at every row count.
  if ((wide_route(w, R) || hybrid_route(w, R)) && (!xout || (ldxo % 4 == 0 && !((uintptr_t)xout & 15)))) {
    size_t wcls;
    pack_layout(w, nullptr, nullptr, &wcls);
    const char* e = getenv("AUDIOCAPTION_DEC_WIDE_CLS_NTB");
    const int cls_ntb = e ? atoi(e) : (R >= 512 ? 4 : 2);   // 64-column groups per workgroup: the LayerNorm of a 32-row tile is redone once per group block
    return ac_dec_wide_gemm(2, fin.x, w->d_model, fin.y, w->d_model, fin.ln_w, fin.ln_b, nullptr, 0, 0, nullptr, nullptr, 0.f,
                            xout, ldxo, w->step_pk + wcls, nullptr, logit, ldl, R, w->vocab, w->d_model, 0,
                            cls_ntb > 0 ? cls_ntb : 1, 0, (void*)s);
  }
  static const bool cls_gemm = !(getenv("AUDIOCAPTION_DEC_CLS_GEMM") && !strcmp(getenv("AUDIOCAPTION_DEC_CLS_GEMM"), "0"));
  if (cls_gemm && R >= 512 && (xout || scratch) && w->cls_w) {
    float* nx = xout ? xout : scratch;
    const long ldn = xout ? ldxo : (long)w->d_model;
    AC_TRY(launch_ln(fin.x, fin.y, fin.ln_w, fin.ln_b, nx, R, w->d_model, w->d_model, w->d_model, ldn, s));
    return ac_gemm(nx, ldn, 1, w->cls_w, 1, w->d_model, logit, ldl, R, w->vocab, w->d_model, nullptr, 0, 0.f, 1, 0.f, 0,
                   nullptr, 0, nullptr, 0, (void*)s);
  }
  DecGemmParams g;
  g.tok = nullptr; g.tok_stride = 0; g.t = 0; g.emb = nullptr; g.pe = nullptr; g.emb_scale = 0.f;
  g.M = R; g.N = w->vocab; g.K = w->d_model; g.relu = 0;
  static const int dev_cls_ntb = getenv("AUDIOCAPTION_DEC_CLS_NTB") ? atoi(getenv("AUDIOCAPTION_DEC_CLS_NTB")) : 4;   // development
  g.ntb = dev_cls_ntb;  // 1092 column tiles: four per block keep the grid near one resident wave of blocks
  g.X = fin.x; g.ldx = w->d_model; g.Y2 = fin.y; g.ldy2 = w->d_model; g.ln_w = fin.ln_w; g.ln_b = fin.ln_b;
  size_t cls_off;
  pack_layout(w, nullptr, &cls_off);
  g.Wp = w->step_pk + cls_off; g.bias = nullptr; g.Y = logit; g.ldy = ldl; g.xout = xout; g.ldxo = ldxo;
  return launch_dec_gemm<PRO_ADDLN>(g, s);
}

}  // namespace

extern "C" int ac_abi_version(void) { return AC_ABI_VERSION; }

extern "C" int ac_add_layernorm(const float* x, const float* y, const float* w, const float* b, float* out,
                                int rows, int d, long ldx, long ldy, long ldo, void* stream) {
  if (!x || !w || !b || !out || rows <= 0 || d <= 0 || d > 1024) return AC_ERR_ARG;
  return launch_ln(x, y, w, b, out, rows, d, ldx, ldy, ldo, (hipStream_t)stream);
}

extern "C" int ac_trm_memory(const ac_trm_weights* w, const float* attn_emb, int R, int Tm, float* memkv,
                             float* tmp, void* stream) {
  AC_TRY(check_weights(w));
  if (!attn_emb || !memkv || !tmp || R <= 0 || Tm <= 0 || Tm > MAX_KEYS) return AC_ERR_ARG;
  const int d = w->d_model, rows = R * Tm;
  hipStream_t s = (hipStream_t)stream;
  // attn_proj: Linear -> ReLU -> (Dropout: identity in eval) -> LayerNorm
  AC_TRY(ac_linear(attn_emb, w->proj_w, w->proj_b, tmp, rows, d, w->attn_emb_dim, w->attn_emb_dim,
                   w->attn_emb_dim, d, 1, stream));
  AC_TRY(launch_ln(tmp, nullptr, w->proj_ln_w, w->proj_ln_b, tmp, rows, d, d, 0, d, s));
  // cross-attention K and V of every layer: rows d..3d-1 of multihead_attn.in_proj
  for (int l = 0; l < w->nlayers; ++l) {
    const ac_trm_layer& L = w->layer[l];
    AC_TRY(ac_linear(tmp, L.ca_in_w + (size_t)d * d, L.ca_in_b + d, memkv + (size_t)l * rows * 2 * d, rows,
                     2 * d, d, d, d, 2 * d, 0, stream));
  }
  return AC_OK;
}

extern "C" long ac_trm_step_pack_floats(const ac_trm_weights* w) {
  if (check_weights(w) != AC_OK) return -1;
  return (long)pack_layout(w, nullptr, nullptr);
}

extern "C" int ac_trm_pack_step_weights(const ac_trm_weights* w, float* out, void* stream) {
  AC_TRY(check_weights(w));
  if (!out || w->d_model % 64 || w->dim_ff % 64) return AC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int d = w->d_model, ff = w->dim_ff;
  PackLayout PL[AC_MAX_LAYERS];
  size_t cls_off, wcls_off;
  pack_layout(w, PL, &cls_off, &wcls_off);
  const bool wide = wide_shape_ok(w);
  auto wpack = [&](const float* W, int N, int K, float* dst) {
    return wide ? ac_dec_wide_pack(W, (long)K, N, K, dst, stream) : AC_OK;
  };
  auto pack = [&](const float* W, int N, int K, float* dst) {
    const size_t n4 = packed_floats(N, K) / 4;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, W, (long)K, N, K, dst);
    return ac_check_launch();
  };
  for (int l = 0; l < w->nlayers; ++l) {
    const ac_trm_layer& L = w->layer[l];
    AC_TRY(pack(L.sa_in_w, 3 * d, d, out + PL[l].sa_in));
    AC_TRY(pack(L.sa_out_w, d, d, out + PL[l].sa_out));
    AC_TRY(pack(L.ca_in_w, d, d, out + PL[l].ca_q));  // rows 0..d-1 of in_proj = the query projection
    AC_TRY(pack(L.ca_out_w, d, d, out + PL[l].ca_out));
    AC_TRY(pack(L.l1_w, ff, d, out + PL[l].l1));
    AC_TRY(pack(L.l2_w, d, ff, out + PL[l].l2));
    AC_TRY(wpack(L.sa_in_w, 3 * d, d, out + PL[l].w_sa_in));
    AC_TRY(wpack(L.sa_out_w, d, d, out + PL[l].w_sa_out));
    AC_TRY(wpack(L.ca_in_w, d, d, out + PL[l].w_ca_q));
    AC_TRY(wpack(L.ca_out_w, d, d, out + PL[l].w_ca_out));
    AC_TRY(wpack(L.l1_w, ff, d, out + PL[l].w_l1));
    AC_TRY(wpack(L.l2_w, d, ff, out + PL[l].w_l2));
    if (d % 32 == 0) {
      const float* src[3] = {L.sa_out_w, L.ca_in_w, L.ca_out_w};
      const size_t dst[3] = {PL[l].sa_outT, PL[l].ca_qT, PL[l].ca_outT};
      for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(transpose_sq_kernel, dim3(d / 32, d / 32), dim3(32, 8), 0, s, src[i], (long)d, d, out + dst[i]);
        AC_TRY(ac_check_launch());
      }
    }
  }
  AC_TRY(pack(w->cls_w, w->vocab, d, out + cls_off));
  return wpack(w->cls_w, w->vocab, d, out + wcls_off);
}

extern "C" long ac_trm_workspace_floats(const ac_trm_weights* w, int rows, int max_len) {
  if (check_weights(w) != AC_OK || rows <= 0 || max_len <= 0) return -1;
  return (long)carve(w, rows, max_len, nullptr).total;
}

extern "C" int ac_trm_greedy(const ac_trm_weights* w, const float* memkv, const int* mem_len, int B, int Tm,
                             int max_len, int start_idx, int end_idx, int pad_idx, int64_t* seq, float* logit,
                             float* logprob, float* embed, int* unfinished_cnt, float* ws_base, void* stream) {
  AC_TRY(check_weights(w));
  if (!memkv || !mem_len || !seq || !logit || !logprob || !embed || !unfinished_cnt || !ws_base) return AC_ERR_ARG;
  if (B <= 0 || Tm <= 0 || Tm > MAX_KEYS || max_len <= 0 || max_len > w->max_pos) return AC_ERR_ARG;
  if (w->vocab > PICK_MAXV) return AC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const Ws ws = carve(w, B, max_len, ws_base);
  const int d = w->d_model, V = w->vocab;
  const int n_init = B * (max_len + 1);
  hipLaunchKernelGGL(greedy_init_kernel, dim3((n_init + 255) / 256), dim3(256), 0, s, seq, logprob, ws.tok,
                     ws.mask, ws.unfinished, unfinished_cnt, B, max_len, start_idx, end_idx, pad_idx);
  AC_TRY(ac_check_launch());
  for (int t = 0; t < max_len; ++t) {
    StepOut fin;
    AC_TRY(decoder_step(w, memkv, mem_len, B, 1, Tm, max_len, t, ws.tok, ws.mask, max_len + 1, ws.cache[0], ws,
                        &fin, s));
    AC_TRY(classifier_step(w, fin, B, embed + (size_t)t * d, (long)max_len * d, logit + (size_t)t * V,
                           (long)max_len * V, s));
    PickParams p;
    p.logit = logit + (size_t)t * V; p.ldl = (long)max_len * V;
    p.V = V; p.t = t; p.max_len = max_len; p.end_idx = end_idx; p.pad_idx = pad_idx;
    p.seq = seq; p.logprob = logprob; p.tok = ws.tok; p.mask = ws.mask; p.unfinished = ws.unfinished;
    p.cnt = unfinished_cnt;
    hipLaunchKernelGGL(greedy_pick_kernel, dim3(B), dim3(256), 0, s, p);
    AC_TRY(ac_check_launch());
  }
  return AC_OK;
}

extern "C" int ac_trm_forward_tokens(const ac_trm_weights* w, const float* memkv, const int* mem_len, int N,
                                     int Tm, const int* tokens, const unsigned char* key_mask, int T,
                                     float* embed, float* logit, float* ws_base, void* stream) {
  AC_TRY(check_weights(w));
  if (!memkv || !mem_len || !tokens || !embed || !logit || !ws_base) return AC_ERR_ARG;
  if (N <= 0 || Tm <= 0 || Tm > MAX_KEYS || T <= 0 || T > w->max_pos) return AC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const Ws ws = carve(w, N, T, ws_base);
  const int d = w->d_model, V = w->vocab;
  for (int t = 0; t < T; ++t) {
    StepOut fin;
    AC_TRY(decoder_step(w, memkv, mem_len, N, 1, Tm, T, t, tokens, key_mask, T, ws.cache[0], ws, &fin, s));
    AC_TRY(classifier_step(w, fin, N, embed + (size_t)t * d, (long)T * d, logit + (size_t)t * V, (long)T * V, s));
  }
  return AC_OK;
}

extern "C" int ac_trm_beam_step(const ac_trm_weights* w, const float* memkv, const int* mem_len, int B, int beam,
                                int Tm, int max_len, int t, float temp, const int* tokens,
                                const unsigned char* key_mask, const float* cum_logprob, float* top_val,
                                int* top_idx, float* ws_base, void* stream) {
  AC_TRY(check_weights(w));
  if (!memkv || !mem_len || !tokens || !cum_logprob || !top_val || !top_idx || !ws_base) return AC_ERR_ARG;
  if (B <= 0 || beam <= 0 || beam > 64 || Tm <= 0 || Tm > MAX_KEYS || t < 0 || t >= max_len) return AC_ERR_ARG;
  if (max_len > w->max_pos || !(temp > 0.f)) return AC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int R = B * beam, d = w->d_model, V = w->vocab;
  const Ws ws = carve(w, R, max_len, ws_base);
  StepOut fin;
  AC_TRY(decoder_step(w, memkv, mem_len, R, beam, Tm, max_len, t, tokens, key_mask, max_len + 1,
                      ws.cache[t & 1], ws, &fin, s));
  AC_TRY(classifier_step(w, fin, R, nullptr, 0, ws.lg, V, s, ws.att));   // ws.att: dead once the last layer has consumed it
  static const bool two_kernels = getenv("AUDIOCAPTION_BEAM_TOPK") && !strcmp(getenv("AUDIOCAPTION_BEAM_TOPK"), "scan");
  if (beam <= 8 && V <= 8192 && !two_kernels) {
    // scores + per-row candidates in one pass over registers, then a one-wave merge per clip; the candidates (2 x R x
    // beam words) go into the QKV scratch of the decoder step, which is dead until the next step starts
    float* cand_val = ws.qkv;
    int* cand_idx = (int*)(cand_val + (size_t)R * beam);
    if (V <= 5120) hipLaunchKernelGGL(beam_row_topk_kernel<20>, dim3(R), dim3(256), 0, s, ws.lg, cum_logprob, temp, beam, V, cand_val, cand_idx);
    else hipLaunchKernelGGL(beam_row_topk_kernel<32>, dim3(R), dim3(256), 0, s, ws.lg, cum_logprob, temp, beam, V, cand_val, cand_idx);
    AC_TRY(ac_check_launch());
    hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(64), 0, s, cand_val, cand_idx, beam, t == 0 ? 1 : beam, top_val, top_idx);
    return ac_check_launch();
  }
  // lp is written over the qkv/ff scratch?  No: it needs R*V floats, reuse a second logits-sized area.
  float* lp = ws.lg;  // in place: every element is read before it is written by the same thread
  hipLaunchKernelGGL(beam_logprob_kernel, dim3(R), dim3(256), 0, s, ws.lg, cum_logprob, temp, lp, V);
  AC_TRY(ac_check_launch());
  hipLaunchKernelGGL(beam_topk_kernel, dim3(B), dim3(256), 0, s, lp, beam, t == 0 ? 1 : beam, V, top_val,
                     top_idx);
  return ac_check_launch();
}

extern "C" int ac_trm_beam_update(const float* top_val, const int* top_idx, const int* tokens_in, int* tokens_out,
                                  unsigned char* key_mask_out, float* cum_logprob, int* active, int* done_count,
                                  int* done_seq, float* done_score, int* src_row, int* n_active, int B, int beam, int V,
                                  int max_len, int t, int end_idx, int pad_idx, int done_capacity, void* stream) {
  if (!top_val || !top_idx || !tokens_in || !tokens_out || !key_mask_out || !cum_logprob || !active || !done_count ||
      !done_seq || !done_score || !src_row || !n_active || B <= 0 || beam <= 0 || beam > 64 || V <= 0 || max_len <= 0 ||
      t < 0 || t >= max_len || done_capacity <= 0)
    return AC_ERR_ARG;
  hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, top_val, top_idx, tokens_in, tokens_out,
                     key_mask_out, cum_logprob, active, done_count, done_seq, done_score, src_row, n_active, beam, V,
                     max_len, t, end_idx, pad_idx, done_capacity);
  return ac_check_launch();
}

extern "C" int ac_trm_beam_reorder(const ac_trm_weights* w, int R, int max_len, int t, const int* src_row,
                                   float* ws_base, void* stream) {
  AC_TRY(check_weights(w));
  if (!src_row || !ws_base || R <= 0 || t < 0 || t >= max_len) return AC_ERR_ARG;
  const Ws ws = carve(w, R, max_len, ws_base);
  hipLaunchKernelGGL(cache_gather_kernel, dim3(R, 2 * w->nlayers), dim3(256), 0, (hipStream_t)stream,
                     ws.cache[t & 1], ws.cache[(t + 1) & 1], src_row, R, max_len, t, w->d_model,
                     ws.cache_set_stride, 2 * w->nlayers);
  return ac_check_launch();
}
