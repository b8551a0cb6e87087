/*
 * C ABI of libaudiocaption_hip.so - the MI355X (gfx950) kernels behind the audio-captioning hot path
 *     wav -> log-mel -> Cnn14 -> bi-GRU -> Transformer decoder (greedy / beam).
 *
 * The reference (wsntxxn/AudioCaption) is pure Python/PyTorch and has no FFI: its boundary is the
 * Python plugin protocol.  This library sits UNDER the Python classes of
 * audiocaption_amd/ that mirror that protocol; each entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host";
 *   - float = IEEE fp32; token ids are int32 on the device side, int64 in the seq output
 *     (the reference returns int64, base.py:122);
 *   - `stream` is a hipStream_t (pass the caller's current PyTorch stream); all work is stream-ordered,
 *     nothing synchronises, nothing allocates; outputs are caller-allocated;
 *   - return value: 0 = ok, AC_ERR_ARG (-1) = rejected arguments, AC_ERR_LAUNCH (-2) = HIP launch error.
 *     Nothing throws across the ABI.
 *   - no hidden global state.
 */
#ifndef AUDIOCAPTION_HIP_H
#define AUDIOCAPTION_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AC_OK 0
#define AC_ERR_ARG (-1)
#define AC_ERR_LAUNCH (-2)

#define AC_ABI_VERSION 1
int ac_abi_version(void);

/* ---- log-mel front-end -------------------------------------------------------------------------
 * Replaces torchaudio MelSpectrogram + AmplitudeToDB + transposes + bn0 as called at
 * cnn_encoder.py:418-429 (Cnn14: n_fft 1024, hop 320) / hf_wrapper.py:292-293 (EffB2: n_fft 512, hop 160).
 * wav (B, L) -> out[b*stride_b + t*stride_t + m*stride_m], t < T = L/hop + 1, m < 64;
 * frames T <= t < rows_per_clip are written as 0.  scale/shift (64 each, may be NULL) fold bn0.
 * window [n_fft], twiddle [n_fft] complex (re,im) = exp(-2 pi i n/n_fft), melfb [n_fft/2+1][64],
 * mel_lo/mel_hi [64] = first/last non-zero bin of each filter. */
int ac_logmel(const float* wav, int B, int L, int n_fft, int hop, const float* window, const float* twiddle,
              const float* melfb, const int* mel_lo, const int* mel_hi, const float* scale, const float* shift,
              float* out, int rows_per_clip, long stride_b, long stride_t, long stride_m, void* stream);

/* ---- Cnn14 conv stack ---------------------------------------------------------------------------
 * Activations are channels-last with row padding: [B*Hp][W][C], rows h >= H of every clip are zero
 * (see csrc/conv3x3.hip).  Replaces ConvBlock.forward cnn_encoder.py:59-75 (conv3x3 + eval BN + ReLU
 * [+ avg_pool2d]) and the mean/transposes of cnn_encoder.py:443-444.
 *   mode 0: out [B*Hp][W][Cout]           (conv1 of a block)
 *   mode 1: out [B*Hp/2][W/2][Cout]       (conv2 + 2x2 average pooling)
 *   mode 2: out [B][H][Cout] dense        (last conv2 + mean over the W == 2 mel columns = attn_emb)
 * wpk = weights packed as [Cin/32][9][Cout][32] (chunk, tap = ky*3+kx, out channel, in channel % 32); scale/shift = folded BatchNorm.
 * map_mode: -1 auto, 0 linear, 1 weight-slab-per-XCD, 2 halo-patch-per-XCD block mapping. */
int ac_conv3x3_bn_relu(const float* in, const float* wpk, const float* scale, const float* shift, float* out,
                       int B, int Hp, int H, int W, int Cin, int Cout, int mode, int map_mode, void* stream);
/* Same operation in Winograd F(2x2,3x3) form (2.25x fewer multiplications, fp32): identical arguments
 * except the weights, upk = U = G g G^T packed as [Cin/32][4 j][4 i][Cout][32] (transform column j, row i);
 * Hp must be even. */
int ac_conv3x3_bn_relu_winograd(const float* in, const float* upk, const float* scale, const float* shift,
                                float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                int map_mode, void* stream);
/* Same operation on split-bf16 operands ("bf16x3": x = hi + lo in bf16, hi*hi + hi*lo + lo*hi accumulated in
 * f32 on v_mfma_f32_32x32x16_bf16; ~2^-16 relative operand error, 5.3x the f32 matrix rate) - the 1e-3-logit
 * precision tier.  Activations in/out stay f32; wpk = weights split offline and packed as
 * [Cin/32][9][2 (hi, lo)][Cout][32] bf16. */
int ac_conv3x3_bn_relu_bf16x3(const float* in, const void* wpk, const float* scale, const float* shift, float* out,
                              int B, int Hp, int H, int W, int Cin, int Cout, int mode, int map_mode, void* stream);
/* bf16x3 with the weight fragments read straight from L2 (no LDS weight ring, no per-tap barrier):
 * wfrag = split weights in MFMA fragment order [Cin/32][9][2 k-steps][Cout/32][2 (hi, lo)][64 lanes][8] bf16,
 * lane = (cout % 32) + 32 * ((cin % 16) / 8), element = cin % 8. */
int ac_conv3x3_bn_relu_bf16x3_gw(const float* in, const void* wfrag, const float* scale, const float* shift,
                                 float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                 int map_mode, void* stream);
/* First conv (Cin = 1): in [B*Hp][64], w [64][9] (OIHW), out [B*Hp][64][64]. */
int ac_conv3x3_first(const float* in, const float* w, const float* scale, const float* shift, float* out,
                     int B, int Hp, int H, int W, void* stream);

/* ---- dense projection ---------------------------------------------------------------------------
 * Y[M,N] = act(X[M,K] W[N,K]^T + bias): every F.linear of the path (rnn_encoder.py:41 input
 * projections, transformer_decoder.py:86,95-101).  K % 32 == 0, ldx/ldw % 4 == 0, 16-byte aligned X/W. */
int ac_linear(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, long ldx,
              long ldw, long ldy, int relu, void* stream);

/* ---- GRU recurrence + pooling -------------------------------------------------------------------
 * One bidirectional layer of the packed GRU (rnn_encoder.py:41 via model_util.py:22-27):
 * gx [B][T][2][3H] = input projections incl. b_ih (gate order r,z,n), whhT [2][H][3H] = W_hh^T,
 * bhh [2][3H], lens [B] int32; out [B][T][2H], zeros at t >= lens[b].  H must be 256. */
int ac_gru_layer(const float* gx, const float* whhT, const float* bhh, const int* lens, float* out, int B,
                 int T, int hidden, void* stream);
/* mean_with_lens (model_util.py:41-63); add_max != 0 adds max_with_lens (cnn_encoder.py:451-453). */
int ac_mean_with_lens(const float* x, const int* lens, float* out, int B, int T, int C, int add_max,
                      void* stream);

/* ---- Transformer decoder ------------------------------------------------------------------------ */
#define AC_MAX_LAYERS 8
typedef struct {
  const float *sa_in_w, *sa_in_b, *sa_out_w, *sa_out_b; /* self_attn.in_proj / out_proj       */
  const float *ca_in_w, *ca_in_b, *ca_out_w, *ca_out_b; /* multihead_attn.in_proj / out_proj  */
  const float *l1_w, *l1_b, *l2_w, *l2_b;               /* linear1 / linear2                  */
  const float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b; /* norm1..3                           */
} ac_trm_layer;

typedef struct {
  int32_t d_model, nhead, nlayers, dim_ff, vocab, max_pos, attn_emb_dim, reserved;
  const float* emb;       /* word_embedding.weight [V][d]            */
  const float* pe;        /* pos_encoder.pe        [max_pos][d]      */
  const float* cls_w;     /* classifier.weight     [V][d] (no bias)  */
  const float* proj_w;    /* attn_proj.0.weight    [d][attn_emb_dim] */
  const float* proj_b;    /* attn_proj.0.bias                        */
  const float* proj_ln_w; /* attn_proj.3 (LayerNorm)                 */
  const float* proj_ln_b;
  const float* step_pk;   /* fragment-packed step weights from ac_trm_pack_step_weights */
  ac_trm_layer layer[AC_MAX_LAYERS];
} ac_trm_weights;

/* Row-wise LayerNorm of (x + y) (y may be NULL), eps 1e-5: the post-LN residual blocks of
 * nn.TransformerDecoderLayer and attn_proj's LayerNorm. */
int ac_add_layernorm(const float* x, const float* y, const float* w, const float* b, float* out, int rows,
                     int d, long ldx, long ldy, long ldo, void* stream);

/* Memory side of the decoder, once per batch (transformer_decoder.py:86 + the K/V in-projections of
 * every layer's cross attention, which the reference recomputes on every call):
 *   attn_emb [R*Tm][attn_emb_dim] -> memkv [nlayers][R*Tm][2*d]  (K then V);  tmp: R*Tm*d floats. */
int ac_trm_memory(const ac_trm_weights* w, const float* attn_emb, int R, int Tm, float* memkv, float* tmp,
                  void* stream);

/* The per-position projections read their weights in MFMA fragment order (one contiguous 1 KiB read per
 * wave and fragment).  ac_trm_step_pack_floats = size of that copy; ac_trm_pack_step_weights writes it from
 * the row-major pointers of `w` (w->step_pk itself is not read); the caller then stores `out` in
 * w->step_pk.  Required by ac_trm_greedy / ac_trm_forward_tokens / ac_trm_beam_step; redo it whenever the
 * weights change. */
long ac_trm_step_pack_floats(const ac_trm_weights* w);
int ac_trm_pack_step_weights(const ac_trm_weights* w, float* out, void* stream);

/* Number of workspace floats ac_trm_greedy / ac_trm_forward_tokens / ac_trm_beam need. */
long ac_trm_workspace_floats(const ac_trm_weights* w, int rows, int max_len);

/* Greedy decoding (base.py:152-218 stepwise_forward + sample_next_word + stepwise_process_step,
 * transformer_model.py:34-57), fully on device with a self-attention KV cache:
 *   memkv from ac_trm_memory, mem_len [B] int32 (= attn_emb_len),
 *   seq [B][max_len] int64, logit [B][max_len][V], logprob [B][max_len], embed [B][max_len][d];
 *   unfinished_cnt [max_len] int32: rows still unfinished after step t (step t+1.. are the steps the
 *   reference would not have executed once this reaches 0; their seq/logprob columns keep the
 *   reference's initial values end_idx / 0).  ws: ac_trm_workspace_floats(w, B, max_len) floats. */
int ac_trm_greedy(const ac_trm_weights* w, const float* memkv, const int* mem_len, int B, int Tm, int max_len,
                  int start_idx, int end_idx, int pad_idx, int64_t* seq, float* logit, float* logprob,
                  float* embed, int* unfinished_cnt, float* ws, void* stream);

/* Decoder forward on given tokens (teacher forcing / plugin call, transformer_decoder.py:80-103):
 * tokens [N][T] int32; key_mask [N][T] uint8 (1 = masked key, the reference's cap_padding_mask /
 * tgt_key_padding_mask, transformer_model.py:22-23,55) or NULL.  embed [N][T][d], logit [N][T][V]. */
int ac_trm_forward_tokens(const ac_trm_weights* w, const float* memkv, const int* mem_len, int N, int Tm,
                          const int* tokens, const unsigned char* key_mask, int T, float* embed, float* logit,
                          float* ws, void* stream);

/* One beam-search step over R = B*beam rows (base.py:269-289): runs the decoder for position t on
 * tokens[R][max_len+1] / key_mask[R][max_len+1] (column t is the input token; row r uses the audio
 * memory of clip r / beam), forms log_softmax(log_softmax(logit)/temp) + cum_logprob[R] and returns,
 * per clip, the `beam` best candidates over the flattened (beam*V) scores (only the clip's first row at
 * t == 0, base.py:285-289): top_val [B][beam], top_idx [B][beam] (flattened index beam_i*V + word).
 * The self-attention KV cache set (t & 1) of the workspace is the active one. */
int ac_trm_beam_step(const ac_trm_weights* w, const float* memkv, const int* mem_len, int B, int beam, int Tm,
                     int max_len, int t, float temp, const int* tokens, const unsigned char* key_mask,
                     const float* cum_logprob, float* top_val, int* top_idx, float* ws, void* stream);
/* Re-gather the beams after selection (base.py:294-302): row r of cache set ((t+1) & 1) takes the
 * self-attention KV cache (positions 0..t) of row src_row[r] of set (t & 1). */
int ac_trm_beam_reorder(const ac_trm_weights* w, int R, int max_len, int t, const int* src_row, float* ws,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOCAPTION_HIP_H */
