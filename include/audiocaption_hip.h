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

#define AC_ABI_VERSION 2
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
/* The same operation (ConvBlock.forward, cnn_encoder.py:59-75) with 1.5x fewer multiplications: 1-D Winograd F(2,3)
 * along the time axis (row pairs) on split-bf16 operands - the input transform V = B^T d (+-1 coefficients) in f32,
 * then hi + lo; the filter transform U = G g evaluated offline in f64, then hi + lo; three bf16 MFMA products per
 * transformed product, f32 accumulation, output transform + BN + ReLU (+ pool / mean) in the epilogue.  Two bf16 MFMA
 * products per f32 product of the direct form (ac_conv3x3_bn_relu_bf16x3_gw: three) at the same f32-grade accuracy.
 * in / out f32 with the layouts of ac_conv3x3_bn_relu.  wfrag = U split and packed in MFMA fragment order
 * [Cin/32][3 kx][4 positions][2 k-steps][Cout/32][2 (hi, lo)][64 lanes][8] bf16, lane = (cout % 32) + 32 * ((cin % 16) / 8),
 * element = cin % 8.  Requires Hp even, W = 2 or a multiple of 4, Cin % 32 == 0, and Cout % 128 == 0 or Cout == 64 with
 * W % 16 == 0 (conv2 of block 1) (AC_ERR_ARG otherwise; mode 1 needs W >= 4, mode 2 needs W == 2), and an input of less
 * than 2 GiB, (B * Hp + 16) * W * Cin * 4 < 2^31 (32-bit byte offsets through one buffer descriptor; AC_ERR_ARG beyond:
 * the caller convolves the batch in clip chunks - the clips of a batch do not interact).
 * Ragged batches (the reference pads every clip to the batch maximum and convolves the padding, collate_func.py:29-32,
 * cnn_encoder.py:446-450): clip_frames (device int32 [B], may be NULL) = every clip's own attn_emb_len; workgroups whose
 * output rows all lie at or beyond need_mul * clip_frames[b] + need_add of their clip(s) skip the convolution and store
 * zeros.  The caller derives (need_mul, need_add) per layer from the receptive field downstream.  With this F(2,3) kernel
 * every output frame below clip_frames[b] - all the temporal encoder reads (model_util.py:10-27) - is then bit-identical
 * to the run that convolves the padding.  The F(4,3) kernels below form a row quad from six input rows: the same
 * (need_mul, need_add) leave valid frames within the tier's own error of the dense run (measured <= 5e-5, token ids
 * equal), and bit-identity needs the wider windows of cnn_encoder.rows_needed(quads=True) (AUDIOCAPTION_RAGGED_EXACT=1). */
int ac_conv3x3_bn_relu_wino1d(const float* in, const void* wfrag, const float* scale, const float* shift,
                              float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                              int map_mode, const int* clip_frames, int need_mul, int need_add, void* stream);
/* The same layer for launches of a few workgroups (single clips: conv2 of block 6 at B = 1 is 16 workgroups streaming
 * 200 MB of weights): the K loop is cut into slices on separate workgroups, every slice stores its transformed sums to
 * workspace[slice][B*Hp][W][Cout], and a second kernel adds the slices IN ORDER (deterministic) and applies BN / ReLU /
 * pool / mean.  ac_conv3x3_wino1d_splitk_floats: floats of workspace the geometry needs, 0 when the launch is not split
 * (the call then equals ac_conv3x3_bn_relu_wino1d).  clip_frames / need_mul / need_add as above: skipped blocks come out
 * as zeros here too. */
/* The same layer followed by F.dropout(drop_p) on its output (the train-mode forward of the frozen network,
 * cnn_encoder.py:431-442), applied in the epilogue: output element i of the output buffer is scaled by the counter-hash
 * mask ac_dropout(seed, seed_dev) gives index i - bit-identical to the layer followed by ac_dropout over the buffer,
 * without the extra pass over it.  Modes 0 and 1 (the mean over mel of the last block comes after its dropout). */
int ac_conv3x3_bn_relu_wino1d_drop(const float* in, const void* wfrag, const float* scale, const float* shift,
                                   float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode, int map_mode,
                                   float drop_p, unsigned long long drop_seed, const unsigned long long* seed_dev,
                                   void* stream);
long ac_conv3x3_wino1d_splitk_floats(int B, int Hp, int W, int Cin, int Cout);
int ac_conv3x3_bn_relu_wino1d_splitk(const float* in, const void* wfrag, const float* scale, const float* shift,
                                     float* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                     int map_mode, const int* clip_frames, int need_mul, int need_add,
                                     float* workspace, long workspace_floats, void* stream);

/* The same layer for FEW PIXELS and heavy weights (conv blocks 5-6 at one to a few clips: conv2 of block 6 is 64 pixels
 * against 151 MB of weights): a K-sliced DIRECT convolution on split-bf16 operands built to stream the weights once at
 * memory bandwidth (csrc/conv3x3_skinny.hip) - 9 taps instead of the 12 / 18 transformed ones, eight fragment groups in
 * flight per wave, all pixels of up to eight 32-pixel tiles per workgroup; the slices' raw sums go to
 * workspace[slice][B*Hp*W][Cout] and a second kernel adds them IN ORDER (deterministic) and applies BN / ReLU / pool /
 * mean and, with drop_p > 0, F.dropout on the output (the mask ac_dropout gives the output buffer; modes 0 and 1).
 * ConvBlock.forward, cnn_encoder.py:59-75 (+ the pooling / mean of :431-444); the arithmetic of
 * ac_conv3x3_bn_relu_bf16x3_gw (2^-16 operand error), wfrag = ITS pack ([Cin/32][9][2][Cout/32][hi, lo][64][8] bf16).
 * W = 4 (modes 0, 1) or 2 (modes 0, 2), Cin % 32 == 0, Cout % 128 == 0, B * Hp * W <= 2048 pixels.
 * ac_conv3x3_skinny_workspace_floats: floats of workspace the geometry needs, 0 when it is not covered (AC_ERR_ARG). */
long ac_conv3x3_skinny_workspace_floats(int B, int Hp, int W, int Cin, int Cout);
int ac_conv3x3_bn_relu_skinny(const float* in, const void* wfrag, const float* scale, const float* shift, float* out, int B,
                              int Hp, int H, int W, int Cin, int Cout, int mode, float* workspace, long workspace_floats,
                              float drop_p, unsigned long long drop_seed, const unsigned long long* seed_dev, void* stream);

/* Second generation of the f32-grade tier: the same layer as a 1-D Winograd F(4,3) along time (row QUADS: 6 transformed
 * positions per 4 output rows, 18 instead of 36 products per (cin, cout)) on split-bf16 operands - 1.5 bf16 MFMA products
 * per f32 product (F(2,3): 2) - with ONE 512-register wave per SIMD (csrc/conv3x3_wino43.hip).  Same layouts, epilogue
 * modes 0 / 1 and ragged-batch arguments as ac_conv3x3_bn_relu_wino1d; replaces the same reference code: ConvBlock.forward,
 * cnn_encoder.py:59-75 (+ the pooling of Cnn14Encoder.forward, :431-441).  Covers the full-width layers W = 32, 16, 8, 4
 * (conv blocks 2-5; modes 0, 1) and W = 2 (block 6: column tiles that skip the taps on the zero padding; modes 0 and
 * 2 = mean over the two mel columns, out (B, H, Cout), cnn_encoder.py:443) with Cout % 128 == 0, Cin % 32 == 0,
 * Hp % 4 == 0; wfrag
 * [Cin/16][3 kx x 6 positions][Cout/32][hi, lo][64 lanes][8] bf16 (U = G g in f64, then split).  tiles_per_wave: 2 (192
 * accumulators, one workgroup per CU), 1 (96 accumulators, <= 256 registers: two workgroups per CU, one's prologue and
 * epilogue under the other's K loop - the short-K layers), or 0 = chosen by the layer's K steps.  The input is
 * addressed through a buffer descriptor rebased per workgroup, so any B * Hp * W * Cin is accepted (B * Hp < 2^23 rows: the
 * epilogue's row -> clip arithmetic; callers chunk clips beyond it).  Ragged batches: see the contract note above - valid
 * frames within 5e-5 of the dense run with the F(2,3) windows, bit-identical with the quad-wide ones.
 * AC_ERR_ARG for shapes outside this list. */
int ac_conv3x3_bn_relu_wino43(const float* in, const void* wfrag, const float* scale, const float* shift, float* out,
                              int B, int Hp, int H, int W, int Cin, int Cout, int mode, int map_mode, int tiles_per_wave,
                              const int* clip_frames, int need_mul, int need_add, void* stream);
/* ... followed by F.dropout in the epilogue (see ac_conv3x3_bn_relu_wino1d_drop). */
int ac_conv3x3_bn_relu_wino43_drop(const float* in, const void* wfrag, const float* scale, const float* shift, float* out,
                                   int B, int Hp, int H, int W, int Cin, int Cout, int mode, int map_mode, float drop_p,
                                   unsigned long long drop_seed, const unsigned long long* seed_dev, void* stream);
/* Workgroups ac_conv3x3_bn_relu_wino43 launches for a geometry at two tiles per wave (0: not covered): callers route
 * launches of a few workgroups (single clips) to the K-sliced F(2,3) form instead. */
long ac_conv3x3_wino43_workgroups(int B, int Hp, int W, int Cout);

/* Conv block 1 at f32 grade in ONE kernel (csrc/conv3x3_block1_w4.hip): conv1 (Cin = 1) + BN + ReLU is computed straight
 * into the staging of conv2, conv2 + BN + ReLU + 2x2 average pool runs as F(4,3) on split-bf16 operands; persistent
 * workgroups (one per CU) walk tiles of 8 rows x 64 columns.  The 64-channel intermediate never exists in HBM.
 * in1 [B*Hp][64] f32 (log-mel after bn0), w1 [64][9], wfrag2 = the F(4,3) pack of conv2 ([4][18][2][hi, lo][64][8] bf16),
 * out [B*Hp/2][32][64] f32.  Hp % 8 == 0.  Replaces ConvBlock.forward of conv_block1 + F.avg_pool2d, cnn_encoder.py:59-75 /
 * :431-432: bit-identical to ac_conv3x3_first followed by ac_conv3x3_block1_conv2_wino43.  clip_frames / need_mul /
 * need_add: ragged batches (see ac_conv3x3_bn_relu_wino1d); drop_p > 0: F.dropout on the pooled output in the epilogue
 * (see ac_conv3x3_bn_relu_wino1d_drop). */
int ac_conv3x3_block1_wino43(const float* in1, const float* w1, const float* scale1, const float* shift1,
                             const void* wfrag2, const float* scale2, const float* shift2, float* out, int B, int Hp,
                             int H, const int* clip_frames, int need_mul, int need_add, float drop_p,
                             unsigned long long drop_seed, const unsigned long long* seed_dev, void* stream);
/* The same kernel with conv1 on the matrix cores as well (the default of the host class): a [32 channels] x [K = 16: nine
 * taps times the BN scale + the BN shift against a constant 1] x [32 columns] split-bf16 product per row, whose result a lane
 * holds in the layout conv2's staging stores - 1760 of the 4700 vector instructions per tile gone for 8 % more MFMAs.  Same
 * arguments; conv1 has the split-bf16 grade of the tier's other layers (not bit-identical to ac_conv3x3_first).
 * cnn_encoder.py:59-75 / :431-432. */
int ac_conv3x3_block1_wino43_mfma(const float* in1, const float* w1, const float* scale1, const float* shift1,
                                  const void* wfrag2, const float* scale2, const float* shift2, float* out, int B, int Hp,
                                  int H, const int* clip_frames, int need_mul, int need_add, float drop_p,
                                  unsigned long long drop_seed, const unsigned long long* seed_dev, void* stream);
/* The conv2 half of the same kernel on a 64-channel input in HBM ([B*Hp][64][64] f32, B*Hp*64*64*4 < 2^31): the unfused
 * form the fused kernel is tested against. */
int ac_conv3x3_block1_conv2_wino43(const float* in64, const void* wfrag2, const float* scale2, const float* shift2,
                                   float* out, int B, int Hp, int H, void* stream);

/* "f16x2" tier of the same kernel.  Activations live in HBM as fp16 (in: [B*Hp][W][Cin] fp16; out: fp16 for modes 0
 * and 1, f32 for mode 2 = the attn_emb the rest of the path consumes), weights as fp16 hi + lo (2^-22) in the same
 * fragment order, two fp16 MFMA products per f32 product, f32 accumulation, fp16 rounding (RNE, 2^-12 relative) once
 * per stored activation.  The caller scales every output channel of the weights by a power of two before splitting
 * (so the lo parts stay normal fp16 numbers) and multiplies `scale` by the inverse.  Same shape arguments and error
 * codes as ac_conv3x3_bn_relu_bf16x3_gw.  Replaces the same reference code: ConvBlock.forward, cnn_encoder.py:318-338.
 * out_f32 != 0 (mode 1 only): the pooled output is written as f32 (it feeds a split-bf16 block: the mixed tier runs
 * conv_block6, K = 9216 / 18432, on ac_conv3x3_bn_relu_bf16x3_gw).  overflow_flag (may be NULL): one device word that is
 * OR-ed with 1 when a value about to be stored as fp16 exceeds the fp16 range (65504) - the caller re-runs the batch
 * on an f32-activation tier instead of returning inf/NaN-poisoned results.
 * AC_ERR_ARG when B*Hp*W*Cin >= 2^32 (32-bit staging offsets). */
int ac_conv3x3_bn_relu_f16x2_gw(const void* in, const void* wfrag, const float* scale, const float* shift,
                                void* out, int B, int Hp, int H, int W, int Cin, int Cout, int mode,
                                int map_mode, int out_f32, unsigned int* overflow_flag, void* stream);
/* Block 1 of the "f16x2" tier in one kernel: conv1 (Cin = 1) + BN + ReLU is computed straight into the LDS patch of
 * conv2 (the 64-channel intermediate never exists in HBM), conv2 + BN + ReLU + 2x2 average pool on the matrix cores.
 * in1 [B*Hp][64] f32 (log-mel after bn0), w1 [64][9], wfrag2 as for ac_conv3x3_bn_relu_f16x2_gw (Cin = Cout = 64),
 * out [B*Hp/2][32][64] fp16.  Replaces ConvBlock.forward of conv_block1, cnn_encoder.py:318-338 / :431. */
int ac_conv3x3_block1_f16x2(const float* in1, const float* w1, const float* scale1, const float* shift1,
                            const void* wfrag2, const float* scale2, const float* shift2, void* out,
                            int B, int Hp, int H, int W, unsigned int* overflow_flag, void* stream);

/* Y[M][N] = act(X[M][K] W[N][K]^T + bias) on the split-bf16 matrix path (2^-16 relative operand error, f32
 * accumulation): the one-tap instance of the "gw" convolution kernel.  X, Y dense row-major f32; wfrag = W split into
 * bf16 hi + lo in fragment order [K/32][1][2 k-steps][N/32][2][64 lanes][8]; ones = N floats of 1.0;
 * K % 32 == 0, N % 64 == 0.  Replaces the same reference code as ac_linear (nn.Linear / the nn.GRU input projections,
 * rnn_encoder.py:34-49) for the large layers. */
int ac_linear_bf16x3(const float* X, const void* wfrag, const float* ones, const float* bias, float* Y,
                     int M, int N, int K, int relu, void* stream);

/* First conv (Cin = 1): in [B*Hp][64], w [64][9] (OIHW), out [B*Hp][64][64]. */
int ac_conv3x3_first(const float* in, const float* w, const float* scale, const float* shift, float* out,
                     int B, int Hp, int H, int W, void* stream);
/* The same with an fp16 output [B*Hp][64][64] (first layer of the "f16x2" tier). */
int ac_conv3x3_first_f16(const float* in, const float* w, const float* scale, const float* shift, void* out,
                         int B, int Hp, int H, int W, unsigned int* overflow_flag, void* stream);

/* ---- dense projection ---------------------------------------------------------------------------
 * Y[M,N] = act(X[M,K] W[N,K]^T + bias): every F.linear of the path (rnn_encoder.py:41 input
 * projections, transformer_decoder.py:86,95-101).  K % 32 == 0, ldx/ldw % 4 == 0, 16-byte aligned X/W. */
int ac_linear(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, long ldx,
              long ldw, long ldy, int relu, void* stream);

/* ---- GRU recurrence + pooling -------------------------------------------------------------------
 * One bidirectional layer of the packed GRU (rnn_encoder.py:41 via model_util.py:22-27):
 * gx [B][T][2][3H] = input projections incl. b_ih (gate order r,z,n), whhT [2][H/4][3H][4] = W_hh packed by ac_gru_pack_whh,
 * bhh [2][3H], lens [B] int32; out [B][T][2H], zeros at t >= lens[b].  H must be 256. */
/* packed[d][k/4][n][k%4] = whh[d][n][k] for nn.GRU's weight_hh of both directions, whh [2][3H][H]. */
int ac_gru_pack_whh(const float* whh, float* packed, int hidden, void* stream);
int ac_gru_layer(const float* gx, const float* whhT, const float* bhh, const int* lens, float* out, int B,
                 int T, int hidden, void* stream);
/* The same layer with every (clip, direction) split over FOUR workgroups of 256 threads, each holding its quarter of W_hh
 * in registers for the whole sequence (nothing is re-streamed per step); the parts trade 64 hidden values each per step
 * through L2 as {tag, value} granules (agent-scope relaxed atomics).  whh = nn.GRU's weight_hh of both directions [2][3H][H], NOT
 * packed.  workspace: ac_gru_split_workspace_bytes(B) bytes, 8-byte aligned, owned by the caller; the call clears what
 * it needs (stream-ordered memset).  Its FIRST 4-byte word is a sticky error flag the caller zeroes once: non-zero when
 * a workgroup's partner never started within 2 s (the output is then invalid).  8*B workgroups of 256 threads.
 * save (may be NULL): [B][T][2][4H] = (r, z, n, W_hn h + b_hn) per cell, what ac_gru_layer_bwd needs (ac_gru_layer_train's
 * layout): the training forward. */
long ac_gru_split_workspace_bytes(int B);
int ac_gru_layer_split(const float* gx, const float* whh, const float* bhh, const int* lens, float* out, float* save,
                       void* workspace, int B, int T, int hidden, void* stream);
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

/* Decode-step projection for WIDE row batches (a greedy chain over several submissions, a beam search over grouped batches:
 * 200 ... 1000 rows per step), csrc/decoder_wide.hip:  Y[M][N] = act(P(X)[M][K] W[N][K]^T + bias), the F.linear calls of
 * nn.TransformerDecoderLayer and the classifier (transformer_decoder.py:92-101) with the producer of the A rows fused in -
 * producer 0: X is a fragment PACK of the rows (ac_dec_wide_pack of a [M][K] matrix, or the output of a split_out launch),
 *             K = 256, 512 or 1024, ntb = 1;
 * producer 1: A = emb[tok[r*tok_stride + t]] * emb_scale + pe[t] (transformer_decoder.py:89-91), K = 256;
 * producer 2: A = LayerNorm(X + Y2) * ln_w + ln_b, eps 1e-5 (the post-LN residual join), K = 256.
 * Producers 1 and 2 also store the produced rows to xout (may be NULL).  Arithmetic: both operands split into three bf16 planes
 * (24 significant bits), the six plane products of order <= 2 on v_mfma_f32_32x32x16_bf16 with f32 accumulation - f32-grade
 * (below the rounding of an f32 dot product) at 2.7x the f32 matrix rate.  Pack layout: [tile of 32 rows][k step of 16][plane]
 * [lane = row % 32 + 32 * (k % 16 / 8)][k % 8] bf16, ac_dec_wide_packed_floats(rows, K) floats, 16-byte aligned; Wp = the pack
 * of W.  split_out = 1 (producers 1 / 2, N % 16 == 0): Y receives the PACK of the [M][N] result instead of f32 rows.
 * ntb: 64-column groups per workgroup.  16-byte aligned rows of X / Y2 / xout (Y: any; 16-byte stores when its rows are aligned). */
long ac_dec_wide_packed_floats(int N, int K);
int ac_dec_wide_pack(const float* W, long ldw, int N, int K, float* out, void* stream);
int ac_dec_wide_gemm(int producer, const float* X, long ldx, const float* Y2, long ldy2, const float* ln_w,
                     const float* ln_b, const int* tok, long tok_stride, int t, const float* emb, const float* pe,
                     float emb_scale, float* xout, long ldxo, const float* Wp, const float* bias, float* Y, long ldy, int M,
                     int N, int K, int relu, int ntb, int split_out, void* stream);

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

/* The same greedy search (same arguments, same outputs; base.py:152-218, transformer_decoder.py:80-103) as ONE persistent
 * launch for the case that nothing else runs on the GPU (the blocking model() call, single clips): a row is decoded by a
 * cluster of four workgroups for all of its steps - part p owns head p of both attention sub-layers (its keys and values
 * stay in LDS: the projected audio memory of the row and the self-attention cache) and quarter p of the feed-forward
 * units and of the vocabulary; the parts meet in seven exchanges per step (tagged 8-byte granules through L2, the split
 * GRU kernel's hand-off) - csrc/decoder_cluster.hip.  ~45 us per step instead of the launch chain's ~90, for up to two
 * rows per CU; beyond that the clusters run in rounds and ac_trm_greedy is the faster call.  Sums are formed in another
 * order than ac_trm_greedy's (logits equal to ~1e-6, the same token ids on every fixture).
 * cluster_pk: ac_trm_cluster_pack_floats(w) floats written by ac_trm_cluster_pack (per-part weight blobs; redo when the
 * weights change).  workspace: ac_trm_cluster_workspace_bytes(B) bytes, 8-byte aligned; its FIRST 4-byte word is a sticky
 * error flag the caller zeroes once: non-zero when a workgroup's partners never started within 2 s (outputs invalid).
 * early_stop != 0: the launch ends one or two steps after every row has emitted <end> (the reference's loop ends there,
 * base.py:206-211; the columns it would not have written keep their initial values either way); 0: all max_len steps run
 * (what ac_trm_greedy's launches do).
 * Shapes: d_model 256, 4 heads, dim_ff 1024, max_len <= 32, nlayers * (Tm + max_len) * 512 + ~20 KB of LDS <= 160 KB;
 * AC_ERR_ARG otherwise (ac_trm_cluster_pack_floats: -1). */
long ac_trm_cluster_pack_floats(const ac_trm_weights* w);
int ac_trm_cluster_pack(const ac_trm_weights* w, float* out, void* stream);
long ac_trm_cluster_workspace_bytes(int B);
int ac_trm_greedy_cluster(const ac_trm_weights* w, const float* cluster_pk, const float* memkv, const int* mem_len, int B,
                          int Tm, int max_len, int start_idx, int end_idx, int pad_idx, int64_t* seq, float* logit,
                          float* logprob, float* embed, int* unfinished_cnt, void* workspace, int early_stop, void* stream);

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
/* Per-clip beam bookkeeping on the device (base.py:290-323), one call per step after ac_trm_beam_step: for every clip
 * still active, row k of tokens_out = row (top_idx / V) of tokens_in with the word (top_idx % V) appended at column
 * t + 1 (key_mask_out = tokens_out == pad_idx); beams whose word is end_idx (all of them at t == max_len - 1) are
 * appended, in beam order, to the clip's finished list done_seq [B][done_capacity][max_len] (padded with end_idx) /
 * done_score (= logprob / (t + 1)), their cumulative score gets the reference's -1000; the clip retires when its
 * finished count EQUALS beam (n_active is decremented).  src_row [B*beam] is what ac_trm_beam_reorder needs. */
int ac_trm_beam_update(const float* top_val, const int* top_idx, const int* tokens_in, int* tokens_out,
                       unsigned char* key_mask_out, float* cum_logprob, int* active, int* done_count, int* done_seq,
                       float* done_score, int* src_row, int* n_active, int B, int beam, int V, int max_len, int t,
                       int end_idx, int pad_idx, int done_capacity, void* stream);
/* Re-gather the beams after selection (base.py:294-302): row r of cache set ((t+1) & 1) takes the
 * self-attention KV cache (positions 0..t) of row src_row[r] of set (t & 1). */
int ac_trm_beam_reorder(const ac_trm_weights* w, int R, int max_len, int t, const int* src_row, float* ws,
                        void* stream);

/* ================================== training step ==================================
 * The reference trains GRU + decoder on the frozen Cnn14 with scheduled sampling: step t runs the decoder on a
 * (N, t+1) prefix and keeps the last position's logit (base.py:131-137,152-199, transformer_model.py:34-57).  The
 * host keeps every prefix pass in one row space (row = one token position of one pass); forward kernels work on the
 * row range of a pass, backward kernels on all rows at once.  Dropout masks are a counter hash of
 * (seed, element index) - splitmix64, keep iff the top 32 bits >= p * 2^32, kept values scaled by 1/(1-p) - so the
 * backward regenerates them and the CPU oracle reproduces them.  Every dropout site takes `seed` plus an optional
 * device word `seed_dev`: the effective seed is seed + (*seed_dev << 16), which lets a captured HIP graph draw fresh
 * masks on every replay (the host bumps the device word between replays).                                          */

/* C[M][N] (row pitch ldc) = epi( sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] ): X W^T / dY W / dY^T X with one kernel
 * (replaces F.linear and its autograd, e.g. transformer_decoder.py:86-101, rnn_encoder.py:41).
 * epi: + bias[n], activation (relu = 1: ReLU, 2: swish x*sigmoid(x), 3: sigmoid), dropout(drop_p, drop_seed, index (row0+m)*N+n),
 * + beta*C.  splitk > 1: K is cut into splitk slices whose partial sums are atomically ADDED to C (weight gradients, and
 * input gradients with few output tiles over a long reduction; beta == 1, or beta == 0: C is zero-filled first; no
 * bias/activation/dropout).  a_scale (optional): A(m,k) is multiplied by
 * a_scale[(m / a_rows)*K + k] on the way in - the squeeze-excite gate of an MBConv block applied inside its 1x1
 * projection (efficientnet_pytorch MBConvBlock.forward; call site hf_wrapper.py:231). */
int ac_gemm(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M, int N,
            int K, const float* bias, int relu, float beta, int splitk, float drop_p, unsigned long long drop_seed,
            const unsigned long long* seed_dev, long row0, const float* a_scale, int a_rows, void* stream);
/* The same contract on split-bf16 operands: every f32 operand is split into bf16 hi + lo
 * when it is staged (16 significant bits), three bf16 MFMAs per product, f32 accumulation - 2^-16 relative operand error
 * at ~5x the f32 matrix rate.  Used for the large GEMMs of the training step (backward dY W / dY^T X over 7 392 rows,
 * teacher-forced forward).  Operands must be contiguous along k or along their row dimension with 16-byte aligned rows;
 * anything else, and products under ~3e7 multiply-adds or under 200 tiles of 64 x 64, are forwarded to ac_gemm (exact f32). */
int ac_gemm_bf16x3(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M,
                   int N, int K, const float* bias, int relu, float beta, int splitk, float drop_p,
                   unsigned long long drop_seed, const unsigned long long* seed_dev, long row0, const float* a_scale,
                   int a_rows, void* stream);
/* y[i] = x[i] * mask(seed, idx0 + i): F.dropout (cnn_encoder.py:432-442, nn.GRU inter-layer dropout); applying it to
 * a gradient with the same seed is its backward. */
int ac_dropout(const float* x, float* y, long n, float p, unsigned long long seed, const unsigned long long* seed_dev,
               long idx0, void* stream);
/* g[i] = h[i] > 0 ? g[i] * scale : 0 - backward of dropout(relu(.)) given its output h (FFN hidden). */
int ac_mask_pos_scale(float* g, const float* h, long n, float scale, void* stream);
/* Decoder input tokens of scheduled-sampling step t (transformer_model.py:44-52): word[row0 + n*L + l] =
 * use_cap[t] ? cap[n][l] : (l == 0 ? start_idx : seq[n][l-1]); cap int64 [N][cap_ld], seq int32 [N][seq_ld]. */
int ac_build_prefix(const long long* cap, int cap_ld, const int* seq, int seq_ld, const int* use_cap, int t,
                    int start_idx, int* word, long row0, int N, int L, void* stream);
/* x[row] = dropB(dropA(E[word[row]]) * sqrt(d) + pe[pos[row]]) for rows row0..row0+rows (transformer_decoder.py:88-90;
 * in_dropout and PositionalEncoding's dropout), and its backward into the embedding table (atomic adds, all rows). */
int ac_embed_fwd(const float* emb, const float* pe, const int* word, const int* pos, float* x, long row0, long rows,
                 int d, float pa, unsigned long long seed_a, float pb, unsigned long long seed_b,
                 const unsigned long long* seed_dev, void* stream);
int ac_embed_bwd(const float* dx, const int* word, float* demb, long rows, int d, float pa, unsigned long long seed_a,
                 float pb, unsigned long long seed_b, const unsigned long long* seed_dev, void* stream);
/* pre = res + dropout(x); y = LayerNorm(pre) over d = 256 columns (post-LN residual blocks of
 * nn.TransformerDecoderLayer; attn_proj's Dropout -> LayerNorm with res = NULL, transformer_decoder.py:38-43).
 * xmod > 0: x has only xmod rows and row r reads x[r % xmod] (the projected audio memory is shared by all passes, only
 * its dropout mask differs). */
int ac_dropadd_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* pre, float* y,
                      long row0, long rows, long xmod, int d, float p, unsigned long long seed,
                      const unsigned long long* seed_dev, float eps, void* stream);
/* Backward over rows 0..rows: dpre -> dres (= or += when accumulate), dpre * mask [* (relu_src > 0)] -> dx,
 * dgamma / dbeta accumulated atomically.  dx or dres may be NULL. */
int ac_dropadd_ln_bwd(const float* dy, const float* pre, const float* gamma, float* dx, float* dres, int accumulate,
                      const float* relu_src, long relu_mod, float* dgamma, float* dbeta, long rows, int d, float p,
                      unsigned long long seed, const unsigned long long* seed_dev, float eps, void* stream);
/* Multi-head attention over short sequences (nn.MultiheadAttention inside nn.TransformerDecoderLayer), head_dim 64,
 * one workgroup per (sequence, head).  Sequence s: qlen[s] queries at rows qrow0[s].., klen[s] keys at rows
 * krow0[s]...  Key j is visible to query i iff j < kvalid[s] (if given), j <= i (if causal) and
 * word[krow0[s] + j] != pad_idx (if word given).  P (softmax before dropout) is kept at
 * P[((s*nhead + h)*pl + i)*ptk + j]; that index also drives the attention dropout.  Launch covers sequences
 * seq0 .. seq0+nseq; lmax / tkmax bound qlen / klen over the launch. */
int ac_attn_seq_fwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o, long ldo,
                    float* P, int pl, int ptk, const int* qrow0, const int* qlen, const int* krow0, const int* klen,
                    const int* kvalid, const int* word, int pad_idx, int causal, int seq0, int nseq, int nhead,
                    int head_dim, int lmax, int tkmax, float drop_p, unsigned long long seed,
                    const unsigned long long* seed_dev, void* stream);
int ac_attn_seq_bwd(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, const float* P, int pl,
                    int ptk, const float* dout, long lddo, float* dq, long lddq, float* dk, long lddk, float* dv,
                    long lddv, const int* qrow0, const int* qlen, const int* krow0, const int* klen, int seq0, int nseq,
                    int nhead, int head_dim, int lmax, int tkmax, float drop_p, unsigned long long seed,
                    const unsigned long long* seed_dev, void* stream);
/* dst[i] = src[index[i]] / dst[index[i]] += src[i] over rows of C floats (classifier on each pass's last position,
 * base.py:181-183). */
int ac_gather_rows(const float* src, const int* index, float* dst, long nrows, int C, void* stream);
int ac_scatter_add_rows(const float* src, const int* index, float* dst, long nrows, int C, void* stream);
/* SpecAugment of the train-mode Cnn14 (torchlibrosa SpecAugmentation as the reference configures it,
 * cnn_encoder.py:352-354,423-425: 2 time stripes of width < 64 frames and 2 mel stripes of width < 8 bins per clip,
 * zeroed before bn0): x [B*rows_per_clip][F] is the bn0-normalised log-mel, stripes [B][n_time + n_freq][2] =
 * (begin, length) drawn on the host; a masked bin is set to fill[mel] = bn0(0) (0 when fill is NULL). */
int ac_specaug(float* x, const int* stripes, const float* fill, int B, int rows_per_clip, int T, int F, int n_time,
               int n_freq, void* stream);
/* out[r] = sum_t x[t*n + r], t < reps (gradient of the projected audio memory shared by all passes). */
int ac_sum_replicas(const float* x, float* out, long n, int reps, void* stream);
/* Cnn14 head when dropout sits between the last block and the mel mean (train mode, cnn_encoder.py:441-444):
 * out[b][h][c] = mean_w x[(b*Hp + h)][w][c] for h < H. */
int ac_rows_mean_w(const float* x, float* out, int B, int Hp, int H, int W, int C, void* stream);
/* dst[b][c][r] = src[b][r][c] (k-major copy of W_hh for the recurrence kernel). */
int ac_transpose(const float* src, float* dst, int batch, int rows, int cols, void* stream);
/* out[n] += sum_m x[m*ld + n] (bias gradients). */
int ac_colsum(const float* x, long ld, float* out, long M, int N, void* stream);
/* out[r*out_ld] = argmax_v logit[r*ld + v], first index on ties (sample_next_word "greedy", base.py:206-209). */
int ac_argmax_rows(const float* logit, long ld, int rows, int V, int* out, long out_ld, void* stream);
/* LabelSmoothingLoss (loss.py:51-74): logit [N][T][V], tgt int64 [N][tgt_ld], tgt_len int32 [N]; row_loss [N*T];
 * loss[0] = inv_count * sum(row_loss); dlogit (optional) = gscale [* gscale_dev[0]] * (softmax - q) on valid rows,
 * 0 elsewhere (gscale_dev: the upstream gradient of the loss when it lives on the device).  inv_count <= 0 /
 * gscale <= 0 mean "1 / sum_n min(tgt_len[n], T)", computed on the device (reduction "mean" inside a HIP graph). */
int ac_label_smoothing_loss(const float* logit, const long long* tgt, long tgt_ld, const int* tgt_len, int N, int T, int V,
                            float smoothing, float inv_count, float* row_loss, float* loss, float* dlogit, float gscale,
                            const float* gscale_dev, void* stream);
/* ac_gru_layer that also keeps (r, z, n, W_hn h + b_hn) per (clip, step, direction): save [B][T][2][4H]. */
int ac_gru_layer_train(const float* gx, const float* whhT, const float* bhh, const int* lens, float* out, float* save,
                       int B, int T, int hidden, void* stream);
/* Backward through time of one bidirectional layer (autograd of nn.GRU under pack_padded_sequence,
 * model_util.py:10-27): dout/out [B][T][2H], whh [2][3H][H] -> gate gradients dgx (input side), dgh (hidden side)
 * [B][T][2][3H] and the previous hidden state of every step hprev [B][T][2][H]; weight gradients are GEMMs on those. */
int ac_gru_layer_bwd(const float* dout, const float* out, const float* save, const float* whh, const int* lens,
                     float* dgx, float* dgh, float* hprev, int B, int T, int hidden, void* stream);
/* Optimiser on flat buffers (run.py:122-126, torch.optim.Adam with L2 weight decay, cnn14rnn_trm.yaml:42-46):
 * norm_state[0] += sum g^2;  ac_clip_coef: [1] = sqrt([0]) / grad_div (total norm of the averaged gradient),
 * [2] = min(1, max_norm / (norm + 1e-6)) / grad_div (max_norm <= 0: no clipping);  ac_scale_by_coef: x *= [2];
 * ac_adam_step: g' = g * [2] (norm_state may be NULL) + wd * p, then Adam's update for 1-based `step`. */
/* norm_state has AC_NORM_STATE_FLOATS = 1032 floats, zero-initialised: [0..3] as described here, the rest is scratch of
 * ac_grad_sumsq (per-workgroup partials summed in a fixed order, so that every rank of a data-parallel job gets the
 * same bits); [3] is set to 1 by ac_clip_coef when the gradient norm is not finite, and ac_adam_step
 * then leaves parameters and moments untouched (the reference's NaN-loss skip, run.py:123, without a host sync).
 * ac_swa_update: avg += (p - avg) / (n_averaged + 1) (AveragedModel.update_parameters, train_util.py:233-253;
 * n_averaged = 0 copies). */
int ac_swa_update(float* avg, const float* p, long n, int n_averaged, void* stream);
int ac_grad_sumsq(const float* g, long n, float* norm_state, void* stream);
int ac_clip_coef(float* norm_state, float max_norm, float grad_div, void* stream);
int ac_scale_by_coef(float* x, long n, const float* norm_state, void* stream);
int ac_adam_step(float* p, const float* g, float* m, float* v, long n, const float* norm_state, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, const int* step_dev, void* stream);
/* step_dev (may be NULL): the number of updates APPLIED so far, kept on the device; when given, the bias correction
 * uses *step_dev + 1 instead of `step`.  ac_adam_commit advances it after the ac_adam_step launches of one optimiser
 * step - unless norm_state[3] says the update was skipped (the reference does not call optimizer.step() then,
 * run.py:123, so its step count does not advance either). */
int ac_adam_commit(int* step_dev, const float* norm_state, void* stream);

/* ============================ EfficientNet-B2 encoder ============================
 * Replaces efficientnet_pytorch==0.7.1 EfficientNet.extract_features as the reference calls it (hf_wrapper.py:229-241,
 * cnn_encoder.py:798-839; construction restated in eff_latent_encoder.py:74-186).  PARITY UNPINNED: that package is
 * not vendored, the kernels follow its published algorithm.  Activations are channels-last [clip][time][mel][C] fp32
 * (time = the reference's W axis, mel = its H axis).  The 1x1 convolutions are ac_gemm (BatchNorm folded into the
 * weight rows, swish in the epilogue, the squeeze-excite gate as a_scale, the residual as beta = 1).               */

/* 1x1 convolution over channels-last rows: y[M][N] = act((x[M][K] .* gate[m / gate_rows][K]) w[N][K]^T + bias) + beta*y
 * (_expand_conv / _project_conv / _conv_head of efficientnet_pytorch with BatchNorm folded into w and bias; act 0 none,
 * 1 ReLU, 2 swish; gate = the squeeze-excite gate or NULL; beta = 1 adds the block input in place).  K % 8 == 0,
 * N % 4 == 0, 16-byte aligned pointers. */
int ac_pointwise_conv(const float* x, const float* w, const float* bias, float* y, long M, int N, int K, int act, float beta,
                      const float* gate, int gate_rows, void* stream);
/* AmplitudeToDB(top_db) on a (batch, mel, time) tensor: x = max(x, max(x over the WHOLE buffer) - top_db)
 * (hf_wrapper.py:279; torchaudio packs the batch axis as channels).  scratch: >= 1 float (<= 1024 used). */
int ac_top_db_clamp(float* x, long n, float top_db, float* scratch, int scratch_floats, void* stream);
/* Stem: 3x3 stride-2 conv of the 1-channel log-mel x [B][T][F] with w [C][3 mel][3 time], static "same" padding
 * (pad_before / pad_after zeros on both axes), BN scale/shift, swish -> y [B][To][Fo][C]. */
int ac_effnet_stem(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int T, int F,
                   int C, int pad_before, int pad_after, void* stream);
/* Depthwise k x k conv (k = 3 / 5, stride 1 / 2) + BN + swish: x [B][T][F][C], w [k time][k mel][C] ->
 * y [B][To][Fo][C], To = (T + pad_before + pad_after - k) / stride + 1; pool [B][C] += pool_scale * sum of y over
 * positions (the squeeze of the squeeze-excite layer: pool_scale = 1 / (To*Fo) gives the mean; zero it first). */
int ac_effnet_depthwise(const float* x, const float* w, const float* scale, const float* shift, float* y, float* pool,
                        float pool_scale, int B, int T, int F, int C, int k, int stride, int pad_before, int pad_after,
                        void* stream);
/* gate[b][c] = sigmoid(w2 swish(w1 (pool[b] * inv_count) + b1) + b2), w1 [S][C], w2 [C][S]. */
int ac_effnet_se_gate(const float* pool, float inv_count, const float* w1, const float* b1, const float* w2,
                      const float* b2, float* gate, int B, int C, int S, void* stream);
/* 1x1 convolution / linear layer with static weights on the split-bf16 matrix path (csrc/pw_gemm.hip), the same contract
 * as ac_pointwise_conv: y[M][N] = act((x[M][K] .* gate[m / gate_rows][K]) w^T + bias) + beta * y, for the matrix-bound
 * layers (thousands of rows against hundreds of channels).  The weights are split into bf16 hi + lo and laid out in MFMA
 * fragment order ONCE by ac_pw_gemm_pack (w [N][K] f32 -> wfrag, ac_pw_gemm_packed_bytes(N, K) bytes, 16-byte aligned);
 * activations are split when staged; three bf16 MFMAs per product, f32 accumulation (2^-16 relative operand error).
 * K % 4 == 0, N % 4 == 0, act 0 none / 1 ReLU / 2 swish, 16-byte aligned pointers. */
long ac_pw_gemm_packed_bytes(int N, int K);
int ac_pw_gemm_pack(const float* w, void* wfrag, int N, int K, void* stream);
int ac_pw_gemm_bf16x3(const float* x, const void* wfrag, const float* bias, float* y, long M, int N, int K, int act,
                      float beta, const float* gate, int gate_rows, void* stream);
/* The same with row strides (ldx, ldy in floats, multiples of 4) and inverted dropout on the output (mask = the counter
 * hash of ac_gemm: element index (row0 + m) * N + n, seed + (*seed_dev << 16)): the training step's x W^T (+ bias, ReLU,
 * dropout) and dy W (beta = 1) products (run.py:77-148 through transformer_model.py:20-32), whose weights change every
 * iteration - ac_pw_gemm_pack_strided packs W(n, k) = w[n * s_n + k * s_k] (W^T for the input-gradient product), and
 * ac_pw_gemm_pack_table repacks a whole device-resident table of layers in ONE launch: `count` records
 * {const float* w; void* wfrag; long s_n; long s_k; int N; int K;} (40 bytes each). */
int ac_pw_gemm_bf16x3_ex(const float* x, long ldx, const void* wfrag, const float* bias, float* y, long ldy, long M, int N,
                         int K, int act, float beta, const float* gate, int gate_rows, float drop_p,
                         unsigned long long drop_seed, const unsigned long long* seed_dev, long row0, void* stream);
int ac_pw_gemm_pack_strided(const float* w, long s_n, long s_k, void* wfrag, int N, int K, void* stream);
int ac_pw_gemm_pack_table(const void* table, int count, void* stream);
/* The same gate with the second matrix transposed, w2t [S][C] (= _se_expand.weight^T): one launch per block, both
 * phases read their weights with coalesced 16-byte loads.  C % 4 == 0, 16-byte aligned pointers. */
int ac_effnet_se_gate_t(const float* pool, float inv_count, const float* w1, const float* b1, const float* w2t,
                        const float* b2, float* gate, int B, int C, int S, void* stream);
/* MBConv head in one kernel (csrc/effnet_fused.hip): _expand_conv + _bn0 + swish -> _depthwise_conv + _bn1 + swish ->
 * squeeze sums, the expanded tensor staying in LDS.  x [B][T][F][Cin]; we [Cmid][Cin] / be [Cmid] = the expand
 * convolution with BatchNorm folded in (as for ac_pointwise_conv); wd [k time][k mel][Cmid], scale / shift [Cmid] = the
 * depthwise convolution and its folded BatchNorm (as for ac_effnet_depthwise); y [B][To][Fo][Cmid] (NULL: only the
 * squeeze sums); pool [B][Cmid] += pool_scale * sum of y over positions (zero it first).  Cin % 8 == 0, Cmid % 4 == 0,
 * k = 3 / 5, stride 1 / 2, 16-byte aligned pointers.  AC_ERR_ARG when one input row band does not fit the LDS budget
 * (the caller then runs the two-kernel chain). */
int ac_effnet_expand_depthwise(const float* x, const float* we, const float* be, const float* wd, const float* scale,
                               const float* shift, float* y, float* pool, float pool_scale, int B, int T, int F, int Cin,
                               int Cmid, int k, int stride, int pad_before, int pad_after, void* stream);

/* ===================================== waveform ingest =====================================
 * B clips stored back to back in src (float16 when src_half, else float32; clip b = src[src_off[b] .. src_off[b+1]))
 * -> out [B][lmax] float32: converted, resampled by the windowed-sinc polyphase filter of
 * torchaudio.functional.resample (call site caption_dataset.py:110-120; kernel [new][2*width + orig] with the non-zero
 * tap range [tap_lo, tap_hi) of each phase; orig / new are the gcd-reduced rates; orig == new: conversion only) and
 * zero-padded (WavPadCollate, inference.py:81-111).  out_len[b] = samples of the RESAMPLED clip; out_start[b] (may be NULL
 * = 0) = its first sample kept: output sample o is resampled sample o + out_start[b], zero where that is >= out_len[b] -
 * the random crop / zero pad to audio_duration of caption_dataset.py:121-129 (the caller draws the offsets).  The filter
 * bank is pinned by tests/golden/g13_resample.npz (an independent float64 evaluation of torchaudio 0.13.1's published
 * windowed-sinc prototype on the fine grid; torchaudio itself is not vendored). */
int ac_ingest_resample(const void* src, int src_half, const long* src_off, const float* kernel, const int* tap_lo,
                       const int* tap_hi, float* out, const int* out_len, const int* out_start, int B, int lmax, int orig,
                       int new_, int width, void* stream);

/* Diagnostic (bench.py, not the hot path): `blocks` workgroups of 4 waves each issue iters x 32 dependent-free
 * v_mfma_f32_32x32x16_bf16 (8 accumulators per wave, no memory traffic); out[blocks * 256] receives the accumulator sums.
 * FLOPs = blocks * 4 * iters * 32 * 32768.  Timed by the caller, it gives the matrix rate the part sustains at the clock it
 * holds under matrix load - the context of `roofline.frac`, whose denominator is the NOMINAL dense peak. */
int ac_mfma_bf16_probe(float* out, int blocks, int iters, void* stream);

/* Diagnostic: `blocks` workgroups of `threads` threads record where they ran - out[2 b] = XCC_ID register, out[2 b + 1] =
 * HW_ID register (cu_id bits 11:8, sh_id bit 12, se_id bits 15:13) - and stay resident for spin_ticks of the 100 MHz clock. */
int ac_placement_probe(int* out, int blocks, int threads, int spin_ticks, void* stream);

/* A HIP stream restricted to n_cus compute units (hipExtStreamCreateWithCUMask, mask bits first_cu .. first_cu + n_cus - 1).  The throughput
 * mode (TransformerModel.forward_async) runs its latency-bound decode chains on such a stream: their small workgroups are
 * packed onto a few CUs instead of each holding a CU that a one-workgroup-per-CU conv kernel of the encoder stream then
 * cannot use.  The only entry points of this library that create / destroy a driver object; the caller owns the stream. */
int ac_stream_create_cu_mask(int first_cu, int n_cus, void** out);
int ac_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUDIOCAPTION_HIP_H */
