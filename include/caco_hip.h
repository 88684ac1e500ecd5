/*
 * caco_hip.h  --  C ABI of libcaco_hip.so: the MI355X (gfx950) drop-in for the Cacophony
 * inference hot path  (waveform -> log-mel patches -> audio / text encoders -> similarity).
 *
 * The reference (gzhu06/Cacophony) has no FFI layer: its boundary for this path is a Python
 * nn.Module method API plus four pre-processing free functions.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference repo root).  The Python binding
 * a maintainer would add is cacophony_amd/_lib.py (ctypes); see INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every pointer whose name ends in _dev is DEVICE memory owned by the
 *     caller (the library never frees caller memory); `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  All work is enqueued on that stream; calls return without
 *     synchronising unless stated.
 *   - every function returns CACO_OK (0) or a negative status; caco_last_error() gives the
 *     message of the calling thread's last failure.  No C++ exceptions cross the boundary.
 *   - a caco_model owns its weights (bf16 GEMM operands, fp32 everything else) and one workspace arena per
 *     (tower, stream), grown on demand: forwards enqueued on DIFFERENT streams (the text tower next to the audio
 *     tower, or two half batches) share no scratch memory and may overlap on the GPU; calls that share a stream
 *     are ordered by it.  Host-side the model is not thread safe: issue all calls from one thread.
 *   - a caco_model lives on the device that was current at caco_create(); every entry point that takes a model returns
 *     CACO_ERR_STATE when another device is current (kernel attributes, CU counts and the mel tables are kept per device,
 *     so several models on several GPUs of one process are fine).
 *   - process-global state: the tuning knobs caco_set_gemm_tile / caco_set_switch / caco_set_ln_fold (the last only as the
 *     default of NEW models; caco_model_set_ln_fold acts on one model) and the caco_profile_* recorder (mutex-guarded).
 *   - row-major everywhere; Linear weights arrive in torch layout [out, in], fp32.
 */
#ifndef CACO_HIP_H
#define CACO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CACO_OK 0
#define CACO_ERR_INVALID (-1)   /* bad argument / shape / missing tensor */
#define CACO_ERR_HIP (-2)       /* a HIP runtime call failed             */
#define CACO_ERR_STATE (-3)     /* call order (e.g. forward before finalize) */

#define CACO_DTYPE_F32 0
#define CACO_DTYPE_BF16 1

typedef struct caco_model caco_model;

/* Hyper-parameter contract.  Mirrors AudioTransformerConfig (src/caco_torch/audio_models/mae.py:9-20),
 * RobertaConfig (src/caco_torch/text_models/roberta.py:11-23), CACOConfig (src/caco_torch/caco.py:17-21);
 * defaults of create_caco_model() (src/caco_torch/caco.py:264-317) via caco_default_config(). */
typedef struct caco_config {
  int32_t audio_hidden, audio_layers, audio_heads, audio_intermediate, patch_size, num_freq_patches;
  float audio_ln_eps;
  int32_t text_vocab, text_hidden, text_layers, text_heads, text_intermediate, text_max_pos, text_type_vocab;
  float text_ln_eps;
  int32_t projection_size, pool_heads;   /* pool_heads: 1 or an even number dividing audio_hidden.  The torch model pools with 2
                                            (caco.py:20,294), the JAX model with 8 (src/caco/load_model.py:46) on the same tensors */
  float logit_scale;
  int32_t has_audio, has_text;      /* which towers to allocate (AudioMAE-only models set has_text = 0) */
  int32_t mae_decoder_layers;       /* > 0: also hold an AudioDecoder (mae.py:151-207) of that depth     */
  int32_t caption_decoder_layers;   /* > 0: also hold the caption decoder (RobertaDecoder, text_models/roberta.py:329-373;
                                       create_caco_model uses 4, caco.py:297-309); widths = the text tower's */
} caco_config;

const char* caco_version(void);
/* sizeof(caco_config) as the library was built: a binding asserts it equals its own struct size before the first
 * caco_default_config() (which memsets that many bytes) - cacophony_amd/_lib.py load() does. */
int32_t caco_config_size(void);
const char* caco_last_error(void);
void caco_default_config(caco_config* cfg);

/* ---- model lifetime + weights: replaces create_caco_model()/load_state_dict
 *      (src/caco_torch/caco.py:264, src/eval/eval_caco_torch.py:154-169) ------------------------ */
int caco_create(const caco_config* cfg, caco_model** out);
void caco_destroy(caco_model* m);
/* Upload one tensor by its reference state-dict key ("audio_module.layers.3.mlp.fc1.weight", ...;
 * AudioMAE keys are "encoder.*" / "decoder.*").  host_f32 is HOST memory, fp32, contiguous.
 * "decoder_module.*" (the caption decoder) is loaded when the model was created with caption_decoder_layers > 0 and
 * accepted-and-ignored otherwise; any other unknown key, or a shape mismatch, is an error. */
int caco_load_tensor(caco_model* m, const char* name, const float* host_f32, const int64_t* shape, int32_t ndim);
/* Verify every required tensor arrived and build the packed bf16 operands.  Synchronises. */
int caco_finalize_weights(caco_model* m);
int caco_set_logit_scale(caco_model* m, float logit_scale);
float caco_get_logit_scale(const caco_model* m);

/* ---- front end --------------------------------------------------------------------------------
 * caco_mel_num_frames: ceil(n_samples / 160), the frame count of compute_mel_spectrogram
 * (src/eval/eval_caco_torch.py:66-72). */
int64_t caco_mel_num_frames(int64_t n_samples);
/* compute_mel_spectrogram (src/eval/eval_caco_torch.py:41-105), batched:
 * wav_dev fp32 [batch, n_samples] -> mel_dev fp32 [batch, frames, 128].
 * Fixed front-end geometry: sr 16000, hop 160, win 400 (periodic Hann, centred in the 512 frame),
 * n_fft 512, 128 HTK mels, magnitude spectrum, log(x + 1e-5) * scale + bias. */
int caco_mel_spectrogram(const float* wav_dev, int32_t batch, int64_t n_samples, float scale, float bias,
                         float* mel_dev, void* stream);
/* compute_mel_spectrogram + spectrogram_to_patches (src/eval/eval_caco_torch.py:41-151) fused,
 * = prepare_audio_batch (:181-206) without the host round trip:
 * wav_dev fp32 [batch, n_samples] -> patches_dev [batch, max_patches, 256] (patch_dtype f32 | bf16),
 * time_inds_dev / freq_inds_dev / mask_dev fp32 [batch, max_patches] (any of the three may be NULL). */
int caco_mel_patches(const float* wav_dev, int32_t batch, int64_t n_samples, int32_t max_patches, float scale,
                     float bias, void* patches_dev, int32_t patch_dtype, float* time_inds_dev, float* freq_inds_dev,
                     float* mask_dev, void* stream);
/* The same for a batch of clips of DIFFERENT lengths in rows of n_samples: lengths_dev int64 [batch] (NULL = all
 * n_samples).  Whatever a row holds past lengths[b] is ignored (read as the STFT's zero padding).  Clip b gets the patches / indices / mask the reference produces when prepare_audio_batch
 * (src/eval/eval_caco_torch.py:181-206) runs on its lengths[b] samples alone: (ceil(len / 160) / 16) * 8 valid patches,
 * the rest zero rows with mask 0. */
int caco_mel_patches_lens(const float* wav_dev, const int64_t* lengths_dev, int32_t batch, int64_t n_samples,
                          int32_t max_patches, float scale, float bias, void* patches_dev, int32_t patch_dtype,
                          float* time_inds_dev, float* freq_inds_dev, float* mask_dev, void* stream);

/* ---- encoders ---------------------------------------------------------------------------------
 * CACO.get_audio_embedding (src/caco_torch/caco.py:123-150):
 * patches [B,S,256] (f32 | bf16), time/freq inds fp32 [B,S], mask fp32 [B,S] (1 = keep)
 *   -> emb_dev fp32 [B, projection_size]; hidden_dev fp32 [B,S,hidden] or NULL. */
int caco_audio_forward(caco_model* m, const void* patches_dev, int32_t patch_dtype, const float* time_inds_dev,
                       const float* freq_inds_dev, const float* mask_dev, int32_t batch, int32_t seq,
                       int32_t normalize, float* emb_dev, float* hidden_dev, void* stream);
/* CACO.get_text_embedding (src/caco_torch/caco.py:152-177): ids int64 [B,T], mask int64 [B,T] (1 = keep),
 * position_ids int64 [B,T] or NULL (= arange(T), text_models/roberta.py:292-293)
 *   -> emb_dev fp32 [B, projection_size]; hidden_dev fp32 [B,T,hidden] or NULL. */
int caco_text_forward(caco_model* m, const int64_t* ids_dev, const int64_t* mask_dev, const int64_t* position_ids_dev,
                      int32_t batch, int32_t seq, int32_t normalize, float* emb_dev, float* hidden_dev, void* stream);
/* encode_audio of BASELINE.json north_star = mel patches (bf16, on device) + get_audio_embedding(normalize=True). */
int caco_encode_audio(caco_model* m, const float* wav_dev, int32_t batch, int64_t n_samples, int32_t max_patches,
                      float* emb_dev, void* stream);
/* ... with per-clip lengths (lengths_dev int64 [batch] or NULL, see caco_mel_patches_lens) and an output row stride
 * ld_emb (elements; 0 = projection_size): both towers can write straight into one packed [B, 2, P] exchange buffer
 * (the all-gather payload of the data-parallel path, src/eval/eval_caco.py:53-64). */
int caco_encode_audio_ex(caco_model* m, const float* wav_dev, const int64_t* lengths_dev, int32_t batch, int64_t n_samples,
                         int32_t max_patches, float* emb_dev, int32_t ld_emb, void* stream);
/* encode_text of north_star = get_text_embedding(normalize=True, position_ids=None) with an output row stride. */
int caco_encode_text(caco_model* m, const int64_t* ids_dev, const int64_t* mask_dev, int32_t batch, int32_t seq,
                     float* emb_dev, int32_t ld_emb, void* stream);

/* ---- scoring ----------------------------------------------------------------------------------
 * out[i, j] = scale * <a_i, t_j>: CACO.get_contrastive_logits' matmul (src/caco_torch/caco.py:208-210)
 * and the callers' similarity (src/eval/eval_caco_torch.py:330,398).  fp32 in, fp32 MFMA, fp32 out.
 * a_dev [na, dim], t_dev [nt, dim], out_dev [na, nt] with row stride ld_out (>= nt). */
int caco_similarity(const float* a_dev, int32_t na, const float* t_dev, int32_t nt, int32_t dim, float scale,
                    float* out_dev, int32_t ld_out, void* stream);
/* ... with row strides lda / ldt (elements, 0 = dim, multiples of 4) for banks that live interleaved in a packed buffer */
int caco_similarity_ld(const float* a_dev, int32_t na, int32_t lda, const float* t_dev, int32_t nt, int32_t ldt, int32_t dim,
                       float scale, float* out_dev, int32_t ld_out, void* stream);
/* Retrieval scoring, device part: the first k columns of argsort(-sim, dim=-1) per row
 * (src/eval/eval_caco_torch.py:402-408; src/eval/eval_utils.py:18-54 reads only the first 10).  sim_dev is read
 * as sim[r * row_stride + c * col_stride] (elements), so audio->text (rows = clips) and text->audio (rows = captions)
 * both run on one stored matrix.  idx_dev int32 [rows, k] (value descending, ties by ascending index, -1 padding when
 * cols < k); val_dev fp32 [rows, k] or NULL.  1 <= k <= 64. */
int caco_topk(const float* sim_dev, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride, int32_t k,
              int32_t* idx_dev, float* val_dev, void* stream);
/* HEAR "event" (timestamp) embeddings: mean over `group` consecutive tokens of the encoder's hidden states,
 * tf.nn.avg_pool(hidden, ksize=8, strides=8, padding='VALID') in
 * src/eval/heareval/embeddings/audio_embedding/caco_embeddings.py:118-124 (group = 8 = the frequency patches of one
 * 160 ms time step).  hidden_dev fp32 [batch, seq, dim] (the `hidden` output of caco_audio_forward) ->
 * out_dev fp32 [batch, seq / group, dim]; the trailing seq % group tokens are dropped. */
int caco_token_group_mean(const float* hidden_dev, int32_t batch, int32_t seq, int32_t dim, int32_t group, float* out_dev,
                          void* stream);
/* x / ||x + 1e-10||_2 per row (src/caco_torch/caco.py:144-146), in place allowed. */
int caco_l2_normalize(const float* x_dev, int32_t rows, int32_t dim, float* out_dev, void* stream);

/* ---- AudioMAE stage-1 forward: AudioMAE.forward (src/caco_torch/audio_models/mae.py:217-247) -------
 * visible patches [B,V,256] + restore index sets -> out_dev fp32 [B, V+R, patch_size]. */
int caco_mae_forward(caco_model* m, const void* patches_dev, int32_t patch_dtype, const float* mask_dev,
                     const float* time_inds_dev, const float* freq_inds_dev, const float* restore_time_inds_dev,
                     const float* restore_freq_inds_dev, const float* restore_mask_dev, int32_t batch,
                     int32_t n_visible, int32_t n_restore, float* out_dev, void* stream);

/* ---- caption decoder logits: CACO.get_decoder_logits (src/caco_torch/caco.py:212-240) ->
 * RobertaDecoder.forward (src/caco_torch/text_models/roberta.py:337-373).  Teacher-forced: every layer is
 * self-attention under the causal AND caption-padding mask, cross-attention over the audio tokens (padded audio
 * tokens masked), MLP, each followed by a post-LayerNorm; then decoder_proj to the vocabulary.
 * text_hidden_dev fp32 [B, T, H]  = the `hidden` output of caco_text_forward on the caption prefix (caco.py:226-230),
 * text_mask_dev int64 [B, T] (1 = keep), audio_hidden_dev fp32 [B, S, H] = the `hidden` output of caco_audio_forward,
 * audio_mask_dev fp32 [B, S] (1 = keep) -> logits_dev fp32 [B, T, vocab].  Needs caption_decoder_layers > 0 and the
 * `decoder_module.*` tensors; otherwise CACO_ERR_INVALID with "Decoder module not initialized" (caco.py:223-224). */
int caco_decoder_forward(caco_model* m, const float* text_hidden_dev, const int64_t* text_mask_dev,
                         const float* audio_hidden_dev, const float* audio_mask_dev, int32_t batch, int32_t seq_text,
                         int32_t seq_audio, float* logits_dev, void* stream);

/* ---- incremental caption decoding with key / value caches: the JAX path's get_next_decoder_logits loop
 * (src/caco/caco.py:154-230; the torch path re-runs the whole prefix every step, eval_caco_torch.py:443-456).
 * caco_decode_begin projects the audio hidden states to every decoder layer's cross-attention keys / values once and
 * allocates the self-attention caches of the 12 text layers and the decoder layers for captions of up to max_len tokens;
 * caco_decode_step feeds ONE token per clip (token_ids_dev int64 [B]: BOS first, then the token chosen from the previous
 * step's logits) and returns logits_dev fp32 [B, vocab] for the next position - equal, position by position, to
 * caco_text_forward + caco_decoder_forward on the prefix with an all-ones text mask; caco_decode_end frees the state.
 * The state owns its device memory; the model must outlive it; steps of one state must be issued in order on one stream. */
typedef struct caco_decode_state caco_decode_state;
int caco_decode_begin(caco_model* m, const float* audio_hidden_dev, const float* audio_mask_dev, int32_t batch,
                      int32_t seq_audio, int32_t max_len, caco_decode_state** out, void* stream);
int caco_decode_step(caco_decode_state* s, const int64_t* token_ids_dev, float* logits_dev, void* stream);
void caco_decode_end(caco_decode_state* s);

/* ---- introspection / measurement ---------------------------------------------------------------- */
int64_t caco_workspace_bytes(const caco_model* m);
/* Tuning knob (process-global): bf16 GEMM kernel choice.  256 (default) = per shape: the persistent 256x256 eight-wave
 * kernel (csrc/gemm_w8.hip) when every CU gets work, else the 256x128 two-workgroups-per-CU kernel (gemm_x.hip), else
 * 128x128; 128 = always 128x128.  Forced kernels for tests / A-B runs: 8256 = gemm_w8, 2256 = gemm_x, 4256 = gemm_w4q
 * (four waves of 128 x 128), 4128 = gemm_w4h (128 x 256 tiles for mid-size M): round-3 experiments, never the default.  A
 * forced kernel that does not support a shape or an epilogue falls back to the default choice.  Returns the mode now in
 * force; any other value only queries. */
int32_t caco_set_gemm_tile(int32_t tile);
/* Run-time switches of the A/B experiments (process-global diagnostics; csrc/kernels.h lists them with their defaults:
 * CACO_PINGPONG, CACO_POS_FUSE, CACO_POOL_FUSE, CACO_ATTN_SMALL, CACO_ATTN_ROWS, CACO_W_NGROUP, CACO_W8_MIN_TILES,
 * CACO_W4H_MAX_TILES).  A switch takes its initial value from the environment variable of the same name ONCE, at its first
 * use in the process; afterwards only caco_set_switch changes it (no launch path reads the environment).  Unknown name:
 * CACO_ERR_INVALID resp. INT32_MIN.  A value outside the switch's range (0 / 1 for the on-off switches, 32 / 64 for
 * CACO_ATTN_ROWS, -1 .. 4096 for CACO_W_NGROUP, >= 0 for the tile thresholds; never INT32_MIN) is CACO_ERR_INVALID and changes
 * nothing; the same ranges apply to the environment's initial value (out of range: the default, with a line on stderr).
 * The reference has no counterpart: its knobs are Python arguments. */
int caco_set_switch(const char* name, int32_t value);
int32_t caco_get_switch(const char* name);
/* Tuning knob: LayerNorm folding in the audio stack (the LayerNorm passes disappear into the neighbouring GEMM
 * epilogues; api.hip run_audio_layers).  0 = separate LayerNorm passes (default: measured slightly faster at batch
 * 256, see api.hip), 1 = always fold, -1 = fold when the batch fills the chip.  Env CACO_LN_FOLD sets the initial
 * mode.  Returns the mode now in force; any other value only queries. */
int32_t caco_set_ln_fold(int32_t mode);             /* the default of models created afterwards */
int32_t caco_model_set_ln_fold(caco_model* m, int32_t mode);   /* one model */
/* Per-stage timing.  While enabled, every forward records a hipEvent pair around each launch group on the
 * caller's stream.  caco_profile_report synchronises on them, writes a JSON object
 * {"audio.gemm_fc1": {"ms": total, "n": launches}, ...} into buf (truncated to buflen) and resets the
 * accumulators; it returns the number of bytes the full report needs.  Not thread safe. */
int caco_profile_enable(int32_t on);
int64_t caco_profile_report(char* buf, int64_t buflen);
/* Run one bf16 GEMM of the encoder's dominant shape class in isolation (bench.py roofline leg):
 * out[M,N] (bf16) = act(A[M,K] (bf16) x W[N,K]^T (bf16) + bias).  act: 0 none, 1 SiLU, 2 erf-GELU. */
int caco_op_gemm_bf16(const void* a_dev, const void* w_dev, const float* bias_dev, int64_t M, int32_t N, int32_t K,
                      int32_t act, void* out_dev, void* stream);
/* same with explicit row strides (elements) for A, W and out: padded leading dimensions */
int caco_op_gemm_bf16_strided(const void* a_dev, int32_t lda, const void* w_dev, int32_t ldw, const float* bias_dev, int64_t M,
                              int32_t N, int32_t K, int32_t act, void* out_dev, int32_t ldc, void* stream);
/* fp32 out[M,N] = A x W^T + bias + resid (resid may alias out, may be NULL) */
int caco_op_gemm_bf16_f32out(const void* a_dev, const void* w_dev, const float* bias_dev, const float* resid_dev,
                             int64_t M, int32_t N, int32_t K, float* out_dev, void* stream);
/* LayerNorm over the last axis of fp32 x[rows, dim]; out_f32_dev / out_bf16_dev may each be NULL. */
int caco_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, int64_t rows, int32_t dim,
                      float eps, float* out_f32_dev, void* out_bf16_dev, void* stream);
/* Fused softmax(QK^T * scale + mask) V for the encoder layout: qkv_dev bf16 [B*S, ld] with Q at column 0, K at
 * column k_off, V at column v_off (head h = columns h*head_dim.. of each), key_mask_dev fp32 [B,S] or NULL,
 * out_dev bf16 [B*S, heads*head_dim].  head_dim in {64, 96}; ld, k_off, v_off multiples of 8. */
int caco_op_attention(const void* qkv_dev, int32_t ld, int32_t k_off, int32_t v_off, const float* key_mask_dev,
                      int32_t batch, int32_t seq, int32_t heads, int32_t head_dim, int32_t causal, void* out_dev,
                      void* stream);
/* The same kernel with separate operands (cross-attention, RobertaSelfAttention with key_value_states,
 * roberta.py:67-104): queries q_dev bf16 [B*seq_q, q_ld] (head h at column h*head_dim), keys / values kv_dev bf16
 * [B*seq, ld] at columns k_off / v_off, key_mask_dev fp32 [B, seq] or NULL -> out_dev bf16 [B*seq_q, heads*head_dim].
 * causal needs seq_q == seq. */
int caco_op_attention_qkv(const void* q_dev, int32_t q_ld, int32_t seq_q, const void* kv_dev, int32_t ld, int32_t k_off,
                          int32_t v_off, const float* key_mask_dev, int32_t batch, int32_t seq, int32_t heads,
                          int32_t head_dim, int32_t causal, void* out_dev, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* CACO_HIP_H */
