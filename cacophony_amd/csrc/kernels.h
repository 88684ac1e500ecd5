// Internal launcher interface between the kernel translation units and the C-ABI layer (api.hip).
#pragma once
#include "common.h"
#include "../../include/caco_hip.h"

namespace caco {

enum { EPI_BF16 = 0, EPI_F32 = 1 };
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2 };

struct GemmArgs {
  const bf16_t* A;      // [M, K]
  const bf16_t* W;      // [N, K]
  const float* bias;    // [N] or null
  const float* resid;   // EPI_F32: [M, ldc] fp32 or null (may alias out)
  void* out;            // EPI_BF16: bf16 [M, ldc]; EPI_F32: fp32 [M, ldc]
  int64_t M;
  int N, K, ldc;
  int lda, ldw;         // row strides of A / W in elements; 0 = K (dense)
  int ngroup;           // gemm_x: n-tiles per L2 group (0 = all)
  int reverse;          // persistent kernels: walk the tile list last to first (ping-pong traversal of consecutive kernels,
                        // api.hip run_audio_layers); the launcher folds it into the sign of ngroup, the kernels read only that
  // LayerNorm folding (8-wave kernel of gemm_w4.hip only; all null = plain GEMM).
  // consumer, EPI_BF16:  out = act( rstd[m] * (acc[m,n] - mean[m] * fold_c1[n]) + bias[n] )
  //   with A = the RAW (un-normalised) rows in bf16, W = gamma-scaled weights, bias = W.beta + b  (see api.hip)
  const float* fold_mr;   // [M] (mean, rstd) pairs
  const float* fold_c1;   // [N] row sums of the gamma-scaled bf16 weights
  // producer, EPI_F32: besides out (fp32) also a bf16 copy of the same rows and per-row partial statistics
  bf16_t* xb_out;         // [M, ldc] bf16 copy of out, or null
  float* stats_part;      // [M, N/64] (sum, sum of squares) pairs over each wave's 64 columns, or null
  // gathered residual (8-wave kernel, EPI_F32): row m adds resid[resid_idx[m], :] - `resid` is then a small TABLE of rows
  // (row stride ldc) instead of an [M, ldc] matrix; an index < 0 adds nothing.  The patch-embed GEMM adds the positional
  // embedding this way (api.hip).  null = `resid` is indexed by the output row as usual.
  const int* resid_idx;
};

// Per-device launch preparation shared by every kernel that needs more than 64 KiB of dynamic LDS or sizes its grid by the
// CU count: sets hipFuncAttributeMaxDynamicSharedMemorySize for `kern` on the CURRENT device (once per kernel and device) and
// returns that device's CU count.  (Round 1 kept this in function-local statics: a second model on another GPU would have
// launched without the attribute and with the first device's CU count.)
constexpr int CACO_MAX_DEVICES = 16;
int prepare_launch(const void* kern, int dyn_lds_bytes, int* num_cu);      // api.hip
#define CACO_TRY_RC(expr)    \
  do {                       \
    int _rc = (expr);        \
    if (_rc) return _rc;     \
  } while (0)

// Run-time switches (A/B experiments and diagnostics; api.hip).  Each takes its initial value from the environment variable of
// the same name ONCE, at first use in the process; afterwards only caco_set_switch (include/caco_hip.h) changes it.  No launch
// path calls getenv.
enum Switch {
  SW_PINGPONG,        // CACO_PINGPONG       0     consecutive kernels of an audio layer walk the rows in opposite directions
  SW_POS_FUSE,        // CACO_POS_FUSE       0     positional embedding inside the patch-embed GEMM's epilogue
  SW_POOL_FUSE,       // CACO_POOL_FUSE      0     final LayerNorm of encode_audio inside the pooling kernel
  SW_ATTN_SMALL,      // CACO_ATTN_SMALL     0     one-wave attention kernel for sequences <= 64 (the text tower)
  SW_ATTN_ROWS,       // CACO_ATTN_ROWS      64    32 = one query block per wave at every sequence length
  SW_W_NGROUP,        // CACO_W_NGROUP       -1    n-tiles per L2 group of the persistent GEMM (-1 = by shape, 0 = one group)
  SW_W8_MIN_TILES,    // CACO_W8_MIN_TILES   128   256 x 128 tile units from which the persistent GEMM is the default
  SW_W4H_MAX_TILES,   // CACO_W4H_MAX_TILES  0     shapes below this many 256 x 256 tiles take 128 x 256 tiles (gemm_w4h.hip)
  SW_COUNT
};
int sw(Switch s);

int gemm_tile_config();
int set_gemm_tile_config(int tile);
int gemm_bf16(const GemmArgs& p, int epi, int act, hipStream_t st);
bool gemm_bf16_picks_w8(const GemmArgs& p, int epi);                      // gemm.hip: would gemm_bf16 send this shape to w8?
int gemm_bf16_x(const GemmArgs& p, int epi, int act, hipStream_t st);   // gemm_x.hip
bool gemm_bf16_w8_ok(const GemmArgs& p, int epi);                         // gemm_w8.hip: persistent 256x256, the default
int gemm_bf16_w8(const GemmArgs& p, int epi, int act, hipStream_t st);
int gemm_bf16_w4q(const GemmArgs& p, int epi, int act, hipStream_t st);
int gemm_bf16_w4h(const GemmArgs& p, int epi, int act, hipStream_t st);   // gemm_w4h.hip: 128 x 256 tiles, four waves (mid-M experiment, tile code 4128)   // gemm_w4q.hip: four waves of 128 x 128 (experiment, tile code 4256)
int gemm_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int ldc, float scale,
             hipStream_t st, int lda = 0, int ldb = 0);     // lda / ldb: row strides of A / B in elements (0 = K)

// norm.hip
// order: 0 = rows in block order; 1 / 2 = the rows as 8 contiguous ranges (workgroup b serves range b % 8: the ranges the
// persistent GEMMs' XCDs own), each walked first to last (1) or last to first (2) - ping-pong traversal, api.hip
int layernorm(const float* x, const float* gamma, const float* beta, int64_t rows, int dim, float eps, float* out_f32,
              bf16_t* out_bf16, hipStream_t st, int order = 0);
int l2_normalize(const float* x, int rows, int dim, float* out, hipStream_t st, int ld_out = 0);   // ld_out: row stride of out (0 = dim)
int text_embed_ln(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type0,
                  const float* gamma, const float* beta, int64_t rows, int seq, int dim, int vocab, int max_pos,
                  float eps, float* out_f32, bf16_t* out_bf16, hipStream_t st, int pos_base = 0);   // position = row % seq + pos_base
// x[m, :] (+)= sincos(time[m]) + freq_table[freq[m], :] (+ base[:] when base != null, replacing x)
int add_pos_embed(float* x, const float* base, const float* time_inds, const float* freq_inds, const float* freq_table,
                  int64_t rows, int dim, int num_freq, hipStream_t st);
// positional embedding as a gathered residual of the patch-embed GEMM (norm.hip): table [tmax * num_freq, dim] and per-row
// table indices (-1 = not an integer position: that row is finished by add_pos_embed_rest afterwards)
int pos_prepare(const float* freq_table, int tmax, int num_freq, int dim, float* table, const float* time_inds,
                const float* freq_inds, int64_t rows, int* idx, hipStream_t st);
int add_pos_embed_rest(float* x, const float* time_inds, const float* freq_inds, const float* freq_table, const int* idx,
                       int64_t rows, int dim, int num_freq, hipStream_t st);
// LayerNorm folding helpers: per-row (mean, rstd) from the GEMM epilogue's partial sums / from fp32 rows (+ bf16 copy)
int ln_stats_finalize(const float* part, int nslot, int64_t rows, int dim, float eps, float* mr, hipStream_t st);
int row_stats_bf16(const float* x, int64_t rows, int dim, float eps, bf16_t* xb, float* mr, hipStream_t st);
int mask_i64_to_f32(const int64_t* in, float* out, int64_t n, hipStream_t st);
int cast_f32_to_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t st);
// dst[b, dst_off + s, :] = src[b, s, :]  (fp32 rows of `dim`; used to concatenate decoder tokens)
int copy_rows(const float* src, float* dst, int batch, int src_seq, int dst_seq, int dst_off, int dim, hipStream_t st);

// attention.hip
// qkv: bf16 [batch*seq, ld], Q at column 0, K at k_off, V at v_off (head h = columns h*head_dim.. of each)
// order: 0 = clip b on XCD b % 8 (default); 1 / 2 = XCD x serves clips [x * B/8, (x + 1) * B/8), first to last / last to first.
// A non-zero order is honoured ONLY by the non-causal head_dim 96 kernels (the audio tower, where the ping-pong traversal of
// api.hip asks for it); every other instantiation is compiled without the argument's use and keeps order 0.
int attention(const bf16_t* qkv, int ld, int k_off, int v_off, const float* key_mask, int batch, int seq, int heads,
              int head_dim, int causal, bf16_t* out, hipStream_t st, int order = 0);
// kv_batch_rows: rows between the first keys of two consecutive clips in `kv` (0 = seq; larger for a KV cache)
int attention_qkv(const bf16_t* q, int q_ld, int seq_q, const bf16_t* kv, int ld, int k_off, int v_off, const float* key_mask,
                  int batch, int seq, int heads, int head_dim, int causal, bf16_t* out, hipStream_t st, int kv_batch_rows = 0,
                  int order = 0);

// attention_small.hip: one wave per (clip, head, 32-query block) for seq_q, seq <= 64 at head_dim 64 (the text tower at T = 32)
bool attention_small_ok(int seq_q, int seq, int head_dim);
int attention_small(const bf16_t* q, int q_ld, int seq_q, const bf16_t* kv, int ld, int k_off, int v_off, const float* key_mask,
                    int batch, int seq, int heads, int causal, bf16_t* out, hipStream_t st, int kv_batch_rows);

// pool.hip: learned-query attention pooling over the encoder output rows themselves (projections folded out):
// x bf16 [B, S, H], wq fp32 [heads, H] -> out fp32 [B, heads, H]
int attn_pool_rows(const bf16_t* x, const float* wq, const float* mask, int batch, int seq, int hidden, int heads, float* out,
                   hipStream_t st);

// ... with the final LayerNorm applied on the way in (x = the fp32 residual rows; nothing normalised is written)
int attn_pool_rows_ln(const float* x, const float* gamma, const float* beta, float eps, const float* wq, const float* mask, int batch,
                      int seq, int hidden, int heads, float* out, hipStream_t st);

// topk.hip: per-row top-k (value desc, index asc) through arbitrary strides
int token_group_mean(const float* x, int batch, int seq, int hidden, int group, float* out, hipStream_t st);
int topk_rows(const float* sim, int rows, int cols, int64_t row_stride, int64_t col_stride, int k, int* idx, float* val,
              hipStream_t st);

// mel.hip
// lengths: int64 [batch] valid samples per clip (device), or null = every clip is n_samples long
int mel_frontend(const float* wav, int batch, int64_t n_samples, int max_patches, float scale, float bias,
                 void* out, int mode, float* tinds, float* finds, float* mask, hipStream_t st, const int64_t* lengths = nullptr);
enum { MEL_NATURAL_F32 = 0, MEL_PATCH_F32 = 1, MEL_PATCH_BF16 = 2 };

}  // namespace caco
