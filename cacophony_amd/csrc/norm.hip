// Row-wise wavefront-reduced kernels: LayerNorm, RoBERTa embedding gather + LayerNorm,
// sin-cos / learned positional embedding add, L2 normalisation, small casts.
// One 64-lane wave owns one row; statistics stay in fp32 registers; every global access is a
// 16-byte float4 (or 8-byte bf16x4) per lane.  These are HBM-bound passes.
//
// Reference ops replaced: nn.LayerNorm (audio_models/mae.py:68,76,123; text_models/roberta.py:32,111,165),
// RobertaEmbeddings.forward (roberta.py:35-53), get_sin_cos_pos_embed + freq gather
// (mae.py:102-109,135-142), x / ||x + 1e-10|| (caco.py:144-146,171-173).
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int MAXC = 4;  // float4 chunks per lane -> dim <= 1024

__device__ __forceinline__ void ln_row(f32x4 (&v)[MAXC], int nchunk, int lane, int dim, const float* gamma,
                                       const float* beta, float eps, float* of, bf16_t* ob) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c * 64 + lane < nchunk) s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c * 64 + lane < nchunk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[c][r] -= mean;
        q += v[c][r] * v[c][r];
      }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nchunk) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + ch * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + ch * 4);
      f32x4 y;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = v[c][r] * rstd * g[r] + b[r];
      if (of) *reinterpret_cast<f32x4*>(of + ch * 4) = y;
      if (ob) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)y[r];
#ifdef LN_ST_NT       // A/B switch (tools/build_variant.sh): streaming stores of the bf16 rows
        __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(ob + ch * 4));
#else
        *reinterpret_cast<bf16x4*>(ob + ch * 4) = o;
#endif
      }
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t rows, int dim, float eps,
                                                        float* __restrict__ of, bf16_t* __restrict__ ob) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = dim >> 2;
  f32x4 v[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c * 64 + lane < nchunk) {
      const f32x4* src = reinterpret_cast<const f32x4*>(x + row * dim + (c * 64 + lane) * 4);
      v[c] = NT ? __builtin_nontemporal_load(src) : *src;
    }
  ln_row(v, nchunk, lane, dim, gamma, beta, eps, of ? of + row * dim : nullptr, ob ? ob + row * dim : nullptr);
}

// The same rows in the order the persistent GEMMs' XCDs own them (ping-pong traversal, api.hip): the row blocks form 8
// contiguous ranges, workgroup b serves range b % 8 (the hardware deals consecutive workgroups to the 8 XCDs round-robin) and
// walks it first to last or last to first (`desc`).  What a kernel wrote last in each range is what the next one reads first.
template <bool NT>
__global__ __launch_bounds__(256) void layernorm_ranges_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int64_t rows, int dim, float eps,
                                                               float* __restrict__ of, bf16_t* __restrict__ ob, int per_range,
                                                               int desc) {
  const int lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int64_t blk = (int64_t)xcd * per_range + (desc ? per_range - 1 - slot : slot);
  const int64_t row = blk * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = dim >> 2;
  f32x4 v[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c * 64 + lane < nchunk) {
      const f32x4* src = reinterpret_cast<const f32x4*>(x + row * dim + (c * 64 + lane) * 4);
      v[c] = NT ? __builtin_nontemporal_load(src) : *src;
    }
  ln_row(v, nchunk, lane, dim, gamma, beta, eps, of ? of + row * dim : nullptr, ob ? ob + row * dim : nullptr);
}

// A/B variant (-DLN_TWO_ROWS, tools/build_variant.sh; round 3, untimed): one wave owns TWO consecutive rows - twice the loads
// in flight per wave (6 x 16 bytes instead of 3 at dim 768), half the waves, the two rows' reductions interleaved.  The
// default kernel above reaches 0.70 of the HBM roofline; this asks whether latency per wave is what is left.
template <bool NT>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int64_t rows, int dim, float eps,
                                                         float* __restrict__ of, bf16_t* __restrict__ ob) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  if (row0 >= rows) return;
  const bool two = row0 + 1 < rows;                      // wave-uniform
  const int nchunk = dim >> 2;
  f32x4 v[2][MAXC];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c * 64 + lane < nchunk && (r == 0 || two)) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x + (row0 + r) * dim + (c * 64 + lane) * 4);
        v[r][c] = NT ? __builtin_nontemporal_load(src) : *src;
      }
  ln_row(v[0], nchunk, lane, dim, gamma, beta, eps, of ? of + row0 * dim : nullptr, ob ? ob + row0 * dim : nullptr);
  if (two) ln_row(v[1], nchunk, lane, dim, gamma, beta, eps, of ? of + (row0 + 1) * dim : nullptr, ob ? ob + (row0 + 1) * dim : nullptr);
}

// LayerNorm folding helpers (kernels.h GemmArgs::fold_*): (mean, rstd) per row
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ part, int nslot, int64_t rows,
                                                                float inv_dim, float eps, float* __restrict__ mr) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float2* p = reinterpret_cast<const float2*>(part) + row * nslot;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < nslot; ++i) {
    const float2 v = p[i];
    s1 += v.x;
    s2 += v.y;
  }
  const float mean = s1 * inv_dim;
  const float var = fmaxf(s2 * inv_dim - mean * mean, 0.f);
  reinterpret_cast<float2*>(mr)[row] = make_float2(mean, rsqrtf(var + eps));
}

__global__ __launch_bounds__(256) void row_stats_bf16_kernel(const float* __restrict__ x, int64_t rows, int dim, float eps,
                                                             bf16_t* __restrict__ xb, float* __restrict__ mr) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = dim >> 2;
  float s = 0.f;
  f32x4 v[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nchunk) {
      v[c] = *reinterpret_cast<const f32x4*>(x + row * dim + ch * 4);
      s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[c][r];
      *reinterpret_cast<bf16x4*>(xb + row * dim + ch * 4) = o;
    }
  }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c * 64 + lane < nchunk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) q += (v[c][r] - mean) * (v[c][r] - mean);
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)dim + eps);
  if (lane == 0) reinterpret_cast<float2*>(mr)[row] = make_float2(mean, rstd);
}

__global__ __launch_bounds__(256) void text_embed_ln_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                            const float* __restrict__ word, const float* __restrict__ pos,
                                                            const float* __restrict__ type0, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int64_t rows, int seq, int dim,
                                                            int vocab, int max_pos, float eps, float* __restrict__ of,
                                                            bf16_t* __restrict__ ob, int pos_base) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  int64_t id = ids[row];
  int64_t pi = pos_ids ? pos_ids[row] : (row % seq) + pos_base;
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);        // host validates; clamp keeps a bad id from faulting
  pi = pi < 0 ? 0 : (pi >= max_pos ? max_pos - 1 : pi);
  const int nchunk = dim >> 2;
  f32x4 v[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nchunk) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(word + id * dim + ch * 4);
      const f32x4 p = *reinterpret_cast<const f32x4*>(pos + pi * dim + ch * 4);
      const f32x4 t = *reinterpret_cast<const f32x4*>(type0 + ch * 4);
      v[c] = (w + p) + t;   // same association as roberta.py:49
    }
  }
  ln_row(v, nchunk, lane, dim, gamma, beta, eps, of ? of + row * dim : nullptr, ob ? ob + row * dim : nullptr);
}

__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, int rows, int dim,
                                                           float* __restrict__ out, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) {
    const float y = x[(int64_t)row * dim + i] + 1e-10f;     // eps joins the vector, not the norm (caco.py:146)
    s += y * y;
  }
  const float inv = 1.0f / sqrtf(wave_sum(s));
  for (int i = lane; i < dim; i += 64) out[(int64_t)row * ld_out + i] = x[(int64_t)row * dim + i] * inv;
}

// x[m, n] = (base ? base[n] : x[m, n]) + sincos(time[m])[n] + freq_table[freq[m]][n]
// sincos: n < dim/2 -> sin(t * w_n), else cos(t * w_{n - dim/2}); w_i = exp(2 i * (-ln 1e4) / dim)  (mae.py:102-109)
__device__ __forceinline__ void pos_embed_row(float* __restrict__ xrow, const float* __restrict__ base, float t, int f,
                                              const float* __restrict__ freq_table, int dim, int lane) {
  const int half = dim >> 1;
  const float kf = -9.210340371976184f / (float)dim;  // -ln(10000) / dim
  for (int ch = lane; ch < (dim >> 2); ch += 64) {
    f32x4 v = base ? *reinterpret_cast<const f32x4*>(base + ch * 4) : *reinterpret_cast<const f32x4*>(xrow + ch * 4);
    const f32x4 fe = *reinterpret_cast<const f32x4*>(freq_table + (int64_t)f * dim + ch * 4);
    f32x4 te;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = ch * 4 + r;
      const int i = n < half ? n : n - half;
      const float ang = t * expf((2.0f * (float)i) * kf);
      te[r] = n < half ? sinf(ang) : cosf(ang);
    }
    v = (v + te) + fe;   // x + time_pos_emb, then + freq_pos_emb (mae.py:141-142)
    *reinterpret_cast<f32x4*>(xrow + ch * 4) = v;
  }
}

__global__ __launch_bounds__(256) void add_pos_embed_kernel(float* __restrict__ x, const float* __restrict__ base,
                                                            const float* __restrict__ time_inds,
                                                            const float* __restrict__ freq_inds,
                                                            const float* __restrict__ freq_table, int64_t rows, int dim,
                                                            int num_freq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float t = time_inds[row];
  int f = (int)freq_inds[row];     // .long() truncation, mae.py:139
  f = f < 0 ? 0 : (f >= num_freq ? num_freq - 1 : f);
  pos_embed_row(x + row * dim, base, t, f, freq_table, dim, lane);
}

// ---- positional embedding through the patch-embed GEMM's epilogue (api.hip, GemmArgs::resid_idx) ------------------------
// The indices a front end produces are small integers (time patch 0 .. S/8, frequency patch 0 .. 7: eval_caco_torch.py
// :139-144), so the [M, dim] embedding is a gather from a table of tmax * num_freq distinct rows:
//   table[t * num_freq + f, :] = (0 + sincos(t)) + freq_table[f]        (the same expressions as pos_embed_row)
//   idx[m] = t * num_freq + f   when time_inds[m] is an integer in [0, tmax);  -1 otherwise
// Rows with idx < 0 (a caller is free to pass any float: get_sin_cos_pos_embed takes float positions, mae.py:102-109) get
// nothing from the GEMM and the exact per-row form afterwards (add_pos_embed_rest_kernel), so the result does not depend on
// the indices being integers - only the speed does.  One launch: blocks [0, table_blocks) build the table, the rest the indices.
__global__ __launch_bounds__(256) void pos_prepare_kernel(const float* __restrict__ freq_table, int tmax, int num_freq, int dim,
                                                          float* __restrict__ table, int table_blocks,
                                                          const float* __restrict__ time_inds, const float* __restrict__ freq_inds,
                                                          int64_t rows, int* __restrict__ idx) {
  if ((int)blockIdx.x < table_blocks) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= tmax * num_freq) return;
    float* trow = table + (int64_t)r * dim;
    for (int ch = lane; ch < (dim >> 2); ch += 64) *reinterpret_cast<f32x4*>(trow + ch * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    pos_embed_row(trow, nullptr, (float)(r / num_freq), r % num_freq, freq_table, dim, lane);   // each lane re-reads its own zeros
    return;
  }
  const int64_t m = (int64_t)(blockIdx.x - table_blocks) * 256 + threadIdx.x;
  if (m >= rows) return;
  const float t = time_inds[m];
  const float ff = freq_inds[m];
  const int f = !(ff >= 0.f) ? 0 : (ff >= (float)num_freq ? num_freq - 1 : (int)ff);   // clamp before the conversion (NaN -> 0)
  int id = -1;
  if (t >= 0.f && t < (float)tmax) {           // the conversion only for a value known to be in range (a NaN fails both tests)
    const int ti = (int)t;
    if ((float)ti == t) id = ti * num_freq + f;
  }
  idx[m] = id;
}

__global__ __launch_bounds__(256) void add_pos_embed_rest_kernel(float* __restrict__ x, const float* __restrict__ time_inds,
                                                                 const float* __restrict__ freq_inds,
                                                                 const float* __restrict__ freq_table, const int* __restrict__ idx,
                                                                 int64_t rows, int dim, int num_freq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows || idx[row] >= 0) return;            // the GEMM epilogue already added this row's embedding
  const float t = time_inds[row];
  int f = (int)freq_inds[row];
  f = f < 0 ? 0 : (f >= num_freq ? num_freq - 1 : f);
  pos_embed_row(x + row * dim, nullptr, t, f, freq_table, dim, lane);
}

__global__ void mask_i64_to_f32_kernel(const int64_t* __restrict__ in, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] != 0 ? 1.0f : 0.0f;
}

__global__ void cast_f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + i * 4);
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
    *reinterpret_cast<bf16x4*>(out + i * 4) = o;
  }
}

__global__ void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int src_seq, int dst_seq,
                                 int dst_off, int dim4, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % dim4);
  const int64_t r = i / dim4;
  const int s = (int)(r % src_seq);
  const int64_t b = r / src_seq;
  reinterpret_cast<f32x4*>(dst)[(b * dst_seq + dst_off + s) * dim4 + c] = reinterpret_cast<const f32x4*>(src)[i];
}

}  // namespace

int layernorm(const float* x, const float* gamma, const float* beta, int64_t rows, int dim, float eps, float* out_f32,
              bf16_t* out_bf16, hipStream_t st, int order) {
  CACO_REQUIRE(dim % 4 == 0 && dim > 0 && dim <= 256 * MAXC, "layernorm: dim %d must be a multiple of 4, <= %d", dim, 256 * MAXC);
  CACO_REQUIRE(rows > 0 && x && gamma && beta && (out_f32 || out_bf16), "layernorm: bad arguments");
  // streaming (nt) reads of the fp32 rows when only the bf16 copy is produced (pre-LN stacks): x is not needed again
  // before the next GEMM rewrites it, and the bf16 rows this kernel writes are what should stay cached
  constexpr int nt_env = 1;                   // measured -1.1 % per step (round 1)
  if (order == 1 || order == 2) {
    const int64_t nblk = (rows + 3) / 4;
    const int per_range = (int)((nblk + 7) / 8);
    const dim3 grid((unsigned)(per_range * 8));
    if (nt_env && !out_f32)
      hipLaunchKernelGGL(layernorm_ranges_kernel<true>, grid, dim3(256), 0, st, x, gamma, beta, rows, dim, eps, out_f32, out_bf16, per_range, order == 2);
    else
      hipLaunchKernelGGL(layernorm_ranges_kernel<false>, grid, dim3(256), 0, st, x, gamma, beta, rows, dim, eps, out_f32, out_bf16, per_range, order == 2);
    return check_hip(hipGetLastError(), "layernorm launch");
  }
#ifdef LN_TWO_ROWS
  if (nt_env && !out_f32)
    hipLaunchKernelGGL(layernorm2_kernel<true>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, x, gamma, beta, rows, dim, eps, out_f32, out_bf16);
  else
    hipLaunchKernelGGL(layernorm2_kernel<false>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, x, gamma, beta, rows, dim, eps, out_f32, out_bf16);
  return check_hip(hipGetLastError(), "layernorm launch");
#endif
  if (nt_env && !out_f32)
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, gamma, beta, rows, dim, eps,
                       out_f32, out_bf16);
  else
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, gamma, beta, rows, dim, eps,
                       out_f32, out_bf16);
  return check_hip(hipGetLastError(), "layernorm launch");
}

int ln_stats_finalize(const float* part, int nslot, int64_t rows, int dim, float eps, float* mr, hipStream_t st) {
  CACO_REQUIRE(part && mr && rows > 0 && nslot > 0 && dim > 0, "ln_stats_finalize: bad arguments");
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, part, nslot, rows,
                     1.0f / (float)dim, eps, mr);
  return check_hip(hipGetLastError(), "ln_stats_finalize launch");
}

int row_stats_bf16(const float* x, int64_t rows, int dim, float eps, bf16_t* xb, float* mr, hipStream_t st) {
  CACO_REQUIRE(dim % 4 == 0 && dim > 0 && dim <= 256 * MAXC, "row_stats_bf16: dim %d must be a multiple of 4, <= %d", dim, 256 * MAXC);
  CACO_REQUIRE(rows > 0 && x && xb && mr, "row_stats_bf16: bad arguments");
  hipLaunchKernelGGL(row_stats_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, rows, dim, eps, xb, mr);
  return check_hip(hipGetLastError(), "row_stats_bf16 launch");
}

int text_embed_ln(const int64_t* ids, const int64_t* pos_ids, const float* word, const float* pos, const float* type0,
                  const float* gamma, const float* beta, int64_t rows, int seq, int dim, int vocab, int max_pos,
                  float eps, float* out_f32, bf16_t* out_bf16, hipStream_t st, int pos_base) {
  CACO_REQUIRE(dim % 4 == 0 && dim <= 256 * MAXC, "text_embed_ln: unsupported dim %d", dim);
  hipLaunchKernelGGL(text_embed_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, ids, pos_ids, word, pos,
                     type0, gamma, beta, rows, seq, dim, vocab, max_pos, eps, out_f32, out_bf16, pos_base);
  return check_hip(hipGetLastError(), "text_embed_ln launch");
}

int l2_normalize(const float* x, int rows, int dim, float* out, hipStream_t st, int ld_out) {
  CACO_REQUIRE(rows > 0 && dim > 0 && x && out && (ld_out == 0 || ld_out >= dim), "l2_normalize: bad arguments");
  hipLaunchKernelGGL(l2_normalize_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, rows, dim, out, ld_out ? ld_out : dim);
  return check_hip(hipGetLastError(), "l2_normalize launch");
}

int add_pos_embed(float* x, const float* base, const float* time_inds, const float* freq_inds, const float* freq_table,
                  int64_t rows, int dim, int num_freq, hipStream_t st) {
  CACO_REQUIRE(dim % 8 == 0, "add_pos_embed: dim %d must be a multiple of 8", dim);
  hipLaunchKernelGGL(add_pos_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, base, time_inds,
                     freq_inds, freq_table, rows, dim, num_freq);
  return check_hip(hipGetLastError(), "add_pos_embed launch");
}

int pos_prepare(const float* freq_table, int tmax, int num_freq, int dim, float* table, const float* time_inds,
                const float* freq_inds, int64_t rows, int* idx, hipStream_t st) {
  CACO_REQUIRE(dim % 8 == 0 && tmax > 0 && num_freq > 0, "pos_prepare: bad table shape");
  const int table_blocks = (tmax * num_freq + 3) / 4;
  const int64_t idx_blocks = (rows + 255) / 256;
  hipLaunchKernelGGL(pos_prepare_kernel, dim3((unsigned)(table_blocks + idx_blocks)), dim3(256), 0, st, freq_table, tmax, num_freq,
                     dim, table, table_blocks, time_inds, freq_inds, rows, idx);
  return check_hip(hipGetLastError(), "pos_prepare launch");
}

int add_pos_embed_rest(float* x, const float* time_inds, const float* freq_inds, const float* freq_table, const int* idx,
                       int64_t rows, int dim, int num_freq, hipStream_t st) {
  hipLaunchKernelGGL(add_pos_embed_rest_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, time_inds, freq_inds,
                     freq_table, idx, rows, dim, num_freq);
  return check_hip(hipGetLastError(), "add_pos_embed_rest launch");
}

int mask_i64_to_f32(const int64_t* in, float* out, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(mask_i64_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
  return check_hip(hipGetLastError(), "mask_i64_to_f32 launch");
}

int cast_f32_to_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t st) {
  CACO_REQUIRE(n % 4 == 0, "cast_f32_to_bf16: n must be a multiple of 4");
  hipLaunchKernelGGL(cast_f32_to_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, in, out, n / 4);
  return check_hip(hipGetLastError(), "cast_f32_to_bf16 launch");
}

int copy_rows(const float* src, float* dst, int batch, int src_seq, int dst_seq, int dst_off, int dim, hipStream_t st) {
  CACO_REQUIRE(dim % 4 == 0, "copy_rows: dim must be a multiple of 4");
  const int64_t total = (int64_t)batch * src_seq * (dim / 4);
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, dst, src_seq, dst_seq,
                     dst_off, dim / 4, total);
  return check_hip(hipGetLastError(), "copy_rows launch");
}

}  // namespace caco
