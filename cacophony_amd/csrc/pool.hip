// Learned-query attention pooling: one query vector per head attends over the S tokens of a clip / caption and
// returns the softmax-weighted mean of the values.
//
//   audio: AudioAttentionPooler.forward, src/caco_torch/caco.py:41-79 (kv_proj 768 -> 1536, 2 heads x 384, q / sqrt(384))
//   text : AttentionPooler.forward, src/caco_torch/text_models/roberta.py:253-271 (key_proj / sqrt(768), value_proj, 1 head)
//
// Both linear maps are moved out of the token dimension (exact algebra, no approximation):
//   scores[j,h] = (x_j Wk_h^T + bk_h) . q_h * s = x_j . (s Wk_h^T q_h) + const_h       softmax is shift-invariant: const_h drops
//   out_h       = sum_j p[j,h] (x_j Wv_h^T + bv_h) = (sum_j p[j,h] x_j) Wv_h^T + bv_h  since sum_j p[j,h] = 1
// so the device work over the tokens is ONE pass over the encoder output x (bf16, 197 MB at batch 256): per head a dot
// product with the pre-folded vector wq_h = s Wk_h^T q_h and an online-softmax weighted sum of the rows themselves;
// the [B*S, 2H] key/value projection GEMM and its 393 MB round trip disappear, and the value projection becomes an
// exact-fp32 GEMM on the pooled [B, heads, H] rows (api.hip).  A clip whose tokens are all masked pools to 0.
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int PW = 8;            // waves per workgroup; one workgroup per clip / caption
constexpr int PMAXC = 4;         // bf16x4 chunks per lane -> hidden <= 1024

template <int HEADS>
__global__ __launch_bounds__(PW * 64) void pool_rows_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wq,
                                                            const float* __restrict__ mask, int S, int H,
                                                            float* __restrict__ out, int out_heads) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // [PW][HEADS][H] partial sums, then [PW][HEADS][2] (m, l)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int nchunk = H >> 2;                                      // 4-element chunks per row
  const bf16_t* xb = x + (int64_t)b * S * H;

  f32x4 w[HEADS][PMAXC];
#pragma unroll
  for (int h = 0; h < HEADS; ++h)
#pragma unroll
    for (int c = 0; c < PMAXC; ++c)
      w[h][c] = (c * 64 + lane < nchunk) ? *reinterpret_cast<const f32x4*>(wq + (int64_t)h * H + (c * 64 + lane) * 4)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[HEADS][PMAXC];
  float m_run[HEADS], l_run[HEADS];
#pragma unroll
  for (int h = 0; h < HEADS; ++h) {
    m_run[h] = -INFINITY;
    l_run[h] = 0.f;
#pragma unroll
    for (int c = 0; c < PMAXC; ++c) acc[h][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // rows of this wave, four in flight: the loads of a group are issued before the first dot product needs them
  constexpr int UN = 4;
  for (int j0 = wave; j0 < S; j0 += PW * UN) {
    bf16x4 raw[UN][PMAXC];
    bool keep[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * PW;
      keep[u] = j < S && !(mask && mask[(int64_t)b * S + j] == 0.f);     // wave-uniform: masked tokens carry no weight
#pragma unroll
      for (int c = 0; c < PMAXC; ++c)
        if (keep[u] && c * 64 + lane < nchunk) raw[u][c] = *reinterpret_cast<const bf16x4*>(xb + (int64_t)j * H + (c * 64 + lane) * 4);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (!keep[u]) continue;
      f32x4 v[PMAXC];
      float dot[HEADS];
#pragma unroll
      for (int h = 0; h < HEADS; ++h) dot[h] = 0.f;
#pragma unroll
      for (int c = 0; c < PMAXC; ++c) {
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c * 64 + lane < nchunk) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[c][e] = (float)raw[u][c][e];
#pragma unroll
          for (int h = 0; h < HEADS; ++h)
            dot[h] += (v[c][0] * w[h][c][0] + v[c][1] * w[h][c][1]) + (v[c][2] * w[h][c][2] + v[c][3] * w[h][c][3]);
        }
      }
#pragma unroll
      for (int h = 0; h < HEADS; ++h) {
        const float s = wave_sum(dot[h]);
        const float m_new = fmaxf(m_run[h], s);
        const float alpha = __expf(m_run[h] - m_new);               // first token: exp(-inf) = 0
        const float p = __expf(s - m_new);
        l_run[h] = l_run[h] * alpha + p;
        m_run[h] = m_new;
#pragma unroll
        for (int c = 0; c < PMAXC; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[h][c][e] = acc[h][c][e] * alpha + p * v[c][e];
      }
    }
  }

  // combine the PW waves' running states
  float* part = sm;                                   // [PW][HEADS][H]
  float* ml = sm + PW * HEADS * H;                    // [PW][HEADS][2]
#pragma unroll
  for (int h = 0; h < HEADS; ++h) {
#pragma unroll
    for (int c = 0; c < PMAXC; ++c)
      if (c * 64 + lane < nchunk) *reinterpret_cast<f32x4*>(part + ((int64_t)wave * HEADS + h) * H + (c * 64 + lane) * 4) = acc[h][c];
    if (lane == 0) {
      ml[(wave * HEADS + h) * 2] = m_run[h];
      ml[(wave * HEADS + h) * 2 + 1] = l_run[h];
    }
  }
  __syncthreads();
  for (int i = tid; i < HEADS * H; i += PW * 64) {
    const int h = i / H, k = i - h * H;
    float mx = -INFINITY;
#pragma unroll
    for (int wv = 0; wv < PW; ++wv) mx = fmaxf(mx, ml[(wv * HEADS + h) * 2]);
    float num = 0.f, den = 0.f;
    if (mx > -INFINITY) {
#pragma unroll
      for (int wv = 0; wv < PW; ++wv) {
        const float f = __expf(ml[(wv * HEADS + h) * 2] - mx);    // a wave that saw no token: exp(-inf) = 0
        den += ml[(wv * HEADS + h) * 2 + 1] * f;
        num += part[((int64_t)wv * HEADS + h) * H + k] * f;
      }
    }
    out[((int64_t)b * out_heads + h) * H + k] = den > 0.f ? num / den : 0.f;       // out_heads: heads per clip in `out`
  }
}


// The same pooling with the encoder's FINAL LayerNorm applied on the way in: x are the fp32 residual rows, each row is
// normalised (two-pass mean / variance, as norm.hip ln_row) in registers and pooled without ever being written - for callers
// that do not ask for the hidden states (encode_audio), this removes the LayerNorm pass' 195 MB bf16 write and this kernel's
// 195 MB read of it per batch of 256.  The pooled rows are sums of fp32 LayerNorm outputs instead of their bf16 roundings.
// Round 3, opt-in (CACO_POOL_FUSE=1, api.hip) until timed on hardware; checked on the wavesim build.
template <int HEADS>
__global__ __launch_bounds__(PW * 64) void pool_rows_ln_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               const float* __restrict__ wq, const float* __restrict__ mask, int S,
                                                               int H, float* __restrict__ out, int out_heads) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int nchunk = H >> 2;
  const float* xb = x + (int64_t)b * S * H;
  const float inv_h = 1.0f / (float)H;

  f32x4 w[HEADS][PMAXC], g[PMAXC], be[PMAXC];
#pragma unroll
  for (int c = 0; c < PMAXC; ++c) {
    const bool in = c * 64 + lane < nchunk;
    g[c] = in ? *reinterpret_cast<const f32x4*>(gamma + (c * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    be[c] = in ? *reinterpret_cast<const f32x4*>(beta + (c * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
      w[h][c] = in ? *reinterpret_cast<const f32x4*>(wq + (int64_t)h * H + (c * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 acc[HEADS][PMAXC];
  float m_run[HEADS], l_run[HEADS];
#pragma unroll
  for (int h = 0; h < HEADS; ++h) {
    m_run[h] = -INFINITY;
    l_run[h] = 0.f;
#pragma unroll
    for (int c = 0; c < PMAXC; ++c) acc[h][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int UN = 2;
  for (int j0 = wave; j0 < S; j0 += PW * UN) {
    f32x4 raw[UN][PMAXC];
    bool keep[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * PW;
      keep[u] = j < S && !(mask && mask[(int64_t)b * S + j] == 0.f);     // wave-uniform
#pragma unroll
      for (int c = 0; c < PMAXC; ++c)
        raw[u][c] = (keep[u] && c * 64 + lane < nchunk) ? *reinterpret_cast<const f32x4*>(xb + (int64_t)j * H + (c * 64 + lane) * 4)
                                                         : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (!keep[u]) continue;
      float s1 = 0.f;
#pragma unroll
      for (int c = 0; c < PMAXC; ++c) s1 += (raw[u][c][0] + raw[u][c][1]) + (raw[u][c][2] + raw[u][c][3]);
      const float mean = wave_sum(s1) * inv_h;
      float s2 = 0.f;
      f32x4 v[PMAXC];
#pragma unroll
      for (int c = 0; c < PMAXC; ++c) {
        const bool in = c * 64 + lane < nchunk;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[c][e] = in ? raw[u][c][e] - mean : 0.f;
          s2 += v[c][e] * v[c][e];
        }
      }
      const float rstd = rsqrtf(wave_sum(s2) * inv_h + eps);
      float dot[HEADS];
#pragma unroll
      for (int h = 0; h < HEADS; ++h) dot[h] = 0.f;
#pragma unroll
      for (int c = 0; c < PMAXC; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[c][e] = v[c][e] * rstd * g[c][e] + be[c][e];      // padded chunks: g = be = 0
#pragma unroll
        for (int h = 0; h < HEADS; ++h)
          dot[h] += (v[c][0] * w[h][c][0] + v[c][1] * w[h][c][1]) + (v[c][2] * w[h][c][2] + v[c][3] * w[h][c][3]);
      }
#pragma unroll
      for (int h = 0; h < HEADS; ++h) {
        const float s = wave_sum(dot[h]);
        const float m_new = fmaxf(m_run[h], s);
        const float alpha = __expf(m_run[h] - m_new);
        const float p = __expf(s - m_new);
        l_run[h] = l_run[h] * alpha + p;
        m_run[h] = m_new;
#pragma unroll
        for (int c = 0; c < PMAXC; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[h][c][e] = acc[h][c][e] * alpha + p * v[c][e];
      }
    }
  }
  float* part = sm;
  float* ml = sm + PW * HEADS * H;
#pragma unroll
  for (int h = 0; h < HEADS; ++h) {
#pragma unroll
    for (int c = 0; c < PMAXC; ++c)
      if (c * 64 + lane < nchunk) *reinterpret_cast<f32x4*>(part + ((int64_t)wave * HEADS + h) * H + (c * 64 + lane) * 4) = acc[h][c];
    if (lane == 0) {
      ml[(wave * HEADS + h) * 2] = m_run[h];
      ml[(wave * HEADS + h) * 2 + 1] = l_run[h];
    }
  }
  __syncthreads();
  for (int i = tid; i < HEADS * H; i += PW * 64) {
    const int h = i / H, k = i - h * H;
    float mx = -INFINITY;
#pragma unroll
    for (int wv = 0; wv < PW; ++wv) mx = fmaxf(mx, ml[(wv * HEADS + h) * 2]);
    float num = 0.f, den = 0.f;
    if (mx > -INFINITY) {
#pragma unroll
      for (int wv = 0; wv < PW; ++wv) {
        const float f = __expf(ml[(wv * HEADS + h) * 2] - mx);
        den += ml[(wv * HEADS + h) * 2 + 1] * f;
        num += part[((int64_t)wv * HEADS + h) * H + k] * f;
      }
    }
    out[((int64_t)b * out_heads + h) * H + k] = den > 0.f ? num / den : 0.f;
  }
}

// Mean over `group` consecutive tokens (the HEAR "event" embedding: tf.nn.avg_pool(hidden, ksize=8, strides=8, 'VALID')
// over the 8 frequency patches of one time step, src/eval/heareval/embeddings/audio_embedding/caco_embeddings.py:118-124).
// One thread per 4 output channels; a warp reads 1 KiB-contiguous row segments.  HBM-bound: seq*hidden*4 B in per clip.
__global__ __launch_bounds__(256) void token_group_mean_kernel(const float* __restrict__ x, int seq, int hidden, int group, int n_out,
                                                               float* __restrict__ out) {
  const int h4 = hidden >> 2;
  const int b = blockIdx.y;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n_out * h4; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(e / h4), c = (int)(e - (int64_t)t * h4);
    const f32x4* src = reinterpret_cast<const f32x4*>(x + ((int64_t)b * seq + (int64_t)t * group) * hidden) + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < group; ++g) acc += src[(int64_t)g * h4];
    const float inv = 1.0f / (float)group;
    reinterpret_cast<f32x4*>(out + ((int64_t)b * n_out + t) * hidden)[c] = acc * inv;
  }
}

}  // namespace

// x bf16 [batch, seq, hidden]; wq fp32 [heads, hidden] (scale and key projection folded in); out fp32 [batch, heads, hidden]
int attn_pool_rows(const bf16_t* x, const float* wq, const float* mask, int batch, int seq, int hidden, int heads, float* out,
                   hipStream_t st) {
  CACO_REQUIRE(x && wq && out && batch > 0 && seq > 0, "attn_pool: bad arguments");
  CACO_REQUIRE(heads == 1 || heads % 2 == 0, "attn_pool: %d pooling heads unsupported (1 or an even number)", heads);
  CACO_REQUIRE(hidden % 4 == 0 && hidden <= 256 * PMAXC, "attn_pool: hidden %d must be a multiple of 4, <= %d", hidden, 256 * PMAXC);
  // two heads per pass over the rows (the reference's pooler); more heads - the JAX checkpoints pool with 8,
  // src/caco/load_model.py:46 - take four per pass when they divide (x is read heads / 4 times instead of heads / 2), each
  // launch writing its heads of the [batch, heads, hidden] output
  const int per = heads == 1 ? 1 : (heads % 4 == 0 ? 4 : 2);
  const size_t smem = (size_t)PW * per * (hidden + 2) * sizeof(float);
  CACO_REQUIRE(smem <= 160 * 1024, "attn_pool: hidden %d too large for the LDS combine buffer", hidden);
  if (heads == 1) {
    CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(pool_rows_kernel<1>), 160 * 1024, nullptr));
    hipLaunchKernelGGL(pool_rows_kernel<1>, dim3(batch), dim3(PW * 64), smem, st, x, wq, mask, seq, hidden, out, 1);
  } else if (per == 4) {
    CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(pool_rows_kernel<4>), 160 * 1024, nullptr));
    for (int h0 = 0; h0 < heads; h0 += 4)
      hipLaunchKernelGGL(pool_rows_kernel<4>, dim3(batch), dim3(PW * 64), smem, st, x, wq + (size_t)h0 * hidden, mask, seq, hidden,
                         out + (size_t)h0 * hidden, heads);
  } else {
    CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(pool_rows_kernel<2>), 160 * 1024, nullptr));
    for (int h0 = 0; h0 < heads; h0 += 2)
      hipLaunchKernelGGL(pool_rows_kernel<2>, dim3(batch), dim3(PW * 64), smem, st, x, wq + (size_t)h0 * hidden, mask, seq, hidden,
                         out + (size_t)h0 * hidden, heads);
  }
  return check_hip(hipGetLastError(), "attn_pool launch");
}

// LayerNorm(x; gamma, beta, eps) rows pooled as above without being written: x fp32 [batch, seq, hidden]
int attn_pool_rows_ln(const float* x, const float* gamma, const float* beta, float eps, const float* wq, const float* mask, int batch,
                      int seq, int hidden, int heads, float* out, hipStream_t st) {
  CACO_REQUIRE(x && gamma && beta && wq && out && batch > 0 && seq > 0, "attn_pool_ln: bad arguments");
  CACO_REQUIRE(heads == 1 || heads % 2 == 0, "attn_pool_ln: %d pooling heads unsupported (1 or an even number)", heads);
  CACO_REQUIRE(hidden % 4 == 0 && hidden <= 256 * PMAXC, "attn_pool_ln: hidden %d must be a multiple of 4, <= %d", hidden, 256 * PMAXC);
  const int per = heads == 1 ? 1 : 2;
  const size_t smem = (size_t)PW * per * (hidden + 2) * sizeof(float);
  CACO_REQUIRE(smem <= 160 * 1024, "attn_pool_ln: hidden %d too large for the LDS combine buffer", hidden);
  if (heads == 1) {
    CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(pool_rows_ln_kernel<1>), 160 * 1024, nullptr));
    hipLaunchKernelGGL(pool_rows_ln_kernel<1>, dim3(batch), dim3(PW * 64), smem, st, x, gamma, beta, eps, wq, mask, seq, hidden, out, 1);
  } else {
    CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(pool_rows_ln_kernel<2>), 160 * 1024, nullptr));
    for (int h0 = 0; h0 < heads; h0 += 2)
      hipLaunchKernelGGL(pool_rows_ln_kernel<2>, dim3(batch), dim3(PW * 64), smem, st, x, gamma, beta, eps, wq + (size_t)h0 * hidden,
                         mask, seq, hidden, out + (size_t)h0 * hidden, heads);
  }
  return check_hip(hipGetLastError(), "attn_pool_ln launch");
}

// x fp32 [batch, seq, hidden] -> out fp32 [batch, seq / group, hidden] (trailing seq % group tokens dropped: 'VALID')
int token_group_mean(const float* x, int batch, int seq, int hidden, int group, float* out, hipStream_t st) {
  CACO_REQUIRE(batch > 0 && seq > 0 && group > 0, "token_group_mean: bad shape");
  CACO_REQUIRE(hidden > 0 && hidden % 4 == 0, "token_group_mean: hidden %d must be a positive multiple of 4", hidden);
  const int n_out = seq / group;
  if (n_out == 0) return CACO_OK;               // fewer tokens than one group: empty output ('VALID')
  CACO_REQUIRE(x && out, "token_group_mean: null argument");
  const int64_t work = (int64_t)n_out * (hidden >> 2);
  const int gx = (int)((work + 255) / 256 < 64 ? (work + 255) / 256 : 64);
  hipLaunchKernelGGL(token_group_mean_kernel, dim3(gx, batch), dim3(256), 0, st, x, seq, hidden, group, n_out, out);
  return check_hip(hipGetLastError(), "token_group_mean launch");
}

}  // namespace caco
