// Learned-query attention pooling: one query vector per head attends over the S tokens of a clip /
// caption and returns the softmax-weighted mean of the values.
//
//   audio: AudioAttentionPooler.forward, src/caco_torch/caco.py:41-79 (2 heads x 384, scale 1/sqrt(384))
//   text : AttentionPooler.forward, src/caco_torch/text_models/roberta.py:253-271 (1 head x 768, key/sqrt(768))
//
// Input is the fused [k | v] projection kv[B*S, 2H] (bf16) written by one GEMM.  One workgroup per
// (head, clip): scores by wave-reduced dot products into LDS, block softmax in fp32, then the
// value reduction with the key loop split over the 4 waves and combined through LDS.
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

__global__ __launch_bounds__(256) void attn_pool_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ query,
                                                        const float* __restrict__ mask, int S, int H, int heads,
                                                        float scale, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sc = sm;                       // [S] scores -> probabilities
  float* red = sm + S;                  // [8] reduction scratch
  float* part = red + 8;                // [4][hd] per-wave partial outputs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int hd = H / heads;
  const bf16_t* kb = kv + (int64_t)b * S * (2 * H) + h * hd;
  const bf16_t* vb = kb + H;
  const float* q = query + h * hd;

  // 1. scores
  for (int j = wave; j < S; j += 4) {
    float acc = 0.f;
    const bf16_t* kr = kb + (int64_t)j * (2 * H);
    for (int d = lane * 2; d < hd; d += 128) {
      const bf16x2 k2 = *reinterpret_cast<const bf16x2*>(kr + d);
      acc += (float)k2[0] * q[d] + (float)k2[1] * q[d + 1];
    }
    acc = wave_sum(acc) * scale;
    if (mask && mask[(int64_t)b * S + j] == 0.f) acc = -INFINITY;
    if (lane == 0) sc[j] = acc;
  }
  __syncthreads();
  // 2. softmax over S
  float m = -INFINITY;
  for (int j = tid; j < S; j += 256) m = fmaxf(m, sc[j]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float m_use = (m == -INFINITY) ? 0.f : m;
  float s = 0.f;
  for (int j = tid; j < S; j += 256) {
    const float p = __expf(sc[j] - m_use);
    sc[j] = p;
    s += p;
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  const float tot = red[4] + red[5] + red[6] + red[7];
  const float inv = tot > 0.f ? 1.0f / tot : 0.f;
  // 3. out[d] = sum_j p_j v[j][d]; wave w takes keys j = w, w+4, ...; lanes own element pairs
  for (int d0 = 0; d0 < hd; d0 += 128) {
    const int d = d0 + lane * 2;
    float a0 = 0.f, a1 = 0.f;
    if (d < hd) {
      for (int j = wave; j < S; j += 4) {
        const bf16x2 v2 = *reinterpret_cast<const bf16x2*>(vb + (int64_t)j * (2 * H) + d);
        const float p = sc[j];
        a0 += p * (float)v2[0];
        a1 += p * (float)v2[1];
      }
      part[wave * hd + d] = a0;
      part[wave * hd + d + 1] = a1;
    }
  }
  __syncthreads();
  for (int d = tid; d < hd; d += 256)
    out[(int64_t)b * H + h * hd + d] = (part[d] + part[hd + d] + part[2 * hd + d] + part[3 * hd + d]) * inv;
}

}  // namespace

int attn_pool(const bf16_t* kv, const float* query, const float* mask, int batch, int seq, int hidden, int heads,
              float scale, float* out, hipStream_t st) {
  CACO_REQUIRE(heads > 0 && hidden % heads == 0 && (hidden / heads) % 2 == 0, "attn_pool: bad hidden/heads %d/%d", hidden, heads);
  const int hd = hidden / heads;
  const size_t smem = (size_t)(seq + 8 + 4 * hd) * sizeof(float);
  CACO_REQUIRE(smem <= 64 * 1024, "attn_pool: sequence %d too long for the LDS score buffer", seq);
  hipLaunchKernelGGL(attn_pool_kernel, dim3(heads, batch), dim3(256), smem, st, kv, query, mask, seq, hidden, heads, scale, out);
  return check_hip(hipGetLastError(), "attn_pool launch");
}

}  // namespace caco
