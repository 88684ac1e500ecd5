// LayerNorm of one row held by one 64-lane wave (lane owns float4 chunks lane, lane + 64, ...): shared by the stand-alone
// LayerNorm kernels (norm.hip) and by the GEMM that normalises its own output rows (gemm_rowln.hip), so that both produce
// the same bits for the same row (same summation order, same two-pass variance).
#pragma once
#include "common.h"

namespace caco {
namespace {

constexpr int MAXC = 4;  // float4 chunks per lane -> dim <= 1024

// The mean and the variance scale by 1 / dim (no division whose expansion could depend on dim being a constant) and the
// output fma is explicit so that the two translation units that inline this function
// cannot make different fusion choices: the stand-alone kernel and the fused GEMM must agree to the bit.
__device__ __forceinline__ void ln_row(f32x4 (&v)[MAXC], int nchunk, int lane, int dim, const float* gamma,
                                       const float* beta, float eps, float* of, bf16_t* ob) {
  const float inv_dim = 1.0f / (float)dim;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c * 64 + lane < nchunk) s += ((v[c][0] + v[c][1]) + v[c][2]) + v[c][3];
  const float mean = wave_sum(s) * inv_dim;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c)
    if (c * 64 + lane < nchunk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[c][r] -= mean;
        q += v[c][r] * v[c][r];      // (an explicit fmaf here makes the compiler keep copies of v: 99 instead of 44 registers)
      }
    }
  const float rstd = rsqrtf(__builtin_fmaf(wave_sum(q), inv_dim, eps));
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nchunk) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + ch * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + ch * 4);
      f32x4 y;
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = __builtin_fmaf(v[c][r] * rstd, g[r], b[r]);
      if (of) *reinterpret_cast<f32x4*>(of + ch * 4) = y;
      if (ob) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)y[r];
        *reinterpret_cast<bf16x4*>(ob + ch * 4) = o;
      }
    }
  }
}

}  // namespace
}  // namespace caco
