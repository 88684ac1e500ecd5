// Row-wise top-k of the similarity matrix: the device part of the reference's retrieval scoring,
//   at_indices = argsort(-logits_ar.T)[:, :10],  ta_indices = argsort(-logits_ar)[:, :10]
// (src/eval/eval_caco_torch.py:402-408; compute_retrieval_metric, src/eval/eval_utils.py:18-54, only ever reads the
// first 10 columns).  The full argsort and the D2H copy of an [N, N] index matrix are replaced by k selection
// passes of one wave per row; the matrix is read through arbitrary (row, column) strides, so the text->audio and
// audio->text directions both run on the single stored matrix (no transpose).
// Order: value descending, ties by ascending index (what a stable argsort of the negated row gives); NaN never wins.
#include <limits.h>

#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ sim, int rows, int cols, int64_t row_stride,
                                                        int64_t col_stride, int k, int* __restrict__ idx, float* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* r = sim + (int64_t)row * row_stride;
  float pv = INFINITY;     // previously selected (value, index); the next pick must come strictly after it
  int pc = -1;
  for (int t = 0; t < k; ++t) {
    float bv = -INFINITY;
    int bc = INT_MAX;
    for (int c = lane; c < cols; c += 64) {
      const float v = r[(int64_t)c * col_stride];
      const bool after = (v < pv) || (v == pv && c > pc);
      const bool better = (v > bv) || (v == bv && c < bc);
      if (after && better) { bv = v; bc = c; }      // NaN compares false everywhere: never selected
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oc = __shfl_xor(bc, o, 64);
      if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
    }
    if (bc == INT_MAX) { bc = -1; bv = -INFINITY; }   // fewer than k selectable entries
    if (lane == 0) {
      idx[(int64_t)row * k + t] = bc;
      if (val) val[(int64_t)row * k + t] = bv;
    }
    if (bc < 0) { pv = -INFINITY; pc = INT_MAX; } else { pv = bv; pc = bc; }
  }
}

}  // namespace

int topk_rows(const float* sim, int rows, int cols, int64_t row_stride, int64_t col_stride, int k, int* idx, float* val,
              hipStream_t st) {
  CACO_REQUIRE(sim && idx, "topk: null argument");
  CACO_REQUIRE(rows > 0 && cols > 0 && k > 0 && k <= 64, "topk: need rows, cols > 0 and 1 <= k <= 64 (got %d, %d, %d)", rows, cols, k);
  hipLaunchKernelGGL(topk_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, sim, rows, cols, row_stride, col_stride, k, idx, val);
  return check_hip(hipGetLastError(), "topk launch");
}

}  // namespace caco
