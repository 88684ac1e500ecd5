// Self-test of the ThreadSanitizer mode (tools/wavesim/tsan_check.py): two kernels that differ by one barrier.
// Not part of the library; compiled into libcaco_sim_tsan.so only (build_sim.py --extra).
#include <hip/hip_runtime.h>

namespace {
template <bool BARRIER>
__global__ void handoff_kernel(const float* in, float* out) {
  __shared__ float box[64];
  const int t = threadIdx.x;
  if (t < 64) box[t] = in[t] * 2.f;                 // wave 0 writes
  if (BARRIER) __syncthreads();
  if (t >= 64) out[t - 64] = box[127 - t];          // wave 1 reads what wave 0 wrote
}
__global__ void overlap_kernel(float* out, int stride) {
  out[blockIdx.x * stride + threadIdx.x] = (float)blockIdx.x;      // stride < blockDim.x: neighbouring workgroups overlap
}
}  // namespace

extern "C" void selftest_handoff(const float* in, float* out, int barrier) {
  if (barrier) hipLaunchKernelGGL(handoff_kernel<true>, dim3(1), dim3(128), 0, 0, in, out);
  else hipLaunchKernelGGL(handoff_kernel<false>, dim3(1), dim3(128), 0, 0, in, out);
}
extern "C" void selftest_overlap(float* out, int stride) {
  hipLaunchKernelGGL(overlap_kernel, dim3(2), dim3(64), 0, 0, out, stride);
}
