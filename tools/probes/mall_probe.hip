// Probe: does data WRITTEN by one kernel stay in the 256 MiB Infinity Cache for the NEXT kernel to read?
// For buffer sizes X: time a streaming write of X, then a streaming read of X (separate launches, same stream).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void wr(float4* p, size_t n, float v) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void rd(const float4* p, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 1234.5f) out[0] = s;
}
__global__ __launch_bounds__(256) void cp(const float4* a, float4* b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main() {
  const size_t maxb = 2048ull << 20;
  float4 *a, *b; float* out;
  hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&out, 64);
  hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
  for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    float tw = 0, tr = 0, tc = 0; const int reps = 10;
    for (int r = 0; r < reps + 2; ++r) {
      hipEventRecord(e[0]);
      wr<<<2048, 256>>>(a, n, (float)r);
      hipEventRecord(e[1]);
      rd<<<2048, 256>>>(a, n, out);
      hipEventRecord(e[2]);
      cp<<<2048, 256>>>(a, b, n / 2);       // read half, write half: working set = X
      hipEventRecord(e[3]);
      hipDeviceSynchronize();
      float x, y, z; hipEventElapsedTime(&x, e[0], e[1]); hipEventElapsedTime(&y, e[1], e[2]); hipEventElapsedTime(&z, e[2], e[3]);
      if (r >= 2) { tw += x; tr += y; tc += z; }
    }
    printf("%5zu MB: write %7.2f TB/s   read-after-write %7.2f TB/s   copy(X/2->X/2) %7.2f TB/s (r+w bytes)\n", mb,
           bytes / (tw / reps) / 1e9, bytes / (tr / reps) / 1e9, bytes / (tc / reps) / 1e9);
  }
  return 0;
}
