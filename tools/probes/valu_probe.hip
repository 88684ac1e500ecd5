// Probe: issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU ops the attention softmax uses.
//   hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
  float a[16];
  f32x2 p[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long m0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
      if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
      if (OP == 5) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
      if (OP == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
      if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
      if (OP == 8) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
    }
  }
  const unsigned long long m1 = __builtin_amdgcn_s_memtime();
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
  if (s == 12345.f) out[threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[256] = (float)(m1 - m0); out[257] = (float)(t1 - t0); }
}
template <int OP>
void run(const char* name, float* out, int threads = 256) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<OP><<<256, threads>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<OP><<<256, threads>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  float h[2]; (void)hipMemcpy(h, out + 256, 8, hipMemcpyDeviceToHost);
  printf("%4d thr/CU %-18s %.3f ms  %.2f ns/instr  s_memtime %.2f ticks/instr  cyclecounter %.2f /instr\n", threads, name, ms, ms * 1e6 / (iters * 16.0),
         h[0] / (iters * 16.0), h[1] / (iters * 16.0));
}
int main() {
  float* out; (void)hipMalloc(&out, 2048);
  run<0>("v_exp_f32", out); run<1>("v_rcp_f32", out); run<2>("v_fma_f32", out); run<3>("v_pk_fma_f32", out);
  run<4>("v_pk_mul_f32", out); run<5>("v_max3_f32", out); run<6>("v_cvt_pk_bf16_f32", out); run<7>("v_pk_add_f32", out);
  run<8>("v_add_f32", out);
  for (int thr : {256, 512, 1024}) { run<2>("v_fma_f32", out, thr); run<0>("v_exp_f32", out, thr); run<3>("v_pk_fma_f32", out, thr); }
  return 0;
}
