// Shared device/host helpers for the Cacophony gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CACO_WAVE 64

// CACO_WAVE_LDS_SYNC(): lanes of ONE wave hand data to each other through LDS at this point (one lane's ds_write, another
// lane's ds_read) with no workgroup barrier in between.  On the hardware a wave executes in lockstep and the compiler keeps
// the program order of the may-alias LDS accesses, so in the product build the macro expands to NOTHING in the kernels that
// were verified on hardware in that form (their ISA keeps the ds_write -> s_waitcnt -> ds_read order); the functional
// simulator (tools/wavesim: a lane is a fiber) makes it a wave rendezvous.  A translation unit that defines
// CACO_WAVE_SYNC_FENCE first (the round-3 kernels no GPU has run yet) gets wavefront-scope fences around a wave barrier
// instead: no instruction is emitted, but the compiler may not move an LDS access across the mark whatever it can prove
// about the addresses.
#if defined(WAVESIM)
#define CACO_WAVE_LDS_SYNC() wavesim_wave_sync()
#elif defined(CACO_WAVE_SYNC_FENCE)
#define CACO_WAVE_LDS_SYNC()                                  \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)
#else
#define CACO_WAVE_LDS_SYNC() ((void)0)
#endif

namespace caco {

// error plumbing shared by all translation units (defined in api.hip)
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define CACO_HIP(call)                                   \
  do {                                                   \
    int _st = caco::check_hip((call), #call);            \
    if (_st) return _st;                                 \
  } while (0)

#define CACO_REQUIRE(cond, ...)                          \
  do {                                                   \
    if (!(cond)) {                                       \
      caco::set_error(__VA_ARGS__);                      \
      return CACO_ERR_INVALID;                           \
    }                                                    \
  } while (0)

enum { CACO_OK_ = 0 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each): the result is rounded to bf16 right after
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf(z) for the erf-GELU of the text stack (F.gelu, exact form: text_models/roberta.py:150-158), branch-free:
// z clamped to [-4, 4] (erf(4) = 1 - 1.5e-8), erf(z) = z P(z^2) / Q(z^2) with the degree-6 / degree-4 minimax pair that
// Eigen / XLA use for fp32 erf.  Checked against scipy.special.erf on 2e6 points of [-6, 6] (tests/test_oracle_golden.py):
// |error| <= 4.5e-7 absolute, <= 2.7e-7 relative for |z| < 1; GELU built on it differs from torch's by <= 1.5e-6 absolute,
// i.e. by less than 1/1000 of the bf16 rounding the activation gets right afterwards.  The device library's erff is a
// two-branch routine (|z| < 1 polynomial, else exp): on 128 independent values per lane both branches run with the lanes
// masked, 5 500 instructions per wave and output tile in the persistent GEMM's epilogue against its 3 600-instruction K-loop
// (profiles/r4_cpu/epilogue_budget.txt).  This form is 17 VALU + 1 v_rcp per value.  Every GEMM kernel uses this one function,
// so a caption's embedding does not depend on which tile shape its batch size selected.
__device__ __forceinline__ float erf_rational_f(float z) {
  z = fminf(fmaxf(z, -4.0f), 4.0f);
  const float z2 = z * z;
  float p = -2.72614225801306e-10f;
  p = __builtin_fmaf(p, z2, 2.77068142495902e-08f);
  p = __builtin_fmaf(p, z2, -2.10102402082508e-06f);
  p = __builtin_fmaf(p, z2, -5.69250639462346e-05f);
  p = __builtin_fmaf(p, z2, -7.34990630326855e-04f);
  p = __builtin_fmaf(p, z2, -2.95459980854025e-03f);
  p = __builtin_fmaf(p, z2, -1.60960333262415e-02f);
  float q = -1.45660718464996e-05f;
  q = __builtin_fmaf(q, z2, -2.13374055278905e-04f);
  q = __builtin_fmaf(q, z2, -1.68282697438203e-03f);
  q = __builtin_fmaf(q, z2, -7.37332916720468e-03f);
  q = __builtin_fmaf(q, z2, -1.42647390514189e-02f);
  return (z * p) * __builtin_amdgcn_rcpf(q);
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float h = 0.5f * x;
#ifdef CACO_LIBM_ERF   // bisection arm `libm_erf` (tools/build_variants.sh): rounds 1-2's form on the device library's erff
  return __builtin_fmaf(h, erff(x * 0.70710678118654752440f), h);
#else
  return __builtin_fmaf(h, erf_rational_f(x * 0.70710678118654752440f), h);
#endif
}

// host: fp32 -> bf16 round-to-nearest-even
static inline uint16_t f32_to_bf16_host(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

}  // namespace caco
