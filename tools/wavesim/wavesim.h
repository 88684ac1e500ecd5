// wavesim: a host-side FUNCTIONAL model of the gfx950 device constructs that cacophony_amd/csrc uses, so that the very
// same kernel sources can be compiled for x86 and executed lane by lane on the CPU.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in cacophony_amd/ loads or links this; the product path is the hipcc build of the
// same sources for gfx950 and fails loudly without it.  The simulator exists because a kernel's index arithmetic (MFMA
// fragment layouts, LDS swizzles, LDS-DMA addressing, hardware transpose reads, buffer-descriptor range checks, counted
// s_waitcnt vmcnt) can be checked without a GPU; it says nothing about speed.
//
// Model
//   * one fiber per lane; a workgroup's lanes run on one OS thread, workgroups are spread over a few OS threads
//   * wave-level operations (MFMA, shuffles, ballot, permlane swap, transpose reads, readfirstlane) are rendezvous points:
//     every live lane of the wave deposits its operands, the last arrival releases the wave, each lane then computes its
//     own share of the result from the deposited operands.  All live lanes of a wave must arrive at the SAME call site
//     (the kernels keep wave-uniform control flow around such operations); anything else aborts with a diagnostic.
//   * s_barrier / __syncthreads: rendezvous of all live waves of the workgroup
//   * vector-memory bookkeeping per lane, in issue order: every buffer / global load, store and LDS-DMA counts one;
//     s_waitcnt vmcnt(N) retires the oldest until N remain.  In the default LATE mode an LDS-DMA's bytes are captured at
//     issue and only written to LDS when a covering s_waitcnt retires it - the latest moment the hardware allows - so a
//     missing or too-loose wait shows up as stale operands.  WAVESIM_DMA=eager writes them at issue.
//   * __syncthreads() = s_waitcnt vmcnt(0) + barrier (what hipcc emits); __builtin_amdgcn_s_barrier() is the bare barrier
//   * LDS is ordinary host memory: `__shared__` becomes a thread_local static (one workgroup per OS thread at a time)
//   * raw buffer descriptors: base + byte range; accesses are range-checked per dword like the hardware (loads of
//     out-of-range dwords return 0, stores are dropped) - on the per-lane offset only: the scalar offset is outside the check
//   * MFMA numerics: exact products of the bf16 / fp32 inputs accumulated in fp32 in k order (the hardware's internal
//     order is unspecified; differences are below the tests' tolerances)
//
// Semantics of the instructions are taken from the CDNA4 ISA as used by the GPU-verified kernels of this repository
// (attention.hip, gemm_w8.hip ... passed `pytest -m gpu` on MI355X): those kernels producing oracle-exact results under
// this model is what pins the model (tests/test_wavesim.py).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <functional>

#define WAVESIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

// ------------------------------------------------------------------------------------------------------------------
// runtime (wavesim.cpp)
// ------------------------------------------------------------------------------------------------------------------
namespace wavesim {

struct idx3 { unsigned x, y, z; };
constexpr int XBYTES = 96;          // operand bytes a lane can deposit at a rendezvous

struct Wave;
struct VmOp { char* lds; int bytes; unsigned char data[16]; };
struct Lane {
  void* sp;
  char* stack;
  idx3 tid;
  int lane;               // index within the wave
  int state;              // 0 ready, 1 waiting at a wave operation, 2 waiting at the workgroup barrier, 3 done
  const void* site;       // call site of the wave operation being waited at
  Wave* wave;
  VmOp vm[256];           // outstanding vector-memory operations, oldest first (ring)
  int vm_head, vm_count;
};
struct Wave {
  Lane* lanes[64];
  int nlanes;
  int gen;                                      // completed rendezvous count
  unsigned long long active[2];                 // lanes that took part in a rendezvous, by parity (the EXEC mask of the operation)
  alignas(16) unsigned char x[2][64][XBYTES];    // deposited operands, by rendezvous parity
};

#define WAVESIM_TLS thread_local
extern WAVESIM_TLS Lane* cur;
extern WAVESIM_TLS idx3 block_idx, block_dim, grid_dim;
extern int dma_late;                            // 1 = LDS-DMA bytes land at the covering s_waitcnt (default)

void wave_rendezvous(const void* site);         // returns when every live lane of the wave has arrived at `site`
void block_barrier();
int block_barrier_or(int pred);                 // __syncthreads_or: barrier + OR of every live lane's predicate
char* dyn_lds();                                // the workgroup's dynamic LDS (160 KiB)
void launch(idx3 grid, idx3 block, size_t lds_bytes, const std::function<void()>& body, const char* what = "", const void* fn = nullptr);
void declare_dyn_lds(const void* fn, int bytes);      // hipFuncSetAttribute(MaxDynamicSharedMemorySize): required above 64 KiB
[[noreturn]] void fail(const char* fmt, ...);

inline unsigned char* xput() { return cur->wave->x[cur->wave->gen & 1][cur->lane]; }
inline const unsigned char* xget(int lane) { return cur->wave->x[(cur->wave->gen - 1) & 1][lane]; }
inline bool xlive(int lane) { return (cur->wave->active[(cur->wave->gen - 1) & 1] >> lane) & 1ull; }

// vector-memory queue
#ifdef WAVESIM_TSAN
extern "C" void __tsan_write_range(void* addr, unsigned long size);
#endif
inline void vm_retire_one(Lane* L) {
  VmOp& o = L->vm[L->vm_head];
  if (o.bytes > 0) memcpy(o.lds, o.data, (size_t)o.bytes);
  L->vm_head = (L->vm_head + 1) & 255;
  --L->vm_count;
}
inline void vm_push(char* lds, const void* data, int bytes) {
  Lane* L = cur;
  if (L->vm_count == 256) fail("more than 256 vector-memory operations outstanding in one lane");
  VmOp& o = L->vm[(L->vm_head + L->vm_count) & 255];
  o.lds = lds;
  o.bytes = 0;
  if (bytes > 0) {
#ifdef WAVESIM_TSAN
    // the hardware may write the bytes at any moment between issue and the covering wait: the race detector is told about a
    // write at both ends (a wave still reading this buffer when another wave issues its refill is a WAR hazard)
    __tsan_write_range(lds, (unsigned long)bytes);
#endif
    if (dma_late) { o.bytes = bytes; memcpy(o.data, data, (size_t)bytes); }
    else memcpy(lds, data, (size_t)bytes);
  }
  ++L->vm_count;
}
inline void vm_wait(int n) {
  Lane* L = cur;
  while (L->vm_count > n) vm_retire_one(L);
}
void s_waitcnt(const char* text);               // "vmcnt(N) lgkmcnt(M)" in any order; lgkmcnt is a no-op here
inline void s_waitcnt_n(const char* text, int n) {     // the "%0" form with an immediate operand: vmcnt(%0)
  if (strstr(text, "vmcnt(%0)")) vm_wait(n);
  else fail("s_waitcnt_n: unsupported form '%s'", text);
}

}  // namespace wavesim

#define threadIdx (wavesim::cur->tid)
#define blockIdx (wavesim::block_idx)
#define blockDim (wavesim::block_dim)
#define gridDim (wavesim::grid_dim)
#define warpSize 64

// ------------------------------------------------------------------------------------------------------------------
// HIP host API (just enough for csrc/api.hip): device memory is host memory, streams are synchronous
// ------------------------------------------------------------------------------------------------------------------
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct wavesim_stream* hipStream_t;
typedef struct wavesim_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; };

hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height,
                                          hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < height; ++r) memmove((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev);
static inline hipError_t hipFuncSetAttribute(const void* fn, hipFuncAttribute, int v) { wavesim::declare_dyn_lds(fn, v); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "wavesim"; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// launches are synchronous here, so an event is the wall clock at the moment it is recorded
struct wavesim_event { double t_ms; };
static inline double wavesim_now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new wavesim_event{0.0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (e) e->t_ms = wavesim_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (a && b) ? (float)(b->t_ms - a->t_ms) : 0.f; return hipSuccess; }

#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...)                                            \
  do {                                                                                                     \
    const dim3 g_ = (grid), b_ = (block);                                                                  \
    wavesim::launch(wavesim::idx3{g_.x, g_.y, g_.z}, wavesim::idx3{b_.x, b_.y, b_.z}, (size_t)(lds),        \
                    [&]() { kern(__VA_ARGS__); }, #kern, reinterpret_cast<const void*>(+kern));                                                 \
  } while (0)

// ------------------------------------------------------------------------------------------------------------------
// device-side helpers of the HIP headers
// ------------------------------------------------------------------------------------------------------------------
template <typename A, typename B>
static inline auto min(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
template <typename A, typename B>
static inline auto max(A a, B b) -> decltype(a + b) { return a > b ? a : b; }
// __expf lowers to v_mul_f32 (by log2 e) + v_exp_f32 on the device: the same two roundings here, so that a kernel that writes
// exp2(x * log2e) by hand agrees bit for bit with one that calls __expf, as it does on the hardware
#define __expf(x) exp2f((x) * 1.4426950408889634f)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
// points where the lanes of a wave exchange data through LDS (common.h CACO_WAVE_LDS_SYNC, __builtin_amdgcn_wave_barrier)
static inline __attribute__((noinline)) void wavesim_wave_sync() { wavesim::wave_rendezvous(__builtin_return_address(0)); }
static inline __attribute__((noinline)) void __builtin_amdgcn_wave_barrier() { wavesim::wave_rendezvous(__builtin_return_address(0)); }
static inline void __builtin_amdgcn_s_barrier() { wavesim::block_barrier(); }
static inline void __syncthreads() { wavesim::vm_wait(0); wavesim::block_barrier(); }
static inline int __syncthreads_or(int pred) { wavesim::vm_wait(0); return wavesim::block_barrier_or(pred); }

template <typename T>
static inline T wavesim_exchange_from(T v, int src_lane, const void* site) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  memcpy(wavesim::xput(), &v, sizeof(T));
  wavesim::wave_rendezvous(site);
  T r = v;
  if (src_lane >= 0 && src_lane < 64 && wavesim::xlive(src_lane)) memcpy(&r, wavesim::xget(src_lane), sizeof(T));
  return r;
}
#define WAVESIM_SITE() __builtin_return_address(0)
template <typename T>
static inline __attribute__((noinline)) T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return wavesim_exchange_from(v, wavesim::cur->lane ^ mask, WAVESIM_SITE());
}
template <typename T>
static inline __attribute__((noinline)) T __shfl(T v, int src, int width = 64) {
  const int base = wavesim::cur->lane & ~(width - 1);
  return wavesim_exchange_from(v, base + (src & (width - 1)), WAVESIM_SITE());
}
template <typename T>
static inline __attribute__((noinline)) T __shfl_down(T v, unsigned d, int width = 64) {
  const int l = wavesim::cur->lane, s = l + (int)d;
  return wavesim_exchange_from(v, (s & ~(width - 1)) == (l & ~(width - 1)) ? s : l, WAVESIM_SITE());
}
static inline __attribute__((noinline)) unsigned long long __ballot(int pred) {
  *wavesim::xput() = pred ? 1 : 0;
  wavesim::wave_rendezvous(WAVESIM_SITE());
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (wavesim::xlive(l) && *wavesim::xget(l)) m |= 1ull << l;
  return m;
}
template <typename T>
static inline __attribute__((noinline)) T __builtin_amdgcn_readfirstlane(T v) {
  memcpy(wavesim::xput(), &v, sizeof(T));
  wavesim::wave_rendezvous(WAVESIM_SITE());
  for (int l = 0; l < 64; ++l)
    if (wavesim::xlive(l)) { T r; memcpy(&r, wavesim::xget(l), sizeof(T)); return r; }
  return v;
}
typedef unsigned int wavesim_u32x2 __attribute__((ext_vector_type(2)));
// v_permlane32_swap vdst, src: lanes 32..63 of vdst trade places with lanes 0..31 of src; returns {vdst, src}
static inline __attribute__((noinline)) wavesim_u32x2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
  unsigned both[2] = {vdst, src};
  memcpy(wavesim::xput(), both, 8);
  wavesim::wave_rendezvous(WAVESIM_SITE());
  const int l = wavesim::cur->lane;
  wavesim_u32x2 r = {vdst, src};
  unsigned o[2];
  if (l >= 32) { memcpy(o, wavesim::xget(l - 32), 8); r[0] = o[1]; }    // vdst[l] <- src[l - 32]
  else { memcpy(o, wavesim::xget(l + 32), 8); r[1] = o[0]; }            // src[l] <- vdst[l + 32]
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
// buffer descriptors, LDS-DMA, transpose reads
// ------------------------------------------------------------------------------------------------------------------
struct __amdgpu_buffer_rsrc_t { char* base; uint32_t num_records; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* base, short stride, int num_records, int flags) {
  (void)stride; (void)flags;
  return __amdgpu_buffer_rsrc_t{reinterpret_cast<char*>(base), (uint32_t)num_records};
}
typedef unsigned int wavesim_u32x4 __attribute__((ext_vector_type(4)));
// Raw buffer addressing: address = base + soffset + (voffset + imm); the RANGE CHECK covers voffset + imm only - the scalar
// offset is "excluded from bounds checking" (LLVM's definition of the raw.buffer intrinsics' soffset operand, and the reason
// composable_kernel's out-of-bounds trick adds 0x80000000 to the per-lane offset, ck/utility/amd_buffer_addressing.hpp).
// A kernel that steps through rows with soffset gets NO protection for the rows it reaches that way: under this model such
// an access goes to memory, where AddressSanitizer (asan_check.py) sees it.
static inline void wavesim_buffer_align(const __amdgpu_buffer_rsrc_t& r, uint32_t checked_off, uint32_t soff) {
  if (((uintptr_t)r.base + checked_off + soff) & 3) wavesim::fail("buffer access at an address that is not dword-aligned");
}
static inline void wavesim_buffer_read(const __amdgpu_buffer_rsrc_t& r, uint32_t checked_off, uint32_t soff, void* dst, int bytes) {
  wavesim_buffer_align(r, checked_off, soff);
  unsigned char* d = reinterpret_cast<unsigned char*>(dst);
  for (int b = 0; b < bytes; b += 4) {
    const int64_t o = (int64_t)checked_off + b;
    if (o + 4 <= (int64_t)r.num_records) memcpy(d + b, r.base + o + soff, 4);
    else memset(d + b, 0, 4);
  }
}
static inline void wavesim_buffer_write(const __amdgpu_buffer_rsrc_t& r, uint32_t checked_off, uint32_t soff, const void* src, int bytes) {
  wavesim_buffer_align(r, checked_off, soff);
  const unsigned char* s = reinterpret_cast<const unsigned char*>(src);
  for (int b = 0; b < bytes; b += 4) {
    const int64_t o = (int64_t)checked_off + b;
    if (o + 4 <= (int64_t)r.num_records) memcpy(r.base + o + soff, s + b, 4);
  }
}
static inline wavesim_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
  (void)aux;
  wavesim_u32x4 v;
  wavesim_buffer_read(r, (uint32_t)voff, (uint32_t)soff, &v, 16);
  wavesim::vm_push(nullptr, nullptr, 0);
  return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b128(wavesim_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
  (void)aux;
  wavesim_buffer_write(r, (uint32_t)voff, (uint32_t)soff, &v, 16);
  wavesim::vm_push(nullptr, nullptr, 0);
}
static inline void __builtin_amdgcn_raw_buffer_store_b64(wavesim_u32x2 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
  (void)aux;
  wavesim_buffer_write(r, (uint32_t)voff, (uint32_t)soff, &v, 8);
  wavesim::vm_push(nullptr, nullptr, 0);
}
typedef __attribute__((address_space(3))) void* wavesim_lds_vptr;
// buffer_load_dword{,x3,x4} ... lds: every lane fetches `size` bytes at voffset + soffset + imm; the wave's data lands in
// LDS at the (wave-uniform) base + lane * size
static inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(__amdgpu_buffer_rsrc_t r, wavesim_lds_vptr lds, int size, int voff,
                                                            int soff, int imm, int aux) {
  (void)aux;
  unsigned char tmp[16];
  if (size > 16) wavesim::fail("buffer_load ... lds of %d bytes", size);
  wavesim_buffer_read(r, (uint32_t)voff + (uint32_t)imm, (uint32_t)soff, tmp, size);
  if (((uintptr_t)lds) & (size - 1) & 15) wavesim::fail("LDS-DMA destination not aligned to its %d-byte elements", size);
  wavesim::vm_push(reinterpret_cast<char*>((uintptr_t)lds) + wavesim::cur->lane * size, tmp, size);
}
typedef __attribute__((address_space(1))) void* wavesim_glb_vptr;
template <typename P>
static inline void __builtin_amdgcn_global_load_lds(P gptr, wavesim_lds_vptr lds, int size, int imm, int aux) {
  (void)aux;
  if (size > 16) wavesim::fail("global_load_lds of %d bytes", size);
  wavesim::vm_push(reinterpret_cast<char*>((uintptr_t)lds) + imm + wavesim::cur->lane * size,
                   reinterpret_cast<const char*>((uintptr_t)gptr) + imm, size);
}

typedef short wavesim_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) wavesim_s16x4* wavesim_lds_s16x4_ptr;
// ds_read_b64_tr_b16: within each group of 16 lanes the 16 x 8-byte reads form a 4 x 16 matrix of 16-bit elements
// distributed row-major (lane = row * 4 + col / 4, element = col % 4); it is returned column-major: lane n of the group
// receives column n, i.e. element e of its result is element n % 4 of what lane e * 4 + n / 4 of the group read.
static inline __attribute__((noinline)) wavesim_s16x4 __builtin_amdgcn_ds_read_tr16_b64_v4i16(wavesim_lds_s16x4_ptr p) {
  memcpy(wavesim::xput(), reinterpret_cast<const void*>((uintptr_t)p), 8);
  wavesim::wave_rendezvous(WAVESIM_SITE());
  const int l = wavesim::cur->lane, g = l & ~15, n = l & 15;
  wavesim_s16x4 r;
  for (int e = 0; e < 4; ++e) {
    short q[4];
    memcpy(q, wavesim::xget(g + e * 4 + (n >> 2)), 8);
    r[e] = q[n & 3];
  }
  return r;
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 wavesim_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wavesim_f32x4 __attribute__((ext_vector_type(4)));
typedef float wavesim_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x32_bf16: A[i][k] from lane i + 16 * (k / 8), element k % 8; B[k][j] likewise from lane j + 16 * (k / 8);
// D[i][j]: lane j + 16 * (i / 4), register i % 4
static inline __attribute__((noinline)) wavesim_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(wavesim_bf16x8 a, wavesim_bf16x8 b, wavesim_f32x4 c,
                                                                                      int, int, int) {
  unsigned char* x = wavesim::xput();
  memcpy(x, &a, 16);
  memcpy(x + 16, &b, 16);
  wavesim::wave_rendezvous(WAVESIM_SITE());
  const int l = wavesim::cur->lane, j = l & 15, ig = l >> 4;
  wavesim_bf16x8 bj[4];
  for (int kg = 0; kg < 4; ++kg) memcpy(&bj[kg], wavesim::xget(j + 16 * kg) + 16, 16);
  for (int r = 0; r < 4; ++r) {
    const int i = ig * 4 + r;
    float acc = c[r];
    for (int kg = 0; kg < 4; ++kg) {
      wavesim_bf16x8 ai;
      memcpy(&ai, wavesim::xget(i + 16 * kg), 16);
      for (int e = 0; e < 8; ++e) acc += (float)ai[e] * (float)bj[kg][e];
    }
    c[r] = acc;
  }
  return c;
}
// v_mfma_f32_32x32x16_bf16: A[i][k] from lane i + 32 * (k / 8), element k % 8; B likewise;
// D[i][j]: lane j + 32 * ((i / 4) % 2), register (i / 8) * 4 + i % 4
static inline __attribute__((noinline)) wavesim_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(wavesim_bf16x8 a, wavesim_bf16x8 b, wavesim_f32x16 c,
                                                                                       int, int, int) {
  unsigned char* x = wavesim::xput();
  memcpy(x, &a, 16);
  memcpy(x + 16, &b, 16);
  wavesim::wave_rendezvous(WAVESIM_SITE());
  const int l = wavesim::cur->lane, j = l & 31, ih = l >> 5;
  wavesim_bf16x8 bj[2];
  for (int kg = 0; kg < 2; ++kg) memcpy(&bj[kg], wavesim::xget(j + 32 * kg) + 16, 16);
  for (int r = 0; r < 16; ++r) {
    const int i = (r >> 2) * 8 + ih * 4 + (r & 3);
    float acc = c[r];
    for (int kg = 0; kg < 2; ++kg) {
      wavesim_bf16x8 ai;
      memcpy(&ai, wavesim::xget(i + 32 * kg), 16);
      for (int e = 0; e < 8; ++e) acc += (float)ai[e] * (float)bj[kg][e];
    }
    c[r] = acc;
  }
  return c;
}
// v_mfma_f32_32x32x2_f32: A[i][k] from lane i + 32 * k; B[k][j] from lane j + 32 * k; D as above.  Exact fp32 fma chain.
static inline __attribute__((noinline)) wavesim_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, wavesim_f32x16 c, int, int, int) {
  unsigned char* x = wavesim::xput();
  memcpy(x, &a, 4);
  memcpy(x + 4, &b, 4);
  wavesim::wave_rendezvous(WAVESIM_SITE());
  const int l = wavesim::cur->lane, j = l & 31, ih = l >> 5;
  float bj[2];
  for (int k = 0; k < 2; ++k) memcpy(&bj[k], wavesim::xget(j + 32 * k) + 4, 4);
  for (int r = 0; r < 16; ++r) {
    const int i = (r >> 2) * 8 + ih * 4 + (r & 3);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float ai;
      memcpy(&ai, wavesim::xget(i + 32 * k), 4);
      acc = fmaf(ai, bj[k], acc);
    }
    c[r] = acc;
  }
  return c;
}
