// NOT BUILT.  out-proj + residual + LayerNorm (ln2) as ONE launch that owns whole 64 x 768 row blocks - built, verified
// (x bit-identical to the w8 GEMM; h bit-identical to layernorm_kernel when both use ln_row.h with explicit roundings) and
// integrated behind a per-model knob at the end of round 2, then parked: it does not win inside the step.
//   isolated, M = 126976 (rowln_bench.py):   w8 GEMM + layernorm 345 us      this kernel 301-305 us
//   inside the step (bench.py stages):       gemm_out 2.91 + ln2 1.22 = 4.13 ms      gemm_out_ln 4.19 ms, and fc1 +0.16 ms
//   (h is no longer the last thing written before fc1 reads it: it falls out of the Infinity Cache) -> 28.99 vs 28.86 ms.
// Why it cannot go much further (rowln_timing.py stamps, PMC): the whole weight matrix streams through LDS once per 64-row
// block = 2.3 GB of L2 -> LDS traffic per launch; with the weights re-packed slab-major (whole cache lines per DMA piece;
// the plain [N][K] layout was 364 us) the K-loop runs at 1.9 k cycles per 32-deep slab = 13 TB/s of L2 -> LDS traffic
// chip-wide, which is the fabric's limit, against 0.8 k cycles of MFMA work.  128-row blocks would halve it but need 192
// accumulator registers per lane; 96-row blocks leave 1323 blocks for 256 CUs (14 % tail).  LDS holds only a ring of three
// 48 KiB W slabs, so 24 barriers per block with both waves of a SIMD in lock step.
// Files next to this one: ln_row.h (the shared row code), integration.diff (api.hip / header / binding / tests / bench),
// common_h.diff (wave_sum_dpp, measured neutral), rowln_bench.py, rowln_timing.py, rowln_bitcmp.py.
//
// gemm_resid_ln: residual GEMM with the FOLLOWING LayerNorm in its epilogue, one pass over the rows:
//   x[M, N] (fp32, in place)  = x + A[M, K] (bf16) . W[N, K]^T (bf16) + bias
//   h[M, N] (bf16)            = LayerNorm(x) * gamma + beta
// for N = 768 (the audio encoder's hidden size) and small K (the attention out-projection, K = 768).
//
// Replaces, per encoder layer, `x = x + attn.out_proj(o)` followed by `ln2(x)` (audio_models/mae.py:92-95): today one
// persistent 256x256 GEMM launch (0.243 ms: 975 MB of HBM traffic, HBM-bound through its fp32 read-modify-write epilogue)
// plus one LayerNorm launch (0.103 ms: the same 390 MB of x read again, 195 MB of h written).  LayerNorm needs whole rows;
// a 256-wide output tile only sees a third of a row, which is why folding LayerNorm into the w8 kernel's epilogue (partial
// statistics, a bf16 copy, a finalize kernel, the consumer GEMM normalising in ITS epilogue) measured a wash twice.
// Here one workgroup owns WHOLE rows: a 64-row x 768-column block, so the new rows are normalised while they are still
// on chip and x is read once and written once: 1170 MB instead of 1560 MB per layer for the pair.
//
// The price is operand reuse: the whole weight matrix (N x K bf16 = 1.2 MB, L2-resident) streams through LDS once per
// 64-row block - 2.3 GB of L2 -> LDS traffic per launch, the same order as the w8 fc1 launch sustains (4.7 GB in 0.58 ms).
// That is affordable because the kernel is HBM-bound with a 2.5x margin on its matrix work (150 GFLOP per launch); it
// would not be for fc2 (K = 3072: 9.4 GB).
//
//   workgroup  8 waves (1 x 8 over N), wave tile 64 rows x 96 columns = 2 x 3 blocks of v_mfma_f32_32x32x16_bf16
//              (96 accumulators; operands swapped as in gemm_w8: a lane owns 4 consecutive n of one output row)
//   K-loop     K-slabs of 32: W slab 768 rows x 64 B = 48 KiB, A slab 64 rows x 64 B = 4 KiB, rings of three, by LDS-DMA
//              (buffer_load_dwordx4 ... lds), two slabs ahead; every wave issues exactly 7 pieces per slab, so "slab s has
//              landed" is the counted wait vmcnt(7) (VMEM retires in order) and ONE barrier per slab.  (The first version
//              had one slab in flight and vmcnt(0): 24 exposed L2 latencies per block, 400 us per launch.)
//   residual   the 8 x 3 float4 of x this lane will need in the epilogue are requested right after the LAST slab's DMA, into
//              the registers the K-loop does not use; the remaining slab waits count them (vmcnt(31), vmcnt(24))
//   W layout   the kernel reads W RE-PACKED slab-major: Wp[K/32][N][32] (pack_rowln_weights, once per weight at load time), so
//              a DMA piece (16 rows x 64 B) is 1 KiB of consecutive memory = 8 whole cache lines.  With the plain [N][K]
//              layout every 64-byte row segment is its own request into a line 1536 bytes from the next one: the launch
//              was bound by L2 requests (9 TB/s of L2 -> LDS traffic, 364 us).
//   LDS image  lane-linear 64-byte rows; 16-byte chunk c of row r lives at c ^ ((r >> 2) & 3) (applied on the source
//              address, undone on the ds_read_b128 side: conflict-free fragment reads)
//   epilogue   per 32-row half: accumulators -> LDS as a [32][N + 4] fp32 stage (aliases the operand ring), then ROW-major:
//              each wave takes 4 rows, a lane 12 columns (3 x float4, fully coalesced): + residual + bias, store x,
//              ln_row() (the code of layernorm_kernel: same bits as the two-launch form), store h
#include "common.h"
#include "kernels.h"
#include "ln_row.h"

namespace caco {
namespace {

typedef __attribute__((address_space(3))) void* lds_vptr;

#ifndef RL_A_AUX
#define RL_A_AUX 0      // cache policy of the activation DMA (2 = nt measured slower)
#endif
#ifndef RL_X_NT
#define RL_X_NT 1       // non-temporal load / store of the residual rows
#endif
constexpr int RL_RM = 64;              // rows per workgroup
constexpr int RL_SL = 32;              // K-slab, elements
constexpr int RL_ROWB = RL_SL * 2;     // bytes per row per slab

struct RowLnArgs {
  const bf16_t* A;
  const bf16_t* W;      // packed: [K / 32][N][32]
  const float* bias;
  float* x;
  const float* gamma;
  const float* beta;
  bf16_t* h;
  int64_t M;
  int K, lda;
  float eps;
};

template <int N>
__device__ __forceinline__ void rowln_body(const RowLnArgs& p, char* smem) {
  constexpr int WSLAB = N * RL_ROWB, ASLAB = RL_RM * RL_ROWB;
  constexpr int WPIECES = WSLAB / 1024, APIECES = ASLAB / 1024;      // 48, 4
  constexpr int WPW = WPIECES / 8;                                   // W pieces per wave and slab (6)
  constexpr int SP = N + 4;                                          // stage pitch, floats
  static_assert(WPIECES % 8 == 0 && APIECES <= 8 && N % 256 == 0 && N / 8 == 96, "geometry");
  static_assert(32 * SP * 4 <= 3 * WSLAB + 3 * ASLAB, "the epilogue stage must fit in the operand ring");
  constexpr int NBUF = 3;
  char* const wbuf = smem;                       // [NBUF][WSLAB]
  char* const abuf = smem + NBUF * WSLAB;        // [NBUF][ASLAB]

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * RL_RM;
  const int rows = (int)min((int64_t)RL_RM, p.M - m0);

  // ---- DMA geometry: piece = 16 rows x 64 B; lane -> row piece*16 + lane/4, 16-byte position lane & 3 -------------------
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  int voff_w[WPW], voff_a;
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int r = (wave + 8 * i) * 16 + (lane >> 2), pos = lane & 3;
    voff_w[i] = r * RL_ROWB + ((pos ^ ((r >> 2) & 3)) << 4);             // inside a packed slab
  }
  {
    const int r = (wave & (APIECES - 1)) * 16 + (lane >> 2), pos = lane & 3;      // waves 4..7 repeat pieces 0..3: the same
    voff_a = min(r, rows - 1) * p.lda * 2 + ((pos ^ ((r >> 2) & 3)) << 4);      // bytes to the same place, 7 loads per wave
  }
  auto issue = [&](int s, int buf) {
#ifdef RL_NODMA
    return;
#endif
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_vptr)(wbuf + buf * WSLAB + (wave + 8 * i) * 1024), 16, voff_w[i] + s * WSLAB, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (lds_vptr)(abuf + buf * ASLAB + (wave & (APIECES - 1)) * 1024), 16,
                                             voff_a + s * RL_ROWB, 0, 0, RL_A_AUX);
  };
  constexpr int LPS = WPW + 1;                   // loads per slab and wave

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int kx = (l31 >> 2) & 3;
  const int w_lane = (wave * 96 + l31) * RL_ROWB, a_lane = l31 * RL_ROWB;
  const int ns = p.K / RL_SL;
  // residual rows of the epilogue: wave w normalises rows w*4 .. w*4+3 of each 32-row half, a lane columns lane*4 + 256 q.
  // The first half's rows are requested right behind the last slab's DMA (the slab waits count them), the second half's at
  // the start of the epilogue (into the registers the first half's accumulators leave behind), under the first half's work.
  f32x4 res[2][4][3];
  auto issue_resid = [&](int i) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int64_t m = min(m0 + i * 32 + wave * 4 + rr, p.M - 1);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#if RL_X_NT
        res[i][rr][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.x + m * N + lane * 4 + 256 * q));
#else
        res[i][rr][q] = *reinterpret_cast<const f32x4*>(p.x + m * N + lane * 4 + 256 * q);
#endif
    }
  };
  constexpr int NRES = 4 * 3;
#ifdef RL_TIMING     // s_memtime stamps past the end of h (tools: allocate 32 bytes per workgroup more): start, K-loop end, epilogue end
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.h + p.M * N) + (int64_t)blockIdx.x * 4;
  if (tid == 0) stamps[0] = __builtin_amdgcn_s_memtime();
#endif
  issue(0, 0);
  if (ns > 1) issue(1, 1);
  if (ns <= 2) issue_resid(0);
  for (int s = 0; s < ns; ++s) {
    // slab s has landed once only what was issued after it is outstanding: slab s+1 and, from the last DMA on, the residual
    if (s + 2 < ns) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else if (s + 1 < ns) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS + NRES) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NRES) : "memory");
    // a bare barrier: __syncthreads() carries a fence that drains vmcnt, i.e. the two slabs in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everybody's pieces of slab s are in; everybody is done with slab s-1
    const char* wb = wbuf + (s % NBUF) * WSLAB + w_lane;
    const char* ab = abuf + (s % NBUF) * ASLAB + a_lane;
    // fragment reads first, the DMA issue (~60 cycles of issue stall per instruction) under their latency
    bf16x8 af[2][2], wf[2][3];
    const int coff0 = ((0 + hf) ^ kx) << 4, coff1 = ((2 + hf) ^ kx) << 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(ab + i * 32 * RL_ROWB + coff0);
#pragma unroll
    for (int j = 0; j < 3; ++j) wf[0][j] = *reinterpret_cast<const bf16x8*>(wb + j * 32 * RL_ROWB + coff0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < ns) {
      issue(s + 2, (s + 2) % NBUF);
      if (s + 3 == ns) issue_resid(0);           // right behind the last DMA
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) af[1][i] = *reinterpret_cast<const bf16x8*>(ab + i * 32 * RL_ROWB + coff1);
#pragma unroll
    for (int j = 0; j < 3; ++j) wf[1][j] = *reinterpret_cast<const bf16x8*>(wb + j * 32 * RL_ROWB + coff1);
#ifndef RL_NOMFMA
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
#else
    acc[0][0][0] += (float)af[0][0][0] + (float)af[1][1][0] + (float)wf[0][0][0] + (float)wf[1][2][0];
#endif
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the ring is dead: it becomes the epilogue stage
#ifdef RL_TIMING
  if (tid == 0) stamps[1] = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue ----------------------------------------------------------------------------------------------------------
  // acc[i][j][g*4 + r] is row m = i*32 + l31, column n = wave*96 + j*32 + g*8 + hf*4 + r
  float* stage = reinterpret_cast<float*>(smem);
  f32x4 bia[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int col = lane * 4 + 256 * q;
    bia[q] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the previous half's rows have been read
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
        *reinterpret_cast<f32x4*>(stage + l31 * SP + wave * 96 + j * 32 + g * 8 + hf * 4) = v;
      }
    if (i == 0) issue_resid(1);                  // the second half's rows: two barriers and four rows of work ahead of their use
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // row-major: x = (acc + bias) + residual in the order the w8 epilogue uses, then ln_row() - the code layernorm_kernel
    // runs - so the fused launch and the two-launch form give the same bits (batch-split invariance stays exact)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int row = wave * 4 + rr;
      const int64_t m = m0 + i * 32 + row;
      if (m >= p.M) continue;                    // wave-uniform
      f32x4 v[MAXC];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int col = lane * 4 + 256 * q;
        v[q] = (*reinterpret_cast<const f32x4*>(stage + row * SP + col) + bia[q]) + res[i][rr][q];
#if RL_X_NT
        __builtin_nontemporal_store(v[q], reinterpret_cast<f32x4*>(p.x + m * N + col));
#else
        *reinterpret_cast<f32x4*>(p.x + m * N + col) = v[q];
#endif
      }
      ln_row(v, N / 4, lane, N, p.gamma, p.beta, p.eps, nullptr, p.h + m * N);
    }
  }
#ifdef RL_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) stamps[2] = __builtin_amdgcn_s_memtime();
#endif
}

template <int N>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(RowLnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char rowln_smem[];
  rowln_body<N>(p, rowln_smem);
}


__global__ void pack_rowln_kernel(const bf16_t* __restrict__ w, int N, int K, bf16_t* __restrict__ out) {
  // one thread per 16-byte chunk of the output: out[s][r][c*8 ..] = w[r][s*32 + c*8 ..]
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)N * K / 8;
  if (i >= total) return;
  const int c = (int)(i & 3);
  const int64_t rs = i >> 2;
  const int r = (int)(rs % N), s = (int)(rs / N);
  *reinterpret_cast<bf16x8*>(out + i * 8) = *reinterpret_cast<const bf16x8*>(w + (int64_t)r * K + s * RL_SL + c * 8);
}

}  // namespace

bool gemm_resid_ln_ok(int N, int K, int lda) { return N == 768 && K > 0 && K % RL_SL == 0 && lda % 8 == 0 && (int64_t)N * K * 2 < (1ll << 31); }

// w [N, K] bf16 -> the slab-major layout gemm_resid_ln reads ([K / 32][N][32], same size)
int pack_rowln_weights(const bf16_t* w, int N, int K, bf16_t* out, hipStream_t st) {
  CACO_REQUIRE(w && out && N > 0 && K > 0 && K % RL_SL == 0, "pack_rowln_weights: bad arguments (N=%d K=%d)", N, K);
  const int64_t total = (int64_t)N * K / 8;
  hipLaunchKernelGGL(pack_rowln_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, N, K, out);
  return check_hip(hipGetLastError(), "pack_rowln_weights launch");
}

int gemm_resid_ln(const bf16_t* A, int lda, const bf16_t* Wp, const float* bias, float* x, int64_t M, int N, int K,
                  const float* gamma, const float* beta, float eps, bf16_t* h, hipStream_t st) {
  if (lda <= 0) lda = K;
  CACO_REQUIRE(A && Wp && x && gamma && beta && h && M > 0, "gemm_resid_ln: null argument or empty shape");
  CACO_REQUIRE(gemm_resid_ln_ok(N, K, lda), "gemm_resid_ln: needs N == 768 and K %% 32 == 0 (got N=%d K=%d lda=%d)", N, K, lda);
  CACO_REQUIRE((M + RL_RM - 1) / RL_RM < (1ll << 31), "gemm_resid_ln: M too large");
  constexpr int LDS = 3 * 768 * RL_ROWB + 3 * RL_RM * RL_ROWB;
  void (*kern)(RowLnArgs) = gemm_rowln_kernel<768>;
  int num_cu = 0;
  CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(kern), LDS, &num_cu));
  RowLnArgs p{A, Wp, bias, x, gamma, beta, h, M, K, lda, eps};
  hipLaunchKernelGGL(kern, dim3((unsigned)((M + RL_RM - 1) / RL_RM)), dim3(512), LDS, st, p);
  return check_hip(hipGetLastError(), "gemm_resid_ln launch");
}

}  // namespace caco
