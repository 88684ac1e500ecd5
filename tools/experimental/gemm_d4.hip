// gemm_bf16_d4: 256x128x32 bf16 MFMA GEMM, persistent, TWO independent 4-wave workgroups per CU.
//   out[M,N] = epilogue( A[M,K] (bf16) x W[N,K]^T (bf16) ), fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// Why: in the one-workgroup-per-CU 256x256 kernel (gemm_w4.hip, w8) all eight waves of a CU reach the
// end of a tile's K-loop together, so the epilogue's vector work (bias, SiLU: ~10 issue slots per output element), its
// LDS transposition and its 128-256 KiB of stores run with the matrix pipe idle: 18-27 % of the kernel at K = 768, and
// every CU of the chip enters that phase at about the same time.  Here a CU hosts two workgroups that share nothing
// (80 KiB of LDS and one wave per SIMD each).  They are started half a tile apart, so while one is in its epilogue the
// other one's wave has the SIMD's matrix pipe to itself and runs its K-loop at up to twice the shared rate: the
// epilogue's VALU / LDS / store work hides under MFMAs of the partner, and a barrier or DMA stall of one workgroup is
// covered by the other.  The price is operand reuse: 85 FLOP per byte pulled from L2 instead of 128.
//
//   tile      256 (M) x 128 (N), 4 waves as 2 x 2, wave tile 128 x 64 = 4 x 2 MFMA 32x32x16 blocks (128 accumulators)
//   LDS       K-slots of 32: A ring 3 x 16 KiB (256 rows x 64 B), W ring 3 x 8 KiB = 72 KiB -> two workgroups per CU
//   loads     buffer_load_dwordx4 ... lds (LDS-DMA), A three K-slots ahead, W two; ONE counted s_waitcnt vmcnt(6) and ONE
//             s_barrier per K-slot, never vmcnt(0); loads stay in flight across barriers, tile boundaries and epilogues
//   LDS image lane-linear (DMA constraint); 16-byte chunk c of row r lives at c ^ ((r >> 2) & 3): applied on the per-lane
//             SOURCE address and undone on the ds_read_b128 side; every ds_read_b128 lane group then touches 16 distinct
//             16-byte slots of the 256-byte bank row
//   K-loop    rotated: body(g) = ks0 | vmcnt(6) lgkmcnt(0) s_barrier | ks1; fragments double-buffered in registers one
//             16-deep step ahead (across K-slots and output tiles)
//   epilogue  gemm_w8_epilogue.h (same wave tile as w8): bias / SiLU / erf-GELU / residual / LayerNorm-fold forms,
//             32-row slabs transposed through the A slot of the K-slot just consumed
//   schedule  persistent; XCD x owns a contiguous run of tiles (n fastest); the second half of an XCD's workgroups (by
//             dispatch order the second resident of each CU) starts half a tile late
//
// Reference ops replaced: nn.Linear + activation + residual add (audio_models/mae.py:55-61,69-74,92-97,133;
// text_models/roberta.py:62-64,110,153,164; caco.py:35-37).
#include "common.h"
#include "kernels.h"
#include "gemm_w8_epilogue.h"

namespace caco {
namespace {

constexpr int DBK = 32;                        // K-slot, bf16 elements
constexpr int DROWB = DBK * 2;                 // 64 bytes per row per K-slot
constexpr int D_ASLOT = 256 * DROWB;           // 16 KiB
constexpr int D_WSLOT = 128 * DROWB;           // 8 KiB
constexpr int D_AOFF = 0;
constexpr int D_WOFF = 3 * D_ASLOT;
constexpr int D_SMEM = 3 * D_ASLOT + 3 * D_WSLOT;   // 73728 = 72 KiB

typedef __attribute__((address_space(3))) void* lds_vptr;

#define D4_SGB_MFMA 0x008
#define D4_SGB_VMEM 0x010
#define D4_SGB_DSRD 0x100

struct D4CurA {
  __amdgpu_buffer_rsrc_t r;
  int voff[4];        // per-lane byte offsets of this wave's four 16-row groups (rows clamped to the last valid row)
  int li, kt;
};
struct D4CurW {
  __amdgpu_buffer_rsrc_t r;
  int voff;           // 16-row group 0; group `it` adds 64 rows through the scalar offset
  int li, kt;
};

__device__ __forceinline__ void d4_decode(int t, int tiles_n, int& tm, int& tn) {
  tm = t / tiles_n;
  tn = t - tm * tiles_n;
}

// One DMA piece = one wave instruction = 16 rows x 64 B.  Row r = (it * 4 + wave) * 16 + (lane >> 2); (r >> 2) & 3 is
// lane >> 4 for every piece, so the source chunk of LDS chunk position (lane & 3) is (lane & 3) ^ (lane >> 4).
__device__ __forceinline__ void d4_setup_a(D4CurA& C, const GemmArgs& p, int t, int tiles_n, int lda, int wave, int lane) {
  int tm, tn;
  d4_decode(t, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256;
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * lda), 0, 0x7fffffff, 0x00020000);
  const int r16 = wave * 16 + (lane >> 2);
  const int chunk = (lane & 3) ^ (lane >> 4);
  const int last = (int)min((int64_t)256, p.M - m0) - 1;
#pragma unroll
  for (int it = 0; it < 4; ++it) C.voff[it] = min(it * 64 + r16, last) * lda * 2 + chunk * 16;
}
__device__ __forceinline__ void d4_setup_w(D4CurW& C, const GemmArgs& p, int t, int tiles_n, int ldw, int wave, int lane) {
  int tm, tn;
  d4_decode(t, tiles_n, tm, tn);
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)tn * 128 * ldw), 0, 0x7fffffff, 0x00020000);
  const int r16 = wave * 16 + (lane >> 2);
  const int chunk = (lane & 3) ^ (lane >> 4);
  C.voff = r16 * ldw * 2 + chunk * 16;
}
__device__ __forceinline__ void d4_piece_a(const D4CurA& C, int it, char* slot, int wave) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.r, (lds_vptr)(slot + (it * 4 + wave) * 1024), 16, C.voff[it], C.kt * DROWB, 0, 0);
}
__device__ __forceinline__ void d4_piece_w(const D4CurW& C, int it, int ldw, char* slot, int wave) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.r, (lds_vptr)(slot + (it * 4 + wave) * 1024), 16, C.voff, C.kt * DROWB + it * 64 * ldw * 2, 0, 0);
}

__device__ __forceinline__ bf16x8 d4_frag(const char* oper, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(oper + row * DROWB + ((chunk ^ ((row >> 2) & 3)) << 4));
}

// One 16-deep step: 8 MFMAs on the CURRENT fragment set while the NEXT set is read (6 ds_read_b128) and up to four DMA
// pieces are issued; reads and DMA pieces alternate in source order, the group barriers spread them over the MFMAs.
#define D4_STEP(XN, WN, XA, WW, KS, XC, WC, DMA0, DMA1, DMA2, DMA3)                               \
  XN[0] = d4_frag(XA, 0 * 32 + frow, (KS) * 2 + fhalf);                                          \
  WN[0] = d4_frag(WW, 0 * 32 + frow, (KS) * 2 + fhalf);                                          \
  DMA0;                                                                                          \
  XN[1] = d4_frag(XA, 1 * 32 + frow, (KS) * 2 + fhalf);                                          \
  DMA1;                                                                                          \
  WN[1] = d4_frag(WW, 1 * 32 + frow, (KS) * 2 + fhalf);                                          \
  XN[2] = d4_frag(XA, 2 * 32 + frow, (KS) * 2 + fhalf);                                          \
  DMA2;                                                                                          \
  XN[3] = d4_frag(XA, 3 * 32 + frow, (KS) * 2 + fhalf);                                          \
  DMA3;                                                                                          \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], XC[0], acc[0][0], 0, 0, 0);         \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], XC[1], acc[1][0], 0, 0, 0);         \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], XC[0], acc[0][1], 0, 0, 0);         \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], XC[1], acc[1][1], 0, 0, 0);         \
  acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], XC[2], acc[2][0], 0, 0, 0);         \
  acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], XC[2], acc[2][1], 0, 0, 0);         \
  acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[0], XC[3], acc[3][0], 0, 0, 0);         \
  acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WC[1], XC[3], acc[3][1], 0, 0, 0);         \
  _Pragma("unroll") for (int n_ = 0; n_ < 8; ++n_) {                                             \
    __builtin_amdgcn_sched_group_barrier(D4_SGB_MFMA, 1, 0);                                     \
    if (n_ < 6) __builtin_amdgcn_sched_group_barrier(D4_SGB_DSRD, 1, 0);                         \
    if (n_ == 1 || n_ == 2 || n_ == 4 || n_ == 5) __builtin_amdgcn_sched_group_barrier(D4_SGB_VMEM, 1, 0); \
  }                                                                                              \
  __builtin_amdgcn_sched_barrier(0);

template <int EPI, int ACT, int MODE>
__device__ __forceinline__ void d4_body(const GemmArgs& p, char* smem, int stagger) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;

  const int tiles_n = p.N / 128;
  const int tiles_m = (int)((p.M + 255) / 256);
  const int nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int cnt = q + (xcd < r ? 1 : 0);
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  if (slot >= cnt) return;
  const int nk = p.K / DBK;

  // The second resident workgroup of a CU starts half a tile late (units of 1024 cycles): from then on one workgroup's
  // epilogue falls into the middle of the other's K-loop.  A wrong guess about who shares a CU costs speed only.
  if (stagger > 0 && slot >= (slots >> 1)) {
    for (int z = 0; z < stagger; ++z) __builtin_amdgcn_s_sleep(16);
  }

  const int frow = lane & 31, fhalf = lane >> 5;
  const int x_off = wm * 128 * DROWB, w_off = wn * 64 * DROWB;

  // Past the last K-slot of its last tile a cursor keeps re-reading that tile (valid memory, into slots nobody reads).
  D4CurA CA;
  D4CurW CW;
  CA.li = CW.li = slot;
  CA.kt = CW.kt = 0;
  d4_setup_a(CA, p, base + slot, tiles_n, lda, wave, lane);
  d4_setup_w(CW, p, base + slot, tiles_n, ldw, wave, lane);
  auto advance_a = [&]() {
    if (++CA.kt == nk) {
      CA.kt = 0;
      if (CA.li + slots < cnt) { CA.li += slots; d4_setup_a(CA, p, base + CA.li, tiles_n, lda, wave, lane); }
    }
  };
  auto advance_w = [&]() {
    if (++CW.kt == nk) {
      CW.kt = 0;
      if (CW.li + slots < cnt) { CW.li += slots; d4_setup_w(CW, p, base + CW.li, tiles_n, ldw, wave, lane); }
    }
  };

  // ring state: slots of K-slots g, g+1, g+2 (A: g+2 in flight; W: the third slot is free until ks0 issues W(g+2) into it)
  int a_c = D_AOFF, a_1 = D_AOFF + D_ASLOT, a_2 = D_AOFF + 2 * D_ASLOT;
  int w_c = D_WOFF, w_1 = D_WOFF + D_WSLOT, w_2 = D_WOFF + 2 * D_WSLOT;

  // prologue: A(0) W(0) | A(1) W(1) | A(2)
#pragma unroll
  for (int it = 0; it < 4; ++it) d4_piece_a(CA, it, smem + a_c, wave);
  advance_a();
#pragma unroll
  for (int it = 0; it < 2; ++it) d4_piece_w(CW, it, ldw, smem + w_c, wave);
  advance_w();
#pragma unroll
  for (int it = 0; it < 4; ++it) d4_piece_a(CA, it, smem + a_1, wave);
  advance_a();
#pragma unroll
  for (int it = 0; it < 2; ++it) d4_piece_w(CW, it, ldw, smem + w_1, wave);
  advance_w();
#pragma unroll
  for (int it = 0; it < 4; ++it) d4_piece_a(CA, it, smem + a_2, wave);
  advance_a();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 x0[4], w0[2], x1[4], w1[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) x0[i] = d4_frag(smem + a_c + x_off, i * 32 + frow, fhalf);
#pragma unroll
  for (int j = 0; j < 2; ++j) w0[j] = d4_frag(smem + w_c + w_off, j * 32 + frow, fhalf);

  // Vector-memory operations retire in issue order: everything the first barrier after an epilogue waits for was issued
  // before that epilogue's stores, so (fp32 + residual epilogue) it may leave them in flight: vmcnt(6 + 4 + NST).
  constexpr int NST = (EPI == EPI_BF16) ? 16 : 32;      // global stores per wave and epilogue (full tile)
  bool stores_pending = false;
  int c_li = slot;
  while (true) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // body(g) = ks0 | wait + barrier | ks1.  LAST = a tile's last K-slot: its ks1 issues no A(g+3), because the A slot
    // it would overwrite becomes the epilogue's transpose slab first.
#define D4_BODY(LAST)                                                                                                      \
    {                                                                                                                      \
      const char* xa = smem + a_c + x_off;                                                                                 \
      const char* ww = smem + w_c + w_off;                                                                                 \
      /* ks0: compute (g,0), read (g,1); W(g+2) -> the free W slot */                                                      \
      D4_STEP(x1, w1, xa, ww, 1, x0, w0, d4_piece_w(CW, 0, ldw, smem + w_2, wave), (void)0,                                \
              d4_piece_w(CW, 1, ldw, smem + w_2, wave), (void)0)                                                           \
      advance_w();                                                                                                         \
      /* A(g+1), W(g+1) have landed once at most A(g+2) (4) + W(g+2) (2) are outstanding - plus, right after an epilogue, */ \
      /* the epilogue's stores (issued between A(1) W(1) and A(2) W(2)) */                                                 \
      if (stores_pending) {                                                                                                \
        if (EPI == EPI_F32 && p.xb_out) asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");                       \
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(6 + NST) : "memory");                                     \
        stores_pending = false;                                                                                            \
      } else {                                                                                                             \
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                                                        \
      }                                                                                                                    \
      __builtin_amdgcn_s_barrier();                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      /* ks1: compute (g,1), read (g+1,0) (possibly of the next output tile); A(g+3) -> the slot of A(g), free since the */ \
      /* barrier */                                                                                                        \
      if constexpr (!(LAST)) {                                                                                             \
        D4_STEP(x0, w0, smem + a_1 + x_off, smem + w_1 + w_off, 0, x1, w1, d4_piece_a(CA, 0, smem + a_c, wave),            \
                d4_piece_a(CA, 1, smem + a_c, wave), d4_piece_a(CA, 2, smem + a_c, wave), d4_piece_a(CA, 3, smem + a_c, wave)) \
        advance_a();                                                                                                       \
      } else {                                                                                                             \
        D4_STEP(x0, w0, smem + a_1 + x_off, smem + w_1 + w_off, 0, x1, w1, (void)0, (void)0, (void)0, (void)0)             \
      }                                                                                                                    \
      { const int t_ = a_c; a_c = a_1; a_1 = a_2; a_2 = t_; }                                                              \
      { const int t_ = w_c; w_c = w_1; w_1 = w_2; w_2 = t_; }                                                              \
    }
    for (int kt = 0; kt + 1 < nk; ++kt) D4_BODY(false)
    D4_BODY(true)
    int tm_, tn_;
    d4_decode(base + c_li, tiles_n, tm_, tn_);
    // a_2 = the A slot of the K-slot just consumed: nothing has been issued into it yet
    w8_epilogue<EPI, ACT, MODE>(acc, p, (int64_t)tm_ * 256, tn_ * 128, wm, wn, lane, smem + a_2 + wave * 4096);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // nobody may DMA into the slab slot while another wave still transposes through it
    c_li += slots;
    if (c_li >= cnt) break;
    // the A K-slot that the tile's last ks1 did not issue
#pragma unroll
    for (int it = 0; it < 4; ++it) d4_piece_a(CA, it, smem + a_2, wave);
    advance_a();
    // the next tile's first fragments are re-read here (the last ks1 already fetched them once): this way they are not
    // live across the epilogue, which needs the registers
#pragma unroll
    for (int i = 0; i < 4; ++i) x0[i] = d4_frag(smem + a_c + x_off, i * 32 + frow, fhalf);
#pragma unroll
    for (int j = 0; j < 2; ++j) w0[j] = d4_frag(smem + w_c + w_off, j * 32 + frow, fhalf);
    // order at the next barrier: [A(1) W(1)] stores x NST [A(2)] [W(2)]; strict form: vmcnt(6) retires the stores too
#ifdef D4_RELAX_BF16
    stores_pending = (MODE != 0 || !p.xb_out == !p.stats_part);
#else
    stores_pending = (EPI == EPI_F32) && (MODE != 0 || !p.xb_out == !p.stats_part);
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup's LDS allocation
}

template <int EPI, int ACT, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_bf16_d4_kernel(GemmArgs p, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  d4_body<EPI, ACT, MODE>(p, smem, stagger);
}

template <int EPI, int ACT, int MODE>
int launch_d4(const GemmArgs& p, hipStream_t st) {
  void (*kern)(GemmArgs, int) = gemm_bf16_d4_kernel<EPI, ACT, MODE>;
  int dev = 0;
  CACO_HIP(hipGetDevice(&dev));
  static bool attr_done[16] = {};
  static int num_cu[16] = {};
  if (dev < 0 || dev >= 16) dev = 0;
  if (!attr_done[dev]) {
    CACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, D_SMEM));
    attr_done[dev] = true;
  }
  if (!num_cu[dev]) {
    hipDeviceProp_t prop;
    CACO_HIP(hipGetDeviceProperties(&prop, dev));
    num_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int tiles = (int)((p.M + 255) / 256) * (p.N / 128);
  const int resident = 2 * num_cu[dev];
  const int grid = tiles < resident ? (tiles + 7) / 8 * 8 : resident / 8 * 8;
  // half a tile in units of 1024 cycles: a tile keeps a SIMD's matrix pipe busy for nk * 16 MFMAs * 32 cycles per wave,
  // two waves share the pipe
  static const int env_s = getenv("CACO_D4_STAGGER") ? atoi(getenv("CACO_D4_STAGGER")) : -1;
  const int nk = p.K / DBK;
  const int stagger = tiles <= grid / 2 ? 0 : (env_s >= 0 ? env_s * nk / 64 : nk / 2);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), D_SMEM, st, p, stagger);
  return check_hip(hipGetLastError(), "gemm_bf16_d4 launch");
}

}  // namespace

bool gemm_bf16_d4_ok(const GemmArgs& p, int epi) {
  return p.N % 128 == 0 && p.K % DBK == 0 && p.K >= 2 * DBK &&
         (int64_t)256 * (p.lda ? p.lda : p.K) * 2 < 0x7fffffff && (int64_t)128 * (p.ldw ? p.ldw : p.K) * 2 < 0x7fffffff;
}

int gemm_bf16_d4(const GemmArgs& p, int epi, int act, hipStream_t st) {
  CACO_REQUIRE(gemm_bf16_d4_ok(p, epi), "gemm_bf16_d4: shape not supported");
  static const bool generic = getenv("CACO_W8_GENERIC") && atoi(getenv("CACO_W8_GENERIC"));
  const bool plain = !generic && p.bias && !p.fold_mr && !p.xb_out && !p.stats_part;
  if (plain && epi == EPI_BF16 && !p.resid) {
    if (act == ACT_NONE) return launch_d4<EPI_BF16, ACT_NONE, 1>(p, st);
    if (act == ACT_SILU) return launch_d4<EPI_BF16, ACT_SILU, 1>(p, st);
    if (act == ACT_GELU) return launch_d4<EPI_BF16, ACT_GELU, 1>(p, st);
  }
  if (plain && epi == EPI_F32 && act == ACT_NONE) {
    if (p.resid) return launch_d4<EPI_F32, ACT_NONE, 2>(p, st);
    return launch_d4<EPI_F32, ACT_NONE, 1>(p, st);
  }
  // LayerNorm-folded stack (api.hip run_audio_layers): consumer = bias + fold, producer = bias + residual + bf16 copy + sums
  if (!generic && p.bias && p.fold_mr && !p.resid && !p.xb_out && !p.stats_part && epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_d4<EPI_BF16, ACT_NONE, 3>(p, st);
    if (act == ACT_SILU) return launch_d4<EPI_BF16, ACT_SILU, 3>(p, st);
  }
  if (!generic && p.bias && p.resid && p.xb_out && p.stats_part && !p.fold_mr && epi == EPI_F32 && act == ACT_NONE)
    return launch_d4<EPI_F32, ACT_NONE, 4>(p, st);
  if (epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_d4<EPI_BF16, ACT_NONE, 0>(p, st);
    if (act == ACT_SILU) return launch_d4<EPI_BF16, ACT_SILU, 0>(p, st);
    if (act == ACT_GELU) return launch_d4<EPI_BF16, ACT_GELU, 0>(p, st);
  } else if (epi == EPI_F32 && act == ACT_NONE) {
    return launch_d4<EPI_F32, ACT_NONE, 0>(p, st);
  }
  set_error("gemm_bf16_d4: unsupported epilogue %d / activation %d", epi, act);
  return CACO_ERR_INVALID;
}

}  // namespace caco
