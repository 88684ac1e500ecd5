// gemm_bf16_w8: 256x256x64 bf16 MFMA GEMM, EIGHT waves per workgroup (2 (M) x 4 (N), wave tile 128 x 64 = two waves per
// SIMD), one persistent workgroup per CU.  The default kernel for chip-filling shapes.
//   out[M,N] = epilogue( A[M,K] (bf16) x W[N,K]^T (bf16) ), fp32 accumulate on v_mfma_f32_16x16x32_bf16 (8 x 4 blocks per
//   wave; chosen over 32x32x16 for its energy per FLOP: these GEMMs run at the socket power cap).
//
//   LDS (160 KiB = the whole CU)   A ring: 3 slots x 32 KiB (256 rows x 128 B), W ring: 2 slots x 32 KiB.
//             The activation operand streams from HBM and is prefetched TWO K-tiles ahead; the weight operand is
//             L2-resident and is prefetched one K-tile ahead.  The slot of the K-tile just consumed doubles as the
//             epilogue's transpose slab (4 KiB per wave).
//   loads     buffer_load_dwordx4 ... lds (16-byte DMA, no VGPR round trip), spread over the K-tile, retired by ONE counted
//             s_waitcnt vmcnt(4) per K-tile: never 0 in the loop, loads stay in flight across the barrier and across
//             output-tile boundaries (the next tile's first K-tiles land while this tile's epilogue runs)
//   LDS image lane-linear (DMA constraint); the bank swizzle (16-byte chunk c ^ ((row >> 1) & 7)) is applied to the
//             per-lane SOURCE offset and undone on the ds_read_b128 side
//   K-loop    rotated, four phases of 16 MFMAs per 64-deep K-tile: P0 P1 P2 | vmcnt(4) lgkmcnt(0) s_barrier | P3 (see
//             w16_body).  Fragments are double-buffered in registers one phase ahead (also across K-tiles and output
//             tiles): two X sets and two W sets of four 16-row fragments.  A wave that is stalled (DMA issue, fragment
//             wait, barrier) leaves the SIMD's matrix pipe to its partner; nothing forces their phase.
//   MFMA      operands swapped (weights = A operand) so that a lane owns 4 CONSECUTIVE n of one output row m
//   epilogue  gemm_w8_epilogue.h: bias / SiLU / erf-GELU / residual / LayerNorm-fold forms in registers; 32-row slabs
//             transposed through the wave's XOR-swizzled LDS slab so that every global access is a whole 128-byte row
//             segment; residual rows are fetched a slab ahead
//   schedule  persistent; XCD x owns a contiguous run of tiles (n fastest), so the tiles that share an A row panel
//             run together on one XCD's L2
//
// Anatomy (measured on the 32x32x16 form in round 2, tools/w8_timing.py): per 256x256 tile at K = 768 the K-loop takes ~28 k cycles,
// the epilogue 9.7 k (bf16) .. 11.6 k (SiLU) .. 44 k (fp32 + residual) with the matrix pipe idle; the epilogue is bound by
// the CU's own issue / store path, not by HBM.  Ablation builds: -DW4_NOEPI, -DW4_NODMA, -DW4_NOREADS
// (tools/build_variant.sh, tools/energy_ab.sh).  Retired siblings: the 32x32x16 form of this kernel with its timing stamps
// (tools/experimental/gemm_w8_mf32.hip), 4-wave 128x128-per-wave form, phased p8, two-accumulator v8, two-workgroups-per-CU
// d4, skewed-row-group s8 (git history / tools/experimental/).
//
// Reference ops replaced: nn.Linear + activation + residual add (audio_models/mae.py:55-61,69-74,92-97,133;
// text_models/roberta.py:62-64,110,153,164; caco.py:35-37).
#include "common.h"
#include "kernels.h"
#include "gemm_w8_common.h"
#include "gemm_w8_epilogue.h"

namespace caco {
namespace {

// ---- K-loop ------------------------------------------------------------------------------------------------------------
// The wave's 128 x 64 part is 8 x 4 blocks of v_mfma_f32_16x16x32_bf16.  On random operand bits the 16x16x32 instruction
// draws less power per FLOP than 32x32x16 (tools/mfma_power.py: 2.04 vs 1.82 PFLOP/s for register-resident loops at the
// throttle point), and that, not a pipe, is what limits these GEMMs.
// A 64-deep K-tile = two 32-deep steps s0, s1, each split by X row blocks into two PHASES of 16 MFMAs:
//   P0: X(s0, blocks 0..3) x W(s0)    reads X(s0, 4..7), W(s1, 0..1)
//   P1: X(s0, 4..7) x W(s0)           reads X(s1, 0..3), W(s1, 2..3)        A(g+2) pieces 0..1
//   P2: X(s1, 0..3) x W(s1)           reads X(s1, 4..7)                     A(g+2) pieces 2..3
//   -- vmcnt(4) lgkmcnt(0) s_barrier: every read of A(g), W(g) is done; A(g+1), W(g+1) have landed --
//   P3: X(s1, 4..7) x W(s1)           reads X'(s0, 0..3), W'(s0) of K-tile g+1   W(g+2) pieces 0..3 -> slot of W(g)
// Fragment registers: two X sets and two W sets of four 16-row fragments (64 VGPRs) + 128 accumulators.
#define W16_MFMAS(XC, WC, IB)                                                                               \
  _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                          \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                          \
    acc[(IB) * 4 + q_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WC[j_], XC[q_], acc[(IB) * 4 + q_][j_], 0, 0, 0);
// the same with C = 0 (inline constant): the first touch of an accumulator block in an output tile
#define W16_MFMAS_Z(XC, WC, IB)                                                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                          \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                          \
    acc[(IB) * 4 + q_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WC[j_], XC[q_], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
// interleave of one phase: 16 MFMAs, the first NRD of them followed by a fragment read, VMEM after the MFMAs in VM_MASK
#define W16_SCHED(NRD, VM_MASK)                                                                             \
  _Pragma("unroll") for (int n_ = 0; n_ < 16; ++n_) {                                                       \
    __builtin_amdgcn_sched_group_barrier(W4_SGB_MFMA, 1, 0);                                                \
    if (n_ < (NRD)) __builtin_amdgcn_sched_group_barrier(W4_SGB_DSRD, 1, 0);                                \
    if (((VM_MASK) >> n_) & 1) __builtin_amdgcn_sched_group_barrier(W4_SGB_VMEM, 1, 0);                     \
  }                                                                                                         \
  __builtin_amdgcn_sched_barrier(0);

template <int EPI, int ACT, int MODE>
__device__ __forceinline__ void w16_body(const GemmArgs& p, char* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lda = p.lda ? p.lda : p.K, ldw = p.ldw ? p.ldw : p.K;

  const int tiles_n = p.N / 256;
  const int tiles_m = (int)((p.M + 255) / 256);
  const int nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int cnt = q + (xcd < r ? 1 : 0);
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  if (slot >= cnt) return;
  const int nk = p.K / WBK;

  const int l16 = lane & 15, lq = lane >> 4;
  const int x_off = wm * 128 * WROWB, w_off = wn * 64 * WROWB;

  W4CurA<8> CA;
  W4CurW CW;
  CA.li = CW.li = slot;
  CA.kt = CW.kt = 0;
  w4_setup_a<8>(CA, p, base + slot, tiles_n, lda, wave, lane);
  w4_setup_w(CW, p, base + slot, tiles_n, ldw, wave, lane);
  auto advance_a = [&]() {
    if (++CA.kt == nk) {
      CA.kt = 0;
      if (CA.li + slots < cnt) { CA.li += slots; w4_setup_a<8>(CA, p, base + CA.li, tiles_n, lda, wave, lane); }
    }
  };
  auto advance_w = [&]() {
    if (++CW.kt == nk) {
      CW.kt = 0;
      if (CW.li + slots < cnt) { CW.li += slots; w4_setup_w(CW, p, base + CW.li, tiles_n, ldw, wave, lane); }
    }
  };

  int a_c = W_AOFF, a_1 = W_AOFF + W_SLOT, a_2 = W_AOFF + 2 * W_SLOT;
  int w_c = W_WOFF, w_1 = W_WOFF + W_SLOT;

  // prologue: A(0) W(0) | A(1) W(1)
#pragma unroll
  for (int it = 0; it < 4; ++it) w4_piece_a<8>(CA, it, smem + a_c, wave);
  advance_a();
#pragma unroll
  for (int it = 0; it < 4; ++it) w4_piece_w<8>(CW, it, ldw, smem + w_c, wave);
  advance_w();
#pragma unroll
  for (int it = 0; it < 4; ++it) w4_piece_a<8>(CA, it, smem + a_1, wave);
  advance_a();
#pragma unroll
  for (int it = 0; it < 4; ++it) w4_piece_w<8>(CW, it, ldw, smem + w_1, wave);
  advance_w();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // fragment of block `blk` (16 rows), 32-deep step s: row blk*16 + l16, 16-byte chunk s*4 + lq
#define W16_F(OPER, BLK, S) w4_frag(OPER, (BLK) * 16 + l16, (S) * 4 + lq)
#define W16_RD(DST, EXPR) if (W4_DO_READS) DST = EXPR;      // -DW4_NOREADS: the K-loop without its fragment reads (ablation)
  bf16x8 xa[4], xb[4], wc[4], wn_[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[i] = W16_F(smem + a_c + x_off, i, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) wc[j] = W16_F(smem + w_c + w_off, j, 0);

  constexpr int NST = (EPI == EPI_BF16) ? 16 : 32;      // global stores per wave and epilogue (full tile)
  bool stores_pending = false;
  int c_li = slot;
  while (true) {
    f32x4 acc[8][4];
#ifdef W8_CLASSIC
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
#endif

#ifndef W8_CLASSIC
    // The first K-tile of an output tile is peeled: its first-touch MFMAs take C = 0 as an inline constant, so the accumulators
    // are never zeroed by VALU moves.  Left as one loop over zero-initialised accumulators, hipcc 7.2 emits 128 v_mov for the
    // zeroing plus 128 more at the tile-loop header (loop-carried copies) per wave and output tile in the bf16-epilogue kernels
    // - 268 instructions next to an issue-bound epilogue (profiles/r4_cpu/epilogue_budget.txt; for the fp32-epilogue kernels it
    // peels by itself).  Default since round 4 on the static count and bitwise-equal simulator results; NOT yet timed: the
    // variant build `classic` (-DW8_CLASSIC: one loop, scalar bias epilogue) is the other arm of the A/B (tools/gpu_session.sh).
    {
#define W16_MF_FIRST W16_MFMAS_Z
#include "gemm_w8_ktile.inc"
#undef W16_MF_FIRST
    }
    for (int kt = 1; kt < nk; ++kt) {
#define W16_MF_FIRST W16_MFMAS
#include "gemm_w8_ktile.inc"
#undef W16_MF_FIRST
    }
#else
    for (int kt = 0; kt < nk; ++kt) {
#define W16_MF_FIRST W16_MFMAS
#include "gemm_w8_ktile.inc"
#undef W16_MF_FIRST
    }
#endif
    const int t = base + c_li;
    int tm_, tn_;
    w4_decode(t, tiles_n, tiles_m, p.ngroup, tm_, tn_);
    const int64_t m_cur = (int64_t)tm_ * 256;
    const int n_cur = tn_ * 256;
#ifdef W4_NOEPI
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
    w16_epilogue<EPI, ACT, MODE>(acc, p, m_cur, n_cur, wm, wn, lane, smem + a_2 + wave * 4096);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // nobody may DMA into the slab slot while another wave still transposes through it
    // the next tile's first fragments are re-read here: they are not live across the epilogue, which needs the registers
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = W16_F(smem + a_c + x_off, i, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) wc[j] = W16_F(smem + w_c + w_off, j, 0);
    stores_pending = (EPI == EPI_F32) && (MODE != 0 || !p.xb_out == !p.stats_part);
#endif
    c_li += slots;
    if (c_li >= cnt) break;
  }
}

template <int EPI, int ACT, int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_w8_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  w16_body<EPI, ACT, MODE>(p, smem);
}

template <int EPI, int ACT, int MODE = 0>
int launch_w8(const GemmArgs& p, hipStream_t st) {
  void (*kern)(GemmArgs) = gemm_bf16_w8_kernel<EPI, ACT, MODE>;
  int num_cu = 0;
  CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(kern), W_SMEM, &num_cu));
  const int tiles = (int)((p.M + 255) / 256) * (p.N / 256);
  const int grid = tiles < num_cu ? (tiles + 7) / 8 * 8 : num_cu / 8 * 8;
  GemmArgs q = p;
  {   // n-tiles per L2 group (w4_decode).  An XCD's 32 workgroups run 32 consecutive tiles at a time; with n fastest over
      // all of N the weight rows they touch (N x K bf16: 4.7 MB for fc1) exceed the XCD's 4 MiB L2 and are re-streamed
      // through it once per M panel.  Groups of 4 n-tiles at K <= 1024 keep the group's weights (1.5 MB) plus the eight
      // A panels in flight (3.1 MB) resident: fabric reads of one fc1 launch 1.15 -> 0.81 GB by the TCC counters
      // (profiles/r2_v3), time -0.5 %.  Bytes are what the power cap prices, so it is on by default since
      // round 3 (NOT re-timed at HEAD: no GPU since); switch SW_W_NGROUP / CACO_W_NGROUP=<n> forces a group width, 0 restores
      // one group.  A caller's explicit p.ngroup (the ping-pong traversal asks for one group) overrides the switch.
    const int env_g = sw(SW_W_NGROUP);
    const int tiles_n = p.N / 256;
    int g = tiles_n;
    if (tiles_n > 4 && p.K <= 1024) {           // equal groups where N allows: fc1 12 -> 3 x 4, QKV 9 -> 3 x 3
      g = 4;
      for (int d = 4; d >= 2; --d)
        if (tiles_n % d == 0) { g = d; break; }
    }
    if (env_g == 0) g = tiles_n;
    else if (env_g > 0) g = env_g < tiles_n ? env_g : tiles_n;
    if (p.ngroup > 0) g = p.ngroup < tiles_n ? p.ngroup : tiles_n;      // the caller's choice (ping-pong traversal: one group)
    q.ngroup = p.reverse ? -g : g;       // the kernels read the direction from the sign (w4_decode)
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), W_SMEM, st, q);
  return check_hip(hipGetLastError(), "gemm_bf16_w8 launch");
}

#ifdef W8_F32_SKEW      // variant build `skew` only (tools/build_variants.sh): the fp32 + residual epilogue under the K-loop
#include "gemm_w8_skew.inc"
#endif

}  // namespace

bool gemm_bf16_w8_ok(const GemmArgs& p, int epi) {
  return p.N % 256 == 0 && p.K % WBK == 0 && p.K >= WBK &&
         (int64_t)256 * (p.lda ? p.lda : p.K) * 2 < 0x7fffffff && (int64_t)256 * (p.ldw ? p.ldw : p.K) * 2 < 0x7fffffff;
}

int gemm_bf16_w8(const GemmArgs& p, int epi, int act, hipStream_t st) {
  CACO_REQUIRE(gemm_bf16_w8_ok(p, epi), "gemm_bf16_w8: shape not supported");
  // compile-time specialised epilogues (w8_epilogue MODE): 1 bias only, 2 bias + residual, 3 bias + LayerNorm-fold consumer,
  // 4 bias + residual + fold producer (bf16 copy + row sums), 5 bias + gathered residual; 0 = generic (tested at run time)
  constexpr bool generic = false;
  const bool plain_args = p.bias && !p.fold_mr && !p.xb_out && !p.stats_part;
  const bool plain = !generic && plain_args;
  if (plain && epi == EPI_BF16 && !p.resid) {
    if (act == ACT_NONE) return launch_w8<EPI_BF16, ACT_NONE, 1>(p, st);
    if (act == ACT_SILU) return launch_w8<EPI_BF16, ACT_SILU, 1>(p, st);
    if (act == ACT_GELU) return launch_w8<EPI_BF16, ACT_GELU, 1>(p, st);
  }
  if (p.resid_idx) {
    CACO_REQUIRE(plain_args && epi == EPI_F32 && act == ACT_NONE && p.resid, "gemm_bf16_w8: a gathered residual needs bias, a residual table and the plain fp32 epilogue");
    CACO_REQUIRE((int64_t)p.ldc * 4 * 65536 < 0x7fffffff, "gemm_bf16_w8: residual table rows too long");
    return launch_w8<EPI_F32, ACT_NONE, 5>(p, st);
  }
  if (plain && epi == EPI_F32 && act == ACT_NONE) {
#ifdef W8_F32_SKEW
    if (p.resid) {
      int num_cu = 0;
      CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(&gemm_bf16_w8s_kernel<false>), W_SMEM, &num_cu));
      if (gemm_bf16_w8s_ok(p, num_cu / 8 * 8)) return launch_w8s<false>(p, st);
    }
#endif
    if (p.resid) return launch_w8<EPI_F32, ACT_NONE, 2>(p, st);
    return launch_w8<EPI_F32, ACT_NONE, 1>(p, st);
  }
  if (!generic && p.bias && p.fold_mr && !p.resid && !p.xb_out && !p.stats_part && epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_w8<EPI_BF16, ACT_NONE, 3>(p, st);
    if (act == ACT_SILU) return launch_w8<EPI_BF16, ACT_SILU, 3>(p, st);
  }
  if (!generic && p.bias && p.resid && p.xb_out && p.stats_part && !p.fold_mr && epi == EPI_F32 && act == ACT_NONE) {
#ifdef W8_F32_SKEW      // the LayerNorm-fold producer under the K-loop as well (api.hip linear_resid_stats, CACO_LN_FOLD=1)
    int num_cu = 0;
    CACO_TRY_RC(prepare_launch(reinterpret_cast<const void*>(&gemm_bf16_w8s_kernel<true>), W_SMEM, &num_cu));
    if (gemm_bf16_w8s_ok(p, num_cu / 8 * 8)) return launch_w8s<true>(p, st);
#endif
    return launch_w8<EPI_F32, ACT_NONE, 4>(p, st);
  }
  if (epi == EPI_BF16) {
    if (act == ACT_NONE) return launch_w8<EPI_BF16, ACT_NONE>(p, st);
    if (act == ACT_SILU) return launch_w8<EPI_BF16, ACT_SILU>(p, st);
    if (act == ACT_GELU) return launch_w8<EPI_BF16, ACT_GELU>(p, st);
  } else if (epi == EPI_F32 && act == ACT_NONE) {
    return launch_w8<EPI_F32, ACT_NONE>(p, st);
  }
  set_error("gemm_bf16_w8: unsupported epilogue %d / activation %d", epi, act);
  return CACO_ERR_INVALID;
}

}  // namespace caco
