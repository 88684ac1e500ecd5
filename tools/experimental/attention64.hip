// Fused attention, 64 query rows per wave: the opt-in experimental form of attention.hip for non-causal shapes
// (caco_set_attention64(1) / CACO_ATTN64=1), built around what the counters and ablations of that kernel showed
// (DESIGN.md 4.2):
//   * co-resident waves barely overlap each other's MFMA and softmax phases (3, 2, 1 workgroups per CU: same time, -16 %
//     at 1): the kernel runs at the SUM of its matrix and vector work, so the overlap has to be arranged INSIDE a wave;
//   * a lone wave issues one vector instruction per ~5 ns whatever its type; two waves per SIMD are needed for the VALU
//     rate.
// So: every wave owns TWO independent 32-row query blocks (A, B); 8 waves = 2 per SIMD = all 512 query rows of one
// (clip, head) per workgroup, every K / V tile fetched once; block B's MFMAs are interleaved with block A's softmax
// and vice versa (sched_group_barrier pins MFMA : fragment-read : VALU patterns, fragments one step ahead); Q is staged
// in LDS once (DMA, swizzled like K) and its fragments re-read per tile; the output leaves through an LDS staging tile as
// whole rows.  Built with -mllvm -amdgpu-mfma-vgpr-form (cacophony_amd/build.py): with the default accumulation-register
// form the compiler shuttles O (online rescale) or S (softmax input) between the two register files, 380-430 copy
// instructions per tile.
//
// Two softmax forms:
//   default            single pass, online softmax (running maximum, conditional O rescale between two regions);
//   -DATTN64_TWO_PASS  exact two-pass: pass 1 row maxima only, pass 2 exp2 with the FINAL maximum - no rescale, no
//                      branch, P <= 1 always; +50 % Q.K^T MFMAs.
// Measured at the encoder shape (B 256, S 500, 8 x 96): default kernel (attention.hip) 385-415 us, this one 394 us
// single-pass / 460 us two-pass (box to box +-5 %): a draw, which is why it is not the default.  Per-region cycle
// stamps (-DATTN_TIMING, tools/attn64_timing.py) show where it goes: the Q.K^T regions take 2-3 x their MFMA time (18
// LDS fragment reads per 12 MFMAs with all 8 waves in the same phase), the second wave of each SIMD runs up to 1.7 x
// slower than the first in the mixed regions, and the first then waits ~15 k of its 78 k cycles at the tile barriers.
//
// Everything else is attention.hip's design: swapped products so that softmax is lane-local, the key permutation
// that makes P an MFMA operand without leaving registers, K / V tiles of 64 keys by 16-byte LDS-DMA into unpadded rows
// (K chunk-swizzled on the source address), V through ds_read_b64_tr_b16, fp32 statistics.
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int KT = 64;         // keys per tile

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int key_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int VP>
__device__ __forceinline__ bf16x8 v_frag_tr(const char* p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * VP));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int HD, int NW>
__device__ __forceinline__ void attention64_body(const bf16_t* __restrict__ qp_, int q_ld, int Sq, const bf16_t* __restrict__ kv, int ld,
                                                 int k_off, int v_off, const float* __restrict__ key_mask, int S, int heads,
                                                 bf16_t* __restrict__ out, float scale_log2, char* smem) {
  constexpr int QB = NW * 64;                  // query rows per workgroup
  constexpr int RP = HD * 2;                   // K and V row pitch in LDS = the unpadded row (192 / 128 bytes)
  constexpr int VP = RP;
  constexpr int KCH = HD / 8;                  // 16-byte chunks per K / V row
  constexpr int NPC = KT * RP / 1024;          // 1 KiB DMA pieces per operand tile (12 / 8)
  constexpr int PPW = 2 * NPC / NW;            // pieces per wave and tile, K and V pieces numbered together
  static_assert((2 * NPC) % NW == 0, "pieces must divide evenly over the waves");
  constexpr int KS = HD / 16;                  // MFMA k-steps over the head dim
  constexpr int DT = HD / 32;                  // 32-row output tiles over the head dim
  constexpr int K_BYTES = KT * RP, V_BYTES = KT * RP, BUF = K_BYTES + V_BYTES + KT * 4 + 16;
  constexpr int OP = RP + 16;                  // staging pitch of the output rows (conflict-free ds_write_b64)

  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int H = heads * HD;
  const int64_t row_base = (int64_t)b * S, qrow_base = (int64_t)b * Sq;
  const bf16_t* q_base = qp_ + qrow_base * q_ld + h * HD;
  const bf16_t* kv_base = kv + row_base * ld + h * HD;
  const int q0 = qb * QB + wave * 64;          // block A = rows q0 .. q0+31, block B = q0+32 .. q0+63
#ifdef ATTN64_PRIO
  // the second wave of every SIMD (waves NW/2 ..) measured up to 1.7x slower than the first in the mixed regions
  // (oldest-first arbitration): give it the higher issue priority.  Measured with 1 and 3: no effect on the kernel time.
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(ATTN64_PRIO);
#endif

  // Q block of this wave (64 rows x HD) -> its own LDS region by DMA, in the K tile's swizzled row format; the B-operand
  // fragments are re-read from there every tile instead of occupying 2 * KS * 4 registers for the whole kernel (with
  // them resident the kernel needs > 256 architectural VGPRs and the compiler shuttles the O accumulators between
  // the two register files around the softmax's VALU work: 380 copy instructions per tile)
  char* qs = smem + 2 * BUF + wave * (64 * RP);
  {
    const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc((void*)q_base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < KCH; ++i) {            // 64 rows x KCH chunks = KCH pieces of 1 KiB
      const int L = i * 64 + lane;
      const int r = L / KCH, pos = L % KCH;
      const int src_chunk = (pos & ~3) | ((pos & 3) ^ ((r >> 2) & 3));
      const int row = min(q0 + r, Sq - 1);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(qr, (lds_vptr)(qs + i * 1024), 16, row * q_ld * 2 + src_chunk * 16, 0, 0, 0);
    }
  }
  const int qx = (l31 >> 2) & 3;
  const char* q_lane = qs + l31 * RP;
  const int ntiles = (S + KT - 1) / KT;

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)kv_base, 0, 0x7fffffff, 0x00020000);
  // piece p = wave + i * NW of a tile: p < NPC is K piece p, else V piece p - NPC.  A piece covers LDS bytes
  // [pc*1024, +1024) of its operand image = linear 16-byte chunks pc*64 + lane -> row L / KCH, chunk position L % KCH;
  // the K source chunk is un-swizzled from the position.
  int d_row[PPW], d_col[PPW], d_dst[PPW];
  bool d_isv[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pcs = wave + i * NW;
    const bool is_v = pcs >= NPC;
    const int pc = is_v ? pcs - NPC : pcs;
    const int L = pc * 64 + lane;
    const int r = L / KCH, pos = L % KCH;
    d_row[i] = r;
    d_col[i] = is_v ? (v_off + pos * 8) * 2 : (k_off + (((pos & ~3) | ((pos & 3) ^ ((r >> 2) & 3))) * 8)) * 2;
    d_dst[i] = (is_v ? K_BYTES : 0) + pc * 1024;
    d_isv[i] = is_v;
  }
  float mreg = 1.f;
  // tile t -> ring slot `buf`: K (+ V when with_v) by DMA; wave 0 also fetches the tile's key mask (consumed in finish_tile,
  // so that no wait on it lands here and drains the DMA just issued)
  auto issue_tile = [&](int t, int buf, bool with_v) {
#ifdef ATTN_SAMETILE
    t = 0;
#endif
    const int key0 = t * KT;
    char* kb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int rowoff = min(key0 + d_row[i], S - 1) * ld * 2;
      if (with_v || !d_isv[i])       // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(kb + d_dst[i]), 16, rowoff + d_col[i], 0, 0, 0);
    }
    if (tid < KT) {
      const int key = key0 + tid;
      mreg = (key < S) ? (key_mask ? key_mask[row_base + key] : 1.f) : 0.f;
    }
  };
  auto finish_tile = [&](int buf) {      // wave 0: per-key additive mask + "this tile has a masked key" flag
    if (tid < KT) {
      float* bias = reinterpret_cast<float*>(smem + buf * BUF + K_BYTES + V_BYTES);
      const float breg = mreg != 0.f ? 0.f : -INFINITY;
      bias[tid] = breg;
      const unsigned long long any = __ballot(breg != 0.f);
      if (tid == 0) reinterpret_cast<int*>(bias + KT)[0] = any != 0ull;
    }
  };
  auto tile_k = [&](int buf) { return smem + buf * BUF; };
  auto tile_v = [&](int buf) { return smem + buf * BUF + K_BYTES; };
  auto tile_bias = [&](int buf) { return reinterpret_cast<const float*>(smem + buf * BUF + K_BYTES + V_BYTES); };
  auto tile_padded = [&](int buf) { return __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(tile_bias(buf) + KT)[0]) != 0; };

  const int v_lane = (8 * hf + ((lane & 15) >> 2)) * VP + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
  const int kx = (key_perm(l31) >> 2) & 3;
  const int k_lane = key_perm(l31) * RP;

  // ---- phases ------------------------------------------------------------------------------------------------------
  // S^T = K Q^T of one query block (2 * KS MFMAs): two accumulator chains, fragments fetched one k-step ahead
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto qk = [&](int sb, const char* kb, f32x16 (&s)[2]) {
    const char* kr = kb + k_lane;
    bf16x8 kf[2][2], qv[2];
    auto load = [&](int ks, int bufi) {
      const int c = ks * 2 + hf;
      const int coff = ((c & ~3) | ((c & 3) ^ kx)) << 4;
      const int qoff = ((c & ~3) | ((c & 3) ^ qx)) << 4;
#pragma unroll
      for (int st = 0; st < 2; ++st) kf[bufi][st] = *reinterpret_cast<const bf16x8*>(kr + st * 32 * RP + coff);
      qv[bufi] = *reinterpret_cast<const bf16x8*>(q_lane + sb * 32 * RP + qoff);
    };
    load(0, 0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) load(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int st = 0; st < 2; ++st)
        s[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks & 1][st], qv[ks & 1], ks == 0 ? zero16 : s[st], 0, 0, 0);
    }
  };
  // key-padding mask of a tile on one block's scores: s[st][g*8 + e] is key st*32 + 16*g + 8*hf + e
  auto add_mask = [&](const float* bias, f32x16 (&s)[2]) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int kl = st * 32 + 16 * g + 8 * hf;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + kl);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + kl + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[st][g * 8 + e] += (e < 4 ? b0[e] : b1[e - 4]);
      }
  };
  auto tile_max = [&](const f32x16 (&s)[2], float (&m)[4]) {  // this lane's keys only, four independent chains; lanes l and
#pragma unroll                                               // l^32 and the chains are combined after pass 1
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) m[(st * 16 + r) & 3] = fmaxf(m[(st * 16 + r) & 3], s[st][r]);
  };
  // P = exp2(S * scale - max * scale) with the final row maximum: one packed FMA + two v_exp per pair, running row sum
  auto exp_step = [&](const f32x16 (&s)[2], float neg, f32x2 (&ps2)[2], bf16x8 (&pf)[4]) {
    const f32x2 sc2 = {scale_log2, scale_log2}, neg2 = {neg, neg};
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 sv = {s[st][g * 8 + e], s[st][g * 8 + e + 1]};
          const f32x2 x = __builtin_elementwise_fma(sv, sc2, neg2);
          const f32x2 p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
          ps2[(e >> 1) & 1] += p;
          pf[st * 2 + g][e] = (bf16_t)p[0];
          pf[st * 2 + g][e + 1] = (bf16_t)p[1];
        }
  };
  f32x16 o[2][DT];
  // O^T += V^T P^T of one query block (4 * DT MFMAs): DT accumulator chains, fragments one ahead
  auto pv = [&](int sb, const char* vb, const bf16x8 (&pf)[4], bool first) {
    bf16x8 vf[2];
    vf[0] = v_frag_tr<VP>(vb + v_lane);
#pragma unroll
    for (int i = 0; i < 4 * DT; ++i) {
      const int sp = i / DT, dt = i % DT;
      if (i + 1 < 4 * DT) vf[(i + 1) & 1] = v_frag_tr<VP>(vb + v_lane + ((i + 1) % DT) * 64 + ((i + 1) / DT) * 16 * VP);
      o[sb][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i & 1], pf[sp], (first && sp == 0) ? zero16 : o[sb][dt], 0, 0, 0);
    }
  };
  // Interleave of a region that holds one block's MFMAs and the OTHER block's exponentials (group counts are exact:
  // an unsatisfiable pipeline makes the scheduler drop it).  QK: 3 fragment reads per k-step, fetched one step ahead;
  // PV: 2 transpose reads per MFMA, one MFMA ahead; NV vector / transcendental fillers in every MFMA's shadow.
#define A64_SGB_QK(NV)                                                                      \
  __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                        \
  _Pragma("unroll") for (int n_ = 0; n_ < 2 * KS; ++n_) {                                   \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
    if (n_ < 2 * KS - 2 && (n_ & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     \
    if (n_ < 2 * KS - 2 && !(n_ & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);                                     \
  }
#define A64_SGB_PV(NV)                                                                      \
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                        \
  _Pragma("unroll") for (int n_ = 0; n_ < 4 * DT; ++n_) {                                   \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                      \
    if (n_ < 4 * DT - 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);                                     \
  }

#ifdef ATTN_TIMING
  unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
#define A64_STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define A64_STAMP(k)
#endif
#ifndef ATTN64_TWO_PASS
  // ---- single pass, online softmax (default) ----------------------------------------------------------------------------
  // One block's step: tile maximum, new running maximum, P = exp2((S - max) * scale), row sum; branch-free (it shares a
  // scheduling region with the other block's MFMAs).  Returns the factor the block's O must be multiplied by before its
  // next P.V and whether any row's maximum moved (the caller rescales only then, between two regions).
  auto softmax_step = [&](const f32x16 (&s)[2], float& m_run, float& l_acc, bf16x8 (&pf)[4], float& alpha, bool& moved) {
    float m4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    tile_max(s, m4);
    float m_tile = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    f32x2 ps2[2] = {{0.f, 0.f}, {0.f, 0.f}};
    exp_step(s, -m_use * scale_log2, ps2, pf);
    moved = __ballot(m_new > m_run) != 0ull;
    alpha = __builtin_amdgcn_exp2f((m_run - m_use) * scale_log2);     // m_run = -inf -> 0; unchanged maximum -> 1
    l_acc = l_acc * alpha + ((ps2[0][0] + ps2[0][1]) + (ps2[1][0] + ps2[1][1]));
    m_run = m_new;
  };
  auto rescale = [&](int sb, float alpha) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[sb][dt][r] *= alpha;
  };
#pragma unroll
  for (int sb = 0; sb < 2; ++sb)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[sb][dt][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  f32x16 sA[2], sB[2];
  bf16x8 pA[4], pB[4];
  float alphaA, alphaB;
  bool movedA, movedB;
  issue_tile(0, 0, true);
  finish_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile 0 and this wave's Q block
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) issue_tile(t + 1, buf ^ 1, true);
    const bool padded = tile_padded(buf);
    // R1: QK_A
    qk(0, tile_k(buf), sA);
    A64_SGB_QK(0)
    __builtin_amdgcn_sched_barrier(0);
    A64_STAMP(3)
    if (padded) add_mask(tile_bias(buf), sA);
    // R2: QK_B || softmax_A
    qk(1, tile_k(buf), sB);
    softmax_step(sA, m_run[0], l_run[0], pA, alphaA, movedA);
    A64_SGB_QK(8)
    // (use the products here: otherwise the compiler sinks the exponentials into the next region, away from these MFMAs)
    asm volatile("" ::"v"(pA[0]), "v"(pA[1]), "v"(pA[2]), "v"(pA[3]), "v"(l_run[0]), "v"(alphaA));
    A64_STAMP(4)
    if (movedA) rescale(0, alphaA);
    if (padded) add_mask(tile_bias(buf), sB);
    // R3: PV_A || softmax_B
    pv(0, tile_v(buf), pA, false);
    softmax_step(sB, m_run[1], l_run[1], pB, alphaB, movedB);
    A64_SGB_PV(8)
    asm volatile("" ::"v"(pB[0]), "v"(pB[1]), "v"(pB[2]), "v"(pB[3]), "v"(l_run[1]), "v"(alphaB));
    A64_STAMP(5)
    if (movedB) rescale(1, alphaB);
    // R4: PV_B
    pv(1, tile_v(buf), pB, false);
    A64_SGB_PV(0)
    A64_STAMP(6)
    if (t + 1 < ntiles) finish_tile(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of tile t+1 have landed
    A64_STAMP(7)
    __syncthreads();
    A64_STAMP(8)
  }
#else
  // ---- pass 1: exact row maxima ----------------------------------------------------------------------------------------
  float mA4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, mB4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  f32x16 sA[2], sB[2];
  issue_tile(0, 0, false);
  finish_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile 0 and this wave's Q block
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    // the last tile of pass 1 already fetches pass 2's first tile (K and V) into the free slot
    // (ONE call site: with two, the mask value of issue_tile becomes a phi whose copy makes the compiler wait for the
    // load - and with it for the DMA just issued - right here)
    const bool last = t + 1 >= ntiles;
    issue_tile(last ? 0 : t + 1, buf ^ 1, last);
    A64_STAMP(9)
    const bool padded = tile_padded(buf);
    qk(0, tile_k(buf), sA);
    A64_SGB_QK(0)
    __builtin_amdgcn_sched_barrier(0);
    A64_STAMP(3)
    if (padded) add_mask(tile_bias(buf), sA);
    qk(1, tile_k(buf), sB);
    tile_max(sA, mA4);                                   // next to block B's MFMAs
    A64_SGB_QK(2)
    __builtin_amdgcn_sched_barrier(0);
    A64_STAMP(4)
    if (padded) add_mask(tile_bias(buf), sB);
    tile_max(sB, mB4);
    A64_STAMP(0)
    finish_tile(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A64_STAMP(1)
    __syncthreads();
    A64_STAMP(2)
  }
  float mA = fmaxf(fmaxf(mA4[0], mA4[1]), fmaxf(mA4[2], mA4[3])), mB = fmaxf(fmaxf(mB4[0], mB4[1]), fmaxf(mB4[2], mB4[3]));
  mA = fmaxf(mA, __shfl_xor(mA, 32, 64));
  mB = fmaxf(mB, __shfl_xor(mB, 32, 64));
  const float negA = (mA == -INFINITY) ? 0.f : -mA * scale_log2;      // a fully masked row: exp2(-inf) = 0 everywhere, output 0
  const float negB = (mB == -INFINITY) ? 0.f : -mB * scale_log2;

  // ---- pass 2: P and O ---------------------------------------------------------------------------------------------------
  f32x2 lA[2] = {{0.f, 0.f}, {0.f, 0.f}}, lB[2] = {{0.f, 0.f}, {0.f, 0.f}};
  bf16x8 pA[4], pB[4];
  for (int t = 0; t < ntiles; ++t) {
    const int buf = (ntiles + t) & 1;
    if (t + 1 < ntiles) issue_tile(t + 1, buf ^ 1, true);
    const bool padded = tile_padded(buf);
    // R1: QK_A
    qk(0, tile_k(buf), sA);
    A64_SGB_QK(0)
    __builtin_amdgcn_sched_barrier(0);
    A64_STAMP(3)
    if (padded) add_mask(tile_bias(buf), sA);
    // R2: QK_B || exp_A
    qk(1, tile_k(buf), sB);
    exp_step(sA, negA, lA, pA);
    A64_SGB_QK(6)
    // (use the products here: otherwise the compiler sinks the exponentials into the next region, away from these MFMAs)
    asm volatile("" ::"v"(pA[0]), "v"(pA[1]), "v"(pA[2]), "v"(pA[3]), "v"(lA[0]), "v"(lA[1]));
    A64_STAMP(4)
    __builtin_amdgcn_sched_barrier(0);
    if (padded) add_mask(tile_bias(buf), sB);
    // R3: PV_A || exp_B
    pv(0, tile_v(buf), pA, t == 0);
    exp_step(sB, negB, lB, pB);
    A64_SGB_PV(6)
    asm volatile("" ::"v"(pB[0]), "v"(pB[1]), "v"(pB[2]), "v"(pB[3]), "v"(lB[0]), "v"(lB[1]));
    A64_STAMP(5)
    __builtin_amdgcn_sched_barrier(0);
    // R4: PV_B
    pv(1, tile_v(buf), pB, t == 0);
    A64_SGB_PV(0)
    A64_STAMP(6)
    if (t + 1 < ntiles) finish_tile(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of tile t+1 have landed
    A64_STAMP(7)
    __syncthreads();
    A64_STAMP(8)
  }
  float l_run[2];
  l_run[0] = (lA[0][0] + lA[0][1]) + (lA[1][0] + lA[1][1]);
  l_run[1] = (lB[0][0] + lB[0][1]) + (lB[1][0] + lB[1][1]);
#endif

#undef A64_SGB_QK
#undef A64_SGB_PV
  // Epilogue: normalise, stage this wave's 64 x HD block in LDS (the K / V ring is dead: the loop's last barrier is
  // behind every wave), write whole rows.  Lane (l31, hf) holds query row l31, columns dt*32 + g*8 + 4*hf .. +3.
  char* stage = smem + wave * (64 * OP);
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const float l_tot = l_run[sb] + __shfl_xor(l_run[sb], 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[sb][dt][g * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(stage + (sb * 32 + l31) * OP + (dt * 32 + g * 8 + 4 * hf) * 2) = v;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-private staging: no barrier needed
  const int rows_valid = min(64, Sq - q0);               // may be <= 0 for a wave past the end
  const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(
      out + (qrow_base + q0) * H + h * HD, 0, rows_valid > 0 ? (rows_valid - 1) * H * 2 + RP : 0, 0x00020000);
#pragma unroll
  for (int it = 0; it < KCH; ++it) {                     // 64 rows x KCH chunks = KCH wave instructions of 16 B per lane
    const int L = it * 64 + lane;
    const int r = L / KCH, c = L % KCH;
    const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * OP + c * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, out_r, r * H * 2 + c * 16, 0, 0);
  }
#ifdef ATTN_TIMING
  A64_STAMP(9)
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 3 && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(out + (int64_t)gridDim.z * Sq * H) + wave * 16;   // past the logical end (tools/attn64_timing.py over-allocates)
    for (int k = 0; k < 10; ++k) dbg[k] = tacc[k];
  }
#endif
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64, 1)        // 8 waves = 2 per SIMD: register budget 256, all architectural (build.py: MFMA VGPR form)
void attention64_kernel(const bf16_t* __restrict__ q, int q_ld, int Sq, const bf16_t* __restrict__ kv, int ld, int k_off, int v_off,
                        const float* __restrict__ key_mask, int S, int heads, bf16_t* __restrict__ out, float scale_log2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attention64_body<HD, NW>(q, q_ld, Sq, kv, ld, k_off, v_off, key_mask, S, heads, out, scale_log2, smem);
}

template <int HD>
int launch64(const bf16_t* q, int q_ld, int seq_q, const bf16_t* kv, int ld, int k_off, int v_off, const float* key_mask, int batch,
             int seq, int heads, bf16_t* out, hipStream_t st) {
  constexpr int NW = 8;                               // 512 query rows per workgroup, two waves per SIMD
  constexpr int RP = HD * 2, BUF = 2 * KT * RP + KT * 4 + 16;     // = attention64_body's BUF
  constexpr int SMEM = 2 * BUF + NW * 64 * RP;        // K/V ring of 2 + Q blocks (the output staging reuses both, >= NW*64*(RP+16))
  static_assert(SMEM >= NW * 64 * (RP + 16), "output staging must fit");
  static bool attr_done = false;
  if (!attr_done) {
    CACO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention64_kernel<HD, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    attr_done = true;
  }
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)HD);
  const dim3 grid((seq_q + NW * 64 - 1) / (NW * 64), heads, batch);
  hipLaunchKernelGGL((attention64_kernel<HD, NW>), grid, dim3(NW * 64), SMEM, st, q, q_ld, seq_q, kv, ld, k_off, v_off, key_mask, seq,
                     heads, out, scale_log2);
  return check_hip(hipGetLastError(), "attention64 launch");
}

}  // namespace

// Same contract as attention_qkv (attention.hip) without the causal option.
int attention64(const bf16_t* q, int q_ld, int seq_q, const bf16_t* kv, int ld, int k_off, int v_off, const float* key_mask,
                int batch, int seq, int heads, int head_dim, bf16_t* out, hipStream_t st) {
  if (head_dim == 96) return launch64<96>(q, q_ld, seq_q, kv, ld, k_off, v_off, key_mask, batch, seq, heads, out, st);
  if (head_dim == 64) return launch64<64>(q, q_ld, seq_q, kv, ld, k_off, v_off, key_mask, batch, seq, heads, out, st);
  set_error("attention64: head_dim %d not in {64, 96}", head_dim);
  return CACO_ERR_INVALID;
}

}  // namespace caco
