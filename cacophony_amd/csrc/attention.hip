// Fused (flash-style) multi-head self-attention for the two encoder stacks: never materialises
// the S x S score matrix.
//
//   audio: 8 heads x 96, S = 500 (496 valid), key-padding mask        (audio_models/mae.py:89-92,
//          torch.nn.MultiheadAttention with key_padding_mask, need_weights=False)
//   text : 12 heads x 64, T = 32, causal AND key-padding mask          (text_models/roberta.py:86-104,297-310)
//
// Layout contract (produced by ONE fused QKV GEMM): qkv[B*S, 3H] bf16, row = token, columns Q | K | V, head h
// owning the contiguous slice h*HD.. of each third.  All three operands are consumed in this natural layout:
// Q and K are k-contiguous MFMA operands as they are; V (key-major) is the transposed operand of P.V and is read
// from LDS with the gfx950 hardware transpose read (ds_read_b64_tr_b16), so no V^T copy is ever written.
//
// Work split: one workgroup = NW waves = 32*QR*NW query rows of one (clip, head); each wave owns QR blocks of 32 query
// rows and the full head dimension.  Every K / V fragment read from LDS is 1 KiB per wave and feeds one 32-cycle MFMA
// per query block: with QR = 1 the four SIMDs of a CU ask the LDS for exactly its 128 bytes/clock at full MFMA rate -
// the kernel is LDS-bound by construction.  QR = 2 (non-causal sequences longer than 128 rows: the audio encoder) uses
// each fragment for two MFMAs: half the LDS bytes per flop, 256 registers per wave, two workgroups (= two independent
// barrier domains) per CU.  QR = 1 (text, caption decoder): 164 registers, three workgroups per CU.
// K / V tiles of 64 keys go HBM/L2 -> LDS by 16-byte DMA (buffer_load ... lds: no staging registers, no ds_write),
// double-buffered, the next tile's DMA issued before the current tile's MFMAs.  The LDS images are lane-linear (a DMA
// constraint), i.e. unpadded 2*HD-byte rows: the V image is conflict-free as it is for the transpose reads; the K image
// is made conflict-free for ds_read_b128 by XOR-ing the low two bits of the 16-byte chunk index with (key >> 2) & 3 on
// the SOURCE address and again on the read address.
//
// Math per 64-key tile, all on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   S^T[key, q] = K Q^T      (operands swapped so that every lane owns ONE query column: the row-wise softmax is
//                             lane-local plus a single lane^32 exchange)
//   O^T[d, q]  += V^T P^T
// The MFMA row <-> key assignment of the first product is permuted (bits 2 and 3 swapped) so that the S^T
// accumulator registers of a lane are, in order, exactly the 8-key groups the P operand of the second MFMA wants:
// P never leaves registers and needs no cross-lane shuffle.
// Softmax statistics are fp32.  The exponent reference of a row is a LAZY running maximum of the raw scores: it only
// moves when the row maximum outgrows it by more than 2^8 (softmax is invariant to the reference; P <= 256 meanwhile), so
// the rescale of O runs about once per row block, not once per tile.  exp is exp2(fma(s, scale*log2e, -ref*scale*log2e)):
// one packed FMA per two scores + one v_exp per score.  Mask work (key padding, causal diagonal) is only executed for
// tiles that contain a masked key.
// A query row whose keys are all masked yields 0 (the reference yields NaN there; it cannot happen with
// right-padded inputs).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

#ifndef ATTN_ST_AUX
#define ATTN_ST_AUX 0      // cache policy of the output stores (2 = nt, 16 = sc1): A/B switch, tools/build_variant.sh
#endif

namespace caco {
namespace {

constexpr int KT = 64;         // keys per tile

typedef __attribute__((address_space(3))) void* lds_vptr;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ int key_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// 8 consecutive keys x one head-dim column per lane = A operand of O^T += V^T P^T, from the key-major V tile
template <int VP>
__device__ __forceinline__ bf16x8 v_frag_tr(const char* p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 4 * VP));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// (the body lives in a __device__ function: the buffer-descriptor builtins it uses are not visible to the host pass)
// ORD: the workgroup order is a run-time argument (ping-pong traversal, api.hip).  It is a template parameter so that the
// default kernels (ORD = false) keep the instruction streams that ran on hardware in round 2 (tools/isa_diff.sh).
template <int HD, bool CAUSAL, int NW, int QR, bool ORD>
__device__ __forceinline__ void attention_body(const bf16_t* __restrict__ qp_, int q_ld, int Sq, const bf16_t* __restrict__ kv, int ld,
                                               int k_off, int v_off, const float* __restrict__ key_mask, int S, int heads,
                                               bf16_t* __restrict__ out, float scale_log2, int kv_rows, int order) {
  constexpr int NT = NW * 64, QB = NW * 32 * QR;
  constexpr int RP = HD * 2;                   // K and V row pitch in LDS = the unpadded row (192 / 128 bytes)
  constexpr int VP = RP;
  constexpr int KCH = HD / 8;                  // 16-byte chunks per K / V row
  constexpr int NPC = KT * RP / 1024;          // 1 KiB DMA pieces per operand tile (12 / 8)
  constexpr int PPW = NPC / NW;                // pieces per wave
  static_assert(NPC % NW == 0, "pieces must divide evenly over the waves");
  constexpr int KS = HD / 16;                  // MFMA k-steps over the head dim
  constexpr int DT = HD / 32;                  // 32-row output tiles over the head dim
  constexpr int K_BYTES = KT * RP, V_BYTES = KT * RP, BUF = K_BYTES + V_BYTES + KT * 4 + 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup -> (query block, head, clip).  The hardware deals workgroups to the 8 XCDs round-robin in linear order, and
  // each XCD has its own L2.  All the workgroups of one clip read the same rows of the QKV buffer (a head's K / V slice is
  // 192 bytes of a 4.6 KB row: neighbouring heads share cache lines, query blocks share whole tiles), so they are given
  // consecutive slots of ONE XCD: clip b lives on XCD b % 8 and its rows cross the fabric once (round 2 counters: 91 % L2
  // misses and 1.3 GB fetched per launch for a 0.59 GB buffer with the plain mapping).
  int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  {
    const int per_clip = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int b8 = (gridDim.z / 8) * 8;                // clips covered by the remap (the tail keeps the plain order)
    if (lin < per_clip * b8) {
      const int xcd = lin & 7, slot = lin >> 3;
      const int w = slot % per_clip;
      // order 0: clip b on XCD b % 8.  order 1 / 2 (ping-pong traversal, api.hip): XCD x serves the contiguous clips
      // [x * b8/8, (x + 1) * b8/8) - the row range the persistent GEMMs' XCD x owns - first to last / last to first
      const int k = slot / per_clip;
      if constexpr (ORD) {
        const int nb = b8 >> 3;
        const int kk = order == 2 ? nb - 1 - k : k;
        b = order == 0 ? k * 8 + xcd : xcd * nb + kk;
      } else {
        b = k * 8 + xcd;
      }
      qb = w % gridDim.x;
      h = w / gridDim.x;
    }
  }
  const int H = heads * HD;
  // queries: rows [b*Sq, b*Sq+Sq) of qp_ (row stride q_ld); keys / values: rows [b*S, b*S+S) of kv (row stride ld), at
  // columns k_off / v_off past the head's first column.  Self-attention passes the same buffer twice (Sq == S).
  const int64_t row_base = (int64_t)b * S, qrow_base = (int64_t)b * Sq;
  const bf16_t* q_base = qp_ + qrow_base * q_ld + h * HD;
  const bf16_t* kv_base = kv + (int64_t)b * kv_rows * ld + h * HD;    // kv_rows >= S: rows between two clips' keys (a KV cache)

  const int q0 = qb * QB + wave * (32 * QR);     // query block x of this wave: rows q0 + 32 x + l31
  const bool wave_active = q0 < Sq;

  // Q fragments (B operand: column j = query, k = 8 contiguous head-dim elements)
  bf16x8 qf[QR][KS];
#pragma unroll
  for (int x = 0; x < QR; ++x) {
    const int q_row = q0 + 32 * x + l31;
    const bf16_t* qp = q_base + (int64_t)(q_row < Sq ? q_row : Sq - 1) * q_ld + hf * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[x][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  int ntiles = (S + KT - 1) / KT;
#ifdef ATTN_ONETILE
  ntiles = 1;
#endif
  if (CAUSAL) {
    const int last_q = min(qb * QB + QB - 1, S - 1);
    ntiles = min(ntiles, last_q / KT + 1);
  }

  // DMA geometry: piece pc of a tile covers LDS bytes [pc*1024, +1024) = linear 16-byte chunks pc*64 + lane.
  // chunk L -> row L / KCH, chunk position L % KCH; the K source chunk is un-swizzled from the position.
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)kv_base, 0, 0x7fffffff, 0x00020000);
#ifdef ATTN_LEAN
  // variant `attn_lean`: row, K chunk and V chunk of a piece packed into ONE register (6 + 8 + 8 bits; the uniform k_off / v_off
  // are added as scalars at use): 3 registers instead of 9 held through the tile loop, 3 more VALU per piece and tile
  int d_pack[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int L = (wave + i * NW) * 64 + lane;
    const int r = L / KCH, pos = L % KCH;
    d_pack[i] = r | ((((pos & ~3) | ((pos & 3) ^ ((r >> 2) & 3))) * 16) << 8) | ((pos * 16) << 16);
  }
#else
  int d_row[PPW], d_kcol[PPW], d_vcol[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int L = (wave + i * NW) * 64 + lane;
    const int r = L / KCH, pos = L % KCH;
    d_row[i] = r;
    d_kcol[i] = (k_off + (((pos & ~3) | ((pos & 3) ^ ((r >> 2) & 3))) * 8)) * 2;
    d_vcol[i] = (v_off + pos * 8) * 2;
  }
#endif
  float mreg = 1.f;
  auto issue_tile = [&](int t, int buf) {
    const int key0 = t * KT;
    char* kb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
#ifdef ATTN_LEAN
      const int rowoff = min(key0 + (d_pack[i] & 0xff), S - 1) * ld * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(kb + (wave + i * NW) * 1024), 16, rowoff + ((d_pack[i] >> 8) & 0xff) + k_off * 2, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(kb + K_BYTES + (wave + i * NW) * 1024), 16, rowoff + ((d_pack[i] >> 16) & 0xff) + v_off * 2, 0, 0, 0);
#else
      const int rowoff = min(key0 + d_row[i], S - 1) * ld * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(kb + (wave + i * NW) * 1024), 16, rowoff + d_kcol[i], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(kb + K_BYTES + (wave + i * NW) * 1024), 16, rowoff + d_vcol[i], 0, 0, 0);
#endif
    }
    // wave 0 fetches the tile's key mask; the value is only consumed in finish_tile, so that no wait on it (it would
    // also drain the DMA just issued) lands here
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

  f32x16 o[QR][DT];
#pragma unroll
  for (int x = 0; x < QR; ++x)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][dt][r] = 0.f;
  float m_run[QR], l_run[QR];               // running max of the RAW scores, running sum of exp
#pragma unroll
  for (int x = 0; x < QR; ++x) { m_run[x] = -INFINITY; l_run[x] = 0.f; }

  // per-lane part of the transposed V fragment address: 16-lane group (lane >> 4) & 1 selects the 16-column half,
  // lane >> 5 the 8-key half, (lane & 15) >> 2 the key within a 4-key block, lane & 3 the 4-column piece
  const int v_lane = (8 * hf + ((lane & 15) >> 2)) * VP + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);

  issue_tile(0, 0);
  finish_tile(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) issue_tile(t + 1, (t + 1) & 1);
    if (wave_active) {
      const char* kb = smem + (t & 1) * BUF;
      const char* vb = kb + K_BYTES;
      const float* bias = reinterpret_cast<const float*>(vb + V_BYTES);
      f32x16 s[QR][2];
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // the 2 QR chains (32-key halves x query blocks) are interleaved so that no MFMA waits on the one issued right
      // before it; each K fragment is read once and used by every query block
#if defined(ATTN_LEAN) && !defined(WAVESIM)
      // variant `attn_lean`: the per-lane K row offset is re-derived per tile from an opaque copy of the lane id (5 VALU per tile)
      // instead of being held through the loop: at 256 VGPRs that register is the one that spilled once the geometry was packed
      int l31 = lane;
      asm volatile("" : "+v"(l31));
      l31 &= 31;
#endif
      const int kx = (key_perm(l31) >> 2) & 3;          // same for both 32-key halves (32 >> 2 is a multiple of 4)
      const char* kr = kb + key_perm(l31) * RP;
#ifdef ATTN_KPIPE
#include "attention_kpipe.inc"
#else
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 2 + hf;       // logical 16-byte chunk; stored at (c & ~3) | ((c & 3) ^ kx)
        const int coff = ((c & ~3) | ((c & 3) ^ kx)) << 4;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kr + st * 32 * RP + coff);
#pragma unroll
          for (int x = 0; x < QR; ++x)
            s[x][st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[x][ks], ks == 0 ? zero16 : s[x][st], 0, 0, 0);   // C = 0 literal
        }
      }
#endif
      const bool pad_tile = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(bias + KT)[0]) != 0;
      const bool diag_tile = CAUSAL && (t * KT + KT - 1 > q0);
      bf16x8 pf[QR][4];
#pragma unroll
      for (int x = 0; x < QR; ++x) {
        const int q_row = q0 + 32 * x + l31;
        // masks (only for tiles that have any): s[x][st][g*8 + e] is key t*64 + st*32 + 16*g + 8*hf + e
        if (pad_tile || diag_tile) {
#pragma unroll
          for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int kl = st * 32 + 16 * g + 8 * hf;
              const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + kl);
              const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + kl + 4);
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float v = s[x][st][g * 8 + e] + (e < 4 ? b0[e] : b1[e - 4]);
                if (CAUSAL && (t * KT + kl + e) > q_row) v = -INFINITY;
                s[x][st][g * 8 + e] = v;
              }
            }
        }
        float m_tile = -INFINITY;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, s[x][st][r]);
        {   // a row's 64 keys sit in lanes l and l + 32: one v_permlane32_swap joins the halves (no LDS round trip)
          const unsigned mu = __builtin_bit_cast(unsigned, m_tile);
          const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
          m_tile = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
        }
        // Lazy exponent reference: softmax is invariant to the reference point, so m_run only moves when the row max
        // outgrows it by more than 2^8 (P <= 256 meanwhile; fp32 sums and bf16 P keep their relative precision).  The
        // O rescale (24 packed multiplies per block) then runs about once per row block instead of once per tile.
        const bool grow = (m_tile - m_run[x]) * scale_log2 > 8.f;     // m_run = -inf: true unless the tile is all masked
        if (__ballot(grow) != 0ull) {
          const float m_new = grow ? m_tile : m_run[x];
          const float alpha = __builtin_amdgcn_exp2f((m_run[x] - m_new) * scale_log2);   // 1 for rows that keep theirs
          const float a = (m_run[x] == -INFINITY) ? 0.f : alpha;      // -inf - -inf
          l_run[x] *= a;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[x][dt][r] *= a;
          m_run[x] = m_new;
        }
        const float m_use = (m_run[x] == -INFINITY) ? 0.f : m_run[x];
        const float neg = -m_use * scale_log2;
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32 work on register pairs) and four independent partial sums
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc2 = {scale_log2, scale_log2}, neg2 = {neg, neg};
        f32x2 ps2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const f32x2 sv = {s[x][st][g * 8 + e], s[x][st][g * 8 + e + 1]};
              const f32x2 xx = __builtin_elementwise_fma(sv, sc2, neg2);
              const f32x2 p = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
              ps2[(e >> 1) & 1] += p;
              pf[x][st * 2 + g][e] = (bf16_t)p[0];
              pf[x][st * 2 + g][e + 1] = (bf16_t)p[1];
            }
        l_run[x] += (ps2[0][0] + ps2[0][1]) + (ps2[1][0] + ps2[1][1]);
      }
      // O^T += V^T P^T: each transposed V fragment is read once and used by every query block
#pragma unroll
      for (int sp = 0; sp < 4; ++sp)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {      // QR * DT independent accumulator chains, interleaved
          const bf16x8 vf = v_frag_tr<VP>(vb + v_lane + dt * 64 + sp * 16 * VP);
#pragma unroll
          for (int x = 0; x < QR; ++x) o[x][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[x][sp], o[x][dt], 0, 0, 0);
        }
    }
    if (t + 1 < ntiles) finish_tile((t + 1) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of tile t+1 have landed
    __syncthreads();
  }

  if (!wave_active) return;
  // The output leaves as whole rows: a 32 x HD block is staged in LDS (the K / V ring is dead: the loop's last barrier is
  // behind every wave; the region is wave-private) and stored 16 bytes per lane, consecutive lanes on consecutive chunks
  // of a row.  Lane (l31, hf) holds query row l31, columns dt*32 + g*8 + 4*hf .. +3; the staging pitch RP + 16 keeps the
  // 8-byte writes conflict-free.  (Direct 8-byte stores at a row stride cost 32 lines per instruction: 57 of 405 us.)
  constexpr int OPITCH = RP + 16;
  static_assert(NW * 32 * OPITCH <= 2 * BUF, "output staging must fit in the K / V ring");
  char* stage = smem + wave * (32 * OPITCH);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifdef ATTN_LEAN
  // Variant build `attn_lean` (round 5, written without a GPU; the default kernels stay the round-2 instruction streams):
  // (1) the lane's staging address is re-derived here from an opaque copy of the lane id instead of `lane & 31` being carried
  // from the prologue - at 256 VGPRs the two-block kernel spilled exactly that register around its tile loop (8 bytes of scratch;
  // the thread id it derives from is live through the loop anyway); (2) the
  // normalisation and bf16 conversion work on register pairs (v_pk_mul_f32 + v_cvt_pk_bf16_f32): the scalar form below compiles
  // to ~150 surplus v_perm / v_alignbit / v_mov per wave around the conversions (profiles/r4_cpu/epilogue_budget.txt).
  int lane_e = lane;
#ifndef WAVESIM
  asm volatile("" : "+v"(lane_e));       // opaque copy of the lane id (the thread id is live through the loop anyway)
#endif
  const int st_off = (lane_e & 31) * OPITCH + (lane_e >> 5) * 8;
#endif
#pragma unroll
  for (int x = 0; x < QR; ++x) {
    const int qx = q0 + 32 * x;
    if (qx >= Sq) break;
    const float l_tot = l_run[x] + __shfl_xor(l_run[x], 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v;
#ifdef ATTN_LEAN
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const f32x2 y = f32x2{o[x][dt][g * 4 + e], o[x][dt][g * 4 + e + 1]} * f32x2{inv, inv};
          v[e] = (bf16_t)y[0];
          v[e + 1] = (bf16_t)y[1];
        }
        *reinterpret_cast<bf16x4*>(stage + st_off + (dt * 32 + g * 8) * 2) = v;
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[x][dt][g * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(stage + l31 * OPITCH + (dt * 32 + g * 8 + 4 * hf) * 2) = v;
#endif
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CACO_WAVE_LDS_SYNC();
    const int rows_valid = min(32, Sq - qx);
    const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(
        out + (qrow_base + qx) * H + h * HD, 0, (rows_valid - 1) * H * 2 + RP, 0x00020000);    // rows past Sq fall outside
#pragma unroll
    for (int it = 0; it < KCH / 2; ++it) {                   // 32 rows x KCH chunks of 16 B = KCH / 2 wave instructions
      const int L = it * 64 + lane;
      const int r = L / KCH, c = L % KCH;
      const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * OPITCH + c * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, out_r, r * H * 2 + c * 16, 0, ATTN_ST_AUX);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the staging block is reused by the next query block
    CACO_WAVE_LDS_SYNC();
  }
}

// workgroups per CU the register budget is held to: 3 (164 VGPRs) with one query block per wave, 2 (256) with two
template <int HD, bool CAUSAL, int NW, int QR, bool ORD = false>
__global__ __launch_bounds__(NW * 64, QR == 1 ? 3 : 2) void attention_kernel(const bf16_t* __restrict__ q, int q_ld, int Sq,
                                                               const bf16_t* __restrict__ kv, int ld, int k_off, int v_off,
                                                               const float* __restrict__ key_mask, int S, int heads,
                                                               bf16_t* __restrict__ out, float scale_log2, int kv_rows, int order) {
  attention_body<HD, CAUSAL, NW, QR, ORD>(q, q_ld, Sq, kv, ld, k_off, v_off, key_mask, S, heads, out, scale_log2, kv_rows, order);
}

// experiment switch: CACO_ATTN_ROWS=32 keeps one query block per wave at every sequence length
int attention_rows_per_wave() { return sw(SW_ATTN_ROWS); }

}  // namespace

int attention(const bf16_t* qkv, int ld, int k_off, int v_off, const float* key_mask, int batch, int seq, int heads,
              int head_dim, int causal, bf16_t* out, hipStream_t st, int order) {
  return attention_qkv(qkv, ld, seq, qkv, ld, k_off, v_off, key_mask, batch, seq, heads, head_dim, causal, out, st, 0, order);
}

// General form: queries [batch, seq_q] rows of `q` (row stride q_ld, head h at column h*head_dim), keys / values
// [batch, seq] rows of `kv` (row stride ld, head h at columns h*head_dim + k_off / v_off).  Cross-attention of the caption
// decoder (RobertaSelfAttention with key_value_states, src/caco_torch/text_models/roberta.py:67-104): seq_q != seq.
int attention_qkv(const bf16_t* q, int q_ld, int seq_q, const bf16_t* qkv, int ld, int k_off, int v_off, const float* key_mask,
                  int batch, int seq, int heads, int head_dim, int causal, bf16_t* out, hipStream_t st, int kv_batch_rows, int order) {
  if (kv_batch_rows <= 0) kv_batch_rows = seq;
  CACO_REQUIRE(kv_batch_rows >= seq, "attention: kv_batch_rows %d < seq %d", kv_batch_rows, seq);
  CACO_REQUIRE(batch > 0 && seq > 0 && seq_q > 0 && heads > 0, "attention: bad shape B=%d Sq=%d S=%d heads=%d", batch, seq_q, seq, heads);
  CACO_REQUIRE(!causal || seq_q == seq, "attention: the causal mask needs seq_q == seq (%d vs %d)", seq_q, seq);
  CACO_REQUIRE(q_ld % 8 == 0, "attention: query row stride must be a multiple of 8 elements");
  CACO_REQUIRE(head_dim == 64 || head_dim == 96, "attention: head_dim %d not in {64, 96}", head_dim);
  CACO_REQUIRE(heads <= 65535 && batch <= 65535, "attention: heads / batch exceed the grid limit");
  CACO_REQUIRE(ld % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0, "attention: row stride / operand offsets must be multiples of 8 elements");
  // short sequences (the text tower at T = 32): one wave per (clip, head, query block), attention_small.hip.  Opt-in
  // (switch SW_ATTN_SMALL / CACO_ATTN_SMALL=1) until it has been timed and verified on hardware.
  if (sw(SW_ATTN_SMALL) != 0 && attention_small_ok(seq_q, seq, head_dim))
    return attention_small(q, q_ld, seq_q, qkv, ld, k_off, v_off, key_mask, batch, seq, heads, causal, out, st, kv_batch_rows);
  const float scale_log2 = 1.4426950408889634f / sqrtf((float)head_dim);
  constexpr int NW = 4;
  // two query blocks per wave once the sequence fills the 256-row workgroups that makes; short sequences (text, decoder)
  // keep 128-row workgroups
  const int qr = (!causal && seq_q > 128 && attention_rows_per_wave() != 32) ? 2 : 1;
  const dim3 grid((seq_q + NW * 32 * qr - 1) / (NW * 32 * qr), heads, batch);
#define CACO_ATTN_(HD_, C_, QR_, O_) \
  hipLaunchKernelGGL((attention_kernel<HD_, C_, NW, QR_, O_>), grid, dim3(NW * 64), 0, st, q, q_ld, seq_q, qkv, ld, k_off, v_off, key_mask, seq, heads, out, scale_log2, kv_batch_rows, order)
  // a non-default workgroup order (ping-pong traversal) exists for the audio tower's shape class only
#define CACO_ATTN(HD_, C_, QR_) do { if (order != 0 && HD_ == 96 && !C_) CACO_ATTN_(96, false, QR_, true); else CACO_ATTN_(HD_, C_, QR_, false); } while (0)
#define CACO_ATTN_QR(HD_, C_) do { if (qr == 2) CACO_ATTN(HD_, C_, 2); else CACO_ATTN(HD_, C_, 1); } while (0)
  if (head_dim == 96) {
    if (causal) CACO_ATTN(96, true, 1); else CACO_ATTN_QR(96, false);
  } else {
    if (causal) CACO_ATTN(64, true, 1); else CACO_ATTN_QR(64, false);
  }
#undef CACO_ATTN_QR
#undef CACO_ATTN
#undef CACO_ATTN_
  return check_hip(hipGetLastError(), "attention launch");
}

}  // namespace caco
