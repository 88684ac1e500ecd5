// NOT BUILT.  Ping-pong (two wave groups one barrier apart) variant of the audio attention kernel, kept for the record.
//
// To try it: paste this block into cacophony_amd/csrc/attention.hip's anonymous namespace (it uses key_perm, v_frag_tr,
// KT and the typedefs there) and add the dispatch branch at the bottom of this file to attention_qkv().
// tools/experimental/attn_timing.py reads its -DATTN_PP_TIMING stamps.
//
// Measured on MI355X, 256 clips x 8 heads x 96, S = 496, random operands (tools/power_probe.py --attn):
//   this kernel                                   313 us   617 TFLOP/s   (299 us on zero operands: not power-bound, stall-bound)
//   free-running, 2 query blocks per wave (product) 264 us   731 TFLOP/s   (1.89 GHz at the 1400 W cap; 220 us on zeros)
//   free-running, 1 query block per wave           275 us   705 TFLOP/s
// Why it loses (stamps, ablations, tools/probes/valu_probe): ONE wave issues at most one instruction per ~12 shader
// cycles, whatever the instruction (v_fma 4.9 ns per instruction with one wave per SIMD, 1.2 ns per instruction and
// SIMD with four).  The softmax of a 64-row x 32-key block is ~150 VALU instructions = 1770 cycles for the one wave
// that is in its softmax phase, the matrix phase (24 MFMAs, 18 fragment reads) 1300 cycles against 768 MFMA cycles; a
// step lasts ~2000 cycles, so the matrix pipe is busy 37 % of the time - by construction only one wave per SIMD feeds the
// VALU at any moment.  Free-running waves do collide, but two (or three) waves issuing at once is worth more than the
// orderly overlap.  Also tried here: 64-key tiles (the compiler spilled the Q fragments: 609 us), __syncthreads() (drains
// vmcnt at every step), DMA distance of one tile (2.5 k cycles per tile in s_waitcnt vmcnt), pairing waves 2k / 2k+1
// instead of w / w+4 (326 us: HW_ID confirms w and w+4 share a SIMD).

// ---------------------------------------------------------------------------------------------------------------------
// Ping-pong variant for long non-causal sequences (the audio encoder: 8 heads x 96, S = 496).
//
// The kernel above runs at the SUM of its matrix and vector work: with two or three waves per SIMD that drift freely, the
// MFMA phases (K Q^T, V^T P^T) and the softmax phase (max, exp2, sum, bf16 pack: ~1100 VALU cycles per 64 x 64 score
// block against 1536 MFMA cycles) of the waves sharing a SIMD collide as often as they interleave (round 2 counters:
// MFMA busy 34 %, VALU busy 42 %, 4500 cycles per wave-tile where either pipe alone needs 1600).
//
// Here one workgroup = 8 waves = two groups of 4 (waves w and w + 4 share a SIMD), 64 query rows per wave, 512 rows per
// workgroup: a whole (clip, head) at S <= 512, so K and V are fetched ONCE.  The two groups run the same per-tile
// sequence  [matrix phase: O += V(t-1) P(t-1), S = K(t) Q^T]  barrier  [softmax phase: P(t) from S]  barrier  but group B
// is one barrier behind group A: in every step each SIMD has one wave in its matrix phase and one in its softmax
// phase - the overlap is built into the schedule instead of left to chance.
//
// Steps (= barrier intervals): A runs matrix(t) in step 2t and softmax(t) in step 2t+1; B one step later.  K(t) is read
// in steps 2t, 2t+1 and V(t) in steps 2t+2, 2t+3, so with two buffers each the DMA of V(t) and K(t+1) is issued at the
// top of step 2t (both targets were last read in step 2t-1) and waited for at the end of step 2t+1.  The running max is
// lazy: the exponent reference of a row only moves when the row max outgrows it by more than 2^8, so the O rescale
// (48 packed multiplies) runs about once per workgroup instead of once per tile; softmax is invariant to the reference.
#ifndef ATTN_PP_GRP_SHIFT
#define ATTN_PP_GRP_SHIFT 2    // which waves share a SIMD: 2: waves w and w+4 (measured better than 0: waves 2k and 2k+1)
#endif
template <int HD>
__device__ __forceinline__ void attention_pp_body(const bf16_t* __restrict__ qp_, int q_ld, int Sq, const bf16_t* __restrict__ kv,
                                                  int ld, int k_off, int v_off, const float* __restrict__ key_mask, int S,
                                                  int heads, bf16_t* __restrict__ out, float scale_log2, int kv_rows) {
  constexpr int NW = 8, QR = 2, QB = NW * 32 * QR;
  constexpr int RP = HD * 2, VP = RP, KCH = HD / 8;
  constexpr int KT2 = 32;                      // keys per tile: S (2 x 16 registers) and P (2 x 8) of a tile stay small enough
                                               // for 96 O + 48 Q registers to leave room to prefetch fragments (with 64-key
                                               // tiles the compiler spilled the Q fragments and waited on every LDS read)
  constexpr int NPC = KT2 * RP / 1024;         // 1 KiB DMA pieces per operand tile (6)
  constexpr int NB = 8, DIST = NB - 2;         // K / V ring depth and DMA distance in tiles: one workgroup per CU means the
                                               // ring alone has to cover the HBM latency (a distance of one 32-key tile left
                                               // the issuing waves 2.5 k cycles per tile in s_waitcnt vmcnt)
  constexpr int NPE = 2 * NPC / 4;             // DMA instructions per event and wave of group B, which issues all of it: its
                                               // even steps are softmax phases (no other memory instruction to wait behind)
  static_assert((2 * NPC) % 4 == 0, "pieces must divide over the four issuing waves");
  constexpr int KS = HD / 16, DT = HD / 32;
  constexpr int T_BYTES = KT2 * RP;            // one K or V tile
  constexpr int BIAS_BYTES = KT2 * 4 + 16;
  constexpr int OPITCH = RP + 16;
  constexpr int STAGE_OFF = 2 * NB * T_BYTES + 2 * BIAS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[STAGE_OFF + NW * 32 * OPITCH];
  static_assert(sizeof(smem) <= 160 * 1024, "LDS");
  char* const kbuf = smem;                     // K tiles [NB]
  char* const vbuf = smem + NB * T_BYTES;      // V tiles [NB]
  char* const bbuf = smem + 2 * NB * T_BYTES;  // key bias + "has a masked key" flag [2]

  const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = (wave >> ATTN_PP_GRP_SHIFT) & 1;   // 0: group A, 1: group B (one step behind)
  const int gidx = ATTN_PP_GRP_SHIFT ? (wave & 3) : (wave >> 1);   // index of the wave within its group
  int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  {   // all workgroups of a clip on one XCD (see attention_body)
    const int per_clip = gridDim.x * gridDim.y;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int b8 = (gridDim.z / 8) * 8;
    if (lin < per_clip * b8) {
      const int xcd = lin & 7, slot = lin >> 3;
      const int w = slot % per_clip;
      b = (slot / per_clip) * 8 + xcd;
      qb = w % gridDim.x;
      h = w / gridDim.x;
    }
  }
  const int H = heads * HD;
  const int64_t row_base = (int64_t)b * S, qrow_base = (int64_t)b * Sq;
  const bf16_t* q_base = qp_ + qrow_base * q_ld + h * HD;
  const bf16_t* kv_base = kv + (int64_t)b * kv_rows * ld + h * HD;

  const int q0 = qb * QB + wave * (32 * QR);
  const bool wave_active = q0 < Sq;
  const int ntiles = (S + KT2 - 1) / KT2;

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)kv_base, 0, 0x7fffffff, 0x00020000);
  // DMA event e = { K(e + DIST), V(e + DIST - 1) } = 12 pieces, three per wave of group B; piece p < 6 is K, else V.
  // Every event has the same number of instructions (tiles outside [0, ntiles) re-read a clamped row into a ring slot
  // nobody reads), so "event e has landed" is the counted wait vmcnt(NPE * events issued since).
  int d_row[NPE], d_col[NPE], d_lds[NPE], d_tile[NPE];
#pragma unroll
  for (int i = 0; i < NPE; ++i) {
    const int pc = gidx + 4 * i;
    const bool is_v = pc >= NPC;
    const int pp = is_v ? pc - NPC : pc;
    const int L = pp * 64 + lane;
    const int r = L / KCH, pos = L % KCH;
    d_row[i] = r;
    d_col[i] = is_v ? (v_off + pos * 8) * 2 : (k_off + (((pos & ~3) | ((pos & 3) ^ ((r >> 2) & 3))) * 8)) * 2;
    d_lds[i] = (is_v ? NB * T_BYTES : 0) + pp * 1024;
    d_tile[i] = is_v ? DIST - 1 : DIST;
  }
  auto issue_event = [&](int e) {
#ifdef ATTN_PP_NODMA
    return;
#endif
    if (!grp) return;
#pragma unroll
    for (int i = 0; i < NPE; ++i) {
      const int tile = e + d_tile[i];
      const int rowoff = max(0, min(tile * KT2 + d_row[i], S - 1)) * ld * 2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(smem + d_lds[i] + (tile & (NB - 1)) * T_BYTES), 16,
                                               rowoff + d_col[i], 0, 0, 0);
    }
  };
  float mreg = 1.f;                            // wave 0: key mask of the tile whose bias it writes next
  auto load_mask = [&](int t) {
    if (tid < KT2) {
      const int key = t * KT2 + tid;
      mreg = (key < S) ? (key_mask ? key_mask[row_base + key] : 1.f) : 0.f;
    }
  };
  auto write_bias = [&](int t) {               // wave 0: per-key additive mask + "this tile has a masked key" flag
    if (tid < KT2) {
      float* bias = reinterpret_cast<float*>(bbuf + (t & 1) * BIAS_BYTES);
      const float breg = mreg != 0.f ? 0.f : -INFINITY;
      bias[tid] = breg;
      const unsigned long long any = __ballot(breg != 0.f);
      if (tid == 0) reinterpret_cast<int*>(bias + KT2)[0] = any != 0ull;
    }
  };

  // ---- prologue: DMA events -DIST .. -1 (K(0 .. DIST-1), V(0 .. DIST-2)), the first key mask, the Q fragments -------------
#pragma unroll
  for (int e = -DIST; e < 0; ++e) issue_event(e);
  load_mask(0);
  bf16x8 qf[QR][KS];
#pragma unroll
  for (int x = 0; x < QR; ++x) {
    const int q_row = q0 + 32 * x + l31;
    const bf16_t* qp = q_base + (int64_t)(q_row < Sq ? q_row : Sq - 1) * q_ld + hf * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[x][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }
  // Everything issued so far is waited for HERE, with the builtin the compiler's counter bookkeeping understands: left to
  // itself it guards the Q registers with vmcnt(small) inside the loop, which drains the DMA ring on every tile.
  __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0), expcnt / lgkmcnt untouched
  f32x16 o[QR][DT];
#pragma unroll
  for (int x = 0; x < QR; ++x)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][dt][r] = 0.f;
  float m_ref[QR], l_run[QR];                  // exponent reference (RAW score units), running sum of exp
#pragma unroll
  for (int x = 0; x < QR; ++x) { m_ref[x] = -INFINITY; l_run[x] = 0.f; }
  const int v_lane = (8 * hf + ((lane & 15) >> 2)) * VP + ((((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
  const int kx = (key_perm(l31) >> 2) & 3;
  const int k_lane = key_perm(l31) * RP;
  f32x16 s[QR];
  bf16x8 pf[QR][2];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // matrix phase of tile t: O^T += V(t-1)^T P(t-1)^T, then S^T = K(t) Q^T.  Every fragment read feeds two MFMAs.
  auto matrix_phase = [&](int t) {
#ifdef ATTN_PP_NOMATRIX
    return;
#endif
    if (!wave_active) return;
    if (t > 0) {
      const char* vb = vbuf + ((t - 1) & (NB - 1)) * T_BYTES;
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 vf = v_frag_tr<VP>(vb + v_lane + dt * 64 + sp * 16 * VP);
#pragma unroll
          for (int x = 0; x < QR; ++x) o[x][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[x][sp], o[x][dt], 0, 0, 0);
        }
    }
    if (t < ntiles) {
      const char* kr = kbuf + (t & (NB - 1)) * T_BYTES + k_lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = ks * 2 + hf;
        const int coff = ((c & ~3) | ((c & 3) ^ kx)) << 4;
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kr + coff);
#pragma unroll
        for (int x = 0; x < QR; ++x)
          s[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[x][ks], ks == 0 ? zero16 : s[x], 0, 0, 0);
      }
    }
  };
  // softmax phase of tile t: P(t) (bf16 MFMA operand) from S, running sums; all VALU
  auto softmax_phase = [&](int t) {
#ifdef ATTN_PP_NOSOFTMAX
    return;
#endif
    if (!wave_active) return;
    const float* bias = reinterpret_cast<const float*>(bbuf + (t & 1) * BIAS_BYTES);
    const bool pad_tile = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(bias + KT2)[0]) != 0;
    if (pad_tile) {         // s[x][g*8 + e] is key t*32 + 16*g + 8*hf + e
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int kl = 16 * g + 8 * hf;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + kl);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + kl + 4);
#pragma unroll
        for (int x = 0; x < QR; ++x)
#pragma unroll
          for (int e = 0; e < 8; ++e) s[x][g * 8 + e] += (e < 4 ? b0[e] : b1[e - 4]);
      }
    }
    // row max of both blocks: lane-local over the 16 keys a lane holds, then one v_permlane32_swap joins the two halves
    // (a row's 32 keys sit in lanes l and l + 32); no LDS round trip, and both blocks' chains are independent
    float m_tile[QR];
#pragma unroll
    for (int x = 0; x < QR; ++x) {
      float m = s[x][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, s[x][r]);
      const unsigned mu = __builtin_bit_cast(unsigned, m);
      const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
      m_tile[x] = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
    }
    // lazy reference: move it only when the row max outgrows it by more than 8 in the log2 domain (P <= 256 otherwise)
    bool grow[QR];
    bool any_grow = false;
#pragma unroll
    for (int x = 0; x < QR; ++x) {
      grow[x] = (m_tile[x] - m_ref[x]) * scale_log2 > 8.f;      // m_ref = -inf: true unless the tile is all masked
      any_grow |= grow[x];
    }
    if (__ballot(any_grow) != 0ull) {
#pragma unroll
      for (int x = 0; x < QR; ++x) {
        const float m_new = grow[x] ? m_tile[x] : m_ref[x];
        if (t > 0) {                                                 // (tile 0: O and l are still zero)
          const float alpha = __builtin_amdgcn_exp2f((m_ref[x] - m_new) * scale_log2);   // 1 for rows that keep theirs
          const float a = (m_ref[x] == -INFINITY) ? 0.f : alpha;     // -inf - -inf
          l_run[x] *= a;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[x][dt][r] *= a;
        }
        m_ref[x] = m_new;
      }
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 sc2 = {scale_log2, scale_log2};
#pragma unroll
    for (int x = 0; x < QR; ++x) {
      const float m_use = (m_ref[x] == -INFINITY) ? 0.f : m_ref[x];
      const float neg = -m_use * scale_log2;
      const f32x2 neg2 = {neg, neg};
      f32x2 ps2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 sv = {s[x][g * 8 + e], s[x][g * 8 + e + 1]};
          const f32x2 xx = __builtin_elementwise_fma(sv, sc2, neg2);
          const f32x2 p = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
          ps2[(e >> 1) & 1] += p;
          pf[x][g][e] = (bf16_t)p[0];
          pf[x][g][e + 1] = (bf16_t)p[1];
        }
      l_run[x] += (ps2[0][0] + ps2[0][1]) + (ps2[1][0] + ps2[1][1]);
    }
  };
  // the oldest event in flight has landed (VMEM returns in order); group A has nothing in flight
  auto wait_dma = [&]() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPE * (DIST - 1)) : "memory"); };
#ifdef ATTN_PP_TIMING      // s_memtime before and after every barrier, per wave, for the first 256 workgroups (tools/attn_timing.py)
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(out + (int64_t)gridDim.z * Sq * H) +
                               ((int64_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NW + wave) * 96;
  const bool stamp_on = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) < 256 && lane == 0;
  int stamp_n = 1;
  if (stamp_on) stamps[0] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
#define ATTN_STAMP() do { if (stamp_on && stamp_n < 96) stamps[stamp_n] = __builtin_amdgcn_s_memtime(); ++stamp_n; } while (0)
#else
#define ATTN_STAMP() do { } while (0)
#endif
  auto step_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    ATTN_STAMP();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ATTN_STAMP();
    __builtin_amdgcn_sched_barrier(0);
  };

  step_barrier();                               // step 0 begins: the ring is primed
  if (grp) {                                    // group B idles through step 0 (the DMA event aside)
    issue_event(0);
    step_barrier();
  }
  for (int t = 0; t <= ntiles; ++t) {
    if (wave == 0 && t < ntiles) { write_bias(t); load_mask(t + 1); }     // top of an even step (group A)
    matrix_phase(t);
    if (grp) wait_dma();                        // end of an odd step: event t + 1 - DIST = K(t + 1), V(t), read from step 2t+2
    step_barrier();
    issue_event(t + 1);                         // top of an even step (group B)
    if (t < ntiles) softmax_phase(t);
    step_barrier();
  }
  if (!grp) step_barrier();                     // both groups execute the same number of barriers
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last events only re-read a clamped row; nothing may land later)

  if (!wave_active) return;
  // ---- output: whole rows, staged per 32-row block through a wave-private LDS region (see attention_body) --------------
  char* stage = smem + STAGE_OFF + wave * (32 * OPITCH);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
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
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[x][dt][g * 4 + e] * inv);
        *reinterpret_cast<bf16x4*>(stage + l31 * OPITCH + (dt * 32 + g * 8 + 4 * hf) * 2) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int rows_valid = min(32, Sq - qx);
    const __amdgpu_buffer_rsrc_t out_r = __builtin_amdgcn_make_buffer_rsrc(
        out + (qrow_base + qx) * H + h * HD, 0, (rows_valid - 1) * H * 2 + RP, 0x00020000);
#pragma unroll
    for (int it = 0; it < KCH / 2; ++it) {
      const int L = it * 64 + lane;
      const int r = L / KCH, c = L % KCH;
      const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * OPITCH + c * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, out_r, r * H * 2 + c * 16, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <int HD>
__global__ __launch_bounds__(512) void attention_pp_kernel(const bf16_t* __restrict__ q, int q_ld, int Sq,
                                                           const bf16_t* __restrict__ kv, int ld, int k_off, int v_off,
                                                           const float* __restrict__ key_mask, int S, int heads,
                                                           bf16_t* __restrict__ out, float scale_log2, int kv_rows) {
  attention_pp_body<HD>(q, q_ld, Sq, kv, ld, k_off, v_off, key_mask, S, heads, out, scale_log2, kv_rows);
}


// ---- dispatch branch for attention_qkv() ----
#if 0
  if (!causal && head_dim == 96 && seq_q > 256 && attention_rows_per_wave() == 64) {   // ping-pong kernel: 512 rows per workgroup
    const dim3 grid((seq_q + 511) / 512, heads, batch);
    hipLaunchKernelGGL((attention_pp_kernel<96>), grid, dim3(512), 0, st, q, q_ld, seq_q, qkv, ld, k_off, v_off, key_mask, seq,
                       heads, out, scale_log2, kv_batch_rows);
    return check_hip(hipGetLastError(), "attention launch");
  }
#endif
