// Pieces shared by the persistent 256x256x64 bf16 GEMM kernels (gemm_w8.hip: eight waves of 128 x 64; gemm_w4q.hip: four
// waves of 128 x 128): LDS ring geometry, tile order, operand DMA cursors, fragment reads.  See gemm_w8.hip for the design.
#pragma once
#include "common.h"
#include "kernels.h"

namespace caco {
namespace {

constexpr int WBK = 64;                       // K-tile, bf16 elements
constexpr int WROWB = WBK * 2;                // 128 bytes per row per K-tile
constexpr int W_SLOT = 256 * WROWB;           // 32 KiB: one operand's K-tile
constexpr int W_AOFF = 0;                     // A ring: slots 0..2
constexpr int W_WOFF = 3 * W_SLOT;            // W ring: slots 0..1
constexpr int W_SMEM = 5 * W_SLOT;            // 163840 = 160 KiB

typedef __attribute__((address_space(3))) void* lds_vptr;

#ifdef W4_NOREADS
#define W4_DO_READS 0
#else
#define W4_DO_READS 1
#endif
#define W4_KOFF(kt) ((kt) * (WBK * 2))
#define W4_SGB_MFMA 0x008
#define W4_SGB_VMEM 0x010
#define W4_SGB_DSRD 0x100

// Tile order.  The n-tiles are processed in groups of G (p.ngroup): all M panels of one group, then the next group,
// n fastest inside a group.  A group's weight rows (G x 256 x K bf16) then stay in the XCD's 4 MiB L2 for the whole
// pass instead of being re-streamed through it once per M panel.  Default: groups of 4 at K <= 1024 (launch_w8).
// G < 0: the same list walked last to first (GemmArgs::reverse).
__device__ __forceinline__ void w4_decode(int t, int tiles_n, int tiles_m, int G, int& tm, int& tn) {
  if (G < 0) {
    G = -G;
    t = tiles_m * tiles_n - 1 - t;
  }
  const int per = tiles_m * G;
  const int g = t / per;                 // groups before the last one are full
  const int n0 = g * G;
  const int gw = min(G, tiles_n - n0);   // width of this group
  const int rem = t - g * per;
  tm = rem / gw;
  tn = n0 + rem % gw;
}

// Load cursors.  Each walks the K-tiles of this workgroup's output tiles in order, ahead of the compute cursor.
template <int NWV>
struct W4CurA {
  __amdgpu_buffer_rsrc_t r;
  int voff[32 / NWV]; // per-lane byte offsets of this wave's row groups (rows clamped to the last valid row)
  int li, kt;
};
struct W4CurW {
  __amdgpu_buffer_rsrc_t r;
  int voff;           // row group 0; group `it` adds it * 32 rows through the scalar offset
  int li, kt;
};

template <int NWV>
__device__ __forceinline__ void w4_setup_a(W4CurA<NWV>& C, const GemmArgs& p, int t, int tiles_n, int lda, int wave, int lane) {
  int tm, tn;
  w4_decode(t, tiles_n, (int)((p.M + 255) / 256), p.ngroup, tm, tn);
  const int64_t m0 = (int64_t)tm * 256;
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + m0 * lda), 0, 0x7fffffff, 0x00020000);
  const int r8 = wave * 8 + (lane >> 3);                         // row within a span of NWV * 8 rows
  const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7); // (row >> 1) & 7 is the same for every span
  const int last = (int)min((int64_t)256, p.M - m0) - 1;
#pragma unroll
  for (int it = 0; it < 32 / NWV; ++it) C.voff[it] = min(it * NWV * 8 + r8, last) * lda * 2 + chunk * 16;
}
__device__ __forceinline__ void w4_setup_w(W4CurW& C, const GemmArgs& p, int t, int tiles_n, int ldw, int wave, int lane) {
  int tm, tn;
  w4_decode(t, tiles_n, (int)((p.M + 255) / 256), p.ngroup, tm, tn);
  const int n0 = tn * 256;
  C.r = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * ldw), 0, 0x7fffffff, 0x00020000);
  const int r8 = wave * 8 + (lane >> 3);
  const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
  C.voff = r8 * ldw * 2 + chunk * 16;
}

// cache policy of the operand DMA loads (0 default, 2 = nt, 16 = sc1): A/B switches, tools/build_variant.sh
#ifndef W8_A_AUX
#define W8_A_AUX 0
#endif
#ifndef W8_W_AUX
#define W8_W_AUX 0
#endif
// piece `it` of this wave's share of an operand K-tile: 8 rows x 128 B = one wave instruction
template <int NWV>
__device__ __forceinline__ void w4_piece_a(const W4CurA<NWV>& C, int it, char* slot, int wave) {
#ifndef W4_NODMA
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.r, (lds_vptr)(slot + (it * NWV + wave) * 1024), 16, C.voff[it], W4_KOFF(C.kt), 0, W8_A_AUX);
#endif
}
template <int NWV>
__device__ __forceinline__ void w4_piece_w(const W4CurW& C, int it, int ldw, char* slot, int wave) {
#ifndef W4_NODMA
  __builtin_amdgcn_raw_ptr_buffer_load_lds(C.r, (lds_vptr)(slot + (it * NWV + wave) * 1024), 16, C.voff,
                                           W4_KOFF(C.kt) + it * NWV * 8 * ldw * 2, 0, W8_W_AUX);
#endif
}

__device__ __forceinline__ bf16x8 w4_frag(const char* oper, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(oper + row * WROWB + ((chunk ^ ((row >> 1) & 7)) << 4));
}

}  // namespace
}  // namespace caco
