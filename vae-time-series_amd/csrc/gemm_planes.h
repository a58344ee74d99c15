// bf16x3 GEMM core over operands that already ARE bf16 planes.  Every fp32 operand value x is stored as a bf16
// head and a bf16 tail (x = hi + lo, two planes), and each plane is cut into 1 KiB tiles of 16 rows x 32 k.
// A product is three MFMAs per fragment pair (hi*lo, lo*hi, hi*hi) on v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//
// Tile format.  Element (row r, k) of a tile sits at byte  slot(k >> 3, r) * 16 + (k & 7) * 2  with
//     slot(g, r) = g * 16 + (r ^ ((g & 1) << 1)).
// Without the XOR this is the operand order of the MFMA (lane 16 g + r holds row r, k = 8 g .. 8 g + 7 as 16
// contiguous bytes).  A tile is one global_load_lds_dwordx4 per wave (global -> LDS, no registers, no conversion
// work, LDS image = memory image) and serves two kinds of product:
//   * contraction over the tile's k (rows of the tile = rows of the operand): lane (g, r) reads its fragment with
//     one ds_read_b128 at slot(g, r) * 16.  The XOR keeps that read conflict-free: the hardware serves the lane
//     groups {g even: r 0-3, 12-15; g odd: r 4-11} and its complement, and r ^ 2 maps each of those sets to itself.
//   * contraction over the tile's ROWS (weight gradients: rows are tokens, k are features): ds_read_b64_tr_b16
//     gathers, per 16 lanes, 4 rows x 16 columns and hands lane i column i.  One MFMA k-step of 32 rows is two
//     tiles; lane group g' of the fragment holds rows {8 (g' & 1) + 4 (q >> 1) + 2 h + (q & 1)} of tile g' >> 1 in
//     element 4 h + q (h = which of the two reads).  The row order inside a k-step is the same for both operands,
//     so it cancels in the sum.  With the XOR the 32 lanes of a half touch all 64 banks once (without it the two
//     16-byte slots a row has in a 32-byte column block share their banks: 2-way).
//
// Workgroup tile 256 x 256 x 32 per k-step (32 + 32 tiles = 64 KiB), 8 waves as 4 (M) x 2 (N), each 64 x 128 of
// the output = 4 x 8 accumulator tiles (128 registers).  Per k-step a wave issues 8 tile loads for the k-step after
// next and 96 MFMAs.  A k-step is 3072 matrix-pipe clocks per SIMD; an HBM miss needs about two of them, hence the
// five-slot ring below.
//
// The kernel body is a template over a policy that says where the tiles of k-step s are; what happens to the
// accumulators is the caller's business.
#pragma once
#include "vts_common.h"

namespace gp {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int TILE_B = 1024;                   // bytes of one tile
constexpr int TM = 256, TN = 256;              // workgroup tile
constexpr int HALF_TILES = 32;                 // tiles of one operand per k-step (16 row tiles x 2 planes, or 2 x 8 x 2)
constexpr int HALF_B = HALF_TILES * TILE_B;    // 32 KiB
constexpr int LDS_B = 5 * HALF_B;              // 160 KiB: the whole LDS of a CU
constexpr int NWAVE = 8;

__host__ __device__ constexpr int slot(int g, int r) { return g * 16 + (r ^ ((g & 1) << 1)); }
// byte offset of element (row r, k) inside a tile
__host__ __device__ constexpr int elem_offset(int r, int k) { return slot(k >> 3, r) * 16 + (k & 7) * 2; }

__device__ __forceinline__ void glds16(const char* g, char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ bf16x8 asfrag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
// Transposed fragment read as inline asm: through the builtin the compiler takes the read for a possible reader of
// what the LDS-DMA loads in flight are writing and puts s_waitcnt vmcnt(0) in front of every one of them, which
// drains the ring.  The price is that the compiler no longer counts these reads: every use sits behind an explicit
// lgkmcnt wait that names the registers (wait_frag), so the MFMAs cannot be scheduled above it.
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ u32x2 lds_tr(unsigned addr, int off) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
  return v;
}
__device__ __forceinline__ void wait_frag(u32x4& x, u32x4& y) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y));
}

// Sequence operand as planes: tokens (window b, step t) in tiles of 16 windows (batch tile bt = b / 16) at one
// step, `ncb` column blocks of 32 features, `ts` step slots per batch tile of which step t sits in slot t0 + t
// (the h sequence keeps a zero tile in slot 0 - the initial state - so that "the step before" is a plain offset).
struct SeqPlanes {
  char* base;
  int ncb, ts, t0;
  __host__ __device__ long tile(long bt, int t, int cb, int p) const {
    return ((((bt * ts) + t0 + t) * ncb + cb) * 2 + p) * (long)TILE_B;
  }
  __host__ __device__ size_t bytes(long nbt) const { return (size_t)nbt * ts * ncb * 2 * TILE_B; }
};

// Policy interface.  pol.advance() is called once per k-step, in order, before that k-step's loads are issued
// (wave-uniform bookkeeping); pol.a_tile(li) / pol.b_tile(li) then give the global address of tile li (0..31) of
// the operand's half stage for that k-step; pol.flush(acc) is called after the products of every nk-th k-step with
// the finished accumulators (which are cleared afterwards).  Order of the 32 tiles:
//   TRANS = false: li = 2 * i + p      row tile i (0..15) of the workgroup's 256 rows, plane p (0 hi, 1 lo);
//   TRANS = true : li = (tt * 8 + cb) * 2 + p   tile tt (0, 1) of the k-step's two row tiles, column block cb
//                  (32 columns each, 0..7) of the workgroup's 256 output rows (A) or columns (B), plane p.
//
// LDS is a ring of five 32 KiB slots that holds the half stages A_0, B_0, A_1, B_1, ... in turn (half stage h in
// slot h % 5), so the loads of k-step s+2 are issued during the products of k-step s: A_{s+2} takes the slot B_{s-1}
// left at the end of the previous k-step, B_{s+2} takes A_s's slot once every wave holds its A fragments in
// registers (the barrier after the fragment reads).  The DMA loads stay in flight across the barriers: waits are
// counted (each wave issues NP = eight per k-step, vmcnt(NP) retires everything but the newest NP) and the barriers
// are bare s_barrier instructions.  A slot is read only after a barrier that follows the wait that retired it.
//
// NTW = column tiles (of 16) per wave: 8 for the 256-column workgroup tile, 4 (transposed mode only) for a workgroup
// tile of 256 x 128 - half the B tiles (li = (tt * 4 + cb) * 2 + p, 16 per k-step, two per wave), half the MFMAs:
// for the last, half-filled column tile of a weight-gradient plan.  pol.flush then gets NTW accumulator columns.
template <class Pol, bool TRANS, int NTW = 8, int DBG = 0>   // DBG (timing experiments only): 1 = no global loads, 2 = no products
__device__ __forceinline__ void mainloop(Pol& pol, int nsteps, int nk, char* lds) {
  static_assert(NTW == 8 || (NTW == 4 && TRANS), "column tiles per wave");
  constexpr int NP = 4 + NTW / 2;     // tile loads a wave issues per k-step
  f32x4 acc[4][NTW];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nsteps <= 0) return;

  // piece q (0..NP-1) of a wave's loads for a k-step: q < 4 -> A tile 4 * wave + q; q >= 4 -> B tile
  // (NTW / 2) * wave + q - 4
  auto issue_piece = [&](int slot_a, int slot_b, int q) {
    if (q < 4) {
      const int li = 4 * wave + q;
      glds16(pol.a_tile(li) + lane * 16, lds + slot_a * HALF_B + li * TILE_B);
    } else {
      const int li = (NTW / 2) * wave + (q - 4);
      glds16(pol.b_tile(li) + lane * 16, lds + slot_b * HALF_B + li * TILE_B);
    }
  };
  auto next2 = [](int sl) { return sl >= 3 ? sl - 3 : sl + 2; };   // (sl + 2) % 5

  // per-lane byte offsets of the fragment reads inside a half stage
  int off0, off1 = 0;   // direct: off0; transposed: off0 (h = 0) and off1 (h = 1)
  int ttb = 0;          // transposed, NTW = 4: what the second row tile's B tiles sit closer than A's (8 tiles, not 16)
  if (!TRANS) {
    off0 = slot(lane >> 4, lane & 15) * 16;
  } else {
    const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int row0 = ((gq & 1) << 3) + ((q >> 1) << 2) + (q & 1);        // h = 0; h = 1 adds 2
    const int base = (gq >> 1) * (16 * TILE_B) + (p >> 1) * 256 + (p & 1) * 8;
    off0 = base + ((row0 ^ ((p >> 1) << 1)) << 4);
    off1 = base + (((row0 + 2) ^ ((p >> 1) << 1)) << 4);
    if (NTW == 4) ttb = (gq >> 1) * (8 * TILE_B);
  }
  // fragment of 16-row (or, transposed, 16-column) block wbase + j of the workgroup's 256, plane p; j and p are
  // compile-time after unrolling (the transposed read takes them as an immediate offset), wbase is the wave's
  const int wa = TRANS ? wm * 4 * TILE_B : wm * 8 * TILE_B;      // byte offset of the wave's first A block
  const int wb = (TRANS ? wn * NTW * TILE_B : wn * 16 * TILE_B) - ttb;     // and of its first B block
  auto frag = [&](const char* half, int woff, int j, int p) -> u32x4 {
    if (!TRANS) return *reinterpret_cast<const u32x4*>(half + woff + (j * 2 + p) * TILE_B + off0);
    const unsigned t = lds_addr(half) + woff;
    const int imm = ((j >> 1) * 2 + p) * TILE_B + (j & 1) * 512;
    const u32x2 lo = lds_tr(t + off0, imm), hi = lds_tr(t + off1, imm);
    return u32x4{lo.x, lo.y, hi.x, hi.y};
  };

  if (DBG != 1) {
    pol.advance();
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_piece(0, 1, q);
    if (nsteps > 1) {
      pol.advance();
#pragma unroll
      for (int q = 0; q < NP; ++q) issue_piece(2, 3, q);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();

  int sa = 0;   // slot of A_s; B_s is in sa + 1 (mod 5)
  int kleft = nk;
  for (int s = 0; s < nsteps; ++s) {
    const int sb = sa == 4 ? 0 : sa + 1;
    const int sa2 = next2(next2(sa)), sb2 = sa;        // slots of A_{s+2}, B_{s+2}
    const bool more = DBG != 1 && s + 2 < nsteps;
    const char* As = lds + sa * HALF_B;
    const char* Bs = lds + sb * HALF_B;
    u32x4 a[4][2];
    u32x4 bh, bl;
    if (DBG != 2) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) a[mt][p] = frag(As, wa, mt, p);
      bh = frag(Bs, wb, 0, 0);
      bl = frag(Bs, wb, 0, 1);
      if (TRANS) {
        wait_frag(a[0][0], a[0][1]); wait_frag(a[1][0], a[1][1]); wait_frag(a[2][0], a[2][1]); wait_frag(a[3][0], a[3][1]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave holds its A fragments: A_s's slot is free
    if (more) pol.advance();
    // B fragments one column tile ahead of their MFMAs; the scheduling fences keep the compiler from hoisting
    // all sixteen fragment reads to the top (128 registers) and spilling the accumulators.
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      if (more) {
        issue_piece(sa2, sb2, nt);
        if (nt + NTW < NP) issue_piece(sa2, sb2, nt + NTW);
      }
      if (DBG != 2) {
        if (TRANS) wait_frag(bh, bl);
        u32x4 nh = bh, nl = bl;
        if (nt < NTW - 1) {
          nh = frag(Bs, wb, nt + 1, 0);
          nl = frag(Bs, wb, nt + 1, 1);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(asfrag(a[mt][0]), asfrag(bl), acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(asfrag(a[mt][1]), asfrag(bh), acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(asfrag(a[mt][0]), asfrag(bh), acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        bh = nh; bl = nl;
      }
    }
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");    // k-step s+1 has landed, s+2 stays in flight
    else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    sa = next2(sa);
    if (--kleft == 0) {
      kleft = nk;
      pol.flush(acc);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

}  // namespace gp
