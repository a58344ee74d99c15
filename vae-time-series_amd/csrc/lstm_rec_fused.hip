// Forward pass of one LSTM layer, bf16x3, hidden 256, as ONE kernel: input projection and recurrence together,
// both weight matrices stationary in registers, no gate pre-activations in memory.
//
// The two-kernel forward path computes G_x = x . W_ih^T + b for all steps (1.6 GB written at 4096 x 96 windows),
// then a recurrent kernel reads it back; the recurrent kernel's matrix pipe is busy a third of the time.  Here
// the x product of step t is done by the recurrent kernel itself, in the time it would otherwise wait for its
// peers: per step gates_t = [x_t | h_{t-1}] . [W_ih | W_hh]^T + b.
//
// Who owns what.  Clusters of eight workgroups (eight CUs) share 128 windows, as in lstm_rec_cluster.hip.  Member c
// owns hidden units [32 c, 32 c + 32) for all four gates; inside a member, wave w owns hidden units 8 w .. 8 w + 7
// of those, i.e. 32 gate columns = two MFMA column tiles, for ALL 128 windows.  The wave keeps its slice of both
// matrices - 32 columns x (Kx + 256) k, bf16 head and tail planes - in registers for the whole sequence (256
// registers per lane at Kx = 256).  The A operand (x_t and h_{t-1} of the cluster's 128 windows: plane tiles of
// gemm_planes.h, 1 KiB each) comes from memory to LDS by DMA, 64 KiB (four k-steps) at a time into two alternating
// halves, and every wave reads all of it.
//
// Column tiles of a wave: tile 0 = [gate i of its 8 units | gate g of its 8 units], tile 1 = [gate f | gate o].
// So lane (g4, r) holds, for window rows 4 g4 .. 4 g4 + 3 of a window tile, i and f of unit r (r < 8) or g and o of
// unit r - 8 (r >= 8): each lane applies its own two activations, the two half rows swap two rows' worth of values
// (one DPP rotation by eight lanes), and each then finishes the cell update for two of the four rows.
//
// Hand-off.  The plane buffer of the layer's h sequence is the exchange medium: member c writes column block c of
// the cluster's eight window tiles for step t (sc1 stores), drains them, and raises its flag to t + 2; the members
// read h_{t-1} - all eight column blocks - with sc1 DMA loads once every flag of the cluster says so.  Every step has
// its own slot, so nothing is ever reused.  Slot 0 is the zero initial state: each member clears its column block
// and raises its flag to 1, which makes step 0 like every other step.  Polls are bounded; a peer that never shows
// up ends the launch with NaN in the outputs and the status word set, as in the cluster kernel.
//
// Order of a step (blocks of four k-steps; "issue" = DMA of the block after next's operand tiles into the half the
// previous block has left):  X blocks of step t (operands prefetched during the previous step) -> poll the flags
// for h_{t-1} before the last X block, whose arithmetic then covers the latency of the first H block's loads ->
// H blocks -> activations, cell update -> h_t to staging, to the plane buffer, flag -> saved activations and cell
// states to HBM from the accumulators (behind the flag, so the hand-off never waits for them).
#include "lstm_rec_common.h"
#include "gemm_planes.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include <utility>

namespace {

constexpr int FH = 256;                  // hidden size this kernel is built for
constexpr int FNC = 8;                   // members per cluster
constexpr int FROWS = 128;               // windows per cluster
constexpr int FRT = FROWS / 16;          // window tiles per cluster
constexpr int F_HALF = 4 * FRT * 2 * gp::TILE_B;   // 64 KiB: four k-steps of the A operand
constexpr int F_STAGE = FRT * 2 * 16 * 16;         // per wave: [window tile][plane][row] 16-byte slots = 4 KiB
constexpr int F_LDS = 2 * F_HALF + 4 * F_STAGE;    // 144 KiB
constexpr int F_FLAG_STRIDE = 32;                  // uint32 words between flags (one 128-byte line each)
constexpr int F_SC1 = 16;                          // cache-policy operand of the buffer builtins: agent scope
#ifndef VTS_F_NT
#define VTS_F_NT 0
#endif
constexpr int F_BULK = VTS_F_NT;                   // ... of the saved tensors' stores (2 = non-temporal)
#ifndef VTS_F_STAGGER
#define VTS_F_STAGGER 0
#endif

template <class F, int... I>
__device__ __forceinline__ void fstatic_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void fstatic_for(F&& f) { fstatic_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ bf16x8 ffrag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
typedef unsigned fvec4 __attribute__((vector_size(16)));
typedef unsigned fvu2 __attribute__((vector_size(8)));

// lane l <-> lane l ^ 8 inside every row of sixteen lanes (DPP row rotation by eight)
__device__ __forceinline__ float swap8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ float fswap1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// one dword to LDS at addr + OFF, OFF in the instruction's offset field
template <int OFF>
__device__ __forceinline__ void lds_write_b32_at(unsigned addr, unsigned v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}

// gate (0 i, 1 f, 2 g, 3 o) of column r of a wave's column tile ct
__host__ __device__ constexpr int fused_gate(int ct, int r) { return ct == 0 ? (r < 8 ? 0 : 2) : (r < 8 ? 1 : 3); }

// image: [member c][wave w][f][lane] 16-byte fragments, f = ((pks * 2 + ct) * 2 + plane), pks = k-step of the input
// part (0 .. KX-1), then KX + k-step of the recurrent part (0 .. 7):
//   W[fused_gate(ct, r) * H + 32 c + 8 w + (r & 7)][32 ks + 8 (lane >> 4) + e],  e = 0..7, r = lane & 15
// (W = W_ih, zero beyond its Cin columns, or W_hh).  Also clears the hand-off flags and the status word.
template <int KX>
__global__ void pack_fused_kernel(const float* __restrict__ W_ih, int Cin, const float* __restrict__ W_hh,
                                  u32x4* __restrict__ out, void* handoff, long n16, unsigned* status) {
  reset_handoff(handoff, n16, 0u, status);
  constexpr int NF = (KX + 8) * 4;
  const int total = FNC * 4 * NF * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int f = rest % NF; rest /= NF;
    const int w = rest & 3;
    const int c = rest >> 2;
    const int p = f & 1, ct = (f >> 1) & 1, pks = f >> 2;
    const int r = lane & 15, gq = lane >> 4;
    const int row = fused_gate(ct, r) * FH + 32 * c + 8 * w + (r & 7);
    const bool xpart = pks < KX;
    const float* src = xpart ? W_ih : W_hh;
    const int ld = xpart ? Cin : FH;
    const int k0 = 32 * (xpart ? pks : pks - KX) + 8 * gq;
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ka = k0 + 2 * q, kb = ka + 1;
      const float a = ka < ld ? src[(long)row * ld + ka] : 0.f;
      const float b = kb < ld ? src[(long)row * ld + kb] : 0.f;
      unsigned hi, lo;
      split2(a, b, hi, lo);
      o[q] = p ? lo : hi;
    }
    out[idx] = o;
  }
}

// xp: the layer's input as planes (KX column blocks of 32); hp: the layer's h sequence as planes (slot 0 = the zero
// state), written here; G, cseq: tile16 buffers that receive the gate activations and the cell states for the
// backward pass; hseq (or null): fp32 rows [B, T, 256]; hlast, clast (or null): [B, 256].  flags: [cluster][member]
// words F_FLAG_STRIDE apart, zero before the launch; err: the launch's status word.
template <int KX>
__global__ __launch_bounds__(256) void lstm_fused_fwd_kernel(
    gp::SeqPlanes xp, float* __restrict__ G, float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hlast,
    float* __restrict__ clast, const u32x4* __restrict__ Wimg, const float* __restrict__ b_ih, const float* __restrict__ b_hh,
    unsigned* __restrict__ flags, unsigned* __restrict__ err, int T, int nclusters, int spin_limit, int mute_member,
    gp::SeqPlanes hp) {
  constexpr int JT = FH / 16;
  constexpr int NXB = (KX + 3) / 4;               // blocks of the input part
  constexpr int NF = (KX + 8) * 4;
#ifndef VTS_F_XACC8
#define VTS_F_XACC8 1
#endif
  constexpr int FX_IN_ACC = KX > 8 ? KX - 8 : (KX == 8 ? VTS_F_XACC8 : 0);   // input k-steps kept in accumulator registers
  extern __shared__ __attribute__((aligned(16))) unsigned char smemf[];
  __shared__ int dead;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  // the eight members of a cluster have equal blockIdx.x % 8: one XCD under round-robin dispatch (speed only)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int c = j & 7;
  const int q = (j >> 3) * 8 + xcd;
  if (q >= nclusters) return;
  if (tid == 0) dead = 0;

  // ---- this wave's slice of both matrices, resident for the whole sequence
  u32x4 wx[KX][2][2], wh[8][2][2];
  {
    const u32x4* src = Wimg + ((size_t)(c * 4 + wave) * NF) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KX; ++ks)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int p = 0; p < 2; ++p) wx[ks][ct][p] = src[((ks * 2 + ct) * 2 + p) * 64];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int p = 0; p < 2; ++p) wh[ks][ct][p] = src[(((KX + ks) * 2 + ct) * 2 + p) * 64];
    // Every fragment is "used" here, so that the compiler waits for these loads NOW.  It does not count the DMA loads
    // issued as inline asm below; a wait of its own for a fragment's first use inside the step loop would be an
    // s_waitcnt vmcnt(small N) that, at run time, drains whatever DMA loads and stores are in flight at that point
    // of every step.  The same statements decide the register file: the recurrent part (and, with twelve input
    // k-steps, a third of the input part) is pinned to the accumulator registers - legal MFMA B operands -, so that
    // the 256 architectural registers are left to the rest of the input part, the A fragments and the cell update;
    // left to itself the allocator keeps every fragment in architectural registers and spills them.
#pragma unroll
    for (int ks = 0; ks < KX; ++ks)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (ks < FX_IN_ACC) asm volatile("" : "+a"(wx[ks][ct][p]));
          else asm volatile("" : "+v"(wx[ks][ct][p]));
        }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int p = 0; p < 2; ++p) asm volatile("" : "+a"(wh[ks][ct][p]));
  }
  const bool lo8 = r < 8;                          // this lane holds gates i, f (else g, o)
  const int hl = r & 7;                            // hidden unit inside the wave's eight
  const int jt = 2 * c + (wave >> 1);              // 16-wide hidden tile of the tile16 buffers
  const int colb = 8 * (wave & 1) + hl;            // column inside that tile
  const int hcol = 32 * c + 8 * wave + hl;         // hidden unit
  const int gate0 = lo8 ? 0 : 2, gate1 = lo8 ? 1 : 3;
  const float bias0 = b_ih[gate0 * FH + hcol] + b_hh[gate0 * FH + hcol];
  const float bias1 = b_ih[gate1 * FH + hcol] + b_hh[gate1 * FH + hcol];
  // activation of tile 0 as m * rcp(1 + exp2(k x)) + a: sigmoid (k = -log2 e, m = 1, a = 0) or tanh (2 log2 e, -2, 1)
  const float ak = lo8 ? -1.44269504088896341f : 2.88539008177792681f;
  const float am = lo8 ? 1.f : -2.f, aa = lo8 ? 0.f : 1.f;

  // ---- the cluster's part of the h planes: sc1 stores through a buffer resource, DMA loads by address
  const long hp_cluster_bytes = (long)FRT * hp.ts * hp.ncb * 2 * gp::TILE_B;
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hp.base + hp.tile((long)q * FRT, -hp.t0, 0, 0), 0,
                                                                         (int)hp_cluster_bytes, 0x00020000);
  auto hp_block = [&](int rt, int t, int cb, int p) -> int {
    return (int)(((((long)rt * hp.ts + hp.t0 + t) * hp.ncb + cb) * 2 + p) * gp::TILE_B);
  };
  unsigned* fq = flags + (size_t)q * FNC * F_FLAG_STRIDE;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(gp::lds_addr(reinterpret_cast<const char*>(smemf)));
  const unsigned voff = lane * 16;
  const unsigned pl_off = gp::slot(g, r) * 16;
  unsigned char* stage = smemf + 2 * F_HALF + wave * F_STAGE;

  // Operand tile n (= window tile * 2 + plane, 0..15) of this wave's k-step of a block into LDS half `half`: wave w
  // fetches k-step w of the block.  At one wave per SIMD every instruction of the stream costs issue time that only
  // the eight free cycles behind each MFMA can hide, and a vector memory instruction costs the most (60 cycles and
  // more): so the tiles are handed out one at a time between the MFMA groups of a block, never in a burst, and
  // their addresses are a scalar base per block plus per-lane offsets that differ by a constant per window tile.
  const long x_q = (long)((long)q * FRT * xp.ts + xp.t0) * xp.ncb * 2 * gp::TILE_B;      // the cluster's first tile, step 0
  const long h_q = (long)((long)q * FRT * hp.ts + hp.t0) * hp.ncb * 2 * gp::TILE_B;
  const unsigned x_rt = (unsigned)(xp.ts * xp.ncb * 2 * gp::TILE_B), h_rt = (unsigned)(hp.ts * hp.ncb * 2 * gp::TILE_B);
  // (the lane offset the DMA loads add their window-tile stride to is re-read through an opaque statement every
  // step: otherwise the sixteen sums are hoisted out of the step loop and, at 512 registers, spilled)
  unsigned voff_t = voff;
  auto issue_x_tile = [&](int t, int blk, int half, int n) {
    if (KX % 4 == 0 || 4 * blk + wave < KX) {
      const char* sb = xp.base + x_q + ((long)t * xp.ncb + 4 * blk + wave) * (2 * gp::TILE_B) + (n & 1) * gp::TILE_B;
      const unsigned ldst = lds0 + half * F_HALF + (wave * 2 * FRT + n) * gp::TILE_B;
      const unsigned vo = voff_t + (unsigned)(n >> 1) * x_rt;
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(ldst), "v"(vo), "s"(sb) : "memory", "m0");
    }
  };
  auto issue_h_tile = [&](int t_of_h, int blk, int half, int n) {       // h_{t_of_h}: t_of_h = -1 is the zero slot
    const char* sb = hp.base + h_q + ((long)t_of_h * hp.ncb + 4 * blk + wave) * (2 * gp::TILE_B) + (n & 1) * gp::TILE_B;
    const unsigned ldst = lds0 + half * F_HALF + (wave * 2 * FRT + n) * gp::TILE_B;
    const unsigned vo = voff_t + (unsigned)(n >> 1) * h_rt;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" :: "s"(ldst), "v"(vo), "s"(sb) : "memory", "m0");
  };
  // everything this wave has in flight has landed, and every wave is past the block before
  auto land = [&] {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // ---- slot 0 of this member's column block: the zero state; flag 1 = "slot 0 is there"
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_raw_buffer_store_b128(fvec4{0u, 0u, 0u, 0u}, hrsrc, hp_block(2 * wave + (i >> 1), -1, c, i & 1) + (int)voff,
                                           0, F_SC1);
#pragma unroll
  for (int n = 0; n < 16; ++n) issue_x_tile(0, 0, 0, n);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && c != mute_member)
    __hip_atomic_store(fq + c * F_FLAG_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  if (VTS_F_STAGGER > 0) {
    // clusters start a fraction of a step apart, so that their bursts of stores do not meet in the memory system
    const long long t0 = clock64();
    while (clock64() - t0 < (long long)(q & 3) * VTS_F_STAGGER) __builtin_amdgcn_s_sleep(8);
  }
  f32x4 acc[FRT][2];
  float cst[FRT][2];
#pragma unroll
  for (int rt = 0; rt < FRT; ++rt) cst[rt][0] = cst[rt][1] = 0.f;
  int half = 0;                                    // LDS half of the block that is computed next

  // Products of one block: A from LDS half `hf`, B = the resident fragments of k-steps 4 blk .. of part IS_H.
  // Groups of twelve MFMAs, one per (pair of window tiles rp, k-step ksl), pair-major: a pair's accumulators are
  // first touched in group rp * NKS and final after group rp * NKS + NKS - 1 of the step's last block, so the work
  // on the finished (or not yet started) pairs can ride between the groups of the others.  side(gi) runs in front of
  // group gi's MFMAs (memory instructions, initialisation, cell arithmetic of other pairs).
  auto compute = [&](auto is_h_c, auto blk_c, int hf, auto&& side) {
    constexpr bool IS_H = decltype(is_h_c)::value;
    constexpr int BLK = decltype(blk_c)::value;
    constexpr int NKS = IS_H ? 4 : (KX - 4 * BLK < 4 ? KX - 4 * BLK : 4);
    constexpr int NG = NKS * (FRT / 2);
    const unsigned char* A = smemf + hf * F_HALF + pl_off;
    u32x4 fa[2][2][2];                             // [slot][window tile of the pair][plane]
    auto fetch = [&](int gi, int slot) {
      const int rp = gi / NKS, ksl = gi % NKS;
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          fa[slot][e][p] = *reinterpret_cast<const u32x4*>(A + ((ksl * FRT + 2 * rp + e) * 2 + p) * gp::TILE_B);
    };
    fetch(0, 0);
    fstatic_for<NG>([&](auto gi_c) {
      constexpr int gi = decltype(gi_c)::value;
      constexpr int rp = gi / NKS, ksl = gi % NKS, slot = gi & 1;
      if (gi + 1 < NG) fetch(gi + 1, slot ^ 1);
      side(gi_c);
      auto B = [&](int ct, int p) -> u32x4 {
        if constexpr (IS_H) return wh[4 * BLK + ksl][ct][p];
        else return wx[4 * BLK + ksl][ct][p];
      };
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[2 * rp + e][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(fa[slot][e][1]), ffrag(B(ct, 0)), acc[2 * rp + e][ct], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[2 * rp + e][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(fa[slot][e][0]), ffrag(B(ct, 1)), acc[2 * rp + e][ct], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
          acc[2 * rp + e][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ffrag(fa[slot][e][0]), ffrag(B(ct, 0)), acc[2 * rp + e][ct], 0, 0, 0);
      // what the side work brought goes BETWEEN the MFMAs (an MFMA leaves the issue port free for half of its
      // sixteen cycles; in a lump in front of them it would cost its full issue time)
      if constexpr (IS_H && BLK == 1) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // saved tensors of step t, window tile rt (gate activations in place in acc, cell states): tile16 blocks, HBM;
  // piece 0 / 1: the wave's two column tiles, 2: the cell states
  // (buffer stores: a descriptor per tensor over the cluster's part, a per-lane offset that never changes, and the
  // (window tile, step) part as the scalar offset - no per-store address arithmetic in vector registers)
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(G + (long)q * FRT * T * (4 * JT) * 256, 0,
                                                                         FRT * T * (4 * JT) * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(cseq + (long)q * FRT * T * JT * 256, 0,
                                                                         FRT * T * JT * 1024, 0x00020000);
  const int gv0 = (gate0 * JT + jt) * 1024 + (g * 16 + colb) * 16, gv1 = (gate1 * JT + jt) * 1024 + (g * 16 + colb) * 16;
  const int cv = jt * 1024 + (g * 16 + colb) * 16 + (lo8 ? 0 : 8);
  auto store_piece = [&](int t, int rt, int piece) {
#ifdef VTS_F_NOSTORE      // timing-only ablation: what the saved tensors' stores cost
    return;
#endif
    const int bt_t = rt * T + t;
    if (piece < 2)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fvec4, acc[rt][piece]), grsrc, piece == 0 ? gv0 : gv1,
                                             bt_t * (4 * JT * 1024), F_BULK);
    else
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fvu2, float2{cst[rt][0], cst[rt][1]}), crsrc, cv, bt_t * (JT * 1024), F_BULK);
  };
  // activations in place, swap, cell update of window tile rt for this lane's two rows (4 g + (lo8 ? 0 : 2) + {0, 1}),
  // h to the staging tile
  // (the staging address is one register per lane plus a constant per (window tile, row): written as compile-time
  // offsets of the LDS instruction; sixteen separate address registers did not fit and were reloaded from scratch
  // memory in the middle of the step, each reload waiting for every DMA load in flight)
  const unsigned stage_lane = gp::lds_addr(reinterpret_cast<const char*>(stage)) + ((hl & 1) ? 256 : 0) +
                              (4 * g + (lo8 ? 0 : 2)) * 16 + (hl & ~1) * 2;
  const int pub_lane = (int)(((long)(lane >> 5) * hp.ts * hp.ncb * 2 + ((lane >> 4) & 1)) * gp::TILE_B) + gp::slot(wave, lane & 15) * 16;
  auto cell_rt = [&](auto rt_c) {
    constexpr int rt = decltype(rt_c)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[rt][0][e] = am * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ak * acc[rt][0][e])) + aa;
      acc[rt][1][e] = fast_sigmoid(acc[rt][1][e]);
    }
    float rcv[4];
    rcv[0] = swap8(lo8 ? acc[rt][0][2] : acc[rt][0][0]);
    rcv[1] = swap8(lo8 ? acc[rt][0][3] : acc[rt][0][1]);
    rcv[2] = swap8(lo8 ? acc[rt][1][2] : acc[rt][1][0]);
    rcv[3] = swap8(lo8 ? acc[rt][1][3] : acc[rt][1][1]);
    fstatic_for<2>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      const float gi = lo8 ? acc[rt][0][k] : rcv[k];
      const float gg = lo8 ? rcv[k] : acc[rt][0][k + 2];
      const float gf = lo8 ? acc[rt][1][k] : rcv[2 + k];
      const float go = lo8 ? rcv[2 + k] : acc[rt][1][k + 2];
      const float cnew = gf * cst[rt][k] + gi * gg;
      cst[rt][k] = cnew;
      const float h = go * fast_tanh(cnew);
      // h -> staging as bf16 head / tail pairs: the even lane of a pair writes the two heads, the odd one the tails
      const float other = fswap1(h);
      const bool odd = hl & 1;
      unsigned hi, lo;
      split2(odd ? other : h, odd ? h : other, hi, lo);
      lds_write_b32_at<rt * 512 + k * 16>(stage_lane, odd ? lo : hi);
    });
  };

#ifdef VTS_F_STAMP
  long long stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define FSTAMP(i) do { const long long now = clock64(); stamp[i] += now - tprev; tprev = now; } while (0)
#else
#define FSTAMP(i)
#endif
  for (int t = 0; t < T; ++t) {
    // ---- first block of the step: its operands were issued a block ago
    FSTAMP(13);
    land();
    asm volatile("" : "+v"(voff_t));
    FSTAMP(0);
    // The saved tensors of the step before leave now, behind the hand-off: a pair's pieces before that pair's
    // accumulators are set to the bias again - pair 0's here, pair rp + 1's between the groups of pair rp.
    if (t > 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) store_piece(t - 1, i / 3, i % 3);
    }
    FSTAMP(12);
    bool gone = false;
    fstatic_for<NXB>([&](auto xb_c) {
      constexpr int xb = decltype(xb_c)::value;
      constexpr bool last = xb == NXB - 1;
      constexpr int NKS = KX - 4 * xb < 4 ? KX - 4 * xb : 4;
      constexpr int NG = NKS * (FRT / 2), PER = 16 / NG;
      constexpr int SPER = (6 + NKS - 1) / NKS;      // store pieces of the next pair per group
      static_assert(NG * PER == 16, "operand tiles divide over the groups");
      if (gone) return;
      if (xb > 0) land();
      FSTAMP(2 * xb);
      if (last) {
        // h_{t-1} is slot t: every member's flag must have reached t + 1 before its tiles are asked for
        if (wave == 0) {
          int spins = 0;
          bool ok = false;
          do {
            const unsigned seen = lane < FNC ? __hip_atomic_load(fq + lane * F_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : 0xffffffffu;
            ok = __all(seen >= (unsigned)(t + 1));
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
          } while (++spins < spin_limit);
          if (!ok && lane == 0) {
            dead = 1;
            atomicOr(err, 1u);
          }
        }
        __syncthreads();
        if (dead) { gone = true; return; }
      }
      FSTAMP(14);
      compute(std::false_type{}, xb_c, half, [&](auto gi_c) {
        constexpr int gi = decltype(gi_c)::value;
        constexpr int rp = gi / NKS, ksl = gi % NKS;
#pragma unroll
        for (int n = gi * PER; n < (gi + 1) * PER; ++n) {
          if (last) issue_h_tile(t - 1, 0, half ^ 1, n);
          else issue_x_tile(t, xb + 1, half ^ 1, n);
        }
        if (xb == 0) {
          if (ksl == 0) {
            acc[2 * rp][0] = acc[2 * rp + 1][0] = f32x4{bias0, bias0, bias0, bias0};
            acc[2 * rp][1] = acc[2 * rp + 1][1] = f32x4{bias1, bias1, bias1, bias1};
          }
          if (t > 0 && rp + 1 < FRT / 2) {
#pragma unroll
            for (int i = ksl * SPER; i < (ksl + 1) * SPER && i < 6; ++i) store_piece(t - 1, 2 * (rp + 1) + i / 3, i % 3);
          }
        }
      });
      FSTAMP(2 * xb + 1);
      half ^= 1;
    });
    if (gone) break;
    land();
    FSTAMP(6);
    compute(std::true_type{}, std::integral_constant<int, 0>{}, half, [&](auto gi_c) {
      issue_h_tile(t - 1, 1, half ^ 1, decltype(gi_c)::value);
    });
    FSTAMP(7);
    half ^= 1;
    land();
    FSTAMP(8);
    // the last block: a pair whose products are complete goes through its cell update between the groups of the next
    compute(std::true_type{}, std::integral_constant<int, 1>{}, half, [&](auto gi_c) {
      constexpr int gi = decltype(gi_c)::value;
      constexpr int rp = gi / 4, ksl = gi % 4;
      issue_x_tile(t + 1 < T ? t + 1 : t, 0, half ^ 1, gi);      // (the last step fetches its own tiles again: no branch here)
      if constexpr (rp > 0 && ksl == 0) cell_rt(std::integral_constant<int, 2 * (rp > 0 ? rp - 1 : 0)>{});
      if constexpr (rp > 0 && ksl == 2) cell_rt(std::integral_constant<int, 2 * (rp > 0 ? rp - 1 : 0) + 1>{});
    });
    FSTAMP(9);
    half ^= 1;
    cell_rt(std::integral_constant<int, FRT - 2>{});
    cell_rt(std::integral_constant<int, FRT - 1>{});
    FSTAMP(10);
    // ---- publish h_t: this wave's k-group (eight units = one 16-byte slot per row) of column block c
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // staging slot i * 64 + lane = [window tile 2 i + (lane >> 5)][plane (lane >> 4) & 1][row lane & 15]: the lane's
      // part of the address never changes, the (i, step) part is the scalar offset
      const u32x4 v = *reinterpret_cast<const u32x4*>(stage + (i * 64 + lane) * 16);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fvec4, v), hrsrc, pub_lane,
                                             (int)(((long)(2 * i) * hp.ts + hp.t0 + t) * hp.ncb + c) * (2 * gp::TILE_B), F_SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && c != mute_member)
      __hip_atomic_store(fq + c * F_FLAG_STRIDE, (unsigned)(t + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    FSTAMP(11);
    // ---- optional fp32 outputs (a caller that wants the sequence as rows; the final state).  h is read back from the
    // staging tile as head + tail - the value every consumer of the plane output sees (h to 2^-17 relative) - so that
    // no copy of it occupies registers through the step.  The addresses are worked out here, from operands the
    // optimiser cannot see through: hoisted out of the step loop they would occupy a hundred registers.
    if (hseq != nullptr || t == T - 1) {
      int hc = hcol, row0 = 4 * g + (lo8 ? 0 : 2);
      asm volatile("" : "+v"(hc), "+v"(row0));
#pragma unroll
      for (int rt = 0; rt < FRT; ++rt)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const long b = ((long)q * FRT + rt) * 16 + row0 + k;
          const unsigned short hh = *reinterpret_cast<const unsigned short*>(stage + ((rt * 2 + 0) * 16 + row0 + k) * 16 + hl * 2);
          const unsigned short hl_ = *reinterpret_cast<const unsigned short*>(stage + ((rt * 2 + 1) * 16 + row0 + k) * 16 + hl * 2);
          const float h = __uint_as_float((unsigned)hh << 16) + __uint_as_float((unsigned)hl_ << 16);
          if (hseq != nullptr) hseq[(b * T + t) * FH + hc] = h;
          if (t == T - 1) {
            if (hlast != nullptr) hlast[b * FH + hc] = h;
            if (clast != nullptr) clast[b * FH + hc] = cst[rt][k];
          }
        }
    }
  }
#ifdef VTS_F_STAMP
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 16; ++i) reinterpret_cast<long long*>(err + 2)[i] = stamp[i];
#endif
  if (!dead) {
#pragma unroll
    for (int i = 0; i < 3 * FRT; ++i) store_piece(T - 1, i / 3, i % 3);
  } else {
    // a peer never published: besides the status word, poison what the layers above read, so that the loss of such
    // a step is NaN instead of a plausible number
    const float poison = __builtin_nanf("");
#pragma unroll
    for (int rt = 0; rt < FRT; ++rt)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const long b = ((long)q * FRT + rt) * 16 + 4 * g + (lo8 ? 0 : 2) + k;
        if (hseq != nullptr) hseq[(b * T + T - 1) * FH + hcol] = poison;
        if (hlast != nullptr) hlast[b * FH + hcol] = poison;
        if (clast != nullptr) clast[b * FH + hcol] = poison;
      }
    // ... and the last step of the plane output, which is what the next layer and the weight gradients read
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = i * 64 + lane;
      const int rt = s >> 5, p = (s >> 4) & 1, row = s & 15;
      __builtin_amdgcn_raw_buffer_store_b128(fvec4{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u}, hrsrc,
                                             hp_block(rt, T - 1, c, p) + gp::slot(wave, row) * 16, 0, F_SC1);
    }
  }
}

std::atomic<int> g_fused_fault{-1};

template <int KX>
int launch_fused(const gp::SeqPlanes& xp, float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* w_ih,
                 int Cin, const float* w_hh, const float* b_ih, const float* b_hh, u32x4* img, unsigned* flags, long flag16,
                 unsigned* err, int B, int T, const gp::SeqPlanes& hp, hipStream_t stream) {
  const int nclusters = B / FROWS;
  const void* kern = reinterpret_cast<const void*>(lstm_fused_fwd_kernel<KX>);
  static vts_lds_mask lds_done{0};
  if (int rc = vts_reserve_lds(kern, F_LDS, &lds_done, "lstm_fused_fwd")) return rc;
  const unsigned grid = 64 * vts_cdiv(nclusters, 8);
  if (vts_stream_is_capturing(stream)) {
    vts_set_error("lstm_fused_fwd: cooperative launches cannot be recorded into a graph");
    return VTS_ERR_UNSUPPORTED;
  }
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, F_LDS) != hipSuccess || (long)per_cu * cus < (long)grid) {
    (void)hipGetLastError();
    vts_set_error("lstm_fused_fwd: %u workgroups cannot be resident at once (%d per CU x %d CUs)", grid, per_cu, cus);
    return VTS_ERR_UNSUPPORTED;
  }
  constexpr int NF = (KX + 8) * 4;
  hipLaunchKernelGGL(pack_fused_kernel<KX>, dim3(vts_cdiv(FNC * 4 * NF * 64, 256)), dim3(256), 0, stream, w_ih, Cin, w_hh, img,
                     flags, flag16, err);
  VTS_CHECK_LAUNCH("pack_fused");
  int mute_member = g_fused_fault.load();
  int spin_limit = mute_member >= 0 ? 1 << 10 : 1 << 21;
  int nc = nclusters;
  gp::SeqPlanes xpa = xp, hpa = hp;
  void* args[] = {&xpa, &G, &cseq, &hseq, &hlast, &clast, &img, &b_ih, &b_hh, &flags, &err, &T, &nc, &spin_limit, &mute_member, &hpa};
  const hipError_t e = hipLaunchCooperativeKernel(kern, dim3(grid), dim3(256), args, F_LDS, stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    vts_set_error("lstm_fused_fwd: cooperative launch refused: %s", hipGetErrorString(e));
    return VTS_ERR_UNSUPPORTED;      // whatever the reason (too large, another process holds the device's cooperative queue): nothing was launched, the caller runs the kernel that waits for nobody
  }
  return VTS_OK;
}

}  // namespace

// Test hook: member `member` (0..7) of every cluster withholds its flags and the polls give up after 1024 rounds;
// negative (the default) = off.  Process-wide.
void vts_lstm_fused_set_fault(int member) { g_fused_fault.store(member < 0 ? -1 : (member & (FNC - 1))); }

// Column blocks (of 32) of the input planes this kernel is built for: 2 (up to 64 features), 8 (256), 12 (384).
static int fused_kx(long Cin) {
  const int ncb = (int)((Cin + 63) / 64 * 64) / 32;
  return (ncb == 2 || ncb == 8 || ncb == 12) ? ncb : 0;
}
// Shape test: hidden 256, whole clusters of 128 windows, a planned input width and a grid that is resident at once.
bool vts_lstm_fused_usable(long B, long T, long Cin, long H) {
  if (H != FH || B < FROWS || (B % FROWS) != 0 || T < 1 || fused_kx(Cin) == 0) return false;
  if (T >= 4000) return false;      // 32-bit offsets into a cluster's part of the planes and of the gate buffer
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  return 64 * vts_cdiv(B / FROWS, 8) <= cus;
}
size_t vts_lstm_fused_image_bytes(long Cin) { return (size_t)FNC * 4 * (fused_kx(Cin) + 8) * 4 * 1024; }
size_t vts_lstm_fused_flag_bytes(long B) { return ((size_t)(B / FROWS) * FNC + 1) * F_FLAG_STRIDE * sizeof(unsigned); }

// Returns VTS_OK, an error, or VTS_ERR_UNSUPPORTED when the runtime will not guarantee that the whole grid is
// resident at once or the stream is recording a graph (the occupancy and capture tests refuse before anything is
// enqueued; a refusal by the cooperative launch itself comes after the packing kernel is in the queue, which has
// then overwritten the image, the flags and the status word - all of them this launch's own scratch).
int vts_lstm_fused_fwd(const void* x_planes, int x_is_hseq, const float* w_ih, const float* b_ih, const float* b_hh,
                       const float* w_hh, float* G, float* cseq, float* hseq, void* h_planes, float* hlast, float* clast,
                       void* image, void* flags, unsigned* err, long B, long T, long Cin, hipStream_t stream) {
  const int kx = fused_kx(Cin);
  gp::SeqPlanes xp{reinterpret_cast<char*>(const_cast<void*>(x_planes)), kx, (int)T + (x_is_hseq ? 1 : 0), x_is_hseq ? 1 : 0};
  gp::SeqPlanes hp{reinterpret_cast<char*>(h_planes), 8, (int)T + 1, 1};
  if (!vts_aligned16(image) || !vts_aligned16(flags)) {
    vts_set_error("lstm_fused_fwd: the weight image and the flag words must be 16-byte aligned");
    return VTS_ERR_ARG;
  }
  u32x4* img = reinterpret_cast<u32x4*>(image);
  unsigned* fl = reinterpret_cast<unsigned*>(flags);
  const long flag16 = (long)(vts_lstm_fused_flag_bytes(B) / 16);
  switch (kx) {
    case 2: return launch_fused<2>(xp, G, cseq, hseq, hlast, clast, w_ih, (int)Cin, w_hh, b_ih, b_hh, img, fl, flag16, err, (int)B, (int)T, hp, stream);
    case 8: return launch_fused<8>(xp, G, cseq, hseq, hlast, clast, w_ih, (int)Cin, w_hh, b_ih, b_hh, img, fl, flag16, err, (int)B, (int)T, hp, stream);
    case 12: return launch_fused<12>(xp, G, cseq, hseq, hlast, clast, w_ih, (int)Cin, w_hh, b_ih, b_hh, img, fl, flag16, err, (int)B, (int)T, hp, stream);
  }
  vts_set_error("lstm_fused_fwd: input width %ld is not planned", Cin);
  return VTS_ERR_UNSUPPORTED;
}
