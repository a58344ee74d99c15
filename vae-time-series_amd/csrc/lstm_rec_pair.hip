// Backward LSTM recurrence, bf16x3, hidden 256: two workgroups (two CUs) share 32 windows and half of W_hh each.
//
// The one-workgroup kernel (lstm_rec3_bwd_staged_kernel) pulls the whole 1 MiB image of W_hh through its CU's
// vector-memory path every step, and that path - 64 bytes per clock - sets its step time.  Here member m of a
// pair owns hidden units [128 m, 128 m + 128): it runs the cell backward of those units for the pair's 32 windows
// (two window tiles: the same 16 blocks of 16 x 16 per workgroup as before) and so holds the gate gradients
// dA_t[:, its 512 columns] in LDS.  Of the product dh_{t-1} = dA_t . W_hh it computes the part its own columns
// contribute - K = 512 of the 1024, all 256 outputs - so it needs only the rows of W_hh that belong to its
// hidden units (512 KiB, a quarter of them resident in registers), and every fragment of W_hh it streams serves
// two window tiles.  What the members owe each other afterwards is the partial sum for the other's 128 outputs:
// 16 KiB of fp32 per member and step through the L2.
//
// Hiding the hand-off: a step's product runs in two passes over the member's K, first the peer's 128 outputs, then
// its own.  The peer's half goes out after the first pass; the other's arrives while the second pass runs.
// There are no flags: a slot of the exchange buffer holds the pattern EMPTY until the producer's 16-byte stores
// land, the consumer reads the slot (sc1 loads, around the L1) until no word of it is EMPTY and stores EMPTY back.
// (Measured alternatives for two members that share an L2: plain stores change nothing - the sc1 read is the
// slow part, about 3 000 clocks; a plain read behind `buffer_inv sc1` drops the W_hh image from the L2 and doubles
// the step; behind `buffer_inv sc0` it returns stale lines of the L1.)
// Where the slots live.  Agent-scope (sc1) stores are written through the L2 to memory and sc1 reads of such lines
// come back from there: 1.2 GB of HBM traffic per launch at 4096 windows, a third on top of what the kernel must
// move.  Two members that share an L2 need none of it - a plain store stops in the L2, where the peer's sc1 read
// (which only goes around its own L1) finds it.  Workgroup b is expected on XCD b mod 8, which puts a pair on one
// XCD, but nothing guarantees it; so the members first tell each other where they run (the XCC_ID hardware register,
// exchanged with sc1 accesses) and use plain stores only if the two values are equal, sc1 stores otherwise.
// Three slots in rotation: between a reset and the producer's next store into the same slot lie a full drain of the
// consumer's memory operations (the wait of the next step's read) and two hand-offs, so the reset cannot overtake it.
// Reads are bounded: when the peer never shows up the wave goes on with NaN and sets the launch's status word.
//
// Wave w of member m: hidden tiles 8 m + 2 w + {0, 1} in the cell phase; in the product the output tiles
// 8 (1 - m) + 2 w + {0, 1} (first pass) and 8 m + 2 w + {0, 1} (second pass) - the same wave of the peer consumes
// what this wave publishes, so the hand-off is wave to wave and needs no workgroup barrier.
// The W_hh ring, the LDS-DMA staging of the saved tensors and the coefficient arithmetic inside the product are
// those of lstm_rec3_bwd_staged_kernel (lstm_rec_bf16x3.hip), where they are described.
#include "lstm_rec_common.h"
#include "gemm_planes.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace {

constexpr int PH = 256;                       // hidden size this kernel is built for
constexpr int PJT = PH / 16;                  // hidden tiles
constexpr int PROWS = 32;                     // windows per pair
constexpr int UT = 8;                         // 16-byte fragments per ring unit and lane: 2 k-steps x (hi, lo) x 2 output tiles
constexpr int NU = 16;                        // units per step and wave: 8 per pass
constexpr int NI = 4;                         // 16 x 16 blocks per wave: i = window tile * 2 + hidden tile
constexpr int NB_G = 4 * NI, NB_C = NI;       // staged tiles per step: gates, cells (and NI more with a dh sequence)
constexpr int ROWB = 512 * 2;                 // bytes per row of a dA plane (the member's 512 columns, bf16)
constexpr int PLANE_M = 16 * ROWB;            // one plane of one window tile
constexpr int DA_LDS = 2 * 2 * PLANE_M;       // [hi | lo][window tile] = 64 KiB
constexpr int STG = 6 * NI * 1024;            // staging bytes per wave
constexpr int P_LDS = DA_LDS + 4 * STG;       // 160 KiB
constexpr int XW = NI * 1024;                 // bytes one wave publishes per step
constexpr int XSLOTS = 3;
constexpr int X_SLOTS_B = XSLOTS * 2 * 4 * XW;   // [slot][destination member][wave] = 96 KiB per pair
constexpr int X_PAIR = X_SLOTS_B + 256;       // ... and, behind them, one word per member: where it runs (below)
// "nothing here yet".  As a float this is a NaN with an all-ones payload, which no arithmetic instruction of this
// GPU produces (they return the canonical quiet NaN 0x7fc00000 or pass an operand's NaN on): a partial sum can only
// carry it if the caller's own tensors held that exact bit pattern, and the launch then ends as a missed hand-off
// does - NaN gradients and bit 1 of the status word - after the bounded reads have run out.
constexpr unsigned EMPTY = 0xffffffffu;
constexpr int SC1 = 16;                       // cache-policy operand of the buffer builtins: agent scope
typedef unsigned pvec4 __attribute__((vector_size(16)));
__device__ __forceinline__ bf16x8 pfrag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// image: [member m][wave w][unit u][f = kl*4 + plane*2 + jl][lane] 16-byte fragments ->
//   W_hh[(k' & 3) * H + (k' >> 2)][nt * 16 + (lane & 15)],  k' = 512 m + 32 ks + 8 (lane >> 4) + e (e = 0..7: the
//   column order 4 * hidden + gate in which the cell phase writes dA), ks = 2 (u & 7) + kl,
//   nt = 8 (u < 8 ? 1 - m : m) + 2 w + jl
__global__ void pack_whh3p_kernel(const float* __restrict__ W, u32x4* __restrict__ out, void* handoff, long n16,
                                  unsigned* status) {
  reset_handoff(handoff, n16, EMPTY, status);
  const int total = PH * PH;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int f = rest & 7; rest >>= 3;
    const int u = rest & 15; rest >>= 4;
    const int w = rest & 3;
    const int m = rest >> 2;
    const int kl = f >> 2, p = (f >> 1) & 1, jl = f & 1;
    const int ks = 2 * (u & 7) + kl;
    const int nt = 8 * (u < 8 ? 1 - m : m) + 2 * w + jl;
    const int col = nt * 16 + (lane & 15);
    const int k0 = 512 * m + 32 * ks + 8 * (lane >> 4);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ka = k0 + 2 * q, kb = ka + 1;
      unsigned hi, lo;
      split2(W[((ka & 3) * PH + (ka >> 2)) * PH + col], W[((kb & 3) * PH + (kb >> 2)) * PH + col], hi, lo);
      o[q] = p ? lo : hi;
    }
    out[idx] = o;
  }
}

// G, cseq, dhseq: tile16 buffers as for lstm_rec3_bwd_kernel; dhlast / dclast: tile16 [B, 256] blocks or null;
// dgp: the gate-gradient planes (the only form in which this kernel writes dA); xchg: X_PAIR bytes per pair, every
// word EMPTY before the launch; err: the launch's status word.  Grid: workgroup b -> XCD b & 7 (the hardware deals
// workgroups round-robin), member (b >> 3) & 1, pair (b >> 4) * 8 + (b & 7): both members on one XCD.
// DH: a dh sequence is given (four more staged tiles per step - the hand-counted waits need the count at compile time).
template <bool DH>
__global__ __launch_bounds__(256) void lstm_rec3p_bwd_kernel(
    const float* __restrict__ G, const float* __restrict__ cseq, const float* __restrict__ dhseq,
    const float* __restrict__ dhlast, const float* __restrict__ dclast, float* __restrict__ dbpart,
    const u32x4* __restrict__ Wp, gp::SeqPlanes dgp, unsigned char* __restrict__ xchg, unsigned* __restrict__ err, int T,
    int npairs, int spin_limit, int mute_member) {
  constexpr int NB = NB_G + NB_C + (DH ? NI : 0);      // LDS-DMA loads per batch
  constexpr int RES = 4;                               // units of the image resident in registers: 4, 5, 14, 15
  constexpr bool RING_AG = true;                       // the ring lives in accumulator registers (the arch VGPRs serve the cell arithmetic)
  constexpr int PAY_AT = 11;                // the read of the peer's half is issued after this unit
  constexpr int PRE = NI - 1;                          // blocks whose coefficients are worked out inside the product
  static_assert(3 * UT + NB <= 63, "vmcnt is a six-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smemp[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int m = jb & 1;
  const int q = (jb >> 1) * 8 + xcd;
  if (q >= npairs) return;
  unsigned char* const da_s = smemp;
  unsigned char* const stg = smemp + DA_LDS + wave * STG;
  const unsigned stg_lds = __builtin_amdgcn_readfirstlane(gp::lds_addr(reinterpret_cast<const char*>(stg)));
  const int jt0 = 8 * m + 2 * wave;                    // first of this wave's two hidden tiles
  const long bt0 = 2L * q;                             // first of the pair's two window tiles

  f32x4 dc[NI], dh[NI];
  float dbs[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const long blk = ((bt0 + (i >> 1)) * PJT + jt0 + (i & 1)) * 256 + lane * 4;
    dh[i] = dhlast != nullptr ? *reinterpret_cast<const f32x4*>(dhlast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
    dc[i] = dclast != nullptr ? *reinterpret_cast<const f32x4*>(dclast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) dbs[i][k] = 0.f;
  }
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(Wp) + (size_t)(m * 4 + wave) * (NU * UT * 1024);
  const unsigned voff = lane * 16;
  const unsigned pl_off = gp::slot(g, r) * 16;
  auto unit_ptr = [&](int u) { return wbase + (size_t)(u * UT) * 1024; };
  // the pair's exchange buffer; this wave writes the peer's wave w and reads its own
  const __amdgpu_buffer_rsrc_t xrsrc =
      __builtin_amdgcn_make_buffer_rsrc(xchg + (size_t)q * X_PAIR, 0, X_PAIR, 0x00020000);
  const int tx0 = ((1 - m) * 4 + wave) * XW + lane * 16, rx0 = (m * 4 + wave) * XW + lane * 16;
  // the same descriptor as four SGPRs, for the read that is issued as inline asm
  const unsigned long long xbase = reinterpret_cast<unsigned long long>(xchg) + (unsigned long long)q * X_PAIR;
  const u32x4 xdesc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xbase),
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xbase >> 32)) & 0xffffu, (unsigned)X_PAIR,
                       0x00020000u};
  bool dead = false;
  // Do the members share an L2?  Every wave asks for itself (no barrier, no LDS): lane 0's value counts.
  bool same_l2;
  {
    unsigned here;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(here));
    here = (here & 0x00ffffffu) | 0x01000000u;                      // never EMPTY
    const int hs = X_SLOTS_B + wave * 16;                           // [wave][member] words
    if (lane == 0 && m != mute_member) __builtin_amdgcn_raw_buffer_store_b32(here, xrsrc, hs + m * 4, 0, SC1);
    unsigned peer = EMPTY;
    for (int spins = 0; spins < spin_limit; ++spins) {
      peer = __builtin_amdgcn_raw_buffer_load_b32(xrsrc, hs + (1 - m) * 4, 0, SC1);
      peer = __builtin_amdgcn_readfirstlane(peer);
      if (peer != EMPTY) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (peer == EMPTY) {
      if (lane == 0) atomicOr(err, 2u);
      dead = true;
    }
    same_l2 = peer == here;
  }

  // LDS traffic of this wave done, then the barrier (no fence: a fence would wait for the ring and the batch)
  auto wg_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto dma = [&](unsigned lds_off, const float* src) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_off), "v"(voff), "s"(src) : "memory", "m0");
  };
  // the batch of step t: G_t, c_{t-1} and dh_t of this wave's four blocks into staging, block by block (the cell
  // phase issues a block's share between its own blocks: issued back to back, the 24 loads of a wave hold up
  // the wave for as long as the CU's memory path needs to take them).  Always NB loads per step.
  auto issue_block = [&](int i, int t) {
    const int tc = t >= 1 ? t - 1 : 0;
    const long bt = bt0 + (i >> 1);
    const int jt = jt0 + (i & 1);
    const float* Gt = G + ((bt * T + t) * (4 * PJT) + jt) * 256;
#pragma unroll
    for (int k = 0; k < 4; ++k) dma(stg_lds + (i * 4 + k) * 1024, Gt + (long)(k * PJT) * 256);
    dma(stg_lds + (NB_G + i) * 1024, cseq + ((bt * T + tc) * PJT + jt) * 256);
    if (DH) dma(stg_lds + (NB_G + NB_C + i) * 1024, dhseq + ((bt * T + t) * PJT + jt) * 256);
  };
  auto issue_batch = [&](int t) {
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_block(i, t);
  };

  // coefficients of the cell backward that do not depend on dh (see lstm_rec3_bwd_staged_kernel)
  f32x4 kA[NI], kO[NI], kI[NI], kF[NI], kG[NI], kf[NI], tcn[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const f32x4 c = ld_stream(cseq + (((bt0 + (i >> 1)) * T + T - 1) * PJT + jt0 + (i & 1)) * 256 + lane * 4);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) tcn[i][reg] = fast_tanh(c[reg]);
  }

  u32x4 w0[UT], w1[UT], w2[UT], w3[UT];
#pragma unroll
  for (int i = 0; i < UT; ++i) w0[i] = w1[i] = w2[i] = w3[i] = u32x4{0u, 0u, 0u, 0u};
  // the resident units, loaded the way the ring is (as plain loads the compiler keeps guarding their registers with
  // waits of its own inside the loop - counted on the operations it knows of, so far too strict)
  u32x4 wr[RES][UT];
#pragma unroll
  for (int i = 0; i < RES; ++i) {
#pragma unroll
    for (int j = 0; j < UT; ++j) wr[i][j] = u32x4{0u, 0u, 0u, 0u};
    w_issue<UT, RING_AG>(wr[i], unit_ptr(i < 2 ? 4 + i : 12 + i), voff);
  }
#pragma unroll
  for (int i = 0; i < RES; ++i) w_wait<0, UT, RING_AG>(wr[i]);
  w_issue<UT, RING_AG>(w0, unit_ptr(0), voff);
  w_issue<UT, RING_AG>(w1, unit_ptr(1), voff);
  w_issue<UT, RING_AG>(w2, unit_ptr(2), voff);
  w_issue<UT, RING_AG>(w3, unit_ptr(3), voff);
  issue_batch(T - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  auto premath = [&](int i, int t) {
    asm volatile("" ::: "memory");     // the staging reads stay behind the wait that proved the batch landed
    const unsigned char* sj = stg + lane * 16;
    const f32x4 gi = *reinterpret_cast<const f32x4*>(sj + (i * 4 + 0) * 1024);
    const f32x4 gf = *reinterpret_cast<const f32x4*>(sj + (i * 4 + 1) * 1024);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(sj + (i * 4 + 2) * 1024);
    const f32x4 go = *reinterpret_cast<const f32x4*>(sj + (i * 4 + 3) * 1024);
    f32x4 cp = *reinterpret_cast<const f32x4*>(sj + (NB_G + i) * 1024);
    if (t < 1) cp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float tc = tcn[i][reg];
      kA[i][reg] = go[reg] * (1.f - tc * tc);
      kO[i][reg] = tc * (go[reg] * (1.f - go[reg]));
      kI[i][reg] = gg[reg] * (gi[reg] * (1.f - gi[reg]));
      kF[i][reg] = cp[reg] * (gf[reg] * (1.f - gf[reg]));
      kG[i][reg] = gi[reg] * (1.f - gg[reg] * gg[reg]);
      kf[i][reg] = gf[reg];
      tcn[i][reg] = fast_tanh(cp[reg]);
    }
    // computed here, not where the cell phase first uses them (the optimiser sinks arithmetic past the branches of
    // the exchange otherwise - out of the product, where it is free, into the critical path behind it)
    asm volatile("" : "+v"(kA[i]), "+v"(kO[i]), "+v"(kI[i]), "+v"(kF[i]), "+v"(kG[i]), "+v"(kf[i]), "+v"(tcn[i]));
  };
  auto take_dh = [&] {
    asm volatile("" ::: "memory");
    if (DH) {
#pragma unroll
      for (int i = 0; i < NI; ++i) dh[i] += *reinterpret_cast<const f32x4*>(stg + (NB_G + NB_C + i) * 1024 + lane * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // cell backward of this wave's four blocks: gate gradients as bf16 planes into LDS (column 4 * local hidden +
  // gate, 16-byte chunks XOR-ed with the row), their plane tiles read back in fragment order and stored
  auto cell_bwd = [&](int t, int tnext) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      issue_block(i, tnext);                     // (the staging area is free: every coefficient is in registers)
      const int mt = i >> 1, jl = i & 1;
      const int jloc = 2 * wave + jl;            // hidden tile within the member's half
      unsigned char* const da_m = da_s + mt * PLANE_M;
      char* const dg_t = dgp.base + dgp.tile(bt0 + mt, t, 16 * m + 4 * wave, 0) + pl_off;
      const f32x4 dht = dh[i];
      f32x4 dai, daf, dag, dao;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float dcc = dc[i][reg] + dht[reg] * kA[i][reg];
        dc[i][reg] = dcc * kf[i][reg];
        dai[reg] = dcc * kI[i][reg];
        daf[reg] = dcc * kF[i][reg];
        dag[reg] = dcc * kG[i][reg];
        dao[reg] = dht[reg] * kO[i][reg];
        uint2 hi, lo;
        split4(dai[reg], daf[reg], dag[reg], dao[reg], hi, lo);
        const int row = 4 * g + reg;
        const int j = jloc * 16 + r;
        const int off = row * ROWB + ((((j >> 1) ^ row) << 4) | ((j & 1) << 3));
        *reinterpret_cast<uint2*>(da_m + off) = hi;
        *reinterpret_cast<uint2*>(da_m + 2 * PLANE_M + off) = lo;
      }
      dbs[i][0] += (dai[0] + dai[1]) + (dai[2] + dai[3]);
      dbs[i][1] += (daf[0] + daf[1]) + (daf[2] + daf[3]);
      dbs[i][2] += (dag[0] + dag[1]) + (dag[2] + dag[3]);
      dbs[i][3] += (dao[0] + dao[1]) + (dao[2] + dao[3]);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        u32x4 pf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          pf[p] = *reinterpret_cast<const u32x4*>(da_m + p * 2 * PLANE_M + r * ROWB + (((4 * (2 * jloc + hf) + g) ^ r) << 4));
#pragma unroll
        for (int p = 0; p < 2; ++p)
          __builtin_nontemporal_store(pf[p], reinterpret_cast<u32x4*>(dg_t + (4 * jl + 2 * hf + p) * gp::TILE_B));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Pairs start a quarter of a step apart: in lockstep every CU of the device would ask the HBM for its saved
  // tensors in the same fifth of the step (and the L2 for W_hh in the rest); staggered, the demand is even.
#ifndef VTS_RP_SKEW
#define VTS_RP_SKEW 78
#endif
  for (int k = 0; k < (q & 3); ++k) __builtin_amdgcn_s_sleep(VTS_RP_SKEW);
#pragma unroll
  for (int i = 0; i < NI; ++i) premath(i, T - 1);
  take_dh();
  cell_bwd(T - 1, T >= 2 ? T - 2 : 0);

#ifdef VTS_RP_STAMP
  // timing experiments: ticks per phase, summed over the steps, written over the first bias-gradient partials
  long long pstamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long pprev = clock64();
#define PSTAMP(i) do { const long long now = clock64(); pstamp[i] += now - pprev; pprev = now; } while (0)
#else
#define PSTAMP(i)
#endif
  int slot = 0;
  for (int t = T - 2; t >= 0; --t) {
    PSTAMP(9);                                    // (cell backward of the step before)
    wg_barrier();                                 // dA_{t+1} is complete
    PSTAMP(0);
    f32x4 acc[NI];
    u32x4 fa_h[2], fa_l[2];                       // A fragments (hi, lo plane) of window tile 0 / 1
    auto lds_a = [&](int plane, int mt, int ks) {
      return *reinterpret_cast<const u32x4*>(da_s + (plane * 2 + mt) * PLANE_M + r * ROWB + (((4 * ks + g) ^ r) << 4));
    };
    // one unit = two k-steps of this wave's two output tiles for both window tiles: 24 MFMAs; allowed < 0: resident.
    // The A fragments run one group of six MFMAs ahead: while window tile 0's products issue, tile 1's fragments
    // of the same k-step are on their way from LDS, and while tile 1's issue, tile 0's of the next k-step (kn = the
    // unit that runs next, -1 = none).  One window tile at a time: its two output tiles alternate, a dependent
    // MFMA is two issues away.
    auto unit = [&](u32x4 (&w)[UT], int u, int kn, auto allowed) {
#pragma unroll
      for (int kl = 0; kl < 2; ++kl) {
        const int ks = 2 * (u & 7) + kl;
        fa_h[1] = lds_a(0, 1, ks);
        fa_l[1] = lds_a(1, 1, ks);
        if constexpr (decltype(allowed)::value >= 0) {
          if (kl == 0) w_wait<decltype(allowed)::value, UT, RING_AG>(w);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jl = 0; jl < 2; ++jl)
            acc[2 * mt + jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag(fa_l[mt]), pfrag(w[kl * 4 + jl]), acc[2 * mt + jl], 0, 0, 0);
#pragma unroll
          for (int jl = 0; jl < 2; ++jl)
            acc[2 * mt + jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag(fa_h[mt]), pfrag(w[kl * 4 + 2 + jl]), acc[2 * mt + jl], 0, 0, 0);
#pragma unroll
          for (int jl = 0; jl < 2; ++jl)
            acc[2 * mt + jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfrag(fa_h[mt]), pfrag(w[kl * 4 + jl]), acc[2 * mt + jl], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (mt == 0) {
            const int nks = kl == 0 ? ks + 1 : (kn >= 0 ? 2 * (kn & 7) : -1);
            if (nks >= 0) {
              fa_h[0] = lds_a(0, 0, nks);
              fa_l[0] = lds_a(1, 0, nks);
            }
          }
        }
      }
    };
    // Queue on entry, oldest first: units 0..3 in w0..w3 (issued during the previous product; landed - the wait
    // of the exchange read drained the queue behind them), then the batch of step t.  The streamed units run
    // 0 1 2 3 6 7 | 8 9 10 11 12 13, each buffer refilled with the unit four streamed ones ahead, the last four
    // with the next step's 0..3: at every wait three units are younger than the awaited one (and during the
    // first four the batch, which unit 6's wait retires).  Loads and stores issued by the compiler in between
    // (the exchange) only make a given count stricter.
    constexpr std::integral_constant<int, 3 * UT> steady{};
    constexpr std::integral_constant<int, 3 * UT + NB> first{};
    constexpr std::integral_constant<int, -1> resident{};
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    fa_h[0] = lds_a(0, 0, 0);
    fa_l[0] = lds_a(1, 0, 0);
    // ---- first pass: the peer's 128 outputs
    unit(w0, 0, 1, first);
    w_issue<UT, RING_AG>(w0, unit_ptr(6), voff);
    unit(w1, 1, 2, first);
    w_issue<UT, RING_AG>(w1, unit_ptr(7), voff);
    unit(w2, 2, 3, first);
    w_issue<UT, RING_AG>(w2, unit_ptr(8), voff);
    unit(w3, 3, 4, first);
    w_issue<UT, RING_AG>(w3, unit_ptr(9), voff);
    PSTAMP(1);
    unit(wr[0], 4, 5, resident);
    unit(wr[1], 5, 6, resident);
    PSTAMP(2);
    unit(w0, 6, 7, steady);
    w_issue<UT, RING_AG>(w0, unit_ptr(10), voff);
    premath(0, t);
    __builtin_amdgcn_sched_barrier(0);
    unit(w1, 7, 8, steady);
    w_issue<UT, RING_AG>(w1, unit_ptr(11), voff);
    // ---- publish it
    const int xs = slot * (2 * 4 * XW);
    if (m != mute_member) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (same_l2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pvec4, acc[i]), xrsrc, xs + tx0 + i * 1024, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pvec4, acc[i]), xrsrc, xs + tx0 + i * 1024, 0, SC1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(3);
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The peer's contribution to this wave's outputs: asked for after unit PAY_AT, looked at when the pass is over.
    // The first read is inline asm with a counted wait, like the ring loads (left to the compiler the wait is
    // vmcnt(0): a full L2 round trip for the two ring units issued behind the read); a read that finds the slot
    // EMPTY is repeated the plain way.  Always issued, so the queue looks the same on every path.
    pvec4 pay[NI];
    auto load_pay = [&] {
#pragma unroll
      for (int i = 0; i < NI; ++i) pay[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xs + rx0 + i * 1024, 0, SC1);
    };
    auto ask_peer = [&](auto after) {
      if constexpr (decltype(after)::value == PAY_AT) {
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(4);
        const int vo = xs + rx0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3 sc1" : "=v"(pay[i]) : "v"(vo), "s"(xdesc), "n"(i * 1024));
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_assert(PAY_AT == 11, "the wait below counts the ring loads issued behind the read: units 2 and 3 of the next step");
    // ---- second pass: the own 128 outputs
    unit(w2, 8, 9, steady);
    w_issue<UT, RING_AG>(w2, unit_ptr(12), voff);
    ask_peer(std::integral_constant<int, 8>{});
    premath(1, t);
    __builtin_amdgcn_sched_barrier(0);
    unit(w3, 9, 10, steady);
    w_issue<UT, RING_AG>(w3, unit_ptr(13), voff);
    ask_peer(std::integral_constant<int, 9>{});
    unit(w0, 10, 11, steady);
    w_issue<UT, RING_AG>(w0, unit_ptr(0), voff);
    ask_peer(std::integral_constant<int, 10>{});
    premath(2, t);
    __builtin_amdgcn_sched_barrier(0);
    unit(w1, 11, 12, steady);
    w_issue<UT, RING_AG>(w1, unit_ptr(1), voff);
    ask_peer(std::integral_constant<int, 11>{});
    unit(w2, 12, 13, steady);
    w_issue<UT, RING_AG>(w2, unit_ptr(2), voff);
    ask_peer(std::integral_constant<int, 12>{});
    unit(w3, 13, 14, steady);
    w_issue<UT, RING_AG>(w3, unit_ptr(3), voff);
    ask_peer(std::integral_constant<int, 13>{});
    unit(wr[2], 14, 15, resident);
    unit(wr[3], 15, -1, resident);
    ask_peer(std::integral_constant<int, 15>{});
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(5);
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(pay[0]), "+v"(pay[1]), "+v"(pay[2]), "+v"(pay[3]) : "n"(2 * UT));
    if (!dead) {
      int spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NI; ++i)
          ok = ok && pay[i][0] != EMPTY && pay[i][1] != EMPTY && pay[i][2] != EMPTY && pay[i][3] != EMPTY;
        if (__all(ok)) break;
        if (++spins >= spin_limit) {
          if (lane == 0) atomicOr(err, 2u);
          dead = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        load_pay();
      }
#ifdef VTS_RP_STAMP
      pstamp[10] += spins;
#endif
    }
    PSTAMP(6);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const f32x4 p = dead ? f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")}
                           : __builtin_bit_cast(f32x4, pay[i]);
      dh[i] = acc[i] + p;
    }
    // The slot is EMPTY again for the step after next but one.  (The sums above come first, pinned by the asm: the
    // compiler waits for `pay` with counts that stores issued in between would turn into a wait for those stores -
    // a write-through round trip.  A wave that gave up resets too: it no longer matters.)
    asm volatile("" : "+v"(dh[0]), "+v"(dh[1]), "+v"(dh[2]), "+v"(dh[3]) :: "memory");
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (same_l2) __builtin_amdgcn_raw_buffer_store_b128(pvec4{EMPTY, EMPTY, EMPTY, EMPTY}, xrsrc, xs + rx0 + i * 1024, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(pvec4{EMPTY, EMPTY, EMPTY, EMPTY}, xrsrc, xs + rx0 + i * 1024, 0, SC1);
    }
    slot = slot == XSLOTS - 1 ? 0 : slot + 1;
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(11);
#pragma unroll
    for (int i = PRE; i < NI; ++i) premath(i, t);
    take_dh();
    PSTAMP(12);
    wg_barrier();                                 // every wave has read the last fragments of dA_{t+1}
    PSTAMP(8);
    cell_bwd(t, t >= 1 ? t - 1 : 0);
  }
  // keep the wrapped-around ring loads' registers live until they have landed
  w_wait<0, UT, RING_AG>(w0);
  w_wait<0, UT, RING_AG>(w1);
  w_wait<0, UT, RING_AG>(w2);
  w_wait<0, UT, RING_AG>(w3);

#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const long bt = bt0 + (i >> 1);
    const int jt = jt0 + (i & 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = dbs[i][k];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (dead) s = __builtin_nanf("");
#ifdef VTS_RP_STAMP
      s = 0.f;
#endif
      if (g == 0) dbpart[bt * (4 * PH) + k * PH + jt * 16 + r] = s;
    }
  }
#ifdef VTS_RP_STAMP
  __syncthreads();
  if (tid == 0)
    for (int i = 0; i < 16; ++i) dbpart[(bt0 + m) * (4 * PH) + 128 * m + i] = (float)pstamp[i] / (float)(T - 1);
#endif
#undef PSTAMP
}

std::atomic<int> g_pair_fault{-1};

}  // namespace

// Test hook: member `member` (0 or 1) of every pair withholds what it owes its peer and the reads give up after
// 1024 rounds, so that the timeout path can be exercised; negative (the default) = off.  Process-wide.
void vts_lstm_rec3p_set_fault(int member) { g_pair_fault.store(member < 0 ? -1 : (member & 1)); }

size_t vts_lstm_rec3p_xchg_bytes(int B) { return (size_t)(B / PROWS) * X_PAIR; }

// Shape test: hidden 256, whole pairs, and a grid that is resident at once at one workgroup per CU (the members
// of a pair wait for each other); whether it really is co-resident is the cooperative launch's business.
bool vts_lstm_rec3p_usable(int B, int T, int H) {
  if (H != PH || B < PROWS || (B % PROWS) != 0 || T < 2) return false;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  return 16 * vts_cdiv(B / PROWS, 8) <= cus;
}

// Same contract as vts_lstm_rec3c_bwd: VTS_OK, an error, or VTS_ERR_UNSUPPORTED when the runtime will not guarantee
// that the whole grid is resident at once.  The occupancy test refuses before anything is enqueued; a refusal by the
// cooperative launch itself comes after the packing kernel is in the queue, which has then overwritten Wp, the
// exchange buffer and the status word - harmless, because the caller's fall-back kernel packs Wp again and clears
// the status word itself.  Under stream capture a cooperative launch cannot be recorded: that, too, is reported as
// VTS_ERR_UNSUPPORTED before anything is enqueued.  xws: vts_lstm_rec3p_xchg_bytes(B) bytes.
int vts_lstm_rec3p_bwd(const float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                       float* dbpart, const float* W_hh, float* Wp, void* dg_planes, void* xws, unsigned* err, int B, int T,
                       hipStream_t stream) {
  const int npairs = B / PROWS;
  const void* kern = dhseq != nullptr ? reinterpret_cast<const void*>(lstm_rec3p_bwd_kernel<true>)
                                      : reinterpret_cast<const void*>(lstm_rec3p_bwd_kernel<false>);
  static vts_lds_mask done_dh{0}, done_nodh{0};
  if (int rc = vts_reserve_lds(kern, P_LDS, dhseq != nullptr ? &done_dh : &done_nodh, "lstm_rec3p_bwd")) return rc;
  const unsigned grid = 16 * vts_cdiv(npairs, 8);
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, P_LDS) != hipSuccess ||
      (long)per_cu * cus < (long)grid) {
    (void)hipGetLastError();
    vts_set_error("lstm_rec3p_bwd: %u workgroups cannot be resident at once (%d per CU x %d CUs)", grid, per_cu, cus);
    return VTS_ERR_UNSUPPORTED;
  }
  if (vts_stream_is_capturing(stream)) {
    vts_set_error("lstm_rec3p_bwd: not recorded into graphs (a replayed grid's residency cannot be checked here)");
    return VTS_ERR_UNSUPPORTED;
  }
  // plain launch by default; VTS_LSTM_PAIR_COOP=1 makes it a cooperative one (see vts_lstm_rec3c_fwd for what that buys)
  static const bool coop = [] { const char* e = getenv("VTS_LSTM_PAIR_COOP"); return e && e[0] == '1'; }();
  u32x4* img = reinterpret_cast<u32x4*>(Wp);
  // (the packing kernel also sets every word of the exchange buffer to EMPTY and clears the launch's status word)
  if (!vts_aligned16(xws)) {
    vts_set_error("lstm_rec3p_bwd: the exchange buffer must be 16-byte aligned");
    return VTS_ERR_ARG;
  }
  hipLaunchKernelGGL(pack_whh3p_kernel, dim3(vts_cdiv(PH * PH, 256)), dim3(256), 0, stream, W_hh, img, xws,
                     (long)(vts_lstm_rec3p_xchg_bytes(B) / 16), err);
  VTS_CHECK_LAUNCH("pack_whh3p");
  int mute_member = g_pair_fault.load();
  int spin_limit = mute_member >= 0 ? 1 << 10 : 1 << 20;
  int np = npairs;
  gp::SeqPlanes dgp{reinterpret_cast<char*>(dg_planes), 32, T, 0};
  unsigned char* xchg = reinterpret_cast<unsigned char*>(xws);
  if (!coop) {
    if (dhseq != nullptr)
      hipLaunchKernelGGL(lstm_rec3p_bwd_kernel<true>, dim3(grid), dim3(256), P_LDS, stream, G, cseq, dhseq, dhlast, dclast, dbpart,
                         img, dgp, xchg, err, T, np, spin_limit, mute_member);
    else
      hipLaunchKernelGGL(lstm_rec3p_bwd_kernel<false>, dim3(grid), dim3(256), P_LDS, stream, G, cseq, dhseq, dhlast, dclast, dbpart,
                         img, dgp, xchg, err, T, np, spin_limit, mute_member);
    VTS_CHECK_LAUNCH("lstm_rec3p_bwd");
    return VTS_OK;
  }
  void* args[] = {&G, &cseq, &dhseq, &dhlast, &dclast, &dbpart, &img, &dgp, &xchg, &err, &T, &np, &spin_limit, &mute_member};
  const hipError_t e = hipLaunchCooperativeKernel(kern, dim3(grid), dim3(256), args, P_LDS, stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    vts_set_error("lstm_rec3p_bwd: cooperative launch refused: %s", hipGetErrorString(e));
    return VTS_ERR_UNSUPPORTED;      // whatever the reason (too large, another process holds the device's cooperative queue): nothing was launched, the caller runs the kernel that waits for nobody
  }
  return VTS_OK;
}
