// Forward LSTM recurrence, bf16x3, with W_hh stationary: eight workgroups (eight CUs) share 128 windows.
//
// The one-workgroup-per-16-windows kernels (lstm_rec_bf16x3.hip) re-read the whole packed W_hh image from the
// XCD's L2 every time step, and that stream - not the matrix cores - sets their step time.  Here member c of a
// cluster owns hidden units [32c, 32c + 32) for all four gates: its slice of W_hh (128 gate columns x 256 k, hi
// and lo planes, 128 KiB) stays in LDS for the whole sequence, and what crosses the L2 each step is h instead:
// every member publishes its 128 x 32 slice of h_t (bf16 hi / lo, A-fragment order, 16 KiB) and reads all of
// h_t (128 KiB) for the next step.
//
// Hand-off: payload fragments are stored and loaded with sc1 buffer accesses (they go around the CU's L1, which
// would otherwise keep the lines of two steps ago: the two exchange buffers alternate with the step's parity);
// every storing wave drains its stores (vmcnt(0)) before the workgroup barrier behind which one lane raises the
// member's flag; one wave of each consumer polls the eight flags, the others wait at a barrier it joins.
// A member reuses a buffer only after it has seen every peer publish the step in between, so no peer can
// still be reading it.  Every poll is bounded: if a peer never shows up (the launcher only takes this kernel
// when the whole grid fits on the device at one workgroup per CU) the kernels leave with an error word set and
// NaN in the last step of the h sequence and in the final states, which turns the step's loss into NaN.
#include "lstm_rec_common.h"
#include "gemm_planes.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include <utility>

namespace {

constexpr int CH = 256;                 // hidden size this kernel is built for
constexpr int NC = 8;                   // members per cluster
constexpr int CROWS = 128;              // windows per cluster
constexpr int CT = 8;                   // column tiles per member: 4 gates x 2 hidden tiles
constexpr int KS = CH / 32;             // k-steps of the recurrent product
constexpr int W_LDS = KS * 2 * CT * 1024;            // resident slice: [ks][hi | lo][ct][64 lanes] x 16 B = 128 KiB
constexpr int S_PLANE = CROWS * 64;                  // staging plane: 128 rows x 32 bf16
constexpr int C_LDS = W_LDS + 2 * S_PLANE;           // 144 KiB
constexpr int X_PLANE = 8 * KS * 1024;               // exchange plane: [row tile 8][ks 8][64 lanes] x 16 B = 64 KiB
constexpr size_t X_CLUSTER = 2ul * 2 * X_PLANE;      // [parity][hi | lo] = 256 KiB per cluster
constexpr int FLAG_STRIDE = 32;                      // uint32 words between flags (one 128-byte line each)

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), unrolled by the front end (the optimiser's own
// full-unroll budget is smaller than the 64 MFMA groups of a step, and their register rotation needs static indices)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Bulk tile16 stores are plain ones here: no image in the L2 needs protecting from them (the one-workgroup kernels
// write them non-temporally for that), and plain stores measured 3 % faster per step.
__device__ __forceinline__ void st_bulk(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ bf16x8 frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ float lane_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// image for this kernel: [member c][ks][hi | lo][ct = gate*2 + jl][lane] ->
//   W_hh[gate*H + (2c + jl)*16 + (lane & 15)][32*ks + 8*(lane >> 4) + e],  e = 0..7
__global__ void pack_whh3c_kernel(const float* __restrict__ W, u32x4* __restrict__ out, void* handoff, long n16,
                                  unsigned* status) {
  reset_handoff(handoff, n16, 0u, status);
  const int total = CH * CH;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int ct = rest & 7; rest >>= 3;
    const int p = rest & 1; rest >>= 1;
    const int ks = rest & 7;
    const int c = rest >> 3;
    const int gate = ct >> 1, jl = ct & 1;
    const int row = gate * CH + (2 * c + jl) * 16 + (lane & 15);
    const int k0 = 32 * ks + 8 * (lane >> 4);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hi, lo;
      split2(W[row * CH + k0 + 2 * q], W[row * CH + k0 + 2 * q + 1], hi, lo);
      o[q] = p ? lo : hi;
    }
    out[idx] = o;
  }
}

typedef unsigned long long u64;
typedef unsigned bvec4 __attribute__((vector_size(16)));
// xchg: [cluster][parity][plane][rt][ks][lane] 16-byte fragments; flags: [cluster][member] words FLAG_STRIDE
// apart, zero before the launch, flag = number of steps published; err: one word, set on a poll timeout.
__global__ __launch_bounds__(256) void lstm_rec3c_fwd_kernel(
    float* __restrict__ G, float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hlast,
    float* __restrict__ clast, const u32x4* __restrict__ Wp, u64* __restrict__ xchg, unsigned* __restrict__ flags,
    unsigned* __restrict__ err, int T, int nclusters, int spin_limit, int mute_member, gp::SeqPlanes hp) {
  constexpr int JT = CH / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smemc[];
  __shared__ int dead;
  u32x4* wl = reinterpret_cast<u32x4*>(smemc);                 // [ks][plane][ct][lane]
  unsigned char* stage = smemc + W_LDS;                        // [plane][128 rows][64 B]
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

  // resident slice of the image
  {
    const u32x4* src = Wp + (size_t)c * (W_LDS / 16);
#pragma unroll 4
    for (int i = tid; i < W_LDS / 16; i += 256) wl[i] = src[i];
  }
  // the cluster's exchange buffers as a buffer resource: 16-byte accesses with the sc1 bit (aux 16), which go
  // around the CU's L1 like the agent-scope atomics the flags use, and which the compiler's wait counting sees
  // With plane output the plane buffer itself is the exchange medium: member c's h_t tiles (column block c of the
  // cluster's eight window tiles) are what its peers load as the fragments of k-step c, and every step has its own
  // slot, so h is written once and no separate exchange buffer is touched.
  const bool planes = hp.base != nullptr;
  const long hp_cluster_bytes = 8L * hp.ts * hp.ncb * 2 * gp::TILE_B;      // the cluster's eight window tiles
  const __amdgpu_buffer_rsrc_t xrsrc =
      planes ? __builtin_amdgcn_make_buffer_rsrc(hp.base + hp.tile((long)q * 8, -hp.t0, 0, 0), 0, (int)hp_cluster_bytes, 0x00020000)
             : __builtin_amdgcn_make_buffer_rsrc(xchg + (size_t)q * (X_CLUSTER / 8), 0, (int)X_CLUSTER, 0x00020000);
  // byte offset in xrsrc of the 1 KiB block (window tile rt of the cluster, k-step / column block ks, plane p) of h_t
  auto hp_block = [&](int rt, int t, int ks, int p) -> int {
    return (int)(((((long)rt * hp.ts + hp.t0 + t) * hp.ncb + ks) * 2 + p) * gp::TILE_B);
  };
  unsigned* fq = flags + (size_t)q * NC * FLAG_STRIDE;

  f32x4 cst[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int jl = 0; jl < 2; ++jl) cst[mt][jl] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Plane output of the h sequence (hp.base != null; gemm_planes.h): member c owns column block c (its 32 hidden
  // units) of every tile and writes it when it publishes h_t; the last step's h (never published) is copied out of
  // the staging planes at the end, and slot 0 (the zero initial state) is cleared here.
  const unsigned pl_off = gp::slot(g, r) * 16;
  auto hp_tile = [&](int mt, int t, int p) -> char* {
    return hp.base + hp.tile((long)q * 8 + 2 * wave + mt, t, c, p);
  };
  if (planes) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<u32x4*>(hp_tile(i >> 1, -1, i & 1) + lane * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // gate block (tile16) of batch tile bt, step t, column tile (gate, hidden tile jt)
  auto gblock = [&](int mt, int t, int gate, int jl) -> long {
    const long bt = (long)q * 8 + 2 * wave + mt;
    return ((bt * T + t) * (4 * JT) + gate * JT + 2 * c + jl) * 256 + lane * 4;
  };
  f32x4 gnext[2][CT];
  auto load_g = [&](int t) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) gnext[mt][ct] = ld_stream(G + gblock(mt, t, ct >> 1, ct & 1));
  };
  auto load_g_one = [&](int t, int i) {          // block i of the sixteen
    gnext[i >> 3][i & 7] = ld_stream(G + gblock(i >> 3, t, (i & 7) >> 1, i & 1));
  };
  load_g(0);
  __syncthreads();   // image slice and `dead` in place

  // bulk stores of step t (gates, cells, h sequence: HBM writes), from the registers the cell update left them in
  f32x4 act[2][CT];
  f32x4 hv[2][2];
  // one sixteenth of the same: sub = mt*8 + jl*4 + {0: h sequence rows, 1: gates i f, 2: gates g o, 3: cell block}
  auto store_part = [&](int t, int sub) {
    const int mt = sub >> 3, jl = (sub >> 2) & 1, what = sub & 3;
    const long bt = (long)q * 8 + 2 * wave + mt;
    const int jt = 2 * c + jl;
    if (what == 0) {
      if (hseq != nullptr) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) hseq[((bt * 16 + 4 * g + reg) * T + t) * CH + jt * 16 + r] = hv[mt][jl][reg];
      }
    } else if (what == 3) {
      st_bulk(cseq + ((bt * T + t) * JT + jt) * 256 + lane * 4, cst[mt][jl]);
    } else {
#pragma unroll
      for (int gate = 2 * (what - 1); gate < 2 * what; ++gate) st_bulk(G + gblock(mt, t, gate, jl), act[mt][gate * 2 + jl]);
    }
  };
  auto bulk_store = [&](int t) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const long bt = (long)q * 8 + 2 * wave + mt;
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const int jt = 2 * c + jl;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const long b = bt * 16 + 4 * g + reg;
          if (hseq != nullptr) hseq[(b * T + t) * CH + jt * 16 + r] = hv[mt][jl][reg];
          if (t == T - 1) {
            if (hlast != nullptr) hlast[b * CH + jt * 16 + r] = hv[mt][jl][reg];
            if (clast != nullptr) clast[b * CH + jt * 16 + r] = cst[mt][jl][reg];
          }
        }
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) st_bulk(G + gblock(mt, t, gate, jl), act[mt][gate * 2 + jl]);
        st_bulk(cseq + ((bt * T + t) * JT + jt) * 256 + lane * 4, cst[mt][jl]);
      }
    }
  };
#ifdef VTS_C_STAMP
  long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define STAMP(i) do { const long long now = clock64(); stamp[i] += now - tprev; tprev = now; } while (0)
#else
#define STAMP(i)
#endif
  for (int t = 0; t < T; ++t) {
    STAMP(6);
    // Order of a step, chosen so that only the hand-off itself sits on the critical path (vector memory
    // instructions complete in issue order, so whatever is issued before a load delays it):
    //   wait for h_{t-1} -> issue the fragment loads of its first two k-steps -> product, with the other
    //   fragment loads, the next step's gate prefetch and the bulk stores of step t-1 spread over it ->
    //   cell update -> publish h_t and raise the flag.
    u32x4 a[3][2][2];    // [ks % 3][mt][plane]: fragments of three k-steps in flight
    if (t > 0) {
#ifndef VTS_C_NOSYNC
      if (wave == 0) {
        int spins = 0;
        bool ok = false;
        do {
          const unsigned seen = lane < NC ? __hip_atomic_load(fq + lane * FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : 0xffffffffu;
          ok = __all(seen >= (unsigned)t);
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
        } while (++spins < spin_limit);
        if (!ok && lane == 0) {
          dead = 1;
          atomicOr(err, 1u);
        }
      }
      __syncthreads();
      if (dead) break;
#endif
      STAMP(0);
    }
    const int xin_off = ((t + 1) & 1) * (2 * X_PLANE);      // buffer of h_{t-1}
    auto load_a = [&](int ks) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int off = planes ? hp_block(2 * wave + mt, t - 1, ks, p) + (int)pl_off
                                 : xin_off + p * X_PLANE + ((2 * wave + mt) * KS + ks) * 1024 + lane * 16;
          a[ks % 3][mt][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 16));
        }
    };
    if (t > 0) {
      STAMP(0);
      load_a(0);
      load_a(1);
      STAMP(1);
    }
    f32x4 acc[2][CT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = gnext[mt][ct];
    STAMP(7);

    if (t == 0 && T > 1) load_g(1);
#ifndef VTS_C_NOPROD
    if (t > 0) {
      // 64 groups of six MFMAs, one per (k-step, column tile); the W fragments of a group are read from LDS two
      // groups ahead (an LDS read takes about as long as one group), in a three-slot register rotation
      u32x4 bh[3], bl[3];
      const int tnext = min(t + 1, T - 1);
      auto fetch_b = [&](int idx, int slot) {
        bh[slot] = wl[(((idx >> 3) * 2 + 0) * CT + (idx & 7)) * 64 + lane];
        bl[slot] = wl[(((idx >> 3) * 2 + 1) * CT + (idx & 7)) * 64 + lane];
      };
      fetch_b(0, 0);
      fetch_b(1, 1);
      static_for<KS * CT>([&](auto idx_c) {
        constexpr int idx = decltype(idx_c)::value;
        constexpr int ks = idx >> 3, ct = idx & 7, slot = idx % 3;
        if (idx + 2 < KS * CT) fetch_b(idx + 2, (idx + 2) % 3);
        if (ct == 0) {
          if (ks + 2 < KS) load_a(ks + 2);
        }
        // The CU's one vector-memory pipe moves 64 bytes a clock, and a wave stalls at a memory instruction while
        // that pipe's queue is full.  A step moves about 300 KiB through it, against 6144 clocks of matrix work:
        // so the memory instructions are spread over the product, one or two per group of MFMAs - the
        // fragments two k-steps ahead, two blocks of the next step's gates and two sixteenths of the previous
        // step's bulk stores (gates, cells, h sequence; they read act, hv and cst, which the cell update
        // further down overwrites) per k-step.
        // The next step's gate pre-activations (HBM): one block per group over the last two k-steps, behind the
        // last h fragment loads - vector memory loads complete in issue order, so earlier in the product they hold
        // up the fragments (L2) queued behind them.  (The last step re-reads its own blocks: no branch here.)
        if (ks >= KS - 2) load_g_one(tnext, (ks - (KS - 2)) * CT + ct);
#ifndef VTS_C_NOSTORE
        if (ct == 4) store_part(t - 1, 2 * ks);
        if (ct == 6) store_part(t - 1, 2 * ks + 1);
#endif
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks % 3][mt][1]), frag(bh[slot]), acc[mt][ct], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks % 3][mt][0]), frag(bl[slot]), acc[mt][ct], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks % 3][mt][0]), frag(bh[slot]), acc[mt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
#endif

    STAMP(2);
    // cell update, lane-local: acc[mt][gate*2 + jl][reg] is gate `gate` of window row 4g + reg of batch tile mt,
    // hidden unit (2c + jl)*16 + r.  The activations go into act, h into hv: the next step's product stores them.
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int jl = 0; jl < 2; ++jl)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float gi = fast_sigmoid(acc[mt][0 + jl][reg]);
          const float gf = fast_sigmoid(acc[mt][2 + jl][reg]);
          const float gg = fast_tanh(acc[mt][4 + jl][reg]);
          const float go = fast_sigmoid(acc[mt][6 + jl][reg]);
          act[mt][0 + jl][reg] = gi; act[mt][2 + jl][reg] = gf; act[mt][4 + jl][reg] = gg; act[mt][6 + jl][reg] = go;
          cst[mt][jl][reg] = gf * cst[mt][jl][reg] + gi * gg;
          const float h = go * fast_tanh(cst[mt][jl][reg]);
          hv[mt][jl][reg] = h;
          // h -> staging as bf16 hi / lo pairs: the even lane of a pair writes the two hi halves, the odd one the lo halves
          const float other = lane_xor1(h);
          const bool odd = r & 1;
          unsigned hi, lo;
          split2(odd ? other : h, odd ? h : other, hi, lo);
          const int row = (2 * wave + mt) * 16 + 4 * g + reg;
          const int col = jl * 16 + (r & ~1);
          *reinterpret_cast<unsigned*>(stage + (odd ? S_PLANE : 0) + row * 64 + col * 2) = odd ? lo : hi;
        }
    STAMP(3);
    if (t + 1 < T) {
      lds_barrier();                      // the staging tile is complete
      // publish this member's k-step (ks = c) of h_t: [plane][rt][lane] fragments, 1 KiB per wave instruction
      const int xout_off = (t & 1) * (2 * X_PLANE);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;    // [plane 2][rt 8][lane 64]
        const int p = idx >> 9, rt = (idx >> 6) & 7, l = idx & 63;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + p * S_PLANE + (rt * 16 + (l & 15)) * 64 + (l >> 4) * 16);
        const int off = planes ? hp_block(rt, t, c, p) + gp::slot(l >> 4, l & 15) * 16
                               : xout_off + p * X_PLANE + (rt * KS + c) * 1024 + l * 16;
        // (sc1 although the members of a cluster share an L2 when dispatch is round-robin: plain stores here
        // measured 10 % slower per launch and moved the same bytes)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bvec4, v), xrsrc, off, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // (mute_member >= 0: fault injection for the test of the timeout path - that member never raises its flag)
      if (tid == 0 && c != mute_member)
        __hip_atomic_store(fq + c * FLAG_STRIDE, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STAMP(4);
  }
  if (!dead) bulk_store(T - 1);
  if (!dead && planes) {
    // h_{T-1} from the staging planes: each wave reads back only rows it wrote itself (LDS operations of a wave
    // execute in order)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        *reinterpret_cast<u32x4*>(hp_tile(mt, T - 1, p) + pl_off) =
            *reinterpret_cast<const u32x4*>(stage + p * S_PLANE + ((2 * wave + mt) * 16 + r) * 64 + g * 16);
  }
  if (dead) {
    // a peer never published: besides the error word, poison what the layers above read (last step of the h
    // sequence, final states), so that the loss of such a step is NaN instead of a plausible number
    const float poison = __builtin_nanf("");
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int jl = 0; jl < 2; ++jl)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const long b = ((long)q * 8 + 2 * wave + mt) * 16 + 4 * g + reg;
          const int col = (2 * c + jl) * 16 + r;
          if (hseq != nullptr) hseq[(b * T + T - 1) * CH + col] = poison;
          if (hlast != nullptr) hlast[b * CH + col] = poison;
          if (clast != nullptr) clast[b * CH + col] = poison;
        }
  }
#ifdef VTS_C_STAMP
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) reinterpret_cast<long long*>(err + 2)[i] = stamp[i];
#endif
}


// ---- forward, one ring of waves per window tile -------------------------------------------------------------------
// Same clusters, same slices, same buffers as lstm_rec3c_fwd_kernel, other division of labour.  The recurrence of a
// window tile (16 windows) needs nothing of the cluster's other seven tiles, so the unit that takes part in a
// hand-off is a WAVE, not a workgroup: wave w of member c owns window tile w of the cluster for the member's 128 gate
// columns, publishes its own 1 KiB block of h_t (both planes) and raises a flag of its own; it waits for the eight
// waves w of the cluster's members and for nobody else.  A workgroup is eight such waves, two per SIMD, and has no
// barrier inside the time loop: while one wave of a SIMD waits for its ring (flag, fragments from the L2, write
// acknowledgements) the other one has the matrix core.  In the kernel above those waits - a third of its step - sit
// on the critical path of the one wave a SIMD runs.  The price: every wave reads the whole resident slice from LDS
// for one window tile instead of two (3 MFMAs per fragment pair instead of 6).
// flags: [cluster][window tile w] lines of FLAG_STRIDE words, word c = number of steps member c's wave w published.
#ifndef VTS_W_ST_KS
#define VTS_W_ST_KS 4      // k-steps over which the previous step's bulk stores are spread (2 or 4); the gate loads follow
#endif
__global__ __launch_bounds__(512) void lstm_rec3w_fwd_kernel(
    float* __restrict__ G, float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hlast,
    float* __restrict__ clast, const u32x4* __restrict__ Wp, u64* __restrict__ xchg, unsigned* __restrict__ flags,
    unsigned* __restrict__ err, int T, int nclusters, int spin_limit, int mute_member, gp::SeqPlanes hp) {
  constexpr int JT = CH / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smemc[];
  u32x4* wl = reinterpret_cast<u32x4*>(smemc);                 // [ks][plane][ct][lane]
  unsigned char* stage = smemc + W_LDS;                        // [plane][128 rows][64 B]; wave w owns rows 16w .. 16w + 15
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = window tile of the cluster
  const int r = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int c = j & 7;
  const int q = (j >> 3) * 8 + xcd;
  if (q >= nclusters) return;
  {
    const u32x4* src = Wp + (size_t)c * (W_LDS / 16);
#pragma unroll 4
    for (int i = tid; i < W_LDS / 16; i += 512) wl[i] = src[i];
  }
  const bool planes = hp.base != nullptr;
  const long hp_cluster_bytes = 8L * hp.ts * hp.ncb * 2 * gp::TILE_B;
  const __amdgpu_buffer_rsrc_t xrsrc =
      planes ? __builtin_amdgcn_make_buffer_rsrc(hp.base + hp.tile((long)q * 8, -hp.t0, 0, 0), 0, (int)hp_cluster_bytes, 0x00020000)
             : __builtin_amdgcn_make_buffer_rsrc(xchg + (size_t)q * (X_CLUSTER / 8), 0, (int)X_CLUSTER, 0x00020000);
  // byte offset, in xrsrc, of the 1 KiB block (this wave's window tile, column block / k-step ks, plane p) of h_t:
  // xbase(t) + ks * kstr + p * pstr in both media (plane buffer: every step its own slot; exchange buffer: two
  // buffers that alternate with the step's parity)
  const int kstr = planes ? 2 * gp::TILE_B : 1024, pstr = planes ? gp::TILE_B : X_PLANE;
  auto xbase = [&](int t) -> int {
    return planes ? ((wave * hp.ts + hp.t0 + t) * hp.ncb) * (2 * gp::TILE_B) : (t & 1) * (2 * X_PLANE) + wave * KS * 1024;
  };
  unsigned* fq = flags + ((size_t)q * NC + wave) * FLAG_STRIDE;
  const long bt = (long)q * 8 + wave;
  const unsigned pl_off = gp::slot(g, r) * 16;
  if (planes) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
      *reinterpret_cast<u32x4*>(hp.base + hp.tile(bt, -1, c, p) + lane * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  f32x4 cst[2];
  cst[0] = cst[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The tile16 blocks of this wave's window tile (gates: T x 64 blocks, cells: T x 16 blocks of 1 KiB) as buffer
  // resources: a block's address is then a scalar offset plus lane * 16, and no vector register holds an address.
  const __amdgpu_buffer_rsrc_t grsrc =
      __builtin_amdgcn_make_buffer_rsrc(G + bt * T * (4 * JT) * 256, 0, T * (4 * JT) * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t crsrc =
      __builtin_amdgcn_make_buffer_rsrc(cseq + bt * T * JT * 256, 0, T * JT * 1024, 0x00020000);
  const int lane16 = lane * 16;
  const int xvoff = planes ? (int)pl_off : lane16;        // a lane's 16 bytes inside a 1 KiB block of the exchange medium
  auto gblock = [&](int t, int gate, int jl) -> int { return (t * (4 * JT) + gate * JT + 2 * c + jl) * 1024; };
  // Two gate buffers that swap roles every step: `acc` enters a step holding the step's gate pre-activations
  // (fetched during the step before), receives the recurrent product and is turned into the gate activations in
  // place by the cell update; `nxt` enters holding the activations of the step before, which leave for memory during
  // the first half of the product, and is then refilled with the next step's pre-activations.
  f32x4 bufa[CT], bufb[CT];
  auto load_g_one = [&](f32x4 (&dst)[CT], int t, int ct) {      // streaming (nt) load, as ld_stream
    dst[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grsrc, lane16, gblock(t, ct >> 1, ct & 1), 2));
  };
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) load_g_one(bufa, 0, ct);
  __syncthreads();   // the image slice is in place; the only workgroup barrier of the kernel

  // an eighth of a step's bulk stores: sub = jl*4 + {1: gates i f, 2: gates g o, 3: cell block} (0: nothing - the
  // fp32 rows of h, where a caller wants them, leave straight from the cell update)
  auto store_part = [&](const f32x4 (&act)[CT], int t, int sub) {
    const int jl = sub >> 2, what = sub & 3;
    const int jt = 2 * c + jl;
    if (what == 3) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bvec4, cst[jl]), crsrc, lane16, (t * JT + jt) * 1024, 0);
    } else if (what != 0) {
#pragma unroll
      for (int gate = 2 * (what - 1); gate < 2 * what; ++gate)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bvec4, act[gate * 2 + jl]), grsrc, lane16, gblock(t, gate, jl), 0);
    }
  };
#ifdef VTS_C_STAMP
  long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#endif
  // one time step; false when the ring gave up waiting for a peer
  auto step = [&](int t, f32x4 (&acc)[CT], f32x4 (&nxt)[CT]) -> bool {
    STAMP(6);
    u32x4 a[KS][2];      // the fragments of h_{t-1} for this wave's window tile, all k-steps, both planes
    if (t > 0) {
#ifndef VTS_C_NOSYNC
      int spins = 0;
      bool ok = false;
      do {
        const unsigned seen = lane < NC ? __hip_atomic_load(fq + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
        ok = __all(seen >= (unsigned)t);
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
      } while (++spins < spin_limit);
      if (!ok) {
        if (lane == 0) atomicOr(err, 1u);
        return false;
      }
#endif
      // Nothing that reads the peers' blocks may be issued before the flags were seen: the loop above has consumed the
      // flag loads, instructions issue in order, and the fragment loads below are sc1 accesses that go to the L2 like
      // the flag loads did - so a compiler barrier is all that is needed.  (An agent-scope acquire fence would be a
      // buffer_inv sc1, which on this multi-XCD part also drops lines from the L2: measured 5 000 - 8 000 ticks per
      // step and wave, and every other access of the kernel slowed down with it.)
      asm volatile("" ::: "memory");
      STAMP(0);
      // Every fragment load of the step goes out now, in front of the step's other memory traffic: vector memory
      // operations complete in issue order, and a fragment queued behind a store's write acknowledgement or a gate
      // block from HBM would hold up the product for that long.
      const int xin = xbase(t - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          a[ks][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xvoff, xin + ks * kstr + p * pstr, 16));
      STAMP(1);
    }
    if (t == 0) {
      if (T > 1) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) load_g_one(nxt, 1, ct);
      }
    } else {
      // 32 groups of six MFMAs, one per (k-step, pair of column tiles): the two tiles alternate, so that an MFMA never
      // follows the one whose accumulator it adds to.  W fragments are read from LDS one group ahead (the SIMD's other wave covers what that leaves).
      u32x4 bh[2][2], bl[2][2];
      const int tnext = min(t + 1, T - 1);
      auto fetch_b = [&](int idx, int slot) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          bh[slot][e] = wl[(((idx >> 2) * 2 + 0) * CT + 2 * (idx & 3) + e) * 64 + lane];
          bl[slot][e] = wl[(((idx >> 2) * 2 + 1) * CT + 2 * (idx & 3) + e) * 64 + lane];
        }
      };
      fetch_b(0, 0);
#ifdef VTS_W_NOLDS
      fetch_b(1, 1);
#endif
      static_for<KS * CT / 2>([&](auto idx_c) {
        constexpr int idx = decltype(idx_c)::value;
        constexpr int ks = idx >> 2, cp = idx & 3, slot = idx & 1;
#ifndef VTS_W_NOLDS
        if (idx + 1 < KS * CT / 2) fetch_b(idx + 1, (idx + 1) & 1);
#endif
        // the previous step's bulk stores over the first four k-steps, then the next step's gate pre-activations
        // into the registers they left (the last step re-reads its own blocks: no branch here)
#ifndef VTS_C_NOSTORE
#ifndef VTS_W_NOST
        if (ks < VTS_W_ST_KS) { if (cp % (VTS_W_ST_KS / 2) == 0 || VTS_W_ST_KS == 2) store_part(nxt, t - 1, ks * (8 / VTS_W_ST_KS) + (VTS_W_ST_KS == 2 ? cp : cp / (VTS_W_ST_KS / 2))); }
#endif
#ifndef VTS_W_NOG
        if (ks == VTS_W_ST_KS || ks == VTS_W_ST_KS + 1) load_g_one(nxt, tnext, (ks - VTS_W_ST_KS) * 4 + cp);
#endif
#endif
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[2 * cp + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks][1]), frag(bh[slot][e]), acc[2 * cp + e], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[2 * cp + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks][0]), frag(bl[slot][e]), acc[2 * cp + e], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 2; ++e)
          acc[2 * cp + e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks][0]), frag(bh[slot][e]), acc[2 * cp + e], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    STAMP(2);
    // cell update, lane-local and in place: acc[gate*2 + jl][reg] is gate `gate` of window row 4g + reg, hidden unit
    // (2c + jl)*16 + r, and holds the gate's activation afterwards
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float gi = fast_sigmoid(acc[0 + jl][reg]);
        const float gf = fast_sigmoid(acc[2 + jl][reg]);
        const float gg = fast_tanh(acc[4 + jl][reg]);
        const float go = fast_sigmoid(acc[6 + jl][reg]);
        acc[0 + jl][reg] = gi; acc[2 + jl][reg] = gf; acc[4 + jl][reg] = gg; acc[6 + jl][reg] = go;
        cst[jl][reg] = gf * cst[jl][reg] + gi * gg;
        const float h = go * fast_tanh(cst[jl][reg]);
        {
          const long b = bt * 16 + 4 * g + reg;
          const int col = (2 * c + jl) * 16 + r;
          if (hseq != nullptr) hseq[(b * T + t) * CH + col] = h;
          if (t == T - 1) {
            if (hlast != nullptr) hlast[b * CH + col] = h;
            if (clast != nullptr) clast[b * CH + col] = cst[jl][reg];
          }
        }
        const float other = lane_xor1(h);
        const bool odd = r & 1;
        unsigned hi, lo;
        split2(odd ? other : h, odd ? h : other, hi, lo);
        const int row = wave * 16 + 4 * g + reg;
        const int col = jl * 16 + (r & ~1);
        *reinterpret_cast<unsigned*>(stage + (odd ? S_PLANE : 0) + row * 64 + col * 2) = odd ? lo : hi;
      }
    STAMP(3);
    if (t + 1 < T) {
      // publish this wave's block of h_t (window tile w, column block c): the rows it has just written itself - LDS
      // operations of a wave execute in order - as fragments, one KiB per plane
      const int xout = xbase(t) + c * kstr;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + p * S_PLANE + (wave * 16 + r) * 64 + g * 16);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bvec4, v), xrsrc, xvoff, xout + p * pstr, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the block has reached the L2 before the flag goes up
      if (lane == 0 && c != mute_member)
        __hip_atomic_store(fq + c, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    STAMP(4);
    return true;
  };
  bool dead = false;
  int t = 0;
  for (; t + 1 < T; t += 2) {
    if (!step(t, bufa, bufb) || !step(t + 1, bufb, bufa)) { dead = true; break; }
  }
  const bool odd_T = (T & 1) != 0;
  if (!dead && odd_T) dead = !step(T - 1, bufa, bufb);
#ifdef VTS_C_STAMP
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) reinterpret_cast<long long*>(err + 2)[i] = stamp[i];
#endif
  if (!dead) {
    // the last step's activations sit in the buffer that step accumulated in
#pragma unroll
    for (int sub = 0; sub < 8; ++sub) {
      if (odd_T) store_part(bufa, T - 1, sub);
      else store_part(bufb, T - 1, sub);
    }
    if (planes) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
        *reinterpret_cast<u32x4*>(hp.base + hp.tile(bt, T - 1, c, p) + pl_off) =
            *reinterpret_cast<const u32x4*>(stage + p * S_PLANE + (wave * 16 + r) * 64 + g * 16);
    }
  } else {
    // a peer never published: besides the error word, poison what the layers above read
    const float poison = __builtin_nanf("");
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const long b = bt * 16 + 4 * g + reg;
        const int col = (2 * c + jl) * 16 + r;
        if (hseq != nullptr) hseq[(b * T + T - 1) * CH + col] = poison;
        if (hlast != nullptr) hlast[b * CH + col] = poison;
        if (clast != nullptr) clast[b * CH + col] = poison;
      }
  }
}


// ---- backward ---------------------------------------------------------------------------------------------
// Reverse recurrence with W_hh stationary, same clusters: member c owns hidden units [32c, 32c + 32).  Per step it
// turns the running gradients of its slice into the gate gradients dA_t[:, 128 columns of its slice] (cell
// backward, lane-local), and needs dh_{t-1}[:, its slice] = dA_t[:, all 1024 columns] . W_hh[:, its slice]: the
// weight slice (1024 x 32, hi and lo planes, 128 KiB) stays in LDS, and what the members exchange is dA - through
// the very buffer the layer's products read afterwards: every member writes its tiles of the gate-gradient planes
// (gemm_planes.h; the output this kernel owes anyway) with sc1 stores and reads its peers' tiles of the same step
// back with sc1 loads once their flags are up.  No exchange buffer, no extra HBM bytes; every step has its own
// tiles, so nothing is ever reused while a peer may still read it.  Column order of the planes: k' = 4 * hidden +
// gate, as the one-workgroup kernel writes them.
constexpr int KSB = 4 * CH / 32;                       // k-steps of the backward product
constexpr int WB_LDS = KSB * 2 * 2 * 1024;             // resident slice: [ks][hi | lo][ct 2][64 lanes] x 16 B = 128 KiB
constexpr int SB_WAVE = 4 * 1024;                      // per-wave staging: 16 windows x 64 columns as 2 tiles x 2 planes
constexpr int CB_LDS = WB_LDS + 4 * SB_WAVE;           // 144 KiB

// image: [member c][ks][hi | lo][ct][lane] -> W_hh[gate_major(k')][(2c + ct)*16 + (lane & 15)], k' = 32 ks + 8 (lane >> 4) + e
__global__ void pack_whh3cb_kernel(const float* __restrict__ W, u32x4* __restrict__ out, void* handoff, long n16,
                                  unsigned* status) {
  reset_handoff(handoff, n16, 0u, status);
  const int total = CH * CH;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int ct = rest & 1; rest >>= 1;
    const int p = rest & 1; rest >>= 1;
    const int ks = rest & (KSB - 1);
    const int c = rest >> 5;
    const int col = (2 * c + ct) * 16 + (lane & 15);
    const int k0 = 32 * ks + 8 * (lane >> 4);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ka = k0 + 2 * q, kb = ka + 1;
      unsigned hi, lo;
      split2(W[((ka & 3) * CH + (ka >> 2)) * CH + col], W[((kb & 3) * CH + (kb >> 2)) * CH + col], hi, lo);
      o[q] = p ? lo : hi;
    }
    out[idx] = o;
  }
}

__global__ __launch_bounds__(256) void lstm_rec3c_bwd_kernel(
    const float* __restrict__ G, const float* __restrict__ cseq, const float* __restrict__ dhseq,
    const float* __restrict__ dhlast, const float* __restrict__ dclast, float* __restrict__ dbpart,
    const u32x4* __restrict__ Wp, gp::SeqPlanes dgp, unsigned* __restrict__ flags, unsigned* __restrict__ err, int T,
    int nclusters, int spin_limit, int mute_member) {
  constexpr int JT = CH / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smemc[];
  u32x4* wl = reinterpret_cast<u32x4*>(smemc);                 // [ks][plane][ct][lane]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* stage = smemc + WB_LDS + wave * SB_WAVE;      // this wave's own
  const int r = lane & 15, g = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int c = j & 7;
  const int q = (j >> 3) * 8 + xcd;
  if (q >= nclusters) return;
  {
    const u32x4* src = Wp + (size_t)c * (WB_LDS / 16);
#pragma unroll 4
    for (int i = tid; i < WB_LDS / 16; i += 256) wl[i] = src[i];
  }
  // the cluster's window tiles of the gate-gradient planes as one buffer: [rt 8][t][cb 32][plane 2] tiles of 1 KiB
  const long cl_bytes = 8L * dgp.ts * dgp.ncb * 2 * gp::TILE_B;
  const __amdgpu_buffer_rsrc_t prsrc =
      __builtin_amdgcn_make_buffer_rsrc(dgp.base + (size_t)q * cl_bytes, 0, (int)cl_bytes, 0x00020000);
  auto tile_off = [&](int rt, int t, int cb, int p) -> int {
    return (((rt * dgp.ts + t) * dgp.ncb + cb) * 2 + p) * gp::TILE_B;
  };
  unsigned* fw = flags + ((size_t)q * 4 + wave) * NC * FLAG_STRIDE;     // [cluster][wave][member]
  const unsigned pl_off = gp::slot(g, r) * 16;
  bool wdead = false;

  // running gradients of this lane's 2 x 2 blocks (window tile 2 wave + mt, hidden tile 2 c + jl)
  f32x4 dc[2][2], dh[2][2];
  float dbs[2][2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int jl = 0; jl < 2; ++jl) {
      const long blk = (((long)q * 8 + 2 * wave + mt) * JT + 2 * c + jl) * 256 + lane * 4;
      dh[mt][jl] = dhlast != nullptr ? *reinterpret_cast<const f32x4*>(dhlast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
      dc[mt][jl] = dclast != nullptr ? *reinterpret_cast<const f32x4*>(dclast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) dbs[mt][jl][k] = 0.f;
    }

  // saved tensors of a step, 24 blocks of 16 bytes per lane: part 0..15 gates (mt, gate, jl), 16..19 c_{t-1},
  // 20..23 the upstream gradient of the h sequence
  f32x4 gn[2][4][2], cprev[2][2], ccur[2][2], dn[2][2];
  auto load_part = [&](int t, int part) {
    if (part < 16) {
      const int mt = part >> 3, gate = (part >> 1) & 3, jl = part & 1;
      const long bt = (long)q * 8 + 2 * wave + mt;
      gn[mt][gate][jl] = ld_stream(G + ((bt * T + t) * (4 * JT) + gate * JT + 2 * c + jl) * 256 + lane * 4);
    } else {
      const int mt = (part >> 1) & 1, jl = part & 1;
      const long bt = (long)q * 8 + 2 * wave + mt;
      if (part < 20) {
        cprev[mt][jl] = t > 0 ? ld_stream(cseq + ((bt * T + t - 1) * JT + 2 * c + jl) * 256 + lane * 4)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        dn[mt][jl] = dhseq != nullptr ? ld_stream(dhseq + ((bt * T + t) * JT + 2 * c + jl) * 256 + lane * 4)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
      ccur[mt][jl] = ld_stream(cseq + ((((long)q * 8 + 2 * wave + mt) * T + T - 1) * JT + 2 * c + jl) * 256 + lane * 4);
  static_for<24>([&](auto pc) { load_part(T - 1, decltype(pc)::value); });
  __syncthreads();   // weight slice in place

#ifdef VTS_CB_STAMP
  long long bstamp[4] = {0, 0, 0, 0};
  long long bprev = clock64();
#define BSTAMP(i) do { const long long now = clock64(); bstamp[i] += now - bprev; bprev = now; } while (0)
#else
#define BSTAMP(i)
#endif
  for (int t = T - 1; t >= 0; --t) {
    BSTAMP(3);
    if (t < T - 1) {
      // ---- wait until wave `wave` of every member has published its rows of dA_{t+1} (flag = steps published).
      // The waves of a workgroup do not wait for each other: a wave's product reads only the window rows that the
      // same-numbered waves of the eight members wrote.
      {
        int spins = 0;
        bool ok = false;
        const unsigned want = (unsigned)(T - 1 - t);
        do {
          const unsigned seen = lane < NC ? __hip_atomic_load(fw + lane * FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : 0xffffffffu;
          ok = __all(seen >= want);
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
        } while (++spins < spin_limit);
        if (!ok) {
          if (lane == 0) atomicOr(err, 1u);
          wdead = true;
          break;
        }
      }
      BSTAMP(0);
      // ---- dh_t[:, slice] = dA_{t+1} . W_hh[:, slice]: A fragments from the planes (own tiles included), seven
      // k-steps ahead in an eight-slot rotation (28 KiB per wave in flight: the L2 round trip is ~1 us and a step
      // has 128 KiB per wave to pull); the weight fragments from LDS two groups ahead; the saved tensors of step t
      // ride along, one block per k-step
      u32x4 a[8][2][2];
      auto load_a = [&](int ks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            a[ks & 7][mt][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                prsrc, tile_off(2 * wave + mt, t + 1, ks, p) + pl_off, 0, 16));
      };
      static_for<7>([&](auto kc) { load_a(decltype(kc)::value); });
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 bh[3], bl[3];
      auto fetch_b = [&](int idx, int sl) {
        bh[sl] = wl[(((idx >> 1) * 2 + 0) * 2 + (idx & 1)) * 64 + lane];
        bl[sl] = wl[(((idx >> 1) * 2 + 1) * 2 + (idx & 1)) * 64 + lane];
      };
      fetch_b(0, 0);
      fetch_b(1, 1);
      static_for<KSB * 2>([&](auto idx_c) {
        constexpr int idx = decltype(idx_c)::value;
        constexpr int ks = idx >> 1, ct = idx & 1, sl = idx % 3;
        if (idx + 2 < KSB * 2) fetch_b(idx + 2, (idx + 2) % 3);
        if (ct == 0) {
          if (ks + 7 < KSB) load_a(ks + 7);
        } else {
          if (ks < 24) load_part(t, ks);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks & 7][mt][1]), frag(bh[sl]), acc[mt][ct], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks & 7][mt][0]), frag(bl[sl]), acc[mt][ct], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(a[ks & 7][mt][0]), frag(bh[sl]), acc[mt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) dh[mt][jl] = acc[mt][jl];
      BSTAMP(1);
    }

    // ---- cell backward of step t, one 16 x 16 block at a time: gate gradients -> this wave's staging tiles (the
    // plane tile format, four gates of a hidden unit = 8 bytes per plane) -> 16-byte fragments -> the planes
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int jl = 0; jl < 2; ++jl) {
        const f32x4 gi = gn[mt][0][jl], gf = gn[mt][1][jl], gg = gn[mt][2][jl], go = gn[mt][3][jl];
        const f32x4 ct_ = ccur[mt][jl], cp = cprev[mt][jl];
        const f32x4 dht = dh[mt][jl] + dn[mt][jl];
        f32x4 dai, daf, dag, dao;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float tc = fast_tanh(ct_[reg]);
          const float d_o = dht[reg] * tc;
          const float dcc = dc[mt][jl][reg] + dht[reg] * go[reg] * (1.f - tc * tc);
          const float d_i = dcc * gg[reg];
          const float d_g = dcc * gi[reg];
          const float d_f = dcc * cp[reg];
          dc[mt][jl][reg] = dcc * gf[reg];
          dai[reg] = d_i * gi[reg] * (1.f - gi[reg]);
          daf[reg] = d_f * gf[reg] * (1.f - gf[reg]);
          dag[reg] = d_g * (1.f - gg[reg] * gg[reg]);
          dao[reg] = d_o * go[reg] * (1.f - go[reg]);
          uint2 hi, lo;
          split4(dai[reg], daf[reg], dag[reg], dao[reg], hi, lo);
          // element (window row 4 g + reg, column 4 r + gate of the block's 64) -> tile r >> 3, k = 4 (r & 7) + gate
          const int row = 4 * g + reg;
          const int off = (r >> 3) * (2 * gp::TILE_B) + gp::slot((r & 7) >> 1, row) * 16 + (r & 1) * 8;
          *reinterpret_cast<uint2*>(stage + off) = hi;
          *reinterpret_cast<uint2*>(stage + gp::TILE_B + off) = lo;
        }
        dbs[mt][jl][0] += (dai[0] + dai[1]) + (dai[2] + dai[3]);
        dbs[mt][jl][1] += (daf[0] + daf[1]) + (daf[2] + daf[3]);
        dbs[mt][jl][2] += (dag[0] + dag[1]) + (dag[2] + dag[3]);
        dbs[mt][jl][3] += (dao[0] + dao[1]) + (dao[2] + dao[3]);
        // the wave reads back what it alone wrote (LDS operations of a wave execute in order)
#pragma unroll
        for (int i = 0; i < 4; ++i) {      // i = tile (r >> 3) * 2 + plane
          const u32x4 v = *reinterpret_cast<const u32x4*>(stage + i * gp::TILE_B + lane * 16);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bvec4, v), prsrc,
                                                 tile_off(2 * wave + mt, t, 4 * c + 2 * jl + (i >> 1), i & 1) + lane * 16, 0, 16);
        }
        ccur[mt][jl] = cp;
        __builtin_amdgcn_sched_barrier(0);
      }
    BSTAMP(2);
    if (t > 0) {
      // ---- publish: this wave's stores drained, then its flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0 && c != mute_member)
        __hip_atomic_store(fw + c * FLAG_STRIDE, (unsigned)(T - t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

#ifdef VTS_CB_STAMP
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 4; ++i) reinterpret_cast<long long*>(err + 2)[i] = bstamp[i];
#endif
  // bias-gradient partials: sum over the 4 row groups (lanes r, r+16, r+32, r+48); poisoned after a timeout
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s = dbs[mt][jl][k];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (wdead) s = __builtin_nanf("");
        if (g == 0) dbpart[((long)q * 8 + 2 * wave + mt) * (4 * CH) + k * CH + (2 * c + jl) * 16 + r] = s;
      }
}

}  // namespace

// The cluster kernel serves the bench-sized launches: hidden 256, whole clusters, and a grid that is resident
// at once at one workgroup per CU (its members wait for each other).  This is the cheap shape test; whether the
// grid really is co-resident is the launch's business (cooperative launch below).
bool vts_lstm_rec3c_usable(int B, int T, int H) {
  if (H != CH || B < CROWS || (B % CROWS) != 0 || T < 2) return false;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  const int nclusters = B / CROWS;
  return 64 * vts_cdiv(nclusters, 8) <= cus;
}

// Test hook: member `member` (0..7) of every cluster withholds its flags and the polls give up after 1024 rounds, so
// that the timeout path can be exercised; -1 (the default) = off.  Process-wide, not for production callers.
static std::atomic<int> g_fault_member{-1};
// Development switch: 1 = the forward cluster kernel with one ring of waves per window tile (VTS_LSTM_CLUSTER_RINGS).
static std::atomic<int> g_wave_rings{[] { const char* e = getenv("VTS_LSTM_CLUSTER_RINGS"); return e ? atoi(e) : 0; }()};
extern "C" void vts_debug_set_cluster_rings(int on) { g_wave_rings.store(on ? 1 : 0); }
extern "C" void vts_debug_set_cluster_fault(int member) { g_fault_member.store(member < 0 ? -1 : (member & (NC - 1))); }

// xws: exchange buffers (X_CLUSTER bytes per cluster) followed by the flags; fws: flag words; err: the launch's
// status word (cleared by the caller).
size_t vts_lstm_rec3c_xchg_bytes(int B) { return (size_t)(B / CROWS) * X_CLUSTER; }
size_t vts_lstm_rec3c_flag_bytes(int B) { return ((size_t)(B / CROWS) * NC + 1) * FLAG_STRIDE * sizeof(unsigned); }

// Returns VTS_OK, an error, or VTS_ERR_UNSUPPORTED when the whole grid cannot be resident at once (the occupancy query
// says so: a CU mask, too few CUs) or the stream is recording a graph: the caller then takes the one-workgroup
// kernel.  Both tests refuse before anything is enqueued.  The launch is a plain one by default.  A cooperative
// launch (VTS_LSTM_CLUSTER_COOP=1) adds nothing the occupancy query has not already established - the runtime's test
// is the same static one; neither form reserves the compute units against kernels of other queues, which is what the
// bounded polls and the status word are for - and costs 12 us of idle device in front of and behind the kernel (it
// travels through a queue of its own), is refused to every process but one per device, and makes rocprofv3 runs end
// in a segmentation fault inside the runtime's exit handler.  A refusal by the cooperative launch comes after the
// packing kernel is in the queue, which has then overwritten Wp, the flags and the status word - harmless, because
// the fall-back kernel packs Wp again and the caller clears the status word.
int vts_lstm_rec3c_fwd(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp,
                       void* xws, void* fws, unsigned* err, int B, int T, void* h_planes, hipStream_t stream) {
  const int nclusters = B / CROWS;
  // the kernel addresses a cluster's part of the plane buffer with 32-bit offsets
  if (h_planes != nullptr && 8L * (T + 1) * 8 * 2 * gp::TILE_B >= (1L << 31)) {
    vts_set_error("lstm_rec3c_fwd: %d steps are too many for the cluster kernel's plane addressing", T);
    return VTS_ERR_UNSUPPORTED;
  }
  static vts_lds_mask lds_done{0}, lds_done_w{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec3c_fwd_kernel), C_LDS, &lds_done, "lstm_rec3c_fwd"))
    return rc;
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec3w_fwd_kernel), C_LDS, &lds_done_w, "lstm_rec3w_fwd"))
    return rc;
  static const bool coop = [] { const char* e = getenv("VTS_LSTM_CLUSTER_COOP"); return e && e[0] == '1'; }();
  // which division of labour: a ring of waves per window tile (lstm_rec3w_fwd_kernel) or the workgroup-wide hand-off
  const bool rings = g_wave_rings.load() != 0;
  const void* kern = rings ? reinterpret_cast<const void*>(lstm_rec3w_fwd_kernel) : reinterpret_cast<const void*>(lstm_rec3c_fwd_kernel);
  const unsigned threads = rings ? 512 : 256;
  const unsigned grid = 64 * vts_cdiv(nclusters, 8);
  {
    if (vts_stream_is_capturing(stream)) {
      vts_set_error("lstm_rec3c_fwd: not recorded into graphs (a replayed grid's residency cannot be checked here)");
      return VTS_ERR_UNSUPPORTED;
    }
    // the whole grid must fit the device at once (what a cooperative launch would be refused for), asked beforehand
    // so that nothing has been enqueued on a refusal
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, (int)threads, C_LDS) != hipSuccess ||
        (long)per_cu * cus < (long)grid) {
      (void)hipGetLastError();
      vts_set_error("lstm_rec3c_fwd: %u workgroups cannot be resident at once (%d per CU x %d CUs)", grid, per_cu, cus);
      return VTS_ERR_UNSUPPORTED;
    }
  }
  u32x4* img = reinterpret_cast<u32x4*>(Wp);
  // (the packing kernel also zeroes the hand-off flags and the launch's status word)
  unsigned* flags = reinterpret_cast<unsigned*>(fws);
  if (!vts_aligned16(fws)) {
    vts_set_error("lstm_rec3c_fwd: the flag words must be 16-byte aligned");
    return VTS_ERR_ARG;
  }
  hipLaunchKernelGGL(pack_whh3c_kernel, dim3(vts_cdiv(CH * CH, 256)), dim3(256), 0, stream, W_hh, img, fws,
                     (long)(vts_lstm_rec3c_flag_bytes(B) / 16), err);
  VTS_CHECK_LAUNCH("pack_whh3c");
  // polls give up after about two seconds (1024 rounds under the test hook above)
  int mute_member = g_fault_member.load();
  int spin_limit = mute_member >= 0 ? 1 << 10 : 1 << 21;
  unsigned long long* xchg = reinterpret_cast<unsigned long long*>(xws);
  int nc = nclusters;
  gp::SeqPlanes hp{reinterpret_cast<char*>(h_planes), 8, T + 1, 1};
  if (coop) {
    void* args[] = {&G, &cseq, &hseq, &hlast, &clast, &img, &xchg, &flags, &err, &T, &nc, &spin_limit, &mute_member, &hp};
    const hipError_t e = hipLaunchCooperativeKernel(kern, dim3(grid), dim3(threads), args, C_LDS, stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      vts_set_error("lstm_rec3c_fwd: cooperative launch refused: %s", hipGetErrorString(e));
      return VTS_ERR_UNSUPPORTED;      // whatever the reason (too large, another process holds the device's cooperative queue): nothing was launched, the caller runs the kernel that waits for nobody
    }
    return VTS_OK;
  }
  if (rings)
    hipLaunchKernelGGL(lstm_rec3w_fwd_kernel, dim3(grid), dim3(512), C_LDS, stream, G, cseq, hseq, hlast, clast, img, xchg,
                       flags, err, T, nclusters, spin_limit, mute_member, hp);
  else
    hipLaunchKernelGGL(lstm_rec3c_fwd_kernel, dim3(grid), dim3(256), C_LDS, stream, G, cseq, hseq, hlast, clast, img, xchg,
                       flags, err, T, nclusters, spin_limit, mute_member, hp);
  VTS_CHECK_LAUNCH("lstm_rec3c_fwd");
  return VTS_OK;
}

size_t vts_lstm_rec3c_bwd_flag_bytes(int B) { return (size_t)(B / CROWS) * 4 * NC * FLAG_STRIDE * sizeof(unsigned); }

// Backward counterpart: same contract as vts_lstm_rec3c_fwd (VTS_ERR_UNSUPPORTED - see there for what may have been enqueued - when the
// grid cannot be guaranteed resident).  dhlast / dclast are tile16 [B, 256] blocks or null; dg_planes is the
// gate-gradient plane buffer (T step slots per window tile, 32 column blocks); fws holds the flag words.
int vts_lstm_rec3c_bwd(const float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                       float* dbpart, const float* W_hh, float* Wp, void* dg_planes, void* fws, unsigned* err, int B, int T,
                       hipStream_t stream) {
  const int nclusters = B / CROWS;
  static vts_lds_mask lds_done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec3c_bwd_kernel), CB_LDS, &lds_done, "lstm_rec3c_bwd"))
    return rc;
  const unsigned grid = 64 * vts_cdiv(nclusters, 8);
  if (vts_stream_is_capturing(stream)) {
    vts_set_error("lstm_rec3c_bwd: cooperative launches cannot be recorded into a graph");
    return VTS_ERR_UNSUPPORTED;
  }
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(lstm_rec3c_bwd_kernel), 256,
                                                   CB_LDS) != hipSuccess ||
      (long)per_cu * cus < (long)grid) {
    (void)hipGetLastError();
    vts_set_error("lstm_rec3c_bwd: %u workgroups cannot be resident at once (%d per CU x %d CUs)", grid, per_cu, cus);
    return VTS_ERR_UNSUPPORTED;
  }
  u32x4* img = reinterpret_cast<u32x4*>(Wp);
  unsigned* flags = reinterpret_cast<unsigned*>(fws);
  if (!vts_aligned16(fws)) {
    vts_set_error("lstm_rec3c_bwd: the flag words must be 16-byte aligned");
    return VTS_ERR_ARG;
  }
  hipLaunchKernelGGL(pack_whh3cb_kernel, dim3(vts_cdiv(CH * CH, 256)), dim3(256), 0, stream, W_hh, img, fws,
                     (long)(vts_lstm_rec3c_bwd_flag_bytes(B) / 16), err);
  VTS_CHECK_LAUNCH("pack_whh3cb");
  int mute_member = g_fault_member.load();
  int spin_limit = mute_member >= 0 ? 1 << 10 : 1 << 21;
  int nc = nclusters;
  gp::SeqPlanes dgp{reinterpret_cast<char*>(dg_planes), 32, T, 0};
  static const bool coop = [] { const char* e = getenv("VTS_LSTM_CLUSTER_COOP"); return e && e[0] == '1'; }();
  if (coop) {
    void* args[] = {&G, &cseq, &dhseq, &dhlast, &dclast, &dbpart, &img, &dgp, &flags, &err, &T, &nc, &spin_limit, &mute_member};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(lstm_rec3c_bwd_kernel), dim3(grid),
                                                    dim3(256), args, CB_LDS, stream);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      vts_set_error("lstm_rec3c_bwd: cooperative launch refused: %s", hipGetErrorString(e));
      return VTS_ERR_UNSUPPORTED;      // whatever the reason (too large, another process holds the device's cooperative queue): nothing was launched, the caller runs the kernel that waits for nobody
    }
    return VTS_OK;
  }
  hipLaunchKernelGGL(lstm_rec3c_bwd_kernel, dim3(grid), dim3(256), CB_LDS, stream, G, cseq, dhseq, dhlast, dclast, dbpart,
                     img, dgp, flags, err, T, nclusters, spin_limit, mute_member);
  VTS_CHECK_LAUNCH("lstm_rec3c_bwd");
  return VTS_OK;
}
