// Persistent LSTM recurrence kernels, bf16x3 on v_mfma_f32_16x16x32_bf16.
//
// Same ownership as the fp32 kernels (lstm_rec_f32.hip): one workgroup = 16 windows for the whole
// sequence, wave w = hidden units [w*JPW*16, (w+1)*JPW*16) for all four gates, tile16 buffers in
// MFMA accumulator order.  The recurrent product is evaluated as
//     h_hi.W_hi + h_lo.W_hi + h_hi.W_lo        (x = x_hi + x_lo, both bf16; fp32 accumulate)
// which costs 3/16 of the fp32 matrix-core time and keeps fp32-level accuracy (the dropped
// lo.lo term is ~2^-16 relative).  W_hh is split once per call by the pack kernels into a
// fragment-ordered bf16 image of the same size as the fp32 one (hi + lo = 4 bytes per weight); h
// (forward) and dA (backward) are split when they are written to LDS, where they live as two bf16
// planes.  With the matrix-core time out of the way the kernels run at the rate the XCD's L2
// delivers the W_hh image to 32 CUs at once (~70 GB/s per CU), so part of the image stays in LDS.
#include "lstm_rec_common.h"
#include "gemm_planes.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

// hidden 256: the backward kernel that stages the saved gates through LDS (lstm_rec3_bwd_staged_kernel) unless
// VTS_LSTM_STAGED_BWD=0 is in the environment (read once) or vts_lstm_rec3_set_staged_bwd(0) was called
std::atomic<int> g_staged_bwd{[] { const char* e = getenv("VTS_LSTM_STAGED_BWD"); return e == nullptr || e[0] != '0' ? 1 : 0; }()};

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// value of the neighbouring lane (lane ^ 1) without touching LDS: DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <int H>
struct Rec3Cfg {
  using C = RecCfg<H>;
  static constexpr int NT = C::JPW * 4;             // forward: accumulator tiles per wave
  static constexpr int KS_F = H / 32;               // forward k-steps (32 k each)
  static constexpr int NU_F = 2 * KS_F;             // forward ring units per step: (k-step, hi | lo)
  static constexpr int KS_B = (4 * H) / 32;         // backward k-steps
  static constexpr int NU_B = KS_B / 2;             // backward ring units per step: 2 k-steps x (hi, lo)
  static constexpr int UNIT = NT * 64;              // 16-byte fragments per unit and wave (both directions)
  static constexpr int CH_F = H / 8;                // 16-byte chunks per row of an h plane
  static constexpr int CH_B = (4 * H) / 8;          // ... of a dA plane
  static constexpr int SWZ_F = (CH_F < 16 ? CH_F : 16) - 1;
  // h planes + one half-unit (NT/2 fragments per lane) for every second k-step
  static constexpr size_t LDS_F = 2ul * 2 * 16 * H * 2 + (size_t)(KS_F / 2) * (NT / 2) * C::NW * 64 * 16;
  static constexpr size_t LDS_B = 2ul * 2 * 16 * 4 * H * 2;
};

// ---- weight packing ---------------------------------------------------------------------------
// One thread per 16-byte fragment element (8 consecutive k of one column, hi or lo plane).
// forward image : [wave][u = 2*ks + p][i = jl*4 + gate][lane]  ->
//                 W_hh[gate*H + (wave*JPW+jl)*16 + (lane&15)][32*ks + 8*(lane>>4) + e],  e = 0..7
// backward image: [wave][u][i = (kl*2 + p)*JPW + jl][lane], ks = 2*u + kl  ->
//                 W_hh[(e&3)*H + 8*ks + 2*(lane>>4) + (e>>2)][(wave*JPW+jl)*16 + (lane&15)]
//                 (k slot 4*j + gate, the order the backward kernel writes dA in)
template <int H, bool BWD>
__global__ void pack_whh3_kernel(const float* __restrict__ W, u32x4* __restrict__ out) {
  using C = RecCfg<H>;
  using R = Rec3Cfg<H>;
  const int total = H * H;   // 4H*H weights * 2 planes / 8 per fragment
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63;
    int rest = idx >> 6;
    const int i = rest % R::NT; rest /= R::NT;
    const int NU = BWD ? R::NU_B : R::NU_F;
    const int u = rest % NU;
    const int wave = rest / NU;
    float v[8];
    int p;
    if (!BWD) {
      const int ks = u >> 1;
      p = u & 1;
      const int jl = i >> 2, gate = i & 3;
      const int row = gate * H + (wave * C::JPW + jl) * 16 + (lane & 15);
      const int k0 = 32 * ks + 8 * (lane >> 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = W[row * H + k0 + e];
    } else {
      const int jl = i % C::JPW;
      const int kp = i / C::JPW;
      p = kp & 1;
      const int ks = 2 * u + (kp >> 1);
      const int col = (wave * C::JPW + jl) * 16 + (lane & 15);
      const int j0 = 8 * ks + 2 * (lane >> 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = W[((e & 3) * H + j0 + (e >> 2)) * H + col];
    }
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hi, lo;
      split2(v[2 * q], v[2 * q + 1], hi, lo);
      o[q] = p ? lo : hi;
    }
    out[idx] = o;
  }
}

// ---- forward ------------------------------------------------------------------------------------
// W_hh stream: the image is consumed in half-units of UT = NT/2 fragments (8 KiB per wave at
// H = 256): per k-step  hi[0..UT), hi[UT..NT), lo[0..UT), lo[UT..NT), through the four-buffer
// register ring of lstm_rec_common.h (w_issue / w_wait), three half-units in flight.
// Arguments as lstm_rec_fwd_kernel.  LDS: h as bf16 planes [buffer 2][hi, lo][16][H], the 16-byte
// chunk index XOR-ed with the row (conflict-free fragment reads without padding), then the
// second lo half-unit of every odd k-step (a quarter of the lo plane, 1/8 of the image).
// PL: the h sequence also goes out as bf16 planes (gemm_planes.h; hp: T + 1 step slots per window tile, slot 0 =
// the zero initial state): the A fragments a step's product reads from LDS are exactly the tiles of h_{t-1}, so
// wave w stores the fragments of k-steps w and w + 4 as it uses them; hseq may then be null.
template <int H, bool PL>
__global__ __launch_bounds__(RecCfg<H>::NW * 64) void lstm_rec3_fwd_kernel(
    float* __restrict__ G, float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hlast,
    float* __restrict__ clast, const u32x4* __restrict__ Wp, int B, int T, gp::SeqPlanes hp) {
  using C = RecCfg<H>;
  using R = Rec3Cfg<H>;
  constexpr int NT = R::NT;
  constexpr int UT = NT / 2;
  constexpr int KS = R::KS_F;
  static_assert(KS % 2 == 0, "k-steps are processed in pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  constexpr int PLANE = 16 * H * 2;                            // bytes of one bf16 plane
  u32x4* wlds = reinterpret_cast<u32x4*>(smem3 + 4 * PLANE);   // [KS/2][UT][NW*64]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int bt = blockIdx.x;
  const int b0 = bt * 16;

  f32x4 c[C::JPW];
#pragma unroll
  for (int jl = 0; jl < C::JPW; ++jl) c[jl] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's part of the image: [ks][hi | lo][NT][64 lanes] fragments of 16 bytes
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(Wp) + (size_t)wave * (2 * KS * NT * 1024);
  const unsigned voff = lane * 16;
  // half-unit q of k-step ks: q = 0, 1 hi halves, 2, 3 lo halves
  auto hu_ptr = [&](int ks, int q) { return wbase + (size_t)((ks * 4 + q) * UT) * 1024; };
  // Hidden 64: a wave's whole share of the image is 16 fragments (64 registers).  It is loaded once and stays; the
  // ring below would wait for the L2 four times a k-step with six MFMAs in between (22 us for 16 steps; 1.4 us a step).
  constexpr bool RESW = H == 64;
  static_assert(!(RESW && PL), "the plane stores ride in the streamed k-steps");
  u32x4 wr[RESW ? 4 * KS : 1][UT];
  if constexpr (RESW) {
#pragma unroll
    for (int f = 0; f < 4 * KS; ++f)
#pragma unroll
      for (int i = 0; i < UT; ++i) wr[f][i] = *reinterpret_cast<const u32x4*>(wbase + (size_t)(f * UT + i) * 1024 + voff);
  }
  // every lane re-reads from LDS only what it wrote itself, so no barrier is needed
#pragma unroll 1
  for (int n = 0; n < (RESW ? 0 : KS / 2); ++n) {
    const u32x4* src = reinterpret_cast<const u32x4*>(hu_ptr(2 * n + 1, 3)) + lane;
#pragma unroll
    for (int i = 0; i < UT; ++i) wlds[(n * UT + i) * (C::NW * 64) + tid] = src[i * 64];
  }
  auto load_w_lds = [&](u32x4 (&w)[UT], int slot) {
#pragma unroll
    for (int i = 0; i < UT; ++i) w[i] = wlds[(slot * UT + i) * (C::NW * 64) + tid];
  };
  auto load_g = [&](f32x4 (&gv)[NT], int t) {
    const float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
#pragma unroll
    for (int jl = 0; jl < C::JPW; ++jl)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        gv[jl * 4 + q] = ld_stream(Gt + (long)(q * C::JT + wave * C::JPW + jl) * 256);
  };

  float* hseq_wg = hseq + (long)b0 * T * H;   // uniform base; lane offsets stay below 16*T*H
  const unsigned pl_off = gp::slot(g, r) * 16;
  if (PL) {
#pragma unroll
    for (int i = 0; i < 2 * KS / C::NW; ++i)   // zero slot: column blocks wave, wave + NW, ..., both planes
      __builtin_nontemporal_store(u32x4{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4*>(
          hp.base + hp.tile(bt, -1, wave + C::NW * (i >> 1), i & 1) + lane * 16));
  }
  f32x4 gnext[NT];
  load_g(gnext, 0);
  u32x4 w0[UT], w1[UT], w2[UT], w3[UT];
#pragma unroll
  for (int i = 0; i < UT; ++i) w0[i] = w1[i] = w2[i] = w3[i] = u32x4{0u, 0u, 0u, 0u};
  if constexpr (!RESW) {
    w_issue<UT>(w0, hu_ptr(0, 0), voff);
    w_issue<UT>(w1, hu_ptr(0, 1), voff);
    w_issue<UT>(w2, hu_ptr(0, 2), voff);
  }

  // byte offset of this lane's A fragment (row r, chunk 4*ks + g) inside a plane
  auto a_off = [&](int ks) { return r * (H * 2) + (((4 * ks + g) ^ (r & R::SWZ_F)) << 4); };

  for (int t = 0; t < T; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    const unsigned char* hcur = smem3 + cur * (2 * PLANE);
    unsigned char* hnxt = smem3 + nxt * (2 * PLANE);
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = gnext[i];
    if (t + 1 < T) load_g(gnext, t + 1);

    if (t > 0) {
      auto lds_a = [&](int plane, int ks) {
        return *reinterpret_cast<const u32x4*>(hcur + plane * PLANE + a_off(ks));
      };
      u32x4 ah = lds_a(0, 0), al = lds_a(1, 0);
      // MFMAs of one half-unit: tiles [half*UT, half*UT + UT); sched_barrier(0) pins the issue order
      auto mma_hi = [&](const u32x4 (&w)[UT], int half) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < UT; ++i)
          acc[half * UT + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), as_bf16x8(w[i]),
                                                                      acc[half * UT + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < UT; ++i)
          acc[half * UT + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(w[i]),
                                                                      acc[half * UT + i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto mma_lo = [&](const u32x4 (&w)[UT], int half) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < UT; ++i)
          acc[half * UT + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(w[i]),
                                                                      acc[half * UT + i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      // One k-step.  RES: its last half-unit comes from LDS instead of the stream.  Younger loads
      // at each wait: the two (RES) or three half-units issued after the one being waited for.
      char* hp_t = PL ? hp.base + hp.tile(bt, t - 1, 0, 0) : nullptr;     // tiles of h_{t-1}
      auto kstep = [&](int ks, auto res_tag) {
        constexpr bool RES = decltype(res_tag)::value;
        constexpr int NY = (RES ? 2 : 3) * UT;
        if (PL && (ks & (C::NW - 1)) == wave) {
          __builtin_nontemporal_store(ah, reinterpret_cast<u32x4*>(hp_t + ks * (2 * gp::TILE_B) + pl_off));
          __builtin_nontemporal_store(al, reinterpret_cast<u32x4*>(hp_t + ks * (2 * gp::TILE_B) + gp::TILE_B + pl_off));
        }
        if constexpr (RES) load_w_lds(w3, ks >> 1);
        else w_issue<UT>(w3, hu_ptr(ks, 3), voff);
        const int kn = (ks + 1) & (KS - 1);
        const u32x4 nh = lds_a(0, kn), nl = lds_a(1, kn);   // A fragments of the next k-step
        w_wait<NY, UT>(w0);
        mma_hi(w0, 0);
        w_issue<UT>(w0, hu_ptr(kn, 0), voff);
        w_wait<NY, UT>(w1);
        mma_hi(w1, 1);
        w_issue<UT>(w1, hu_ptr(kn, 1), voff);
        w_wait<NY, UT>(w2);
        mma_lo(w2, 0);
        w_issue<UT>(w2, hu_ptr(kn, 2), voff);
        if constexpr (!RES) w_wait<3 * UT, UT>(w3);
        mma_lo(w3, 1);
        ah = nh;
        al = nl;
      };
      if constexpr (RESW) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int kn = (ks + 1) & (KS - 1);
          const u32x4 nh = lds_a(0, kn), nl = lds_a(1, kn);
          mma_hi(wr[ks * 4 + 0], 0);
          mma_hi(wr[ks * 4 + 1], 1);
          mma_lo(wr[ks * 4 + 2], 0);
          mma_lo(wr[ks * 4 + 3], 1);
          ah = nh;
          al = nl;
        }
      } else {
#pragma unroll 1
        for (int ks = 0; ks < KS; ks += 2) {
          kstep(ks, std::false_type{});
          kstep(ks + 1, std::true_type{});
        }
      }
    }

    // cell update, lane-local: acc[jl*4+q][reg] is gate q of window row 4g+reg, hidden jt*16+r
    float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
#pragma unroll
    for (int jl = 0; jl < C::JPW; ++jl) {
      const int jt = wave * C::JPW + jl;
      f32x4 gi, gf, gg, go;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        gi[reg] = fast_sigmoid(acc[jl * 4 + 0][reg]);
        gf[reg] = fast_sigmoid(acc[jl * 4 + 1][reg]);
        gg[reg] = fast_tanh(acc[jl * 4 + 2][reg]);
        go[reg] = fast_sigmoid(acc[jl * 4 + 3][reg]);
        c[jl][reg] = gf[reg] * c[jl][reg] + gi[reg] * gg[reg];
        const float hv = go[reg] * fast_tanh(c[jl][reg]);
        // h -> LDS as bf16 hi / lo pairs: the even lane of each pair writes the two hi halves,
        // the odd lane the two lo halves (one 4-byte store per lane)
        const float other = lane_xor1(hv);
        const bool odd = r & 1;
        unsigned hi, lo;
        split2(odd ? other : hv, odd ? hv : other, hi, lo);
        const int row = 4 * g + reg;
        const int col = jt * 16 + (r & ~1);
        const int off = row * (H * 2) + ((((col >> 3) ^ (row & R::SWZ_F)) << 4) | ((col & 7) << 1));
        *reinterpret_cast<unsigned*>(hnxt + (odd ? PLANE : 0) + off) = odd ? lo : hi;
        if (b0 + row < B) {
          if (hseq != nullptr) hseq_wg[(unsigned)((row * T + t) * H + jt * 16 + r)] = hv;
          if (t == T - 1) {
            if (hlast != nullptr) hlast[(long)(b0 + row) * H + jt * 16 + r] = hv;
            if (clast != nullptr) clast[(long)(b0 + row) * H + jt * 16 + r] = c[jl][reg];
          }
        }
      }
      st_stream(Gt + (long)(0 * C::JT + jt) * 256, gi);
      st_stream(Gt + (long)(1 * C::JT + jt) * 256, gf);
      st_stream(Gt + (long)(2 * C::JT + jt) * 256, gg);
      st_stream(Gt + (long)(3 * C::JT + jt) * 256, go);
      st_stream(cseq + ((long)(bt * T + t) * C::JT + jt) * 256 + lane * 4, c[jl]);
    }
    lds_barrier();
  }
  // The ring wrapped into a step that does not exist.  Waiting on the buffers here keeps them live
  // to the end: were they dead after their last MFMA, the allocator would reuse registers that
  // these loads still write to.
  if constexpr (!RESW) {
    w_wait<0, UT>(w0);
    w_wait<0, UT>(w1);
    w_wait<0, UT>(w2);
  }
  if (PL) {   // h_{T-1}: no product reads it, copy its tiles out of LDS (the barrier that ended the last step covers it)
    const unsigned char* hl = smem3 + (T & 1) * (2 * PLANE);
    char* hp_t = hp.base + hp.tile(bt, T - 1, 0, 0);
    for (int ks = wave; ks < KS; ks += C::NW)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(hl + p * PLANE + a_off(ks)),
                                    reinterpret_cast<u32x4*>(hp_t + (ks * 2 + p) * gp::TILE_B + pl_off));
  }
}

// ---- backward -----------------------------------------------------------------------------------
// Arguments as lstm_rec_bwd_kernel, except that dhlast / dclast are tile16 [Bp, H] (zero rows past
// B).  dh_rec[16 x H-slice] = dA[16 x 4H] . W_hh[4H x H-slice] with
// dA split into bf16 planes in LDS ([buffer 2][hi, lo][16][4H], chunk index XOR-ed with the row,
// k slot = 4*hidden + gate) and the image streamed one k-step (JPW hi + JPW lo fragments) per
// ring buffer, three k-steps in flight.
// PL: the gate gradients go out as bf16 planes (dgp: T step slots per window tile, columns in the k' = 4 * hidden +
// gate order of the LDS image) instead of fp32 tile16 blocks into G, which is then left as it was: the A fragments
// the product of step t + 1 reads from LDS are the tiles of dA_{t+1}, and wave w stores those of k-steps = w (mod 4).
template <int H, bool PL>
__global__ __launch_bounds__(RecCfg<H>::NW * 64) void lstm_rec3_bwd_kernel(
    float* __restrict__ G, const float* __restrict__ cseq, const float* __restrict__ dhseq,
    const float* __restrict__ dhlast, const float* __restrict__ dclast, float* __restrict__ dbpart,
    const u32x4* __restrict__ Wp, int B, int T, gp::SeqPlanes dgp) {
  using C = RecCfg<H>;
  using R = Rec3Cfg<H>;
  constexpr int JPW = C::JPW;
  constexpr int UT = 2 * JPW;                  // fragments per k-step and wave
  constexpr int KS = R::KS_B;
  constexpr bool RING_AG = false;              // register file of the ring buffers
  static_assert(KS % 4 == 0, "ring of four k-steps");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  constexpr int ROWB = 4 * H * 2;              // bytes per row of a plane
  constexpr int PLANE = 16 * ROWB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int bt = blockIdx.x;

  // running gradients start from the gradients w.r.t. h_T and c_T, given as tile16 blocks
  f32x4 dc[JPW], dh[JPW];
  float dbs[JPW][4];
#pragma unroll
  for (int jl = 0; jl < JPW; ++jl) {
    const long blk = ((long)bt * C::JT + wave * JPW + jl) * 256 + lane * 4;
    dh[jl] = dhlast != nullptr ? *reinterpret_cast<const f32x4*>(dhlast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
    dc[jl] = dclast != nullptr ? *reinterpret_cast<const f32x4*>(dclast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) dbs[jl][q] = 0.f;
  }
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(Wp) + (size_t)wave * (KS * UT * 1024);
  const unsigned voff = lane * 16;
  const unsigned pl_off = gp::slot(g, r) * 16;
  auto ks_ptr = [&](int ks) { return wbase + (size_t)(ks * UT) * 1024; };

  auto load_step = [&](f32x4 (&gv)[JPW * 4], f32x4 (&dv)[JPW], int t) {
    const float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) {
      const int jt = wave * JPW + jl;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        gv[jl * 4 + q] = ld_stream(Gt + (long)(q * C::JT + jt) * 256);
      if (dhseq != nullptr)
        dv[jl] = ld_stream(dhseq + ((long)(bt * T + t) * C::JT + jt) * 256 + lane * 4);
      else
        dv[jl] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto load_cells = [&](f32x4 (&cv)[JPW], int t) {
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) {
      if (t >= 0)
        cv[jl] = ld_stream(cseq + ((long)(bt * T + t) * C::JT + wave * JPW + jl) * 256 + lane * 4);
      else
        cv[jl] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x4 gn[JPW * 4], dn[JPW], cvt[JPW], cvp[JPW];
  load_step(gn, dn, T - 1);
  load_cells(cvt, T - 1);
  load_cells(cvp, T - 2);

  u32x4 w0[UT], w1[UT], w2[UT], w3[UT];
#pragma unroll
  for (int i = 0; i < UT; ++i) w0[i] = w1[i] = w2[i] = w3[i] = u32x4{0u, 0u, 0u, 0u};
  // hidden 64: the wave's share of the image (16 fragments) stays in registers, as in the forward kernel
  constexpr bool RESW = H == 64;
  static_assert(!(RESW && PL), "the plane stores ride in the streamed k-steps");
  u32x4 wr[RESW ? KS : 1][UT];
  if constexpr (RESW) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < UT; ++i) wr[ks][i] = *reinterpret_cast<const u32x4*>(ks_ptr(ks) + (size_t)i * 1024 + voff);
  } else {
    // the whole ring lives in AGPRs: the cell backward needs the arch VGPRs for its VALU operands
    w_issue<UT, RING_AG>(w0, ks_ptr(0), voff);
    w_issue<UT, RING_AG>(w1, ks_ptr(1), voff);
    w_issue<UT, RING_AG>(w2, ks_ptr(2), voff);
  }

  // gate gradients of step t from the running dh / dc: dA to LDS (bf16 planes, buffer t & 1) and
  // to the gates buffer, bias-gradient partial sums in registers
  auto cell_bwd = [&](int t) {
    const int cur = t & 1;
    float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
    unsigned char* da_s = smem3 + cur * (2 * PLANE);
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) {
      const int jt = wave * JPW + jl;
      const f32x4 gi = gn[jl * 4 + 0], gf = gn[jl * 4 + 1], gg = gn[jl * 4 + 2], go = gn[jl * 4 + 3];
      const f32x4 ct = cvt[jl], cp = cvp[jl];
      const f32x4 dht = dh[jl] + dn[jl];
      f32x4 dai, daf, dag, dao;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float tc = fast_tanh(ct[reg]);
        const float d_o = dht[reg] * tc;
        const float dcc = dc[jl][reg] + dht[reg] * go[reg] * (1.f - tc * tc);
        const float d_i = dcc * gg[reg];
        const float d_g = dcc * gi[reg];
        const float d_f = dcc * cp[reg];
        dc[jl][reg] = dcc * gf[reg];
        dai[reg] = d_i * gi[reg] * (1.f - gi[reg]);
        daf[reg] = d_f * gf[reg] * (1.f - gf[reg]);
        dag[reg] = d_g * (1.f - gg[reg] * gg[reg]);
        dao[reg] = d_o * go[reg] * (1.f - go[reg]);
        // the four gates of (window 4g+reg, hidden jt*16+r) are four consecutive k slots: 8 bytes
        // per plane
        uint2 hi, lo;
        split4(dai[reg], daf[reg], dag[reg], dao[reg], hi, lo);
        const int row = 4 * g + reg;
        const int j = jt * 16 + r;
        const int off = row * ROWB + ((((j >> 1) ^ (row & 15)) << 4) | ((j & 1) << 3));
        *reinterpret_cast<uint2*>(da_s + off) = hi;
        *reinterpret_cast<uint2*>(da_s + PLANE + off) = lo;
      }
      dbs[jl][0] += (dai[0] + dai[1]) + (dai[2] + dai[3]);
      dbs[jl][1] += (daf[0] + daf[1]) + (daf[2] + daf[3]);
      dbs[jl][2] += (dag[0] + dag[1]) + (dag[2] + dag[3]);
      dbs[jl][3] += (dao[0] + dao[1]) + (dao[2] + dao[3]);
      if (!PL) {
        st_stream(Gt + (long)(0 * C::JT + jt) * 256, dai);
        st_stream(Gt + (long)(1 * C::JT + jt) * 256, daf);
        st_stream(Gt + (long)(2 * C::JT + jt) * 256, dag);
        st_stream(Gt + (long)(3 * C::JT + jt) * 256, dao);
      }
      // one hidden tile at a time: interleaving the four tiles' arithmetic quadruples the
      // temporaries live at once, and this kernel has no registers to spare
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Loop shape: [product of step t+1 -> dh][cell backward of step t], the first cell backward
  // peeled off.  One loop exit and no branch around the product: any other shape lets the
  // compiler reconcile the ring buffers' registers with moves while their loads are in flight.
  cell_bwd(T - 1);
  for (int t = T - 2; t >= 0; --t) {
    // saved tensors of step t: in flight during the product below
    load_step(gn, dn, t);
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) cvt[jl] = cvp[jl];
    load_cells(cvp, t - 1);
    lds_barrier();
    const unsigned char* da_s = smem3 + ((t + 1) & 1) * (2 * PLANE);
    f32x4 acc[JPW];
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) acc[jl] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto lds_a = [&](int plane, int ks) {
      return *reinterpret_cast<const u32x4*>(da_s + plane * PLANE + r * ROWB + (((4 * ks + g) ^ r) << 4));
    };
    u32x4 ah = lds_a(0, 0), al = lds_a(1, 0);
    // one k-step: fragments w[0..JPW) = hi, w[JPW..2*JPW) = lo of this wave's JPW output tiles; `allowed` = how
    // many younger vector-memory loads may still be outstanding when this buffer's loads have landed
    char* dg_t = PL ? dgp.base + dgp.tile(bt, t + 1, 0, 0) : nullptr;     // tiles of dA_{t+1}
    auto kstep = [&](u32x4 (&w)[UT], int ks, auto allowed) {
      const int kn = (ks + 1) & (KS - 1);
      const u32x4 nh = lds_a(0, kn), nl = lds_a(1, kn);
      if (PL && (ks & 3) == wave) {
        __builtin_nontemporal_store(ah, reinterpret_cast<u32x4*>(dg_t + ks * (2 * gp::TILE_B) + pl_off));
        __builtin_nontemporal_store(al, reinterpret_cast<u32x4*>(dg_t + ks * (2 * gp::TILE_B) + gp::TILE_B + pl_off));
      }
      w_wait<decltype(allowed)::value, UT, RING_AG>(w);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl)
        acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), as_bf16x8(w[jl]), acc[jl], 0, 0, 0);
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl)
        acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(w[JPW + jl]), acc[jl], 0, 0, 0);
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl)
        acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(w[jl]), acc[jl], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ah = nh;
      al = nl;
    };
    // In steady state three ring buffers' loads (3 * UT) are younger than the one being waited for.  The first
    // three k-steps of a product are different: the step's 4 * JPW = 2 * UT gate loads (load_step above, always
    // issued; loads return in order) also sit between the ring loads issued before the cell backward and the
    // ones issued here, so 5 * UT loads may stay outstanding.  Waiting for 3 * UT there would drain the gate
    // loads - a whole HBM latency - and the cell backward's stores before the product's first MFMA.
    constexpr std::integral_constant<int, 3 * UT> steady{};
    constexpr std::integral_constant<int, 5 * UT> first{};
    if constexpr (RESW) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kn = (ks + 1) & (KS - 1);
        const u32x4 nh = lds_a(0, kn), nl = lds_a(1, kn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jl = 0; jl < JPW; ++jl)
          acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), as_bf16x8(wr[ks][jl]), acc[jl], 0, 0, 0);
#pragma unroll
        for (int jl = 0; jl < JPW; ++jl)
          acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(wr[ks][JPW + jl]), acc[jl], 0, 0, 0);
#pragma unroll
        for (int jl = 0; jl < JPW; ++jl)
          acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(wr[ks][jl]), acc[jl], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ah = nh;
        al = nl;
      }
    } else {
    w_issue<UT, RING_AG>(w3, ks_ptr(3), voff);
    kstep(w0, 0, first);
    w_issue<UT, RING_AG>(w0, ks_ptr(4 & (KS - 1)), voff);
    kstep(w1, 1, first);
    w_issue<UT, RING_AG>(w1, ks_ptr(5 & (KS - 1)), voff);
    kstep(w2, 2, first);
    w_issue<UT, RING_AG>(w2, ks_ptr(6 & (KS - 1)), voff);
    kstep(w3, 3, steady);
#pragma unroll 1
    for (int ks = 4; ks < KS; ks += 4) {
      w_issue<UT, RING_AG>(w3, ks_ptr(ks + 3), voff);
      kstep(w0, ks, steady);
      w_issue<UT, RING_AG>(w0, ks_ptr((ks + 4) & (KS - 1)), voff);
      kstep(w1, ks + 1, steady);
      w_issue<UT, RING_AG>(w1, ks_ptr((ks + 5) & (KS - 1)), voff);
      kstep(w2, ks + 2, steady);
      w_issue<UT, RING_AG>(w2, ks_ptr((ks + 6) & (KS - 1)), voff);
      kstep(w3, ks + 3, steady);
    }
    }
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) dh[jl] = acc[jl];
    cell_bwd(t);
  }
  // keep the wrapped-around ring loads' registers live until they have landed (see the forward kernel)
  if constexpr (!RESW) {
    w_wait<0, UT, RING_AG>(w0);
    w_wait<0, UT, RING_AG>(w1);
    w_wait<0, UT, RING_AG>(w2);
  }
  if (PL) {   // dA_0: no product reads it, copy its tiles out of LDS
    lds_barrier();
    const unsigned char* da_s = smem3;   // buffer 0 & 1
    char* dg_t = dgp.base + dgp.tile(bt, 0, 0, 0);
    for (int ks = wave; ks < KS; ks += 4)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        __builtin_nontemporal_store(
            *reinterpret_cast<const u32x4*>(da_s + p * PLANE + r * ROWB + (((4 * ks + g) ^ r) << 4)),
            reinterpret_cast<u32x4*>(dg_t + (ks * 2 + p) * gp::TILE_B + pl_off));
  }

  // bias-gradient partials: sum over the 4 row groups (lanes r, r+16, r+32, r+48)
#pragma unroll
  for (int jl = 0; jl < JPW; ++jl) {
    const int jt = wave * JPW + jl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s = dbs[jl][q];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (g == 0) dbpart[(long)bt * (4 * H) + q * H + jt * 16 + r] = s;
    }
  }
}

// ---- backward, saved gates staged through LDS -----------------------------------------------------
// Same product and cell arithmetic as lstm_rec3_bwd_kernel; what differs is when the step's saved tensors are
// fetched.  Vector memory loads complete in issue order, so the 20-odd HBM loads of a step's gates, cells and dh,
// issued at the top of a product, hold up every W_hh ring load issued behind them until the slowest of them is
// back (a quarter of the kernel's time at the bench shape).  Here the batch for step t - 1 is issued at the start
// of step t's cell phase, behind ring loads for the next product's first FOUR k-steps: the next ring load behind
// the batch is consumed a whole cell phase and four k-steps later.  The gates travel by LDS-DMA into a per-wave
// staging area (no registers, so the batch can fly while the cell phase still uses the previous one; the cell
// phase first copies its gates from staging to registers, then issues the next batch), cells and dh into a second
// set of registers by inline-asm loads.  dA is single-buffered in LDS to make room: two barriers per step.
// DH: a dh sequence is given (four more loads per batch - the hand-counted waits need the count at compile time).
// LDS: dA planes [hi, lo][16][4H] bf16, then staging [wave][JPW * 4 gate tiles][1 KiB].
template <int H, bool PL, bool DH>
__global__ __launch_bounds__(RecCfg<H>::NW * 64) void lstm_rec3_bwd_staged_kernel(
    float* __restrict__ G, const float* __restrict__ cseq, const float* __restrict__ dhseq,
    const float* __restrict__ dhlast, const float* __restrict__ dclast, float* __restrict__ dbpart,
    const u32x4* __restrict__ Wp, int B, int T, gp::SeqPlanes dgp) {
  using C = RecCfg<H>;
  using R = Rec3Cfg<H>;
  constexpr int JPW = C::JPW;
  constexpr int UT = 2 * JPW;                  // fragments per k-step and wave
  constexpr int KS = R::KS_B;
  constexpr int NB = 4 * JPW + JPW + (DH ? JPW : 0);   // loads per batch: gate tiles, cells, dh
  constexpr bool RING_AG = false;              // (true: the allocator spills and writes into ring registers in flight)
  constexpr int PRE = JPW - 1;                 // hidden tiles whose coefficients are worked out inside the product
  constexpr int RES = 4;                       // k-steps of the image resident in registers
  static_assert(KS % 4 == 0 && KS >= RES + 8, "ring of four k-steps, batch retired by the fifth streamed one");
  static_assert(3 * UT + NB <= 63, "vmcnt is a six-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  constexpr int ROWB = 4 * H * 2;              // bytes per row of a plane
  constexpr int PLANE = 16 * ROWB;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int bt = blockIdx.x;
  unsigned char* const da_s = smem3;
  constexpr int STG = 6 * JPW * 1024;          // staging bytes per wave
  unsigned char* const stg = smem3 + 2 * PLANE + wave * STG;
  const unsigned stg_lds = __builtin_amdgcn_readfirstlane(gp::lds_addr(reinterpret_cast<const char*>(stg)));

  f32x4 dc[JPW], dh[JPW];
  float dbs[JPW][4];
#pragma unroll
  for (int jl = 0; jl < JPW; ++jl) {
    const long blk = ((long)bt * C::JT + wave * JPW + jl) * 256 + lane * 4;
    dh[jl] = dhlast != nullptr ? *reinterpret_cast<const f32x4*>(dhlast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
    dc[jl] = dclast != nullptr ? *reinterpret_cast<const f32x4*>(dclast + blk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) dbs[jl][q] = 0.f;
  }
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(Wp) + (size_t)wave * (KS * UT * 1024);
  const unsigned voff = lane * 16;
  const unsigned pl_off = gp::slot(g, r) * 16;
  auto ks_ptr = [&](int ks) { return wbase + (size_t)(ks * UT) * 1024; };
  // LDS traffic of this wave done, then the barrier.  Not lds_barrier(): with LDS-DMA in the kernel its release
  // fence becomes s_waitcnt vmcnt(0), which would drain the ring and the batch in flight at every barrier.
  auto wg_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // the batch of step t: G_t, c_{t-1} (c_0 when there is none: replaced by zeros when it is taken over) and dh_t by
  // LDS-DMA into staging.  Always NB loads.
  // LDS-DMA as inline asm: through the builtin the compiler knows that vector memory is writing LDS and puts
  // s_waitcnt vmcnt(0) in front of the cell phase's first LDS write (and of every fenced barrier)
  auto dma = [&](unsigned lds_off, const float* src) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_off), "v"(voff), "s"(src) : "memory", "m0");
  };
  auto issue_batch = [&](int t) {
    const float* Gt = G + ((long)(bt * T + t) * (4 * C::JT) + wave * JPW) * 256;
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl)
#pragma unroll
      for (int q = 0; q < 4; ++q) dma(stg_lds + (jl * 4 + q) * 1024, Gt + (long)(q * C::JT + jl) * 256);
    const int tc = t >= 1 ? t - 1 : 0;
    const float* cb = cseq + ((long)(bt * T + tc) * C::JT + wave * JPW) * 256;
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) dma(stg_lds + (4 * JPW + jl) * 1024, cb + jl * 256);
    if (DH) {
      const float* db = dhseq + ((long)(bt * T + t) * C::JT + wave * JPW) * 256;
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl) dma(stg_lds + (5 * JPW + jl) * 1024, db + jl * 256);
    }
  };

  // What the cell phase of a step needs of its saved tensors does not depend on dh, so it is worked out ahead, inside
  // the product that computes dh (where the VALU idles under the W_hh stream): per element
  //   kA = o (1 - tanh(c_t)^2)        dc_total = dc + dh kA
  //   kO = tanh(c_t) o (1 - o)        dA_o = dh kO
  //   kI = g i (1 - i)                dA_i = dc_total kI
  //   kF = c_{t-1} f (1 - f)          dA_f = dc_total kF
  //   kG = i (1 - g^2)                dA_g = dc_total kG
  //   kf = f                          dc   = dc_total kf
  // and tanh(c_{t-1}) is kept for the step after (tcn).
  f32x4 kA[JPW], kO[JPW], kI[JPW], kF[JPW], kG[JPW], kf[JPW], tcn[JPW];
#pragma unroll
  for (int jl = 0; jl < JPW; ++jl) {
    const f32x4 c = ld_stream(cseq + ((long)(bt * T + T - 1) * C::JT + wave * JPW + jl) * 256 + lane * 4);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) tcn[jl][reg] = fast_tanh(c[reg]);
  }

  u32x4 w0[UT], w1[UT], w2[UT], w3[UT];
#pragma unroll
  for (int i = 0; i < UT; ++i) w0[i] = w1[i] = w2[i] = w3[i] = u32x4{0u, 0u, 0u, 0u};
  // k-steps 0..RES-1 of the image stay in registers for the whole sequence: an eighth less to stream per step
  u32x4 wr[RES][UT];
#pragma unroll
  for (int i = 0; i < RES; ++i)
#pragma unroll
    for (int j = 0; j < UT; ++j)
      wr[i][j] = *reinterpret_cast<const u32x4*>(ks_ptr(i) + j * 1024 + voff);
  w_issue<UT, RING_AG>(w0, ks_ptr(RES + 0), voff);
  w_issue<UT, RING_AG>(w1, ks_ptr(RES + 1), voff);
  w_issue<UT, RING_AG>(w2, ks_ptr(RES + 2), voff);
  w_issue<UT, RING_AG>(w3, ks_ptr(RES + 3), voff);
  issue_batch(T - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // hidden tile jl of the staged batch of step t -> the coefficients above
  auto premath = [&](int jl, int t) {
    asm volatile("" ::: "memory");     // the staging reads stay behind the wait that proved the batch landed
    const unsigned char* sj = stg + lane * 16;
    const f32x4 gi = *reinterpret_cast<const f32x4*>(sj + (jl * 4 + 0) * 1024);
    const f32x4 gf = *reinterpret_cast<const f32x4*>(sj + (jl * 4 + 1) * 1024);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(sj + (jl * 4 + 2) * 1024);
    const f32x4 go = *reinterpret_cast<const f32x4*>(sj + (jl * 4 + 3) * 1024);
    f32x4 cp = *reinterpret_cast<const f32x4*>(sj + (4 * JPW + jl) * 1024);
    if (t < 1) cp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float tc = tcn[jl][reg];
      kA[jl][reg] = go[reg] * (1.f - tc * tc);
      kO[jl][reg] = tc * (go[reg] * (1.f - go[reg]));
      kI[jl][reg] = gg[reg] * (gi[reg] * (1.f - gi[reg]));
      kF[jl][reg] = cp[reg] * (gf[reg] * (1.f - gf[reg]));
      kG[jl][reg] = gi[reg] * (1.f - gg[reg] * gg[reg]);
      kf[jl][reg] = gf[reg];
      tcn[jl][reg] = fast_tanh(cp[reg]);
    }
  };
  // dh_t of the staged batch joins the recurrent part; after it the staging area is free for the next batch
  auto take_dh = [&] {
    asm volatile("" ::: "memory");
    if (DH) {
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl) dh[jl] += *reinterpret_cast<const f32x4*>(stg + (5 * JPW + jl) * 1024 + lane * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  auto cell_bwd = [&](int t) {
    float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
    char* dg_t = PL ? dgp.base + dgp.tile(bt, t, 2 * wave * JPW, 0) + pl_off : nullptr;
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) {
      const int jt = wave * JPW + jl;
      const f32x4 dht = dh[jl];
      f32x4 dai, daf, dag, dao;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float dcc = dc[jl][reg] + dht[reg] * kA[jl][reg];
        dc[jl][reg] = dcc * kf[jl][reg];
        dai[reg] = dcc * kI[jl][reg];
        daf[reg] = dcc * kF[jl][reg];
        dag[reg] = dcc * kG[jl][reg];
        dao[reg] = dht[reg] * kO[jl][reg];
        uint2 hi, lo;
        split4(dai[reg], daf[reg], dag[reg], dao[reg], hi, lo);
        const int row = 4 * g + reg;
        const int j = jt * 16 + r;
        const int off = row * ROWB + ((((j >> 1) ^ (row & 15)) << 4) | ((j & 1) << 3));
        *reinterpret_cast<uint2*>(da_s + off) = hi;
        *reinterpret_cast<uint2*>(da_s + PLANE + off) = lo;
      }
      dbs[jl][0] += (dai[0] + dai[1]) + (dai[2] + dai[3]);
      dbs[jl][1] += (daf[0] + daf[1]) + (daf[2] + daf[3]);
      dbs[jl][2] += (dag[0] + dag[1]) + (dag[2] + dag[3]);
      dbs[jl][3] += (dao[0] + dao[1]) + (dao[2] + dao[3]);
      if (!PL) {
        st_stream(Gt + (long)(0 * C::JT + jt) * 256, dai);
        st_stream(Gt + (long)(1 * C::JT + jt) * 256, daf);
        st_stream(Gt + (long)(2 * C::JT + jt) * 256, dag);
        st_stream(Gt + (long)(3 * C::JT + jt) * 256, dao);
      } else {
        // The plane tiles of this hidden tile (k-steps 2 * jt and 2 * jt + 1, both planes) are the wave's own LDS
        // writes read back in fragment order - LDS operations of a wave execute in order, no barrier needed - and go
        // out at once, while the W_hh stream pauses anyway.
        // (Stored from inside the k-steps, as the product reads the fragments, they cost the stream a tenth of
        // its rate.)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {     // one k-step (two tiles) at a time: eight registers, not sixteen
          u32x4 pf[2];
#pragma unroll
          for (int p = 0; p < 2; ++p)
            pf[p] = *reinterpret_cast<const u32x4*>(da_s + p * PLANE + r * ROWB + (((4 * (2 * jt + hf) + g) ^ r) << 4));
#pragma unroll
          for (int p = 0; p < 2; ++p)
            __builtin_nontemporal_store(pf[p], reinterpret_cast<u32x4*>(dg_t + (4 * jl + 2 * hf + p) * gp::TILE_B));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // (the last step's "next batch" is step 0 once more: an unconditional issue keeps the count of loads in the
  // queue the same on every path, which is what the hand-counted waits - and their checker - rest on)
#ifdef VTS_RB_STAMP
  // timing experiments: ticks per phase, summed over the steps, written over the first bias-gradient partials
  long long rstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long rprev = clock64();
#define RSTAMP(i) do { const long long now = clock64(); rstamp[i] += now - rprev; rprev = now; } while (0)
#else
#define RSTAMP(i)
#endif
#pragma unroll
  for (int jl = 0; jl < JPW; ++jl) premath(jl, T - 1);
  take_dh();
  issue_batch(T >= 2 ? T - 2 : 0);
  cell_bwd(T - 1);
  for (int t = T - 2; t >= 0; --t) {
    RSTAMP(6);                                    // (cell backward of the step before)
    wg_barrier();                                 // dA_{t+1} is complete
    RSTAMP(0);
    f32x4 acc[JPW];
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) acc[jl] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto lds_a = [&](int plane, int ks) {
      return *reinterpret_cast<const u32x4*>(da_s + plane * PLANE + r * ROWB + (((4 * ks + g) ^ r) << 4));
    };
    // one k-step (kn, the k-step that runs next, is not used: the A fragments are fetched just in time - their
    // LDS latency hides behind the wait for the ring buffer, and a prefetched pair would cost eight registers this
    // kernel does not have); allowed < 0: a resident buffer, nothing to wait for
    auto kstep = [&](u32x4 (&w)[UT], int ks, int kn, auto allowed) {
      (void)kn;
      const u32x4 ah = lds_a(0, ks), al = lds_a(1, ks);
      if constexpr (decltype(allowed)::value >= 0) w_wait<decltype(allowed)::value, UT, RING_AG>(w);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl)
        acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), as_bf16x8(w[jl]), acc[jl], 0, 0, 0);
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl)
        acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(w[JPW + jl]), acc[jl], 0, 0, 0);
#pragma unroll
      for (int jl = 0; jl < JPW; ++jl)
        acc[jl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), as_bf16x8(w[jl]), acc[jl], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // On entry the queue holds, oldest first: w0 w1 w2 w3 (k-steps RES..RES+3, issued at the end of the previous
    // product), then the batch of step t (NB loads).  Each streamed k-step waits for its buffer, then refills it
    // four streamed k-steps ahead: three buffers (3 * UT loads) are younger than the one waited for, and during
    // the first four so is the batch.  The resident k-steps run after those four (more time for the batch); the
    // fifth streamed k-step's wait retires the batch.
    constexpr std::integral_constant<int, 3 * UT> steady{};
    constexpr std::integral_constant<int, 3 * UT + NB> first{};
    constexpr std::integral_constant<int, -1> resident{};
    kstep(w0, RES + 0, RES + 1, first);
    w_issue<UT, RING_AG>(w0, ks_ptr(RES + 4), voff);
    kstep(w1, RES + 1, RES + 2, first);
    w_issue<UT, RING_AG>(w1, ks_ptr(RES + 5), voff);
    kstep(w2, RES + 2, RES + 3, first);
    w_issue<UT, RING_AG>(w2, ks_ptr(RES + 6), voff);
    kstep(w3, RES + 3, 0, first);
    w_issue<UT, RING_AG>(w3, ks_ptr(RES + 7), voff);
    RSTAMP(1);
#pragma unroll
    for (int i = 0; i < RES; ++i) kstep(wr[i], i, i + 1 < RES ? i + 1 : RES + 4, resident);
    RSTAMP(2);
    // four streamed k-steps ks..ks+3; refills four streamed k-steps ahead, wrapping to the next step's first ones
    auto group = [&](int ks) {
      const int nx = ks + 4 < KS ? ks + 4 : RES;
      kstep(w0, ks, ks + 1, steady);
      w_issue<UT, RING_AG>(w0, ks_ptr(nx), voff);
      kstep(w1, ks + 1, ks + 2, steady);
      w_issue<UT, RING_AG>(w1, ks_ptr(nx + 1), voff);
      kstep(w2, ks + 2, ks + 3, steady);
      w_issue<UT, RING_AG>(w2, ks_ptr(nx + 2), voff);
      kstep(w3, ks + 3, ks + 4 < KS ? ks + 4 : RES, steady);
      w_issue<UT, RING_AG>(w3, ks_ptr(nx + 3), voff);
    };
    // the first group's first wait retires the batch of step t: each of the next JPW groups is followed by the
    // coefficient arithmetic of one hidden tile, which the VALU does while the matrix pipe and the ring are busy
    // (PRE of the JPW tiles: the registers do not hold the coefficients of all of them beside the ring and the
    // resident part of the image; the others follow when the product is over)
    static_assert(RES + 4 + 4 * PRE <= KS, "a group of k-steps per hidden tile");
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
      group(RES + 4 + 4 * i);
      premath(i, t);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int ks = RES + 4 + 4 * PRE; ks < KS; ks += 4) group(ks);
    RSTAMP(3);
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) dh[jl] = acc[jl];
#pragma unroll
    for (int i = PRE; i < JPW; ++i) premath(i, t);
    take_dh();
    issue_batch(t >= 1 ? t - 1 : 0);
    RSTAMP(4);
    wg_barrier();                                 // every wave has read the last fragments of dA_{t+1}
    RSTAMP(5);
    cell_bwd(t);
  }
  // keep the wrapped-around ring loads' registers live until they have landed
  w_wait<0, UT, RING_AG>(w0);
  w_wait<0, UT, RING_AG>(w1);
  w_wait<0, UT, RING_AG>(w2);
  w_wait<0, UT, RING_AG>(w3);

#pragma unroll
  for (int jl = 0; jl < JPW; ++jl) {
    const int jt = wave * JPW + jl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float s = dbs[jl][q];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (g == 0) dbpart[(long)bt * (4 * H) + q * H + jt * 16 + r] = s;
    }
  }
#ifdef VTS_RB_STAMP
  __syncthreads();
  if (tid == 0)
    for (int i = 0; i < 8; ++i) dbpart[(long)bt * (4 * H) + i] = (float)rstamp[i] / (float)(T - 1);
#endif
#undef RSTAMP
}

template <int H, bool PL>
int run_fwd3(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp, int B,
             int T, gp::SeqPlanes hp, hipStream_t stream) {
  using C = RecCfg<H>;
  using R = Rec3Cfg<H>;
  u32x4* img = reinterpret_cast<u32x4*>(Wp);
  hipLaunchKernelGGL((pack_whh3_kernel<H, false>), dim3(vts_cdiv(H * H, 256)), dim3(256), 0, stream, W_hh, img);
  VTS_CHECK_LAUNCH("pack_whh3_fwd");
  static vts_lds_mask lds_done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec3_fwd_kernel<H, PL>), R::LDS_F, &lds_done, "lstm_rec3_fwd")) return rc;
  hipLaunchKernelGGL((lstm_rec3_fwd_kernel<H, PL>), dim3((B + 15) / 16), dim3(C::NW * 64), R::LDS_F, stream, G, cseq, hseq,
                     hlast, clast, img, B, T, hp);
  VTS_CHECK_LAUNCH("lstm_rec3_fwd");
  return VTS_OK;
}

template <int H, bool PL>
int run_bwd3(float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
             float* dbpart, const float* W_hh, float* Wp, int B, int T, gp::SeqPlanes dgp, hipStream_t stream) {
  using C = RecCfg<H>;
  using R = Rec3Cfg<H>;
  u32x4* img = reinterpret_cast<u32x4*>(Wp);
  hipLaunchKernelGGL((pack_whh3_kernel<H, true>), dim3(vts_cdiv(H * H, 256)), dim3(256), 0, stream, W_hh, img);
  VTS_CHECK_LAUNCH("pack_whh3_bwd");
  if constexpr (H == 256) {
    if (g_staged_bwd.load(std::memory_order_relaxed)) {
      constexpr size_t LDS = R::LDS_B / 2 + (size_t)C::NW * 6 * C::JPW * 1024;
      static vts_lds_mask done_dh{0}, done_nodh{0};
      if (dhseq != nullptr) {
        if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec3_bwd_staged_kernel<H, PL, true>), LDS, &done_dh, "lstm_rec3_bwd_staged")) return rc;
        hipLaunchKernelGGL((lstm_rec3_bwd_staged_kernel<H, PL, true>), dim3((B + 15) / 16), dim3(C::NW * 64), LDS, stream, G,
                           cseq, dhseq, dhlast, dclast, dbpart, img, B, T, dgp);
      } else {
        if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec3_bwd_staged_kernel<H, PL, false>), LDS, &done_nodh, "lstm_rec3_bwd_staged")) return rc;
        hipLaunchKernelGGL((lstm_rec3_bwd_staged_kernel<H, PL, false>), dim3((B + 15) / 16), dim3(C::NW * 64), LDS, stream, G,
                           cseq, dhseq, dhlast, dclast, dbpart, img, B, T, dgp);
      }
      VTS_CHECK_LAUNCH("lstm_rec3_bwd_staged");
      return VTS_OK;
    }
  }
  static vts_lds_mask lds_done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec3_bwd_kernel<H, PL>), R::LDS_B, &lds_done, "lstm_rec3_bwd")) return rc;
  hipLaunchKernelGGL((lstm_rec3_bwd_kernel<H, PL>), dim3((B + 15) / 16), dim3(C::NW * 64), R::LDS_B, stream, G, cseq, dhseq,
                     dhlast, dclast, dbpart, img, B, T, dgp);
  VTS_CHECK_LAUNCH("lstm_rec3_bwd");
  return VTS_OK;
}

}  // namespace

void vts_lstm_rec3_set_staged_bwd(int on) { g_staged_bwd.store(on ? 1 : 0, std::memory_order_relaxed); }

bool vts_lstm_rec3_supported(int H) { return H == 64 || H == 128 || H == 256; }

// h_planes / dg_planes (hidden 256 only): plane outputs as described at the kernels; null = the fp32 outputs only
int vts_lstm_rec3_fwd(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp,
                      int B, int T, int H, void* h_planes, hipStream_t stream) {
  const gp::SeqPlanes none{nullptr, 0, 0, 0};
  if (h_planes != nullptr) {
    if (H != 256) { vts_set_error("lstm bf16x3 recurrence: plane output needs hidden size 256"); return VTS_ERR_UNSUPPORTED; }
    return run_fwd3<256, true>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T,
                               gp::SeqPlanes{reinterpret_cast<char*>(h_planes), 8, T + 1, 1}, stream);
  }
  switch (H) {
    case 64: return run_fwd3<64, false>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, none, stream);
    case 128: return run_fwd3<128, false>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, none, stream);
    case 256: return run_fwd3<256, false>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, none, stream);
  }
  vts_set_error("lstm bf16x3 recurrence: hidden size %d not supported (64, 128, 256)", H);
  return VTS_ERR_UNSUPPORTED;
}

int vts_lstm_rec3_bwd(float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                      float* dbpart, const float* W_hh, float* Wp, int B, int T, int H, void* dg_planes,
                      hipStream_t stream) {
  const gp::SeqPlanes none{nullptr, 0, 0, 0};
  if (dg_planes != nullptr) {
    if (H != 256) { vts_set_error("lstm bf16x3 recurrence: plane output needs hidden size 256"); return VTS_ERR_UNSUPPORTED; }
    return run_bwd3<256, true>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T,
                               gp::SeqPlanes{reinterpret_cast<char*>(dg_planes), 32, T, 0}, stream);
  }
  switch (H) {
    case 64: return run_bwd3<64, false>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, none, stream);
    case 128: return run_bwd3<128, false>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, none, stream);
    case 256: return run_bwd3<256, false>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, none, stream);
  }
  vts_set_error("lstm bf16x3 recurrence: hidden size %d not supported (64, 128, 256)", H);
  return VTS_ERR_UNSUPPORTED;
}
