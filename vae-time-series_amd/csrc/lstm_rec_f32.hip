// Persistent LSTM recurrence kernels, fp32 on v_mfma_f32_16x16x4_f32.
//
// One workgroup owns 16 windows (one 16-row batch tile) for the whole sequence, so there is no
// inter-workgroup communication at all: h_t and c_t never leave the CU between steps.  Wave w
// owns hidden units [w*JPW*16, (w+1)*JPW*16) for ALL four gates, so the cell update is
// lane-local in the MFMA accumulator layout.  W_hh is streamed from L2 every step in a
// pre-packed, fragment-ordered image (one coalesced 1 KiB load per MFMA quad).
//
// Gate pre-activations / activations / gate gradients live in the tile16 layout
// (vts_common.h) of a [Bp*T, 4H] matrix with logical row m' = (bt*T + t)*16 + bi.
#include "lstm_rec_common.h"

namespace {

// ---- weight packing ---------------------------------------------------------------------------
// forward image: [wave][kq][jl][gate][lane][s] = W_hh[gate*H + (wave*JPW+jl)*16 + (lane&15)][16*kq + 4*(lane>>4) + s]
template <int H>
__global__ void pack_whh_fwd_kernel(const float* __restrict__ W, float* __restrict__ out) {
  using C = RecCfg<H>;
  const int total = 4 * H * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int s = idx & 3;
    const int lane = (idx >> 2) & 63;
    int rest = idx >> 8;
    const int gate = rest & 3; rest >>= 2;
    const int jl = rest % C::JPW; rest /= C::JPW;
    const int kq = rest % C::KQ_F; rest /= C::KQ_F;
    const int wave = rest;
    const int row = gate * H + (wave * C::JPW + jl) * 16 + (lane & 15);
    const int col = 16 * kq + 4 * (lane >> 4) + s;
    out[idx] = W[row * H + col];
  }
}

// backward image: [wave][kq][jl][lane][s] = W_hh[s*H + 4*kq + (lane>>4)][(wave*JPW+jl)*16 + (lane&15)]
template <int H>
__global__ void pack_whh_bwd_kernel(const float* __restrict__ W, float* __restrict__ out) {
  using C = RecCfg<H>;
  const int total = 4 * H * H;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int s = idx & 3;
    const int lane = (idx >> 2) & 63;
    int rest = idx >> 8;
    const int jl = rest % C::JPW; rest /= C::JPW;
    const int kq = rest % C::KQ_B; rest /= C::KQ_B;
    const int wave = rest;
    const int row = s * H + 4 * kq + (lane >> 4);
    const int col = (wave * C::JPW + jl) * 16 + (lane & 15);
    out[idx] = W[row * H + col];
  }
}

// ---- forward ------------------------------------------------------------------------------------
// G     : in  gate pre-activations x.W_ih^T + b (tile16 [Bp*T, 4H]); out activated gates i,f,g,o
// cseq  : out cell states (tile16 [Bp*T, H])
// hseq  : out hidden states, row-major [B, T, H]
// hlast, clast : out optional h_T, c_T row-major [B, H]
//
// Per step and wave: NT = 4*JPW accumulator tiles, KQ_F k-quads; the W_hh fragments of k-quad
// kq+1 are in flight (registers w1 / w0 alternately) while the MFMAs of k-quad kq issue, and the
// gate pre-activations of step t+1 are in flight during the whole product of step t.
template <int H>
__global__ __launch_bounds__(RecCfg<H>::NW * 64) void lstm_rec_fwd_kernel(
    float* __restrict__ G, float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hlast,
    float* __restrict__ clast, const float* __restrict__ Wp, int B, int T) {
  using C = RecCfg<H>;
  constexpr int NT = C::JPW * 4;
  // LDS: the double-buffered h tile, [2][16][H] floats with the 16-byte column index XOR-ed with
  // the row (conflict-free fragment reads without padding: the tile is a power of two and the
  // rest of the 160 KiB holds W_hh), then NLB resident W_hh k-quads in fragment order.
  constexpr int NLB = C::KQ_F / 8;                  // k-quads of W_hh kept in LDS (2 at H = 256)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hl0 = smem;
  f32x4* wlds = reinterpret_cast<f32x4*>(smem + 2 * 16 * H);   // [NLB][NT][NW*64]
  constexpr int SWZ = (H / 4 < 16 ? H / 4 : 16) - 1;
  auto hidx = [](int row, int col) { return row * H + ((((col >> 2) ^ (row & SWZ)) << 2) | (col & 3)); };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int bt = blockIdx.x;
  const int b0 = bt * 16;

  f32x4 c[C::JPW];
#pragma unroll
  for (int jl = 0; jl < C::JPW; ++jl) c[jl] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* Wv = reinterpret_cast<const f32x4*>(Wp) + (long)wave * C::KQ_F * NT * 64 + lane;
  auto load_w = [&](f32x4 (&w)[NT], int kq) {
#pragma unroll
    for (int i = 0; i < NT; ++i) w[i] = Wv[(kq * NT + i) * 64];   // i = jl*4 + gate
  };
  auto load_g = [&](f32x4 (&gv)[NT], int t) {
    const float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
#pragma unroll
    for (int jl = 0; jl < C::JPW; ++jl)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        gv[jl * 4 + q] = ld_stream(Gt + (long)(q * C::JT + wave * C::JPW + jl) * 256);
  };

  // k-quad 8*n + 7 lives in LDS slot n (each wave keeps and re-reads only its own fragments)
  auto load_w_lds = [&](f32x4 (&w)[NT], int slot) {
#pragma unroll
    for (int i = 0; i < NT; ++i) w[i] = wlds[(slot * NT + i) * (C::NW * 64) + tid];
  };
  if constexpr (NLB > 0) {
#pragma unroll 1
    for (int n = 0; n < NLB; ++n)
#pragma unroll
      for (int i = 0; i < NT; ++i) wlds[(n * NT + i) * (C::NW * 64) + tid] = Wv[((8 * n + 7) * NT + i) * 64];
  }

  f32x4 gnext[NT];
  load_g(gnext, 0);

  // W_hh fragment ring: three k-quads ahead of the MFMAs, wrapping into the next step (the
  // weights do not change with t), so ~48 KiB per wave stay in flight to cover the L2 latency.
  constexpr bool RING = (C::KQ_F % 4 == 0);
  f32x4 w0[NT], w1[NT], w2[NT], w3[NT];
  // Four k-quads in all (hidden 64): the four ring buffers hold the wave's whole share of the image, which is
  // loaded once - re-loading the same fragments every step put an L2 round trip in front of each block of MFMAs.
  constexpr bool RESW = RING && C::KQ_F == 4;
  if constexpr (RING) {
    load_w(w0, 0);
    load_w(w1, 1);
    load_w(w2, 2);
    if constexpr (RESW) load_w(w3, 3);
  }

  for (int t = 0; t < T; ++t) {
    const int cur = t & 1, nxt = cur ^ 1;
    float* hcur = hl0 + cur * (16 * H);
    float* hnxt = hl0 + nxt * (16 * H);
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = gnext[i];
    if (t + 1 < T) load_g(gnext, t + 1);

    if (t > 0) {
      auto lds_a = [&](int kq) { return *reinterpret_cast<const f32x4*>(&hcur[hidx(r, 16 * kq + 4 * g)]); };
      auto mma_a = [&](const f32x4 (&w)[NT], const f32x4 a) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < NT; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], w[i][s], acc[i], 0, 0, 0);
      };
      auto mma = [&](const f32x4 (&w)[NT], int kq) { mma_a(w, lds_a(kq)); };
      if constexpr (RING) {
        // the A fragment (h rows) of block kq+1 is read from LDS before the 64 MFMAs of block kq:
        // with one wave per SIMD an exposed LDS round trip is an idle matrix pipe
        f32x4 a = lds_a(0);
        // sched_barrier(0) pins the issue order: left alone, the scheduler sinks every load down
        // to its first use and the ring degenerates to load-wait-compute.
        auto block = [&](const f32x4 (&w)[NT], int kq) {
          const f32x4 an = lds_a((kq + 1) & (C::KQ_F - 1));
          __builtin_amdgcn_sched_barrier(0);
          mma_a(w, a);
          __builtin_amdgcn_sched_barrier(0);
          a = an;
        };
#pragma unroll 1
        for (int kq = 0; kq < C::KQ_F; kq += 4) {
          if constexpr (!RESW) {
            if (NLB > 0 && (kq & 4)) load_w_lds(w3, kq >> 3);   // k-quads 7, 15: LDS-resident
            else load_w(w3, kq + 3);
          }
          block(w0, kq);
          if constexpr (!RESW) load_w(w0, (kq + 4) & (C::KQ_F - 1));
          block(w1, kq + 1);
          if constexpr (!RESW) load_w(w1, (kq + 5) & (C::KQ_F - 1));
          block(w2, kq + 2);
          if constexpr (!RESW) load_w(w2, (kq + 6) & (C::KQ_F - 1));
          block(w3, kq + 3);
        }
      } else if constexpr (C::KQ_F == 1) {
        load_w(w0, 0);
        mma(w0, 0);
      } else {
        load_w(w0, 0);
#pragma unroll 1
        for (int kq = 0; kq < C::KQ_F; kq += 2) {
          load_w(w1, kq + 1);
          mma(w0, kq);
          if (kq + 2 < C::KQ_F) load_w(w0, kq + 2);
          mma(w1, kq + 1);
        }
      }
    }

    // cell update, lane-local: acc[jl*4+q][reg] is gate q of window row 4g+reg, hidden jt*16+r
    float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
#pragma unroll
    for (int jl = 0; jl < C::JPW; ++jl) {
      const int jt = wave * C::JPW + jl;
      f32x4 gi, gf, gg, go, hv;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        gi[reg] = fast_sigmoid(acc[jl * 4 + 0][reg]);
        gf[reg] = fast_sigmoid(acc[jl * 4 + 1][reg]);
        gg[reg] = fast_tanh(acc[jl * 4 + 2][reg]);
        go[reg] = fast_sigmoid(acc[jl * 4 + 3][reg]);
        c[jl][reg] = gf[reg] * c[jl][reg] + gi[reg] * gg[reg];
        hv[reg] = go[reg] * fast_tanh(c[jl][reg]);
        hnxt[hidx(4 * g + reg, jt * 16 + r)] = hv[reg];
      }
      st_stream(Gt + (long)(0 * C::JT + jt) * 256, gi);
      st_stream(Gt + (long)(1 * C::JT + jt) * 256, gf);
      st_stream(Gt + (long)(2 * C::JT + jt) * 256, gg);
      st_stream(Gt + (long)(3 * C::JT + jt) * 256, go);
      st_stream(cseq + ((long)(bt * T + t) * C::JT + jt) * 256 + lane * 4, c[jl]);
      if (clast != nullptr && t == T - 1) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int b = b0 + 4 * g + reg;
          if (b < B) clast[(long)b * H + jt * 16 + r] = c[jl][reg];
        }
      }
    }
    lds_barrier();
    // h_t tile -> hseq[b, t, :], whole rows, 16 B per lane
    for (int idx = tid; idx < 16 * (H / 4); idx += C::NW * 64) {
      const int row = idx / (H / 4), c4 = idx % (H / 4);
      const int b = b0 + row;
      if (b < B) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&hnxt[hidx(row, 4 * c4)]);
        st_stream(hseq + ((long)b * T + t) * H + 4 * c4, v);
        if (hlast != nullptr && t == T - 1) *reinterpret_cast<f32x4*>(hlast + (long)b * H + 4 * c4) = v;
      }
    }
  }
}

// ---- backward -----------------------------------------------------------------------------------
// G       : in  activated gates (tile16 [Bp*T, 4H]); out gate pre-activation gradients dA
// cseq    : in  cell states (tile16 [Bp*T, H])
// dhseq   : in  optional gradient w.r.t. every h_t from the layer above (tile16 [Bp*T, H])
// dhlast  : in  optional gradient w.r.t. h_T, row-major [B, H]
// dclast  : in  optional gradient w.r.t. c_T, row-major [B, H]
// dbpart  : out per-batch-tile column sums of dA: [Bp/16][4H]
//
// Same pipelining as the forward kernel: W_hh fragments one block (KB k-quads) ahead of the
// MFMAs, and the gates / cell states / upstream gradient of step t-1 in flight during the
// product of step t.
template <int H>
__global__ __launch_bounds__(RecCfg<H>::NW * 64) void lstm_rec_bwd_kernel(
    float* __restrict__ G, const float* __restrict__ cseq, const float* __restrict__ dhseq,
    const float* __restrict__ dhlast, const float* __restrict__ dclast, float* __restrict__ dbpart,
    const float* __restrict__ Wp, int B, int T) {
  using C = RecCfg<H>;
  constexpr int JPW = C::JPW;
  constexpr int KB = 4;                        // k-quads per register block
  constexpr int NBLK = C::KQ_B / KB;           // 4H/64 blocks: 1, 2, 4, 8, 16
  extern __shared__ __attribute__((aligned(16))) float dal[];  // [2][16 * LDA]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int bt = blockIdx.x;
  const int b0 = bt * 16;

  f32x4 dc[JPW], dh[JPW];
  float dbs[JPW][4];
#pragma unroll
  for (int jl = 0; jl < JPW; ++jl) {
    dc[jl] = f32x4{0.f, 0.f, 0.f, 0.f};
    dh[jl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) dbs[jl][q] = 0.f;
  }
  const f32x4* Wv = reinterpret_cast<const f32x4*>(Wp) + (long)wave * C::KQ_B * JPW * 64 + lane;
  auto load_w = [&](f32x4 (&w)[KB * JPW], int blk) {
#pragma unroll
    for (int i = 0; i < KB * JPW; ++i) w[i] = Wv[(blk * KB * JPW + i) * 64];   // i = kq_local*JPW + jl
  };
  // saved tensors of one step: gates [jl][4] and upstream dh [jl]; cell states separately
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

  // gates / dh of the step about to be processed, c_t and c_{t-1}: loaded one step ahead
  f32x4 gn[JPW * 4], dn[JPW], cvt[JPW], cvp[JPW];
  load_step(gn, dn, T - 1);
  load_cells(cvt, T - 1);
  load_cells(cvp, T - 2);

  // W_hh fragment ring, three blocks ahead, wrapping into the next step (see the forward kernel)
  constexpr bool RING = (NBLK % 4 == 0);
  f32x4 w0[KB * JPW], w1[KB * JPW], w2[KB * JPW], w3[KB * JPW];
  constexpr bool RESW = RING && NBLK == 4;     // hidden 64: the four buffers hold the whole share, loaded once
  if constexpr (RING) {
    load_w(w0, 0);
    load_w(w1, 1);
    load_w(w2, 2);
    if constexpr (RESW) load_w(w3, 3);
  }

  for (int t = T - 1; t >= 0; --t) {
    const int cur = t & 1;
    float* Gt = G + ((long)(bt * T + t) * (4 * C::JT)) * 256 + lane * 4;
    float* da_s = dal + cur * (16 * C::LDA);
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) {
      const int jt = wave * JPW + jl;
      const f32x4 gi = gn[jl * 4 + 0], gf = gn[jl * 4 + 1], gg = gn[jl * 4 + 2], go = gn[jl * 4 + 3];
      const f32x4 ct = cvt[jl], cp = cvp[jl];
      f32x4 dht = dh[jl] + dn[jl];
      if (t == T - 1) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int b = b0 + 4 * g + reg;
          if (b < B) {
            if (dhlast != nullptr) dht[reg] += dhlast[(long)b * H + jt * 16 + r];
            if (dclast != nullptr) dc[jl][reg] += dclast[(long)b * H + jt * 16 + r];
          }
        }
      }
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
        // k-slot order of the backward product: slot = 4*j + gate
        *reinterpret_cast<f32x4*>(&da_s[(4 * g + reg) * C::LDA + 4 * (jt * 16 + r)]) =
            f32x4{dai[reg], daf[reg], dag[reg], dao[reg]};
      }
      dbs[jl][0] += (dai[0] + dai[1]) + (dai[2] + dai[3]);
      dbs[jl][1] += (daf[0] + daf[1]) + (daf[2] + daf[3]);
      dbs[jl][2] += (dag[0] + dag[1]) + (dag[2] + dag[3]);
      dbs[jl][3] += (dao[0] + dao[1]) + (dao[2] + dao[3]);
      st_stream(Gt + (long)(0 * C::JT + jt) * 256, dai);
      st_stream(Gt + (long)(1 * C::JT + jt) * 256, daf);
      st_stream(Gt + (long)(2 * C::JT + jt) * 256, dag);
      st_stream(Gt + (long)(3 * C::JT + jt) * 256, dao);
    }
    if (t == 0) break;  // dh_{-1} is not needed
    // saved tensors of step t-1: in flight during the product below
    load_step(gn, dn, t - 1);
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) cvt[jl] = cvp[jl];
    load_cells(cvp, t - 2);
    lds_barrier();
    // dh_rec[16 x H-slice] = dA[16 x 4H] . W_hh[4H x H-slice]
    f32x4 acc[JPW];
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) acc[jl] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* arow = &da_s[r * C::LDA + 4 * g];
    auto mma = [&](const f32x4 (&w)[KB * JPW], int blk) {
#pragma unroll
      for (int kl = 0; kl < KB; ++kl) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow + 16 * (blk * KB + kl));
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int jl = 0; jl < JPW; ++jl)
            acc[jl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], w[kl * JPW + jl][s], acc[jl], 0, 0, 0);
      }
    };
    if constexpr (RING) {
#pragma unroll 1
      for (int blk = 0; blk < NBLK; blk += 4) {
        if constexpr (!RESW) load_w(w3, blk + 3);
        mma(w0, blk);
        if constexpr (!RESW) load_w(w0, (blk + 4) & (NBLK - 1));
        mma(w1, blk + 1);
        if constexpr (!RESW) load_w(w1, (blk + 5) & (NBLK - 1));
        mma(w2, blk + 2);
        if constexpr (!RESW) load_w(w2, (blk + 6) & (NBLK - 1));
        mma(w3, blk + 3);
      }
    } else if constexpr (NBLK == 1) {
      load_w(w0, 0);
      mma(w0, 0);
    } else {
      load_w(w0, 0);
#pragma unroll 1
      for (int blk = 0; blk < NBLK; blk += 2) {
        load_w(w1, blk + 1);
        mma(w0, blk);
        if (blk + 2 < NBLK) load_w(w0, blk + 2);
        mma(w1, blk + 1);
      }
    }
#pragma unroll
    for (int jl = 0; jl < JPW; ++jl) dh[jl] = acc[jl];
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

// out[n] (+)= sum_p part[p * pitch + n], and the same into out2 when given (the two bias gradients of an LSTM layer
// are the same sums).  One workgroup per 64 columns; its 16 waves split the
// partials (coalesced 256-byte rows, four loads in flight per wave) and combine through LDS in a
// fixed order (deterministic).
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ part,
                                                               float* __restrict__ out, float* __restrict__ out2,
                                                               int nparts, int N, long pitch, int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    const float* p0 = part + n;
    int p = wave;
    for (; p + 48 < nparts; p += 64) {
      s0 += p0[(long)p * pitch];
      s1 += p0[(long)(p + 16) * pitch];
      s2 += p0[(long)(p + 32) * pitch];
      s3 += p0[(long)(p + 48) * pitch];
    }
    for (; p < nparts; p += 16) s0 += p0[(long)p * pitch];
  }
  red[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && n < N) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w][lane];
    out[n] = accumulate ? out[n] + tot : tot;
    if (out2 != nullptr) out2[n] = accumulate ? out2[n] + tot : tot;
  }
}

template <int H>
int run_fwd(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp, int B,
            int T, hipStream_t stream) {
  using C = RecCfg<H>;
  hipLaunchKernelGGL(pack_whh_fwd_kernel<H>, dim3(vts_cdiv(4 * H * H, 256)), dim3(256), 0, stream, W_hh, Wp);
  VTS_CHECK_LAUNCH("pack_whh_fwd");
  const int nbt = (B + 15) / 16;
  const size_t lds = (2ul * 16 * H + (size_t)(C::KQ_F / 8) * C::JPW * 4 * C::NW * 64 * 4) * sizeof(float);
  static vts_lds_mask lds_done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec_fwd_kernel<H>), lds, &lds_done, "lstm_rec_fwd"))
    return rc;
  hipLaunchKernelGGL(lstm_rec_fwd_kernel<H>, dim3(nbt), dim3(C::NW * 64), lds, stream, G, cseq, hseq, hlast, clast,
                     Wp, B, T);
  VTS_CHECK_LAUNCH("lstm_rec_fwd");
  return VTS_OK;
}

template <int H>
int run_bwd(float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
            float* dbpart, const float* W_hh, float* Wp, int B, int T, hipStream_t stream) {
  using C = RecCfg<H>;
  hipLaunchKernelGGL(pack_whh_bwd_kernel<H>, dim3(vts_cdiv(4 * H * H, 256)), dim3(256), 0, stream, W_hh, Wp);
  VTS_CHECK_LAUNCH("pack_whh_bwd");
  const int nbt = (B + 15) / 16;
  const size_t lds = 2ul * 16 * C::LDA * sizeof(float);
  static vts_lds_mask lds_done{0};
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(lstm_rec_bwd_kernel<H>), lds, &lds_done, "lstm_rec_bwd"))
    return rc;
  hipLaunchKernelGGL(lstm_rec_bwd_kernel<H>, dim3(nbt), dim3(C::NW * 64), lds, stream, G, cseq, dhseq, dhlast,
                     dclast, dbpart, Wp, B, T);
  VTS_CHECK_LAUNCH("lstm_rec_bwd");
  return VTS_OK;
}

}  // namespace

bool vts_lstm_hidden_supported(int H) { return H == 16 || H == 32 || H == 64 || H == 128 || H == 256; }

int vts_lstm_rec_fwd(float* G, float* cseq, float* hseq, float* hlast, float* clast, const float* W_hh, float* Wp,
                     int B, int T, int H, hipStream_t stream) {
  switch (H) {
    case 16: return run_fwd<16>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, stream);
    case 32: return run_fwd<32>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, stream);
    case 64: return run_fwd<64>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, stream);
    case 128: return run_fwd<128>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, stream);
    case 256: return run_fwd<256>(G, cseq, hseq, hlast, clast, W_hh, Wp, B, T, stream);
  }
  vts_set_error("lstm: hidden size %d not supported (16, 32, 64, 128, 256)", H);
  return VTS_ERR_UNSUPPORTED;
}

int vts_lstm_rec_bwd(float* G, const float* cseq, const float* dhseq, const float* dhlast, const float* dclast,
                     float* dbpart, const float* W_hh, float* Wp, int B, int T, int H, hipStream_t stream) {
  switch (H) {
    case 16: return run_bwd<16>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, stream);
    case 32: return run_bwd<32>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, stream);
    case 64: return run_bwd<64>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, stream);
    case 128: return run_bwd<128>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, stream);
    case 256: return run_bwd<256>(G, cseq, dhseq, dhlast, dclast, dbpart, W_hh, Wp, B, T, stream);
  }
  vts_set_error("lstm: hidden size %d not supported (16, 32, 64, 128, 256)", H);
  return VTS_ERR_UNSUPPORTED;
}

int vts_colsum_partials2(const float* part, float* out, float* out2, int nparts, int N, long pitch, int accumulate,
                         hipStream_t stream) {
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(vts_cdiv(N, 64)), dim3(1024), 0, stream, part, out, out2, nparts, N,
                     pitch, accumulate);
  VTS_CHECK_LAUNCH("colsum_partials");
  return VTS_OK;
}

int vts_colsum_partials(const float* part, float* out, int nparts, int N, long pitch, int accumulate,
                        hipStream_t stream) {
  return vts_colsum_partials2(part, out, nullptr, nparts, N, pitch, accumulate, stream);
}
