// Fused head and decoder chains: up to three Linear + activation layers ("trunk") followed by one or more Linear
// heads that all read the last trunk output, in one launch forward and two launches backward.
//   encoder heads  h -> relu(fc1) -> dropout(relu(fc2)) -> {mean, logvar, exp(scale), clamp(shape)}
//                  (gpd-lstm-DE.py:29-33, mixture-vae-DE.py:62-69; gaussian-lstm-DE.py:30-31 has no trunk)
//   decoders       z -> relu(fc1) -> dropout(relu(fc2)) -> out        (gpd-lstm-DE.py:45-47, mixture-vae-DE.py:81-83)
//                  z -> relu(dense1) -> out                            (gaussian-lstm-DE.py:43-45)
// These products are tiny (a few hundred MFLOP at 4096 windows): run layer by layer they are 9 + 9 GEMM launches
// and some 30 small ones for activations and bias sums, each waiting on memory latency.  Here a workgroup owns 16
// rows and walks the whole chain with the activations in LDS; the weights (under 0.5 MB for a model) come from the
// L2.  Arithmetic: v_mfma_f32_16x16x4_f32, plain fp32 in every mode.
//
// Fragment convention of the 16x16x4 fp32 MFMA: lane l supplies A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15] and receives C[i = 4 * (l >> 4) + reg][j = l & 15].  A lane fetches four consecutive
// k of its row at a time (one float4) and feeds them to four MFMAs, so the k index a lane covers in a group of 16
// is 4 * (l >> 4) + e - the same (permuted) order for both operands.
#include "vts_common.h"
#include <stdlib.h>

namespace {

constexpr int TM = 16;            // rows per workgroup
constexpr int MAXT = 3;           // trunk layers
constexpr int MAXH = 8;           // heads
constexpr int MAXW = 512;         // widest activation kept in LDS
constexpr int LDA = MAXW + 4;     // LDS row stride (floats)
constexpr size_t CHAIN_LDS = 2ul * TM * LDA * sizeof(float);   // two activation buffers

struct Chain {
  int n_trunk, n_heads, in_dim, head_dim;
  int dim[MAXT], act[MAXT];
  int hact[MAXH];
  float keep_scale;
  const float* w[MAXT + MAXH];    // trunk layers first, then heads; [N][K] row-major
  const float* b[MAXT + MAXH];
  const float* keep[MAXT];        // [B][dim] 0/1 masks of the VTS_ACT_RELU_DROPOUT layers (null: no dropout)
  float* tout[MAXT];              // post-activation trunk outputs [B][dim] (saved for backward)
  float* hout[MAXH];              // head outputs [B][head_dim]
  float* haux[MAXH];              // VTS_ACT_CLAMP_MIN heads: 1 where the gradient passes
};

struct ChainBwd {
  const float* dhead[MAXH];       // gradients w.r.t. the head outputs (null: zero)
  float* dpre_t[MAXT];            // gradients w.r.t. the trunk pre-activations [B][dim]
  float* dpre_h;                  // ... and the heads', side by side [B][n_heads * head_dim]
  float* dx;                      // [B][in_dim] row stride lddx (null: not needed)
  long lddx;
  float* dw[MAXT + MAXH];
  float* db[MAXT + MAXH];
  int tile0[MAXT + MAXH + 1];     // weight-gradient kernel: first 16x16 output tile of every matrix
};

__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
  return c;
}

// ---- forward -------------------------------------------------------------------------------------------------
// NW waves per workgroup.  A wave walks its column tiles one after the other and each costs a memory latency (the
// weights), so a batch of few row tiles - few workgroups - runs with more waves than four (WIDE_NW below).
template <int NW>
__global__ __launch_bounds__(NW * 64) void mlp_chain_fwd_kernel(const Chain D, const float* __restrict__ x, long ldx, int B) {
  extern __shared__ __attribute__((aligned(16))) float chain_lds[];
  float (*sa)[TM][LDA] = reinterpret_cast<float (*)[TM][LDA]>(chain_lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * TM;
  {
    const int c4n = D.in_dim >> 2;
    for (int idx = tid; idx < TM * c4n; idx += NW * 64) {
      const int rr = idx / c4n, c4 = idx - rr * c4n;
      const long row = m0 + rr;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < B) v = *reinterpret_cast<const f32x4*>(x + row * ldx + c4 * 4);
      *reinterpret_cast<f32x4*>(&sa[0][rr][c4 * 4]) = v;
    }
  }
  __syncthreads();
  int cur = 0, K = D.in_dim;
  for (int li = 0; li <= D.n_trunk; ++li) {
    const bool heads = li == D.n_trunk;
    const int Nw = heads ? D.head_dim : D.dim[li];           // width of one matrix of this stage
    const int tpm = (Nw + 15) >> 4;                          // column tiles per matrix
    const int ntiles = heads ? D.n_heads * tpm : tpm;
    for (int tile = wave; tile < ntiles; tile += NW) {
      const int mi = heads ? D.n_trunk + tile / tpm : li;    // matrix index
      const int n0 = (heads ? tile % tpm : tile) << 4;
      const int n = n0 + r;
      const bool nvalid = n < Nw;
      const float* wrow = D.w[mi] + (long)(nvalid ? n : Nw - 1) * K + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < K; kc += 128) {                    // eight groups of 16 k: all weight loads first
        f32x4 av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k0 = kc + 16 * u;
          av[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          bv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (k0 + 4 * q < K) {                                // K is a multiple of 4: a float4 is all in or all out
            bv[u] = *reinterpret_cast<const f32x4*>(wrow + k0);
            av[u] = *reinterpret_cast<const f32x4*>(&sa[cur][r][k0 + 4 * q]);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = mfma4(av[u], bv[u], acc);
      }
      const float bias = D.b[mi] != nullptr ? D.b[mi][nvalid ? n : Nw - 1] : 0.f;
      const int act = heads ? D.hact[mi - D.n_trunk] : D.act[li];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = 4 * q + reg;
        const long row = m0 + m;
        const bool live = nvalid && row < B;
        float v = acc[reg] + bias;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = expf(v);
        else if (act == 3) {
          if (live && D.haux[mi - D.n_trunk] != nullptr) D.haux[mi - D.n_trunk][row * Nw + n] = v >= 1e-6f ? 1.f : 0.f;
          v = fmaxf(v, 1e-6f);
        } else if (act == 4) {
          v = fmaxf(v, 0.f);
          if (!heads && D.keep[li] != nullptr) v *= live ? D.keep[li][row * Nw + n] * D.keep_scale : 0.f;
        }
        if (heads) {
          if (live) D.hout[mi - D.n_trunk][row * Nw + n] = v;
        } else {
          if (live) D.tout[li][row * Nw + n] = v;
          if (nvalid) sa[cur ^ 1][m][n] = row < B ? v : 0.f;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    K = Nw;
  }
}

// ---- backward 1: gradients w.r.t. the pre-activations and the chain input ---------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void mlp_chain_bwd_dx_kernel(const Chain D, const ChainBwd R, int B) {
  extern __shared__ __attribute__((aligned(16))) float chain_lds[];
  float (*sd)[TM][LDA] = reinterpret_cast<float (*)[TM][LDA]>(chain_lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.x * TM;
  const int L = D.head_dim, NH = D.n_heads * L;
  const int NHp = (NH + 15) & ~15;
  // head stage: d pre = d out * act'
  for (int idx = tid; idx < TM * NHp; idx += NW * 64) {
    const int m = idx / NHp, c = idx - m * NHp;
    const long row = m0 + m;
    float v = 0.f;
    if (c < NH && row < B) {
      const int h = c / L, j = c - h * L;
      const float d = R.dhead[h] != nullptr ? R.dhead[h][row * L + j] : 0.f;
      const int act = D.hact[h];
      if (act == 0) v = d;
      else if (act == 1) v = D.hout[h][row * L + j] > 0.f ? d : 0.f;
      else if (act == 2) v = d * D.hout[h][row * L + j];
      else if (act == 3) v = d * D.haux[h][row * L + j];
      else v = D.hout[h][row * L + j] != 0.f ? d * D.keep_scale : 0.f;
      R.dpre_h[row * NH + c] = v;
    }
    sd[0][m][c] = v;
  }
  __syncthreads();
  int cur = 0, Nin = NH;
  for (int li = D.n_trunk - 1; li >= -1; --li) {
    // out[m][k] = sum_n din[m][n] * W[n][k]; the incoming stage is the heads (first pass) or trunk layer li + 1
    const bool from_heads = li == D.n_trunk - 1;
    const int K = li >= 0 ? D.dim[li] : D.in_dim;
    if (li >= 0 || R.dx != nullptr) {
      const int ktiles = (K + 15) >> 4;
      for (int tile = wave; tile < ktiles; tile += NW) {
        const int k = (tile << 4) + r;
        const int kc = k < K ? k : K - 1;
        // this tile's saved activations (for the activation derivative) are fetched with the weights, not after
        f32x4 aprev = {0.f, 0.f, 0.f, 0.f};
        if (li >= 0 && k < K) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const long row = m0 + 4 * q + reg;
            if (row < B) aprev[reg] = D.tout[li][row * K + k];
          }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // eight groups of 16 n at a time: their 32 weight loads (4 bytes per lane, rows K floats apart) are all
        // issued before the first product, so a tile costs a memory latency per eight groups, not per group
        for (int nc = 0; nc < Nin; nc += 128) {
          f32x4 av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            av[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            bv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int nb = nc + 16 * u + 4 * q;
            // the 16 n of a group belong to one matrix (several heads: their width is a multiple of 16), so the
            // pointer is picked with a wave-uniform index - a per-lane index into the argument struct would send
            // the struct to scratch memory
            const int hh = (from_heads && D.n_heads > 1) ? (nc + 16 * u) / L : 0;
            const float* Wb = (from_heads ? D.w[D.n_trunk + hh] : D.w[li + 1]) + kc;
            if (nb < Nin) {
              av[u] = *reinterpret_cast<const f32x4*>(&sd[cur][r][nb]);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int n = nb + e;
                if (n < Nin) bv[u][e] = Wb[(long)(n - hh * L) * K];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = mfma4(av[u], bv[u], acc);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int m = 4 * q + reg;
          const long row = m0 + m;
          const bool live = k < K && row < B;
          float v = acc[reg];
          if (li >= 0) {
            const float a = aprev[reg];
            const int act = D.act[li];
            if (act == 1) v = a > 0.f ? v : 0.f;
            else if (act == 4) v = a != 0.f ? v * (D.keep[li] != nullptr ? D.keep_scale : 1.f) : 0.f;
            if (!live) v = 0.f;
            if (live) R.dpre_t[li][row * K + k] = v;
            if (k < K) sd[cur ^ 1][m][k] = v;
          } else if (live) {
            R.dx[row * R.lddx + k] = v;
          }
        }
      }
      // columns of the next stage's input beyond K, up to the next multiple of 16, must read as zero
      if (li >= 0) {
        const int Kp = (K + 15) & ~15;
        for (int idx = tid; idx < TM * (Kp - K); idx += NW * 64) sd[cur ^ 1][idx / (Kp - K)][K + idx % (Kp - K)] = 0.f;
      }
    }
    __syncthreads();
    cur ^= 1;
    Nin = K;
  }
}

#ifndef VTS_CHAIN_NW
#define VTS_CHAIN_NW 8
#endif
constexpr int WIDE_NW = VTS_CHAIN_NW;
// WIDE_NW waves per workgroup up to this many rows (VTS_CHAIN_WIDE_ROWS overrides it, for measurements).
bool chain_wide(long B) {
  static const long lim = [] { const char* e = getenv("VTS_CHAIN_WIDE_ROWS"); return e ? atol(e) : 8192L; }();
  return B <= lim;
}

// ---- backward 2: weight and bias gradients of every matrix of the chain -----------------------------------------
// One workgroup per 16 x 16 tile of some dW = dpre^T . input, contracted over all rows: its four waves take every
// fourth group of 16 rows (four groups in flight each) and their partial tiles are added up through LDS in a fixed
// order.
// Batches beyond DW_ROWS rows: blockIdx.y takes a chunk of the rows and the tiles go to `part` as
// [chunk][tile][256 accumulator values in lane order, 16 bias sums] for mlp_chain_dw_reduce_kernel.
constexpr int DW_ROWS = 8192;
constexpr int DW_PART = 272;
struct TileAt { int mi, N, K, n0, k0; };
__device__ __forceinline__ TileAt tile_at(const Chain& D, const ChainBwd& R, int tile) {
  TileAt t;
  t.mi = 0;
  while (tile >= R.tile0[t.mi + 1]) ++t.mi;
  const bool head = t.mi >= D.n_trunk;
  t.N = head ? D.head_dim : D.dim[t.mi];
  const int srcl = head ? D.n_trunk - 1 : t.mi - 1;
  t.K = srcl >= 0 ? D.dim[srcl] : D.in_dim;
  const int ktiles = (t.K + 15) >> 4;
  const int i = tile - R.tile0[t.mi];
  t.n0 = (i / ktiles) << 4;
  t.k0 = (i % ktiles) << 4;
  return t;
}

__global__ __launch_bounds__(256) void mlp_chain_bwd_dw_kernel(const Chain D, const ChainBwd R, const float* __restrict__ x,
                                                               long ldx, int B, float* __restrict__ part) {
  __shared__ float red[3][64][5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x;
  int mi = 0;
  while (tile >= R.tile0[mi + 1]) ++mi;
  const bool head = mi >= D.n_trunk;
  const int N = head ? D.head_dim : D.dim[mi];
  const int srcl = head ? D.n_trunk - 1 : mi - 1;             // trunk layer that feeds this matrix (-1: the chain input)
  const int K = srcl >= 0 ? D.dim[srcl] : D.in_dim;
  const float* src = srcl >= 0 ? D.tout[srcl] : x;
  const long lds_ = srcl >= 0 ? K : ldx;
  const float* dp = head ? R.dpre_h + (mi - D.n_trunk) * D.head_dim : R.dpre_t[mi];
  const long ldd = head ? (long)D.n_heads * D.head_dim : N;
  const int ktiles = (K + 15) >> 4;
  const int t = tile - R.tile0[mi];
  const int n0 = (t / ktiles) << 4, k0 = (t % ktiles) << 4;
  const int n = n0 + r, k = k0 + r;
  const float* dpc = dp + (n < N ? n : N - 1);
  const float* srcc = src + (k < K ? k : K - 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int row_beg = blockIdx.y * DW_ROWS;
  const int row_end = part != nullptr ? min(B, row_beg + DW_ROWS) : B;
  auto fetch = [&](int row0, f32x4& av, f32x4& bv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long row = row0 + 4 * q + e;
      const bool in = row < row_end;
      av[e] = in ? dpc[row * ldd] : 0.f;
      bv[e] = in ? srcc[row * lds_] : 0.f;
    }
  };
  for (int row0 = row_beg + 16 * wave; row0 < row_end; row0 += 256) {
    f32x4 av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(row0 + 64 * u, av[u], bv[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bsum += (av[u][0] + av[u][1]) + (av[u][2] + av[u][3]);
      acc = mfma4(av[u], bv[u], acc);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave - 1][lane][i] = acc[i];
    red[wave - 1][lane][4] = bsum;
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += red[w][lane][i];
    bsum += red[w][lane][4];
  }
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  if (part != nullptr) {
    float* p = part + ((long)blockIdx.y * gridDim.x + tile) * DW_PART;
    *reinterpret_cast<f32x4*>(p + lane * 4) = acc;
    if (q == 0) p[256 + r] = bsum;
    return;
  }
  float* dw = R.dw[mi];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int nn = n0 + 4 * q + reg;
    if (nn < N && k < K) dw[(long)nn * K + k] = acc[reg];
  }
  if (k0 == 0 && R.db[mi] != nullptr && q == 0 && n < N) R.db[mi][n] = bsum;
}

// chunks of one tile added up in a fixed order and written where the one-chunk kernel writes them
__global__ __launch_bounds__(320) void mlp_chain_dw_reduce_kernel(const Chain D, const ChainBwd R, const float* __restrict__ part,
                                                                  int ntiles, int nchunks) {
  const int tile = blockIdx.x, e = threadIdx.x;
  if (e >= DW_PART) return;
  const TileAt t = tile_at(D, R, tile);
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += part[((long)c * ntiles + tile) * DW_PART + e];
  if (e < 256) {
    const int lane = e >> 2, reg = e & 3;
    const int nn = t.n0 + 4 * (lane >> 4) + reg, k = t.k0 + (lane & 15);
    if (nn < t.N && k < t.K) R.dw[t.mi][(long)nn * t.K + k] = s;
  } else {
    const int n = t.n0 + (e - 256);
    if (t.k0 == 0 && R.db[t.mi] != nullptr && n < t.N) R.db[t.mi][n] = s;
  }
}

}  // namespace

// Description of a chain as the C ABI passes it (include/vts.h: vts_mlp_chain).
struct vts_mlp_chain_desc {
  int32_t n_trunk, n_heads, in_dim, head_dim;
  int32_t trunk_dim[MAXT], trunk_act[MAXT];
  int32_t head_act[MAXH];
  float keep_scale;
};

static int chain_check(const vts_mlp_chain_desc* d, long B) {
  VTS_REQUIRE(d != nullptr && B >= 1 && B < (1l << 31) - 16, "mlp chain: bad batch");
  VTS_REQUIRE(d->n_trunk >= 0 && d->n_trunk <= MAXT && d->n_heads >= 1 && d->n_heads <= MAXH, "mlp chain: 0..%d trunk layers, 1..%d heads", MAXT, MAXH);
  VTS_REQUIRE(d->in_dim >= 4 && d->in_dim <= MAXW && d->in_dim % 4 == 0, "mlp chain: input width %d (a multiple of 4, at most %d)", d->in_dim, MAXW);
  VTS_REQUIRE(d->head_dim >= 1 && d->n_heads * d->head_dim <= MAXW, "mlp chain: heads too wide");
  VTS_REQUIRE(d->n_heads == 1 || d->head_dim % 16 == 0, "mlp chain: several heads need a head width that is a multiple of 16");
  for (int i = 0; i < d->n_trunk; ++i) {
    VTS_REQUIRE(d->trunk_dim[i] >= 4 && d->trunk_dim[i] <= MAXW && d->trunk_dim[i] % 4 == 0, "mlp chain: trunk width %d (a multiple of 4, at most %d)", d->trunk_dim[i], MAXW);
    VTS_REQUIRE(d->trunk_act[i] == 0 || d->trunk_act[i] == 1 || d->trunk_act[i] == 4, "mlp chain: trunk activation %d", d->trunk_act[i]);
  }
  for (int i = 0; i < d->n_heads; ++i) VTS_REQUIRE(d->head_act[i] >= 0 && d->head_act[i] <= 3, "mlp chain: head activation %d", d->head_act[i]);
  return VTS_OK;
}

// 16 x 16 output tiles of all weight gradients of a chain
static int chain_tiles(const vts_mlp_chain_desc* d) {
  int tiles = 0;
  for (int i = 0; i < d->n_trunk + d->n_heads; ++i) {
    const int N = i < d->n_trunk ? d->trunk_dim[i] : d->head_dim;
    const int K = i == 0 ? d->in_dim : (i <= d->n_trunk ? d->trunk_dim[i - 1] : (d->n_trunk ? d->trunk_dim[d->n_trunk - 1] : d->in_dim));
    tiles += ((N + 15) / 16) * ((K + 15) / 16);
  }
  return tiles;
}

static void chain_fill(Chain& D, const vts_mlp_chain_desc* d) {
  D = Chain{};
  D.n_trunk = d->n_trunk; D.n_heads = d->n_heads; D.in_dim = d->in_dim; D.head_dim = d->head_dim;
  for (int i = 0; i < MAXT; ++i) { D.dim[i] = d->trunk_dim[i]; D.act[i] = d->trunk_act[i]; }
  for (int i = 0; i < MAXH; ++i) D.hact[i] = d->head_act[i];
  D.keep_scale = d->keep_scale;
}

int vts_mlp_chain_supported_impl(const vts_mlp_chain_desc* d, long B) { return chain_check(d, B) == VTS_OK; }

int vts_mlp_chain_fwd_impl(const vts_mlp_chain_desc* d, const float* x, long ldx, long B, const float* const* w,
                           const float* const* b, const float* const* keep, float* const* trunk_out,
                           float* const* head_out, float* const* head_aux, hipStream_t st) {
  if (int rc = chain_check(d, B)) return rc;
  VTS_REQUIRE(x != nullptr && vts_aligned16(x) && ldx % 4 == 0 && ldx >= d->in_dim, "mlp chain: input must be 16-byte aligned with a row stride that is a multiple of 4");
  Chain D;
  chain_fill(D, d);
  const int nm = d->n_trunk + d->n_heads;
  for (int i = 0; i < nm; ++i) {
    VTS_REQUIRE(w[i] != nullptr && vts_aligned16(w[i]), "mlp chain: weight %d missing or not 16-byte aligned", i);
    D.w[i] = w[i];
    D.b[i] = b != nullptr ? b[i] : nullptr;
  }
  for (int i = 0; i < d->n_trunk; ++i) {
    VTS_REQUIRE(trunk_out != nullptr && trunk_out[i] != nullptr, "mlp chain: trunk output %d missing", i);
    D.tout[i] = trunk_out[i];
    D.keep[i] = (keep != nullptr && d->trunk_act[i] == 4) ? keep[i] : nullptr;
  }
  for (int i = 0; i < d->n_heads; ++i) {
    VTS_REQUIRE(head_out != nullptr && head_out[i] != nullptr, "mlp chain: head output %d missing", i);
    D.hout[i] = head_out[i];
    D.haux[i] = head_aux != nullptr ? head_aux[i] : nullptr;
    VTS_REQUIRE(d->head_act[i] != 3 || D.haux[i] != nullptr, "mlp chain: head %d clamps and needs its aux buffer", i);
  }
  static vts_lds_mask lds_done{0};
  if (chain_wide(B)) {
    static vts_lds_mask lds_done_w{0};
    if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(mlp_chain_fwd_kernel<WIDE_NW>), CHAIN_LDS, &lds_done_w, "mlp_chain_fwd")) return rc;
    hipLaunchKernelGGL(mlp_chain_fwd_kernel<WIDE_NW>, dim3((unsigned)((B + TM - 1) / TM)), dim3(WIDE_NW * 64), CHAIN_LDS, st, D, x, ldx, (int)B);
    VTS_CHECK_LAUNCH("mlp_chain_fwd");
    return VTS_OK;
  }
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(mlp_chain_fwd_kernel<4>), CHAIN_LDS, &lds_done, "mlp_chain_fwd")) return rc;
  hipLaunchKernelGGL(mlp_chain_fwd_kernel<4>, dim3((unsigned)((B + TM - 1) / TM)), dim3(256), CHAIN_LDS, st, D, x, ldx, (int)B);
  VTS_CHECK_LAUNCH("mlp_chain_fwd");
  return VTS_OK;
}

size_t vts_mlp_chain_bwd_workspace_bytes_impl(const vts_mlp_chain_desc* d, long B) {
  if (chain_check(d, B) != VTS_OK) return 0;
  size_t n = (size_t)d->n_heads * d->head_dim;
  for (int i = 0; i < d->n_trunk; ++i) n += (size_t)d->trunk_dim[i];
  size_t bytes = n * (size_t)B * sizeof(float) + 512;
  const long nchunks = (B + DW_ROWS - 1) / DW_ROWS;
  if (nchunks > 1) bytes += (size_t)nchunks * chain_tiles(d) * DW_PART * sizeof(float);
  return bytes;
}

int vts_mlp_chain_bwd_impl(const vts_mlp_chain_desc* d, const float* x, long ldx, long B, const float* const* w,
                           const float* const* keep, float* const* trunk_out, float* const* head_out,
                           float* const* head_aux, const float* const* d_head, float* dx, long lddx, float* const* dw,
                           float* const* db, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (int rc = chain_check(d, B)) return rc;
  VTS_REQUIRE(workspace != nullptr && workspace_bytes >= vts_mlp_chain_bwd_workspace_bytes_impl(d, B), "mlp chain: workspace too small");
  VTS_REQUIRE(x != nullptr && w != nullptr && dw != nullptr && d_head != nullptr && head_out != nullptr &&
                  (d->n_trunk == 0 || trunk_out != nullptr),
              "mlp chain: missing argument");
  VTS_REQUIRE(vts_aligned16(x) && ldx % 4 == 0 && ldx >= d->in_dim, "mlp chain: input must be 16-byte aligned with a row stride that is a multiple of 4");
  VTS_REQUIRE(dx == nullptr || lddx >= d->in_dim, "mlp chain: dx row stride shorter than the input width");
  Chain D;
  chain_fill(D, d);
  ChainBwd R{};
  const int nm = d->n_trunk + d->n_heads;
  float* ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
  int tiles = 0;
  for (int i = 0; i < nm; ++i) {
    VTS_REQUIRE(w[i] != nullptr && vts_aligned16(w[i]) && dw[i] != nullptr, "mlp chain: weight %d missing or not 16-byte aligned, or its gradient missing", i);
    D.w[i] = w[i];
    R.dw[i] = dw[i];
    R.db[i] = db != nullptr ? db[i] : nullptr;
    const int N = i < d->n_trunk ? d->trunk_dim[i] : d->head_dim;
    const int K = i == 0 ? d->in_dim : (i <= d->n_trunk ? d->trunk_dim[i - 1] : (d->n_trunk ? d->trunk_dim[d->n_trunk - 1] : d->in_dim));
    R.tile0[i] = tiles;
    tiles += ((N + 15) / 16) * ((K + 15) / 16);
  }
  R.tile0[nm] = tiles;
  for (int i = 0; i < d->n_trunk; ++i) {
    VTS_REQUIRE(trunk_out[i] != nullptr, "mlp chain: trunk output %d missing", i);
    D.tout[i] = trunk_out[i];
    D.keep[i] = (keep != nullptr && d->trunk_act[i] == 4) ? keep[i] : nullptr;
    R.dpre_t[i] = ws;
    ws += (size_t)B * d->trunk_dim[i];
  }
  R.dpre_h = ws;
  for (int i = 0; i < d->n_heads; ++i) {
    VTS_REQUIRE(head_out[i] != nullptr, "mlp chain: head output %d missing", i);
    D.hout[i] = head_out[i];
    D.haux[i] = head_aux != nullptr ? head_aux[i] : nullptr;
    VTS_REQUIRE(d->head_act[i] != 3 || D.haux[i] != nullptr, "mlp chain: head %d clamps and needs its aux buffer", i);
    R.dhead[i] = d_head[i];              // (a head without an upstream gradient passes NULL)
  }
  R.dx = dx;
  R.lddx = lddx;
  static vts_lds_mask lds_done{0};
  if (chain_wide(B)) {
    static vts_lds_mask lds_done_w{0};
    if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(mlp_chain_bwd_dx_kernel<WIDE_NW>), CHAIN_LDS, &lds_done_w, "mlp_chain_bwd_dx")) return rc;
    hipLaunchKernelGGL(mlp_chain_bwd_dx_kernel<WIDE_NW>, dim3((unsigned)((B + TM - 1) / TM)), dim3(WIDE_NW * 64), CHAIN_LDS, st, D, R, (int)B);
  } else {
    if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(mlp_chain_bwd_dx_kernel<4>), CHAIN_LDS, &lds_done, "mlp_chain_bwd_dx")) return rc;
    hipLaunchKernelGGL(mlp_chain_bwd_dx_kernel<4>, dim3((unsigned)((B + TM - 1) / TM)), dim3(256), CHAIN_LDS, st, D, R, (int)B);
  }
  VTS_CHECK_LAUNCH("mlp_chain_bwd_dx");
  const int nchunks = (int)((B + DW_ROWS - 1) / DW_ROWS);
  float* part = nullptr;
  if (nchunks > 1) part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws + (size_t)B * d->n_heads * d->head_dim) + 255) & ~uintptr_t(255));
  hipLaunchKernelGGL(mlp_chain_bwd_dw_kernel, dim3((unsigned)tiles, (unsigned)nchunks), dim3(256), 0, st, D, R, x, ldx, (int)B, part);
  VTS_CHECK_LAUNCH("mlp_chain_bwd_dw");
  if (nchunks > 1) {
    hipLaunchKernelGGL(mlp_chain_dw_reduce_kernel, dim3((unsigned)tiles), dim3(320), 0, st, D, R, part, tiles, nchunks);
    VTS_CHECK_LAUNCH("mlp_chain_dw_reduce");
  }
  return VTS_OK;
}
