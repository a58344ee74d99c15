// The LSTM layer's three big products over bf16 plane operands (gemm_planes.h): input projection, data gradient
// and the paired weight gradients, plus the conversions that put an fp32 input sequence and the weights into
// plane form.  The recurrent kernels write the h sequence and the gate gradients as planes themselves, so none of
// these products converts or transposes anything: tiles go from HBM to LDS by DMA and from LDS to the matrix cores.
//
// Column order of the gate gradients.  The backward recurrence keeps dA in LDS with the four gates of a hidden
// unit side by side (k' = 4 * j + gate for gate-major column n = gate * H + j) and writes its planes in that
// order.  The data-gradient product contracts over k', so its weight operand is packed in the same order; the
// weight-gradient product has k' as its output row and un-permutes when it reduces the split-K slabs.
#include "gemm_planes.h"

namespace {

using gp::SeqPlanes;
using gp::TILE_B;

__device__ __forceinline__ int gate_major(int kp, int H) { return (kp & 3) * H + (kp >> 2); }

// ---- conversions ------------------------------------------------------------------------------------------
// One thread per 16-byte slot pair (hi and lo): row r of the tile, k-group g (8 consecutive features).
__global__ __launch_bounds__(256) void rows_to_planes_kernel(const float* __restrict__ x, SeqPlanes out, int B, int T,
                                                             int C, long nslots) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (long)gridDim.x * blockDim.x) {
    const int l = i & 63;
    long rest = i >> 6;
    const int cb = rest % out.ncb; rest /= out.ncb;
    const int t = rest % T;
    const long bt = rest / T;
    const int g = l & 3, r = l >> 2;          // 4 consecutive threads read 128 contiguous bytes of one row
    const long b = bt * 16 + r;
    const int c0 = cb * 32 + g * 8;
    float v[8];
    const float* px = x + (b * T + t) * C + c0;
    if (b < B && c0 + 8 <= C && (reinterpret_cast<uintptr_t>(px) & 7) == 0) {     // the usual case: four 8-byte loads
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float2 w2 = *reinterpret_cast<const float2*>(px + 2 * e);
        v[2 * e] = w2.x;
        v[2 * e + 1] = w2.y;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (b < B && c0 + e < C) ? px[e] : 0.f;
    }
    gp::u32x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h_, l_;
      split2(v[2 * q], v[2 * q + 1], h_, l_);
      hi[q] = h_; lo[q] = l_;
    }
    char* tile = out.base + out.tile(bt, t, cb, 0);
    *reinterpret_cast<gp::u32x4*>(tile + gp::slot(g, r) * 16) = hi;
    *reinterpret_cast<gp::u32x4*>(tile + TILE_B + gp::slot(g, r) * 16) = lo;
  }
}

// Weight operand planes [row tile][cb][p].  kind 0: rows = gate column n (gate-major), k = input feature c
// (input projection: W[n][c]); kind 1: rows = input feature c, k = k' (data gradient: W[gate_major(k')][c]).
__global__ __launch_bounds__(256) void weights_to_planes_kernel(const float* __restrict__ W, int N4H, int C, int H,
                                                                int kind, char* __restrict__ out, int nrt, int ncb) {
  const long nslots = (long)nrt * ncb * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (long)gridDim.x * blockDim.x) {
    const int l = i & 63;
    long rest = i >> 6;
    const int cb = rest % ncb;
    const int rt = rest / ncb;
    const int g = l >> 4, r = l & 15;
    const int row = rt * 16 + r, k0 = cb * 32 + g * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      if (kind == 0) v[e] = (row < N4H && k < C) ? W[(long)row * C + k] : 0.f;
      else           v[e] = (row < C && k < N4H) ? W[(long)gate_major(k, H) * C + row] : 0.f;
    }
    gp::u32x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h_, l_;
      split2(v[2 * q], v[2 * q + 1], h_, l_);
      hi[q] = h_; lo[q] = l_;
    }
    char* tile = out + ((long)(rt * ncb + cb) * 2) * TILE_B;
    *reinterpret_cast<gp::u32x4*>(tile + gp::slot(g, r) * 16) = hi;
    *reinterpret_cast<gp::u32x4*>(tile + TILE_B + gp::slot(g, r) * 16) = lo;
  }
}

__global__ void zero_slot0_kernel(SeqPlanes p, long nbt) {   // the zero "step before the first" of an h sequence
  const long n16 = (long)p.ncb * 2 * (TILE_B / 16);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nbt * n16; i += (long)gridDim.x * blockDim.x) {
    const long bt = i / n16, o = i % n16;
    *reinterpret_cast<gp::u32x4*>(p.base + (bt * p.ts) * p.ncb * 2 * TILE_B + o * 16) = gp::u32x4{0u, 0u, 0u, 0u};
  }
}

// ---- rows x weights: input projection and data gradient ------------------------------------------------------
// out[tile16 of [tokens, N]] = A[tokens, K] . Wt[N, K]^T (+ bias0 + bias1).  A workgroup owns a sequence of output
// tiles (256 tokens x 256 columns; tile ids start, start + step, ...: id = token block * ntn + column tile) and
// streams their k-steps as one sequence through the ring: the loads of the next tile's first k-steps are in flight
// while a tile's accumulators are written out.  Two ways to hand out the tiles (rowgemm_planes_kernel): a run of
// consecutive ids (a token block's column tiles one after the other in one workgroup), or - several column tiles -
// the column tiles of a token block side by side on ntn workgroups of one XCD, which walk the same token blocks in
// step so that a block's A tiles come from HBM once and from that XCD's L2 the other ntn - 1 times.
struct RowPol {
  SeqPlanes A;
  const char* W; int w_ncb;
  int T, nk, ntn, N;
  float* out; const float* bias0; const float* bias1;
  int wave, lane;
  int is_tile, is_ks, fl_tile, step;
  const char *a0, *a1, *b0, *b1;
  __device__ __forceinline__ void advance() {
    if (++is_ks == nk) {
      is_ks = 0;
      is_tile += step;
      const int tb = is_tile / ntn, tn = is_tile - tb * ntn;
      const int idx = tb * 16 + 2 * wave;             // first of this wave's two token tiles (T is even: same window tile)
      const int bt = idx / T, t = idx - bt * T;
      a0 = A.base + A.tile(bt, t, 0, 0);
      a1 = a0 + A.ncb * 2 * TILE_B;
      b0 = W + (long)(tn * 16 + 2 * wave) * w_ncb * 2 * TILE_B;
      b1 = b0 + w_ncb * 2 * TILE_B;
    }
  }
  // li = 4 * wave + q: row tile 2 * wave + (q >> 1), plane q & 1
  __device__ __forceinline__ const char* a_tile(int li) const { return ((li & 2) ? a1 : a0) + (is_ks * 2 + (li & 1)) * TILE_B; }
  __device__ __forceinline__ const char* b_tile(int li) const { return ((li & 2) ? b1 : b0) + (is_ks * 2 + (li & 1)) * TILE_B; }
  __device__ __forceinline__ void flush(f32x4 (&acc)[4][8]) {
    const int tile = fl_tile;
    fl_tile += step;
    const int tb = tile / ntn, tn = tile - tb * ntn;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 15;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const int ct = tn * 16 + wn * 8 + nt;
      // (The bias loads make the compiler wait for every older memory operation once per column tile, so the 32
      // stores of a flush leave four at a time.  Measured: with the biases fetched through the scalar cache and the
      // stores issued back to back - with or without letting them stay in flight through the next k-step - the
      // input projection is 4 % slower: the burst holds up the DMA loads of the next tile.)
      float bias = 0.f;
      if (bias0 != nullptr) bias = bias0[ct * 16 + r] + bias1[ct * 16 + r];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const long tt = (long)tb * 16 + wm * 4 + mt;
        f32x4 v = acc[mt][nt];
        v += bias;
        // streaming store: the output is read again a kernel later at the earliest, and allocated in the L2 it
        // would push out the operand tiles the column tiles of a token block share (input projection 0.72 -> 0.66 ms)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (tt * (N / 16) + ct) * 256 + lane * 4));
      }
    }
  }
};

__global__ __launch_bounds__(512) void rowgemm_planes_kernel(SeqPlanes A, const char* W, int w_ncb, int T, int nk,
                                                             int ntn, int N, float* out, const float* bias0,
                                                             const float* bias1, int ntiles_all, int colsplit) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int start, step, n;
  if (colsplit) {
    // workgroup b runs on XCD b % 8: the ntn workgroups j = ntn * g' + tn of an XCD form a group that shares its
    // token blocks, one column tile each
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, gpx = (int)(gridDim.x >> 3) / ntn;
    if (j >= gpx * ntn) return;
    const int g = xcd * gpx + j / ntn, ng = 8 * gpx, ntb = ntiles_all / ntn;
    const long lo = (long)g * ntb / ng, hi = (long)(g + 1) * ntb / ng;
    start = (int)lo * ntn + j % ntn; step = ntn; n = (int)(hi - lo);
  } else {
    // a run of consecutive tiles per workgroup
    const long lo = (long)blockIdx.x * ntiles_all / gridDim.x, hi = (long)(blockIdx.x + 1) * ntiles_all / gridDim.x;
    start = (int)lo; step = 1; n = (int)(hi - lo);
  }
  if (n <= 0) return;
  RowPol pol;
  pol.A = A; pol.W = W; pol.w_ncb = w_ncb; pol.T = T; pol.nk = nk; pol.ntn = ntn; pol.N = N;
  pol.out = out; pol.bias0 = bias0; pol.bias1 = bias1;
  pol.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  pol.lane = threadIdx.x & 63;
  pol.step = step; pol.is_tile = start - step; pol.is_ks = nk - 1; pol.fl_tile = start;
  gp::mainloop<RowPol, false>(pol, n * nk, nk, lds);
}

// ---- weight gradients ---------------------------------------------------------------------------------------
// slab[z][m'][n] = sum over the k-steps of slice z of dG[token][m'] * B[token][n], B = [layer input | h one step
// back].  Both operands are read with the transposed fragment read; a k-step is two token tiles.  B's column
// blocks (32 columns each) are cut into full column tiles of eight and, when four or fewer remain, one half tile
// (NTW = 4: half the B loads and MFMAs per k-step, so its workgroups take longer k-slices).
template <int NTW>
struct WgradPol {
  SeqPlanes dG, X, Hs;          // Hs.t0 = 0: slot t holds h_{t-1}
  int T;
  int ncb_x, ncb_h;
  const char* zero;
  float* slab; int ldn;         // this workgroup's slab tile origin and row pitch
  int tm, cb0, wave, lane;      // cb0: first B column block of this workgroup
  long bt; int t;               // token tiles (bt, t), (bt, t + 1) of the k-step being issued
  const char *a, *b;
  __device__ __forceinline__ void advance() {
    t += 2;
    if (t >= T) { t = 0; ++bt; }
    const int tt = wave >> 2;                               // this wave's token tile of the pair
    a = dG.base + dG.tile(bt, t + tt, tm * 8 + 2 * (wave & 3), 0);
    const int cbg = cb0 + (NTW == 8 ? 2 * (wave & 3) : (wave & 3));
    if (cbg < ncb_x) b = X.base + X.tile(bt, t + tt, cbg, 0);
    else if (cbg - ncb_x < ncb_h) b = Hs.base + Hs.tile(bt, t + tt, cbg - ncb_x, 0);
    else b = nullptr;
  }
  // A: li = 4 * wave + q: column blocks 2 (wave & 3) + (q >> 1), plane q & 1: four consecutive KiB.
  // B: NTW 8 the same; NTW 4: li = 2 * wave + q: column block wave & 3, plane q.
  __device__ __forceinline__ const char* a_tile(int li) const { return a + (li & 3) * TILE_B; }
  __device__ __forceinline__ const char* b_tile(int li) const {
    return b != nullptr ? b + (li & (NTW == 8 ? 3 : 1)) * TILE_B : zero;
  }
  __device__ __forceinline__ void flush(f32x4 (&acc)[4][NTW]) {
    const int wm = wave >> 1, wn = wave & 1, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = tm * 256 + (wm * 4 + mt) * 16 + 4 * g + reg;
          const int col = (wn * NTW + nt) * 16 + r;
          slab[(long)row * ldn + col] = acc[mt][nt][reg];
        }
  }
};

// Plan of a launch: per XCD `fa` k-slices of the full column tiles (each slice = tiles_m x nfull workgroups) and
// `ha` k-slices of the half tile (tiles_m workgroups each); slices of one XCD's share sit on that XCD together with
// all their tiles, so an operand tile is fetched from HBM once and served to the slice's other workgroups by the L2.
struct WgradPlan {
  int M, nfull, has_half, fa, ha, nsteps;
  __host__ __device__ int ncols_full() const { return nfull * 256; }
  int grid() const { return 8 * (fa * (M / 256) * nfull + ha * (M / 256) * has_half); }
  size_t slab_floats() const { return (size_t)8 * fa * M * ncols_full() + (size_t)8 * ha * has_half * M * 128; }
};

template <int NTW>
__device__ __forceinline__ void wgrad_run(SeqPlanes dG, SeqPlanes X, SeqPlanes Hs, int T, int ncb_x, int ncb_h,
                                          const char* zero, float* slab, int ldn, int tm, int cb0, int s0, int n, char* lds) {
  WgradPol<NTW> pol;
  pol.dG = dG; pol.X = X; pol.Hs = Hs; pol.T = T; pol.ncb_x = ncb_x; pol.ncb_h = ncb_h; pol.zero = zero;
  pol.slab = slab; pol.ldn = ldn; pol.tm = tm; pol.cb0 = cb0;
  pol.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  pol.lane = threadIdx.x & 63;
  const long idx = 2L * s0 - 2;                // advance() steps to 2 * s0 first
  pol.bt = idx >= 0 ? idx / T : -1;
  pol.t = idx >= 0 ? (int)(idx - pol.bt * T) : T - 2;
  if (n <= 0) {                                // an empty slice still owns a slab: zero it
    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < NTW; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    pol.flush(acc);
    return;
  }
  gp::mainloop<WgradPol<NTW>, true, NTW>(pol, n, n, lds);
}

__global__ __launch_bounds__(512) void wgrad_planes_kernel(SeqPlanes dG, SeqPlanes X, SeqPlanes Hs, int T, int ncb_x,
                                                           int ncb_h, const char* zero, float* slabs, WgradPlan P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tiles_m = P.M / gp::TM, per = tiles_m * P.nfull;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  if (j < P.fa * per) {
    const int ksplit = 8 * P.fa, z = (j / per) * 8 + xcd, tile = j % per;
    const int chunk = (P.nsteps + ksplit - 1) / ksplit, s0 = z * chunk;
    const int tm = tile / P.nfull, tn = tile % P.nfull, ldn = P.ncols_full();
    wgrad_run<8>(dG, X, Hs, T, ncb_x, ncb_h, zero, slabs + (size_t)z * P.M * ldn + tn * 256, ldn, tm, tn * 8, s0,
                 min(chunk, P.nsteps - s0), lds);
  } else {
    const int jj = j - P.fa * per;
    const int ksplit = 8 * P.ha, z = (jj / tiles_m) * 8 + xcd, tm = jj % tiles_m;
    const int chunk = (P.nsteps + ksplit - 1) / ksplit, s0 = z * chunk;
    float* half = slabs + (size_t)8 * P.fa * P.M * P.ncols_full() + (size_t)z * P.M * 128;
    wgrad_run<4>(dG, X, Hs, T, ncb_x, ncb_h, zero, half, 128, tm, P.nfull * 8, s0, min(chunk, P.nsteps - s0), lds);
  }
}

// dw_ih[gate_major(m')][c] (+)= sum_z slab[z][m'][c], c < F;  dw_hh[gate_major(m')][j] (+)= sum_z slab[z][m'][Fp + j]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, WgradPlan P, int F, int Fp,
                                                           int H, float* __restrict__ dw_ih, float* __restrict__ dw_hh,
                                                           int accumulate) {
  const long total = (long)P.M * (Fp + H);
  const int nfc = P.ncols_full();
  const float* halves = slabs + (size_t)8 * P.fa * P.M * nfc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = i / (Fp + H), n = i % (Fp + H);
    if (n >= F && n < Fp) continue;
    const bool full = n < nfc;
    const float* p = full ? slabs + (long)m * nfc + n : halves + (long)m * 128 + (n - nfc);
    const long pitch = full ? (long)P.M * nfc : (long)P.M * 128;
    const int ns = 8 * (full ? P.fa : P.ha);
    float s0 = 0.f, s1 = 0.f;
    int z = 0;
    for (; z + 1 < ns; z += 2) {
      s0 += p[z * pitch];
      s1 += p[(z + 1) * pitch];
    }
    if (z < ns) s0 += p[z * pitch];
    const float s = s0 + s1;
    const int row = gate_major(m, H);
    float* dst = n < F ? dw_ih + (long)row * F + n : dw_hh + (long)row * H + (n - Fp);
    *dst = accumulate ? *dst + s : s;
  }
}

vts_lds_mask g_lds_row{0}, g_lds_wgrad{0};

int grid_cus() {
  // development switch: the persistent grids sized for a share of the device (two half batches side by side)
  if (const char* e = getenv("VTS_GRID_CUS")) {
    const int want = atoi(e);
    if (want >= 8 && want <= 1024 && want % 8 == 0) return want;
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    cus = 256;
  return cus > 0 ? cus : 256;
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------------------------
// Shapes the plane path is built for: hidden 256 (gate columns fill whole 256-row tiles), whole clusters of 128
// windows, an even number of steps (a weight-gradient k-step is two token tiles of one window tile).
bool vts_planes_usable(long B, long T, long H) { return H == 256 && B > 0 && (B % 128) == 0 && T >= 2 && (T % 2) == 0; }
int vts_planes_fpad(long F) { return (int)((F + 63) / 64 * 64); }
size_t vts_planes_seq_bytes(long B, long T, long F, int with_zero_slot) {
  const long nbt = (B + 15) / 16;
  return (size_t)nbt * (T + (with_zero_slot ? 1 : 0)) * (vts_planes_fpad(F) / 32) * 2 * TILE_B;
}
size_t vts_planes_weight_bytes(long rows, long K) {
  return (size_t)((rows + 15) / 16) * (vts_planes_fpad(K) / 32) * 2 * TILE_B;
}
static SeqPlanes seq_planes(void* base, long T, long F, int with_zero_slot) {
  return SeqPlanes{reinterpret_cast<char*>(base), vts_planes_fpad(F) / 32, (int)T + (with_zero_slot ? 1 : 0),
                   with_zero_slot ? 1 : 0};
}

int vts_planes_from_rows(const float* x, void* planes, long B, long T, long C, hipStream_t st) {
  SeqPlanes P = seq_planes(planes, T, C, 0);
  const long nbt = (B + 15) / 16, nslots = nbt * T * P.ncb * 64;
  hipLaunchKernelGGL(rows_to_planes_kernel, dim3((unsigned)((nslots + 255) / 256 > 65536 * 4 ? 65536 * 4 : (nslots + 255) / 256)),
                     dim3(256), 0, st, x, P, (int)B, (int)T, (int)C, nslots);
  VTS_CHECK_LAUNCH("rows_to_planes");
  return VTS_OK;
}

int vts_planes_zero_slot0(void* planes, long B, long T, long F, hipStream_t st) {
  SeqPlanes P = seq_planes(planes, T, F, 1);
  const long nbt = (B + 15) / 16;
  hipLaunchKernelGGL(zero_slot0_kernel, dim3((unsigned)((nbt * P.ncb * 128 + 255) / 256)), dim3(256), 0, st, P, nbt);
  VTS_CHECK_LAUNCH("zero_slot0");
  return VTS_OK;
}

// gates[tile16 of [B*T, 4H]] = S . W_ih^T + b_ih + b_hh; S planes with (a_zero_slot ? the h-sequence layout : T slots)
int vts_planes_inproj(const void* s_planes, int a_zero_slot, const float* w_ih, const float* b_ih, const float* b_hh,
                      float* gates, void* wplanes, long B, long T, long Cin, long H, hipStream_t st) {
  const int N = (int)(4 * H), Fp = vts_planes_fpad(Cin), ncb = Fp / 32;
  hipLaunchKernelGGL(weights_to_planes_kernel, dim3((N / 16) * ncb / 4), dim3(256), 0, st, w_ih, N, (int)Cin, (int)H, 0,
                     reinterpret_cast<char*>(wplanes), N / 16, ncb);
  VTS_CHECK_LAUNCH("weights_to_planes");
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(rowgemm_planes_kernel), gp::LDS_B, &g_lds_row, "rowgemm_planes")) return rc;
  SeqPlanes A = seq_planes(const_cast<void*>(s_planes), T, Cin, a_zero_slot);
  const int ntb = (int)(B / 16 * T / 16), ntn = N / 256;
  const int grid = min(grid_cus(), ntb);
  // column tiles side by side when every XCD gets whole groups with at least one token block each
  static const int want_split = getenv("VTS_ROW_COLSPLIT") ? atoi(getenv("VTS_ROW_COLSPLIT")) : 1;
  const int colsplit = want_split && grid % (8 * ntn) == 0 && ntb >= grid / ntn;
  hipLaunchKernelGGL(rowgemm_planes_kernel, dim3(grid), dim3(512), gp::LDS_B, st, A, reinterpret_cast<const char*>(wplanes),
                     ncb, (int)T, ncb, ntn, N, gates, b_ih, b_hh, ntb * ntn, colsplit);
  VTS_CHECK_LAUNCH("rowgemm_planes(inproj)");
  return VTS_OK;
}

// dx[tile16 of [B*T, Cin]] = dG . W_ih  (dG planes in k' order, Cin a multiple of 256)
int vts_planes_dgrad(const void* dg_planes, const float* w_ih, float* dx_t16, void* wplanes, long B, long T, long Cin,
                     long H, hipStream_t st) {
  const int N4H = (int)(4 * H), ncb = N4H / 32;
  hipLaunchKernelGGL(weights_to_planes_kernel, dim3((int)(Cin / 16) * ncb / 4), dim3(256), 0, st, w_ih, N4H, (int)Cin, (int)H, 1,
                     reinterpret_cast<char*>(wplanes), (int)(Cin / 16), ncb);
  VTS_CHECK_LAUNCH("weights_to_planes");
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(rowgemm_planes_kernel), gp::LDS_B, &g_lds_row, "rowgemm_planes")) return rc;
  SeqPlanes A = seq_planes(const_cast<void*>(dg_planes), T, N4H, 0);
  const int ntb = (int)(B / 16 * T / 16), ntn = (int)(Cin / 256);
  const int grid = min(grid_cus(), ntb);
  hipLaunchKernelGGL(rowgemm_planes_kernel, dim3(grid), dim3(512), gp::LDS_B, st, A, reinterpret_cast<const char*>(wplanes),
                     ncb, (int)T, ncb, ntn, (int)Cin, dx_t16, nullptr, nullptr, ntb * ntn, 0);
  VTS_CHECK_LAUNCH("rowgemm_planes(dgrad)");
  return VTS_OK;
}

// Split-K plan of the paired weight gradients.  Column blocks of [Fp | H]: full tiles of eight, and a half tile
// when at most four remain.  Per XCD (a share of cus / 8 CUs) fa slices of the full tiles and ha of the half tile are
// chosen so that every workgroup is resident at once and both kinds finish together (a half-tile k-step is taken as
// 0.6 of a full one: half the matrix work, three quarters of the loads).
static WgradPlan wgrad_plan(long B, long T, long F, long H) {
  WgradPlan P{};
  P.M = (int)(4 * H);
  const int ncb = vts_planes_fpad(F) / 32 + (int)(H / 32);
  P.nfull = ncb / 8;
  const int rem = ncb % 8;
  P.has_half = 0;
  if (rem > 4) P.nfull += 1;
  else if (rem > 0) P.has_half = 1;
  P.nsteps = (int)(B / 16 * T / 2);
  const int tiles_m = P.M / 256, share = grid_cus() / 8;
  double best = 1e30;
  P.fa = 1; P.ha = P.has_half;
  for (int fa = 1; fa * tiles_m * P.nfull <= share; ++fa)
    for (int ha = P.has_half; ha <= (P.has_half ? share : 0); ++ha) {
      if (fa * tiles_m * P.nfull + ha * tiles_m * P.has_half > share) break;
      if (P.nsteps / (8 * fa) < 4) continue;
      const double tf = (double)P.nsteps / (8 * fa), th = P.has_half ? 0.6 * P.nsteps / (8 * ha) : 0.0;
      const double t = tf > th ? tf : th;
      if (t < best) { best = t; P.fa = fa; P.ha = ha; }
    }
  return P;
}
size_t vts_planes_wgrad_slab_bytes(long B, long T, long F, long H) {
  return wgrad_plan(B, T, F, H).slab_floats() * sizeof(float);
}

// dw_ih[4H, F] (+)= dG^T . S,  dw_hh[4H, H] (+)= dG^T . h(one step back): one launch + the slab reduction.
// x_zero_slot: S is an h sequence (slot 0 = zeros) rather than a converted input.
int vts_planes_wgrad(const void* dg_planes, const void* s_planes, int x_zero_slot, const void* h_planes, float* dw_ih,
                     float* dw_hh, int accumulate, float* slabs, long B, long T, long F, long H, hipStream_t st) {
  const WgradPlan P = wgrad_plan(B, T, F, H);
  const int Fp = vts_planes_fpad(F);
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(wgrad_planes_kernel), gp::LDS_B, &g_lds_wgrad, "wgrad_planes")) return rc;
  SeqPlanes dG = seq_planes(const_cast<void*>(dg_planes), T, P.M, 0);
  SeqPlanes X = seq_planes(const_cast<void*>(s_planes), T, F, x_zero_slot);
  SeqPlanes Hs = seq_planes(const_cast<void*>(h_planes), T, H, 1);
  Hs.t0 = 0;                                        // slot t = h_{t-1}; slot 0 = the zero initial state
  const char* zero = Hs.base;                       // (bt 0, slot 0, cb 0): 2 KiB of zeros
  hipLaunchKernelGGL(wgrad_planes_kernel, dim3(P.grid()), dim3(512), gp::LDS_B, st, dG, X, Hs, (int)T, Fp / 32,
                     (int)(H / 32), zero, slabs, P);
  VTS_CHECK_LAUNCH("wgrad_planes");
  const long total = (long)P.M * (Fp + H);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slabs, P, (int)F, Fp,
                     (int)H, dw_ih, dw_hh, accumulate);
  VTS_CHECK_LAUNCH("wgrad_reduce");
  return VTS_OK;
}
