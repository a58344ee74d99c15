// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 fmaf chain, one
// rounding per product) with LDS-staged 128x128x16 tiles, 4 waves per workgroup, each wave a
// 64x64 sub-tile = 4x4 MFMA tiles (64 accumulator registers).
//
// One kernel template serves every dense contraction of the train step; what changes is how the
// logical operands A[m][k], B[k][n], C[m][n] map to memory:
//   A_ROW  A[m][k] = a[row(m)*lda + k]        (k contiguous; optional window row map)
//   A_COL  A[m][k] = a[k*lda + m]             (m contiguous)
//   A_T16  A[m][k] = tile16(a)[m][k]          (tile16 layout of an [M, K] matrix)
//   A_T16T A[m][k] = tile16(a)[k][m]          (tile16 layout of a [K, M] matrix, transposed use)
//   B_NK   B[k][n] = b[n*ldb + k]             (k contiguous: a Linear weight [N][K])
//   B_KN   B[k][n] = b[row(k)*ldb + n]        (n contiguous; optional window row map on k)
//   C_ROW  row-major with bias/activation epilogue, or split-K slabs
//   C_T16  tile16 layout (+ bias)
//
// Window row map (LSTM sequences stored batch-first [B, T, F]): logical row
// m' = (bt*T + t)*16 + bi  <->  window b = bt*16 + bi at step t (+ tshift), memory row b*T + t.
// Rows with b >= B or a shifted step outside [0, T) read as zero.
#include "vts_common.h"
#include <type_traits>

namespace {

enum { A_ROW = 0, A_COL = 1, A_T16 = 2, A_T16T = 3 };
enum { B_NK = 0, B_KN = 1 };
enum { C_ROW = 0, C_T16 = 1 };

#ifndef VTS_GEMM_WPS
#define VTS_GEMM_WPS 3
#endif
#ifndef VTS_GEMM_NSUB
#define VTS_GEMM_NSUB 1
#endif
constexpr int BM = 128, BN = 128, BKT = 16;
constexpr int NSUB = VTS_GEMM_NSUB;          // 16-deep k sub-chunks staged per barrier
constexpr int BKS = BKT * NSUB;              // k elements per main-loop iteration
constexpr size_t GEMM_LDS_BYTES = 2ul * NSUB * 2 * BKT * 144 * sizeof(float);
static_assert(GEMM_LDS_BYTES >= 4 * 16 * 68 * sizeof(float), "row-major epilogue stages 4 x [16][68] floats in LDS");
constexpr int LDS_LD = 144;  // 144 % 32 == 16: the (k, k+1) x 16-column b32 fragment reads hit 32 distinct banks

struct GemmParams {
  const float* A;
  const float* B;
  float* C;          // final output, or slab base when ksplit > 1
  int M, N, K;
  long lda, ldb, ldc;
  // window row maps
  int a_map, b_map;  // 0 = identity, 1 = window map
  int T, Bwin;       // steps per window, number of real windows
  int a_tshift, b_tshift;
  // epilogue
  const float* bias0;  // [N] or null
  const float* bias1;  // [N] or null (second bias, added too)
  int act;             // 0 none, 1 relu, 2 exp, 3 clamp_min(1e-6), 4 relu * keep * scale
  const float* keep;   // [M][N] 0/1 keep mask for act 4 (row stride ldc)
  float keep_scale;
  unsigned long long drop_seed;   // act 4 without a keep mask: counter-hash dropout over index row*N + col
  unsigned drop_thresh;           // (0 = none: plain relu)
  const float* mask_src;  // [M][N] (row stride ld_mask) or null: result = mask_src != 0 ? result * mask_scale : 0,
  long ld_mask;           // i.e. the gradient of a relu(+dropout) whose OUTPUT is mask_src, applied to a data gradient
  float mask_scale;
  const unsigned long long* mask_bits;   // the same mask as packed bits written by bits_out (N % 64 == 0)
  unsigned long long* bits_out;          // or null: [M][N/64] words, bit set where the finished output is non-zero
  float* colpart;         // or null: [2 * ceil(M/128)][N] column sums of the finished output, one row per
                          // (128-row tile, upper/lower 64 rows): the bias gradient of whatever produced the
                          // operand this output is the gradient of, without another pass over it
  float* aux;          // act 3: [M][N] 1.0 where pre-activation >= 1e-6 (gradient passes)
  int accumulate;      // C += result (ksplit == 1 only)
  int ksplit;          // >1: grid.z slabs of M*N floats at C
  int kchunk;          // K elements per split (multiple of 32)
  int mode;            // 0 = f32 MFMA, 1 = bf16x3 split on the bf16 MFMA
  int big_split;       // split-K plan made for the 256 x 128 tile kernel
  // Two products over the same A in one launch of the 256 x 128 kernel (both weight gradients of an LSTM layer):
  // column tiles below n_cut (a multiple of 128) are product 1 = (B, ldb, b_tshift, N1 columns, slabs at C),
  // the others product 2 = (B2, ldb2, b_tshift2, N2 columns, slabs at C2); N = n_cut + N2.  n_cut 0 = off.
  const float* B2;
  float* C2;
  long ldb2;
  int b_tshift2, N1, N2, n_cut;
};

__device__ __forceinline__ long window_row(int m, int T, int Bwin, int tshift, bool& valid) {
  const int bi = m & 15;
  const int q = m >> 4;
  const int t = q % T + tshift;
  const int b = (q / T) * 16 + bi;
  valid = (b < Bwin) && (t >= 0) && (t < T);
  return (long)b * T + t;
}

// Load 8 consecutive floats p[0..7] with validity count nvalid (elements >= nvalid read as 0).
// Full loads are two 16-byte accesses that only need 4-byte alignment (rows of 370 floats start
// on 8-byte boundaries): one instruction stream for every lane, no alignment branches.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void load8(const float* __restrict__ p, int nvalid, float (&v)[8]) {
  if (nvalid >= 8) {
    const f32x4_u x = *reinterpret_cast<const f32x4_u*>(p);
    const f32x4_u y = *reinterpret_cast<const f32x4_u*>(p + 4);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3]; v[4] = y[0]; v[5] = y[1]; v[6] = y[2]; v[7] = y[3];
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (i < nvalid) ? p[i] : 0.0f;
}

// A_ROW: the memory row of this thread's logical row does not depend on k: resolved once.
__device__ __forceinline__ const float* a_row_base(const GemmParams& P, int m0, int tid) {
  const int m = m0 + (tid >> 1);
  bool valid = m < P.M;
  long row = m;
  if (P.a_map && valid) row = window_row(m, P.T, P.Bwin, P.a_tshift, valid);
  return valid ? P.A + row * P.lda : nullptr;
}

template <int AK>
__device__ __forceinline__ void load_a(const GemmParams& P, const float* arow, int m0, int k0, int kend, int tid,
                                       float (&v)[8]) {
  if constexpr (AK == A_ROW) {
    const int k = k0 + (tid & 1) * 8;
    const int nv = (arow != nullptr) ? min(8, kend - k) : 0;
    if (nv > 0) load8(arow + k, nv, v);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
  } else if constexpr (AK == A_COL) {
    const int k = k0 + (tid >> 4);
    const int m = m0 + (tid & 15) * 8;
    const int nv = (k < kend) ? min(8, P.M - m) : 0;
    if (nv > 0) load8(P.A + (long)k * P.lda + m, nv, v);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
  } else if constexpr (AK == A_T16) {
    // tile16 of an [M, K] matrix; this thread takes two float4 (rows 4g..4g+3 of column c) of the
    // block (m-tile m0/16 + blk, k-tile k0/16).
    const int blk = tid >> 5;
    const int mt = (m0 >> 4) + blk;
    const bool valid = (mt * 16 < P.M) && (k0 < kend);
    if (valid) {
      const float* p = P.A + ((long)mt * (P.K >> 4) + (k0 >> 4)) * 256 + (tid & 31) * 8;
      load8(p, 8, v);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
  } else {  // A_T16T: memory is tile16 of a [K, M] matrix
    const int blk = tid >> 5;
    const int mt = (m0 >> 4) + blk;  // column tile of the stored matrix
    const bool valid = (mt * 16 < P.M) && (k0 < kend);
    if (valid) {
      const float* p = P.A + ((long)(k0 >> 4) * (P.M >> 4) + mt) * 256 + (tid & 31) * 8;
      load8(p, 8, v);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
  }
}

template <int AK>
__device__ __forceinline__ void store_a(float* __restrict__ As, int tid, const float (&v)[8]) {
  if constexpr (AK == A_ROW) {
    const int m = tid >> 1;
    const int kb = (tid & 1) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) As[(kb + s) * LDS_LD + m] = v[s];
  } else if constexpr (AK == A_COL) {
    const int k = tid >> 4;
    const int m = (tid & 15) * 8;
    *reinterpret_cast<float4*>(&As[k * LDS_LD + m]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&As[k * LDS_LD + m + 4]) = make_float4(v[4], v[5], v[6], v[7]);
  } else if constexpr (AK == A_T16) {
    // float4 j (j = 0,1) is lane l = (tid&31)*2 + j of the block: rows 4g..4g+3 (g = l>>4) of
    // column c = l&15; the block's column index is k here, its row index is m.
    const int blk = tid >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = (tid & 31) * 2 + j;
      const int g = l >> 4, c = l & 15;
      *reinterpret_cast<float4*>(&As[c * LDS_LD + blk * 16 + 4 * g]) =
          make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
  } else {  // A_T16T: block row index is k, column index is m
    const int blk = tid >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int l = (tid & 31) * 2 + j;
      const int g = l >> 4, c = l & 15;
#pragma unroll
      for (int s = 0; s < 4; ++s) As[(4 * g + s) * LDS_LD + blk * 16 + c] = v[4 * j + s];
    }
  }
}

__device__ __forceinline__ const float* b_row_base(const GemmParams& P, int n0, int tid) {
  const int n = n0 + (tid >> 1);
  return (n < P.N) ? P.B + (long)n * P.ldb : nullptr;
}

template <int BKIND>
__device__ __forceinline__ void load_b(const GemmParams& P, const float* brow, int n0, int k0, int kend, int tid,
                                       float (&v)[8]) {
  if constexpr (BKIND == B_NK) {
    const int k = k0 + (tid & 1) * 8;
    const int nv = (brow != nullptr) ? min(8, kend - k) : 0;
    if (nv > 0) load8(brow + k, nv, v);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
  } else {
    const int k = k0 + (tid >> 4);
    const int n = n0 + (tid & 15) * 8;
    bool valid = k < kend;
    long row = k;
    if (P.b_map && valid) row = window_row(k, P.T, P.Bwin, P.b_tshift, valid);
    const int nv = valid ? min(8, P.N - n) : 0;
    if (nv > 0) load8(P.B + row * P.ldb + n, nv, v);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
  }
}

template <int BKIND>
__device__ __forceinline__ void store_b(float* __restrict__ Bs, int tid, const float (&v)[8]) {
  if constexpr (BKIND == B_NK) {
    const int n = tid >> 1;
    const int kb = (tid & 1) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) Bs[(kb + s) * LDS_LD + n] = v[s];
  } else {
    const int k = tid >> 4;
    const int n = (tid & 15) * 8;
    *reinterpret_cast<float4*>(&Bs[k * LDS_LD + n]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&Bs[k * LDS_LD + n + 4]) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// Workgroup -> (m tile, n tile, k slice).  Workgroups whose ids are equal mod 8 share an XCD
// (and its L2): all tiles that read the same operand panel - the n tiles of one m tile, or
// every tile of one k slice - are given ids equal mod 8 and adjacent in time, so the panel is
// fetched into one L2 once instead of into all eight.  Returns false for padding workgroups.
template <int TBM = 128>
__device__ __forceinline__ bool map_workgroup(const GemmParams& P, int& m0, int& n0, int& zslice, int& kbeg,
                                              int& kend) {
  const int tiles_n = (P.N + BN - 1) / BN;
  const int tiles_m = (P.M + TBM - 1) / TBM;
  int tile;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  if (P.ksplit > 1) {
    const int per = tiles_m * tiles_n;
    zslice = (j / per) * 8 + xcd;
    tile = j % per;
    if (zslice >= P.ksplit) return false;
  } else {
    const int mt = (j / tiles_n) * 8 + xcd;
    if (mt >= tiles_m) return false;
    tile = mt * tiles_n + j % tiles_n;
    zslice = 0;
  }
  m0 = (tile / tiles_n) * TBM;
  n0 = (tile % tiles_n) * BN;
  kbeg = zslice * P.kchunk;
  kend = min(P.K, kbeg + P.kchunk);
  return true;
}

// Accumulator tiles -> memory.  acc[i][j] is the 16x16 tile (i, j) of this wave's 64x64 block in the
// v_mfma 16x16 C/D layout (lane l: column l&15, rows 4*(l>>4)..+3), the same for every dtype.
// PLAIN: the launch has no bias, activation or mask (weight gradients, split-K slabs): a bare store
// (or accumulate) straight from the accumulator layout, which keeps these kernels' register budget
// for their operand streams.
template <int CK, bool PLAIN>
__device__ __forceinline__ void epilogue(const GemmParams& P, f32x4 (&acc)[4][4], int m0, int n0, int zslice,
                                         int wm, int wn, int lane, float* stage) {
  const int r = lane & 15, g = lane >> 4;
  if constexpr (CK == C_ROW && PLAIN) {
    float* Cbase = P.C + (P.ksplit > 1 ? (long)zslice * P.M * P.N : 0);
    const long ldc = (P.ksplit > 1) ? P.N : P.ldc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + r;
        if (col >= P.N) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = m0 + wm * 64 + i * 16 + 4 * g + reg;
          if (row >= P.M) continue;
          float v = acc[i][j][reg];
          if (P.ksplit == 1 && P.accumulate) v += Cbase[(long)row * ldc + col];
          Cbase[(long)row * ldc + col] = v;
        }
      }
    }
  } else if constexpr (CK == C_T16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt = (m0 + wm * 64 + i * 16) >> 4;
      if (mt * 16 >= P.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nbase = n0 + wn * 64 + j * 16;
        if (nbase >= P.N) continue;
        f32x4 v = acc[i][j];
        float bsum = 0.f;
        if (P.bias0) bsum += P.bias0[nbase + r];
        if (P.bias1) bsum += P.bias1[nbase + r];
        v += bsum;
        float* dst = P.C + ((long)mt * (P.N >> 4) + (nbase >> 4)) * 256 + lane * 4;
        if (P.accumulate) {
          const f32x4 old = *reinterpret_cast<const f32x4*>(dst);
          v += old;
        }
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  } else {
    // Row-major output.  In the accumulator layout a store instruction would touch 16 consecutive
    // floats of four different rows; instead each wave passes one 16-row block at a time through a
    // private LDS patch ([16][64] floats, row pitch 68) and writes it back out as float4 per lane:
    // 16 lanes cover 256 contiguous bytes of a row, and bias, activation, dropout and the masks are
    // applied on that side with vector loads.  The k loop ended with a workgroup barrier, so the
    // operand tiles in LDS are dead.
    // (split-K launches carry no bias, activation or mask and take the PLAIN variant above)
    float* Cbase = P.C;
    const long ldc = P.ldc;
    constexpr int SP = 68;
    float* patch = stage + (wm * 2 + wn) * (16 * SP);
    const bool vec_ok = ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cbase) & 15) == 0) &&
                        (!P.keep || (reinterpret_cast<uintptr_t>(P.keep) & 15) == 0) &&
                        (!P.aux || (reinterpret_cast<uintptr_t>(P.aux) & 15) == 0) &&
                        (!P.mask_src || ((reinterpret_cast<uintptr_t>(P.mask_src) & 15) == 0 && (P.ld_mask & 3) == 0));
    const int cg = lane & 15;             // column group: columns col0 .. col0+3
    const int rq = lane >> 4;             // row within a group of four
    const int c4 = cg * 4;
    const int col0 = n0 + wn * 64 + c4;
    const long bitcol = (n0 + wn * 64) >> 6;   // 64-column block index of this wave (mask words)
    const long bitld = P.N >> 6;
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (col0 + e < P.N) {
        if (P.bias0) bias4[e] += P.bias0[col0 + e];
        if (P.bias1) bias4[e] += P.bias1[col0 + e];
      }
    }
    auto finish = [&](float v, long off, int row, int col) -> float {   // off = row * ldc + col
      if (P.act == 1) v = fmaxf(v, 0.f);
      else if (P.act == 2) v = expf(v);
      else if (P.act == 3) {
        if (P.aux) P.aux[off] = (v >= 1e-6f) ? 1.f : 0.f;
        v = fmaxf(v, 1e-6f);
      } else if (P.act == 4) {
        v = fmaxf(v, 0.f);
        if (P.keep) v = v * P.keep[off] * P.keep_scale;
        else if (P.drop_thresh)
          v = vts_drop_keep(P.drop_seed, (unsigned long long)row * P.N + col, P.drop_thresh) ? v * P.keep_scale : 0.f;
      }
      if (P.mask_src) v = (P.mask_src[(long)row * P.ld_mask + col] != 0.f) ? v * P.mask_scale : 0.f;
      if (P.accumulate) v += Cbase[off];
      return v;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) patch[(4 * g + reg) * SP + j * 16 + r] = acc[i][j][reg];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rl = it * 4 + rq;
        const int row = m0 + wm * 64 + i * 16 + rl;
        f32x4 v = *reinterpret_cast<const f32x4*>(patch + rl * SP + c4);
        const bool valid = row < P.M && col0 < P.N;
        unsigned nz = 0;                  // bit e: element e of this lane is non-zero in the output
        if (valid) {
          const long off = (long)row * ldc + col0;
          v += bias4;
          if (vec_ok && col0 + 3 < P.N && P.act != 3 && !(P.act == 4 && P.keep == nullptr && P.drop_thresh)) {
            // vector forms of the element transforms (clamp-with-aux and hashed dropout take the paths below)
            if (P.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            else if (P.act == 2) { v[0] = expf(v[0]); v[1] = expf(v[1]); v[2] = expf(v[2]); v[3] = expf(v[3]); }
            else if (P.act == 4) {
              v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
              if (P.keep) v = v * *reinterpret_cast<const f32x4*>(P.keep + off) * P.keep_scale;
            }
            if (P.mask_bits) {
              const unsigned long long word = P.mask_bits[(long)row * bitld + bitcol];
              const unsigned nib = (unsigned)(word >> (16 * (cg & 3) + 4 * (cg >> 2))) & 15u;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = ((nib >> e) & 1u) ? v[e] * P.mask_scale : 0.f;
            } else if (P.mask_src) {
              const f32x4 mk = *reinterpret_cast<const f32x4*>(P.mask_src + (long)row * P.ld_mask + col0);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (mk[e] != 0.f) ? v[e] * P.mask_scale : 0.f;
            }
            if (P.accumulate) v += *reinterpret_cast<const f32x4*>(Cbase + off);
            *reinterpret_cast<f32x4*>(Cbase + off) = v;
            csum += v;
          } else if (vec_ok && col0 + 3 < P.N && P.act == 4) {
            // hashed dropout: element hashes, one vector store
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = finish(v[e], off + e, row, col0 + e);
            *reinterpret_cast<f32x4*>(Cbase + off) = v;
            csum += v;
          } else {
            for (int e = 0; e < 4; ++e) {
              if (col0 + e < P.N) {
                v[e] = finish(v[e], off + e, row, col0 + e);
                Cbase[off + e] = v[e];
                csum[e] += v[e];
              } else {
                v[e] = 0.f;
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) nz |= (v[e] != 0.f ? 1u : 0u) << e;
        }
        if (P.bits_out) {
          // One 64-bit word per (row, 64-column block).  Column c of the block sits at bit
          // 16*((c>>2)&3) + 4*(c>>4) + (c&3): this lane's four columns are one nibble, and the 16 lanes of
          // the row are OR-ed together with xor-shuffles (every lane of the wave reaches them: the branch
          // above has rejoined).  The fused feed-forward kernels use the same layout.
          const int sh = 16 * (cg & 3) + 4 * (cg >> 2);
          unsigned lo32 = sh < 32 ? nz << sh : 0u, hi32 = sh >= 32 ? nz << (sh - 32) : 0u;
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { lo32 |= __shfl_xor(lo32, o); hi32 |= __shfl_xor(hi32, o); }
          if (cg == 0 && row < P.M && n0 + wn * 64 < P.N)
            P.bits_out[(long)row * bitld + bitcol] = ((unsigned long long)hi32 << 32) | lo32;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
    if (P.colpart) {   // the four row groups of the wave hold the same columns
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        csum[e] += __shfl_xor(csum[e], 16);
        csum[e] += __shfl_xor(csum[e], 32);
      }
      if (lane < 16) {
        float* dst = P.colpart + ((long)(m0 / BM) * 2 + wm) * P.N;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col0 + e < P.N) dst[col0 + e] = csum[e];
      }
    }
  }
}

template <int AK, int BKIND, int CK, bool PLAIN>
__global__ __launch_bounds__(256, VTS_GEMM_WPS) void gemm_f32_kernel(const GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];   // [buf][sub][A|B][BKT * LDS_LD]
  auto lds = [&](int buf, int sub, int op) -> float* { return lds_raw + ((buf * NSUB + sub) * 2 + op) * (BKT * LDS_LD); };
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  int m0, n0, zslice, kbeg, kend;
  if (!map_workgroup(P, m0, n0, zslice, kbeg, kend)) return;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float va[NSUB][8], vb[NSUB][8];
  const float* arow = nullptr;
  const float* brow = nullptr;
  if constexpr (AK == A_ROW) arow = a_row_base(P, m0, tid);
  if constexpr (BKIND == B_NK) brow = b_row_base(P, n0, tid);
  const int nk = (kend - kbeg + BKS - 1) / BKS;
  if (nk > 0) {
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
      load_a<AK>(P, arow, m0, kbeg + u * BKT, kend, tid, va[u]);
      load_b<BKIND>(P, brow, n0, kbeg + u * BKT, kend, tid, vb[u]);
    }
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
      store_a<AK>(lds(0, u, 0), tid, va[u]);
      store_b<BKIND>(lds(0, u, 1), tid, vb[u]);
    }
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) {
#pragma unroll
      for (int u = 0; u < NSUB; ++u) {
        load_a<AK>(P, arow, m0, kbeg + (kt + 1) * BKS + u * BKT, kend, tid, va[u]);
        load_b<BKIND>(P, brow, n0, kbeg + (kt + 1) * BKS + u * BKT, kend, tid, vb[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
      const float* As = lds(cur, u, 0);
      const float* Bs = lds(cur, u, 1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[(4 * kk + g) * LDS_LD + wm * 64 + i * 16 + r];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[(4 * kk + g) * LDS_LD + wn * 64 + j * 16 + r];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < NSUB; ++u) {
        store_a<AK>(lds(cur ^ 1, u, 0), tid, va[u]);
        store_b<BKIND>(lds(cur ^ 1, u, 1), tid, vb[u]);
      }
    }
    __syncthreads();
  }

  epilogue<CK, PLAIN>(P, acc, m0, n0, zslice, wm, wn, lane, lds_raw);
}

// =================================================================================================
// bf16x3 variant: the same contraction on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, 16x the
// f32 MFMA rate) at fp32-level accuracy.  Every fp32 operand x is split while it is staged into
// LDS, x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and each product is evaluated as
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation (the dropped a_lo*b_lo term and the
// split residuals are ~2^-16 relative).  Tile 128 x 128 x 32, LDS rows hold 32 k values (k
// contiguous, 80-byte row pitch) in two planes (hi, lo) per operand; the C/D layout of the 16x16
// bf16 MFMA equals the f32 one, so mapping and epilogue are shared with the f32 kernel.
constexpr int BK3 = 32;                 // k elements per main-loop iteration
constexpr int ROWB = 64;                // LDS row pitch in bytes (32 bf16, no padding)
constexpr int PLANE = 128 * ROWB;       // one plane (hi or lo) of one operand tile

// Byte offset of (row, byte kbyte of the row's 64) inside a plane.  Rows are 64 bytes = 16 banks, so
// an unswizzled tile maps rows r, r+4, r+8, ... onto the same banks.  Two XORs spread them:
//   the row's low two bits        ^= bits 4-5 of the row   (separates rows 16 apart)
//   the 16-byte chunk of the row  ^= bits 2-3 of the row   (separates rows 4 apart)
// With it the MFMA fragment reads (16 consecutive rows, one chunk, 16 bytes per lane) touch every
// bank once, and the stores of all four operand kinds (row-pair halves, 4-row columns of 8
// bytes, tile16 blocks either way round) hit each bank the minimum number of times or twice that.
__device__ __forceinline__ int lds3_off(int row, int kbyte) {
  const int prow = (row & ~3) | ((row & 3) ^ ((row >> 4) & 3));
  return prow * ROWB + ((((kbyte >> 4) ^ ((row >> 2) & 3)) << 4) | (kbyte & 15));
}
#ifndef VTS_GEMM3_NSUB
#define VTS_GEMM3_NSUB 1
#endif
#ifndef VTS_GEMM3_PIPE
#define VTS_GEMM3_PIPE 0
#endif
#ifndef VTS_GEMM3_WPS
#define VTS_GEMM3_WPS (VTS_GEMM3_PIPE ? 2 : 3)
#endif
constexpr int NSUB3 = VTS_GEMM3_NSUB;   // 32-deep k sub-chunks staged per barrier pair
constexpr bool PIPE3 = VTS_GEMM3_PIPE;  // software-pipelined k loop (two LDS buffers) or one buffer and more workgroups per CU
constexpr size_t GEMM3_LDS_BYTES = (PIPE3 ? 2ul : 1ul) * 4ul * PLANE * NSUB3;   // per buffer: A hi, A lo, B hi, B lo
static_assert(GEMM3_LDS_BYTES >= 4 * 16 * 68 * sizeof(float), "row-major epilogue stages 4 x [16][68] floats in LDS");

__device__ __forceinline__ void load4(const float* __restrict__ p, int nvalid, float (&v)[4]) {
  if (nvalid >= 4) {
    const f32x4_u x = *reinterpret_cast<const f32x4_u*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (i < nvalid) ? p[i] : 0.f;
  }
}

// Row pointer of logical row `row` of a k-contiguous operand (A_ROW with the optional window-row
// mapping, or B_NK), or nullptr when the row is outside the matrix.
__device__ __forceinline__ const float* row_ptr3(const GemmParams& P, bool is_a, int row) {
  if (is_a) {
    bool valid = row < P.M;
    long rr = row;
    if (P.a_map && valid) rr = window_row(row, P.T, P.Bwin, P.a_tshift, valid);
    return valid ? P.A + rr * P.lda : nullptr;
  }
  return (row < P.N) ? P.B + (long)row * P.ldb : nullptr;
}

// 16 fp32 values per thread for one 128 x 32 operand tile as four 16-byte pieces v[c][0..3]; every
// piece is one load instruction, and the lane -> address maps are chosen so that an instruction's
// 64 lanes read whole cache lines (a strided map costs several hundred cycles of issue
// back-pressure per instruction).  OPK:
//  0 = k-contiguous rows (A_ROW, B_NK): row tid/2, piece c = k 16*(tid%2) + 4c..+3 (64 contiguous
//      bytes per thread; a lane-contiguous map measured slower here: it needs four window-row
//      pointers per thread in the prologue of a kernel whose k loop is only 8-12 steps long);
//  1 = row-contiguous (A_COL, B_KN): piece c = k row 4*(tid/32) + c, rows 4*(tid%32)..+3
//      (32 lanes x 16 B contiguous);
//  2 = tile16 [rows, K] (A_T16): piece c = slot 16*g + 4*cq + c of block tid/16 (see store3);
//  3 = tile16 [K, rows] (A_T16T): piece c = block 4*wave + c, slot lane (one whole 1 KiB block
//      per instruction): k = 4*(lane/16)..+3 of row lane%16.
template <int OPK>
__device__ __forceinline__ void load3(const GemmParams& P, bool is_a, const float* const (&rows)[4], int r0, int k0,
                                      int kend, int tid, float (&v)[4][4]) {
  const float* base = is_a ? P.A : P.B;
  const long ld = is_a ? P.lda : P.ldb;
  const int R = is_a ? P.M : P.N;          // extent of the row (m or n) index
  if constexpr (OPK == 0) {
    const int k = k0 + (tid & 1) * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int nv = (rows[0] != nullptr) ? min(4, kend - (k + 4 * c)) : 0;
      if (nv > 0) load4(rows[0] + k + 4 * c, nv, v[c]);
      else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
    }
  } else if constexpr (OPK == 1) {
    const int rr = r0 + (tid & 31) * 4;
    const int kb = k0 + (tid >> 5) * 4;
    const int map = is_a ? 0 : P.b_map;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = kb + s;
      bool valid = k < kend;
      long row = k;
      if (map && valid) row = window_row(k, P.T, P.Bwin, P.b_tshift, valid);
      const int nv = valid ? min(4, R - rr) : 0;
      if (nv > 0) load4(base + row * ld + rr, nv, v[s]);
      else { v[s][0] = v[s][1] = v[s][2] = v[s][3] = 0.f; }
    }
  } else if constexpr (OPK == 2) {
    const int blk = tid >> 4;
    const int rt = (r0 >> 4) + (blk >> 1), kt = (k0 >> 4) + (blk & 1);
    const int g = (tid & 15) >> 2, cq = tid & 3;
    const bool valid = (rt * 16 < R) && (kt * 16 < kend);
    const float* p = base + ((long)rt * (P.K >> 4) + kt) * 256 + (g * 16 + 4 * cq) * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (valid) load4(p + 4 * s, 4, v[s]);
      else { v[s][0] = v[s][1] = v[s][2] = v[s][3] = 0.f; }
    }
  } else {
    const int wave = tid >> 6, lane = tid & 63;
    const int kt = (k0 >> 4) + (wave >> 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int rt = (r0 >> 4) + 4 * (wave & 1) + c;
      const bool valid = (rt * 16 < R) && (kt * 16 < kend);
      if (valid) load4(base + ((long)kt * (R >> 4) + rt) * 256 + lane * 4, 4, v[c]);
      else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
    }
  }
}

// Per-thread streaming state of one operand for the interior of the k loop: everything that does
// not change from one 32-deep k-step to the next (row validity, base pointers, strides) is worked
// out once, so that a full k-step costs four 16-byte loads and a pointer bump.  load3 above stays
// the path for the last, partial k-step and for threads on the ragged edge of the matrix.
template <int OPK>
struct Stream3 {
  const float* p = nullptr;   // next k-step's first element for this thread (nullptr: contributes zeros)
  const float* rows[4] = {nullptr, nullptr, nullptr, nullptr};   // OPK 0: rows[0] = this thread's row
  long step = 0;              // elements between consecutive k-steps
  long ld = 0;                // OPK 1: row stride
  int nv = 4;                 // OPK 1: valid elements of this thread's 4-wide row pieces (N % 4 != 0 edge)
  // OPK 1 with the window-row mapping: position of this thread's 16-row group
  bool map = false;
  int T = 1, Bwin = 0, tshift = 0, tile = 0, t = 0, bi = 0;
  const float* base = nullptr;
  int rr = 0;

  __device__ __forceinline__ void init(const GemmParams& P, bool is_a, int r0, int kbeg, int tid) {
    const float* b = is_a ? P.A : P.B;
    ld = is_a ? P.lda : P.ldb;
    const int R = is_a ? P.M : P.N;
    if constexpr (OPK == 0) {
      rows[0] = row_ptr3(P, is_a, r0 + (tid >> 1));
      p = rows[0] ? rows[0] + kbeg + (tid & 1) * 16 : nullptr;
      step = BK3;
    } else if constexpr (OPK == 1) {
      rr = r0 + (tid & 31) * 4;
      nv = min(4, max(0, R - rr));
      map = !is_a && P.b_map;
      const int k = kbeg + (tid >> 5) * 4;
      base = b;
      if (map) {
        T = P.T; Bwin = P.Bwin; tshift = P.b_tshift;
        bi = k & 15;
        tile = (k >> 4) / T;
        t = (k >> 4) % T;
      } else {
        p = nv > 0 ? b + (long)k * ld + rr : nullptr;
        step = (long)BK3 * ld;
      }
    } else if constexpr (OPK == 2) {
      const int blk = tid >> 4;
      const int rt = (r0 >> 4) + (blk >> 1), kt = (kbeg >> 4) + (blk & 1);
      if (rt * 16 < R) p = b + ((long)rt * (P.K >> 4) + kt) * 256 + (((tid & 15) >> 2) * 16 + 4 * (tid & 3)) * 4;
      step = 2 * 256;
    } else {
      const int wave = tid >> 6, lane = tid & 63;
      const int kt = (kbeg >> 4) + (wave >> 1), rt = (r0 >> 4) + 4 * (wave & 1);
      if ((rt + 4) * 16 <= R) p = b + ((long)kt * (R >> 4) + rt) * 256 + lane * 4;   // four row blocks in range
      step = 2L * (R >> 4) * 256;
    }
  }

  // one full k-step (all 32 k inside the slice); v as load3 fills it
  __device__ __forceinline__ void load_full(float (&v)[4][4]) {
    const float* q = p;
    long rs = 4;                     // distance between the four 16-byte pieces
    if constexpr (OPK == 3) rs = 256;
    if constexpr (OPK == 1) {
      rs = ld;
      if (map) {
        // rows b = tile*16 + bi + s of time slice t (+ tshift): all four in range, or zeros
        const int tt = t + tshift;
        const bool ok = nv > 0 && tt >= 0 && tt < T && (tile * 16 + bi + 4 <= Bwin);
        q = ok ? base + ((long)(tile * 16 + bi) * T + tt) * ld + rr : nullptr;
        rs = (long)T * ld;
      }
    }
    if (q != nullptr && (OPK != 1 || nv == 4)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x4_u x = *reinterpret_cast<const f32x4_u*>(q + s * rs);
        v[s][0] = x[0]; v[s][1] = x[1]; v[s][2] = x[2]; v[s][3] = x[3];
      }
    } else if (q != nullptr) {       // the one thread whose piece crosses the last column
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[s][i] = (i < nv) ? q[s * rs + i] : 0.f;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) v[s][0] = v[s][1] = v[s][2] = v[s][3] = 0.f;
    }
  }
  __device__ __forceinline__ void advance() {
    if constexpr (OPK == 1) {
      if (map) {                     // 32 logical rows further: two time slices of the same 16 windows
        t += 2;
        if (t >= T) { t -= T; ++tile; }
        return;
      }
    }
    if (p != nullptr) p += step;
  }
  // threads that must take the checked path for this k-step even though it is a full one
  __device__ __forceinline__ bool ragged() const {
    if constexpr (OPK == 1) return map && (tile * 16 + 16 > Bwin) && (tile * 16 < Bwin);
    if constexpr (OPK == 3) return p == nullptr;   // a row block past the edge: per-block checks
    return false;
  }
};

// Split and write four of the thread's 16 values (chunk c = 0..3) into the hi / lo planes of the
// operand's LDS tile; the four chunks together cover the tile.  Chunked so that the caller can
// spread the conversion work between MFMAs.
template <int OPK>
__device__ __forceinline__ void store3_chunk(char* __restrict__ hi, char* __restrict__ lo, int tid,
                                             const float (&v)[4][4], int c) {
  uint2 h, l;
  if constexpr (OPK == 0) {          // v[c][i] = (row tid>>1, k = 16*(tid&1) + 4c + i)
    const int off = lds3_off(tid >> 1, (tid & 1) * 32 + 8 * c);
    split4(v[c][0], v[c][1], v[c][2], v[c][3], h, l);
    *reinterpret_cast<uint2*>(hi + off) = h;
    *reinterpret_cast<uint2*>(lo + off) = l;
  } else if constexpr (OPK == 1 || OPK == 2) {   // v[s][i] = (row base + i, k = kbase + s)
    int rowb, kbyte;
    if constexpr (OPK == 1) { rowb = (tid & 31) * 4; kbyte = (tid >> 5) * 8; }
    else { const int blk = tid >> 4; rowb = (blk >> 1) * 16 + ((tid & 15) >> 2) * 4; kbyte = ((blk & 1) * 16 + (tid & 3) * 4) * 2; }
    split4(v[0][c], v[1][c], v[2][c], v[3][c], h, l);
    *reinterpret_cast<uint2*>(hi + lds3_off(rowb + c, kbyte)) = h;
    *reinterpret_cast<uint2*>(lo + lds3_off(rowb + c, kbyte)) = l;
  } else {                           // v[c][i] = (row 16*(4*(wave&1) + c) + lane%16, k = 16*(wave/2) + 4*(lane/16) + i)
    const int wave = tid >> 6, lane = tid & 63;
    const int row = (4 * (wave & 1) + c) * 16 + (lane & 15);
    const int kbyte = (wave >> 1) * 32 + (lane >> 4) * 8;
    split4(v[c][0], v[c][1], v[c][2], v[c][3], h, l);
    *reinterpret_cast<uint2*>(hi + lds3_off(row, kbyte)) = h;
    *reinterpret_cast<uint2*>(lo + lds3_off(row, kbyte)) = l;
  }
}
template <int OPK>
__device__ __forceinline__ void store3(char* __restrict__ hi, char* __restrict__ lo, int tid, const float (&v)[4][4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) store3_chunk<OPK>(hi, lo, tid, v, c);
}

template <int AK> struct OpKindA { static constexpr int v = (AK == A_ROW) ? 0 : (AK == A_COL) ? 1 : (AK == A_T16) ? 2 : 3; };
template <int BKIND> struct OpKindB { static constexpr int v = (BKIND == B_NK) ? 0 : 1; };

template <int AK, int BKIND, int CK, bool PLAIN>
__global__ __launch_bounds__(256, VTS_GEMM3_WPS) void gemm_bf16x3_kernel(const GemmParams P) {
  extern __shared__ __attribute__((aligned(16))) char lds3[];   // [sub][A hi | A lo | B hi | B lo]
  constexpr int OA = OpKindA<AK>::v, OB = OpKindB<BKIND>::v;
  constexpr int BKL = BK3 * NSUB3;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  int m0, n0, zslice, kbeg, kend;
  if (!map_workgroup(P, m0, n0, zslice, kbeg, kend)) return;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  static_assert(NSUB3 == 1, "the pipeline below assumes one 32-deep chunk per step");
  // raw fp32 values of two tiles in flight: tile j lives in register set j & 1
  float va[2][4][4], vb[2][4][4];
  Stream3<OA> sa;
  Stream3<OB> sb;
  sa.init(P, true, m0, kbeg, tid);
  sb.init(P, false, n0, kbeg, tid);
  // k0 .. k0+31 entirely inside the slice: unchecked loads, except for threads on a ragged edge
  auto stage_load = [&](int k0, float (&ra)[4][4], float (&rb)[4][4]) {
    const bool full = k0 + BK3 <= kend;
    if (full && !sa.ragged()) sa.load_full(ra);
    else load3<OA>(P, true, sa.rows, m0, k0, kend, tid, ra);
    if (full && !sb.ragged()) sb.load_full(rb);
    else load3<OB>(P, false, sb.rows, n0, k0, kend, tid, rb);
    sa.advance();
    sb.advance();
  };
  // Software pipeline over 32-deep k-steps, two LDS buffers, two register sets, one barrier per step:
  //   step kt:  global loads of tile kt+2 -> register set kt&1   (a whole step ahead of their use)
  //             MFMAs on LDS buffer kt&1   ||   split + store of tile kt+1 (register set (kt+1)&1)
  //             into LDS buffer (kt+1)&1
  // The conversion VALU work and the LDS stores have no dependence on the step's MFMAs, so they
  // issue in the shadow of the matrix pipe (a 16x16x32 MFMA occupies it for 16 cycles but takes
  // one issue slot).  The barrier orders LDS only: a full __syncthreads() would also drain the
  // loads in flight.
  const int nk = (kend - kbeg + BKL - 1) / BKL;
  if (nk > 0) {
    stage_load(kbeg, va[0], vb[0]);
    store3<OA>(lds3, lds3 + PLANE, tid, va[0]);
    store3<OB>(lds3 + 2 * PLANE, lds3 + 3 * PLANE, tid, vb[0]);
    if (nk > 1) stage_load(kbeg + BKL, va[1], vb[1]);
  }
  vts_lds_barrier();
  int a_off[4], b_off[4];               // this lane's fragment of each of the wave's 4 + 4 row blocks
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_off[i] = lds3_off(wm * 64 + i * 16 + r, g * 16);
    b_off[i] = lds3_off(wn * 64 + i * 16 + r, g * 16);
  }
  auto step = [&](int kt, auto par) {
    constexpr int PAR = decltype(par)::value;         // kt & 1
    if (kt + 2 < nk) stage_load(kbeg + (kt + 2) * BKL, va[PAR], vb[PAR]);
    const char* a_hi = lds3 + PAR * 4 * PLANE;
    const char* a_lo = a_hi + PLANE;
    const char* b_hi = a_hi + 2 * PLANE;
    const char* b_lo = a_hi + 3 * PLANE;
    bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + a_off[i]);
      al[i] = *reinterpret_cast<const bf16x8*>(a_lo + a_off[i]);
      bh[i] = *reinterpret_cast<const bf16x8*>(b_hi + b_off[i]);
      bl[i] = *reinterpret_cast<const bf16x8*>(b_lo + b_off[i]);
    }
    // 48 MFMAs in 16 groups of three (one accumulator tile each); after every second group one
    // eighth of the next tile's conversion (12 VALU, 2 LDS stores) issues in the matrix pipe's
    // shadow.  sched_barrier(0) pins that order.  The stores are unconditional: on the last step
    // they re-store stale registers into the idle buffer, which keeps this block branch-free.
    char* nbase = lds3 + (1 - PAR) * 4 * PLANE;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = q & 3, j = q >> 2;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      if (q & 1) {
        const int c = q >> 1;          // 0..7: four chunks of A, then four of B
        if (c < 4) store3_chunk<OA>(nbase, nbase + PLANE, tid, va[1 - PAR], c);
        else store3_chunk<OB>(nbase + 2 * PLANE, nbase + 3 * PLANE, tid, vb[1 - PAR], c - 4);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    vts_lds_barrier();
  };
  if constexpr (PIPE3) {
    for (int kt = 0; kt < nk; kt += 2) {
      step(kt, std::integral_constant<int, 0>{});
      if (kt + 1 < nk) step(kt + 1, std::integral_constant<int, 1>{});
    }
  } else {
    // One LDS buffer: the next tile's loads fly during the MFMAs, its conversion and stores sit
    // between two barriers.  Latency is hidden by the other workgroups of the CU (3 resident)
    // rather than inside the wave.
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more && kt > 0) stage_load(kbeg + (kt + 1) * BKL, va[1], vb[1]);
      bf16x8 ah[4], al[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(lds3 + a_off[i]);
        al[i] = *reinterpret_cast<const bf16x8*>(lds3 + PLANE + a_off[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(lds3 + 2 * PLANE + b_off[j]);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(lds3 + 3 * PLANE + b_off[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
        }
      }
      vts_lds_barrier();           // every wave is done reading this k chunk
      if (more) {
        store3<OA>(lds3, lds3 + PLANE, tid, va[1]);
        store3<OB>(lds3 + 2 * PLANE, lds3 + 3 * PLANE, tid, vb[1]);
      }
      vts_lds_barrier();
    }
  }
  epilogue<CK, PLAIN>(P, acc, m0, n0, zslice, wm, wn, lane, reinterpret_cast<float*>(lds3));
}

// ---- 256 x 128 tile variant ---------------------------------------------------------------------
// The 128 x 128 kernel above runs at the rate one CU can pull operand bytes through its L1 (fp32
// operands: 32 KiB per 32-deep k-step and workgroup), not at the matrix-core rate.  A 256 x 128
// tile needs 48 KiB for twice the products, i.e. 25 % fewer bytes per FLOP.  8 waves: the two
// halves of the workgroup (threads 0-255 / 256-511) each stage their own 128 rows of A with the
// 256-thread lane maps above; the B tile of k-step j is staged by half j & 1, so a thread issues
// four A loads every step and four B loads every other step.  One workgroup per CU (two waves per
// SIMD, 96 KiB of LDS): two LDS stages and two register sets, loads two k-steps ahead of their
// use, conversion and LDS stores in the shadow of the MFMAs, one LDS-only barrier per step.
constexpr int BMB = 256;
constexpr int STAGE_B = 2 * 2 * PLANE + 2 * PLANE;        // A hi, A lo (256 rows each), B hi, B lo
constexpr size_t GEMM3B_LDS_BYTES = 2ul * STAGE_B;
static_assert(GEMM3B_LDS_BYTES >= 8 * 16 * 68 * sizeof(float), "row-major epilogue stages 8 x [16][68] floats in LDS");

template <int AK, int BKIND, int CK, bool PLAIN>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_big_kernel(const GemmParams Pin) {
  extern __shared__ __attribute__((aligned(16))) char lds3[];   // [stage][A hi | A lo | B hi | B lo]
  constexpr int OA = OpKindA<AK>::v, OB = OpKindB<BKIND>::v;
  const int tid = threadIdx.x;
  const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
  const int t2 = tid & 255;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;          // wave tile (wm, wn) of the 4 x 2 grid of 64 x 64 blocks
  const int r = lane & 15, g = lane >> 4;

  int m0, n0, zslice, kbeg, kend;
  if (!map_workgroup<BMB>(Pin, m0, n0, zslice, kbeg, kend)) return;
  // a paired launch: this workgroup's column tile belongs to one of the two products
  GemmParams P = Pin;
  if (Pin.n_cut > 0) {
    if (n0 >= Pin.n_cut) {
      P.B = Pin.B2; P.ldb = Pin.ldb2; P.b_tshift = Pin.b_tshift2; P.C = Pin.C2; P.N = Pin.N2;
      n0 -= Pin.n_cut;
    } else {
      P.N = Pin.N1;
    }
  }
  const int ma = m0 + half * 128;                    // first A row this half stages

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float va[2][4][4], vb[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) vb[i][0] = vb[i][1] = vb[i][2] = vb[i][3] = 0.f;
  Stream3<OA> sa;
  Stream3<OB> sb;
  sa.init(P, true, ma, kbeg, t2);
  sb.init(P, false, n0, kbeg + half * BK3, t2);     // this half's B tiles: k-steps half, half + 2, ...
  auto load_a = [&](int j, float (&ra)[4][4]) {
    const int k0 = kbeg + j * BK3;
    if (k0 + BK3 <= kend && !sa.ragged()) sa.load_full(ra);
    else load3<OA>(P, true, sa.rows, ma, k0, kend, t2, ra);
    sa.advance();
  };
  auto load_b = [&](int j) {                         // j & 1 == half
    const int k0 = kbeg + j * BK3;
    if (k0 + BK3 <= kend && !sb.ragged()) sb.load_full(vb);
    else load3<OB>(P, false, sb.rows, n0, k0, kend, t2, vb);
    sb.advance();
    sb.advance();
  };
  const int nk = (kend - kbeg + BK3 - 1) / BK3;
  char* const a_st = lds3 + half * PLANE;            // this half's rows inside the A planes
  if (nk > 0) {
    load_a(0, va[0]);
    if (half == 0) load_b(0);
    store3<OA>(a_st, a_st + 2 * PLANE, t2, va[0]);
    if (half == 0) store3<OB>(lds3 + 4 * PLANE, lds3 + 5 * PLANE, t2, vb);
    if (nk > 1) {
      load_a(1, va[1]);
      if (half == 1) load_b(1);
    }
  }
  vts_lds_barrier();
  int a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_off[i] = lds3_off(wm * 64 + i * 16 + r, g * 16);
    b_off[i] = lds3_off(wn * 64 + i * 16 + r, g * 16);
  }
  auto step = [&](int kt, auto par) {
    constexpr int PAR = decltype(par)::value;         // kt & 1
    if (kt + 2 < nk) {
      load_a(kt + 2, va[PAR]);
      if (half == PAR) load_b(kt + 2);
    }
    const char* a_hi = lds3 + PAR * STAGE_B;
    const char* a_lo = a_hi + 2 * PLANE;
    const char* b_hi = a_hi + 4 * PLANE;
    const char* b_lo = a_hi + 5 * PLANE;
    bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(a_hi + a_off[i]);
      al[i] = *reinterpret_cast<const bf16x8*>(a_lo + a_off[i]);
      bh[i] = *reinterpret_cast<const bf16x8*>(b_hi + b_off[i]);
      bl[i] = *reinterpret_cast<const bf16x8*>(b_lo + b_off[i]);
    }
    // next tile -> the other stage: A chunks after MFMA groups 1, 3, 5, 7, this half's B chunks (every
    // other step) after 9, 11, 13, 15.  On the last step stale registers go into the idle stage.
    char* nbase = lds3 + (1 - PAR) * STAGE_B;
    char* na = nbase + half * PLANE;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = q & 3, j = q >> 2;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      if (q & 1) {
        const int c = q >> 1;
        if (c < 4) store3_chunk<OA>(na, na + 2 * PLANE, t2, va[1 - PAR], c);
        else if (half != PAR) store3_chunk<OB>(nbase + 4 * PLANE, nbase + 5 * PLANE, t2, vb, c - 4);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    vts_lds_barrier();
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nk) step(kt + 1, std::integral_constant<int, 1>{});
  }
  epilogue<CK, PLAIN>(P, acc, m0, n0, zslice, wm, wn, lane, reinterpret_cast<float*>(lds3));
}

// Sum ksplit slabs [S][M*N] into C (row stride ldc), optionally accumulating onto C: few slabs of a
// large output, one thread per element.
__global__ void reduce_few_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N,
                                        long ldc, int S, int accumulate) {
  const long total = (long)M * N;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[(long)z * total + idx];
    float* dst = C + (idx / N) * ldc + idx % N;
    *dst = accumulate ? (*dst + s) : s;
  }
}

// The same for many slabs of a small output (weight gradients contracted over ~10^5 rows).  A workgroup owns
// 64 consecutive output elements; its 16 waves split the slabs (four loads in flight each) and
// combine through LDS in a fixed order.
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ C,
                                                            int M, int N, long ldc, int S, int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long total = (long)M * N;
  for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
    const long idx = base + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
      const float* p0 = slabs + idx;
      int z = wave;
      for (; z + 48 < S; z += 64) {
        s0 += p0[(long)z * total];
        s1 += p0[(long)(z + 16) * total];
        s2 += p0[(long)(z + 32) * total];
        s3 += p0[(long)(z + 48) * total];
      }
      for (; z < S; z += 16) s0 += p0[(long)z * total];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && idx < total) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) s += red[w][lane];
      float* dst = C + (idx / N) * ldc + idx % N;
      *dst = accumulate ? (*dst + s) : s;
    }
    __syncthreads();
  }
}

static bool plain_epilogue(const GemmParams& P) {
  return !P.bias0 && !P.bias1 && P.act == 0 && !P.mask_src && !P.mask_bits && !P.bits_out && !P.colpart;
}

// Split-K launches planned for it (vts_gemm_pick_ksplit) take the 256 x 128 tile.  Launches with a short
// contraction (the LSTM input projection and data gradient: 8 to 32 k-steps per workgroup) do not: there a
// workgroup's life is mostly prologue and epilogue, which only other resident workgroups can hide
// (measured at 4096 x 96 windows: 13 % slower with one big workgroup per CU, against 6-11 % faster for
// the weight gradients' 384-step loops).
static bool use_big_tile(const GemmParams& P) { return P.mode == 1 && P.ksplit > 1 && P.big_split != 0; }

template <int AK, int BKIND, int CK, bool PLAIN>
int launch3big(const GemmParams& P, hipStream_t stream) {
  const int tiles_m = vts_cdiv(P.M, BMB), tiles_n = vts_cdiv(P.N, BN);
  const long nwg = (P.ksplit > 1) ? 8L * vts_cdiv(P.ksplit, 8) * tiles_m * tiles_n : 8L * vts_cdiv(tiles_m, 8) * tiles_n;
  static vts_lds_mask lds_done{0};   // per instantiation
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(gemm_bf16x3_big_kernel<AK, BKIND, CK, PLAIN>),
                               GEMM3B_LDS_BYTES, &lds_done, "gemm_bf16x3_big"))
    return rc;
  hipLaunchKernelGGL((gemm_bf16x3_big_kernel<AK, BKIND, CK, PLAIN>), dim3((unsigned)nwg), dim3(512), GEMM3B_LDS_BYTES,
                     stream, P);
  VTS_CHECK_LAUNCH("gemm_bf16x3_big");
  return VTS_OK;
}

template <int AK, int BKIND, int CK, bool PLAIN>
int launch3p(const GemmParams& P, hipStream_t stream) {
  if (use_big_tile(P)) return launch3big<AK, BKIND, CK, PLAIN>(P, stream);
  const int tiles_m = vts_cdiv(P.M, BM), tiles_n = vts_cdiv(P.N, BN);
  const long nwg = (P.ksplit > 1) ? 8L * vts_cdiv(P.ksplit, 8) * tiles_m * tiles_n : 8L * vts_cdiv(tiles_m, 8) * tiles_n;
  static vts_lds_mask lds_done{0};   // per instantiation
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(gemm_bf16x3_kernel<AK, BKIND, CK, PLAIN>), GEMM3_LDS_BYTES,
                               &lds_done, "gemm_bf16x3"))
    return rc;
  hipLaunchKernelGGL((gemm_bf16x3_kernel<AK, BKIND, CK, PLAIN>), dim3((unsigned)nwg), dim3(256), GEMM3_LDS_BYTES, stream, P);
  VTS_CHECK_LAUNCH("gemm_bf16x3");
  return VTS_OK;
}

template <int AK, int BKIND, int CK, bool PLAIN>
int launchp(const GemmParams& P, hipStream_t stream) {
  const int tiles_m = vts_cdiv(P.M, BM), tiles_n = vts_cdiv(P.N, BN);
  const long nwg = (P.ksplit > 1) ? 8L * vts_cdiv(P.ksplit, 8) * tiles_m * tiles_n : 8L * vts_cdiv(tiles_m, 8) * tiles_n;
  dim3 grid((unsigned)nwg, 1, 1);
  static vts_lds_mask lds_done{0};   // per instantiation
  if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(gemm_f32_kernel<AK, BKIND, CK, PLAIN>), GEMM_LDS_BYTES,
                               &lds_done, "gemm_f32"))
    return rc;
  hipLaunchKernelGGL((gemm_f32_kernel<AK, BKIND, CK, PLAIN>), grid, dim3(256), GEMM_LDS_BYTES, stream, P);
  VTS_CHECK_LAUNCH("gemm_f32");
  return VTS_OK;
}

template <int AK, int BKIND, int CK>
int launch(const GemmParams& P, hipStream_t stream) {
  if (P.ksplit > 1 && !plain_epilogue(P)) {
    vts_set_error("gemm: split-K launches cannot carry a bias, activation, mask or column sums");
    return VTS_ERR_ARG;
  }
  if constexpr (CK == C_ROW) {
    if (plain_epilogue(P))
      return P.mode == 1 ? launch3p<AK, BKIND, CK, true>(P, stream) : launchp<AK, BKIND, CK, true>(P, stream);
  }
  return P.mode == 1 ? launch3p<AK, BKIND, CK, false>(P, stream) : launchp<AK, BKIND, CK, false>(P, stream);
}

}  // namespace

// ---- internal C++ entry points used by vts_api.hip -------------------------------------------

// Number of K splits for a contraction of length K into an M x N output.  Workgroups are dealt
// round-robin to the 8 XCDs and the kernel keeps every tile of one k slice on one XCD, so the unit
// of balance is the XCD: 32 CUs x 4 resident workgroups of this kernel (105 VGPRs, 36 KiB LDS) =
// 128 slots.  Pick as many whole slices per XCD as fit ONE residency round (a partly filled
// second round would run alone for a whole slice's duration), the same number on every XCD.
// Long contractions into a small output whose height fills whole 256-row tiles are planned for the
// 256 x 128 kernel (one workgroup per CU).
static bool split_plan_big(int M, int N, long K, int mode) {
  (void)N;
  return mode == 1 && K >= 32768 && (M % BMB) == 0;
}

int vts_gemm_pick_ksplit(int M, int N, long K, int mode) {
  const bool big = split_plan_big(M, N, K, mode);
  const int tiles = vts_cdiv(M, big ? BMB : BM) * vts_cdiv(N, BN);
  const int slots_per_xcd = big ? 32 : 32 * (mode == 1 ? 3 : 4);   // bf16x3 kernel: 160 VGPRs -> 3 workgroups per CU
  if (tiles >= 4 * slots_per_xcd) return 1;
  // A handful of output tiles and a short contraction (the reference's own sizes: 256 tokens): one workgroup per
  // tile would walk the k loop alone, a memory latency per iteration (29 us for 256 x 64 x 256); slices of 64 run
  // side by side instead and a small reduction adds them up.
  if (K <= 512) return (tiles <= 8 && K >= 128) ? (int)((K + 63) / 64) : 1;
  long per_xcd = slots_per_xcd / tiles;
  if (per_xcd < 1) per_xcd = 1;
  long s = 8 * per_xcd;
  const long maxs = (K + 255) / 256;  // at least 256 contraction elements per split
  if (s > maxs) s = maxs >= 8 ? maxs / 8 * 8 : maxs;
  if (s < 1) s = 1;
  return (int)s;
}

size_t vts_gemm_slab_bytes(int M, int N, int ksplit) {
  return ksplit > 1 ? (size_t)ksplit * M * N * sizeof(float) : 0;
}

static int finish_split(GemmParams P, float* Cfinal, long ldc, int accumulate, hipStream_t stream) {
  if (P.ksplit <= 1) return VTS_OK;
  const long total = (long)P.M * P.N;
  if (P.ksplit <= 32) {
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_few_slabs_kernel, dim3(blocks), dim3(256), 0, stream, P.C, Cfinal, P.M, P.N, ldc,
                       P.ksplit, accumulate);
    VTS_CHECK_LAUNCH("reduce_few_slabs");
    return VTS_OK;
  }
  int blocks = (int)((total + 63) / 64);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(1024), 0, stream, P.C, Cfinal, P.M, P.N, ldc,
                     P.ksplit, accumulate);
  VTS_CHECK_LAUNCH("reduce_slabs");
  return VTS_OK;
}

static void set_split(GemmParams& P, int ksplit) {
  P.ksplit = ksplit < 1 ? 1 : ksplit;
  long chunk = (P.K + P.ksplit - 1) / P.ksplit;
  chunk = (chunk + 63) / 64 * 64;
  P.kchunk = (int)chunk;
  P.ksplit = (int)((P.K + chunk - 1) / chunk);
  if (P.ksplit < 1) P.ksplit = 1;
}

// LSTM input projection: G[tile16 of [Mp, 4H]] = X[window rows] . W^T + b0 + b1
int vts_gemm_inproj(const float* X, const float* W, const float* b0, const float* b1, float* G, int Bwin, int T,
                    int Cin, int N4H, int mode, hipStream_t stream) {
  GemmParams P{};
  P.mode = mode;
  const int Bp = (Bwin + 15) / 16 * 16;
  P.A = X; P.B = W; P.C = G;
  P.M = Bp * T; P.N = N4H; P.K = Cin;
  P.lda = Cin; P.ldb = Cin; P.ldc = 0;
  P.a_map = 1; P.T = T; P.Bwin = Bwin;
  P.bias0 = b0; P.bias1 = b1;
  set_split(P, 1);
  return launch<A_ROW, B_NK, C_T16>(P, stream);
}

// dX[tile16 of [Mp, Cin]] (+)= dG[tile16 of [Mp, 4H]] . W[4H][Cin]
int vts_gemm_dgrad_t16(const float* dG, const float* W, float* dX, int Bwin, int T, int Cin, int N4H,
                       int accumulate, int mode, hipStream_t stream) {
  GemmParams P{};
  P.mode = mode;
  const int Bp = (Bwin + 15) / 16 * 16;
  P.A = dG; P.B = W; P.C = dX;
  P.M = Bp * T; P.N = Cin; P.K = N4H;
  P.ldb = Cin;
  P.accumulate = accumulate;
  set_split(P, 1);
  return launch<A_T16, B_KN, C_T16>(P, stream);
}

// dW[4H][F] = sum over window rows m' of dG[m'][:]^T . X[window row(m', tshift)][:F]
// columns [0, F) of a weight gradient whose rows are ldc apart; X points at the first of these columns
static int wgrad_t16_cols(const float* dG, const float* X, float* dW, float* slabs, int ksplit, int Bwin, int T, int F,
                          long ldx, long ldc, int N4H, int tshift, int accumulate, int mode, hipStream_t stream) {
  GemmParams P{};
  P.mode = mode;
  const int Bp = (Bwin + 15) / 16 * 16;
  P.A = dG; P.B = X;
  P.M = N4H; P.N = F; P.K = Bp * T;
  P.ldb = ldx; P.ldc = ldc;
  P.b_map = 1; P.T = T; P.Bwin = Bwin; P.b_tshift = tshift;
  set_split(P, ksplit);
  P.big_split = split_plan_big(P.M, P.N, P.K, mode);
  P.C = (P.ksplit > 1) ? slabs : dW;
  if (P.ksplit > 1) P.ldc = F;          // slabs are dense [M][F]
  P.accumulate = (P.ksplit > 1) ? 0 : accumulate;
  int rc = launch<A_T16T, B_KN, C_ROW>(P, stream);
  if (rc) return rc;
  return finish_split(P, dW, ldc, accumulate, stream);
}

// Where the 256 x 128 plan of a weight gradient is cut in two along its columns, or 0.  That plan runs one
// workgroup per CU, so its tile count x k-slices has to come out near the CU count with the same number of slices
// on every XCD: 4H = 1024 rows x 256 columns is 8 tiles x 32 slices = 256 workgroups, but 370 columns are 12 tiles
// x 16 slices = 192 (a quarter of the chip idle; 24 slices would start a second round).  Cutting at the last
// multiple of 256 gives 8 tiles x 32 slices plus 4 tiles x 64 slices: two full launches.
int vts_gemm_wgrad_cut(int M, int N, long K, int mode) {
  if (!split_plan_big(M, N, K, mode) || N <= 256 || (N % 256) == 0) return 0;
  return N / 256 * 256;
}

int vts_gemm_wgrad_t16(const float* dG, const float* X, float* dW, float* slabs, int ksplit, int Bwin, int T,
                       int F, long ldx, int N4H, int tshift, int accumulate, int mode, hipStream_t stream) {
  const long K = (long)((Bwin + 15) / 16 * 16) * T;
  const int cut = vts_gemm_wgrad_cut(N4H, F, K, mode);
  if (cut == 0)
    return wgrad_t16_cols(dG, X, dW, slabs, ksplit, Bwin, T, F, ldx, F, N4H, tshift, accumulate, mode, stream);
  // the two launches run one after the other on the stream and share the slab buffer
  int rc = wgrad_t16_cols(dG, X, dW, slabs, vts_gemm_pick_ksplit(N4H, cut, K, mode), Bwin, T, cut, ldx, F, N4H, tshift,
                          accumulate, mode, stream);
  if (rc) return rc;
  return wgrad_t16_cols(dG, X + cut, dW + cut, slabs, vts_gemm_pick_ksplit(N4H, F - cut, K, mode), Bwin, T, F - cut, ldx,
                        F, N4H, tshift, accumulate, mode, stream);
}

// Both weight gradients of a layer whose input is H wide, dW_ih = dG^T . X and dW_hh = dG^T . Hprev (the h
// sequence one step back), as one launch of the 256 x 128 kernel: 16 column tiles x 16 k-slices instead of twice
// 8 x 32, i.e. k loops twice as long per workgroup, half the slab traffic and dG read by one launch.
// A wider input (the first layer's 370 channels) pairs its first 256 columns with the recurrent product and
// sends the rest through a launch of their own (4 column tiles x 64 k-slices for 114 columns).
bool vts_gemm_wgrad_pair_usable(int N4H, int F, int Hh, long K, int mode) {
  return mode == 1 && F >= 256 && Hh == 256 && split_plan_big(N4H, 256 + Hh, K, mode);
}
size_t vts_gemm_wgrad_pair_slab_bytes(int N4H, int F, int Hh, long K, int mode) {
  const size_t pair = vts_gemm_slab_bytes(N4H, 256 + Hh, vts_gemm_pick_ksplit(N4H, 256 + Hh, K, mode));
  const size_t rest = F > 256 ? vts_gemm_slab_bytes(N4H, F - 256, vts_gemm_pick_ksplit(N4H, F - 256, K, mode)) : 0;
  return pair > rest ? pair : rest;     // the launches follow each other on the stream and share the buffer
}
int vts_gemm_wgrad_pair_t16(const float* dG, const float* X, long ldx, const float* Hs, float* dWih, float* dWhh,
                            float* slabs, int Bwin, int T, int F, int Hh, int N4H, int accumulate, int mode,
                            hipStream_t stream) {
  constexpr int F1 = 256;                // columns of X in the paired launch
  GemmParams P{};
  P.mode = mode;
  const int Bp = (Bwin + 15) / 16 * 16;
  P.A = dG; P.B = X; P.B2 = Hs;
  P.M = N4H; P.K = Bp * T;
  P.N1 = F1; P.N2 = Hh; P.n_cut = F1; P.N = P.n_cut + Hh;
  P.ldb = ldx; P.ldb2 = Hh;
  P.b_map = 1; P.T = T; P.Bwin = Bwin; P.b_tshift = 0; P.b_tshift2 = -1;
  set_split(P, vts_gemm_pick_ksplit(N4H, F1 + Hh, P.K, mode));
  if (P.ksplit <= 1) {
    vts_set_error("gemm: the paired weight gradient needs a split-K plan");
    return VTS_ERR_ARG;
  }
  P.big_split = 1;
  P.C = slabs;
  P.C2 = slabs + (size_t)P.ksplit * N4H * F1;
  int rc = launch<A_T16T, B_KN, C_ROW>(P, stream);
  if (rc) return rc;
  GemmParams R = P;
  R.N = F1;
  rc = finish_split(R, dWih, F, accumulate, stream);
  if (rc) return rc;
  R.C = P.C2; R.N = Hh;
  rc = finish_split(R, dWhh, Hh, accumulate, stream);
  if (rc || F == F1) return rc;
  return wgrad_t16_cols(dG, X + F1, dWih + F1, slabs, vts_gemm_pick_ksplit(N4H, F - F1, P.K, mode), Bwin, T, F - F1, ldx,
                        F, N4H, 0, accumulate, mode, stream);
}

// Y[M][N] = act(X[M][K] . W[N][K]^T + bias)
int vts_gemm_linear_fwd(const float* X, long ldx, const float* W, const float* bias, float* Y, long ldy, int M,
                        int N, int K, int act, const float* keep, float keep_scale, unsigned long long drop_seed,
                        unsigned drop_thresh, float* aux, unsigned long long* bits_out, int mode,
                        hipStream_t stream) {
  GemmParams P{};
  P.mode = mode;
  P.bits_out = bits_out;
  P.drop_seed = drop_seed; P.drop_thresh = drop_thresh;
  P.A = X; P.B = W; P.C = Y;
  P.M = M; P.N = N; P.K = K;
  P.lda = ldx; P.ldb = K; P.ldc = ldy;
  P.bias0 = bias; P.act = act; P.keep = keep; P.keep_scale = keep_scale; P.aux = aux;
  set_split(P, 1);
  return launch<A_ROW, B_NK, C_ROW>(P, stream);
}

// dX[M][K] (+)= dY[M][N] . W[N][K]
int vts_gemm_linear_dx(const float* dY, long lddy, const float* W, float* dX, long lddx, int M, int N, int K,
                       int accumulate, const float* mask_src, long ld_mask, const unsigned long long* mask_bits,
                       float mask_scale, float* colpart, int mode, hipStream_t stream) {
  GemmParams P{};
  P.mode = mode;
  P.mask_src = mask_bits ? nullptr : mask_src; P.ld_mask = ld_mask; P.mask_bits = mask_bits; P.mask_scale = mask_scale;
  P.colpart = colpart;
  P.A = dY; P.B = W; P.C = dX;
  P.M = M; P.N = K; P.K = N;
  P.lda = lddy; P.ldb = K; P.ldc = lddx;
  P.accumulate = accumulate;
  set_split(P, 1);
  return launch<A_ROW, B_KN, C_ROW>(P, stream);
}

// dW[N][K] (+)= dY[M][N]^T . X[M][K]
int vts_gemm_linear_dw(const float* dY, long lddy, const float* X, long ldx, float* dW, float* slabs, int ksplit,
                       int M, int N, int K, int accumulate, int mode, hipStream_t stream) {
  GemmParams P{};
  P.mode = mode;
  P.A = dY; P.B = X;
  P.M = N; P.N = K; P.K = M;
  P.lda = lddy; P.ldb = ldx; P.ldc = K;
  set_split(P, ksplit);
  P.C = (P.ksplit > 1) ? slabs : dW;
  P.accumulate = (P.ksplit > 1) ? 0 : accumulate;
  int rc = launch<A_COL, B_KN, C_ROW>(P, stream);
  if (rc) return rc;
  return finish_split(P, dW, K, accumulate, stream);
}
