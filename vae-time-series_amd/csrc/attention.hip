// Multi-head self-attention of nn.TransformerEncoderLayer (forward and backward) and the
// residual-dropout sites around it, for short sequences and narrow heads (T ~ 100, head width 8 or
// 16): the attention of vae-models/mixture-vae-transformer.py:39-40,49.
//
// One wave owns one (window, head) pair.  Its K and V rows (T x DH floats each) live in LDS, a lane
// owns a query row (forward, dQ pass) or a key row (dK/dV pass) and walks the other index
// sequentially, so every LDS read is a broadcast and no cross-lane reduction is needed.  Scores are
// never written to memory: forward keeps only the row log-sum-exp, backward recomputes the weights
// from it twice (once per ownership), which is cheaper than a T x T round trip through HBM and
// needs no atomics.  The contraction length is the head width (8), too short for the matrix cores to
// pay for their operand shuffles: this is VALU work.
#include "vts_common.h"

namespace {

struct MhaParams {
  const float* q;     // row b*Tq + i, head h at column h*DH; row stride ldq
  const float* k;     // row b*T + j
  const float* v;
  long ldq, ldkv;
  float* out;         // [B*Tq][ldo]
  long ldo;
  float* lse;         // [B*nhead][Tq]  max + log(sum exp) of the scaled scores
  VtsDrop drop;       // attention-weight dropout, element index ((b*nhead+h)*Tq + i)*T + j
  int B, T, Tq, nhead;
  float scale;        // 1/sqrt(DH)
  // backward only
  const float* dout;  // [B*Tq][lddo]
  long lddo;
  float* dq; long lddq;
  float* dk; float* dv; long lddkv;
};

template <int DH>
__device__ __forceinline__ void load_row(const float* p, float (&r)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(p + c);
    r[c] = x[0]; r[c + 1] = x[1]; r[c + 2] = x[2]; r[c + 3] = x[3];
  }
}
template <int DH>
__device__ __forceinline__ void store_row(float* p, const float (&r)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) *reinterpret_cast<f32x4*>(p + c) = f32x4{r[c], r[c + 1], r[c + 2], r[c + 3]};
}
template <int DH>
__device__ __forceinline__ float dot(const float (&a)[DH], const float* b) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(b + c);
    s = fmaf(a[c], x[0], s); s = fmaf(a[c + 1], x[1], s); s = fmaf(a[c + 2], x[2], s); s = fmaf(a[c + 3], x[3], s);
  }
  return s;
}
template <int DH>
__device__ __forceinline__ void axpy(float a, const float* x, float (&y)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(x + c);
    y[c] = fmaf(a, t[0], y[c]); y[c + 1] = fmaf(a, t[1], y[c + 1]);
    y[c + 2] = fmaf(a, t[2], y[c + 2]); y[c + 3] = fmaf(a, t[3], y[c + 3]);
  }
}

// Copy `rows` rows of DH floats (row stride ld in memory) into a dense [rows][DH] LDS block.
template <int DH>
__device__ __forceinline__ void stage_rows(const float* src, long ld, int rows, float* dst, int lane) {
  constexpr int PER = DH / 4;
  for (int c = lane; c < rows * PER; c += 64) {
    const int row = c / PER, part = c % PER;
    *reinterpret_cast<f32x4*>(dst + row * DH + part * 4) =
        *reinterpret_cast<const f32x4*>(src + (long)row * ld + part * 4);
  }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const MhaParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * 4 + wave;
  const bool active = item < (long)P.B * P.nhead;
  const int b = active ? (int)(item / P.nhead) : 0, h = active ? (int)(item % P.nhead) : 0;
  float* Ks = lds + (long)wave * 2 * P.T * DH;
  float* Vs = Ks + (long)P.T * DH;
  if (active) {
    stage_rows<DH>(P.k + (long)b * P.T * P.ldkv + h * DH, P.ldkv, P.T, Ks, lane);
    stage_rows<DH>(P.v + (long)b * P.T * P.ldkv + h * DH, P.ldkv, P.T, Vs, lane);
  }
  __syncthreads();
  if (!active) return;
  for (int i = lane; i < P.Tq; i += 64) {
    float q[DH];
    load_row<DH>(P.q + ((long)b * P.Tq + i) * P.ldq + h * DH, q);
#pragma unroll
    for (int c = 0; c < DH; ++c) q[c] *= P.scale;
    float m = -INFINITY;
    for (int j = 0; j < P.T; ++j) m = fmaxf(m, dot<DH>(q, Ks + j * DH));
    float l = 0.f, acc[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] = 0.f;
    const unsigned long long mbase = ((unsigned long long)item * P.Tq + i) * P.T;
    // keys two at a time where the mask index of the first is even: one hash draws both keep decisions
    int j = 0;
    if ((mbase & 1) == 0) {
      for (; j + 1 < P.T; j += 2) {
        float m0, m1;
        P.drop.mult2(mbase + j, m0, m1);
        const float p0 = __expf(dot<DH>(q, Ks + j * DH) - m);
        const float p1 = __expf(dot<DH>(q, Ks + (j + 1) * DH) - m);
        l += p0;
        axpy<DH>(p0 * m0, Vs + j * DH, acc);
        l += p1;
        axpy<DH>(p1 * m1, Vs + (j + 1) * DH, acc);
      }
    }
    for (; j < P.T; ++j) {
      const float p = __expf(dot<DH>(q, Ks + j * DH) - m);
      l += p;
      axpy<DH>(p * P.drop.mult(mbase + j), Vs + j * DH, acc);
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int c = 0; c < DH; ++c) acc[c] *= inv;
    store_row<DH>(P.out + ((long)b * P.Tq + i) * P.ldo + h * DH, acc);
    P.lse[item * P.Tq + i] = m + __logf(l);
  }
}

// Backward.  With W = softmax(S), Wd = W * mask/(1-p), O = Wd V:
//   dV_j = sum_i Wd_ij dO_i ;  dWd_ij = dO_i . V_j ;  dS_ij = W_ij * (mask_ij/(1-p) * dWd_ij - D_i),
//   D_i = sum_j Wd_ij dWd_ij = dO_i . O_i ;  dQ_i = scale * sum_j dS_ij K_j ;  dK_j = scale * sum_i dS_ij Q_i.
template <int DH>
__global__ __launch_bounds__(256) void mha_bwd_kernel(const MhaParams P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * 4 + wave;
  const bool active = item < (long)P.B * P.nhead;
  const int b = active ? (int)(item / P.nhead) : 0, h = active ? (int)(item % P.nhead) : 0;
  const int per_wave = 2 * P.T * DH + 2 * P.Tq * DH + 2 * P.Tq;
  float* Ks = lds + (long)wave * per_wave;
  float* Vs = Ks + (long)P.T * DH;
  float* Qs = Vs + (long)P.T * DH;
  float* Gs = Qs + (long)P.Tq * DH;     // dO rows
  float* Ls = Gs + (long)P.Tq * DH;     // log-sum-exp per query
  float* Ds = Ls + P.Tq;                // D_i
  if (active) {
    stage_rows<DH>(P.k + (long)b * P.T * P.ldkv + h * DH, P.ldkv, P.T, Ks, lane);
    stage_rows<DH>(P.v + (long)b * P.T * P.ldkv + h * DH, P.ldkv, P.T, Vs, lane);
    stage_rows<DH>(P.q + (long)b * P.Tq * P.ldq + h * DH, P.ldq, P.Tq, Qs, lane);
    stage_rows<DH>(P.dout + (long)b * P.Tq * P.lddo + h * DH, P.lddo, P.Tq, Gs, lane);
    for (int i = lane; i < P.Tq; i += 64) {
      float g[DH];
      load_row<DH>(P.dout + ((long)b * P.Tq + i) * P.lddo + h * DH, g);
      Ds[i] = dot<DH>(g, P.out + ((long)b * P.Tq + i) * P.ldo + h * DH);
      Ls[i] = P.lse[item * P.Tq + i];
    }
  }
  __syncthreads();
  if (!active) return;
  const unsigned long long mitem = (unsigned long long)item * P.Tq;
  // queries own lanes: dQ
  for (int i = lane; i < P.Tq; i += 64) {
    float q[DH], g[DH], dq[DH];
    load_row<DH>(Qs + i * DH, q);
    load_row<DH>(Gs + i * DH, g);
#pragma unroll
    for (int c = 0; c < DH; ++c) { q[c] *= P.scale; dq[c] = 0.f; }
    const float lse = Ls[i], D = Ds[i];
    const unsigned long long mrow = (mitem + i) * P.T;
    int j = 0;
    if ((mrow & 1) == 0) {                         // as in the forward kernel: one hash per two keys
      for (; j + 1 < P.T; j += 2) {
        float m0, m1;
        P.drop.mult2(mrow + j, m0, m1);
        const float p0 = __expf(dot<DH>(q, Ks + j * DH) - lse);
        const float ds0 = p0 * (m0 * dot<DH>(g, Vs + j * DH) - D);
        axpy<DH>(ds0, Ks + j * DH, dq);
        const float p1 = __expf(dot<DH>(q, Ks + (j + 1) * DH) - lse);
        const float ds1 = p1 * (m1 * dot<DH>(g, Vs + (j + 1) * DH) - D);
        axpy<DH>(ds1, Ks + (j + 1) * DH, dq);
      }
    }
    for (; j < P.T; ++j) {
      const float p = __expf(dot<DH>(q, Ks + j * DH) - lse);
      const float dwd = dot<DH>(g, Vs + j * DH);
      const float ds = p * (P.drop.mult(mrow + j) * dwd - D);
      axpy<DH>(ds, Ks + j * DH, dq);
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] *= P.scale;
    store_row<DH>(P.dq + ((long)b * P.Tq + i) * P.lddq + h * DH, dq);
  }
  // keys own lanes: dK, dV
  for (int j = lane; j < P.T; j += 64) {
    float k[DH], v[DH], dk[DH], dv[DH];
    load_row<DH>(Ks + j * DH, k);
    load_row<DH>(Vs + j * DH, v);
#pragma unroll
    for (int c = 0; c < DH; ++c) { k[c] *= P.scale; dk[c] = 0.f; dv[c] = 0.f; }
    for (int i = 0; i < P.Tq; ++i) {
      const float p = __expf(dot<DH>(k, Qs + i * DH) - Ls[i]);
      const float mult = P.drop.mult((mitem + i) * P.T + j);
      const float dwd = dot<DH>(v, Gs + i * DH);
      axpy<DH>(p * mult, Gs + i * DH, dv);
      axpy<DH>(p * (mult * dwd - Ds[i]), Qs + i * DH, dk);
    }
#pragma unroll
    for (int c = 0; c < DH; ++c) dk[c] *= P.scale;
    store_row<DH>(P.dk + ((long)b * P.T + j) * P.lddkv + h * DH, dk);
    store_row<DH>(P.dv + ((long)b * P.T + j) * P.lddkv + h * DH, dv);
  }
}

// One query per window (Tq == 1, the last-position-only layer): lanes own keys, the softmax
// statistics and the DH-wide sums are wave reductions, nothing is staged in LDS.
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <int DH>
__global__ __launch_bounds__(256) void mha_fwd_one_kernel(const MhaParams P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * 4 + wave;
  if (item >= (long)P.B * P.nhead) return;
  const int b = (int)(item / P.nhead), h = (int)(item % P.nhead);
  float q[DH];
  load_row<DH>(P.q + (long)b * P.ldq + h * DH, q);
#pragma unroll
  for (int c = 0; c < DH; ++c) q[c] *= P.scale;
  const float* kb = P.k + (long)b * P.T * P.ldkv + h * DH;
  const float* vb = P.v + (long)b * P.T * P.ldkv + h * DH;
  float m = -INFINITY;
  for (int j = lane; j < P.T; j += 64) m = fmaxf(m, dot<DH>(q, kb + (long)j * P.ldkv));
  m = wave_max_f(m);
  float l = 0.f, acc[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] = 0.f;
  for (int j = lane; j < P.T; j += 64) {
    const float p = __expf(dot<DH>(q, kb + (long)j * P.ldkv) - m);
    l += p;
    axpy<DH>(p * P.drop.mult((unsigned long long)item * P.T + j), vb + (long)j * P.ldkv, acc);
  }
  l = wave_sum_f(l);
  const float inv = 1.f / l;
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] = wave_sum_f(acc[c]) * inv;
  if (lane == 0) {
    store_row<DH>(P.out + (long)b * P.ldo + h * DH, acc);
    P.lse[item] = m + __logf(l);
  }
}

template <int DH>
__global__ __launch_bounds__(256) void mha_bwd_one_kernel(const MhaParams P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * 4 + wave;
  if (item >= (long)P.B * P.nhead) return;
  const int b = (int)(item / P.nhead), h = (int)(item % P.nhead);
  float q[DH], g[DH], dq[DH];
  load_row<DH>(P.q + (long)b * P.ldq + h * DH, q);
  load_row<DH>(P.dout + (long)b * P.lddo + h * DH, g);
  const float D = dot<DH>(g, P.out + (long)b * P.ldo + h * DH);
  const float lse = P.lse[item];
#pragma unroll
  for (int c = 0; c < DH; ++c) { q[c] *= P.scale; dq[c] = 0.f; }
  const long kvrow = (long)b * P.T;
  for (int j = lane; j < P.T; j += 64) {
    const float* kj = P.k + (kvrow + j) * P.ldkv + h * DH;
    const float* vj = P.v + (kvrow + j) * P.ldkv + h * DH;
    const float p = __expf(dot<DH>(q, kj) - lse);
    const float mult = P.drop.mult((unsigned long long)item * P.T + j);
    const float ds = p * (mult * dot<DH>(g, vj) - D);
    axpy<DH>(ds, kj, dq);
    float dk[DH], dv[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) { dk[c] = ds * q[c]; dv[c] = p * mult * g[c]; }   // q already carries the scale
    store_row<DH>(P.dk + (kvrow + j) * P.lddkv + h * DH, dk);
    store_row<DH>(P.dv + (kvrow + j) * P.lddkv + h * DH, dv);
  }
#pragma unroll
  for (int c = 0; c < DH; ++c) dq[c] = wave_sum_f(dq[c]) * P.scale;
  if (lane == 0) store_row<DH>(P.dq + (long)b * P.lddq + h * DH, dq);
}

// out = residual + dropout(x)   /   dx = dropout mask applied to dout   /   the mask itself
__global__ void dropout_add_kernel(const float* __restrict__ residual, const float* __restrict__ x, VtsDrop drop,
                                   float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = (residual ? residual[i] : 0.f) + x[i] * drop.mult((unsigned long long)i);
}
__global__ void dropout_mask_kernel(unsigned long long seed, unsigned thresh, unsigned char* __restrict__ keep, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    keep[i] = vts_drop_keep(seed, (unsigned long long)i, thresh) ? 1 : 0;
}

int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

template <int DH>
int launch_mha(const MhaParams& P, bool bwd, hipStream_t st) {
  if (P.Tq == 1) {
    const unsigned grid1 = (unsigned)(((long)P.B * P.nhead + 3) / 4);
    if (bwd) hipLaunchKernelGGL(mha_bwd_one_kernel<DH>, dim3(grid1), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(mha_fwd_one_kernel<DH>, dim3(grid1), dim3(256), 0, st, P);
    VTS_CHECK_LAUNCH("mha_one");
    return VTS_OK;
  }
  const size_t per_wave = bwd ? (size_t)2 * P.T * DH + (size_t)2 * P.Tq * DH + 2 * P.Tq : (size_t)2 * P.T * DH;
  const size_t bytes = 4 * per_wave * sizeof(float);
  if (bytes > 160 * 1024) {
    vts_set_error("mha: T %d x head width %d needs %zu B of LDS (limit 160 KiB)", P.T, DH, bytes);
    return VTS_ERR_UNSUPPORTED;
  }
  const unsigned grid = (unsigned)(((long)P.B * P.nhead + 3) / 4);
  // the LDS request varies with T, so the opt-in above 64 KiB is renewed per launch rather than cached
  vts_lds_mask fresh{0};
  if (bwd) {
    if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(mha_bwd_kernel<DH>), bytes, &fresh, "mha_bwd")) return rc;
    hipLaunchKernelGGL(mha_bwd_kernel<DH>, dim3(grid), dim3(256), bytes, st, P);
    VTS_CHECK_LAUNCH("mha_bwd");
  } else {
    if (int rc = vts_reserve_lds(reinterpret_cast<const void*>(mha_fwd_kernel<DH>), bytes, &fresh, "mha_fwd")) return rc;
    hipLaunchKernelGGL(mha_fwd_kernel<DH>, dim3(grid), dim3(256), bytes, st, P);
    VTS_CHECK_LAUNCH("mha_fwd");
  }
  return VTS_OK;
}

}  // namespace

int vts_attn_mha(int bwd, const float* q, long ldq, const float* k, const float* v, long ldkv, float* out, long ldo,
                 float* lse, const unsigned char* keep, float drop_p, unsigned long long seed, const float* dout,
                 long lddo, float* dq, long lddq, float* dk, float* dv, long lddkv, int B, int T, int Tq, int nhead,
                 int dh, hipStream_t st) {
  MhaParams P{};
  P.q = q; P.k = k; P.v = v; P.ldq = ldq; P.ldkv = ldkv;
  P.out = out; P.ldo = ldo; P.lse = lse;
  P.drop = vts_make_drop(keep, drop_p, seed);
  P.B = B; P.T = T; P.Tq = Tq; P.nhead = nhead;
  P.scale = 1.0f / sqrtf((float)dh);
  P.dout = dout; P.lddo = lddo; P.dq = dq; P.lddq = lddq; P.dk = dk; P.dv = dv; P.lddkv = lddkv;
  if (dh == 8) return launch_mha<8>(P, bwd != 0, st);
  if (dh == 16) return launch_mha<16>(P, bwd != 0, st);
  if (dh == 32) return launch_mha<32>(P, bwd != 0, st);
  vts_set_error("mha: head width %d not supported (8, 16, 32)", dh);
  return VTS_ERR_UNSUPPORTED;
}

int vts_attn_dropout_add(const float* residual, const float* x, const unsigned char* keep, float drop_p,
                         unsigned long long seed, float* out, long n, hipStream_t st) {
  hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for(n)), dim3(256), 0, st, residual, x,
                     vts_make_drop(keep, drop_p, seed), out, n);
  VTS_CHECK_LAUNCH("dropout_add");
  return VTS_OK;
}

int vts_attn_dropout_mask(unsigned long long seed, float drop_p, unsigned char* keep, long n, hipStream_t st) {
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, st, seed, vts_drop_thresh(drop_p), keep, n);
  VTS_CHECK_LAUNCH("dropout_mask");
  return VTS_OK;
}
