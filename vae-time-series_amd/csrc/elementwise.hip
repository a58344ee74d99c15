// HBM-bound pieces of the train step: reparameterisation (Gaussian / GPD / mixture row select),
// the three ELBO losses with their gradients, LayerNorm, activation gradients, column sums and
// the Adam update.  All fp32, 16-byte accesses where the shapes allow, wave64 shuffle reductions.
#include "vts_common.h"
#include "vts_gpd_pow.h"
#include <initializer_list>
#include <mutex>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Block-wide sum of NV values per thread; result valid in thread 0.  blockDim.x == 256.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem /* [4*NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) smem[wave * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = smem[i] + smem[NV + i] + smem[2 * NV + i] + smem[3 * NV + i];
  }
}

// ------------------------------------------------------------------------------ reparameterise
// softmax(pi)[0] the way torch computes a 2-element fp32 softmax (subtract max, exp, normalise).
__device__ __forceinline__ float softmax0(const float* pi) {
  const float a = pi[0], b = pi[1];
  const float m = fmaxf(a, b);
  const float ea = expf(a - m), eb = expf(b - m);
  return ea / (ea + eb);
}

// GPD inverse-CDF sample, same operation order as the reference expression
// scale / shape * ((1 - u) ** (-shape) - 1).  The power is evaluated in fp64 and rounded once:
// near the shape clamp (1e-6) the result sits a few ulps above 1 and the subtraction that
// follows amplifies any last-bit difference by ~1e6, so the power must be correctly rounded
// (torch's CPU fp32 pow is, in that regime); vts_gpd_pow.h holds the short fp64 evaluation.
__device__ __forceinline__ float gpd_pow(float base, float expo) { return vts_gpd_pow(base, expo); }

// mode 0: z = mean + eps*exp(0.5*logvar)
// mode 1: z = GPD sample
// mode 2: z = row-select(choice < softmax(pi)[0] ? gaussian : gpd)
__global__ void reparam_fwd_kernel(int mode, const float* __restrict__ mean, const float* __restrict__ logvar,
                                   const float* __restrict__ scale, const float* __restrict__ shape,
                                   const float* __restrict__ eps, const float* __restrict__ u,
                                   const float* __restrict__ choice, const float* __restrict__ pi,
                                   float* __restrict__ z, long n, int L) {
  float p0 = 0.f;
  if (mode == 2) p0 = softmax0(pi);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    bool gauss = (mode == 0);
    if (mode == 2) gauss = choice[i / L] < p0;
    float out;
    if (gauss) {
      out = mean[i] + eps[i] * expf(0.5f * logvar[i]);
    } else {
      const float sc = scale[i], sh = shape[i];
      const float p = gpd_pow(1.0f - u[i], -sh);
      out = (sc / sh) * (p - 1.0f);
    }
    z[i] = out;
  }
}

// Gradients w.r.t. mean, logvar, scale, shape given dz (unselected family gets 0 in mode 2).
// GPD branch mirrors autograd's decomposition of scale/shape*((1-u)**(-shape)-1):
//   a = scale/shape, q = p-1, z = a*q;  da = dz*q, dq = dz*a
//   dscale = da/shape,  dshape = -da*(a/shape) - dq*p*log(1-u)
__global__ void reparam_bwd_kernel(int mode, const float* __restrict__ dz, const float* __restrict__ logvar,
                                   const float* __restrict__ scale, const float* __restrict__ shape,
                                   const float* __restrict__ eps, const float* __restrict__ u,
                                   const float* __restrict__ choice, const float* __restrict__ pi,
                                   float* __restrict__ dmean, float* __restrict__ dlogvar,
                                   float* __restrict__ dscale, float* __restrict__ dshape, long n, int L) {
  float p0 = 0.f;
  if (mode == 2) p0 = softmax0(pi);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    bool gauss = (mode == 0);
    if (mode == 2) gauss = choice[i / L] < p0;
    const float d = dz[i];
    if (mode != 1) {
      float dm = 0.f, dl = 0.f;
      if (gauss) {
        dm = d;
        dl = d * eps[i] * expf(0.5f * logvar[i]) * 0.5f;
      }
      dmean[i] = dm;
      dlogvar[i] = dl;
    }
    if (mode != 0) {
      float dsc = 0.f, dsh = 0.f;
      if (!gauss) {
        const float sc = scale[i], sh = shape[i];
        const float base = 1.0f - u[i];
        const float p = gpd_pow(base, -sh);
        const float a = sc / sh;
        const float q = p - 1.0f;
        const float da = d * q, dq = d * a;
        dsc = da / sh;
        const float de = dq * (p * logf(base));   // d/d(exponent); exponent = -shape
        dsh = -da * (a / sh) - de;
      }
      dscale[i] = dsc;
      dshape[i] = dsh;
    }
  }
}

// ----------------------------------------------------------- fused sampling + KL (one HBM pass)
// Forward: z as reparam_fwd_kernel computes it, plus this block's share of the two KL sums
//   part[blk][0] = sum 1 + lv - mu^2 - e^lv      part[blk][1] = sum log(scale) + log(1 + shape*thr/scale)
// so mean / logvar / scale / shape are read once for both.  Backward: the sampling gradient
// (from dz) and the KL gradient (from d loss / d part) in one pass, written once.
// Elements are owned in chunks of 1024: chunk c belongs to block c % gridDim.x in both kernels and
// for both access widths, so dpart[blk] is the coefficient of exactly the elements summed into
// part[blk].  V = 4: 16-byte accesses (L % 4 == 0, so the four elements share a row).
// In the mixture a row reads only the noise of the family it selected (eps or u).
template <int V> struct VecT { using type = float; };
template <> struct VecT<4> { using type = f32x4; };

template <int V>
__device__ __forceinline__ void ld_vec(float (&d)[V], const float* p) {
  if constexpr (V == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  } else {
    d[0] = *p;
  }
}
template <int V>
__device__ __forceinline__ void st_vec(float* p, const float (&d)[V]) {
  if constexpr (V == 4) *reinterpret_cast<f32x4*>(p) = f32x4{d[0], d[1], d[2], d[3]};
  else *p = d[0];
}

template <int MODE, int V>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(
    const float* __restrict__ mean, const float* __restrict__ logvar, const float* __restrict__ scale,
    const float* __restrict__ shape, const float* __restrict__ eps, const float* __restrict__ u,
    const float* __restrict__ choice, const float* __restrict__ pi, float thr, float* __restrict__ z,
    float* __restrict__ part, long n, int L) {
  __shared__ float smem[8];
  float p0 = 0.f;
  if (MODE == 2) p0 = softmax0(pi);
  float s[2] = {0.f, 0.f};
  const long nchunk = (n + 1023) >> 10;
  for (long c = blockIdx.x; c < nchunk; c += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 4 / V; ++j) {
      const long e = (c << 10) + (V == 4 ? threadIdx.x * 4 : j * 256 + threadIdx.x);
      if (e >= n) continue;
      bool gauss = (MODE == 0);
      if (MODE == 2) gauss = choice[e / L] < p0;
      float mu[V], lv[V], sc[V], sh[V], nz[V], out[V];
      if (MODE != 1) { ld_vec<V>(mu, mean + e); ld_vec<V>(lv, logvar + e); }
      if (MODE != 0) { ld_vec<V>(sc, scale + e); ld_vec<V>(sh, shape + e); }
      ld_vec<V>(nz, (gauss ? eps : u) + e);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        if (MODE != 1) s[0] += 1.f + lv[k] - mu[k] * mu[k] - expf(lv[k]);
        if (MODE != 0) s[1] += logf(sc[k]) + logf(1.f + sh[k] * thr / sc[k]);
        if (gauss) {
          out[k] = mu[k] + nz[k] * expf(0.5f * lv[k]);
        } else {
          const float p = gpd_pow(1.0f - nz[k], -sh[k]);
          out[k] = (sc[k] / sh[k]) * (p - 1.0f);
        }
      }
      st_vec<V>(z + e, out);
    }
  }
  block_sum<2>(s, smem);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = s[0]; part[blockIdx.x * 2 + 1] = s[1]; }
}

template <int MODE, int V>
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ dpart, const float* __restrict__ mean,
    const float* __restrict__ logvar, const float* __restrict__ scale, const float* __restrict__ shape,
    const float* __restrict__ eps, const float* __restrict__ u, const float* __restrict__ choice,
    const float* __restrict__ pi, float thr, float* __restrict__ dmean, float* __restrict__ dlogvar,
    float* __restrict__ dscale, float* __restrict__ dshape, long n, int L) {
  float p0 = 0.f;
  if (MODE == 2) p0 = softmax0(pi);
  const float c_klg = dpart ? dpart[blockIdx.x * 2] : 0.f;
  const float c_klp = dpart ? dpart[blockIdx.x * 2 + 1] : 0.f;
  const long nchunk = (n + 1023) >> 10;
  for (long c = blockIdx.x; c < nchunk; c += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 4 / V; ++j) {
      const long e = (c << 10) + (V == 4 ? threadIdx.x * 4 : j * 256 + threadIdx.x);
      if (e >= n) continue;
      bool gauss = (MODE == 0);
      if (MODE == 2) gauss = choice[e / L] < p0;
      float d[V], mu[V], lv[V], sc[V], sh[V], nz[V];
      if (dz) {
        ld_vec<V>(d, dz + e);
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) d[k] = 0.f;
      }
      if (MODE != 1) { ld_vec<V>(mu, mean + e); ld_vec<V>(lv, logvar + e); }
      if (MODE != 0) { ld_vec<V>(sc, scale + e); ld_vec<V>(sh, shape + e); }
      ld_vec<V>(nz, (gauss ? eps : u) + e);
      float dm[V], dl[V], dsc[V], dsh[V];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        if (MODE != 1) {
          // KL part as elbo_grads_kernel, sampling part as reparam_bwd_kernel
          dm[k] = c_klg * (-2.f * mu[k]);
          dl[k] = c_klg * (1.f - expf(lv[k]));
          if (gauss) {
            dm[k] += d[k];
            dl[k] += d[k] * nz[k] * expf(0.5f * lv[k]) * 0.5f;
          }
        }
        if (MODE != 0) {
          const float w = 1.f + sh[k] * thr / sc[k];
          dsc[k] = c_klp * (1.f / sc[k] - (sh[k] * thr / (sc[k] * sc[k])) / w);
          dsh[k] = c_klp * ((thr / sc[k]) / w);
          if (!gauss) {
            const float base = 1.0f - nz[k];
            const float p = gpd_pow(base, -sh[k]);
            const float a = sc[k] / sh[k];
            const float q = p - 1.0f;
            const float da = d[k] * q, dq = d[k] * a;
            dsc[k] += da / sh[k];
            const float de = dq * (p * logf(base));
            dsh[k] += -da * (a / sh[k]) - de;
          }
        }
      }
      if (MODE != 1) { st_vec<V>(dmean + e, dm); st_vec<V>(dlogvar + e, dl); }
      if (MODE != 0) { st_vec<V>(dscale + e, dsc); st_vec<V>(dshape + e, dsh); }
    }
  }
}

// ------------------------------------------------------------------------------------- ELBO
// kind 0 gaussian, 1 gpd, 2 mixture, 3 mixture whose GPD component is scored on the rows with
// |y| > thr and carries the Gaussian-form KL of (scale, shape) (mixture-vae-transformer.py:100-113).
// Partial sums per block, 6 slots:
//   0: sum (recon-y)^2          1: same over |y| <= thr      2: count |y| <= thr
//   3: sum 1 + lv - mu^2 - e^lv 4: sum log(scale) + log(1 + shape*thr/scale)   (kind 3: sum 1 + sc - sh^2 - e^sc)
//   5: sum (recon-y)^2 over |y| > thr
__global__ __launch_bounds__(256) void elbo_partials_kernel(
    int kind, const float* __restrict__ recon, const float* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ logvar, const float* __restrict__ scale, const float* __restrict__ shape,
    float thr, long nrec, long nlat, float* __restrict__ part) {
  __shared__ float smem[24];
  float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long stride = (long)gridDim.x * blockDim.x;
  const long start = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = start; i < nrec; i += stride) {
    const float yv = y[i];
    const float d = recon[i] - yv;
    const float d2 = d * d;
    s[0] += d2;
    if (kind >= 2) {
      if (fabsf(yv) <= thr) { s[1] += d2; s[2] += 1.f; }
      else s[5] += d2;
    }
  }
  for (long i = start; i < nlat; i += stride) {
    if (kind != 1) {
      const float lv = logvar[i], mu = mean[i];
      s[3] += 1.f + lv - mu * mu - expf(lv);
    }
    if (kind == 3) {
      const float sc = scale[i], sh = shape[i];
      s[4] += 1.f + sc - sh * sh - expf(sc);
    } else if (kind != 0) {
      const float sc = scale[i], sh = shape[i];
      s[4] += logf(sc) + logf(1.f + sh * thr / sc);
    }
  }
  block_sum<6>(s, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) part[(long)blockIdx.x * 6 + i] = s[i];
  }
}

// out[0] = loss; out[1..7] = coefficients used by elbo_grads_kernel; out[8], out[9] = d loss / d pi_params
//   1: c_all   d loss / d recon = c_all * (recon-y) (+ c_norm * (recon-y) on normal rows)
//   2: c_norm  3: c_klg (multiplies d/dx of sum(1+lv-mu^2-e^lv))  4: c_klp
//   5: c_ext (kind 3: extra coefficient of (recon-y) on the rows with |y| > thr)
// klpart / nkl: optional [nkl][2] partial sums of slots 3 and 4 produced by reparam_kl_fwd_kernel.
// Data-parallel form: with sums_out the six sums (and the row count, slot 6) are written out instead of the
// loss, to be added up over ranks by the caller; with sums_in the loss is formed from such totals (B = the
// total row count they carry, nrec / B = fs, nlat / B = L) and every per-row gradient coefficient is
// multiplied by gscale = the number of ranks, whose gradients the caller averages afterwards.
__global__ void elbo_finalize_kernel(int kind, const float* __restrict__ part, int nparts,
                                     const float* __restrict__ klpart, int nkl, const float* __restrict__ pi,
                                     float beta, long B, long nrec, long nlat, float* __restrict__ out,
                                     const float* __restrict__ sums_in, float* __restrict__ sums_out, float gscale) {
  __shared__ float smem[24];
  float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (sums_in == nullptr) {
    for (int p = threadIdx.x; p < nparts; p += blockDim.x)
#pragma unroll
      for (int i = 0; i < 6; ++i) s[i] += part[(long)p * 6 + i];
    for (int p = threadIdx.x; p < nkl; p += blockDim.x) {
      s[3] += klpart[2 * p];
      s[4] += klpart[2 * p + 1];
    }
    block_sum<6>(s, smem);
  }
  if (threadIdx.x != 0) return;
  if (sums_out != nullptr) {
#pragma unroll
    for (int i = 0; i < 6; ++i) sums_out[i] = s[i];
    sums_out[6] = (float)B;
    sums_out[7] = 0.f;
    return;
  }
  if (sums_in != nullptr) {
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] = sums_in[i];
    const long rows = (long)sums_in[6];
    nrec = rows * (nrec / B);
    nlat = rows * (nlat / B);
    B = rows;
  }
  const float R = s[0] / (float)nrec;
  const float klg = -0.5f * s[3] / (float)B;
  const float klp = s[4] / (float)nlat;
  float loss, c_all, c_norm = 0.f, c_klg = 0.f, c_klp = 0.f, c_ext = 0.f, dpi0 = 0.f, dpi1 = 0.f;
  if (kind == 0) {
    loss = R + beta * klg;
    c_all = 2.f / (float)nrec;
    c_klg = beta * -0.5f / (float)B;
  } else if (kind == 1) {
    loss = R + beta * klp;
    c_all = 2.f / (float)nrec;
    c_klp = beta / (float)nlat;
  } else if (kind == 3) {
    const float a = pi[0], b = pi[1];
    const float m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    const float p0 = ea / (ea + eb), p1 = eb / (ea + eb);
    const float next = (float)nrec - s[2];
    const float Rg = s[1] / s[2];  // 0/0 = NaN when either selection is empty, as mse_loss of an empty tensor
    const float Rp = s[5] / next;
    const float klq = -0.5f * s[4] / (float)B;
    const float Ag = Rg + beta * klg;
    const float Ap = Rp + beta * klq;
    loss = R + p0 * Ag + p1 * Ap;
    c_all = 2.f / (float)nrec;
    c_norm = p0 * 2.f / s[2];
    c_ext = p1 * 2.f / next;
    c_klg = p0 * beta * -0.5f / (float)B;
    c_klp = p1 * beta * -0.5f / (float)B;
    const float sg = Ag * p0 + Ap * p1;
    dpi0 = p0 * (Ag - sg);
    dpi1 = p1 * (Ap - sg);
  } else {
    const float a = pi[0], b = pi[1];
    const float m = fmaxf(a, b);
    const float ea = expf(a - m), eb = expf(b - m);
    const float p0 = ea / (ea + eb), p1 = eb / (ea + eb);
    const float Rg = s[1] / s[2];  // 0/0 = NaN when no row is "normal", as mse_loss of an empty tensor
    const float Ag = Rg + beta * klg;
    const float Ap = R + beta * klp;
    loss = R + p0 * Ag + p1 * Ap;
    c_all = (1.f + p1) * 2.f / (float)nrec;
    c_norm = p0 * 2.f / s[2];
    c_klg = p0 * beta * -0.5f / (float)B;
    c_klp = p1 * beta / (float)nlat;
    // softmax backward in the form autograd evaluates it: p_i * (g_i - sum_j p_j g_j)
    const float sg = Ag * p0 + Ap * p1;
    dpi0 = p0 * (Ag - sg);
    dpi1 = p1 * (Ap - sg);
  }
  out[0] = loss; out[1] = c_all * gscale; out[2] = c_norm * gscale; out[3] = c_klg * gscale; out[4] = c_klp * gscale;
  // d loss / d pi_params is a whole-batch quantity that every rank holds in full: averaging leaves it as it is
  out[5] = c_ext * gscale; out[6] = 0.f; out[7] = 0.f; out[8] = dpi0; out[9] = dpi1;
}

// Elementwise ELBO gradients scaled by the upstream scalar *gout (or 1 when gout == nullptr).
__global__ void elbo_grads_kernel(int kind, const float* __restrict__ recon, const float* __restrict__ y,
                                  const float* __restrict__ mean, const float* __restrict__ logvar,
                                  const float* __restrict__ scale, const float* __restrict__ shape,
                                  const float* __restrict__ coef, const float* __restrict__ gout, float thr,
                                  long nrec, long nlat, float* __restrict__ drecon, float* __restrict__ dmean,
                                  float* __restrict__ dlogvar, float* __restrict__ dscale,
                                  float* __restrict__ dshape, float* __restrict__ dpi,
                                  float* __restrict__ dklpart, int nkl) {
  const float go = gout ? gout[0] : 1.f;
  const float c_all = coef[1] * go, c_norm = coef[2] * go, c_klg = coef[3] * go, c_klp = coef[4] * go;
  const float c_ext = coef[5] * go;
  const long stride = (long)gridDim.x * blockDim.x;
  const long start = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (start == 0 && kind >= 2 && dpi) { dpi[0] = coef[8] * go; dpi[1] = coef[9] * go; }
  // d loss / d (KL partial sums): the same two coefficients for every block of the fused kernel
  for (long i = start; i < nkl; i += stride) { dklpart[2 * i] = c_klg; dklpart[2 * i + 1] = c_klp; }
  for (long i = start; i < nrec; i += stride) {
    const float yv = y[i];
    const float d = recon[i] - yv;
    float gr = c_all * d;
    if (kind >= 2) gr += (fabsf(yv) <= thr ? c_norm : c_ext) * d;
    drecon[i] = gr;
  }
  for (long i = start; i < nlat; i += stride) {
    if (kind != 1) {
      const float lv = logvar[i], mu = mean[i];
      dmean[i] = c_klg * (-2.f * mu);
      dlogvar[i] = c_klg * (1.f - expf(lv));
    }
    if (kind == 3) {
      dscale[i] = c_klp * (1.f - expf(scale[i]));
      dshape[i] = c_klp * (-2.f * shape[i]);
    } else if (kind != 0) {
      const float sc = scale[i], sh = shape[i];
      const float w = 1.f + sh * thr / sc;
      dscale[i] = c_klp * (1.f / sc - (sh * thr / (sc * sc)) / w);
      dshape[i] = c_klp * ((thr / sc) / w);
    }
  }
}

// ----------------------------------------------------------------------------------- LayerNorm
// One wave per row; H <= 1024.  Saves mean and rstd per row.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int M, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long)row * ldx;
  float s = 0.f;
  for (int j = lane; j < H; j += 64) s += xr[j];
  const float mu = wave_sum(s) / (float)H;
  float v = 0.f;
  for (int j = lane; j < H; j += 64) { const float d = xr[j] - mu; v += d * d; }
  const float rs = 1.0f / sqrtf(wave_sum(v) / (float)H + eps);
  for (int j = lane; j < H; j += 64) y[(long)row * H + j] = (xr[j] - mu) * rs * gamma[j] + beta[j];
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx (row stride lddx) = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), dxhat = dy * gamma.
// dgamma / dbeta partials per block: part[blk][2][H].
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            long lddx, float* __restrict__ part, int M, int H,
                                                            int rows_per_block) {
  extern __shared__ float sm[];  // [2][H] accumulators
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < 2 * H; j += blockDim.x) sm[j] = 0.f;
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int row = r0 + wave; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + (long)row * ldx;
    const float* dyr = dy + (long)row * H;
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < H; j += 64) {
      const float xh = (xr[j] - mu) * rs;
      const float dxh = dyr[j] * gamma[j];
      s1 += dxh;
      s2 += dxh * xh;
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    for (int j = lane; j < H; j += 64) {
      const float xh = (xr[j] - mu) * rs;
      const float dxh = dyr[j] * gamma[j];
      dx[(long)row * lddx + j] = rs * (dxh - s1 - xh * s2);
      atomicAdd(&sm[j], dyr[j] * xh);      // LDS atomics, 4 waves share the accumulators
      atomicAdd(&sm[H + j], dyr[j]);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * H; j += blockDim.x) part[(long)blockIdx.x * 2 * H + j] = sm[j];
}

// Rows of NC * 64 columns (the encoder trunk's 256): a lane always sees the same NC columns, so the gamma / beta
// gradients accumulate in registers and the four waves of a block meet once, in LDS, at the end (the kernel above
// pays two LDS atomics per element).
template <int NC>
__global__ __launch_bounds__(256) void layernorm_bwd_cols_kernel(const float* __restrict__ dy,
                                                                 const float* __restrict__ x, long ldx,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                                 long lddx, float* __restrict__ part, int M,
                                                                 int rows_per_block) {
  constexpr int H = NC * 64;
  __shared__ float sm[4][2 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[NC], ag[NC], ab[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) { g[k] = gamma[lane + 64 * k]; ag[k] = 0.f; ab[k] = 0.f; }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int row = r0 + wave; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    const float* xr = x + (long)row * ldx;
    const float* dyr = dy + (long)row * H;
    float xh[NC], d[NC], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      xh[k] = (xr[lane + 64 * k] - mu) * rs;
      d[k] = dyr[lane + 64 * k];
      const float dxh = d[k] * g[k];
      s1 += dxh;
      s2 += dxh * xh[k];
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      dx[(long)row * lddx + lane + 64 * k] = rs * (d[k] * g[k] - s1 - xh[k] * s2);
      ag[k] += d[k] * xh[k];
      ab[k] += d[k];
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) { sm[wave][lane + 64 * k] = ag[k]; sm[wave][H + lane + 64 * k] = ab[k]; }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * H; j += blockDim.x)
    part[(long)blockIdx.x * 2 * H + j] = (sm[0][j] + sm[1][j]) + (sm[2][j] + sm[3][j]);
}

// Narrow rows (H = 4 * LPR <= 64, the transformer's d_model): LPR lanes share a row with one float4
// each, so a wave covers 64/LPR rows per load instruction instead of one, the row statistics are
// xor-shuffles inside the lane group, and the gamma/beta gradients accumulate in registers (each lane
// always sees the same four columns) with one cross-lane and one LDS reduction at the end.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_narrow_kernel(const float* __restrict__ x, long ldx,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   float* __restrict__ y, float* __restrict__ mean,
                                                                   float* __restrict__ rstd, int M, float eps) {
  constexpr int H = 4 * LPR, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, slot = lane / LPR;
  const int row = (blockIdx.x * 4 + wave) * RPW + slot;
  const bool valid = row < M;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (valid) v = *reinterpret_cast<const f32x4*>(x + (long)row * ldx + sub * 4);
  const float mu = group_sum<LPR>((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / H);
  const f32x4 d = v - mu;
  const float var = group_sum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / H);
  const float rs = 1.0f / sqrtf(var + eps);
  if (!valid) return;
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + sub * 4);
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + sub * 4);
  *reinterpret_cast<f32x4*>(y + (long)row * H + sub * 4) = d * rs * g4 + b4;
  if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <int LPR>
__global__ __launch_bounds__(256) void layernorm_bwd_narrow_kernel(const float* __restrict__ dy,
                                                                   const float* __restrict__ x, long ldx,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   float* __restrict__ dx, long lddx,
                                                                   float* __restrict__ part, int M,
                                                                   int rows_per_block) {
  constexpr int H = 4 * LPR, RPW = 64 / LPR;
  __shared__ float sm[4][2][H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, slot = lane / LPR;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + sub * 4);
  f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f}, db = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int base = r0 + wave * RPW; base < r1; base += 4 * RPW) {
    const int row = base + slot;
    const bool valid = row < r1;
    f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f}, dyv = xv;
    float mu = 0.f, rs = 0.f;
    if (valid) {
      xv = *reinterpret_cast<const f32x4*>(x + (long)row * ldx + sub * 4);
      dyv = *reinterpret_cast<const f32x4*>(dy + (long)row * H + sub * 4);
      mu = mean[row]; rs = rstd[row];
    }
    const f32x4 xh = (xv - mu) * rs;
    const f32x4 dxh = dyv * g4;
    const float s1 = group_sum<LPR>((dxh[0] + dxh[1]) + (dxh[2] + dxh[3])) * (1.0f / H);
    const f32x4 t = dxh * xh;
    const float s2 = group_sum<LPR>((t[0] + t[1]) + (t[2] + t[3])) * (1.0f / H);
    if (valid) {
      *reinterpret_cast<f32x4*>(dx + (long)row * lddx + sub * 4) = (dxh - s1 - xh * s2) * rs;
      dg += dyv * xh;
      db += dyv;
    }
  }
  // lanes with the same `sub` hold the same columns: fold the row slots, then the four waves
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[e] += __shfl_xor(dg[e], o); db[e] += __shfl_xor(db[e], o); }
  }
  if (slot == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { sm[wave][0][sub * 4 + e] = dg[e]; sm[wave][1][sub * 4 + e] = db[e]; }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * H; j += blockDim.x) {
    const int which = j / H, col = j % H;
    part[(long)blockIdx.x * 2 * H + j] = (sm[0][which][col] + sm[1][which][col]) + (sm[2][which][col] + sm[3][which][col]);
  }
}

static bool ln_narrow_ok(const void* a, const void* b, const void* c, long ld0, long ld1, int H) {
  return (H == 16 || H == 32 || H == 64) && (ld0 % 4) == 0 && (ld1 % 4) == 0 && vts_aligned16(a) && vts_aligned16(b) &&
         vts_aligned16(c);
}

// -------------------------------------------------------------------------- activation gradient
// kind 1 relu: dy*(y>0) | 2 exp: dy*y | 3 clamp: dy*aux | 4 relu+dropout: dy*scale*(y!=0)
__global__ void act_bwd_kernel(int kind, const float* __restrict__ dy, const float* __restrict__ y,
                               const float* __restrict__ aux, float scale, float* __restrict__ dpre, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = dy[i];
    float o;
    if (kind == 1) o = (y[i] > 0.f) ? d : 0.f;
    else if (kind == 2) o = d * y[i];
    else if (kind == 3) o = d * aux[i];
    else o = (y[i] != 0.f) ? d * scale : 0.f;
    dpre[i] = o;
  }
}

// Column sums of a row-major [M][N] matrix over a chunk of rows: part[chunk][N].  A block covers 64
// columns; its four waves take every fourth row of the chunk (each wave reads 256 contiguous bytes
// per row, four rows in flight) and are summed through LDS.
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ a, long lda, int M, int N,
                                                          int rows_per_chunk, float* __restrict__ part) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(M, r0 + rows_per_chunk);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    const float* p = a + n;
    int r = r0 + wave;
    for (; r + 12 < r1; r += 16) {
      s0 += p[(long)r * lda];
      s1 += p[(long)(r + 4) * lda];
      s2 += p[(long)(r + 8) * lda];
      s3 += p[(long)(r + 12) * lda];
    }
    for (; r < r1; r += 4) s0 += p[(long)r * lda];
  }
  sm[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && n < N) part[(long)blockIdx.y * N + n] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

__global__ void scale_kernel(const float* __restrict__ in, const float* __restrict__ s, float* __restrict__ out,
                             long n) {
  const float f = s[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = in[i] * f;
}

// ---------------------------------------------------------------------------------------- Adam
// torch.optim.Adam single-tensor update order: m.lerp_(g, 1-b1); v = v*b2 + (1-b2)*g*g;
// denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float gscale) {
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// The same update behind a guard: n_guard words of device memory (status words of this step's launches; at most 64);
// if any of them is non-zero the kernel returns without touching a buffer.  The decision is made on the device, so
// the host need not wait for the step to finish before it enqueues the update.
__global__ void adam_guarded_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                                    float bc2_sqrt, float gscale, const int* __restrict__ guard, int n_guard) {
  int any = 0;
  for (int k = 0; k < n_guard; ++k) any |= guard[k];
  if (any != 0) return;
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// Step counter on the device, for callers that replay a recorded launch sequence (HIP graphs): one thread
// advances the counter and leaves this step's bias corrections beside it; the update kernel reads them.
// state: [int64 step][float 1 - b1^t][float sqrt(1 - b2^t)]
__global__ void adam_advance_kernel(long long* state, float b1, float b2) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long t = state[0] + 1;
    state[0] = t;
    float* coef = reinterpret_cast<float*>(state + 1);
    coef[0] = (float)(1.0 - pow((double)b1, (double)t));
    coef[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
  }
}
__global__ void adam_counted_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                    const long long* __restrict__ state, float gscale) {
  const float* coef = reinterpret_cast<const float*>(state + 1);
  const float step_size = lr / coef[0];
  const float bc2_sqrt = coef[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// tile16 [Mp*T, N] (window row order) <-> row-major [B, T, N]
__global__ void tile16_from_rows_kernel(const float* __restrict__ rows, float* __restrict__ tiled, int B, int T,
                                        int N, int to_rows) {
  const int Bp = (B + 15) / 16 * 16;
  const long total = (long)Bp * T * N;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int n = idx % N;
    const long bt_ = idx / N;
    const int t = bt_ % T;
    const int b = bt_ / T;
    const long mprime = ((long)(b >> 4) * T + t) * 16 + (b & 15);
    const long off = vts_tile16_offset(mprime, n, N);
    if (to_rows) {
      if (b < B) const_cast<float*>(rows)[idx] = tiled[off];
    } else {
      tiled[off] = (b < B) ? rows[idx] : 0.f;
    }
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

int vts_ew_reparam_fwd(int mode, const float* mean, const float* logvar, const float* scale, const float* shape,
                       const float* eps, const float* u, const float* choice, const float* pi, float* z, long B,
                       int L, hipStream_t st) {
  const long n = B * L;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, mode, mean, logvar, scale, shape, eps,
                     u, choice, pi, z, n, L);
  VTS_CHECK_LAUNCH("reparam_fwd");
  return VTS_OK;
}

int vts_ew_reparam_bwd(int mode, const float* dz, const float* logvar, const float* scale, const float* shape,
                       const float* eps, const float* u, const float* choice, const float* pi, float* dmean,
                       float* dlogvar, float* dscale, float* dshape, long B, int L, hipStream_t st) {
  const long n = B * L;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, mode, dz, logvar, scale, shape, eps,
                     u, choice, pi, dmean, dlogvar, dscale, dshape, n, L);
  VTS_CHECK_LAUNCH("reparam_bwd");
  return VTS_OK;
}

int vts_ew_elbo_blocks(long nrec, long nlat) {
  long n = nrec > nlat ? nrec : nlat;
  long b = (n + 1023) / 1024;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

int vts_ew_elbo_fwd(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                    const float* scale, const float* shape, const float* pi, float thr, float beta, long B, int fs,
                    int L, float* part, float* out10, hipStream_t st) {
  const long nrec = B * fs, nlat = B * L;
  const int blocks = vts_ew_elbo_blocks(nrec, nlat);
  hipLaunchKernelGGL(elbo_partials_kernel, dim3(blocks), dim3(256), 0, st, kind, recon, y, mean, logvar, scale,
                     shape, thr, nrec, nlat, part);
  VTS_CHECK_LAUNCH("elbo_partials");
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(1), dim3(256), 0, st, kind, part, blocks, (const float*)nullptr, 0,
                     pi, beta, B, nrec, nlat, out10, (const float*)nullptr, (float*)nullptr, 1.f);
  VTS_CHECK_LAUNCH("elbo_finalize");
  return VTS_OK;
}

// ---- the same loss for callers that cannot allocate ------------------------------------------------
// The block partials of elbo_partials_kernel go into a file-scope device array instead of a caller
// workspace.  `scratch` is a host-side stand-in for the symbol, not a device address: each device
// that loads the code object holds its own copy, whose address hipGetSymbolAddress returns with that
// device current.  The address is looked up once per device and kept; every launch takes it as an
// ordinary kernel argument.  Every call on a device shares that device's array: the partials kernel
// rewrites every slot the finalize kernel reads, so nothing has to start at zero, but two calls
// that overlap on one device (different streams) race on it.
constexpr int ELBO_STATIC_MAX_BLOCKS = 1024;                       // 1024 elements per block -> 2^20 elements
constexpr int VTS_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[ELBO_STATIC_MAX_BLOCKS * 6];   // 24 KiB per device
static float* g_scratch_ptr[VTS_MAX_DEVICES];
static std::mutex g_scratch_mutex;

long vts_ew_elbo_static_max_elements() { return (long)ELBO_STATIC_MAX_BLOCKS * 1024; }

int vts_ew_elbo_fwd_static(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                           const float* scale, const float* shape, const float* pi, float thr, float beta, long B,
                           int fs, int L, float* out10, hipStream_t st) {
  const long nrec = B * fs, nlat = B * L;
  const long nmax = nrec > nlat ? nrec : nlat;
  if (nmax > vts_ew_elbo_static_max_elements()) {
    vts_set_error("elbo_fwd_static: %ld elements exceed the static scratch buffer's %ld (B * max(fs, L)); use "
                  "vts_elbo_fwd with a caller workspace", nmax, vts_ew_elbo_static_max_elements());
    return VTS_ERR_ARG;
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VTS_MAX_DEVICES) {
    vts_set_error("elbo_fwd_static: no current device (or device index above %d)", VTS_MAX_DEVICES - 1);
    return VTS_ERR_HIP;
  }
  float* part = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
      void* p = nullptr;
      const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      if (e != hipSuccess || p == nullptr) {
        vts_set_error("elbo_fwd_static: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                      hipGetErrorString(e));
        return VTS_ERR_HIP;
      }
      g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    part = g_scratch_ptr[dev];
  }
  const int blocks = vts_ew_elbo_blocks(nrec, nlat);      // <= ELBO_STATIC_MAX_BLOCKS by the check above
  hipLaunchKernelGGL(elbo_partials_kernel, dim3(blocks), dim3(256), 0, st, kind, recon, y, mean, logvar, scale,
                     shape, thr, nrec, nlat, part);
  VTS_CHECK_LAUNCH("elbo_partials(static)");
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(1), dim3(256), 0, st, kind, (const float*)part, blocks,
                     (const float*)nullptr, 0, pi, beta, B, nrec, nlat, out10, (const float*)nullptr, (float*)nullptr,
                     1.f);
  VTS_CHECK_LAUNCH("elbo_finalize(static)");
  return VTS_OK;
}

int vts_ew_reparam_kl_blocks(long B, int L) {
  long c = (B * L + 1023) / 1024;
  if (c > 2048) c = 2048;
  return c < 1 ? 1 : (int)c;
}

namespace {
template <int V>
int reparam_kl_fwd_v(int mode, const float* mean, const float* logvar, const float* scale, const float* shape,
                     const float* eps, const float* u, const float* choice, const float* pi, float thr, float* z,
                     float* part, long n, int L, int blocks, hipStream_t st) {
#define VTS_RKF(M)                                                                                                 \
  hipLaunchKernelGGL((reparam_kl_fwd_kernel<M, V>), dim3(blocks), dim3(256), 0, st, mean, logvar, scale, shape, eps, \
                     u, choice, pi, thr, z, part, n, L)
  if (mode == 0) VTS_RKF(0);
  else if (mode == 1) VTS_RKF(1);
  else VTS_RKF(2);
#undef VTS_RKF
  VTS_CHECK_LAUNCH("reparam_kl_fwd");
  return VTS_OK;
}
template <int V>
int reparam_kl_bwd_v(int mode, const float* dz, const float* dpart, const float* mean, const float* logvar,
                     const float* scale, const float* shape, const float* eps, const float* u, const float* choice,
                     const float* pi, float thr, float* dmean, float* dlogvar, float* dscale, float* dshape, long n,
                     int L, int blocks, hipStream_t st) {
#define VTS_RKB(M)                                                                                              \
  hipLaunchKernelGGL((reparam_kl_bwd_kernel<M, V>), dim3(blocks), dim3(256), 0, st, dz, dpart, mean, logvar, scale, \
                     shape, eps, u, choice, pi, thr, dmean, dlogvar, dscale, dshape, n, L)
  if (mode == 0) VTS_RKB(0);
  else if (mode == 1) VTS_RKB(1);
  else VTS_RKB(2);
#undef VTS_RKB
  VTS_CHECK_LAUNCH("reparam_kl_bwd");
  return VTS_OK;
}
inline bool aligned16(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
  return true;
}
}  // namespace

int vts_ew_reparam_kl_fwd(int mode, const float* mean, const float* logvar, const float* scale, const float* shape,
                          const float* eps, const float* u, const float* choice, const float* pi, float thr, float* z,
                          float* part, long B, int L, hipStream_t st) {
  const long n = B * L;
  const int blocks = vts_ew_reparam_kl_blocks(B, L);
  const bool v4 = (L % 4 == 0) && aligned16({mean, logvar, scale, shape, eps, u, z});
  return v4 ? reparam_kl_fwd_v<4>(mode, mean, logvar, scale, shape, eps, u, choice, pi, thr, z, part, n, L, blocks, st)
            : reparam_kl_fwd_v<1>(mode, mean, logvar, scale, shape, eps, u, choice, pi, thr, z, part, n, L, blocks, st);
}

int vts_ew_reparam_kl_bwd(int mode, const float* dz, const float* dpart, const float* mean, const float* logvar,
                          const float* scale, const float* shape, const float* eps, const float* u,
                          const float* choice, const float* pi, float thr, float* dmean, float* dlogvar,
                          float* dscale, float* dshape, long B, int L, hipStream_t st) {
  const long n = B * L;
  const int blocks = vts_ew_reparam_kl_blocks(B, L);
  const bool v4 = (L % 4 == 0) && aligned16({dz, mean, logvar, scale, shape, eps, u, dmean, dlogvar, dscale, dshape});
  return v4 ? reparam_kl_bwd_v<4>(mode, dz, dpart, mean, logvar, scale, shape, eps, u, choice, pi, thr, dmean,
                                  dlogvar, dscale, dshape, n, L, blocks, st)
            : reparam_kl_bwd_v<1>(mode, dz, dpart, mean, logvar, scale, shape, eps, u, choice, pi, thr, dmean,
                                  dlogvar, dscale, dshape, n, L, blocks, st);
}

// Loss from the reconstruction pair and KL partial sums already produced by vts_ew_reparam_kl_fwd.
int vts_ew_elbo_recon_fwd(int kind, const float* recon, const float* y, const float* klpart, int nkl,
                          const float* pi, float thr, float beta, long B, int fs, int L, float* part, float* out10,
                          hipStream_t st) {
  const long nrec = B * fs, nlat = B * L;
  const int blocks = vts_ew_elbo_blocks(nrec, 0);
  hipLaunchKernelGGL(elbo_partials_kernel, dim3(blocks), dim3(256), 0, st, kind, recon, y, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, thr, nrec, 0L, part);
  VTS_CHECK_LAUNCH("elbo_partials");
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(1), dim3(256), 0, st, kind, part, blocks, klpart, nkl, pi, beta, B,
                     nrec, nlat, out10, (const float*)nullptr, (float*)nullptr, 1.f);
  VTS_CHECK_LAUNCH("elbo_finalize");
  return VTS_OK;
}

// This rank's additive statistics of the loss: sums[0..5] as elbo_partials_kernel's slots, sums[6] = rows.
// The KL slots come from the latent tensors, or from klpart (reparam_kl_fwd's partial sums) when given.
int vts_ew_elbo_sums(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                     const float* scale, const float* shape, const float* klpart, int nkl, float thr, long B, int fs,
                     int L, float* part, float* sums, hipStream_t st) {
  const long nrec = B * fs, nlat = klpart ? 0L : B * L;
  const int blocks = vts_ew_elbo_blocks(nrec, nlat);
  hipLaunchKernelGGL(elbo_partials_kernel, dim3(blocks), dim3(256), 0, st, kind, recon, y, mean, logvar, scale, shape,
                     thr, nrec, nlat, part);
  VTS_CHECK_LAUNCH("elbo_partials");
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(1), dim3(256), 0, st, kind, part, blocks, klpart, klpart ? nkl : 0,
                     (const float*)nullptr, 0.f, B, nrec, B * L, (float*)nullptr, (const float*)nullptr, sums, 1.f);
  VTS_CHECK_LAUNCH("elbo_sums");
  return VTS_OK;
}

int vts_ew_elbo_from_sums(int kind, const float* sums, const float* pi, float beta, float gscale, int fs, int L,
                          float* out10, hipStream_t st) {
  hipLaunchKernelGGL(elbo_finalize_kernel, dim3(1), dim3(64), 0, st, kind, (const float*)nullptr, 0,
                     (const float*)nullptr, 0, pi, beta, 1L, (long)fs, (long)L, out10, sums, (float*)nullptr, gscale);
  VTS_CHECK_LAUNCH("elbo_from_sums");
  return VTS_OK;
}

int vts_ew_elbo_recon_bwd(int kind, const float* recon, const float* y, const float* coef, const float* gout,
                          float thr, long B, int fs, float* drecon, float* dklpart, int nkl, float* dpi,
                          hipStream_t st) {
  const long nrec = B * fs;
  const long m = nrec > nkl ? nrec : (long)nkl;
  hipLaunchKernelGGL(elbo_grads_kernel, dim3(grid_for(m)), dim3(256), 0, st, kind, recon, y, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, coef, gout, thr, nrec, 0L,
                     drecon, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, dpi, dklpart, nkl);
  VTS_CHECK_LAUNCH("elbo_grads");
  return VTS_OK;
}

int vts_ew_elbo_bwd(int kind, const float* recon, const float* y, const float* mean, const float* logvar,
                    const float* scale, const float* shape, const float* coef, const float* gout, float thr, long B,
                    int fs, int L, float* drecon, float* dmean, float* dlogvar, float* dscale, float* dshape,
                    float* dpi, hipStream_t st) {
  const long nrec = B * fs, nlat = B * L;
  hipLaunchKernelGGL(elbo_grads_kernel, dim3(grid_for(nlat > nrec ? nlat : nrec)), dim3(256), 0, st, kind, recon, y,
                     mean, logvar, scale, shape, coef, gout, thr, nrec, nlat, drecon, dmean, dlogvar, dscale, dshape,
                     dpi, (float*)nullptr, 0);
  VTS_CHECK_LAUNCH("elbo_grads");
  return VTS_OK;
}

int vts_ew_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, float* mean,
                         float* rstd, int M, int H, float eps, hipStream_t st) {
  if (ln_narrow_ok(x, y, gamma, ldx, H, H) && vts_aligned16(beta)) {
    const int rpw = 256 / H;   // rows per wave
    const dim3 grid(vts_cdiv(M, 4 * rpw));
    if (H == 64) hipLaunchKernelGGL(layernorm_fwd_narrow_kernel<16>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, mean, rstd, M, eps);
    else if (H == 32) hipLaunchKernelGGL(layernorm_fwd_narrow_kernel<8>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, mean, rstd, M, eps);
    else hipLaunchKernelGGL(layernorm_fwd_narrow_kernel<4>, grid, dim3(256), 0, st, x, ldx, gamma, beta, y, mean, rstd, M, eps);
    VTS_CHECK_LAUNCH("layernorm_fwd_narrow");
    return VTS_OK;
  }
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(vts_cdiv(M, 4)), dim3(256), 0, st, x, ldx, gamma, beta, y, mean,
                     rstd, M, H, eps);
  VTS_CHECK_LAUNCH("layernorm_fwd");
  return VTS_OK;
}

int vts_ew_layernorm_bwd_blocks(int M) {
  int b = vts_cdiv(M, 16);
  if (b > 1024) b = 1024;
  return b < 1 ? 1 : b;
}

int vts_ew_layernorm_bwd(const float* dy, const float* x, long ldx, const float* gamma, const float* mean,
                         const float* rstd, float* dx, long lddx, float* part, int M, int H, hipStream_t st) {
  const int blocks = vts_ew_layernorm_bwd_blocks(M);
  const int rpb = vts_cdiv(M, blocks);
  if (ln_narrow_ok(x, dy, dx, ldx, lddx, H) && vts_aligned16(gamma)) {
    if (H == 64) hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<16>, dim3(blocks), dim3(256), 0, st, dy, x, ldx, gamma, mean, rstd, dx, lddx, part, M, rpb);
    else if (H == 32) hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<8>, dim3(blocks), dim3(256), 0, st, dy, x, ldx, gamma, mean, rstd, dx, lddx, part, M, rpb);
    else hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<4>, dim3(blocks), dim3(256), 0, st, dy, x, ldx, gamma, mean, rstd, dx, lddx, part, M, rpb);
    VTS_CHECK_LAUNCH("layernorm_bwd_narrow");
    return VTS_OK;
  }
  if (H == 256 || H == 128 || H == 64) {
    if (H == 256) hipLaunchKernelGGL(layernorm_bwd_cols_kernel<4>, dim3(blocks), dim3(256), 0, st, dy, x, ldx, gamma, mean, rstd, dx, lddx, part, M, rpb);
    else if (H == 128) hipLaunchKernelGGL(layernorm_bwd_cols_kernel<2>, dim3(blocks), dim3(256), 0, st, dy, x, ldx, gamma, mean, rstd, dx, lddx, part, M, rpb);
    else hipLaunchKernelGGL(layernorm_bwd_cols_kernel<1>, dim3(blocks), dim3(256), 0, st, dy, x, ldx, gamma, mean, rstd, dx, lddx, part, M, rpb);
    VTS_CHECK_LAUNCH("layernorm_bwd_cols");
    return VTS_OK;
  }
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 2 * H * sizeof(float), st, dy, x, ldx, gamma,
                     mean, rstd, dx, lddx, part, M, H, rpb);
  VTS_CHECK_LAUNCH("layernorm_bwd");
  return VTS_OK;
}

int vts_ew_act_bwd(int kind, const float* dy, const float* y, const float* aux, float scale, float* dpre, long n,
                   hipStream_t st) {
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, kind, dy, y, aux, scale, dpre, n);
  VTS_CHECK_LAUNCH("act_bwd");
  return VTS_OK;
}

int vts_ew_colsum_chunks(int M) {
  int c = vts_cdiv(M, 256);
  if (c > 1024) c = 1024;
  return c < 1 ? 1 : c;
}

int vts_ew_colsum_rows(const float* a, long lda, int M, int N, float* part, hipStream_t st) {
  const int chunks = vts_ew_colsum_chunks(M);
  const int rpc = vts_cdiv(M, chunks);
  hipLaunchKernelGGL(colsum_rows_kernel, dim3(vts_cdiv(N, 64), chunks), dim3(256), 0, st, a, lda, M, N, rpc, part);
  VTS_CHECK_LAUNCH("colsum_rows");
  return VTS_OK;
}

// ---- gradients of many parameters into one flat buffer -----------------------------------------------------------
// Up to GATHER_MAX chunks a launch: dst[off[c] .. off[c] + n[c]) = src[c] (zeros where src[c] is null).  blockIdx.y
// = chunk, so the table is indexed with a wave-uniform value and stays in the kernel-argument segment.
namespace {
constexpr int GATHER_MAX = 40;
struct GatherArgs {
  const float* src[GATHER_MAX];
  long n[GATHER_MAX];
  long off[GATHER_MAX];
};
__global__ __launch_bounds__(256) void gather_chunks_kernel(const GatherArgs A, float* __restrict__ dst) {
  const int c = blockIdx.y;
  const float* __restrict__ s = A.src[c];
  float* __restrict__ d = dst + A.off[c];
  const long n = A.n[c];
  const long stride = (long)gridDim.x * blockDim.x;
  const long start = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s == nullptr) {
    for (long i = start; i < n; i += stride) d[i] = 0.f;
    return;
  }
  if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
    const long n4 = n >> 2;
    for (long i = start; i < n4; i += stride)
      reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(s)[i];
    for (long i = (n4 << 2) + start; i < n; i += stride) d[i] = s[i];
  } else {
    for (long i = start; i < n; i += stride) d[i] = s[i];
  }
}
}  // namespace

int vts_ew_gather_chunks(const float* const* src, const long long* numel, const long long* dst_off, long count,
                         float* dst, hipStream_t st) {
  for (long c0 = 0; c0 < count; c0 += GATHER_MAX) {
    const int m = (int)((count - c0) < GATHER_MAX ? (count - c0) : GATHER_MAX);
    GatherArgs A{};
    long nmax = 1;
    for (int i = 0; i < m; ++i) {
      A.src[i] = src[c0 + i];
      A.n[i] = (long)numel[c0 + i];
      A.off[i] = (long)dst_off[c0 + i];
      if (A.n[i] > nmax) nmax = A.n[i];
    }
    int bx = vts_cdiv(nmax, 4 * 256 * 4);      // four float4 per thread in the largest chunk
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(gather_chunks_kernel, dim3(bx, m), dim3(256), 0, st, A, dst);
    VTS_CHECK_LAUNCH("gather_chunks");
  }
  return VTS_OK;
}

int vts_ew_scale(const float* in, const float* s, float* out, long n, hipStream_t st) {
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, s, out, n);
  VTS_CHECK_LAUNCH("scale");
  return VTS_OK;
}

int vts_ew_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                float bc1, float bc2_sqrt, float gscale, hipStream_t st) {
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1,
                     bc2_sqrt, gscale);
  VTS_CHECK_LAUNCH("adam");
  return VTS_OK;
}

int vts_ew_adam_guarded(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        float bc1, float bc2_sqrt, float gscale, const int* guard, int n_guard, hipStream_t st) {
  hipLaunchKernelGGL(adam_guarded_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1,
                     bc2_sqrt, gscale, guard, n_guard);
  VTS_CHECK_LAUNCH("adam_guarded");
  return VTS_OK;
}

int vts_ew_adam_counted(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                        long long* state, float gscale, hipStream_t st) {
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, st, state, b1, b2);
  VTS_CHECK_LAUNCH("adam_advance");
  hipLaunchKernelGGL(adam_counted_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, state,
                     gscale);
  VTS_CHECK_LAUNCH("adam_counted");
  return VTS_OK;
}

int vts_ew_tile16_convert(const float* rows, float* tiled, int B, int T, int N, int to_rows, hipStream_t st) {
  const long total = (long)((B + 15) / 16 * 16) * T * N;
  hipLaunchKernelGGL(tile16_from_rows_kernel, dim3(grid_for(total)), dim3(256), 0, st, rows, tiled, B, T, N,
                     to_rows);
  VTS_CHECK_LAUNCH("tile16_convert");
  return VTS_OK;
}
