// Three-component latent mixture (Gaussian / generalized Pareto / Bernoulli) of
// vae-models/graph/vae_mixture_graph_model.py as its LD2011 driver configures it
// (use_gcn = use_d_knn = False, use_gpd = use_bernoulli = True: graph/train_ld_2011.py:91):
//   reparameterize :182-209   per-row pick among the three samples by one uniform draw against the
//                             cumulative softmax(pi_params)
//   loss_function  :222-285   pi_g*(MSE over normal rows + beta*KL_gauss) + pi_p*GPD NLL of the
//                             exceedances y - u over the extreme rows + pi_b*Bernoulli NLL over the
//                             rows with y == 0 (future_steps = 1: y is one value per row)
// HBM-bound elementwise passes with wave64 shuffle reductions; per-block partial sums are combined
// by a single workgroup in fp64 in a fixed order, so the loss is deterministic.
#include "vts_common.h"
#include "vts_gpd_pow.h"
#include "../../include/vts.h"

namespace {

// softmax of the three mixture logits the way torch computes it (subtract max, exp, normalise)
__device__ __forceinline__ void softmax3(const float* pi, float (&p)[3]) {
  const float m = fmaxf(pi[0], fmaxf(pi[1], pi[2]));
  const float e0 = expf(pi[0] - m), e1 = expf(pi[1] - m), e2 = expf(pi[2] - m);
  const float s = e0 + e1 + e2;
  p[0] = e0 / s; p[1] = e1 / s; p[2] = e2 / s;
}

// component of a row: 0 gaussian, 1 GPD, 2 bernoulli (:195-198: choice < p0, p0 <= choice < p0 + p1, rest)
__device__ __forceinline__ int pick(float choice, const float (&p)[3]) {
  if (choice < p[0]) return 0;
  return (choice < p[0] + p[1]) ? 1 : 2;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void reparam3_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ logvar,
                                    const float* __restrict__ scale, const float* __restrict__ shape,
                                    const float* __restrict__ logits, const float* __restrict__ bern, int bern_is_uniform,
                                    const float* __restrict__ eps, const float* __restrict__ u,
                                    const float* __restrict__ choice, const float* __restrict__ pi,
                                    float* __restrict__ z, long n, int L) {
  float p[3];
  softmax3(pi, p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = pick(choice[i / L], p);
    float out;
    if (k == 0) {
      out = mean[i] + eps[i] * expf(0.5f * logvar[i]);
    } else if (k == 1) {
      const float sc = scale[i], sh = shape[i];
      out = (sc / sh) * (vts_gpd_pow(1.0f - u[i], -sh) - 1.0f);
    } else {
      // torch.bernoulli(sigmoid(logits)) (:21-23): 1 where the uniform draw is below the probability
      out = bern_is_uniform ? (bern[i] < sigmoidf_(logits[i]) ? 1.0f : 0.0f) : bern[i];
    }
    z[i] = out;
  }
}

// Sampling gradients (the Bernoulli sample and the row pick carry none).
__global__ void reparam3_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ logvar,
                                    const float* __restrict__ scale, const float* __restrict__ shape,
                                    const float* __restrict__ eps, const float* __restrict__ u,
                                    const float* __restrict__ choice, const float* __restrict__ pi,
                                    float* __restrict__ dmean, float* __restrict__ dlogvar, float* __restrict__ dscale,
                                    float* __restrict__ dshape, long n, int L) {
  float p[3];
  softmax3(pi, p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = pick(choice[i / L], p);
    const float d = dz[i];
    float dm = 0.f, dl = 0.f, dsc = 0.f, dsh = 0.f;
    if (k == 0) {
      dm = d;
      dl = d * eps[i] * expf(0.5f * logvar[i]) * 0.5f;
    } else if (k == 1) {
      // autograd's decomposition of scale/shape*((1-u)**(-shape)-1): a = scale/shape, q = p-1
      const float sc = scale[i], sh = shape[i];
      const float base = 1.0f - u[i];
      const float pw = vts_gpd_pow(base, -sh);
      const float a = sc / sh, q = pw - 1.0f;
      const float da = d * q, dq = d * a;
      dsc = da / sh;
      dsh = -da * (a / sh) - dq * (pw * logf(base));
    }
    dmean[i] = dm; dlogvar[i] = dl; dscale[i] = dsc; dshape[i] = dsh;
  }
}

// ---- loss -----------------------------------------------------------------------------------------
// row class from y and the threshold (classify_observations, :26-50): 0 normal, 1 extreme, 2 zero, 3 none (NaN)
__device__ __forceinline__ int row_class(float y, float thr) {
  if (y == 0.0f) return 2;
  if (y <= thr) return 0;
  if (y > thr) return 1;
  return 3;
}

constexpr int NSUM = 7;   // sq err over normal rows, KL inner sum, GPD NLL sum, Bernoulli log-lik sum, three row counts
enum { S_SQ = 0, S_KL, S_GPD, S_BER, S_NN, S_NE, S_NZ };
// state written by the forward for the backward (floats)
enum { ST_PG = 0, ST_PP, ST_PB, ST_NN, ST_NE, ST_NZ, ST_A, ST_GPD, ST_BER, ST_N };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void elbo3_partials_kernel(
    const float* __restrict__ recon, const float* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ logvar, const float* __restrict__ scale, const float* __restrict__ shape,
    const float* __restrict__ logits, float thr, float* __restrict__ part, long B, int L) {
  __shared__ float red[4][NSUM];
  float s[NSUM];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) s[i] = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long b = tid; b < B; b += stride) {
    const float yv = y[b];
    const int k = row_class(yv, thr);
    if (k == 0) {
      const float d = recon[b] - yv;
      s[S_SQ] += d * d;
      s[S_NN] += 1.f;
    } else if (k == 1) {
      s[S_NE] += 1.f;
    } else if (k == 2) {
      s[S_NZ] += 1.f;
    }
  }
  const long n = B * L;
  for (long i = tid; i < n; i += stride) {
    const float yv = y[i / L];
    const int k = row_class(yv, thr);
    const float mu = mean[i], lv = logvar[i];
    s[S_KL] += 1.0f + lv - mu * mu - expf(lv);
    if (k == 1) {
      const float sc = scale[i], sh = shape[i];
      const float excess = yv - thr;
      s[S_GPD] += logf(sc) + (1.0f + 1.0f / sh) * logf(1.0f + sh * excess / sc);
    } else if (k == 2) {
      // y_zero*log(sigmoid) + (1 - y_zero)*log(1 - sigmoid) with y_zero == 0, spelled out so that an
      // underflowed sigmoid gives the reference's 0 * -inf = NaN
      const float sg = sigmoidf_(logits[i]);
      s[S_BER] += yv * logf(sg) + (1.0f - yv) * logf(1.0f - sg);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NSUM; ++i) {
    const float v = wave_sum(s[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NSUM)
    part[(long)blockIdx.x * NSUM + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                                  (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one workgroup of 256: fp64 sums of the partials in a fixed order, then the scalar loss and the state
__global__ __launch_bounds__(256) void elbo3_finalize_kernel(const float* __restrict__ part, int nparts,
                                                             const float* __restrict__ pi, float beta, long B, int L,
                                                             float* __restrict__ loss, float* __restrict__ state) {
  __shared__ double red[256];
  __shared__ double tot[NSUM];
  for (int q = 0; q < NSUM; ++q) {
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += (double)part[(long)i * NSUM + q];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) tot[q] = red[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float p[3];
    softmax3(pi, p);
    const double nn = tot[S_NN], ne = tot[S_NE], nz = tot[S_NZ];
    const float r_g = (float)(tot[S_SQ] / nn);                       // mse over an empty selection is NaN, as in torch
    const float kl_g = (float)(-0.5 * tot[S_KL] / (double)B);
    const float a = r_g + beta * kl_g;
    const float gpd = ne > 0.0 ? (float)(tot[S_GPD] / (ne * L)) : 0.f;
    const float ber = nz > 0.0 ? (float)(-tot[S_BER] / (nz * L)) : 0.f;
    float total = p[0] * a;
    if (ne > 0.0) total += p[1] * gpd;
    if (nz > 0.0) total += p[2] * ber;
    loss[0] = total;
    state[ST_PG] = p[0]; state[ST_PP] = p[1]; state[ST_PB] = p[2];
    state[ST_NN] = (float)nn; state[ST_NE] = (float)ne; state[ST_NZ] = (float)nz;
    state[ST_A] = a; state[ST_GPD] = gpd; state[ST_BER] = ber;
  }
}

__global__ void elbo3_grads_kernel(const float* __restrict__ recon, const float* __restrict__ y,
                                   const float* __restrict__ mean, const float* __restrict__ logvar,
                                   const float* __restrict__ scale, const float* __restrict__ shape,
                                   const float* __restrict__ logits, float thr, float beta,
                                   const float* __restrict__ state, const float* __restrict__ gout,
                                   float* __restrict__ drecon, float* __restrict__ dmean, float* __restrict__ dlogvar,
                                   float* __restrict__ dscale, float* __restrict__ dshape, float* __restrict__ dlogits,
                                   float* __restrict__ dpi, long B, int L) {
  const float go = gout[0];
  const float pg = state[ST_PG], pp = state[ST_PP], pb = state[ST_PB];
  const float nn = state[ST_NN], ne = state[ST_NE], nz = state[ST_NZ];
  const long stride = (long)gridDim.x * blockDim.x;
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid == 0) {
    // softmax backward over the three component losses (a component without rows is the constant 0)
    const float g[3] = {state[ST_A], ne > 0.f ? state[ST_GPD] : 0.f, nz > 0.f ? state[ST_BER] : 0.f};
    const float p[3] = {pg, pp, pb};
    const float dot = p[0] * g[0] + p[1] * g[1] + p[2] * g[2];
    for (int j = 0; j < 3; ++j) dpi[j] = go * p[j] * (g[j] - dot);
  }
  const float wr = go * pg * 2.0f / nn;
  for (long b = tid; b < B; b += stride) {
    const float yv = y[b];
    drecon[b] = row_class(yv, thr) == 0 ? wr * (recon[b] - yv) : 0.f;
  }
  const float wkl = go * pg * beta / (float)B;
  const float wgp = ne > 0.f ? go * pp / (ne * L) : 0.f;
  const float wbe = nz > 0.f ? go * pb / (nz * L) : 0.f;
  const long n = B * L;
  for (long i = tid; i < n; i += stride) {
    const float yv = y[i / L];
    const int k = row_class(yv, thr);
    const float mu = mean[i], lv = logvar[i];
    dmean[i] = wkl * mu;
    dlogvar[i] = wkl * (-0.5f) * (1.0f - expf(lv));
    float dsc = 0.f, dsh = 0.f, dlg = 0.f;
    if (k == 1) {
      const float sc = scale[i], sh = shape[i];
      const float excess = yv - thr;
      const float t = 1.0f + sh * excess / sc;
      const float c = 1.0f + 1.0f / sh;
      dsc = wgp * (1.0f / sc - c * (sh * excess / (sc * sc)) / t);
      dsh = wgp * (-(1.0f / (sh * sh)) * logf(t) + c * (excess / sc) / t);
    } else if (k == 2) {
      // -d/dl [y*log(s) + (1-y)*log(1-s)], s = sigmoid(l): y*(1-s) - (1-y)*s
      const float sg = sigmoidf_(logits[i]);
      dlg = -wbe * (yv * (1.0f - sg) - (1.0f - yv) * sg);
    }
    dscale[i] = dsc; dshape[i] = dsh; dlogits[i] = dlg;
  }
}

int blocks_for(long n) {
  long b = (n + 255) / 256;
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int vts_reparam3_fwd(const float* mean, const float* logvar, const float* scale, const float* shape,
                     const float* logits, const float* bern, int bern_is_uniform, const float* eps, const float* u,
                     const float* choice, const float* pi, float* z, int64_t B, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(mean && logvar && scale && shape && bern && eps && u && choice && pi && z && B > 0 && L > 0,
              "reparam3_fwd: bad arguments");
  VTS_REQUIRE(!bern_is_uniform || logits, "reparam3_fwd: a uniform Bernoulli draw needs the logits");
  hipLaunchKernelGGL(reparam3_fwd_kernel, dim3(blocks_for(B * L)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), mean,
                     logvar, scale, shape, logits, bern, bern_is_uniform, eps, u, choice, pi, z, (long)(B * L), (int)L);
  VTS_CHECK_LAUNCH("reparam3_fwd");
  return VTS_OK;
}

int vts_reparam3_bwd(const float* dz, const float* logvar, const float* scale, const float* shape, const float* eps,
                     const float* u, const float* choice, const float* pi, float* dmean, float* dlogvar, float* dscale,
                     float* dshape, int64_t B, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(dz && logvar && scale && shape && eps && u && choice && pi && dmean && dlogvar && dscale && dshape &&
                  B > 0 && L > 0, "reparam3_bwd: bad arguments");
  hipLaunchKernelGGL(reparam3_bwd_kernel, dim3(blocks_for(B * L)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dz,
                     logvar, scale, shape, eps, u, choice, pi, dmean, dlogvar, dscale, dshape, (long)(B * L), (int)L);
  VTS_CHECK_LAUNCH("reparam3_bwd");
  return VTS_OK;
}

size_t vts_elbo3_workspace_bytes(int64_t B, int64_t L) {
  return ((size_t)blocks_for(B * L) * NSUM * sizeof(float) + 255) / 256 * 256;
}

int vts_elbo3_fwd(const float* recon, const float* y, const float* mean, const float* logvar, const float* scale,
                  const float* shape, const float* logits, const float* pi, float threshold, float beta, float* loss,
                  float* state, void* workspace, size_t workspace_bytes, int64_t B, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(recon && y && mean && logvar && scale && shape && logits && pi && loss && state && workspace && B > 0 &&
                  L > 0, "elbo3_fwd: bad arguments");
  VTS_REQUIRE(workspace_bytes >= vts_elbo3_workspace_bytes(B, L), "elbo3_fwd: workspace %zu B too small", workspace_bytes);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = blocks_for(B * L);
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(elbo3_partials_kernel, dim3(nb), dim3(256), 0, st, recon, y, mean, logvar, scale, shape, logits,
                     threshold, part, (long)B, (int)L);
  VTS_CHECK_LAUNCH("elbo3_partials");
  hipLaunchKernelGGL(elbo3_finalize_kernel, dim3(1), dim3(256), 0, st, part, nb, pi, beta, (long)B, (int)L, loss, state);
  VTS_CHECK_LAUNCH("elbo3_finalize");
  return VTS_OK;
}

int vts_elbo3_bwd(const float* recon, const float* y, const float* mean, const float* logvar, const float* scale,
                  const float* shape, const float* logits, float threshold, float beta, const float* state,
                  const float* gout, float* drecon, float* dmean, float* dlogvar, float* dscale, float* dshape,
                  float* dlogits, float* dpi, int64_t B, int64_t L, vts_stream_t stream) {
  VTS_REQUIRE(recon && y && mean && logvar && scale && shape && logits && state && gout && drecon && dmean && dlogvar &&
                  dscale && dshape && dlogits && dpi && B > 0 && L > 0, "elbo3_bwd: bad arguments");
  hipLaunchKernelGGL(elbo3_grads_kernel, dim3(blocks_for(B * L)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), recon,
                     y, mean, logvar, scale, shape, logits, threshold, beta, state, gout, drecon, dmean, dlogvar, dscale,
                     dshape, dlogits, dpi, (long)B, (int)L);
  VTS_CHECK_LAUNCH("elbo3_grads");
  return VTS_OK;
}

}  // extern "C"
