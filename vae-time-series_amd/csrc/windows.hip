// Training windows cut on the device from a series that stays resident in HBM: the step in front of
// the train step.  The reference builds every window on the host, standardises the copies and uploads
// them (vae-models/graph/train_ld_2011.py:11-17 create_sliding_windows, :47-59 StandardScaler
// fit/transform; vae-models/graph/training_block_max_hurdat.py:81-86,100-106 block-maximum targets).
// Here a batch is a gather: window b is the T consecutive series rows from starts[b], which is one
// contiguous run of T*C floats, so a batch costs one coalesced HBM write of the windows and reads
// that mostly hit the L2 (the series is a few MB), instead of a host-to-device copy of B*T*C floats.
//
// Standardisation reproduces sklearn's StandardScaler.transform on a float32 array bit for bit: the
// scaler keeps float64 statistics, `X -= mean_` and `X /= scale_` each run in float64 and round
// back to float32, so x -> f32(f64(f32(f64(x) - mean)) / scale).  The quotient is formed as
// q0 = x1 * r, q = fma(fma(-scale, q0, x1), r, q0) with r = 1/scale (one correctly rounded division
// per column and workgroup, kept in LDS): with r correctly rounded this is the last step of the
// IEEE division sequence and gives the correctly rounded quotient, at 3 fp64 operations per element
// instead of a full division each.
#include "vts_common.h"
#include "../../include/vts.h"
#include <math.h>

namespace {

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

struct ColStat {
  double mean, scale, rcp;
};

__device__ __forceinline__ float standardise(float x, const ColStat& s) {
  const float x1 = (float)((double)x - s.mean);
  const double xd = (double)x1;
  const double q0 = xd * s.rcp;
  const double q = fma(fma(-s.scale, q0, xd), s.rcp, q0);
  return (float)q;
}

constexpr int WB_THREADS = 256;
constexpr int WB_VEC = 4;          // float4 pieces per thread
constexpr int WB_SPAN = WB_THREADS * WB_VEC * 4;   // floats of one window handled by a workgroup

// grid (windows side by side, ceil(T*C / WB_SPAN)).  Dynamic LDS: C ColStat entries when standardising.
__global__ __launch_bounds__(WB_THREADS) void windows_x_kernel(const float* __restrict__ series, long Nt, int C,
                                                                const long long* __restrict__ starts, long B, int T,
                                                                const double* __restrict__ mean,
                                                                const double* __restrict__ scale,
                                                                float* __restrict__ X) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  ColStat* stat = reinterpret_cast<ColStat*>(wsm);
  const bool std_on = mean != nullptr;
  if (std_on) {
    for (int c = threadIdx.x; c < C; c += WB_THREADS) {
      const double s = scale[c];
      stat[c] = ColStat{mean[c], s, 1.0 / s};
    }
    __syncthreads();
  }
  const long TC = (long)T * C;
  const float qnan = __builtin_nanf("");
  const bool vec_ok = (TC & 3) == 0;                     // every window then starts on a 16-byte boundary
  const bool pair_ok = (C & 1) == 0 && (reinterpret_cast<uintptr_t>(series) & 7) == 0;
  // T*C < 2^31 (checked by the caller): 32-bit offsets; the channel of the thread's first element is
  // found once and then stepped, so the loops carry no division
  const unsigned o0 = blockIdx.y * (unsigned)WB_SPAN + threadIdx.x * 4u;
  const unsigned cfirst = std_on ? o0 % (unsigned)C : 0u;
  const unsigned cstep = (unsigned)(WB_THREADS * 4) % (unsigned)C;
  // a workgroup walks over windows blockIdx.x, + gridDim.x, ...: the column table above is built once for all of them
  for (long b = blockIdx.x; b < B; b += gridDim.x) {
  const long start = starts[b];
  const bool inside = start >= 0 && start + T <= Nt;     // a window that leaves the series reads as NaN
  const float* src = series + start * C;
  float* dst = X + b * TC;
  unsigned c0 = cfirst;
#pragma unroll
  for (int it = 0; it < WB_VEC; ++it) {
    const long o = (long)o0 + (long)it * (WB_THREADS * 4);
    if (o >= TC) break;
    const int n = (int)min(4L, TC - o);
    float v[4];
    if (!inside) {
      v[0] = v[1] = v[2] = v[3] = qnan;
    } else if (n == 4 && pair_ok) {
      // an even channel count puts every window start on an 8-byte boundary: two aligned 8-byte loads (a 16-byte
      // load that straddles its natural boundary is served at a lower rate)
      const float2 x0 = *reinterpret_cast<const float2*>(src + o), x1 = *reinterpret_cast<const float2*>(src + o + 2);
      v[0] = x0.x; v[1] = x0.y; v[2] = x1.x; v[3] = x1.y;
    } else if (n == 4) {
      const f32x4_u x = *reinterpret_cast<const f32x4_u*>(src + o);   // rows of C floats: 4-byte alignment only
      v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else {
      for (int j = 0; j < 4; ++j) v[j] = j < n ? src[o + j] : 0.f;
    }
    if (std_on && inside) {
      unsigned c = c0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = standardise(v[j], stat[c]);
        if (++c == (unsigned)C) c = 0;
      }
    }
    c0 += cstep;
    if (c0 >= (unsigned)C) c0 -= (unsigned)C;
    if (n == 4 && vec_ok) {
      __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(dst + o));
    } else {
      for (int j = 0; j < n; ++j) dst[o + j] = v[j];
    }
  }
  }
}

// One thread per window: the P rows after the window, column `col`.
__global__ void windows_y_kernel(const float* __restrict__ series, long Nt, int C, const long long* __restrict__ starts,
                                 long B, int T, int P, int target, int col, const double* __restrict__ ystat,
                                 float* __restrict__ y) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const long first = starts[b] + T;
  const bool inside = starts[b] >= 0 && first + P <= Nt;
  const float qnan = __builtin_nanf("");
  auto finish = [&](float v) -> float {
    if (!inside) return qnan;
    if (ystat == nullptr) return v;
    const float v1 = (float)((double)v - ystat[0]);
    return (float)((double)v1 / ystat[1]);
  };
  if (target == VTS_TARGET_BLOCK_MAX) {
    float m = qnan;
    if (inside) {
      m = series[first * C + col];
      for (int p = 1; p < P; ++p) {                 // numpy.max: a NaN wins
        const float v = series[(first + p) * C + col];
        if (v > m || v != v) m = (m != m) ? m : v;
      }
    }
    y[b] = finish(m);
  } else {
    for (int p = 0; p < P; ++p) y[b * P + p] = finish(inside ? series[(first + p) * C + col] : 0.f);
  }
}

// ---- statistics of the windows' rows (StandardScaler.fit on X.reshape(-1, C)) -------------------
// cover[r] = number of windows that contain series row r.
__global__ void windows_cover_kernel(const long long* __restrict__ starts, long B, int T, long Nt,
                                     unsigned* __restrict__ cover) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const long s = starts[i / T];
  if (s < 0 || s + T > Nt) return;
  atomicAdd(&cover[s + i % T], 1u);
}

// One workgroup per column: weighted mean, then weighted sum of squared deviations, in fp64 with a
// fixed reduction order (deterministic).  scale = sqrt(var), 1 where the column is constant
// (sklearn's _handle_zeros_in_scale); stats[c] = mean, stats[C + c] = scale.
__global__ __launch_bounds__(256) void windows_stats_kernel(const float* __restrict__ series, long Nt, int C,
                                                            const unsigned* __restrict__ cover, int col_only,
                                                            double* __restrict__ stats, int nstat) {
  __shared__ double red[256];
  __shared__ double bcast;
  const int c = col_only >= 0 ? col_only : blockIdx.x;
  const int slot = col_only >= 0 ? 0 : blockIdx.x;
  auto block_sum = [&](double v) -> double {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) bcast = red[0];
    __syncthreads();
    return bcast;
  };
  double n = 0.0, sx = 0.0;
  for (long r = threadIdx.x; r < Nt; r += 256) {
    const double w = (double)cover[r];
    n += w;
    sx += w * (double)series[r * C + c];
  }
  const double ntot = block_sum(n);
  const double mean = ntot > 0.0 ? block_sum(sx) / ntot : 0.0;
  double ss = 0.0;
  for (long r = threadIdx.x; r < Nt; r += 256) {
    const double d = (double)series[r * C + c] - mean;
    ss += (double)cover[r] * d * d;
  }
  const double var = ntot > 0.0 ? block_sum(ss) / ntot : 0.0;
  if (threadIdx.x == 0) {
    double sc = sqrt(var);
    if (!(sc > 10.0 * 2.220446049250313e-16)) sc = 1.0;
    stats[slot] = mean;
    stats[nstat + slot] = sc;
  }
}

}  // namespace

extern "C" {

int vts_windows_build(const float* series, int64_t Nt, int64_t C, const int64_t* starts, int64_t B, int64_t T,
                      int64_t P, int target, const double* x_mean, const double* x_scale, const double* y_stats,
                      float* X, float* y, vts_stream_t stream) {
  VTS_REQUIRE(series && starts && Nt > 0 && C > 0 && B > 0 && T > 0, "windows_build: bad arguments");
  VTS_REQUIRE((x_mean == nullptr) == (x_scale == nullptr), "windows_build: x_mean and x_scale go together");
  VTS_REQUIRE(X || y, "windows_build: nothing to produce");
  VTS_REQUIRE(!y || P > 0, "windows_build: targets need predict_steps > 0");
  VTS_REQUIRE(target == VTS_TARGET_NEXT || target == VTS_TARGET_BLOCK_MAX, "windows_build: unknown target %d", target);
  VTS_REQUIRE(C <= 2048, "windows_build: at most 2048 channels (got %lld)", (long long)C);
  VTS_REQUIRE(T * C < (1ll << 31) && B < (1ll << 31), "windows_build: window or batch too large");
  VTS_REQUIRE(!X || vts_aligned16(X), "windows_build: X must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (X) {
    const long TC = (long)T * C;
    const unsigned gy = (unsigned)vts_cdiv(TC, WB_SPAN);
    VTS_REQUIRE(gy <= 65535u, "windows_build: window of %ld floats is too long", TC);
    // about eight workgroups per compute unit; each builds its column table once and then walks over windows
    long gx = 2048 / gy;
    gx = gx < 1 ? 1 : (gx > B ? B : gx);
    const dim3 grid((unsigned)gx, gy);
    const size_t lds = x_mean ? (size_t)C * sizeof(ColStat) : 0;
    hipLaunchKernelGGL(windows_x_kernel, grid, dim3(WB_THREADS), lds, st, series, (long)Nt, (int)C,
                       reinterpret_cast<const long long*>(starts), (long)B, (int)T, x_mean, x_scale, X);
    VTS_CHECK_LAUNCH("windows_x");
  }
  if (y) {
    hipLaunchKernelGGL(windows_y_kernel, dim3((unsigned)vts_cdiv(B, 256)), dim3(256), 0, st, series, (long)Nt, (int)C,
                       reinterpret_cast<const long long*>(starts), (long)B, (int)T, (int)P, target, 0, y_stats, y);
    VTS_CHECK_LAUNCH("windows_y");
  }
  return VTS_OK;
}

size_t vts_windows_stats_workspace_bytes(int64_t Nt) { return ((size_t)Nt * sizeof(unsigned) + 255) / 256 * 256; }

int vts_windows_stats(const float* series, int64_t Nt, int64_t C, const int64_t* starts, int64_t B, int64_t T,
                      int64_t P, int target, double* x_stats, double* y_stats, void* workspace, size_t workspace_bytes,
                      vts_stream_t stream) {
  VTS_REQUIRE(series && starts && workspace && Nt > 0 && C > 0 && B > 0 && T > 0, "windows_stats: bad arguments");
  VTS_REQUIRE(x_stats || y_stats, "windows_stats: nothing to produce");
  VTS_REQUIRE(workspace_bytes >= vts_windows_stats_workspace_bytes(Nt), "windows_stats: workspace %zu B too small",
              workspace_bytes);
  VTS_REQUIRE(!y_stats || (target == VTS_TARGET_NEXT && P > 0),
              "windows_stats: target statistics are offered for the next-steps target only");
  VTS_REQUIRE(B * T < (1ll << 40), "windows_stats: batch too large");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned* cover = reinterpret_cast<unsigned*>(workspace);
  const long long* s64 = reinterpret_cast<const long long*>(starts);
  if (x_stats) {
    if (hipMemsetAsync(cover, 0, (size_t)Nt * sizeof(unsigned), st) != hipSuccess) {
      vts_set_error("windows_stats: memset failed");
      return VTS_ERR_HIP;
    }
    hipLaunchKernelGGL(windows_cover_kernel, dim3((unsigned)vts_cdiv(B * T, 256)), dim3(256), 0, st, s64, (long)B, (int)T,
                       (long)Nt, cover);
    VTS_CHECK_LAUNCH("windows_cover");
    hipLaunchKernelGGL(windows_stats_kernel, dim3((unsigned)C), dim3(256), 0, st, series, (long)Nt, (int)C, cover, -1,
                       x_stats, (int)C);
    VTS_CHECK_LAUNCH("windows_stats");
  }
  if (y_stats) {
    // the targets of window b are rows starts[b] + T .. + P - 1 of column 0: the same cover count, P rows
    // long, over starts shifted by T (the kernel tests s + T' <= Nt with T' = P on the shifted start)
    if (hipMemsetAsync(cover, 0, (size_t)Nt * sizeof(unsigned), st) != hipSuccess) {
      vts_set_error("windows_stats: memset failed");
      return VTS_ERR_HIP;
    }
    hipLaunchKernelGGL(windows_cover_kernel, dim3((unsigned)vts_cdiv(B * P, 256)), dim3(256), 0, st, s64, (long)B, (int)P,
                       (long)(Nt - T), cover + T);
    VTS_CHECK_LAUNCH("windows_cover");
    hipLaunchKernelGGL(windows_stats_kernel, dim3(1), dim3(256), 0, st, series, (long)Nt, (int)C, cover, 0, y_stats, 1);
    VTS_CHECK_LAUNCH("windows_stats");
  }
  return VTS_OK;
}

}  // extern "C"
