// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 from one wave per SIMD, with and without
// operand streaming from L2.  Development tool (not part of the library).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ W, float* out, int steps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 w[16];
  for (int i = 0; i < 16; ++i) w[i] = W[(wave * 16 + i) * 64 + lane];
  f32x4 a = W[lane];
  const f32x4* Wv = W + (long)wave * 16 * 16 * 64 + lane;
  if (MODE < 2) for (int t = 0; t < steps; ++t) {
#pragma unroll 1
    for (int kq = 0; kq < 16; ++kq) {
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = Wv[(kq * 16 + i) * 64];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], w[i][s], acc[i], 0, 0, 0);
    }
  }
  if (MODE == 2 || MODE == 3) {
    f32x4 w0[16], w1[16], w2[16], w3[16];
    auto load_w = [&](f32x4 (&ww)[16], int kq) {
#pragma unroll
      for (int i = 0; i < 16; ++i) ww[i] = Wv[(kq * 16 + i) * 64];
    };
    auto mma = [&](const f32x4 (&ww)[16]) {
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], ww[i][s], acc[i], 0, 0, 0);
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
    };
    load_w(w0, 0); load_w(w1, 1); load_w(w2, 2);
    for (int t = 0; t < steps; ++t) {
#pragma unroll 1
      for (int kq = 0; kq < 16; kq += 4) {
        load_w(w3, kq + 3); mma(w0);
        load_w(w0, (kq + 4) & 15); mma(w1);
        load_w(w1, (kq + 5) & 15); mma(w2);
        load_w(w2, (kq + 6) & 15); mma(w3);
      }
    }
  }
  f32x4 r = acc[0];
  for (int i = 1; i < 16; ++i) r += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}

int main() {
  f32x4* W; float* out;
  hipMalloc(&W, 1 << 20); hipMalloc(&out, 256 * 256 * 4);
  hipMemset(W, 0, 1 << 20);
  const int steps = 96;
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(s);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, W, out, steps);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, W, out, steps);
      else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, W, out, steps);
      else hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, W, out, steps);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      const double mfma = 1024.0 * steps;  // per wave
      printf("mode %d: %.3f ms, %.1f cycles/MFMA at 2.4 GHz, %.1f TFLOP/s\n", mode, ms, ms * 1e-3 * 2.4e9 / mfma,
             mfma * 1024 * 2048 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
