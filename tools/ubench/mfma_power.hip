// Micro-benchmark: sustained rate of the two bf16 MFMA shapes on operands held in registers - zeros against random
// values, one or two waves per SIMD, every CU busy for about a millisecond.  The question it answers: how much of the
// gap between the plane GEMM kernels and the bf16 peak is the matrix cores' clock under load, and does the
// 32 x 32 x 16 shape (half the operand-register reads per FLOP) sustain more than the 16 x 16 x 32 one.
// Development tool (not part of the library):  hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 16x16x32, 4 x 8 accumulator tiles; 1: 32x32x16, 2 x 4 accumulator tiles (128 registers either way)
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, float* out, int iters, long long* clk) {
  const int tid = threadIdx.x;
  const long long c0 = clock64(), w0 = wall_clock64();
  u32x4 a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = src[(i * 512 + tid) & 8191];
  for (int i = 0; i < 8; ++i) b[i] = src[((4 + i) * 512 + tid) & 8191];
  float r = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)      // two k-halves: the same 32 k per iteration as SHAPE 0
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[2 * kk + i]), __builtin_bit_cast(bf16x8, b[4 * kk + j]), acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][15];
  }
  out[blockIdx.x * 512 + tid] = r;
  // shader-clock ticks (s_memtime) against the constant 100 MHz counter over the loop: the clock the CU really ran at
  if (tid == 0 && blockIdx.x == 0) {
    clk[0] = clock64() - c0;
    clk[1] = wall_clock64() - w0;
  }
}

int main() {
  u32x4* src; float* out; long long* clk;
  hipMalloc(&src, 8192 * 16); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&clk, 16);
  std::vector<unsigned> h(8192 * 4);
  for (int data = 0; data < 2; ++data) {
    for (auto& v : h) {
      if (!data) { v = 0; continue; }
      // two random bf16 values of magnitude ~1 with full mantissas
      auto one = [] { const unsigned m = rand() & 0x7f, e = 126 + (rand() & 1), s = rand() & 1; return (s << 15) | (e << 7) | m; };
      v = one() | (one() << 16);
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape)
      for (int threads = 256; threads <= 512; threads += 256) {
        const int iters = 6000;
        hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          hipEventRecord(s);
          if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, src, out, iters, clk);
          else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, src, out, iters, clk);
          hipEventRecord(e); hipEventSynchronize(e);
          float ms; hipEventElapsedTime(&ms, s, e);
          if (rep > 0 && ms < best) best = ms;
        }
        const double flops = 256.0 * (threads / 64) * iters * 32.0 * 16384.0;     // 32 MFMAs of 16x16x32 (or 16 of 32x32x16) per iteration
        long long hc[2];
        hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        printf("%s operands, %s, %d waves/SIMD: %.3f ms  %.0f TFLOP/s   s_memtime ticks / 100 MHz ticks of the last run: %.1f\n",
               data ? "random" : "zero  ", shape ? "32x32x16" : "16x16x32", threads / 256, best, flops / (best * 1e-3) / 1e12,
               (double)hc[0] / (double)hc[1]);
      }
  }
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return 0;
}
