// Micro-benchmark: what do the LDS fragment reads of the plane GEMM's main loop cost the matrix cores?  A wave owns a
// 4 x 8 grid of 16 x 16 accumulator tiles (the plane GEMM's wave tile) and runs 96 MFMAs per "k-step" - three per
// fragment pair, as bf16x3 does - on random bf16 operands.  Before each k-step it re-reads FR of its 24 operand
// fragments from LDS with ds_read_b128 (FR = 24: what the GEMM does, 256 bytes per MFMA; 16: what a 128 x 128 wave
// tile would need per 96 MFMAs; 0: operands stay in registers).  Two waves per SIMD, every CU busy.
// Development tool:  hipcc -O3 --offload-arch=gfx950 mfma_lds_power.hip -o bin/mfma_lds_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int FR>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, float* out, int iters, long long* clk) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 6 * 24 * 64; i += 512) lds[i] = src[i & 8191];
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  u32x4 f[24];     // [0, 8): A hi / lo of four row tiles; [8, 24): B hi / lo of eight column tiles
  for (int i = 0; i < 24; ++i) f[i] = lds[((wave % 6) * 24 + i) * 64 + lane];
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < FR; ++i) f[i] = lds[(((wave + it) % 6) * 24 + i) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f[2 * i]), __builtin_bit_cast(bf16x8, f[8 + 2 * j + 1]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f[2 * i + 1]), __builtin_bit_cast(bf16x8, f[8 + 2 * j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f[2 * i]), __builtin_bit_cast(bf16x8, f[8 + 2 * j]), acc[i][j], 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = r;
  if (tid == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int FR>
void run(const u32x4* src, float* out, long long* clk, const char* what) {
  const int iters = 2000, lds_bytes = 6 * 24 * 64 * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<FR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(s);
    hipLaunchKernelGGL(k<FR>, dim3(256), dim3(512), lds_bytes, 0, src, out, iters, clk);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    if (rep > 0 && ms < best) best = ms;
  }
  long long hc[2];
  hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
  const double flops = 256.0 * 8 * iters * 96.0 * 16384.0;
  printf("%s operands, %2d of 24 fragments re-read per 96 MFMAs (%3d B / MFMA): %.3f ms  %.0f TFLOP/s  clock %.2f GHz\n", what, FR,
         FR * 1024 / 96, best, flops / (best * 1e-3) / 1e12, 0.1 * (double)hc[0] / (double)hc[1]);
}

int main() {
  u32x4* src; float* out; long long* clk;
  hipMalloc(&src, 8192 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
  std::vector<unsigned> h(8192 * 4);
  for (int data = 0; data < 2; ++data) {
    for (auto& v : h) {
      if (!data) { v = 0; continue; }
      auto one = [] { const unsigned m = rand() & 0x7f, e = 126 + (rand() & 1), s = rand() & 1; return (s << 15) | (e << 7) | m; };
      v = one() | (one() << 16);
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* what = data ? "random" : "zero  ";
    run<0>(src, out, clk, what);
    run<8>(src, out, clk, what);
    run<16>(src, out, clk, what);
    run<24>(src, out, clk, what);
  }
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
