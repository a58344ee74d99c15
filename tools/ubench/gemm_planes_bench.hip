// Stand-alone check + timing of the plane GEMM main loop (csrc/gemm_planes.h) on a weight-gradient shaped problem:
// C[M][N] = sum_k A[m][k] B[n][k], M 1024, N 512, K = 32 * ksteps, split over `ksplit` slices into fp32 slabs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vae-time-series_amd/csrc tools/ubench/gemm_planes_bench.hip -o gemm_planes_bench
//   ./gemm_planes_bench            (validates a small K against a host fp64 sum, then times K = 393216)
#include "gemm_planes.h"
#include <stdlib.h>
#include <vector>
#include <math.h>

void vts_set_error(const char*, ...) {}

namespace {

// What both policies do with the finished accumulators: this workgroup's 256 x 256 tile of its slice's slab.
struct SlabOut {
  float* C; int N, tm, tn;
  __device__ __forceinline__ void store(f32x4 (&acc)[4][8]) const {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = tm * 256 + (wm * 4 + mt) * 16 + 4 * g + reg;
          const int col = tn * 256 + (wn * 8 + nt) * 16 + r;
          C[(long)row * N + col] = acc[mt][nt][reg];
        }
  }
};
// direct: operand [rows][K] as tiles: tile(rt, s, p) = base + ((s * 2 + p) * ntiles + rt) * 1024
struct Pol {
  const char* A; const char* B;
  int mt_all, nt_all;   // row tiles of the whole operands
  int m0t, n0t;         // first row tile of this workgroup
  int s;                // k-step whose loads are being issued (advance() steps to the first one)
  SlabOut out;
  int wrap;             // timing only: k-step s reads the tiles of k-step s % wrap (operands stay in the L2)
  __device__ __forceinline__ void advance() { ++s; }
  __device__ __forceinline__ const char* a_tile(int li) const {
    return A + ((long)((s % wrap) * 2 + (li & 1)) * mt_all + m0t + (li >> 1)) * gp::TILE_B;
  }
  __device__ __forceinline__ const char* b_tile(int li) const {
    return B + ((long)((s % wrap) * 2 + (li & 1)) * nt_all + n0t + (li >> 1)) * gp::TILE_B;
  }
  __device__ __forceinline__ void flush(f32x4 (&acc)[4][8]) { out.store(acc); }
};
// transposed: operand [tokens][features] as tiles of 16 tokens x 32 features: tile(tt, cb, p) = base + ((tt * ncb + cb) * 2 + p) * 1024
struct PolT {
  const char* A; const char* B;
  int acb_all, bcb_all;  // 32-column blocks of the whole operands
  int acb0, bcb0;        // first column block of this workgroup
  int s;
  SlabOut out;
  int wrap;
  __device__ __forceinline__ void advance() { ++s; }
  __device__ __forceinline__ const char* a_tile(int li) const {
    const int tt = li >> 4, cb = (li >> 1) & 7, p = li & 1;
    return A + (((long)(2 * (s % wrap) + tt) * acb_all + acb0 + cb) * 2 + p) * gp::TILE_B;
  }
  __device__ __forceinline__ const char* b_tile(int li) const {
    const int tt = li >> 4, cb = (li >> 1) & 7, p = li & 1;
    return B + (((long)(2 * (s % wrap) + tt) * bcb_all + bcb0 + cb) * 2 + p) * gp::TILE_B;
  }
  __device__ __forceinline__ void flush(f32x4 (&acc)[4][8]) { out.store(acc); }
};

__device__ long long g_clk[2];     // shader-clock ticks (s_memtime) and 100 MHz ticks of workgroup 0's main loop

template <int DBG, bool TRANS>
__global__ __launch_bounds__(512) void wgrad_like_kernel(const char* A, const char* B, float* slabs, int M, int N,
                                                            int ksteps, int ksplit, int wrap) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const long long c0 = clock64(), w0 = wall_clock64();
  const int tiles_m = M / gp::TM, tiles_n = N / gp::TN, per = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int z = (j / per) * 8 + xcd, tile = j % per;
  if (z >= ksplit) return;
  const int chunk = ksteps / ksplit;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const SlabOut out{slabs + (long)z * M * N, N, tm, tn};
  if (TRANS) {
    PolT pol{A, B, M / 32, N / 32, tm * 8, tn * 8, z * chunk - 1, out, wrap};
    gp::mainloop<PolT, true, 8, DBG>(pol, chunk, chunk, lds);
  } else {
    Pol pol{A, B, M / 16, N / 16, tm * 16, tn * 16, z * chunk - 1, out, wrap};
    gp::mainloop<Pol, false, 8, DBG>(pol, chunk, chunk, lds);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk[0] = clock64() - c0;
    g_clk[1] = wall_clock64() - w0;
  }
}

__host__ __device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// value of logical element (row, k) of operand `which`: roughly N(0,1)-sized, full-range mantissas
__host__ __device__ inline float elem(int which, int row, long k) {
  const unsigned h = hash32((unsigned)(k * 2654435761u) ^ hash32((unsigned)row * 40503u + which * 7919u));
  const float u = (float)(h & 0xffffff) / 16777216.0f - 0.5f;
  const float v = (float)((h >> 24) & 0xff) / 256.0f - 0.5f;
  return 2.0f * u + 0.37f * v;
}
__host__ __device__ inline unsigned short bf16_rne(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_to_f(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}

// one thread per bf16 element of the plane buffer; trans = 0: tiles (s, p, rt) of [rows][K]; 1: tiles (tt, cb, p) of [tokens][features]
__global__ void fill_kernel(unsigned short* buf, int which, int rows, int ksteps, int trans) {
  const long total = (long)ksteps * 2 * (rows / 16) * 512;   // bf16 elements
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = i & 511;           // element inside the tile: tile row r = e >> 5, tile k = e & 31
    const int r = e >> 5, k = e & 31;
    long rest = i >> 9;
    int row, pl; long kk;
    if (!trans) {
      const int ntiles = rows / 16;
      const int rt = rest % ntiles; rest /= ntiles;
      pl = rest & 1; const long st = rest >> 1;
      row = rt * 16 + r; kk = st * 32 + k;
    } else {
      const int ncb = rows / 32;
      pl = rest & 1; rest >>= 1;
      const int cb = rest % ncb; const long tt = rest / ncb;
      row = cb * 32 + k; kk = tt * 16 + r;
    }
    const float x = elem(which, row, kk);
    const unsigned short hi = bf16_rne(x);
    buf[(i >> 9) * 512 + gp::elem_offset(r, k) / 2] = pl ? bf16_rne(x - bf16_to_f(hi)) : hi;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

}  // namespace

int main(int argc, char** argv) {
  const int M = 1024, N = 512;
  const int big_steps = argc > 1 ? atoi(argv[1]) : 12288, ksplit_big = argc > 2 ? atoi(argv[2]) : 32;
  // argv[4]: timing only - every k-step reads the operand tiles of k-step s % wrap, so that they come out of the L2
  // (16 k-steps = 3 MiB of operands) instead of HBM: the same instruction stream without the memory system's share
  const int wrap = argc > 4 && atoi(argv[4]) > 0 ? atoi(argv[4]) : 1 << 30;
#define SETLDS(D, T) CK(hipFuncSetAttribute((const void*)wgrad_like_kernel<D, T>, hipFuncAttributeMaxDynamicSharedMemorySize, gp::LDS_B))
  SETLDS(0, false); SETLDS(1, false); SETLDS(2, false); SETLDS(0, true); SETLDS(1, true); SETLDS(2, true);
  char *A, *B; float* slabs;
  const size_t abytes = (size_t)big_steps * 2 * (M / 16) * 1024, bbytes = (size_t)big_steps * 2 * (N / 16) * 1024;
  CK(hipMalloc(&A, abytes)); CK(hipMalloc(&B, bbytes));
  CK(hipMalloc(&slabs, (size_t)ksplit_big * M * N * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int trans = 0; trans < 2; ++trans) {
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned short*)A, 0, M, big_steps, trans);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (unsigned short*)B, 1, N, big_steps, trans);
    CK(hipDeviceSynchronize());
    // ---- validation: 80 k-steps in 8 slices of 10, then 40 in 8 slices of 5 (even and odd step counts)
    for (int vs : {80, 40}) {
      const int ks = 8;
      if (trans) hipLaunchKernelGGL((wgrad_like_kernel<0, true>), dim3(8 * 8), dim3(512), gp::LDS_B, 0, A, B, slabs, M, N, vs, ks, 1 << 30);
      else       hipLaunchKernelGGL((wgrad_like_kernel<0, false>), dim3(8 * 8), dim3(512), gp::LDS_B, 0, A, B, slabs, M, N, vs, ks, 1 << 30);
      CK(hipDeviceSynchronize());
      std::vector<float> h((size_t)ks * M * N);
      CK(hipMemcpy(h.data(), slabs, h.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0, scale = 0;
      for (int t = 0; t < 400; ++t) {
        const int m = (t * 131 + 7) % M, n = (t * 71 + 3) % N;
        double ref = 0, refabs = 0;
        for (long k = 0; k < (long)vs * 32; ++k) {
          const float a = elem(0, m, k), b = elem(1, n, k);
          const float ah = bf16_to_f(bf16_rne(a)), bh = bf16_to_f(bf16_rne(b));
          const float al = bf16_to_f(bf16_rne(a - ah)), bl = bf16_to_f(bf16_rne(b - bh));
          ref += (double)ah * bh + (double)ah * bl + (double)al * bh;
          refabs += fabs((double)a * b);
        }
        double got = 0;
        for (int z = 0; z < ks; ++z) got += h[(size_t)z * M * N + (size_t)m * N + n];
        worst = fmax(worst, fabs(got - ref));
        scale = fmax(scale, refabs);
      }
      printf("trans %d validate ksteps %d: worst |err| %.3e against sum|a b| %.3e -> rel %.2e %s\n", trans, vs, worst, scale,
             worst / scale, worst / scale < 2e-6 ? "OK" : "FAIL");
      if (!(worst / scale < 2e-6)) return 1;
    }
    if (argc > 3 && atoi(argv[3]) == 1) {   // zero operands: the data-dependent share of the matrix cores' power goes away
      CK(hipMemset(A, 0, abytes)); CK(hipMemset(B, 0, bbytes)); printf("operands zeroed\n");
    }
    // ---- timing
    const int grid = 8 * (ksplit_big / 8) * 8;
    for (int dbg = 0; dbg < 3; ++dbg)
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, 0));
      const int iters = 5;
      for (int i = 0; i < iters; ++i) {
#define GO(D, T) hipLaunchKernelGGL((wgrad_like_kernel<D, T>), dim3(grid), dim3(512), gp::LDS_B, 0, A, B, slabs, M, N, big_steps, ksplit_big, wrap)
        if (trans) { if (dbg == 0) GO(0, true); if (dbg == 1) GO(1, true); if (dbg == 2) GO(2, true); }
        else       { if (dbg == 0) GO(0, false); if (dbg == 1) GO(1, false); if (dbg == 2) GO(2, false); }
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
      const double flop = 2.0 * M * N * 32.0 * big_steps;
      long long hc[2];
      CK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
      if (wrap < (1 << 30) && dbg == 0 && rep == 0) printf("operand k-steps wrap at %d\n", wrap);
      printf("trans %d dbg %d M %d N %d K %ld ksplit %d: %.3f ms, %.1f TFLOP/s algorithmic, %.1f executed = %.3f of 2500; shader clock %.2f GHz\n",
             trans, dbg, M, N, 32L * big_steps, ksplit_big, ms, flop / ms * 1e-9, 3 * flop / ms * 1e-9, 3 * flop / ms * 1e-9 / 2500.0,
             0.1 * (double)hc[0] / (double)hc[1]);
    }
  }
  return 0;
}
