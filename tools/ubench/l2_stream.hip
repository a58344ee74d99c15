// Development aid: how fast do the 32 CUs of an XCD pull one shared 1 MiB image out of their L2, again and again?
// This is the access pattern of the one-workgroup recurrent kernels (the packed W_hh image, once per time step).
// Variants: register ring or LDS-DMA ring, ring depth, waves per workgroup, every workgroup in the same order or each
// one starting at its own unit of the image.
//   hipcc -O3 --offload-arch=gfx950 -o bin/l2_stream l2_stream.hip && bin/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// UT 1 KiB fragments per unit and wave; a wave owns units_per_wave consecutive units of the image.
template <int DEPTH, int UT, int NW>
__global__ __launch_bounds__(NW * 64) void stream_regs(const u32x4* __restrict__ img, int upw, int reps, int rotate,
                                                       unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32x4* base = img + (size_t)wave * upw * UT * 64 + lane;
  const int rot = rotate ? ((blockIdx.x >> 3) * upw / 32) : 0;
  u32x4 w[DEPTH][UT];
  u32x4 acc = {0u, 0u, 0u, 0u};
  auto issue = [&](u32x4 (&b)[UT], int n) {
    const u32x4* p = base + (size_t)((n + rot) & (upw - 1)) * UT * 64;
#pragma unroll
    for (int i = 0; i < UT; ++i) b[i] = __builtin_nontemporal_load(p + i * 64);
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) issue(w[d], d);
  __builtin_amdgcn_sched_barrier(0);
  const int total = reps * upw;
#pragma unroll 1
  for (int n = 0; n < total; n += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      issue(w[(d + DEPTH - 1) % DEPTH], n + d + DEPTH - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < UT; ++i) acc ^= w[d][i];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, int UT, int NW>
__global__ __launch_bounds__(NW * 64) void stream_lds(const u32x4* __restrict__ img, int upw, int reps, int rotate,
                                                      unsigned* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = reinterpret_cast<const char*>(img) + (size_t)wave * upw * UT * 1024 + lane * 16;
  char* ring = lds + (size_t)wave * DEPTH * UT * 1024;
  const unsigned ring_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring + lane * 16;
  const int rot = rotate ? ((blockIdx.x >> 3) * upw / 32) : 0;
  u32x4 acc = {0u, 0u, 0u, 0u};
  auto issue = [&](int slot, int n) {
    const char* p = base + (size_t)((n + rot) & (upw - 1)) * UT * 1024;
#pragma unroll
    for (int i = 0; i < UT; ++i) glds16(p + i * 1024, ring + (slot * UT + i) * 1024);
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
  const int total = reps * upw;
#pragma unroll 1
  for (int n = 0; n < total; n += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      issue((d + DEPTH - 1) % DEPTH, n + d + DEPTH - 1);
      wait_vm<(DEPTH - 1) * UT>();
      u32x4 v[UT];
#pragma unroll
      for (int i = 0; i < UT; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(ring_a), "n"((d * UT + i) * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < UT; ++i) {
        asm volatile("" : "+v"(v[i]));
        acc ^= v[i];
      }
    }
  }
  wait_vm<0>();
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[blockIdx.x] = 1;
}

template <class K>
static void run(const char* name, K kern, int nw, size_t lds_bytes, const u32x4* img, int upw, unsigned* out, int nwg) {
  if (lds_bytes > 65536)
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const int reps = 96;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rotate = 0; rotate < 2; ++rotate) {
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kern, dim3(nwg), dim3(nw * 64), lds_bytes, 0, img, upw, reps, rotate, out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double bytes = (double)nwg * reps * (1 << 20);
    printf("%-28s wgs %4d rotate %d: %7.3f ms  %7.1f GB/s per workgroup  %6.2f TB/s\n", name, nwg, rotate, best,
           bytes / nwg / best * 1e-6, bytes / best * 1e-9);
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  u32x4* img;
  unsigned* out;
  CK(hipMalloc(&img, 1 << 20));
  CK(hipMalloc(&out, 4096 * 4));
  std::vector<unsigned> h((1 << 20) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
  CK(hipMemcpy(img, h.data(), 1 << 20, hipMemcpyHostToDevice));
  // 4 waves, 8 KiB units: 32 units per wave
  run("regs depth 3 ut 8 nw 4", stream_regs<4, 8, 4>, 4, 0, img, 32, out, nwg);       // 4 buffers, 3 in flight
  run("regs depth 7 ut 8 nw 4", stream_regs<8, 8, 4>, 4, 0, img, 32, out, nwg);
  run("regs depth 3 ut 4 nw 8", stream_regs<4, 4, 8>, 8, 0, img, 32, out, nwg);
  run("regs depth 7 ut 4 nw 8", stream_regs<8, 4, 8>, 8, 0, img, 32, out, nwg);
  run("regs depth 11 ut 4 nw 8", stream_regs<12, 4, 8>, 8, 0, img, 32, out, nwg);
  run("lds  depth 3 ut 8 nw 4", stream_lds<4, 8, 4>, 4, 4 * 4 * 8 * 1024, img, 32, out, nwg);
  run("lds  depth 4 ut 8 nw 4", stream_lds<5, 8, 4>, 4, 4 * 5 * 8 * 1024, img, 32, out, nwg);   // 32 % 5 != 0: 160 steps
  run("lds  depth 7 ut 4 nw 4", stream_lds<8, 4, 4>, 4, 4 * 8 * 4 * 1024, img, 64, out, nwg);
  run("lds  depth 3 ut 4 nw 8", stream_lds<4, 4, 8>, 8, 8 * 4 * 4 * 1024, img, 32, out, nwg);
  run("lds  depth 4 ut 4 nw 8", stream_lds<5, 4, 8>, 8, 8 * 5 * 4 * 1024, img, 32, out, nwg);
  return 0;
}
